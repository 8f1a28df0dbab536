// Shared device-side definitions for the BoxDreamer gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/boxdreamer_hip.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
// 16-byte register quantum for global/LDS moves (a native vector: HIP's struct uint4 defeats SROA
// and lands staging registers in scratch)
typedef __attribute__((__vector_size__(16))) unsigned int u128;

// 16-bit MFMA operand types.  Both run v_mfma_f32_32x32x16_* at the same rate; the C/D
// fragment layout is dtype-independent (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
template <class T> struct Op16;
template <> struct Op16<__bf16> {
    typedef bf16x8 vec8;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Op16<_Float16> {
    typedef f16x8 vec8;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// OCP e4m3 element (gfx950 fp8; NOT the MI300 fnuz encoding).  Only ever moved around as raw bytes.
struct fp8e4 { unsigned char v; };
typedef __attribute__((__vector_size__(8 * sizeof(int)))) int i32x8;

// fp8 operands go through the block-scaled MFMA with unit (E8M0 = 127) scales: 32x32x64 per instruction at twice the
// bf16 rate.  A lane supplies 32 consecutive k of its row (8 VGPRs) for its lane half; since A and W fragments are
// built by the same loader the k labelling inside the instruction is immaterial (tools/mfma_layout_probe.hip).
template <> struct Op16<fp8e4> {
    typedef i32x8 vec8;     // "fragment" type of this operand class
    static __device__ __forceinline__ f32x16 mfma(i32x8 a, i32x8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
};
// F16C8 (defined below): the hi plane is an f16 operand; the e4m3 correction pass is issued explicitly by the GEMM
struct f16c8;
template <> struct Op16<f16c8> {
    typedef f16x8 vec8;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
// element size, K per MFMA, 16-byte chunks per fragment per lane
template <class T> struct OpGeom { static constexpr int ESZ = 2, KSTEP = 16, CPF = 1; };
template <> struct OpGeom<fp8e4> { static constexpr int ESZ = 1, KSTEP = 64, CPF = 2; };
template <> struct OpGeom<f16c8> { static constexpr int ESZ = 2, KSTEP = 16, CPF = 1; };

template <class T> __device__ __forceinline__ T from_f32(float x) { return (T)x; }
template <class T> __device__ __forceinline__ float to_f32(T x) { return (float)x; }

// store N (4 or 8) consecutive elements converted from fp32
template <class T, int N> __device__ __forceinline__ void store_cvt(T* dst, const float (&v)[N]) {
    typedef __attribute__((__vector_size__(N * sizeof(T)))) T vecN;
    vecN o;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        // The fp32 value is rounded to T as a SEPARATE step, everywhere: left alone, hipcc fuses a trailing multiply with the
        // conversion (v_fma_mixlo_f16: one rounding instead of two) in some inlining contexts and not in others, and the same
        // row then differs in near-tie cases between two kernels that compute the same fp32 result (found by the tile-shape
        // invariance property: 63 of 3.5M f16 GELU outputs).  The empty asm makes the value opaque; it emits nothing.
        float x = v[j];
        asm("" : "+v"(x));
        o[j] = (T)x;
    }
    // BD_STORE_NT (set by gemm.hip only): non-temporal stores for the GEMM's 16/8-bit outputs.  qkv and the MLP hidden
    // are 226-302 MB per launch, written once and read by a LATER kernel; kept out of the 4 MB L2s' write-back path they
    // no longer evict the operand tiles of the running GEMM: whole step +3.2 %.  The LayerNorm / layout / attention
    // outputs measured neutral to negative with the same hint (-6 % for the attention output) and keep plain stores.
#if defined(BD_STORE_NT) && BD_STORE_NT
    __builtin_nontemporal_store(o, (vecN*)dst);
#else
    *(vecN*)dst = o;
#endif
}
template <> __device__ __forceinline__ void store_cvt<fp8e4, 8>(fp8e4* dst, const float (&v)[8]) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    *(uint2*)dst = make_uint2((unsigned)lo, (unsigned)hi);
}
template <> __device__ __forceinline__ void store_cvt<fp8e4, 4>(fp8e4* dst, const float (&v)[4]) {
    int lo = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    *(int*)dst = lo;
}

// ---- BD_PREC_F16C8 operand class: "f16 + e4m3 corrections" (the round-2 strict mode).
//   x  ~=  hi + lo,   hi = f16(x),   lo = x - hi  (|lo| <= 2^-12 |x|)
//   A . W  ~=  hi_A . hi_W  (one f16 MFMA pass)  +  lo_A . q_W  +  q_A . lo_W   (ONE e4m3 pass over a doubled K on the
//   block-scaled MFMA at twice the f16 rate: positions [lo_A | q_A] against [q_W | lo_W]),  q = e4m3 image of hi.
// Two MFMA pass-equivalents instead of split-bf16's three AND 3 bytes per element instead of 4 through LDS-DMA; heatmap-logit
// error 1.7e-4 at full depth (oracle/numerics_sim.py "f16c8fix"; split-bf16: 0.7e-4; bar: 1e-3).  Storage:
//   plane 0: f16 hi [rows][K]  (2 bytes / element);
//   plane 1: lo8 = e4m3(lo * 2^(E + D)) [rows][K] (1 byte / element), starting `plane` 2-byte units after plane 0, the 32 k of
//            every block stored in the order the MFMA lanes want them: byte 16 h + 8 a + j holds k = 16 a + 8 h + j
//            (a = f16 k-step of the block, h = lane half, j < 8), so that lane (row, h) reads ITS sixteen lo8 bytes with one
//            ds_read_b128 and in the same order as the sixteen q8 bytes it derives from its own two f16 fragments
//            (v_cvt_scalef32_pk_fp8_f16: q8 = e4m3(hi * 2^E), nothing stored).
// Scales are fixed powers of two: E = 0 for activations, one exponent per weight tensor (bd_linear.w_qexp: max|w| 2^E <= 448),
// D = 11; both cross terms carry 2^(E_w + D), undone by the MFMA's E8M0 block scales.
// RANGE (round 3; round 2 clamped the whole activation to +-448, which is catastrophic on trained-like statistics -- LayerNorm
// gains and massive-activation channels push single A-operand elements past 448, oracle/numerics_sim.py --outliers): the hi
// plane keeps the full f16 range (producers clamp to +-65504 only), and the two e4m3 images SATURATE instead:
//   q8 of A (derived in the GEMM's registers) = e4m3(clamp(hi, +-448)): an element beyond 448 loses accuracy only in its
//       q_A . lo_W term, i.e. degrades to what a single f16 pass gives for that one element;
//   lo8 = e4m3(clamp((x - hi) 2^D, +-448)): exact for |x| < 512, partially saturated beyond (f16's ulp there exceeds 448 / 2^D).
// Saturation costs nothing: with MODE.FP16_OVFL set, gfx950's fp8 conversions (v_cvt_scalef32_pk_fp8_f16, v_cvt_pk_fp8_f32) clamp
// to +-448 and f32 -> f16 conversions to +-65504 instead of producing NaN / inf (measured: tools/fp8_sat_probe.hip; an explicit
// v_pk_max / v_pk_min pair per f16 pair in the GEMM's K loop cost 3.2 % of the strict step).  Every kernel that produces or consumes
// the class calls bd_saturating_conversions() first; the mode is per-wave state and dies with the wave.
__device__ __forceinline__ void bd_saturating_conversions() {
    __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);      // hwreg(HW_REG_MODE, 23, 1) = MODE.FP16_OVFL
}
struct f16c8 { unsigned short v; };              // storage element of either plane (never used arithmetically)
#define BD_F16C8_D 11

template <class T> __device__ __forceinline__ typename Op16<T>::vec8 as_vec8(u128 u);
template <> __device__ __forceinline__ bf16x8 as_vec8<__bf16>(u128 u) { return __builtin_bit_cast(bf16x8, u); }
template <> __device__ __forceinline__ f16x8 as_vec8<_Float16>(u128 u) { return __builtin_bit_cast(f16x8, u); }
template <> __device__ __forceinline__ f16x8 as_vec8<f16c8>(u128 u) { return __builtin_bit_cast(f16x8, u); }

// load one element of a runtime-typed input tensor (bf16 / f16 / f32) as fp32
__device__ __forceinline__ float load_any(const void* p, size_t i, int dtype) {
    if (dtype == BD_DTYPE_F32) return ((const float*)p)[i];
    if (dtype == BD_DTYPE_BF16) return (float)((const __bf16*)p)[i];
    return (float)((const _Float16*)p)[i];
}

// Wave-wide sum, result in every lane.  On the VALU: four DPP butterfly steps leave every lane of a 16-lane row with its row's sum
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), four v_readlane + three adds combine the rows.  (__shfl_xor is
// ds_bpermute_b32 + lgkmcnt(0) per step: six dependent LDS round trips per reduction, twelve per LayerNorm row.)
__device__ __forceinline__ float wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// Exact-erf GELU  0.5 x (1 + erf(x / sqrt2))  evaluated in erfc form so the negative tail has no cancellation:
//   1 + erf(z) = erfc(-z);  erfc(|z|) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),  t = 1 / (1 + p |z|)
// (Abramowitz & Stegun 7.1.26, |abs err| <= 1.5e-7 -- fp32-rounding class).  ~12 VALU ops + v_exp + v_rcp per
// element instead of ocml erff's branchy ~40: the fc1 epilogue was costing 40 % of that GEMM.
// (All GELU forms: every fused multiply-add is written out and contraction is off inside, so that the arithmetic does not
// depend on the inlining context -- two kernels computing the same row must agree bit for bit.)
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(t, poly, 1.421413741f);
    poly = fmaf(t, poly, -0.284496736f);
    poly = fmaf(t, poly, 0.254829592f);
    const float erfc_abs = poly * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
    return 0.5f * x * (x > 0.f ? 2.0f - erfc_abs : erfc_abs);
}

// The same arithmetic for TWO elements per instruction where the VALU has a packed fp32 form (v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32: IEEE results, bit-identical to the scalar instructions): 18 regular + 4 quarter-rate instructions per pair instead of
// 2 x (16 + 2).  The strict classes' fc1 epilogue spends ~9k of its ~12k VALU cycles per wave and tile in gelu_erf (round 5:
// profiles/r5_gelu_packed.md); every operation below is the scalar form's, in the scalar form's order, so gelu_erf2(x).x == gelu_erf(x.x).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
#pragma clang fp contract(off)
    auto c2 = [](float c) { return (f32x2){c, c}; };
    f32x2 z;
    z.x = fabsf(x.x) * 0.70710678118654752f;
    z.y = fabsf(x.y) * 0.70710678118654752f;
    const f32x2 d = __builtin_elementwise_fma(c2(0.3275911f), z, c2(1.0f));
    f32x2 t;
    t.x = __builtin_amdgcn_rcpf(d.x);
    t.y = __builtin_amdgcn_rcpf(d.y);
    f32x2 poly = __builtin_elementwise_fma(t, c2(1.061405429f), c2(-1.453152027f));
    poly = __builtin_elementwise_fma(t, poly, c2(1.421413741f));
    poly = __builtin_elementwise_fma(t, poly, c2(-0.284496736f));
    poly = __builtin_elementwise_fma(t, poly, c2(0.254829592f));
    const f32x2 a = (-z * z) * c2(1.4426950408889634f);
    f32x2 ex;
    ex.x = __builtin_amdgcn_exp2f(a.x);
    ex.y = __builtin_amdgcn_exp2f(a.y);
    const f32x2 erfc_abs = (poly * t) * ex;
    const f32x2 two_minus = c2(2.0f) - erfc_abs;
    f32x2 sel;
    sel.x = x.x > 0.f ? two_minus.x : erfc_abs.x;
    sel.y = x.y > 0.f ? two_minus.y : erfc_abs.y;
    return (c2(0.5f) * x) * sel;
}

// GELU for results that are about to be rounded to 16 (or 8) bits:  x * Phi(x)  with  Phi(x) ~ 1 / (1 + 2^(-x p(x^2))),
// p minimax-fitted against the exact erf form on [-9, 9] (tools/fit_gelu.py): |abs err| <= 5.4e-5, below half an ulp of
// bf16 for |gelu| > 0.03 and of f16 for |gelu| > 0.2.  7 VALU + v_exp + v_rcp per element against 15 + 2 for the erfc
// form: the fc1 epilogue is VALU-bound (64 GELUs per lane per 256x128 tile against 192 MFMAs), so this is what lets it
// hide under the next tile's MFMAs.  Only the single-pass modes use it; the strict split-bf16 mode and every fp32
// output keep gelu_erf.  The argument is clamped to +-8 where the fitted polynomial is still monotone (Phi saturates).
__device__ __forceinline__ float gelu_fast(float x) {
#pragma clang fp contract(off)
    const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
    const float x2 = xc * xc;
    const float pz = fmaf(x2, fmaf(x2, -1.10189899e-03f, 1.07380689e-01f), 2.30034092f);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-(xc * pz)));
}
// Transcendental-free variant for bf16 / e4m3 results, two elements per instruction (v_pk_fma_f32):
//   gelu(x) ~ x * (0.5 + xc * P(xc^2)),  xc = clamp(x, -4, 4),  P of degree 7 in xc^2, minimax-fitted (tools/fit_gelu.py);
// |abs err| <= 1.6e-4 in fp32 arithmetic = half a bf16 ulp at |gelu| = 0.08.  v_exp_f32 / v_rcp_f32 issue at quarter
// rate, so gelu_fast spends most of its time in them; this form is ~2x cheaper.  fp16 results keep gelu_fast (an f16 ulp
// is 8x finer), fp32 / split-bf16 results keep gelu_erf.
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
#pragma clang fp contract(off)
    f32x2 xc;
    xc.x = __builtin_amdgcn_fmed3f(x.x, -4.0f, 4.0f);
    xc.y = __builtin_amdgcn_fmed3f(x.y, -4.0f, 4.0f);
    const f32x2 x2 = xc * xc;
    auto c2 = [](float c) { return (f32x2){c, c}; };
    f32x2 pz = __builtin_elementwise_fma(x2, c2(-1.411566826e-09f), c2(1.110951978e-07f));
    pz = __builtin_elementwise_fma(pz, x2, c2(-3.829025890e-06f));
    pz = __builtin_elementwise_fma(pz, x2, c2(7.702150218e-05f));
    pz = __builtin_elementwise_fma(pz, x2, c2(-1.020914998e-03f));
    pz = __builtin_elementwise_fma(pz, x2, c2(9.552915274e-03f));
    pz = __builtin_elementwise_fma(pz, x2, c2(-6.594778014e-02f));
    pz = __builtin_elementwise_fma(pz, x2, c2(3.986759341e-01f));
    return x * __builtin_elementwise_fma(xc, pz, c2(0.5f));
}
// GELU of N (even) values headed for a 16/8-bit store: KIND 0 exact erf, 1 exp-based fit, 2 packed polynomial
template <int KIND, int N> __device__ __forceinline__ void gelu_n(float (&v)[N]) {
    if constexpr (KIND == 2) {
#pragma unroll
        for (int e = 0; e < N; e += 2) {
            const f32x2 y = gelu_poly2((f32x2){v[e], v[e + 1]});
            v[e] = y.x; v[e + 1] = y.y;
        }
    } else if constexpr (KIND == 0 && N % 2 == 0) {
#pragma unroll
        for (int e = 0; e < N; e += 2) {
            const f32x2 y = gelu_erf2((f32x2){v[e], v[e + 1]});
            v[e] = y.x; v[e + 1] = y.y;
        }
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = KIND == 1 ? gelu_fast(v[e]) : gelu_erf(v[e]);
    }
}
// which form a narrow (16/8-bit) result of operand class T in an NS-plane mode takes
template <class T, int NS> struct GeluKind { static constexpr int value = NS == 2 ? 0 : 2; };
template <> struct GeluKind<_Float16, 1> { static constexpr int value = 1; };
template <> struct GeluKind<f16c8, 2> { static constexpr int value = 0; };     // strict: exact erf

template <bool FAST> __device__ __forceinline__ float gelu_sel(float x) { return FAST ? gelu_fast(x) : gelu_erf(x); }

// ---- 8 consecutive elements of a GEMM A-operand from fp32 (every producer: LayerNorm, GEMM epilogue, attention output,
// im2col / patchify / gather): base = plane 0, e = element index (multiple of 8), plane = 2-byte units between the planes
// (hi, lo) of a split 16-bit class: hi = T(v), lo = v - hi, both as fp32.  The value is made opaque first (as in store_cvt): hi must
// be the rounding of the SAME fp32 number lo is taken against.  Left alone, hipcc folds the producer's last multiply into the f16
// conversion (v_fma_mix*: one rounding) at one use and not at the other, and in near-tie cases the stored hi and the hi that lo was
// computed against differ by an f16 ulp (found by tests/test_gpu_ops.py::test_attention_prefix_split, round 4; bf16 has no such
// instruction and is unaffected).
template <class T> __device__ __forceinline__ void split_hi_lo(float v, float& hi, float& lo) {
    asm("" : "+v"(v));
    hi = to_f32<T>(from_f32<T>(v));
    asm("" : "+v"(hi));
    lo = v - hi;
}
template <class T, int NS>
__device__ __forceinline__ void store_operand8(T* base, int64_t plane, int64_t e, const float (&v)[8]) {
    T* dst = base + e;
    if constexpr (NS == 2) {
        float hi8[8], lo8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) split_hi_lo<T>(v[j], hi8[j], lo8[j]);
        store_cvt<T, 8>(dst, hi8);
        store_cvt<T, 8>(dst + plane, lo8);
    } else {
        store_cvt<T, 8>(dst, v);
    }
}
// F16C8: 8 consecutive elements starting at element index e (e % 8 == 0, rows are multiples of 32 long)
__device__ __forceinline__ int64_t f16c8_lo_index(int64_t e) {            // byte index in the lo8 plane of element e's 8-group
    const int g = (int)(e >> 3) & 3;                                         // 8-group inside the 32-block: k = 8 g ..
    return (e & ~(int64_t)31) + (((g & 1) << 4) | ((g >> 1) << 3));          // a = g >> 1, h = g & 1  ->  16 h + 8 a
}
template <int N>
__device__ __forceinline__ void f16c8_split(const float (&v)[N], _Float16 (&hi)[N], float (&lo_scaled)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        // (MODE.FP16_OVFL is set by the calling kernel: hi saturates at +-65504, the e4m3 image of lo at +-448.)  The value is made
        // opaque first, as in store_cvt: hi must be the rounding of the SAME fp32 number lo is taken against -- left alone, hipcc folds
        // the producer's last multiply into the conversion in some contexts and not into the subtraction (op test: lo planes off by
        // an f16 ulp in the attention epilogue).
        float c = v[j];
        asm("" : "+v"(c));
        hi[j] = (_Float16)c;
        lo_scaled[j] = (c - (float)hi[j]) * (float)(1 << BD_F16C8_D);
    }
}
template <> __device__ __forceinline__ void store_operand8<f16c8, 2>(f16c8* base, int64_t plane, int64_t e, const float (&v)[8]) {
    _Float16 h[8];
    float lo[8];
    f16c8_split<8>(v, h, lo);
    f16x8 hv;
#pragma unroll
    for (int j = 0; j < 8; ++j) hv[j] = h[j];
    int l0 = 0, l1 = 0;
    l0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], l0, false); l0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], l0, true);
    l1 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4], lo[5], l1, false); l1 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[6], lo[7], l1, true);
    unsigned char* lo_plane = (unsigned char*)(base + plane);
#if defined(BD_STORE_NT) && BD_STORE_NT
    __builtin_nontemporal_store(__builtin_bit_cast(u128, hv), (u128*)(base + e));
#else
    *(u128*)(base + e) = __builtin_bit_cast(u128, hv);
#endif
    *(uint2*)(lo_plane + f16c8_lo_index(e)) = make_uint2((unsigned)l0, (unsigned)l1);
}
// 4 consecutive elements (LayerNorm, attention outputs): e % 4 == 0
__device__ __forceinline__ void f16c8_store4(f16c8* base, int64_t plane, int64_t e, const float (&v)[4]) {
    _Float16 h[4];
    float lo[4];
    f16c8_split<4>(v, h, lo);
    typedef __attribute__((__vector_size__(4 * sizeof(_Float16)))) _Float16 f16x4;
    f16x4 hv = {h[0], h[1], h[2], h[3]};
    int l0 = 0;
    l0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], l0, false); l0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], l0, true);
    *(f16x4*)(base + e) = hv;
    *(int*)((unsigned char*)(base + plane) + f16c8_lo_index(e & ~(int64_t)7) + (e & 4)) = l0;
}

// ---- LayerNorm of ONE row by ONE wave64 (fp32 statistics, two-pass variance as nn.LayerNorm computes it): a row of <= 1024 fp32
// lives in 4 float4 registers per lane (lane l owns columns (i*64 + l)*4 .. +3).  Shared by the stand-alone kernel (norm.hip) and
// by the persistent GEMMs' fused form (gemm.hip: the workgroup that completes a 256-row panel of the residual stream normalises
// it), so both produce the same bits.  NT: non-temporal row loads (stand-alone kernel: the stream is far larger than the L2s).
template <bool NT>
__device__ __forceinline__ void ln_row_load(const float* __restrict__ xr, int cols, int lane, f32x4 (&t)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            if constexpr (NT) t[i] = __builtin_nontemporal_load((const f32x4*)(xr + c));
            else t[i] = *(const f32x4*)(xr + c);
        }
    }
}
// The same row fetched with device-coherent (sc1) loads: they bypass this CU's L1 and the XCD's L2, so rows that OTHER workgroups
// wrote with write-through stores earlier in the same launch are read correctly without an acquire fence (buffer_inv sc1 drops the
// whole L2's clean lines: every CU of the XCD then re-fetches its GEMM operands -- measured 3.4x on the launch).  Inline asm: the
// caller waits with ln_rows_wait() before it touches the registers.
__device__ __forceinline__ void ln_row_load_sc1(const float* xr, int cols, int lane, f32x4 (&t)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(t[i]) : "v"(xr + c) : "memory");
    }
}
__device__ __forceinline__ void ln_rows_wait(f32x4 (&a)[4], f32x4 (&b)[4]) {      // the registers pass THROUGH the wait: nothing moves above it
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : : "memory");
}
template <class T, int NS>
__device__ __forceinline__ void ln_row_finish(const f32x4 (&t)[4], const float* __restrict__ gamma, const float* __restrict__ beta,
                                              float eps, T* __restrict__ out16, int64_t out16_plane, float* __restrict__ out32_row,
                                              int64_t orow, int cols, int lane) {
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            v[i] = make_float4(t[i][0], t[i][1], t[i][2], t[i][3]);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < cols) {
            float y[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd,
                          (v[i].w - mean) * rstd};
            if (gamma) {
                const float4 g = *(const float4*)(gamma + c);
                y[0] *= g.x; y[1] *= g.y; y[2] *= g.z; y[3] *= g.w;
            }
            if (beta) {
                const float4 bb = *(const float4*)(beta + c);
                y[0] += bb.x; y[1] += bb.y; y[2] += bb.z; y[3] += bb.w;
            }
            if (out32_row) *(float4*)(out32_row + c) = make_float4(y[0], y[1], y[2], y[3]);
            if (out16) {
                if constexpr (__is_same(T, f16c8)) {
                    f16c8_store4(out16, out16_plane, orow * cols + c, y);
                } else if constexpr (NS == 2) {
                    float hi4[4], lo4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) split_hi_lo<T>(y[j], hi4[j], lo4[j]);
                    store_cvt<T, 4>(out16 + orow * cols + c, hi4);
                    store_cvt<T, 4>(out16 + out16_plane + orow * cols + c, lo4);
                } else {
                    store_cvt<T, 4>(out16 + orow * cols + c, y);
                }
            }
        }
    }
}

template <class T, int NS, bool NT>
__device__ __forceinline__ void ln_row(const float* __restrict__ xr, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float eps, T* __restrict__ out16, int64_t out16_plane, float* __restrict__ out32_row,
                                       int64_t orow, int cols, int lane) {
    f32x4 t[4];
    ln_row_load<NT>(xr, cols, lane, t);
    ln_row_finish<T, NS>(t, gamma, beta, eps, out16, out16_plane, out32_row, orow, cols, lane);
}

// gemm_f16c8.hip: how many launches of the same shape the calling thread is enqueueing SIDE BY SIDE on different streams (forward.hip's
// sub-batch lanes set it around their per-lane calls; 1 otherwise).  A host-side hint for tile-form choices that count free CUs; it never
// changes a result (every row is independent of the tile form).
int& bd_concurrent_launches();

// attention.hip: bd_attention_q with the opt-in latency forms (bd_*_weights.latency_mode): launches whose 256-query workgroups would
// occupy at most a quarter of the CUs take the 128-query kernel instead (twice the workgroups; not bit-identical to the 256-query form)
int bd_attention_q_forms(const void* qkv, int64_t qkv_plane, void* out, int64_t out_plane, int batch, int seq, int heads, int head_dim,
                         float scale, const int32_t* q_view, int q_len, int prec, int latency_forms, void* stream);

// trace.hip
int bd_trace_open(hipStream_t s, int kind, int M, int N, int K);
void bd_trace_close(hipStream_t s, int slot);

#define BD_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
