// Fused multi-head self-attention  softmax(scale * Q K^T) V  on packed qkv, gfx950.
//
// Replaces flash_attn_func / F.scaled_dot_product_attention (BETR, head_dim 96, seq T*256) and
// xformers memory_efficient_attention / the naive softmax path (DINOv2, head_dim 64, seq 261).
//
// Layout trick (all three MFMAs keep "query = lane & 31"):
//   S^T = K . Q^T      A = K tile rows from LDS (key = lane&31, 8 d per lane), B = Q^T from registers.
//                      Lane (q, h) ends up holding scores of ITS query for 16 keys per 32-key tile:
//                      key = (r&3) + 8*(r>>2) + 4*h.  Row max / sum are therefore in-lane reductions
//                      plus ONE exchange with lane^32.
//   O^T = V^T . P^T    B = P^T: the lane's own probabilities, regs 8t..8t+7, used directly as the 8
//                      k-slots of 16-key group t -- no cross-lane movement at all, because the V^T
//                      image in LDS is written with the same key permutation (quads 1 and 2 of every
//                      16-key group swapped).  A = V^T rows (d = lane&31) read with one ds_read_b128.
//                      The O^T accumulator again has query = lane&31, so the online-softmax rescale
//                      and the final 1/l are per-lane scalars.
// K tile rows are padded by 16 B and the V^T image is XOR-swizzled so that each ds_read_b128 lane
// group hits 16 distinct bank slots.  K/V tiles of 64 keys are register-prefetched one tile ahead.
// NS = 2 is the split-bf16 strict mode: Q, K, V, P each carry (hi, lo) and every product issues
// hi*hi + hi*lo + lo*hi.
#include "bd_common.h"
#include <type_traits>

#ifdef BD_ATTN_PROBE
// Measurement build only (tools/attn_phase_probe.py): per-wave shader-clock stamps of the ping-pong kernel's segments.
__device__ unsigned* bd_attn_probe_buf = nullptr;
extern "C" int bd_attn_probe_set(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(bd_attn_probe_buf), &buf, sizeof(buf)); }
#define AP(idx) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if ((idx) < 64) probe_ts = (lane == (idx)) ? (unsigned)t__ : probe_ts; }
#else
#define AP(idx) {}
#endif

namespace {

struct AttnArgs {
    const void* qkv; int64_t qkv_plane;
    void* out; int64_t out_plane;
    int batch, seq, heads;
    float scale_log2e;
    const int32_t* q_view;   // optional: queries are rows [q_view[b]*q_len, +q_len) of every sequence, output compact
    int q_len;               // number of query rows per sequence (== seq when q_view is NULL)
    int q_off;               // (q_view == NULL) first query row inside every sequence: queries are rows [q_off, q_off + q_len)
    int out_rows;            // rows per sample in `out` (q_len: compact; seq with q_off: the result lands in rows [q_off, q_off + q_len))
};

constexpr int KT = 64;   // keys per tile

// 16-byte chunk swizzle of V^T row d (128-byte rows: every row aliases the same 32 banks).
//  * ds_read_b128 (O^T A operand): a 16-lane group reads 16 consecutive rows, same logical chunk -> the 8 rows of
//    equal parity must land on 8 distinct chunks: (d >> 1) & 7 does that, and XOR-ing in (d >> 4), constant over an
//    aligned 16-row group, keeps it a bijection;
//  * ds_write_b64 (transposing stage): the lanes of a group hold d = 8*dc + const for 12 consecutive dc -> with
//    (d >> 1) alone only 2 of 8 chunks are used (6-way conflict: 43 % of LDS cycles in profiles/r1_attn_pmc.md);
//    the (d >> 4) term spreads them over all 8 (<= 2-way).
__device__ __forceinline__ int vswz(int d) { return ((d >> 1) ^ (d >> 4)) & 7; }

// low / high 16-bit halves of two dwords as one dword (the 4 x 4 register transposition of the V staging): one v_perm_b32 each
// (written as and / shift / or hipcc emits two VALU instructions per pair: 32 instead of 16 per staged micro-tile)
__device__ __forceinline__ unsigned pack_lo16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }   // (a & 0xffff) | (b << 16)
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // (a >> 16) | (b & 0xffff0000)

// OUTSPLIT: single-pass f16 attention whose result is written as split-bf16 (hi, lo) planes -- the strict mode's
// attention (its GEMMs stay split-bf16 x3): q/k are RMS-normalised and P is in [0, 1], so one f16 pass costs ~1e-4 on
// the logits while the x3 attention kernel is register-bound at one wave per SIMD.
template <class T, int NS, int HD, int NW, int OUTMODE = 0>
__global__ __launch_bounds__(NW * 64, NS == 1 && HD == 64 ? 3 : 2) void attn_kernel(const AttnArgs p) {      // (HIP: 2nd argument = min waves per SIMD)
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    typedef typename Op16<T>::vec8 vec8;
    constexpr int NT = NW * 64;
    constexpr int DCH = HD / 8;                  // 16-byte chunks per K/V row
    constexpr int KSTRIDE = HD * 2 + 16;         // padded K row (bytes)
    constexpr int K_BYTES = KT * KSTRIDE;
    constexpr int V_BYTES = HD * 128;            // V^T image: HD rows x 64 keys x 2 B
    constexpr int PLANE_BYTES = K_BYTES + V_BYTES;
    constexpr int KCH = KT * DCH;                // K chunks per tile
    constexpr int KCPT = (KCH + NT - 1) / NT;
    constexpr int VMT = (KT / 4) * DCH;          // V micro-tiles (4 keys x 8 d)
    static_assert(VMT <= NT, "one V micro-tile per thread");
    constexpr int QB = NW * 32;
    constexpr int DM = HD / 32;                  // O^T M-tiles
    constexpr int KS = HD / 16;                  // k-steps of S^T
    constexpr int BUF_BYTES = NS * PLANE_BYTES;
    constexpr int NBUF = NS == 1 ? 2 : 1;      // strict mode keeps one buffer (two planes already fill the LDS budget)
    __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * BUF_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
#ifdef BD_ATTN_PROBE
    unsigned probe_ts = 0;
    AP(60)
#endif
    const int lq = lane & 31, lh = lane >> 5;
    const int seq = p.seq, heads = p.heads;
    const int q_len = p.q_len;
    const int nqb = (q_len + QB - 1) / QB;

    // XCD-aware remap: consecutive work items (same batch*head, consecutive q-blocks) share K/V and
    // are kept on one XCD's L2.
    int wg;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int qb = wg % nqb;
    const int bh = wg / nqb;
    const int head = bh % heads, b = bh / heads;

    const int ld = 3 * heads * HD;                         // elements per token row (32-bit offsets inside a sample)
    const T* base = (const T*)p.qkv + (int64_t)b * seq * ld + head * HD;
    const T* qbase = base;
    const T* kbase = base + heads * HD;
    const T* vbase = base + 2 * heads * HD;
    const int64_t plane = p.qkv_plane;

    // ---- Q fragments (B operand of S^T): lane (q, h) holds Q[q][ks*16 + h*8 .. +7]
    const int q0 = qb * QB + wid * 32;                     // local query index (within the query range)
    const int qbase_row = p.q_view ? p.q_view[b] * q_len : p.q_off;
    int qrow = q0 + lq; qrow = (qrow < q_len ? qrow : q_len - 1) + qbase_row;
    vec8 qf[NS][KS];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[s][ks] = as_vec8<T>(*(const u128*)(qbase + s * plane + (unsigned)(qrow * ld + ks * 16 + lh * 8)));

    // ---- loader coordinates
    int kc_row[KCPT], kc_col[KCPT];
    unsigned kc_off[KCPT];                                  // element offset of the chunk in tile 0
#pragma unroll
    for (int i = 0; i < KCPT; ++i) {
        const int c = tid + NT * i;
        kc_row[i] = c / DCH;
        kc_col[i] = c % DCH;
        kc_off[i] = (unsigned)(kc_row[i] * ld + kc_col[i] * 8);
    }
    const int vm_kq = tid / DCH, vm_dc = tid % DCH;     // key quad (0..15), d chunk
    const bool vm_active = tid < VMT;
    const unsigned vm_off = (unsigned)(vm_kq * 4 * ld + vm_dc * 8);
    const bool ragged = (seq % KT) != 0;
    // destination of the micro-tile in the V^T image: 16-key group g, permuted quad
    int vdst;
    {
        const int g = vm_kq >> 2, qi = vm_kq & 3;
        const int qp = (qi == 1) ? 2 : (qi == 2 ? 1 : qi);
        vdst = ((g * 2 + (qp >> 1)) << 4) | ((qp & 1) << 3);   // chunk<<4 | 8-byte half  (pre-swizzle)
    }

    u128 rk[NS][KCPT], rv[NS][4];
    // K / V tiles by BUFFER loads (round 5, as attn_kernel_pp): one wave-uniform descriptor per plane over this (sample, head)'s seq rows, the
    // tile / row origin in the scalar offset, loop-invariant 32-bit per-lane offsets.  Rows past a ragged sequence's end are OUT OF RANGE of the
    // descriptor and read as ZERO (hardware bounds check) -- no per-lane row clamp; their scores are masked to -inf and their P is exactly 0,
    // so every valid output keeps its bits (the clamped form read row seq - 1 there, equally without effect).
    __amdgpu_buffer_rsrc_t rsrc[NS];
    {
        const int bytes = __builtin_amdgcn_readfirstlane(seq * ld * (int)sizeof(T));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uintptr_t a = (uintptr_t)(base + s * plane);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
            rsrc[s] = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
        }
    }
    const int row_bytes = __builtin_amdgcn_readfirstlane(ld * (int)sizeof(T));
    int k_voff[KCPT];
#pragma unroll
    for (int i = 0; i < KCPT; ++i) k_voff[i] = ((int)kc_off[i] + heads * HD) * (int)sizeof(T);
    const int v_voff = ((int)vm_off + 2 * heads * HD) * (int)sizeof(T);
#define LOAD_TILE(kt)                                                                         \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                          \
        _Pragma("unroll") for (int i = 0; i < KCPT; ++i) {                                    \
            if (KCH % NT == 0 || tid + NT * i < KCH)                                          \
                rk[s][i] = __builtin_bit_cast(u128, __builtin_amdgcn_raw_buffer_load_b128(rsrc[s], k_voff[i], (kt) * KT * row_bytes, 0)); \
        }                                                                                     \
        if (vm_active) {                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                     \
                rv[s][j] = __builtin_bit_cast(u128, __builtin_amdgcn_raw_buffer_load_b128(rsrc[s], v_voff, ((kt) * KT + j) * row_bytes, 0)); \
        }                                                                                     \
    }
#define STORE_TILE(buf)                                                                        \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                          \
        unsigned char* kl = lds + (buf) * BUF_BYTES + s * PLANE_BYTES;                        \
        unsigned char* vl = kl + K_BYTES;                                                     \
        _Pragma("unroll") for (int i = 0; i < KCPT; ++i)                                      \
            if (KCH % NT == 0 || tid + NT * i < KCH)                                          \
                *(u128*)(kl + kc_row[i] * KSTRIDE + kc_col[i] * 16) = rk[s][i];               \
        if (vm_active) {                                                                      \
            _Pragma("unroll") for (int w = 0; w < 4; ++w) {                                   \
                const unsigned a0 = rv[s][0][w], a1 = rv[s][1][w], a2 = rv[s][2][w], a3 = rv[s][3][w]; \
                const int d = vm_dc * 8 + 2 * w;                                              \
                uint2 lo, hi;                                                                 \
                lo.x = pack_lo16(a0, a1); lo.y = pack_lo16(a2, a3);       \
                hi.x = pack_hi16(a0, a1); hi.y = pack_hi16(a2, a3); \
                *(uint2*)(vl + d * 128 + (vdst ^ (vswz(d) << 4))) = lo;                \
                *(uint2*)(vl + (d + 1) * 128 + (vdst ^ (vswz(d + 1) << 4))) = hi;    \
            }                                                                                 \
        }                                                                                     \
    }

    f32x16 oacc[DM];
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.scale_log2e;

    const int nt = (seq + KT - 1) / KT;
    LOAD_TILE(0)
    STORE_TILE(0)
    __syncthreads();
    AP(61)
    // One key tile.  Two instances: TAIL 0 = any tile (a ragged last one masks its scores), TAIL 2 = a ragged last tile of <= 16 keys
    // (DINOv2: 261 = 4 x 64 + 5), which skips the second 32-key M-tile of S^T and the three 16-key P.V groups that lie wholly past
    // the sequence -- their scores would be masked to -inf and their P exactly 0, so the result is bit-identical and the tile costs
    // 6 instead of 16 MFMAs per plane product.  The tail is PEELED: the same skip as wave-uniform branches inside one loop body
    // costs the main loop its schedule (seq 261: 121 -> 182 us), and so does skipping the staging of the tail's unused rows
    // (profiles/r3_attention_tail.md).  hd 64 only: the hd-96 split-plane instances have no registers for a second tile instance.
    const bool small_tail = HD == 64 && nt > 1 && seq % KT != 0 && seq % KT <= 16;
    auto tile = [&](const int kt, auto tail_c) {
        constexpr int TAIL = decltype(tail_c)::value;
        constexpr int KMN = TAIL == 2 ? 1 : 2, GN = TAIL == 2 ? 1 : 4;
        if (kt + 1 < nt) { LOAD_TILE(kt + 1) }
        const unsigned char* cur = lds + (NBUF == 2 ? (kt & 1) : 0) * BUF_BYTES;
        if (kt < 7) AP(kt * 8)

        // ---- S^T = K . Q^T  (two 32-key M-tiles)
        f32x16 sacc[KMN];
#pragma unroll
        for (int km = 0; km < KMN; ++km) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[km][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                vec8 kf[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    kf[s] = as_vec8<T>(*(const u128*)(cur + s * PLANE_BYTES + (km * 32 + lq) * KSTRIDE + (ks * 2 + lh) * 16));
                if (NS == 2) {
                    sacc[km] = Op16<T>::mfma(kf[NS - 1], qf[0][ks], sacc[km]);
                    sacc[km] = Op16<T>::mfma(kf[0], qf[NS - 1][ks], sacc[km]);
                }
                sacc[km] = Op16<T>::mfma(kf[0], qf[0][ks], sacc[km]);
            }
        }
        // ---- online softmax (base-2), per query = per lane pair (l, l^32).  Scores stay raw; the softmax scale
        // (times log2 e) is folded into the exponent:  p = exp2(s*sc - m*sc)  = one FMA + v_exp_f32 per score.
        if (kt < 7) AP(kt * 8 + 1)
        float tmax = -INFINITY;
        if (TAIL == 2 || (kt + 1) * KT > seq) {
#pragma unroll
            for (int km = 0; km < KMN; ++km)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * KT + km * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= seq) sacc[km][r] = -INFINITY;
                }
        }
#pragma unroll
        for (int km = 0; km < KMN; ++km)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, sacc[km][r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);                     // raw-score units (sc > 0)
        if (!__all(m_new == m_run)) {                               // some row's max moved: rescale O and l
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
            m_run = m_new;
        }
        const float mneg = -m_run * sc;
        float psum = 0.f;
#pragma unroll
        for (int km = 0; km < KMN; ++km)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[km][r], sc, mneg));
                sacc[km][r] = pv;
                psum += pv;
            }
        l_run += psum;
        if (kt < 7) AP(kt * 8 + 2)

        // ---- O^T += V^T . P^T   (four 16-key groups)
#pragma unroll
        for (int g = 0; g < GN; ++g) {
            vec8 pf[NS];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float pv = sacc[g >> 1][(g & 1) * 8 + j];
                const T hi = from_f32<T>(pv);
                pf[0][j] = hi;
                if (NS == 2) pf[NS - 1][j] = from_f32<T>(pv - to_f32<T>(hi));
            }
#pragma unroll
            for (int dm = 0; dm < DM; ++dm) {
                const int d = dm * 32 + lq;
                vec8 vf[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    vf[s] = as_vec8<T>(*(const u128*)(cur + s * PLANE_BYTES + K_BYTES + d * 128 +
                                                        (((g * 2 + lh) ^ vswz(d)) << 4)));
                if (NS == 2) {
                    oacc[dm] = Op16<T>::mfma(vf[NS - 1], pf[0], oacc[dm]);
                    oacc[dm] = Op16<T>::mfma(vf[0], pf[NS - 1], oacc[dm]);
                }
                oacc[dm] = Op16<T>::mfma(vf[0], pf[0], oacc[dm]);
            }
        }
        // the other buffer was last read in iteration kt-1, which every wave left through the barrier below
        if (kt < 7) AP(kt * 8 + 3)
        if (NBUF == 1) __syncthreads();                      // single buffer: everyone must be done reading first
        if (kt < 7) AP(kt * 8 + 4)
        if (kt + 1 < nt) { STORE_TILE(NBUF == 2 ? ((kt + 1) & 1) : 0) }
        if (kt < 7) AP(kt * 8 + 5)
        __syncthreads();
        if (kt < 7) AP(kt * 8 + 6)
    };
    for (int kt = 0; kt < nt - (small_tail ? 1 : 0); ++kt) tile(kt, std::integral_constant<int, 0>{});
    if (small_tail) tile(nt - 1, std::integral_constant<int, 2>{});
#undef LOAD_TILE
#undef STORE_TILE

    // ---- finalise: O[q][d] = O^T / l ; lane (q, h) owns d = dm*32 + 8*(r>>2) + 4*h + (r&3)
    AP(62)
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + lq;
    if (q < q_len) {
        if (OUTMODE == 2) {
            fp8e4* orow = (fp8e4*)p.out + ((int64_t)b * p.out_rows + (p.out_rows == q_len ? 0 : p.q_off) + q) * (heads * HD) + head * HD;
#pragma unroll
            for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    float v4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = oacc[dm][rq * 4 + j] * inv;
                    store_cvt<fp8e4, 4>(orow + dm * 32 + 8 * rq + 4 * lh, v4);
                }
        } else if (OUTMODE == 3) {
            // F16C8 operand (the proj GEMM's A in the round-2 strict mode): f16 hi plane + k-permuted lo8 plane
            const int64_t e0 = ((int64_t)b * p.out_rows + (p.out_rows == q_len ? 0 : p.q_off) + q) * (heads * HD) + head * HD;
#pragma unroll
            for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    float v4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = oacc[dm][rq * 4 + j] * inv;
                    f16c8_store4((f16c8*)p.out, p.out_plane, e0 + dm * 32 + 8 * rq + 4 * lh, v4);
                }
        } else if (OUTMODE == 1 || OUTMODE == 4) {
            // (hi, lo) planes of ANOTHER 16-bit type than the kernel's operands: 1 = split-bf16, 4 = split-f16 (BD_PREC_F16X3)
            typedef typename std::conditional<OUTMODE == 1, __bf16, _Float16>::type OT;
            OT* orow = (OT*)p.out + ((int64_t)b * p.out_rows + (p.out_rows == q_len ? 0 : p.q_off) + q) * (heads * HD) + head * HD;
#pragma unroll
            for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d0 = dm * 32 + 8 * rq + 4 * lh;
                    typedef __attribute__((__vector_size__(4 * sizeof(OT)))) OT ovec4;
                    ovec4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float h, l;
                        split_hi_lo<OT>(oacc[dm][rq * 4 + j] * inv, h, l);
                        hi[j] = from_f32<OT>(h);
                        lo[j] = from_f32<OT>(l);
                    }
                    *(ovec4*)(orow + d0) = hi;
                    *(ovec4*)(orow + p.out_plane + d0) = lo;
                }
        } else {
            T* orow = (T*)p.out + ((int64_t)b * p.out_rows + (p.out_rows == q_len ? 0 : p.q_off) + q) * (heads * HD) + head * HD;
#pragma unroll
            for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d0 = dm * 32 + 8 * rq + 4 * lh;
                    typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4;
                    vec4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (NS == 2) {
                            float h, l;
                            split_hi_lo<T>(oacc[dm][rq * 4 + j] * inv, h, l);
                            hi[j] = from_f32<T>(h);
                            lo[j] = from_f32<T>(l);
                        } else {
                            hi[j] = from_f32<T>(oacc[dm][rq * 4 + j] * inv);
                        }
                    }
                    *(vec4*)(orow + d0) = hi;
                    if (NS == 2) *(vec4*)(orow + p.out_plane + d0) = lo;
                }
        }
    }
#ifdef BD_ATTN_PROBE
    AP(63)
    if (bd_attn_probe_buf && blockIdx.x < 512) bd_attn_probe_buf[((size_t)blockIdx.x * 8 + wid) * 64 + lane] = probe_ts;
#endif
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined form (round 2) for the single-plane operand classes: 8 waves, 256 queries per workgroup, one per CU.
//
// Why (tools/attn_phase_probe.py, profiles/r2_attention.md): attn_kernel runs S(t), softmax(t), PV(t) one after the other in
// every wave; with two independent 4-wave workgroups per CU the two waves of a SIMD fell into the same phase (VALU 55 % + MFMA
// 29 % busy adding up instead of overlapping).  A first round-2 attempt put the two waves of a SIMD into ONE workgroup, a
// barrier-enforced segment apart (one in its MFMA segment while the other does its softmax): measured 1376-1759 cycles for a
// 768-cycle MFMA segment and 504 vs 1419 cycles for the same softmax depending on which wave wins the issue arbitration --
// a dense VALU stream and an MFMA stream of two different waves do not share a SIMD's issue port gracefully.  What does work
// is interleaving them inside ONE instruction stream: an MFMA occupies the issue port for 4 of its 32 cycles, and the same
// wave's next 6-7 VALU instructions ride in the rest.  So each wave here runs, per key tile t,
//     24 MFMA slots = PV(t-1) (12) + S(t+1) (12),   with the VALU work of softmax(t) spread over the slots,
// i.e. the softmax of a tile executes under the matrix work of its two neighbours (two S accumulator sets, two P fragment
// sets; order pinned with sched_barrier).  Operand fragments are read three slots ahead into a ring of four registers.
// Staging: all 512 threads fetch tile t+2 (global -> registers) at the top of iteration t and store it (K as is, V transposed,
// attn_kernel's LDS images) at its end; K ring 2 tiles, V^T ring 4 tiles, ONE barrier per tile.
template <class T, int HD, int OUTMODE>
__global__ __launch_bounds__(512, 1) void attn_kernel_pp(const AttnArgs p) {
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    typedef typename Op16<T>::vec8 vec8;
    constexpr int NT = 512;
    constexpr int DCH = HD / 8;
    constexpr int KSTRIDE = HD * 2 + 16;
    constexpr int K_BYTES = KT * KSTRIDE;
    constexpr int V_BYTES = HD * 128;
    constexpr int KBUF = 2, VBUF = 4;
    constexpr int VMT = (KT / 4) * DCH;
    static_assert(VMT <= NT, "one V micro-tile per thread");
    constexpr int QB = 256;
    constexpr int DM = HD / 32;
    constexpr int KS = HD / 16;
    constexpr int NPV = 4 * DM, NS_ = 2 * KS, NSLOT = NPV + NS_;
    static_assert(NSLOT >= 22, "softmax schedule below needs 22 slots");
    // output staging (plain 16-bit and F16C8 results): one row image per query, [hi plane | lo8 plane], 16 bytes of padding
    constexpr bool OUT_VIA_LDS = (OUTMODE == 0 || OUTMODE == 3);
    constexpr int OROW = (OUTMODE == 3) ? HD * 3 : HD * 2;
    constexpr int OSTRIDE = OROW + 16;
    constexpr int RING_BYTES = KBUF * K_BYTES + VBUF * V_BYTES;
    constexpr int LDS_BYTES = (OUT_VIA_LDS && 8 * 32 * OSTRIDE > RING_BYTES) ? 8 * 32 * OSTRIDE : RING_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    unsigned char* const vring = lds;
    unsigned char* const kring = lds + VBUF * V_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, lh = lane >> 5;
    const int seq = p.seq, heads = p.heads;
    const int q_len = p.q_len;
    const int nqb = (q_len + QB - 1) / QB;

    int wg;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int qb = wg % nqb;
    const int bh = wg / nqb;
    const int head = bh % heads, b = bh / heads;

    const int ld = 3 * heads * HD;
    const T* base = (const T*)p.qkv + (int64_t)b * seq * ld + head * HD;
    const T* kbase = base + heads * HD;
    const T* vbase = base + 2 * heads * HD;

    const int q0 = qb * QB + wid * 32;
    const int qbase_row = p.q_view ? p.q_view[b] * q_len : p.q_off;
    int qrow = q0 + lq; qrow = (qrow < q_len ? qrow : q_len - 1) + qbase_row;
    vec8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = as_vec8<T>(*(const u128*)(base + (unsigned)(qrow * ld + ks * 16 + lh * 8)));

    // ---- staging coordinates
    // K tile: 8 threads per key row; thread (row, j) moves 16-byte pieces j and j + 8 (< DCH) of the row, so that the second piece's
    // global and LDS addresses are the first one's plus an immediate (no second set of address registers: the loop's budget is full)
    static_assert(KT * 8 == NT && DCH > 8 && DCH <= 16, "K staging: 8 threads per key row, two pieces each");
    const int kc_row = tid >> 3, kc_col = tid & 7;
    const bool kc_two = kc_col < DCH - 8;
    const int vm_kq = tid / DCH, vm_dc = tid % DCH;
    const bool vm_active = tid < VMT;
    // The host launches this kernel for whole key tiles only (dispatch: q_len % 256 == 0 and seq % 64 == 0).  K / V tiles are fetched with
    // BUFFER loads: one wave-uniform descriptor over this (sample, head)'s rows, the tile / row origin in the SCALAR offset, and ONE
    // loop-invariant 32-bit byte offset per lane for K and one for V -- no vector address arithmetic per tile (round 5: the flat form
    // cost five 64-bit multiply-adds + shifts per tile and wave in a loop whose VALU is the bound; MI355X guide T8 / T20).
    int vdst;
    {
        const int g = vm_kq >> 2, qi = vm_kq & 3;
        const int qp = (qi == 1) ? 2 : (qi == 2 ? 1 : qi);
        vdst = ((g * 2 + (qp >> 1)) << 4) | ((qp & 1) << 3);
    }
    u128 rk[2], rv[4];
    __amdgpu_buffer_rsrc_t rsrc;
    {
        const uintptr_t a = (uintptr_t)base;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(seq * ld * (int)sizeof(T)), 0x00020000);
    }
    const int k_voff = (kc_row * ld + kc_col * 8 + heads * HD) * (int)sizeof(T);
    const int v_voff = (vm_kq * 4 * ld + vm_dc * 8 + 2 * heads * HD) * (int)sizeof(T);
    const int row_bytes = __builtin_amdgcn_readfirstlane(ld * (int)sizeof(T));
    auto load_tile = [&](int kt) {
        const int t_off = kt * KT * row_bytes;                        // scalar
        rk[0] = __builtin_bit_cast(u128, __builtin_amdgcn_raw_buffer_load_b128(rsrc, k_voff, t_off, 0));
        if (kc_two) rk[1] = __builtin_bit_cast(u128, __builtin_amdgcn_raw_buffer_load_b128(rsrc, k_voff + 128, t_off, 0));
        if (vm_active) {
#pragma unroll
            for (int j = 0; j < 4; ++j) rv[j] = __builtin_bit_cast(u128, __builtin_amdgcn_raw_buffer_load_b128(rsrc, v_voff, t_off + j * row_bytes, 0));
        }
    };
    auto store_tile = [&](int kt, int kbi, int vbi) {
        (void)kt;
        unsigned char* kl = kring + kbi * K_BYTES;
        unsigned char* vl = vring + vbi * V_BYTES;
        *(u128*)(kl + kc_row * KSTRIDE + kc_col * 16) = rk[0];
        if (kc_two) *(u128*)(kl + kc_row * KSTRIDE + kc_col * 16 + 128) = rk[1];
        if (vm_active) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const unsigned a0 = rv[0][w], a1 = rv[1][w], a2 = rv[2][w], a3 = rv[3][w];
                const int d = vm_dc * 8 + 2 * w;
                uint2 lo, hi;
                lo.x = pack_lo16(a0, a1); lo.y = pack_lo16(a2, a3);
                hi.x = pack_hi16(a0, a1); hi.y = pack_hi16(a2, a3);
                *(uint2*)(vl + d * 128 + (vdst ^ (vswz(d) << 4))) = lo;
                *(uint2*)(vl + (d + 1) * 128 + (vdst ^ (vswz(d + 1) << 4))) = hi;
            }
        }
    };

    f32x16 oacc[DM];
#pragma unroll
    for (int i = 0; i < DM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = p.scale_log2e;
    const int nt = (seq + KT - 1) / KT;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // ---- prologue: tiles 0 and 1 staged; the V^T buffer of "tile -1" zeroed (iteration 0 runs PV(-1) = 0 . 0 through it)
    load_tile(0);
    store_tile(0, 0, 0);
    if (nt > 1) { load_tile(1); store_tile(1, 1, 1); }
    for (int i = tid * 16; i < V_BYTES; i += NT * 16) *(u128*)(vring + (VBUF - 1) * V_BYTES + i) = (u128){0u, 0u, 0u, 0u};
    __syncthreads();

    // S[t & 1] = scores of tile t, P[t & 1] = probability fragments of tile t.  The loop is unrolled over the ring period (4)
    // so that every LDS address is a loop-invariant per-lane register plus an immediate and no register set is ever copied.
    // The P fragments are ONE set, rewritten in place: PV(t-1) reads group g in slots 3 g .. 3 g + 2 and softmax(t) writes group g in
    // slots 5 + 4 g .. 8 + 4 g, always later.
    f32x16 S[2][2];
    vec8 P[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) P[g][j] = from_f32<T>(0.f);                    // "P(-1)"
    const unsigned kfrag = (unsigned)(lq * KSTRIDE + lh * 16);                      // K fragment: + km*32*KSTRIDE + ks*32 (immediates)
    unsigned vfrag[DM];               // V^T fragment of (dm, 16-key group g) = vfrag[dm] ^ (g << 5): chunk (2 g + lh) ^ vswz(d)
#pragma unroll
    for (int dm = 0; dm < DM; ++dm) {
        const int d = dm * 32 + lq;
        vfrag[dm] = (unsigned)(d * 128 + ((lh ^ vswz(d)) << 4));
    }
    {   // S(0)
#pragma unroll
        for (int km = 0; km < 2; ++km)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const vec8 kf = as_vec8<T>(*(const u128*)(kring + kfrag + km * 32 * KSTRIDE + ks * 32));
                S[0][km] = Op16<T>::mfma(kf, qf[ks], ks == 0 ? zero16 : S[0][km]);
            }
    }
#ifdef BD_ATTN_PROBE
    unsigned probe_ts = 0;
#endif
    // one key tile; TI = t % 4 at compile time
#define PP_LOADF(dst, n)                                                                                         \
    {                                                                                                             \
        if ((n) < NPV) {                                                                                          \
            dst = as_vec8<T>(*(const u128*)(vring + ((TI + 3) % VBUF) * V_BYTES + (vfrag[(n) % DM] ^ (unsigned)(((n) / DM) << 5)))); \
        } else {                                                                                                  \
            const int ks_ = ((n) - NPV) / 2, km_ = ((n) - NPV) % 2;    /* alternate the two S accumulators */      \
            dst = as_vec8<T>(*(const u128*)(kring + ((TI + 1) % KBUF) * K_BYTES + kfrag + km_ * 32 * KSTRIDE + ks_ * 32)); \
        }                                                                                                         \
    }
#define PP_MFMA(n, f)                                                                                            \
    {                                                                                                             \
        if ((n) < NPV) {                                                                                          \
            oacc[(n) % DM] = Op16<T>::mfma(f, P[(n) / DM], oacc[(n) % DM]);                                       \
        } else {                                                                                                  \
            const int ks_ = ((n) - NPV) / 2, km_ = ((n) - NPV) % 2;                                               \
            S[(TI + 1) & 1][km_] = Op16<T>::mfma(f, qf[ks_], ks_ == 0 ? zero16 : S[(TI + 1) & 1][km_]);          \
        }                                                                                                         \
    }
#define PP_ITER(TI_)                                                                                             \
    {                                                                                                             \
        constexpr int TI = (TI_);                                                                                 \
        if (t >= 4 && t < 16) AP((t - 4) * 5)                                                                     \
        const bool stage = t + 2 < nt;                                                                            \
        if (stage) load_tile(t + 2);                                                                              \
        vec8 fr[4];                                                                                               \
        float tmax = -INFINITY, psum = 0.f, mneg = 0.f, alpha = 1.f;                                              \
        bool moved = false;                                                                                       \
        _Pragma("unroll") for (int n = 0; n < 3; ++n) PP_LOADF(fr[n & 3], n)                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        _Pragma("unroll") for (int n = 0; n < NSLOT; ++n) {                                                       \
            PP_MFMA(n, fr[n & 3])                                                                                 \
            if (n + 3 < NSLOT) PP_LOADF(fr[(n + 3) & 3], n + 3)                                                   \
            if (n < 4) {                        /* row max, 8 scores per slot */                                  \
                _Pragma("unroll") for (int e = n * 8; e < n * 8 + 8; ++e) tmax = fmaxf(tmax, S[TI & 1][e >> 4][e & 15]); \
                asm volatile("" : "+v"(tmax));   /* pin: IR passes sink side-effect-free code past sched_barrier */  \
            } else if (n == 4) {                /* the other half of the query's keys lives in lane ^ 32 */       \
                float other;      /* two DISTINCT registers: the builtin given one value twice swaps nothing */    \
                asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(tmax), "=&v"(other)); \
                tmax = fmaxf(tmax, other);                                                                        \
                const float m_new = fmaxf(m_run, tmax);                                                           \
                moved = !__all(m_new == m_run);                                                                   \
                alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);   /* 1 where the max did not move */        \
                m_run = m_new;                                                                                    \
                mneg = -m_new * sc;                                                                               \
            } else if (n < 21) {                /* 2 probabilities per slot: exp2, row sum, conversion */         \
                _Pragma("unroll") for (int e = (n - 5) * 2; e < (n - 5) * 2 + 2; ++e) {                           \
                    float pv = __builtin_amdgcn_exp2f(fmaf(S[TI & 1][e >> 4][e & 15], sc, mneg));                \
                    asm volatile("" : "+v"(pv));                                                                  \
                    psum += pv;                                                                                   \
                    P[e >> 3][e & 7] = from_f32<T>(pv);                                                           \
                }                                                                                                 \
            }                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
        }                                                                                                         \
        if (moved) {                    /* O and l follow the new max (PV(t-1) was added under the old one) */    \
            _Pragma("unroll") for (int i = 0; i < DM; ++i)                                                        \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;                               \
            l_run *= alpha;                                                                                       \
        }                                                                                                         \
        l_run += psum;                                                                                            \
        if (t >= 4 && t < 16) AP((t - 4) * 5 + 1)                                                                 \
        if (stage) store_tile(t + 2, (TI + 2) % KBUF, (TI + 2) % VBUF);                                           \
        if (t >= 4 && t < 16) AP((t - 4) * 5 + 2)                                                                 \
        __syncthreads();                                                                                          \
        if (t >= 4 && t < 16) AP((t - 4) * 5 + 3)                                                                 \
        ++t;                                                                                                      \
    }
    int t = 0;
    while (t + 4 <= nt) { PP_ITER(0) PP_ITER(1) PP_ITER(2) PP_ITER(3) }
    const int rem = nt - t;                 // t % 4 == 0 here
    if (rem > 0) PP_ITER(0)
    if (rem > 1) PP_ITER(1)
    if (rem > 2) PP_ITER(2)
#undef PP_ITER
#undef PP_LOADF
#undef PP_MFMA
#ifdef BD_ATTN_PROBE
    if (bd_attn_probe_buf && blockIdx.x < 512) bd_attn_probe_buf[((size_t)blockIdx.x * 8 + wid) * 64 + lane] = probe_ts;
#endif
    {   // PV(nt-1): V^T buffer (nt-1) % 4 -- a run-time index here, once per kernel
        const unsigned char* vcur = vring + ((nt - 1) % VBUF) * V_BYTES;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int dm = 0; dm < DM; ++dm) {
                const vec8 vf = as_vec8<T>(*(const u128*)(vcur + (vfrag[dm] ^ (unsigned)(g << 5))));
                oacc[dm] = Op16<T>::mfma(vf, P[g], oacc[dm]);
            }
    }

    // ---- finalise: O[q][d] = O^T / l ; lane (q, h) owns d = dm*32 + 8*(r>>2) + 4*h + (r&3)
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if constexpr (OUT_VIA_LDS) {
        // The lanes hold O^T (a query per lane, 4 consecutive d per register quad): written out directly that is 32 rows x 16
        // bytes per store instruction, ~10k cycles of a workgroup's ~100k (profiles/r2_attention.md).  Instead every wave lays its
        // 32 query rows out in LDS (the rings are dead after the barrier) and stores whole 16-byte pieces of full rows.
        __syncthreads();
        // the lane id is re-derived HERE from an opaque instruction: nothing of this block's addressing can then be hoisted above
        // the key loop, whose register budget is full (hoisted, it put a spill reload = a vector-memory load into every iteration)
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const int lq2 = ln & 31, lh2 = ln >> 5;
        unsigned char* const wl = lds + wid * (32 * OSTRIDE);
        unsigned char* const rowp = wl + lq2 * OSTRIDE;
#pragma unroll
        for (int dm = 0; dm < DM; ++dm)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float v4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v4[j] = oacc[dm][rq * 4 + j] * inv;
                const int d = dm * 32 + 8 * rq + 4 * lh2;
                if constexpr (OUTMODE == 3) f16c8_store4((f16c8*)rowp, HD, d, v4);
                else store_cvt<T, 4>((T*)rowp + d, v4);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int CPR = OROW / 16, HIC = HD / 8;           // 16-byte pieces per row image / of its hi plane
        static_assert((32 * CPR) % 64 == 0, "pieces per wave");
        const int64_t row0 = (int64_t)b * p.out_rows + (p.out_rows == q_len ? 0 : p.q_off) + q0;
        unsigned char* const out_hi = (unsigned char*)p.out;
        unsigned char* const out_lo = out_hi + 2 * p.out_plane;
#pragma unroll
        for (int k = 0; k < 32 * CPR / 64; ++k) {
            const int c = ln + 64 * k, r = c / CPR, col = c % CPR;
            if (q0 + r >= q_len) continue;
            const u128 piece = *(const u128*)(wl + r * OSTRIDE + col * 16);
            const int64_t e = (row0 + r) * (heads * HD) + head * HD;
            if (OUTMODE == 3 && col >= HIC) *(u128*)(out_lo + e + (col - HIC) * 16) = piece;
            else *(u128*)(out_hi + 2 * e + col * 16) = piece;
        }
        return;
    }
    const int q = q0 + lq;
    if (q < q_len) {
        const int64_t e0 = ((int64_t)b * p.out_rows + (p.out_rows == q_len ? 0 : p.q_off) + q) * (heads * HD) + head * HD;
#pragma unroll
        for (int dm = 0; dm < DM; ++dm)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float v4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v4[j] = oacc[dm][rq * 4 + j] * inv;
                const int64_t e = e0 + dm * 32 + 8 * rq + 4 * lh;
                if constexpr (OUTMODE == 2) {
                    store_cvt<fp8e4, 4>((fp8e4*)p.out + e, v4);
                } else if constexpr (OUTMODE == 3) {
                    f16c8_store4((f16c8*)p.out, p.out_plane, e, v4);
                } else if constexpr (OUTMODE == 1 || OUTMODE == 4) {
                    typedef typename std::conditional<OUTMODE == 1, __bf16, _Float16>::type OT;
                    float hi4[4], lo4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) split_hi_lo<OT>(v4[j], hi4[j], lo4[j]);
                    store_cvt<OT, 4>((OT*)p.out + e, hi4);
                    store_cvt<OT, 4>((OT*)p.out + p.out_plane + e, lo4);
                } else {
                    store_cvt<T, 4>((T*)p.out + e, v4);
                }
            }
    }
}

template <class T, int HD, int OUTMODE> int launch_pp(const AttnArgs& a, hipStream_t s) {
    const int nqb = (a.q_len + 255) / 256;
    const int slot = bd_trace_open(s, 1, a.batch * a.heads, a.seq, HD);
    hipLaunchKernelGGL((attn_kernel_pp<T, HD, OUTMODE>), dim3(nqb * a.heads * a.batch), dim3(512), 0, s, a);
    bd_trace_close(s, slot);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

template <class T, int NS, int HD, int NW, int OUTMODE = 0> int launch(const AttnArgs& a, hipStream_t s) {
    const int nqb = (a.q_len + NW * 32 - 1) / (NW * 32);
    const int slot = bd_trace_open(s, 1, a.batch * a.heads, a.seq, HD);
    hipLaunchKernelGGL((attn_kernel<T, NS, HD, NW, OUTMODE>), dim3(nqb * a.heads * a.batch), dim3(NW * 64), 0, s, a);
    bd_trace_close(s, slot);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

// (file-local) opt-in latency forms of the call being dispatched: set by bd_attention_q_forms around its dispatch, on the calling thread
static thread_local bool g_latency_forms = false;

template <class T, int NS, int OUTMODE = 0> int dispatch(const AttnArgs& a, int head_dim, hipStream_t s) {
    // 3 waves (96-query blocks) when that tiles the sequence with less waste (DINOv2: 261 -> 3 x 96)
    const int waste4 = ((a.q_len + 127) / 128) * 128 - a.q_len, waste3 = ((a.q_len + 95) / 96) * 96 - a.q_len;
    const bool use3 = waste3 < waste4;
    if constexpr (NS == 1) {
        // ping-pong kernel where its 256-query blocks tile the query range without waste (BETR: 1536 = 6 x 256; last block: 256)
        // (latency forms, round 6: one pose at a time the 256-query blocks are 48 workgroups on 256 CUs, 36 us; the 128-query kernel's 96
        // take 31 us -- and lose from two poses on: profiles/r6_attn_b1_probe.txt)
        const bool few = g_latency_forms && 4 * ((a.q_len + 255) / 256) * a.heads * a.batch <= 256;      // (a quarter of MI355X's 256 CUs)
        if (head_dim == 96 && a.q_len % 256 == 0 && a.seq % KT == 0 && !few) return launch_pp<T, 96, OUTMODE>(a, s);      // (whole key tiles only: the kernel has no tail mask)
    }
    if (head_dim == 96) return use3 ? launch<T, NS, 96, 3, OUTMODE>(a, s) : launch<T, NS, 96, 4, OUTMODE>(a, s);
    if (head_dim == 64) return use3 ? launch<T, NS, 64, 3, OUTMODE>(a, s) : launch<T, NS, 64, 4, OUTMODE>(a, s);
    return BD_ERR_SHAPE;
}

// the K / V buffer descriptors span ONE sample's rows with a 32-bit byte count (and 32-bit per-lane offsets inside it)
inline bool sample_bytes_out_of_range(int seq, int heads, int head_dim) {
    return head_dim <= 0 || (int64_t)seq * 3 * heads * head_dim * 2 >= ((int64_t)1 << 31);
}

}  // namespace

int bd_attention_q_forms(const void* qkv, int64_t qkv_plane, void* out, int64_t out_plane, int batch, int seq, int heads, int head_dim,
                         float scale, const int32_t* q_view, int q_len, int prec, int latency_forms, void* stream) {
    g_latency_forms = latency_forms != 0;
    const int rc = bd_attention_q(qkv, qkv_plane, out, out_plane, batch, seq, heads, head_dim, scale, q_view, q_len, prec, stream);
    g_latency_forms = false;
    return rc;
}

extern "C" int bd_attention_q(const void* qkv, int64_t qkv_plane, void* out, int64_t out_plane, int batch, int seq,
                              int heads, int head_dim, float scale, const int32_t* q_view, int q_len, int prec,
                              void* stream) {
    if (!qkv || !out) return BD_ERR_NULL;
    if (batch <= 0 || seq <= 0 || heads <= 0) return BD_ERR_SHAPE;
    if (q_view ? (q_len <= 0 || q_len > seq || seq % q_len) : (q_len != seq)) return BD_ERR_SHAPE;
    if (sample_bytes_out_of_range(seq, heads, head_dim)) return BD_ERR_SHAPE;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) return BD_ERR_ALIGN;
    AttnArgs a{qkv, qkv_plane, out, out_plane, batch, seq, heads, scale * 1.4426950408889634f, q_view, q_len, 0, q_len};
    hipStream_t s = (hipStream_t)stream;
    switch (prec) {
        case BD_PREC_BF16: return dispatch<__bf16, 1>(a, head_dim, s);
        case BD_PREC_F16: return dispatch<_Float16, 1>(a, head_dim, s);
        case BD_PREC_BF16X3: return dispatch<__bf16, 2>(a, head_dim, s);
        case BD_PREC_F16_OUT_BF16X3: return dispatch<_Float16, 1, 1>(a, head_dim, s);
        case BD_PREC_BF16_OUT_FP8: return dispatch<__bf16, 1, 2>(a, head_dim, s);
        case BD_PREC_F16_OUT_F16C8: return dispatch<_Float16, 1, 3>(a, head_dim, s);
        case BD_PREC_BF16X3_OUT_F16C8: return dispatch<__bf16, 2, 3>(a, head_dim, s);
        case BD_PREC_F16_OUT_F16X3: return dispatch<_Float16, 1, 4>(a, head_dim, s);
        case BD_PREC_BF16X3_OUT_F16X3: return dispatch<__bf16, 2, 4>(a, head_dim, s);
        default: return BD_ERR_DTYPE;
    }
}

// prec -> (T, NS, OUTMODE) for both the tiled and the prefix kernel
#define BD_ATTN_PREC_SWITCH(CALL)                                                       \
    switch (prec) {                                                                     \
        case BD_PREC_BF16: return CALL(__bf16, 1, 0);                                   \
        case BD_PREC_F16: return CALL(_Float16, 1, 0);                                  \
        case BD_PREC_BF16X3: return CALL(__bf16, 2, 0);                                 \
        case BD_PREC_F16_OUT_BF16X3: return CALL(_Float16, 1, 1);                       \
        case BD_PREC_BF16_OUT_FP8: return CALL(__bf16, 1, 2);                           \
        case BD_PREC_F16_OUT_F16C8: return CALL(_Float16, 1, 3);                        \
        case BD_PREC_BF16X3_OUT_F16C8: return CALL(__bf16, 2, 3);                       \
        case BD_PREC_F16_OUT_F16X3: return CALL(_Float16, 1, 4);                        \
        case BD_PREC_BF16X3_OUT_F16X3: return CALL(__bf16, 2, 4);                       \
        default: return BD_ERR_DTYPE;                                                   \
    }

extern "C" int bd_attention_prefix(const void* qkv, int64_t qkv_plane, void* out, int64_t out_plane, int batch, int seq, int heads,
                                   int head_dim, float scale, int n_prefix, int prefix_queries, int prec, void* stream) {
    if (!qkv || !out) return BD_ERR_NULL;
    if (batch <= 0 || seq <= 0 || heads <= 0 || n_prefix <= 0 || n_prefix >= seq) return BD_ERR_SHAPE;
    if (sample_bytes_out_of_range(seq, heads, head_dim)) return BD_ERR_SHAPE;
    // With the prefix queries wanted, one launch over all seq queries IS the fastest form measured (profiles/r4_attention.md: a separate
    // launch for the prefix rows re-reads the pair's K / V -- 154 MB per DINOv2 launch -- and costs more than the ninth query wave it
    // saves); without them, the patch queries alone tile exactly.
    if (prefix_queries) return bd_attention_q(qkv, qkv_plane, out, out_plane, batch, seq, heads, head_dim, scale, nullptr, seq, prec, stream);
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) return BD_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    // the seq - n_prefix queries behind the prefix, all keys; result rows [n_prefix, seq) of every sample, the others untouched
    AttnArgs a{qkv, qkv_plane, out, out_plane, batch, seq, heads, scale * 1.4426950408889634f, nullptr, seq - n_prefix, n_prefix, seq};
#define BD_CALL_MAIN(T_, NS_, OM_) dispatch<T_, NS_, OM_>(a, head_dim, s)
    BD_ATTN_PREC_SWITCH(BD_CALL_MAIN)
#undef BD_CALL_MAIN
}

extern "C" int bd_attention(const void* qkv, int64_t qkv_plane, void* out, int64_t out_plane, int batch,
                            int seq, int heads, int head_dim, float scale, int prec, void* stream) {
    return bd_attention_q(qkv, qkv_plane, out, out_plane, batch, seq, heads, head_dim, scale, nullptr, seq, prec, stream);
}
