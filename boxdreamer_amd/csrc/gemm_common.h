// Shared pieces of the GEMM kernels (gemm.hip: the 16-bit, split and e4m3 operand classes; gemm_f16c8.hip: the F16C8 class): tile
// coordinates, LDS-DMA wrappers, the accumulator-start convention, every epilogue and the launch-geometry predicates.
// Header-only; included by exactly those two translation units.
#pragma once
#include "bd_common.h"
#include <type_traits>

#ifdef BD_GEMM_PROBE
// Measurement build only (tools/gemm_phase_probe.py; never part of libboxdreamer_hip.so): per-wave shader-clock stamps of the
// mainloop phases, kept in the lanes of one VGPR (lane i = stamp i) and written out once at kernel end.  One buffer pointer per
// translation unit (no relocatable device code): bd_gemm_probe_set / bd_gemm_f16c8_probe_set.
static __device__ unsigned* bd_probe_buf = nullptr;
#define BD_PROBE(idx) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if ((idx) < 64) probe_ts = (lane == (idx)) ? (unsigned)t__ : probe_ts; }
#define BD_PROBE_IF(c, idx) { if (c) BD_PROBE(idx) }
#define BD_PROBE_RT(idx) { const unsigned long long t__ = __builtin_amdgcn_s_memrealtime(); probe_ts = (lane == (idx)) ? (unsigned)t__ : probe_ts; }
#else
#define BD_PROBE(idx)
#define BD_PROBE_IF(c, idx)
#define BD_PROBE_RT(idx)
#endif


namespace {


// chunk swizzle for a tile whose rows hold CH 16-byte chunks: rows that share a 256-byte LDS bank row are separated
template <int CH> __device__ __forceinline__ int swz_chunk(int row, int c) {
    constexpr int RPB = 16 / CH;          // tile rows per 256-byte LDS bank row
    return c ^ ((row / RPB) & (CH - 1));
}

// Workgroup -> output tile.  (1) XCD-aware bijective remap: XCD x (= blockIdx % 8, private 4 MiB L2) owns a contiguous
// run of logical ids.  (2) Grouped raster inside that run: ids walk group_m M-tiles down, then step one N-tile across,
// so the workgroups resident on an XCD cover a compact patch whose A row-panels and W tiles stay L2-resident.
template <int BM_, int BN_>
__device__ __forceinline__ void tile_coords_t(int M, int N, int group_m, int& m0, int& n0) {
    const int tilesM = (M + BM_ - 1) / BM_, tilesN = (N + BN_ - 1) / BN_;
    const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int per_group = group_m * tilesN;
    const int g = wg / per_group, in_g = wg % per_group;
    const int gm0 = g * group_m;
    const int gh = (tilesM - gm0) < group_m ? (tilesM - gm0) : group_m;
    m0 = (gm0 + in_g % gh) * BM_;
    n0 = (in_g / gh) * BN_;
}

// LDS-DMA from inline asm.  With the builtin, hipcc's waitcnt pass sees an LDS write it cannot place: it then (a) puts
// s_waitcnt vmcnt(0) in front of the first ds_read that follows a pending DMA in the same block (the prefetch latency is
// exposed every slab) and (b) degrades every ds_read wait to lgkmcnt(0) (no counted waits, so the fragment reads of the
// next k-step cannot stay in flight under the MFMAs of this one).  Hidden from the compiler the prefetch flies under the
// slab's MFMAs, fragment waits are counted, and slab_barrier() does the one wait that is really needed.
// (M0 has no other user in these kernels: gfx9+ DS ops do not read it.)
__device__ __forceinline__ void glds16(const unsigned char* g, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_off) : "memory");
}

__device__ __forceinline__ void slab_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces (and earlier stores) have landed
    __builtin_amdgcn_s_barrier();                         // ... and everyone else's; the other buffer is free
    asm volatile("" ::: "memory");
}
// LDS-DMA with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: the producer wave of
// gemm_kernel_pc keeps one offset VGPR per piece and advances K on the scalar side.
__device__ __forceinline__ void glds16_s(unsigned voff, const unsigned char* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}
// workgroup barrier without any counter wait of its own (consumers have nothing outstanding that matters; the producer waits
// for its DMA explicitly): a raw s_barrier fenced against compiler motion of LDS accesses
__device__ __forceinline__ void pc_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ unsigned lds_offset_of(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p);
}

// out_f32 codes
enum { OUT_OPERAND = 0, OUT_F32 = 1, OUT_F16 = 2, OUT_BF16 = 3, OUT_BF16X2 = 4, OUT_F16X2 = 5 };

// Accumulator start values and where the epilogue terms enter -- ONE convention for every kernel, so that a row's result
// does not depend on the tile shape that computed it (the property tests compare a sample run alone with the same sample
// inside a batch, bit for bit):
//   * a plain fp32 residual (identity row map) is loaded INTO the accumulators (divided by the column's weight scale in the e4m3
//     class, whose epilogue multiplies by it again) before the first
//     MFMA (C fragment layout: 2 rows x 128 contiguous bytes per load instruction), so no epilogue reads global memory for
//     it -- on gfx9 loads and stores share the in-order vmcnt, and an epilogue that loads after it has stored waits for its
//     own stores to be acknowledged (measured: 13.8k cycles per 256x192 tile for a LONE workgroup, profiles/r2_gemm_epilogue.md);
//   * everything else starts at zero;
//   * scale * acc + bias is applied when the accumulators leave the registers (per-column values, one register per 32-column
//     tile), then the activation, the table rows and a remapped / scaled-mode residual.
__device__ __forceinline__ bool resid_in_acc(const bd_gemm_args& p) { return p.resid && p.rpg_in <= 0; }

template <int MI, int NI>
__device__ __forceinline__ void acc_init(const bd_gemm_args& p, f32x16 (&acc)[MI][NI], int wm0, int wn0, int lane) {
    if (resid_in_acc(p)) {
        // rows / columns past the edge are clamped: those accumulators are never stored
        const int lrow = lane & 31, lhalf = lane >> 5;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int gr = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                gr = gr < p.M ? gr : p.M - 1;
                const float* rp = p.resid + (int64_t)gr * p.ldr;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    int gc = wn0 + j * 32 + lrow;
                    gc = gc < p.N ? gc : p.N - 1;
                    acc[i][j][r] = p.wscale ? rp[gc] / p.wscale[gc] : rp[gc];
                }
            }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
}

// ---- scalar fallback epilogue (N not a multiple of 8 / unaligned pointers): C fragment = (col = lane & 31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5)); operand-dtype (16-bit) or fp32 outputs only
template <class T, int NS, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const bd_gemm_args& p, f32x16 (&acc)[MI][NI], int wm0, int wn0, int lane) {
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int M = p.M, N = p.N;
    const float* bias = p.bias;
    const float* resid = resid_in_acc(p) ? nullptr : p.resid;      // else already in the accumulators (acc_init)
    const float* addtab = p.addtab;
    const float* wscale = p.wscale;
    const int act = p.act, out_f32 = p.out_f32, rpg_in = p.rpg_in, rpg_out = p.rpg_out, row_off = p.row_off;
    const int tab_rows = p.tab_rows;
    const int64_t ldr = p.ldr, ldo = p.ldo, out_plane = p.out_plane;
    float bj[NI], sj[NI];
    int gcs[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        gcs[j] = wn0 + j * 32 + lrow;
        bj[j] = (bias && gcs[j] < N) ? bias[gcs[j]] : 0.f;
        sj[j] = (wscale && gcs[j] < N) ? wscale[gcs[j]] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
            if (gr < M) {
                int64_t orow = gr;
                if (rpg_in > 0) orow = (int64_t)(gr / rpg_in) * rpg_out + gr % rpg_in + row_off;
                const float* tab = addtab ? addtab + (int64_t)(gr % tab_rows) * N : nullptr;
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int gc = gcs[j];
                    if (gc < N) {
                        float v = fmaf(acc[i][j][r], sj[j], bj[j]);
                        if (act == BD_ACT_GELU) v = gelu_erf(v);
                        if (tab) v += tab[gc];
                        if (resid) v += resid[orow * ldr + gc];
                        if (out_f32 == OUT_F32) {
                            ((float*)p.out)[orow * ldo + gc] = v;
                        } else if constexpr (sizeof(T) == 2) {
                            T* o = (T*)p.out + orow * ldo + gc;
                            if (NS == 2) {
                                float h, l;
                                split_hi_lo<T>(v, h, l);
                                o[0] = from_f32<T>(h);
                                o[out_plane] = from_f32<T>(l);
                            } else {
                                o[0] = from_f32<T>(v);
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---- wide epilogue: accumulators -> this wave's private LDS scratch -> row-contiguous 16-byte accesses.
// The MFMA C fragment gives each lane one column and 16 scattered rows (64 narrow stores per lane, each half-wave
// touching half a cache line): measured store-issue- and load-latency-bound (~27 us per tile round vs 1.3 us per
// K-slab).  Here every 32-row chunk of the wave tile is written to LDS with conflict-free ds_write_b32 (one row per
// half-wave), read back with ds_read_b128 as 4 (fp32 out) or 8 (narrow out) consecutive columns per lane, the residual
// / table rows are fetched as 16-byte vectors in batches, and the result leaves as full-line stores.  Same-wave LDS
// traffic is ordered, so no workgroup barrier is needed between chunks.
// SR: rows of the wave tile staged per pass through the scratch (32 = one MFMA row chunk; 16 = half of it, for kernels whose
// scratch must fit a smaller LDS region: registers r with (r >> 2) in {2 hc, 2 hc + 1} are exactly rows 16 hc .. 16 hc + 15).
// LEAN: register-frugal form for kernels capped at 168 VGPRs (gemm_kernel_pc: three waves on one SIMD): per-column bias /
// scale vectors are re-loaded per column block instead of kept live across the whole tile, and the fp32 path batches its
// global reads two passes at a time instead of four.
template <class T, int NS, int MI, int NI, int SR = 32, bool LEAN = false>
__device__ __forceinline__ void gemm_epilogue_lds(const bd_gemm_args& p, f32x16 (&acc)[MI][NI], unsigned char* scratch,
                                                  int wm0, int wn0, int lane_) {
    constexpr int COLS = NI * 32;                      // wave-tile width (fp32 words per scratch row)
    int lane = lane_;
    if constexpr (LEAN) {     // persistent kernels: keep the epilogue's lane-dependent addressing out of the K loop's live ranges (pc_epilogue)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    }
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int M = p.M, N = p.N;
    const float* resid = resid_in_acc(p) ? nullptr : p.resid;      // else already in the accumulators (acc_init)
    const float* addtab = p.addtab;
    const int act = p.act, rpg_in = p.rpg_in, rpg_out = p.rpg_out, row_off = p.row_off, tab_rows = p.tab_rows;
    const int64_t ldr = p.ldr, ldo = p.ldo, out_plane = p.out_plane;
    float* sc = (float*)scratch;
    // scale * acc + bias on the way INTO the scratch: one column per lane and 32-column tile
    float bj[NI], sj[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int gc = wn0 + j * 32 + lrow;
        bj[j] = (p.bias && gc < N) ? p.bias[gc] : 0.f;
        sj[j] = (p.wscale && gc < N) ? p.wscale[gc] : 1.f;
    }
    auto to_scratch = [&](int i, int hc) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = hc * (SR / 2); r < (hc + 1) * (SR / 2); ++r)
                sc[((r & 3) + 8 * ((r >> 2) - hc * (SR / 8)) + 4 * lhalf) * COLS + j * 32 + lrow] = fmaf(acc[i][j][r], sj[j], bj[j]);
    };
    // the wave tile is flushed in column blocks of CW columns (all of it when it is 32 or 64 wide; 3 x 32 for the 96-wide
    // wave tile of the 256 x 192 workgroup tile) so that the lanes of a pass always cover whole rows of a block
    constexpr int CW = (NI % 2 == 0) ? 64 : 32, NCB = COLS / CW;
    if (p.out_f32 == OUT_F32) {
        constexpr int LPR = CW / 4;                    // lanes per row (4 floats each)
        constexpr int RPI = 64 / LPR;                  // rows per pass
        constexpr int PASSES = SR / RPI;
        const int c4 = lane % LPR, rsub = lane / LPR;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ih = 0; ih < MI * (32 / SR); ++ih) {
            const int i = ih / (32 / SR), hc = ih % (32 / SR);
            to_scratch(i, hc);
            // global reads are issued in batches of PB passes (register budget); native vector types only -- HIP's
            // float4 struct in a local array lands in scratch
            constexpr int PB = LEAN ? (PASSES > 2 ? 2 : PASSES) : (PASSES > 4 ? 4 : PASSES);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int gc = wn0 + cb * CW + c4 * 4;
                const bool cok = gc < N;
#pragma unroll
                for (int t0 = 0; t0 < PASSES; t0 += PB) {
                    f32x4 rv[PB], tv[PB];
                    int64_t orow[PB];
                    bool ok[PB];
#pragma unroll
                    for (int u = 0; u < PB; ++u) {
                        const int gr = wm0 + i * 32 + hc * SR + (t0 + u) * RPI + rsub;
                        ok[u] = cok && gr < M;
                        const int grc = gr < M ? gr : M - 1;
                        orow[u] = rpg_in > 0 ? (int64_t)(grc / rpg_in) * rpg_out + grc % rpg_in + row_off : (int64_t)grc;
                        rv[u] = zero4;
                        tv[u] = zero4;
                        if (resid && ok[u]) rv[u] = *(const f32x4*)(resid + orow[u] * ldr + gc);
                        if (addtab && ok[u]) tv[u] = *(const f32x4*)(addtab + (int64_t)(grc % tab_rows) * N + gc);
                    }
#pragma unroll
                    for (int u = 0; u < PB; ++u) {
                        f32x4 v = *(const f32x4*)(sc + ((t0 + u) * RPI + rsub) * COLS + cb * CW + c4 * 4);
                        if (act == BD_ACT_GELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                        }
                        v = v + tv[u] + rv[u];
                        if (ok[u]) *(f32x4*)((float*)p.out + orow[u] * ldo + gc) = v;
                    }
                }
            }
        }
    } else {
        constexpr int LPR = CW / 8;                    // lanes per row (8 output columns each)
        constexpr int RPI = 64 / LPR;
        constexpr int PASSES = SR / RPI;
        const int c8 = lane % LPR, rsub = lane / LPR;
        const int out_mode = p.out_f32;
        auto store8 = [&](int64_t orow, int gc, const float (&v)[8]) {
            if (out_mode == OUT_F16) {            // f16 single plane (optional f16 attention of the strict mode)
                store_cvt<_Float16, 8>((_Float16*)p.out + orow * ldo + gc, v);
            } else if (out_mode == OUT_BF16) {    // bf16 single plane (fp8 mode: attention operands stay bf16)
                store_cvt<__bf16, 8>((__bf16*)p.out + orow * ldo + gc, v);
            } else if (out_mode == OUT_BF16X2) {  // split-bf16 planes (F16C8 mode: DINOv2's split-bf16 attention)
                store_operand8<__bf16, 2>((__bf16*)p.out, out_plane, orow * ldo + gc, v);
            } else if (out_mode == OUT_F16X2) {   // split-f16 planes (an F16C8 Linear feeding a promoted, split-f16 one)
                store_operand8<_Float16, 2>((_Float16*)p.out, out_plane, orow * ldo + gc, v);
            } else {
                store_operand8<T, NS>((T*)p.out, out_plane, orow * ldo + gc, v);
            }
        };
        if constexpr (LEAN && NI == 3 && PASSES == 1) {
            if (p.rms_wq) {
                // Fused q/k RMSNorm: the 96-column wave tile IS one head (host-checked).  A lane owns 3 x 8 columns of one row per
                // 16-row pass and the 4 lanes of a row (one quad) combine their sums of squares: fp32 mean / rsqrt on the
                // accumulators themselves, then the learned weight, then the 16-bit store.  Which third of the output this
                // wave tile lies in (q: normalise with wq, k: with wk, v: untouched) is wave-uniform.
                const int part = wn0 / (N / (p.rms_parts == 2 ? 2 : 3));
                const float* rw = part == 0 ? p.rms_wq : (part == 1 ? p.rms_wk : nullptr);
                float wv[3][8];
#pragma unroll
                for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                    for (int e = 0; e < 8; ++e) wv[cb][e] = rw ? rw[cb * 32 + c8 * 8 + e] : 1.f;
                const float eps = p.rms_eps;
#pragma unroll
                for (int ih = 0; ih < MI * (32 / SR); ++ih) {
                    const int i = ih / (32 / SR), hc = ih % (32 / SR);
                    to_scratch(i, hc);
                    const int gr = wm0 + i * 32 + hc * SR + rsub;
                    float v[3][8];
                    float ss = 0.f;
#pragma unroll
                    for (int cb = 0; cb < 3; ++cb) {
                        const float* src = sc + rsub * COLS + cb * CW + c8 * 8;
                        const f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[cb][e] = a0[e]; v[cb][4 + e] = a1[e]; }
#pragma unroll
                        for (int e = 0; e < 8; ++e) ss = fmaf(v[cb][e], v[cb][e], ss);
                    }
                    ss += __shfl_xor(ss, 1);
                    ss += __shfl_xor(ss, 2);
                    const float inv = rw ? rsqrtf(ss * (1.0f / 96.0f) + eps) : 1.f;
                    if (gr < M) {
#pragma unroll
                        for (int cb = 0; cb < 3; ++cb) {
                            float o8[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) o8[e] = wv[cb][e] * (v[cb][e] * inv);     // w * (x * rsqrt(..)): blocks.py:51-56
                            store8((int64_t)gr, wn0 + cb * CW + c8 * 8, o8);
                        }
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int ih = 0; ih < MI * (32 / SR); ++ih) {
            const int i = ih / (32 / SR), hc = ih % (32 / SR);
            to_scratch(i, hc);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int gc = wn0 + cb * CW + c8 * 8;
#pragma unroll
                for (int t = 0; t < PASSES; ++t) {
                    const int gr = wm0 + i * 32 + hc * SR + t * RPI + rsub;
                    const float* src = sc + (t * RPI + rsub) * COLS + cb * CW + c8 * 8;
                    const f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = a0[e]; v[4 + e] = a1[e]; }
                    if (act == BD_ACT_GELU) gelu_n<GeluKind<T, NS>::value, 8>(v);   // 16/8-bit result: fitted forms (bd_common.h)
                    if (gc < N && gr < M) {
                        const int64_t orow = rpg_in > 0 ? (int64_t)(gr / rpg_in) * rpg_out + gr % rpg_in + row_off : (int64_t)gr;
                        if (addtab) {
                            const float* tp = addtab + (int64_t)(gr % tab_rows) * N + gc;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += tp[e];
                        }
                        if (resid) {
                            const float* rp = resid + orow * ldr + gc;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += rp[e];
                        }
                        store8(orow, gc, v);
                    }
                }
            }
        }
    }
}

// quad all-reduce (lanes 4k .. 4k+3) on the VALU: two DPP quad_perm adds
__device__ __forceinline__ float quad_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // [2,3,0,1]
    return x;
}

// the value of the neighbouring lane (lane ^ 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ unsigned dpp_xor1(unsigned x) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);
}
// all-reduce over 8 consecutive lanes (8k .. 8k+7): the quad sums, then the other quad of the half-row (DPP row_half_mirror: lane i <- 7 - i)
__device__ __forceinline__ float oct_sum(float x) {
    x = quad_sum(x);
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
    return x;
}

// Epilogue of gemm_kernel_pc for the 64 x 96 wave tile (MI = 2, NI = 3), 16 rows per pass through this wave's 6-KiB scratch.
//   colp: this tile's per-column vectors in LDS (bias at [0, 256), weight scale at [256, 512)), indexed by tile column
//   rmsw: q weights at [0, 96), k weights at [256, 352)
//   wcol: first column of the wave tile inside the workgroup tile;  (wm0, wn0): its global origin
//   next: the wave tile origin of this workgroup's next tile (EP 3 / 4 pre-load its residual), has_next = there is one
// LayerNorm fold (ABI 8, bd_gemm_args.ln_*):
//   EP 4 = EP 3 + the producer side: the registers that just left as an fp32 row piece (8 lanes x 12 values = the row's 96 columns) give
//          (mean, M2) -- two 8-lane reductions -- stored by one lane at ln_stats_out[row][wave tile], and leave once more as the F16C8
//          operand copy of the row (ln_op_out);
//   LNF  = the consumer side: rowp = this tile's row statistics in LDS, (rstd, -mean rstd) per tile row (the producer waves combined the
//          row's partial pairs, ln_rows_combine below), colp[256 ..] = the column sums s[n]; applied where the accumulators leave the
//          registers: rstd * acc + (-mean rstd) * s[n] + bias[n].  wrow: first row of the wave tile inside the workgroup tile.
template <class T, int NS, int EP, int OUTK, bool GELU, int MI = 2, bool LNF = false>
__device__ __forceinline__ void pc_epilogue(const bd_gemm_args& p, f32x16 (&acc)[MI][3], float* sc, float* sc_hi, const float* colp, const float* colp_next,
                                            const float* rmsw, int wcol, int wm0, int wn0, int lane_, bool has_next, int nwm0, int nwn0,
                                            const float* rowp = nullptr, int wrow = 0) {
    static_assert(!LNF || (sizeof(T) == 2 && EP != 3 && EP != 4), "the LayerNorm fold's consumer side: 16-bit results of the 16-bit / F16C8 classes");
    static_assert((EP != 4 && EP != 5) || std::is_same<T, f16c8>::value, "the LayerNorm fold's producer side emits the F16C8 operand class");
    constexpr int COLS = 96;
    // The lane id is re-derived HERE from an opaque instruction pair, so that none of the epilogue's lane-dependent addressing can be
    // hoisted above the K loop, whose register budget (168) is full: hoisted, three of those values were spilled in the F16C8 / e4m3
    // instances and RELOADED inside the epilogue -- a vector-memory load whose s_waitcnt vmcnt(0) also waited for the 24 residual
    // pre-loads of the chunk before the first row could be stored (round 4; the build now fails on a spill, boxdreamer_amd/build.py).
    (void)lane_;
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int M = p.M;
    float bj[3], sj[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        bj[j] = p.bias ? colp[wcol + j * 32 + lrow] : 0.f;
        sj[j] = 1.f;
        if constexpr (sizeof(T) == 1) sj[j] = p.wscale ? colp[256 + wcol + j * 32 + lrow] : 1.f;
        if constexpr (LNF) sj[j] = colp[256 + wcol + j * 32 + lrow];          // the column sums s[n] (their slot is the e4m3 class's scale slot)
    }
    // LNF: (rstd, -mean rstd) of a chunk's eight register rows (the same address for the 32 lanes of a half-wave: LDS broadcasts), read as a
    // block BEFORE the chunk's writes: interleaved, every read would wait for the writes in front of it (LDS operations return in order and
    // the compiler cannot tell the two regions apart) -- measured ~3000 cycles per tile
    float2 lnrs[LNF ? 8 : 1];
    auto load_row_stats = [&](int i, int hc) {
        if constexpr (LNF) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = hc * 8 + q;
                lnrs[q] = *(const float2*)(rowp + 2 * (wrow + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf));
            }
        }
    };
    load_row_stats(0, 0);
    auto to_scratch = [&](int i, int hc) {          // rows 16 hc .. 16 hc + 15 of 32-row block i: registers r with (r >> 2) in {2 hc, 2 hc + 1}
        if constexpr (LNF) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = hc * 8 + q;
                float* const dst = ((r >> 2) - hc * 2) == 0 ? sc : sc_hi;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    dst[((r & 3) + 4 * lhalf) * COLS + j * 32 + lrow] = fmaf(acc[i][j][r], lnrs[q].x, fmaf(lnrs[q].y, sj[j], bj[j]));
            }
            // the NEXT chunk's row statistics into the registers this chunk just released: they arrive under this chunk's read-back and
            // stores (only the first chunk's reads, issued before the loop, are waited for)
            if (i * 2 + hc + 1 < 2 * MI) load_row_stats((i * 2 + hc + 1) >> 1, (i * 2 + hc + 1) & 1);
            return;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = hc * 8; r < hc * 8 + 8; ++r) {
                const float a = acc[i][j][r];
                // rows 0-7 of the 16-row chunk live at sc, rows 8-15 at sc_hi (one contiguous block, or two regions of the operand ring)
                float* const dst = ((r >> 2) - hc * 2) == 0 ? sc : sc_hi;
                dst[((r & 3) + 4 * lhalf) * COLS + j * 32 + lrow] = sizeof(T) == 1 ? fmaf(a, sj[j], bj[j]) : a + bj[j];
            }
    };
    if constexpr (EP == 3 || EP == 4) {
        // fp32 rows: 8 lanes x 16 bytes = one 128-byte line per row and 32-column block; 8 rows per pass, 2 passes per chunk
        const int c4 = lane & 7, rsub = lane >> 3;
        const bool pre = has_next && p.resid != nullptr;
        float rsn[3] = {1.f, 1.f, 1.f};       // e4m3 class: the NEXT tile's column scales (its side-buffer slot landed with its first slab)
        if constexpr (sizeof(T) == 1) {
            if (pre && p.wscale) {
#pragma unroll
                for (int j = 0; j < 3; ++j) rsn[j] = colp_next[256 + wcol + j * 32 + lrow];
            }
        }
        float keep_mean = 0.f, keep_m2 = 0.f;            // EP 4: the (mean, M2) pair of pass c4 of this lane's row group
#pragma unroll
        for (int ih = 0; ih < 2 * MI; ++ih) {
            const int i = ih >> 1, hc = ih & 1;
            to_scratch(i, hc);
            // these accumulator registers are free now: the next tile's residual (or zero) goes in
            if (pre) {
#pragma unroll
                for (int r = hc * 8; r < hc * 8 + 8; ++r) {
                    int gr = nwm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
                    gr = gr < M ? gr : M - 1;
                    const float* rp = p.resid + (int64_t)gr * p.ldr + nwn0 + lrow;
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i][j][r] = (sizeof(T) == 1 && p.wscale) ? rp[j * 32] / rsn[j] : rp[j * 32];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int r = hc * 8; r < hc * 8 + 8; ++r) acc[i][j][r] = 0.f;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int gr = wm0 + i * 32 + hc * 16 + t * 8 + rsub;
                f32x4 v[3];
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) v[cb] = *(const f32x4*)((t == 0 ? sc : sc_hi) + rsub * COLS + cb * 32 + c4 * 4);
                if (gr < M) {
                    float* op = (float*)p.out + (int64_t)gr * p.ldo + wn0 + c4 * 4;
#pragma unroll
                    for (int cb = 0; cb < 3; ++cb) {
                        *(f32x4*)(op + cb * 32) = v[cb];
                    }
                }
                if constexpr (EP == 4) {
                    // LayerNorm fold, producer side, on the registers that just left as fp32: the 8 lanes of a row hold its 96 values (3 x 4
                    // each).  (mean, M2) by two 8-lane reductions on the VALU (DPP); lane c4 == pass keeps the pair, so that all eight passes
                    // of the tile leave in ONE full-wave store at the end.  The F16C8 operand copy leaves in 16-byte (f16 plane) / 8-byte (lo8
                    // plane) pieces: neighbouring lanes swap halves (DPP) so that the even lane of a pair owns 8 consecutive columns of block
                    // 0 and the odd lane 8 of block 1; block 2's halves both go to the even lane.
                    // (4- and 8-byte pieces -- f16c8_store4 -- cost 10 store instructions per pass against the 3 of the fp32 rows and made the
                    // launch 31 us slower: the CU's vector-memory path is what the eight epilogues of a workgroup share.)
                    const int pass = ih * 2 + t;
                    float sum = 0.f;
#pragma unroll
                    for (int cb = 0; cb < 3; ++cb) sum += (v[cb][0] + v[cb][1]) + (v[cb][2] + v[cb][3]);
                    const float mean = oct_sum(sum) * (1.0f / 96.0f);
                    float m2 = 0.f;
#pragma unroll
                    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float d = v[cb][e] - mean; m2 = fmaf(d, d, m2); }
                    m2 = oct_sum(m2);
                    if (c4 == pass) { keep_mean = mean; keep_m2 = m2; }
                    unsigned hh[3][2], ll[3];
#pragma unroll
                    for (int cb = 0; cb < 3; ++cb) {
                        const float v4[4] = {v[cb][0], v[cb][1], v[cb][2], v[cb][3]};
                        _Float16 h[4];
                        float lo[4];
                        f16c8_split<4>(v4, h, lo);
                        typedef _Float16 h2_ __attribute__((ext_vector_type(2)));
                        hh[cb][0] = __builtin_bit_cast(unsigned, (h2_){h[0], h[1]});
                        hh[cb][1] = __builtin_bit_cast(unsigned, (h2_){h[2], h[3]});
                        int l0 = 0;
                        l0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], l0, false);
                        l0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], l0, true);
                        ll[cb] = (unsigned)l0;
                    }
                    const bool odd = c4 & 1;
                    auto emit8 = [&](int row, int col, unsigned a0, unsigned a1, unsigned b0, unsigned b1, unsigned la, unsigned lb) {
                        // 8 consecutive elements of row `row` starting at column `col` (col % 8 == 0): (a0, a1 | b0, b1) f16 pairs, (la | lb) lo8
                        // (32-bit element offsets against wave-uniform bases: one address VGPR per store; the host checks M * ld < 2^31)
                        if (row < M) {
                            const unsigned e = (unsigned)row * (unsigned)p.ln_op_ld + (unsigned)col;
                            const unsigned g8 = (e >> 3) & 3u;                                        // f16c8_lo_index, 32-bit
                            const unsigned el = (e & ~31u) + (((g8 & 1u) << 4) | ((g8 >> 1) << 3));
                            unsigned char* const b0p = (unsigned char*)p.ln_op_out;
                            unsigned char* const b1p = b0p + 2 * p.ln_op_plane;
                            *(u128*)(b0p + (size_t)(2u * e)) = (u128){a0, a1, b0, b1};
                            *(uint2*)(b1p + (size_t)el) = make_uint2(la, lb);
                        }
                    };
                    {   // blocks 0 / 1: the even lane sends its block-1 half and receives the odd lane's block-0 half, and vice versa
                        const unsigned r0 = dpp_xor1(odd ? hh[0][0] : hh[1][0]), r1 = dpp_xor1(odd ? hh[0][1] : hh[1][1]), rl = dpp_xor1(odd ? ll[0] : ll[1]);
                        const int col = wn0 + (odd ? 32 : 0) + (c4 >> 1) * 8;
                        if (odd) emit8(gr, col, r0, r1, hh[1][0], hh[1][1], rl, ll[1]);
                        else emit8(gr, col, hh[0][0], hh[0][1], r0, r1, ll[0], rl);
                    }
                    {   // block 2: the even lane of a pair takes both halves (the odd lane's store slots stay empty: 32 lanes x 16 / 8 bytes)
                        const unsigned r0 = dpp_xor1(hh[2][0]), r1 = dpp_xor1(hh[2][1]), rl = dpp_xor1(ll[2]);
                        if (!odd) emit8(gr, wn0 + 64 + (c4 >> 1) * 8, hh[2][0], hh[2][1], r0, r1, ll[2], rl);
                    }
                }
            }
        }
        if constexpr (EP == 4) {
            // the eight passes' (mean, M2) pairs: lane (rsub, c4) holds pass c4 = (32-row block, 16-row chunk, 8-row pass) of row group rsub
            static_assert(MI == 2, "eight passes, eight lanes per row");
            const int c4 = lane & 7, rsub = lane >> 3;
            const int row = wm0 + (c4 >> 2) * 32 + ((c4 >> 1) & 1) * 16 + (c4 & 1) * 8 + rsub;
            if (row < M) *(float2*)((unsigned char*)p.ln_stats_out + (size_t)(((unsigned)row * (unsigned)(p.N / 96) + (unsigned)(wn0 / 96)) * 8u)) = make_float2(keep_mean, keep_m2);
        }
    } else if constexpr (EP == 5) {
        // LayerNorm fold, producer side with the 3-byte residual stream (bd_gemm_args.ln_resid_in_op): the residual rows are the F16C8 operand
        // copy the PREVIOUS residual Linear left at ln_op_out, the sum goes back there in place (+ its row statistics), and fp32 rows are
        // written only for OUTK == OUT_F32 (someone reads the stream as fp32 before the next residual Linear).  Everything happens in the
        // 16-bit row layout: a lane owns 3 x 8 columns of one row per 16-row chunk.  The residual pieces of chunk c + 1 (3 x 16 + 3 x 8
        // bytes per lane) are requested BEFORE chunk c's stores, into the second of two register sets: on gfx9 loads and stores share
        // the in-order vmcnt, so a load issued behind a store could only be waited for together with that store's acknowledgement.
        static_assert(MI == 2, "four 16-row chunks, four lanes per row");
        const int c8 = lane & 3, rsub = lane >> 2;
        const unsigned ld = (unsigned)p.ln_op_ld;
        unsigned char* const b0p = (unsigned char*)p.ln_op_out;
        unsigned char* const b1p = b0p + 2 * p.ln_op_plane;
        auto lo_index32 = [](unsigned e) { const unsigned g8 = (e >> 3) & 3u; return (e & ~31u) + (((g8 & 1u) << 4) | ((g8 >> 1) << 3)); };
        u128 rh[2][3];
        uint2 rl[2][3];
        auto load_resid = [&](int buf, int ih) {
            int row = wm0 + (ih >> 1) * 32 + (ih & 1) * 16 + rsub;
            row = row < M ? row : M - 1;                    // (rows past the edge read a valid row; their results are never stored)
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const unsigned e = (unsigned)row * ld + (unsigned)(wn0 + cb * 32 + c8 * 8);
                rh[buf][cb] = *(const u128*)(b0p + (size_t)(2u * e));
                rl[buf][cb] = *(const uint2*)(b1p + (size_t)lo_index32(e));
            }
        };
        float keep_mean = 0.f, keep_m2 = 0.f;
        load_resid(0, 0);
#pragma unroll
        for (int ih = 0; ih < 2 * MI; ++ih) {
            const int i = ih >> 1, hc = ih & 1, buf = ih & 1;
            to_scratch(i, hc);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = hc * 8; r < hc * 8 + 8; ++r) acc[i][j][r] = 0.f;
            if (ih + 1 < 2 * MI) load_resid(buf ^ 1, ih + 1);
            const int gr = wm0 + i * 32 + hc * 16 + rsub;
            float v[3][8];
            float sum = 0.f;
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const float* src = (rsub < 8 ? sc : sc_hi - 8 * COLS) + rsub * COLS + cb * 32 + c8 * 8;
                const f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
                const f16x8 h = __builtin_bit_cast(f16x8, rh[buf][cb]);
                const int l0 = (int)rl[buf][cb].x, l1 = (int)rl[buf][cb].y;
                const float lo[8] = {__builtin_amdgcn_cvt_f32_fp8(l0, 0), __builtin_amdgcn_cvt_f32_fp8(l0, 1), __builtin_amdgcn_cvt_f32_fp8(l0, 2),
                                     __builtin_amdgcn_cvt_f32_fp8(l0, 3), __builtin_amdgcn_cvt_f32_fp8(l1, 0), __builtin_amdgcn_cvt_f32_fp8(l1, 1),
                                     __builtin_amdgcn_cvt_f32_fp8(l1, 2), __builtin_amdgcn_cvt_f32_fp8(l1, 3)};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = fmaf(lo[e], 1.0f / (float)(1 << BD_F16C8_D), (float)h[e]);      // the residual element: hi + lo8 2^-D (exact in fp32)
                    v[cb][e] = (e < 4 ? a0[e] : a1[e - 4]) + x;
                    sum += v[cb][e];
                }
            }
            const float mean = quad_sum(sum) * (1.0f / 96.0f);
            float m2 = 0.f;
#pragma unroll
            for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[cb][e] - mean; m2 = fmaf(d, d, m2); }
            m2 = quad_sum(m2);
            if (c8 == ih) { keep_mean = mean; keep_m2 = m2; }
            if (gr < M) {
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) {
                    const unsigned e = (unsigned)gr * ld + (unsigned)(wn0 + cb * 32 + c8 * 8);
                    if constexpr (OUTK == OUT_F32) {
                        float* op = (float*)p.out + (int64_t)gr * p.ldo + wn0 + cb * 32 + c8 * 8;
                        *(f32x4*)op = (f32x4){v[cb][0], v[cb][1], v[cb][2], v[cb][3]};
                        *(f32x4*)(op + 4) = (f32x4){v[cb][4], v[cb][5], v[cb][6], v[cb][7]};
                    }
                    _Float16 hh[8];
                    float lo[8];
                    f16c8_split<8>(v[cb], hh, lo);
                    f16x8 hv;
#pragma unroll
                    for (int q = 0; q < 8; ++q) hv[q] = hh[q];
                    int l0 = 0, l1 = 0;
                    l0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], l0, false); l0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], l0, true);
                    l1 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4], lo[5], l1, false); l1 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[6], lo[7], l1, true);
                    *(u128*)(b0p + (size_t)(2u * e)) = __builtin_bit_cast(u128, hv);
                    *(uint2*)(b1p + (size_t)lo_index32(e)) = make_uint2((unsigned)l0, (unsigned)l1);
                }
            }
        }
        {   // the four chunks' (mean, M2) pairs in one store: lane (rsub, c8) holds chunk c8 of row group rsub
            const int row = wm0 + (c8 >> 1) * 32 + (c8 & 1) * 16 + rsub;
            if (row < M) *(float2*)((unsigned char*)p.ln_stats_out + (size_t)(((unsigned)row * (unsigned)(p.N / 96) + (unsigned)(wn0 / 96)) * 8u)) = make_float2(keep_mean, keep_m2);
        }
    } else if constexpr (EP == 1 && NS == 1 && sizeof(T) == 2 && OUTK == OUT_OPERAND) {
        // Plain bf16 / f16 result (optional GELU): rounded to 16 bits BEFORE the LDS round trip, two adjacent rows per dword
        // (registers r, r + 1 of a C fragment are rows 2 k, 2 k + 1 of the same column), so the transposition moves half the bytes:
        // per 32-row block 24 ds_write_b32 + 6 ds_read_b128 per lane instead of 48 + 12.  A lane then owns 8 consecutive columns of
        // one row PAIR, splits the dwords with two byte permutes each and stores two 16-byte row pieces; 12 lanes cover a row
        // (192-byte runs).  Same values, same roundings as the fp32 staging (the conversion is the separate step of store_cvt).
        unsigned* sp = (unsigned*)sc;
        unsigned* sp_hi = (unsigned*)sc_hi;          // pair-rows 8-15
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) {
                    float v2[2] = {acc[i][j][2 * pp] + bj[j], acc[i][j][2 * pp + 1] + bj[j]};
                    if constexpr (GELU) gelu_n<GeluKind<T, NS>::value, 2>(v2);
                    float x0 = v2[0], x1 = v2[1];
                    asm("" : "+v"(x0));          // separate fp32 -> 16-bit rounding (store_cvt's rule)
                    asm("" : "+v"(x1));
                    typedef T pair_t __attribute__((ext_vector_type(2)));
                    const pair_t pr = {(T)x0, (T)x1};
                    const int prow = (pp & 1) + 4 * (pp >> 1) + 2 * lhalf;
                    ((pp >> 1) < 2 ? sp : sp_hi - 8 * COLS)[prow * COLS + j * 32 + lrow] = __builtin_bit_cast(unsigned, pr);
                }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int t = u * 64 + lane, prow = t / 12, cg = t % 12;
                const unsigned* const spr = (prow < 8 ? sp : sp_hi - 8 * COLS) + prow * COLS + cg * 8;
                const u128 d0 = *(const u128*)spr, d1 = *(const u128*)(spr + 4);
                u128 ra, rb;          // even row: low halves, odd row: high halves
                ra[0] = __builtin_amdgcn_perm(d0[1], d0[0], 0x05040100u); ra[1] = __builtin_amdgcn_perm(d0[3], d0[2], 0x05040100u);
                ra[2] = __builtin_amdgcn_perm(d1[1], d1[0], 0x05040100u); ra[3] = __builtin_amdgcn_perm(d1[3], d1[2], 0x05040100u);
                rb[0] = __builtin_amdgcn_perm(d0[1], d0[0], 0x07060302u); rb[1] = __builtin_amdgcn_perm(d0[3], d0[2], 0x07060302u);
                rb[2] = __builtin_amdgcn_perm(d1[1], d1[0], 0x07060302u); rb[3] = __builtin_amdgcn_perm(d1[3], d1[2], 0x07060302u);
                const int gr = wm0 + i * 32 + 2 * prow;
                T* op = (T*)p.out + (int64_t)gr * p.ldo + wn0 + cg * 8;
                if (gr < M) __builtin_nontemporal_store(ra, (u128*)op);
                if (gr + 1 < M) __builtin_nontemporal_store(rb, (u128*)(op + p.ldo));
            }
        }
    } else {
        // 16-bit rows: 4 lanes x 16 bytes per row and 32-column block, 16 rows per pass
        const int c8 = lane & 3, rsub = lane >> 2;
        auto store8 = [&](int64_t e, const float (&v)[8]) {
            if constexpr (OUTK == OUT_F16) store_cvt<_Float16, 8>((_Float16*)p.out + e, v);
            else if constexpr (OUTK == OUT_BF16) store_cvt<__bf16, 8>((__bf16*)p.out + e, v);
            else if constexpr (OUTK == OUT_BF16X2) store_operand8<__bf16, 2>((__bf16*)p.out, p.out_plane, e, v);
            else if constexpr (OUTK == OUT_F16X2) store_operand8<_Float16, 2>((_Float16*)p.out, p.out_plane, e, v);
            else store_operand8<T, NS>((T*)p.out, p.out_plane, e, v);
        };
        float wv[EP == 2 ? 3 : 1][8];
        bool norm = false;
        if constexpr (EP == 2) {
            // the 96-column wave tile IS one head (host-checked); which third of the output it lies in is wave-uniform
            const int part = wn0 / (p.N / (p.rms_parts == 2 ? 2 : 3));
            norm = part < 2;
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const float* src = rmsw + (part & 1) * 256 + cb * 32 + c8 * 8;
                const f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { wv[cb][e] = a0[e]; wv[cb][4 + e] = a1[e]; }
            }
        }
#pragma unroll
        for (int ih = 0; ih < 2 * MI; ++ih) {
            const int i = ih >> 1, hc = ih & 1;
            to_scratch(i, hc);
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = hc * 8; r < hc * 8 + 8; ++r) acc[i][j][r] = 0.f;
            const int gr = wm0 + i * 32 + hc * 16 + rsub;
            float v[3][8];
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const float* src = (rsub < 8 ? sc : sc_hi - 8 * COLS) + rsub * COLS + cb * 32 + c8 * 8;
                const f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[cb][e] = a0[e]; v[cb][4 + e] = a1[e]; }
            }
            if constexpr (GELU) {
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) gelu_n<GeluKind<T, NS>::value, 8>(v[cb]);   // 16/8-bit result: fitted forms (bd_common.h)
            }
            if constexpr (EP == 2) {
                float ss = 0.f;
#pragma unroll
                for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss = fmaf(v[cb][e], v[cb][e], ss);
                ss = quad_sum(ss);
                const float inv = rsqrtf(ss * (1.0f / 96.0f) + p.rms_eps);
                if (norm) {
#pragma unroll
                    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[cb][e] = wv[cb][e] * (v[cb][e] * inv);     // w * (x * rsqrt(..)): blocks.py:51-56
                }
            }
            if (gr < M) {
                const int64_t e0 = (int64_t)gr * p.ldo + wn0 + c8 * 8;
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) store8(e0 + cb * 32, v[cb]);
            }
        }
    }
}


// LayerNorm fold, consumer side: one PRODUCER-wave lane per tile row turns the row's eight (mean, M2) pairs -- one per 96 columns of the K = 768
// row, written by the launch that produced the row (pc_epilogue EP 4) -- into (rstd, -mean rstd).  Chan's combination for equal counts, in a
// fixed order: the result does not depend on which launch form or tile shape wrote or reads the pairs.
__device__ __forceinline__ float2 ln_rows_combine(const f32x4 (&st)[4], float eps) {
    const float m[8] = {st[0][0], st[0][2], st[1][0], st[1][2], st[2][0], st[2][2], st[3][0], st[3][2]};
    const float q[8] = {st[0][1], st[0][3], st[1][1], st[1][3], st[2][1], st[2][3], st[3][1], st[3][3]};
    float mean = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) mean += m[i];
    mean *= 0.125f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = m[i] - mean; m2 += q[i] + 96.0f * (d * d); }
    const float rstd = rsqrtf(m2 * (1.0f / 768.0f) + eps);
    return make_float2(rstd, -mean * rstd);
}
// host side: do these arguments ask for the fold, and are they well-formed for it (the kernel forms are checked by the launchers)
inline bool ln_fold_producer(const bd_gemm_args& a) { return a.ln_stats_out != nullptr || a.ln_op_out != nullptr || a.ln_resid_in_op != 0; }
inline bool ln_fold_consumer(const bd_gemm_args& a) { return a.ln_stats_in != nullptr || a.ln_colsum != nullptr; }
inline bool ln_fold_producer_ok(const bd_gemm_args& a) {
    if (a.ln_resid_in_op ? (a.resid != nullptr || (a.out_f32 != OUT_F32 && a.out_f32 != OUT_OPERAND)) : a.out_f32 != OUT_F32) return false;
    return a.ln_stats_out && a.ln_op_out && a.N == 768 && a.act == BD_ACT_NONE && !a.rms_wq && !a.addtab && a.rpg_in <= 0 &&
           !a.wscale && a.ln_op_ld % 32 == 0 && a.ln_op_plane % 8 == 0 && (((uintptr_t)a.ln_op_out | (uintptr_t)a.ln_stats_out) & 15) == 0 &&
           (int64_t)a.M * a.ln_op_ld < ((int64_t)1 << 30);        // (the epilogue addresses the copy with 32-bit byte offsets)
}
inline bool ln_fold_consumer_ok(const bd_gemm_args& a) {
    return a.ln_stats_in && a.ln_colsum && a.K == 768 && a.N % 192 == 0 && a.out_f32 != OUT_F32 && !a.resid && !a.addtab && a.rpg_in <= 0 && !a.wscale &&
           a.bias && (((uintptr_t)a.ln_stats_in | (uintptr_t)a.ln_colsum | (uintptr_t)a.bias) & 15) == 0;
}

// the wide (LDS-staged, 16-byte) epilogue needs 16-byte aligned rows: N % 8 == 0 and aligned leading dimensions / pointers
inline bool wide_epilogue_ok(const bd_gemm_args& p, int ns) {
    return (p.N % 8 == 0) && (p.ldo % 8 == 0) && (((uintptr_t)p.out & 15) == 0) &&
           (!p.resid || ((p.ldr % 4 == 0) && ((uintptr_t)p.resid & 15) == 0)) &&
           (!p.bias || ((uintptr_t)p.bias & 15) == 0) && (!p.addtab || ((uintptr_t)p.addtab & 15) == 0) &&
           (!p.wscale || ((uintptr_t)p.wscale & 15) == 0) && (p.out_f32 || ns == 1 || (p.out_plane % 8 == 0));
}


// Compute units of the current device (MI355X: 256; partitioned / harvested parts differ): the tile-choice model counts
// resident workgroup slots per round (one 256x256 workgroup per CU, two 128x128, four 64x64).  Immutable device property,
// looked up once per device.
inline int cu_count() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// rows [row0, row0 + rows) of the problem as its own launch (no row remap / table: checked by the caller)
inline bool rms_geometry_ok(const bd_gemm_args& a) {
    return a.rms_wq && a.rms_wk && a.out_f32 != OUT_F32 && a.act == BD_ACT_NONE && !a.resid && !a.addtab && a.rpg_in <= 0 &&
           (a.rms_parts == 0 || a.rms_parts == 2 || a.rms_parts == 3) &&
           a.N % (a.rms_parts == 2 ? 2 : 3) == 0 && (a.N / (a.rms_parts == 2 ? 2 : 3)) % 96 == 0 && a.N % 192 == 0 && (((uintptr_t)a.rms_wq | (uintptr_t)a.rms_wk) & 3) == 0;
}

}  // namespace

// Rows the persistent 256 x 192 kernel takes when the tiles of ALL rows would end in a nearly empty round: DINOv2's M = 50112 is
// 195.75 row tiles -- N = 768: 784 tiles = 3 rounds + 16 tiles, a fourth round on 16 of 256 CUs.  The rows of the k full rounds stay
// here, the rest (960 rows) goes to the one-tile kernels as a second, sparse launch (fc2 266 -> 231 us, proj 95 -> 90 us;
// profiles/r4_gemm_tail_rows.md).  Only where a round is a large part of the launch (k <= 4): behind 9 or 12 rounds (N = 2304, 3072)
// the second launch costs what the empty round did.  A row's result does not depend on the kernel form (same K order, same
// epilogue arithmetic: tests/test_gpu_ops.py::test_gemm_row_result_independent_of_tile_shape).
inline int64_t pc192_main_rows(const bd_gemm_args& a, int cus) {
    if (a.rms_wq || a.addtab || a.rpg_in > 0) return a.M;        // (row_slice does not re-phase the table / row-group maps)
    const int64_t tn = a.N / 192, mt = (a.M + 255) / 256, nt = mt * tn;
    const int64_t k = nt / cus, rem = nt % cus;
    if (k < 1 || k > 4 || rem == 0 || rem * 4 > cus) return a.M;  // the last round is at least a quarter full: leave it
    const int64_t rows = (k * cus / tn) * 256;      // 0 when cus < tn (a CU partition / mask narrower than one row of tiles): no split then
    return rows > 0 ? rows : a.M;
}

// the F16C8 class has its own persistent kernel (gemm_f16c8.hip); every shape goes through it
int bd_launch_gemm_f16c8(const bd_gemm_args& a, hipStream_t s);
bool bd_f16c8_takes_ln_fold(const bd_gemm_args& a);
