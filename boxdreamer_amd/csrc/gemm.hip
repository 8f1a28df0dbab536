// MFMA GEMM with fused epilogue for every nn.Linear on the corner-heatmap path.
//
//   out[map(r), :] = act(wscale * (A[r, :] . W^T) + bias) + addtab[r % tab_rows, :] + resid[map(r), :]
//
// gfx950 design.  Both operands are K-contiguous ("A . W^T", the nn.Linear layout), so A and W tiles share one
// loader.  Tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 (LDS-DMA: no staging VGPRs, no ds_write pass).  An
// LDS-DMA wave-instruction writes 1 KiB at  M0-base + lane*16  (lane-linear), so the XOR swizzle that makes the
// ds_read_b128 fragment reads conflict-free is applied on the SOURCE side: the lane that owns LDS slot (row, c')
// fetches logical chunk c = c' ^ swz(row) of that row (same 128-byte line: coalescing unchanged), and fragment reads
// apply the same involution.  One barrier per K-slab (s_waitcnt vmcnt(0) + s_barrier): slab kt landed, and the
// buffer about to be overwritten is free; the DMA for slab kt+1 is issued right after it and flies under slab kt.
// Workgroup ids are remapped (XCD-aware, grouped raster) so that each XCD's L2 sees a compact patch of tiles.
//
// Operand classes (template T):
//   bf16 / f16   v_mfma_f32_32x32x16, slab = 64 k  (128-byte tile rows), a fragment = one 16-byte chunk per lane
//   split-bf16   NS = 2 planes (hi, lo): hi*hi + hi*lo + lo*hi per fragment pair, slab = 32 k
//   e4m3         v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (2x rate), slab = 128 k (128-byte rows),
//                a fragment = two chunks per lane; per-output-channel weight scale applied in the epilogue
// Tile geometry (template): WM x WN waves, each owning an (MI*32) x (NI*32) sub-tile:
//   <2,4,4,2> 256x256, 8 waves, 128 KiB LDS, 1 workgroup/CU (wide outputs: 128 FLOP per LDS-DMA byte)
//   <2,2,2,2> 128x128, 4 waves, 64 KiB, 2 workgroups/CU (N = 768 outputs)      <2,2,1,1> 64x64 (latency mode)
// Experiments that did NOT pay (deeper LDS-DMA rings, mid-slab barriers, ping-pong wave groups, 256x128 tiles at one or
// two workgroups per CU, a persistent tile loop with the epilogue overlapped, pinned fragment prefetch) and the counters
// behind the choices: profiles/r1_gemm_experiments.md, profiles/r1_gemm_pmc.md.
#include "bd_common.h"
#include <type_traits>

#ifdef BD_GEMM_PROBE
// Measurement build only (tools/gemm_phase_probe.py; never part of libboxdreamer_hip.so): per-wave shader-clock stamps of
// the mainloop phases, kept in the lanes of one VGPR (lane i = stamp i) and written out once at kernel end.
__device__ unsigned* bd_probe_buf = nullptr;
extern "C" int bd_gemm_probe_set(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(bd_probe_buf), &buf, sizeof(buf));
}
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
                            const T hi = from_f32<T>(v);
                            o[0] = hi;
                            if (NS == 2) o[out_plane] = from_f32<T>(v - to_f32<T>(hi));
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

// ------------------------------------------------------------------------------------------------
// NSTG > 2 (small-batch launches: fewer tiles than the chip has workgroup slots, so a tile's time is the chain of its slabs' DMA
// latencies, not the matrix work): a ring of NSTG stages with NSTG - 1 slabs in flight and counted vmcnt waits; the arithmetic
// (K order per row) is that of the 2-stage form, so a row's result does not depend on which form a launch takes.
template <class T, int NS, int BK, int WM, int WN, int MI, int NI, int NSTG = 2>
__global__ __launch_bounds__(WM * WN * 64, NSTG == 2 ? 2 : 1) void gemm_kernel_glds(const bd_gemm_args p) {
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    typedef typename Op16<T>::vec8 frag_t;
    constexpr int ESZ = OpGeom<T>::ESZ, KSTEP = OpGeom<T>::KSTEP, CPF = OpGeom<T>::CPF;
    constexpr int NWAVE = WM * WN;
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32;
    constexpr int ROWB = BK * ESZ;                    // bytes per tile row per slab
    constexpr int CH = ROWB / 16;                     // 16-byte chunks per tile row
    constexpr int RPP = 64 / CH;                      // tile rows per 1-KiB DMA piece
    constexpr int A_BYTES = TBM * ROWB, W_BYTES = TBN * ROWB;
    constexpr int PPW_A = A_BYTES / 1024 / NWAVE, PPW_W = W_BYTES / 1024 / NWAVE;
    static_assert(PPW_A >= 1 && PPW_W >= 1 && A_BYTES % (1024 * NWAVE) == 0 && W_BYTES % (1024 * NWAVE) == 0, "tile/DMA split");
    constexpr int STAGE_BYTES = (A_BYTES + W_BYTES) * NS;   // A planes then W planes
    constexpr int KS = BK / KSTEP;
    static_assert(KS >= 1 && CH * 16 == ROWB && (CH == 4 || CH == 8), "slab geometry");
    constexpr int EPI_SCRATCH = NWAVE * 32 * NI * 32 * 4;   // the wide epilogue's per-wave transpose scratch
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NSTG * STAGE_BYTES > EPI_SCRATCH ? NSTG * STAGE_BYTES : EPI_SCRATCH];
    constexpr int PPW = (PPW_A + PPW_W) * NS;         // DMA pieces this wave issues per slab
    static_assert(NSTG >= 2 && NSTG <= 4 && (NSTG - 2) * PPW < 64, "ring depth / vmcnt range");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    int m0, n0;
    tile_coords_t<TBM, TBN>(p.M, p.N, TBM >= 256 ? 4 : 8, m0, n0);

    // per-lane DMA sources (byte pointers): piece j covers tile rows j*RPP .. j*RPP+RPP-1
    const unsigned char* ga[PPW_A];
    const unsigned char* gw[PPW_W];
#pragma unroll
    for (int i = 0; i < PPW_A; ++i) {
        const int row = (wid * PPW_A + i) * RPP + lane / CH;
        int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
        ga[i] = (const unsigned char*)p.A + ((int64_t)ar * p.lda) * ESZ + swz_chunk<CH>(row, lane % CH) * 16;
    }
#pragma unroll
    for (int i = 0; i < PPW_W; ++i) {
        const int row = (wid * PPW_W + i) * RPP + lane / CH;
        int wr = n0 + row; wr = wr < p.N ? wr : p.N - 1;
        gw[i] = (const unsigned char*)p.W + ((int64_t)wr * p.ldw) * ESZ + swz_chunk<CH>(row, lane % CH) * 16;
    }
    const int64_t a_plane = p.a_plane * ESZ, w_plane = p.w_plane * ESZ;
    const unsigned lds_off = lds_offset_of(lds);
#define DMA_SLAB(buf, kb)                                                                                     \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                          \
        _Pragma("unroll") for (int i = 0; i < PPW_A; ++i)                                                     \
            glds16(ga[i] + s * a_plane + (kb), lds_off + (buf) * STAGE_BYTES + s * A_BYTES + (wid * PPW_A + i) * 1024); \
        _Pragma("unroll") for (int i = 0; i < PPW_W; ++i)                                                     \
            glds16(gw[i] + s * w_plane + (kb), lds_off + (buf) * STAGE_BYTES + NS * A_BYTES + s * W_BYTES + (wid * PPW_W + i) * 1024); \
    }

    f32x16 acc[MI][NI];
    acc_init<MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);

    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nk = p.K / BK;
#ifdef BD_GEMM_PROBE
    unsigned probe_ts = 0;
#endif
    BD_PROBE(60)
    DMA_SLAB(0, 0)
    if constexpr (NSTG > 2) {
#pragma unroll
        for (int s2 = 1; s2 < NSTG - 1; ++s2)
            if (s2 < nk) { DMA_SLAB(s2, s2 * ROWB) }
    }
    for (int kt = 0; kt < nk; ++kt) {
        BD_PROBE(kt * 3)
        if constexpr (NSTG == 2) {
            slab_barrier();                                // slab kt landed; buffer (kt+1)&1 is free
        } else {
            // everything but the (up to NSTG - 2) younger slabs' pieces of this wave has landed; then everyone else's
            const int younger = nk - 1 - kt < NSTG - 2 ? nk - 1 - kt : NSTG - 2;
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        BD_PROBE(kt * 3 + 1)
        if (kt + NSTG - 1 < nk) { DMA_SLAB((kt + NSTG - 1) % NSTG, (kt + NSTG - 1) * ROWB) }      // into the stage of slab kt - 1
        BD_PROBE(kt * 3 + 2)
        const unsigned char* base = lds + (kt % NSTG) * STAGE_BYTES;
        // fragments are double-buffered in registers (one plane only: two sets next to 128 accumulators would
        // spill in the split mode): the LDS reads of k-step ks+1 are in flight while the MFMAs of ks issue
        constexpr int FB = NS == 1 ? 2 : 1;
        frag_t a[FB][NS][MI], b[FB][NS][NI];
        // a fragment = CPF consecutive chunks starting at chunk (ks*2 + lane_half) * CPF of the row
#define LOAD_ONE(dst, ptr, row, ks)                                                                           \
        {                                                                                                     \
            const int c0_ = ((ks) * 2 + lhalf) * CPF;                                                         \
            if constexpr (CPF == 1) {                                                                         \
                dst = __builtin_bit_cast(frag_t, *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_) << 4))); \
            } else {                                                                                          \
                const u128 lo_ = *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_) << 4));     \
                const u128 hi_ = *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_ + 1) << 4)); \
                dst = (frag_t){(int)lo_[0], (int)lo_[1], (int)lo_[2], (int)lo_[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]}; \
            }                                                                                                 \
        }
#define LOAD_FRAGS(ks, slot)                                                                                  \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                      \
            _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                    \
                LOAD_ONE(a[slot][s][i], base + s * A_BYTES, wm * (MI * 32) + i * 32 + lrow, ks)               \
            _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                    \
                LOAD_ONE(b[slot][s][j], base + NS * A_BYTES + s * W_BYTES, wn * (NI * 32) + j * 32 + lrow, ks) \
        }
        if (FB == 2) { LOAD_FRAGS(0, 0) }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cur = FB == 2 ? (ks & 1) : 0;
            if (FB == 2) { if (ks + 1 < KS) { LOAD_FRAGS(ks + 1, (cur ^ 1) & (FB - 1)) } }
            else { LOAD_FRAGS(ks, 0) }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if (NS == 2) {
                        acc[i][j] = Op16<T>::mfma(a[cur][NS - 1][i], b[cur][0][j], acc[i][j]);   // lo * hi
                        acc[i][j] = Op16<T>::mfma(a[cur][0][i], b[cur][NS - 1][j], acc[i][j]);   // hi * lo
                    }
                    acc[i][j] = Op16<T>::mfma(a[cur][0][i], b[cur][0][j], acc[i][j]);            // hi * hi
                }
        }
#undef LOAD_FRAGS
#undef LOAD_ONE
    }
#undef DMA_SLAB
    // wide epilogue needs 16-byte aligned rows: N % 8 == 0 and aligned leading dimensions (else scalar path)
    const bool wide = (p.N % 8 == 0) && (p.ldo % 8 == 0) && (((uintptr_t)p.out & 15) == 0) &&
                      (!p.resid || ((p.ldr % 4 == 0) && ((uintptr_t)p.resid & 15) == 0)) &&
                      (!p.bias || ((uintptr_t)p.bias & 15) == 0) && (!p.addtab || ((uintptr_t)p.addtab & 15) == 0) &&
                      (!p.wscale || ((uintptr_t)p.wscale & 15) == 0) &&
                      (p.out_f32 || NS == 1 || (p.out_plane % 8 == 0));
    BD_PROBE(61)
    if (wide) {
        __syncthreads();                                   // every wave is done with the operand slabs
        BD_PROBE(62)
        gemm_epilogue_lds<T, NS, MI, NI>(p, acc, lds + wid * (32 * NI * 32 * 4), m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
    } else {
        gemm_epilogue<T, NS, MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
    }
#ifdef BD_GEMM_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the epilogue's stores have left
    BD_PROBE(63)
    if (bd_probe_buf && blockIdx.x < 1024) bd_probe_buf[((size_t)blockIdx.x * NWAVE + wid) * 64 + lane] = probe_ts;
#endif
}

// ------------------------------------------------------------------------------------------------
// Producer / consumer, persistent form of the same GEMM  (round 2).
//
// Why (tools/gemm_phase_probe.py, profiles/r2_gemm_phase_probe.md): in gemm_kernel_glds every wave issues its share of the
// next slab's LDS-DMA right after the slab barrier.  The CU's texture-address path accepts one 1-KiB global_load_lds per ~23
// cycles, so the 64 pieces of a 256x256x64 slab keep the ISSUING waves blocked for 580 (older wave of a SIMD) to 1470 cycles
// (younger wave) per slab, during which they issue no MFMA: a slab takes ~3300 cycles for 2048 cycles of matrix work per
// SIMD, and the two waves of a SIMD end up running their MFMA phases one after the other.
//
// Here the workgroup has WM*WN consumer waves (MFMA + epilogue, never a VMEM instruction inside the K loop) and ONE producer
// wave that issues every LDS-DMA piece (SGPR base + 32-bit VGPR offset form) and is the only wave that waits on vmcnt.  One
// s_barrier per slab:  B(kt) = "slab kt has landed (the producer waited for it) and every consumer is done with slab kt-1";
// after B(kt) the producer refills the buffer slab kt-1 lived in.  The kernel is persistent (grid = CUs, static XCD-aware
// tile lists), which buys two more overlaps: the first slab of the NEXT tile is fetched under the last slab of this one, and
// the epilogue's global stores are not waited for by anyone -- they drain under the next tile's MFMAs (in the one-tile kernel
// all CUs write their tiles at the same moment: QKV's 128 KiB per CU sit at the HBM write floor of ~10k cycles while the
// matrix pipes idle).  Barriers per tile: nk (B) + 1 (X: every consumer is done reading the last slab).
//
// Consumers execute NO vector-memory LOAD anywhere in the specialised epilogues (EP 1-3 below): on gfx9 loads and stores share
// the in-order vmcnt, so a load issued after the epilogue's stores (a bias vector, a residual row, a register-spill reload)
// can only be waited for together with those stores' acknowledgements.  The first version of this kernel had exactly that
// (generic epilogue under the 168-VGPR cap: scratch reloads and per-column-block bias loads between the stores) and a LONE
// workgroup spent 13.8k cycles per tile in its epilogue against 1.9k for the bare stores (profiles/r2_gemm_epilogue.md).  Now:
//   * per-column vectors (bias, e4m3 weight scale) of a tile arrive in a small LDS side buffer with the tile's first slab
//     (one extra 1-KiB DMA piece each, producer wave 0; double-buffered by tile parity); the q/k RMSNorm weights once per kernel;
//   * a plain fp32 residual is loaded INTO the accumulators of the NEXT tile (acc_init convention) from inside this tile's
//     epilogue, chunk by chunk as the accumulator registers are flushed to the LDS scratch -- before that chunk's stores;
//   * the epilogue kind is a template parameter (EP), so no path carries the other paths' live values or branches.
// EP: 0 = generic (gemm_epilogue_lds: table add, row remap, residual in scaled modes, odd output kinds),
//     1 = 16-bit / 8-bit result (OUTK: operand-native, f16, bf16), optional GELU,
//     2 = 16-bit result with the fused q/k RMSNorm,
//     3 = fp32 result (+ fp32 residual through the accumulators).
// What still bounds it (profiles/r2_gemm_phase_probe.md): per slab the producers need ~1000 cycles to issue + ~840 to land
// the next slab (a 2-stage ring cannot hide that latency: 2045 cycles per slab for 1536 of matrix work per SIMD).
// Negative results kept in the profile notes: a row-per-lane epilogue on transposed accumulators (no LDS, 2.1x slower: a wave
// store touching 32 rows x 32 bytes is issue-bound in the texture path), non-temporal vs plain stores (equal), start-up
// staggers that spread the CUs' epilogues over the tile period (+1 % / -9 % QKV, -7 % / -2 % fc1).

// quad all-reduce (lanes 4k .. 4k+3) on the VALU: two DPP quad_perm adds
__device__ __forceinline__ float quad_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // [2,3,0,1]
    return x;
}

// Epilogue of gemm_kernel_pc for the 64 x 96 wave tile (MI = 2, NI = 3), 16 rows per pass through this wave's 6-KiB scratch.
//   colp: this tile's per-column vectors in LDS (bias at [0, 256), weight scale at [256, 512)), indexed by tile column
//   rmsw: q weights at [0, 96), k weights at [256, 352)
//   wcol: first column of the wave tile inside the workgroup tile;  (wm0, wn0): its global origin
//   next: the wave tile origin of this workgroup's next tile (EP 3 pre-loads its residual), has_next = there is one
template <class T, int NS, int EP, int OUTK, bool GELU, int MI = 2>
__device__ __forceinline__ void pc_epilogue(const bd_gemm_args& p, f32x16 (&acc)[MI][3], float* sc, const float* colp, const float* colp_next,
                                            const float* rmsw, int wcol, int wm0, int wn0, int lane_, bool has_next, int nwm0, int nwn0) {
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
    }
    auto to_scratch = [&](int i, int hc) {          // rows 16 hc .. 16 hc + 15 of 32-row block i: registers r with (r >> 2) in {2 hc, 2 hc + 1}
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = hc * 8; r < hc * 8 + 8; ++r) {
                const float a = acc[i][j][r];
                sc[((r & 3) + 8 * ((r >> 2) - hc * 2) + 4 * lhalf) * COLS + j * 32 + lrow] = sizeof(T) == 1 ? fmaf(a, sj[j], bj[j]) : a + bj[j];
            }
    };
    if constexpr (EP == 3) {
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
                for (int cb = 0; cb < 3; ++cb) v[cb] = *(const f32x4*)(sc + (t * 8 + rsub) * COLS + cb * 32 + c4 * 4);
                if (gr < M) {
                    float* op = (float*)p.out + (int64_t)gr * p.ldo + wn0 + c4 * 4;
#pragma unroll
                    for (int cb = 0; cb < 3; ++cb) {
                        *(f32x4*)(op + cb * 32) = v[cb];
                    }
                }
            }
        }
    } else if constexpr (EP == 1 && NS == 1 && sizeof(T) == 2 && OUTK == OUT_OPERAND) {
        // Plain bf16 / f16 result (optional GELU): rounded to 16 bits BEFORE the LDS round trip, two adjacent rows per dword
        // (registers r, r + 1 of a C fragment are rows 2 k, 2 k + 1 of the same column), so the transposition moves half the bytes:
        // per 32-row block 24 ds_write_b32 + 6 ds_read_b128 per lane instead of 48 + 12.  A lane then owns 8 consecutive columns of
        // one row PAIR, splits the dwords with two byte permutes each and stores two 16-byte row pieces; 12 lanes cover a row
        // (192-byte runs).  Same values, same roundings as the fp32 staging (the conversion is the separate step of store_cvt).
        unsigned* sp = (unsigned*)sc;
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
                    sp[prow * COLS + j * 32 + lrow] = __builtin_bit_cast(unsigned, pr);
                }
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int t = u * 64 + lane, prow = t / 12, cg = t % 12;
                const u128 d0 = *(const u128*)(sp + prow * COLS + cg * 8), d1 = *(const u128*)(sp + prow * COLS + cg * 8 + 4);
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
                const float* src = sc + rsub * COLS + cb * 32 + c8 * 8;
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

template <class T, int NS, int BK, int WM, int WN, int MI, int NI, int NPW, int EP, int OUTK, bool GELU>
__global__ __launch_bounds__((WM * WN + NPW) * 64, 1) void gemm_kernel_pc(const bd_gemm_args p) {
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    typedef typename Op16<T>::vec8 frag_t;
    constexpr int ESZ = OpGeom<T>::ESZ, KSTEP = OpGeom<T>::KSTEP, CPF = OpGeom<T>::CPF;
    constexpr int NCW = WM * WN;                      // consumer waves; waves NCW .. NCW + NPW - 1 are producers
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32;
    constexpr int ROWB = BK * ESZ;
    constexpr int CH = ROWB / 16;
    constexpr int RPP = 64 / CH;
    constexpr int A_BYTES = TBM * ROWB, W_BYTES = TBN * ROWB;
    constexpr int PA = A_BYTES / 1024, PW = W_BYTES / 1024;       // 1-KiB DMA pieces per plane per slab
    constexpr int STAGE_BYTES = (A_BYTES + W_BYTES) * NS;
    constexpr int KS = BK / KSTEP;
    static_assert(KS >= 1 && CH * 16 == ROWB && (CH == 4 || CH == 8) && A_BYTES % 1024 == 0 && W_BYTES % 1024 == 0, "slab geometry");
    static_assert(PA % NPW == 0 && PW % NPW == 0, "pieces split evenly over the producer waves");
    constexpr int SR = (EP == 0 && NCW * 32 * NI * 32 * 4 <= STAGE_BYTES) ? 32 : 16;      // LDS-staged epilogue: scratch rows per pass
    static_assert(NCW * SR * NI * 32 * 4 <= STAGE_BYTES, "the epilogue scratch must fit one stage");
    static_assert(EP == 0 || (MI == 2 && NI == 3 && SR == 16 && TBN <= 256), "pc_epilogue is written for 96-column wave tiles");
    // side buffer behind the ring: per-column vectors of the current / next tile (2 x [bias 1 KiB | weight scale 1 KiB]) and
    // the q / k RMSNorm weights (2 x 1 KiB)
    constexpr int AUX_COLP = 2 * STAGE_BYTES, AUX_RMS = AUX_COLP + 4096, AUX_BYTES = EP == 0 ? 0 : 6144;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE_BYTES + AUX_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, N = p.N;
    const int nk = p.K / BK;

    // static tile list: XCD x (= blockIdx % 8) owns a contiguous run of logical tile ids (grouped raster, as in
    // tile_coords_t); its workgroups walk that run with stride = workgroups on the XCD, so the CUs of an XCD always work on
    // consecutive tiles (a compact patch of the output: shared A panels / W tiles stay in that XCD's L2).
    const int tilesM = (M + TBM - 1) / TBM, tilesN = (N + TBN - 1) / TBN, nt = tilesM * tilesN;
    const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7;
    const int tq = nt >> 3, tr = nt & 7;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_end = t_begin + tq + (xcd < tr ? 1 : 0);
    const int stride = (nwg - xcd + 7) >> 3;          // workgroups living on this XCD
    constexpr int GROUP_M = TBM >= 256 ? 4 : 8;
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int per_group = GROUP_M * tilesN;
        const int g = t / per_group, in_g = t % per_group;
        const int gm0 = g * GROUP_M;
        const int gh = (tilesM - gm0) < GROUP_M ? (tilesM - gm0) : GROUP_M;
        m0 = (gm0 + in_g % gh) * TBM;
        n0 = (in_g / gh) * TBN;
    };
    const unsigned lds_off = lds_offset_of(lds);

    if (wid >= NCW) {
        // ------------------------------------------------------------------ producers (wave pw takes pieces pw, pw + NPW, ...)
        const int pw = wid - NCW;
        // Per-lane byte offsets of the 1-KiB pieces inside a tile are the same for every tile (piece j = tile rows
        // j*RPP .. j*RPP+RPP-1, source-side swizzled chunk); the tile origin, the K position and the plane go into the
        // scalar base.  Rows past the edge of the tensor are clamped with one v_min against the tile's last valid byte
        // (they load in-bounds garbage whose results the epilogue discards).
        const unsigned lda_b = (unsigned)(p.lda * ESZ), ldw_b = (unsigned)(p.ldw * ESZ);
        unsigned offA[PA / NPW], offW[PW / NPW];
#pragma unroll
        for (int j = 0; j < PA / NPW; ++j) {
            const int row = (j * NPW + pw) * RPP + lane / CH;
            offA[j] = (unsigned)row * lda_b + swz_chunk<CH>(row, lane % CH) * 16;
        }
#pragma unroll
        for (int j = 0; j < PW / NPW; ++j) {
            const int row = (j * NPW + pw) * RPP + lane / CH;
            offW[j] = (unsigned)row * ldw_b + swz_chunk<CH>(row, lane % CH) * 16;
        }
        const int64_t a_plane = p.a_plane * ESZ, w_plane = p.w_plane * ESZ;
        auto issue = [&](int stage, int m0, int n0, int kt) {
            const int rows_a = (M - m0) < TBM ? (M - m0) : TBM, rows_w = (N - n0) < TBN ? (N - n0) : TBN;
            const unsigned lim_a = (unsigned)(rows_a - 1) * lda_b + (CH - 1) * 16, lim_w = (unsigned)(rows_w - 1) * ldw_b + (CH - 1) * 16;
            const unsigned st = lds_off + stage * STAGE_BYTES + pw * 1024;
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                const unsigned char* ba = (const unsigned char*)p.A + (int64_t)m0 * lda_b + (int64_t)kt * ROWB + s2 * a_plane;
                const unsigned char* bw = (const unsigned char*)p.W + (int64_t)n0 * ldw_b + (int64_t)kt * ROWB + s2 * w_plane;
#pragma unroll
                for (int j = 0; j < PA / NPW; ++j)
                    glds16_s(offA[j] < lim_a ? offA[j] : lim_a, ba, st + s2 * A_BYTES + j * NPW * 1024);
#pragma unroll
                for (int j = 0; j < PW / NPW; ++j)
                    glds16_s(offW[j] < lim_w ? offW[j] : lim_w, bw, st + NS * A_BYTES + s2 * W_BYTES + j * NPW * 1024);
            }
        };
#ifdef BD_GEMM_PROBE
        unsigned probe_ts = 0;
#endif
        // Issue cursor: slab number `ig` of this workgroup's slab sequence (all its tiles, K-slab by K-slab) is the next one
        // to fetch, into stage ig & 1.  Consumers are released into slab g by barrier B(g); after B(g) the stage of slab g-1
        // is free, so slab g+1 may be fetched.  At a tile boundary the consumers signal "done with the tile's last slab" with
        // one extra barrier X before their epilogue, whose scratch is that slab's stage: the next tile's FIRST slab (fetched after
        // B of the last slab, into the other stage) lands while the epilogue runs; its second slab follows B of the first, when
        // the scratch is free again.
        int ig = 0, it = t_begin + (bid >> 3), ikt = 0, im0 = 0, in0 = 0, itn = 0;
        if (it < t_end) tile_origin(it, im0, in0);
        if constexpr (EP == 2) {
            if (pw == 1) {        // q / k RMSNorm weights (96 floats each), once
                const unsigned off = (unsigned)lane * 16 < 368u ? (unsigned)lane * 16 : 368u;
                glds16_s(off, (const unsigned char*)p.rms_wq, lds_off + AUX_RMS);
                glds16_s(off, (const unsigned char*)p.rms_wk, lds_off + AUX_RMS + 1024);
            }
        }
        auto issue_next = [&]() {
            if (it >= t_end) return;
            if constexpr (EP != 0) {
                if (ikt == 0) {   // the tile's per-column vectors ride with its first slab (N % TBN == 0: TBN floats are in bounds)
                    if (pw == 0) {
                        const unsigned off = (unsigned)lane * 16 < (unsigned)(TBN * 4 - 16) ? (unsigned)lane * 16 : (unsigned)(TBN * 4 - 16);
                        if (p.bias) glds16_s(off, (const unsigned char*)(p.bias + in0), lds_off + AUX_COLP + (itn & 1) * 2048);
                        if (sizeof(T) == 1 && p.wscale) glds16_s(off, (const unsigned char*)(p.wscale + in0), lds_off + AUX_COLP + (itn & 1) * 2048 + 1024);
                    }
                    ++itn;
                }
            }
            issue(ig & 1, im0, in0, ikt);
            ++ig;
            if (++ikt == nk) {
                ikt = 0;
                it += stride;
                if (it < t_end) tile_origin(it, im0, in0);
            }
        };
        issue_next();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int g = 0;
        for (int t = t_begin + (bid >> 3); t < t_end; t += stride) {
            for (int kt = 0; kt < nk; ++kt) {
                BD_PROBE_IF(g < 20, g * 3)
                pc_barrier();                          // B(g): slab g landed; slab g-1 is dead
                BD_PROBE_IF(g < 20, g * 3 + 1)
                if (ig == g + 1) issue_next();         // slab g+1 (unless it was fetched at the tile boundary already)
                BD_PROBE_IF(g < 20, g * 3 + 2)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ++g;
            }
            pc_barrier();                              // X: every consumer is done with the tile's last slab
        }
#ifdef BD_GEMM_PROBE
        if (bd_probe_buf && blockIdx.x < 1024) bd_probe_buf[((size_t)blockIdx.x * 16 + wid) * 64 + lane] = probe_ts;
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wid / WN, wn = wid % WN;
    const int lrow = lane & 31, lhalf = lane >> 5;
#ifdef BD_GEMM_PROBE
    unsigned probe_ts = 0;
#endif
    BD_PROBE(58) BD_PROBE_RT(56)
    int g = 0;       // (the host launches this kernel only when the wide, 16-byte epilogue applies: wide_epilogue_ok)
    int ti = 0;      // tiles done by this workgroup (parity = side-buffer slot of the tile's column vectors)
    f32x16 acc[MI][NI];
    if constexpr (EP != 0) {      // first tile; later tiles are initialised inside the previous tile's epilogue
        int m0, n0;
        tile_origin(t_begin + (bid >> 3), m0, n0);
        acc_init<MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
    }
    for (int t = t_begin + (bid >> 3); t < t_end; t += stride, ++ti) {
        int m0, n0;
        tile_origin(t, m0, n0);
        if constexpr (EP == 0) acc_init<MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
        for (int kt = 0; kt < nk; ++kt, ++g) {
            BD_PROBE_IF(g < 20, g * 3)
            pc_barrier();                              // B(kt)
            BD_PROBE_IF(g < 20, g * 3 + 1)
            const unsigned char* base = lds + (g & 1) * STAGE_BYTES;
            constexpr int FB = NS == 1 ? 2 : 1;
            frag_t a[FB][NS][MI], b[FB][NS][NI];
#define LOAD_ONE(dst, ptr, row, ks)                                                                           \
            {                                                                                                 \
                const int c0_ = ((ks) * 2 + lhalf) * CPF;                                                     \
                if constexpr (CPF == 1) {                                                                     \
                    dst = __builtin_bit_cast(frag_t, *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_) << 4))); \
                } else {                                                                                      \
                    const u128 lo_ = *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_) << 4)); \
                    const u128 hi_ = *(const u128*)((ptr) + (row) * ROWB + (swz_chunk<CH>((row), c0_ + 1) << 4)); \
                    dst = (frag_t){(int)lo_[0], (int)lo_[1], (int)lo_[2], (int)lo_[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]}; \
                }                                                                                             \
            }
#define LOAD_FRAGS(ks, slot)                                                                                  \
            _Pragma("unroll") for (int s2 = 0; s2 < NS; ++s2) {                                               \
                _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                \
                    LOAD_ONE(a[slot][s2][i], base + s2 * A_BYTES, wm * (MI * 32) + i * 32 + lrow, ks)         \
                _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                \
                    LOAD_ONE(b[slot][s2][j], base + NS * A_BYTES + s2 * W_BYTES, wn * (NI * 32) + j * 32 + lrow, ks) \
            }
            if constexpr (NS == 1) {
                // Software pipeline inside the slab, order PINNED (sched_barrier(0) after every MFMA / ds_read pair): the
                // fragment reads of k-step ks+1 are interleaved one-for-one with the MFMAs of k-step ks, so a consumer wave
                // keeps its matrix pipe fed on its own.  (With producers doing the DMA, both consumer waves of a SIMD leave the
                // slab barrier at the same instant; left to itself hipcc re-associates the unrolled k-steps into chains of
                // dependent MFMAs on one accumulator, each behind an lgkmcnt(0) -- 3100-3600 cycles per slab for 2048 cycles of
                // matrix work per SIMD, profiles/r2_gemm_phase_probe.md.)  MFMA q of a k-step works on tile (q % MI, q / MI);
                // reads go a0, b0, a1, b1, ... so that the operands the next k-step needs first arrive first; hipcc keeps the
                // lgkmcnt waits counted (its own ds_reads, in order).
                constexpr int NRD = MI + NI, NMM = MI * NI;
                auto load_q = [&](int slot, int ks, int q) {
                    const int which = (q < 2 * (MI < NI ? MI : NI)) ? (q & 1) : (MI > NI ? 0 : 1);   // 0: an A fragment, 1: a W fragment
                    const int idx = (q < 2 * (MI < NI ? MI : NI)) ? (q >> 1) : (q - (MI < NI ? MI : NI));
                    if (which == 0) LOAD_ONE(a[slot][0][idx], base, wm * (MI * 32) + idx * 32 + lrow, ks)
                    else LOAD_ONE(b[slot][0][idx], base + NS * A_BYTES, wn * (NI * 32) + idx * 32 + lrow, ks)
                };
#pragma unroll
                for (int q = 0; q < NRD; ++q) load_q(0, 0, q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int cur = ks & 1;
#pragma unroll
                    for (int q = 0; q < NMM; ++q) {
                        const int i = q % MI, j = q / MI;
                        acc[i][j] = Op16<T>::mfma(a[cur][0][i], b[cur][0][j], acc[i][j]);
                        if (ks + 1 < KS && q < NRD) load_q(cur ^ 1, ks + 1, q);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                // Split-bf16 (NS == 2): per k-step and output tile three MFMAs -- lo(A) hi(W), hi(A) lo(W), hi(A) hi(W) -- on ten
                // fragments (A hi/lo x MI, W hi/lo x NI = 40 VGPRs).  Order pinned so that registers rotate without a second
                // full fragment set (96 accumulator + 60 fragment registers fit the 168-VGPR budget):
                //   phase 1  lo(A) * hi(W)   -- afterwards the lo(A) registers are dead: the NEXT k-step's lo(A) loads go there
                //   phase 2  hi(A) * lo(W)   -- lo(A) loads in flight; afterwards lo(W) is dead: next lo(W) loads go there
                //   phase 3  hi(A) * hi(W)   -- next lo(W), then next hi(W), hi(A) loads (into the spare hi set) in flight
                // so one ds_read rides under almost every MFMA and no wave depends on its SIMD partner to cover its reads.
                constexpr int NMM = MI * NI;
                frag_t ah[2][MI], wh[2][NI], al[MI], wl[NI];
                const unsigned char* wbase = base + NS * A_BYTES;
#define LD_A(dst, plane, idx, ks) LOAD_ONE(dst, base + (plane) * A_BYTES, wm * (MI * 32) + (idx) * 32 + lrow, ks)
#define LD_W(dst, plane, idx, ks) LOAD_ONE(dst, wbase + (plane) * W_BYTES, wn * (NI * 32) + (idx) * 32 + lrow, ks)
#pragma unroll
                for (int j = 0; j < NI; ++j) LD_W(wh[0][j], 0, j, 0)
#pragma unroll
                for (int i = 0; i < MI; ++i) LD_A(al[i], 1, i, 0)
#pragma unroll
                for (int i = 0; i < MI; ++i) LD_A(ah[0][i], 0, i, 0)
#pragma unroll
                for (int j = 0; j < NI; ++j) LD_W(wl[j], 1, j, 0)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int cur = ks & 1, nxt = cur ^ 1;
                    const bool more = ks + 1 < KS;
                    // phase 1: lo(A) * hi(W)
#pragma unroll
                    for (int q = 0; q < NMM; ++q) {
                        const int i = q % MI, j = q / MI;
                        acc[i][j] = Op16<T>::mfma(al[i], wh[cur][j], acc[i][j]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // phase 2: hi(A) * lo(W), next lo(A) loads riding under the first MFMAs
#pragma unroll
                    for (int q = 0; q < NMM; ++q) {
                        const int i = q % MI, j = q / MI;
                        acc[i][j] = Op16<T>::mfma(ah[cur][i], wl[j], acc[i][j]);
                        if (more && q < MI) LD_A(al[q], 1, q, ks + 1)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // phase 3: hi(A) * hi(W), next lo(W), hi(W), hi(A) loads under it
#pragma unroll
                    for (int q = 0; q < NMM; ++q) {
                        const int i = q % MI, j = q / MI;
                        acc[i][j] = Op16<T>::mfma(ah[cur][i], wh[cur][j], acc[i][j]);
                        if (more) {
                            if (q < NI) LD_W(wl[q], 1, q, ks + 1)
                            else if (q < 2 * NI) LD_W(wh[nxt][q - NI], 0, q - NI, ks + 1)
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (more) {
#pragma unroll
                        for (int i = 0; i < MI; ++i) LD_A(ah[nxt][i], 0, i, ks + 1)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#undef LD_A
#undef LD_W
            }
#undef LOAD_FRAGS
#undef LOAD_ONE
        }
        BD_PROBE_IF(g == nk, 60)
        pc_barrier();                                  // X: this wave is done reading the tile's last slab
        BD_PROBE_IF(g == nk, 61)
        {   // LDS-staged epilogue; its scratch is the stage the tile's last slab lived in (free after X)
            unsigned char* scratch = lds + ((g - 1) & 1) * STAGE_BYTES + wid * (SR * NI * 32 * 4);
            if constexpr (EP == 0) {
                gemm_epilogue_lds<T, NS, MI, NI, SR, true>(p, acc, scratch, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
            } else {
                const bool has_next = t + stride < t_end;
                int nm0 = 0, nn0 = 0;
                if (has_next) tile_origin(t + stride, nm0, nn0);
                pc_epilogue<T, NS, EP, OUTK, GELU, MI>(p, acc, (float*)scratch, (const float*)(lds + AUX_COLP + (ti & 1) * 2048),
                                                  (const float*)(lds + AUX_COLP + ((ti + 1) & 1) * 2048), (const float*)(lds + AUX_RMS), wn * (NI * 32), m0 + wm * (MI * 32), n0 + wn * (NI * 32),
                                                  lane, has_next, nm0 + wm * (MI * 32), nn0 + wn * (NI * 32));
            }
        }
        BD_PROBE_IF(g == nk, 62)
    }
    BD_PROBE(59) BD_PROBE_RT(57)
#ifdef BD_GEMM_PROBE
    if (bd_probe_buf && blockIdx.x < 1024) bd_probe_buf[((size_t)blockIdx.x * 16 + wid) * 64 + lane] = probe_ts;
#endif
}

// the wide (LDS-staged, 16-byte) epilogue needs 16-byte aligned rows: N % 8 == 0 and aligned leading dimensions / pointers
inline bool wide_epilogue_ok(const bd_gemm_args& p, int ns) {
    return (p.N % 8 == 0) && (p.ldo % 8 == 0) && (((uintptr_t)p.out & 15) == 0) &&
           (!p.resid || ((p.ldr % 4 == 0) && ((uintptr_t)p.resid & 15) == 0)) &&
           (!p.bias || ((uintptr_t)p.bias & 15) == 0) && (!p.addtab || ((uintptr_t)p.addtab & 15) == 0) &&
           (!p.wscale || ((uintptr_t)p.wscale & 15) == 0) && (p.out_f32 || ns == 1 || (p.out_plane % 8 == 0));
}

// ------------------------------------------------------------------------------------------------
// BD_PREC_F16C8 GEMM (bd_common.h: f16 hi pass + one e4m3 correction pass; 3 bytes per element through LDS-DMA).
// Same producer / consumer persistent structure and 256 x 192 tile as gemm_kernel_pc, with
//   * a stage of four sub-planes per 32-deep slab: A hi (256 x 64 B), W hi (192 x 64 B), A lo8 (256 x 32 B), W lo8 (192 x 32 B)
//     = 42 KiB, so THREE stages fit next to nothing else (the LDS-staged epilogue's 48 KiB scratch overlays stage 2, which is
//     where every tile's last slab lives when K / 32 is a multiple of 3 -- K = 768 and 3072 -- and is free while the next
//     tile's first two slabs land in stages 0 and 1).  With three stages the producers run TWO slabs ahead and wait with a
//     counted vmcnt: the ~1100-cycle issue-to-landing latency of a slab, which bounds the 2-stage kernels at ~2800 cycles per
//     slab in this operand class, is off the critical path.  NSTAGE = 2 (scratch separate) serves the other K.
//   * the e4m3 image q8 of an f16 fragment derived in registers (v_cvt_scalef32_pk_fp8_f16, 4 per fragment), so the
//     correction pass costs one extra ds_read_b128 per 32-row fragment pair instead of two.
template <int NSTAGE, int EP, int OUTK, bool GELU>
__global__ __launch_bounds__(768, 1) void gemm_kernel_pc_f16c8(const bd_gemm_args p) {
    constexpr int WM = 4, WN = 2, MI = 2, NI = 3, NPW = 4, NCW = WM * WN;
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32, BK = 32;
    constexpr int A0 = TBM * 64, W0 = TBN * 64, A1 = TBM * 32, W1 = TBN * 64;     // W's e4m3 plane carries q8 AND lo8 (weights: packed once)
    constexpr int OFF_W0 = A0, OFF_A1 = A0 + W0, OFF_W1 = A0 + W0 + A1, STAGE = A0 + W0 + A1 + W1;
    constexpr int SR = 16, SCRATCH = NCW * SR * NI * 32 * 4;
    static_assert(SCRATCH == STAGE, "the epilogue scratch overlays stage 2 exactly");
    // side buffer behind the ring + scratch (gemm_kernel_pc): per-column vectors of the current / next tile, q / k RMSNorm weights
    constexpr int AUX_COLP = 2 * STAGE + SCRATCH, AUX_RMS = AUX_COLP + 4096, AUX_BYTES = EP == 0 ? 0 : 6144;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE + SCRATCH + AUX_BYTES];   // stages at 0, STAGE, 2*STAGE; scratch at 2*STAGE

    bd_saturating_conversions();      // q8 images (K loop) and F16C8 / f16 results (epilogue) saturate instead of turning NaN / inf
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = p.M, N = p.N;
    const int nk = p.K / BK;
    const int tilesM = (M + TBM - 1) / TBM, tilesN = (N + TBN - 1) / TBN, nt = tilesM * tilesN;
    const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7;
    const int tq = nt >> 3, tr = nt & 7;
    const int t_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int t_end = t_begin + tq + (xcd < tr ? 1 : 0);
    const int stride = (nwg - xcd + 7) >> 3;
    constexpr int GROUP_M = 4;
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int per_group = GROUP_M * tilesN;
        const int g = t / per_group, in_g = t % per_group;
        const int gm0 = g * GROUP_M;
        const int gh = (tilesM - gm0) < GROUP_M ? (tilesM - gm0) : GROUP_M;
        m0 = (gm0 + in_g % gh) * TBM;
        n0 = (in_g / gh) * TBN;
    };
    const unsigned lds_off = lds_offset_of(lds);

    if (wid >= NCW) {
        // ------------------------------------------------------------------ producers
        const int pw = wid - NCW;
        const unsigned lda0 = (unsigned)(p.lda * 2), ldw0 = (unsigned)(p.ldw * 2), lda1 = (unsigned)p.lda, ldw1 = (unsigned)(p.ldw * 2);
        unsigned oA0[4], oW0[3], oA1[2], oW1[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int row = (pw + 4 * i) * 16 + lane / 4; oA0[i] = row * lda0 + swz_chunk<4>(row, lane % 4) * 16; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int row = (pw + 4 * i) * 16 + lane / 4; oW0[i] = row * ldw0 + swz_chunk<4>(row, lane % 4) * 16; }
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int row = (pw + 4 * i) * 32 + lane / 2; oA1[i] = row * lda1 + swz_chunk<2>(row, lane % 2) * 16; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int row = (pw + 4 * i) * 16 + lane / 4; oW1[i] = row * ldw1 + swz_chunk<4>(row, lane % 4) * 16; }
        const unsigned char* pA0 = (const unsigned char*)p.A;
        const unsigned char* pW0 = (const unsigned char*)p.W;
        const unsigned char* pA1 = pA0 + p.a_plane * 2;
        const unsigned char* pW1 = pW0 + p.w_plane * 2;
        auto issue = [&](int stage, int m0, int n0, int kt) {
            const int rows_a = (M - m0) < TBM ? (M - m0) : TBM, rows_w = (N - n0) < TBN ? (N - n0) : TBN;
            const unsigned la0 = (unsigned)(rows_a - 1) * lda0 + 48, lw0 = (unsigned)(rows_w - 1) * ldw0 + 48;
            const unsigned la1 = (unsigned)(rows_a - 1) * lda1 + 16, lw1 = (unsigned)(rows_w - 1) * ldw1 + 48;
            const unsigned st = lds_off + stage * STAGE + pw * 1024;
            const unsigned char* ba0 = pA0 + (int64_t)m0 * lda0 + (int64_t)kt * 64;
            const unsigned char* bw0 = pW0 + (int64_t)n0 * ldw0 + (int64_t)kt * 64;
            const unsigned char* ba1 = pA1 + (int64_t)m0 * lda1 + (int64_t)kt * 32;
            const unsigned char* bw1 = pW1 + (int64_t)n0 * ldw1 + (int64_t)kt * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_s(oA0[i] < la0 ? oA0[i] : la0, ba0, st + i * 4096);
#pragma unroll
            for (int i = 0; i < 3; ++i) glds16_s(oW0[i] < lw0 ? oW0[i] : lw0, bw0, st + OFF_W0 + i * 4096);
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16_s(oA1[i] < la1 ? oA1[i] : la1, ba1, st + OFF_A1 + i * 4096);
#pragma unroll
            for (int i = 0; i < 3; ++i) glds16_s(oW1[i] < lw1 ? oW1[i] : lw1, bw1, st + OFF_W1 + i * 4096);
        };
        // issue cursor over this workgroup's slab sequence (all tiles, slab by slab); slab number ig goes to stage ig % NSTAGE
        int ig = 0, ist = 0, it = t_begin + (bid >> 3), ikt = 0, im0 = 0, in0 = 0, itn = 0;
        if (it < t_end) tile_origin(it, im0, in0);
        if constexpr (EP == 2) {
            if (pw == 1) {        // q / k RMSNorm weights (96 floats each), once
                const unsigned off = (unsigned)lane * 16 < 368u ? (unsigned)lane * 16 : 368u;
                glds16_s(off, (const unsigned char*)p.rms_wq, lds_off + AUX_RMS);
                glds16_s(off, (const unsigned char*)p.rms_wk, lds_off + AUX_RMS + 1024);
            }
        }
        auto issue_next = [&]() {
            if (it >= t_end) return;
            if constexpr (EP != 0) {
                // the tile's bias vector rides in FRONT of its first slab (the counted vmcnt below covers everything but the
                // most recent slab's 12 pieces); the host sends K >= 128 here, so a slot is rewritten only after its tile is done
                if (ikt == 0) {
                    if (pw == 0 && p.bias) {
                        const unsigned off = (unsigned)lane * 16 < (unsigned)(TBN * 4 - 16) ? (unsigned)lane * 16 : (unsigned)(TBN * 4 - 16);
                        glds16_s(off, (const unsigned char*)(p.bias + in0), lds_off + AUX_COLP + (itn & 1) * 2048);
                    }
                    ++itn;
                }
            }
            issue(ist, im0, in0, ikt);
            ++ig;
            ist = ist + 1 == NSTAGE ? 0 : ist + 1;
            if (++ikt == nk) {
                ikt = 0;
                it += stride;
                if (it < t_end) tile_origin(it, im0, in0);
            }
        };
        // wait until at most `slabs` of this wave's most recent slab fetches are still in flight (12 pieces each)
        auto wait_landed = [&](int slabs) {
            if (slabs <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        };
#ifdef BD_GEMM_PROBE
        unsigned probe_ts = 0;
#endif
#pragma unroll
        for (int i = 0; i < NSTAGE - 1; ++i) issue_next();
        int g = 0;
        for (int t = t_begin + (bid >> 3); t < t_end; t += stride) {
            for (int kt = 0; kt < nk; ++kt) {
                BD_PROBE_IF(g < 20, g * 3 + 2)
                wait_landed(ig - g - 1);                  // slab g has landed (later slabs may still fly)
                BD_PROBE_IF(g < 20, g * 3)
                pc_barrier();                             // B(g): releases the consumers into slab g; slab g-1 is dead
                BD_PROBE_IF(g < 20, g * 3 + 1)
                if (ig <= g + NSTAGE - 1) issue_next();   // refill the stage slab g-1 lived in
                ++g;
            }
            pc_barrier();                                 // X: consumers are done with the tile's last slab (scratch = stage 2 ..)
        }
#ifdef BD_GEMM_PROBE
        if (bd_probe_buf && blockIdx.x < 1024) bd_probe_buf[((size_t)blockIdx.x * 16 + wid) * 64 + lane] = probe_ts;
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = wid / WN, wn = wid % WN;
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int scale_a = 127, scale_w = 127 - (p.w_qexp + BD_F16C8_D);           // E8M0: the cross terms carry 2^(E + D)
    int g_st = 0;
#ifdef BD_GEMM_PROBE
    unsigned probe_ts = 0;
    int g = 0;
#endif
    BD_PROBE(58) BD_PROBE_RT(56)
    int ti = 0;
    f32x16 acc[MI][NI];
    if constexpr (EP != 0) {      // first tile; later tiles are initialised inside the previous tile's epilogue
        int m0, n0;
        tile_origin(t_begin + (bid >> 3), m0, n0);
        acc_init<MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
    }
    for (int t = t_begin + (bid >> 3); t < t_end; t += stride, ++ti) {
        int m0, n0;
        tile_origin(t, m0, n0);
        if constexpr (EP == 0) acc_init<MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
        for (int kt = 0; kt < nk; ++kt) {
            BD_PROBE_IF(g < 20, g * 3)
            pc_barrier();                                 // B
            BD_PROBE_IF(g < 20, g * 3 + 1)
#ifdef BD_GEMM_PROBE
            ++g;
#endif
            const unsigned char* base = lds + g_st * STAGE;
            g_st = g_st + 1 == NSTAGE ? 0 : g_st + 1;
            f16x8 ah[2][MI], wh[2][NI];
            u128 al[MI];
            i32x8 w8[NI];
#define LD16(dst, ptr, row, ks) dst = __builtin_bit_cast(f16x8, *(const u128*)((ptr) + (row) * 64 + (swz_chunk<4>((row), (ks) * 2 + lhalf) << 4)));
#define LDLO(dst, ptr, row) dst = *(const u128*)((ptr) + (row) * 32 + (swz_chunk<2>((row), lhalf) << 4));
            // W's e4m3 operand straight from LDS: 32 bytes = [q8 x 16 | lo8 x 16] of this lane half
#define LDW8(dst, ptr, row)                                                                                    \
            {                                                                                                     \
                const u128 lo_ = *(const u128*)((ptr) + (row) * 64 + (swz_chunk<4>((row), 2 * lhalf) << 4));        \
                const u128 hi_ = *(const u128*)((ptr) + (row) * 64 + (swz_chunk<4>((row), 2 * lhalf + 1) << 4));    \
                dst = (i32x8){(int)lo_[0], (int)lo_[1], (int)lo_[2], (int)lo_[3], (int)hi_[0], (int)hi_[1], (int)hi_[2], (int)hi_[3]}; \
            }
            // e4m3 image of one f16 A fragment (8 values of a k-step), 2 dwords; scale 1.0.  The wave runs with MODE.FP16_OVFL set
            // (bd_saturating_conversions): the conversion clamps to +-448, so an activation beyond 448 keeps its full value in the f16
            // pass and only its q_A . lo_W correction term is computed from 448 (bd_common.h, RANGE).
            // (pairs built element-wise: __builtin_bit_cast of a vector ELEMENT to a 2 x f16 vector is miscompiled by hipcc 7.2
            // here -- every conversion then reads the first dword)
#define Q8P(f, a, b) ((h2_){f[a], f[b]})
#define Q8H(d0, d1, f)                                                                                        \
            {                                                                                                     \
                typedef _Float16 h2_ __attribute__((ext_vector_type(2)));                                         \
                typedef short s2_ __attribute__((ext_vector_type(2)));                                            \
                s2_ a_ = {0, 0}, b_ = {0, 0};                                                                     \
                a_ = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(a_, Q8P(f, 0, 1), 1.0f, false);                     \
                a_ = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(a_, Q8P(f, 2, 3), 1.0f, true);                      \
                b_ = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(b_, Q8P(f, 4, 5), 1.0f, false);                     \
                b_ = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(b_, Q8P(f, 6, 7), 1.0f, true);                      \
                d0 = __builtin_bit_cast(unsigned, a_);                                                            \
                d1 = __builtin_bit_cast(unsigned, b_);                                                            \
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) LD16(wh[0][j], base + OFF_W0, wn * (NI * 32) + j * 32 + lrow, 0)
#pragma unroll
            for (int i = 0; i < MI; ++i) LD16(ah[0][i], base, wm * (MI * 32) + i * 32 + lrow, 0)
            __builtin_amdgcn_sched_barrier(0);
            constexpr int NMM = MI * NI;
            static_assert(MI == 2 && NI == 3, "the interleave below is written for 2 x 3 MFMA tiles per wave");
            unsigned qd[MI][4];
#pragma unroll
            for (int q = 0; q < NMM; ++q) {                       // f16, k-step 0
                const int i = q % MI, j = q / MI;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0][i], wh[0][j], acc[i][j], 0, 0, 0);
                if (q < MI) LD16(ah[1][q], base, wm * (MI * 32) + q * 32 + lrow, 1)
                else if (q < MI + NI) LD16(wh[1][q - MI], base + OFF_W0, wn * (NI * 32) + (q - MI) * 32 + lrow, 1)
                if (q == 3) Q8H(qd[0][0], qd[0][1], ah[0][0])
                if (q == 4) Q8H(qd[1][0], qd[1][1], ah[0][1])
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < NMM; ++q) {                       // f16, k-step 1
                const int i = q % MI, j = q / MI;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1][i], wh[1][j], acc[i][j], 0, 0, 0);
                if (q < MI) LDLO(al[q], base + OFF_A1, wm * (MI * 32) + q * 32 + lrow)
                if (q == 2) LDW8(w8[0], base + OFF_W1, wn * (NI * 32) + lrow)
                if (q == 4) LDW8(w8[1], base + OFF_W1, wn * (NI * 32) + 32 + lrow)
                if (q == 2) Q8H(qd[0][2], qd[0][3], ah[1][0])
                if (q == 3) Q8H(qd[1][2], qd[1][3], ah[1][1])
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < NMM; ++q) {                       // e4m3 correction pass: [lo_A | q_A] . [q_W | lo_W]
                const int i = q % MI, j = q / MI;
                const i32x8 a8 = {(int)al[i][0], (int)al[i][1], (int)al[i][2], (int)al[i][3], (int)qd[i][0], (int)qd[i][1], (int)qd[i][2], (int)qd[i][3]};
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, w8[j], acc[i][j], 0, 0, 0, scale_a, 0, scale_w);
                if (q == 0) LDW8(w8[2], base + OFF_W1, wn * (NI * 32) + 64 + lrow)
                __builtin_amdgcn_sched_barrier(0);
            }
#undef LDW8
#undef LD16
#undef LDLO
#undef Q8H
#undef Q8P
        }
        BD_PROBE_IF(g == nk, 60)
        pc_barrier();                                     // X
        BD_PROBE_IF(g == nk, 61)
        unsigned char* scratch = lds + 2 * STAGE + wid * (SR * NI * 32 * 4);
        if constexpr (EP == 0) {
            gemm_epilogue_lds<f16c8, 2, MI, NI, SR, true>(p, acc, scratch, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
        } else {
            const bool has_next = t + stride < t_end;
            int nm0 = 0, nn0 = 0;
            if (has_next) tile_origin(t + stride, nm0, nn0);
            pc_epilogue<f16c8, 2, EP, OUTK, GELU, 2>(p, acc, (float*)scratch, (const float*)(lds + AUX_COLP + (ti & 1) * 2048),
                                                 (const float*)(lds + AUX_COLP + ((ti + 1) & 1) * 2048), (const float*)(lds + AUX_RMS), wn * (NI * 32), m0 + wm * (MI * 32), n0 + wn * (NI * 32),
                                                 lane, has_next, nm0 + wm * (MI * 32), nn0 + wn * (NI * 32));
        }
        BD_PROBE_IF(g == nk, 62)
    }
    BD_PROBE(59) BD_PROBE_RT(57)
#ifdef BD_GEMM_PROBE
    if (bd_probe_buf && blockIdx.x < 1024) bd_probe_buf[((size_t)blockIdx.x * 16 + wid) * 64 + lane] = probe_ts;
#endif
}

// Which epilogue specialisation of gemm_kernel_pc serves these arguments (0 = the generic one): see the kernel's header
template <class T, int NS> int pc_epilogue_kind(const bd_gemm_args& a, int& outk, bool& gelu) {
    outk = a.out_f32;
    gelu = a.act == BD_ACT_GELU;
    if (a.addtab || a.rpg_in > 0) return 0;
    if (a.bias && ((uintptr_t)a.bias & 15)) return 0;
    if (a.out_f32 == OUT_F32) {
        if (gelu || a.rms_wq) return 0;
        if (a.wscale && (sizeof(T) != 1 || ((uintptr_t)a.wscale & 15))) return 0;
        return 3;
    }
    if (a.resid) return 0;
    if (a.wscale && (sizeof(T) != 1 || ((uintptr_t)a.wscale & 15))) return 0;
    // output kinds instantiated per operand class: native everywhere; f16 from split-bf16 (strict f16 attention), bf16 from e4m3
    const bool outk_ok = outk == OUT_OPERAND || (outk == OUT_F16 && NS == 2 && sizeof(T) == 2) || (outk == OUT_BF16 && sizeof(T) == 1) ||
                         (outk == OUT_BF16X2 && NS == 2 && std::is_same<T, _Float16>::value);   // split-f16 QKV feeding split-bf16 attention
    if (!outk_ok) return 0;
    if (a.rms_wq) return gelu ? 0 : 2;
    if (gelu && outk != OUT_OPERAND) return 0;
    return 1;
}

template <class T, int NS, int BK, int WM, int WN, int MI, int NI> void launch_pc(const bd_gemm_args& a, hipStream_t s, int cus) {
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32;
    constexpr int NPW = 4;        // one producer wave per SIMD: a single wave issues one LDS-DMA piece per ~70 cycles, the CU ~23
    const int tiles = ((a.M + TBM - 1) / TBM) * ((a.N + TBN - 1) / TBN);
    const int grid = tiles < cus ? tiles : cus;
    const dim3 g(grid), b((WM * WN + NPW) * 64);
    int outk = 0;
    bool gelu = false;
    const int ep = (a.N % TBN == 0) ? pc_epilogue_kind<T, NS>(a, outk, gelu) : 0;
#define BD_PC_LAUNCH(EP_, OUTK_, GELU_) hipLaunchKernelGGL((gemm_kernel_pc<T, NS, BK, WM, WN, MI, NI, NPW, EP_, OUTK_, GELU_>), g, b, 0, s, a)
    constexpr int ALT = (NS == 2 && sizeof(T) == 2) ? OUT_F16 : (sizeof(T) == 1 ? OUT_BF16 : OUT_OPERAND);   // the one non-native 16-bit kind
    constexpr bool X2 = NS == 2 && std::is_same<T, _Float16>::value;     // split-f16 also emits split-bf16 planes (attention input)
    if (ep == 3) BD_PC_LAUNCH(3, OUT_F32, false);
    else if (ep == 2 && outk == OUT_BF16X2) { if constexpr (X2) BD_PC_LAUNCH(2, OUT_BF16X2, false); }
    else if (ep == 2) { if (outk == OUT_OPERAND) BD_PC_LAUNCH(2, OUT_OPERAND, false); else BD_PC_LAUNCH(2, ALT, false); }
    else if (ep == 1 && gelu && outk == OUT_OPERAND) BD_PC_LAUNCH(1, OUT_OPERAND, true);
    else if (ep == 1 && !gelu && outk == OUT_BF16X2) { if constexpr (X2) BD_PC_LAUNCH(1, OUT_BF16X2, false); }
    else if (ep == 1 && !gelu) { if (outk == OUT_OPERAND) BD_PC_LAUNCH(1, OUT_OPERAND, false); else BD_PC_LAUNCH(1, ALT, false); }
    else BD_PC_LAUNCH(0, OUT_OPERAND, false);
#undef BD_PC_LAUNCH
}

template <class T, int NS, int BK, int WM, int WN, int MI, int NI, int NSTG = 2> void launch_glds(const bd_gemm_args& a, hipStream_t s) {
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32;
    const int tiles = ((a.M + TBM - 1) / TBM) * ((a.N + TBN - 1) / TBN);
    hipLaunchKernelGGL((gemm_kernel_glds<T, NS, BK, WM, WN, MI, NI, NSTG>), dim3(tiles), dim3(WM * WN * 64), 0, s, a);
}

// Compute units of the current device (MI355X: 256; partitioned / harvested parts differ): the tile-choice model counts
// resident workgroup slots per round (one 256x256 workgroup per CU, two 128x128, four 64x64).  Immutable device property,
// looked up once per device.
int cu_count() {
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
template <class T> bd_gemm_args row_slice(const bd_gemm_args& a, int64_t row0, int rows) {
    bd_gemm_args sub = a;
    constexpr int ESZ = OpGeom<T>::ESZ;
    const int osz = a.out_f32 == OUT_F32 ? 4 : (a.out_f32 == OUT_OPERAND ? ESZ : 2);
    sub.A = (const unsigned char*)a.A + row0 * a.lda * ESZ;
    sub.out = (unsigned char*)a.out + row0 * a.ldo * osz;
    if (a.resid) sub.resid = a.resid + row0 * a.ldr;
    sub.M = rows;
    return sub;
}

// does this problem go to the persistent 256 x 192 kernel (every wave tile = 96 consecutive output columns)?
inline bool pc192_possible(const bd_gemm_args& a, int ns, int esz) {
    return wide_epilogue_ok(a, ns) && 256 * a.lda * esz < ((int64_t)1 << 31) && 256 * a.ldw * esz < ((int64_t)1 << 31);
}
inline bool uses_pc192(const bd_gemm_args& a, int ns, int esz, int cus) {
    if (!pc192_possible(a, ns, esz)) return false;
    if (a.rms_wq) return true;       // a fused q/k RMSNorm pins the tile shape for EVERY batch size: a sample's q, k must not
                                     // depend on whether its batch filled the CUs (bit-exact batch independence is tested)
    if (a.N % 192 || a.M < 1024) return false;
    const int64_t t192 = (int64_t)((a.M + 255) / 256) * (a.N / 192);
    // (0.75: DINOv2's N = 768 GEMMs at M = 50112 are 784 tiles = 3.06 rounds, fill 0.77.  One batch at a time the persistent kernel
    // is then +1.8 % on the step against the one-tile kernels' hybrid split; with two batches in flight -- the other batch's kernels
    // run on the CUs this kernel's tail leaves idle -- +5 %: 1221 -> 1282 poses/s same box.)
    return (double)t192 / (double)(((t192 + cus - 1) / cus) * cus) >= 0.75;       // last-round occupancy of the CUs
}
// a fused q/k RMSNorm needs: 16-bit output of a plain Linear, N = 3 x heads x 96, row-identity output map, and N a multiple of
// the 192-column workgroup tile: with an odd head count (N = 288 h, h odd) the last column tile's second wave tile would lie
// past N, and the fused branch stores whole 96-column heads without a column guard
inline bool rms_geometry_ok(const bd_gemm_args& a) {
    return a.rms_wq && a.rms_wk && a.out_f32 != OUT_F32 && a.act == BD_ACT_NONE && !a.resid && !a.addtab && a.rpg_in <= 0 &&
           (a.rms_parts == 0 || a.rms_parts == 2 || a.rms_parts == 3) &&
           a.N % (a.rms_parts == 2 ? 2 : 3) == 0 && (a.N / (a.rms_parts == 2 ? 2 : 3)) % 96 == 0 && a.N % 192 == 0 && (((uintptr_t)a.rms_wq | (uintptr_t)a.rms_wk) & 3) == 0;
}

template <class T, int NS, int BK> int launch(const bd_gemm_args& a, hipStream_t s) {
    const int kCUs = cu_count();
    if (a.rms_wq && !(rms_geometry_ok(a) && pc192_possible(a, NS, OpGeom<T>::ESZ))) return BD_ERR_SHAPE;
    const int slot = bd_trace_open(s, 0, a.M, a.N, a.K);
    constexpr int ESZ_ = OpGeom<T>::ESZ;
    {
        // 256 x 192 tiles (8 consumer waves of 64 x 96: 96 accumulator + 2 x 20 fragment registers fit the 168-VGPR budget
        // of three waves per SIMD with the fragment double-buffering intact) whenever they tile N exactly -- every Linear of
        // both stacks except the head (N = 2304, 3072, 768 are multiples of 192).
        if (uses_pc192(a, NS, ESZ_, kCUs)) {
            launch_pc<T, NS, BK, 4, 2, 2, 3>(a, s, kCUs);
            bd_trace_close(s, slot);
            BD_CHECK_LAUNCH();
            return BD_OK;
        }
    }
    {
        // Tile choice = best estimated efficiency: wave quantisation over the resident slots (256x256: one workgroup
        // per CU; 128x128: two; 64x64: four) times the measured relative mainloop efficiency of the tile
        // (profiles/r1_gemm_experiments.md: 128x128 ~0.87 of 256x256 on the wide shapes; 64x64 ~0.55).
        auto eff = [&](int tm, int tn, int slots, double base) {
            const double tiles = (double)((a.M + tm - 1) / tm) * ((a.N + tn - 1) / tn);
            const double rounds = (double)(((int64_t)tiles + slots - 1) / slots);
            return base * tiles / (rounds * slots);
        };
        const double e256 = a.N >= 1536 ? eff(256, 256, kCUs, 1.0) : 0.0;
        const double e128 = eff(128, 128, 2 * kCUs, 0.87);
        const double e64 = eff(64, 64, 4 * kCUs, 0.55);
        // Narrow outputs (N = 768: proj / fc2) have too few 256x256 tiles for whole rounds (576 tiles = 2.25 rounds) and
        // pay 4.5 -> 5 rounds plus the weaker mainloop with 128x128 tiles.  Hybrid: k FULL rounds of 256x256 tiles over
        // the first rows, the remaining rows as 128x128 tiles -- two launches over disjoint row ranges, no split-K, no
        // cross-workgroup fix-up, bit-identical per-row arithmetic order (the K order of a row does not depend on the
        // tile).  Cost model in units of one 256x256 round; a 128x128 round is 2 tiles per CU at 0.87 efficiency.
        // Only for deep K (fc2: 324 -> 297 us): at K = 768 (proj) the GEMM is bound by its fp32 residual epilogue, which
        // two co-resident 128x128 workgroups overlap better than one 256x256 workgroup (117 vs 125 us).
        int k256 = 0;
        int64_t rows256 = 0;
        double hybrid_cost = 1e30;
        if (a.N % 256 == 0 && a.N < 1536 && a.K >= 2048 && a.rpg_in <= 0 && !a.addtab && a.M >= 1024) {
            const int tn = a.N / 256;
            const int64_t mtiles = (a.M + 255) / 256;
            auto cost128 = [&](int64_t rows) {
                if (rows <= 0) return 0.0;
                const int64_t t = ((rows + 127) / 128) * (a.N / 128);
                const int64_t full = t / (2 * kCUs), rem = t % (2 * kCUs);
                // a partial last round runs with one workgroup on most CUs: faster than a full round
                return full * 0.575 + (rem == 0 ? 0.0 : (rem <= kCUs ? 0.36 : 0.575));
            };
            double best = cost128(a.M);
            for (int k = 1; k <= 64; ++k) {
                const int64_t mt = (int64_t)k * kCUs / tn;
                if (mt > mtiles) break;
                const int64_t r256 = mt * 256 < a.M ? mt * 256 : a.M;
                const double c = k + cost128(a.M - r256);
                if (c < best - 1e-9) { best = c; k256 = k; rows256 = r256; }
            }
            hybrid_cost = best;
        }
        // 256 x 192 tiles (8 waves of 64 x 96): N = 768 in 4 columns of tiles, 192 row tiles x 4 = exactly 3 rounds at
        // M = 49152 (BETR fc2) -- whole rounds of a big tile without any row split.
        bool use192 = false;
        if (a.N % 192 == 0 && a.N < 1536 && a.K >= 2048 && a.M >= 1024) {
            const double e192 = eff(256, 192, kCUs, 0.93);
            const double ehyb = k256 > 0 ? (double)((a.M + 255) / 256) * (a.N / 256) / kCUs / hybrid_cost : e128;
            use192 = e192 > ehyb && e192 > e128;
        }
        if constexpr (NS == 1) {
            if (use192) {
                launch_glds<T, NS, BK, 4, 2, 2, 3>(a, s);
                bd_trace_close(s, slot);
                BD_CHECK_LAUNCH();
                return BD_OK;
            }
        }
        if (k256 > 0) {
            const bd_gemm_args big = row_slice<T>(a, 0, (int)rows256);
            launch_glds<T, NS, BK, 2, 4, 4, 2>(big, s);
            if (rows256 < a.M) {
                const bd_gemm_args rest = row_slice<T>(a, rows256, (int)(a.M - rows256));
                launch_glds<T, NS, BK, 2, 2, 2, 2>(rest, s);
            }
        } else if (e256 >= e128 && e256 >= e64)
            launch_glds<T, NS, BK, 2, 4, 4, 2>(a, s);             // 256 x 256, one tile per workgroup
        else {
            // Sparse launches (small batch: every tile gets its own workgroup slot at once, so the launch lasts one tile's chain of
            // slab latencies): the 4-stage ring form of the same tile (three slabs in flight; 128 / 64 KiB of LDS: one / two per CU).
            const int64_t t128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128), t64 = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
            if (e128 >= e64) {
                if (t128 <= kCUs) launch_glds<T, NS, BK, 2, 2, 2, 2, 4>(a, s);
                else launch_glds<T, NS, BK, 2, 2, 2, 2>(a, s);    // 128 x 128, 2 workgroups / CU
            } else {
                if (t64 <= 2 * kCUs) launch_glds<T, NS, BK, 2, 2, 1, 1, 4>(a, s);
                else launch_glds<T, NS, BK, 2, 2, 1, 1>(a, s);    // 64 x 64 (latency mode: batch 1, M = 1536)
            }
        }
    }
    bd_trace_close(s, slot);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

// F16C8 has its own persistent kernel; every shape goes through it
int launch_f16c8(const bd_gemm_args& a, hipStream_t s) {
    if (!wide_epilogue_ok(a, 2) || 256 * a.lda * 2 >= ((int64_t)1 << 31) || 256 * a.ldw * 2 >= ((int64_t)1 << 31)) return BD_ERR_ALIGN;
    if ((a.K % 32) || (a.lda % 32) || (a.ldw % 32)) return BD_ERR_SHAPE;            // the lo8 planes are laid out in 32-element blocks
    if (a.out_f32 == OUT_OPERAND && (a.ldo % 32)) return BD_ERR_SHAPE;
    if (a.w_qexp + BD_F16C8_D < -100 || a.w_qexp + BD_F16C8_D > 120) return BD_ERR_SHAPE;
    if (a.rms_wq && !rms_geometry_ok(a)) return BD_ERR_SHAPE;
    const int slot = bd_trace_open(s, 0, a.M, a.N, a.K);
    const int cus = cu_count();
    const int tiles = ((a.M + 255) / 256) * ((a.N + 191) / 192);
    const int grid = tiles < cus ? tiles : cus;
    // epilogue specialisation (gemm_kernel_pc's header): native / f16 / split-bf16 16-bit results, fp32 (+ residual)
    int ep = 0, outk = a.out_f32;
    const bool gelu = a.act == BD_ACT_GELU;
    if (!a.addtab && a.rpg_in <= 0 && !a.wscale && a.N % 192 == 0 && a.K >= 128 && !(a.bias && ((uintptr_t)a.bias & 15))) {
        if (a.out_f32 == OUT_F32) ep = (gelu || a.rms_wq) ? 0 : 3;
        else if (!a.resid && (outk == OUT_OPERAND || outk == OUT_F16 || outk == OUT_BF16X2)) ep = a.rms_wq ? (gelu ? 0 : 2) : 1;
    }
    if (ep == 1 && gelu && outk != OUT_OPERAND) ep = 0;
    const bool s3 = (a.K / 32) % 3 == 0;
    const dim3 g(grid), b(768);
#define BD_C8_LAUNCH(EP_, OUTK_, GELU_)                                                                     \
    { if (s3) hipLaunchKernelGGL((gemm_kernel_pc_f16c8<3, EP_, OUTK_, GELU_>), g, b, 0, s, a);                 \
      else hipLaunchKernelGGL((gemm_kernel_pc_f16c8<2, EP_, OUTK_, GELU_>), g, b, 0, s, a); }
    if (ep == 3) BD_C8_LAUNCH(3, OUT_F32, false)
    else if (ep == 2 && outk == OUT_OPERAND) BD_C8_LAUNCH(2, OUT_OPERAND, false)
    else if (ep == 2 && outk == OUT_F16) BD_C8_LAUNCH(2, OUT_F16, false)
    else if (ep == 2) BD_C8_LAUNCH(2, OUT_BF16X2, false)
    else if (ep == 1 && gelu) BD_C8_LAUNCH(1, OUT_OPERAND, true)
    else if (ep == 1 && outk == OUT_OPERAND) BD_C8_LAUNCH(1, OUT_OPERAND, false)
    else if (ep == 1 && outk == OUT_F16) BD_C8_LAUNCH(1, OUT_F16, false)
    else if (ep == 1) BD_C8_LAUNCH(1, OUT_BF16X2, false)
    else BD_C8_LAUNCH(0, OUT_OPERAND, false)
#undef BD_C8_LAUNCH
    bd_trace_close(s, slot);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

}  // namespace

extern "C" int bd_gemm_fuses_qk_rmsnorm(const bd_gemm_args* args, int prec) {
    if (!args || !rms_geometry_ok(*args)) return 0;
    switch (prec) {
        case BD_PREC_BF16: case BD_PREC_F16: return pc192_possible(*args, 1, 2) ? 1 : 0;
        case BD_PREC_BF16X3: case BD_PREC_F16X3: return pc192_possible(*args, 2, 2) ? 1 : 0;
        case BD_PREC_FP8: return pc192_possible(*args, 1, 1) ? 1 : 0;
        case BD_PREC_F16C8: return wide_epilogue_ok(*args, 2) ? 1 : 0;
        default: return 0;
    }
}

extern "C" int bd_gemm(const bd_gemm_args* args, int prec, void* stream) {
    if (!args || !args->A || !args->W || !args->out) return BD_ERR_NULL;
    const bd_gemm_args& a = *args;
    const int kmult = prec == BD_PREC_FP8 ? 128 : 64;
    const int esz = prec == BD_PREC_FP8 ? 1 : 2;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % kmult) != 0) return BD_ERR_SHAPE;
    if (((a.lda * esz) % 16) || ((a.ldw * esz) % 16) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15)) return BD_ERR_ALIGN;
    if ((prec == BD_PREC_BF16X3 || prec == BD_PREC_F16X3 || prec == BD_PREC_F16C8) && ((a.a_plane % 8) || (a.w_plane % 8))) return BD_ERR_ALIGN;
    if (a.addtab && a.tab_rows <= 0) return BD_ERR_SHAPE;
    if (a.out_f32 < 0 || a.out_f32 > 5) return BD_ERR_DTYPE;
    if (a.out_f32 >= 4 && (a.out_plane % 8)) return BD_ERR_ALIGN;
    // the single-plane 16-bit outputs and every fp8-mode output exist only in the wide (16-byte) epilogue
    const bool needs_wide = a.out_f32 >= 2 || (prec == BD_PREC_FP8 && a.out_f32 == 0);
    if (needs_wide && ((a.N % 8) || (a.ldo % 8) || ((uintptr_t)a.out & 15))) return BD_ERR_ALIGN;
    if (a.wscale && ((uintptr_t)a.wscale & 15)) return BD_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    switch (prec) {
        case BD_PREC_BF16: return launch<__bf16, 1, 64>(a, s);
        case BD_PREC_F16: return launch<_Float16, 1, 64>(a, s);
        case BD_PREC_BF16X3: return launch<__bf16, 2, 32>(a, s);
        case BD_PREC_F16X3: return launch<_Float16, 2, 32>(a, s);
        case BD_PREC_FP8: return launch<fp8e4, 1, 128>(a, s);
        case BD_PREC_F16C8: return launch_f16c8(a, s);
        default: return BD_ERR_DTYPE;
    }
}
