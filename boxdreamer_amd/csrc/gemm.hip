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
#define BD_STORE_NT 1   // non-temporal 16/8-bit epilogue stores (bd_common.h: store_cvt)
#include "bd_common.h"

#ifdef BD_GEMM_PROBE
// Measurement build only (tools/gemm_phase_probe.py; never part of libboxdreamer_hip.so): per-wave shader-clock stamps of
// the mainloop phases, kept in the lanes of one VGPR (lane i = stamp i) and written out once at kernel end.
__device__ unsigned* bd_probe_buf = nullptr;
extern "C" int bd_gemm_probe_set(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(bd_probe_buf), &buf, sizeof(buf));
}
#define BD_PROBE(idx) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if ((idx) < 64) probe_ts = (lane == (idx)) ? (unsigned)t__ : probe_ts; }
#else
#define BD_PROBE(idx)
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
__device__ __forceinline__ unsigned lds_offset_of(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p);
}

// out_f32 codes
enum { OUT_OPERAND = 0, OUT_F32 = 1, OUT_F16 = 2, OUT_BF16 = 3 };

// ---- scalar fallback epilogue (N not a multiple of 8 / unaligned pointers): C fragment = (col = lane & 31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5)); operand-dtype (16-bit) or fp32 outputs only
template <class T, int NS, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const bd_gemm_args& p, f32x16 (&acc)[MI][NI], int wm0, int wn0, int lane) {
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int M = p.M, N = p.N;
    const float* bias = p.bias;
    const float* resid = p.resid;
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
                        float v = acc[i][j][r] * sj[j] + bj[j];
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
template <class T, int NS, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue_lds(const bd_gemm_args& p, f32x16 (&acc)[MI][NI], unsigned char* scratch,
                                                  int wm0, int wn0, int lane) {
    constexpr int COLS = NI * 32;                      // wave-tile width (fp32 words per scratch row)
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int M = p.M, N = p.N;
    const float* bias = p.bias;
    const float* resid = p.resid;
    const float* addtab = p.addtab;
    const float* wscale = p.wscale;
    const int act = p.act, rpg_in = p.rpg_in, rpg_out = p.rpg_out, row_off = p.row_off, tab_rows = p.tab_rows;
    const int64_t ldr = p.ldr, ldo = p.ldo, out_plane = p.out_plane;
    float* sc = (float*)scratch;
    // the wave tile is flushed in column blocks of CW columns (all of it when it is 32 or 64 wide; 3 x 32 for the 96-wide
    // wave tile of the 256 x 192 workgroup tile) so that the lanes of a pass always cover whole rows of a block
    constexpr int CW = (NI % 2 == 0) ? 64 : 32, NCB = COLS / CW;
    if (p.out_f32 == OUT_F32) {
        constexpr int LPR = CW / 4;                    // lanes per row (4 floats each)
        constexpr int RPI = 64 / LPR;                  // rows per pass
        constexpr int PASSES = 32 / RPI;
        const int c4 = lane % LPR, rsub = lane / LPR;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
        f32x4 bv4[NCB], sv4[NCB];
        bool cok[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int gc = wn0 + cb * CW + c4 * 4;
            cok[cb] = gc < N;
            bv4[cb] = zero4; sv4[cb] = one4;
            if (bias && cok[cb]) bv4[cb] = *(const f32x4*)(bias + gc);
            if (wscale && cok[cb]) sv4[cb] = *(const f32x4*)(wscale + gc);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sc[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * COLS + j * 32 + lrow] = acc[i][j][r];
            // global reads are issued in batches of PB passes (register budget); native vector types only -- HIP's
            // float4 struct in a local array lands in scratch
            constexpr int PB = PASSES > 4 ? 4 : PASSES;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int gc = wn0 + cb * CW + c4 * 4;
#pragma unroll
                for (int t0 = 0; t0 < PASSES; t0 += PB) {
                    f32x4 rv[PB], tv[PB];
                    int64_t orow[PB];
                    bool ok[PB];
#pragma unroll
                    for (int u = 0; u < PB; ++u) {
                        const int gr = wm0 + i * 32 + (t0 + u) * RPI + rsub;
                        ok[u] = cok[cb] && gr < M;
                        const int grc = gr < M ? gr : M - 1;
                        orow[u] = rpg_in > 0 ? (int64_t)(grc / rpg_in) * rpg_out + grc % rpg_in + row_off : (int64_t)grc;
                        rv[u] = zero4;
                        tv[u] = zero4;
                        if (resid && ok[u]) rv[u] = *(const f32x4*)(resid + orow[u] * ldr + gc);
                        if (addtab && ok[u]) tv[u] = *(const f32x4*)(addtab + (int64_t)(grc % tab_rows) * N + gc);
                    }
#pragma unroll
                    for (int u = 0; u < PB; ++u) {
                        f32x4 v = *(const f32x4*)(sc + ((t0 + u) * RPI + rsub) * COLS + cb * CW + c4 * 4) * sv4[cb] + bv4[cb];
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
        constexpr int PASSES = 32 / RPI;
        const int c8 = lane % LPR, rsub = lane / LPR;
        float bv[NCB][8], sv[NCB][8];
        bool cok[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int gc = wn0 + cb * CW + c8 * 8;
            cok[cb] = gc < N;
#pragma unroll
            for (int e = 0; e < 8; ++e) { bv[cb][e] = 0.f; sv[cb][e] = 1.f; }
            if (bias && cok[cb]) {
                const f32x4 b0 = *(const f32x4*)(bias + gc), b1 = *(const f32x4*)(bias + gc + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bv[cb][e] = b0[e]; bv[cb][4 + e] = b1[e]; }
            }
            if (wscale && cok[cb]) {
                const f32x4 s0 = *(const f32x4*)(wscale + gc), s1 = *(const f32x4*)(wscale + gc + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { sv[cb][e] = s0[e]; sv[cb][4 + e] = s1[e]; }
            }
        }
        const int out_mode = p.out_f32;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sc[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * COLS + j * 32 + lrow] = acc[i][j][r];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int gc = wn0 + cb * CW + c8 * 8;
#pragma unroll
                for (int t = 0; t < PASSES; ++t) {
                    const int gr = wm0 + i * 32 + t * RPI + rsub;
                    const float* src = sc + (t * RPI + rsub) * COLS + cb * CW + c8 * 8;
                    const f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = a0[e] * sv[cb][e] + bv[cb][e]; v[4 + e] = a1[e] * sv[cb][4 + e] + bv[cb][4 + e]; }
                    if (act == BD_ACT_GELU) gelu_n<GeluKind<T, NS>::value, 8>(v);   // 16/8-bit result: fitted forms (bd_common.h)
                    if (cok[cb] && gr < M) {
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
                        if (out_mode == OUT_F16) {            // f16 single plane (optional f16 attention of the strict mode)
                            store_cvt<_Float16, 8>((_Float16*)p.out + orow * ldo + gc, v);
                        } else if (out_mode == OUT_BF16) {    // bf16 single plane (fp8 mode: attention operands stay bf16)
                            store_cvt<__bf16, 8>((__bf16*)p.out + orow * ldo + gc, v);
                        } else {
                            T* o = (T*)p.out + orow * ldo + gc;
                            if constexpr (NS == 2) {
                                float hi8[8], lo8[8];
#pragma unroll
                                for (int e = 0; e < 8; ++e) { hi8[e] = to_f32<T>(from_f32<T>(v[e])); lo8[e] = v[e] - hi8[e]; }
                                store_cvt<T, 8>(o, hi8);
                                store_cvt<T, 8>(o + out_plane, lo8);
                            } else {
                                store_cvt<T, 8>(o, v);
                            }
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <class T, int NS, int BK, int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_kernel_glds(const bd_gemm_args p) {
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
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE_BYTES > EPI_SCRATCH ? 2 * STAGE_BYTES : EPI_SCRATCH];

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
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lrow = lane & 31, lhalf = lane >> 5;
    const int nk = p.K / BK;
#ifdef BD_GEMM_PROBE
    unsigned probe_ts = 0;
#endif
    BD_PROBE(60)
    DMA_SLAB(0, 0)
    for (int kt = 0; kt < nk; ++kt) {
        BD_PROBE(kt * 3)
        slab_barrier();                                    // slab kt landed; buffer (kt+1)&1 is free
        BD_PROBE(kt * 3 + 1)
        if (kt + 1 < nk) { DMA_SLAB((kt + 1) & 1, (kt + 1) * ROWB) }
        BD_PROBE(kt * 3 + 2)
        const unsigned char* base = lds + (kt & 1) * STAGE_BYTES;
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

template <class T, int NS, int BK, int WM, int WN, int MI, int NI> void launch_glds(const bd_gemm_args& a, hipStream_t s) {
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32;
    const int tiles = ((a.M + TBM - 1) / TBM) * ((a.N + TBN - 1) / TBN);
    hipLaunchKernelGGL((gemm_kernel_glds<T, NS, BK, WM, WN, MI, NI>), dim3(tiles), dim3(WM * WN * 64), 0, s, a);
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

template <class T, int NS, int BK> int launch(const bd_gemm_args& a, hipStream_t s) {
    const int slot = bd_trace_open(s, 0, a.M, a.N, a.K);
    const int kCUs = cu_count();
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
                launch_glds<T, NS, BK, 4, 2, 2, 3>(a, s);         // 256 x 192, 1 workgroup / CU
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
            launch_glds<T, NS, BK, 2, 4, 4, 2>(a, s);             // 256 x 256, 1 workgroup / CU
        else if (e128 >= e64)
            launch_glds<T, NS, BK, 2, 2, 2, 2>(a, s);             // 128 x 128, 2 workgroups / CU
        else
            launch_glds<T, NS, BK, 2, 2, 1, 1>(a, s);             // 64 x 64 (latency mode: batch 1, M = 1536)
    }
    bd_trace_close(s, slot);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

}  // namespace

extern "C" int bd_gemm(const bd_gemm_args* args, int prec, void* stream) {
    if (!args || !args->A || !args->W || !args->out) return BD_ERR_NULL;
    const bd_gemm_args& a = *args;
    const int kmult = prec == BD_PREC_FP8 ? 128 : 64;
    const int esz = prec == BD_PREC_FP8 ? 1 : 2;
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % kmult) != 0) return BD_ERR_SHAPE;
    if (((a.lda * esz) % 16) || ((a.ldw * esz) % 16) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15)) return BD_ERR_ALIGN;
    if (prec == BD_PREC_BF16X3 && ((a.a_plane % 8) || (a.w_plane % 8))) return BD_ERR_ALIGN;
    if (a.addtab && a.tab_rows <= 0) return BD_ERR_SHAPE;
    if (a.out_f32 < 0 || a.out_f32 > 3) return BD_ERR_DTYPE;
    // the single-plane 16-bit outputs and every fp8-mode output exist only in the wide (16-byte) epilogue
    const bool needs_wide = a.out_f32 >= 2 || (prec == BD_PREC_FP8 && a.out_f32 == 0);
    if (needs_wide && ((a.N % 8) || (a.ldo % 8) || ((uintptr_t)a.out & 15))) return BD_ERR_ALIGN;
    if (a.wscale && ((uintptr_t)a.wscale & 15)) return BD_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    switch (prec) {
        case BD_PREC_BF16: return launch<__bf16, 1, 64>(a, s);
        case BD_PREC_F16: return launch<_Float16, 1, 64>(a, s);
        case BD_PREC_BF16X3: return launch<__bf16, 2, 32>(a, s);
        case BD_PREC_FP8: return launch<fp8e4, 1, 128>(a, s);
        default: return BD_ERR_DTYPE;
    }
}
