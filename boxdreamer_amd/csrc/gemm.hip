// MFMA GEMM with fused epilogue for every nn.Linear on the corner-heatmap path.
//
//   out[map(r), :] = act(A[r, :] . W^T + bias) + addtab[r % tab_rows, :] + resid[map(r), :]
//
// gfx950 design: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave a 64x64
// sub-tile = 2x2 v_mfma_f32_32x32x16 fragments, 64 fp32 accumulators per lane); K is streamed in
// BK-deep slabs through a double-buffered LDS image.  Both operands are K-contiguous ("A . W^T",
// the nn.Linear layout), so A and W tiles use the same loader: 16-byte global loads (8 lanes cover
// one 128-byte row segment), register-staged one slab ahead, ds_write_b128 into an XOR-swizzled
// image ( 16-byte chunk index ^= row-derived bits ) so that the 16 rows a ds_read_b128 lane group
// touches fall on 16 distinct 16-byte bank slots.  Workgroup ids are remapped so that each XCD
// (private L2) owns a contiguous run of tiles sharing A row-panels.
//
// NS = 1: one operand plane (bf16 or f16).  NS = 2: split-bf16 "x3" mode -- A and W each carry a
// hi and a lo plane and every fragment pair issues hi*hi + hi*lo + lo*hi (fp32-class accuracy
// from bf16 MFMA), BK halves so the LDS image stays 64 KiB.
#include <stdlib.h>

#include "bd_common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;


// Workgroup -> output tile.  (1) XCD-aware bijective remap: XCD x (= blockIdx % 8, private 4 MiB L2) owns a
// contiguous run of logical ids.  (2) Grouped raster inside that run: ids walk GROUP_M M-tiles down, then step
// one N-tile across, so the ~64 workgroups resident on an XCD cover a compact ~8 x 8 patch -- 8 A row-panels
// + 8 W tiles (~3 MB at K = 768) stay L2-resident instead of streaming every W tile through it.
constexpr int GROUP_M = 8;
__device__ __forceinline__ void tile_coords(int M, int N, int& m0, int& n0) {
    const int tilesM = (M + BM - 1) / BM, tilesN = (N + BN - 1) / BN;
    const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int per_group = GROUP_M * tilesN;
    const int g = wg / per_group, in_g = wg % per_group;
    const int gm0 = g * GROUP_M;
    const int gh = (tilesM - gm0) < GROUP_M ? (tilesM - gm0) : GROUP_M;     // last group may be shorter
    m0 = (gm0 + in_g % gh) * BM;
    n0 = (in_g / gh) * BN;
}

template <int BK> __device__ __forceinline__ int swz_chunk(int row, int c) {
    constexpr int CH = BK / 8;            // 16-byte chunks per tile row
    constexpr int RPB = 16 / CH;          // tile rows per 256-byte LDS bank row
    return c ^ ((row / RPB) & (CH - 1));
}

// ---- epilogue shared by both mainloops: C fragment (col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
template <class T, int NS, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const bd_gemm_args& p, f32x16 (&acc)[MI][NI], int wm0, int wn0, int lane) {
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int M = p.M, N = p.N;
    const float* bias = p.bias;
    const float* resid = p.resid;
    const float* addtab = p.addtab;
    const int act = p.act, out_f32 = p.out_f32, rpg_in = p.rpg_in, rpg_out = p.rpg_out, row_off = p.row_off;
    const int tab_rows = p.tab_rows;
    const int64_t ldr = p.ldr, ldo = p.ldo, out_plane = p.out_plane;
    float bj[NI];
    int gcs[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        gcs[j] = wn0 + j * 32 + lrow;
        bj[j] = (bias && gcs[j] < N) ? bias[gcs[j]] : 0.f;
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
                        float v = acc[i][j][r] + bj[j];
                        if (act == BD_ACT_GELU) v = gelu_erf(v);
                        if (tab) v += tab[gc];
                        if (resid) v += resid[orow * ldr + gc];
                        if (out_f32 == 1) {
                            ((float*)p.out)[orow * ldo + gc] = v;
                        } else {
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



// ---- wide epilogue (LDS-DMA kernels): accumulators -> this wave's private LDS scratch -> row-contiguous
// 16-byte accesses.  The MFMA C fragment gives each lane one column and 16 scattered rows (64 two-/four-byte
// stores per lane, each half-wave touching half a cache line): measured store-issue- and load-latency-bound
// (~27 us per tile round vs 1.3 us per K-slab).  Here every 32-row chunk of the wave tile is written to LDS with
// conflict-free ds_write_b32 (one row per half-wave), read back with ds_read_b128 as 4 (fp32 out) or 8 (16-bit
// out) consecutive columns per lane, the residual / table rows are fetched as float4 in one batch, and the result
// leaves as full-line 16-byte stores.  Same-wave LDS traffic is ordered, so no workgroup barrier is needed.
template <class T, int NS, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue_lds(const bd_gemm_args& p, f32x16 (&acc)[MI][NI], unsigned char* scratch,
                                                  int wm0, int wn0, int lane) {
    constexpr int COLS = NI * 32;                      // wave-tile width (fp32 words per scratch row)
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int M = p.M, N = p.N;
    const float* bias = p.bias;
    const float* resid = p.resid;
    const float* addtab = p.addtab;
    const int act = p.act, rpg_in = p.rpg_in, rpg_out = p.rpg_out, row_off = p.row_off, tab_rows = p.tab_rows;
    const int64_t ldr = p.ldr, ldo = p.ldo, out_plane = p.out_plane;
    float* sc = (float*)scratch;
    if (p.out_f32 == 1) {
        constexpr int LPR = COLS / 4;                  // lanes per row (float4 each)
        constexpr int RPI = 64 / LPR;                  // rows per pass
        constexpr int PASSES = 32 / RPI;
        const int c4 = lane % LPR, rsub = lane / LPR;
        const int gc = wn0 + c4 * 4;
        const bool cok = gc < N;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && cok) bv = *(const float4*)(bias + gc);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sc[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * COLS + j * 32 + lrow] = acc[i][j][r];
            // global reads are issued in batches of PB passes (register budget: 2 x 4 floats per pass in flight);
            // native vector types only -- HIP's float4 struct in a local array lands in scratch
            constexpr int PB = PASSES > 4 ? 4 : PASSES;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 bv4 = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int t0 = 0; t0 < PASSES; t0 += PB) {
                f32x4 rv[PB], tv[PB];
                int64_t orow[PB];
                bool ok[PB];
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    const int gr = wm0 + i * 32 + (t0 + u) * RPI + rsub;
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
                    f32x4 v = *(const f32x4*)(sc + ((t0 + u) * RPI + rsub) * COLS + c4 * 4) + bv4;
                    if (act == BD_ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                    }
                    v = v + tv[u] + rv[u];
                    if (ok[u]) *(f32x4*)((float*)p.out + orow[u] * ldo + gc) = v;
                }
            }
        }
    } else {
        constexpr int LPR = COLS / 8;                  // lanes per row (8 columns = 16 bytes of output each)
        constexpr int RPI = 64 / LPR;
        constexpr int PASSES = 32 / RPI;
        typedef typename Op16<T>::vec8 vec8;
        const int c8 = lane % LPR, rsub = lane / LPR;
        const int gc = wn0 + c8 * 8;
        const bool cok = gc < N;
        float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (bias && cok) {
            const float4 b0 = *(const float4*)(bias + gc), b1 = *(const float4*)(bias + gc + 4);
            bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sc[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * COLS + j * 32 + lrow] = acc[i][j][r];
#pragma unroll
            for (int t = 0; t < PASSES; ++t) {
                const int gr = wm0 + i * 32 + t * RPI + rsub;
                const float* src = sc + (t * RPI + rsub) * COLS + c8 * 8;
                const float4 a0 = *(const float4*)src, a1 = *(const float4*)(src + 4);
                float v[8] = {a0.x + bv[0], a0.y + bv[1], a0.z + bv[2], a0.w + bv[3],
                              a1.x + bv[4], a1.y + bv[5], a1.z + bv[6], a1.w + bv[7]};
                if (act == BD_ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
                }
                if (cok && gr < M) {
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
                    if (p.out_f32 == 2) {                 // f16 single plane (strict mode's attention operands)
                        f16x8 h;
#pragma unroll
                        for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[e];
                        *(f16x8*)((_Float16*)p.out + orow * ldo + gc) = h;
                    } else {
                        vec8 hi, lo;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            hi[e] = from_f32<T>(v[e]);
                            if (NS == 2) lo[e] = from_f32<T>(v[e] - to_f32<T>(hi[e]));
                        }
                        T* o = (T*)p.out + orow * ldo + gc;
                        *(vec8*)o = hi;
                        if (NS == 2) *(vec8*)(o + out_plane) = lo;
                    }
                }
            }
        }
    }
}

template <class T, int NS, int BK>
__global__ __launch_bounds__(256, 2) void gemm_kernel_regstage(const bd_gemm_args p) {
    typedef typename Op16<T>::vec8 vec8;
    constexpr int CH = BK / 8;
    constexpr int CPT = BM * CH / 256;               // chunks per thread per plane tile
    constexpr int TILE_BYTES = BM * BK * 2;
    constexpr int STAGE_BYTES = TILE_BYTES * 2 * NS; // A planes then W planes
    constexpr int KS = BK / 16;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    int m0, n0;
    tile_coords(p.M, p.N, m0, n0);

    const T* Ap = (const T*)p.A;
    const T* Wp = (const T*)p.W;

    // per-thread chunk coordinates (same for the A and W loaders)
    int crow[CPT], ccol[CPT], clds[CPT];
    const T* ga[CPT];
    const T* gw[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int q = tid + 256 * i;
        crow[i] = q / CH;
        ccol[i] = q % CH;
        clds[i] = crow[i] * (BK * 2) + (swz_chunk<BK>(crow[i], ccol[i]) << 4);
        int ar = m0 + crow[i]; ar = ar < p.M ? ar : p.M - 1;
        int wr = n0 + crow[i]; wr = wr < p.N ? wr : p.N - 1;
        ga[i] = Ap + (int64_t)ar * p.lda + ccol[i] * 8;
        gw[i] = Wp + (int64_t)wr * p.ldw + ccol[i] * 8;
    }

    u128 ra[NS][CPT], rw[NS][CPT];
    const int64_t a_plane = p.a_plane, w_plane = p.w_plane;
// (macros, not lambdas: a by-reference capture of the kernel-argument struct forces it to scratch)
#define LOAD_SLAB(k0)                                                           \
    _Pragma("unroll") for (int s = 0; s < NS; ++s)                              \
    _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                           \
        ra[s][i] = *(const u128*)(ga[i] + s * a_plane + (k0));                 \
        rw[s][i] = *(const u128*)(gw[i] + s * w_plane + (k0));                 \
    }
#define STORE_SLAB(buf)                                                         \
    _Pragma("unroll") for (int s = 0; s < NS; ++s)                              \
    _Pragma("unroll") for (int i = 0; i < CPT; ++i) {                           \
        *(u128*)(lds + (buf) * STAGE_BYTES + s * TILE_BYTES + clds[i]) = ra[s][i];        \
        *(u128*)(lds + (buf) * STAGE_BYTES + (NS + s) * TILE_BYTES + clds[i]) = rw[s][i]; \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: lane l reads row (l & 31), 16-byte chunk 2*ks + (l >> 5)
    int fa[2], fb[2];
    const int lrow = lane & 31, lhalf = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fa[i] = wm * 64 + i * 32 + lrow;
        fb[i] = wn * 64 + i * 32 + lrow;
    }

    const int nk = p.K / BK;
    LOAD_SLAB(0)
    STORE_SLAB(0)
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) { LOAD_SLAB((kt + 1) * BK) }
        const unsigned char* base = lds + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            vec8 a[NS][2], b[NS][2];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ca = swz_chunk<BK>(fa[i], ks * 2 + lhalf);
                    const int cb = swz_chunk<BK>(fb[i], ks * 2 + lhalf);
                    a[s][i] = as_vec8<T>(*(const u128*)(base + s * TILE_BYTES + fa[i] * (BK * 2) + (ca << 4)));
                    b[s][i] = as_vec8<T>(*(const u128*)(base + (NS + s) * TILE_BYTES + fb[i] * (BK * 2) + (cb << 4)));
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (NS == 2) {
                        acc[i][j] = Op16<T>::mfma(a[NS - 1][i], b[0][j], acc[i][j]);   // lo * hi
                        acc[i][j] = Op16<T>::mfma(a[0][i], b[NS - 1][j], acc[i][j]);   // hi * lo
                    }
                    acc[i][j] = Op16<T>::mfma(a[0][i], b[0][j], acc[i][j]);            // hi * hi
                }
        }
        if (kt + 1 < nk) { STORE_SLAB((kt + 1) & 1) }
        __syncthreads();
    }

    gemm_epilogue<T, NS, 2, 2>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

#undef LOAD_SLAB
#undef STORE_SLAB

// ------------------------------------------------------------------------------------------------
// LDS-DMA mainloop (default): tiles go HBM/L2 -> LDS with global_load_lds_dwordx4 -- no staging VGPRs, no
// ds_write pass.  An LDS-DMA wave-instruction writes 1 KiB at  M0-base + lane*16  (lane-linear), so the XOR
// swizzle is applied on the SOURCE side: the lane that owns LDS slot (row, c') fetches logical chunk
// c = c' ^ swz(row) of that row (same 128-byte line, so coalescing is unchanged), and fragment reads apply the
// same involution.  One __syncthreads() per K-slab: it waits vmcnt(0) (slab kt landed) and fences the
// buffer about to be overwritten; the DMA for slab kt+1 is issued right after it and flies under slab kt's MFMAs.
//
// Tile geometry is a template: WM x WN waves, each owning an (MI*32) x (NI*32) sub-tile.
//   <2,4,4,2>: 256x256 tile, 8 waves, 128 KiB LDS, 1 workgroup/CU -- 128 FLOP per LDS-DMA byte: the CU's
//              64 B/clk vector-memory path needs only half the MFMA time (the 128x128 tile needs all of it,
//              which is what capped the first version at ~20 % MFMA utilisation, profiles/r1_gemm_pmc.md)
//   <2,4,4,1>: 256x128 for N = 768 outputs (better wave quantisation: 1152 instead of 576 tiles)
//   <2,2,2,2>: 128x128 for small problems
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

template <class T, int NS, int BK, int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN >= 8 ? 2 : 2)) void gemm_kernel_glds(const bd_gemm_args p) {
    typedef typename Op16<T>::vec8 vec8;
    constexpr int NWAVE = WM * WN;
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32;
    constexpr int CH = BK / 8;                        // 16-byte chunks per tile row
    constexpr int RPP = 64 / CH;                      // tile rows per 1-KiB DMA piece
    constexpr int A_BYTES = TBM * BK * 2, W_BYTES = TBN * BK * 2;
    constexpr int PPW_A = A_BYTES / 1024 / NWAVE, PPW_W = W_BYTES / 1024 / NWAVE;
    static_assert(PPW_A >= 1 && PPW_W >= 1 && A_BYTES % (1024 * NWAVE) == 0 && W_BYTES % (1024 * NWAVE) == 0, "tile/DMA split");
    constexpr int STAGE_BYTES = (A_BYTES + W_BYTES) * NS;   // A planes then W planes
    constexpr int KS = BK / 16;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    int m0, n0;
    tile_coords_t<TBM, TBN>(p.M, p.N, TBM >= 256 ? 4 : 8, m0, n0);

    // per-lane DMA sources: piece j covers tile rows j*RPP .. j*RPP+RPP-1
    const T* ga[PPW_A];
    const T* gw[PPW_W];
#pragma unroll
    for (int i = 0; i < PPW_A; ++i) {
        const int row = (wid * PPW_A + i) * RPP + lane / CH;
        int ar = m0 + row; ar = ar < p.M ? ar : p.M - 1;
        ga[i] = (const T*)p.A + (int64_t)ar * p.lda + swz_chunk<BK>(row, lane % CH) * 8;
    }
#pragma unroll
    for (int i = 0; i < PPW_W; ++i) {
        const int row = (wid * PPW_W + i) * RPP + lane / CH;
        int wr = n0 + row; wr = wr < p.N ? wr : p.N - 1;
        gw[i] = (const T*)p.W + (int64_t)wr * p.ldw + swz_chunk<BK>(row, lane % CH) * 8;
    }
    const int64_t a_plane = p.a_plane, w_plane = p.w_plane;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#define DMA_SLAB(buf, k0)                                                                                     \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                          \
        _Pragma("unroll") for (int i = 0; i < PPW_A; ++i)                                                     \
            __builtin_amdgcn_global_load_lds((gptr_t)(ga[i] + s * a_plane + (k0)),                            \
                (lptr_t)(lds + (buf) * STAGE_BYTES + s * A_BYTES + (wid * PPW_A + i) * 1024), 16, 0, 0);      \
        _Pragma("unroll") for (int i = 0; i < PPW_W; ++i)                                                     \
            __builtin_amdgcn_global_load_lds((gptr_t)(gw[i] + s * w_plane + (k0)),                            \
                (lptr_t)(lds + (buf) * STAGE_BYTES + NS * A_BYTES + s * W_BYTES + (wid * PPW_W + i) * 1024), 16, 0, 0); \
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
    DMA_SLAB(0, 0)
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                                   // slab kt landed; buffer (kt+1)&1 is free
        if (kt + 1 < nk) { DMA_SLAB((kt + 1) & 1, (kt + 1) * BK) }
        const unsigned char* base = lds + (kt & 1) * STAGE_BYTES;
        // fragments are double-buffered in registers: the ds_read_b128s of k-step ks+1 are in flight while the
        // MFMAs of k-step ks issue (the compiler's counted lgkmcnt keeps them apart)
        constexpr int FB = NS == 1 ? 2 : 1;              // x3 mode: one fragment set (two would spill next to 128 accumulators)
        vec8 a[FB][NS][MI], b[FB][NS][NI];
#define LOAD_FRAGS(ks, slot)                                                                                   \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                       \
            _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                   \
                const int r = wm * (MI * 32) + i * 32 + lrow;                                                  \
                a[slot][s][i] = as_vec8<T>(*(const u128*)(base + s * A_BYTES + r * (BK * 2) + (swz_chunk<BK>(r, (ks) * 2 + lhalf) << 4))); \
            }                                                                                                  \
            _Pragma("unroll") for (int j = 0; j < NI; ++j) {                                                   \
                const int r = wn * (NI * 32) + j * 32 + lrow;                                                  \
                b[slot][s][j] = as_vec8<T>(*(const u128*)(base + NS * A_BYTES + s * W_BYTES + r * (BK * 2) + (swz_chunk<BK>(r, (ks) * 2 + lhalf) << 4))); \
            }                                                                                                  \
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
    }
#undef DMA_SLAB
    // wide epilogue needs 16-byte aligned rows: N % 8 == 0 and aligned leading dimensions (else scalar path)
    const bool wide = (p.N % 8 == 0) && (p.ldo % 8 == 0) && (((uintptr_t)p.out & 15) == 0) &&
                      (!p.resid || ((p.ldr % 4 == 0) && ((uintptr_t)p.resid & 15) == 0)) &&
                      (!p.bias || ((uintptr_t)p.bias & 15) == 0) && (!p.addtab || ((uintptr_t)p.addtab & 15) == 0) &&
                      (p.out_f32 || NS == 1 || (p.out_plane % 8 == 0));
    if (wide) {
        __syncthreads();                                   // every wave is done with the operand slabs
        gemm_epilogue_lds<T, NS, MI, NI>(p, acc, lds + wid * (32 * NI * 32 * 4), m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
    } else {
        gemm_epilogue<T, NS, MI, NI>(p, acc, m0 + wm * (MI * 32), n0 + wn * (NI * 32), lane);
    }
}



int gemm_impl() {   // BD_GEMM_IMPL=0 selects the register-staged mainloop (A/B measurements only)
    static const int impl = [] { const char* e = getenv("BD_GEMM_IMPL"); return e ? atoi(e) : 1; }();
    return impl;
}

template <class T, int NS, int BK, int WM, int WN, int MI, int NI> void launch_glds(const bd_gemm_args& a, hipStream_t s) {
    constexpr int TBM = WM * MI * 32, TBN = WN * NI * 32;
    const int tiles = ((a.M + TBM - 1) / TBM) * ((a.N + TBN - 1) / TBN);
    hipLaunchKernelGGL((gemm_kernel_glds<T, NS, BK, WM, WN, MI, NI>), dim3(tiles), dim3(WM * WN * 64), 0, s, a);
}

template <class T, int NS, int BK> int launch(const bd_gemm_args& a, hipStream_t s) {
    const int slot = bd_trace_open(s, 0, a.M, a.N, a.K);
    const int impl = gemm_impl();
    if (impl == 0) {
        const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
        hipLaunchKernelGGL((gemm_kernel_regstage<T, NS, BK>), dim3(tiles), dim3(256), 0, s, a);
    } else if (impl == 3) {
        launch_glds<T, NS, BK, 2, 4, 4, 1>(a, s);                 // 256 x 128 (measurement only)
    } else {
        // Tile choice = best estimated efficiency: wave quantisation over the resident slots (256x256: one workgroup
        // per CU; 128x128: two; 64x64: four) times the measured relative mainloop efficiency of the tile
        // (profiles/r1_gemm_experiments.md: 128x128 ~0.87 of 256x256 on the wide shapes; 64x64 ~0.55).
        auto eff = [&](int tm, int tn, int slots, double base) {
            const double tiles = (double)((a.M + tm - 1) / tm) * ((a.N + tn - 1) / tn);
            const double rounds = (double)(((int64_t)tiles + slots - 1) / slots);
            return base * tiles / (rounds * slots);
        };
        const double e256 = (a.N >= 1536 && impl != 2) ? eff(256, 256, 256, 1.0) : 0.0;
        const double e128 = eff(128, 128, 512, 0.87);
        const double e64 = impl == 2 ? 0.0 : eff(64, 64, 1024, 0.55);
        if (e256 >= e128 && e256 >= e64)
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
    if (a.M <= 0 || a.N <= 0 || a.K <= 0 || (a.K % 64) != 0) return BD_ERR_SHAPE;
    if ((a.lda % 8) || (a.ldw % 8) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15)) return BD_ERR_ALIGN;
    if (prec == BD_PREC_BF16X3 && ((a.a_plane % 8) || (a.w_plane % 8))) return BD_ERR_ALIGN;
    if (a.addtab && a.tab_rows <= 0) return BD_ERR_SHAPE;
    if (a.out_f32 < 0 || a.out_f32 > 2) return BD_ERR_DTYPE;
    // the f16 single-plane output exists only in the wide (16-byte) epilogue
    if (a.out_f32 == 2 && ((a.N % 8) || (a.ldo % 8) || ((uintptr_t)a.out & 15) || a.resid || a.addtab)) return BD_ERR_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    switch (prec) {
        case BD_PREC_BF16: return launch<__bf16, 1, 64>(a, s);
        case BD_PREC_F16: return launch<_Float16, 1, 64>(a, s);
        case BD_PREC_BF16X3: return launch<__bf16, 2, 32>(a, s);
        default: return BD_ERR_DTYPE;
    }
}
