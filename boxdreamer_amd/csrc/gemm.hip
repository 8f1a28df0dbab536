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
#include "bd_common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;

template <int BK> __device__ __forceinline__ int swz_chunk(int row, int c) {
    constexpr int CH = BK / 8;            // 16-byte chunks per tile row
    constexpr int RPB = 16 / CH;          // tile rows per 256-byte LDS bank row
    return c ^ ((row / RPB) & (CH - 1));
}

template <class T, int NS, int BK>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const bd_gemm_args p) {
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

    // XCD-aware bijective remap of the 1-D grid: XCD x (= bid % 8) gets a contiguous run of tiles.
    const int tilesN = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int wg;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = (wg / tilesN) * BM;
    const int n0 = (wg % tilesN) * BN;

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

    // ---- epilogue: C fragment (col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    const int M = p.M, N = p.N;
    const float* bias = p.bias;
    const float* resid = p.resid;
    const float* addtab = p.addtab;
    const int act = p.act, out_f32 = p.out_f32, rpg_in = p.rpg_in, rpg_out = p.rpg_out, row_off = p.row_off;
    const int tab_rows = p.tab_rows;
    const int64_t ldr = p.ldr, ldo = p.ldo, out_plane = p.out_plane;
    float bj[2];
    int gcs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        gcs[j] = n0 + wn * 64 + j * 32 + lrow;
        bj[j] = (bias && gcs[j] < N) ? bias[gcs[j]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gr = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhalf;
            if (gr < M) {
                int64_t orow = gr;
                if (rpg_in > 0) orow = (int64_t)(gr / rpg_in) * rpg_out + gr % rpg_in + row_off;
                const float* tab = addtab ? addtab + (int64_t)(gr % tab_rows) * N : nullptr;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int gc = gcs[j];
                    if (gc < N) {
                        float v = acc[i][j][r] + bj[j];
                        if (act == BD_ACT_GELU) v = gelu_erf(v);
                        if (tab) v += tab[gc];
                        if (resid) v += resid[orow * ldr + gc];
                        if (out_f32) {
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

#undef LOAD_SLAB
#undef STORE_SLAB

template <class T, int NS, int BK> int launch(const bd_gemm_args& a, hipStream_t s) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const int slot = bd_trace_open(s, 0, a.M, a.N, a.K);
    hipLaunchKernelGGL((gemm_kernel<T, NS, BK>), dim3(tiles), dim3(256), 0, s, a);
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
    hipStream_t s = (hipStream_t)stream;
    switch (prec) {
        case BD_PREC_BF16: return launch<__bf16, 1, 64>(a, s);
        case BD_PREC_F16: return launch<_Float16, 1, 64>(a, s);
        case BD_PREC_BF16X3: return launch<__bf16, 2, 32>(a, s);
        default: return BD_ERR_DTYPE;
    }
}
