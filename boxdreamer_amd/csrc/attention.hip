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

namespace {

struct AttnArgs {
    const void* qkv; int64_t qkv_plane;
    void* out; int64_t out_plane;
    int batch, seq, heads;
    float scale_log2e;
    const int32_t* q_view;   // optional: queries are rows [q_view[b]*q_len, +q_len) of every sequence, output compact
    int q_len;               // number of query rows per sequence (== seq when q_view is NULL)
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

// OUTSPLIT: single-pass f16 attention whose result is written as split-bf16 (hi, lo) planes -- the strict mode's
// attention (its GEMMs stay split-bf16 x3): q/k are RMS-normalised and P is in [0, 1], so one f16 pass costs ~1e-4 on
// the logits while the x3 attention kernel is register-bound at one wave per SIMD.
template <class T, int NS, int HD, int NW, int OUTMODE = 0>
__global__ __launch_bounds__(NW * 64, 2) void attn_kernel(const AttnArgs p) {
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
    const int qbase_row = p.q_view ? p.q_view[b] * q_len : 0;
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
#define LOAD_TILE(kt)                                                                         \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                          \
        _Pragma("unroll") for (int i = 0; i < KCPT; ++i) {                                    \
            if (KCH % NT == 0 || tid + NT * i < KCH) {                                        \
                unsigned off = kc_off[i] + (unsigned)((kt) * KT * ld);                        \
                if (ragged && (kt) * KT + kc_row[i] >= seq) off = (unsigned)((seq - 1) * ld + kc_col[i] * 8); \
                rk[s][i] = *(const u128*)(kbase + s * plane + off);                           \
            }                                                                                 \
        }                                                                                     \
        if (vm_active) {                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                   \
                unsigned off = vm_off + (unsigned)(((kt) * KT + j) * ld);                     \
                if (ragged && (kt) * KT + vm_kq * 4 + j >= seq) off = (unsigned)((seq - 1) * ld + vm_dc * 8); \
                rv[s][j] = *(const u128*)(vbase + s * plane + off);                           \
            }                                                                                 \
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
                lo.x = (a0 & 0xffffu) | (a1 << 16); lo.y = (a2 & 0xffffu) | (a3 << 16);       \
                hi.x = (a0 >> 16) | (a1 & 0xffff0000u); hi.y = (a2 >> 16) | (a3 & 0xffff0000u); \
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
    for (int kt = 0; kt < nt; ++kt) {
        if (kt + 1 < nt) { LOAD_TILE(kt + 1) }
        const unsigned char* cur = lds + (NBUF == 2 ? (kt & 1) : 0) * BUF_BYTES;

        // ---- S^T = K . Q^T  (two 32-key M-tiles)
        f32x16 sacc[2];
#pragma unroll
        for (int km = 0; km < 2; ++km) {
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
        float tmax = -INFINITY;
        const bool tail = (kt + 1) * KT > seq;
        if (tail) {
#pragma unroll
            for (int km = 0; km < 2; ++km)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * KT + km * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= seq) sacc[km][r] = -INFINITY;
                }
        }
#pragma unroll
        for (int km = 0; km < 2; ++km)
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
        for (int km = 0; km < 2; ++km)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[km][r], sc, mneg));
                sacc[km][r] = pv;
                psum += pv;
            }
        l_run += psum;

        // ---- O^T += V^T . P^T   (four 16-key groups)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
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
        if (NBUF == 1) __syncthreads();                      // single buffer: everyone must be done reading first
        if (kt + 1 < nt) { STORE_TILE(NBUF == 2 ? ((kt + 1) & 1) : 0) }
        __syncthreads();
    }
#undef LOAD_TILE
#undef STORE_TILE

    // ---- finalise: O[q][d] = O^T / l ; lane (q, h) owns d = dm*32 + 8*(r>>2) + 4*h + (r&3)
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + lq;
    if (q < q_len) {
        if (OUTMODE == 2) {
            fp8e4* orow = (fp8e4*)p.out + ((int64_t)b * q_len + q) * (heads * HD) + head * HD;
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
            const int64_t e0 = ((int64_t)b * q_len + q) * (heads * HD) + head * HD;
#pragma unroll
            for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    float v4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v4[j] = oacc[dm][rq * 4 + j] * inv;
                    f16c8_store4((f16c8*)p.out, p.out_plane, e0 + dm * 32 + 8 * rq + 4 * lh, v4);
                }
        } else if (OUTMODE == 1) {
            __bf16* orow = (__bf16*)p.out + ((int64_t)b * q_len + q) * (heads * HD) + head * HD;
#pragma unroll
            for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d0 = dm * 32 + 8 * rq + 4 * lh;
                    typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bvec4;
                    bvec4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = oacc[dm][rq * 4 + j] * inv;
                        hi[j] = (__bf16)v;
                        lo[j] = (__bf16)(v - (float)hi[j]);
                    }
                    *(bvec4*)(orow + d0) = hi;
                    *(bvec4*)(orow + p.out_plane + d0) = lo;
                }
        } else {
            T* orow = (T*)p.out + ((int64_t)b * q_len + q) * (heads * HD) + head * HD;
#pragma unroll
            for (int dm = 0; dm < DM; ++dm)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d0 = dm * 32 + 8 * rq + 4 * lh;
                    typedef __attribute__((__vector_size__(4 * sizeof(T)))) T vec4;
                    vec4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = oacc[dm][rq * 4 + j] * inv;
                        hi[j] = from_f32<T>(v);
                        if (NS == 2) lo[j] = from_f32<T>(v - to_f32<T>(hi[j]));
                    }
                    *(vec4*)(orow + d0) = hi;
                    if (NS == 2) *(vec4*)(orow + p.out_plane + d0) = lo;
                }
        }
    }
}

template <class T, int NS, int HD, int NW, int OUTMODE = 0> int launch(const AttnArgs& a, hipStream_t s) {
    const int nqb = (a.q_len + NW * 32 - 1) / (NW * 32);
    const int slot = bd_trace_open(s, 1, a.batch * a.heads, a.seq, HD);
    hipLaunchKernelGGL((attn_kernel<T, NS, HD, NW, OUTMODE>), dim3(nqb * a.heads * a.batch), dim3(NW * 64), 0, s, a);
    bd_trace_close(s, slot);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

template <class T, int NS, int OUTMODE = 0> int dispatch(const AttnArgs& a, int head_dim, hipStream_t s) {
    // 3 waves (96-query blocks) when that tiles the sequence with less waste (DINOv2: 261 -> 3 x 96)
    const int waste4 = ((a.q_len + 127) / 128) * 128 - a.q_len, waste3 = ((a.q_len + 95) / 96) * 96 - a.q_len;
    const bool use3 = waste3 < waste4;
    if (head_dim == 96) return use3 ? launch<T, NS, 96, 3, OUTMODE>(a, s) : launch<T, NS, 96, 4, OUTMODE>(a, s);
    if (head_dim == 64) return use3 ? launch<T, NS, 64, 3, OUTMODE>(a, s) : launch<T, NS, 64, 4, OUTMODE>(a, s);
    return BD_ERR_SHAPE;
}

}  // namespace

extern "C" int bd_attention_q(const void* qkv, int64_t qkv_plane, void* out, int64_t out_plane, int batch, int seq,
                              int heads, int head_dim, float scale, const int32_t* q_view, int q_len, int prec,
                              void* stream) {
    if (!qkv || !out) return BD_ERR_NULL;
    if (batch <= 0 || seq <= 0 || heads <= 0) return BD_ERR_SHAPE;
    if (q_view ? (q_len <= 0 || q_len > seq || seq % q_len) : (q_len != seq)) return BD_ERR_SHAPE;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) return BD_ERR_ALIGN;
    AttnArgs a{qkv, qkv_plane, out, out_plane, batch, seq, heads, scale * 1.4426950408889634f, q_view, q_len};
    hipStream_t s = (hipStream_t)stream;
    switch (prec) {
        case BD_PREC_BF16: return dispatch<__bf16, 1>(a, head_dim, s);
        case BD_PREC_F16: return dispatch<_Float16, 1>(a, head_dim, s);
        case BD_PREC_BF16X3: return dispatch<__bf16, 2>(a, head_dim, s);
        case BD_PREC_F16_OUT_BF16X3: return dispatch<_Float16, 1, 1>(a, head_dim, s);
        case BD_PREC_BF16_OUT_FP8: return dispatch<__bf16, 1, 2>(a, head_dim, s);
        case BD_PREC_F16_OUT_F16C8: return dispatch<_Float16, 1, 3>(a, head_dim, s);
        case BD_PREC_BF16X3_OUT_F16C8: return dispatch<__bf16, 2, 3>(a, head_dim, s);
        default: return BD_ERR_DTYPE;
    }
}

extern "C" int bd_attention(const void* qkv, int64_t qkv_plane, void* out, int64_t out_plane, int batch,
                            int seq, int heads, int head_dim, float scale, int prec, void* stream) {
    return bd_attention_q(qkv, qkv_plane, out, out_plane, batch, seq, heads, head_dim, scale, nullptr, seq, prec, stream);
}
