// Dense-reference mode: DINO-feature reference selection.
//
// The reference (src/models/utils/matching.py:64-174) scores every (query, reference) pair with the mean over all
// L x L patch pairs of the cosine similarity of the foreground patches, pairs without two foreground patches counting
// -1e4 (its later "== -1e9" filter never fires).  That mean needs no L x L product:
//
//     mean = ( s_q . s_r  -  1e4 * (L^2 - c_q c_r) ) / L^2
//     s_v  = sum over foreground patches of f_v[l] / max(|f_v[l]|, 1e-12),   c_v = number of foreground patches
//
// so the work is ONE pass over the encoder's patch features per view (HBM-bound: L*D*4 bytes per view, read twice, the
// second time from L2) plus a dot product per pair, instead of B*N bmm's of (L x D) x (D x L).
// Foreground = luminance(0.299 R + 0.587 G + 0.114 B) > threshold at the nearest-resized pixel of the patch grid
// (F.interpolate(mode="nearest"): source index floor(i * H / g)).
#include "bd_common.h"

namespace {

constexpr int MAXL = 1024;

// one workgroup (4 waves) per view
__global__ __launch_bounds__(256) void match_sums_kernel(const float* __restrict__ feats, const void* __restrict__ images,
                                                          int img_dtype, int L, int D, int H, int W, float thr,
                                                          float* __restrict__ sums, float* __restrict__ counts) {
    __shared__ float inv[MAXL];
    __shared__ int cnt;
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) cnt = 0;
    __syncthreads();
    const float* f = feats + (int64_t)v * L * D;
    int g = 1;
    while ((g + 1) * (g + 1) <= L) ++g;                 // patch grid side
    const size_t plane = (size_t)H * W;
    for (int l = wid; l < L; l += 4) {
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) { const float x = f[(int64_t)l * D + d]; ss += x * x; }
        ss = wave_sum(ss);
        if (lane == 0) {
            const int py = (int)floorf((float)(l / g) * ((float)H / (float)g)), px = (int)floorf((float)(l % g) * ((float)W / (float)g));
            const size_t o = (size_t)v * 3 * plane + (size_t)(py < H ? py : H - 1) * W + (px < W ? px : W - 1);
            const float lum = 0.299f * load_any(images, o, img_dtype) + 0.587f * load_any(images, o + plane, img_dtype) +
                              0.114f * load_any(images, o + 2 * plane, img_dtype);
            const bool fg = lum > thr;
            inv[l] = fg ? 1.0f / fmaxf(sqrtf(ss), 1e-12f) : 0.f;
            if (fg) atomicAdd(&cnt, 1);
        }
    }
    __syncthreads();
    for (int d = tid; d < D; d += 256) {
        float s = 0.f;
        for (int l = 0; l < L; ++l) s = fmaf(f[(int64_t)l * D + d], inv[l], s);
        sums[(int64_t)v * D + d] = s;
    }
    if (tid == 0) counts[v] = (float)cnt;
}

// one wave per (sample, reference): references are the views != query_view[b], in view order
__global__ __launch_bounds__(64) void match_scores_kernel(const float* __restrict__ sums, const float* __restrict__ counts,
                                                           const int32_t* __restrict__ query_view, int T, int L, int D,
                                                           float* __restrict__ scores) {
    const int b = blockIdx.x / (T - 1), n = blockIdx.x % (T - 1), lane = threadIdx.x;
    const int q = query_view[b];
    const int r = n < q ? n : n + 1;
    const float* sq = sums + ((int64_t)b * T + q) * D;
    const float* sr = sums + ((int64_t)b * T + r) * D;
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot = fmaf(sq[d], sr[d], dot);
    dot = wave_sum(dot);
    if (lane == 0) {
        const float ll = (float)L * (float)L;
        const float invalid = ll - counts[b * T + q] * counts[b * T + r];
        float m = (dot - 1e4f * invalid) / ll;
        if (!(m == m) || fabsf(m) == INFINITY) m = 0.f;            // nan_to_num(0, 0, 0)
        scores[b * (T - 1) + n] = m;
    }
}

// top-k mask per row (N <= 1024): k rounds of (largest value, then lowest index), one workgroup per row
__global__ __launch_bounds__(256) void topk_mask_kernel(const float* __restrict__ scores, int N, int k,
                                                         unsigned char* __restrict__ mask) {
    __shared__ float val[MAXL];
    __shared__ float wv[4];
    __shared__ int wi[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < N; i += 256) { val[i] = scores[(int64_t)b * N + i]; mask[(int64_t)b * N + i] = 0; }
    __syncthreads();
    for (int round = 0; round < k; ++round) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = tid; i < N; i += 256) if (val[i] > bv || (val[i] == bv && i < bi)) { bv = val[i]; bi = i; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { wv[wid] = bv; wi[wid] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) if (wv[w] > bv || (wv[w] == bv && wi[w] < bi)) { bv = wv[w]; bi = wi[w]; }
            if (bi < N) { mask[(int64_t)b * N + bi] = 1; val[bi] = -INFINITY; }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int bd_dino_match_scores(const float* feats, const void* images, int img_dtype, const int32_t* query_view, int B,
                                    int T, int L, int D, int H, int W, float lum_threshold, float* sums, float* counts,
                                    float* scores, void* stream) {
    if (!feats || !images || !query_view || !sums || !counts || !scores) return BD_ERR_NULL;
    if (B <= 0 || T < 2 || L <= 0 || L > MAXL || D <= 0 || H <= 0 || W <= 0) return BD_ERR_SHAPE;
    if (img_dtype != BD_DTYPE_F32 && img_dtype != BD_DTYPE_BF16 && img_dtype != BD_DTYPE_F16) return BD_ERR_DTYPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(match_sums_kernel, dim3(B * T), dim3(256), 0, s, feats, images, img_dtype, L, D, H, W, lum_threshold,
                       sums, counts);
    hipLaunchKernelGGL(match_scores_kernel, dim3(B * (T - 1)), dim3(64), 0, s, sums, counts, query_view, T, L, D, scores);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

extern "C" int bd_topk_mask(const float* scores, int B, int N, int k, unsigned char* mask, void* stream) {
    if (!scores || !mask) return BD_ERR_NULL;
    if (B <= 0 || N <= 0 || N > MAXL || k <= 0 || k > N) return BD_ERR_SHAPE;
    hipLaunchKernelGGL(topk_mask_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, scores, N, k, mask);
    BD_CHECK_LAUNCH();
    return BD_OK;
}
