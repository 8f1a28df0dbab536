// Data-movement kernels either side of the GEMMs (HBM-bound): image normalise + im2col,
// heatmap patchify, token-row bookkeeping, unpatchify + sigmoid.
#include "bd_common.h"

namespace {

// one thread per 8-wide chunk of an output row [n*grid*grid, kpad]; k = c*p*p + py*p + px
template <class T, int NS>
__global__ __launch_bounds__(256) void im2col_kernel(const void* __restrict__ img, int dtype, void* __restrict__ out_,
                                                     int64_t plane, int n_images, int size, int patch, int kpad) {
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    T* out = (T*)out_;
    const int grid = size / patch, pp = patch * patch, kreal = 3 * pp, cpr = kpad / 8;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n_images * grid * grid * cpr;
    if (t >= total) return;
    const int kc = (int)(t % cpr);
    const int64_t row = t / cpr;
    const int gx = (int)(row % grid), gy = (int)((row / grid) % grid);
    const int64_t n = row / (grid * grid);
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kc * 8 + j;
        float y = 0.f;
        if (k < kreal) {
            const int c = k / pp, rem = k % pp, py = rem / patch, px = rem % patch;
            const int64_t src = ((n * 3 + c) * size + gy * patch + py) * size + gx * patch + px;
            y = (load_any(img, src, dtype) - mean[c]) / stdv[c];   // encoder/dinov2.py:45-46
        }
        v[j] = y;
    }
    store_operand8<T, NS>(out, plane, row * kpad + kc * 8, v);
}

// BETR.patchify: chunk j of a row holds the `channels` (= 8) values of pixel (py, px), j = py*p + px
template <class T, int NS>
__global__ __launch_bounds__(256) void patchify_kernel(const void* __restrict__ heat, int dtype, void* __restrict__ out_,
                                                       int64_t plane, int n_images, int size, int patch, int kpad) {
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    T* out = (T*)out_;
    const int grid = size / patch, pp = patch * patch, cpr = kpad / 8;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n_images * grid * grid * cpr;
    if (t >= total) return;
    const int j = (int)(t % cpr);
    const int64_t row = t / cpr;
    const int gx = (int)(row % grid), gy = (int)((row / grid) % grid);
    const int64_t n = row / (grid * grid);
    float v[8];
    if (j < pp) {
        const int py = j / patch, px = j % patch;
        const int64_t pix = (int64_t)(gy * patch + py) * size + gx * patch + px;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = load_any(heat, (n * 8 + c) * (int64_t)size * size + pix, dtype);
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = 0.f;
    }
    store_operand8<T, NS>(out, plane, row * kpad + j * 8, v);
}

__global__ __launch_bounds__(256) void prefix_kernel(float* __restrict__ x, const float* __restrict__ prefix,
                                                     int n_images, int tpi, int n_prefix, int dim) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)n_images * n_prefix * dim;
    if (t >= total) return;
    const int d = (int)(t % dim), pfx = (int)((t / dim) % n_prefix);
    const int64_t n = t / ((int64_t)dim * n_prefix);
    x[(n * tpi + pfx) * dim + d] = prefix[pfx * dim + d];
}

__global__ __launch_bounds__(256) void query_sub_kernel(float* __restrict__ x, const float* __restrict__ rgb,
                                                        const float* __restrict__ pos, const float* __restrict__ qtok,
                                                        const int32_t* __restrict__ qidx, int B, int T, int P, int dim) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * P * dim;
    if (t >= total) return;
    const int d = (int)(t % dim), tok = (int)((t / dim) % P);
    const int b = (int)(t / ((int64_t)dim * P));
    const int64_t row = ((int64_t)b * T + qidx[b]) * P + tok;
    // betr.py:288-290 + :367,399:  (query + rgb) + pos  -- same association order as the reference
    x[row * dim + d] = (qtok[d] + rgb[row * dim + d]) + pos[tok * dim + d];
}

template <class T, int NS>
__global__ __launch_bounds__(256) void gather_query_kernel(const float* __restrict__ x, const int32_t* __restrict__ qidx,
                                                           void* __restrict__ out_, int64_t plane, int B, int T_, int P, int dim) {
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    T* out = (T*)out_;
    const int cpr = dim / 8;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * P * cpr;
    if (t >= total) return;
    const int c = (int)(t % cpr), tok = (int)((t / cpr) % P);
    const int b = (int)(t / ((int64_t)cpr * P));
    const float* src = x + (((int64_t)b * T_ + (qidx ? qidx[b] : 0)) * P + tok) * dim + c * 8;
    const float4 a = *(const float4*)src, bb = *(const float4*)(src + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, bb.x, bb.y, bb.z, bb.w};
    store_operand8<T, NS>(out, plane, ((int64_t)b * P + tok) * dim + c * 8, v);
}

__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const float* __restrict__ x, const int32_t* __restrict__ qidx,
                                                              float* __restrict__ out, int B, int T_, int P, int dim) {
    const int cpr = dim / 4;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * P * cpr;
    if (t >= total) return;
    const int c = (int)(t % cpr), tok = (int)((t / cpr) % P);
    const int b = (int)(t / ((int64_t)cpr * P));
    *(float4*)(out + ((int64_t)b * P + tok) * dim + c * 4) =
        *(const float4*)(x + (((int64_t)b * T_ + qidx[b]) * P + tok) * dim + c * 4);
}

// one thread per output pixel (b, y, x): reads the pixel's 8 consecutive channel features, writes 8 planes
__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ proj, float* __restrict__ logits,
                                                         float* __restrict__ heat, int B, int size, int patch) {
    const int grid = size / patch, F = patch * patch * 8;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)B * size * size;
    if (t >= total) return;
    const int x = (int)(t % size), y = (int)((t / size) % size);
    const int64_t b = t / ((int64_t)size * size);
    const int gx = x / patch, px = x % patch, gy = y / patch, py = y % patch;
    const float* src = proj + (b * grid * grid + gy * grid + gx) * F + (py * patch + px) * 8;
    const float4 a = *(const float4*)src, c = *(const float4*)(src + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
        const int64_t o = ((b * 8 + ch) * size + y) * size + x;
        if (logits) logits[o] = v[ch];
        if (heat) heat[o] = 2.0f * (1.0f / (1.0f + expf(-v[ch]))) - 1.0f;   // betr.py:432-435
    }
}


// Corner heatmap rendering -- the dataset-side producer of `bbox_feat` ("next" row f2):
// make_bbox_features(type='heatmap'), src/datasets/utils/base/bbox_utils.py:263-303.
// corners [groups*group, 8, 2] pixel (x, y) -> out [groups*group, 8, H, W] in [-1, 1].
// v = exp(-dist / (dist_to_centroid/10)^2); divided by the max over the whole call-group for that corner index
// (the reference's bbox_map[..., i].max() spans all views of the sample); then 2v - 1.  The max is the value at the
// pixel nearest to the corner (exp(-d/s) is monotone in d), so it is evaluated analytically with the SAME arithmetic
// instead of a reduction over H*W pixels.
__device__ __forceinline__ float corner_value(float bx, float by, float px, float py, float scale) {
    const float dx = bx - px, dy = by - py;
    const float d = sqrtf(dx * dx + dy * dy);
    return expf(-d / scale);
}

__global__ __launch_bounds__(256) void render_heat_kernel(const float* __restrict__ corners, int group, int H, int W,
                                                          void* __restrict__ out, int out_dtype) {
    __shared__ float s_max;
    const int map = blockIdx.y;                       // (view, corner)
    const int view = map >> 3, ci = map & 7;
    const int g0 = (view / group) * group;            // first view of this call-group
    if (threadIdx.x == 0) {
        float gmax = -INFINITY;
        for (int v = g0; v < g0 + group; ++v) {
            const float* c = corners + (int64_t)v * 16;
            float cx = 0.f, cy = 0.f;
            for (int k = 0; k < 8; ++k) { cx += c[2 * k]; cy += c[2 * k + 1]; }
            cx /= 8.0f; cy /= 8.0f;                   // bbox.mean(dim=1)
            const float bx = c[2 * ci], by = c[2 * ci + 1];
            const float ex = cx - bx, ey = cy - by;
            const float dis = sqrtf(ex * ex + ey * ey);
            const float sc = (dis / 10.0f) * (dis / 10.0f);
            const float nx = fminf(fmaxf(rintf(bx), 0.f), (float)(W - 1));
            const float ny = fminf(fmaxf(rintf(by), 0.f), (float)(H - 1));
            float best = corner_value(bx, by, nx, ny, sc);
            // rint ties / clamping: also probe the 4 neighbours so the true arg-min pixel is always covered
            const float ox[4] = {-1.f, 1.f, 0.f, 0.f}, oy[4] = {0.f, 0.f, -1.f, 1.f};
            for (int k = 0; k < 4; ++k) {
                const float qx = nx + ox[k], qy = ny + oy[k];
                if (qx >= 0.f && qx <= (float)(W - 1) && qy >= 0.f && qy <= (float)(H - 1))
                    best = fmaxf(best, corner_value(bx, by, qx, qy, sc));
            }
            gmax = fmaxf(gmax, best);
        }
        s_max = gmax;
    }
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const float* c = corners + (int64_t)view * 16;
    float cx = 0.f, cy = 0.f;
    for (int k = 0; k < 8; ++k) { cx += c[2 * k]; cy += c[2 * k + 1]; }
    cx /= 8.0f; cy /= 8.0f;
    const float bx = c[2 * ci], by = c[2 * ci + 1];
    const float ex = cx - bx, ey = cy - by;
    const float dis = sqrtf(ex * ex + ey * ey);
    const float sc = (dis / 10.0f) * (dis / 10.0f);
    const float v = corner_value(bx, by, (float)(pix % W), (float)(pix / W), sc) / s_max * 2.0f - 1.0f;
    const int64_t o = (int64_t)map * H * W + pix;
    if (out_dtype == BD_DTYPE_F32) ((float*)out)[o] = v;
    else if (out_dtype == BD_DTYPE_BF16) ((__bf16*)out)[o] = (__bf16)v;
    else ((_Float16*)out)[o] = (_Float16)v;
}

inline unsigned nblk(int64_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace

#define BD_PREC_SWITCH(FN, ...)                                                        \
    switch (prec) {                                                                    \
        case BD_PREC_BF16: hipLaunchKernelGGL((FN<__bf16, 1>), __VA_ARGS__); break;    \
        case BD_PREC_F16: hipLaunchKernelGGL((FN<_Float16, 1>), __VA_ARGS__); break;   \
        case BD_PREC_BF16X3: hipLaunchKernelGGL((FN<__bf16, 2>), __VA_ARGS__); break;  \
        case BD_PREC_FP8: hipLaunchKernelGGL((FN<fp8e4, 1>), __VA_ARGS__); break;      \
        case BD_PREC_F16C8: hipLaunchKernelGGL((FN<f16c8, 2>), __VA_ARGS__); break;    \
        default: return BD_ERR_DTYPE;                                                  \
    }

extern "C" int bd_im2col_images(const void* images, int img_dtype, void* out16, int64_t out_plane, int n_images,
                                int size, int patch, int kpad, int prec, void* stream) {
    if (!images || !out16) return BD_ERR_NULL;
    if (n_images <= 0 || size % patch || kpad % 8 || kpad < 3 * patch * patch) return BD_ERR_SHAPE;
    if (img_dtype < 0 || img_dtype > 2) return BD_ERR_DTYPE;
    const int grid = size / patch;
    const int64_t total = (int64_t)n_images * grid * grid * (kpad / 8);
    hipStream_t s = (hipStream_t)stream;
    BD_PREC_SWITCH(im2col_kernel, dim3(nblk(total)), dim3(256), 0, s, images, img_dtype, out16, out_plane,
                   n_images, size, patch, kpad)
    BD_CHECK_LAUNCH();
    return BD_OK;
}

extern "C" int bd_patchify_heatmaps(const void* heat, int in_dtype, void* out16, int64_t out_plane, int n_images,
                                    int channels, int size, int patch, int kpad, int prec, void* stream) {
    if (!heat || !out16) return BD_ERR_NULL;
    if (channels != 8 || n_images <= 0 || size % patch || kpad % 8 || kpad < patch * patch * 8) return BD_ERR_SHAPE;
    if (in_dtype < 0 || in_dtype > 2) return BD_ERR_DTYPE;
    const int grid = size / patch;
    const int64_t total = (int64_t)n_images * grid * grid * (kpad / 8);
    hipStream_t s = (hipStream_t)stream;
    BD_PREC_SWITCH(patchify_kernel, dim3(nblk(total)), dim3(256), 0, s, heat, in_dtype, out16, out_plane,
                   n_images, size, patch, kpad)
    BD_CHECK_LAUNCH();
    return BD_OK;
}

extern "C" int bd_write_prefix_tokens(float* x, const float* prefix, int n_images, int tokens_per_image,
                                      int n_prefix, int dim, void* stream) {
    if (!x || !prefix) return BD_ERR_NULL;
    if (n_images <= 0 || n_prefix <= 0 || n_prefix > tokens_per_image || dim <= 0) return BD_ERR_SHAPE;
    const int64_t total = (int64_t)n_images * n_prefix * dim;
    hipLaunchKernelGGL(prefix_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, prefix, n_images,
                       tokens_per_image, n_prefix, dim);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

extern "C" int bd_query_substitute(float* x, const float* rgb, const float* pos, const float* query_token,
                                   const int32_t* query_idx, int B, int T, int P, int dim, void* stream) {
    if (!x || !rgb || !pos || !query_token || !query_idx) return BD_ERR_NULL;
    if (B <= 0 || T <= 0 || P <= 0 || dim <= 0) return BD_ERR_SHAPE;
    const int64_t total = (int64_t)B * P * dim;
    hipLaunchKernelGGL(query_sub_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, rgb, pos,
                       query_token, query_idx, B, T, P, dim);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

extern "C" int bd_gather_query_rows_f32(const float* x, const int32_t* query_idx, float* out, int B, int T, int P,
                                        int dim, void* stream) {
    if (!x || !query_idx || !out) return BD_ERR_NULL;
    if (B <= 0 || T <= 0 || P <= 0 || dim % 4) return BD_ERR_SHAPE;
    const int64_t total = (int64_t)B * P * (dim / 4);
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, query_idx, out,
                       B, T, P, dim);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

extern "C" int bd_gather_query_tokens(const float* x, const int32_t* query_idx, void* out16, int64_t out_plane,
                                      int B, int T, int P, int dim, int prec, void* stream) {
    if (!x || !out16) return BD_ERR_NULL;                   /* query_idx == NULL: view 0 (x already compact) */
    if (B <= 0 || T <= 0 || P <= 0 || dim % 8) return BD_ERR_SHAPE;
    const int64_t total = (int64_t)B * P * (dim / 8);
    hipStream_t s = (hipStream_t)stream;
    BD_PREC_SWITCH(gather_query_kernel, dim3(nblk(total)), dim3(256), 0, s, x, query_idx, out16, out_plane, B, T, P, dim)
    BD_CHECK_LAUNCH();
    return BD_OK;
}

extern "C" int bd_unpatchify_sigmoid(const float* proj, float* logits, float* heat, int B, int channels, int size,
                                     int patch, void* stream) {
    if (!proj || (!logits && !heat)) return BD_ERR_NULL;
    if (channels != 8 || B <= 0 || size % patch) return BD_ERR_SHAPE;
    const int64_t total = (int64_t)B * size * size;
    hipLaunchKernelGGL(unpatchify_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, proj, logits, heat,
                       B, size, patch);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

extern "C" int bd_render_corner_heatmaps(const float* corners, int n_groups, int group, int height, int width,
                                         void* out, int out_dtype, void* stream) {
    if (!corners || !out) return BD_ERR_NULL;
    if (n_groups <= 0 || group <= 0 || height <= 0 || width <= 0) return BD_ERR_SHAPE;
    if (out_dtype < 0 || out_dtype > 2) return BD_ERR_DTYPE;
    const dim3 grid((unsigned)((height * width + 255) / 256), (unsigned)(n_groups * group * 8));
    hipLaunchKernelGGL(render_heat_kernel, grid, dim3(256), 0, (hipStream_t)stream, corners, group, height, width, out,
                       out_dtype);
    BD_CHECK_LAUNCH();
    return BD_OK;
}
