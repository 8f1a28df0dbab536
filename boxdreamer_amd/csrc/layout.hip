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

// ---------------------------------------------------------------- row-strip forms of im2col / patchify (patch 14: the path's geometry)
// The one-thread-per-chunk kernels above gather 2-byte elements (14-pixel runs per plane and patch row: 1.4 / 2.2 TB/s on the
// path's shapes).  im2col: one workgroup takes a horizontal strip of <= 8 patches of one patch row: every (plane, pixel row) of the
// strip is one contiguous run of <= 112 pixels, loaded in 4-byte pieces into an LDS image [patch][plane][14 x 14] of fp32 values,
// and the operand rows leave as whole 16-byte chunks.  patchify: see patchify_pair_kernel.  Same arithmetic per element as the
// kernels above (same bits: tools/layout_hash.py).
constexpr int STRIP = 8, P14 = 14, PP14 = P14 * P14;

__device__ __forceinline__ void load_pair_f32(const void* src, int64_t e, int dtype, float& a, float& b) {       // elements e, e + 1 (e even)
    if (dtype == BD_DTYPE_F32) {
        const float2 v = *(const float2*)((const float*)src + e);
        a = v.x; b = v.y;
    } else {
        const unsigned u = *(const unsigned*)((const unsigned short*)src + e);
        if (dtype == BD_DTYPE_BF16) {
            a = __builtin_bit_cast(float, u << 16); b = __builtin_bit_cast(float, u & 0xffff0000u);
        } else {
            a = (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); b = (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
        }
    }
}

// planes: 3 (images, normalised) or 8 (heatmaps).  Fills tile[p][c][py * 14 + px] for the strip's np patches.
template <int NC, bool NORMALISE>
__device__ __forceinline__ void strip_load(const void* src, int dtype, int64_t n, int size, int gy, int x0, int np, float* tile) {
    const float mean[3] = {0.485f, 0.456f, 0.406f};
    const float stdv[3] = {0.229f, 0.224f, 0.225f};
    // lane = 4-byte piece of the run (<= 56 of 64 lanes), wave = (plane, pixel row) mod 4: no per-element index division
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane >= np * (P14 / 2)) return;
    const int x = 2 * lane, p = x / P14, px = x % P14;  // (14 is even: both pixels of a piece lie in the same patch)
    float* dcol = tile + p * NC * PP14 + px;
    const int64_t base = ((n * NC) * size + gy * P14) * (int64_t)size + x0 + x;
#pragma unroll 4
    for (int cy = wid; cy < NC * P14; cy += 4) {
        const int c = cy / P14, py = cy - c * P14;
        float a, b;
        load_pair_f32(src, base + ((int64_t)c * size + py) * size, dtype, a, b);
        if constexpr (NORMALISE) { a = (a - mean[c < 3 ? c : 0]) / stdv[c < 3 ? c : 0]; b = (b - mean[c < 3 ? c : 0]) / stdv[c < 3 ? c : 0]; }   // encoder/dinov2.py:45-46
        float* d = dcol + c * PP14 + py * P14;
        d[0] = a; d[1] = b;
    }
}

template <class T, int NS>
__global__ __launch_bounds__(256) void im2col_strip_kernel(const void* __restrict__ img, int dtype, void* __restrict__ out_,
                                                           int64_t plane, int n_images, int size, int kpad) {
    bd_saturating_conversions();
    __shared__ __attribute__((aligned(16))) float tile[STRIP * 3 * PP14];
    const int grid = size / P14, nsb = (grid + STRIP - 1) / STRIP, cpr = kpad / 8, kreal = 3 * PP14;
    const int sb = blockIdx.x % nsb, gy = (blockIdx.x / nsb) % grid;
    const int64_t n = blockIdx.x / (nsb * grid);
    const int gx0 = sb * STRIP, np = grid - gx0 < STRIP ? grid - gx0 : STRIP;
    strip_load<3, true>(img, dtype, n, size, gy, gx0 * P14, np, tile);
    __syncthreads();
    T* out = (T*)out_;
    for (int i = threadIdx.x; i < np * cpr; i += 256) {
        const int p = i / cpr, kc = i % cpr;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int k = kc * 8 + j; v[j] = k < kreal ? tile[p * kreal + k] : 0.f; }     // k = c*196 + py*14 + px
        const int64_t row = (n * grid + gy) * grid + gx0 + p;
        store_operand8<T, NS>(out, plane, row * kpad + kc * 8, v);
    }
}

// patchify without the LDS: a thread takes the two pixels (y, x), (y, x + 1) of ALL 8 channel planes (eight 4-byte loads, 256
// contiguous bytes per wave instruction) -- exactly the two adjacent 8-channel chunks (py, px), (py, px + 1) of one operand row.
// (The strip form with an LDS image measured slower than the gather it replaces: 161 against 114 us.)
template <class T, int NS>
__global__ __launch_bounds__(256) void patchify_pair_kernel(const void* __restrict__ heat, int dtype, void* __restrict__ out_,
                                                            int64_t plane, int n_images, int size, int kpad) {
    bd_saturating_conversions();
    const int half = size / 2, grid = size / P14, cpr = kpad / 8;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)n_images * size * half) return;
    const int xp = (int)(t % half), y = (int)((t / half) % size);
    const int64_t n = t / ((int64_t)half * size);
    const int x = 2 * xp, gx = x / P14, px = x % P14, gy = y / P14, py = y % P14;
    float a[8], b[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) load_pair_f32(heat, ((n * 8 + c) * size + y) * (int64_t)size + x, dtype, a[c], b[c]);
    T* out = (T*)out_;
    const int64_t row = (n * grid + gy) * grid + gx;
    const int j = py * P14 + px;
    store_operand8<T, NS>(out, plane, row * kpad + j * 8, a);
    store_operand8<T, NS>(out, plane, row * kpad + (j + 1) * 8, b);
    if (py == 0) {                                       // the row's zero padding chunks [196, kpad / 8): by the patch's first pixel row
        const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int jj = PP14 + px / 2; jj < cpr; jj += P14 / 2) store_operand8<T, NS>(out, plane, row * kpad + jj * 8, z);
    }
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
        case BD_PREC_F16X3: hipLaunchKernelGGL((FN<_Float16, 2>), __VA_ARGS__); break; \
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
    // (4-byte input pieces: even image width and an aligned base)
    if (patch == P14 && size % 2 == 0 && ((uintptr_t)images & 7) == 0) {
        const int nsb = (grid + STRIP - 1) / STRIP;
        BD_PREC_SWITCH(im2col_strip_kernel, dim3((unsigned)((int64_t)n_images * grid * nsb)), dim3(256), 0, s, images, img_dtype, out16,
                       out_plane, n_images, size, kpad)
    } else {
        BD_PREC_SWITCH(im2col_kernel, dim3(nblk(total)), dim3(256), 0, s, images, img_dtype, out16, out_plane,
                       n_images, size, patch, kpad)
    }
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
    if (patch == P14 && size % 2 == 0 && ((uintptr_t)heat & 7) == 0) {
        BD_PREC_SWITCH(patchify_pair_kernel, dim3(nblk((int64_t)n_images * size * (size / 2))), dim3(256), 0, s, heat, in_dtype, out16,
                       out_plane, n_images, size, kpad)
    } else {
        BD_PREC_SWITCH(patchify_kernel, dim3(nblk(total)), dim3(256), 0, s, heat, in_dtype, out16, out_plane,
                       n_images, size, patch, kpad)
    }
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
