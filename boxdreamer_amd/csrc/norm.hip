// Row normalisations of the corner-heatmap path (HBM-bound; one wave per row, 16-byte accesses).
#include "bd_common.h"

namespace {

// ---------------------------------------------------------------- LayerNorm (fp32 stats)
// One wave64 per row; a row of <= 1024 fp32 lives in 4 float4 registers per lane.
template <class T, int NS>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        T* __restrict__ out16, int64_t out16_plane,
                                                        float* __restrict__ out32, int64_t ldo, int rows,
                                                        int cols, int rpg_in, int rpg_out, int row_off) {
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    int64_t ir = r;
    if (rpg_in > 0) ir = (int64_t)(r / rpg_in) * rpg_out + r % rpg_in + row_off;
    // non-temporal row loads: the residual stream (151 MB) is far larger than the L2s and is not read again by this kernel
    // (whole step +0.6 %; the same hint on the residual loads of the GEMM epilogue, on q/k RMSNorm and on the attention K/V
    // loads measured neutral to -3 %)
    ln_row<T, NS, true>(x + ir * ldx, gamma, beta, eps, out16, out16_plane, out32 ? out32 + (int64_t)r * ldo : nullptr, (int64_t)r, cols, lane);
}

// ---------------------------------------------------------------- q/k RMSNorm, in place
// One thread per (row, q|k, head) vector of HD 16-bit values (HD/8 16-byte chunks in registers).
template <class T, int NS, int HD>
__global__ __launch_bounds__(256) void qk_rmsnorm_kernel(T* __restrict__ qkv, int64_t plane,
                                                         const float* __restrict__ wq,
                                                         const float* __restrict__ wk, float eps,
                                                         int rows, int heads) {
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    typedef typename Op16<T>::vec8 vec8;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)rows * 2 * heads;
    if (t >= total) return;
    const int head = (int)(t % heads);
    const int which = (int)((t / heads) % 2);
    const int64_t row = t / (2 * heads);
    T* ptr = qkv + row * (3 * heads * HD) + (int64_t)which * heads * HD + head * HD;
    const float* w = which ? wk : wq;
    constexpr int NC = HD / 8;
    float f[NC][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        vec8 hi = as_vec8<T>(*(const u128*)(ptr + c * 8));
        vec8 lo;
        if (NS == 2) lo = as_vec8<T>(*(const u128*)(ptr + plane + c * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = to_f32<T>(hi[j]);
            if (NS == 2) a += to_f32<T>(lo[j]);
            f[c][j] = a;
            ss += a * a;
        }
    }
    const float r = rsqrtf(ss / (float)HD + eps);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        vec8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = w[c * 8 + j] * (f[c][j] * r);
            if (NS == 2) {
                float h, l;
                split_hi_lo<T>(y, h, l);
                hi[j] = from_f32<T>(h);
                lo[j] = from_f32<T>(l);
            } else {
                hi[j] = from_f32<T>(y);
            }
        }
        *(vec8*)(ptr + c * 8) = hi;
        if (NS == 2) *(vec8*)(ptr + plane + c * 8) = lo;
    }
}

// Coalesced variant: one wave per token row.  The q|k part of a row is 2*heads*HD contiguous elements; lane l owns
// EPL = 2*heads*HD/64 consecutive ones (three 16-byte chunks for 8 heads x 96), so a wave's loads and stores cover the
// 3 KiB row back to back (the thread-per-vector kernel above strides 192 B between lanes), and the LPV = HD/EPL lanes of
// one (q|k, head) vector combine their sums of squares with LPV-1 xor-shuffles.
template <class T, int NS, int HD, int EPL>
__global__ __launch_bounds__(256) void qk_rmsnorm_row_kernel(T* __restrict__ qkv, int64_t plane,
                                                             const float* __restrict__ wq,
                                                             const float* __restrict__ wk, float eps,
                                                             int rows, int heads) {
    bd_saturating_conversions();      // fp8 / f16 results saturate (bd_common.h: RANGE)
    typedef typename Op16<T>::vec8 vec8;
    constexpr int NC = EPL / 8, LPV = HD / EPL;
    static_assert(EPL % 8 == 0 && HD % EPL == 0 && (LPV & (LPV - 1)) == 0, "row split");
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    T* ptr = qkv + row * (3 * heads * HD) + lane * EPL;
    const int vec = lane / LPV;                         // 0 .. 2*heads-1: q heads then k heads
    const float* w = (vec < heads ? wq : wk) + (lane % LPV) * EPL;
    float f[NC][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        vec8 hi = as_vec8<T>(*(const u128*)(ptr + c * 8));
        vec8 lo;
        if (NS == 2) lo = as_vec8<T>(*(const u128*)(ptr + plane + c * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = to_f32<T>(hi[j]);
            if (NS == 2) a += to_f32<T>(lo[j]);
            f[c][j] = a;
            ss += a * a;
        }
    }
#pragma unroll
    for (int o = 1; o < LPV; o <<= 1) ss += __shfl_xor(ss, o);
    const float r = rsqrtf(ss / (float)HD + eps);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const f32x4 w0 = *(const f32x4*)(w + c * 8), w1 = *(const f32x4*)(w + c * 8 + 4);
        vec8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = (j < 4 ? w0[j] : w1[j - 4]) * (f[c][j] * r);
            if (NS == 2) {
                float h, l;
                split_hi_lo<T>(y, h, l);
                hi[j] = from_f32<T>(h);
                lo[j] = from_f32<T>(l);
            } else {
                hi[j] = from_f32<T>(y);
            }
        }
        *(vec8*)(ptr + c * 8) = hi;
        if (NS == 2) *(vec8*)(ptr + plane + c * 8) = lo;
    }
}

template <class T, int NS>
int launch_ln(const float* x, int64_t ldx, const float* g, const float* b, float eps, void* o16, int64_t o16p,
              float* o32, int64_t ldo, int rows, int cols, int rpg_in, int rpg_out, int row_off, hipStream_t s) {
    hipLaunchKernelGGL((layernorm_kernel<T, NS>), dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, g, b, eps,
                       (T*)o16, o16p, o32, ldo, rows, cols, rpg_in, rpg_out, row_off);
    BD_CHECK_LAUNCH();
    return BD_OK;
}

template <class T, int NS>
int launch_rms(void* qkv, int64_t plane, const float* wq, const float* wk, float eps, int rows, int heads,
               int hd, hipStream_t s) {
    const int64_t total = (int64_t)rows * 2 * heads;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (hd == 96 && heads == 8) {                      // BETR: the whole q|k row across one wave
        hipLaunchKernelGGL((qk_rmsnorm_row_kernel<T, NS, 96, 24>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (T*)qkv,
                           plane, wq, wk, eps, rows, heads);
        BD_CHECK_LAUNCH();
        return BD_OK;
    }
    if (hd == 96)
        hipLaunchKernelGGL((qk_rmsnorm_kernel<T, NS, 96>), grid, dim3(256), 0, s, (T*)qkv, plane, wq, wk, eps, rows, heads);
    else if (hd == 64)
        hipLaunchKernelGGL((qk_rmsnorm_kernel<T, NS, 64>), grid, dim3(256), 0, s, (T*)qkv, plane, wq, wk, eps, rows, heads);
    else
        return BD_ERR_SHAPE;
    BD_CHECK_LAUNCH();
    return BD_OK;
}

}  // namespace

extern "C" int bd_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                            void* out16, int64_t out16_plane, float* out32, int64_t ldo, int rows, int cols,
                            int rpg_in, int rpg_out, int row_off, int prec, void* stream) {
    if (!x || (!out16 && !out32)) return BD_ERR_NULL;
    if (rows <= 0 || cols <= 0 || cols > 1024 || (cols % 4) || (ldx % 4) || (out32 && (ldo % 4))) return BD_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    switch (prec) {
        case BD_PREC_BF16: return launch_ln<__bf16, 1>(x, ldx, gamma, beta, eps, out16, out16_plane, out32, ldo, rows, cols, rpg_in, rpg_out, row_off, s);
        case BD_PREC_F16: return launch_ln<_Float16, 1>(x, ldx, gamma, beta, eps, out16, out16_plane, out32, ldo, rows, cols, rpg_in, rpg_out, row_off, s);
        case BD_PREC_BF16X3: return launch_ln<__bf16, 2>(x, ldx, gamma, beta, eps, out16, out16_plane, out32, ldo, rows, cols, rpg_in, rpg_out, row_off, s);
        case BD_PREC_F16X3: return launch_ln<_Float16, 2>(x, ldx, gamma, beta, eps, out16, out16_plane, out32, ldo, rows, cols, rpg_in, rpg_out, row_off, s);
        case BD_PREC_FP8: return launch_ln<fp8e4, 1>(x, ldx, gamma, beta, eps, out16, out16_plane, out32, ldo, rows, cols, rpg_in, rpg_out, row_off, s);
        case BD_PREC_F16C8:
            if (out16 && cols % 32) return BD_ERR_SHAPE;      // the lo8 plane is laid out in 32-element blocks
            return launch_ln<f16c8, 2>(x, ldx, gamma, beta, eps, out16, out16_plane, out32, ldo, rows, cols, rpg_in, rpg_out, row_off, s);
        default: return BD_ERR_DTYPE;
    }
}

extern "C" int bd_qk_rmsnorm(void* qkv, int64_t plane, const float* wq, const float* wk, float eps, int rows,
                             int heads, int head_dim, int prec, void* stream) {
    if (!qkv || !wq || !wk) return BD_ERR_NULL;
    if (rows <= 0 || heads <= 0) return BD_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    switch (prec) {
        case BD_PREC_BF16: return launch_rms<__bf16, 1>(qkv, plane, wq, wk, eps, rows, heads, head_dim, s);
        case BD_PREC_F16: return launch_rms<_Float16, 1>(qkv, plane, wq, wk, eps, rows, heads, head_dim, s);
        case BD_PREC_BF16X3: return launch_rms<__bf16, 2>(qkv, plane, wq, wk, eps, rows, heads, head_dim, s);
        case BD_PREC_F16X3: return launch_rms<_Float16, 2>(qkv, plane, wq, wk, eps, rows, heads, head_dim, s);
        default: return BD_ERR_DTYPE;
    }
}
