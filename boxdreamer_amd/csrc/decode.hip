// Heatmap -> 2-D corner decode (reference: src/models/utils/box_utils.py:75-110).
//
// Per (sample, corner) map: h = (heat + 1) / 2 in fp32, top-k (k = 20) over H*W, corner = the
// unweighted mean of the k integer (x, y) = (idx % W, idx / W).  `torch.topk` leaves the order of
// equal values unspecified; this kernel pins: larger value first, then LOWER index first.
//
// One 256-thread workgroup per map.  Every thread owns a CONTIGUOUS slice of the map and keeps one candidate:
// the best element of its slice that comes strictly after the last global pick in (value desc, index asc) order.
// A round = one workgroup arg-max over the 256 candidates (wave shuffles + 4 LDS slots); only the thread that won
// rescans its slice (float4 loads, L1/L2-resident) for its next candidate.  k rounds -> one full pass over the map
// plus k short slice rescans, instead of k full passes (first version: 1.34 ms for 256 maps; profiles/).
// Deterministic, no atomics, exact tie rule.
#include "bd_common.h"

namespace {

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
    return (v > bv) || (v == bv && i < bi);
}

// best element of [lo, hi) strictly after (pv, pi); slices are 16-byte aligned when VEC
template <bool VEC>
__device__ __forceinline__ void scan_slice(const float* __restrict__ h, int lo, int hi, float pv, int pi, float& bv, int& bi) {
    bv = -INFINITY;
    bi = 0x7fffffff;
    if (VEC) {
        // 7 independent 16-byte loads in flight per step (a 196-float slice = 49 float4 = 7 x 7); a plain loop
        // serialises on one L2 round trip per float4 (0.48 ms per decode in profiles/r1_bench_bf16_kernel_stats.csv)
        for (int i0 = lo; i0 < hi; i0 += 28) {
            float4 q[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int i = i0 + 4 * u;
                q[u] = *(const float4*)(h + (i < hi ? i : lo));
            }
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const int i = i0 + 4 * u;
                if (i < hi) {
                    const float e[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = (e[j] + 1.0f) / 2.0f;              // box_utils.py:79
                        const bool after = (v < pv) || (v == pv && i + j > pi);
                        if (after && better(v, i + j, bv, bi)) { bv = v; bi = i + j; }
                    }
                }
            }
        }
    } else {
        for (int i = lo; i < hi; ++i) {
            const float v = (h[i] + 1.0f) / 2.0f;
            const bool after = (v < pv) || (v == pv && i > pi);
            if (after && better(v, i, bv, bi)) { bv = v; bi = i; }
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ heat, int hw, int width, int height,
                                                     int k, float* __restrict__ kp_px, float* __restrict__ kp_norm,
                                                     int32_t* __restrict__ topk_idx) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* h = heat + (int64_t)blockIdx.x * hw;
    const int per = VEC ? hw / 256 : (hw + 255) / 256;
    const int lo = tid * per, hi = (lo + per) < hw ? (lo + per) : hw;
    float cv;
    int ci;
    scan_slice<VEC>(h, lo, hi, INFINITY, -1, cv, ci);
    float sx = 0.f, sy = 0.f;
    for (int round = 0; round < k; ++round) {
        float bv = cv;
        int bi = ci;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
        __syncthreads();
        float fv = sv[0];
        int fi = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (better(sv[w], si[w], fv, fi)) { fv = sv[w]; fi = si[w]; }
        __syncthreads();
        if (tid == 0 && topk_idx) topk_idx[(int64_t)blockIdx.x * k + round] = fi;
        sx += (float)(fi % width);
        sy += (float)(fi / width);
        // refill the winner's candidate: its WAVE rescans that thread's slice cooperatively (one batch of coalesced
        // loads + a shuffle reduce) -- a single thread walking its 196 floats costs ~7 dependent L2 round trips per round
        const int owner = fi / per;
        if ((owner >> 6) == wid) {
            const int slo = owner * per, shi = (slo + per) < hw ? (slo + per) : hw;
            float rv = -INFINITY;
            int ri = 0x7fffffff;
            for (int i = slo + lane; i < shi; i += 64) {
                const float v = (h[i] + 1.0f) / 2.0f;
                const bool after = (v < fv) || (v == fv && i > fi);
                if (after && better(v, i, rv, ri)) { rv = v; ri = i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(rv, o);
                const int oi = __shfl_xor(ri, o);
                if (better(ov, oi, rv, ri)) { rv = ov; ri = oi; }
            }
            if (lane == (owner & 63)) { cv = rv; ci = ri; }
        }
    }
    if (tid == 0) {
        const float mx = sx / (float)k, my = sy / (float)k;    // xs.float().mean(dim=2)
        kp_px[blockIdx.x * 2 + 0] = mx;
        kp_px[blockIdx.x * 2 + 1] = my;
        if (kp_norm) {
            kp_norm[blockIdx.x * 2 + 0] = (mx / (float)width) * 2.0f - 1.0f;   // box_utils.py:105-108
            kp_norm[blockIdx.x * 2 + 1] = (my / (float)height) * 2.0f - 1.0f;
        }
    }
}

}  // namespace

extern "C" int bd_decode_topk(const float* heat, int n_maps, int height, int width, int k, float* kp_px,
                              float* kp_norm, int32_t* topk_idx, void* stream) {
    if (!heat || !kp_px) return BD_ERR_NULL;
    if (n_maps <= 0 || height <= 0 || width <= 0 || k <= 0 || (int64_t)k > (int64_t)height * width) return BD_ERR_SHAPE;
    const int hw = height * width;
    if (hw % 1024 == 0 && ((uintptr_t)heat & 15) == 0)      // 256 slices of a multiple of 4 floats: float4 scans
        hipLaunchKernelGGL(decode_kernel<true>, dim3(n_maps), dim3(256), 0, (hipStream_t)stream, heat, hw, width, height,
                           k, kp_px, kp_norm, topk_idx);
    else
        hipLaunchKernelGGL(decode_kernel<false>, dim3(n_maps), dim3(256), 0, (hipStream_t)stream, heat, hw, width, height,
                           k, kp_px, kp_norm, topk_idx);
    BD_CHECK_LAUNCH();
    return BD_OK;
}
