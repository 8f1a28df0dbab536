// Heatmap -> 2-D corner decode (reference: src/models/utils/box_utils.py:75-110).
//
// Per (sample, corner) map: h = (heat + 1) / 2 in fp32, top-k (k = 20) over H*W, corner = the
// unweighted mean of the k integer (x, y) = (idx % W, idx / W).  `torch.topk` leaves the order of
// equal values unspecified; this kernel pins: larger value first, then LOWER index first.
//
// One 256-thread workgroup per map, k selection rounds.  In each round every thread scans its
// strided share of the map (coalesced, L2-resident: 200 KB per map) for the best element that comes
// strictly AFTER the previous pick in (value desc, index asc) order, then a wave shuffle + LDS
// reduction picks the workgroup's winner.  Deterministic, no atomics, no sorting network.
#include "bd_common.h"

namespace {

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) {
    return (v > bv) || (v == bv && i < bi);
}

__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ heat, int hw, int width, int height,
                                                     int k, float* __restrict__ kp_px, float* __restrict__ kp_norm,
                                                     int32_t* __restrict__ topk_idx) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float pick_v;
    __shared__ int pick_i;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* h = heat + (int64_t)blockIdx.x * hw;
    float pv = INFINITY;
    int pi = -1;
    float sx = 0.f, sy = 0.f;
    for (int round = 0; round < k; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < hw; i += 256) {
            const float v = (h[i] + 1.0f) / 2.0f;             // box_utils.py:79
            const bool after = (v < pv) || (v == pv && i > pi);
            if (after && better(v, i, bv, bi)) { bv = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { sv[wid] = bv; si[wid] = bi; }
        __syncthreads();
        if (tid == 0) {
            float fv = sv[0]; int fi = si[0];
            for (int w = 1; w < 4; ++w) if (better(sv[w], si[w], fv, fi)) { fv = sv[w]; fi = si[w]; }
            pick_v = fv; pick_i = fi;
            if (topk_idx) topk_idx[(int64_t)blockIdx.x * k + round] = fi;
        }
        __syncthreads();
        pv = pick_v; pi = pick_i;
        sx += (float)(pi % width);
        sy += (float)(pi / width);
        __syncthreads();
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
    hipLaunchKernelGGL(decode_kernel, dim3(n_maps), dim3(256), 0, (hipStream_t)stream, heat, height * width, width,
                       height, k, kp_px, kp_norm, topk_idx);
    BD_CHECK_LAUNCH();
    return BD_OK;
}
