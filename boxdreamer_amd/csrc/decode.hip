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

// Round 6: the map held in REGISTERS.  decode_kernel above walks k = 20 dependent rounds of (workgroup arg-max over ds_bpermute shuffles and
// two barriers, rescan of the winner's slice from L2): ~2.8 us per round, 56 us per launch whatever the number of maps -- 1.5 % of a one-pose
// step.  Here 1024 threads load the whole map once, element i to thread i % 1024 (coalesced; the pixels of a heat-map peak -- neighbours in
// a row, rows 224 apart -- land in different threads), keep their NE values in registers and their two best candidates in (value desc,
// index asc) order.  A round is a wave arg-max on DPP row shifts / broadcasts (no LDS round trips), 16 LDS slots double-buffered by round
// parity (one barrier), a 16-lane DPP combine; the winner promotes its second candidate and rescans its registers only after it has won
// twice (behind a wave-uniform branch).  The picks collect in LDS; their (x, y) sums -- integers below 2^24: exact in any order -- and the
// index list leave at the end, so no integer division sits in the loop.  All per-lane conditions are bitwise, not short-circuit (see
// better_nb).  A one-wave-per-SIMD variant (256 threads x 196 registers) was slower: the scans and the rounds are dependent chains of
// VALU -> SALU -> VALU hand-offs (~10-20 cycles per instruction for a lone wave; a build cut after each phase: loads 5 us, first scan 27 us, rounds 31 us),
// which four waves per SIMD interleave.  Same total order, same tie rule, same arithmetic as decode_kernel: bit-identical corners and
// index lists (tests/test_gpu_ops.py, tests/test_gpu_fuzz.py).
// `better` without short-circuit evaluation: hipcc compiles && / || on per-lane conditions into EXEC-mask branches (a dozen scalar
// instructions and a branch per element of the unrolled register scans below); bitwise forms stay two compares and an s_and / s_or
__device__ __forceinline__ bool better_nb(float v, int i, float bv, int bi) { return (v > bv) | ((v == bv) & (i < bi)); }

// (value, index) arg-max steps on the VALU's data-parallel primitives: lane l takes the better of its pair and the pair of the lane the DPP
// control names (row_shr:n = 0x110 + n inside a row of 16 lanes, row_bcast15 / row_bcast31 across rows; lanes without a source keep their own)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_best(float& v, int& i) {
    const int vb = __builtin_bit_cast(int, v);
    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(vb, vb, CTRL, ROW_MASK, 0xF, false));
    const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, ROW_MASK, 0xF, false);
    const bool b = better_nb(ov, oi, v, i);
    v = b ? ov : v;
    i = b ? oi : i;
}
// best pair of a row of 16 lanes, in its lane 15
__device__ __forceinline__ void row_best(float& v, int& i) {
    dpp_best<0x111, 0xF>(v, i);
    dpp_best<0x112, 0xF>(v, i);
    dpp_best<0x114, 0xF>(v, i);
    dpp_best<0x118, 0xF>(v, i);
}
// best pair of the wave, broadcast to every lane (ds_bpermute shuffles cost an LDS round trip per step: six dependent ones per round)
__device__ __forceinline__ void wave_best(float& v, int& i) {
    row_best(v, i);
    dpp_best<0x142, 0xA>(v, i);         // row_bcast15 into rows 1, 3
    dpp_best<0x143, 0xC>(v, i);         // row_bcast31 into rows 2, 3
    v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    i = __builtin_amdgcn_readlane(i, 63);
}

// the two best of a thread's register-resident elements strictly after (pv, pi); element j of thread t is map index t + 1024 j
template <int NE>
__device__ __forceinline__ void regs_rescan(const float (&v)[NE], int tid, float pv, int pi, float& c1v, int& c1i, float& c2v, int& c2i) {
    c1v = c2v = -INFINITY;
    c1i = c2i = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int i = tid + 1024 * j;
        const float x = v[j];                                   // (elements past the map are NaN: every comparison below is false)
        const bool ok = (x < pv) | ((x == pv) & (i > pi));
        const bool b1 = ok & better_nb(x, i, c1v, c1i), b2 = ok & !b1 & better_nb(x, i, c2v, c2i);
        c2v = b1 ? c1v : (b2 ? x : c2v);
        c2i = b1 ? c1i : (b2 ? i : c2i);
        c1v = b1 ? x : c1v;
        c1i = b1 ? i : c1i;
    }
}

template <int NE>
__global__ __launch_bounds__(1024) void decode_kernel_regs(const float* __restrict__ heat, int hw, int width, int height, int k,
                                                          float* __restrict__ kp_px, float* __restrict__ kp_norm, int32_t* __restrict__ topk_idx) {
    __shared__ float sv[2][16];
    __shared__ int si[2][16];
    __shared__ int s_idx[64];                  // the picks, in order (k <= 64: host-checked)
    __shared__ float s_xy[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float* h = heat + (int64_t)blockIdx.x * hw;
    float v[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int i = tid + 1024 * j;
        const float x = (h[i < hw ? i : tid] + 1.0f) / 2.0f;     // box_utils.py:79
        v[j] = i < hw ? x : __builtin_nanf("");
    }
    float c1v, c2v;
    int c1i, c2i;
    // first scan, unfiltered: indices ascend with j, so STRICT comparisons keep the lower index among equal values (NaN past the map: false)
    c1v = c2v = -INFINITY;
    c1i = c2i = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int i = tid + 1024 * j;
        const float x = v[j];
        const bool b1 = x > c1v, b2 = !b1 & (x > c2v);
        c2v = b1 ? c1v : (b2 ? x : c2v);
        c2i = b1 ? c1i : (b2 ? i : c2i);
        c1v = b1 ? x : c1v;
        c1i = b1 ? i : c1i;
    }
    for (int round = 0; round < k; ++round) {
        float bv = c1v;
        int bi = c1i;
        wave_best(bv, bi);
        const int par = round & 1;
        if (lane == 0) { sv[par][wid] = bv; si[par][wid] = bi; }
        __syncthreads();
        // the 16 wave results: every row of 16 lanes reads them (one slot per lane) and reduces them with four DPP steps
        float fv = sv[par][lane & 15];
        int fi = si[par][lane & 15];
        row_best(fv, fi);
        fv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fv), 15));
        fi = __builtin_amdgcn_readlane(fi, 15);
        if (tid == 0) s_idx[round] = fi;
        // this thread's candidate won: its second best moves up; none left -> look through the registers again.  The rescan sits behind a
        // WAVE-UNIFORM branch: as a per-lane branch around side-effect-free code hipcc turned it into straight-line code + selects, i.e.
        // every wave rescanned its NE registers in every round (measured: 108 us per launch instead of 15)
        const bool won = c1i == fi, dry = won && c2i == 0x7fffffff;
        if (won && !dry) { c1v = c2v; c1i = c2i; c2v = -INFINITY; c2i = 0x7fffffff; }
        if (__builtin_amdgcn_ballot_w64(dry) != 0) {
            float n1v, n2v;
            int n1i, n2i;
            regs_rescan<NE>(v, tid, fv, fi, n1v, n1i, n2v, n2i);
            if (dry) { c1v = n1v; c1i = n1i; c2v = n2v; c2i = n2i; }
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int fi = tid < k ? s_idx[tid] : 0;
        if (topk_idx && tid < k) topk_idx[(int64_t)blockIdx.x * k + tid] = fi;
        s_xy[0][tid] = tid < k ? (float)(fi % width) : 0.f;
        s_xy[1][tid] = tid < k ? (float)(fi / width) : 0.f;
    }
    __syncthreads();
    if (tid == 0) {
        float sx = 0.f, sy = 0.f;              // (xs.float().mean(dim=2): sums of integers below 2^24 -- exact, order-free)
        for (int r = 0; r < k; ++r) { sx += s_xy[0][r]; sy += s_xy[1][r]; }
        const float mx = sx / (float)k, my = sy / (float)k;
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
#ifndef BD_DECODE_SLICES      // (measurement builds only: tools/r6_decode_probe.py times the round-1 slice kernel against the register-resident one)
    // the register-resident form: at most 49 elements per thread (224 x 224), k small against a thread's share
    if (hw <= 49 * 1024 && k <= 64) {
        if (hw <= 16 * 1024)
            hipLaunchKernelGGL(decode_kernel_regs<16>, dim3(n_maps), dim3(1024), 0, (hipStream_t)stream, heat, hw, width, height, k, kp_px, kp_norm, topk_idx);
        else
            hipLaunchKernelGGL(decode_kernel_regs<49>, dim3(n_maps), dim3(1024), 0, (hipStream_t)stream, heat, hw, width, height, k, kp_px, kp_norm, topk_idx);
        BD_CHECK_LAUNCH();
        return BD_OK;
    }
#endif
    if (hw % 1024 == 0 && ((uintptr_t)heat & 15) == 0)      // 256 slices of a multiple of 4 floats: float4 scans
        hipLaunchKernelGGL(decode_kernel<true>, dim3(n_maps), dim3(256), 0, (hipStream_t)stream, heat, hw, width, height,
                           k, kp_px, kp_norm, topk_idx);
    else
        hipLaunchKernelGGL(decode_kernel<false>, dim3(n_maps), dim3(256), 0, (hipStream_t)stream, heat, hw, width, height,
                           k, kp_px, kp_norm, topk_idx);
    BD_CHECK_LAUNCH();
    return BD_OK;
}
