// Round 5: what does the matrix pipe sustain under the 1400 W cap when its OPERANDS CHANGE every instruction, as they do in a GEMM?
// tools/clock_probe.hip (round 2: "a pure MFMA loop sustains 1.9-2.0 PFLOP/s") feeds every MFMA the same two fragments: nothing toggles on the
// operand paths.  Variants here, all without global-memory traffic inside the loop, 256 workgroups x 8 waves (2 per SIMD, the GEMM's consumer count):
//   0  constant operands                                   (round 2's loop)
//   1  operands rotate through 8 + 8 random bf16 fragments held in registers (a new pair every MFMA)
//   2  as 1, and every MFMA is accompanied by one ds_read_b128 of random data from LDS (the GEMM's 0.83 fragment reads per MFMA, rounded up)
//   3  as 2 with ZERO data in registers and LDS          (same instruction stream, nothing toggles)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_probe/mfma_toggle_probe tools/mfma_toggle_probe.hip ;  tools/_probe/mfma_toggle_probe <variant> <seconds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef unsigned u128 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// a random bf16 in roughly N(0, 1): sign | exponent 120..127 | 7 random mantissa bits
__device__ __forceinline__ unsigned short rnd_bf16(unsigned h) { return (unsigned short)(((h & 1u) << 15) | ((120u + ((h >> 1) & 7u)) << 7) | ((h >> 8) & 0x7fu)); }

template <int V>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* st, int iters, unsigned seed) {
    __shared__ u128 lds[6144];                       // 96 KiB (64 KiB of fragment data used): ONE workgroup per CU in every variant
    const int tid = threadIdx.x, lane = tid & 63;
    bf16x8 a[8], b[8];
    for (int f = 0; f < 8; ++f)
        for (int i = 0; i < 8; ++i) {
            const unsigned h = hash(seed + tid * 131u + f * 17u + i);
            unsigned short ua = V == 3 ? 0 : rnd_bf16(h), ub = V == 3 ? 0 : rnd_bf16(hash(h));
            if (V == 0) { ua = rnd_bf16(hash(tid)); ub = rnd_bf16(hash(tid + 7u)); }
            a[f][i] = __builtin_bit_cast(__bf16, ua);
            b[f][i] = __builtin_bit_cast(__bf16, ub);
        }
    for (int i = tid; i < 4096; i += 512) {
        u128 v;
        for (int e = 0; e < 4; ++e) { const unsigned h = hash(seed * 3u + i * 4u + e); v[e] = V == 3 ? 0u : ((unsigned)rnd_bf16(h) | ((unsigned)rnd_bf16(hash(h)) << 16)); }
        lds[i] = v;
    }
    f32x16 c[4] = {{0}, {0}, {0}, {0}};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned idx = (unsigned)(tid * 5) & 4095u;
    for (int it8 = 0; it8 < iters; it8 += 8) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            bf16x8 af = a[V == 0 ? 0 : f], bf = b[V == 0 ? 0 : (f + it) & 7];
            if (V >= 2) {
                const u128 d = lds[idx];
                idx = (idx + 64u) & 4095u;
                af = __builtin_bit_cast(bf16x8, d);   // the fragment read from LDS IS the A operand: the read cannot be dropped
            }
            c[f & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, c[f & 3], 0, 0, 0);
        }
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x16 s = c[0] + c[1] + c[2] + c[3];
    float acc = 0; for (int i = 0; i < 16; ++i) acc += s[i];
    out[blockIdx.x * 512 + tid] = acc;
    if (tid == 0) { st[blockIdx.x * 2] = t1 - t0; st[blockIdx.x * 2 + 1] = r1 - r0; }
    (void)lane;
}

template <int V> void run(float seconds) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    unsigned long long* st; hipMallocManaged(&st, 256 * 16);
    const int iters = 40000, wgs = 256;                 // 8 MFMAs per iteration and wave
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 8; ++r) k<V><<<wgs, 512>>>(out, st, iters, 1u + r);      // warm the clock governor (~100 ms)
    hipDeviceSynchronize();
    int reps = 0;
    const auto w0 = std::chrono::steady_clock::now();
    hipEventRecord(e0);
    while (std::chrono::duration<float>(std::chrono::steady_clock::now() - w0).count() < seconds) {
        for (int r = 0; r < 4; ++r) k<V><<<wgs, 512>>>(out, st, iters, 100u + reps + r);
        reps += 4;
        hipStreamSynchronize(0);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double cyc = 0, rt = 0; for (int b = 0; b < wgs; ++b) { cyc += st[b * 2]; rt += st[b * 2 + 1]; }
    // 8 waves per workgroup = 2 per SIMD: per-SIMD cycles per MFMA = wave cycles / (MFMAs per wave x 2 waves)
    // in-kernel rate: every workgroup's 8 waves x 8 iters MFMAs of 32768 flop during its own (s_memrealtime, 100 MHz) loop time
    const double loop_s = rt / wgs / 1e8;
    printf("variant %d: %d launches of %.2f ms (loop %.2f ms), %.2f shader cycles per MFMA and SIMD, delivered %.3f GHz, in-loop %.0f TFLOP/s, wall %.0f TFLOP/s (dense bf16 peak at 2.4 GHz: 2517)\n",
           V, reps, ms / reps, loop_s * 1e3, cyc / wgs / (8.0 * iters) / 2.0, cyc / rt * 0.1, (double)wgs * 8 * 8.0 * iters * 32768.0 / loop_s / 1e12,
           (double)wgs * 8 * 8.0 * iters * 32768.0 * reps / (ms * 1e9));
}

int main(int argc, char** argv) {
    char bus[64] = {0};
    hipDeviceGetPCIBusId(bus, 63, 0);
    printf("pci %s\n", bus);
    const int v = argc > 1 ? atoi(argv[1]) : 1;
    const float sec = argc > 2 ? (float)atof(argv[2]) : 3.0f;
    if (sec <= 0) return 0;
    if (v == 0) run<0>(sec); else if (v == 1) run<1>(sec); else if (v == 2) run<2>(sec); else run<3>(sec);
    return 0;
}
