// What does s_memtime count?  N back-to-back independent v_mfma_f32_32x32x16_bf16 per wave (4 accumulators) cost 32 shader
// cycles each (8 passes x 4): if the s_memtime delta per MFMA stays ~32 when every SIMD of the chip runs the loop (power-limited
// clocks), s_memtime IS the delivered shader clock and its ratio to s_memrealtime (100 MHz) is the delivered frequency.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_probe/clock_probe tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__global__ __launch_bounds__(256) void k(float* out, unsigned long long* st, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed * (threadIdx.x % 13 + i) * 0.37f - 1.f); b[i] = (__bf16)(seed * ((threadIdx.x * 7 + i) % 11) * 0.21f - 1.f); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x16 s = c0 + c1 + c2 + c3;
    float acc = 0; for (int i = 0; i < 16; ++i) acc += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { st[blockIdx.x * 2] = t1 - t0; st[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    unsigned long long* st; hipMallocManaged(&st, 4096 * 16);
    printf("workgroups(4 waves each),waves/SIMD,launches back to back,shader cycles per MFMA,delivered GHz (s_memtime/s_memrealtime),TFLOP/s\n");
    for (int wgs : {1, 8, 256, 512}) {
        const int iters = 20000;
        const int reps = wgs >= 256 ? 30 : 3;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) k<<<wgs, 256>>>(out, st, iters, 1.0f + r);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double cyc = 0, rt = 0; for (int b = 0; b < wgs; ++b) { cyc += st[b * 2]; rt += st[b * 2 + 1]; }
        const double per = cyc / wgs / (4.0 * iters) / (wgs == 512 ? 2 : 1);   // 2 waves share a SIMD at 512 workgroups
        printf("%d,%d,%d,%.2f,%.3f,%.0f\n", wgs, wgs == 512 ? 2 : 1, reps, per, cyc / rt * 0.1, (double)wgs * 4 * 4.0 * iters * 32768.0 * reps / (ms * 1e9));
    }
    return 0;
}
