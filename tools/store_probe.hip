// Micro-benchmark: how fast can a CU push 16-byte global stores, alone and with every other CU doing the same?
// (Measurement tool for the GEMM epilogue analysis; not part of the library.)   hipcc --offload-arch=gfx950 -O3 -o store_probe store_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(16))) unsigned int u128;

// each workgroup (nw waves) writes `iters` rounds of nw KiB; consecutive rounds advance through its private region
template <bool NT>
__global__ void store_kernel(u128* out, int iters, size_t wg_stride16, int spread) {
    const int tid = threadIdx.x;
    u128 v = {(unsigned)tid, 1u, 2u, 3u};
    u128* base = out + (size_t)blockIdx.x * spread * wg_stride16;
    for (int i = 0; i < iters; ++i) {
        u128* dst = base + (size_t)i * blockDim.x + tid;
        if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
}
__global__ void load_kernel(const u128* in, u128* sink, int iters, size_t wg_stride16, int spread) {
    const int tid = threadIdx.x;
    const u128* base = in + (size_t)blockIdx.x * spread * wg_stride16;
    u128 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) acc += base[(size_t)i * blockDim.x + tid];
    if (acc[0] == 0x12345) sink[tid] = acc;
}

int main(int argc, char** argv) {
    const size_t total = (size_t)2 << 30;          // 2 GiB buffer
    u128* buf; hipMalloc(&buf, total);
    hipMemset(buf, 0, total);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cus = 256;
    printf("kind,nt,wgs,waves_per_wg,KiB_per_wg,us,GB/s,B_per_clk_per_active_CU(@2.1GHz)\n");
    for (int kind = 0; kind < 2; ++kind)
    for (int nt = 0; nt < (kind == 0 ? 2 : 1); ++nt)
    for (int wgs : {256, 128, 64, 32, 8})
    for (int nw : {1, 4, 8}) {
        const int kib = 96;                          // one 256x192 bf16 tile per workgroup
        const int iters = kib / nw;
        const size_t stride16 = (size_t)kib * 1024 / 16;
        const int spread = cus / wgs;                // keep blocks on distinct CUs: block b -> region b*spread (placement is the HW's)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int r = 0; r < 20; ++r) {
                if (kind == 0) { if (nt) store_kernel<true><<<wgs, nw * 64>>>(buf + (size_t)r * cus * stride16, iters, stride16, spread);
                                 else store_kernel<false><<<wgs, nw * 64>>>(buf + (size_t)r * cus * stride16, iters, stride16, spread); }
                else load_kernel<<<wgs, nw * 64>>>(buf + (size_t)r * cus * stride16, buf, iters, stride16, spread);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) {
                const double us = ms * 1e3 / 20, bytes = (double)wgs * kib * 1024;
                printf("%s,%d,%d,%d,%d,%.2f,%.0f,%.1f\n", kind ? "load" : "store", nt, wgs, nw, kib, us, bytes / us / 1e3,
                       bytes / wgs / (us * 2100.0));
            }
        }
    }
    return 0;
}
