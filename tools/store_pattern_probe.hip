// Micro-benchmark: cycles a workgroup of 8 waves needs to push one 256x192 16-bit tile (96 KiB) to global memory, by store
// pattern, alone on the chip and with all 256 CUs doing the same.  In-kernel s_memtime, many tiles per workgroup.
// (Measurement tool for the GEMM epilogue analysis; not part of the library.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_probe/store_pattern_probe tools/store_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(16))) unsigned int u128;

// output matrix: rows x 2304 columns of 2 bytes (row pitch 4608 B); tile = 256 rows x 192 columns (384 B per row)
// wave w: wm = w & 3 (64 rows each), wn = w >> 2 (96 columns = 192 B each)
template <int PAT>
__global__ __launch_bounds__(512) void k(unsigned char* out, int tiles, int wgs, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wm = w & 3, wn = w >> 2;
    const size_t pitch = 4608;
    u128 v = {(unsigned)threadIdx.x, 1u, 2u, 3u};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
        const int tile = t * wgs + blockIdx.x;                 // tile raster: 12 column tiles per 256-row band
        unsigned char* base = out + (size_t)(tile / 12) * 256 * pitch + (size_t)(tile % 12) * 384;
        if (PAT == 0) {            // hypothetical: contiguous 96 KiB
            unsigned char* b = out + (size_t)tile * 98304 + w * 12288;
#pragma unroll
            for (int i = 0; i < 12; ++i) *(u128*)(b + i * 1024 + lane * 16) = v;
        } else if (PAT == 1) {     // current epilogue: 16 rows x 64 B per instruction
            unsigned char* b = base + (size_t)(wm * 64) * pitch + wn * 192;
#pragma unroll
            for (int ih = 0; ih < 4; ++ih)
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) *(u128*)(b + (size_t)(ih * 16 + (lane >> 2)) * pitch + cb * 64 + (lane & 3) * 16) = v;
        } else if (PAT == 2) {     // row runs of the wave tile: 5.33 rows x 192 B per instruction
            unsigned char* b = base + (size_t)(wm * 64) * pitch + wn * 192;
#pragma unroll
            for (int ih = 0; ih < 4; ++ih)
#pragma unroll
                for (int u = 0; u < 3; ++u) { const int id = u * 64 + lane; *(u128*)(b + (size_t)(ih * 16 + id / 12) * pitch + (id % 12) * 16) = v; }
        } else if (PAT == 3) {     // whole tile rows: wave w owns rows 32 w .. 32 w + 31, 2.67 rows x 384 B per instruction
            unsigned char* b = base + (size_t)(w * 32) * pitch;
#pragma unroll
            for (int u = 0; u < 12; ++u) { const int id = u * 64 + lane; *(u128*)(b + (size_t)(id / 24) * pitch + (id % 24) * 16) = v; }
        } else if (PAT == 4) {     // 8 rows x 128 B aligned lines (what a 64-column-aligned layout would give)
            unsigned char* b = out + (size_t)(tile / 18) * 256 * pitch + (size_t)(tile % 18) * 256 + (size_t)(wm * 64) * pitch + wn * 128;
#pragma unroll
            for (int i = 0; i < 8; ++i) *(u128*)(b + (size_t)(i * 8 + (lane >> 3)) * pitch + (lane & 7) * 16) = v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const size_t total = (size_t)3 << 30;
    unsigned char* buf; hipMalloc(&buf, total); hipMemset(buf, 0, total);
    unsigned long long* cyc; hipMallocManaged(&cyc, 256 * 8);
    printf("pattern,wgs,tiles_per_wg,cycles_per_tile(s_memtime 100MHz ticks x clk?),us_total,GB/s\n");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 5; ++pat)
        for (int wgs : {1, 8, 32, 256}) {
            const int tiles = wgs == 256 ? 96 : 400;
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                switch (pat) {
                    case 0: k<0><<<wgs, 512>>>(buf, tiles, wgs, cyc); break;
                    case 1: k<1><<<wgs, 512>>>(buf, tiles, wgs, cyc); break;
                    case 2: k<2><<<wgs, 512>>>(buf, tiles, wgs, cyc); break;
                    case 3: k<3><<<wgs, 512>>>(buf, tiles, wgs, cyc); break;
                    case 4: k<4><<<wgs, 512>>>(buf, tiles, wgs, cyc); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            double mean = 0; for (int b = 0; b < wgs; ++b) mean += (double)cyc[b]; mean /= wgs;
            const double bytes = (double)wgs * tiles * (pat == 4 ? 65536.0 : 98304.0);
            printf("%d,%d,%d,%.0f ticks/tile,%.1f us,%.0f GB/s, %.2f us/tile\n", pat, wgs, tiles, mean / tiles, ms * 1e3, bytes / (ms * 1e6), ms * 1e3 / tiles);
        }
    return 0;
}
