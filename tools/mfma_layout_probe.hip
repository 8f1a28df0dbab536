// Empirical layout probe for the gfx950 fp8 MFMAs (no ISA document in this image).
// Packs small-integer (exactly representable in e4m3) A[32][K], B[K][32] under several per-lane layout hypotheses and
// checks D = A.B against the host.  Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_layout_probe.hip -o /tmp/probe_mfma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((__vector_size__(8 * sizeof(int)))) int i32x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__global__ void k_mx(const i32x8* a, const i32x8* b, float* c) {
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) c[threadIdx.x * 16 + r] = acc[r];
}
__global__ void k_f8(const long* a, const long* b, float* c) {
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) c[threadIdx.x * 16 + r] = acc[r];
}
__global__ void k_cvt(const float* x, uint8_t* o, int n) {
    int i = threadIdx.x;
    if (i * 4 < n) {
        int r = 0;
        r = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i], x[4 * i + 1], r, false);
        r = __builtin_amdgcn_cvt_pk_fp8_f32(x[4 * i + 2], x[4 * i + 3], r, true);
        ((int*)o)[i] = r;
    }
}

static uint8_t f8[64];   // e4m3 encodings of the integers -8..8 obtained from the hardware converter
static uint8_t enc(int v) { return f8[v + 8]; }

int kmap(int hyp, int half, int byte, int K) {           // k index held in `byte` of lane-half `half`
    if (K == 16) return 8 * half + byte;                  // 32x32x16: 8 bytes per lane
    switch (hyp) {
        case 0: return 32 * half + byte;                                        // contiguous 32 per half
        case 1: return (byte < 16) ? 16 * half + byte : 32 + 16 * half + (byte - 16);
        case 2: return 8 * half + 16 * (byte / 8) + (byte % 8);                 // four stacked K=16 steps
        case 3: return 16 * half + 32 * (byte / 16) + (byte % 16);              // two stacked K=32 steps
        default: return 2 * byte + half;
    }
}

int main() {
    // fp8 encodings from the hardware converter
    float hx[20]; for (int i = 0; i < 20; ++i) hx[i] = (i < 17) ? (float)(i - 8) : 0.f;
    float* dx; uint8_t* df; hipMalloc(&dx, sizeof(hx)); hipMalloc(&df, 64);
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    k_cvt<<<1, 64>>>(dx, df, 20); hipMemcpy(f8, df, 20, hipMemcpyDeviceToHost);
    printf("e4m3 codes for -8..8:"); for (int i = 0; i < 17; ++i) printf(" %02x", f8[i]); printf("\n");
    for (int K : {16, 64}) {
        int A[32][64], B[64][32];
        srand(7);
        for (int i = 0; i < 32; ++i) for (int k = 0; k < K; ++k) A[i][k] = rand() % 9 - 4;
        for (int k = 0; k < K; ++k) for (int j = 0; j < 32; ++j) B[k][j] = rand() % 7 - 3;
        float ref[32][32];
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int k = 0; k < K; ++k) s += A[i][k] * B[k][j]; ref[i][j] = (float)s; }
        const int nb = K == 16 ? 8 : 32;
        for (int hyp = 0; hyp < (K == 16 ? 1 : 5); ++hyp) {
            uint8_t ha[64 * 32] = {0}, hb[64 * 32] = {0};
            for (int l = 0; l < 64; ++l) for (int b = 0; b < nb; ++b) {
                const int k = kmap(hyp, l >> 5, b, K);
                ha[l * nb + b] = enc(A[l & 31][k]);
                hb[l * nb + b] = enc(B[k][l & 31]);
            }
            uint8_t *da, *db; float* dc; hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dc, 64 * 16 * 4);
            hipMemcpy(da, ha, 64 * nb, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64 * nb, hipMemcpyHostToDevice);
            if (K == 16) k_f8<<<1, 64>>>((const long*)da, (const long*)db, dc);
            else k_mx<<<1, 64>>>((const i32x8*)da, (const i32x8*)db, dc);
            float hc[64 * 16]; hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost);
            int bad = 0;
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;     // C/D layout (dtype independent)
                if (hc[l * 16 + r] != ref[row][col]) ++bad;
            }
            printf("K=%d hypothesis %d: %s (%d mismatches)\n", K, hyp, bad ? "no" : "MATCH", bad);
        }
    }
    return 0;
}
