// What does v_cvt_scalef32_pk_fp8_f16 do with its scale operand?  (measurement tool; hipcc --offload-arch=gfx950 -o cvt_probe cvt_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out) {
    const float vals[8] = {1.0f, 2.0f, 0.3f, -5.0f, 300.0f, 500.0f, 0.001f, 0.0f};
    for (int i = 0; i < 8; i += 2) {
        h2 v = {(_Float16)vals[i], (_Float16)vals[i + 1]};
        for (int si = 0; si < 3; ++si) {
            const float sc = si == 0 ? 1.0f : (si == 1 ? 4.0f : 0.25f);
            s2 o = {0, 0};
            o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, v, sc, false);
            o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, v, sc, true);
            const unsigned u = __builtin_bit_cast(unsigned, o);
            // decode with the non-scaled fp8 -> f32 conversion
            out[(i / 2 * 3 + si) * 4 + 0] = __builtin_amdgcn_cvt_f32_fp8((int)u, 0);
            out[(i / 2 * 3 + si) * 4 + 1] = __builtin_amdgcn_cvt_f32_fp8((int)u, 1);
            out[(i / 2 * 3 + si) * 4 + 2] = __builtin_amdgcn_cvt_f32_fp8((int)u, 2);
            out[(i / 2 * 3 + si) * 4 + 3] = __builtin_amdgcn_cvt_f32_fp8((int)u, 3);
        }
    }
}
int main() {
    float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    k<<<1, 1>>>(d);
    float h[48]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const float vals[8] = {1.0f, 2.0f, 0.3f, -5.0f, 300.0f, 500.0f, 0.001f, 0.0f};
    for (int i = 0; i < 4; ++i) for (int si = 0; si < 3; ++si) {
        const float* p = h + (i * 3 + si) * 4;
        printf("in (%g, %g) scale %g -> lo half (%g, %g) hi half (%g, %g)\n", vals[2 * i], vals[2 * i + 1],
               si == 0 ? 1.0f : (si == 1 ? 4.0f : 0.25f), p[0], p[1], p[2], p[3]);
    }
    return 0;
}
