// Do gfx950's fp8 conversions saturate?  v_cvt_scalef32_pk_fp8_f16 (f16 pair -> e4m3 pair, the GEMM's in-register q8 plane) and
// v_cvt_pk_fp8_f32, with MODE.FP16_OVFL clear and set (s_setreg hwreg(HW_REG_MODE, 23, 1)), plus what v_cvt_f16_f32 does to 1e6
// in either mode.  (measurement tool: hipcc --offload-arch=gfx950 -o fp8_sat_probe fp8_sat_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(float* out, const float* in, int set_ovfl) {
    if (set_ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);      // MODE[23] = FP16_OVFL
    for (int i = 0; i < 8; i += 2) {
        float a = in[i], b = in[i + 1];
        h2 v = {(_Float16)a, (_Float16)b};
        asm volatile("" : "+v"(v));
        s2 o = {0, 0};
        o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, v, 1.0f, false);
        const unsigned u = __builtin_bit_cast(unsigned, o);
        out[i * 4 + 0] = __builtin_amdgcn_cvt_f32_fp8((int)u, 0);
        out[i * 4 + 1] = __builtin_amdgcn_cvt_f32_fp8((int)u, 1);
        int p = 0;
        p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, p, false);
        out[i * 4 + 2] = __builtin_amdgcn_cvt_f32_fp8(p, 0);
        out[i * 4 + 3] = __builtin_amdgcn_cvt_f32_fp8(p, 1);
        out[i * 4 + 4] = (float)v.x;       // the f16 image itself (v_cvt_f16_f32 of a large value: inf or 65504?)
        out[i * 4 + 5] = (float)v.y;
        // packed clamp (the alternative to the mode bit): v_pk_max_f16 / v_pk_min_f16
        h2 c = __builtin_elementwise_min(__builtin_elementwise_max(v, (h2){(_Float16)-448.0f, (_Float16)-448.0f}), (h2){(_Float16)448.0f, (_Float16)448.0f});
        s2 oc = {0, 0};
        oc = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(oc, c, 1.0f, false);
        const unsigned uc = __builtin_bit_cast(unsigned, oc);
        out[i * 4 + 6] = __builtin_amdgcn_cvt_f32_fp8((int)uc, 0);
        out[i * 4 + 7] = __builtin_amdgcn_cvt_f32_fp8((int)uc, 1);
    }
}
int main() {
    const float vals[8] = {300.0f, 500.0f, -449.0f, 1000.0f, 60000.0f, -70000.0f, 1e6f, 0.3f};
    float *d, *di; hipMalloc(&d, 4096); hipMalloc(&di, 64); hipMemcpy(di, vals, sizeof(vals), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(d, 0, 4096);
        k<<<1, 1>>>(d, di, mode);
        float h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("MODE.FP16_OVFL = %d\n", mode);
        for (int i = 0; i < 8; i += 2) {
            const float* p = h + i * 4;
            printf("  in (%g, %g): f16 image (%g, %g)  cvt_scalef32_pk_fp8_f16 -> (%g, %g)  cvt_pk_fp8_f32 -> (%g, %g)  after packed clamp -> (%g, %g)\n",
                   vals[i], vals[i + 1], p[4], p[5], p[0], p[1], p[2], p[3], p[6], p[7]);
        }
    }
    return 0;
}
