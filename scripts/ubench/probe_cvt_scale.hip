// probe_cvt_scale.hip — does v_cvt_scalef32_pk_{f32,bf16,f16}_fp8 multiply by the FULL fp32 scale or only by its exponent?
// (round 4: if the mantissa counts, bf16(q * s) of two 4-bit codes is ONE instruction instead of cvt + fma + cvt_pk.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t* src, const float* scale, uint32_t* out) {
    const int l = threadIdx.x;
    const uint32_t s = src[l];
    const float sc = scale[l];
    const f2 a = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(s, sc, false);
    out[l * 4 + 0] = __builtin_bit_cast(uint32_t, a.x);
    out[l * 4 + 1] = __builtin_bit_cast(uint32_t, a.y);
    out[l * 4 + 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(s, sc, false));
    out[l * 4 + 3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(s, sc, false));
}
int main() {
    uint32_t hs[64]; float hsc[64]; uint32_t ho[256];
    const float scales[8] = {512.f, 768.f, 640.f, 512.f * 1.0009765625f, 1.5f, 3.f, 0.75f, 1000.f};
    for (int i = 0; i < 64; ++i) { hs[i] = (uint32_t)(i & 15) | ((uint32_t)((i * 7) & 15) << 8); hsc[i] = scales[i >> 3]; }
    uint32_t *ds, *dout; float* dsc;
    hipMalloc(&ds, sizeof(hs)); hipMalloc(&dsc, sizeof(hsc)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice); hipMemcpy(dsc, hsc, sizeof(hsc), hipMemcpyHostToDevice);
    k<<<1, 64>>>(ds, dsc, dout);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    int full = 0, expo = 0, other = 0;
    for (int i = 0; i < 64; ++i) {
        const int q0 = hs[i] & 15, q1 = (hs[i] >> 8) & 15;
        const float got0 = __builtin_bit_cast(float, ho[i * 4]), got1 = __builtin_bit_cast(float, ho[i * 4 + 1]);
        const float want_full0 = q0 / 512.f * hsc[i];
        int e; frexpf(hsc[i], &e); const float p2 = ldexpf(1.f, e - 1);
        const float want_exp0 = q0 / 512.f * p2;
        if (q0 == 0) continue;
        if (got0 == want_full0 && want_full0 != want_exp0) full++; else if (got0 == want_exp0 && want_full0 != want_exp0) expo++; else if (want_full0 != want_exp0) other++;
        if ((i & 7) == 3) printf("lane %2d scale %g: codes (%d, %d) -> f32 (%g, %g) bf16pk %08x f16pk %08x | full-scale model %g, exponent-only model %g\n", i, hsc[i], q0, q1, got0, got1, ho[i * 4 + 2], ho[i * 4 + 3], want_full0, want_exp0);
    }
    printf("RESULT cvt_scalef32 scale semantics: full-mantissa matches %d, exponent-only matches %d, neither %d\n", full, expo, other);
    return 0;
}
