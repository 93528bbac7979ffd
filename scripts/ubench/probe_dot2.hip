// Probe: does v_dot2c_f32_f16 keep fp16 SUBNORMAL inputs (needed for the "no magic offset" unpack)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__global__ void k(float* o) {
    uint32_t a = 0x00050003u;  // two fp16 subnormals: 5 * 2^-24, 3 * 2^-24
    uint32_t b = 0x3C003C00u;  // (1.0, 1.0)
    uint32_t c = 0x44004000u;  // (4.0, 2.0)
    o[0] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a), __builtin_bit_cast(h2_t, b), 0.f, false);  // 8 * 2^-24
    o[1] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a), __builtin_bit_cast(h2_t, c), 0.f, false);  // 26 * 2^-24
    o[2] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, 0x64056403u), __builtin_bit_cast(h2_t, b), 0.f, false);  // 2056
}
int main() {
    float* d; float h[3];
    hipMalloc(&d, 12); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("dot2 subnormal probe: %.9g (expect %.9g)  %.9g (expect %.9g)  %.9g (expect 2056)\n", h[0], 8.0 / 16777216.0, h[1], 26.0 / 16777216.0, h[2]);
    return 0;
}
