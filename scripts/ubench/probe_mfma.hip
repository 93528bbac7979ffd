// Probe: does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL B inputs?  A = ones, B = subnormal codes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* o) {
    u32x4 a = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};  // ones
    u32x4 b = {0x00050003u, 0x00F00010u, 0x00010001u, 0x000F000Fu};  // subnormals: 5,3,240,16,1,1,15,15 (x 2^-24)
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
    if (threadIdx.x == 0) o[0] = c[0];  // sum over 4 k-blocks x 8 = 4 * 296 * 2^-24
}
int main() {
    float* d; float h;
    hipMalloc(&d, 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("mfma f16 subnormal probe: %.9g (expect %.9g)\n", h, 4.0 * 296.0 / 16777216.0);
    return 0;
}
