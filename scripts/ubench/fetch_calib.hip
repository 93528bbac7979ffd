// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this library: every
// kernel reads (writes) a KNOWN number of bytes from a buffer larger than the 256 MiB Infinity Cache, once, and prints it; the
// counter pass (`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- scripts/ubench/fetch_calib`, then WRITE_SIZE) gives the counter
// per dispatch in KiB.  Patterns: 16 B / lane coalesced (the GEMV weight stream, LDS-DMA of x), 4 B / lane with 32 lanes on one
// 128-byte row segment and the two lane halves on adjacent rows (the B-operand loads of gemm_wn_mma), 8 B / lane, and 16-byte
// write-through (sc1) stores (the split-K partial tiles).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void k_read16(const u32x4* p, size_t n, uint32_t* out) {  // n = 16-byte elements
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
__global__ void k_read8(const u32x2* p, size_t n, uint32_t* out) {
    u32x2 acc = {0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if ((acc[0] ^ acc[1]) == 0x12345678u) out[0] = 1;
}
// the tile kernel's B pattern: wave = 32 columns x 2 rows per instruction (lane & 31 = column, lane >> 5 = row), rows `pitch` words apart
__global__ void k_read4_rows(const uint32_t* p, int rows, int pitch, uint32_t* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col0 = (blockIdx.x * (blockDim.x >> 6) + wave) * 32;
    uint32_t acc = 0;
    for (int r = 0; r < rows; r += 2) acc ^= p[(size_t)(r + (lane >> 5)) * pitch + col0 + (lane & 31)];
    if (acc == 0x12345678u) out[0] = 1;
}
__global__ void k_write16_sc1(u32x4* p, size_t n) {  // write-through stores, as the split-K partial tiles travel
    const u32x4 v = {1, 2, 3, 4};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p, (short)0, (int)(n * 16), 0x00020000);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(i * 16), 0, 16);  // aux 16 = sc1
}
__global__ void k_write16(u32x4* p, size_t n) {
    const u32x4 v = {1, 2, 3, 4};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

int main() {
    const size_t bytes = 512ull << 20;  // 512 MiB: twice the Infinity Cache
    uint8_t* buf; hipMalloc(&buf, bytes); hipMemset(buf, 0x5a, bytes);
    uint32_t* out; hipMalloc(&out, 4);
    hipDeviceSynchronize();
    const size_t kib = bytes >> 10;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_read16, dim3(2048), dim3(256), 0, 0, (const u32x4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(k_read8, dim3(2048), dim3(256), 0, 0, (const u32x2*)buf, bytes / 8, out);
        // 16384 columns x 8192 rows of 4-byte words = 512 MiB; 512 waves x 32 columns
        hipLaunchKernelGGL(k_read4_rows, dim3(128), dim3(256), 0, 0, (const uint32_t*)buf, 8192, 16384, out);
        hipLaunchKernelGGL(k_write16, dim3(2048), dim3(256), 0, 0, (u32x4*)buf, bytes / 16);
        hipLaunchKernelGGL(k_write16_sc1, dim3(2048), dim3(256), 0, 0, (u32x4*)buf, bytes / 16);
        hipDeviceSynchronize();
    }
    printf("every kernel moves %zu KiB once (k_read16, k_read8, k_read4_rows read; k_write16, k_write16_sc1 write)\n", kib);
    return 0;
}
