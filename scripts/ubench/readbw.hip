// Micro-benchmark: read bandwidth of the access patterns the GEMV uses on a [rows, pitch] int32 matrix.
//   mode 0: linear      — each block streams a contiguous chunk with 16-byte loads
//   mode 1: strip       — block owns `tc` columns (tc*4 bytes per row) and walks all rows (like gemv narrow/wide)
// Usage: readbw <rows> <cols> <pitch_words> <mode> <tc> <rows_per_lane R> <blocks_per_tile_split>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_linear(const u32x4* __restrict__ p, size_t n16, uint32_t* out) {
    size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) { size_t k = i + (size_t)j * 256; if (k < n16) { u32x4 v = p[k]; acc ^= v; } }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

template <int CQ, int R>
__global__ __launch_bounds__(256) void k_strip(const uint32_t* __restrict__ w, int rows, int pitch, int splitk, uint32_t* out) {
    constexpr int G = 64 >> CQ, TC = 4 << CQ, CHUNK = G * R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & ((1 << CQ) - 1), g = lane >> CQ;
    const int tile = blockIdx.x, slice = blockIdx.y;
    const int rows_slice = rows / splitk, rows_wave = rows_slice / 4;
    const uint32_t* base = w + (size_t)(slice * rows_slice + wave * rows_wave + g * R) * pitch + tile * TC + c * 4;
    u32x4 acc = {0, 0, 0, 0};
    const int nch = rows_wave / CHUNK;
    for (int ch = 0; ch < nch; ++ch) {
        u32x4 v[R];
#pragma unroll
        for (int i = 0; i < R; ++i) v[i] = *(const u32x4*)(base + (size_t)(ch * CHUNK + i) * pitch);
#pragma unroll
        for (int i = 0; i < R; ++i) acc ^= v[i];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

int main(int argc, char** argv) {
    int rows = atoi(argv[1]), cols = atoi(argv[2]), pitch = atoi(argv[3]), mode = atoi(argv[4]);
    int tc = argc > 5 ? atoi(argv[5]) : 16, R = argc > 6 ? atoi(argv[6]) : 8, splitk = argc > 7 ? atoi(argv[7]) : 1;
    const int NBUF = 8;  // rotate buffers so that reads come from HBM, not the 256 MiB Infinity Cache
    size_t words = (size_t)rows * pitch;
    size_t need = words * 4;
    int nbuf = NBUF; while ((size_t)nbuf * need < (600u << 20) && nbuf < 64) nbuf *= 2;
    uint32_t* buf; uint32_t* out;
    hipMalloc(&buf, need * nbuf); hipMalloc(&out, 4);
    hipMemset(buf, 0x5a, need * nbuf);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 40;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a, 0);
        for (int it = 0; it < iters; ++it) {
            const uint32_t* p = buf + (size_t)(it % nbuf) * words;
            if (mode == 0) {
                size_t n16 = words / 4;
                hipLaunchKernelGGL(k_linear, dim3((unsigned)((n16 + 2047) / 2048)), dim3(256), 0, 0, (const u32x4*)p, n16, out);
            } else {
                dim3 grid(cols / tc, splitk);
#define L(CQ, RR) hipLaunchKernelGGL((k_strip<CQ, RR>), grid, dim3(256), 0, 0, p, rows, pitch, splitk, out)
                if (tc == 16 && R == 8) L(2, 8); else if (tc == 16 && R == 4) L(2, 4); else if (tc == 16 && R == 16) L(2, 16);
                else if (tc == 64 && R == 4) L(4, 4); else if (tc == 64 && R == 8) L(4, 8); else if (tc == 64 && R == 16) L(4, 16);
                else if (tc == 32 && R == 8) L(3, 8); else if (tc == 32 && R == 16) L(3, 16);
                else { printf("unsupported tc/R\n"); return 1; }
            }
        }
        hipEventRecord(b, 0); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    double bytes = (double)rows * cols * 4;
    printf("rows=%d cols=%d pitch=%d mode=%d tc=%d R=%d splitk=%d : %.2f us/iter  %.1f GB/s (incl. launch gaps)\n", rows, cols, pitch, mode, tc, R,
           splitk, ms * 1e3 / iters, bytes * iters / (ms * 1e-3) / 1e9);
    return 0;
}
