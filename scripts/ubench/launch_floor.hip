// Micro-benchmark: what a kernel launch costs before any work is done, and how fast the GEMV's strip pattern can be
// READ (no arithmetic) for different block shapes / cache policies.  Per-launch device durations come from
// hipExtLaunchKernel start/stop events (the same clock bench.py uses for roofline.kernel_us).
//   launch_floor                -> table of { kernel, grid, block, mean us, min us }
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void k_empty(uint32_t* out) {
    if (out == (uint32_t*)1) out[0] = 1;
}

// strip read: block = NW waves owns TC = 4 << CQ columns; lane (g = lane >> CQ, c) reads R rows x 16 bytes per chunk
template <int CQ, int R, int NW, bool NT, bool PAIR>
__global__ __launch_bounds__(NW * 64) void k_strip(const uint32_t* __restrict__ w, int rows, int pitch, uint32_t* out) {
    constexpr int G = 64 >> CQ, TC = 4 << CQ, CHUNK = G * R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & ((1 << CQ) - 1), g = lane >> CQ;
    int tile = blockIdx.x;
    if (PAIR && (gridDim.x & 15) == 0) {  // adjacent half-line tiles on one XCD
        const int xcd = tile & 7, idx = tile >> 3;
        tile = (((idx >> 1) << 3) + xcd) * 2 + (idx & 1);
    }
    const int slice = blockIdx.y, splitk = gridDim.y;
    const int rows_slice = rows / splitk;
    const int nch = rows_slice / CHUNK;  // chunks of the slice, dealt round-robin to the waves
    const uint32_t* base = w + (size_t)(slice * rows_slice + g * R) * pitch + tile * TC + c * 4;
    u32x4 acc = {0, 0, 0, 0};
    for (int ch = wave; ch < nch; ch += NW) {
        u32x4 v[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const u32x4* p = (const u32x4*)(base + (size_t)(ch * CHUNK + i) * pitch);
            v[i] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) acc ^= v[i];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

struct Timer {
    std::vector<hipEvent_t> a, b;
    Timer(int n) : a(n), b(n) { for (int i = 0; i < n; ++i) { hipEventCreate(&a[i]); hipEventCreate(&b[i]); } }
};

template <typename F>
static void run(const char* name, dim3 grid, dim3 block, F&& launch, int iters = 60) {
    static Timer t(64);
    for (int i = 0; i < 5; ++i) launch(i, nullptr, nullptr);
    hipDeviceSynchronize();
    for (int i = 0; i < iters; ++i) launch(i, t.a[i], t.b[i]);
    hipDeviceSynchronize();
    std::vector<float> us;
    for (int i = 0; i < iters; ++i) { float ms = 0; hipEventElapsedTime(&ms, t.a[i], t.b[i]); us.push_back(ms * 1e3f); }
    std::sort(us.begin(), us.end());
    double mean = 0; for (float v : us) mean += v; mean /= us.size();
    // wall time of the same launches back to back (no events): launch-gap-inclusive
    hipEvent_t e0 = t.a[0], e1 = t.b[0];
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) launch(i, nullptr, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float wall = 0; hipEventElapsedTime(&wall, e0, e1);
    printf("%-44s grid=(%4u,%2u) block=%4u  kernel mean %6.2f us  min %6.2f  med %6.2f | back-to-back %6.2f us/launch\n", name, grid.x, grid.y,
           block.x, mean, us.front(), us[us.size() / 2], wall * 1e3 / iters);
}

int main() {
    uint32_t* out; hipMalloc(&out, 4);
    // ---- launch floor ---------------------------------------------------------------------------------------
    const int gs[] = {64, 256, 256, 256, 512, 1024, 2048, 4096};
    const int bs[] = {256, 256, 512, 1024, 512, 256, 256, 64};
    for (int i = 0; i < 8; ++i) {
        dim3 g(gs[i]), b(bs[i]);
        run("empty", g, b, [&](int, hipEvent_t s, hipEvent_t e) { hipExtLaunchKernelGGL(k_empty, g, b, 0, 0, s, e, 0, out); });
    }
    // ---- strip reads: 4096 x 4096 4-bit (512 packed rows x 4096 words) and 16384^2 (2048 x 16384) ---------------
    for (int big = 0; big < 2; ++big) {
        const int rows = big ? 2048 : 512, cols = big ? 16384 : 4096, pitch = cols;
        const size_t words = (size_t)rows * pitch;
        int nbuf = 1; while ((size_t)nbuf * words * 4 < (700u << 20)) nbuf *= 2;
        uint32_t* buf; hipMalloc(&buf, words * 4 * nbuf); hipMemset(buf, 0x5a, words * 4 * nbuf);
        printf("---- %d x %d words (%.1f MB), %d rotating buffers\n", rows, cols, words * 4 / 1e6, nbuf);
#define RUN(CQ, R, NW, NT, PAIR, SK)                                                                                   \
    {                                                                                                                  \
        dim3 g(cols / (4 << CQ), SK), b(NW * 64);                                                                      \
        char nm[96]; snprintf(nm, 96, "strip tc=%d R=%d waves=%d %s%s sk=%d", 4 << CQ, R, NW, NT ? "nt " : "", PAIR ? "pair" : "", SK); \
        run(nm, g, b, [&](int it, hipEvent_t s, hipEvent_t e) {                                                        \
            hipExtLaunchKernelGGL((k_strip<CQ, R, NW, NT, PAIR>), g, b, 0, 0, s, e, 0, buf + (size_t)(it % nbuf) * words, rows, pitch, out); \
        });                                                                                                            \
    }
        if (!big) {
            RUN(2, 2, 16, false, true, 1) RUN(2, 2, 16, true, true, 1) RUN(2, 4, 8, false, true, 1) RUN(2, 4, 8, true, true, 1)
            RUN(2, 8, 4, false, true, 1) RUN(2, 8, 4, true, true, 1) RUN(2, 2, 16, true, false, 1)
            RUN(3, 2, 16, true, false, 2) RUN(3, 4, 8, true, false, 2) RUN(3, 4, 4, true, false, 4) RUN(3, 8, 4, false, false, 2)
            RUN(4, 4, 4, true, false, 8) RUN(4, 4, 4, false, false, 8) RUN(4, 2, 8, true, false, 8) RUN(4, 4, 8, true, false, 4)
            RUN(4, 8, 4, true, false, 4)
        } else {
            RUN(4, 4, 4, false, false, 1) RUN(4, 4, 4, true, false, 1) RUN(4, 8, 4, true, false, 1) RUN(4, 4, 8, true, false, 1)
            RUN(4, 8, 8, true, false, 1) RUN(4, 4, 4, true, false, 2) RUN(3, 4, 4, true, false, 1) RUN(3, 8, 4, true, false, 1)
            RUN(4, 16, 4, true, false, 1) RUN(4, 4, 16, true, false, 1)
        }
        hipFree(buf);
    }
    return 0;
}
