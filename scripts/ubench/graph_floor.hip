// Micro-benchmark: per-launch cost INSIDE A REPLAYED hipGraph (the way bench.py times the decode step) of an empty kernel and
// of pure strip reads of the GEMV's access pattern.  launch_floor.hip measured the same kernels from an eager host loop, which
// is host-bound at ~2.5 us per launch (MI355X_MICROARCH.md, rows "boundary" / "graph-replay-floor") — VERDICT r2 asked for the
// device-side floor.  Run it plain (graph wall time / launches) and under `rocprofv3 --kernel-trace --stats` (device duration per
// kernel name: every variant is its own template instantiation).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <functional>
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
    const int nch = rows_slice / CHUNK;
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

static void graph_run(const char* name, int nlaunch, const std::function<void(int, hipStream_t)>& launch) {
    hipStream_t st; hipStreamCreate(&st);
    for (int i = 0; i < 4; ++i) launch(i, st);
    hipStreamSynchronize(st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < nlaunch; ++i) launch(i, st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, sum = 0;
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, st);
        hipGraphLaunch(ge, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    printf("%-46s graph of %3d launches: %7.3f us/launch (mean) %7.3f (best replay)\n", name, nlaunch, sum / reps * 1e3 / nlaunch, best * 1e3 / nlaunch);
    hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(st);
}

int main() {
    uint32_t* out; hipMalloc(&out, 4);
    const int NL = 64;
    const int gs[] = {256, 256, 1024}, bs[] = {256, 1024, 256};
    for (int i = 0; i < 3; ++i) {
        dim3 g(gs[i]), b(bs[i]);
        char nm[64]; snprintf(nm, 64, "empty grid=%d block=%d", gs[i], bs[i]);
        graph_run(nm, NL, [&](int, hipStream_t s) { hipLaunchKernelGGL(k_empty, g, b, 0, s, out); });
    }
    for (int big = 0; big < 3; ++big) {
        const int rows = big == 2 ? 2048 : (big ? 1024 : 512), cols = big == 2 ? 16384 : (big ? 8192 : 4096), pitch = cols;
        const size_t words = (size_t)rows * pitch;
        int nbuf = 1; while ((size_t)nbuf * words * 4 < (600u << 20)) nbuf *= 2;
        uint32_t* buf; hipMalloc(&buf, words * 4 * nbuf); hipMemset(buf, 0x5a, words * 4 * nbuf);
        printf("---- %d x %d words (%.1f MB), %d rotating buffers (HBM-cold)\n", rows, cols, words * 4 / 1e6, nbuf);
#define RUN(CQ, R, NW, NT, PAIR, SK)                                                                                   \
    {                                                                                                                  \
        dim3 g(cols / (4 << CQ), SK), b(NW * 64);                                                                      \
        char nm[96]; snprintf(nm, 96, "strip tc=%d R=%d waves=%d %s%s sk=%d", 4 << CQ, R, NW, NT ? "nt " : "", PAIR ? "pair" : "", SK); \
        graph_run(nm, NL, [&](int it, hipStream_t s) {                                                                 \
            hipLaunchKernelGGL((k_strip<CQ, R, NW, NT, PAIR>), g, b, 0, s, buf + (size_t)(it % nbuf) * words, rows, pitch, out); \
        });                                                                                                            \
    }
        if (big == 0) {
            RUN(2, 2, 16, false, true, 1) RUN(2, 2, 16, true, true, 1) RUN(2, 4, 8, true, true, 1) RUN(2, 8, 4, true, true, 1)
            RUN(3, 2, 16, true, false, 1) RUN(3, 4, 8, true, false, 1) RUN(3, 2, 16, true, false, 2) RUN(3, 4, 8, true, false, 2)
            RUN(4, 4, 4, true, false, 4) RUN(4, 2, 8, true, false, 4) RUN(4, 4, 8, true, false, 1) RUN(4, 2, 16, true, false, 1)
        } else if (big == 1) {
            RUN(3, 4, 4, false, false, 1) RUN(3, 4, 4, true, false, 1) RUN(3, 8, 4, true, false, 1) RUN(3, 4, 8, true, false, 1) RUN(3, 2, 16, true, false, 1)
            RUN(4, 4, 4, true, false, 2) RUN(4, 8, 4, true, false, 2) RUN(4, 4, 8, true, false, 2) RUN(4, 4, 8, true, false, 1)
        } else {
            RUN(4, 4, 4, false, false, 1) RUN(4, 4, 4, true, false, 1) RUN(4, 8, 4, true, false, 1) RUN(4, 4, 8, true, false, 1) RUN(4, 8, 8, true, false, 1)
        }
        hipFree(buf);
    }
    return 0;
}
