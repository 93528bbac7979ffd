// ldsfill.hip — how fast can ONE CU fill LDS from L2 / the Infinity Cache with LDS-DMA (buffer_load_dwordx4 ... lds, 1-KiB pieces)?
// The K-unsplit tile kernels of round 4 (gemm_a8w8_sq_kernel, the 64-column W4 tiles) all run at ~60 GB/s per CU of operand fill
// (cfgA: 640 KB per block in 10.7 us; A8W8 4096^2: 512 KB in ~9 us; FP8 16384^2: 2 MB per round in 35 us) whatever they compute —
// is that the hardware's rate for this access pattern or the kernels' pipelining?  256 blocks x 8 waves; every wave streams rows of a
// source matrix as 1-KiB pieces (4 rows x 256 B, like the kernels' x tiles) into a ring of D LDS slots with D - 1 pieces in flight,
// nothing is computed.  Patterns: 0 = every block reads the SAME 64 rows x K (an x tile: L2-resident after the first block of an XCD),
// 1 = every block its own 64 rows (a weight tile: each byte once), 2 = both (x tile + own tile alternating, the kernels' mix).
// Also the same traffic as plain 16-byte global loads into registers (mode r) for comparison.
//   usage: ldsfill [K bytes per row = 8192] [reps = 20]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t srd_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ srd_t make_srd(const void* base, uint32_t bytes) {
    const uint64_t b = (uint64_t)base;
    srd_t r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xFFFFu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void req_lds16(srd_t rs, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// PAT: 0 shared rows, 1 own rows, 2 alternate.  D pieces in flight per wave.  rowbytes = K.  Each wave owns 8 of the block's 64 rows
// (two pieces of 4 rows per 256-byte column step).
template <int PAT, int D>
__global__ __launch_bounds__(512, 1) void k_fill(const unsigned char* shared_rows, const unsigned char* own_rows, int rowbytes, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // PAT 3 = the unsplit 64 x 64 tile kernels' real sharing: 4 row tiles of x (each read by 64 blocks) and 64 weight column tiles (each
    // read by the 4 blocks of one XCD, back to back in its dispatch order: block b runs on XCD b % 8), pieces alternating like PAT 2
    int st = 0, ot = blockIdx.x;
    if (PAT == 3) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        st = idx % 4;
        ot = (idx / 4) * 8 + xcd;
    }
    const srd_t rsS = make_srd(shared_rows + (size_t)st * 64 * rowbytes, (uint32_t)(64 * rowbytes));
    const srd_t rsO = make_srd(own_rows + (size_t)ot * 64 * rowbytes, (uint32_t)(64 * rowbytes));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)smem + (uint32_t)(wave * D) * 1024u;
    // piece p of this wave: rows wave * 8 + 4 (p & 1) + lane / 16, 16-byte slot lane % 16 of the 256-byte column step p >> 1
    const uint32_t voff0 = (uint32_t)((wave * 8 + (lane >> 4)) * rowbytes + (lane & 15) * 16);
    const int npieces = 2 * (rowbytes / 256);
    int issued = 0;
    for (int p = 0; p < npieces; ++p) {
        const uint32_t voff = voff0 + (uint32_t)((p & 1) * 4 * rowbytes);
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((p >> 1) * 256);
        const bool own = PAT == 1 || (PAT >= 2 && (p & 2));
        req_lds16(own ? rsO : rsS, lds0 + (uint32_t)(p % D) * 1024u, voff, soff);
        if (++issued >= D) wait_vm<D - 1>();
    }
    wait_vm<0>();
    __syncthreads();
    if (threadIdx.x == 0 && smem[17] == 0x5A && smem[1025] == 0x77) sink[0] = 1;
}
template <int PAT, int D>
__global__ __launch_bounds__(512, 1) void k_regs(const unsigned char* shared_rows, const unsigned char* own_rows, int rowbytes, uint32_t* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* own = own_rows + (size_t)blockIdx.x * 64 * rowbytes;
    const size_t voff0 = (size_t)(wave * 8 + (lane >> 4)) * rowbytes + (lane & 15) * 16;
    const int npieces = 2 * (rowbytes / 256);
    u32x4 acc = {0, 0, 0, 0};
    for (int p0 = 0; p0 < npieces; p0 += D) {
        u32x4 v[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int p = p0 + j;
            const bool o = PAT == 1 || (PAT == 2 && (p & 2));
            v[j] = *(const u32x4*)((o ? own : shared_rows) + voff0 + (size_t)(p & 1) * 4 * rowbytes + (size_t)(p >> 1) * 256);
        }
#pragma unroll
        for (int j = 0; j < D; ++j) acc ^= v[j];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <typename F>
static float time_us(F launch, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int rowbytes = argc > 1 ? atoi(argv[1]) : 8192, reps = argc > 2 ? atoi(argv[2]) : 20;
    const int blocks = 256;
    unsigned char *sh, *own; uint32_t* sink;
    CHECK(hipMalloc(&sh, (size_t)256 * rowbytes));
    CHECK(hipMalloc(&own, (size_t)blocks * 64 * rowbytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(sh, 1, (size_t)256 * rowbytes)); CHECK(hipMemset(own, 2, (size_t)blocks * 64 * rowbytes));
    const double bytes_per_block = 64.0 * rowbytes;
    printf("256 blocks x 8 waves, %d bytes per row, 64 rows per block = %.0f KB per block\n", rowbytes, bytes_per_block / 1024);
#define RUN(KERN, PAT, D, LDS, NAME) { \
        if (LDS > 65536) CHECK(hipFuncSetAttribute((const void*)KERN<PAT, D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); \
        float us = time_us([&] { KERN<PAT, D><<<blocks, 512, LDS>>>(sh, own, rowbytes, sink); }, reps); \
        printf("%-10s pattern %d  %2d pieces in flight per wave (%3d KB per CU): %7.2f us per launch  ->  %6.1f GB/s per CU, %5.2f TB/s chip (launch overhead included)\n", \
               NAME, PAT, D, D * 8, us, bytes_per_block / us / 1e3, bytes_per_block * blocks / us / 1e6); }
    RUN(k_fill, 0, 2, 16384, "lds-dma") RUN(k_fill, 0, 4, 32768, "lds-dma") RUN(k_fill, 0, 8, 65536, "lds-dma") RUN(k_fill, 0, 16, 131072, "lds-dma")
    RUN(k_fill, 1, 2, 16384, "lds-dma") RUN(k_fill, 1, 4, 32768, "lds-dma") RUN(k_fill, 1, 8, 65536, "lds-dma") RUN(k_fill, 1, 16, 131072, "lds-dma")
    RUN(k_fill, 2, 4, 32768, "lds-dma") RUN(k_fill, 2, 8, 65536, "lds-dma") RUN(k_fill, 2, 16, 131072, "lds-dma")
    RUN(k_fill, 3, 4, 32768, "lds-dma") RUN(k_fill, 3, 8, 65536, "lds-dma") RUN(k_fill, 3, 12, 98304, "lds-dma") RUN(k_fill, 3, 16, 131072, "lds-dma")
    RUN(k_regs, 0, 4, 0, "registers") RUN(k_regs, 0, 8, 0, "registers") RUN(k_regs, 0, 16, 0, "registers")
    RUN(k_regs, 1, 4, 0, "registers") RUN(k_regs, 1, 8, 0, "registers") RUN(k_regs, 1, 16, 0, "registers")
    return 0;
}
