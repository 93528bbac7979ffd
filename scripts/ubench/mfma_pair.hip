// Development microbenchmark: two waves per SIMD (512-thread block) running the SAME mixed stream — one
// v_mfma_f32_32x32x16_bf16, one ds_read_b128 and NV plain VALU per slot — against one wave per SIMD (256 threads).
// Question: does the VALU of both waves hide under the shared matrix pipe, or does it add?  Prints shader cycles per
// MFMA *per SIMD* (32 = the pipe is full).   DEP: the VALU produce the B operand of the MFMA MI slots later.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define SB() __builtin_amdgcn_sched_barrier(0)
#define OPAQUE(x) asm volatile("" : "+v"(x))

template <int NV, bool DSR, bool DEP, int PRIO>
__global__ __launch_bounds__(512, 1) void k(unsigned long long* out, const uint32_t* src, int iters) {
    __shared__ u32x4 lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    __syncthreads();
    if (PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);
    constexpr int MI = 4;
    f32x16 acc[MI];
    for (int i = 0; i < MI; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 af[2][MI];
    for (int i = 0; i < MI; ++i) af[0][i] = af[1][i] = lds[lane + 64 * i];
    u32x4 bf[2] = {lds[lane], lds[lane + 64]};
    uint32_t w = src[tid];
    float A = 1.5f, Bq = 0.25f;
    OPAQUE(A); OPAQUE(Bq);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const int ks = m / MI, mi = m % MI;
            acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, af[ks & 1][mi]),
                                                               __builtin_bit_cast(b8_t, bf[ks & 1]), acc[mi], 0, 0, 0);
            if (DSR) af[(ks + 1) & 1][mi] = lds[lane + 64 * m];
            // NV VALU: pattern of the bf16 dequant (cvt_ubyte, fma, cvt_pk), 6 per "pair + extract"
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int r = q % 6;
                if (r == 0) { OPAQUE(w); v[0] = (float)((w >> (8 * (mi & 3))) & 0xFFu); }
                else if (r == 1) { v[1] = (float)(((w >> 4) >> (8 * (mi & 3))) & 0xFFu); }
                else if (r == 2) v[2] = __builtin_fmaf(v[0], A, Bq);
                else if (r == 3) v[3] = __builtin_fmaf(v[1], A, Bq);
                else if (r == 4) {
                    const b2_t p = {(__bf16)v[2], (__bf16)v[3]};
                    uint32_t u = __builtin_bit_cast(uint32_t, p);
                    if (DEP) bf[(ks + 1) & 1][mi] = u; else asm volatile("" ::"v"(u));
                } else { uint32_t x = w & 0x0F0F0F0Fu; asm volatile("" ::"v"(x)); }
            }
            SB();
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = v[2] + v[3];
    for (int i = 0; i < MI; ++i) s += acc[i][0] + __builtin_bit_cast(float, af[0][i][0]) + __builtin_bit_cast(float, af[1][i][1]);
    s += __builtin_bit_cast(float, bf[0][0]) + __builtin_bit_cast(float, bf[1][1]);
    if (lane == 0) {
        out[(blockIdx.x * 8 + wave) * 2] = c1 - c0;
        out[(blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)s;
    }
}

template <int NV, bool DSR, bool DEP, int PRIO>
static void run(int threads) {
    const int blocks = 256, iters = 200, waves = threads / 64;
    unsigned long long* d; uint32_t* src;
    hipMalloc(&d, blocks * 16 * sizeof(unsigned long long)); hipMalloc(&src, 4096);
    hipMemset(src, 0x5A, 4096);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NV, DSR, DEP, PRIO>), dim3(blocks), dim3(threads), 0, 0, d, src, iters);
    hipDeviceSynchronize();
    unsigned long long* h = (unsigned long long*)malloc(blocks * 16 * sizeof(unsigned long long));
    hipMemcpy(h, d, blocks * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (int b = 0; b < blocks; ++b)
        for (int wv = 0; wv < waves; ++wv) { const double c = (double)h[2 * (b * 8 + wv)] / (iters * 32.0); mean += c; if (c > mx) mx = c; }
    mean /= blocks * waves;
    const double per_simd = mean / (waves / 4);  // waves/4 waves share one SIMD's pipe
    printf("waves/SIMD %d  VALU/slot %2d  ds_read %d  dep %d  prio %d : %6.1f cycles per MFMA per SIMD (wave sees %6.1f, max %6.1f)\n",
           waves / 4, NV, (int)DSR, (int)DEP, PRIO, per_simd, mean, mx);
    hipFree(d); hipFree(src); free(h);
}


// Role split: waves 0..3 (one per SIMD) issue only MFMA (+ ds_read), waves 4..7 only VALU (NV per "slot", no MFMA):
// do a partner's VALU run at full rate beside a busy matrix pipe?
template <int NV, bool DSR, bool ILP = false>
__global__ __launch_bounds__(512, 1) void k_roles(unsigned long long* out, const uint32_t* src, int iters) {
    __shared__ u32x4 lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    __syncthreads();
    constexpr int MI = 4;
    f32x16 acc[MI];
    for (int i = 0; i < MI; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 af[2][MI];
    for (int i = 0; i < MI; ++i) af[0][i] = af[1][i] = lds[lane + 64 * i];
    u32x4 bf[2] = {lds[lane], lds[lane + 64]};
    uint32_t w = src[tid];
    float A = 1.5f, Bq = 0.25f;
    OPAQUE(A); OPAQUE(Bq);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t sink = 0;
    const unsigned long long c0 = __builtin_readcyclecounter();
    if (wave < 4) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                const int ks = m / MI, mi = m % MI;
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, af[ks & 1][mi]),
                                                                   __builtin_bit_cast(b8_t, bf[ks & 1]), acc[mi], 0, 0, 0);
                if (DSR) af[(ks + 1) & 1][mi] = lds[lane + 64 * m];
                SB();
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 32; ++m) {
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    if (ILP) { v[q & 7] = __builtin_fmaf(v[q & 7], A, Bq); continue; }  // 8 independent chains
                    const int r = q % 5;
                    if (r == 0) { OPAQUE(w); v[0] = (float)((w >> (8 * (m & 3))) & 0xFFu); }
                    else if (r == 1) { v[1] = (float)(((w >> 4) >> (8 * (m & 3))) & 0xFFu); }
                    else if (r == 2) v[2] = __builtin_fmaf(v[0], A, Bq);
                    else if (r == 3) v[3] = __builtin_fmaf(v[1], A, Bq);
                    else { const b2_t p = {(__bf16)v[2], (__bf16)v[3]}; uint32_t u = __builtin_bit_cast(uint32_t, p); asm volatile("" ::"v"(u)); }
                }
                SB();
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] + (float)sink;
    for (int i = 0; i < MI; ++i) s += acc[i][0] + __builtin_bit_cast(float, af[0][i][0]) + __builtin_bit_cast(float, af[1][i][1]);
    if (lane == 0) {
        out[(blockIdx.x * 8 + wave) * 2] = c1 - c0;
        out[(blockIdx.x * 8 + wave) * 2 + 1] = (unsigned long long)s;
    }
}

template <int NV, bool DSR, bool ILP = false>
static void run_roles() {
    const int blocks = 256, iters = 200;
    unsigned long long* d; uint32_t* src;
    hipMalloc(&d, blocks * 16 * sizeof(unsigned long long)); hipMalloc(&src, 4096);
    hipMemset(src, 0x5A, 4096);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k_roles<NV, DSR, ILP>), dim3(blocks), dim3(512), 0, 0, d, src, iters);
    hipDeviceSynchronize();
    unsigned long long* h = (unsigned long long*)malloc(blocks * 16 * sizeof(unsigned long long));
    hipMemcpy(h, d, blocks * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mm = 0, vv = 0;
    for (int b = 0; b < blocks; ++b)
        for (int wv = 0; wv < 8; ++wv) { const double c = (double)h[2 * (b * 8 + wv)] / (iters * 32.0); if (wv < 4) mm += c; else vv += c; }
    mm /= blocks * 4; vv /= blocks * 4;
    printf("roles: MFMA wave %6.1f cycles per MFMA | VALU wave %6.1f cycles per %2d VALU = %5.2f per VALU  (ds_read in MFMA wave %d, independent VALU %d)\n",
           mm, vv, NV, NV ? vv / NV : 0.0, (int)DSR, (int)ILP);
    hipFree(d); hipFree(src); free(h);
}

template <int NV>
static void sweep() {
    run<NV, true, false, 0>(256);
    run<NV, true, false, 0>(512);
    run<NV, true, true, 0>(512);
    run<NV, true, true, 1>(512);
    run<NV, false, false, 0>(512);
}

int main() {
    run_roles<5, true>(); run_roles<10, true>(); run_roles<15, true>(); run_roles<20, true>(); run_roles<30, true>(); run_roles<10, false>(); run_roles<8, true, true>(); run_roles<16, true, true>(); run_roles<32, true, true>(); run_roles<64, true, true>();
    sweep<0>(); sweep<2>(); sweep<4>(); sweep<6>(); sweep<8>(); sweep<12>(); sweep<16>();
    return 0;
}
