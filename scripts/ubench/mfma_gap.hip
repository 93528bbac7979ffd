// Development microbenchmark: how many / which "filler" instructions hide behind a v_mfma_f32_32x32x16_bf16 when a
// SIMD runs ONE wave (the 512-register tiled GEMM).  Each variant issues 32 MFMAs per loop trip (8 accumulators x 4)
// and a fixed filler pattern per gap; prints shader cycles per MFMA.   hipcc --offload-arch=gfx950 -O3 mfma_gap.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define SB() __builtin_amdgcn_sched_barrier(0)
#define USE(x) asm volatile("" ::"v"(x))
#define OPAQUE(x) asm volatile("" : "+v"(x))

template <int V>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, const uint32_t* src, int iters) {
    __shared__ u32x4 lds[4096];  // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    __syncthreads();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 af[2][8];
    for (int i = 0; i < 8; ++i) af[0][i] = af[1][i] = lds[lane + 64 * i];
    u32x4 bfr = lds[lane];
    uint32_t w = src[tid];
    uint32_t t0 = w & 0x0F0F0F0Fu, t1 = (w >> 4) & 0x0F0F0F0Fu;
    float A = 1.5f, Bq = 0.25f, lo = 0.f, hi = 0.f;
    OPAQUE(A); OPAQUE(Bq);
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            const int ks = m >> 3, mi = m & 7;
            acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, af[ks & 1][mi]), __builtin_bit_cast(b8_t, bfr), acc[mi], 0, 0, 0);
            constexpr bool DSR = (V == 1 || V == 3 || V == 4 || V == 5 || V == 6 || V == 7 || V == 8 || V == 10);
            if (DSR) af[(ks + 1) & 1][mi] = lds[lane + 64 * m];  // base VGPR + immediate offset: no address VALU
            auto cvt = [&](int b) {
                OPAQUE(t0); OPAQUE(t1);
                lo = (float)((t0 >> (8 * b)) & 0xFFu);
                hi = (float)((t1 >> (8 * b)) & 0xFFu);
            };
            auto fin = [&]() {
                const b2_t v = {(__bf16)__builtin_fmaf(lo, A, Bq), (__bf16)__builtin_fmaf(hi, A, Bq)};
                USE(__builtin_bit_cast(uint32_t, v));
            };
            if (V == 2 || V == 3) { cvt(m & 3); fin(); }                       // full pair (5 VALU) every gap
            if (V == 4) { if (m & 1) { cvt(m & 3); fin(); } }                  // pair every 2nd gap: 1 / 6 fillers
            if (V == 5 || V == 7) {                                            // balanced: 4 / 4 fillers
                if (m & 1) fin();
                else { cvt(m & 3); uint32_t x = w & (0x0F0F0F0Fu << (m & 3)); USE(x); }
            }
            if (V == 6) { uint32_t x = w & 0xFFu, y = w >> (m & 7), z = w ^ (uint32_t)m; OPAQUE(x); OPAQUE(y); OPAQUE(z); USE(x + y + z); }  // ~5 int VALU
            if (V == 7 && (m & 3) == 3) lds[tid + 256 * (m >> 2) + 2048] = af[0][0];  // + ds_write_b128 every 4th gap
            if (V == 8) {                                                      // pair through v_pk_fma_f32
                cvt(m & 3);
                f32x2 p = {lo, hi};
                p = __builtin_elementwise_fma(p, (f32x2){A, A}, (f32x2){Bq, Bq});
                const b2_t v = {(__bf16)p[0], (__bf16)p[1]};
                USE(__builtin_bit_cast(uint32_t, v));
            }
            if (V == 9) { lo = __builtin_fmaf(lo, A, Bq); hi = __builtin_fmaf(hi, A, Bq); }  // 2 fma per gap
            if (V == 10) { if (m & 1) { lo = __builtin_fmaf(lo, A, Bq); hi = __builtin_fmaf(hi, A, Bq); } }  // dsR + 2 fma every 2nd gap
            SB();
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = lo + hi;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + __builtin_bit_cast(float, af[0][i][0]) + __builtin_bit_cast(float, af[1][i][1]);
    if (lane == 0) {
        out[(blockIdx.x * 4 + (tid >> 6)) * 2] = c1 - c0;
        out[(blockIdx.x * 4 + (tid >> 6)) * 2 + 1] = (unsigned long long)s;
    }
}

template <int V>
static void run(const char* what, int blocks) {
    unsigned long long* d; uint32_t* src;
    hipMalloc(&d, blocks * 8 * sizeof(unsigned long long)); hipMalloc(&src, 1024);
    hipMemset(src, 0x5A, 1024);
    const int iters = 200;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, src, iters);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, src, iters);
    hipDeviceSynchronize();
    unsigned long long* h = (unsigned long long*)malloc(blocks * 8 * sizeof(unsigned long long));
    hipMemcpy(h, d, blocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (int i = 0; i < blocks * 4; ++i) { const double c = (double)h[2 * i] / (iters * 32.0); mean += c; if (c > mx) mx = c; }
    printf("variant %2d blocks %3d: %6.1f cycles/MFMA (max wave %6.1f)  %s\n", V, blocks, mean / (blocks * 4), mx, what);
    hipFree(d); hipFree(src); free(h);
}

int main() {
    for (int blocks : {1, 256}) {
        run<0>("MFMA only", blocks);
        run<1>("+ ds_read_b128 every gap", blocks);
        run<9>("+ 2 v_fma_f32 every gap", blocks);
        run<2>("+ bf16 dequant pair (2 cvt_ubyte, 2 fma, cvt_pk) every gap", blocks);
        run<6>("+ ds_read + ~5 integer VALU every gap", blocks);
        run<3>("+ ds_read + pair every gap (6 fillers)", blocks);
        run<4>("+ ds_read every gap, pair every 2nd gap (1 / 6)", blocks);
        run<5>("+ ds_read every gap, pair split over two gaps (4 / 4)", blocks);
        run<7>("  same + ds_write_b128 every 4th gap", blocks);
        run<8>("+ ds_read + pair via v_pk_fma_f32 every gap", blocks);
        run<10>("+ ds_read every gap, 2 fma every 2nd gap", blocks);
    }
    return 0;
}
