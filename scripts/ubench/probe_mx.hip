// probe_mx.hip — operand layout / scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 and of the scaled converters
// (v_cvt_scalef32_pk_{bf16,f16}_{fp8,fp4}) on gfx950, checked against a host model.  One wave per launch.
//
// Hypothesis H (what gemm_mx.hip assumes):
//   A: lane l holds row (l & 31), its 32 elements e = 0..31 are k = 32 * (l >> 5) + e; fp8: element e = byte e of the
//      lane's 8 dwords; fp4: element e = nibble e (low nibble first) of the lane's first 4 dwords.  B: the same with
//      column (l & 31).  Scale: byte 0 of the lane's scale register, e8m0, applies to the lane's own 32 elements.
//   D: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int FA, int FB, int OA, int OB>  // formats (0 fp8 e4m3, 4 fp4 e2m1), opsel of the two scales
__global__ void mx_kernel(const v8i* a, const v8i* b, const int* sa, const int* sb, v16f* d) {
    const int l = threadIdx.x;
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, FA, FB, OA, sa[l], OB, sb[l]);
    d[l] = acc;
}

__global__ void cvt_kernel(const uint32_t* src, const float* scale, uint32_t* out) {
    const int l = threadIdx.x;
    const uint32_t s = src[l];
    const float sc = scale[l];
    out[l * 12 + 0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(s, sc, false));
    out[l * 12 + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(s, sc, true));
    out[l * 12 + 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(s, sc, false));
    out[l * 12 + 3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(s, sc, true));
    out[l * 12 + 4] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(s, sc, 0));
    out[l * 12 + 5] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(s, sc, 1));
    out[l * 12 + 6] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(s, sc, 2));
    out[l * 12 + 7] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(s, sc, 3));
    out[l * 12 + 8] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(s, sc, 0));
    out[l * 12 + 9] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(s, sc, 1));
    out[l * 12 + 10] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(s, sc, 2));
    out[l * 12 + 11] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(s, sc, 3));
}

static float fp8_to_f(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) f = m * 0.001953125f;
    else f = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
static float fp4_to_f(uint8_t v) {
    static const float t[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    return (v & 8) ? -t[v & 7] : t[v & 7];
}
static float bf16_to_f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static float f16_to_f(uint16_t v) {
    const int s = v >> 15, e = (v >> 10) & 31, m = v & 1023;
    float f = e == 0 ? ldexpf((float)m, -24) : (e == 31 ? INFINITY : ldexpf(1.0f + m / 1024.0f, e - 15));
    return s ? -f : f;
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

// element e of lane l under hypothesis H
static float elem(const uint32_t* regs, int l, int e, int fmt) {
    if (fmt == 0) return fp8_to_f((regs[l * 8 + e / 4] >> (8 * (e & 3))) & 0xFF);
    return fp4_to_f((regs[l * 8 + e / 8] >> (4 * (e & 7))) & 0xF);
}

template <int FA, int FB, int OA, int OB>
static double run_case(const char* name, int scale_byte_a, int scale_byte_b, bool dump) {
    uint32_t ha[64 * 8], hb[64 * 8];
    int hsa[64], hsb[64];
    for (int i = 0; i < 64 * 8; ++i) {
        // fp8 bytes: keep exponents moderate and avoid NaN (0x7F / 0xFF): e in 4..10, random mantissa / sign
        auto fp8b = []() -> uint32_t { return ((rnd() & 1) << 7) | ((4 + rnd() % 7) << 3) | (rnd() & 7); };
        ha[i] = FA == 0 ? (fp8b() | fp8b() << 8 | fp8b() << 16 | fp8b() << 24) : (rnd() * 2654435761u);
        hb[i] = FB == 0 ? (fp8b() | fp8b() << 8 | fp8b() << 16 | fp8b() << 24) : (rnd() * 2246822519u);
    }
    for (int l = 0; l < 64; ++l) {
        // the wanted scale sits in byte `scale_byte`, every other byte holds a very different exponent
        const uint32_t wa = 120 + rnd() % 14, wb = 120 + rnd() % 14;
        uint32_t ra = 0x60606060u, rb = 0x90909090u;
        ra = (ra & ~(0xFFu << (8 * scale_byte_a))) | (wa << (8 * scale_byte_a));
        rb = (rb & ~(0xFFu << (8 * scale_byte_b))) | (wb << (8 * scale_byte_b));
        hsa[l] = (int)ra;
        hsb[l] = (int)rb;
    }
    void *da, *db, *dsa, *dsb, *dd;
    CHECK(hipMalloc(&da, sizeof(ha))); CHECK(hipMalloc(&db, sizeof(hb)));
    CHECK(hipMalloc(&dsa, sizeof(hsa))); CHECK(hipMalloc(&dsb, sizeof(hsb))); CHECK(hipMalloc(&dd, 64 * 16 * 4));
    CHECK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dsa, hsa, sizeof(hsa), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dsb, hsb, sizeof(hsb), hipMemcpyHostToDevice));
    mx_kernel<FA, FB, OA, OB><<<1, 64>>>((const v8i*)da, (const v8i*)db, (const int*)dsa, (const int*)dsb, (v16f*)dd);
    CHECK(hipDeviceSynchronize());
    float hd[64 * 16];
    CHECK(hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost));
    double worst = 0, ref_max = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            double ref = 0;
            for (int h = 0; h < 2; ++h) {
                const int la = row + 32 * h, lb = col + 32 * h;
                const double s = ldexp(1.0, (int)(((uint32_t)hsa[la] >> (8 * scale_byte_a)) & 0xFF) - 127) *
                                 ldexp(1.0, (int)(((uint32_t)hsb[lb] >> (8 * scale_byte_b)) & 0xFF) - 127);
                double part = 0;
                for (int e = 0; e < 32; ++e) part += (double)elem(ha, la, e, FA) * (double)elem(hb, lb, e, FB);
                ref += part * s;
            }
            const double err = fabs(ref - (double)hd[l * 16 + r]);
            if (err > worst) worst = err;
            if (fabs(ref) > ref_max) ref_max = fabs(ref);
            if (dump && l < 2 && r < 4) printf("    lane %d reg %d: got %.6g want %.6g\n", l, r, hd[l * 16 + r], ref);
        }
    printf("%-44s max|err| %.3e (max|ref| %.3e) -> %s\n", name, worst, ref_max, worst <= 2e-5 * ref_max ? "MATCH" : "MISMATCH");
    hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dd);
    return worst / (ref_max + 1e-30);
}

int main() {
    printf("== v_mfma_scale_f32_32x32x64_f8f6f4 vs hypothesis H ==\n");
    run_case<0, 0, 0, 0>("fp8 x fp8, scales in byte 0, opsel 0", 0, 0, true);
    run_case<0, 4, 0, 0>("fp8 x fp4, scales in byte 0, opsel 0", 0, 0, false);
    run_case<4, 4, 0, 0>("fp4 x fp4, scales in byte 0, opsel 0", 0, 0, false);
    run_case<4, 0, 0, 0>("fp4 x fp8, scales in byte 0, opsel 0", 0, 0, false);
    // does opsel pick another byte of the scale register?
    run_case<0, 0, 1, 1>("fp8 x fp8, scales in byte 1, opsel 1", 1, 1, false);
    run_case<0, 0, 2, 2>("fp8 x fp8, scales in byte 2, opsel 2", 2, 2, false);
    run_case<0, 0, 3, 3>("fp8 x fp8, scales in byte 3, opsel 3", 3, 3, false);
    run_case<0, 0, 1, 1>("fp8 x fp8, scales in byte 0, opsel 1 (ignored?)", 0, 0, false);
    run_case<4, 4, 2, 1>("fp4 x fp4, scales in bytes 2 / 1, opsel 2 / 1", 2, 1, false);

    printf("== scaled converters ==\n");
    uint32_t hs[64];
    float hsc[64];
    for (int l = 0; l < 64; ++l) {
        hs[l] = l == 0 ? 0x76543210u : (l == 1 ? 0xFEDCBA98u : (l == 2 ? 0x48403830u : (rnd() * 2654435761u)));
        hsc[l] = l < 3 ? 1.0f : ldexpf(1.0f, (int)(rnd() % 9) - 4);
    }
    void *ds, *dsc, *dout;
    CHECK(hipMalloc(&ds, sizeof(hs))); CHECK(hipMalloc(&dsc, sizeof(hsc))); CHECK(hipMalloc(&dout, 64 * 12 * 4));
    CHECK(hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dsc, hsc, sizeof(hsc), hipMemcpyHostToDevice));
    cvt_kernel<<<1, 64>>>((const uint32_t*)ds, (const float*)dsc, (uint32_t*)dout);
    CHECK(hipDeviceSynchronize());
    uint32_t ho[64 * 12];
    CHECK(hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost));
    // hypothesis: fp8 pair = bytes (2 sel, 2 sel + 1) -> (lo16, hi16); fp4 pair = byte `sel`: (low nibble -> lo16, high nibble -> hi16)
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        for (int sel = 0; sel < 2; ++sel) {
            const float w0 = fp8_to_f((hs[l] >> (16 * sel)) & 0xFF) * hsc[l], w1 = fp8_to_f((hs[l] >> (16 * sel + 8)) & 0xFF) * hsc[l];
            const uint32_t gb = ho[l * 12 + sel], gh = ho[l * 12 + 2 + sel];
            const bool okb = (isnan(w0) || bf16_to_f(gb & 0xFFFF) == w0) && (isnan(w1) || bf16_to_f(gb >> 16) == w1);
            const bool okh = (isnan(w0) || fabsf(w0) > 60000.f || f16_to_f(gh & 0xFFFF) == w0) && (isnan(w1) || fabsf(w1) > 60000.f || f16_to_f(gh >> 16) == w1);
            if (!okb || !okh) {
                if (bad < 8) printf("  fp8 cvt mismatch lane %d sel %d: src %08x scale %g bf16 %08x f16 %08x want (%g, %g)\n", l, sel, hs[l], hsc[l], gb, gh, w0, w1);
                ++bad;
            }
        }
        for (int sel = 0; sel < 4; ++sel) {
            const uint32_t byte = (hs[l] >> (8 * sel)) & 0xFF;
            const float w0 = fp4_to_f(byte & 15) * hsc[l], w1 = fp4_to_f(byte >> 4) * hsc[l];
            const uint32_t gb = ho[l * 12 + 4 + sel], gh = ho[l * 12 + 8 + sel];
            const bool okb = bf16_to_f(gb & 0xFFFF) == w0 && bf16_to_f(gb >> 16) == w1;
            const bool okh = f16_to_f(gh & 0xFFFF) == w0 && f16_to_f(gh >> 16) == w1;
            if (!okb || !okh) {
                if (bad < 16) printf("  fp4 cvt mismatch lane %d sel %d: src %08x scale %g bf16 %08x f16 %08x want (%g, %g)\n", l, sel, hs[l], hsc[l], gb, gh, w0, w1);
                ++bad;
            }
        }
    }
    printf("converters: %d mismatches against (fp8: bytes 2s, 2s+1 -> lo, hi; fp4: byte s, low nibble -> lo)\n", bad);
    for (int l = 0; l < 3; ++l) {
        printf("  lane %d src %08x scale %g:", l, hs[l], hsc[l]);
        for (int j = 0; j < 12; ++j) printf(" %08x", ho[l * 12 + j]);
        printf("\n");
    }
    return 0;
}
