// probe_mx2.hip — DISCOVERY of the fp8 operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (probe_mx.hip showed that fp4 lanes hold
// k = 32 * (l >> 5) + nibble index, and that fp8 lanes do NOT hold k = 32 * (l >> 5) + byte index).
//   1. pairing: A one-hot at (lane half ha, byte ja) of row 0, B[(col 0, half hb), byte jb] = a distinct fp8 value per (hb, jb):
//      D[0][0] names the B position that meets the A position in the contraction;
//   2. scale association: which lane's scale register multiplies the element at (ha, ja);
//   3. the same pairing for A fp8 x B fp4 (base-7 digits over three B patterns) and A fp4 x B fp8.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// case c: A registers a[c][lane], B registers b[lane] (shared), scales sa[c][lane], sb[lane]; out[c] = D[row 0][col 0]
template <int FA, int FB>
__global__ void k(const v8i* a, const v8i* b, const int* sa, const int* sb, float* out, int ncases) {
    const int l = threadIdx.x;
    for (int c = 0; c < ncases; ++c) {
        v16f acc = {};
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[c * 64 + l], b[l], acc, FA, FB, 0, sa[c * 64 + l], 0, sb[l]);
        if (l == 0) out[c] = acc[0];
    }
}

static float fp8_to_f(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? m * 0.001953125f : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
static const float FP4T[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};

template <int FA, int FB>
static void run(const uint32_t* ha, const uint32_t* hb, const int* hsa, const int* hsb, float* hout, int ncases) {
    void *da, *db, *dsa, *dsb, *dout;
    CHECK(hipMalloc(&da, ncases * 64 * 32)); CHECK(hipMalloc(&db, 64 * 32)); CHECK(hipMalloc(&dsa, ncases * 64 * 4));
    CHECK(hipMalloc(&dsb, 64 * 4)); CHECK(hipMalloc(&dout, ncases * 4));
    CHECK(hipMemcpy(da, ha, ncases * 64 * 32, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, 64 * 32, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dsa, hsa, ncases * 64 * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dsb, hsb, 64 * 4, hipMemcpyHostToDevice));
    k<FA, FB><<<1, 64>>>((const v8i*)da, (const v8i*)db, (const int*)dsa, (const int*)dsb, (float*)dout, ncases);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hout, dout, ncases * 4, hipMemcpyDeviceToHost));
    hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dout);
}

static void set_byte(uint32_t* regs, int lane, int byte, uint32_t v) { regs[lane * 8 + byte / 4] |= v << (8 * (byte & 3)); }
static void set_nib(uint32_t* regs, int lane, int nib, uint32_t v) { regs[lane * 8 + nib / 8] |= v << (4 * (nib & 7)); }

int main() {
    static uint32_t ha[64 * 64 * 8], hb[64 * 8];
    static int hsa[64 * 64], hsb[64];
    static float out[64];
    // ---- 1. fp8 x fp8 pairing ----------------------------------------------------------------------------------------------
    memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
    for (int i = 0; i < 64 * 64; ++i) hsa[i] = 127;
    for (int i = 0; i < 64; ++i) hsb[i] = 127;
    for (int c = 0; c < 64; ++c) set_byte(ha + c * 64 * 8, /*lane*/ 32 * (c >> 5), c & 31, 0x38);  // 1.0 at (half c >> 5, byte c & 31) of row 0
    for (int hb_ = 0; hb_ < 2; ++hb_)
        for (int jb = 0; jb < 32; ++jb) set_byte(hb, 32 * hb_, jb, 0x08 + hb_ * 32 + jb);  // column 0 only
    run<0, 0>(ha, hb, hsa, hsb, out, 64);
    printf("== fp8 x fp8: A (half, byte) meets B (half, byte) ==\n");
    for (int c = 0; c < 64; ++c) {
        int found = -1;
        for (int q = 0; q < 64; ++q) if (fp8_to_f(0x08 + q) == out[c]) found = q;
        printf("A(%d,%2d)->B(%d,%2d)%s", c >> 5, c & 31, found >> 5, found & 31, (c & 3) == 3 ? "\n" : "   ");
    }
    // ---- 2. scale association (fp8 A): all ones; case c doubles the scale of lane (row 0, half c >> 6 ... ) ------------------
    printf("== fp8: which A elements does the scale register of lane (row 0, half s) multiply? ==\n");
    for (int s = 0; s < 2; ++s) {
        memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
        for (int i = 0; i < 64 * 64; ++i) hsa[i] = 127;
        for (int c = 0; c < 64; ++c) {
            set_byte(ha + c * 64 * 8, 32 * (c >> 5), c & 31, 0x38);
            hsa[c * 64 + 32 * s] = 129;  // x4 on lane (row 0, half s)
        }
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) set_byte(hb, l, j, 0x38);
        run<0, 0>(ha, hb, hsa, hsb, out, 64);
        printf("scale of half %d scales A elements:", s);
        for (int c = 0; c < 64; ++c) if (out[c] == 4.0f) printf(" (%d,%d)", c >> 5, c & 31);
        printf("\n");
    }
    // ---- 3. A fp8 x B fp4 pairing (three base-7 digit patterns) ------------------------------------------------------------
    printf("== fp8 (A) x fp4 (B): A (half, byte) meets B (half, nibble) ==\n");
    {
        int digit[3][64];
        for (int p = 0; p < 3; ++p) {
            memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
            for (int i = 0; i < 64 * 64; ++i) hsa[i] = 127;
            for (int c = 0; c < 64; ++c) set_byte(ha + c * 64 * 8, 32 * (c >> 5), c & 31, 0x38);
            for (int q = 0; q < 64; ++q) {
                int d = q; for (int t = 0; t < p; ++t) d /= 7;
                set_nib(hb, 32 * (q >> 5), q & 31, 1 + d % 7);
            }
            run<0, 4>(ha, hb, hsa, hsb, out, 64);
            for (int c = 0; c < 64; ++c) { digit[p][c] = -1; for (int v = 1; v < 8; ++v) if (FP4T[v] == out[c]) digit[p][c] = v - 1; }
        }
        for (int c = 0; c < 64; ++c) {
            const int q = (digit[0][c] < 0 || digit[1][c] < 0 || digit[2][c] < 0) ? -1 : digit[0][c] + 7 * digit[1][c] + 49 * digit[2][c];
            printf("A(%d,%2d)->B(%d,%2d)%s", c >> 5, c & 31, q < 0 ? -1 : q >> 5, q < 0 ? -1 : q & 31, (c & 3) == 3 ? "\n" : "   ");
        }
    }
    // ---- 4. A fp4 x B fp8 pairing: A nibble one-hot (value 1.0 = code 2) ---------------------------------------------------
    printf("== fp4 (A) x fp8 (B): A (half, nibble) meets B (half, byte) ==\n");
    memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
    for (int i = 0; i < 64 * 64; ++i) hsa[i] = 127;
    for (int c = 0; c < 64; ++c) set_nib(ha + c * 64 * 8, 32 * (c >> 5), c & 31, 2);
    for (int q = 0; q < 64; ++q) set_byte(hb, 32 * (q >> 5), q & 31, 0x08 + q);
    run<4, 0>(ha, hb, hsa, hsb, out, 64);
    for (int c = 0; c < 64; ++c) {
        int found = -1;
        for (int q = 0; q < 64; ++q) if (fp8_to_f(0x08 + q) == out[c]) found = q;
        printf("A(%d,%2d)->B(%d,%2d)%s", c >> 5, c & 31, found >> 5, found & 31, (c & 3) == 3 ? "\n" : "   ");
    }
    // ---- 5. scale association for B fp8 with A fp4 and for fp4 (sanity) ----------------------------------------------------
    printf("== fp4: which A elements does the scale register of lane (row 0, half s) multiply? ==\n");
    for (int s = 0; s < 2; ++s) {
        memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
        for (int i = 0; i < 64 * 64; ++i) hsa[i] = 127;
        for (int c = 0; c < 64; ++c) {
            set_nib(ha + c * 64 * 8, 32 * (c >> 5), c & 31, 2);
            hsa[c * 64 + 32 * s] = 129;
        }
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) set_nib(hb, l, j, 2);
        run<4, 4>(ha, hb, hsa, hsb, out, 64);
        printf("scale of half %d scales A elements:", s);
        for (int c = 0; c < 64; ++c) if (out[c] == 4.0f) printf(" (%d,%d)", c >> 5, c & 31);
        printf("\n");
    }
    return 0;
}
