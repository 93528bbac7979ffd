// probe_mx16.hip — DISCOVERY of the operand and scale layout of v_mfma_scale_f32_16x16x128_f8f6f4 (prepared in round 3, not run yet:
// the GPU budget of the round was spent).  The 32x32x64 form's layout was measured in round 2 (probe_mx2.hip: an fp8 lane holds
// 16 bytes of EACH 32-k block, block s takes its scale from lane half s); a few-row kernel for the MX formats in the shape of
// a8w8_rows_kernel (16-column blocks, one 16-row MFMA per 128-k chunk straight from 32-byte loads: DESIGN.md §8) needs the same facts for
// the 16-row form: which (lane group q = lane >> 4, byte j) of A meets which (group, byte) of B, and whose scale register multiplies it.
//   1. pairing fp8 x fp8: A one-hot at (q, j) of row 0; B column 0 carries position codes in two base-64 digit passes
//   2. scale association: scale of lane (row 0, group s) x4 — which A elements quadruple
//   3. the same two questions for fp4 x fp4 (16 bytes = 32 nibbles per lane)
// Build: make -C scripts/ubench probe_mx16    Run on the MI355X: scripts/ubench/probe_mx16
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// case c: A registers a[c][lane], B registers b[lane] (shared), scales sa[c][lane], sb[lane]; out[c] = D[row 0][col 0] (lane 0, register 0)
template <int FA, int FB>
__global__ void k(const v8i* a, const v8i* b, const int* sa, const int* sb, float* out, int ncases) {
    const int l = threadIdx.x;
    for (int c = 0; c < ncases; ++c) {
        v4f acc = {};
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[c * 64 + l], b[l], acc, FA, FB, 0, sa[c * 64 + l], 0, sb[l]);
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

constexpr int NC = 128;  // A positions of one row: 4 lane groups x 32 bytes (fp8) / 4 x 32 nibbles (fp4)
int main() {
    static uint32_t ha[NC * 64 * 8], hb[64 * 8];
    static int hsa[NC * 64], hsb[64];
    static float out[2][NC];
    // ---- 1. fp8 x fp8 pairing: two passes, B position p = 32 qb + jb carries code 0x08 + (p % 64) / 0x08 + (p / 64) --------------
    for (int pass = 0; pass < 2; ++pass) {
        memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
        for (int i = 0; i < NC * 64; ++i) hsa[i] = 127;
        for (int i = 0; i < 64; ++i) hsb[i] = 127;
        for (int c = 0; c < NC; ++c) set_byte(ha + c * 64 * 8, 16 * (c >> 5), c & 31, 0x38);  // 1.0 at (group c >> 5, byte c & 31) of row 0
        for (int p = 0; p < NC; ++p) set_byte(hb, 16 * (p >> 5), p & 31, 0x08 + (pass ? p / 64 : p % 64));  // column 0 only
        run<0, 0>(ha, hb, hsa, hsb, out[pass], NC);
    }
    printf("== fp8 x fp8 (16x16x128): A (group, byte) meets B (group, byte) ==\n");
    for (int c = 0; c < NC; ++c) {
        int lo = -1, hi = -1;
        for (int q = 0; q < 64; ++q) { if (fp8_to_f(0x08 + q) == out[0][c]) lo = q; if (fp8_to_f(0x08 + q) == out[1][c]) hi = q; }
        const int p = (lo < 0 || hi < 0) ? -1 : hi * 64 + lo;
        printf("A(%d,%2d)->B(%d,%2d)%s", c >> 5, c & 31, p < 0 ? -1 : p >> 5, p < 0 ? -1 : p & 31, (c & 3) == 3 ? "\n" : "   ");
    }
    // ---- 2. scale association (fp8): the scale register of lane (row 0, group s) set to 129 (x4) ---------------------------------
    printf("== fp8: which A elements does the scale register of lane (row 0, group s) multiply?  (op_sel 0: scale byte 0) ==\n");
    for (int s = 0; s < 4; ++s) {
        memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
        for (int i = 0; i < NC * 64; ++i) hsa[i] = 127;
        for (int i = 0; i < 64; ++i) hsb[i] = 127;
        for (int c = 0; c < NC; ++c) {
            set_byte(ha + c * 64 * 8, 16 * (c >> 5), c & 31, 0x38);
            hsa[c * 64 + 16 * s] = 129;
        }
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) set_byte(hb, l, j, 0x38);
        run<0, 0>(ha, hb, hsa, hsb, out[0], NC);
        printf("scale of group %d scales A elements:", s);
        for (int c = 0; c < NC; ++c) if (out[0][c] == 4.0f) printf(" (%d,%d)", c >> 5, c & 31);
        printf("\n");
    }
    // ---- 3. fp4 x fp4: pairing through three base-7 digit passes over the 128 nibble positions, then the scale association ---------
    printf("== fp4 x fp4 (16x16x128): A (group, nibble) meets B (group, nibble) ==\n");
    {
        static float o3[3][NC];
        for (int pass = 0; pass < 3; ++pass) {
            memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
            for (int i = 0; i < NC * 64; ++i) hsa[i] = 127;
            for (int i = 0; i < 64; ++i) hsb[i] = 127;
            for (int c = 0; c < NC; ++c) set_nib(ha + c * 64 * 8, 16 * (c >> 5), c & 31, 2);  // 1.0
            for (int p = 0; p < NC; ++p) {
                int d = p; for (int t = 0; t < pass; ++t) d /= 7;
                set_nib(hb, 16 * (p >> 5), p & 31, 1 + d % 7);
            }
            run<4, 4>(ha, hb, hsa, hsb, o3[pass], NC);
        }
        for (int c = 0; c < NC; ++c) {
            int d[3];
            for (int pass = 0; pass < 3; ++pass) { d[pass] = -1; for (int v = 1; v < 8; ++v) if (FP4T[v] == o3[pass][c]) d[pass] = v - 1; }
            const int p = (d[0] < 0 || d[1] < 0 || d[2] < 0) ? -1 : d[0] + 7 * d[1] + 49 * d[2];
            printf("A(%d,%2d)->B(%d,%2d)%s", c >> 5, c & 31, p < 0 ? -1 : p >> 5, p < 0 ? -1 : p & 31, (c & 3) == 3 ? "\n" : "   ");
        }
        for (int s = 0; s < 4; ++s) {
            memset(ha, 0, sizeof(ha)); memset(hb, 0, sizeof(hb));
            for (int i = 0; i < NC * 64; ++i) hsa[i] = 127;
            for (int c = 0; c < NC; ++c) {
                set_nib(ha + c * 64 * 8, 16 * (c >> 5), c & 31, 2);
                hsa[c * 64 + 16 * s] = 129;
            }
            for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) set_nib(hb, l, j, 2);
            run<4, 4>(ha, hb, hsa, hsb, o3[0], NC);
            printf("fp4: scale of group %d scales A elements:", s);
            for (int c = 0; c < NC; ++c) if (o3[0][c] == 4.0f) printf(" (%d,%d)", c >> 5, c & 31);
            printf("\n");
        }
    }
    return 0;
}
