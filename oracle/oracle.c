/* oracle.c — CPU ORACLE, TEST INFRASTRUCTURE ONLY (never linked into libgemlite_hip.so).
 *
 * Plain-C restatement of the reference's packed-weight matmul, used as a second, independent checker next
 * to oracle/gemlite_oracle.py and as the scalar leg of bench.py's cpu_baseline:
 *   unpack    : element k of column n = (W[k/e][n] >> ((k%e)*b)) & (2^b-1)
 *               (gemlite/bitpack.py:36-60, triton_kernels/gemm_kernels.py:327-328)
 *   dequantize: W_group_mode 0 none | 1 q-z | 2 q*s | 3 (q-z)*s | 4 fma(q,s,z')
 *               (gemlite/triton_kernels/utils.py:57-89), metadata [K/group, N]
 *   matmul    : y[m][n] = sum_k x[m][k] * w[k][n] in double, then channel scaling
 *               (gemm_kernels.py:392-404): mode 1 *sw[n], 2 *sx[m], 3 *sx[m]*sw[n]
 * Inputs are already widened to float / int32 by the caller (tests do that with numpy).
 * Parity of this file is pinned through tests/test_oracle_golden.py::test_c_oracle_matches_numpy_oracle,
 * the numpy oracle itself being pinned to the reference's golden outputs.
 */
#include <stdint.h>
#include <stddef.h>

int oracle_forward_packed(const float* x, const int32_t* w_packed, const float* scales, const float* zeros,
                          const float* scales_w_channel, const float* scales_x, double* y, int64_t M, int64_t N,
                          int64_t K, int nbits, int pack_bits, int group_size, int w_mode, int c_mode,
                          int zero_is_scalar) {
    if (nbits <= 0 || pack_bits % nbits != 0 || pack_bits > 32) return -1;
    const int e = pack_bits / nbits;
    const uint32_t mask = nbits >= 32 ? 0xFFFFFFFFu : ((1u << nbits) - 1u);
    if (K % e != 0 || group_size <= 0) return -2;
    for (int64_t m = 0; m < M; ++m) {
        for (int64_t n = 0; n < N; ++n) {
            double acc = 0.0;
            for (int64_t k = 0; k < K; ++k) {
                const uint32_t word = (uint32_t)w_packed[(k / e) * N + n];
                const double q = (double)((word >> ((k % e) * nbits)) & mask);
                const int64_t g = k / group_size;
                const double s = (w_mode >= 2) ? (double)scales[g * N + n] : 1.0;
                double z = 0.0;
                if (w_mode == 1 || w_mode >= 3) z = zero_is_scalar ? (double)zeros[0] : (double)zeros[g * N + n];
                double w;
                switch (w_mode) {
                    case 1: w = q - z; break;
                    case 2: w = q * s; break;
                    case 3: w = (q - z) * s; break;
                    case 4: w = q * s + z; break;
                    default: w = q; break;
                }
                acc += (double)x[m * K + k] * w;
            }
            if (c_mode == 1) acc *= (double)scales_w_channel[n];
            else if (c_mode == 2) acc *= (double)scales_x[m];
            else if (c_mode == 3) acc *= (double)scales_x[m] * (double)scales_w_channel[n];
            y[m * N + n] = acc;
        }
    }
    return 0;
}
