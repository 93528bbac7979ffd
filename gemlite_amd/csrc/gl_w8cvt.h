// gl_w8cvt.h — 8 unpacked 8-bit weights (two dwords, k ascending) -> one MFMA B fragment of eight fp16 / bf16 values.
// Shared by a16w8_rows_kernel (gemm_a8w8.hip, round 4) and w8_rows_lds_kernel (gemm_w8_rows.hip, round 6): the two must convert identically.
//   int8 -> fp16: 0x6400 | (b ^ 0x80) = 1024 + (b + 128) exactly, minus 1152 (packed fp16 subtract: exact);
//   int8 -> bf16: through fp32 (exact, |b| <= 128 has 8 significant bits);  fp8 (e4m3 / e5m2) -> either: hardware converters (exact: both
//   16-bit types hold every fp8 value).
#pragma once
#include "gl_common.h"

namespace gl {

template <typename Tag, int WDT>
__device__ __forceinline__ u32x4 w8_to_frag(uint32_t lo, uint32_t hi) {
    using TR = F16Traits<Tag>;
    u32x4 f = {0u, 0u, 0u, 0u};
    const uint32_t d[2] = {lo, hi};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if constexpr (WDT == GEMLITE_DT_INT8) {
            if constexpr (TR::DT == GEMLITE_DT_FP16) {
                const uint32_t u = d[h] ^ 0x80808080u;  // b + 128 as an unsigned byte
                const h2_t off = {(_Float16)1152.0f, (_Float16)1152.0f};
                // {0x64, u.b1, 0x64, u.b0} / {0x64, u.b3, 0x64, u.b2}: 1024 + (b + 128)
                const uint32_t p0 = __builtin_amdgcn_perm(0x64646464u, u, 0x04010400u), p1 = __builtin_amdgcn_perm(0x64646464u, u, 0x04030402u);
                f[2 * h] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2_t, p0) - off);
                f[2 * h + 1] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2_t, p1) - off);
            } else {
                const int v = (int)d[h];
                const b2_t a = {(__bf16)(float)(int8_t)(v & 0xFF), (__bf16)(float)(int8_t)((v >> 8) & 0xFF)};
                const b2_t b = {(__bf16)(float)(int8_t)((v >> 16) & 0xFF), (__bf16)(float)(int8_t)((v >> 24) & 0xFF)};
                f[2 * h] = __builtin_bit_cast(uint32_t, a);
                f[2 * h + 1] = __builtin_bit_cast(uint32_t, b);
            }
        } else {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            constexpr bool E5 = WDT == GEMLITE_DT_FP8E5;
            const f32x2 a = E5 ? __builtin_amdgcn_cvt_pk_f32_bf8((int)d[h], false) : __builtin_amdgcn_cvt_pk_f32_fp8((int)d[h], false);
            const f32x2 b = E5 ? __builtin_amdgcn_cvt_pk_f32_bf8((int)d[h], true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)d[h], true);
            f[2 * h] = (uint32_t)TR::from_float(a[0]) | ((uint32_t)TR::from_float(a[1]) << 16);
            f[2 * h + 1] = (uint32_t)TR::from_float(b[0]) | ((uint32_t)TR::from_float(b[1]) << 16);
        }
    }
    return f;
}

}  // namespace gl
