// gemm_mx.hip — block-scaled ("microscaling") formats: MXFP8 / MXFP4 weights and activations with one e8m0 scale per
// 32 k, NVFP4 (e4m3 scale per 16 k).  Replaces gemm_MX_kernel (gemlite/triton_kernels/gemm_kernels.py:422-547),
// gemm_splitK_MX_kernel (gemm_splitK_kernels.py:458-593) and the activation quantisers
// scale_activations_{mxfp8,mxfp4,nvfp4}_triton_v2 (gemlite/quant_utils.py:502-590, 769-855, 859-954).
//
// tl.dot_scaled(a, sa, fmt_a, b, sb, fmt_b, acc) is, on gfx950, ONE instruction: v_mfma_scale_f32_32x32x64_f8f6f4 multiplies a
// 32 x 64 by a 64 x 32 tile of fp8 / fp4 elements, each 32-k block of every row / column carrying its own e8m0 scale.
// Operand layout (measured: scripts/ubench/probe_mx.hip, probe_mx2.hip; profiles/r02/mx/):
//   fp4: lane l = row / column (l & 31); its 32 nibbles (4 dwords, low nibble first) are k = 32 (l >> 5) + e;
//        the lane's own scale register (byte 0) scales them;
//   fp8: lane l = row / column (l & 31); byte b of its 8 dwords is k = 32 (b >> 4) + 16 (l >> 5) + (b & 15): a lane holds
//        16 bytes of EACH of the two 32-k blocks; block s takes its scale from the lane with (l >> 5) == s;
//   D:   col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5), like every 32x32 MFMA.
// Memory layout fixed by the reference's pack() (core.py:363-398, 489-497): weights K-CONTIGUOUS per output column — fp8
// [N][K] bytes, fp4 [N][K/2] bytes (k even in the low nibble) — weight scales [K/32][N] bytes; activations row-major in
// the same element formats with scales [M_pad][K/32].  Both operands are therefore read as 16-byte pieces of rows and
// NOTHING is unpacked or dequantised in the K loop.
//
//   gemm_mx_mma_kernel   8 waves, tile (32 MI) x 128, the skeleton of gemm_a8w8_mma_kernel: x by LDS-DMA (XOR-swizzled
//                        through the source address), weights + both scale streams HBM -> registers, counted waits,
//                        split-K slabs.  fp8 / fp4 activations (A8W8 / A8W4 / A4W4 MXFP dynamic).
//   mx_generic_kernel    coverage: lane = output column, any strides, 16-bit activations x MX weights, and NVFP4 (an
//                        e4m3-scaled 16-k format the gfx950 matrix core has no instruction for).
//   act_quant_mx_kernel  the three activation quantisers, bit-exact restatements (thread = one block of 32 / 16 k).
#include <type_traits>

#include "gl_common.h"
#include "gl_async.h"

namespace gl {

typedef int v8i __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float fp4_to_float(uint32_t c) {  // e2m1: 0, 0.5, 1, 1.5, 2, 3, 4, 6; bit 3 = sign
    const uint32_t m = c & 7u;
    const float a = m < 2u ? 0.5f * (float)m : __builtin_bit_cast(float, ((126u + (m >> 1)) << 23) | ((m & 1u) << 22));
    return (c & 8u) ? -a : a;
}
__device__ __forceinline__ float e8m0_to_float(uint32_t b) {  // 2^(b - 127); 0 -> 2^-127 (an fp32 subnormal)
    return b == 0 ? __builtin_bit_cast(float, 0x00400000u) : __builtin_bit_cast(float, b << 23);
}

// ---------------------------------------------------------------------------------------------------------------------
// activation quantisers.  MODE 0: MXFP8 (e4m3 elements), 1: MXFP4 (e2m1 codes, two per byte), 2: NVFP4 (e2m1 codes, e4m3
// scale per 16 k, meta scale 0.05).  Arithmetic of the reference kernels, operation by operation:
//   MX:  s = 2^ceil(log2(amax / qmax)) through the exponent bits (next_power_of_2_bitwise_triton, quant_utils.py:383-392),
//        exponent clamped to [127 - 30, 254];  q = x / s  (clamped to +-448 and rounded to nearest even for fp8)
//   fp4: code = #{thresholds 0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5, 7 below |q|}, + 8 when q < 0  (quant_utils.py:800-802)
//   NV:  s8 = e4m3(min(amax / (6 * 0.05), 448));  q = x / max(float(s8) * 0.05, 1e-6)       (quant_utils.py:893-900)
// Rows M .. M_pad - 1 of the scale tensor (the reference pads M to a multiple of the group size and its programs store
// the scale of an all-zero block there) are written with that value; no element output exists for them.
// ---------------------------------------------------------------------------------------------------------------------
// One block of G values -> its scale byte and packed elements (MODE 0: 8 dwords of e4m3, MODE 1 / 2: G / 8 dwords of e2m1 codes).  The
// arithmetic of act_quant_mx_kernel, shared with the kernels that quantise a row themselves (mx_rows_kernel<..., FQ>).
template <int MODE>
__device__ __forceinline__ uint8_t mx_quant_block(const float (&v)[MODE == 2 ? 16 : 32], float amax, uint32_t (&o)[8]) {
    constexpr int G = MODE == 2 ? 16 : 32;
    float s, rinv = 1.f;
    bool use_mul = false;
    uint8_t sb;
    if (MODE == 2) {
        const float s32 = fminf(__fdiv_rn(amax, 0.3f), 448.f);  // 6 * 0.05 folded to the fp32 constant 0.3f
        sb = float_to_fp8e4m3(s32);
        s = fmaxf(fp8e4m3_to_float(sb) * 0.05f, 1e-6f);
    } else {
        const uint32_t xi = __builtin_bit_cast(uint32_t, __fdiv_rn(amax, MODE == 0 ? 448.f : 6.f));
        int ex = (int)((xi >> 23) & 0xFFu) + ((xi & 0x7FFFFFu) != 0u ? 1 : 0);
        ex = ex > 254 ? 254 : (ex < 97 ? 97 : ex);
        sb = (uint8_t)ex;
        s = __builtin_bit_cast(float, (uint32_t)ex << 23);
        // (round 3) x / 2^k == x * 2^-k bit for bit — one exact real value, one rounding — so the IEEE division (~12 VALU per
        // element) becomes a multiplication whenever 2^-k is a normal float (ex <= 253; 254 keeps the division)
        rinv = __builtin_bit_cast(float, (uint32_t)(254 - (ex > 253 ? 253 : ex)) << 23);
        use_mul = ex <= 253;
    }
    if (MODE == 0) {
        // hardware e4m3 converter (RNE, subnormals kept; the clamp keeps it away from overflow): two values per instruction
#pragma unroll
        for (int e4 = 0; e4 < 8; ++e4) {
            float q[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                q[t] = fminf(fmaxf(use_mul ? v[4 * e4 + t] * rinv : __fdiv_rn(v[4 * e4 + t], s), -448.f), 448.f);
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w, true);
            o[e4] = (uint32_t)w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < G; ++e) {
            const float q = (MODE == 1 && use_mul) ? v[e] * rinv : __fdiv_rn(v[e], s), a = fabsf(q);
            uint32_t c = (a > 0.25f) + (a > 0.75f) + (a > 1.25f) + (a > 1.75f) + (a > 2.5f) + (a > 3.5f) + (a > 5.0f) + (a > 7.0f);
            if (!(q >= 0.f)) c += 8u;
            // byte = lo | (hi << 4) in uint8 arithmetic, like the reference's pack (quant_utils.py:805-806): a code above 15 —
            // only when the block scale was floored (|q| > 7) — spills into / out of the byte exactly as it does there
            const uint32_t part = ((e & 1) ? ((c << 4) & 0xFFu) : c) << (8 * ((e & 7) >> 1));
            if ((e & 7) == 0) o[e >> 3] = part;
            else o[e >> 3] |= part;
        }
    }
    return sb;
}

template <int MODE>
__global__ __launch_bounds__(256) void act_quant_mx_kernel(const void* x, uint8_t* y, uint8_t* scales, int64_t M, int64_t M_pad,
                                                          int64_t K, int64_t stride_xm, int in_dt) {
    constexpr int G = MODE == 2 ? 16 : 32;
    const int64_t gpr = K / G;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M_pad * gpr) return;
    const int64_t m = idx / gpr, g = idx - m * gpr;
    float v[G];
    float amax = 0.f;
    if (m < M) {
        const int64_t base = m * stride_xm + g * G;
        const bool vec = (in_dt == GEMLITE_DT_FP16 || in_dt == GEMLITE_DT_BF16) && ((((uintptr_t)x) | (uintptr_t)(stride_xm * 2)) % 16 == 0);
        if (vec) {
#pragma unroll
            for (int q = 0; q < G / 8; ++q) {
                const u32x4 d = *(const u32x4*)((const uint16_t*)x + base + 8 * q);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint16_t hb = (uint16_t)(d[e >> 1] >> (16 * (e & 1)));
                    v[8 * q + e] = in_dt == GEMLITE_DT_FP16 ? F16Traits<half_tag>::to_float(hb) : F16Traits<bf16_tag>::to_float(hb);
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < G; ++e) v[e] = load_as_float(x, base + e, in_dt);
        }
#pragma unroll
        for (int e = 0; e < G; ++e) amax = fmaxf(amax, fabsf(v[e]));
    } else {
#pragma unroll
        for (int e = 0; e < G; ++e) v[e] = 0.f;
    }
    uint32_t o[8];
    const uint8_t sb = mx_quant_block<MODE>(v, amax, o);
    scales[idx] = sb;
    if (m >= M) return;
    if (MODE == 0) {
        u32x4* dst = (u32x4*)(y + m * K + g * 32);
        dst[0] = (u32x4){o[0], o[1], o[2], o[3]};
        dst[1] = (u32x4){o[4], o[5], o[6], o[7]};
    } else {
        uint8_t* dst = y + m * (K / 2) + g * (G / 2);
        if (G == 32) *(u32x4*)dst = (u32x4){o[0], o[1], o[2], o[3]};
        else *(u32x2*)dst = (u32x2){o[0], o[1]};
    }
}
const void* act_quant_mx_kernel_fn(int mode) {
    return mode == 0 ? (const void*)act_quant_mx_kernel<0> : (mode == 1 ? (const void*)act_quant_mx_kernel<1> : (const void*)act_quant_mx_kernel<2>);
}

// ---------------------------------------------------------------------------------------------------------------------
// coverage kernel: lane = output column, one row m per blockIdx.y; walks K block by block.  Any strides.
//   out[m, n] = post * sum_blocks  sx(m, blk) * sw(blk, n) * sum_{k in blk} x[m, k] * w[k, n]      (fp32 accumulation)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mx_elem(const void* base, int64_t row_off_bytes, int64_t k, int fmt) {
    if (fmt == MX_FP4) {
        const uint8_t b = ((const uint8_t*)base)[row_off_bytes + (k >> 1)];
        return fp4_to_float((k & 1) ? (b >> 4) : (b & 15));
    }
    if (fmt == MX_FP8) return fp8e4m3_to_float(((const uint8_t*)base)[row_off_bytes + k]);
    const uint16_t hbits = *(const uint16_t*)((const uint8_t*)base + row_off_bytes + 2 * k);
    return fmt == MX_F16 ? F16Traits<half_tag>::to_float(hbits) : F16Traits<bf16_tag>::to_float(hbits);
}

__global__ __launch_bounds__(256) void mx_generic_kernel(const GenericParams p) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t m = blockIdx.y;
    if (n >= p.N) return;
    const int G = p.group_size;
    // weights: element (k, n) at k * stride_wk + n * stride_wn (fp8) or byte (k / e) * stride_wk + n * stride_wn (fp4, e = 2)
    const bool w_kmajor = p.stride_wk == 1;
    const int xbytes = (p.mx_x == MX_F16 || p.mx_x == MX_BF16) ? 2 : 1;
    const int64_t xrow = m * p.stride_xm * xbytes;  // stride in elements of x's own dtype (fp4: bytes)
    float acc = 0.f;
    for (int64_t k0 = 0; k0 < p.K; k0 += G) {
        const int64_t blk = k0 / G;
        const uint8_t sb = ((const uint8_t*)p.scales)[blk * p.stride_meta_g + n * p.stride_meta_n];
        float s = p.mx_scale_e4m3 ? fp8e4m3_to_float(sb) : e8m0_to_float(sb);
        if (p.sx_blocks) {
            const uint8_t sa = ((const uint8_t*)p.sx_blocks)[m * p.stride_sx_blk_m + blk];
            s *= p.mx_scale_e4m3 ? fp8e4m3_to_float(sa) : e8m0_to_float(sa);
        }
        float part = 0.f;
        for (int k = 0; k < G; ++k) {
            const int64_t kk = k0 + k;
            float w;
            if (p.mx_w == MX_FP4) {
                const uint8_t b = ((const uint8_t*)p.w)[(kk >> 1) * p.stride_wk + n * p.stride_wn];
                w = fp4_to_float((kk & 1) ? (b >> 4) : (b & 15));
            } else {
                w = fp8e4m3_to_float(((const uint8_t*)p.w)[kk * p.stride_wk + n * p.stride_wn]);
            }
            part = __builtin_fmaf(mx_elem(p.x, xrow, kk, p.mx_x), w, part);
        }
        acc = __builtin_fmaf(part, s, acc);
    }
    (void)w_kmajor;
    epilogue_store(p.epi, acc * p.mx_post, m, n);
}
const void* mx_generic_kernel_fn() { return (const void*)mx_generic_kernel; }

// ---------------------------------------------------------------------------------------------------------------------
// NVFP4 activations -> fp16 (round 4).  gfx950's scaled MFMA takes e8m0 block-32 scales only; an e2m1 code times its e4m3 block-16
// scale has at most 5 significant bits and lies in [2^-10, 2688], i.e. it is EXACT in fp16 — so the NVFP4 x NVFP4 contraction
// (gemm_MX_kernel with the e4m3 scales, gemlite/triton_kernels/gemm_kernels.py:422-547: dequantise both operands block-wise, multiply,
// fp32 accumulate) runs on the fp16 MFMA tile kernel (gemm_wn_mma_kernel.inc, Geo<NVW4>) with both operands expanded exactly: the
// weights in its K loop, x here — M x K x 2 bytes into the workspace (2 MB at M = 256, K = 4096).  Thread = one 16-k block: 8 code
// bytes + 1 scale byte in, 32 bytes out.  `post[m]` receives the layer's constant output factor (meta_scale_norm^2,
// gemm_kernels.py:461, 530-531), which the tile kernel applies as a per-row scale in its epilogue.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nvfp4_expand_f16_kernel(const uint8_t* __restrict__ xq, const uint8_t* __restrict__ sx, uint16_t* __restrict__ out,
                                                              float* __restrict__ post, int M, int K, int64_t stride_xm, int64_t stride_sx_m, float post_v) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const int kb_per_row = K / 16;
    const int64_t bi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (bi >= (int64_t)M * kb_per_row) return;
    const int m = (int)(bi / kb_per_row), kb = (int)(bi % kb_per_row);
    const u32x2 codes = *(const u32x2*)(xq + (int64_t)m * stride_xm + kb * 8);
    const _Float16 hs = (_Float16)__builtin_amdgcn_cvt_f32_fp8((int)sx[(int64_t)m * stride_sx_m + kb], 0);
    const h2_t s2 = {hs, hs};
    uint32_t o[8];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        o[4 * d + 0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[d], 1.0f, 0) * s2);
        o[4 * d + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[d], 1.0f, 1) * s2);
        o[4 * d + 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[d], 1.0f, 2) * s2);
        o[4 * d + 3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[d], 1.0f, 3) * s2);
    }
    u32x4* dst = (u32x4*)(out + (int64_t)m * K + kb * 16);
    dst[0] = (u32x4){o[0], o[1], o[2], o[3]};
    dst[1] = (u32x4){o[4], o[5], o[6], o[7]};
    if (kb == 0) post[m] = post_v;
}
const void* nvfp4_expand_f16_kernel_fn() { return (const void*)nvfp4_expand_f16_kernel; }

// ---------------------------------------------------------------------------------------------------------------------
// decode kernel (M <= 4): one wave per output column, lanes along K with 16-byte pieces of the K-contiguous weight row (the
// streaming form of kmajor_matmul_kernel).  A piece is 32 fp4 / 16 fp8 weights of ONE microscaling block: the hardware
// converters (v_cvt_scalef32_pk_*) turn it into scaled 16-bit pairs (exact: e2m1 / e4m3 times 2^e fits bf16), the same
// for fp8 / fp4 activations with their block scale, and v_dot2_f32_{bf16,f16} accumulates in fp32.  Replaces the
// GEMM_SPLITK route the reference takes for MX decode (core.py:100-105).
// ---------------------------------------------------------------------------------------------------------------------
template <int XF, int WF, int MB>
__global__ __launch_bounds__(1024) void mx_gemv_kernel(const GenericParams p) {
    constexpr bool H16 = XF == MX_F16;            // arithmetic in fp16 pairs only when x is fp16; bf16 otherwise
    constexpr int CK = WF == MX_FP4 ? 32 : 16;    // k per 16-byte weight piece
    constexpr int NP = CK / 2;                    // 16-bit pairs per piece
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;  // 4 or 16 waves = adjacent columns: they share the scale lines
    const int64_t m0 = (int64_t)blockIdx.y * MB;
    if (n >= p.N) return;
    const uint8_t* wrow = (const uint8_t*)p.w + n * p.stride_wn;
    const uint8_t* srow = (const uint8_t*)p.scales + n * p.stride_meta_n;
    float acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i] = 0.f;
    auto dot = [](uint32_t a, uint32_t b, float c) -> float {
        if constexpr (H16) return F16Traits<half_tag>::dot2(a, b, c);
        else return F16Traits<bf16_tag>::dot2(a, b, c);
    };
    const int64_t pieces = p.K / CK;
    for (int64_t c = lane; c < pieces; c += 64) {
        const int64_t k0 = c * CK, kb = k0 >> 5;
        const u32x4 wv = *(const u32x4*)(wrow + c * 16);
        const float sw = __builtin_bit_cast(float, (uint32_t)srow[kb * p.stride_meta_g] << 23);
        uint32_t wp[NP];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if constexpr (WF == MX_FP4) {
                if constexpr (H16) {
                    wp[4 * d + 0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(wv[d], sw, 0));
                    wp[4 * d + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(wv[d], sw, 1));
                    wp[4 * d + 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(wv[d], sw, 2));
                    wp[4 * d + 3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(wv[d], sw, 3));
                } else {
                    wp[4 * d + 0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(wv[d], sw, 0));
                    wp[4 * d + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(wv[d], sw, 1));
                    wp[4 * d + 2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(wv[d], sw, 2));
                    wp[4 * d + 3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(wv[d], sw, 3));
                }
            } else {
                if constexpr (H16) {
                    wp[2 * d + 0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(wv[d], sw, false));
                    wp[2 * d + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(wv[d], sw, true));
                } else {
                    wp[2 * d + 0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wv[d], sw, false));
                    wp[2 * d + 1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wv[d], sw, true));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int64_t m = m0 + i;
            if (m >= p.M) continue;
            if constexpr (XF == MX_F16 || XF == MX_BF16) {
                const uint8_t* xr = (const uint8_t*)p.x + (m * p.stride_xm + k0) * 2;
#pragma unroll
                for (int q = 0; q < NP / 4; ++q) {
                    const u32x4 xv = *(const u32x4*)(xr + 16 * q);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i] = dot(wp[4 * q + t], xv[t], acc[i]);
                }
            } else {
                float sx = 1.0f;
                if (p.sx_blocks) sx = __builtin_bit_cast(float, (uint32_t)((const uint8_t*)p.sx_blocks)[m * p.stride_sx_blk_m + kb] << 23);
                if constexpr (XF == MX_FP8) {
                    const uint8_t* xr = (const uint8_t*)p.x + m * p.stride_xm + k0;
#pragma unroll
                    for (int q = 0; q < CK / 16; ++q) {
                        const u32x4 xv = *(const u32x4*)(xr + 16 * q);
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const uint32_t lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(xv[d], sx, false));
                            const uint32_t hi = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(xv[d], sx, true));
                            acc[i] = dot(wp[8 * q + 2 * d], lo, acc[i]);
                            acc[i] = dot(wp[8 * q + 2 * d + 1], hi, acc[i]);
                        }
                    }
                } else {  // fp4 activations (with fp4 weights: 32 k per piece = 16 bytes of x)
                    const u32x4 xv = *(const u32x4*)((const uint8_t*)p.x + m * p.stride_xm + (k0 >> 1));
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        acc[i] = dot(wp[4 * d + 0], __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(xv[d], sx, 0)), acc[i]);
                        acc[i] = dot(wp[4 * d + 1], __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(xv[d], sx, 1)), acc[i]);
                        acc[i] = dot(wp[4 * d + 2], __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(xv[d], sx, 2)), acc[i]);
                        acc[i] = dot(wp[4 * d + 3], __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(xv[d], sx, 3)), acc[i]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && m0 + i < p.M) epilogue_store(p.epi, v * p.mx_post, m0 + i, n);
    }
}

typedef void (*mx_gemv_fn)(const GenericParams);
template <int XF, int WF>
static const void* mx_gemv_pick(int mb) {
    mx_gemv_fn f = mb == 1 ? mx_gemv_kernel<XF, WF, 1> : mx_gemv_kernel<XF, WF, 4>;
    return (const void*)f;
}

// M <= 4, e8m0 scales per 32 k, K-contiguous weights.  (NVFP4 stays on the coverage kernel.)
bool plan_mx_gemv(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp) {
    if (a.M > 4 || g.mx_scale_e4m3 || g.group_size != 32) return false;
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.K % 32 != 0) return false;
    if (g.mx_x == MX_FP4 && g.mx_w != MX_FP4) return false;
    const int xb = (g.mx_x == MX_F16 || g.mx_x == MX_BF16) ? 2 : 1;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0 || (a.stride_xm * xb) % 16 != 0 || a.stride_wn % 16 != 0) return false;
    const int mb = a.M == 1 ? 1 : 4;
    const void* fn = nullptr;
    const bool w8 = g.mx_w == MX_FP8;
    switch (g.mx_x) {
        case MX_F16: fn = w8 ? mx_gemv_pick<MX_F16, MX_FP8>(mb) : mx_gemv_pick<MX_F16, MX_FP4>(mb); break;
        case MX_BF16: fn = w8 ? mx_gemv_pick<MX_BF16, MX_FP8>(mb) : mx_gemv_pick<MX_BF16, MX_FP4>(mb); break;
        case MX_FP8: fn = w8 ? mx_gemv_pick<MX_FP8, MX_FP8>(mb) : mx_gemv_pick<MX_FP8, MX_FP4>(mb); break;
        default: fn = mx_gemv_pick<MX_FP4, MX_FP4>(mb); break;
    }
    lp.fn = fn;
    lp.name = w8 ? "mx_gemv_w8_kernel" : "mx_gemv_w4_kernel";
    // waves (= adjacent columns) per block: measured at 4096^2, M = 1 (profiles/r02/mx): fp4 weights 16 waves 5.9 vs 6.9 us with 4
    // (the columns share the lines of the [K/32][N] scale bytes), fp8 weights 4 waves 8.1 vs 8.9 us.  tuning[3]: 1 = 4, 2 = 16.
    const int nw = a.tuning[3] == 1 ? 4 : (a.tuning[3] == 2 ? 16 : (w8 ? 4 : 16));
    lp.grid = dim3((unsigned)((a.N + nw - 1) / nw), (unsigned)((a.M + mb - 1) / mb), 1);
    lp.block = dim3(64 * nw, 1, 1);
    lp.lds_bytes = 0;
    lp.ws_bytes = 0;
    lp.slab_bytes = 0;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// Few rows (round 4; replaces gemm_splitK_MX_kernel, gemlite/triton_kernels/gemm_splitK_kernels.py:458-593, for 5 .. 64 rows — they
// ran on the 32-row tile of the 8-wave kernel below: 17-18 us at 4096^2, the chip 7/8 empty): the shape of a8w8_rows_kernel
// (gemm_a8w8.hip).  Block = 16 output columns (16 K-contiguous weight rows) x all of K, 8 waves dealing the 128-k chunks round-robin;
// a chunk is ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 16 rows of x whose operands come straight from 16-byte loads, nothing is
// unpacked.  Operand layout of that instruction (measured: scripts/ubench/probe_mx16.hip, profiles/r04/probe_mx16_layout.log), lane
// (r = lane & 15, q = lane >> 4) of row / column r:
//   fp4: its 16 bytes are k = 32 q + e (low nibble first); its scale register (byte 0) scales them — MX block q of the chunk;
//   fp8: byte b of its 32 is k = 64 (b >> 4) + 16 q + (b & 15) — 16 bytes of two different MX blocks — and block s of the chunk takes
//        its scale from the lane with q == s: the lane LOADS bytes of blocks q >> 1 and 2 + (q >> 1) and CARRIES the scale of block q;
//   D:   column lane & 15, rows 4 (lane >> 4) + reg.
// Mixed formats (fp8 x rows against fp4 weight rows) pair by k, each side in its own layout.  c_mode 2 (per-token fp32 scale in the
// epilogue): the activation block scale is the constant 127.  Rows >= M read zeros through the descriptor's range check.
// ---------------------------------------------------------------------------------------------------------------------
// FQ (round 4, one row): p.x is the UNQUANTISED 16-bit row; the block requests its first weights, then quantises the row block by block
// (mx_quant_block: the arithmetic of act_quant_mx_kernel, thread = one 32-k block) into LDS and reads its x fragments from there —
// layer(x) of the MXFP8 / MXFP4 dynamic layers at M = 1 in one launch instead of quantiser + matmul.
// FQ = 2: the same for the per-token form of the fp8 activations (channel_scale_mode 2, the processors' default `post_scale=True`): one
// fp32 scale for the row (amax / 448 over the block, IEEE divisions: the arithmetic of act_quant_per_token_kernel), unit block scales,
// the row scale applied in the epilogue.
template <int XF, int WF, int MT, int FQ = 0>
__global__ __launch_bounds__(512) void mx_rows_kernel(const GenericParams p) {
    static_assert(!FQ || MT == 1, "in-launch activation quantisation: one row");
    static_assert(FQ != 2 || XF == 0, "per-token scales: fp8 activations");
    constexpr int XV = XF == 0 ? 2 : 1, WV = WF == 0 ? 2 : 1;  // 16-byte pieces per fragment
    __shared__ __attribute__((aligned(16))) float red[MT][8][64][4];
    extern __shared__ __attribute__((aligned(16))) unsigned char xlds[];  // FQ: [row bytes of quantised x][K / 32 scale bytes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    const int mbase = (int)blockIdx.y * (16 * MT);  // (more than 64 rows: row tiles along grid.y — the fall-back below the tile kernels)
    const int nchunks = p.K / 128;
    const int xk_bytes = XF == 0 ? p.K : p.K / 2, wk_bytes = WF == 0 ? p.K : p.K / 2;  // bytes of one row
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)((int64_t)(p.N - 1) * p.stride_wn + wk_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, (short)0, (int)((int64_t)(p.M - 1) * p.stride_xm + xk_bytes), 0x00020000);
    const int blocks_k = p.K / 32;
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.scales, (short)0, (int)((int64_t)(blocks_k - 1) * p.stride_meta_g + (int64_t)(p.N - 1) * p.stride_meta_n + 1), 0x00020000);
    const bool blk_x = p.sx_blocks != nullptr;
    const int m_pad = (p.M + 31) / 32 * 32;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(blk_x ? p.sx_blocks : p.w), (short)0, blk_x ? (int)((int64_t)(m_pad - 1) * p.stride_sx_blk_m + blocks_k) : 4, 0x00020000);
    // byte offsets inside a 128-k chunk: fp8 pieces at 16 q and 64 + 16 q, the fp4 piece at 16 q (= k 32 q)
    const uint32_t wvoff = (uint32_t)((n0 + c) * p.stride_wn + q * 16);
    const uint32_t svoff = (uint32_t)((n0 + c) * p.stride_meta_n + (int64_t)q * p.stride_meta_g);
    uint32_t xvoff[MT], avoff[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = mbase + c + 16 * t;
        xvoff[t] = m < p.M ? (uint32_t)((int64_t)m * p.stride_xm + q * 16) : 0x80000000u;  // rows >= M: zeros
        avoff[t] = (uint32_t)((int64_t)m * p.stride_sx_blk_m + q);                            // (rows < m_pad exist)
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sx_row = 1.f;  // FQ = 2: the row's per-token scale
    constexpr int D = MT == 1 ? 4 : (MT == 2 ? 3 : 2);  // chunks in flight per wave
    struct Chunk { u32x4 w[WV]; uint32_t sw; u32x4 x[MT][XV]; uint32_t sx[MT]; };
    Chunk ring[D];
    const int mine = (nchunks - wave + 7) >> 3;  // chunks wave, wave + 8, ...
    auto load_w = [&](int slot, int i) __attribute__((always_inline)) {
        const int ch = wave + 8 * i;
        Chunk& k = ring[slot];
        const uint32_t wo = (uint32_t)__builtin_amdgcn_readfirstlane(ch * (WF == 0 ? 128 : 64));
        k.w[0] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wvoff, wo, 0);
        if constexpr (WV == 2) k.w[1] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wvoff, wo + 64u, 0);
        k.sw = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rsS, svoff, (uint32_t)__builtin_amdgcn_readfirstlane(ch * 4 * (int)p.stride_meta_g), 0);
    };
    auto load_x = [&](int slot, int i) __attribute__((always_inline)) {
        const int ch = wave + 8 * i;
        Chunk& k = ring[slot];
        if constexpr (FQ) {  // the one row, from LDS (lanes of row 0 only; every other row of the 16-row operand is zero)
            const unsigned char* src = xlds + ch * (XF == 0 ? 128 : 64) + q * 16;
            const u32x4 zero = {0u, 0u, 0u, 0u};
            k.x[0][0] = c == 0 ? *(const u32x4*)src : zero;
            if constexpr (XV == 2) k.x[0][1] = c == 0 ? *(const u32x4*)(src + 64) : zero;
            k.sx[0] = (FQ == 1 && c == 0) ? (uint32_t)xlds[xk_bytes + ch * 4 + q] : 127u;
            return;
        }
        const uint32_t xo = (uint32_t)__builtin_amdgcn_readfirstlane(ch * (XF == 0 ? 128 : 64));
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            k.x[t][0] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xvoff[t], xo, 0);
            if constexpr (XV == 2) k.x[t][1] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xvoff[t], xo + 64u, 0);
            k.sx[t] = blk_x ? (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rsA, avoff[t], (uint32_t)__builtin_amdgcn_readfirstlane(ch * 4), 0) : 127u;
        }
    };
    auto load = [&](int slot, int i) __attribute__((always_inline)) {
        load_w(slot, i);
        load_x(slot, i);
    };
    auto mma = [&](int slot) __attribute__((always_inline)) {
        const Chunk& k = ring[slot];
        v8i bv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bv[i] = (int)k.w[0][i];
            bv[4 + i] = WV == 2 ? (int)k.w[WV - 1][i] : 0;
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            v8i av;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                av[i] = (int)k.x[t][0][i];
                av[4 + i] = XV == 2 ? (int)k.x[t][XV - 1][i] : 0;
            }
            acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[t], XF, WF, 0, (int)k.sx[t], 0, (int)k.sw);
        }
    };
    if constexpr (FQ) {
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < mine) load_w(j, j);
        // thread = one 32-k block of the row: 64 bytes in, 32 (fp8) / 16 (fp4) bytes + the scale byte out
        const uint16_t* xr = (const uint16_t*)p.x;
        const bool f16 = p.x_dt == GEMLITE_DT_FP16;
        if constexpr (FQ == 2) {
            float* wmax = (float*)(xlds + ((xk_bytes + 15) & ~15));
            float amax = 0.f;
            for (int k = tid * 8; k < p.K; k += 512 * 8) {
                const u32x4 d = *(const u32x4*)(xr + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint16_t hb = (uint16_t)(d[e >> 1] >> (16 * (e & 1)));
                    amax = fmaxf(amax, fabsf(f16 ? F16Traits<half_tag>::to_float(hb) : F16Traits<bf16_tag>::to_float(hb)));
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
            if (lane == 0) wmax[wave] = amax;
            __syncthreads();
            amax = fmaxf(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])), fmaxf(fmaxf(wmax[4], wmax[5]), fmaxf(wmax[6], wmax[7])));
            sx_row = fmaxf(__fdiv_rn(amax, 448.f), 1e-6f);
            for (int k = tid * 8; k < p.K; k += 512 * 8) {
                const u32x4 d = *(const u32x4*)(xr + k);
                float tq[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint16_t hb = (uint16_t)(d[e >> 1] >> (16 * (e & 1)));
                    const float f = f16 ? F16Traits<half_tag>::to_float(hb) : F16Traits<bf16_tag>::to_float(hb);
                    tq[e] = fminf(fmaxf(__fdiv_rn(f, sx_row), -448.f), 448.f);
                }
                int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(tq[0], tq[1], 0, false);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(tq[2], tq[3], w0, true);
                int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(tq[4], tq[5], 0, false);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(tq[6], tq[7], w1, true);
                *(u32x2*)(xlds + k) = (u32x2){(uint32_t)w0, (uint32_t)w1};
            }
        } else
        for (int b = tid; b < blocks_k; b += 512) {
            float v[32];
            float amax = 0.f;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const u32x4 d = *(const u32x4*)(xr + b * 32 + 8 * qd);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint16_t hb = (uint16_t)(d[e >> 1] >> (16 * (e & 1)));
                    v[8 * qd + e] = f16 ? F16Traits<half_tag>::to_float(hb) : F16Traits<bf16_tag>::to_float(hb);
                    amax = fmaxf(amax, fabsf(v[8 * qd + e]));
                }
            }
            uint32_t o[8];
            const uint8_t sb = mx_quant_block<(XF == 0 ? 0 : 1)>(v, amax, o);
            xlds[xk_bytes + b] = sb;
            if constexpr (XF == 0) {
                u32x4* dst = (u32x4*)(xlds + b * 32);
                dst[0] = (u32x4){o[0], o[1], o[2], o[3]};
                dst[1] = (u32x4){o[4], o[5], o[6], o[7]};
            } else {
                *(u32x4*)(xlds + b * 16) = (u32x4){o[0], o[1], o[2], o[3]};
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < mine) load_x(j, j);
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < mine) load(j, j);
    }
    for (int base = 0; base < mine; base += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (base + j < mine) {
                mma(j);
                if (base + j + D < mine) load(j, base + j + D);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) *(f32x4*)&red[t][wave][lane][0] = acc[t];
    __syncthreads();
    for (int u = tid; u < MT * 256; u += 512) {
        const int t = u >> 8, l = u & 63, r = (u >> 6) & 3;
        const int m = mbase + 16 * t + 4 * (l >> 4) + r;  // C fragment of a 16 x 16 MFMA: column lane & 15, rows 4 (lane >> 4) + r
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[t][w][l][r];
        if constexpr (FQ == 2) {  // out = acc * s_x (channel_scale_mode 2: epilogue_scale() with the scale computed here)
            if (m < p.M) store_from_float(p.epi.out, (int64_t)m * p.epi.stride_om + (n0 + (l & 15)) * p.epi.stride_on, p.epi.out_dt, v * p.mx_post * sx_row);
        } else {
            if (m < p.M) epilogue_store(p.epi, v * p.mx_post, m, n0 + (l & 15));
        }
    }
}

// 1 <= M <= 64, same conditions as the scaled-MFMA kernel below plus K % 128 == 0 and the x re-read budget of a8w8_rows_kernel
// (8192^2 fp8: M = 32 25.7 vs 24.2 us, M = 64 42.0 vs 25.1 for the tile kernel).  tuning[0] = 4 forces it past the budget.
// any_m: the fall-back for shapes no tile kernel takes (fp4 activations with K % 512 != 0 — K = 11008 — ran on the coverage kernel:
// 4.5 ms at 4096 x 11008, M = 1): 64-row tiles along grid.y, any M.
// fq: one row, the activation quantiser inside the launch (`a` describes the call as the matmul sees it; the launch gets the raw row)
bool plan_mx_rows(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp, bool any_m, bool fq) {
    if (fq && (a.M != 1 || a.K > 49152)) return false;
    const bool fq_token = fq && a.channel_scale_mode == 2;  // per-token scale (fp8 activations only)
    if (fq_token && g.mx_x != MX_FP8) return false;
    if ((a.M > 64 && !any_m) || a.M < 1 || a.M > 65535 * 64 || g.mx_scale_e4m3 || g.group_size != 32) return false;
    if (!(g.mx_x == MX_FP8 || g.mx_x == MX_FP4) || (g.mx_x == MX_FP4 && g.mx_w != MX_FP4)) return false;
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.K % 128 != 0 || a.N % 16 != 0) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0 || a.stride_xm % 16 != 0 || a.stride_wn % 16 != 0) return false;
    if ((int64_t)a.M * a.stride_xm + a.K >= (1ll << 31) || (int64_t)a.N * a.stride_wn + a.K >= (1ll << 31)) return false;
    if ((int64_t)(a.K / 32) * a.stride_meta_g + (int64_t)a.N * a.stride_meta_n >= (1ll << 31)) return false;
    const int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
    const int64_t xrow = g.mx_x == MX_FP8 ? a.K : a.K / 2;
    if (mt > 1 && a.tuning[0] != 4 && !any_m && (int64_t)a.M * xrow * (a.N / 16) > (88ll << 20)) return false;  // every block re-reads its rows of x from L2
    // round 4 (profiles/r04/probe_rows_vs_tiles*.log), against the unsplit 64 x 64 tiles (K % 256 == 0, N % 64 == 0): fp4 x fp4 — ahead only
    // up to ~22 rows with fewer than 128 column tiles (4096^2 M = 40: 14.8 vs 12.7 us; 4096 x 14336: 33.6 vs 24.4) and up to 2 rows from 128
    // (8192^2 M = 4: 18.6 vs 16.2; 14336 x 4096 M = 40: 36.5 vs 14.5); fp8 activations — M N K <= 1.1 G from 128 column tiles (8192^2:
    // M = 16), <= 200 M from 192 (14336 x 4096: M = 3)
    if (a.M >= 2 && a.tuning[0] != 4 && !any_m && !fq && a.N % 64 == 0 && a.K % 256 == 0) {
        const int64_t ct = a.N / 64, mnk = (int64_t)a.M * a.N * a.K;
        if (g.mx_x == MX_FP4 ? (ct >= 128 ? a.M > 2 : a.M > 22) : (ct >= 192 ? mnk > 200000000ll : (ct >= 128 && mnk > 1100000000ll))) return false;
    }
    typedef void (*fn_t)(const GenericParams);
    fn_t fn = nullptr;
    auto pick = [&](auto xf, auto wf) -> fn_t {
        constexpr int XF = decltype(xf)::value, WF = decltype(wf)::value;
        if (fq) {
            if constexpr (XF == 0) { if (fq_token) return mx_rows_kernel<XF, WF, 1, 2>; }
            return mx_rows_kernel<XF, WF, 1, 1>;
        }
        return mt == 1 ? mx_rows_kernel<XF, WF, 1> : (mt == 2 ? mx_rows_kernel<XF, WF, 2> : mx_rows_kernel<XF, WF, 4>);
    };
    typedef std::integral_constant<int, 0> F8;
    typedef std::integral_constant<int, 4> F4;
    const bool x8 = g.mx_x == MX_FP8, w8 = g.mx_w == MX_FP8;
    fn = x8 ? (w8 ? pick(F8{}, F8{}) : pick(F8{}, F4{})) : pick(F4{}, F4{});
    lp.fn = (const void*)fn;
    static const char* names[3][3] = {{"mx_rows_a8w8_kernel<16x16>", "mx_rows_a8w8_kernel<32x16>", "mx_rows_a8w8_kernel<64x16>"},
                                      {"mx_rows_a8w4_kernel<16x16>", "mx_rows_a8w4_kernel<32x16>", "mx_rows_a8w4_kernel<64x16>"},
                                      {"mx_rows_a4w4_kernel<16x16>", "mx_rows_a4w4_kernel<32x16>", "mx_rows_a4w4_kernel<64x16>"}};
    static const char* fq_names[3] = {"mx_rows_a8w8_fused_quant_kernel<16x16>", "mx_rows_a8w4_fused_quant_kernel<16x16>", "mx_rows_a4w4_fused_quant_kernel<16x16>"};
    lp.name = fq ? fq_names[x8 ? (w8 ? 0 : 1) : 2] : names[x8 ? (w8 ? 0 : 1) : 2][mt == 1 ? 0 : (mt == 2 ? 1 : 2)];
    lp.grid = dim3((unsigned)(a.N / 16), (unsigned)((a.M + 16 * mt - 1) / (16 * mt)), 1);
    lp.block = dim3(512, 1, 1);
    lp.lds_bytes = fq ? (size_t)(xrow + a.K / 32 + 64) : 0;
    lp.ws_bytes = 0;
    lp.slab_bytes = 0;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// NVFP4 x NVFP4, few rows (round 4): the rows shape with BOTH operands expanded in registers.  An e2m1 code times its e4m3 block-16 scale is
// exact in fp16 (nvfp4_expand_f16_kernel above), a lane's 16 k of a 64-k chunk are exactly ONE block of either operand, so lane (c, q)
// loads 8 code bytes + 1 scale byte of weight row n0 + c and of x row c (+ 16 t), converts each to 16 fp16 values (v_cvt_scalef32_pk_f16_fp4
// at scale 1 + v_pk_mul_f16 by the block scale) and issues two v_mfma_f32_16x16x32_f16 per row tile.  The layer's constant 0.05^2 multiplies
// the fp32 sum.  FQ (one row): p.x is the unquantised 16-bit row, quantised per 16-k block into LDS first (mx_quant_block<2>: the arithmetic
// of the NVFP4 quantiser kernel) under the weight round trip.  The 32-row tile of gemm_nvfp4_f16_kernel took 15-17 us at these sizes.
// ---------------------------------------------------------------------------------------------------------------------
template <int MT, bool FQ>
__global__ __launch_bounds__(512) void nvfp4_rows_kernel(const GenericParams p) {
    static_assert(!FQ || MT == 1, "in-launch activation quantisation: one row");
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    typedef _Float16 h8v __attribute__((ext_vector_type(8)));
    __shared__ __attribute__((aligned(16))) float red[MT][8][64][4];
    extern __shared__ __attribute__((aligned(16))) unsigned char xlds[];  // FQ: [K / 2 code bytes][K / 16 scale bytes]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    const int mbase = (int)blockIdx.y * (16 * MT);
    const int nchunks = p.K / 64, blocks_k = p.K / 16, row_bytes = p.K / 2;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)((int64_t)(p.N - 1) * p.stride_wn + row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.scales, (short)0, (int)((int64_t)(blocks_k - 1) * p.stride_meta_g + (int64_t)(p.N - 1) * p.stride_meta_n + 1), 0x00020000);
    const int m_pad = (p.M + 15) / 16 * 16;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(FQ ? p.w : p.x), (short)0, FQ ? 4 : (int)((int64_t)(p.M - 1) * p.stride_xm + row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(FQ ? p.w : p.sx_blocks), (short)0, FQ ? 4 : (int)((int64_t)(m_pad - 1) * p.stride_sx_blk_m + blocks_k), 0x00020000);
    const uint32_t wvoff = (uint32_t)((n0 + c) * p.stride_wn + q * 8);
    const uint32_t svoff = (uint32_t)((n0 + c) * p.stride_meta_n + (int64_t)q * p.stride_meta_g);
    uint32_t xvoff[MT], avoff[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = mbase + c + 16 * t;
        xvoff[t] = m < p.M ? (uint32_t)((int64_t)m * p.stride_xm + q * 8) : 0x80000000u;  // rows >= M: zeros
        avoff[t] = m < p.M ? (uint32_t)((int64_t)m * p.stride_sx_blk_m + q) : 0x80000000u;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int D = MT == 1 ? 8 : (MT == 2 ? 6 : 4);  // chunks in flight per wave (8 + 8 MT bytes per lane each)
    struct Chunk { u32x2 w; uint32_t sw; u32x2 x[MT]; uint32_t sx[MT]; };
    Chunk ring[D];
    const int mine = (nchunks - wave + 7) >> 3;
    auto load_w = [&](int slot, int i) __attribute__((always_inline)) {
        const int ch = wave + 8 * i;
        ring[slot].w = __builtin_amdgcn_raw_buffer_load_b64(rsW, wvoff, (uint32_t)__builtin_amdgcn_readfirstlane(ch * 32), 0);
        ring[slot].sw = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rsS, svoff, (uint32_t)__builtin_amdgcn_readfirstlane(ch * 4 * (int)p.stride_meta_g), 0);
    };
    auto load_x = [&](int slot, int i) __attribute__((always_inline)) {
        const int ch = wave + 8 * i;
        if constexpr (FQ) {
            const u32x2 zero = {0u, 0u};
            ring[slot].x[0] = c == 0 ? *(const u32x2*)(xlds + ch * 32 + q * 8) : zero;
            ring[slot].sx[0] = c == 0 ? (uint32_t)xlds[row_bytes + ch * 4 + q] : 0u;
            return;
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            ring[slot].x[t] = __builtin_amdgcn_raw_buffer_load_b64(rsX, xvoff[t], (uint32_t)__builtin_amdgcn_readfirstlane(ch * 32), 0);
            ring[slot].sx[t] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rsA, avoff[t], (uint32_t)__builtin_amdgcn_readfirstlane(ch * 4), 0);
        }
    };
    // 8 code bytes + the e4m3 block scale -> 16 fp16 values (two MFMA operands of 8)
    auto expand = [&](u32x2 codes, uint32_t sbyte, u32x4& lo, u32x4& hi) __attribute__((always_inline)) {
        const _Float16 hs = (_Float16)__builtin_amdgcn_cvt_f32_fp8((int)sbyte, 0);
        const h2v s2 = {hs, hs};
        lo[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[0], 1.0f, 0) * s2);
        lo[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[0], 1.0f, 1) * s2);
        lo[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[0], 1.0f, 2) * s2);
        lo[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[0], 1.0f, 3) * s2);
        hi[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[1], 1.0f, 0) * s2);
        hi[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[1], 1.0f, 1) * s2);
        hi[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[1], 1.0f, 2) * s2);
        hi[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(codes[1], 1.0f, 3) * s2);
    };
    auto mma = [&](int slot) __attribute__((always_inline)) {
        const Chunk& k = ring[slot];
        u32x4 b0, b1;
        expand(k.w, k.sw, b0, b1);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            u32x4 a0, a1;
            expand(k.x[t], k.sx[t], a0, a1);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a0), __builtin_bit_cast(h8v, b0), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a1), __builtin_bit_cast(h8v, b1), acc[t], 0, 0, 0);
        }
    };
    if constexpr (FQ) {
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < mine) load_w(j, j);
        const uint16_t* xr = (const uint16_t*)p.x;
        const bool f16 = p.x_dt == GEMLITE_DT_FP16;
        for (int b = tid; b < blocks_k; b += 512) {  // thread = one 16-k block: 32 bytes in, 8 code bytes + the scale byte out
            float v[16];
            float amax = 0.f;
#pragma unroll
            for (int qd = 0; qd < 2; ++qd) {
                const u32x4 d = *(const u32x4*)(xr + b * 16 + 8 * qd);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint16_t hb = (uint16_t)(d[e >> 1] >> (16 * (e & 1)));
                    v[8 * qd + e] = f16 ? F16Traits<half_tag>::to_float(hb) : F16Traits<bf16_tag>::to_float(hb);
                    amax = fmaxf(amax, fabsf(v[8 * qd + e]));
                }
            }
            uint32_t o[8];
            xlds[row_bytes + b] = mx_quant_block<2>(v, amax, o);
            *(u32x2*)(xlds + b * 8) = (u32x2){o[0], o[1]};
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < mine) load_x(j, j);
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < mine) { load_w(j, j); load_x(j, j); }
    }
    for (int base = 0; base < mine; base += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (base + j < mine) {
                mma(j);
                if (base + j + D < mine) { load_w(j, base + j + D); load_x(j, base + j + D); }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) *(f32x4*)&red[t][wave][lane][0] = acc[t];
    __syncthreads();
    for (int u = tid; u < MT * 256; u += 512) {
        const int t = u >> 8, l = u & 63, r = (u >> 6) & 3;
        const int m = mbase + 16 * t + 4 * (l >> 4) + r;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[t][w][l][r];
        if (m < p.M) epilogue_store(p.epi, v * p.mx_post, m, n0 + (l & 15));
    }
}

// NVFP4 layers, 1 <= M <= 64 (x re-read budget like the other rows kernels); fq: one row, the quantiser inside
bool plan_nvfp4_rows(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp, bool fq) {
    if (!g.mx_scale_e4m3 || g.group_size != 16 || g.mx_x != MX_FP4 || g.mx_w != MX_FP4) return false;
    if (a.M < 1 || a.M > 64 || (fq && (a.M != 1 || a.K > 65536))) return false;
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.K % 64 != 0 || a.N % 16 != 0) return false;
    if (((uintptr_t)a.w_q % 8) != 0 || a.stride_wn % 8 != 0) return false;
    if (!fq && (((uintptr_t)a.x % 8) != 0 || a.stride_xm % 8 != 0)) return false;
    if ((int64_t)a.M * a.stride_xm + a.K >= (1ll << 31) || (int64_t)a.N * a.stride_wn + a.K >= (1ll << 31)) return false;
    if ((int64_t)(a.K / 16) * a.stride_meta_g + (int64_t)a.N * a.stride_meta_n >= (1ll << 31)) return false;
    const int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
    if (mt > 1 && a.tuning[0] != 4 && (int64_t)a.M * (a.K / 2) * (a.N / 16) > (88ll << 20)) return false;
    // round 4 (profiles/r04/probe_rows_vs_tiles*.log): against the fp16 tile kernel the crossover sits at M N K ~ 600 M (4096^2: M = 36 of the
    // measured 44, 8192^2: 9, 14336 x 4096: 10 — M = 48 there: 60.6 vs 24.7 us)
    if (a.M >= 2 && a.tuning[0] != 4 && !fq && a.N % 128 == 0 && a.K % 128 == 0 && (int64_t)a.M * a.N * a.K > 600000000ll) return false;
    typedef void (*fn_t)(const GenericParams);
    fn_t fn = fq ? nvfp4_rows_kernel<1, true> : (mt == 1 ? nvfp4_rows_kernel<1, false> : (mt == 2 ? nvfp4_rows_kernel<2, false> : nvfp4_rows_kernel<4, false>));
    lp.fn = (const void*)fn;
    lp.name = fq ? "nvfp4_rows_fused_quant_kernel<16x16>" : (mt == 1 ? "nvfp4_rows_kernel<16x16>" : (mt == 2 ? "nvfp4_rows_kernel<32x16>" : "nvfp4_rows_kernel<64x16>"));
    lp.grid = dim3((unsigned)(a.N / 16), 1, 1);
    lp.block = dim3(512, 1, 1);
    lp.lds_bytes = fq ? (size_t)(a.K / 2 + a.K / 16 + 16) : 0;
    lp.ws_bytes = 0;
    lp.slab_bytes = 0;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// scaled-MFMA kernel.  AF / BF: element format of x / w as the instruction's cbsz / blgp code (0 = fp8 e4m3, 4 = fp4 e2m1).
// A K step moves 256 BYTES of every x row (256 k of fp8, 512 k of fp4); wave (cg, kh) owns all rows x 32 columns x the
// kh-th half of the step = NS slices of 64 k.  Per slice and lane: A fragment = 16 (fp4) or 2 x 16 (fp8) bytes read from
// the LDS stage, B fragment = the same amount straight from the weight row (tracked buffer loads, PD steps ahead), one
// weight-scale byte, and per row block one dword of activation scales (4 blocks; shifted so that this lane's block is
// byte 0).  c_mode 2 (per-token fp32 scale applied in the epilogue): the activation block scale is the constant 127.
// ---------------------------------------------------------------------------------------------------------------------
template <int AF, int BF, int MI, int RD, int NST>
__global__ __launch_bounds__(512, 2) void gemm_mx_mma_kernel(const GenericParams p) {
    using namespace async;
    constexpr int AV = AF == 0 ? 2 : 1, BV = BF == 0 ? 2 : 1;      // 16-byte pieces per fragment
    constexpr int BM = 32 * MI, BN = 128;
    constexpr int PITCH = 256, STAGE = BM * PITCH;                 // bytes of x per row and step / per stage
    constexpr int KSTEP = AF == 0 ? 256 : 512, KW = KSTEP / 2;     // k per step / per wave half
    constexpr int NS = KW / 64;                                    // 64-k slices per wave and step (2 or 4)
    constexpr int NSA = NS / 2;                                    // dwords of activation scales per row block and step
    constexpr int PIECES = STAGE / 1024 / 8;                       // LDS-DMA pieces per wave and stage (= MI)
    constexpr int NQ = NS * MI, L = NQ / 2 < 4 ? NQ / 2 : 4;
    constexpr int C_ROWS = 128, C_PITCH = BN + 4;
    constexpr int PD = RD - 2;
    constexpr int WB64 = BF == 0 ? 64 : 32;                        // weight bytes per 64 k
    static_assert(PIECES >= 1 && NQ >= 2 * L && L >= 1 && RD % NST == 0 && RD >= 4 && NST >= 2, "tile too small for the slot schedule");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, kh = wave >> 2;
    const int col = lane & 31, h = lane >> 5;
    const int mtiles = (p.M + BM - 1) / BM;
    const int bid = blockIdx.x, slice = blockIdx.y;
    const int mt = bid % mtiles, nt = bid / mtiles;
    const int m0 = mt * BM;
    const int n = nt * BN + cg * 32 + col;

    const int units = p.K / KSTEP;
    const int s_begin = (int)((int64_t)slice * units / p.splitk), s_end = (int)((int64_t)(slice + 1) * units / p.splitk);
    const int nsteps = s_end - s_begin;
    const int k_s0 = s_begin * KSTEP;
    const int xk_bytes = AF == 0 ? p.K : p.K / 2, wk_bytes = BF == 0 ? p.K : p.K / 2;  // bytes of one row
    const int xs0 = AF == 0 ? k_s0 : k_s0 / 2, ws0 = BF == 0 ? k_s0 : k_s0 / 2;       // byte offset of the slice inside a row

    const srd_t rsX = make_srd(p.x, (uint32_t)((int64_t)(p.M - 1) * p.stride_xm + xk_bytes));
    const __amdgpu_buffer_rsrc_t brW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.w, (short)0, (int)((int64_t)(p.N - 1) * p.stride_wn + wk_bytes), 0x00020000);
    const int blocks_k = p.K / 32;
    const __amdgpu_buffer_rsrc_t brS = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.scales, (short)0, (int)((int64_t)(blocks_k - 1) * p.stride_meta_g + (int64_t)(p.N - 1) * p.stride_meta_n + 1), 0x00020000);
    // activation block scales: rows >= M_pad read zero (2^-127) through the range check — their x rows are zero anyway
    const bool blk_x = p.sx_blocks != nullptr;
    const int m_pad = (p.M + 31) / 32 * 32;
    const __amdgpu_buffer_rsrc_t brA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(blk_x ? p.sx_blocks : p.w), (short)0, blk_x ? (int)((int64_t)(m_pad - 1) * p.stride_sx_blk_m + blocks_k) : 4, 0x00020000);

    // B fragment of slice g: fp8: bytes [16 h, +16) and [32 + 16 h, +16) of the 64 bytes of the slice; fp4: bytes [16 h, +16) of its 32
    const uint32_t wvoff = (uint32_t)((int64_t)n * p.stride_wn + ws0 + kh * (KW * WB64 / 64) + h * 16);
    const uint32_t svoff = (uint32_t)((int64_t)n * p.stride_meta_n + (int64_t)(k_s0 / 32 + kh * (KW / 32) + h) * p.stride_meta_g);
    uint32_t avoff[MI];  // dword of 4 activation block scales of row (mi * 32 + col) for this wave half
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int r = m0 + mi * 32 + col;
        avoff[mi] = blk_x ? (uint32_t)((int64_t)r * p.stride_sx_blk_m + k_s0 / 32 + kh * (KW / 32)) : 0u;
    }
    const uint32_t sh_h = 8u * (uint32_t)h;
    struct BStep { u32x4 w[NS][BV]; uint32_t s[NS]; uint32_t a[MI][NSA]; };
    constexpr int NLB = NS * (BV + 1) + MI * NSA;  // tracked requests per step and wave
    auto req_b = [&](BStep& b, int step, int it) {
        if (it < NS * (BV + 1)) {
            const int g = it / (BV + 1), i = it % (BV + 1);
            if (i < BV) {
                b.w[g][i] = __builtin_amdgcn_raw_buffer_load_b128(
                    brW, wvoff, (uint32_t)__builtin_amdgcn_readfirstlane(step * (KSTEP * WB64 / 64) + g * WB64 + i * 32), 0);
            } else {
                b.s[g] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(
                    brS, svoff, (uint32_t)__builtin_amdgcn_readfirstlane((step * (KSTEP / 32) + 2 * g) * (int)p.stride_meta_g), 0);
            }
        } else {
            const int j = it - NS * (BV + 1), mi = j / NSA, d = j % NSA;
            b.a[mi][d] = __builtin_amdgcn_raw_buffer_load_b32(
                brA, avoff[mi], (uint32_t)__builtin_amdgcn_readfirstlane(blk_x ? step * (KSTEP / 32) + 4 * d : 0), 0);
        }
    };
    // x: piece j of wave w covers LDS bytes [(w * PIECES + j) * 1024, +1024) of a stage: row = byte / 256, physical 16-byte slot
    // (byte % 256) / 16 holds the logical slot phys ^ (row & 15)
    uint32_t xvoff[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const int byte = (wave * PIECES + j) * 1024 + lane * 16;
        const int r = byte / PITCH, phys = (byte % PITCH) / 16;
        const int logical = phys ^ (r & 15);
        xvoff[j] = m0 + r < p.M ? (uint32_t)((int64_t)(m0 + r) * p.stride_xm + xs0 + logical * 16) : 0x80000000u;
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (uint32_t)(wave * PIECES) * 1024u);
    auto req_x = [&](int stage, int step, int j) {
        req_lds16(rsX, lds0 + (uint32_t)(stage * STAGE + j * 1024), xvoff[j], (uint32_t)__builtin_amdgcn_readfirstlane(step * PITCH));
    };
    // A fragment of (slice g, row block mi), piece v: row mi*32 + col, logical slot (kh*128 + g*(64 AV/2...) ...
    //   fp8: slice g spans bytes [kh*128 + g*64, +64): slots 4g' + {h, 2 + h};  fp4: bytes [kh*128 + g*32, +32): slot 2g' + h
    int fbase[NST][NS][AV];
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int g = 0; g < NS; ++g)
#pragma unroll
            for (int v = 0; v < AV; ++v) {
                const int byte = kh * 128 + (AF == 0 ? g * 64 + v * 32 + h * 16 : g * 32 + h * 16);
                const int slot = byte >> 4;
                fbase[st][g][v] = st * STAGE + col * PITCH + (((slot ^ col) & 15) << 4);
            }
    struct AFrag { u32x4 v[AV]; };
    auto read_frag = [&](int stage, int q) -> AFrag {
        AFrag f;
#pragma unroll
        for (int v = 0; v < AV; ++v) f.v[v] = *(const u32x4*)(smem + fbase[stage][q / MI][v] + (q % MI) * 32 * PITCH);
        return f;
    };
    auto mma = [&](const AFrag& a, const BStep& b, int g, int mi, f32x16 c) -> f32x16 {
        v8i av, bv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            av[i] = (int)a.v[0][i];
            av[4 + i] = AV == 2 ? (int)a.v[AV - 1][i] : 0;
            bv[i] = (int)b.w[g][0][i];
            bv[4 + i] = BV == 2 ? (int)b.w[g][BV - 1][i] : 0;
        }
        // this lane's activation block scale: block (2 g + h) of the wave half = byte (2 (g & 1) + h) of dword g / 2
        const uint32_t sa = blk_x ? (b.a[mi][g >> 1] >> (sh_h + 16u * (uint32_t)(g & 1))) : 127u;
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, AF, BF, 0, (int)sa, 0, (int)b.s[g]);
    };

    f32x16 acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    BStep ring[RD];
    AFrag af[L];

    // ---- prologue: x of step 0, weights / scales of steps 0 .. PD-1 (see gemm_wn_mma.hip) ----------------------------------
#pragma unroll
    for (int j = 0; j < PIECES; ++j) req_x(0, 0, j);
#pragma unroll
    for (int it = 0; it < NLB; ++it) req_b(ring[0], 0, it);
#pragma unroll
    for (int st = 1; st < NST - 1; ++st)
#pragma unroll
        for (int j = 0; j < PIECES; ++j) req_x(st, st < nsteps ? st : nsteps - 1, j);
#pragma unroll
    for (int r = 1; r < PD; ++r)
#pragma unroll
        for (int it = 0; it < NLB; ++it) req_b(ring[r], r < nsteps ? r : nsteps - 1, it);
    {
        constexpr int AFTER = (NST - 2) * PIECES + PD * NLB;
        wait_vm<(AFTER < 63 ? AFTER : 63)>();
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < L; ++q) af[q] = read_frag(0, q);
    __builtin_amdgcn_sched_barrier(0);

    constexpr int NL = NLB + PIECES, NQI = NQ - L;
    constexpr int RPS = (NL + NQI - 1) / NQI;
    static_assert((NST - 2) * PIECES + (NST - 1) * NLB + NL <= 63, "vmcnt is a 6-bit counter");
    auto do_step = [&](auto Jc, int step) {
        constexpr int J = decltype(Jc)::value;
        constexpr int stage = J % NST, stage_next = (J + 1) % NST, stage_fill = (J + NST - 1) % NST;
        const BStep& bc = ring[J];
        BStep& bl = ring[(J + PD) % RD];
        const int lstep = step + PD < nsteps ? step + PD : nsteps - 1;  // run-ahead repeats the last step (never consumed)
        const int xstep = step + NST - 1 < nsteps ? step + NST - 1 : nsteps - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int g = q / MI, mi = q % MI;
            acc[mi] = mma(af[q % L], bc, g, mi, acc[mi]);
            if (q == NQI) {
                wait_vm<(NST - 2) * PIECES + (NST - 1) * NLB>();  // the x DMA of step + 1 has landed
                __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): this wave's reads of the current stage are complete
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (q + L < NQ) af[q % L] = read_frag(stage, q + L);
            else af[q % L] = read_frag(stage_next, q + L - NQ);
#pragma unroll
            for (int it = q * RPS; it < (q + 1) * RPS && it < NL; ++it) {
                if (it < PIECES) req_x(stage_fill, xstep, it);
                else req_b(bl, lstep, it - PIECES);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto chain = [&](auto self, auto Jc, int s0) -> void {
        constexpr int J = decltype(Jc)::value;
        do_step(std::integral_constant<int, J % RD>{}, s0 + J);
        if constexpr (J + 1 < RD) {
            if (s0 + J + 1 < nsteps) self(self, std::integral_constant<int, J + 1>{}, s0);
        }
    };
    for (int s0 = 0; s0 < nsteps; s0 += RD) chain(chain, std::integral_constant<int, 0>{}, s0);
    wait_vm<0>();

    // ---- epilogue 1: add the two K halves through LDS ------------------------------------------------------------------------
    __syncthreads();
    {
        f32x16* xch = (f32x16*)smem;  // [cg][mi][lane] whole accumulators
        if (kh == 1) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xch[(cg * MI + mi) * 64 + lane] = acc[mi];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] += xch[(cg * MI + mi) * 64 + lane];
        }
    }
    // ---- epilogue 2: transpose through LDS; slabs / output move as 16-byte row segments ---------------------------------------
    float* ct = (float*)smem;  // [PASS_ROWS][C_PITCH]
    constexpr int PASS_ROWS = BM < C_ROWS ? BM : C_ROWS;
    unsigned* flag = (unsigned*)(smem + PASS_ROWS * C_PITCH * 4);
    constexpr int NPASS = BM / PASS_ROWS, MIP = PASS_ROWS / 32;
    constexpr int UNITS = (PASS_ROWS * BN / 4 + 511) / 512;
    constexpr int NOUT = BM * BN;
    const int64_t ncol0 = (int64_t)nt * BN;
    float* slab = p.slabs + ((int64_t)bid * p.splitk) * NOUT;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, (short)0, p.splitk * NOUT * 4, 0x00020000);
    const float post = p.mx_post;
    auto finish = [&](f32x4 v, int m, int c4) { store_out4_any(p.epi, v * (f32x4){post, post, post, post}, m, ncol0 + c4); };
    f32x4 own[NPASS][UNITS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int mi = 0; mi < MIP; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    ct[r * C_PITCH + cg * 32 + col] = acc[ps * MIP + mi][e];
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = tid + 512 * i, r = u >> 5, c4 = (u & 31) * 4;
            const int m = m0 + ps * PASS_ROWS + r;
            own[ps][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (r < PASS_ROWS && m < p.M) {
                const f32x4 v = *(const f32x4*)(ct + r * C_PITCH + c4);
                own[ps][i] = v;
                if (p.splitk == 1) finish(v, m, c4);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                            (slice * NOUT + (ps * PASS_ROWS + r) * BN + c4) * 4, 0, 16);  // sc1
            }
        }
    }
    if (p.splitk == 1) return;
    __syncthreads();
    if (!splitk_arrive_is_last(p.counters + bid, p.splitk, flag)) return;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        f32x4 sum[UNITS];
#pragma unroll
        for (int i = 0; i < UNITS; ++i) sum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int sl = 0; sl < p.splitk; ++sl) {
            if (sl == slice) {  // own partial from registers; the fixed slice order keeps the sum deterministic
#pragma unroll
                for (int i = 0; i < UNITS; ++i) sum[i] += own[ps][i];
                continue;
            }
            u32x4 t[UNITS];
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                const int u = tid + 512 * i, r = u >> 5, c4 = (u & 31) * 4;
                t[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (sl * NOUT + (ps * PASS_ROWS + r) * BN + c4) * 4, 0, 16);
            }
#pragma unroll
            for (int i = 0; i < UNITS; ++i) sum[i] += __builtin_bit_cast(f32x4, t[i]);
        }
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = tid + 512 * i, r = u >> 5, c4 = (u & 31) * 4;
            const int m = m0 + ps * PASS_ROWS + r;
            if (r < PASS_ROWS && m < p.M) finish(sum[i], m, c4);
        }
    }
    if (tid == 0) splitk_reset(p.counters + bid);
}

typedef void (*mx_kernel_fn_t)(const GenericParams);
// ---------------------------------------------------------------------------------------------------------------------
// prefill kernel (M >= 512): 256 x 256 tiles, BOTH operands through LDS, 8 waves (2 x 4) with 128 x 64 wave tiles.
// The 8-wave kernel above fetches its weight fragments as 64 rows x 16 bytes per request and moves (128 + 128) rows x 256
// bytes per 128 x 128 x 256 step — 128 flop per operand byte, which at the block-scaled MFMA rate (8 kflop / clk / CU for
// fp8, 16 for fp4) asks more than the 64 B/clk a CU gets from L2.  Here a block owns 256 x 256 outputs (512 flop per byte):
//   * x AND w tiles go global -> LDS by LDS-DMA as 64-byte row segments (1 KiB = 16 rows per request), XOR-swizzled through
//     the source address so that the fragment reads are conflict-free (PMC: 1.6 % of LDS cycles conflict); the block scales
//     travel the same way (4 bytes per lane).  EVERY request of the K loop is an asm DMA: with compiler-tracked loads in the
//     loop hipcc waited for all older DMAs before their first use (210 -> 151 us at 8192^2, M = 2048 when the scale loads became
//     DMAs), and a single scratch reload in the loop costs a vmcnt(0), i.e. the whole pipeline (a 4-wave variant with 128 x 128
//     wave tiles spilled accumulators: 611 us) — hence per-lane base addresses + immediates + one opaque SGPR per stage;
//   * NST = 4 stages of 64 bytes per row (3 in flight); a stage is NSL * 4 "rows" of 2 MFMAs, the operands of row r + 1 are
//     read while the MFMAs of row r issue; one counted s_waitcnt + one s_barrier per stage, before its last row.
// K is not split.  Measured (profiles/r02/mx): 8192^2, M = 2048: fp8 154 us = 1.78 PFLOP/s (0.36 of the MX-fp8 peak), fp4
// 88 us = 3.14 PFLOP/s (0.31); the 128-row kernel above: 342 / 251 us.  PMC: the matrix pipe is busy 40 % of the time, waves
// wait 36 % (barrier + the DMA of HBM-cold panels, 2.4 us ahead) — not LDS, not the fabric (XCD-aware tile order: 1 %).
// ---------------------------------------------------------------------------------------------------------------------
// row r, logical 16-byte slot s of a tile with P bytes per row: rows 256 / P apart share an LDS bank period, so the slot is
// XOR-ed with (r / (256 / P)) — 16 consecutive rows then read 16 different bank groups
template <int P>
__device__ __forceinline__ int mxt_slot(int r, int s) { return r * P + ((s ^ ((r / (256 / P)) & (P / 16 - 1))) << 4); }

// AF / BF: element format of x / w (0 fp8 e4m3, 4 fp4 e2m1): fp8 x fp8, fp4 x fp4 and fp8 x fp4 (A8W4).  NST LDS stages.
// BLKX: activation block scales (channel_scale_mode 4), else the constant 127.
template <int AF, int BF, int NST, bool BLKX>
__global__ __launch_bounds__(512, 2) void gemm_mx_tile_kernel(const GenericParams p) {
    using namespace async;
    constexpr int BM = 256, BN = 256;
    constexpr int MI = 4, NI = 2;
    constexpr int KSTAGE = AF == 0 ? 64 : 128;                        // k per stage (64 bytes of every x row)
    constexpr int PA = 64, PB = BF == 0 ? KSTAGE : KSTAGE / 2;        // bytes per row and stage of the x / w tile (w: 64, or 32 for A8W4)
    constexpr int A_BYTES = BM * PA, TILES = A_BYTES + BN * PB;       // operand tiles per stage: 32 KiB (24 KiB)
    constexpr int SC_A = TILES, SC_B = TILES + 1024, STAGE = TILES + 2048;  // + 1 KiB activation scales + 1 KiB weight scales
    constexpr int NSL = KSTAGE / 64;                                  // 64-k slices per stage (1 or 2)
    constexpr int KBLK = KSTAGE / 32;                                 // scale blocks per stage (2 or 4)
    constexpr int FVA = AF == 0 ? 2 : 1, FVB = BF == 0 ? 2 : 1;       // 16-byte pieces per fragment
    constexpr int PIECES = TILES / 1024 / 8;                          // tile pieces per wave and stage (4 or 3)
    constexpr int R = PIECES + 1;                                     // + one scale piece: requests per wave and stage
    static_assert((NST - 1) * R <= 63 && NST >= 3 && NST % 2 == 0, "vmcnt is a 6-bit counter; stage parity must be static");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NST][STAGE], later the per-wave output tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int col = lane & 31, h = lane >> 5;
    const int mtiles = (p.M + BM - 1) / BM, ntiles = p.N / BN;
    // Block b runs on XCD b % 8 and every XCD has its own 4 MiB L2: give each XCD a compact sub-grid (ntiles / 8 weight panels
    // x all row panels) so that a panel stage is fetched into an L2 once and shared (tuning[3] & 16: plain M-fastest order)
    int mt = blockIdx.x % mtiles, nt = blockIdx.x / mtiles;
    if ((ntiles & 7) == 0 && !(p.flags & 16)) {
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = ntiles >> 3;
        mt = idx % mtiles;
        nt = xcd * per + (idx / mtiles) % per;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int nstages = p.K / KSTAGE;
    const int xrow_bytes = AF == 0 ? p.K : p.K / 2, wrow_bytes = BF == 0 ? p.K : p.K / 2;
    const int blocks_k = p.K / 32;
    constexpr bool blk_x = BLKX;
    const int m_pad = (p.M + 31) / 32 * 32;

    const srd_t rsX = make_srd(p.x, (uint32_t)((int64_t)(p.M - 1) * p.stride_xm + xrow_bytes));
    const srd_t rsW = make_srd(p.w, (uint32_t)((int64_t)(p.N - 1) * p.stride_wn + wrow_bytes));
    const srd_t rsSW = make_srd(p.scales, (uint32_t)((int64_t)(blocks_k - 1) * p.stride_meta_g + p.N));
    const srd_t rsSA = blk_x ? make_srd(p.sx_blocks, (uint32_t)((int64_t)(m_pad - 1) * p.stride_sx_blk_m + blocks_k)) : rsSW;

    // Per wave and stage: PIECES tile pieces (1 KiB = 16 rows x 64 bytes; waves 0..3 move x, waves 4..7 move w; lane i's 16 bytes
    // land at +16 i = row i / 4, physical slot i % 4 = logical slot ^ ((row >> 2) & 3)) and ONE 256-byte scale piece: waves 0..3
    // the activation-scale dwords of rows [64 wave, +64), waves 4..7 one 32-k block row of weight scales (fp8: rows repeat).
    uint32_t dvoff[PIECES], dsoff_step[PIECES];
    bool d_is_w[PIECES];  // wave-uniform per piece (A8W4: 24 pieces, the x / w border falls inside wave 5)
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const int piece = wave * PIECES + j;
        d_is_w[j] = piece * 1024 >= A_BYTES;
        const int tb = (d_is_w[j] ? piece * 1024 - A_BYTES : piece * 1024) + lane * 16;
        if (d_is_w[j]) {
            const int r = tb / PB, phys = (tb % PB) / 16;
            const int logical = phys ^ ((r / (256 / PB)) & (PB / 16 - 1));
            dvoff[j] = (uint32_t)((int64_t)(n0 + r) * p.stride_wn + logical * 16);
        } else {
            const int r = tb / PA, phys = (tb % PA) / 16;
            const int logical = phys ^ ((r / (256 / PA)) & (PA / 16 - 1));
            dvoff[j] = m0 + r < p.M ? (uint32_t)((int64_t)(m0 + r) * p.stride_xm + logical * 16) : 0x80000000u;
        }
        dsoff_step[j] = d_is_w[j] ? PB : PA;
    }
    const bool is_w = wave >= 4;  // scale pieces: waves 0..3 activation scales, 4..7 weight scales
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const uint32_t lds0 = lds_base + (uint32_t)(wave * PIECES) * 1024u;
    const int brow = (wave & 3) % KBLK;
    const srd_t rsS = is_w ? rsSW : rsSA;
    const uint32_t s_voff = is_w ? (uint32_t)((int64_t)brow * p.stride_meta_g + n0 + lane * 4)
                                 : (blk_x ? (uint32_t)((int64_t)(m0 + wave * 64 + lane) * p.stride_sx_blk_m) : 0u);
    const uint32_t s_lds = lds_base + (is_w ? (uint32_t)(SC_B + brow * 256) : (uint32_t)(SC_A + wave * 256));
    auto request = [&](int buf, int stage) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j)
            req_lds16(d_is_w[j] ? rsW : rsX, lds0 + (uint32_t)(buf * STAGE + j * 1024), dvoff[j],
                      (uint32_t)__builtin_amdgcn_readfirstlane(stage * (int)dsoff_step[j]));
        // activation scales: the ALIGNED dword that holds this stage's KBLK bytes (fp8: two stages share a dword);
        // weight scales: block row stage * KBLK + brow
        const int so = is_w ? (stage * KBLK) * (int)p.stride_meta_g : (blk_x ? ((stage * KBLK) & ~3) : 0);
        req_lds4(rsS, s_lds + (uint32_t)(buf * STAGE), s_voff, (uint32_t)__builtin_amdgcn_readfirstlane(so));
    };
    // Fragment addresses inside a stage: row block mi adds mi * 32 * PITCH — an immediate — to ONE per-lane base per (slice, piece)
    // (fp8: slots {h, 2 + h}; fp4: slot 2g + h; the swizzle (row >> 2) & 3 = (col >> 2) & 3 does not depend on the row block),
    // and the stage's LDS offset is added from an SGPR the optimiser cannot see through: folded into per-stage address
    // registers, the 4 x 24 addresses spilled, and a scratch reload inside the loop costs a vmcnt(0) — the whole DMA pipeline.
    int abase[NSL][FVA], bbase[NSL][FVB];
#pragma unroll
    for (int g = 0; g < NSL; ++g) {
#pragma unroll
        for (int v = 0; v < FVA; ++v) abase[g][v] = mxt_slot<PA>(wm * 128 + col, AF == 0 ? 4 * g + 2 * v + h : 2 * g + h);
#pragma unroll
        for (int v = 0; v < FVB; ++v) bbase[g][v] = A_BYTES + mxt_slot<PB>(wn * 64 + col, BF == 0 ? 4 * g + 2 * v + h : 2 * g + h);
    }
    const int sa_base = SC_A + (wm * 128 + col) * 4 + (KBLK == 2 ? h : 0);
    const int sb_base = SC_B + h * 256 + wn * 64 + col;
    const uint32_t sh_h = 8u * (uint32_t)h;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // operands: the 4 weight fragments of a 64-k slice (+ this lane's weight block scales), one activation fragment (+ scale)
    typedef typename std::conditional<AF == 0, v8i, u32x4>::type afrag_t;  // fp4 fragments are 4 registers
    typedef typename std::conditional<BF == 0, v8i, u32x4>::type bfrag_t;
    struct BSet { bfrag_t b[NI]; uint32_t s[NI]; };
    struct AFr { afrag_t a; uint32_t s; };
    auto mk8 = [](u32x4 v0, u32x4 v1) -> v8i {
        return (v8i){(int)v0[0], (int)v0[1], (int)v0[2], (int)v0[3], (int)v1[0], (int)v1[1], (int)v1[2], (int)v1[3]};
    };
    auto wide4 = [](u32x4 f) -> v8i { return (v8i){(int)f[0], (int)f[1], (int)f[2], (int)f[3], 0, 0, 0, 0}; };
    auto stage_ptr = [&](int buf) -> const unsigned char* {
        int off = buf * STAGE;
        asm volatile("" : "+s"(off));  // opaque: one s-register per stage, added at every read
        return smem + off;
    };
    auto load_b = [&](BSet& bs, const unsigned char* sb, int g) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const u32x4 v0 = *(const u32x4*)(sb + bbase[g][0] + ni * 32 * PB);
            if constexpr (BF == 0) bs.b[ni] = mk8(v0, *(const u32x4*)(sb + bbase[g][FVB - 1] + ni * 32 * PB));
            else bs.b[ni] = v0;
            bs.s[ni] = sb[sb_base + 2 * g * 256 + ni * 32];
        }
    };
    auto load_a = [&](AFr& af, const unsigned char* sb, int g, int mi, int par) {
        const u32x4 v0 = *(const u32x4*)(sb + abase[g][0] + mi * 32 * PA);
        if constexpr (AF == 0) af.a = mk8(v0, *(const u32x4*)(sb + abase[g][FVA - 1] + mi * 32 * PA));
        else af.a = v0;
        if constexpr (!blk_x) af.s = 127u;
        else if constexpr (KBLK == 2) af.s = sb[sa_base + mi * 128 + par * 2];                               // block h of this stage
        else af.s = *(const uint32_t*)(sb + sa_base + mi * 128) >> (sh_h + 16u * (uint32_t)g);              // block 2g + h
    };
    auto mma_row = [&](const AFr& af, const BSet& bs, int mi) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
        {
            v8i av, bv;
            if constexpr (AF == 0) av = af.a; else av = wide4(af.a);
            if constexpr (BF == 0) bv = bs.b[ni]; else bv = wide4(bs.b[ni]);
            acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[mi][ni], AF, BF, 0, (int)af.s, 0, (int)bs.s[ni]);
        }
    };

#pragma unroll
    for (int s0 = 0; s0 < NST - 1; ++s0) request(s0, s0 < nstages ? s0 : nstages - 1);
    wait_vm<(NST - 2) * R>();
    __builtin_amdgcn_s_barrier();
    BSet bs[2];
    AFr af[2];
    load_b(bs[0], stage_ptr(0), 0);
    load_a(af[0], stage_ptr(0), 0, 0, 0);

    // Stage st (J = st % NST): request stage st + NST - 1 into the buffer stage st - 1 used (every wave passed the barrier after
    // its last read of it).  The stage is NSL * MI "rows" of 4 MFMAs (one activation fragment against the slice's 4 weight
    // fragments); the operands of row r + 1 are read while the MFMAs of row r issue.  Before the LAST row: "stage st + 1 has
    // landed" (counted wait + barrier), then the first operands of stage st + 1, then the last row's MFMAs.
    auto do_stage = [&](auto Jc, int st) {
        constexpr int J = decltype(Jc)::value;  // NST is even: J & 1 == st & 1
        const int nx = st + NST - 1 < nstages ? st + NST - 1 : nstages - 1;  // past the end: repeat the last stage (never read)
        request((J + NST - 1) % NST, nx);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sb_cur = stage_ptr(J);
        const unsigned char* sb_nxt = stage_ptr((J + 1) % NST);
        constexpr int ROWS = NSL * MI;
        // (ROWS is even and MI is even: the parities of the A ring and of the B sets are the same in every stage)
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int g = r / MI, mi = r % MI;
            if (r + 1 < ROWS) {
                const int g1 = (r + 1) / MI, mi1 = (r + 1) % MI;
                if (mi1 == 0) load_b(bs[g1 & 1], sb_cur, g1);  // (NSL == 2: slices alternate between the two sets)
                load_a(af[(r + 1) & 1], sb_cur, g1, mi1, J & 1);
            } else {
                wait_vm<(NST - 2) * R>();            // everything but the newest NST - 2 stages: stage st + 1 has landed
                __builtin_amdgcn_s_waitcnt(0xC07F);  // this wave's LDS reads of stage st are complete
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                load_b(bs[NSL == 1 ? ((J + 1) & 1) : 0], sb_nxt, 0);
                load_a(af[0], sb_nxt, 0, 0, (J + 1) & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_row(af[r & 1], bs[NSL == 1 ? (J & 1) : (g & 1)], mi);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto chain = [&](auto self, auto Jc, int s0) -> void {
        constexpr int J = decltype(Jc)::value;
        do_stage(Jc, s0 + J);
        if constexpr (J + 1 < NST) {
            if (s0 + J + 1 < nstages) self(self, std::integral_constant<int, J + 1>{}, s0);
        }
    };
    for (int s0 = 0; s0 < nstages; s0 += NST) chain(chain, std::integral_constant<int, 0>{}, s0);
    wait_vm<0>();

    // ---- epilogue: every wave transposes its 32 x 32 blocks through a private LDS tile, rows leave as 8-byte (4-column) stores
    __syncthreads();
    float* ct = (float*)smem + wave * (32 * 36);
    const float post = p.mx_post;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int e = 0; e < 16; ++e) ct[((e & 3) + 8 * (e >> 2) + 4 * h) * 36 + col] = acc[mi][ni][e];
            __builtin_amdgcn_s_waitcnt(0xC07F);
            const int r = lane >> 1, c0 = (lane & 1) * 16;
            const int m = m0 + wm * 128 + mi * 32 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *(const f32x4*)(ct + r * 36 + c0 + 4 * q);
                if (m < p.M) store_out4_any(p.epi, v * (f32x4){post, post, post, post}, m, (int64_t)n0 + wn * 64 + ni * 32 + c0 + 4 * q);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
        }
}

// M >= 512 (or tuning[0] == 3), fp8 x fp8 or fp4 x fp4, N % 256 == 0
bool plan_gemm_mx_tile(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp) {
    if (g.mx_scale_e4m3 || g.group_size != 32) return false;
    const bool f8 = g.mx_x == MX_FP8 && g.mx_w == MX_FP8, f4 = g.mx_x == MX_FP4 && g.mx_w == MX_FP4;
    const bool f84 = g.mx_x == MX_FP8 && g.mx_w == MX_FP4;
    if (!f8 && !f4 && !f84) return false;
    // worth it from ~100 tiles of 256 x 256 (sweep: M = 512: 8192^2 = 64 tiles 161 vs 110 us for the 128-row kernel, 14336 x 4096 =
    // 112 tiles 90 vs 90.5; M = 2048, 8192^2 = 256 tiles 154 vs 342)
    const int64_t tiles256 = (int64_t)(a.N / 256) * ((a.M + 255) / 256);
    if (a.tuning[0] != 3 && (a.tuning[0] != 0 || tiles256 < 96)) return false;
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.stride_on != 1 || a.N % 256 != 0 || a.K % (f4 ? 128 : 64) != 0) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0 || a.stride_xm % 16 != 0 || a.stride_wn % 16 != 0) return false;
    if ((int64_t)a.M * a.stride_xm >= (1ll << 31) || (int64_t)a.N * a.stride_wn >= (1ll << 31)) return false;
    if ((int64_t)(a.K / 32) * a.stride_meta_g + (int64_t)a.N * a.stride_meta_n >= (1ll << 31)) return false;
    if (!(a.output_dtype == GEMLITE_DT_FP32 || a.output_dtype == GEMLITE_DT_FP16 || a.output_dtype == GEMLITE_DT_BF16)) return false;
    const int oal = a.output_dtype == GEMLITE_DT_FP32 ? 16 : 8;
    if (((uintptr_t)a.out % oal) != 0 || (a.stride_om * (oal / 4)) % oal != 0) return false;
    if (a.channel_scale_mode == 4) {
        if (!g.sx_blocks || g.stride_sx_blk_m % 4 != 0 || ((uintptr_t)g.sx_blocks % 4) != 0) return false;
        if ((int64_t)((a.M + 31) / 32 * 32) * g.stride_sx_blk_m >= (1ll << 31)) return false;
    } else if (a.channel_scale_mode != 2 && a.channel_scale_mode != 0) {
        return false;
    }
    const int64_t tiles = (int64_t)(a.N / 256) * ((a.M + 255) / 256);
    if (tiles > 0x7FFFFFFF) return false;
    constexpr int nst = 4;  // LDS stages of 34 KiB (tiles + scales): 136 KiB
    if (a.stride_meta_n != 1 || ((uintptr_t)a.scales % 4) != 0 || a.stride_meta_g % 4 != 0) return false;  // 4-byte scale pieces
    const bool bx = a.channel_scale_mode == 4;
    mx_kernel_fn_t f = f8 ? (bx ? gemm_mx_tile_kernel<0, 0, nst, true> : gemm_mx_tile_kernel<0, 0, nst, false>)
                      : (f4 ? (bx ? gemm_mx_tile_kernel<4, 4, nst, true> : gemm_mx_tile_kernel<4, 4, nst, false>)
                            : (bx ? gemm_mx_tile_kernel<0, 4, nst, true> : gemm_mx_tile_kernel<0, 4, nst, false>));  // (6 stages of 26 KiB for A8W4: 142 vs 139 us — depth is not the limit)
    g.splitk = 1;
    g.flags = a.tuning[3];  // & 16: plain M-fastest tile order (A/B runs)
    lp.fn = (const void*)f;
    lp.name = f8 ? "gemm_mx_a8w8_tile_kernel<256x256>" : (f4 ? "gemm_mx_a4w4_tile_kernel<256x256>" : "gemm_mx_a8w4_tile_kernel<256x256>");
    lp.grid = dim3((unsigned)tiles, 1, 1);
    lp.block = dim3(512, 1, 1);
    lp.lds_bytes = (size_t)nst * ((256 + 256) * 64 + 2048);
    lp.ws_bytes = 0;
    lp.slab_bytes = 0;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: 64 x 64 tiles, K NOT split ("gemm_mx_sq_kernel") — the shape of gemm_a8w8_sq_kernel (gemm_a8w8.hip) for the
// block-scaled pairs.  Between 65 and ~256 rows the 128-column kernel above needs K slices to fill 256 CUs (4096^2, M = 256:
// 64 tiles x 4 slices, 22-26 us: prologue + a combine through memory around a 2-us loop); nothing is dequantised here either, so
// a small tile costs operand traffic only.  8 waves: wave (rb, cb, kh) owns the 32 x 32 block (rb, cb) and half of every
// 256-k step (two v_mfma_scale_f32_32x32x64_f8f6f4); x and w rows travel as 1-KiB LDS-DMA pieces into NST stages of
// [64 x rows | 64 w rows] x (256 B fp8 | 128 B fp4), 16-byte slots swizzled like the 256 x 256 kernel's (mxt_slot); the step's
// block scales ride as four 256-byte pieces (activation scale dwords [K half][row], weight scale bytes [block row][column]);
// one counted wait + one barrier per step.  The blocks sharing a weight column tile sit on one XCD.
// ---------------------------------------------------------------------------------------------------------------------
template <int AF, int BF, int NST, bool BLKX>
__global__ __launch_bounds__(512, (NST <= 2 ? 2 : 1)) void gemm_mx_sq_kernel(const GenericParams p) {
    using namespace async;
    constexpr int BM = 64, BN = 64, KSTEP = 256;
    constexpr int PA = AF == 0 ? KSTEP : KSTEP / 2, PB = BF == 0 ? KSTEP : KSTEP / 2;  // bytes per row and step
    constexpr int A_BYTES = BM * PA, TILES = A_BYTES + BN * PB;
    constexpr int SC_A = TILES, SC_B = TILES + 512, STAGE = TILES + 1024;
    constexpr int PX = A_BYTES / 1024 / 8, PW = BN * PB / 1024 / 8;  // tile pieces per wave and stage (2 or 1 each)
    constexpr int R = PX + PW + 1;                                     // + one scale piece
    constexpr int FVA = AF == 0 ? 2 : 1, FVB = BF == 0 ? 2 : 1;       // 16-byte pieces per fragment
    static_assert(NST >= 2 && (NST - 1) * R <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NST][STAGE], later the epilogue tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = (wave >> 1) & 1, cb = wave & 1, kh = wave >> 2;
    const int col = lane & 31, h = lane >> 5;
    const int mtiles = (p.M + BM - 1) / BM, ntiles = p.N / BN;
    int mt, nt;
    {
        const int lin = blockIdx.x;
        if ((ntiles & 7) == 0) {  // the row tiles of one weight column tile on one XCD, back to back in its dispatch order
            const int xcd = lin & 7, idx = lin >> 3;
            mt = idx % mtiles;
            nt = (idx / mtiles) * 8 + xcd;
        } else {
            mt = lin % mtiles;
            nt = lin / mtiles;
        }
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int nsteps = p.K / KSTEP;
    const int xrow_bytes = AF == 0 ? p.K : p.K / 2, wrow_bytes = BF == 0 ? p.K : p.K / 2;
    const int blocks_k = p.K / 32;
    const int m_pad = (p.M + 31) / 32 * 32;

    const srd_t rsX = make_srd(p.x, (uint32_t)((int64_t)(p.M - 1) * p.stride_xm + xrow_bytes));
    const srd_t rsW = make_srd(p.w, (uint32_t)((int64_t)(p.N - 1) * p.stride_wn + wrow_bytes));
    const srd_t rsSW = make_srd(p.scales, (uint32_t)((int64_t)(blocks_k - 1) * p.stride_meta_g + p.N));
    const srd_t rsSA = BLKX ? make_srd(p.sx_blocks, (uint32_t)((int64_t)(m_pad - 1) * p.stride_sx_blk_m + blocks_k)) : rsSW;

    uint32_t xvoff[PX], wvoff[PW];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int byte = (wave * PX + j) * 1024 + lane * 16;
        const int r = byte / PA, phys = (byte % PA) / 16;
        const int logical = phys ^ ((r / (256 / PA)) & (PA / 16 - 1));
        xvoff[j] = m0 + r < p.M ? (uint32_t)((int64_t)(m0 + r) * p.stride_xm + logical * 16) : 0x80000000u;  // rows >= M: zeros
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int byte = (wave * PW + j) * 1024 + lane * 16;
        const int r = byte / PB, phys = (byte % PB) / 16;
        const int logical = phys ^ ((r / (256 / PB)) & (PB / 16 - 1));
        wvoff[j] = (uint32_t)((int64_t)(n0 + r) * p.stride_wn + logical * 16);
    }
    // scale piece sp = wave & 3 (waves 4 .. 7 repeat the requests of waves 0 .. 3: same bytes to the same place — every wave then counts
    // the same number of requests per stage).  sp 0 / 1: dword sp of the step's 8 activation-scale bytes of rows [m0, +64);
    // sp 2 / 3: weight-scale block rows [4 (sp - 2), +4) x 64 columns
    const int sp = wave & 3;
    const bool s_is_w = sp >= 2;
    const srd_t rsS = s_is_w ? rsSW : rsSA;
    const uint32_t s_voff = s_is_w ? (uint32_t)((int64_t)((sp - 2) * 4 + (lane >> 4)) * p.stride_meta_g + n0 + (lane & 15) * 4)
                                   : (BLKX ? (uint32_t)((int64_t)(m0 + lane) * p.stride_sx_blk_m + sp * 4) : 0u);
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const uint32_t ldsx = lds_base + (uint32_t)(wave * PX) * 1024u;
    const uint32_t ldsw = lds_base + (uint32_t)A_BYTES + (uint32_t)(wave * PW) * 1024u;
    const uint32_t ldss = lds_base + (uint32_t)(s_is_w ? SC_B + (sp - 2) * 256 : SC_A + sp * 256);
    // K rotation (round 6, see gemm_a8w8_sq_kernel): the row tiles that share a weight column tile (same XCD) start their K loops mtiles-ths apart, so
    // that each sibling CU pulls a different part of the HBM-cold weights and finds the rest in L2 (MXFP8 4096^2 M = 256 16.1 -> 12.3 us, MXFP4 9.5 ..
    // 10.6 -> 9.0, 4096 x 8192 27.8 -> 25.1: profiles/r06/probe_k_rotation_mx_and_int8_64x64.log).  The planner sets bit 30 of flags where the weight
    // tiles an XCD works on at a time fit its L2 (k_rotation_pays(), gemm_a8w8.hip); tuning[3] & 4194304 = never (A/B runs)
    KOrder kord;
    kord.init(mt, mtiles, nsteps, p.flags);
    auto request = [&](int stage, int step) __attribute__((always_inline)) {
        step = kord.at(step);
#pragma unroll
        for (int j = 0; j < PX; ++j) req_lds16(rsX, ldsx + (uint32_t)(stage * STAGE + j * 1024), xvoff[j], (uint32_t)__builtin_amdgcn_readfirstlane(step * PA));
#pragma unroll
        for (int j = 0; j < PW; ++j) req_lds16(rsW, ldsw + (uint32_t)(stage * STAGE + j * 1024), wvoff[j], (uint32_t)__builtin_amdgcn_readfirstlane(step * PB));
        const int so = s_is_w ? step * 8 * (int)p.stride_meta_g : (BLKX ? step * 8 : 0);
        req_lds4(rsS, ldss + (uint32_t)(stage * STAGE), s_voff, (uint32_t)__builtin_amdgcn_readfirstlane(so));
    };
    // fragments of slice g (64 k of this wave's K half): fp8 slots {4 g + h, 4 g + 2 + h} of the half's 8, fp4 slot 2 g + h of its 4
    const int ra = rb * 32 + col, rw = cb * 32 + col;
    int fa[2][FVA], fb[2][FVB];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int v = 0; v < FVA; ++v) fa[g][v] = mxt_slot<PA>(ra, AF == 0 ? kh * 8 + 4 * g + 2 * v + h : kh * 4 + 2 * g + h);
#pragma unroll
        for (int v = 0; v < FVB; ++v) fb[g][v] = A_BYTES + mxt_slot<PB>(rw, BF == 0 ? kh * 8 + 4 * g + 2 * v + h : kh * 4 + 2 * g + h);
    }
    const int sa_off = SC_A + kh * 256 + ra * 4;
    const int sb_off = SC_B + (kh * 4 + h) * 64 + rw;
    const uint32_t sh_h = 8u * (uint32_t)h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    auto mk8 = [](u32x4 v0, u32x4 v1) -> v8i {
        return (v8i){(int)v0[0], (int)v0[1], (int)v0[2], (int)v0[3], (int)v1[0], (int)v1[1], (int)v1[2], (int)v1[3]};
    };
    auto wide4 = [](u32x4 f) -> v8i { return (v8i){(int)f[0], (int)f[1], (int)f[2], (int)f[3], 0, 0, 0, 0}; };

#pragma unroll
    for (int st = 0; st < NST - 1; ++st) request(st, st < nsteps ? st : nsteps - 1);
    auto do_step = [&](auto Jc, int step) __attribute__((always_inline)) {
        constexpr int stage = decltype(Jc)::value, stage_fill = (stage + NST - 1) % NST;
        wait_vm<(NST - 2) * R>();      // this step's pieces have landed (the later stages' stay in flight)
        __builtin_amdgcn_s_barrier();  // ... everybody's have, and everybody is done reading the stage refilled next
        asm volatile("" ::: "memory");
        request(stage_fill, step + NST - 1 < nsteps ? step + NST - 1 : nsteps - 1);  // past the end: repeat the last step (never consumed)
        const unsigned char* sb = smem + stage * STAGE;
        v8i av[2], bv[2];
        uint32_t sa[2], sw[2];
        const uint32_t sa_dw = BLKX ? *(const uint32_t*)(sb + sa_off) : 0u;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const u32x4 a0 = *(const u32x4*)(sb + fa[g][0]);
            if constexpr (AF == 0) av[g] = mk8(a0, *(const u32x4*)(sb + fa[g][FVA - 1])); else av[g] = wide4(a0);
            const u32x4 b0 = *(const u32x4*)(sb + fb[g][0]);
            if constexpr (BF == 0) bv[g] = mk8(b0, *(const u32x4*)(sb + fb[g][FVB - 1])); else bv[g] = wide4(b0);
            sa[g] = BLKX ? (sa_dw >> (sh_h + 16u * (uint32_t)g)) : 127u;  // block 2 g + h of the half
            sw[g] = sb[sb_off + 2 * g * 64];
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av[g], bv[g], acc, AF, BF, 0, (int)sa[g], 0, (int)sw[g]);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of the stage are complete before it reaches the next barrier
    };
    auto chain = [&](auto self, auto Jc, int s0) -> void {
        constexpr int J = decltype(Jc)::value;
        do_step(std::integral_constant<int, J>{}, s0 + J);
        if constexpr (J + 1 < NST) {
            if (s0 + J + 1 < nsteps) self(self, std::integral_constant<int, J + 1>{}, s0);
        }
    };
    for (int s0 = 0; s0 < nsteps; s0 += NST) chain(chain, std::integral_constant<int, 0>{}, s0);
    wait_vm<0>();
    __syncthreads();

    // ---- epilogue: add the two K halves, transpose through LDS, 4 outputs per store
    constexpr int C_PITCH = BN + 4;
    // (round 6: both K halves drop their partial tile into LDS in output order, ONE barrier, every thread adds the two while it reads its row
    //  segment — see gemm_a8w8_sq_kernel; same addition, same order: bit-identical to the three-barrier form of rounds 4-5)
    float* ct = (float*)smem;  // [2 K halves][64][C_PITCH]
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        ct[(kh * 64 + r) * C_PITCH + cb * 32 + col] = acc[e];
    }
    __syncthreads();
    const float post = p.mx_post;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + 512 * i, r = u >> 4, c4 = (u & 15) * 4;
        const int m = m0 + r;
        if (m < p.M) {
            const f32x4 v = *(const f32x4*)(ct + r * C_PITCH + c4) + *(const f32x4*)(ct + (64 + r) * C_PITCH + c4);
            store_out4_any(p.epi, v * (f32x4){post, post, post, post}, m, (int64_t)n0 + c4);
        }
    }
}

// 65 .. ~256 rows whose 64 x 64 tiles fill the chip about once (the caller's rule), fp8 x fp8 / fp4 x fp4 / fp8 x fp4; tuning[0] = 6 forces it
int k_order_flags(const gemlite_hip_forward_args& a, int64_t tile_bytes);  // gemm_a8w8.hip

bool plan_gemm_mx_sq(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp) {
    if (g.mx_scale_e4m3 || g.group_size != 32 || a.M < 1) return false;
    const bool f8 = g.mx_x == MX_FP8 && g.mx_w == MX_FP8, f4 = g.mx_x == MX_FP4 && g.mx_w == MX_FP4;
    const bool f84 = g.mx_x == MX_FP8 && g.mx_w == MX_FP4;
    if (!f8 && !f4 && !f84) return false;
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.stride_on != 1 || a.N % 64 != 0 || a.K % 256 != 0) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0 || a.stride_xm % 16 != 0 || a.stride_wn % 16 != 0) return false;
    if ((int64_t)a.M * a.stride_xm >= (1ll << 31) || (int64_t)a.N * a.stride_wn >= (1ll << 31)) return false;
    if ((int64_t)(a.K / 32) * a.stride_meta_g + (int64_t)a.N * a.stride_meta_n >= (1ll << 31)) return false;
    if (!(a.output_dtype == GEMLITE_DT_FP32 || a.output_dtype == GEMLITE_DT_FP16 || a.output_dtype == GEMLITE_DT_BF16)) return false;
    const int oal = a.output_dtype == GEMLITE_DT_FP32 ? 16 : 8;
    if (((uintptr_t)a.out % oal) != 0 || (a.stride_om * (oal / 4)) % oal != 0) return false;
    if (a.channel_scale_mode == 4) {
        if (!g.sx_blocks || g.stride_sx_blk_m % 4 != 0 || ((uintptr_t)g.sx_blocks % 4) != 0) return false;
        if ((int64_t)((a.M + 31) / 32 * 32) * g.stride_sx_blk_m >= (1ll << 31)) return false;
    } else if (a.channel_scale_mode != 2 && a.channel_scale_mode != 0) {
        return false;
    }
    if (a.stride_meta_n != 1 || ((uintptr_t)a.scales % 4) != 0 || a.stride_meta_g % 4 != 0) return false;  // 4-byte scale pieces
    const int64_t tiles = (int64_t)(a.N / 64) * ((a.M + 63) / 64);
    if (tiles > 0x7FFFFFFF) return false;
    // one round of tiles: 4 stages in flight per block; more: 2 stages, two co-resident blocks per CU (the A8W8 kernel's measurements);
    // tuning[2] = 2 / 3 / 4 picks the depth under tuning[0] = 6
    const int nst = (a.tuning[0] == 6 && (a.tuning[2] == 2 || a.tuning[2] == 3 || a.tuning[2] == 4)) ? a.tuning[2] : (tiles <= 256 ? 4 : 2);
    const bool bx = a.channel_scale_mode == 4;
    auto pick = [&](auto af, auto bf) -> mx_kernel_fn_t {
        constexpr int A = decltype(af)::value, B = decltype(bf)::value;
        if (bx) return nst == 2 ? gemm_mx_sq_kernel<A, B, 2, true> : (nst == 3 ? gemm_mx_sq_kernel<A, B, 3, true> : gemm_mx_sq_kernel<A, B, 4, true>);
        return nst == 2 ? gemm_mx_sq_kernel<A, B, 2, false> : (nst == 3 ? gemm_mx_sq_kernel<A, B, 3, false> : gemm_mx_sq_kernel<A, B, 4, false>);
    };
    typedef std::integral_constant<int, 0> F8;
    typedef std::integral_constant<int, 4> F4;
    mx_kernel_fn_t f = f8 ? pick(F8{}, F8{}) : (f4 ? pick(F4{}, F4{}) : pick(F8{}, F4{}));
    g.splitk = 1;
    const int64_t wtile_bytes = (int64_t)64 * a.K / (g.mx_w == MX_FP4 ? 2 : 1);  // the weight bytes of a 64-column tile (fp8: K per column, fp4: K / 2)
    g.flags = (a.tuning[3] & ~((1 << 30) | 0x0F000000)) | k_order_flags(a, wtile_bytes);
    lp.fn = (const void*)f;
    lp.name = f8 ? "gemm_mx_a8w8_sq_kernel<64x64>" : (f4 ? "gemm_mx_a4w4_sq_kernel<64x64>" : "gemm_mx_a8w4_sq_kernel<64x64>");
    lp.grid = dim3((unsigned)tiles, 1, 1);
    lp.block = dim3(512, 1, 1);
    const size_t stage = (size_t)64 * ((f4 ? 128 : 256) + (f8 ? 256 : 128)) + 1024;
    lp.lds_bytes = (size_t)nst * stage;
    if (lp.lds_bytes < 2 * 64 * 68 * 4) lp.lds_bytes = 2 * 64 * 68 * 4;  // (the two K halves of the epilogue)
    lp.ws_bytes = 0;
    lp.slab_bytes = 0;
    return true;
}

typedef void (*mx_kernel_fn)(const GenericParams);
template <int AF, int BF>
static const void* mx_pick(int mi) {
    mx_kernel_fn f = nullptr;
    switch (mi) {
        case 4: f = gemm_mx_mma_kernel<AF, BF, 4, 4, 2>; break;
        case 2: f = gemm_mx_mma_kernel<AF, BF, 2, 6, 3>; break;
        case 1: f = gemm_mx_mma_kernel<AF, BF, 1, 8, 4>; break;
        default: break;
    }
    return (const void*)f;
}

// fp8 / fp4 activations x fp8 / fp4 weights, e8m0 scales per 32 k on the weights and (channel_scale_mode 4) on the activations,
// any M >= 1.  tuning[1] = K slices, tuning[2] = tile rows / 32 (1 / 2 / 4).
bool plan_gemm_mx_mma(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp) {
    if (g.mx_scale_e4m3 || g.group_size != 32) return false;
    if (!(g.mx_x == MX_FP8 || g.mx_x == MX_FP4) || !(g.mx_w == MX_FP8 || g.mx_w == MX_FP4)) return false;
    if (g.mx_x == MX_FP4 && g.mx_w == MX_FP8) return false;  // not a combination any processor produces
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.stride_on != 1 || a.N % 128 != 0) return false;
    const int kstep = g.mx_x == MX_FP8 ? 256 : 512;
    if (a.K % kstep != 0) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0 || a.stride_xm % 16 != 0 || a.stride_wn % 16 != 0) return false;
    if ((int64_t)a.M * a.stride_xm >= (1ll << 31) || (int64_t)a.N * a.stride_wn >= (1ll << 31)) return false;
    if ((int64_t)(a.K / 32) * a.stride_meta_g + (int64_t)a.N * a.stride_meta_n >= (1ll << 31)) return false;
    if (!(a.output_dtype == GEMLITE_DT_FP32 || a.output_dtype == GEMLITE_DT_FP16 || a.output_dtype == GEMLITE_DT_BF16)) return false;
    if (a.channel_scale_mode == 4) {
        if (!g.sx_blocks || g.stride_sx_blk_m % 4 != 0 || ((uintptr_t)g.sx_blocks % 4) != 0) return false;
        if ((int64_t)((a.M + 31) / 32 * 32) * g.stride_sx_blk_m >= (1ll << 31)) return false;
    } else if (a.channel_scale_mode != 2 && a.channel_scale_mode != 0) {
        return false;
    }
    const int units = (int)(a.K / kstep);
    const int cap = a.M > 64 ? 4 : (a.M > 32 ? 2 : 1);
    // Tile rows and K slices (sweep: scripts/probe_mx.py, profiles/r02/mx/probe_mx_tilings.jsonl — 4 shapes x 6 batch sizes x 18
    // candidates, HBM-cold): nothing is dequantised here, so a tall tile only saves weight re-reads, and slices are cheap as long
    // as each keeps >= 1024 k.  Take the TALLEST tile (<= the rows M fills) that reaches >= 224 blocks with the slices it may
    // use, then the fewest slices that get there (mean regret against the measured best 3 %; the rule this replaces, tallest
    // tile with >= 112 tiles: 12 % incl. long-K shapes such as 4096 x 14336, M = 128: 63 -> 43 us).
    const int sk_max = units * kstep / 1024 < 1 ? 1 : (units * kstep / 1024 > 8 ? 8 : units * kstep / 1024);
    int mi = 1, splitk = 0;
    for (int c = cap; c >= 1; c >>= 1) {
        const int64_t t = (int64_t)(a.N / 128) * ((a.M + 32 * c - 1) / (32 * c));
        if (t * sk_max >= 224 || c == 1) {
            mi = c;
            for (int sk = 1; sk <= sk_max && sk <= units; ++sk) {
                splitk = sk;
                if (t * sk >= 224) break;
            }
            break;
        }
    }
    if (a.tuning[2] == 1 || a.tuning[2] == 2 || a.tuning[2] == 4) mi = a.tuning[2];
    const int bm = 32 * mi;
    const int64_t tiles = (int64_t)(a.N / 128) * ((a.M + bm - 1) / bm);
    if (a.tuning[1] > 0) {
        if (a.tuning[1] > units) return false;
        splitk = a.tuning[1];
    } else if (a.tuning[2] != 0 || !splitk) {  // forced tile height: the fewest slices that fill the chip
        splitk = 1;
        for (int sk = 1; sk <= sk_max && sk <= units; ++sk) {
            splitk = sk;
            if (tiles * sk >= 224) break;
        }
    }
    // 128-row tiles, late round 6 (profiles/r06/scan_mx_*.log): two blocks share a CU, and blocks past one per CU cost a second round — 5120 x 13824 M = 512,
    // 160 tiles: one slice 131 us, two (320 blocks) 142, three (480) 104.  The slice count that minimises rounds x slice length + 3.3 us per slice
    // (slab traffic and the combine; 9.5 us per block and 1024 k) — equal to the rule above wherever that one stays within one round.
    if (mi == 4 && a.tuning[1] == 0 && !(a.tuning[3] & 16384)) {
        const int64_t cus = resident_block_limit();
        double best = 0;
        for (int sk = 1; sk <= sk_max && sk <= units; ++sk) {
            const double est = (double)((tiles * sk + cus - 1) / cus) * 9.5 * (double)a.K / 1024.0 / sk + 3.3 * sk;
            if (sk == 1 || est < best) { best = est; splitk = sk; }
        }
    }
    if (splitk > 1 && tiles > MAX_SPLITK_COUNTERS) return false;
    if ((uint64_t)splitk * bm * 128 * 4 >= (1ull << 31)) return false;
    const void* fn = g.mx_x == MX_FP8 ? (g.mx_w == MX_FP8 ? mx_pick<0, 0>(mi) : mx_pick<0, 4>(mi)) : mx_pick<4, 4>(mi);
    if (!fn) return false;
    g.splitk = splitk;
    g.flags = a.tuning[3];
    lp.fn = fn;
    static const char* names[3][3] = {
        {"gemm_mx_a8w8_kernel<32x128>", "gemm_mx_a8w8_kernel<64x128>", "gemm_mx_a8w8_kernel<128x128>"},
        {"gemm_mx_a8w4_kernel<32x128>", "gemm_mx_a8w4_kernel<64x128>", "gemm_mx_a8w4_kernel<128x128>"},
        {"gemm_mx_a4w4_kernel<32x128>", "gemm_mx_a4w4_kernel<64x128>", "gemm_mx_a4w4_kernel<128x128>"}};
    lp.name = names[g.mx_x == MX_FP4 ? 2 : (g.mx_w == MX_FP4 ? 1 : 0)][mi == 4 ? 2 : (mi == 2 ? 1 : 0)];
    lp.grid = dim3((unsigned)tiles, splitk, 1);
    lp.block = dim3(512, 1, 1);
    const size_t stages = (size_t)(mi == 4 ? 2 : (mi == 2 ? 3 : 4)) * bm * 256, xch = (size_t)4 * mi * 64 * 64;
    const size_t c_b = (size_t)(bm < 128 ? bm : 128) * 132 * 4 + 16;
    lp.lds_bytes = stages > xch ? stages : xch;
    if (lp.lds_bytes < c_b) lp.lds_bytes = c_b;
    lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * bm * 128 * 4 : 0;
    lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

}  // namespace gl
