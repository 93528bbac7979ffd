// generic.hip — coverage kernels: every (W_group_mode, channel_scale_mode, dtype, stride) combination of
// the reference's five kernels that has no specialised CDNA4 kernel yet lands here.  Still native HIP on
// the GPU (there is no CPU fallback anywhere in the product), just not tuned.
//
//   generic_matmul_kernel : lane = output column; walks K; any packed width / unpacked dtype / strides.
//   kmajor_matmul_kernel  : unpacked K-contiguous weights (W_q = W.t() view, strides (1, K),
//                           core.py:369-381): one wave per output column, lanes along K with 16-byte
//                           loads — the streaming GEMV for A8W8 / FP8xFP8 / FP16xFP16 at small M.
//
// Numerics follow triton_kernels/utils.py:57-89 + gemm_kernels.py:347-413 with fp32 (int32 for int8 x int8)
// accumulation.  One deliberate fidelity point: for fp8 activations the reference casts the dequantised
// weight to the input dtype before the dot (`b.to(input_dtype)`, gemm_kernels.py:384) — so do we.
#include "gl_common.h"

namespace gl {


__device__ __forceinline__ uint32_t load_word(const void* w, int64_t idx, int pack_bits) {
    switch (pack_bits) {
        case 8: return ((const uint8_t*)w)[idx];
        case 16: return ((const uint16_t*)w)[idx];
        default: return ((const uint32_t*)w)[idx];
    }
}

__device__ __forceinline__ float round_to_input(float v, int x_dt) {
    if (x_dt == GEMLITE_DT_FP8E4) return fp8e4m3_to_float(float_to_fp8e4m3(v));
    if (x_dt == GEMLITE_DT_FP8E5) return fp8e5m2_to_float(float_to_fp8e5m2(v));
    return v;
}

__global__ __launch_bounds__(256) void generic_matmul_kernel(const GenericParams p) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t m = blockIdx.y;
    if (n >= p.N) return;
    const float zs = p.zero_is_scalar ? load_as_float(p.zeros, 0, p.zeros_dt) : 0.f;
    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3);
    const bool packed64 = p.pack_bits == 64;
    float accf = 0.f;
    int acci = 0;
    int64_t cur_g = -1;
    float s = 1.f, z = 0.f;
    const uint32_t mask = p.nbits >= 32 ? 0xFFFFFFFFu : ((1u << p.nbits) - 1u);
    for (int64_t k = 0; k < p.K; ++k) {
        float q;
        if (p.e > 1) {
            const int64_t j = k / p.e;
            const int sh = (int)(k - j * p.e) * p.nbits;
            if (packed64) {
                const uint64_t wd = ((const uint64_t*)p.w)[j * p.stride_wk + n * p.stride_wn];
                q = (float)(uint32_t)((wd >> sh) & mask);
            } else {
                const uint32_t wd = load_word(p.w, j * p.stride_wk + n * p.stride_wn, p.pack_bits);
                q = (float)((wd >> sh) & mask);
            }
        } else {
            q = load_as_float(p.w, k * p.stride_wk + n * p.stride_wn, p.w_dt);
        }
        if (need_s || (need_z && !p.zero_is_scalar)) {
            const int64_t gidx = k / p.group_size;
            if (gidx != cur_g) {
                cur_g = gidx;
                if (need_s) s = load_as_float(p.scales, gidx * p.stride_meta_g + n * p.stride_meta_n, p.meta_dt);
                if (need_z && !p.zero_is_scalar)
                    z = load_as_float(p.zeros, gidx * p.stride_meta_g + n * p.stride_meta_n, p.zeros_dt);
            }
        }
        if (p.zero_is_scalar) z = zs;
        const float wv = dequant_f32(q, s, z, p.w_mode);
        if (p.int_acc) {
            acci += (int)load_as_float(p.x, m * p.stride_xm + k * p.stride_xk, p.x_dt) * (int)wv;
        } else {
            const float xv = load_as_float(p.x, m * p.stride_xm + k * p.stride_xk, p.x_dt);
            accf = __builtin_fmaf(xv, round_to_input(wv, p.x_dt), accf);
        }
    }
    epilogue_store(p.epi, p.int_acc ? (float)acci : accf, m, n);
}

// ---------------------------------------------------------------------------------------------
// unpacked, K-contiguous weights, W_group_mode 0: wave per column, 16-byte loads along K
// ---------------------------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// 4 packed fp8 (OCP e4m3 / e5m2) -> 4 floats with the hardware converters (v_cvt_pk_f32_fp8 / _bf8): 2 instructions
// instead of ~40 for the bit-twiddling form (FP8 x FP8 16384^2 at M = 1 ran at 1.2 TB/s on the latter)
__device__ __forceinline__ void fp8x4_to_f32(uint32_t v, bool e5m2, float (&o)[4]) {
    const f32x2_t lo = e5m2 ? __builtin_amdgcn_cvt_pk_f32_bf8((int)v, false) : __builtin_amdgcn_cvt_pk_f32_fp8((int)v, false);
    const f32x2_t hi = e5m2 ? __builtin_amdgcn_cvt_pk_f32_bf8((int)v, true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)v, true);
    o[0] = lo[0]; o[1] = lo[1]; o[2] = hi[0]; o[3] = hi[1];
}

template <int MB>
__global__ __launch_bounds__(256) void kmajor_matmul_kernel(const GenericParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    const int64_t m0 = (int64_t)blockIdx.y * MB;
    if (n >= p.N) return;
    const int esz = (p.w_dt == GEMLITE_DT_FP32) ? 4 : ((p.w_dt == GEMLITE_DT_FP16 || p.w_dt == GEMLITE_DT_BF16) ? 2 : 1);
    const int per = 16 / esz;  // elements per 16-byte load
    const uint8_t* wcol = (const uint8_t*)p.w + n * p.stride_wn * esz;
    float accf[MB];
    int acci[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) { accf[i] = 0.f; acci[i] = 0; }
    for (int64_t k = (int64_t)lane * per; k < p.K; k += 64 * per) {
        const u32x4 wv = *(const u32x4*)(wcol + k * esz);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            if (m0 + i >= p.M) continue;
            const uint8_t* xrow = (const uint8_t*)p.x + ((m0 + i) * p.stride_xm + k) * esz;
            const u32x4 xv = *(const u32x4*)xrow;
            if (p.w_dt == GEMLITE_DT_INT8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acci[i] = __builtin_amdgcn_sdot4((int)xv[q], (int)wv[q], acci[i], false);
            } else if (p.w_dt == GEMLITE_DT_FP16) {
#pragma unroll
                for (int q = 0; q < 4; ++q) accf[i] = F16Traits<half_tag>::dot2(xv[q], wv[q], accf[i]);
            } else if (p.w_dt == GEMLITE_DT_BF16) {
#pragma unroll
                for (int q = 0; q < 4; ++q) accf[i] = F16Traits<bf16_tag>::dot2(xv[q], wv[q], accf[i]);
            } else if (p.w_dt == GEMLITE_DT_FP32) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    accf[i] = __builtin_fmaf(__builtin_bit_cast(float, xv[q]), __builtin_bit_cast(float, wv[q]), accf[i]);
            } else if constexpr (MB == 1) {  // fp8 e4m3 / e5m2, decode: hardware converters
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float xf[4], wf[4];
                    fp8x4_to_f32(xv[q], p.w_dt != GEMLITE_DT_FP8E4, xf);
                    fp8x4_to_f32(wv[q], p.w_dt != GEMLITE_DT_FP8E4, wf);
#pragma unroll
                    for (int b = 0; b < 4; ++b) accf[i] = __builtin_fmaf(xf[b], wf[b], accf[i]);
                }
            }
            // (fp8 with MB = 4 is not planned: api.hip sends fp8 x fp8 at M > 1 to the MFMA kernels, manual GEMV types at M > 1 to
            //  the coverage kernel.  Round 2 carried a software-conversion fallback here because the hardware-converter form of this
            //  instantiation faulted on the MI355X for a reason that was never found; the variant is gone instead of parked.)
        }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        float v = p.w_dt == GEMLITE_DT_INT8 ? 0.f : accf[i];
        int vi = acci[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            v += __shfl_xor(v, off);
            vi += __shfl_xor(vi, off);
        }
        if (lane == 0 && m0 + i < p.M) epilogue_store(p.epi, p.w_dt == GEMLITE_DT_INT8 ? (float)vi : v, m0 + i, n);
    }
}

// kmajor_w8a16_kernel: 8-bit weight-only (A16W8: int8 / fp8 unpacked K-contiguous weights, fp16 / bf16 activations,
// helper.py:88-171): same one-wave-per-column streaming as kmajor_matmul_kernel, a lane converts its 16 weights of a
// 16-byte load to fp32 and multiplies them with the 32 bytes of x that face them; fp32 accumulation.  A channel-wise
// pre-scale (W_group_mode 2 with one group = K) is a per-column constant, applied once to the sum.
template <int MB>
__global__ __launch_bounds__(256) void kmajor_w8a16_kernel(const GenericParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    const int64_t m0 = (int64_t)blockIdx.y * MB;
    if (n >= p.N) return;
    const uint8_t* wcol = (const uint8_t*)p.w + n * p.stride_wn;
    float acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) acc[i] = 0.f;
    for (int64_t k = (int64_t)lane * 16; k < p.K; k += 64 * 16) {
        const u32x4 wv = *(const u32x4*)(wcol + k);
        float wf[16];
        if (p.w_dt == GEMLITE_DT_INT8) {
#pragma unroll
            for (int e = 0; e < 16; ++e) wf[e] = (float)(int8_t)(uint8_t)(wv[e >> 2] >> (8 * (e & 3)));
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t[4];
                fp8x4_to_f32(wv[q], p.w_dt != GEMLITE_DT_FP8E4, t);
#pragma unroll
                for (int b = 0; b < 4; ++b) wf[4 * q + b] = t[b];
            }
        }
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            if (m0 + i >= p.M) continue;
            const uint16_t* xrow = (const uint16_t*)p.x + (m0 + i) * p.stride_xm + k;
            const u32x4 x0 = *(const u32x4*)xrow, x1 = *(const u32x4*)(xrow + 8);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t d = (e < 8 ? x0 : x1)[(e & 7) >> 1];
                const uint16_t h = (uint16_t)(d >> (16 * (e & 1)));
                const float xf = p.x_dt == GEMLITE_DT_FP16 ? F16Traits<half_tag>::to_float(h) : F16Traits<bf16_tag>::to_float(h);
                acc[i] = __builtin_fmaf(xf, wf[e], acc[i]);
            }
        }
    }
    const float sc = p.w_mode == 2 ? load_as_float(p.scales, n, p.meta_dt) : 1.f;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        float v = acc[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0 && m0 + i < p.M) epilogue_store(p.epi, v * sc, m0 + i, n);
    }
}
// kmajor_fused_quant_kernel: M = 1 decode of a dynamically quantised layer (A8W8 int8 / fp8, helper.py:420-500) with the
// per-token activation quantisation FUSED into the prologue (reference: core.py:155-175 runs
// scale_activations_per_token as a separate launch in front of every matmul).  x is 8 KB at K = 4096: every block
// re-derives the row scale (amax / qmax, bit-identical to act_quant_per_token_kernel) and quantises x into LDS, then its
// 8 waves stream one weight row each (16-byte loads along K, v_dot4_i32_i8 against the LDS copy of x_q) — one launch
// instead of two, no x_q / scales_x round trip through memory.
template <int QDT>
__global__ __launch_bounds__(512) void kmajor_fused_quant_kernel(const GenericParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [K] quantised x, then 8 floats
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wmax = (float*)(smem + ((p.K + 15) & ~15));
    float amax = 0.f;
    for (int k = tid * 8; k < p.K; k += 512 * 8) {  // K % 8 == 0 (planner)
        const u32x4 v = *(const u32x4*)((const uint16_t*)p.x + k);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint16_t hbits = (uint16_t)(v[e >> 1] >> (16 * (e & 1)));
            const float f = p.x_dt == GEMLITE_DT_FP16 ? F16Traits<half_tag>::to_float(hbits) : F16Traits<bf16_tag>::to_float(hbits);
            amax = fmaxf(amax, fabsf(f));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if (lane == 0) wmax[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])), fmaxf(fmaxf(wmax[4], wmax[5]), fmaxf(wmax[6], wmax[7])));
    float qmin, qmax;
    if (QDT == GEMLITE_DT_INT8) { qmin = -128.f; qmax = 127.f; }
    else if (QDT == GEMLITE_DT_FP8E4) { qmin = -448.f; qmax = 448.f; }
    else { qmin = -57344.f; qmax = 57344.f; }
    const float sx = fmaxf(__fdiv_rn(amax, qmax), 1e-6f);
    for (int k = tid * 8; k < p.K; k += 512 * 8) {
        const u32x4 v = *(const u32x4*)((const uint16_t*)p.x + k);
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint16_t hbits = (uint16_t)(v[e >> 1] >> (16 * (e & 1)));
            const float f = p.x_dt == GEMLITE_DT_FP16 ? F16Traits<half_tag>::to_float(hbits) : F16Traits<bf16_tag>::to_float(hbits);
            t[e] = fminf(fmaxf(__fdiv_rn(f, sx), qmin), qmax);
        }
        uint32_t q[2] = {0u, 0u};
        if constexpr (QDT == GEMLITE_DT_INT8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e >> 2] |= (uint32_t)(uint8_t)(int8_t)floorf(t[e] + 0.5f) << (8 * (e & 3));
        } else {  // hardware converters, as in act_quant_per_token_vec_kernel
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int w = 0;
                if constexpr (QDT == GEMLITE_DT_FP8E4) {
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(t[4 * h], t[4 * h + 1], w, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(t[4 * h + 2], t[4 * h + 3], w, true);
                } else {
                    w = __builtin_amdgcn_cvt_pk_bf8_f32(t[4 * h], t[4 * h + 1], w, false);
                    w = __builtin_amdgcn_cvt_pk_bf8_f32(t[4 * h + 2], t[4 * h + 3], w, true);
                }
                q[h] = (uint32_t)w;
            }
        }
        *(u32x2*)(smem + k) = (u32x2){q[0], q[1]};
    }
    __syncthreads();
    const int64_t n = (int64_t)blockIdx.x * 8 + wave;
    if (n >= p.N) return;
    const uint8_t* wcol = (const uint8_t*)p.w + n * p.stride_wn;
    float accf = 0.f;
    int acci = 0;
    for (int k = lane * 16; k < p.K; k += 64 * 16) {  // K % 16 == 0 (planner)
        const u32x4 wv = *(const u32x4*)(wcol + k);
        const u32x4 xv = *(const u32x4*)(smem + k);
        if (QDT == GEMLITE_DT_INT8) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) acci = __builtin_amdgcn_sdot4((int)xv[qd], (int)wv[qd], acci, false);
        } else {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                float xf[4], wf[4];
                fp8x4_to_f32(xv[qd], QDT != GEMLITE_DT_FP8E4, xf);
                fp8x4_to_f32(wv[qd], QDT != GEMLITE_DT_FP8E4, wf);
#pragma unroll
                for (int b = 0; b < 4; ++b) accf = __builtin_fmaf(xf[b], wf[b], accf);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        accf += __shfl_xor(accf, off);
        acci += __shfl_xor(acci, off);
    }
    if (lane == 0) {
        float v = QDT == GEMLITE_DT_INT8 ? (float)acci : accf;
        // same arithmetic as epilogue_scale(): acc * (s_x * s_w[n]) for mode 3, acc * s_x for mode 2
        if (p.epi.c_mode == 3) v *= sx * load_as_float(p.epi.scales_w, n, p.epi.meta_dt);
        else v *= sx;
        store_from_float(p.epi.out, n * p.epi.stride_on, p.epi.out_dt, v);
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// a8w8_decode_kernel (round 4): M = 1 of the unpacked 8-bit layers (A8W8 int8 / fp8 dynamic, BASELINE configs[3] / [4]) built like
// gemv_w4_decode3_kernel — a launch this size is latency, not arithmetic.  Replaces gemv_INT_splitK_kernel (gemlite/triton_kernels/
// gemv_splitK_kernels.py:240-420) for these layers and, in its FUSED form, scale_activations_per_token (quant_utils.py:268-347) in the
// same launch (core.py:155-175 runs it as a launch of its own).  Block = 16 waves = 16 output columns; a wave owns ONE K-contiguous
// weight row and requests it as 16-byte loads per lane (k = j * 1024 + lane * 16) in batches of up to 8 — at K = 4096 the whole row is
// in flight before anything else happens.  FUSED: the block then quantises the one row of x into LDS (amax over the block, IEEE
// divisions: bit-identical to act_quant_per_token_kernel) UNDER that round trip — kmajor_fused_quant_kernel quantised first and only
// then asked for its weights.  v_dot4_i32_i8 (exact) / hardware fp8 converters + fp32 fma; one wave = one column: no cross-wave sum.
// ---------------------------------------------------------------------------------------------------------------------
template <int QDT, bool FUSED>
__global__ __launch_bounds__(1024, 1) void a8w8_decode_kernel(const GenericParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // FUSED: [K] quantised x, then 16 floats
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n = (int64_t)blockIdx.x * 16 + wave;
    const uint8_t* wcol = (const uint8_t*)p.w + n * p.stride_wn;
    constexpr int B = 8;  // 16-byte pieces of the row in flight per lane
    const int npieces = p.K >> 10;  // K % 1024 == 0 (planner)
    u32x4 wv[B];
#pragma unroll
    for (int j = 0; j < B; ++j)
        if (j < npieces) wv[j] = __builtin_nontemporal_load((const u32x4*)(wcol + (j << 10) + lane * 16));

    float sx = 1.f;
    const uint8_t* xq;
    if constexpr (FUSED) {
        float* wmax = (float*)(smem + p.K);
        const uint16_t* xr = (const uint16_t*)p.x;
        const bool f16 = p.x_dt == GEMLITE_DT_FP16;
        float amax = 0.f;
        for (int k = tid * 8; k < p.K; k += 1024 * 8) {
            const u32x4 v = *(const u32x4*)(xr + k);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint16_t hbits = (uint16_t)(v[e >> 1] >> (16 * (e & 1)));
                amax = fmaxf(amax, fabsf(f16 ? F16Traits<half_tag>::to_float(hbits) : F16Traits<bf16_tag>::to_float(hbits)));
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        if (lane == 0) wmax[wave] = amax;
        __syncthreads();
        amax = wmax[lane & 15];
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        constexpr float qmin = QDT == GEMLITE_DT_INT8 ? -128.f : (QDT == GEMLITE_DT_FP8E4 ? -448.f : -57344.f);
        constexpr float qmax = QDT == GEMLITE_DT_INT8 ? 127.f : (QDT == GEMLITE_DT_FP8E4 ? 448.f : 57344.f);
        sx = fmaxf(__fdiv_rn(amax, qmax), 1e-6f);
        for (int k = tid * 8; k < p.K; k += 1024 * 8) {
            const u32x4 v = *(const u32x4*)(xr + k);
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint16_t hbits = (uint16_t)(v[e >> 1] >> (16 * (e & 1)));
                const float f = f16 ? F16Traits<half_tag>::to_float(hbits) : F16Traits<bf16_tag>::to_float(hbits);
                t[e] = fminf(fmaxf(__fdiv_rn(f, sx), qmin), qmax);
            }
            uint32_t q[2] = {0u, 0u};
            if constexpr (QDT == GEMLITE_DT_INT8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e >> 2] |= (uint32_t)(uint8_t)(int8_t)floorf(t[e] + 0.5f) << (8 * (e & 3));
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int w = 0;
                    if constexpr (QDT == GEMLITE_DT_FP8E4) {
                        w = __builtin_amdgcn_cvt_pk_fp8_f32(t[4 * h], t[4 * h + 1], w, false);
                        w = __builtin_amdgcn_cvt_pk_fp8_f32(t[4 * h + 2], t[4 * h + 3], w, true);
                    } else {
                        w = __builtin_amdgcn_cvt_pk_bf8_f32(t[4 * h], t[4 * h + 1], w, false);
                        w = __builtin_amdgcn_cvt_pk_bf8_f32(t[4 * h + 2], t[4 * h + 3], w, true);
                    }
                    q[h] = (uint32_t)w;
                }
            }
            *(u32x2*)(smem + k) = (u32x2){q[0], q[1]};
        }
        __syncthreads();
        xq = smem;
    } else {
        xq = (const uint8_t*)p.x;
    }

    float accf = 0.f;
    int acci = 0;
    auto consume = [&](u32x4 w, int j) __attribute__((always_inline)) {
        const u32x4 xv = *(const u32x4*)(xq + (j << 10) + lane * 16);
        if constexpr (QDT == GEMLITE_DT_INT8) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) acci = __builtin_amdgcn_sdot4((int)xv[qd], (int)w[qd], acci, false);
        } else {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                float xf[4], wf[4];
                fp8x4_to_f32(xv[qd], QDT != GEMLITE_DT_FP8E4, xf);
                fp8x4_to_f32(w[qd], QDT != GEMLITE_DT_FP8E4, wf);
#pragma unroll
                for (int b = 0; b < 4; ++b) accf = __builtin_fmaf(xf[b], wf[b], accf);
            }
        }
    };
    for (int base = 0; base < npieces; base += B) {
#pragma unroll
        for (int j = 0; j < B; ++j) {
            if (base + j < npieces) {
                const u32x4 w = wv[j];
                if (base + j + B < npieces) wv[j] = __builtin_nontemporal_load((const u32x4*)(wcol + ((base + j + B) << 10) + lane * 16));
                consume(w, base + j);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        accf += __shfl_xor(accf, off);
        acci += __shfl_xor(acci, off);
    }
    if (lane == 0) {
        float v = QDT == GEMLITE_DT_INT8 ? (float)acci : accf;
        if constexpr (FUSED) {  // same arithmetic as epilogue_scale(): acc * (s_x * s_w[n]) for mode 3, acc * s_x for mode 2
            if (p.epi.c_mode == 3) v *= sx * load_as_float(p.epi.scales_w, n, p.epi.meta_dt);
            else v *= sx;
            store_from_float(p.epi.out, n * p.epi.stride_on, p.epi.out_dt, v);
        } else {
            epilogue_store(p.epi, v, 0, n);
        }
    }
}
const void* a8w8_decode_kernel_fn(int qdt, bool fused) {
    typedef void (*fn_t)(const GenericParams);
    fn_t f = nullptr;
    if (qdt == GEMLITE_DT_INT8) f = fused ? a8w8_decode_kernel<GEMLITE_DT_INT8, true> : a8w8_decode_kernel<GEMLITE_DT_INT8, false>;
    else if (qdt == GEMLITE_DT_FP8E4) f = fused ? a8w8_decode_kernel<GEMLITE_DT_FP8E4, true> : a8w8_decode_kernel<GEMLITE_DT_FP8E4, false>;
    else if (qdt == GEMLITE_DT_FP8E5) f = fused ? a8w8_decode_kernel<GEMLITE_DT_FP8E5, true> : a8w8_decode_kernel<GEMLITE_DT_FP8E5, false>;
    return (const void*)f;
}

// ---------------------------------------------------------------------------------------------------------------------
// a16w8_decode_kernel (round 4): M = 1 of the 8-bit weight-only layers (A16W8 int8 / fp8, helper.py:88-171) in the shape of
// a8w8_decode_kernel: block = 16 waves = 16 output columns, a wave requests its whole K-contiguous weight row (16-byte non-temporal loads,
// up to 8 per lane in flight) before anything else, the block then copies the one row of x into LDS (K x 2 bytes) and every lane reads the
// 32 bytes of x that face each of its 16-byte weight pieces from there.  fp16: weights -> fp16 pairs (int8 through 0x6400 | (b ^ 0x80) =
// 1024 + (b + 128), minus 1152: exact; fp8 by the hardware converter) and v_dot2_f32_f16; bf16: fp32 fma.  One wave = one column: no
// cross-wave sum; the channel scale multiplies the sum once.  a16w8_rows_kernel took 7.3 us at 4096^2 for the same bytes.
// ---------------------------------------------------------------------------------------------------------------------
template <typename Tag, int WDT>
__global__ __launch_bounds__(1024, 1) void a16w8_decode_kernel(const GenericParams p) {
    using TR = F16Traits<Tag>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [K] 16-bit x
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n = (int64_t)blockIdx.x * 16 + wave;
    const uint8_t* wcol = (const uint8_t*)p.w + n * p.stride_wn;
    constexpr int B = (TR::DT == GEMLITE_DT_BF16 && WDT == GEMLITE_DT_INT8) ? 4 : 8;  // (8 spilled 73 registers in the bf16 x int8 form)
    const int npieces = p.K >> 10;  // K % 1024 == 0 (planner)
    u32x4 wv[B];
#pragma unroll
    for (int j = 0; j < B; ++j)
        if (j < npieces) wv[j] = __builtin_nontemporal_load((const u32x4*)(wcol + (j << 10) + lane * 16));
    for (int k = tid * 8; k < p.K; k += 1024 * 8) *(u32x4*)(smem + 2 * k) = *(const u32x4*)((const uint16_t*)p.x + k);
    __syncthreads();
    float acc = 0.f;
    auto consume = [&](u32x4 w, int j) __attribute__((always_inline)) {
        const u32x4 x0 = *(const u32x4*)(smem + 2 * ((j << 10) + lane * 16)), x1 = *(const u32x4*)(smem + 2 * ((j << 10) + lane * 16) + 16);
#pragma unroll
        for (int d = 0; d < 4; ++d) {  // dword d = weights k 4 d .. 4 d + 3 of the piece; x pairs (4 d, 4 d + 1), (4 d + 2, 4 d + 3)
            const uint32_t xa = (d < 2 ? x0 : x1)[2 * (d & 1)], xb = (d < 2 ? x0 : x1)[2 * (d & 1) + 1];
            if constexpr (TR::DT == GEMLITE_DT_FP16) {
                uint32_t pa, pb;
                if constexpr (WDT == GEMLITE_DT_INT8) {
                    const uint32_t u = w[d] ^ 0x80808080u;
                    const h2_t off = {(_Float16)1152.0f, (_Float16)1152.0f};
                    pa = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2_t, __builtin_amdgcn_perm(0x64646464u, u, 0x04010400u)) - off);
                    pb = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2_t, __builtin_amdgcn_perm(0x64646464u, u, 0x04030402u)) - off);
                } else {
                    pa = WDT == GEMLITE_DT_FP8E4 ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w[d], 1.0f, false))
                                                 : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(w[d], 1.0f, false));
                    pb = WDT == GEMLITE_DT_FP8E4 ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w[d], 1.0f, true))
                                                 : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(w[d], 1.0f, true));
                }
                acc = TR::dot2(xa, pa, acc);
                acc = TR::dot2(xb, pb, acc);
            } else {
                float wf[4];
                if constexpr (WDT == GEMLITE_DT_INT8) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) wf[b] = (float)(int8_t)(uint8_t)(w[d] >> (8 * b));
                } else {
                    fp8x4_to_f32(w[d], WDT != GEMLITE_DT_FP8E4, wf);
                }
                acc = __builtin_fmaf(__builtin_bit_cast(float, xa << 16), wf[0], acc);
                acc = __builtin_fmaf(__builtin_bit_cast(float, xa & 0xFFFF0000u), wf[1], acc);
                acc = __builtin_fmaf(__builtin_bit_cast(float, xb << 16), wf[2], acc);
                acc = __builtin_fmaf(__builtin_bit_cast(float, xb & 0xFFFF0000u), wf[3], acc);
            }
        }
    };
    for (int base = 0; base < npieces; base += B) {
#pragma unroll
        for (int j = 0; j < B; ++j) {
            if (base + j < npieces) {
                const u32x4 w = wv[j];
                if (base + j + B < npieces) wv[j] = __builtin_nontemporal_load((const u32x4*)(wcol + ((base + j + B) << 10) + lane * 16));
                consume(w, base + j);
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        const float sc = p.w_mode == 2 ? load_as_float(p.scales, n, p.meta_dt) : 1.f;  // per-channel pre-scale: once, on the sum
        epilogue_store(p.epi, acc * sc, 0, n);
    }
}
const void* a16w8_decode_kernel_fn(int x_dt, int w_dt) {
    typedef void (*fn_t)(const GenericParams);
    fn_t f = nullptr;
    const bool h = x_dt == GEMLITE_DT_FP16;
    if (w_dt == GEMLITE_DT_INT8) f = h ? a16w8_decode_kernel<half_tag, GEMLITE_DT_INT8> : a16w8_decode_kernel<bf16_tag, GEMLITE_DT_INT8>;
    else if (w_dt == GEMLITE_DT_FP8E4) f = h ? a16w8_decode_kernel<half_tag, GEMLITE_DT_FP8E4> : a16w8_decode_kernel<bf16_tag, GEMLITE_DT_FP8E4>;
    else if (w_dt == GEMLITE_DT_FP8E5) f = h ? a16w8_decode_kernel<half_tag, GEMLITE_DT_FP8E5> : a16w8_decode_kernel<bf16_tag, GEMLITE_DT_FP8E5>;
    return (const void*)f;
}

const void* kmajor_fused_quant_kernel_fn(int qdt) {
    return qdt == GEMLITE_DT_INT8 ? (const void*)kmajor_fused_quant_kernel<GEMLITE_DT_INT8>
           : (qdt == GEMLITE_DT_FP8E4 ? (const void*)kmajor_fused_quant_kernel<GEMLITE_DT_FP8E4>
                                       : (const void*)kmajor_fused_quant_kernel<GEMLITE_DT_FP8E5>);
}

const void* kmajor_w8a16_kernel_fn(int mb) {
    return mb == 1 ? (const void*)kmajor_w8a16_kernel<1> : (const void*)kmajor_w8a16_kernel<4>;
}

const void* generic_kernel_fn() { return (const void*)generic_matmul_kernel; }
const void* kmajor_kernel_fn(int mb) {
    return mb == 1 ? (const void*)kmajor_matmul_kernel<1> : (const void*)kmajor_matmul_kernel<4>;
}

// ---------------------------------------------------------------------------------------------
// per-token dynamic activation quantisation  (spec: gemlite/quant_utils.py:231-253; the Triton kernel
// :268-305 uses floor(x + 0.5) on AMD, :259-266).  One block per row; x is read twice (second pass from L2).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void act_quant_per_token_kernel(const void* x, void* y, float* scales, int64_t K,
                                                                 int64_t stride_xm, int in_dt, int out_dt) {
    __shared__ float wmax[4];
    const int64_t m = blockIdx.x;
    const int tid = threadIdx.x;
    float amax = 0.f;
    for (int64_t k = tid; k < K; k += 256) amax = fmaxf(amax, fabsf(load_as_float(x, m * stride_xm + k, in_dt)));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if ((tid & 63) == 0) wmax[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    float qmin, qmax;
    if (out_dt == GEMLITE_DT_INT8) { qmin = -128.f; qmax = 127.f; }
    else if (out_dt == GEMLITE_DT_FP8E4) { qmin = -448.f; qmax = 448.f; }
    else { qmin = -57344.f; qmax = 57344.f; }
    const float s = fmaxf(__fdiv_rn(amax, qmax), 1e-6f);
    if (tid == 0) scales[m] = s;
    for (int64_t k = tid; k < K; k += 256) {
        float v = __fdiv_rn(load_as_float(x, m * stride_xm + k, in_dt), s);
        v = fminf(fmaxf(v, qmin), qmax);
        if (out_dt == GEMLITE_DT_INT8) {
            ((int8_t*)y)[m * K + k] = (int8_t)floorf(v + 0.5f);
        } else if (out_dt == GEMLITE_DT_FP8E4) {
            ((uint8_t*)y)[m * K + k] = float_to_fp8e4m3(v);
        } else {
            ((uint8_t*)y)[m * K + k] = float_to_fp8e5m2(v);
        }
    }
}
const void* act_quant_kernel_fn() { return (const void*)act_quant_per_token_kernel; }

// Round 3: the same arithmetic with the row held in registers.  The kernel above walks a row with 2-byte loads, twice, one
// element per thread and iteration — ~10 us for M = 16 .. 256 rows of 4096 (two chains of 16 dependent memory round trips); the
// dynamic A8W8 processors (helper.py:420-500) pay that in front of EVERY matmul (same-method comparison with the reference on the
// MI355X, profiles/r03/: int8 4096^2 M = 16 layer(x) 17.4 us of which the matmul is 7).  Here a thread loads R x 16 bytes of its row
// once (R = K / 2048 <= 8), the block reduces |x| max through LDS, and the quantised bytes leave as 8-byte stores: one memory round
// trip per row.  16-bit inputs, K % 8 == 0, K <= 16384, 16-byte aligned rows; anything else takes the kernel above.
template <typename Tag, int ODT, int R>
__global__ __launch_bounds__(256) void act_quant_per_token_vec_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ y,
                                                                     float* __restrict__ scales, int K, int64_t stride_xm) {
    __shared__ float wmax[4];
    const int64_t m = blockIdx.x;
    const int tid = threadIdx.x;
    const uint16_t* row = x + m * stride_xm;
    u32x4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = (r * 256 + tid) * 8;
        v[r] = k < K ? *(const u32x4*)(row + k) : (u32x4){0u, 0u, 0u, 0u};
    }
    float amax = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            amax = fmaxf(amax, fabsf(F16Traits<Tag>::to_float((uint16_t)(v[r][e >> 1] >> (16 * (e & 1))))));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if ((tid & 63) == 0) wmax[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    constexpr float qmin = ODT == GEMLITE_DT_INT8 ? -128.f : (ODT == GEMLITE_DT_FP8E4 ? -448.f : -57344.f);
    constexpr float qmax = ODT == GEMLITE_DT_INT8 ? 127.f : (ODT == GEMLITE_DT_FP8E4 ? 448.f : 57344.f);
    const float s = fmaxf(__fdiv_rn(amax, qmax), 1e-6f);
    if (tid == 0) scales[m] = s;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = (r * 256 + tid) * 8;
        if (k < K) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = F16Traits<Tag>::to_float((uint16_t)(v[r][e >> 1] >> (16 * (e & 1))));
                t[e] = fminf(fmaxf(__fdiv_rn(f, s), qmin), qmax);
            }
            uint32_t q[2] = {0u, 0u};
            if constexpr (ODT == GEMLITE_DT_INT8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e >> 2] |= (uint32_t)(uint8_t)(int8_t)floorf(t[e] + 0.5f) << (8 * (e & 3));
            } else {
                // the hardware converters (round to nearest even, subnormals kept; the clamp above keeps them away from overflow):
                // two values per instruction instead of ~15 VALU each for the software form — bit-identical, tested against the oracle
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int w = 0;
                    if constexpr (ODT == GEMLITE_DT_FP8E4) {
                        w = __builtin_amdgcn_cvt_pk_fp8_f32(t[4 * h], t[4 * h + 1], w, false);
                        w = __builtin_amdgcn_cvt_pk_fp8_f32(t[4 * h + 2], t[4 * h + 3], w, true);
                    } else {
                        w = __builtin_amdgcn_cvt_pk_bf8_f32(t[4 * h], t[4 * h + 1], w, false);
                        w = __builtin_amdgcn_cvt_pk_bf8_f32(t[4 * h + 2], t[4 * h + 3], w, true);
                    }
                    q[h] = (uint32_t)w;
                }
            }
            *(u32x2*)(y + m * K + k) = (u32x2){q[0], q[1]};
        }
    }
}
typedef void (*act_quant_vec_fn)(const uint16_t*, uint8_t*, float*, int, int64_t);
template <typename Tag, int ODT>
static act_quant_vec_fn act_quant_vec_pick_r(int r) {
    switch (r) {
        case 1: return act_quant_per_token_vec_kernel<Tag, ODT, 1>;
        case 2: return act_quant_per_token_vec_kernel<Tag, ODT, 2>;
        case 4: return act_quant_per_token_vec_kernel<Tag, ODT, 4>;
        default: return act_quant_per_token_vec_kernel<Tag, ODT, 8>;
    }
}
// nullptr when the vector form does not apply
const void* act_quant_vec_kernel_fn(int in_dt, int out_dt, int64_t K, int64_t stride_xm, const void* x, const void* y) {
    if (!(in_dt == GEMLITE_DT_FP16 || in_dt == GEMLITE_DT_BF16) || K % 8 != 0 || K > 16384 || stride_xm % 8 != 0) return nullptr;
    if (((uintptr_t)x % 16) != 0 || ((uintptr_t)y % 8) != 0) return nullptr;
    const int need = (int)((K + 2047) / 2048), r = need <= 1 ? 1 : (need <= 2 ? 2 : (need <= 4 ? 4 : 8));
    act_quant_vec_fn f = nullptr;
    if (in_dt == GEMLITE_DT_FP16) {
        f = out_dt == GEMLITE_DT_INT8 ? act_quant_vec_pick_r<half_tag, GEMLITE_DT_INT8>(r)
            : (out_dt == GEMLITE_DT_FP8E4 ? act_quant_vec_pick_r<half_tag, GEMLITE_DT_FP8E4>(r) : act_quant_vec_pick_r<half_tag, GEMLITE_DT_FP8E5>(r));
    } else {
        f = out_dt == GEMLITE_DT_INT8 ? act_quant_vec_pick_r<bf16_tag, GEMLITE_DT_INT8>(r)
            : (out_dt == GEMLITE_DT_FP8E4 ? act_quant_vec_pick_r<bf16_tag, GEMLITE_DT_FP8E4>(r) : act_quant_vec_pick_r<bf16_tag, GEMLITE_DT_FP8E5>(r));
    }
    return (const void*)f;
}

// ---------------------------------------------------------------------------------------------
// bit packing along K, output transposed [K/e, N]  (layout spec: gemlite/bitpack.py:36-60 + core.py:384-398)
// thread = one packed word; consecutive threads = consecutive n (coalesced stores; the uint8 reads of a
// wave are 64 rows x e contiguous bytes)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_over_cols_kernel(const uint8_t* w, void* out, int64_t N, int64_t K,
                                                            int64_t ld_in, int nbits, int pack_bits) {
    const int e = pack_bits / nbits;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (K / e) * N;
    if (idx >= total) return;
    const int64_t j = idx / N, n = idx - j * N;
    const uint8_t* src = w + n * ld_in + j * e;
    uint64_t word = 0;
    for (int i = 0; i < e; ++i) word |= (uint64_t)src[i] << (nbits * i);
    switch (pack_bits) {
        case 8: ((uint8_t*)out)[idx] = (uint8_t)word; break;
        case 16: ((uint16_t*)out)[idx] = (uint16_t)word; break;
        case 32: ((uint32_t*)out)[idx] = (uint32_t)word; break;
        default: ((uint64_t*)out)[idx] = word; break;
    }
}

__global__ __launch_bounds__(256) void unpack_over_cols_kernel(const void* packed, uint8_t* out, int64_t N, int64_t K,
                                                              int nbits, int pack_bits) {
    const int e = pack_bits / nbits;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (K / e) * N;
    if (idx >= total) return;
    const int64_t j = idx / N, n = idx - j * N;
    uint64_t word;
    switch (pack_bits) {
        case 8: word = ((const uint8_t*)packed)[idx]; break;
        case 16: word = ((const uint16_t*)packed)[idx]; break;
        case 32: word = ((const uint32_t*)packed)[idx]; break;
        default: word = ((const uint64_t*)packed)[idx]; break;
    }
    const uint64_t mask = (1ull << nbits) - 1ull;
    uint8_t* dst = out + n * K + j * e;
    for (int i = 0; i < e; ++i) dst[i] = (uint8_t)((word >> (nbits * i)) & mask);
}
// 32-bit words, the layout pack() produces (round 5; VERDICT r4: the kernel above reads 8 bytes per lane from a different row each — one
// cache line per lane, 97 us average for inputs whose bytes are 4 us of HBM).  Tile = 64 columns (n) x 256 k: the [n][k] bytes come in as
// 16-byte pieces, consecutive lanes along k (256 contiguous bytes per row), turn around in LDS (row pitch 260 bytes = 65 dwords: the 64
// lanes of a word row hit 64 different banks) and leave as words [k / e][n] with consecutive lanes along n (256 contiguous bytes per row).
__global__ __launch_bounds__(256) void pack_over_cols32_kernel(const uint8_t* w, uint32_t* out, int64_t N, int64_t K, int64_t ld_in, int nbits) {
    constexpr int TN = 64, TK = 256, PITCH = 260;
    __shared__ __attribute__((aligned(16))) unsigned char tile[TN * PITCH];
    const int e = 32 / nbits;
    const int64_t n0 = (int64_t)blockIdx.x * TN, k0 = (int64_t)blockIdx.y * TK;
    const int tid = threadIdx.x;
    const bool vec = (ld_in % 16 == 0) && (((uintptr_t)w) % 16 == 0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 256 + tid, r = idx >> 4, pc = idx & 15;
        const int64_t n = n0 + r, k = k0 + pc * 16;
        uint32_t v[4] = {0u, 0u, 0u, 0u};
        if (n < N && k < K) {
            if (vec && k + 16 <= K) {
                const u32x4 t = *(const u32x4*)(w + n * ld_in + k);
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
            } else {
                for (int b = 0; b < 16 && k + b < K; ++b) v[b >> 2] |= (uint32_t)w[n * ld_in + k + b] << (8 * (b & 3));
            }
        }
        uint32_t* dst = (uint32_t*)(tile + r * PITCH + pc * 16);  // (PITCH % 4 == 0: dword stores)
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
    }
    __syncthreads();
    const int wpt = TK / e;  // words per column in this tile
    for (int o = tid; o < wpt * TN; o += 256) {
        const int jl = o / TN, nl = o % TN;
        const int64_t n = n0 + nl, kk = k0 + (int64_t)jl * e;
        if (n >= N || kk >= K) continue;
        const unsigned char* src = tile + nl * PITCH + jl * e;
        uint32_t word = 0;
        for (int i = 0; i < e; i += 4) {
            const uint32_t q = *(const uint32_t*)(src + i);
            word |= ((q & 0xFFu) << (nbits * i)) | (((q >> 8) & 0xFFu) << (nbits * (i + 1))) | (((q >> 16) & 0xFFu) << (nbits * (i + 2))) |
                    ((q >> 24) << (nbits * (i + 3)));
        }
        out[(kk / e) * N + n] = word;
    }
}

const void* pack32_kernel_fn() { return (const void*)pack_over_cols32_kernel; }
const void* pack_kernel_fn() { return (const void*)pack_over_cols_kernel; }
const void* unpack_kernel_fn() { return (const void*)unpack_over_cols_kernel; }

}  // namespace gl
