// gemv_a8wn.hip — decode-size matmul (M <= 4) for 8-bit activations x packed 4- / 2-bit weights:
// the reference's A8Wn_HQQ_INT_dynamic (fp8 e4m3 activations, helper.py:502-615) and A8W158_INT_dynamic (BitNet, int8
// activations x ternary 2-bit codes, helper.py:1006-1062) through gemv_INT_*_kernel (gemv_kernels.py / gemv_revsplitK_kernels.py).
//
// Numerics are the reference's, per kernel FAMILY: from 2 rows on (GEMM_SPLITK) the dequantised weight is cast to the ACTIVATION
// type before the product (`b.to(a.dtype)`, gemm_kernels.py:384) — for fp8 that is a real rounding step per weight; at ONE row
// (GEMV family, dot_prod_mode 0) it stays in the metadata type (round 4: found against the reference's own M = 1 output).  Either way, so the group-factored form of gemv_wn.hip (sum x q first, scale once per group) does not apply here: every
// weight is dequantised, rounded to e4m3 with the hardware converter and multiplied as fp32; int8 activations use exact
// integer codes on v_dot4_i32_i8.
//
// Layout (HBM-bound; one block per 16 columns so that N = 4096 gives 256 blocks without splitting K across blocks):
//   block = 16 waves; lane (c4 = lane & 3, r = lane >> 2) owns columns [4 c4, 4 c4 + 4) of the tile and packed row r of a
//   16-row group; wave w takes the groups w, w + 16, ...  One 16-byte load per lane and group = 4 columns x one word; the
//   activations of that packed row (8 or 16 bytes per batch row) and the group's (scale, zero) come from L2 (shared by
//   all blocks).  Partial sums: 4 xor-shuffles inside the wave, then 16 waves through LDS in fixed order (deterministic).
#include "gl_common.h"

#include <type_traits>

namespace gl {

typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace a8 {

template <int NBITS>
struct Codes;  // integer codes of one 8-k slice as bytes: ev = codes 0, 2, 4, 6; od = codes 1, 3, 5, 7
template <>
struct Codes<4> {
    static constexpr int SLICES = 1;
    static __device__ __forceinline__ void run(uint32_t w, int, uint32_t& ev, uint32_t& od) {
        ev = w & 0x0F0F0F0Fu;
        od = (w >> 4) & 0x0F0F0F0Fu;
    }
};
template <>
struct Codes<2> {
    static constexpr int SLICES = 2;  // 16 codes per word
    static __device__ __forceinline__ void run(uint32_t w, int t, uint32_t& ev, uint32_t& od) {
        const uint32_t m0 = w & 0x03030303u, m1 = (w >> 2) & 0x03030303u, m2 = (w >> 4) & 0x03030303u, m3 = (w >> 6) & 0x03030303u;
        const uint32_t sel = t ? 0x07030602u : 0x05010400u;  // byte b of m_a = code 4b + a
        ev = __builtin_amdgcn_perm(m2, m0, sel);
        od = __builtin_amdgcn_perm(m3, m1, sel);
    }
};

}  // namespace a8

// FQ (round 4, MB = 1): the launch takes the UNQUANTISED 16-bit row of x and quantises it per token itself — both of a wave's weight
// requests go out first, then the block computes amax / the scale and writes the quantised row into LDS under that round trip (same
// arithmetic as act_quant_per_token_kernel: bit-identical to quantiser + this kernel), reads x from LDS instead of L2 and applies the
// row scale in its epilogue.  layer(x) of A8W4 / A8W2 fp8-dynamic and BitNet int8-dynamic at M = 1: one launch instead of two
// (10.4 / 9.2 / 8.5 us -> see profiles/r04/probe_processors*.log).
template <typename Tag, int NBITS, int XDT, int MB, bool FQ = false>
__global__ __launch_bounds__(1024) void gemv_a8wn_kernel(const WnParams p) {
    static_assert(!FQ || MB == 1, "in-launch activation quantisation: one row");
    extern __shared__ __attribute__((aligned(16))) unsigned char xq_lds[];  // FQ: [K] quantised x, then 16 floats
    using TR = F16Traits<Tag>;
    using CD = a8::Codes<NBITS>;
    constexpr bool INT = XDT == GEMLITE_DT_INT8;
    constexpr int E = 32 / NBITS;       // k per packed word
    constexpr int XB = E;               // activation bytes per packed row
    constexpr int NW = 16;
    __shared__ float red[NW][MB][16];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c4 = lane & 3, r = lane >> 2;
    const int n0 = blockIdx.x * 16 + 4 * c4;  // first of this lane's 4 columns
    const int rows = p.K / E, ngroups = rows >> 4;
    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    const float scalar_zero = p.zero_is_scalar ? (float)((const int32_t*)p.zeros)[0] : 0.f;
    const float u13 = (p.w_mode == 1 || p.w_mode == 3) ? 1.f : 0.f, u4 = p.w_mode == 4 ? 1.f : 0.f;
    const uint32_t* wq = p.w;
    const uint8_t* xb = (const uint8_t*)p.x;

    float accf[MB][4];
    int acci[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) { accf[m][c] = 0.f; acci[m][c] = 0; }

    struct Req { u32x4 w; u32x2 s, z; uint32_t x[MB][XB / 4]; };
    auto request = [&](Req& q, int g) {
        const int row = g * 16 + r;
        q.w = *(const u32x4*)(wq + (int64_t)row * p.stride_wk + n0);
        const int64_t mg = (int64_t)(((int64_t)row * E) >> p.gs_shift) * p.stride_meta_g + n0;
        if (need_s) q.s = *(const u32x2*)((const uint16_t*)p.scales + mg);
        if (need_z) q.z = *(const u32x2*)((const uint16_t*)p.zeros + mg);
        if constexpr (FQ) {  // x comes from LDS at consume time
            q.x[0][0] = (uint32_t)row;
            return;
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const uint8_t* xr = xb + (int64_t)(m < p.M ? m : 0) * p.stride_xm + (int64_t)row * XB;
            if constexpr (XB == 8) {
                const u32x2 v = *(const u32x2*)xr;
                q.x[m][0] = v[0]; q.x[m][1] = v[1];
            } else {
                const u32x4 v = *(const u32x4*)xr;
#pragma unroll
                for (int i = 0; i < XB / 4; ++i) q.x[m][i] = v[i];
            }
        }
    };
    auto consume = [&](const Req& q) {
#pragma unroll
        for (int t = 0; t < CD::SLICES; ++t) {
            // activations of this 8-k slice: bytes [8t, 8t + 8) of the row
            float xf[MB][8];
            uint32_t xe[MB], xo[MB];
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                uint32_t lo, hi;
                if constexpr (FQ) {
                    const u32x2 v = *(const u32x2*)(xq_lds + (size_t)q.x[0][0] * XB + 8 * t);
                    lo = v[0]; hi = v[1];
                } else {
                    lo = q.x[m][2 * t]; hi = q.x[m][2 * t + 1];
                }
                if (INT) {  // even / odd k, like the weight codes
                    xe[m] = __builtin_amdgcn_perm(hi, lo, 0x06040200u);
                    xo[m] = __builtin_amdgcn_perm(hi, lo, 0x07050301u);
                } else {
                    const f32x2 a0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)lo, true);
                    const f32x2 a2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, false), a3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)hi, true);
                    xf[m][0] = a0[0]; xf[m][1] = a0[1]; xf[m][2] = a1[0]; xf[m][3] = a1[1];
                    xf[m][4] = a2[0]; xf[m][5] = a2[1]; xf[m][6] = a3[0]; xf[m][7] = a3[1];
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint16_t sb = (uint16_t)(q.s[c >> 1] >> (16 * (c & 1))), zb = (uint16_t)(q.z[c >> 1] >> (16 * (c & 1)));
                const float s = need_s ? TR::to_float(sb) : 1.f;
                const float z = need_z ? TR::to_float(zb) : scalar_zero;
                uint32_t ev, od;
                CD::run(q.w[c], t, ev, od);
                if (INT) {
                    const uint32_t zz = 0x01010101u * ((uint32_t)(int)(z * u13) & 0xFFu);
                    ev = ((ev | 0x80808080u) - zz) ^ 0x80808080u;  // bytewise q - z, no borrows between bytes
                    od = ((od | 0x80808080u) - zz) ^ 0x80808080u;
#pragma unroll
                    for (int m = 0; m < MB; ++m)
                        acci[m][c] = __builtin_amdgcn_sdot4((int)od, (int)xo[m], __builtin_amdgcn_sdot4((int)ev, (int)xe[m], acci[m][c], false), false);
                } else {
                    const float A = s, B = z * __builtin_fmaf(-u13, s, u4);  // w = fma(q, A, B): all five W_group_modes
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float lo = (float)((ev >> (8 * j)) & 0xFFu), hi = (float)((od >> (8 * j)) & 0xFFu);
                        f32x2 wv;
                        if constexpr (MB == 1) {
                            // ONE row = the reference's GEMV family: its dot product is sum(a.to(acc) * b.to(acc)) (dot_prod_mode 0,
                            // gemv_revsplitK_kernels.py:331-332, fp32 accumulation for fp8 inputs) — the dequantised weight stays in the
                            // METADATA type, it is NOT rounded to e4m3 there.  (Rounds 2-3 rounded it like the GEMM family does: 2.9 % off
                            // the reference's own M = 1 output on the MI355X, tests/golden/fullsize_ref_r4.npz a8w4_fp8dyn_m1.)
                            wv[0] = TR::to_float(TR::from_float(__builtin_fmaf(lo, A, B)));
                            wv[1] = TR::to_float(TR::from_float(__builtin_fmaf(hi, A, B)));
                        } else {
                            // 2 .. 4 rows = the reference's GEMM_SPLITK family: `b.to(a.dtype)` before tl.dot (gemm_splitK_kernels.py) —
                            // the weight as the e4m3 value the reference multiplies
                            const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_fmaf(lo, A, B), __builtin_fmaf(hi, A, B), 0, false);
                            wv = __builtin_amdgcn_cvt_pk_f32_fp8(pk, false);
                        }
#pragma unroll
                        for (int m = 0; m < MB; ++m) {
                            accf[m][c] = __builtin_fmaf(xf[m][2 * j], wv[0], accf[m][c]);
                            accf[m][c] = __builtin_fmaf(xf[m][2 * j + 1], wv[1], accf[m][c]);
                        }
                    }
                }
            }
        }
    };

    // two requests in flight per wave
    Req qa, qb;
    qa.s = qa.z = qb.s = qb.z = (u32x2){0, 0};
    int g = wave;
    if (g < ngroups) request(qa, g);
    float sx_row = 1.f;
    if constexpr (FQ) {
        if (g + NW < ngroups) request(qb, g + NW);  // both of the wave's first requests are out before x is touched
        float* wmax = (float*)(xq_lds + p.K);
        const uint16_t* xr = (const uint16_t*)p.x;
        float amax = 0.f;
        for (int k = tid * 8; k < p.K; k += 1024 * 8) {
            const u32x4 v = *(const u32x4*)(xr + k);
#pragma unroll
            for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(TR::to_float((uint16_t)(v[e >> 1] >> (16 * (e & 1))))));
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        if (lane == 0) wmax[wave] = amax;
        __syncthreads();
        amax = wmax[lane & 15];
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        constexpr float qmin = INT ? -128.f : -448.f, qmax = INT ? 127.f : 448.f;
        sx_row = fmaxf(__fdiv_rn(amax, qmax), 1e-6f);
        for (int k = tid * 8; k < p.K; k += 1024 * 8) {
            const u32x4 v = *(const u32x4*)(xr + k);
            float tq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) tq[e] = fminf(fmaxf(__fdiv_rn(TR::to_float((uint16_t)(v[e >> 1] >> (16 * (e & 1)))), sx_row), qmin), qmax);
            uint32_t o[2] = {0u, 0u};
            if constexpr (INT) {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e >> 2] |= (uint32_t)(uint8_t)(int8_t)floorf(tq[e] + 0.5f) << (8 * (e & 3));
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int w = __builtin_amdgcn_cvt_pk_fp8_f32(tq[4 * h], tq[4 * h + 1], 0, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(tq[4 * h + 2], tq[4 * h + 3], w, true);
                    o[h] = (uint32_t)w;
                }
            }
            *(u32x2*)(xq_lds + k) = (u32x2){o[0], o[1]};
        }
        __syncthreads();
    }
    for (; g < ngroups; g += 2 * NW) {
        const bool more = g + NW < ngroups;
        if (more && !(FQ && g == wave)) request(qb, g + NW);
        consume(qa);
        if (more) {
            if (g + 2 * NW < ngroups) request(qa, g + 2 * NW);
            consume(qb);
        }
    }

    // ---- reduce: the 16 row lanes of a column group, then the 16 waves ------------------------------------------------------
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = INT ? (float)acci[m][c] : accf[m][c];  // int32 sums are exact in fp32 below 2^24
            if (INT) {
                int iv = acci[m][c];
#pragma unroll
                for (int off = 4; off < 64; off <<= 1) iv += __shfl_xor(iv, off);
                v = (float)iv;
            } else {
#pragma unroll
                for (int off = 4; off < 64; off <<= 1) v += __shfl_xor(v, off);
            }
            if (r == 0) red[wave][m][4 * c4 + c] = v;
        }
    __syncthreads();
    if (tid < MB * 16) {
        const int m = tid >> 4, c = tid & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][m][c];
        if constexpr (FQ) {  // same arithmetic as epilogue_scale(): acc * (s_x * s_w[n]) for mode 3, acc * s_x for mode 2
            const int64_t n = (int64_t)blockIdx.x * 16 + c;
            if (p.epi.c_mode == 3) v *= sx_row * load_as_float(p.epi.scales_w, n, p.epi.meta_dt);
            else v *= sx_row;
            store_from_float(p.epi.out, n * p.epi.stride_on, p.epi.out_dt, v);
        } else {
            if (m < p.M) epilogue_store(p.epi, v, m, (int64_t)blockIdx.x * 16 + c);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
typedef void (*a8wn_kernel_fn)(const WnParams);
template <typename Tag, int NBITS, int XDT>
static const void* a8wn_pick_mb(int mb) {
    a8wn_kernel_fn f = nullptr;
    switch (mb) {
        case -1: f = gemv_a8wn_kernel<Tag, NBITS, XDT, 1, true>; break;  // one row, activation quantiser inside
        case 1: f = gemv_a8wn_kernel<Tag, NBITS, XDT, 1>; break;
        case 2: f = gemv_a8wn_kernel<Tag, NBITS, XDT, 2>; break;
        case 4: f = gemv_a8wn_kernel<Tag, NBITS, XDT, 4>; break;
        default: break;
    }
    return (const void*)f;
}
template <typename Tag>
static const void* a8wn_pick(int nbits, int xdt, int mb) {
    if (xdt == GEMLITE_DT_FP8E4) return nbits == 4 ? a8wn_pick_mb<Tag, 4, GEMLITE_DT_FP8E4>(mb) : a8wn_pick_mb<Tag, 2, GEMLITE_DT_FP8E4>(mb);
    return nbits == 4 ? a8wn_pick_mb<Tag, 4, GEMLITE_DT_INT8>(mb) : a8wn_pick_mb<Tag, 2, GEMLITE_DT_INT8>(mb);
}

// fq: `a` describes the call AS THE MATMUL SEES IT (8-bit input dtype, placeholder x / scales_x); the launch gets the raw 16-bit row
bool plan_gemv_a8wn(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp, bool fq) {
    if (fq && (a.M != 1 || a.K > 61440 || a.K % 8 != 0)) return false;
    const int nbits = a.W_nbits;
    if (nbits != 4 && nbits != 2) return false;
    if (a.M < 1 || a.M > 4) return false;
    if (a.input_dtype != GEMLITE_DT_FP8E4 && a.input_dtype != GEMLITE_DT_INT8) return false;
    if (a.output_dtype != GEMLITE_DT_FP16 && a.output_dtype != GEMLITE_DT_BF16) return false;
    const bool loop_s = a.W_group_mode >= 2;
    const bool has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (loop_s && a.meta_dtype != a.output_dtype) return false;  // 16-bit metadata of the output's type inside the loop
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != a.output_dtype) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    if (a.input_dtype == GEMLITE_DT_INT8 && (a.W_group_mode >= 2 || (has_z && !a.zero_is_scalar))) return false;  // integer codes only
    const int e = 32 / nbits;
    if (a.N % 16 != 0 || a.N / 16 < 64) return false;        // one block per 16 columns, no K split across blocks
    if (a.K % (16 * e) != 0) return false;                    // whole 16-row groups
    if (p.gs_shift < 0 || p.group_size < e) return false;
    if (((uintptr_t)a.w_q % 16) != 0 || (a.stride_wk % 4) != 0) return false;
    if (((uintptr_t)a.x % 16) != 0 || (a.stride_xm % 16) != 0) return false;
    if ((loop_s && ((uintptr_t)a.scales % 8) != 0) || (has_z && !a.zero_is_scalar && ((uintptr_t)a.zeros % 8) != 0)) return false;
    if ((loop_s || (has_z && !a.zero_is_scalar)) && p.stride_meta_g % 4 != 0) return false;
    const int mb = fq ? -1 : (a.M == 1 ? 1 : (a.M == 2 ? 2 : 4));
    const void* fn = a.output_dtype == GEMLITE_DT_FP16 ? a8wn_pick<half_tag>(nbits, a.input_dtype, mb)
                                                       : a8wn_pick<bf16_tag>(nbits, a.input_dtype, mb);
    if (!fn) return false;
    p.splitk = 1;
    p.rows_per_slice = (int)(a.K / e);
    lp.fn = fn;
    lp.name = fq ? (nbits == 4 ? "gemv_a8w4_fused_quant_kernel<tile16,16w>" : "gemv_a8w2_fused_quant_kernel<tile16,16w>")
                 : (nbits == 4 ? "gemv_a8w4_kernel<tile16,16w>" : "gemv_a8w2_kernel<tile16,16w>");
    lp.grid = dim3((unsigned)(a.N / 16), 1, 1);
    lp.block = dim3(1024, 1, 1);
    lp.lds_bytes = fq ? (size_t)a.K + 64 : 0;
    lp.slab_bytes = 0;
    lp.ws_bytes = 0;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// a8wn_rows_kernel (round 4): 2 .. 64 rows of the same layers (A8W4 / A8W2 fp8-dynamic, BitNet int8-dynamic; reference: gemm_splitK_INT_kernel,
// gemlite/triton_kernels/gemm_splitK_kernels.py:277-450) — they ran on the 32-row tile of the 8-wave MFMA kernel, one eighth of the chip's
// blocks, 11 us at 4096^2 whatever M.  The rows shape: block = 16 output columns x all of K, 8 waves dealing the chunks; lane (c = lane & 15,
// q = lane >> 4) owns column n0 + c and KL consecutive k of every chunk (chunk = 4 KL k):
//   fp8 activations: KL = 32 / NBITS = the k of ONE packed word.  The word is dequantised weight by weight — fma(q, s, -z s) in fp32, ONE
//     rounding to e4m3 with the hardware converter: the reference's `b.to(a.dtype)` (gemm_kernels.py:384) — and its 8 (4-bit) or 2 x 8
//     (2-bit) values ARE the B fragment(s) of v_mfma_f32_16x16x32_fp8_fp8; A = 8 bytes of x row c (+ 16 t).
//   int8 activations (integer codes minus an integer zero): KL = 16 = one 2-bit word or two 4-bit words = the B fragment of
//     v_mfma_i32_16x16x64_i8, exact.
// Words are 4 bytes per lane, 64-byte row segments per lane group: adjacent tiles are placed on ONE XCD so that the two halves of a
// 128-byte line meet in one L2 (the pairing of gemv_w4_decode3_kernel).  (scale, zero) of the lane's group come as 2-byte loads per chunk.
// ---------------------------------------------------------------------------------------------------------------
template <typename Tag, int NBITS, int XDT, int MT>
__global__ __launch_bounds__(512) void a8wn_rows_kernel(const WnParams p) {
    using TR = F16Traits<Tag>;
    using CD = a8::Codes<NBITS>;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    constexpr bool INT = XDT == GEMLITE_DT_INT8;
    constexpr int E = 32 / NBITS;            // k per packed word
    constexpr int KL = INT ? 16 : E;          // k per lane and chunk
    constexpr int WPL = KL / E;               // packed words per lane and chunk (1, or 2 for int8 x 4-bit)
    constexpr int NF = INT ? 1 : KL / 8;      // MFMAs per chunk and row tile
    constexpr int CK = 4 * KL;                // k per chunk
    __shared__ __attribute__((aligned(16))) float red[MT][8][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    int tile = blockIdx.x;
    if ((gridDim.x & 15) == 0) {  // adjacent half-line tiles on one XCD (speed only; any mapping is correct)
        const int xcd = tile & 7, idx = tile >> 3;
        tile = (((idx >> 1) << 3) + xcd) * 2 + (idx & 1);
    }
    const int n = tile * 16 + c;
    const int nchunks = p.K / CK;
    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    const float scalar_zero = p.zero_is_scalar ? (float)((const int32_t*)p.zeros)[0] : 0.f;
    const float u13 = (p.w_mode == 1 || p.w_mode == 3) ? 1.f : 0.f, u4 = p.w_mode == 4 ? 1.f : 0.f;
    const int rows = p.K / E;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)(((int64_t)(rows - 1) * p.stride_wk + p.N) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, (short)0, (int)((int64_t)(p.M - 1) * p.stride_xm + p.K), 0x00020000);
    const int meta_rows = p.gs_shift >= 31 ? 1 : (p.K >> p.gs_shift);
    const int meta_bytes = (int)(((int64_t)(meta_rows - 1) * p.stride_meta_g + p.N) * 2);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(need_s ? p.scales : (const void*)p.w), (short)0, need_s ? meta_bytes : 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void*)(need_z ? p.zeros : (const void*)p.w), (short)0, need_z ? meta_bytes : 4, 0x00020000);
    const uint32_t wvoff = (uint32_t)(((int64_t)(q * WPL) * p.stride_wk + n) * 4);
    uint32_t xvoff[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) xvoff[t] = c + 16 * t < p.M ? (uint32_t)((int64_t)(c + 16 * t) * p.stride_xm + q * KL) : 0x80000000u;  // rows >= M: zeros
    typedef typename std::conditional<INT, i32x4, f32x4>::type acc_t;
    acc_t acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = acc_t{0, 0, 0, 0};
    constexpr int D = MT == 1 ? 8 : (MT == 2 ? 6 : 4);
    struct Chunk { uint32_t w[WPL]; uint32_t s, z; u32x4 x[MT]; };
    Chunk ring[D];
    const int mine = (nchunks - wave + 7) >> 3;
    auto load = [&](int slot, int i) __attribute__((always_inline)) {
        const int ch = wave + 8 * i;
        Chunk& k = ring[slot];
        const uint32_t wo = (uint32_t)__builtin_amdgcn_readfirstlane((int)((int64_t)ch * (CK / E) * p.stride_wk * 4));
#pragma unroll
        for (int i2 = 0; i2 < WPL; ++i2) k.w[i2] = __builtin_amdgcn_raw_buffer_load_b32(rsW, wvoff + (uint32_t)(i2 * (int)p.stride_wk * 4), wo, 2);
        const uint32_t mo = (uint32_t)((((int64_t)ch * CK + q * KL) >> p.gs_shift) * p.stride_meta_g + n) * 2u;
        k.s = need_s ? (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsS, mo, 0, 0) : 0u;
        k.z = need_z ? (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsZ, mo, 0, 0) : 0u;
        const uint32_t xo = (uint32_t)__builtin_amdgcn_readfirstlane(ch * CK);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            if constexpr (KL == 16) k.x[t] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xvoff[t], xo, 0);
            else {
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsX, xvoff[t], xo, 0);
                k.x[t] = (u32x4){v[0], v[1], 0u, 0u};
            }
        }
    };
    auto mma = [&](int slot) __attribute__((always_inline)) {
        const Chunk& k = ring[slot];
        const float s = need_s ? TR::to_float((uint16_t)k.s) : 1.f;
        const float z = need_z ? TR::to_float((uint16_t)k.z) : scalar_zero;
        if constexpr (INT) {
            const uint32_t zz = 0x01010101u * ((uint32_t)(int)(z * u13) & 0xFFu);
            uint32_t b[4];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {  // the lane's 16 k = 2 slices of 8: slice sl of the 2-bit word, or word sl (4-bit)
                uint32_t ev, od;
                CD::run(k.w[WPL == 2 ? sl : 0], WPL == 2 ? 0 : sl, ev, od);
                ev = ((ev | 0x80808080u) - zz) ^ 0x80808080u;  // bytewise q - z, no borrows between bytes
                od = ((od | 0x80808080u) - zz) ^ 0x80808080u;
                b[2 * sl] = __builtin_amdgcn_perm(od, ev, 0x05010400u);      // k 0 .. 3 of the slice in natural order
                b[2 * sl + 1] = __builtin_amdgcn_perm(od, ev, 0x07030602u);  // k 4 .. 7
            }
            const i32x4 bv = {(int)b[0], (int)b[1], (int)b[2], (int)b[3]};
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, k.x[t]), bv, acc[t], 0, 0, 0);
        } else {
            const float A = s, B = z * __builtin_fmaf(-u13, s, u4);  // w = fma(q, A, B): all five W_group_modes
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                uint32_t ev, od;
                CD::run(k.w[0], f, ev, od);
                int lo = 0, hi = 0;
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_fmaf((float)(ev & 0xFFu), A, B), __builtin_fmaf((float)(od & 0xFFu), A, B), lo, false);
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_fmaf((float)((ev >> 8) & 0xFFu), A, B), __builtin_fmaf((float)((od >> 8) & 0xFFu), A, B), lo, true);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_fmaf((float)((ev >> 16) & 0xFFu), A, B), __builtin_fmaf((float)((od >> 16) & 0xFFu), A, B), hi, false);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_fmaf((float)(ev >> 24), A, B), __builtin_fmaf((float)(od >> 24), A, B), hi, true);
                const long bw = (long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const long aw = (long)(((uint64_t)k.x[t][2 * f + 1] << 32) | k.x[t][2 * f]);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(aw, bw, acc[t], 0, 0, 0);
                }
            }
        }
    };
#pragma unroll
    for (int j = 0; j < D; ++j)
        if (j < mine) load(j, j);
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
    for (int t = 0; t < MT; ++t) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (float)acc[t][r];  // int32 sums: exact in fp32 below 2^24 (K <= 65536 codes of |q - z| <= 255 ... checked by the planner)
        *(f32x4*)&red[t][wave][lane][0] = v;
    }
    __syncthreads();
    for (int u = tid; u < MT * 256; u += 512) {
        const int t = u >> 8, l = u & 63, r = (u >> 6) & 3;
        const int m = 16 * t + 4 * (l >> 4) + r;  // C fragment of a 16 x 16 MFMA: column lane & 15, rows 4 (lane >> 4) + r
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[t][w][l][r];
        if (m < p.M) epilogue_store(p.epi, v, m, (int64_t)tile * 16 + (l & 15));
    }
}

// 2 <= M <= 64; the conditions of plan_gemv_a8wn plus whole chunks and the x re-read budget of the rows kernels
bool plan_a8wn_rows(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    const int nbits = a.W_nbits;
    if (nbits != 4 && nbits != 2) return false;
    if (a.M < 2 || a.M > 64) return false;
    if (a.input_dtype != GEMLITE_DT_FP8E4 && a.input_dtype != GEMLITE_DT_INT8) return false;
    if (a.output_dtype != GEMLITE_DT_FP16 && a.output_dtype != GEMLITE_DT_BF16) return false;
    const bool isint = a.input_dtype == GEMLITE_DT_INT8;
    const bool loop_s = a.W_group_mode >= 2;
    const bool has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (loop_s && a.meta_dtype != a.output_dtype) return false;
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != a.output_dtype) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    if (isint && (a.W_group_mode >= 2 || (has_z && !a.zero_is_scalar))) return false;  // integer codes only
    const int e = 32 / nbits, kl = isint ? 16 : e, ck = 4 * kl;
    if (a.N % 16 != 0 || a.K % ck != 0 || a.K > (isint ? 65536 : (1 << 30))) return false;
    if (p.gs_shift < 0 || p.group_size < kl || p.group_size % kl != 0) return false;  // a lane's k of a chunk lie inside one group
    if (((uintptr_t)a.w_q % 4) != 0 || ((uintptr_t)a.x % 16) != 0 || (a.stride_xm % 16) != 0) return false;
    if ((loop_s && ((uintptr_t)a.scales % 2) != 0) || (has_z && !a.zero_is_scalar && ((uintptr_t)a.zeros % 2) != 0)) return false;
    if (((int64_t)(a.K / e) * a.stride_wk + a.N) * 4 >= (1ll << 31) || (int64_t)a.M * a.stride_xm + a.K >= (1ll << 31)) return false;
    if (((int64_t)(a.K / (p.group_size > 0 ? p.group_size : a.K)) * p.stride_meta_g + a.N) * 2 >= (1ll << 31)) return false;
    const int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
    if (mt > 1 && a.tuning[0] != 4 && (int64_t)a.M * a.K * (a.N / 16) > (88ll << 20)) return false;
    // round 4 (profiles/r04/probe_rows_vs_tiles*.log): against the 32- / 64-row tiles of the 8-wave kernel the crossover sits at
    // M N K ~ 600 M (4096^2: M = 36; 8192^2: 8 — M = 16 there: 27.5 vs 22.0 us; 4096 x 14336: 11)
    if (a.M >= 2 && a.tuning[0] != 4 && a.N % 128 == 0 && a.K % 256 == 0 && (int64_t)a.M * a.N * a.K > 600000000ll) return false;
    typedef void (*fn_t)(const WnParams);
    fn_t fn = nullptr;
    auto pick = [&](auto tag, auto nb, auto xd) -> fn_t {
        using T = decltype(tag);
        constexpr int NB = decltype(nb)::value, XD = decltype(xd)::value;
        return mt == 1 ? a8wn_rows_kernel<T, NB, XD, 1> : (mt == 2 ? a8wn_rows_kernel<T, NB, XD, 2> : a8wn_rows_kernel<T, NB, XD, 4>);
    };
    typedef std::integral_constant<int, 4> N4;
    typedef std::integral_constant<int, 2> N2;
    typedef std::integral_constant<int, GEMLITE_DT_FP8E4> XF;
    typedef std::integral_constant<int, GEMLITE_DT_INT8> XI;
    const bool f16 = a.output_dtype == GEMLITE_DT_FP16;
    if (isint) fn = nbits == 4 ? (f16 ? pick(half_tag{}, N4{}, XI{}) : pick(bf16_tag{}, N4{}, XI{})) : (f16 ? pick(half_tag{}, N2{}, XI{}) : pick(bf16_tag{}, N2{}, XI{}));
    else fn = nbits == 4 ? (f16 ? pick(half_tag{}, N4{}, XF{}) : pick(bf16_tag{}, N4{}, XF{})) : (f16 ? pick(half_tag{}, N2{}, XF{}) : pick(bf16_tag{}, N2{}, XF{}));
    p.splitk = 1;
    p.rows_per_slice = (int)(a.K / e);
    lp.fn = (const void*)fn;
    static const char* names[2][3] = {{"a8w4_rows_kernel<16x16>", "a8w4_rows_kernel<32x16>", "a8w4_rows_kernel<64x16>"},
                                      {"a8w2_rows_kernel<16x16>", "a8w2_rows_kernel<32x16>", "a8w2_rows_kernel<64x16>"}};
    lp.name = names[nbits == 4 ? 0 : 1][mt == 1 ? 0 : (mt == 2 ? 1 : 2)];
    lp.grid = dim3((unsigned)(a.N / 16), 1, 1);
    lp.block = dim3(512, 1, 1);
    lp.lds_bytes = 0;
    lp.slab_bytes = 0;
    lp.ws_bytes = 0;
    return true;
}

}  // namespace gl
