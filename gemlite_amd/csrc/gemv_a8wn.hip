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

}  // namespace gl
