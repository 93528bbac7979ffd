// gemm_wn_stream.hip — fused unpack + group-scale + MFMA matmul for packed low-bit weights in the weight-
// streaming regime: decode and small batches (1 <= M <= 64; larger M through gridDim.z row tiles).
// Replaces gemm_splitK_INT_kernel (gemlite/triton_kernels/gemm_splitK_kernels.py:277-450), and for M = 1 the
// GEMV family as well when it is faster (the dispatcher decides).
//
// Idea: the K-packed layout makes one packed int32 word exactly one lane's B fragment of
// v_mfma_f32_16x16x32_{f16,bf16} (8 consecutive k of ONE column), and inside one quantisation group
//        sum_k x_k (s q_k + z') = s * sum_k x_k q_k + z' * sum_k x_k ,
// so the matrix core can multiply x by the RAW integer codes and the scale / zero are applied once per group
// to the fp32 accumulators.  Per packed weight that leaves one AND (plus one shift per byte window):
//   * fp16: the masked bits are used directly as fp16 SUBNORMALS (q * 2^(4i) * 2^-24, exact; gfx950 MFMA and
//     v_dot2 keep fp16 subnormals — scripts/ubench/probe_*.hip); the 2^24 is folded into the scale;
//   * bf16: 7 mantissa bits and no usable subnormal range, so (bits | 0x4300) = 128 + q and the 128 * sum(x)
//     term is removed together with the zero-point term.
//   x is staged in LDS pair-permuted and pre-scaled by 2^-(4i) per window position (exact), together with
//   per-32-k partial sums of x (true and as-stored) that the group epilogue needs.
//
// Mapping (CDNA4): block = 4 waves, 64-column tile x BM = 16*MT rows x one K slice; a lane loads 16 bytes =
// 4 adjacent columns x 8 k (4 packed rows x 256 B per wave-level load -> 4 MFMAs with column sets {4n + j});
// K is walked in pieces of 512 (wave w owns k in [128w, 128w+128): one group at group_size 128); weights and
// their metadata for the next piece are requested before the current piece is consumed; the 4 waves' partial
// sums are combined in LDS and K slices with the write-through slab + ticket protocol (gl_common.h).
#include "gl_common.h"

namespace gl {

template <typename Tag>
__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma16<half_tag>(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16<bf16_tag>(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
}

constexpr int PIECE_K = 512;             // k per block iteration
constexpr int XPITCH = PIECE_K / 2 + 4;  // LDS dwords per x row (16-byte pad against bank conflicts)
constexpr int NS32 = PIECE_K / 32;       // 32-k partial sums of x per row and piece

// window geometry shared with the GEMV: how many bit fields fit the mantissa next to each other
template <typename Tag, int NBITS>
struct StreamWin {
    static constexpr bool SUBN = F16Traits<Tag>::DT == GEMLITE_DT_FP16;  // subnormal unpack (no magic OR)
    static constexpr int MANT = SUBN ? 10 : 7;
    static constexpr int HALF = 16 / NBITS;
    static constexpr int fit() {
        int wp = 1;
        while (wp * 2 <= HALF && (((1 << NBITS) - 1) << (NBITS * (wp * 2 - 1))) < (1 << MANT)) wp *= 2;
        return wp;
    }
    static constexpr int WP = fit();
};

// SPG = MFMA k-steps (32 k each) per quantisation group inside a wave's 128-k span: 4 (group >= 128), 2, 1
template <typename Tag, int NBITS, int MT, int SPG>
__global__ __launch_bounds__(256, (MT == 1 ? 2 : 1)) void gemm_wn_stream_kernel(const WnParams p) {
    using TR = F16Traits<Tag>;
    using SW = StreamWin<Tag, NBITS>;
    constexpr bool SUBN = SW::SUBN;
    constexpr int E = 32 / NBITS, HALF = E / 2, WP = SW::WP;
    constexpr int NF = E / 8;                   // 32-k MFMA steps fed by one 4-row wave load
    constexpr int ROWS_WP = (PIECE_K / 4) / E;  // packed rows per wave per piece
    constexpr int U = ROWS_WP / 4;              // 16-byte loads per lane per piece (U * NF == 4 k-steps)
    constexpr int NGRP = 4 / SPG;               // groups inside the wave's span
    constexpr int BM = 16 * MT;
    static_assert(E >= 8 && U >= 1 && U * NF == 4, "8-bit words take another path");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool DBUF = MT <= 2;                       // two x buffers: one barrier per piece
    constexpr int NBUF = DBUF ? 2 : 1;
    constexpr int XBUF_DW = BM * XPITCH + 2 * BM * NS32;  // dwords per buffer: x image + true / stored 32-k sums
    uint32_t* xbuf = (uint32_t*)smem;                    // [NBUF][XBUF_DW]
    constexpr size_t XS_BYTES = (size_t)NBUF * XBUF_DW * 4;
    constexpr size_t RED_BYTES = (size_t)4 * BM * 64 * 4;
    unsigned* flag = (unsigned*)(smem + (XS_BYTES > RED_BYTES ? XS_BYTES : RED_BYTES));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x, slice = blockIdx.y, mtile = blockIdx.z;
    const int n0 = tile * 64 + c * 4;
    const int m0 = mtile * BM;

    const int rows_slice = p.rows_per_slice;  // multiple of 4*ROWS_WP
    const int row_s0 = slice * rows_slice;
    const int npieces = rows_slice / (4 * ROWS_WP);  // 1, or even (planner)

    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    const uint16_t* sp = need_s ? (const uint16_t*)p.scales : (const uint16_t*)p.w;  // dummy source keeps the
    const uint16_t* zp = need_z ? (const uint16_t*)p.zeros : (const uint16_t*)p.w;   // loop free of branches
    const int64_t mstride = (need_s || need_z) ? p.stride_meta_g : 0;

    // lane's packed row inside a piece for load u: wave*ROWS_WP + 4u + g
    const uint32_t* wbase = p.w + (int64_t)(row_s0 + wave * ROWS_WP + g) * p.stride_wk + n0;
    struct Piece { u32x4 w[U]; u32x2 s[NGRP], z[NGRP]; };
    auto load_piece = [&](Piece& pc, int piece) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            pc.w[u] = *(const u32x4*)(wbase + (int64_t)(piece * 4 * ROWS_WP + 4 * u) * p.stride_wk);
        const int k_w = (row_s0 + piece * 4 * ROWS_WP + wave * ROWS_WP) * E;  // first k of the span
#pragma unroll
        for (int q = 0; q < NGRP; ++q) {
            const int64_t grp = group_of(k_w + q * 32 * SPG, p.gs_shift);
            pc.s[q] = *(const u32x2*)(sp + grp * mstride + n0);
            pc.z[q] = *(const u32x2*)(zp + grp * mstride + n0);
        }
    };

    f32x4 tot[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) tot[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (read into a scalar register and waited for HERE: a vector load left pending into the K loop is answered with `s_waitcnt vmcnt(0)` at its
    //  first use in EVERY iteration — the loop-head drain of rounds 3-5, see gemv_wn.hip)
    float scalar_zero = 0.f;
    if (p.zero_is_scalar) scalar_zero = (float)__builtin_amdgcn_readfirstlane(((const int32_t*)p.zeros)[0]);
    const float bz = (p.w_mode == 1 || p.w_mode == 3) ? -1.f : (p.w_mode == 4 ? 1.f : 0.f);
    const bool b_times_s = p.w_mode == 3;
    constexpr float QSCALE = SUBN ? 16777216.0f : 1.0f;  // 2^24: undo the subnormal interpretation
    constexpr float OFF = SUBN ? 0.0f : TR::OFF;
    uint32_t wmask[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) wmask[i] = (((1u << NBITS) - 1u) * 0x00010001u) << (NBITS * i);

    // ---- x staging, split so the global loads of piece i+1 fly during the MFMAs of piece i ------------------
    // task = (row r, 32-k span): BM*16 tasks = MT per thread.  fetch: 64 bytes -> registers; put: pair-permute,
    // pre-scale, 32-k partial sums -> LDS.  Rows >= M are zero-filled once (they never change).
    constexpr int XT = MT;
    struct XRegs { u32x4 v[XT][4]; };
    const uint16_t* xg = (const uint16_t*)p.x;
    auto fetch_x = [&](XRegs& xr, int piece) {
        const int64_t k0 = (int64_t)(row_s0 + piece * 4 * ROWS_WP) * E;
#pragma unroll
        for (int t = 0; t < XT; ++t) {
            const int idx = tid + t * 256, r = idx / NS32, spn = idx - r * NS32;
            if (m0 + r < p.M) {
                const uint16_t* src = xg + (int64_t)(m0 + r) * p.stride_xm + k0 + (int64_t)spn * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) xr.v[t][q] = *(const u32x4*)(src + 8 * q);
            }
        }
    };
    auto put_x = [&](const XRegs& xr, int buf) {
        constexpr int WPS = 32 / E;  // packed words per 32-k span
        uint32_t* xs_w = xbuf + buf * XBUF_DW;
        float* xsum_t_w = (float*)(xs_w + BM * XPITCH);
        float* xsum_s_w = xsum_t_w + BM * NS32;
#pragma unroll
        for (int t = 0; t < XT; ++t) {
            const int idx = tid + t * 256, r = idx / NS32, spn = idx - r * NS32;
            if (m0 + r < p.M) {
                uint16_t v[32];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int z = 0; z < 4; ++z) {
                        v[8 * q + 2 * z] = (uint16_t)(xr.v[t][q][z] & 0xFFFFu);
                        v[8 * q + 2 * z + 1] = (uint16_t)(xr.v[t][q][z] >> 16);
                    }
                float sum_t = 0.f, sum_s = 0.f;
                uint32_t outv[16];
#pragma unroll
                for (int wdi = 0; wdi < WPS; ++wdi)
#pragma unroll
                    for (int d = 0; d < HALF; ++d) {
                        const float lo = TR::to_float(v[wdi * E + d]), hi = TR::to_float(v[wdi * E + d + HALF]);
                        sum_t += lo + hi;
                        if constexpr (WP > 1) {
                            const float sc = __builtin_bit_cast(float, (uint32_t)(127 - NBITS * (d % WP)) << 23);
                            const uint16_t slo = TR::from_float(lo * sc), shi = TR::from_float(hi * sc);
                            sum_s += TR::to_float(slo) + TR::to_float(shi);
                            outv[wdi * HALF + d] = (uint32_t)slo | ((uint32_t)shi << 16);
                        } else {
                            outv[wdi * HALF + d] = (uint32_t)v[wdi * E + d] | ((uint32_t)v[wdi * E + d + HALF] << 16);
                        }
                    }
                if constexpr (WP == 1) sum_s = sum_t;
                uint32_t* dst = xs_w + r * XPITCH + spn * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(u32x4*)(dst + 4 * q) = (u32x4){outv[4 * q], outv[4 * q + 1], outv[4 * q + 2], outv[4 * q + 3]};
                xsum_t_w[idx] = sum_t;
                xsum_s_w[idx] = sum_s;
            }
        }
    };
    // zero rows (m >= M) of every buffer once
    for (int b = 0; b < NBUF; ++b) {
        uint32_t* xs_w = xbuf + b * XBUF_DW;
        for (int idx = tid; idx < BM * NS32; idx += 256) {
            const int r = idx / NS32, spn = idx - r * NS32;
            if (m0 + r >= p.M) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *(u32x4*)(xs_w + r * XPITCH + spn * 16 + 4 * q) = (u32x4){0u, 0u, 0u, 0u};
                ((float*)(xs_w + BM * XPITCH))[idx] = 0.f;
                ((float*)(xs_w + BM * XPITCH))[BM * NS32 + idx] = 0.f;
            }
        }
    }

    auto unpack4 = [](const u32x2 raw) -> f32x4 {
        f32x4 r;
        const uint32_t v0 = raw[0], v1 = raw[1];
        r[0] = TR::to_float((uint16_t)(v0 & 0xFFFFu)); r[1] = TR::to_float((uint16_t)(v0 >> 16));
        r[2] = TR::to_float((uint16_t)(v1 & 0xFFFFu)); r[3] = TR::to_float((uint16_t)(v1 >> 16));
        return r;
    };

    auto compute = [&](const Piece& pc, int buf) {
        const uint32_t* xs = xbuf + buf * XBUF_DW;
        const float* xsum_t = (const float*)(xs + BM * XPITCH);
        const float* xsum_s = xsum_t + BM * NS32;
        f32x4 acc[MT][4];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // all A fragments of the piece are requested up front (LDS latency overlaps the unpack VALU work)
        u32x4 afr[U][NF][MT];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    afr[u][f][t] = *(const u32x4*)(xs + (t * 16 + c) * XPITCH + (wave * ROWS_WP + 4 * u + g) * HALF + 4 * f);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int ks = u * NF + f;  // MFMA k-step 0..3 inside the wave's 128-k span
                u32x4 bfrag[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) {
                        const int d = 4 * f + dd, win = d / WP, wi = d % WP;
                        uint32_t h = (pc.w[u][j] >> (NBITS * WP * win)) & wmask[wi];
                        if constexpr (!SUBN) h |= TR::MAGIC2;
                        bfrag[j][dd] = h;
                    }
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][j] = mfma16<Tag>(afr[u][f][t], bfrag[j], acc[t][j]);
                if ((ks + 1) % SPG == 0) {  // end of a quantisation group: fold scale / zero into the totals
                    const int q = ks / SPG;
                    f32x4 s = unpack4(pc.s[q]), z = unpack4(pc.z[q]);
                    if (!need_s) s = (f32x4){1.f, 1.f, 1.f, 1.f};
                    if (!need_z) z = (f32x4){scalar_zero, scalar_zero, scalar_zero, scalar_zero};
                    const int s32 = wave * 4 + (ks + 1 - SPG);  // first 32-k partial sum of this group
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        float xt[4], xst[4];  // per output row of this lane: sum of x, sum of stored x
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int m = t * 16 + 4 * g + rg;
                            float a_t = 0.f, a_s = 0.f;
#pragma unroll
                            for (int e2 = 0; e2 < SPG; ++e2) {
                                a_t += xsum_t[m * NS32 + s32 + e2];
                                if constexpr (!SUBN) a_s += xsum_s[m * NS32 + s32 + e2];
                            }
                            xt[rg] = a_t;
                            xst[rg] = a_s;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float a = s[j] * QSCALE;
                            const float b = bz * z[j] * (b_times_s ? s[j] : 1.f);
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) {
                                float v = acc[t][j][rg];
                                if constexpr (!SUBN) v -= OFF * xst[rg];
                                tot[t][j][rg] += a * v + b * xt[rg];
                                acc[t][j][rg] = 0.f;
                            }
                        }
                    }
                }
            }
        }
    };

    Piece A, B;
    XRegs X;
    fetch_x(X, 0);  // x first: its (L2) latency must not queue behind the weight stream
    load_piece(A, 0);
    if (npieces > 1) load_piece(B, 1);
    put_x(X, 0);
    __syncthreads();
    if constexpr (DBUF) {
        int pc = 0;
        for (; pc + 2 <= npieces; pc += 2) {  // the tail re-requests the last piece (harmless)
            fetch_x(X, pc + 1);
            compute(A, 0);
            load_piece(A, pc + 2 < npieces ? pc + 2 : npieces - 1);
            put_x(X, 1);
            __syncthreads();
            fetch_x(X, pc + 2 < npieces ? pc + 2 : npieces - 1);
            compute(B, 1);
            load_piece(B, pc + 3 < npieces ? pc + 3 : npieces - 1);
            put_x(X, 0);
            __syncthreads();
        }
        if (npieces & 1) compute(A, 0);
    } else {
        int pc = 0;
        for (; pc + 2 <= npieces; pc += 2) {
            fetch_x(X, pc + 1);
            compute(A, 0);
            load_piece(A, pc + 2 < npieces ? pc + 2 : npieces - 1);
            __syncthreads();
            put_x(X, 0);
            __syncthreads();
            fetch_x(X, pc + 2 < npieces ? pc + 2 : npieces - 1);
            compute(B, 0);
            load_piece(B, pc + 3 < npieces ? pc + 3 : npieces - 1);
            __syncthreads();
            put_x(X, 0);
            __syncthreads();
        }
        if (npieces & 1) compute(A, 0);
    }

    // ---- combine the 4 waves (disjoint K) through LDS ------------------------------------------------------
    __syncthreads();
    float* red = (float*)smem;  // [4][BM][64]
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = t * 16 + 4 * g + rg;
            const f32x4 v = {tot[t][0][rg], tot[t][1][rg], tot[t][2][rg], tot[t][3][rg]};
            *(f32x4*)(red + ((wave * BM + row) * 64 + 4 * c)) = v;
        }
    __syncthreads();
    constexpr int NOUT = BM * 64, OPT = NOUT / 256;
    float part[OPT];
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * 256;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += red[w * NOUT + o];
        part[it] = v;
    }
    const int tile_lin = mtile * gridDim.x + tile;
    if (p.splitk == 1) {
#pragma unroll
        for (int it = 0; it < OPT; ++it) {
            const int o = tid + it * 256, m = m0 + (o >> 6);
            if (m < p.M) store_out_t<Tag>(p.epi, part[it], m, (int64_t)tile * 64 + (o & 63));
        }
        return;
    }
    float* slab = p.slabs + ((int64_t)tile_lin * p.splitk) * NOUT;
#pragma unroll
    for (int it = 0; it < OPT; ++it) {  // only rows that exist travel through the slabs
        const int o = tid + it * 256;
        if (m0 + (o >> 6) < p.M) slab_store(slab + (int64_t)slice * NOUT + o, part[it]);
    }
    if (!splitk_arrive_is_last(p.counters + tile_lin, p.splitk, flag)) return;
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * 256, m = m0 + (o >> 6);
        if (m < p.M) {
            float v = 0.f;
            for (int s = 0; s < p.splitk; ++s) v += slab_load(slab + (int64_t)s * NOUT + o);
            store_out_t<Tag>(p.epi, v, m, (int64_t)tile * 64 + (o & 63));
        }
    }
    if (tid == 0) splitk_reset(p.counters + tile_lin);
}

// ---------------------------------------------------------------------------------------------
template <typename Tag, int NBITS, int MT>
static const void* pick_spg(int spg) {
    switch (spg) {
        case 4: return (const void*)gemm_wn_stream_kernel<Tag, NBITS, MT, 4>;
        case 2: return (const void*)gemm_wn_stream_kernel<Tag, NBITS, MT, 2>;
        default: return (const void*)gemm_wn_stream_kernel<Tag, NBITS, MT, 1>;
    }
}
template <typename Tag, int NBITS>
static const void* pick_mt(int mt, int spg) {
    switch (mt) {
        case 1: return pick_spg<Tag, NBITS, 1>(spg);
        case 2: return pick_spg<Tag, NBITS, 2>(spg);
        default: return pick_spg<Tag, NBITS, 4>(spg);
    }
}
template <typename Tag>
static const void* pick_bits_s(int nbits, int mt, int spg) {
    switch (nbits) {
        case 1: return pick_mt<Tag, 1>(mt, spg);
        case 2: return pick_mt<Tag, 2>(mt, spg);
        case 4: return pick_mt<Tag, 4>(mt, spg);
        default: return nullptr;
    }
}

// tuning[1]: 0 auto | n force split-K n
bool plan_gemm_wn_stream(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    const int nbits = a.W_nbits;
    if (nbits != 1 && nbits != 2 && nbits != 4) return false;
    const int e = 32 / nbits;
    if (a.N % 64 != 0 || a.K % PIECE_K != 0) return false;
    if (a.output_dtype != a.input_dtype) return false;  // typed epilogue / metadata
    const bool uses_s = a.W_group_mode >= 2 || a.channel_scale_mode == 1 || a.channel_scale_mode == 3;
    const bool has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (uses_s && a.meta_dtype != a.input_dtype) return false;
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != a.input_dtype) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    if ((a.stride_xm * 2) % 16 != 0 || ((uintptr_t)a.x % 16) != 0) return false;  // 16-byte x loads
    // 16-byte weight loads and 8-byte metadata loads: sliced / offset views that break the alignment go to the
    // coverage kernel instead of faulting
    if (((uintptr_t)a.w_q % 16) != 0 || (a.stride_wk % 4) != 0) return false;
    if ((uses_s && ((uintptr_t)a.scales % 8) != 0) || (has_z && !a.zero_is_scalar && ((uintptr_t)a.zeros % 8) != 0)) return false;
    if (p.stride_meta_g % 4 != 0) return false;
    // a quantisation group must cover whole 32-k MFMA steps and divide or contain the wave's 128-k span
    const int64_t gs = p.group_size;
    int spg;
    if (gs % 128 == 0) spg = 4;
    else if (gs == 64) spg = 2;
    else if (gs == 32) spg = 1;
    else return false;
    if (gs < 4 * e) return false;  // a group must contain whole 4-row wave loads (their k-steps are interleaved)
    const int rows = (int)(a.K / e);
    const int piece_rows = PIECE_K / e;
    const int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
    const int bm = 16 * mt;
    const void* fn = a.input_dtype == GEMLITE_DT_FP16 ? pick_bits_s<half_tag>(nbits, mt, spg)
                                                       : pick_bits_s<bf16_tag>(nbits, mt, spg);
    if (!fn) return false;
    const int tiles = (int)(a.N / 64), mtiles = (int)((a.M + bm - 1) / bm);
    const int units = rows / piece_rows;
    auto ok = [&](int sk) { return sk >= 1 && units % sk == 0; };
    int splitk = 0;
    if (a.tuning[1] > 0) {
        if (!ok(a.tuning[1])) return false;
        splitk = a.tuning[1];
    } else {
        for (int sk = 1; sk <= units; sk *= 2) {
            if (!ok(sk)) continue;
            splitk = sk;
            if ((int64_t)tiles * mtiles * sk >= 256) break;
        }
        if (!splitk) return false;
    }
    const uint64_t ntl = (uint64_t)tiles * mtiles;
    if (splitk > 1 && ntl > (uint64_t)MAX_SPLITK_COUNTERS) return false;
    p.splitk = splitk;
    p.rows_per_slice = rows / splitk;
    lp.fn = fn;
    lp.name = "gemm_wn_stream_kernel";
    lp.grid = dim3(tiles, splitk, mtiles);
    lp.block = dim3(256, 1, 1);
    const size_t xs_b = (size_t)(mt <= 2 ? 2 : 1) * ((size_t)bm * XPITCH * 4 + (size_t)2 * bm * NS32 * 4), red_b = (size_t)4 * bm * 64 * 4;
    lp.lds_bytes = (xs_b > red_b ? xs_b : red_b) + 16;
    lp.slab_bytes = splitk > 1 ? ntl * splitk * bm * 64 * 4 : 0;
    lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

}  // namespace gl
