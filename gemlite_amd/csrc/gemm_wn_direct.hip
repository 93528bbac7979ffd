// gemm_wn_direct.hip — fused unpack + group-scale + MFMA matmul for packed low-bit weights and a handful of
// activation rows (2 <= M <= 32; more rows through gridDim.z): the batched-decode regime where the weights are
// streamed once and nothing else should be on the critical path.  Replaces gemm_splitK_INT_kernel
// (gemlite/triton_kernels/gemm_splitK_kernels.py:277-450) for those M.
//
// Same arithmetic as gemm_wn_stream.hip (one packed int32 word == one lane's B fragment of
// v_mfma_f32_16x16x32, raw integer codes through the matrix core, scale / zero applied once per group to the
// fp32 accumulators), but with NO shared-memory staging and NO barrier before the final reduction:
//   * the A fragment of lane (m = lane & 15, kb = lane >> 4) is the 8 (E for narrower codes) activations
//     x[m][8 row .. 8 row + 7] that face the lane's own packed word; they come straight from global memory
//     (L2-resident after the first touch) with one 16-byte load, are pair-permuted with v_perm_b32 and
//     pre-scaled with v_pk_mul_f16 in registers;
//   * the per-group sum of x that the zero-point term needs is produced by the matrix core as well: one extra
//     MFMA per k-step against a constant B fragment (the inverse window scales), which lands in exactly the
//     accumulator layout of the product — no VALU work, no LDS;
//   * the tile width is a template parameter V (packed words per lane and load: 1, 2 or 4 -> 16, 32 or 64
//     columns), so that N / (16 V) alone fills the 256 CUs and K never has to be split across blocks for the
//     usual 4096..16384 shapes (the cross-block combine costs ~3 us on MI355X, most of a decode GEMV).
// The bytes in flight per wave are kept constant (two pieces of 16 dwords per lane) by walking 16 / V k-steps
// per piece.
#include "gl_common.h"

namespace gl {

template <typename Tag>
__device__ __forceinline__ f32x4 mfma16d(u32x4 a, u32x4 b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mfma16d<half_tag>(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mfma16d<bf16_tag>(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
}

// how many bit fields fit the mantissa next to each other (same geometry as the GEMV / streaming kernels)
template <typename Tag, int NBITS>
struct DirWin {
    static constexpr bool SUBN = F16Traits<Tag>::DT == GEMLITE_DT_FP16;  // fp16: masked bits used as subnormals
    static constexpr int MANT = SUBN ? 10 : 7;
    static constexpr int HALF = 16 / NBITS;
    static constexpr int fit() {
        int wp = 1;
        while (wp * 2 <= HALF && (((1 << NBITS) - 1) << (NBITS * (wp * 2 - 1))) < (1 << MANT)) wp *= 2;
        return wp;
    }
    // bf16 keeps one field per window: a second field would need a (missing) packed bf16 multiply for x
    static constexpr int WP = SUBN ? fit() : 1;
};

template <int V> struct DirVec;
template <> struct DirVec<1> { typedef uint32_t W; typedef uint16_t M; };
template <> struct DirVec<2> { typedef u32x2 W; typedef uint32_t M; };
template <> struct DirVec<4> { typedef u32x4 W; typedef u32x2 M; };

template <int V>
__device__ __forceinline__ uint32_t dir_word(const typename DirVec<V>::W& w, int j) {
    if constexpr (V == 1) return w; else return w[j];
}
template <int V>
__device__ __forceinline__ uint16_t dir_meta(const typename DirVec<V>::M& m, int j) {
    if constexpr (V == 1) return m;
    else if constexpr (V == 2) return (uint16_t)(m >> (16 * j));
    else return (uint16_t)(m[j >> 1] >> (16 * (j & 1)));
}

// SPG = MFMA k-steps (32 k each) between two applications of scale / zero: 4 (group >= 128), 2 (64), 1 (32)
// NW = waves per block (round 3): 4 = one wave per SIMD with two pieces of K in flight (round 1); 8 = two per SIMD, each with half
// the K range — at 4096^2 every wave then has its WHOLE K range in flight from the start and the unpack / MFMA issue of one wave
// overlaps the other's waits (the 4-wave kernel spends ~2 us of its 8.4 us at M = 16 issuing ~1000 VALU per wave one after the other)
template <typename Tag, int NBITS, int V, int MT, int SPG, int NW = 4>
__global__ __launch_bounds__(NW * 64, 1) void gemm_wn_direct_kernel(const WnParams p) {
    using TR = F16Traits<Tag>;
    using DW = DirWin<Tag, NBITS>;
    using WT = typename DirVec<V>::W;
    using MV = typename DirVec<V>::M;
    constexpr bool SUBN = DW::SUBN;
    constexpr int E = 32 / NBITS, HALF = E / 2, WP = DW::WP;
    constexpr int NF = E / 8;             // 32-k MFMA steps fed by one packed word
    constexpr int KS = 16 / V;            // k-steps per wave and piece
    constexpr int U = KS / NF;            // row-steps (4 packed rows, one per lane quarter) per wave and piece
    constexpr int NGRP = KS / SPG;        // metadata rows per wave and piece
    constexpr int BM = 16 * MT, TN = 16 * V;
    constexpr int ROWS_WP = 4 * U, ROWS_PIECE = NW * ROWS_WP, NT = NW * 64;
    static_assert(E >= 8 && U >= 1 && NGRP >= 1 && SPG % NF == 0, "unsupported geometry");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = (float*)smem;  // [NW][BM][TN]
    unsigned* flag = (unsigned*)(smem + (size_t)NW * BM * TN * 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x, slice = blockIdx.y, mtile = blockIdx.z;
    const int n0 = tile * TN + c * V;
    const int m0 = mtile * BM;
    const int row_s0 = slice * p.rows_per_slice;
    const int npieces = p.rows_per_slice / ROWS_PIECE;

#ifdef GL_DIRECT_TIMELINE  // development build only (scripts/timeline_direct.py): wave 0 of every block stamps the 100 MHz clock
    const int lin_blk = tile + gridDim.x * (slice + gridDim.y * mtile);
    const bool probe = (p.flags & 4) && p.counters && wave == 0 && lane == 0 && lin_blk < 512;
    unsigned long long* stamps = (unsigned long long*)(p.counters + MAX_SPLITK_COUNTERS) + lin_blk * 8;
    auto stamp = [&](int i) { if (probe) stamps[i] = __builtin_amdgcn_s_memrealtime(); };
#else
    auto stamp = [](int) {};
#endif
    stamp(0);
    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    const uint16_t* sp = need_s ? (const uint16_t*)p.scales : (const uint16_t*)p.w;  // dummy source keeps the
    const uint16_t* zp = need_z ? (const uint16_t*)p.zeros : (const uint16_t*)p.w;   // loop free of branches
    const int64_t mstride = (need_s || need_z) ? p.stride_meta_g : 0;

    // lane's packed row inside a piece for row-step u: wave * ROWS_WP + 4u + g
    const int row_l0 = row_s0 + wave * ROWS_WP + g;
    const uint32_t* wbase = p.w + (int64_t)row_l0 * p.stride_wk + n0;
    const uint16_t* xrow[MT];
    bool xvalid[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        xvalid[t] = m0 + t * 16 + c < p.M;
        xrow[t] = (const uint16_t*)p.x + (int64_t)(xvalid[t] ? m0 + t * 16 + c : 0) * p.stride_xm + (int64_t)row_l0 * E;
    }

    struct Piece {
        WT w[U];
        u32x4 x[U][NF][MT];
        MV s[NGRP], z[NGRP];
    };
    auto load_piece = [&](Piece& pc, int piece) {
        const int rbase = piece * ROWS_PIECE;
#pragma unroll
        for (int u = 0; u < U; ++u) pc.w[u] = *(const WT*)(wbase + (int64_t)(rbase + 4 * u) * p.stride_wk);
#pragma unroll
        for (int t = 0; t < MT; ++t)
            if (xvalid[t]) {
                const uint16_t* src = xrow[t] + (int64_t)rbase * E;
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int q = 0; q < NF; ++q) pc.x[u][q][t] = *(const u32x4*)(src + u * 4 * E + 8 * q);
            }
        const int k_w = (row_s0 + rbase + wave * ROWS_WP) * E;  // first k of the wave's span
#pragma unroll
        for (int q = 0; q < NGRP; ++q) {
            const int64_t grp = group_of(k_w + q * 32 * SPG, p.gs_shift);
            pc.s[q] = *(const MV*)(sp + grp * mstride + n0);
            pc.z[q] = *(const MV*)(zp + grp * mstride + n0);
        }
    };

    f32x4 tot[MT][V];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < V; ++j) tot[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

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

    auto compute = [&](const Piece& pc, int) {
        f32x4 acc[MT][V], ones[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            ones[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < V; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int ks = u * NF + f;
                // A fragments: (x[a], x[a + HALF]) pairs, a = 4f + dd, scaled by 2^-(NBITS * (a % WP))
                u32x4 afr[MT], onesb;
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    const int a = 4 * f + dd, b = a + HALF, wi = a % WP;
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const uint32_t lo = pc.x[u][a / 8][t][(a % 8) / 2], hi = pc.x[u][b / 8][t][(b % 8) / 2];
                        uint32_t r = __builtin_amdgcn_perm(hi, lo, (a & 1) ? 0x07060302u : 0x05040100u);
                        if constexpr (SUBN) {
                            if (wi != 0) {
                                const _Float16 sc = (_Float16)(1.0f / (float)(1u << (NBITS * wi)));
                                r = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2_t, r) * (h2_t){sc, sc});
                            }
                        }
                        afr[t][dd] = r;
                    }
                    // constant B fragment that turns the stored x back into x: 2^(NBITS * wi) (fp16) / 1.0 (bf16)
                    onesb[dd] = SUBN ? (uint32_t)((15 + NBITS * wi) << 10) * 0x00010001u : TR::ONES2;
                }
#pragma unroll
                for (int t = 0; t < MT; ++t) ones[t] = mfma16d<Tag>(afr[t], onesb, ones[t]);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const uint32_t w = dir_word<V>(pc.w[u], j);
                    u32x4 bfrag;
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) {
                        const int d = 4 * f + dd, win = d / WP, wi = d % WP;
                        uint32_t h = (w >> (NBITS * WP * win)) & wmask[wi];
                        if constexpr (!SUBN) h |= TR::MAGIC2;
                        bfrag[dd] = h;
                    }
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t][j] = mfma16d<Tag>(afr[t], bfrag, acc[t][j]);
                }
                if ((ks + 1) % SPG == 0) {  // end of a quantisation group: fold scale / zero into the totals
                    const int q = ks / SPG;
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        const float s = need_s ? TR::to_float(dir_meta<V>(pc.s[q], j)) : 1.f;
                        const float z = need_z ? TR::to_float(dir_meta<V>(pc.z[q], j)) : scalar_zero;
                        const float a = s * QSCALE;
                        const float b = bz * z * (b_times_s ? s : 1.f) - a * OFF;  // OFF: the bf16 magic offset
#pragma unroll
                        for (int t = 0; t < MT; ++t)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) {
                                tot[t][j][rg] += a * acc[t][j][rg] + b * ones[t][rg];
                                acc[t][j][rg] = 0.f;
                            }
                    }
#pragma unroll
                    for (int t = 0; t < MT; ++t) ones[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    };

    Piece A, B;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int q = 0; q < NF; ++q)
#pragma unroll
            for (int t = 0; t < MT; ++t) A.x[u][q][t] = B.x[u][q][t] = (u32x4){0u, 0u, 0u, 0u};  // rows >= M stay zero
    pipeline2_prime(npieces, A, B, load_piece);
    stamp(1);
    pipeline2_run(npieces, A, B, load_piece, [&](const Piece& pc, int i) {
        compute(pc, i);
        if (i == 0) stamp(2);
    });
    stamp(3);

    // ---- combine the NW waves (disjoint K) through LDS ------------------------------------------------------
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = t * 16 + 4 * g + rg;
#pragma unroll
            for (int j = 0; j < V; ++j) red[(wave * BM + row) * TN + c * V + j] = tot[t][j][rg];
        }
    __syncthreads();
    stamp(4);
    constexpr int NOUT = BM * TN, OPT = NOUT / NT;
    static_assert(NOUT % NT == 0 && OPT >= 1, "every thread owns OPT outputs of the tile");
    float part[OPT];
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * NT;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w * NOUT + o];
        part[it] = v;
    }
    const int tile_lin = mtile * gridDim.x + tile;
    if (p.splitk == 1) {
#pragma unroll
        for (int it = 0; it < OPT; ++it) {
            const int o = tid + it * NT, m = m0 + o / TN;
            if (m < p.M) store_out_t<Tag>(p.epi, part[it], m, (int64_t)tile * TN + (o % TN));
        }
        stamp(5);
        return;
    }
    float* slab = p.slabs + ((int64_t)tile_lin * p.splitk) * NOUT;
#pragma unroll
    for (int it = 0; it < OPT; ++it) {  // only rows that exist travel through the slabs
        const int o = tid + it * NT;
        if (m0 + o / TN < p.M) slab_store(slab + (int64_t)slice * NOUT + o, part[it]);
    }
    stamp(5);
    if (!splitk_arrive_is_last(p.counters + tile_lin, p.splitk, flag)) return;
    stamp(6);
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * NT, m = m0 + o / TN;
        if (m < p.M) {
            float v = 0.f;
            for (int s = 0; s < p.splitk; ++s) v += slab_load(slab + (int64_t)s * NOUT + o);
            store_out_t<Tag>(p.epi, v, m, (int64_t)tile * TN + (o % TN));
        }
    }
    if (tid == 0) splitk_reset(p.counters + tile_lin);
    stamp(7);
}

// ---------------------------------------------------------------------------------------------
template <typename Tag, int NBITS, int V, int MT>
static const void* dpick_spg(int spg, int nw) {
    // group size 64 (two metadata rows per 128 k) fits the 512-VGPR budget for one row tile only (M <= 16);
    // group size 32 and 17..32 rows at group size 64 stay with the LDS-staged streaming kernel
    if (spg == 4) {
        if constexpr (V >= 2 && MT == 1) {  // 8 waves: 16 x 16 V outputs over 512 threads; two row tiles spill at 256 registers per wave
            if (nw == 8) return (const void*)gemm_wn_direct_kernel<Tag, NBITS, V, MT, 4, 8>;
        }
        return nw == 4 ? (const void*)gemm_wn_direct_kernel<Tag, NBITS, V, MT, 4> : nullptr;
    }
    if constexpr (MT == 1) {
        if (spg == 2 && nw == 4) return (const void*)gemm_wn_direct_kernel<Tag, NBITS, V, 1, 2>;
    }
    return nullptr;
}
template <typename Tag, int NBITS, int V>
static const void* dpick_mt(int mt, int spg, int nw) {
    if (mt == 1) return dpick_spg<Tag, NBITS, V, 1>(spg, nw);
    if constexpr (V >= 2) { if (mt == 2) return dpick_spg<Tag, NBITS, V, 2>(spg, nw); }
    return nullptr;
}
template <typename Tag, int NBITS>
static const void* dpick_v(int v, int mt, int spg, int nw) {
    switch (v) {
        case 1: return dpick_mt<Tag, NBITS, 1>(mt, spg, nw);
        case 2: return dpick_mt<Tag, NBITS, 2>(mt, spg, nw);
        case 4: return dpick_mt<Tag, NBITS, 4>(mt, spg, nw);
        default: return nullptr;
    }
}
template <typename Tag>
static const void* dpick_bits(int nbits, int v, int mt, int spg, int nw) {
    switch (nbits) {
        case 2: return dpick_v<Tag, 2>(v, mt, spg, nw);
        case 4: return dpick_v<Tag, 4>(v, mt, spg, nw);
        default: return nullptr;
    }
}

// tuning[0]: 0 auto | 1, 2, 4 force the words-per-lane V (16 V columns per block)
// tuning[1]: 0 auto | n force split-K n;  tuning[2]: 0 auto | 4 / 8 waves per block (8: one row tile, >= 32-column tiles, K slice of whole 8-wave pieces)
bool plan_gemm_wn_direct(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    const int nbits = a.W_nbits;
    if (nbits != 2 && nbits != 4) return false;
    const int e = 32 / nbits;
    if (a.M > 32) return false;
    if (a.output_dtype != a.input_dtype) return false;  // typed epilogue / metadata
    // metadata read in the K loop: the kernel's 16-bit type; channel scales of the epilogue alone: any float type (store_out_t; round 4:
    // BitNet's fp32 scale, A16W158_INT)
    const bool loop_s = a.W_group_mode >= 2, post_s = a.channel_scale_mode == 1 || a.channel_scale_mode == 3;
    const bool uses_s = loop_s;
    const bool has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (loop_s && a.meta_dtype != a.input_dtype) return false;
    if (post_s && !loop_s && a.meta_dtype != a.input_dtype && a.meta_dtype != GEMLITE_DT_FP32 && a.meta_dtype != GEMLITE_DT_FP16 && a.meta_dtype != GEMLITE_DT_BF16) return false;
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != a.input_dtype) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    if ((a.stride_xm * 2) % 16 != 0 || ((uintptr_t)a.x % 16) != 0) return false;  // 16-byte x loads
    if (((uintptr_t)a.w_q % 16) != 0 || (a.stride_wk % 4) != 0) return false;
    if ((uses_s && ((uintptr_t)a.scales % 8) != 0) || (has_z && !a.zero_is_scalar && ((uintptr_t)a.zeros % 8) != 0)) return false;
    if (p.stride_meta_g % 4 != 0) return false;
    const int64_t gs = p.group_size;
    int spg;
    if (gs % 128 == 0) spg = 4;       // scale / zero folded in every 128 k
    else if (gs == 64) spg = 2;       // ... every 64 k
    else return false;
    if (gs < 4 * e) return false;     // a group must contain whole 4-row wave loads (their k-steps are interleaved)
    const int mt = a.M <= 16 ? 1 : 2;
    const int bm = 16 * mt;
    const int mtiles = (int)((a.M + bm - 1) / bm);
    const bool f16 = a.input_dtype == GEMLITE_DT_FP16;
    auto fn_nw = [&](int v, int nw) { return f16 ? dpick_bits<half_tag>(nbits, v, mt, spg, nw) : dpick_bits<bf16_tag>(nbits, v, mt, spg, nw); };
    auto fn_of = [&](int v) { return fn_nw(v, 4); };
    auto fits = [&](int v) { return a.N % (16 * v) == 0 && a.K % (2048 / v) == 0 && fn_of(v) != nullptr; };
    int v = 0, force_sk = 0;
    if (a.tuning[0] == 1 || a.tuning[0] == 2 || a.tuning[0] == 4) {
        if (!fits(a.tuning[0])) return false;
        v = a.tuning[0];
    } else {
        // widest tile that still gives most CUs a block without splitting K (>= 160 of 256: N = 11008 / 13824 / 14336 with
        // 64-column tiles measured 12-48 % faster than 32-column tiles, profiles/r02/autotune_report.json)
        for (int cand : {4, 2, 1})
            if (fits(cand) && (a.N / (16 * cand)) * mtiles >= 160) { v = cand; break; }
        if (!v)
            for (int cand : {1, 2, 4})
                if (fits(cand)) { v = cand; break; }
        if (!v) return false;
        // every block re-reads its K range of all M rows of x from L2: with 16-column tiles that is 4x the weight
        // bytes at M = 16.  From 8 rows on, 32-column tiles with K split in two were measured faster
        // (4096 x 4096, M = 16: 7.9 us vs 8.9 us) although they pay the cross-block combine.
        // (round 3: from 5 rows — M = 6 at 4096^2 7.7 -> 6.5 us, 4096 x 14336 16.6 -> 11.8, profiles/r03/probe_m6_llm_shapes.log)
        if (v == 1 && a.M >= 5 && a.tuning[1] == 0 && fits(2) && (a.K / 1024) % 2 == 0 && (a.N / 32) * mtiles * 2 >= 256) {
            v = 2;
            force_sk = 2;
        }
        // round 3 (profiles/r03/probe_fewrows_llm_shapes.log): the same step from 32- to 64-column tiles where 64-column tiles x 2
        // slices still fill the chip — with 8 waves per block: 8192^2 M = 8 / 16 11.9 / 13.7 -> 10.7 / 12.3 us, 8192 x 28672 27.1 / 35.2
        // -> 23.3 / 24.5; N = 6144 (192 blocks) loses (7.8 / 8.9 -> 8.0 / 9.8) and keeps the unsplit 32-column tiles
        else if (v == 2 && mt == 1 && a.M >= 5 && a.tuning[1] == 0 && a.tuning[2] == 0 && fits(4) && a.K % 2048 == 0 && (a.N / 64) * 2 >= 256 &&
                 fn_nw(4, 8) != nullptr) {
            v = 4;
            force_sk = 2;
        }
    }
    const int tiles = (int)(a.N / (16 * v));
    const int units = (int)(a.K / (2048 / v));
    auto ok = [&](int sk) { return sk >= 1 && units % sk == 0; };
    int splitk = 0;
    if (a.tuning[1] > 0 || force_sk > 0) {
        const int want = a.tuning[1] > 0 ? a.tuning[1] : force_sk;
        if (!ok(want)) return false;
        splitk = want;
    } else {
        for (int sk = 1; sk <= units; sk *= 2) {
            if (!ok(sk)) continue;
            splitk = sk;
            if ((int64_t)tiles * mtiles * sk >= 160) break;  // a K slice costs a combine: 172 unsplit blocks beat 344 split ones
        }
        if (!splitk) return false;
    }
    const uint64_t ntl = (uint64_t)tiles * mtiles;
    if (splitk > 1 && ntl > (uint64_t)MAX_SPLITK_COUNTERS) return false;
    p.splitk = splitk;
    p.rows_per_slice = (int)(a.K / e) / splitk;
    // waves per block: tuning[2] = 4 / 8 forces; 8 needs a K slice of whole 8-wave pieces (4096 / V k) and V >= 2
    int nw = 4;
    const bool nw8_ok = fn_nw(v, 8) != nullptr && (a.K / splitk) % (4096 / v) == 0;
    if (a.tuning[2] == 8) { if (!nw8_ok) return false; nw = 8; }
    // default: with two K slices (4096^2 M = 8 / 16, 32-column tiles x 2: 7.1 / 8.3 -> 6.6 / 7.4 us; the 8192-wide shapes above);
    // unsplit tiles are a wash (N = 14336: 9.8 / 11.8 -> 9.5 / 10.3, N = 28672: 22.8 / 25.6 -> 23.9 / 27.2) and keep 4 waves
    else if (a.tuning[2] != 4 && nw8_ok && splitk == 2) nw = 8;
    lp.fn = fn_nw(v, nw);
    static const char* names[2][3] = {{"gemm_wn_direct_kernel<tile16>", "gemm_wn_direct_kernel<tile32>", "gemm_wn_direct_kernel<tile64>"},
                                      {"gemm_wn_direct_kernel<tile16,8w>", "gemm_wn_direct_kernel<tile32,8w>", "gemm_wn_direct_kernel<tile64,8w>"}};
    lp.name = names[nw == 8][v == 1 ? 0 : (v == 2 ? 1 : 2)];
    lp.grid = dim3(tiles, splitk, mtiles);
    lp.block = dim3(nw * 64, 1, 1);
    lp.lds_bytes = (size_t)nw * bm * 16 * v * 4 + 16;
    lp.slab_bytes = splitk > 1 ? ntl * splitk * bm * 16 * v * 4 : 0;
    lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

}  // namespace gl
