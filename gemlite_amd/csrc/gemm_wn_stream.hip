// gemm_wn_stream.hip — fused unpack + dequant + MFMA GEMM for packed low-bit weights, weight-streaming
// regime (small / medium M).  Replaces gemm_splitK_INT_kernel (gemlite/triton_kernels/
// gemm_splitK_kernels.py:277-450) and, through gridDim.z row tiles, is the first correct MFMA path for
// gemm_INT_kernel (gemm_kernels.py:248-413).
//
// Mapping (CDNA4): block = 4 waves, 64-column tile x BM = 16*MT rows x one K slice.
//   * v_mfma_f32_16x16x32_{f16,bf16}: the B operand of lane (g = lane>>4, n = lane&15) is 8 consecutive
//     k of ONE column — with the K-packed int32 layout that is a slice of one packed word, so weights go
//     HBM -> VGPR -> (unpack, dequant) -> MFMA with no LDS transpose.  A lane loads 16 bytes = 4 adjacent
//     columns, giving 4 MFMAs per load whose column sets are {4n + j}; the C fragment of lane (g, n) is
//     then rows 4g..4g+3 x columns 4n..4n+3 — contiguous again.
//   * the K order inside an MFMA is free as long as A and B agree: both use the "pair-permuted" order
//     (k_d, k_{d+e/2}) the AND/OR unpack produces, so x is stored that way in LDS (one ds_read_b128 per
//     A fragment, rows padded by 16 B against bank conflicts).
//   * K is walked in pieces of 512: wave w owns k in [128w, 128w+128) of each piece (one group for
//     group_size 128), so the 4 waves hold partial sums over disjoint K that are combined in LDS, then
//     across K slices with the same write-through slab + ticket protocol as the GEMV.
//   * dequant follows triton_kernels/utils.py:73-87: fp16 uses packed fp16 ops exactly as the reference
//     (q exact, then fma/sub/mul in fp16); bf16 has no packed VALU on gfx950, so it evaluates
//     fma(q + 128, A, B) in fp32 with the 128 offset folded into B, and rounds once to bf16.
#include "gl_common.h"

namespace gl {

template <typename Tag>
struct Dequant2;

// fp16: returns two dequantised weights as packed fp16 (bit pattern)
template <>
struct Dequant2<half_tag> {
    h2_t s2, z2;  // per column: scale splat, zero (or folded zero) splat
    __device__ __forceinline__ void set(float s, float z, int) {
        s2 = (h2_t){(_Float16)s, (_Float16)s};
        z2 = (h2_t){(_Float16)z, (_Float16)z};
    }
    __device__ __forceinline__ uint32_t apply(uint32_t h, int w_mode) const {
        h2_t q = __builtin_bit_cast(h2_t, h) - (h2_t){(_Float16)1024.0f, (_Float16)1024.0f};  // exact
        switch (w_mode) {
            case 1: q = q - z2; break;
            case 2: q = q * s2; break;
            case 3: q = (q - z2) * s2; break;
            case 4: q = __builtin_elementwise_fma(q, s2, z2); break;
            default: break;
        }
        return __builtin_bit_cast(uint32_t, q);
    }
};

// bf16: w = fma(f, A, B) in fp32 where f = 128 + q, rounded once to bf16
template <>
struct Dequant2<bf16_tag> {
    float A, B;
    __device__ __forceinline__ void set(float s, float z, int w_mode) {
        switch (w_mode) {
            case 1: A = 1.0f; B = -(z + 128.0f); break;
            case 2: A = s; B = -128.0f * s; break;
            case 3: A = s; B = -(z + 128.0f) * s; break;
            case 4: A = s; B = __builtin_fmaf(-128.0f, s, z); break;
            default: A = 1.0f; B = -128.0f; break;
        }
    }
    __device__ __forceinline__ uint32_t apply(uint32_t h, int) const {
        const float lo = __builtin_bit_cast(float, h << 16);
        const float hi = __builtin_bit_cast(float, h & 0xFFFF0000u);
        const b2_t r = {(__bf16)__builtin_fmaf(lo, A, B), (__bf16)__builtin_fmaf(hi, A, B)};
        return __builtin_bit_cast(uint32_t, r);
    }
};

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

constexpr int PIECE_K = 512;                 // k per block iteration
constexpr int XPITCH = PIECE_K / 2 + 4;      // LDS dwords per x row (16-byte pad)

template <typename Tag, int NBITS, int MT>
__global__ __launch_bounds__(256, 2) void gemm_wn_stream_kernel(const WnParams p) {
    using TR = F16Traits<Tag>;
    constexpr int E = 32 / NBITS, HALF = E / 2;
    constexpr uint32_t QMASK2 = ((1u << NBITS) - 1u) * 0x00010001u;
    constexpr int NF = E / 8;                 // 32-k MFMA steps fed by one 4-row wave load
    constexpr int ROWS_WP = (PIECE_K / 4) / E;  // packed rows per wave per piece
    constexpr int U = ROWS_WP / 4;            // 16-byte loads per lane per piece
    constexpr int BM = 16 * MT;
    static_assert(E >= 8 && U >= 1, "8-bit words take the generic path");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* xs = (uint32_t*)smem;  // [BM][XPITCH] dwords, later reused as red[4][BM][64] floats
    unsigned* flag = (unsigned*)(smem + (size_t)(BM * XPITCH * 4 > 4 * BM * 64 * 4 ? BM * XPITCH * 4 : 4 * BM * 64 * 4));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x, slice = blockIdx.y, mtile = blockIdx.z;
    const int n0 = tile * 64 + c * 4;
    const int m0 = mtile * BM;

    const int rows_slice = p.rows_per_slice;  // multiple of 4*ROWS_WP
    const int row_s0 = slice * rows_slice;
    const int npieces = rows_slice / (4 * ROWS_WP);

    // lane's packed row inside a piece for load u: wave*ROWS_WP + 4u + g
    const uint32_t* wbase = p.w + (int64_t)(row_s0 + wave * ROWS_WP + g) * p.stride_wk + n0;
    u32x4 wa[U], wb[U];
    auto load_w = [&](u32x4 (&dst)[U], int piece) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            dst[u] = *(const u32x4*)(wbase + (int64_t)(piece * 4 * ROWS_WP + 4 * u) * p.stride_wk);
    };
    load_w(wa, 0);

    f32x4 acc[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int scalar_zero = p.zero_is_scalar ? ((const int32_t*)p.zeros)[0] : 0;
    Dequant2<Tag> dq[4];
    int64_t cur_grp = -1;

    // x staging: one packed-word worth of k (E halfs) per step, pair-permuted into HALF dwords
    auto stage_x = [&](int piece) {
        const uint16_t* xg = (const uint16_t*)p.x;
        const int64_t k0 = (int64_t)(row_s0 + piece * 4 * ROWS_WP) * E;
        constexpr int WORDS_PER_ROW = PIECE_K / E;
        for (int idx = tid; idx < BM * WORDS_PER_ROW; idx += 256) {
            const int r = idx / WORDS_PER_ROW, wd = idx - r * WORDS_PER_ROW;
            uint32_t outv[HALF];
            if (m0 + r < p.M) {
                const uint16_t* src = xg + (int64_t)(m0 + r) * p.stride_xm + (k0 + (int64_t)wd * E) * p.stride_xk;
                uint16_t v[E];
                if (p.stride_xk == 1) {
#pragma unroll
                    for (int q = 0; q < E / 8; ++q) {
                        const u32x4 t4 = *(const u32x4*)(src + 8 * q);
#pragma unroll
                        for (int z = 0; z < 4; ++z) {
                            v[8 * q + 2 * z] = (uint16_t)(t4[z] & 0xFFFFu);
                            v[8 * q + 2 * z + 1] = (uint16_t)(t4[z] >> 16);
                        }
                    }
                } else {
#pragma unroll
                    for (int z = 0; z < E; ++z) v[z] = src[(int64_t)z * p.stride_xk];
                }
#pragma unroll
                for (int d = 0; d < HALF; ++d) outv[d] = (uint32_t)v[d] | ((uint32_t)v[d + HALF] << 16);
            } else {
#pragma unroll
                for (int d = 0; d < HALF; ++d) outv[d] = 0u;
            }
            uint32_t* dst = xs + r * XPITCH + wd * HALF;
#pragma unroll
            for (int q = 0; q < HALF / 4; ++q)
                *(u32x4*)(dst + 4 * q) = (u32x4){outv[4 * q], outv[4 * q + 1], outv[4 * q + 2], outv[4 * q + 3]};
        }
    };

    auto compute = [&](const u32x4 (&wv)[U], int piece) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row_p = wave * ROWS_WP + 4 * u + g;  // packed row inside the piece
            const int64_t row = (int64_t)row_s0 + piece * 4 * ROWS_WP + row_p;
            const int64_t grp = (row * E) / p.group_size;
            if (grp != cur_grp) {
                cur_grp = grp;
                f32x4 s = {1.f, 1.f, 1.f, 1.f}, z = {0.f, 0.f, 0.f, 0.f};
                if (p.w_mode >= 2) s = load4_t<Tag>(p.scales, grp * p.stride_meta_g + n0);
                if (p.w_mode == 1 || p.w_mode >= 3) {
                    if (p.zero_is_scalar) z[0] = z[1] = z[2] = z[3] = (float)scalar_zero;
                    else z = load4_t<Tag>(p.zeros, grp * p.stride_meta_g + n0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) dq[j].set(s[j], z[j], p.w_mode);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                u32x4 bfrag[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) {
                        const int d = 4 * f + dd;
                        const uint32_t h = ((wv[u][j] >> (NBITS * d)) & QMASK2) | TR::MAGIC2;
                        bfrag[j][dd] = dq[j].apply(h, p.w_mode);
                    }
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const u32x4 afrag = *(const u32x4*)(xs + (t * 16 + c) * XPITCH + row_p * HALF + 4 * f);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[t][j] = mfma16<Tag>(afrag, bfrag[j], acc[t][j]);
                }
            }
        }
    };

    for (int pc = 0; pc < npieces; pc += 2) {
        __syncthreads();
        stage_x(pc);
        if (pc + 1 < npieces) load_w(wb, pc + 1);
        __syncthreads();
        compute(wa, pc);
        if (pc + 1 < npieces) {
            __syncthreads();
            stage_x(pc + 1);
            if (pc + 2 < npieces) load_w(wa, pc + 2);
            __syncthreads();
            compute(wb, pc + 1);
        }
    }

    // ---- combine the 4 waves (disjoint K) through LDS ------------------------------------------
    __syncthreads();
    float* red = (float*)smem;  // [4][BM][64]
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int row = t * 16 + 4 * g + rg;
            const f32x4 v = {acc[t][0][rg], acc[t][1][rg], acc[t][2][rg], acc[t][3][rg]};
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
    for (int it = 0; it < OPT; ++it) slab_store(slab + (int64_t)slice * NOUT + tid + it * 256, part[it]);
    if (!splitk_arrive_is_last(p.counters + tile_lin, p.splitk, flag)) return;
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * 256, m = m0 + (o >> 6);
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += slab_load(slab + (int64_t)s * NOUT + o);
        if (m < p.M) store_out_t<Tag>(p.epi, v, m, (int64_t)tile * 64 + (o & 63));
    }
    if (tid == 0) splitk_reset(p.counters + tile_lin);
}

// ---------------------------------------------------------------------------------------------
template <typename Tag, int NBITS>
static const void* pick_mt(int mt) {
    switch (mt) {
        case 1: return (const void*)gemm_wn_stream_kernel<Tag, NBITS, 1>;
        case 2: return (const void*)gemm_wn_stream_kernel<Tag, NBITS, 2>;
        default: return (const void*)gemm_wn_stream_kernel<Tag, NBITS, 4>;
    }
}
template <typename Tag>
static const void* pick_bits_s(int nbits, int mt) {
    switch (nbits) {
        case 1: return pick_mt<Tag, 1>(mt);
        case 2: return pick_mt<Tag, 2>(mt);
        case 4: return pick_mt<Tag, 4>(mt);
        default: return nullptr;
    }
}

bool plan_gemm_wn_stream(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    const int nbits = a.W_nbits;
    if (nbits != 1 && nbits != 2 && nbits != 4) return false;
    const int e = 32 / nbits;
    if (a.N % 64 != 0 || a.K % PIECE_K != 0) return false;
    if (p.group_size % e != 0) return false;
    if (a.output_dtype != a.input_dtype) return false;  // typed epilogue / metadata
    const bool uses_s = a.W_group_mode >= 2 || a.channel_scale_mode == 1 || a.channel_scale_mode == 3;
    const bool has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (uses_s && a.meta_dtype != a.input_dtype) return false;
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != a.input_dtype) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    const int rows = (int)(a.K / e);
    const int piece_rows = PIECE_K / e;
    const int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
    const int bm = 16 * mt;
    const void* fn = a.input_dtype == GEMLITE_DT_FP16 ? pick_bits_s<half_tag>(nbits, mt)
                                                       : pick_bits_s<bf16_tag>(nbits, mt);
    if (!fn) return false;
    const int tiles = (int)(a.N / 64), mtiles = (int)((a.M + bm - 1) / bm);
    const int units = rows / piece_rows;
    int splitk = a.tuning[1] > 0 ? a.tuning[1] : 1;
    if (a.tuning[1] <= 0) {
        while (splitk < units && (int64_t)tiles * mtiles * splitk < 512 && units % (splitk * 2) == 0) splitk *= 2;
    }
    if (units % splitk != 0) return false;
    p.splitk = splitk;
    p.rows_per_slice = rows / splitk;
    lp.fn = fn;
    lp.name = "gemm_wn_stream_kernel";
    lp.grid = dim3(tiles, splitk, mtiles);
    lp.block = dim3(256, 1, 1);
    const size_t xs_b = (size_t)bm * XPITCH * 4, red_b = (size_t)4 * bm * 64 * 4;
    lp.lds_bytes = (xs_b > red_b ? xs_b : red_b) + 16;
    const uint64_t ntl = (uint64_t)tiles * mtiles;
    lp.slab_bytes = splitk > 1 ? ntl * splitk * bm * 64 * 4 : 0;
    if (splitk > 1 && ntl > (uint64_t)MAX_SPLITK_COUNTERS) return false;
    lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

}  // namespace gl
