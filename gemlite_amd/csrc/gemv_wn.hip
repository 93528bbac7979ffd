// gemv_wn.hip — fused unpack + group-dequant + GEMV for packed low-bit weights, M <= 8 (decode).
//
// Replaces the reference's gemv_INT_revsplitK_kernel / gemv_INT_kernel / gemv_INT_splitK_kernel
// (gemlite/triton_kernels/gemv_revsplitK_kernels.py:226-462, gemv_kernels.py:230-388,
// gemv_splitK_kernels.py:240-420) for packed int32 weights.  HBM-bound: the only large stream is W_q.
//
// Mapping (CDNA4, 64-wide waves, block = 4 waves):
//   * a block owns a 64-column tile and one K slice (gridDim.y slices, combined in-launch);
//   * lane = (g = lane>>4, c = lane&15): it owns the 4 adjacent columns 4c..4c+3 (one 16-byte
//     global_load_dwordx4 per packed row) and packed rows  wave_base + chunk*4R + R*g + i, i < R (R=4).
//     One wave-level load instruction therefore reads 4 row segments of 256 contiguous bytes;
//     a lane's 4 rows are consecutive, so for group_size 128 they share one (scale, zero) pair;
//   * weights go HBM -> VGPR directly (no LDS round trip: every byte is used by exactly one lane);
//     the K slice of x (<= 8 KB per row) is staged in LDS once, pair-permuted so one 32-bit LDS
//     word is exactly the (k, k + e/2) pair that one AND/OR "magic number" unpack produces;
//   * math: inside one quantisation group  sum_k x_k (a q_k + b) = a * sum_k x_k q_k + b * sum_k x_k,
//     so the inner loop is  AND/OR -> v_dot2(c)_f32_{f16,bf16}  on (OFF + q) pairs with fp32
//     accumulation; the OFF * sum(x) term is removed once per run of rows;
//   * reduction: 2 ds_bpermute/DPP xor-shuffles across the 4 row sub-groups of a wave, LDS across
//     the 4 waves, then split-K slabs: write-through (sc1) stores + arrival ticket, last block
//     sums the slabs in fixed slice order (deterministic), applies the epilogue and stores.
#include "gl_common.h"

namespace gl {

template <typename Tag, int NBITS, int MB, int R>
__global__ __launch_bounds__(256, (MB * (16 / NBITS) >= 32 ? 1 : 2)) void gemv_wn_kernel(const WnParams p) {
    using TR = F16Traits<Tag>;
    constexpr int E = 32 / NBITS;    // elements per packed word
    constexpr int HALF = E / 2;      // (k, k+HALF) pairs per word == LDS dwords per packed row
    constexpr uint32_t QMASK2 = ((1u << NBITS) - 1u) * 0x00010001u;
    constexpr int CHUNK = 4 * R;     // packed rows one wave consumes per iteration (4 sub-groups x R)
    static_assert(NBITS <= TR::MAX_QBITS, "q + OFF must be exact in the 16-bit float type");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x, slice = blockIdx.y;
    const int n0 = tile * 64 + c * 4;

    const int rows_slice = p.rows_per_slice;          // multiple of 4*CHUNK (= 16*R)
    const int rows_wave = rows_slice >> 2;            // each wave takes a contiguous quarter
    const int row_s0 = slice * rows_slice;            // first packed row of the slice
    const int row_w0 = wave * rows_wave;              // wave start, relative to the slice
    const int pairs = rows_slice * HALF;              // LDS dwords per x row

    uint32_t* xs = (uint32_t*)smem;                               // [MB][pairs]
    float* red = (float*)(smem + (size_t)MB * pairs * 4);          // [4][MB][64]
    unsigned* flag = (unsigned*)(red + 4 * MB * 64);

    const uint32_t* wbase = p.w + (int64_t)(row_s0 + row_w0 + g * R) * p.stride_wk + n0;
    const int nchunks = rows_wave / CHUNK;

    u32x4 wa[R], wb[R];
    auto load_w = [&](u32x4 (&dst)[R], int chunk) {
#pragma unroll
        for (int i = 0; i < R; ++i)
            dst[i] = *(const u32x4*)(wbase + (int64_t)(chunk * CHUNK + i) * p.stride_wk);
    };
    // issue the first weight loads before anything else: they are the long pole
    load_w(wa, 0);
    if (nchunks > 1) load_w(wb, 1);

    // ---- stage x[k-slice] into LDS, pair-permuted -------------------------------------------
    {
        const uint16_t* xg = (const uint16_t*)p.x;
        const int64_t k0 = (int64_t)row_s0 * E;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            for (int pi = tid; pi < pairs; pi += 256) {
                const int word = pi / HALF, d = pi - word * HALF;
                uint32_t v = 0;
                if (m < p.M) {
                    const int64_t k = k0 + (int64_t)word * E + d;
                    const uint32_t lo = xg[m * p.stride_xm + k * p.stride_xk];
                    const uint32_t hi = xg[m * p.stride_xm + (k + HALF) * p.stride_xk];
                    v = lo | (hi << 16);
                }
                xs[m * pairs + pi] = v;
            }
        }
    }
    __syncthreads();

    float tot[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) tot[m][j] = 0.f;

    const int scalar_zero = p.zero_is_scalar ? ((const int32_t*)p.zeros)[0] : 0;

    auto compute = [&](const u32x4 (&wv)[R], int chunk) {
        const int row_rel = row_w0 + chunk * CHUNK + g * R;  // first of this lane's R rows (slice-relative)
        // metadata of the (single) group these R rows live in
        float a[4], b[4];
        {
            const int64_t grp = ((int64_t)(row_s0 + row_rel) * E) / p.group_size;
            f32x4 s = {1.f, 1.f, 1.f, 1.f}, z = {0.f, 0.f, 0.f, 0.f};
            if (p.w_mode >= 2) s = load_meta4(p.scales, grp * p.stride_meta_g + n0, p.meta_dt);
            if (p.w_mode == 1 || p.w_mode >= 3) {
                if (p.zero_is_scalar) {
                    z[0] = z[1] = z[2] = z[3] = (float)scalar_zero;
                } else {
                    z = load_meta4(p.zeros, grp * p.stride_meta_g + n0, p.zeros_dt);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) group_affine(s[j], z[j], p.w_mode, a[j], b[j]);
        }
        float acc[MB][4], accx[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            accx[m] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            constexpr int XW = HALF >= 4 ? 4 : HALF;  // x dwords fetched per LDS read
#pragma unroll
            for (int q = 0; q < HALF / XW; ++q) {
                uint32_t xr[MB][XW];
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const uint32_t* src = xs + m * pairs + (row_rel + i) * HALF + q * XW;
                    if constexpr (XW == 4) {
                        const u32x4 v = *(const u32x4*)src;
                        xr[m][0] = v[0]; xr[m][1] = v[1]; xr[m][2] = v[2]; xr[m][3] = v[3];
                    } else {  // HALF == 2 (8-bit words)
                        const u32x2 v = *(const u32x2*)src;
                        xr[m][0] = v[0]; xr[m][1] = v[1];
                    }
                }
#pragma unroll
                for (int dd = 0; dd < XW; ++dd) {
                    const int d = q * XW + dd;
#pragma unroll
                    for (int m = 0; m < MB; ++m) accx[m] = TR::dot2(xr[m][dd], TR::ONES2, accx[m]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // (OFF + q_d, OFF + q_{d+HALF}) as two 16-bit floats
                        const uint32_t h = ((wv[i][j] >> (NBITS * d)) & QMASK2) | TR::MAGIC2;
#pragma unroll
                        for (int m = 0; m < MB; ++m) acc[m][j] = TR::dot2(h, xr[m][dd], acc[m][j]);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                tot[m][j] += a[j] * (acc[m][j] - TR::OFF * accx[m]) + b[j] * accx[m];
    };

    for (int ch = 0; ch < nchunks; ch += 2) {
        compute(wa, ch);
        if (ch + 2 < nchunks) load_w(wa, ch + 2);
        if (ch + 1 < nchunks) {
            compute(wb, ch + 1);
            if (ch + 3 < nchunks) load_w(wb, ch + 3);
        }
    }

    // ---- reduce over the 4 row sub-groups of the wave (lanes l, l^16, l^32, l^48) --------------
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = tot[m][j];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            tot[m][j] = v;
        }
    if (g == 0) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            f32x4 v = {tot[m][0], tot[m][1], tot[m][2], tot[m][3]};
            *(f32x4*)(red + (wave * MB + m) * 64 + c * 4) = v;
        }
    }
    __syncthreads();

    // ---- across the 4 waves; output o = m*64 + col, o < MB*64, strided over the 256 threads ----
    constexpr int NOUT = MB * 64;
    constexpr int OPT = (NOUT + 255) / 256;  // outputs per thread
    float part[OPT];
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * 256;
        float v = 0.f;
        if (o < NOUT) {
            const int m = o >> 6, col = o & 63;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[(w * MB + m) * 64 + col];
        }
        part[it] = v;
    }
    if (p.splitk == 1) {
#pragma unroll
        for (int it = 0; it < OPT; ++it) {
            const int o = tid + it * 256;
            if (o < NOUT && (o >> 6) < p.M) epilogue_store(p.epi, part[it], o >> 6, (int64_t)tile * 64 + (o & 63));
        }
        return;
    }
    float* slab = p.slabs + ((int64_t)tile * p.splitk) * NOUT;
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * 256;
        if (o < NOUT) slab_store(slab + (int64_t)slice * NOUT + o, part[it]);
    }
    if (!splitk_arrive_is_last(p.counters + tile, p.splitk, flag)) return;
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * 256;
        if (o < NOUT) {
            float v = 0.f;
            for (int s = 0; s < p.splitk; ++s) v += slab_load(slab + (int64_t)s * NOUT + o);
            if ((o >> 6) < p.M) epilogue_store(p.epi, v, o >> 6, (int64_t)tile * 64 + (o & 63));
        }
    }
    if (tid == 0) splitk_reset(p.counters + tile);
}

// ---------------------------------------------------------------------------------------------
// host-side planning
// ---------------------------------------------------------------------------------------------
template <typename Tag, int NBITS, int MB>
static const void* pick_r(int r) {
    return r == 4 ? (const void*)gemv_wn_kernel<Tag, NBITS, MB, 4> : (const void*)gemv_wn_kernel<Tag, NBITS, MB, 1>;
}
template <typename Tag, int NBITS>
static const void* pick_mb(int mb, int r) {
    switch (mb) {
        case 1: return pick_r<Tag, NBITS, 1>(r);
        case 2: return pick_r<Tag, NBITS, 2>(r);
        case 4: return pick_r<Tag, NBITS, 4>(r);
        default: return pick_r<Tag, NBITS, 8>(r);
    }
}
template <typename Tag>
static const void* pick_bits(int nbits, int mb, int r) {
    switch (nbits) {
        case 1: return pick_mb<Tag, 1>(mb, r);
        case 2: return pick_mb<Tag, 2>(mb, r);
        case 4: return pick_mb<Tag, 4>(mb, r);
        case 8:
            if constexpr (F16Traits<Tag>::MAX_QBITS >= 8) return pick_mb<Tag, 8>(mb, r);
            return nullptr;
        default: return nullptr;
    }
}

// Decide grid / split-K / LDS for the GEMV kernel.  Returns false if this shape is not covered.
bool plan_gemv_wn(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    const int nbits = a.W_nbits, e = 32 / nbits;
    if (a.M > 8 || a.N % 64 != 0 || a.K % e != 0) return false;
    const int rows = (int)(a.K / e);
    const int64_t gs = p.group_size;
    if (gs % e != 0) return false;
    const int rpg = (int)(gs / e);  // packed rows per group
    // rows a lane consumes per chunk must stay inside one group
    const int r = (rpg % 4 == 0 && rows % 64 == 0) ? 4 : 1;
    const int chunk_rows = 16 * r;  // 4 waves * 4 sub-groups * r packed rows per block iteration
    if (rows % chunk_rows != 0) return false;
    const int mb = a.M <= 1 ? 1 : (a.M <= 2 ? 2 : (a.M <= 4 ? 4 : 8));
    const void* fn = a.input_dtype == GEMLITE_DT_FP16 ? pick_bits<half_tag>(nbits, mb, r)
                                                       : pick_bits<bf16_tag>(nbits, mb, r);
    if (!fn) return false;

    // split K so that (a) >= ~2 blocks per CU exist, (b) the x slice fits LDS comfortably
    const int tiles = (int)(a.N / 64);
    const int units = rows / chunk_rows;  // max number of slices
    int splitk = a.tuning[1] > 0 ? a.tuning[1] : 1;
    if (a.tuning[1] <= 0) {
        const int target_blocks = 512;
        while (splitk < units && tiles * splitk < target_blocks && (units % (splitk * 2) == 0)) splitk *= 2;
    }
    if (units % splitk != 0) return false;
    // LDS cap: MB * k_slice * 2 bytes <= 64 KiB
    while (((int64_t)mb * (rows / splitk) * e * 2 > 65536) && (units % (splitk * 2) == 0)) splitk *= 2;
    if ((int64_t)mb * (rows / splitk) * e * 2 > 65536) return false;

    p.splitk = splitk;
    p.rows_per_slice = rows / splitk;
    lp.fn = fn;
    lp.name = "gemv_wn_kernel";
    lp.grid = dim3(tiles, splitk, 1);
    lp.block = dim3(256, 1, 1);
    lp.lds_bytes = (size_t)mb * p.rows_per_slice * (e / 2) * 4 + (size_t)4 * mb * 64 * 4 + 16;
    lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * mb * 64 * 4 : 0;
    lp.ws_bytes = lp.slab_bytes + (splitk > 1 ? (uint64_t)tiles * 4 : 0);
    return true;
}

}  // namespace gl
