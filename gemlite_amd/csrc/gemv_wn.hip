// gemv_wn.hip — fused unpack + group-dequant + GEMV for packed low-bit weights, M <= 8 (decode).
//
// Replaces the reference's gemv_INT_revsplitK_kernel / gemv_INT_kernel / gemv_INT_splitK_kernel
// (gemlite/triton_kernels/gemv_revsplitK_kernels.py:226-462, gemv_kernels.py:230-388,
// gemv_splitK_kernels.py:240-420) for packed int32 weights.  HBM-bound: the only large stream is W_q.
//
// Mapping (CDNA4, 64-wide waves, block = 4 waves).  A lane owns 4 adjacent columns (one 16-byte
// global_load_dwordx4 per packed row = 4 columns x e k-values) and R consecutive packed rows per step:
//     lane = (g = lane >> CQ, c = lane & (2^CQ - 1));  columns 4c..4c+3 of the tile;
//     rows  wave_base + step*G*R + g*R + i,  i < R,  G = 64 >> CQ row sub-groups per wave.
//   CQ = 2  ("narrow"): 16-column tiles (64-byte row segments), 16 row sub-groups per wave.  N/16 tiles fill
//           the chip without splitting K (4096 columns -> 256 blocks): no cross-block reduction at all.
//           Tiles 2p, 2p+1 (the two halves of one 128-byte line) are mapped to the same XCD (block b runs on
//           XCD b % 8) so the line is fetched into one L2 only.
//   CQ = 4  ("wide"): 64-column tiles (256-byte row segments), 4 row sub-groups; used when N/16 is too small or
//           K too large for the LDS copy of x; K is then split over gridDim.y and combined in-launch.
//   * weights go HBM -> VGPR directly (no LDS round trip: every byte is used by exactly one lane); loads for
//     the next step are issued before the current one is consumed (two register sets);
//   * x[k-slice] is staged in LDS once, "pair-permuted": one 32-bit LDS word is exactly the (k, k + e/2)
//     pair that one AND/OR magic-number unpack of a packed word produces (x is loaded before W is issued so
//     that its latency is not queued behind the weight stream);
//   * math: inside one quantisation group  sum_k x_k (a q_k + b) = a * sum_k x_k q_k + b * sum_k x_k, so the
//     inner loop is  shift -> AND/OR -> v_dot2c_f32_{f16,bf16}  on (OFF + q) pairs, fp32 accumulation; the
//     OFF * sum(x) term is removed once per run of R rows; (a, b) per W_group_mode from group_affine();
//   * reduction: xor-shuffles (ds_bpermute/DPP) over the row sub-groups of a wave, LDS over the 4 waves, and —
//     only if K was split — write-through (sc1) slabs + an arrival ticket; the last block adds the slabs in
//     fixed slice order (run-to-run deterministic), applies the epilogue and stores.
#include "gl_common.h"

namespace gl {

// Unpack geometry.  One AND/OR turns packed bits into two 16-bit floats (OFF + q * 2^(NBITS*i)) where i is
// the position of the element inside a WINDOW of consecutive bit fields that still fits the mantissa; the
// matching x pair is stored pre-scaled by 2^-(NBITS*i) (exact power of two), so only one shift per window
// (not per element) is needed:  fp16 (10-bit mantissa): 2 x 4-bit / 4 x 2-bit / 8 x 1-bit fields per window;
// bf16 (7-bit): 1 x 4-bit / 2 x 2-bit / 4 x 1-bit.
template <typename Tag, int NBITS>
struct Window {
    static constexpr int MANT = F16Traits<Tag>::DT == GEMLITE_DT_FP16 ? 10 : 7;
    static constexpr int HALF = 16 / NBITS;
    static constexpr int fit() {
        int wp = 1;
        while (wp * 2 <= HALF && (((1 << NBITS) - 1) << (NBITS * (wp * 2 - 1))) < (1 << MANT)) wp *= 2;
        return wp;
    }
    static constexpr int WP = fit();            // pairs per window
    static constexpr int NWIN = HALF / WP;      // windows (= shifts + 1) per packed word
};

__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t magic) {
    // plain C on purpose: an inline-asm v_and_or_b32 (one op instead of two) made hipcc's scheduler hoist all
    // unpacks ahead of their uses: 256 VGPRs + scratch spills instead of ~110 VGPRs
    return (a & mask) | magic;
}

template <typename Tag, int NBITS, int MB, int R, int CQ>
__global__ __launch_bounds__(256, ((MB * (16 / NBITS) >= 32 || R * MB >= 32) ? 1 : 2)) void gemv_wn_kernel(const WnParams p) {
    using TR = F16Traits<Tag>;
    using WN = Window<Tag, NBITS>;
    constexpr int E = 32 / NBITS;    // elements per packed word
    constexpr int HALF = E / 2;      // (k, k+HALF) pairs per word == LDS dwords per packed row
    constexpr int WP = WN::WP;
    constexpr int G = 64 >> CQ;      // row sub-groups per wave
    constexpr int CHUNK = G * R;     // packed rows one wave consumes per step
    constexpr int TC = 4 << CQ;      // tile columns
    static_assert(NBITS <= TR::MAX_QBITS, "q + OFF must be exact in the 16-bit float type");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & ((1 << CQ) - 1), g = lane >> CQ;
    int tile = blockIdx.x;
    if constexpr (CQ == 2) {  // adjacent half-line tiles on one XCD (speed only; any mapping is correct)
        const int nb = gridDim.x;
        if ((nb & 15) == 0) {
            const int xcd = tile & 7, idx = tile >> 3;
            tile = (((idx >> 1) << 3) + xcd) * 2 + (idx & 1);
        }
    }
    const int slice = blockIdx.y;
    const int n0 = tile * TC + c * 4;

    const int rows_slice = p.rows_per_slice;          // multiple of 4*CHUNK
    const int rows_wave = rows_slice >> 2;            // each wave takes a contiguous quarter
    const int row_s0 = slice * rows_slice;            // first packed row of the slice
    const int row_w0 = wave * rows_wave;              // wave start, relative to the slice
    const int pairs = rows_slice * HALF;              // LDS dwords per x row

    uint32_t* xs = (uint32_t*)smem;                               // [MB][pairs]
    float* red = (float*)(smem + (size_t)MB * pairs * 4);          // [4][MB][TC]
    unsigned* flag = (unsigned*)(red + 4 * MB * TC);

    // ---- x[k-slice]: global -> registers (pair-permuted, window-prescaled), issued AHEAD of the weights -----
    constexpr int XPT = 4;  // LDS dwords staged per thread per pass
    const uint16_t* xg = (const uint16_t*)p.x;
    const int64_t k0 = (int64_t)row_s0 * E;
    const int xtotal = MB * pairs;
    const int npass = (xtotal + 256 * XPT - 1) / (256 * XPT);
    auto fetch_x = [&](uint32_t (&v)[XPT], int pass) {
#pragma unroll
        for (int t = 0; t < XPT; ++t) {
            const int idx = (pass * XPT + t) * 256 + tid;
            v[t] = 0;
            if (idx < xtotal) {
                const int m = idx / pairs, pi = idx - m * pairs;
                const int word = pi / HALF, d = pi - word * HALF;
                if (m < p.M) {
                    const int64_t k = k0 + (int64_t)word * E + d;
                    v[t] = (uint32_t)xg[m * p.stride_xm + k] | ((uint32_t)xg[m * p.stride_xm + k + HALF] << 16);
                }
            }
        }
    };
    auto put_x = [&](const uint32_t (&v)[XPT], int pass) {
#pragma unroll
        for (int t = 0; t < XPT; ++t) {
            const int idx = (pass * XPT + t) * 256 + tid;
            if (idx < xtotal) {
                uint32_t val = v[t];
                if constexpr (WP > 1) {  // pre-scale the pair by 2^-(NBITS * position-in-window): exact
                    const int i = (idx % HALF) % WP;
                    const float sc = __builtin_bit_cast(float, (uint32_t)(127 - NBITS * i) << 23);
                    const float lo = TR::to_float((uint16_t)(val & 0xFFFFu)) * sc, hi = TR::to_float((uint16_t)(val >> 16)) * sc;
                    val = (uint32_t)TR::from_float(lo) | ((uint32_t)TR::from_float(hi) << 16);
                }
                xs[idx] = val;
            }
        }
    };

    // ---- weight + metadata stream: everything a chunk needs is requested together, one chunk ahead --------
    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    // unused metadata is still "loaded" (from the weight buffer, always in bounds) so the loop stays branch-free
    const uint16_t* sp = need_s ? (const uint16_t*)p.scales : (const uint16_t*)p.w;
    const uint16_t* zp = need_z ? (const uint16_t*)p.zeros : (const uint16_t*)p.w;
    const int64_t mstride = (need_s || need_z) ? p.stride_meta_g : 0;
    const uint32_t* wbase = p.w + (int64_t)(row_s0 + row_w0 + g * R) * p.stride_wk + n0;
    const int nchunks = rows_wave / CHUNK;  // 1, or even (planner)
    struct Chunk { u32x4 w[R]; u32x2 s, z; };
    auto load_chunk = [&](Chunk& ck, int chunk) {
        const int row = row_s0 + row_w0 + chunk * CHUNK + g * R;
        const int64_t grp = ((int64_t)row * E) / p.group_size;
#pragma unroll
        for (int i = 0; i < R; ++i) ck.w[i] = *(const u32x4*)(wbase + (int64_t)(chunk * CHUNK + i) * p.stride_wk);
        ck.s = *(const u32x2*)(sp + grp * mstride + n0);
        ck.z = *(const u32x2*)(zp + grp * mstride + n0);
    };

    Chunk A, B;
    {
        uint32_t xv[XPT];
        fetch_x(xv, 0);
        load_chunk(A, 0);
        if (nchunks > 1) load_chunk(B, 1);
        put_x(xv, 0);
        for (int pass = 1; pass < npass; ++pass) {  // large K * MB only
            fetch_x(xv, pass);
            put_x(xv, pass);
        }
    }
    __syncthreads();

    float tot[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) tot[m][j] = 0.f;

    const float scalar_zero = p.zero_is_scalar ? (float)((const int32_t*)p.zeros)[0] : 0.f;
    // (a, b) of group_affine() without per-column branches: a = s (s == 1 when unused),
    // b = bz * z * (mode 3 ? s : 1) with bz = -1 (modes 1, 3), +1 (mode 4), 0 otherwise
    const float bz = (p.w_mode == 1 || p.w_mode == 3) ? -1.f : (p.w_mode == 4 ? 1.f : 0.f);
    const bool b_times_s = p.w_mode == 3;
    const uint32_t magic = TR::MAGIC2;
    uint32_t wmask[WP];  // wave-uniform masks (SGPRs): field i of a window, both halves
#pragma unroll
    for (int i = 0; i < WP; ++i) wmask[i] = (((1u << NBITS) - 1u) * 0x00010001u) << (NBITS * i);

    auto unpack4 = [](const u32x2 raw) -> f32x4 {
        f32x4 r;
        const uint32_t v0 = raw[0], v1 = raw[1];
        r[0] = TR::to_float((uint16_t)(v0 & 0xFFFFu)); r[1] = TR::to_float((uint16_t)(v0 >> 16));
        r[2] = TR::to_float((uint16_t)(v1 & 0xFFFFu)); r[3] = TR::to_float((uint16_t)(v1 >> 16));
        return r;
    };

    auto compute = [&](const Chunk& ck, int chunk) {
        const int row_rel = row_w0 + chunk * CHUNK + g * R;  // first of this lane's R rows (slice-relative)
        float acc[MB][4], accx[MB][WP];  // accx[m][i]: sum of the STORED (pre-scaled) x of window position i
#pragma unroll
        for (int m = 0; m < MB; ++m) {
#pragma unroll
            for (int i = 0; i < WP; ++i) accx[m][i] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            constexpr int XW = HALF >= 4 ? 4 : HALF;  // x dwords fetched per LDS read
#pragma unroll
            for (int q = 0; q < HALF / XW; ++q) {
                // keep hipcc from hoisting every row's LDS reads + unpacks to the top (VGPR blow-up / spills)
                __builtin_amdgcn_sched_barrier(0);
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
                    const int d = q * XW + dd;          // pair index inside the word
                    const int win = d / WP, wi = d % WP;  // window, position in window
#pragma unroll
                    for (int m = 0; m < MB; ++m) accx[m][wi] = TR::dot2(xr[m][dd], TR::ONES2, accx[m][wi]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // two 16-bit floats  OFF + q_d * 2^(NBITS*wi),  OFF + q_{d+HALF} * 2^(NBITS*wi)
                        const uint32_t h = and_or(ck.w[i][j] >> (NBITS * WP * win), wmask[wi], magic);
#pragma unroll
                        for (int m = 0; m < MB; ++m) acc[m][j] = TR::dot2(h, xr[m][dd], acc[m][j]);
                    }
                }
            }
        }
        f32x4 s = unpack4(ck.s), z = unpack4(ck.z);
        if (!need_s) s = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (!need_z) z = (f32x4){scalar_zero, scalar_zero, scalar_zero, scalar_zero};
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float xs_stored = 0.f, xs_true = 0.f;  // sum of stored x (for the OFF term), sum of real x
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                xs_stored += accx[m][i];
                xs_true += accx[m][i] * (float)(1u << (NBITS * i));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = s[j];
                const float b = bz * z[j] * (b_times_s ? s[j] : 1.f);
                tot[m][j] += a * (acc[m][j] - TR::OFF * xs_stored) + b * xs_true;
            }
        }
    };

    if (nchunks == 1) {
        compute(A, 0);
    } else {
        for (int ch = 0; ch < nchunks; ch += 2) {  // nchunks is even; the tail re-requests the last chunk (unused)
            compute(A, ch);
            load_chunk(A, ch + 2 < nchunks ? ch + 2 : nchunks - 1);
            compute(B, ch + 1);
            load_chunk(B, ch + 3 < nchunks ? ch + 3 : nchunks - 1);
        }
    }

    // ---- reduce over the G row sub-groups of the wave (lane bits CQ..5) --------------------------------
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = tot[m][j];
#pragma unroll
            for (int off = 1 << CQ; off < 64; off <<= 1) v += __shfl_xor(v, off);
            tot[m][j] = v;
        }
    if (g == 0) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const f32x4 v = {tot[m][0], tot[m][1], tot[m][2], tot[m][3]};
            *(f32x4*)(red + (wave * MB + m) * TC + c * 4) = v;
        }
    }
    __syncthreads();

    // ---- across the 4 waves; output o = m*TC + col, o < MB*TC, strided over the 256 threads ------------
    constexpr int NOUT = MB * TC;
    constexpr int OPT = (NOUT + 255) / 256;  // outputs per thread
    float part[OPT];
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * 256;
        float v = 0.f;
        if (o < NOUT) {
            const int m = o / TC, col = o % TC;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[(w * MB + m) * TC + col];
        }
        part[it] = v;
    }
    if (p.splitk == 1) {
#pragma unroll
        for (int it = 0; it < OPT; ++it) {
            const int o = tid + it * 256;
            if (o < NOUT && (o / TC) < p.M) store_out_t<Tag>(p.epi, part[it], o / TC, (int64_t)tile * TC + (o % TC));
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
            if ((o / TC) < p.M) store_out_t<Tag>(p.epi, v, o / TC, (int64_t)tile * TC + (o % TC));
        }
    }
    if (tid == 0) splitk_reset(p.counters + tile);
}

// ---------------------------------------------------------------------------------------------
// host-side planning
// ---------------------------------------------------------------------------------------------
template <typename Tag, int NBITS, int MB>
static const void* pick_shape(int cq, int r) {
    if (cq == 2) {
        if constexpr (NBITS == 4 || NBITS == 2) {
            if (r == 8) return (const void*)gemv_wn_kernel<Tag, NBITS, MB, 8, 2>;
        }
        return nullptr;
    }
    if (cq == 3) {
        if constexpr (NBITS == 4 || NBITS == 2) {
            if (r == 4) return (const void*)gemv_wn_kernel<Tag, NBITS, MB, 4, 3>;
            if (r == 2) return (const void*)gemv_wn_kernel<Tag, NBITS, MB, 2, 3>;
        }
        return nullptr;
    }
    if (r == 4) return (const void*)gemv_wn_kernel<Tag, NBITS, MB, 4, 4>;
    if (r == 1) return (const void*)gemv_wn_kernel<Tag, NBITS, MB, 1, 4>;
    return nullptr;
}
template <typename Tag, int NBITS>
static const void* pick_mb(int mb, int cq, int r) {
    switch (mb) {
        case 1: return pick_shape<Tag, NBITS, 1>(cq, r);
        case 2: return pick_shape<Tag, NBITS, 2>(cq, r);
        default: return pick_shape<Tag, NBITS, 4>(cq, r);
    }
}
template <typename Tag>
static const void* pick_bits(int nbits, int mb, int cq, int r) {
    switch (nbits) {
        case 1: return pick_mb<Tag, 1>(mb, cq, r);
        case 2: return pick_mb<Tag, 2>(mb, cq, r);
        case 4: return pick_mb<Tag, 4>(mb, cq, r);
        case 8:
            if constexpr (F16Traits<Tag>::MAX_QBITS >= 8) return pick_mb<Tag, 8>(mb, cq, r);
            return nullptr;
        default: return nullptr;
    }
}

// Decide variant / grid / split-K / LDS for the GEMV kernel.  Returns false if this shape is not covered.
bool plan_gemv_wn(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    const int nbits = a.W_nbits;
    if (nbits != 1 && nbits != 2 && nbits != 4 && nbits != 8) return false;
    const int e = 32 / nbits;
    if (a.M > 4 || a.K % e != 0) return false;  // M >= 5 goes to the MFMA streaming kernel (16-row tiles)
    if (a.output_dtype != a.input_dtype) return false;  // typed epilogue
    const bool uses_s = a.W_group_mode >= 2 || a.channel_scale_mode == 1 || a.channel_scale_mode == 3;
    const bool has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (uses_s && a.meta_dtype != a.input_dtype) return false;
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != a.input_dtype) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    const int rows = (int)(a.K / e);
    const int64_t gs = p.group_size;
    if (gs % e != 0) return false;
    const int rpg = (int)(gs / e);  // packed rows per group
    const int mb = a.M <= 1 ? 1 : (a.M <= 2 ? 2 : 4);

    auto try_plan = [&](int cq, int want_splitk) -> bool {
        const int tc = 4 << cq, G = 64 >> cq;
        if (a.N % tc != 0) return false;
        int r;
        if (cq == 2) {
            r = 8;
        } else if (cq == 3) {
            r = (rpg % 4 == 0 && rows % (4 * G * 4) == 0) ? 4 : 2;
        } else {
            r = (rpg % 4 == 0 && rows % (4 * G * 4) == 0) ? 4 : 1;
            if (nbits == 1) r = 1;  // register budget (16 x-dwords per packed row)
        }
        if (rpg % r != 0) return false;
        const int block_rows = 4 * G * r;  // packed rows per block step
        if (rows % block_rows != 0) return false;
        const void* fn = a.input_dtype == GEMLITE_DT_FP16 ? pick_bits<half_tag>(nbits, mb, cq, r)
                                                           : pick_bits<bf16_tag>(nbits, mb, cq, r);
        if (!fn) return false;
        const int tiles = (int)(a.N / tc);
        const int units = rows / block_rows;  // max number of K slices
        auto ok = [&](int sk) {  // slices must divide the steps, leave 1 or an even number of steps per wave,
            if (sk < 1 || units % sk != 0) return false;  // and keep the LDS copy of x within 64 KiB
            const int steps = units / sk;
            if (!(steps == 1 || steps % 2 == 0)) return false;
            return (int64_t)mb * (rows / sk) * e * 2 <= 65536;
        };
        int splitk = 0;
        if (want_splitk > 0) {
            if (!ok(want_splitk)) return false;
            splitk = want_splitk;
        } else {
            const int target = cq == 2 ? 1 : 384;  // narrow tiles exist to avoid the split
            for (int sk = 1; sk <= units; sk *= 2) {
                if (!ok(sk)) continue;
                splitk = sk;
                if (tiles * sk >= target) break;
            }
            if (!splitk) return false;
        }
        if (tiles > MAX_SPLITK_COUNTERS && splitk > 1) return false;
        p.splitk = splitk;
        p.rows_per_slice = rows / splitk;
        lp.fn = fn;
        lp.name = cq == 2 ? "gemv_wn_kernel<tile16>" : (cq == 3 ? "gemv_wn_kernel<tile32>" : "gemv_wn_kernel<tile64>");
        lp.grid = dim3(tiles, splitk, 1);
        lp.block = dim3(256, 1, 1);
        lp.lds_bytes = (size_t)mb * p.rows_per_slice * (e / 2) * 4 + (size_t)4 * mb * tc * 4 + 16;
        lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * mb * tc * 4 : 0;
        lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
        return true;
    };

    // tuning[0]: 0 auto | 2 / 3 / 4 force 16- / 32- / 64-column tiles;  tuning[1]: 0 auto | n force split-K n
    const int force = a.tuning[0], sk = a.tuning[1];
    if (force >= 2 && force <= 4) return try_plan(force, sk);
    // auto: 128-byte row segments (32-column tiles) stream best (scripts/ubench/readbw.hip); 64-column tiles when
    // N is not a multiple of 32 columns... and 16-column tiles only when that is what makes the shape fit
    if (try_plan(3, sk)) return true;
    if (try_plan(4, sk)) return true;
    return try_plan(2, sk);
}

}  // namespace gl
