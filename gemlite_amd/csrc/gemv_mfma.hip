// gemv_mfma.hip — decode (M <= 16) for packed 4- / 2-bit weights ON THE MATRIX CORE (round 3).
//
// Replaces gemv_INT_revsplitK_kernel / gemv_INT_kernel (gemlite/triton_kernels/gemv_revsplitK_kernels.py:226-462,
// gemv_kernels.py:230-388) at M = 1 and gemm_splitK_INT_kernel (gemm_splitK_kernels.py:277-450) for a handful of rows.
//
// Why: the round-3 timeline of the dot-product GEMV (scripts/timeline_decode.py, profiles/r03/timeline_decode_v1.log) showed
// that at 4096 x 4096 a block's first wave has finished its arithmetic 1.2 us after the block started, but the block lives
// 2.5 us: the weights arrive within ~1.2 us at close to the HBM rate, and then the SIMD still has to issue the ~260 VALU
// instructions of each of its four waves (unpack, v_dot2c, pairing / pre-scaling x per lane, group epilogue).  The kernel
// was VALU-bound, not bandwidth-bound.  Here
//   * one packed word IS one lane's B fragment of v_mfma_f32_16x16x32 (8 consecutive k of one column): 5 VALU turn it into
//     8 raw integer codes (fp16: the masked bit fields used as subnormals, exact; bf16: 128 + q through OR 0x4300), the
//     matrix core multiplies and adds, and scale / zero are applied once per quantisation group to the accumulator;
//   * x is prepared ONCE per block: the rows are pair-permuted / pre-scaled into LDS (16 bytes per packed row and
//     activation row, the exact A fragment of lane (m, k-quarter)) together with the per-group sums of x that the zero term
//     needs; the A fragment of a k-step is then one ds_read_b128 — no per-lane x arithmetic, no x traffic per MFMA
//     (gemm_wn_direct.hip fetches 16 bytes of x per lane and MFMA from L2: 4x the weight bytes through the address path);
//   * tile = 16 V columns (V = 1 / 2 / 4 packed words per lane: 64- / 128- / 256-byte row segments), K is never split
//     across blocks; the NW waves of a block deal the quantisation groups round-robin, every wave keeps GB groups of
//     requests in flight (all of its K range at 4096), non-temporal loads.
// Numerics: fp32 accumulation in the matrix core, raw codes are exact, scale / zero in fp32 per group.
#include "gl_common.h"

namespace gl {

namespace gmf {

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

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int V> struct Vec;
template <> struct Vec<1> { typedef uint32_t W; typedef uint16_t M; };
template <> struct Vec<2> { typedef u32x2 W; typedef uint32_t M; };
template <> struct Vec<4> { typedef u32x4 W; typedef u32x2 M; };
// V packed words / V 16-bit metadata values of one lane: one buffer load each (voffset per lane, soffset per request)
template <int V>
__device__ __forceinline__ typename Vec<V>::W ldw(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    if constexpr (V == 1) return __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 2);
    else if constexpr (V == 2) return __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 2);
    else return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 2);
}
template <int V>
__device__ __forceinline__ typename Vec<V>::M ldm(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    if constexpr (V == 1) return (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0);
    else if constexpr (V == 2) return __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
    else return __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
}
template <int V>
__device__ __forceinline__ uint16_t meta_of(const typename Vec<V>::M& m, int j) {
    if constexpr (V == 1) return m;
    else if constexpr (V == 2) return (uint16_t)(m >> (16 * j));
    else return (uint16_t)(m[j >> 1] >> (16 * (j & 1)));
}
template <int V>
__device__ __forceinline__ uint32_t word_of(const typename Vec<V>::W& w, int j) {
    if constexpr (V == 1) return w; else return w[j];
}

// B fragments (round 6: EXACT in fp16 — rounds 3-5 divided the matching x slots by 16 / 4^a in fp16, which lost mantissa bits of small
// activations).  A lane (column c, k-quarter kg) holds the words wa, wb of TWO packed rows (4 apart) of its column; one MFMA
// takes 8 of their k-values, chosen so that every code of the fragment sits at the SAME bit offset inside its 16-bit half:
//   4-bit, plane 0: {wa, wa >> 8, wb, wb >> 8} & 0x000F000F = (q0,q4)(q2,q6) of row a, then of row b            [fp16: x 2^-24, exact subnormals]
//          plane 1: the same & 0x00F000F0 = 16 (q1,q5)(q3,q7)                                                    [fp16: x 16 x 2^-24]
//   2-bit, plane a = 0..3: {wa, wa >> 8, wb, wb >> 8} & (0x00030003 << 2a) = 4^a (q_a, q_{a+8})(q_{a+4}, q_{a+12})
// fp16 keeps ONE fp32 accumulator per plane and applies the plane's 16^-a / 4^-a once to the sum — nothing is rounded before the matrix
// core's fp32 accumulation.  bf16 (no subnormal trick: 8-bit mantissa): the fields shifted down and OR-ed into 128.0 (0x4300 | q = 128 + q,
// exact), all planes into one accumulator, the 128 * sum(x) removed per group as before.  The A fragment of (row pair, plane, kg) is the
// matching x pairs of the two rows, laid out by the staging lanes (below) so that it is ONE ds_read_b128.
template <typename Tag, int NBITS>
struct Planes {
    static constexpr bool SUBN = F16Traits<Tag>::DT == GEMLITE_DT_FP16;
    static constexpr int NA = NBITS == 4 ? 2 : 4;          // planes per row pair
    static constexpr int FW = NBITS;                       // field width
    static __device__ __forceinline__ u32x4 frag(uint32_t wa, uint32_t wb, int a) {
        u32x4 r;
        if constexpr (SUBN) {
            const uint32_t m = (((1u << FW) - 1u) * 0x00010001u) << (FW * a);
            r[0] = wa & m; r[1] = (wa >> 8) & m; r[2] = wb & m; r[3] = (wb >> 8) & m;
        } else {
            const uint32_t m = ((1u << FW) - 1u) * 0x00010001u;
            r[0] = ((wa >> (FW * a)) & m) | F16Traits<Tag>::MAGIC2;
            r[1] = ((wa >> (FW * a + 8)) & m) | F16Traits<Tag>::MAGIC2;
            r[2] = ((wb >> (FW * a)) & m) | F16Traits<Tag>::MAGIC2;
            r[3] = ((wb >> (FW * a + 8)) & m) | F16Traits<Tag>::MAGIC2;
        }
        return r;
    }
    static __device__ __forceinline__ constexpr float inv_scale(int a) { return SUBN ? 1.0f / (float)(1 << (FW * a)) : 1.0f; }
};

}  // namespace gmf

// MB = 1: a single activation row (every lane reads row 0's fragment: LDS broadcast); MB = 4: up to 4 rows.
// GB = group units of requests a wave keeps in flight per batch; SPG = MFMA k-steps (32 k) per quantisation group (4: >= 128, 2: 64).
//
// x never crosses waves: a wave needs exactly the x of ITS k range, 16 bytes per (row, packed row) = CPB <= 64 chunks per batch,
// so lane L fetches chunk L of the batch (first request of the batch, ahead of the weights), pair-permutes / pre-scales it in
// registers, drops the pairs into the wave's private LDS slot in A-fragment order and reads the fragments back as broadcasts.  No block barrier before
// the arithmetic: the first version staged all of x per BLOCK, and its barrier waited 3.3 us — the x requests of the later
// waves queue in the CU's in-order memory path behind the weight requests of the earlier ones (profiles/r03/timeline_decode_v2.log).
template <typename Tag, int NBITS, int V, int NW, int MB, int GB, int SPG>
__global__ __launch_bounds__(NW * 64, 1) void gemv_mfma_kernel(const WnParams p) {
    using namespace gmf;
    using TR = F16Traits<Tag>;
    using PL = Planes<Tag, NBITS>;
    using WT = typename Vec<V>::W;
    using MT = typename Vec<V>::M;
    constexpr bool SUBN = PL::SUBN;
    constexpr int E = 32 / NBITS;
    static_assert(NBITS == 4 || NBITS == 2, "x staging below: 4-bit words (one chunk per packed row) or 2-bit words (two)");
    constexpr int TN = 16 * V, NT = NW * 64;
    constexpr int RSTEPS = SPG * 8 / E;        // wave loads (4 packed rows = 4 E k each) per group unit: 4 / 2 (4-bit), 2 (2-bit)
    constexpr int P = RSTEPS / 2;              // row pairs per lane and unit
    constexpr int NA = PL::NA, NACC = SUBN ? NA : 1;
    constexpr int CPG = 4 * SPG;               // x chunks (8 k) per group unit
    constexpr int CPB = GB * CPG;              // ... per batch: one per lane
    static_assert(RSTEPS >= 2 && RSTEPS % 2 == 0 && CPB <= 64 && GB * P * NA * 64 <= 1024, "a batch's x chunks are dealt one per lane; its fragments fill <= 1 KiB per row");
    constexpr float QSCALE = SUBN ? 16777216.0f : 1.0f;
    constexpr float OFF = SUBN ? 0.0f : TR::OFF;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kg = lane >> 4;
    const int tile = blockIdx.x;
    const int n0 = tile * TN + c * V;
    const int K = p.K, M = p.M;
    const int ngroups = K >> 5 >> (SPG == 4 ? 2 : 1);  // group units of SPG k-steps
    unsigned char* xw = smem + (size_t)wave * (2 * MB * 1024);            // this wave's x slots: [2 buffers][MB rows][64 chunks x 16 B]
    float* red = (float*)(smem + (size_t)NW * (2 * MB * 1024));           // [NW][64][4 V]

    // opt-in timeline (tuning[3] & 4, needs a workspace): wave 0 of every block stores the 100 MHz global clock at
    // [start, first batch requested, arithmetic done, output stored] (scripts/timeline_decode.py)
    const bool probe = (p.flags & 4) && p.counters && wave == 0 && lane == 0 && blockIdx.x < 1024;
    unsigned long long* stamps = (unsigned long long*)(p.counters + MAX_SPLITK_COUNTERS) + blockIdx.x * 4;
    auto stamp = [&](int i) {
        if (probe) stamps[i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);

    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    const uint16_t* sp = need_s ? (const uint16_t*)p.scales : (const uint16_t*)p.w;
    const uint16_t* zp = need_z ? (const uint16_t*)p.zeros : (const uint16_t*)p.w;
    const uint32_t mstride = (need_s || need_z) ? (uint32_t)p.stride_meta_g : 0u;
    const uint32_t sw4 = (uint32_t)p.stride_wk * 4u;
    // Buffer loads: the per-lane part of every address is ONE register computed once, the per-request part (group unit, wave
    // load) is a scalar offset — no vector arithmetic per request (the first version computed a 64-bit address per load: ~250
    // instructions per wave before its last request was out).  Loads are unconditional (a branch around a load makes every
    // later wait vmcnt(0)); units past the end re-read the wave's last unit and count zero.  aux 2 = nt.
    const int rows_x = M < MB ? M : MB;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, (short)0, (int)(((int64_t)(rows_x - 1) * p.stride_xm + K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)(uint32_t)((int64_t)(K / E) * sw4), 0x00020000);
    const int meta_rows = p.gs_shift >= 31 ? 1 : (K >> p.gs_shift);
    const int meta_bytes = (int)(((int64_t)(meta_rows - 1) * mstride + p.N) * 2);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)sp, (short)0, need_s ? meta_bytes : 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void*)zp, (short)0, need_z ? meta_bytes : 16, 0x00020000);
    const uint32_t wvoff = (uint32_t)kg * sw4 + (uint32_t)n0 * 4u;
    const uint32_t mvoff = (need_s || need_z) ? (uint32_t)n0 * 2u : 0u;
    // x chunk of lane L in a batch starting at the wave's unit index i0: unit gi = L / CPG of the batch, chunk L % CPG of it;
    // unit u = wave + NW (i0 + gi) covers chunks [u CPG, (u + 1) CPG) of the row
    const int xgi = lane / CPG, xwi = lane % CPG;
    // 4-bit: chunk xwi of the unit = the 8 k of packed row xwi, 16 contiguous bytes.  2-bit: chunk (k-step ks = xwi >> 2, quarter kq)
    // = fragment f = ks & 1 of packed row (ks >> 1) 4 + kq (16 k): its k are {4 f .. 4 f + 3} and {8 + 4 f .. 8 + 4 f + 3} — two 8-byte pieces
    const uint32_t xvoff = lane >= CPB ? 0x80000000u  // (lanes past CPB: no request)
                           : (NBITS == 4 ? (uint32_t)(((wave + NW * xgi) * CPG + xwi) * 16)
                                         : (uint32_t)((wave + NW * xgi) * CPG * 16 + (((xwi >> 3) * 4 + (xwi & 3)) * 32) + ((xwi >> 2) & 1) * 8));
    // where this lane's pairs go: fragment (unit gi, row pair t, plane a, quarter kq) at ((((gi P + t) NA + a) 4 + kq) 16 bytes; the first 8 bytes
    // belong to the pair's first row, the last 8 to its second
    //   4-bit chunk = packed row rs 4 + kq (rs = xwi >> 2): t = rs >> 1, second row if rs & 1; pairs (x0,x4)(x2,x6) -> plane 0, (x1,x5)(x3,x7) -> plane 1
    //   2-bit chunk (row rs 4 + kq, half f): t = 0, second row if rs; pair i = (x_{4f+i}, x_{4f+i+8}) -> plane i, dword f of the row's 8 bytes
    const int st_rs = NBITS == 4 ? (xwi >> 2) : (xwi >> 3), st_kq = xwi & 3, st_f = (xwi >> 2) & 1;
    const uint32_t st_base = (uint32_t)((((xgi * P + (NBITS == 4 ? (st_rs >> 1) : 0)) * NA) * 4 + st_kq) * 16 + (st_rs & 1) * 8 + (NBITS == 4 ? 0 : st_f * 4));
    const int my_units = (ngroups - wave + NW - 1) / NW;

    struct Batch { u32x4 x[MB]; WT w[GB][RSTEPS]; MT s[GB], z[GB]; };
    auto load_batch = [&](Batch& b, int i0) {
        // x of units past the wave's end: offsets beyond K read zeros through the descriptor's range check (rows share one range:
        // a chunk index past the row's end may land in the next row — harmless, that unit counts zero)
        const uint32_t xso = (uint32_t)__builtin_amdgcn_readfirstlane(NW * i0 * CPG * 16);
#pragma unroll
        for (int r = 0; r < MB; ++r) {
            const uint32_t rso = (uint32_t)__builtin_amdgcn_readfirstlane((int)((int64_t)(r < rows_x ? r : rows_x - 1) * p.stride_xm * 2));
            if constexpr (NBITS == 4) b.x[r] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xvoff, xso + rso, 0);
            else {
                const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(rsX, xvoff, xso + rso, 0);
                const u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(rsX, xvoff + 16u, xso + rso, 0);
                b.x[r] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
            }
        }
#pragma unroll
        for (int gi = 0; gi < GB; ++gi) {
            const int iu = i0 + gi < my_units ? i0 + gi : my_units - 1;
            const int u = wave + NW * iu;
            const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(u * (4 * RSTEPS)) * sw4;
#pragma unroll
            for (int rs = 0; rs < RSTEPS; ++rs) b.w[gi][rs] = gmf::ldw<V>(rsW, wvoff, so + (uint32_t)(4 * rs) * sw4);
            const uint32_t mo = (uint32_t)__builtin_amdgcn_readfirstlane(group_of(u * 32 * SPG, p.gs_shift)) * mstride * 2u;
            b.s[gi] = gmf::ldm<V>(rsS, mvoff, (need_s || need_z) ? mo : 0u);
            b.z[gi] = gmf::ldm<V>(rsZ, mvoff, (need_s || need_z) ? mo : 0u);
        }
    };
    // pair-permute (+ pre-scale) the lane's chunk into A-fragment order, store it in buffer `buf`; returns the sum of the TRUE
    // x over the chunk's group unit (butterfly over the CPG lanes of the unit: fixed order, every lane of the unit gets it)
    auto stage = [&](const u32x4 v, int buf, int r) -> float {
        // 4-bit: v = the row's 8 k (x0x1)(x2x3)(x4x5)(x6x7) -> pairs (x_a, x_{a+4});  2-bit: v = {low piece (x0x1)(x2x3), high piece
        // (x8x9)(x10x11)} of the fragment -> pairs (x_a, x_{a+8}): the same four permutes
        uint32_t q[4];
        q[0] = __builtin_amdgcn_perm(v[2], v[0], 0x05040100u);
        q[1] = __builtin_amdgcn_perm(v[2], v[0], 0x07060302u);
        q[2] = __builtin_amdgcn_perm(v[3], v[1], 0x05040100u);
        q[3] = __builtin_amdgcn_perm(v[3], v[1], 0x07060302u);
        float sm = 0.f;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) sm = TR::dot2(q[dd], TR::ONES2, sm);
        if (lane < CPB) {
            unsigned char* dst = xw + (size_t)(buf * MB + r) * 1024 + st_base;
            if constexpr (NBITS == 4) {
                *(u32x2*)(dst) = (u32x2){q[0], q[2]};
                *(u32x2*)(dst + 64) = (u32x2){q[1], q[3]};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) *(uint32_t*)(dst + i * 64) = q[i];
            }
        }
        sm += gmf::dppf<0xB1>(sm);   // quad_perm [1,0,3,2]
        sm += gmf::dppf<0x4E>(sm);   // quad_perm [2,3,0,1]: every lane holds its quad's sum
        sm += gmf::dppf<0x141>(sm);  // row_half_mirror: + the other quad of the 8-lane half
        if constexpr (CPG == 16) sm += gmf::dppf<0x140>(sm);  // row_mirror: + the other half
        return sm;
    };

    f32x4 tot[V];
#pragma unroll
    for (int j = 0; j < V; ++j) tot[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (read into a scalar register and waited for HERE: a vector load left pending into the K loop is answered with `s_waitcnt vmcnt(0)` at its
    //  first use in EVERY iteration — the loop-head drain of rounds 3-5, see gemv_wn.hip)
    float scalar_zero = 0.f;
    if (p.zero_is_scalar) scalar_zero = (float)__builtin_amdgcn_readfirstlane(((const int32_t*)p.zeros)[0]);
    const float bz = (p.w_mode == 1 || p.w_mode == 3) ? -1.f : (p.w_mode == 4 ? 1.f : 0.f);
    const bool b_times_s = p.w_mode == 3;
    const int mrow = MB == 1 ? 0 : (c < MB ? c : MB - 1);  // A row of this lane (rows past the tile repeat the last one; never stored)

    // sums of x per (row, group unit of the batch): wave-uniform -> scalar registers
    float gs[MB][GB];
    auto stage_batch = [&](const Batch& b, int buf) {
#pragma unroll
        for (int r = 0; r < MB; ++r) {
            const float sm = stage(b.x[r], buf, r);
#pragma unroll
            for (int gi = 0; gi < GB; ++gi)
                gs[r][gi] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sm), gi * CPG));
        }
    };
    auto compute = [&](const Batch& b, int i0, int buf) {
        const unsigned char* xa = xw + (size_t)(buf * MB + mrow) * 1024 + (size_t)kg * 16;
#pragma unroll
        for (int gi = 0; gi < GB; ++gi) {
            f32x4 acc[V][NACC];
#pragma unroll
            for (int t = 0; t < P; ++t)
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const u32x4 af = *(const u32x4*)(xa + (((gi * P + t) * NA + a) * 4) * 16);  // fragment (unit gi, pair t, plane a, quarter kg)
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        const u32x4 bf = PL::frag(word_of<V>(b.w[gi][2 * t], j), word_of<V>(b.w[gi][2 * t + 1], j), a);
                        const int ai = SUBN ? a : 0;
                        const bool first = t == 0 && (SUBN || a == 0);
                        acc[j][ai] = mfma16<Tag>(af, bf, first ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[j][ai]);
                    }
                }
            const float live = i0 + gi < my_units ? 1.f : 0.f;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float s = need_s ? TR::to_float(gmf::meta_of<V>(b.s[gi], j)) : 1.f;
                const float z = need_z ? TR::to_float(gmf::meta_of<V>(b.z[gi], j)) : scalar_zero;
                const float a = s * QSCALE * live;
                const float bb = (bz * z * (b_times_s ? s : 1.f) - s * OFF) * live;
                // C layout: rows 4 kg + r — the tile's rows (< MB <= 4) live in the registers of the kg = 0 lanes
#pragma unroll
                for (int r = 0; r < MB; ++r) {
                    float v = acc[j][0][r];
#pragma unroll
                    for (int ai = 1; ai < NACC; ++ai) v = __builtin_fmaf(acc[j][ai][r], PL::inv_scale(ai), v);
                    tot[j][r] += a * v + bb * gs[r][gi];
                }
            }
        }
    };

    // Two batches alternate STATICALLY (A <-> LDS buffer 0, B <-> buffer 1): the requests of the batch after next go out as soon as
    // the arithmetic on a batch has released its registers.  (The first version copied `cur = nxt` at the end of every iteration:
    // the copy reads the registers of requests still in flight, so every batch was waited for in full before the next one was
    // issued — one batch of requests per latency.  Measured effect of the change: none beyond noise (16384^2: 25.3 -> 25.8 us, 8192^2 9.9 -> 9.8);
    // kept because it is the structure the comment above describes.)
    Batch A, B;
    load_batch(A, 0);
    stamp(1);
    stage_batch(A, 0);
    int i0 = 0;
#pragma unroll 1
    while (true) {
        const bool more_b = i0 + GB < my_units;
        if (more_b) load_batch(B, i0 + GB);
        compute(A, i0, 0);
        if (!more_b) break;
        stage_batch(B, 1);
        i0 += GB;
        const bool more_a = i0 + GB < my_units;
        if (more_a) load_batch(A, i0 + GB);
        compute(B, i0, 1);
        if (!more_a) break;
        stage_batch(A, 0);
        i0 += GB;
    }
    stamp(2);

    // ---- the NW waves (disjoint K) meet in LDS; C layout of 16x16: column c, rows 4 kg + r ----------------------------------
#pragma unroll
    for (int j = 0; j < V; ++j) *(f32x4*)(red + ((wave * 64 + lane) * V + j) * 4) = tot[j];
    __syncthreads();
    for (int o = tid; o < MB * TN; o += NT) {
        // output (m, n): column n = cc V + j lives in lane cc (kg = 0), register m
        const int m = o / TN, n = o - m * TN;
        const int cc = n / V, j = n - cc * V;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[((w * 64 + cc) * V + j) * 4 + m];
        if (m < M) store_out_t<Tag>(p.epi, v, m, (int64_t)tile * TN + n);
    }
    stamp(3);
}

// ---------------------------------------------------------------------------------------------------------------------
// host-side planning.  tuning[0]: 0 auto | 21 / 22 / 24 = 1 / 2 / 4 words per lane (16 / 32 / 64 columns);  tuning[2]: 0 auto | 4 / 8 / 16 waves
// ---------------------------------------------------------------------------------------------------------------------
typedef void (*gmf_fn)(const WnParams);
template <typename Tag, int NB, int V, int MB, int SPG>
static gmf_fn gmf_pick_nw(int nw) {
    // group units per batch: what keeps >= 64 KB of weight requests in flight per CU (8 waves) within the register budget —
    // a unit is RSTEPS = 4 (4-bit) / 2 (2-bit) wave loads of V words
    constexpr int GB = V == 4 ? 2 : 4;  // (4 units with 2-bit 64-column tiles measured slower: 20.1 vs 18.0 us at 16384^2)
    switch (nw) {
        case 4: return gemv_mfma_kernel<Tag, NB, V, 4, MB, GB, SPG>;
        case 8: return gemv_mfma_kernel<Tag, NB, V, 8, MB, GB, SPG>;
        case 16:  // 1024 threads: 128 registers per lane — only the one-word tile fits without spilling
            if constexpr (V == 1 && NB == 4 && MB == 1) return gemv_mfma_kernel<Tag, NB, V, 16, MB, GB, SPG>;
            else return nullptr;
        default: return nullptr;
    }
}
template <typename Tag, int NB, int MB, int SPG>
static gmf_fn gmf_pick_v(int v, int nw) {
    switch (v) {
        case 1: return gmf_pick_nw<Tag, NB, 1, MB, SPG>(nw);
        case 2: return gmf_pick_nw<Tag, NB, 2, MB, SPG>(nw);
        case 4: return gmf_pick_nw<Tag, NB, 4, MB, SPG>(nw);
        default: return nullptr;
    }
}
template <typename Tag>
static gmf_fn gmf_pick(int nbits, int v, int nw, int mb, int spg) {
    if (nbits == 2) {  // 2-bit words: groups of >= 128 only (a 64-k group is half a wave load), one activation row
        if (spg != 4 || mb != 1) return nullptr;
        return gmf_pick_v<Tag, 2, 1, 4>(v, nw);
    }
    if (spg == 4) return mb == 1 ? gmf_pick_v<Tag, 4, 1, 4>(v, nw) : gmf_pick_v<Tag, 4, 4, 4>(v, nw);
    return mb == 1 ? gmf_pick_v<Tag, 4, 1, 2>(v, nw) : gmf_pick_v<Tag, 4, 4, 2>(v, nw);
}

bool plan_gemv_mfma(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    if (a.W_nbits != 4 && a.W_nbits != 2) return false;
    if (a.M < 1 || a.M > 4) return false;
    if (a.output_dtype != a.input_dtype) return false;  // typed epilogue / metadata
    if (a.input_dtype != GEMLITE_DT_FP16 && a.input_dtype != GEMLITE_DT_BF16) return false;
    // metadata read in the K loop: the kernel's 16-bit type; channel scales of the epilogue alone: any float type (store_out_t; round 4:
    // BitNet's fp32 scale, A16W158_INT)
    const bool loop_s = a.W_group_mode >= 2, post_s = a.channel_scale_mode == 1 || a.channel_scale_mode == 3;
    (void)post_s;
    const bool has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (loop_s && a.meta_dtype != a.input_dtype) return false;
    if (post_s && !loop_s && a.meta_dtype != a.input_dtype && a.meta_dtype != GEMLITE_DT_FP32 && a.meta_dtype != GEMLITE_DT_FP16 && a.meta_dtype != GEMLITE_DT_BF16) return false;
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != a.input_dtype) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    if ((a.stride_xm * 2) % 16 != 0 || ((uintptr_t)a.x % 16) != 0) return false;  // 16-byte x chunks
    if (((uintptr_t)a.w_q % 16) != 0 || (a.stride_wk % 4) != 0) return false;      // vector weight loads
    const int64_t gs = p.group_size;
    int spg;
    if (gs % 128 == 0) spg = 4;
    else if (gs == 64) spg = 2;
    else return false;
    if (a.K % (32 * spg) != 0) return false;
    const int64_t rows = a.K / (32 / a.W_nbits);
    if (rows * a.stride_wk * 4 + a.N * 4 >= (1ll << 32) || (int64_t)(a.K / gs) * p.stride_meta_g + a.N >= (1ll << 32)) return false;  // 32-bit offsets
    const int mb = a.M == 1 ? 1 : 4;
    const int ngroups = (int)(a.K / (32 * spg));
    // tile width: the widest tile that gives >= 256 blocks (one per CU) — K is never split across blocks
    int v = 0;
    if (a.tuning[0] == 21 || a.tuning[0] == 22 || a.tuning[0] == 24) v = a.tuning[0] - 20;
    else if (a.tuning[0] != 0) return false;  // the dot-product family's tile codes (2 / 3 / 4)
    else {
        // round 3 sweep (profiles/r03/probe_m1_llm_shapes_*.log): for 4-bit decode fewer, wider blocks win — 160 blocks, 128 over a
        // short K (8960 x 1536: 64-column tiles 5.1 vs 8.1 us for 32-column ones; 6144 x 4096: 32-column 5.6 vs 8.1 for 16-column)
        const int64_t want_blocks = a.K <= 2048 ? 128 : 160;
        for (int cand : {4, 2, 1})
            if (a.N % (16 * cand) == 0 && a.N / (16 * cand) >= want_blocks) { v = cand; break; }
        // narrow matrices: 16-column tiles whatever the block count for 2 .. 4 rows (1536 x 8960, M = 4: 10.7 us against 14.9 for the
        // 32-row MFMA tile the K = 8960 shapes used to fall to; 2560 x 9728: 12.0 vs 15.9) — at M = 1 the K-splitting kernels
        // (2-bit, M = 1: the same — 1024 x 4096 3.7 vs 8.1 us, 3072 x 8192 6.2 vs 8.1, profiles/r03/probe_w2_m1_llm_shapes_*.log)
        if (!v && ((a.W_nbits == 4 && a.M >= 2) || a.W_nbits == 2) && a.N % 16 == 0) v = 1;
        if (!v) return false;
    }
    if (a.N % (16 * v) != 0) return false;
    int nw = a.tuning[2] == 4 || a.tuning[2] == 8 || a.tuning[2] == 16 ? a.tuning[2] : 8;
    if (a.tuning[2] != 0 && nw != a.tuning[2]) return false;
    if (nw == 16 && (v != 1 || mb != 1 || a.W_nbits != 4)) nw = 8;
    while (nw > 4 && ngroups < nw) nw >>= 1;
    const size_t lds = (size_t)nw * 2 * mb * 1024 + (size_t)nw * 64 * 4 * v * 4;
    const bool f16 = a.input_dtype == GEMLITE_DT_FP16;
    gmf_fn fn = f16 ? gmf_pick<half_tag>(a.W_nbits, v, nw, mb, spg) : gmf_pick<bf16_tag>(a.W_nbits, v, nw, mb, spg);
    if (!fn) return false;
    p.splitk = 1;
    p.rows_per_slice = (int)rows;
    lp.fn = (const void*)fn;
    static const char* names2[3] = {"gemv_w2_mfma_kernel<tile16>", "gemv_w2_mfma_kernel<tile32>", "gemv_w2_mfma_kernel<tile64>"};
    static const char* names[2][3] = {{"gemv_mfma_kernel<tile16>", "gemv_mfma_kernel<tile32>", "gemv_mfma_kernel<tile64>"},
                                      {"gemv_mfma_kernel<tile16,rows4>", "gemv_mfma_kernel<tile32,rows4>", "gemv_mfma_kernel<tile64,rows4>"}};
    lp.name = names[mb == 1 ? 0 : 1][v == 1 ? 0 : (v == 2 ? 1 : 2)];
    if (a.W_nbits == 2) lp.name = names2[v == 1 ? 0 : (v == 2 ? 1 : 2)];
    lp.grid = dim3((unsigned)(a.N / (16 * v)), 1, 1);
    lp.block = dim3(64 * nw, 1, 1);
    lp.lds_bytes = lds;
    lp.slab_bytes = 0;
    lp.ws_bytes = 0;
    return true;
}

}  // namespace gl
