// gemv_wn.hip — fused unpack + group-dequant + GEMV for packed low-bit weights, M = 1 (decode).
//
// Replaces the reference's gemv_INT_revsplitK_kernel / gemv_INT_kernel / gemv_INT_splitK_kernel
// (gemlite/triton_kernels/gemv_revsplitK_kernels.py:226-462, gemv_kernels.py:230-388,
// gemv_splitK_kernels.py:240-420) for packed int32 weights.  HBM-bound: the only large stream is W_q.
//
// Mapping (CDNA4, 64-wide waves, block = 4 waves; 8 / 16 for the short-K variant, see NW below).  A lane owns 4 adjacent columns (one 16-byte
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

#ifndef GL_GEMV_NT
#define GL_GEMV_NT 1  // -DGL_GEMV_NT=0: default-policy weight loads (A/B builds)
#endif

namespace gl {

const void* gemv_w4_decode3_fn(int tag, bool nt);  // gemv_decode.hip

// Round 6 ("GEMV family onto counted asm loads"): the chunk requests of gemv_wn_kernel are inline-asm global loads
// retired with hand-counted s_waitcnt (gl_async.h has the reasoning; scripts/isa_asmloads.py replays the generated code).  With the loads
// left to hipcc, every instantiation carried a full `s_waitcnt vmcnt(0)` at the head of its two-chunk loop: the chunk requested a moment
// before was waited for in full before the arithmetic on the OLDER chunk could start — one memory latency per loop iteration with nothing
// overlapped.  Uniform base + 32-bit byte offset per lane, as before.
namespace gvw {
template <bool NT>
__device__ __forceinline__ void gld128(u32x4& d, const char* base, uint32_t voff) {
    if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(d) : "v"(voff), "s"(base) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void gld64(u32x2& d, const char* base, uint32_t voff) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <typename T>
__device__ __forceinline__ void tie(T& v) { asm volatile("" : "+v"(v)); }

// Two chunk buffers A / B, one chunk in flight while the other is computed on.  NLC = requests per chunk.  On entry A holds chunk 0 and B
// chunk 1 (requested by the caller, those that exist).  Steady state: unconditional requests, ONE wait value ("everything but the newest
// chunk has landed": loads return in order); the last one to three chunks are peeled, their waits picked by uniform branches.
template <int NLC, typename Buf, typename Issue, typename Tie, typename Compute>
__device__ __forceinline__ void ring2_run(int n, Buf& A, Buf& B, Issue&& issue, Tie&& tie_buf, Compute&& compute) {
    int i = 0;
#pragma unroll 1
    for (; i + 4 <= n; i += 2) {
        wait_vm<NLC>();
        tie_buf(A);
        compute(A, i);
        issue(A, i + 2);
        wait_vm<NLC>();
        tie_buf(B);
        compute(B, i + 1);
        issue(B, i + 3);
    }
    const int rem = n - i;  // 0 (no chunk at all), 1, 2 or 3
    if (rem >= 1) {
        if (rem >= 2) wait_vm<NLC>(); else wait_vm<0>();
        tie_buf(A);
        compute(A, i);
        if (rem >= 2) {
            if (rem == 3) { issue(A, i + 2); wait_vm<NLC>(); } else wait_vm<0>();
            tie_buf(B);
            compute(B, i + 1);
            if (rem == 3) {
                wait_vm<0>();
                tie_buf(A);
                compute(A, i + 2);
            }
        }
    }
}
}  // namespace gvw

// Unpack geometry.  One AND turns packed bits into two 16-bit floats q * 2^(NBITS*i) (i = position of the
// element inside a WINDOW of consecutive bit fields that still fits the mantissa); the matching x pair is stored
// pre-scaled by 2^-(NBITS*i) (an exact power of two), so only one shift per window (not per element) is needed.
//   fp16 (10-bit mantissa): the masked bits are used as fp16 SUBNORMALS (value * 2^-24, exact; v_dot2c keeps
//         subnormals — scripts/ubench/probe_dot2.hip); windows hold 2 x 4-bit / 4 x 2-bit / 8 x 1-bit fields;
//   bf16 (7-bit mantissa, no usable subnormal range): (bits | 0x4300) = 128 + value, and the 128 * sum(x) term
//         is removed per run of rows; windows hold 1 x 4-bit / 2 x 2-bit / 4 x 1-bit fields.
template <typename Tag, int NBITS>
struct Window {
    static constexpr bool SUBN = F16Traits<Tag>::DT == GEMLITE_DT_FP16;
    static constexpr int MANT = SUBN ? 10 : 7;
    static constexpr int HALF = 16 / NBITS;
    static constexpr int fit() {
        int wp = 1;
        while (wp * 2 <= HALF && (((1 << NBITS) - 1) << (NBITS * (wp * 2 - 1))) < (1 << MANT)) wp *= 2;
        return wp;
    }
    static constexpr int WP = fit();  // pairs per window
};

// XD ("x direct", 4-bit only): x is not staged in LDS; every lane loads the 16 bytes of x that belong to each of
// its packed rows straight from global memory (L1/L2 hits, requested together with the weights) and pairs /
// pre-scales them in registers (4 v_perm + 2 v_pk_mul + 4 v_dot2 per row).  This removes the x -> LDS -> barrier
// prologue from the critical path of short K slices (decode shapes such as 4096 x 4096).
// NW = waves per block (4; 8 for the short-K direct-x variant: with one chunk per wave the only way to overlap the
// unpack arithmetic of one part of K with the weight stream of another is a second wave on the same SIMD).
template <typename Tag, int NBITS, int MB, int R, int CQ, bool XD = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, ((MB * (16 / NBITS) >= 32 || R * MB >= 32 || NW > 4) ? 1 : 2)) void gemv_wn_kernel(const WnParams p) {
    using TR = F16Traits<Tag>;
    using WN = Window<Tag, NBITS>;
    constexpr bool SUBN = WN::SUBN;
    constexpr int E = 32 / NBITS;    // elements per packed word
    constexpr int HALF = E / 2;      // (k, k+HALF) pairs per word == LDS dwords per packed row
    constexpr int WP = WN::WP;
    // fp16, 4-bit (two fields per 16-bit window): the field read 4 bits up gets its OWN fp32 accumulator and the 2^-4 is applied once to the sum
    // (round 5; rounds 2-4 pre-scaled the matching x by 2^-4 in fp16: mantissa bits lost below |x| = 2^-10).
    // (A/B on one box, scripts/r5/run_j_ab_gemv_split.sh: 16384^2 M = 1 27.9 vs 28.3 us in the single-workload bench — no cost)
    // Round 6: 2- and 1-bit words too — four / eight accumulator sets, one per field position, combined per chunk with
    // 2^-(NBITS * position): one accuracy bound for every decode kernel (bf16 never needed it: its pre-scale by a power of two is exact over
    // the whole fp32-sized exponent range).
    constexpr bool SPLIT = SUBN && WP > 1;
    constexpr int NACC = SPLIT ? WP : 1;
    constexpr int G = 64 >> CQ;      // row sub-groups per wave
    constexpr int CHUNK = G * R;     // packed rows one wave consumes per step
    constexpr int TC = 4 << CQ;      // tile columns
    constexpr int RUN_SPANS = R * E / 32;  // 32-k spans of x covered by a lane's run of R rows
    constexpr int NT = NW * 64;           // threads per block
    static_assert(XD || R * E % 32 == 0, "a run of rows must cover whole 32-k spans of x");
    static_assert(!XD || (NBITS == 4 && MB == 1), "direct x loads: one 16-byte x chunk per packed row");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: the branches on a wave's chunk count are scalar branches)
    const int c = lane & ((1 << CQ) - 1), g = lane >> CQ;
    // The scalar zero point is read — and waited for — HERE, ahead of every request of the K loop (round 6).  Left next to its first use
    // inside the chunk arithmetic, it was what put the `s_waitcnt vmcnt(0)` at the head of the two-chunk loop in rounds 2-5: hipcc's
    // waitcnt pass carries "this load may be outstanding" around the back edge and answers it with a full drain in EVERY iteration.
    float scalar_zero = 0.f;
    if (p.zero_is_scalar) scalar_zero = (float)__builtin_amdgcn_readfirstlane(((const int32_t*)p.zeros)[0]);
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

    const int rows_slice = p.rows_per_slice;          // multiple of CHUNK
    const int row_s0 = slice * rows_slice;            // first packed row of the slice
    // chunks of the slice are dealt round-robin to the waves (wave w: chunks w, w + NW, ...), so any chunk count
    // works — K = 11008 or 8960 (43 / 35 chunks of 32 rows) as well as the powers of two
    const int row_w0 = wave * CHUNK;                  // first row of this wave's chunk 0, relative to the slice
    constexpr int CSTRIDE = NW * CHUNK;               // rows between consecutive chunks of one wave
    const int pairs = rows_slice * HALF;              // LDS dwords per x row
    const int nspans = rows_slice * E / 32;           // 32-k spans per x row

    uint32_t* xs = (uint32_t*)smem;                                   // [MB][pairs]  pair-permuted, pre-scaled x
    float* xsum_t = (float*)(smem + (size_t)MB * pairs * 4);           // [MB][nspans] sum of x over each span
    float* xsum_s = xsum_t + MB * nspans;                              // [MB][nspans] sum of x as stored (bf16 path)
    float* red = xsum_s + MB * nspans;                                 // [NW][MB][TC]
    unsigned* flag = (unsigned*)(red + NW * MB * TC);

    // ---- weight + metadata stream: everything a chunk needs is requested together, one chunk ahead --------
    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    // unused metadata is still "loaded" (from the weight buffer, always in bounds) so the loop stays branch-free
    const uint16_t* sp = need_s ? (const uint16_t*)p.scales : (const uint16_t*)p.w;
    const uint16_t* zp = need_z ? (const uint16_t*)p.zeros : (const uint16_t*)p.w;
    const uint32_t mstride = (need_s || need_z) ? (uint32_t)p.stride_meta_g : 0u;
    // Addresses are a kernel-uniform base plus a 32-bit BYTE offset per lane (global_load saddr + voffset): the
    // 64-bit row * stride products of the straightforward form were a third of the loop's VALU instructions.
    // (The planner keeps every tensor below 4 GiB.)
    const uint32_t sw4 = (uint32_t)p.stride_wk * 4u;
    uint32_t wo[R];  // byte offset of this lane's R rows in chunk 0
#pragma unroll
    for (int i = 0; i < R; ++i) wo[i] = (uint32_t)(row_s0 + row_w0 + g * R + i) * sw4 + (uint32_t)n0 * 4u;
    const uint32_t xo = (uint32_t)(row_s0 + row_w0 + g * R) * (uint32_t)(E * 2);  // XD: byte offset of row 0's x chunk
    const int nch_slice = rows_slice / CHUNK;
    const int nchunks = (nch_slice - wave + NW - 1) / NW;  // may be 0 for the last waves of a short slice
    struct Chunk { u32x4 w[R]; u32x2 s, z; u32x4 x[XD ? R : 1]; };
    const uint16_t* xg = (const uint16_t*)p.x;
    const char* wb = (const char*)p.w;
    constexpr int NLC = R + 2 + (XD ? R : 0);  // requests per chunk
    // (8-bit words with 8 rows per lane: 32 weight registers per chunk buffer — with the asm requests the allocator spilled, and a spilled
    //  destination of a request in flight is a wrong result; that one variant keeps the compiler-tracked loads and pipeline2_run)
    constexpr bool ASMQ = !(NBITS == 8 && R == 8);
    auto load_chunk = [&](Chunk& ck, int chunk) __attribute__((always_inline)) {
        const int row = row_s0 + row_w0 + chunk * CSTRIDE + g * R;
        const uint32_t mo = ((uint32_t)group_of(row * E, p.gs_shift) * mstride + (uint32_t)n0) * 2u;
        if constexpr (!ASMQ) {
            const uint32_t co = (uint32_t)(chunk * CSTRIDE) * sw4;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const u32x4* src = (const u32x4*)(wb + (wo[i] + co));
                ck.w[i] = GL_GEMV_NT ? __builtin_nontemporal_load(src) : *src;
            }
            ck.s = *(const u32x2*)((const char*)sp + mo);
            ck.z = *(const u32x2*)((const char*)zp + mo);
            return;
        }
        if constexpr (XD) {  // x first: it is the cheaper (cached) request and is needed together with w
            const uint32_t xc = xo + (uint32_t)(chunk * CSTRIDE) * (uint32_t)(E * 2);
#pragma unroll
            for (int i = 0; i < R; ++i) gvw::gld128<false>(ck.x[i], (const char*)xg, xc + (uint32_t)(i * E * 2));
        }
        const uint32_t co = (uint32_t)(chunk * CSTRIDE) * sw4;  // uniform
#pragma unroll
        for (int i = 0; i < R; ++i)  // streamed once by one CU: non-temporal (guide row nt-weights; strip reads 23.6 -> 20.7 us at 16384^2 with R = 8)
            gvw::gld128<(GL_GEMV_NT != 0)>(ck.w[i], wb, wo[i] + co);
        gvw::gld64(ck.s, (const char*)sp, mo);
        gvw::gld64(ck.z, (const char*)zp, mo);
    };
    auto tie_chunk = [&](Chunk& ck) __attribute__((always_inline)) {
        if constexpr (XD) {
#pragma unroll
            for (int i = 0; i < R; ++i) gvw::tie(ck.x[i]);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) gvw::tie(ck.w[i]);
        gvw::tie(ck.s);
        gvw::tie(ck.z);
    };

    // ---- x[k-slice] -> LDS.  task = (row m, 32-k span): 64 bytes in, 16 pair-permuted / pre-scaled dwords and
    //      the span sums out.  The first task's loads are issued AHEAD of the weight stream. ---------------------
    const int64_t k0 = (int64_t)row_s0 * E;
    const int ntasks = MB * nspans;
    // the first task of every thread: four asm requests, unconditional (a task past the end re-reads the last one and is dropped by put_x), so
    // that every wave's queue holds the same number of requests ahead of its first chunks
    auto fetch_x_first = [&](u32x4 (&v)[4], int task) __attribute__((always_inline)) {
        const int t2 = task < ntasks ? task : ntasks - 1;
        const int m = t2 / nspans, spn = t2 - m * nspans;
        const int m2 = m < p.M ? m : p.M - 1;
        const uint32_t off = (uint32_t)(((int64_t)m2 * p.stride_xm + k0 + (int64_t)spn * 32) * 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) gvw::gld128<false>(v[q], (const char*)xg, off + (uint32_t)(16 * q));
    };
    auto fetch_x = [&](u32x4 (&v)[4], int task) {
        if (task < ntasks) {
            const int m = task / nspans, spn = task - m * nspans;
            if (m < p.M) {
                const uint16_t* src = xg + (int64_t)m * p.stride_xm + k0 + (int64_t)spn * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *(const u32x4*)(src + 8 * q);
            }
        }
    };
    auto put_x = [&](const u32x4 (&vin)[4], int task) {
        if (task >= ntasks) return;
        constexpr int WPS = 32 / E;  // packed words per span (E <= 32)
        const int m = task / nspans, spn = task - m * nspans;
        uint32_t outv[16];
        float sum_t = 0.f, sum_s = 0.f;
        if (m < p.M) {
            uint16_t v[32];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int z = 0; z < 4; ++z) {
                    v[8 * q + 2 * z] = (uint16_t)(vin[q][z] & 0xFFFFu);
                    v[8 * q + 2 * z + 1] = (uint16_t)(vin[q][z] >> 16);
                }
#pragma unroll
            for (int wdi = 0; wdi < WPS; ++wdi)
#pragma unroll
                for (int d = 0; d < HALF; ++d) {
                    const float lo = TR::to_float(v[wdi * E + d]), hi = TR::to_float(v[wdi * E + d + HALF]);
                    sum_t += lo + hi;
                    if constexpr (WP > 1 && !SPLIT) {  // pre-scale by 2^-(NBITS * position-in-window): exact
                        const float sc = __builtin_bit_cast(float, (uint32_t)(127 - NBITS * (d % WP)) << 23);
                        const uint16_t slo = TR::from_float(lo * sc), shi = TR::from_float(hi * sc);
                        sum_s += TR::to_float(slo) + TR::to_float(shi);
                        outv[wdi * HALF + d] = (uint32_t)slo | ((uint32_t)shi << 16);
                    } else {
                        outv[wdi * HALF + d] = (uint32_t)v[wdi * E + d] | ((uint32_t)v[wdi * E + d + HALF] << 16);
                    }
                }
            if constexpr (WP == 1 || SPLIT) sum_s = sum_t;
        } else {
#pragma unroll
            for (int d = 0; d < 16; ++d) outv[d] = 0u;
        }
        uint32_t* dst = xs + m * pairs + spn * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(u32x4*)(dst + 4 * q) = (u32x4){outv[4 * q], outv[4 * q + 1], outv[4 * q + 2], outv[4 * q + 3]};
        xsum_t[task] = sum_t;
        xsum_s[task] = sum_s;
    };

    // opt-in timeline (tuning[3] & 4, needs a workspace): lane 0 of every wave of block (0,0) stores s_memtime stamps
    // (round 3: scripts/isa_loops.py shows hipcc waiting with s_waitcnt vmcnt(0) at the head of the steady-state loop below, and
    //  these conditional stamp STORES are one of its reasons (stores count in vmcnt on gfx9).  Compiling them out, moving the
    //  scalar-zero load above the priming and dropping the in-loop stamp turned the wait into vmcnt(6) + a vmcnt(0) a few
    //  instructions later AND changed the schedule for the worse: 16384^2 23.7 -> 28.0 us, 8192^2 10.4 -> 11.3 us, 4096^2 4.80 ->
    //  4.95 us (profiles/r03/probe_gemv3_v8*.log).  The version with the stamps is the one that ships.)
    const bool probe = (p.flags & 4) && p.counters && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0;
    unsigned long long* stamps = (unsigned long long*)(p.counters + MAX_SPLITK_COUNTERS) + wave * 16;
    auto stamp = [&](int i) {
        if (probe) stamps[i] = __builtin_readcyclecounter();
    };
    stamp(0);
    Chunk A, B;
    if constexpr (XD) {
        if (nchunks > 0) load_chunk(A, 0);
        if (nchunks > 1) load_chunk(B, 1);
    } else {
        u32x4 xv[4];
        if constexpr (ASMQ) {
            fetch_x_first(xv, tid);  // x FIRST in every wave's queue (requests return in order), then the first two chunks
            if (nchunks > 0) load_chunk(A, 0);
            if (nchunks > 1) load_chunk(B, 1);
            // x has landed when only the chunk requests are outstanding
            if (nchunks > 1) gvw::wait_vm<2 * NLC>();
            else if (nchunks > 0) gvw::wait_vm<NLC>();
            else gvw::wait_vm<0>();
#pragma unroll
            for (int q = 0; q < 4; ++q) gvw::tie(xv[q]);
        } else {
            fetch_x(xv, tid);
            if (nchunks > 0) pipeline2_prime(nchunks, A, B, load_chunk);
        }
        put_x(xv, tid);
        for (int task = tid + NT; task < ntasks; task += NT) {  // large K * MB only
            if constexpr (ASMQ) {
                // asm requests here too (their wait drains the first chunks as well — this path is rare): ONE compiler-tracked load ahead of the
                // K loop is enough to bring the drain back — hipcc answers "that load may still be writing v[a:b]" with s_waitcnt vmcnt(0) in
                // front of the first instruction of the loop that re-uses those registers, in every iteration (round 6: that, not the chunk
                // requests themselves, was the vmcnt(0) rounds 3-5 could not get rid of)
                fetch_x_first(xv, task);
                gvw::wait_vm<0>();
#pragma unroll
                for (int q = 0; q < 4; ++q) gvw::tie(xv[q]);
            } else {
                fetch_x(xv, task);
            }
            put_x(xv, task);
        }
        __syncthreads();
    }

    stamp(1);  // loads issued (and, without XD, x staged + barrier passed)
    float tot[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) tot[m][j] = 0.f;

    // (a, b) of group_affine() without per-column branches: a = s (s == 1 when unused),
    // b = bz * z * (mode 3 ? s : 1) with bz = -1 (modes 1, 3), +1 (mode 4), 0 otherwise
    const float bz = (p.w_mode == 1 || p.w_mode == 3) ? -1.f : (p.w_mode == 4 ? 1.f : 0.f);
    const bool b_times_s = p.w_mode == 3;
    constexpr float QSCALE = SUBN ? 16777216.0f : 1.0f;  // 2^24: undo the subnormal interpretation
    uint32_t wmask[WP];  // field i of a window, both 16-bit halves
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
        const int row_rel = row_w0 + chunk * CSTRIDE + g * R;  // first of this lane's R rows (slice-relative)
        float acc[NACC][MB][4];
        float xd_sum = 0.f;  // XD: sum of the run's true x
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[a][m][j] = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            constexpr int XW = HALF >= 4 ? 4 : HALF;  // x dwords fetched per LDS read
#pragma unroll
            for (int q = 0; q < HALF / XW; ++q) {
                uint32_t xr[MB][XW];
                if constexpr (XD) {
                    // natural (x0x1)(x2x3)(x4x5)(x6x7) -> pairs (x0,x4)(x1,x5)(x2,x6)(x3,x7); run sums on the TRUE x,
                    // then the window pre-scale (fp16: odd pairs * 2^-4, exact)
                    const uint32_t d0 = ck.x[i][0], d1 = ck.x[i][1], d2 = ck.x[i][2], d3 = ck.x[i][3];
                    xr[0][0] = __builtin_amdgcn_perm(d2, d0, 0x05040100u);
                    xr[0][1] = __builtin_amdgcn_perm(d2, d0, 0x07060302u);
                    xr[0][2] = __builtin_amdgcn_perm(d3, d1, 0x05040100u);
                    xr[0][3] = __builtin_amdgcn_perm(d3, d1, 0x07060302u);
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) xd_sum = TR::dot2(xr[0][dd], TR::ONES2, xd_sum);
                    if constexpr (WP > 1 && !SPLIT) {
#pragma unroll
                        for (int dd = 0; dd < 4; ++dd)
                            if (dd % WP) {
                                const h2_t v = __builtin_bit_cast(h2_t, xr[0][dd]) *
                                               (h2_t){(_Float16)(1.0f / (1 << (NBITS * (dd % WP)))), (_Float16)(1.0f / (1 << (NBITS * (dd % WP))))};
                                xr[0][dd] = __builtin_bit_cast(uint32_t, v);
                            }
                    }
                } else
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
                    const int d = q * XW + dd;            // pair index inside the word
                    const int win = d / WP, wi = d % WP;  // window, position in window
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        uint32_t h = (ck.w[i][j] >> (NBITS * WP * win)) & wmask[wi];
                        if constexpr (!SUBN) h |= TR::MAGIC2;
#pragma unroll
                        for (int m = 0; m < MB; ++m) acc[SPLIT ? wi : 0][m][j] = TR::dot2(h, xr[m][dd], acc[SPLIT ? wi : 0][m][j]);
                    }
                }
            }
        }
        f32x4 s = unpack4(ck.s), z = unpack4(ck.z);
        if (!need_s) s = (f32x4){1.f, 1.f, 1.f, 1.f};
        if (!need_z) z = (f32x4){scalar_zero, scalar_zero, scalar_zero, scalar_zero};
        const int span0 = (row_rel * E) / 32;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float xt = 0.f, xst = 0.f;  // sum of x over the run; sum of x as stored (magic-offset removal)
            if constexpr (XD) {
                xt = xd_sum;
                xst = xd_sum;  // bf16 has one 4-bit field per window: stored x == x
            } else {
#pragma unroll
                for (int e2 = 0; e2 < RUN_SPANS; ++e2) {
                    xt += xsum_t[m * nspans + span0 + e2];
                    if constexpr (!SUBN) xst += xsum_s[m * nspans + span0 + e2];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = s[j] * QSCALE;
                const float b = bz * z[j] * (b_times_s ? s[j] : 1.f);
                float v = acc[0][m][j];
                if constexpr (SPLIT) {
#pragma unroll
                    for (int wi = 1; wi < NACC; ++wi) v = __builtin_fmaf(acc[wi][m][j], 1.0f / (float)(1 << (NBITS * wi)), v);
                }
                if constexpr (!SUBN) v -= TR::OFF * xst;
                tot[m][j] += a * v + b * xt;
            }
        }
    };

    // (no stamp inside the ring: a store is a vector-memory request and would shift the counted waits)
    if constexpr (ASMQ) gvw::ring2_run<NLC>(nchunks, A, B, load_chunk, tie_chunk, compute);
    else if (nchunks > 0) pipeline2_run(nchunks, A, B, load_chunk, compute);
    stamp(3);  // all chunks consumed

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
    stamp(4);  // wave-level shuffles done
    if (g == 0) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const f32x4 v = {tot[m][0], tot[m][1], tot[m][2], tot[m][3]};
            *(f32x4*)(red + (wave * MB + m) * TC + c * 4) = v;
        }
    }
    __syncthreads();
    stamp(5);  // block barrier passed

    // ---- across the waves; output o = m*TC + col, o < MB*TC, strided over the threads ---------------------
    constexpr int NOUT = MB * TC;
    constexpr int OPT = (NOUT + NT - 1) / NT;  // outputs per thread
    float part[OPT];
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * NT;
        float v = 0.f;
        if (o < NOUT) {
            const int m = o / TC, col = o % TC;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[(w * MB + m) * TC + col];
        }
        part[it] = v;
    }
    if (p.splitk == 1) {
#pragma unroll
        for (int it = 0; it < OPT; ++it) {
            const int o = tid + it * NT;
            if (o < NOUT && (o / TC) < p.M) store_out_t<Tag>(p.epi, part[it], o / TC, (int64_t)tile * TC + (o % TC));
        }
        stamp(6);
        return;
    }
    float* slab = p.slabs + ((int64_t)tile * p.splitk) * NOUT;
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * NT;
        if (o < NOUT) slab_store(slab + (int64_t)slice * NOUT + o, part[it]);
    }
    if (!splitk_arrive_is_last(p.counters + tile, p.splitk, flag)) return;
#pragma unroll
    for (int it = 0; it < OPT; ++it) {
        const int o = tid + it * NT;
        if (o < NOUT) {
            float v = 0.f;
            for (int s = 0; s < p.splitk; ++s) v += slab_load(slab + (int64_t)s * NOUT + o);
            if ((o / TC) < p.M) store_out_t<Tag>(p.epi, v, o / TC, (int64_t)tile * TC + (o % TC));
        }
    }
    if (tid == 0) splitk_reset(p.counters + tile);
}

// ---------------------------------------------------------------------------------------------
// Decode kernel for the short-K 4-bit shapes (round 3): 16-column tiles, K not split, NW waves of one or two chunks each.
// Same arithmetic as gemv_wn_kernel<Tag, 4, 1, R, 2, true, NW> (bit-identical partial sums per lane), rebuilt around what the
// round-2 timeline of that kernel showed (profiles/r01_gemv_timeline.json: 1.7k of a block's 6.3k cycles go into ISSUING its
// requests — the CU's address path moves 64 B per clock and the direct x loads asked for as many bytes as the weights — and
// 0.9k into the 16 ds_bpermute + 16-thread tail after the last wave's data has arrived):
//   * x: lane (g, c) loads only DWORD c of its packed row's 16 bytes (4 B per lane instead of 16: the four lanes of a quad own
//     the four column groups of the SAME row) and the quad exchanges them with DPP quad_perm broadcasts — no LDS, no barrier;
//   * weights: non-temporal loads (every byte is read once by one CU; guide row nt-weights: -18 % issue-to-landed);
//   * reduction: the 4 row sub-groups of a 16-lane DPP row are summed with two row_ror adds per value (VALU, no LDS pipe), the
//     4 DPP rows of every wave go to LDS as one 16-byte store per quad leader, and after the only barrier ONE wave sums the
//     4 NW partials of the 16 outputs (16 conflict-free reads per lane + two cross-row shuffles) and stores.
// Measured (profiles/r03/probe_gemv3_*.log, timeline_decode_*.log): 5.14-5.19 -> 4.80 us per launch in the bench's replayed
// graph.  What the per-block timelines then showed: every variant tried afterwards — 8 waves x 4 rows with x / scales / zeros
// fetched once per wave through LDS (half the memory instructions and half the VALU work per CU), the matrix-core kernel of
// gemv_mfma.hip — leaves a block alive for the same 2.5 us: blocks start within 0.3 us, the first bytes come back ~0.9 us
// later and the CU's 40 KB are complete ~1.3 us after that (6.8 TB/s chip-wide while it streams); the launch boundary in the
// graph is another 1.5 us.  At this size the kernel is bound by HBM LATENCY + the launch boundary, not by issue or arithmetic.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_movf(float v) {
    return __builtin_bit_cast(float, dpp_mov<CTRL>(__builtin_bit_cast(uint32_t, v)));
}

template <typename Tag, int R, int NW, bool NT>
__global__ __launch_bounds__(NW * 64, 1) void gemv_w4_decode_kernel(const WnParams p) {
    using TR = F16Traits<Tag>;
    using WN = Window<Tag, 4>;
    constexpr bool SUBN = WN::SUBN;
    constexpr int WP = WN::WP;
    constexpr int G = 16, CHUNK = G * R, TC = 16, CSTRIDE = NW * CHUNK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = (float*)smem;  // [NW * 4 DPP rows][16 columns]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 3, g = lane >> 2;
    int tile = blockIdx.x;
    {  // adjacent half-line tiles on one XCD (speed only; any mapping is correct)
        const int nb = gridDim.x;
        if ((nb & 15) == 0) {
            const int xcd = tile & 7, idx = tile >> 3;
            tile = (((idx >> 1) << 3) + xcd) * 2 + (idx & 1);
        }
    }
    const int n0 = tile * TC + c * 4;
    const int nch_total = p.rows_per_slice / CHUNK;          // K is not split: rows_per_slice = all packed rows
    const int nchunks = (nch_total - wave + NW - 1) / NW;    // chunks wave, wave + NW, ...

    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    const uint16_t* sp = need_s ? (const uint16_t*)p.scales : (const uint16_t*)p.w;
    const uint16_t* zp = need_z ? (const uint16_t*)p.zeros : (const uint16_t*)p.w;
    const uint32_t mstride = (need_s || need_z) ? (uint32_t)p.stride_meta_g : 0u;
    const uint32_t sw4 = (uint32_t)p.stride_wk * 4u;
    const char* wb = (const char*)p.w;
    const char* xb = (const char*)p.x;
    const uint32_t row0 = (uint32_t)(wave * CHUNK + g * R);  // this lane's first row of chunk 0
    const uint32_t wo0 = row0 * sw4 + (uint32_t)n0 * 4u;
    const uint32_t xo0 = row0 * 16u + (uint32_t)c * 4u;      // 8 k x 2 bytes per packed row; dword c of it

    struct Chunk { u32x4 w[R]; u32x2 s, z; uint32_t xq[R]; };
    auto load_chunk = [&](Chunk& ck, int chunk) {
        const uint32_t row = row0 + (uint32_t)(chunk * CSTRIDE);
        const uint32_t mo = ((uint32_t)group_of((int)row * 8, p.gs_shift, p.gs_magic) * mstride + (uint32_t)n0) * 2u;
        const uint32_t xo = xo0 + (uint32_t)(chunk * CSTRIDE) * 16u;
#pragma unroll
        for (int i = 0; i < R; ++i) ck.xq[i] = *(const uint32_t*)(xb + (xo + (uint32_t)i * 16u));
        ck.s = *(const u32x2*)((const char*)sp + mo);
        ck.z = *(const u32x2*)((const char*)zp + mo);
        const uint32_t wo = wo0 + (uint32_t)(chunk * CSTRIDE) * sw4;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const u32x4* src = (const u32x4*)(wb + (wo + (uint32_t)i * sw4));
            ck.w[i] = NT ? __builtin_nontemporal_load(src) : *src;
        }
    };

    // opt-in timeline (tuning[3] & 4, needs a workspace): wave 0 of EVERY block stores the constant-rate global clock
    // (s_memrealtime, 100 MHz) at four points, slot [block][i] — when blocks start, how long they wait for their data, when the
    // last one leaves
    const bool probe = (p.flags & 4) && p.counters && wave == 0 && lane == 0 && blockIdx.x < 1024;
    unsigned long long* stamps = (unsigned long long*)(p.counters + MAX_SPLITK_COUNTERS) + blockIdx.x * 4;
    auto stamp = [&](int i) {
        if (probe && i < 4) stamps[i] = __builtin_amdgcn_s_memrealtime();
    };
    stamp(0);
    Chunk cur, nxt;  // the wave's first two chunks are requested up front (all there is at K <= 8192)
    if (nchunks > 0) load_chunk(cur, 0);
    if (nchunks > 1) load_chunk(nxt, 1);
    stamp(1);

    float tot[4] = {0.f, 0.f, 0.f, 0.f};
    const float scalar_zero = p.zero_is_scalar ? (float)((const int32_t*)p.zeros)[0] : 0.f;
    const float bz = (p.w_mode == 1 || p.w_mode == 3) ? -1.f : (p.w_mode == 4 ? 1.f : 0.f);
    const bool b_times_s = p.w_mode == 3;
    constexpr float QSCALE = SUBN ? 16777216.0f : 1.0f;
    uint32_t wmask[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) wmask[i] = (15u * 0x00010001u) << (4 * i);

    auto compute = [&](const Chunk& ck) {
        // fp16: the second field of a 16-bit window is read 4 bits up; its products go to their OWN accumulator and the 2^-4 is applied once to
        // the fp32 sum (round 5, ADVICE r4: rounds 2-4 scaled the matching x pairs by 2^-4 in fp16 here, which lost mantissa bits below |x| = 2^-10;
        // the same scheme as gemv_w4_decode3_kernel)
        float acc[WP][4];
#pragma unroll
        for (int wi = 0; wi < WP; ++wi)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[wi][j] = 0.f;
        float xsum = 0.f;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            // the quad's four dwords = the row's 8 x values (x0x1)(x2x3)(x4x5)(x6x7) -> pairs (x0,x4)(x1,x5)(x2,x6)(x3,x7)
            const uint32_t d0 = dpp_mov<0x00>(ck.xq[i]), d1 = dpp_mov<0x55>(ck.xq[i]);
            const uint32_t d2 = dpp_mov<0xAA>(ck.xq[i]), d3 = dpp_mov<0xFF>(ck.xq[i]);
            uint32_t xr[4];
            xr[0] = __builtin_amdgcn_perm(d2, d0, 0x05040100u);
            xr[1] = __builtin_amdgcn_perm(d2, d0, 0x07060302u);
            xr[2] = __builtin_amdgcn_perm(d3, d1, 0x05040100u);
            xr[3] = __builtin_amdgcn_perm(d3, d1, 0x07060302u);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) xsum = TR::dot2(xr[dd], TR::ONES2, xsum);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const int win = dd / WP, wi = dd % WP;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t h = (ck.w[i][j] >> (4 * WP * win)) & wmask[wi];
                    if constexpr (!SUBN) h |= TR::MAGIC2;
                    acc[wi][j] = TR::dot2(h, xr[dd], acc[wi][j]);
                }
            }
        }
        const uint32_t s0 = ck.s[0], s1 = ck.s[1], z0 = ck.z[0], z1 = ck.z[1];
        float s[4] = {TR::to_float((uint16_t)(s0 & 0xFFFFu)), TR::to_float((uint16_t)(s0 >> 16)), TR::to_float((uint16_t)(s1 & 0xFFFFu)), TR::to_float((uint16_t)(s1 >> 16))};
        float z[4] = {TR::to_float((uint16_t)(z0 & 0xFFFFu)), TR::to_float((uint16_t)(z0 >> 16)), TR::to_float((uint16_t)(z1 & 0xFFFFu)), TR::to_float((uint16_t)(z1 >> 16))};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!need_s) s[j] = 1.f;
            if (!need_z) z[j] = scalar_zero;
            const float a = s[j] * QSCALE;
            const float b = bz * z[j] * (b_times_s ? s[j] : 1.f);
            float v = acc[0][j];
            if constexpr (WP > 1) v = __builtin_fmaf(acc[WP - 1][j], 1.0f / 16.0f, v);
            if constexpr (!SUBN) v -= TR::OFF * xsum;
            tot[j] += a * v + b * xsum;
        }
    };
    // (a wave has 1 .. 3 chunks on the shapes this kernel takes); ONE copy of the arithmetic in the binary —
    // the five inlined copies of the generic two-buffer pipeline needed > 128 registers at 16 waves per block
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
        compute(cur);
        if (ch + 1 < nchunks) {
            cur = nxt;
            if (ch + 2 < nchunks) load_chunk(nxt, ch + 2);
        }
    }
    stamp(2);

    // ---- the 4 row sub-groups of every 16-lane DPP row (lane bits 2, 3): two rotations, every lane ends with the row's sum --
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = tot[j];
        v += dpp_movf<0x124>(v);  // row_ror:4
        v += dpp_movf<0x128>(v);  // row_ror:8
        tot[j] = v;
    }
    if ((lane & 12) == 0) *(f32x4*)(red + ((wave * 4 + (lane >> 4)) * TC + c * 4)) = (f32x4){tot[0], tot[1], tot[2], tot[3]};
    __syncthreads();
    if (wave == 0) {
        constexpr int PER = NW;  // partial rows per lane quarter: 4 NW rows over 4 quarters
        const int o = lane & 15, part = lane >> 4;
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < PER; ++r) v += red[(part * PER + r) * TC + o];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) store_out_t<Tag>(p.epi, v, 0, (int64_t)tile * TC + o);
    }
    stamp(3);
}

// ---------------------------------------------------------------------------------------------
// host-side planning
// ---------------------------------------------------------------------------------------------
template <typename Tag, int NBITS, int MB, int R, int CQ, bool XD = false, int NW = 4>
static const void* inst() {
    // only the (bits, rows, tile) combinations the planner can pick are instantiated
    constexpr bool used = (CQ == 2 && R == 4 && (NBITS == 4 || NBITS == 2)) ||
                          (CQ == 3 && ((R == 4 && (NBITS == 4 || NBITS == 2)) || (R == 2 && NBITS == 2))) ||
                          (CQ == 4 && ((R == 8 && (NBITS == 8 || NBITS == 4)) || (R == 4 && (NBITS == 4 || NBITS == 2)) ||
                                       (R == 2 && NBITS == 2) || (R == 1 && NBITS == 1)));
    if constexpr (used && (R * (32 / NBITS)) % 32 == 0 && NBITS <= F16Traits<Tag>::MAX_QBITS) {
        return (const void*)gemv_wn_kernel<Tag, NBITS, MB, R, CQ, XD, NW>;
    } else {
        return nullptr;
    }
}
template <typename Tag, int NBITS, int MB>
static const void* pick_shape(int cq, int r, bool xd, int nw) {
    if (xd) {  // direct-x variants exist for 4-bit, 4 rows per lane
        if constexpr (NBITS == 4 && MB == 1) {
            if (nw == 8 && r == 2) return cq == 2 ? (const void*)gemv_wn_kernel<Tag, 4, 1, 2, 2, true, 8> : nullptr;
            if (nw == 16 && r == 2) return cq == 2 ? (const void*)gemv_wn_kernel<Tag, 4, 1, 2, 2, true, 16> : nullptr;
            if (r != 4) return nullptr;
            if (nw == 8) return cq == 2 ? (const void*)gemv_wn_kernel<Tag, 4, 1, 4, 2, true, 8> : (cq == 3 ? (const void*)gemv_wn_kernel<Tag, 4, 1, 4, 3, true, 8> : nullptr);
            return cq == 2 ? inst<Tag, 4, 1, 4, 2, true>() : (cq == 3 ? inst<Tag, 4, 1, 4, 3, true>() : inst<Tag, 4, 1, 4, 4, true>());
        } else {
            return nullptr;
        }
    }
    if (nw == 8) {  // two waves per SIMD: one wave's unpack arithmetic overlaps the other's memory wait (long K)
        if (r != 4) return nullptr;
        return cq == 4 ? inst<Tag, NBITS, MB, 4, 4, false, 8>() : (cq == 3 ? inst<Tag, NBITS, MB, 4, 3, false, 8>() : inst<Tag, NBITS, MB, 4, 2, false, 8>());
    }
    if (cq == 2) return r == 4 ? inst<Tag, NBITS, MB, 4, 2>() : nullptr;
    if (cq == 3) return r == 4 ? inst<Tag, NBITS, MB, 4, 3>() : (r == 2 ? inst<Tag, NBITS, MB, 2, 3>() : nullptr);
    switch (r) {
        case 8: return inst<Tag, NBITS, MB, 8, 4>();
        case 4: return inst<Tag, NBITS, MB, 4, 4>();
        case 2: return inst<Tag, NBITS, MB, 2, 4>();
        case 1: return inst<Tag, NBITS, MB, 1, 4>();
        default: return nullptr;
    }
}
template <typename Tag, int NBITS>
static const void* pick_mb(int mb, int cq, int r, bool xd, int nw) {
    return mb == 1 ? pick_shape<Tag, NBITS, 1>(cq, r, xd, nw) : nullptr;  // M >= 2 runs on the MFMA streaming kernel
}
template <typename Tag>
static const void* pick_bits(int nbits, int mb, int cq, int r, bool xd, int nw = 4) {
    switch (nbits) {
        case 1: return pick_mb<Tag, 1>(mb, cq, r, xd, nw);
        case 2: return pick_mb<Tag, 2>(mb, cq, r, xd, nw);
        case 4: return pick_mb<Tag, 4>(mb, cq, r, xd, nw);
        case 8: return pick_mb<Tag, 8>(mb, cq, r, xd, nw);
        default: return nullptr;
    }
}

// measured (profiles/r02/probe_gemv.log): A16W2 16384^2 20.0 -> 17.8 us with 8 waves (16 k per packed word: twice the unpack
// arithmetic per byte); 4-bit shapes do not gain (16384^2: 24.6 vs 25.8 us, 8192^2: equal) and keep 4 waves
constexpr bool GEMV_AUTO_8W = true;
// Decide variant / grid / split-K / LDS for the GEMV kernel.  Returns false if this shape is not covered.
bool plan_gemv_wn(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    const int nbits = a.W_nbits;
    if (nbits != 1 && nbits != 2 && nbits != 4 && nbits != 8) return false;
    const int e = 32 / nbits;
    if (a.M > 1 || a.K % e != 0) return false;  // M >= 2 goes to the MFMA streaming kernel (16-row tiles)
    if (a.output_dtype != a.input_dtype) return false;  // typed epilogue
    const bool loop_s = a.W_group_mode >= 2, post_s = a.channel_scale_mode == 1 || a.channel_scale_mode == 3;
    const bool uses_s = loop_s || post_s;
    const bool has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    // metadata read in the K loop: the kernel's 16-bit type.  Channel scales of the epilogue alone (round 4: BitNet's fp32 scale,
    // A16W158_INT — it ran on the 32-row MFMA tile at M = 1, 12.6 us at 4096^2): any float type (store_out_t)
    if (loop_s && a.meta_dtype != a.input_dtype) return false;
    if (post_s && !loop_s && a.meta_dtype != a.input_dtype && a.meta_dtype != GEMLITE_DT_FP32 && a.meta_dtype != GEMLITE_DT_FP16 && a.meta_dtype != GEMLITE_DT_BF16) return false;
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != a.input_dtype) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    if ((a.stride_xm * 2) % 16 != 0 || ((uintptr_t)a.x % 16) != 0 || a.K % 32 != 0) return false;  // 16-byte x loads
    // 16-byte weight loads and 8-byte metadata loads: sliced / offset views that break the alignment go to the
    // coverage kernel instead of faulting
    if (((uintptr_t)a.w_q % 16) != 0 || (a.stride_wk % 4) != 0) return false;
    if ((loop_s && ((uintptr_t)a.scales % 8) != 0) || (has_z && !a.zero_is_scalar && ((uintptr_t)a.zeros % 8) != 0)) return false;
    (void)uses_s;
    if (p.stride_meta_g % 4 != 0) return false;
    const int rows = (int)(a.K / e);
    const int64_t gs = p.group_size;
    if (gs % e != 0) return false;
    // 32-bit byte offsets in the kernel
    if ((int64_t)rows * a.stride_wk * 4 >= (1ll << 32) || (int64_t)(a.K / gs) * p.stride_meta_g * 2 + a.N * 2 >= (1ll << 32)) return false;
    const int rpg = (int)(gs / e);  // packed rows per group
    const int mb = a.M <= 1 ? 1 : (a.M <= 2 ? 2 : 4);

    auto try_plan = [&](int cq, int want_splitk) -> bool {
        const int tc = 4 << cq, G = 64 >> cq;
        if (a.N % tc != 0) return false;
        // rows per lane and step: must divide a group, cover whole 32-k spans of x, and fit the register budget
        int r = 0;
        const int cand[4] = {8, 4, 2, 1};
        for (int ci = 0; ci < 4 && !r; ++ci) {
            const int rr = cand[ci];
            if (cq == 2 && rr != 4) continue;
            if (cq == 3 && rr != 4 && rr != 2) continue;
            // 8 rows per lane: 8-bit words.  4-bit only on request (tuning[2] = 48): a pure strip read gains from the deeper
            // queue (scripts/ubench/graph_floor.hip: 23.6 -> 20.8 us at 16384^2), the real kernel LOSES (27.4 vs 23.8 us with
            // 4 rows + non-temporal loads: 185 registers, one wave per SIMD — profiles/r03/probe_gemv3_v3)
            if (rr == 8 && cq == 4 && nbits != 8 && !(nbits == 4 && a.tuning[2] == 48)) continue;
            if (nbits == 1 && rr != 1) continue;                   // 16 x-dwords per packed row
            if ((rr * e) % 32 != 0 || rpg % rr != 0 || rows % (G * rr) != 0) continue;
            r = rr;
        }
        // K = 11008 / 8960 (Llama-2-7B down_proj, Qwen2.5-1.5B): 16-column tiles need 64-row chunks with 4 rows per
        // lane; the 16-wave direct-x variant takes 2 rows per lane (32-row chunks: 43 / 35 of them)
        bool force_xd16 = false;
        if (!r && cq == 2 && nbits == 4 && mb == 1 && rows % (G * 2) == 0 && rpg % 2 == 0 && (a.tuning[3] & 3) != 1) {
            r = 2;
            force_xd16 = true;
        }
        if (!r) return false;
        const int chunk_rows = G * r;  // packed rows one wave consumes per step; chunks are dealt round-robin to the waves
        if (rows % chunk_rows != 0) return false;
        const int tiles = (int)(a.N / tc);
        const int units = rows / chunk_rows;  // chunks = max number of K slices
        auto ok = [&](int sk) {  // slices must divide the steps and keep the LDS copy of x within 64 KiB
            if (sk < 1 || units % sk != 0) return false;
            return (int64_t)mb * (rows / sk) * e * 2 <= 65536;
        };
        int splitk = 0;
        if (want_splitk > 0) {
            if (!ok(want_splitk)) return false;
            splitk = want_splitk;
        } else {
            const int target = 256;  // >= one block per CU; beyond that the in-launch combine costs more than it buys
            for (int sk = 1; sk <= units; sk *= 2) {
                if (!ok(sk)) continue;
                splitk = sk;
                if (tiles * sk >= target) break;
            }
            if (!splitk) return false;
        }
        if (tiles > MAX_SPLITK_COUNTERS && splitk > 1) return false;
        // tuning[3]: 0 auto | 1 stage x in LDS | 2 load x directly.  Direct x pays when a wave has <= 2 steps.
        const int steps = (units / splitk + 3) / 4;  // chunks per wave (4 waves)
        const int xmode = a.tuning[3] & 3;
        bool xd = force_xd16 || (nbits == 4 && mb == 1 && r == 4 && (xmode == 2 || (xmode == 0 && steps <= 2)));
        // tuning[2]: 0 auto | 4 | 8 waves per block.  8 waves: one chunk per wave, two waves per SIMD (short K only)
        int nw = force_xd16 ? 16 : 4;
        if (force_xd16 && splitk != 1) return false;
        if (xd && !force_xd16 && cq <= 3 && splitk == 1) {
            const int want = a.tuning[2];
            // measured at 4096 x 4096 (profiles/r01_run21): 4 waves 5.28 us, 8 waves 4.69-4.80 us, 16 waves 4.48 us
            const int opt_nw[3] = {16, 8, 8}, opt_r[3] = {2, 4, 2}, opt_key[3] = {16, 8, 82};
            for (int oi = 0; oi < 3; ++oi) {
                if (!(want == opt_key[oi] || (want == 0 && oi < 2))) continue;
                const int cr = G * opt_r[oi];  // chunk rows; at most two chunks per wave
                if (rows % cr != 0 || (rows / cr + opt_nw[oi] - 1) / opt_nw[oi] > 2 || rpg % opt_r[oi] != 0) continue;
                nw = opt_nw[oi];
                r = opt_r[oi];
                break;
            }
        }
        // LDS-x path with 8 waves (tuning[2] == 8; auto when every wave still gets >= 4 chunks: the long-K shapes, where
        // one wave per SIMD spends ~2/3 of its time in unpack arithmetic that nothing overlaps with the weight stream)
        if (!xd && r == 4 && (cq >= 3 || a.tuning[2] == 8) && (nbits == 4 || nbits == 2) &&
            (a.tuning[2] == 8 || (a.tuning[2] == 0 && GEMV_AUTO_8W && (units / splitk) >= (nbits == 2 ? 32 : (cq == 4 ? 32 : 128)))))  // (round 6, counted asm loads: 4-bit 16384^2 24.5 vs 25.0 us with 8 waves;
            // late round 6, 64-column tiles from 32 chunks per slice: 14336 x 4096 8.05 -> 7.72 us, 5120 x 13824 12.55 -> 11.58, 13824 x 5120 9.17 -> 8.75, 8192 x 28672 24.0 -> 22.5; 32-column tiles
            // LOSE with 8 waves (8192^2 9.13 -> 9.79, 8192 x 4096 6.19 -> 6.72) and keep the 128-chunk rule: profiles/r06/probe_m1_8waves_w4.log)
            nw = 8;
        const void* fn = a.input_dtype == GEMLITE_DT_FP16 ? pick_bits<half_tag>(nbits, mb, cq, r, xd, nw)
                                                           : pick_bits<bf16_tag>(nbits, mb, cq, r, xd, nw);
        if (!fn && !xd && nw == 8) {
            nw = 4;
            fn = a.input_dtype == GEMLITE_DT_FP16 ? pick_bits<half_tag>(nbits, mb, cq, r, false) : pick_bits<bf16_tag>(nbits, mb, cq, r, false);
        }
        if (!fn && xd) {
            xd = false;
            nw = 4;
            fn = a.input_dtype == GEMLITE_DT_FP16 ? pick_bits<half_tag>(nbits, mb, cq, r, false)
                                                  : pick_bits<bf16_tag>(nbits, mb, cq, r, false);
        }
        if (!fn) return false;
        p.splitk = splitk;
        p.rows_per_slice = rows / splitk;
        lp.fn = fn;
        // round 3: the short-K decode kernel (quad-shared x dwords, non-temporal weights, DPP + one-wave reduction) takes the
        // direct-x 16-column shapes; tuning[3] & 16 keeps the round-2 kernel (A/B runs), & 32 = default-policy weight loads
        if (xd && cq == 2 && splitk == 1 && mb == 1 && nbits == 4 && !(a.tuning[3] & 16) && ((nw == 16 && r == 2) || (nw == 8 && r == 4))) {
            const bool f16 = a.input_dtype == GEMLITE_DT_FP16, nt = !(a.tuning[3] & 32);
            typedef void (*kfn)(const WnParams);
            kfn k;
            if (nw == 16) k = f16 ? (nt ? gemv_w4_decode_kernel<half_tag, 2, 16, true> : gemv_w4_decode_kernel<half_tag, 2, 16, false>)
                                  : (nt ? gemv_w4_decode_kernel<bf16_tag, 2, 16, true> : gemv_w4_decode_kernel<bf16_tag, 2, 16, false>);
            else k = f16 ? (nt ? gemv_w4_decode_kernel<half_tag, 4, 8, true> : gemv_w4_decode_kernel<half_tag, 4, 8, false>)
                         : (nt ? gemv_w4_decode_kernel<bf16_tag, 4, 8, true> : gemv_w4_decode_kernel<bf16_tag, 4, 8, false>);
            lp.fn = (const void*)k;
            lp.name = nw == 16 ? "gemv_w4_decode_kernel<tile16,16w>" : "gemv_w4_decode_kernel<tile16,8w>";
            lp.grid = dim3(tiles, 1, 1);
            lp.block = dim3(64 * nw, 1, 1);
            lp.lds_bytes = (size_t)nw * 4 * 16 * 4;
            lp.slab_bytes = 0;
            lp.ws_bytes = 0;
            // round 4: the same arithmetic behind preloaded scalar arguments, weights requested first (gemv_decode.hip); the common
            // case only — no channel scales in the epilogue, contiguous output row, group size a power of two.  tuning[3] & 4096
            // keeps the round-3 kernel (A/B runs).
            if (nw == 16 && !(a.tuning[3] & 4096) && a.channel_scale_mode == 0 && a.stride_on == 1 && p.gs_shift >= 0 && p.gs_shift < 32 &&
                (int64_t)rows * 16 < (1ll << 31)) {
                // (the scalar-zero bit and pointer only when the mode HAS a zero point — a C-ABI caller may pass zero_is_scalar
                //  with zeros = NULL for W_group_mode 0 / 2, and the kernel reads zp[0] whenever the bit is set)
                const bool has_z2 = p.w_mode == 1 || p.w_mode >= 3, z_scalar2 = has_z2 && p.zero_is_scalar;
                const bool need_s2 = p.w_mode >= 2, need_z2 = has_z2 && !p.zero_is_scalar;
                lp.arg_kind = 1;
                lp.fn = gemv_w4_decode3_fn(f16 ? 0 : 1, nt);
                lp.name = "gemv_w4_decode3_kernel<tile16,16w>";
                lp.lds_bytes = 0;  // static LDS
                lp.d3.w = (const char*)p.w;
                lp.d3.x = (const char*)p.x;
                lp.d3.s = need_s2 ? (const char*)p.scales : (const char*)p.w;
                lp.d3.z = (need_z2 || z_scalar2) ? (const char*)p.zeros : (const char*)p.w;
                lp.d3.out = (uint16_t*)p.epi.out;
                lp.d3.sw4 = (uint32_t)p.stride_wk * 4u;
                lp.d3.mstride2 = (need_s2 || need_z2) ? (uint32_t)p.stride_meta_g * 2u : 0u;
                lp.d3.nch_total = rows / 32;
                lp.d3.modes = (uint32_t)p.w_mode | (z_scalar2 ? 16u : 0u) | (((tiles & 15) == 0) ? 32u : 0u) | ((a.tuning[3] & 4) ? 64u : 0u) |
                              ((uint32_t)p.gs_shift << 8);
                lp.d3.counters = nullptr;
            }
            return true;
        }
        // group sizes that are not a power of two (gs_shift < 0, gs_magic): only gemv_w4_decode_kernel above finds the group by multiply-high —
        // the same two instructions in gemv_wn_kernel's chunk loader pushed its counted-asm-load variants into spills (scripts/isa_guard.py)
        if (p.gs_shift < 0) return false;
        lp.name = (xd && nw == 16) ? "gemv_wn_kernel<tile16,xdirect,16w>"
                  : (xd && nw == 8) ? (cq == 2 ? "gemv_wn_kernel<tile16,xdirect,8w>" : "gemv_wn_kernel<tile32,xdirect,8w>")
                  : xd ? (cq == 2 ? "gemv_wn_kernel<tile16,xdirect>" : (cq == 3 ? "gemv_wn_kernel<tile32,xdirect>" : "gemv_wn_kernel<tile64,xdirect>"))
                     : (nw == 8 ? (cq == 2 ? "gemv_wn_kernel<tile16,8w>" : (cq == 3 ? "gemv_wn_kernel<tile32,8w>" : "gemv_wn_kernel<tile64,8w>"))
                                : (cq == 2 ? "gemv_wn_kernel<tile16>" : (cq == 3 ? "gemv_wn_kernel<tile32>" : "gemv_wn_kernel<tile64>")));
        lp.grid = dim3(tiles, splitk, 1);
        lp.block = dim3(64 * nw, 1, 1);
        lp.lds_bytes = (size_t)mb * p.rows_per_slice * (e / 2) * 4 + (size_t)2 * mb * (p.rows_per_slice * e / 32) * 4 +
                       (size_t)nw * mb * tc * 4 + 16;
        lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * mb * tc * 4 : 0;
        lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
        return true;
    };

    // tuning[0]: 0 auto | 2 / 3 / 4 force 16- / 32- / 64-column tiles;  tuning[1]: 0 auto | n force split-K n
    const int force = a.tuning[0], sk = a.tuning[1];
    if (force >= 2 && force <= 4) return try_plan(force, sk);
    // auto (measured, profiles/r01_*): the widest tile that still gives >= 256 blocks WITHOUT splitting K wins —
    // 256-byte row segments stream best, 64-byte ones worst, but any in-launch split-K combine costs ~3 us.
    // Round 3 (4-bit, 22 LLM layer shapes x every tile / slice combination, profiles/r03/probe_m1_llm_shapes_*.log): 160 blocks are
    // enough — 5120^2 32-column tiles 7.6 vs 8.9 us for 16-column ones, 5120 x 13824 13.6 vs 18.3, N = 11008 .. 14336 over K = 4096
    // 64-column tiles 8.5 .. 8.9 vs 9.0 .. 9.3 — and narrow matrices (N < 2048) want 16-column tiles UNSPLIT on the decode kernel
    // (1024 x 4096: 4.5 vs 7.4 us, 1536 x 8960: 7.5 vs 9.4) rather than 32-column tiles with K slices.
    // 16-column tiles keep the 256-block rule (2560 x 9728 / 3072 x 8192: 160 .. 192 unsplit blocks of 64-byte segments 10.0 / 8.5 us
    // against 8.9 / 7.8 for 32-column tiles with K slices).  A long K over a narrow N (K >= 12288, fewer than 160 tiles of 64 columns)
    // takes 64-column tiles with the fewest K slices that give 160 blocks: 8192 x 28672 23.1 vs 26.2 us, 4096 x 14336 10.6 vs 12.2,
    // 5120 x 13824 12.9 vs 13.6.
    // Round 6 (profiles/r06/probe_m1_shapes_w2.log): 2-bit words follow the same rule — the widest tile with >= 160 blocks (128 over a short K:
    // 8960 x 1536 64-column tiles 4.41 / 4.75 us for 2- / 4-bit against 5.05 / 6.54 for 32-column ones) — 5120^2 5.65 (32-column) vs 6.54 us,
    // 13824 x 5120 7.26 vs 10.3, 28672 x 8192 16.2 vs 21.9; and narrow layers up to N = 3072 over a long K take 16-column tiles unsplit
    // (2048 x 8192: 5.44 / 6.78 vs 6.5 / 7.2; 3072 x 8192: 5.56 / 6.95 vs 6.5 / 7.8).
    const int64_t want_blocks = (nbits == 4 || nbits == 2) ? (a.K <= 2048 ? 128 : 160) : 256;
    const bool narrow16 = a.N < 2048 || (a.N <= 3072 && a.K >= 8192 && a.K % 2048 == 0);   // (2560 x 9728 keeps 32-column tiles x 2 slices: 7.88 vs 8.96)
    if ((nbits == 4 || (nbits == 2 && a.K % 2048 == 0)) && sk == 0 && narrow16 && a.K <= 12288 && try_plan(2, 1)) return true;
    if (nbits == 4 && sk == 0 && a.K >= 12288 && a.N % 64 == 0 && a.N / 64 < 160 && a.N / 64 >= 40) {
        int lsk = 2;
        while ((a.N / 64) * lsk < 160) lsk *= 2;
        if (try_plan(4, lsk)) return true;
    }
    if (a.N % 64 == 0 && a.N / 64 >= want_blocks && try_plan(4, sk > 0 ? sk : 1)) return true;
    if (a.N % 32 == 0 && a.N / 32 >= want_blocks && try_plan(3, sk > 0 ? sk : 1)) return true;
    if (a.N % 16 == 0 && a.N / 16 >= 256 && try_plan(2, sk > 0 ? sk : 1)) return true;
    if (try_plan(3, sk)) return true;
    if (try_plan(4, sk)) return true;
    return try_plan(2, sk);
}

}  // namespace gl
