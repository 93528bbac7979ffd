// gemm_a8w8.hip — tiled MFMA matmul for 8-bit activations x 8-bit UNPACKED weights (A8W8 int8 / fp8 dynamic
// quantisation, BASELINE config 4) at M >= 32.  Replaces gemm_INT_kernel / gemm_splitK_INT_kernel
// (gemlite/triton_kernels/gemm_kernels.py:248-413, gemm_splitK_kernels.py:277-450) for the unpacked case:
// W_q is the [K, N] transposed VIEW of an [N, K] row-major tensor (strides (1, K), core.py:369-381), x_q is the
// per-token quantised [M, K] tensor (quant_utils.py:268-347), so BOTH operands are K-contiguous and a lane's MFMA
// fragment is 16 consecutive bytes of one row — no unpack, no dequant, the only VALU work is address arithmetic.
//   int8 : v_mfma_i32_32x32x32_i8, exact int32 accumulation (core.py:48), fp32 epilogue scale (mode 3:
//          acc * scales_x[m] * scales_w[n], gemm_kernels.py:396-404);
//   fp8  : v_mfma_f32_32x32x16_{fp8_fp8, bf8_bf8} on the two 8-byte halves of the same 16-byte fragments.
// Tile (64 MI) x (64 NI), K step 128 bytes, 4 waves as 2 x 2 (wave tile 32 MI x 32 NI: MI A fragments and NI B
// fragments feed MI * NI MFMAs per 32 k).  Both tiles go global -> registers -> XOR-swizzled LDS (the same conflict-free 128-byte-row
// layout as the W4 tiled kernel), double buffered, with the tiles of the next THREE steps in flight in registers
// (one step ahead left every step waiting a full HBM round trip: 39.5 us at 4096^2, M = 256); rows >= M are read
// as zeros through the buffer descriptor.  K is not split; the planner takes the largest of 128x128 / 64x128 / 64x64
// that gives every CU a block (4096 x 4096, M = 256: 64x64 -> 256 blocks).
#include "gl_common.h"
#include "gl_async.h"
#include "gl_coopquant.h"
#include "gl_w8cvt.h"

#include <type_traits>

namespace gl {

bool plan_w8_rows_lds(const gemlite_hip_forward_args& a, LaunchPlan& lp, int mt, bool two);  // gemm_w8_rows.hip
// row tiles per block of that kernel and whether the two-blocks-per-CU form applies: layers with more 16-column blocks than CUs, up to 32 rows
static inline int w8_lds_mt(int64_t M) { return M <= 16 ? 1 : (M <= 32 ? 2 : (M <= 48 ? 3 : 4)); }
static inline bool w8_lds_two(const gemlite_hip_forward_args& a) { return a.N / 16 > resident_block_limit() && a.M <= 32; }

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int a8_slot(int r, int s) { return r * 128 + ((s ^ ((r >> 1) & 7)) << 4); }

template <int DT>
struct A8Acc {  // accumulator type + one 32-k multiply-accumulate on 16-byte fragments
    typedef f32x16 T;
    static __device__ __forceinline__ T zero() {
        T v;
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0.f;
        return v;
    }
    static __device__ __forceinline__ T mma(u32x4 a, u32x4 b, T c) {
        const long a0 = (long)(((uint64_t)a[1] << 32) | a[0]), a1 = (long)(((uint64_t)a[3] << 32) | a[2]);
        const long b0 = (long)(((uint64_t)b[1] << 32) | b[0]), b1 = (long)(((uint64_t)b[3] << 32) | b[2]);
        if constexpr (DT == GEMLITE_DT_FP8E4) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a0, b0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1, b1, c, 0, 0, 0);
        } else {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(a0, b0, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(a1, b1, c, 0, 0, 0);
        }
        return c;
    }
    // 64 k in ONE instruction: the 32x32x16 fp8 forms above run at the bf16 rate on gfx950 (32 cycles for 16 k; PMC on FP8 x FP8
    // 16384^2, M = 256: SQ_VALU_MFMA_BUSY_CYCLES = 32 per instruction, matrix pipes 63 % busy at 110 us) — the fp8 peak
    // belongs to v_mfma_f32_32x32x64_f8f6f4 (64 cycles for 64 k).  Two consecutive 16-byte fragments of a lane are exactly
    // its 32-byte operand (k = 32 (b >> 4) + 16 (lane >> 5) + (b & 15)); unit block scales (e8m0 127).
    static __device__ __forceinline__ T mma64(u32x4 a0, u32x4 a1, u32x4 b0, u32x4 b1, T c) {
        typedef int v8i __attribute__((ext_vector_type(8)));
        const v8i av = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
        const v8i bv = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
        constexpr int FMT = DT == GEMLITE_DT_FP8E4 ? 0 : 1;  // e4m3 / e5m2
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, FMT, FMT, 0, 127, 0, 127);
    }
    static __device__ __forceinline__ float to_float(T c, int e) { return c[e]; }
};
template <>
struct A8Acc<GEMLITE_DT_INT8> {
    typedef i32x16 T;
    static __device__ __forceinline__ T zero() {
        T v;
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0;
        return v;
    }
    static __device__ __forceinline__ T mma(u32x4 a, u32x4 b, T c) {
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_float(T c, int e) { return (float)c[e]; }
};

template <int DT, int MI, int NI>
__global__ __launch_bounds__(256, 2) void gemm_a8w8_kernel(const GenericParams p) {
    using AC = A8Acc<DT>;
    constexpr int BM = 64 * MI, BN = 64 * NI, BK = 128;
    constexpr int A_BYTES = BM * BK, B_BYTES = BN * BK, STAGE = A_BYTES + B_BYTES;
    constexpr int SA = BM * 8 / 256, SB = BN * 8 / 256;  // 16-byte staging slots per thread and step
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, col = lane & 31, kb = lane >> 5;
    const int mtiles = (p.M + BM - 1) / BM;
    const int mt = blockIdx.x % mtiles, nt = blockIdx.x / mtiles;  // M tiles fastest: neighbours share the weight tile
    const int m0 = mt * BM, n0 = nt * BN;
    const int ksteps = p.K / BK;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, (short)0, (int)((int64_t)(p.M - 1) * p.stride_xm + p.K), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.w, (short)0, (int)((int64_t)(p.N - 1) * p.stride_wn + p.K), 0x00020000);
    uint32_t aoff[SA], boff[SB];
    int awr[SA], bwr[SB];
#pragma unroll
    for (int i = 0; i < SA; ++i) {
        const int u = tid + 256 * i, r = u >> 3, s = u & 7;
        aoff[i] = m0 + r < p.M ? (uint32_t)((int64_t)(m0 + r) * p.stride_xm + s * 16) : 0x80000000u;
        awr[i] = a8_slot(r, s);
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) {
        const int u = tid + 256 * i, r = u >> 3, s = u & 7;
        boff[i] = (uint32_t)((int64_t)(n0 + r) * p.stride_wn + s * 16);
        bwr[i] = A_BYTES + a8_slot(r, s);
    }
    // PF register sets: the tile of step st + PF is requested when step st starts and is written to LDS at the end
    // of step st + PF - 1.  An HBM round trip is ~1-2 us while a step is only 4 MI NI MFMAs per wave (0.05-0.2 us),
    // so the small tiles keep six steps in flight (measured at 4096^2, M = 256: 1 step 39.5 us, 3 steps 27.4 us).
    constexpr int PF = MI * NI == 4 ? 3 : 6;
    static_assert(6 % PF == 0, "the K loop is unrolled by 6");
    u32x4 ra[PF][SA], rb[PF][SB];
    auto fetch = [&](int set, int step) {
        const uint32_t so = step < ksteps ? (uint32_t)(step * BK) : 0x80000000u;  // past K: zeros, no traffic
#pragma unroll
        for (int i = 0; i < SA; ++i) ra[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[i] + so, 0, 0);
#pragma unroll
        for (int i = 0; i < SB; ++i) rb[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, boff[i] + so, 0, 0);
    };
    auto stash = [&](int set, int buf) {
#pragma unroll
        for (int i = 0; i < SA; ++i) *(u32x4*)(smem + buf * STAGE + awr[i]) = ra[set][i];
#pragma unroll
        for (int i = 0; i < SB; ++i) *(u32x4*)(smem + buf * STAGE + bwr[i]) = rb[set][i];
    };

    typename AC::T acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = AC::zero();

    int fa[MI][4], fb[NI][4];  // fragment byte offsets inside a stage, per k32 step
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi][ks] = a8_slot(wm * 32 * MI + mi * 32 + col, ks * 2 + kb);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni][ks] = A_BYTES + a8_slot(wn * 32 * NI + ni * 32 + col, ks * 2 + kb);
    }

#pragma unroll
    for (int j = 0; j < PF; ++j) fetch(j, j);
    stash(0, 0);
    __syncthreads();
    for (int st0 = 0; st0 < ksteps; st0 += 6) {  // 6 = lcm(register sets, LDS stages): every index below is static
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int st = st0 + j;
            if (st >= ksteps) break;
            const unsigned char* sb = smem + (j & 1) * STAGE;
            fetch(j % PF, st + PF);  // that set's previous content (step st) is already in LDS
            __builtin_amdgcn_sched_barrier(0);  // keep the requests ahead of the MFMAs (the scheduler sinks them)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                u32x4 af[MI], bf[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[mi] = *(const u32x4*)(sb + fa[mi][ks]);
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bf[ni] = *(const u32x4*)(sb + fb[ni][ks]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = AC::mma(af[mi], bf[ni], acc[mi][ni]);
            }
            __builtin_amdgcn_sched_barrier(0);
            stash((j + 1) % PF, (j + 1) & 1);  // step st + 1; that stage was last read in step st - 1
            __syncthreads();
        }
    }

    // epilogue: C fragment of a 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 32 * MI + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kb;
                const int n = n0 + wn * 32 * NI + ni * 32 + col;
                if (m < p.M) epilogue_store(p.epi, AC::to_float(acc[mi][ni], e), m, n);
            }
}


// ---------------------------------------------------------------------------------------------------------------
// Shared epilogue of the 8-wave kernels: K halves through LDS, transpose, split-K slabs + combine, scaled output.
// ---------------------------------------------------------------------------------------------------------------
template <int DT, int MI>
__device__ __forceinline__ void a8_epilogue(const GenericParams& p, typename A8Acc<DT>::T (&acc)[MI], unsigned char* smem, int tid,
                                            int lane, int cg, int kh, int col, int h, int bid, int slice, int m0, int nt) {
    using AC = A8Acc<DT>;
    typedef typename AC::T acc_t;
    constexpr bool INT = DT == GEMLITE_DT_INT8;
    constexpr int BM = 32 * MI, BN = 128;
    constexpr int C_ROWS = 128, C_PITCH = BN + 4;
    // ---- epilogue 1: add the two K halves through LDS (raw 32-bit accumulator words: int32 stays exact) -------------
    __syncthreads();
    {
        acc_t* xch = (acc_t*)smem;  // [cg][mi][lane] whole accumulators (64 bytes per lane)
        if (kh == 1) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xch[(cg * MI + mi) * 64 + lane] = acc[mi];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] += xch[(cg * MI + mi) * 64 + lane];
        }
    }
    // ---- epilogue 2: transpose through LDS (128 rows per pass): slabs / output move as 16-byte row segments ------------
    typedef typename std::conditional<INT, int, float>::type word_t;
    typedef word_t word4 __attribute__((ext_vector_type(4)));
    word_t* ct = (word_t*)smem;  // [PASS_ROWS][C_PITCH]
    constexpr int PASS_ROWS = BM < C_ROWS ? BM : C_ROWS;
    unsigned* flag = (unsigned*)(smem + PASS_ROWS * C_PITCH * 4);
    constexpr int NPASS = BM / PASS_ROWS, MIP = PASS_ROWS / 32;
    constexpr int UNITS = (PASS_ROWS * BN / 4 + 511) / 512;
    constexpr int NOUT = BM * BN;
    const int64_t ncol0 = (int64_t)nt * BN;
    float* slab = p.slabs + ((int64_t)bid * p.splitk) * NOUT;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, (short)0, p.splitk * NOUT * 4, 0x00020000);
    auto finish = [&](word4 v, int m, int c4) {
        f32x4 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = (float)v[j];
        store_out4_any(p.epi, f, m, ncol0 + c4);
    };
    word4 own[NPASS][UNITS];  // this block's partial tile, kept for the combine (its own slab is not read back)
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int mi = 0; mi < MIP; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    ct[r * C_PITCH + cg * 32 + col] = acc[ps * MIP + mi][e];
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = tid + 512 * i, r = u >> 5, c4 = (u & 31) * 4;
            const int m = m0 + ps * PASS_ROWS + r;
            own[ps][i] = (word4){0, 0, 0, 0};
            if (r < PASS_ROWS && m < p.M) {
                const word4 v = *(const word4*)(ct + r * C_PITCH + c4);
                own[ps][i] = v;
                if (p.splitk == 1) finish(v, m, c4);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                            (slice * NOUT + (ps * PASS_ROWS + r) * BN + c4) * 4, 0, 16);  // sc1
            }
        }
    }
    if (p.splitk == 1) return;
    __syncthreads();
    if (!splitk_arrive_is_last(p.counters + bid, p.splitk, flag)) return;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        word4 sum[UNITS];
#pragma unroll
        for (int i = 0; i < UNITS; ++i) sum[i] = (word4){0, 0, 0, 0};
        for (int sl = 0; sl < p.splitk; ++sl) {
            if (sl == slice) {  // own partial: from registers (fixed slice order keeps the sum deterministic)
#pragma unroll
                for (int i = 0; i < UNITS; ++i) sum[i] += own[ps][i];
                continue;
            }
            u32x4 t[UNITS];
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                const int u = tid + 512 * i, r = u >> 5, c4 = (u & 31) * 4;
                t[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (sl * NOUT + (ps * PASS_ROWS + r) * BN + c4) * 4, 0, 16);
            }
#pragma unroll
            for (int i = 0; i < UNITS; ++i) sum[i] += __builtin_bit_cast(word4, t[i]);
        }
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = tid + 512 * i, r = u >> 5, c4 = (u & 31) * 4;
            const int m = m0 + ps * PASS_ROWS + r;
            if (r < PASS_ROWS && m < p.M) finish(sum[i], m, c4);
        }
    }
    if (tid == 0) splitk_reset(p.counters + bid);
}

// ---------------------------------------------------------------------------------------------------------------
// 8-wave kernel (default).  Same skeleton as gemm_wn_mma.hip: tile (32 MI) x 128, wave (cg, kh) owns all rows x 32
// columns x one half of every 256-byte K step; x goes global -> LDS by LDS-DMA (XOR-swizzled through the source
// address), the wave's B fragments — 16 consecutive bytes of ONE weight row = exactly a lane's operand of
// v_mfma_i32_32x32x32_i8 — go HBM -> registers with one 16-byte load per lane and 32-k slice, two steps ahead, and feed
// MI MFMAs each; nothing is dequantised.  Requests come from inline asm with counted waits (gl_async.h); K may be split
// over gridDim.y (raw int32 / fp32 accumulator words travel through the slabs, so int8 stays exact).
template <int DT, int MI, int RD, int NST>
__global__ __launch_bounds__(512, 2) void gemm_a8w8_mma_kernel(const GenericParams p) {
    using namespace async;
    using AC = A8Acc<DT>;
    typedef typename AC::T acc_t;
    constexpr bool INT = DT == GEMLITE_DT_INT8;
    constexpr int BM = 32 * MI, BN = 128, KSTEP = 256, KW = 128;
    constexpr int PITCH = KSTEP, STAGE = BM * PITCH;
    constexpr int PIECES = STAGE / 1024 / 8;  // LDS-DMA pieces per wave and stage (= MI)
    constexpr int NS = KW / 32;               // 32-k slices per wave and step
    constexpr int NQ = NS * MI, L = MI >= 4 ? 4 : (MI == 2 ? 4 : 2);
    constexpr int PD = RD - 2;  // weights are requested PD steps ahead: a step is only 4 MI MFMAs per wave (128 MI cycles)
                                // and an HBM round trip under load 2-3k cycles, so the small tiles need a deep ring
    // NST LDS stages of x (the tile of step s + NST - 1 is requested during step s): see gemm_wn_mma.hip
    static_assert(PIECES >= 1 && NQ >= 2 * L && RD % NST == 0 && RD >= 4 && NST >= 2, "tile too small for the slot schedule");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NST][STAGE], later the epilogue tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, kh = wave >> 2;
    const int col = lane & 31, h = lane >> 5;
    const int mtiles = (p.M + BM - 1) / BM;
    int bid = blockIdx.x, slice = blockIdx.y;
    {  // blocks of one XCD (block b runs on XCD b % 8) share a K slice: its x rows stay in that XCD's L2 (speed only)
        const int T = gridDim.x, S = gridDim.y;
        if ((p.flags & 8) && S > 1 && (8 % S) == 0 && ((T * S) & 7) == 0) {
            const int lin = blockIdx.x + T * blockIdx.y, xcd = lin & 7, idx = lin >> 3;
            slice = xcd % S;
            bid = idx * (8 / S) + xcd / S;
        }
    }
    const int mt = bid % mtiles, nt = bid / mtiles;
    const int m0 = mt * BM;
    const int n = nt * BN + cg * 32 + col;  // this lane's output column = its weight row

    const int units = p.K / KSTEP;
    const int s_begin = (int)((int64_t)slice * units / p.splitk), s_end = (int)((int64_t)(slice + 1) * units / p.splitk);
    const int nsteps = s_end - s_begin;
    const int k_s0 = s_begin * KSTEP;

    const srd_t rsX = make_srd(p.x, (uint32_t)((int64_t)(p.M - 1) * p.stride_xm + p.K));
    const __amdgpu_buffer_rsrc_t brW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.w, (short)0, (int)((int64_t)(p.N - 1) * p.stride_wn + p.K), 0x00020000);
    // B fragment of slice g: 16 bytes at weight row n, k = k_s0 + step * 256 + kh * 128 + g * 32 + h * 16
    const uint32_t wvoff = (uint32_t)((int64_t)n * p.stride_wn + k_s0 + kh * KW + h * 16);
    struct BStep { u32x4 w[NS]; };
    auto req_b = [&](BStep& b, int step, int g) {
        // a tracked buffer load: the compiler retires it with its own counted vmcnt (see gemm_wn_mma.hip)
        b.w[g] = __builtin_amdgcn_raw_buffer_load_b128(brW, wvoff, (uint32_t)__builtin_amdgcn_readfirstlane(step * KSTEP + g * 32), 0);
    };
    // x: piece j of wave w covers LDS bytes [(w * PIECES + j) * 1024, +1024) of a stage: row = byte / 256, physical
    // 16-byte slot (byte % 256) / 16 holds the logical slot phys ^ (row & 15)
    uint32_t xvoff[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const int byte = (wave * PIECES + j) * 1024 + lane * 16;
        const int r = byte / PITCH, phys = (byte % PITCH) / 16;
        const int logical = phys ^ (r & 15);
        xvoff[j] = m0 + r < p.M ? (uint32_t)((int64_t)(m0 + r) * p.stride_xm + k_s0 + logical * 16) : 0x80000000u;
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (uint32_t)(wave * PIECES) * 1024u);
    auto req_x = [&](int stage, int step, int j) {
        req_lds16(rsX, lds0 + (uint32_t)(stage * STAGE + j * 1024), xvoff[j], (uint32_t)__builtin_amdgcn_readfirstlane(step * KSTEP));
    };
    int fbase[NST][NS];  // A fragment of (slice g, row block mi): row mi*32 + col, byte kh*128 + g*32 + h*16
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            const int slot = (kh * KW + g * 32 + h * 16) >> 4;
            fbase[st][g] = st * STAGE + col * PITCH + (((slot ^ col) & 15) << 4);
        }
    // slot q -> (slice g, row block mi).  fp8: the slices are consumed in PAIRS by one 64-k MFMA, so the two slices of a pair
    // are adjacent slots: q = (pair * MI + mi) * 2 + (g & 1)
    auto slot_g = [&](int q) { return INT ? q / MI : 2 * (q / (2 * MI)) + (q & 1); };
    auto slot_mi = [&](int q) { return INT ? q % MI : (q >> 1) % MI; };
    auto read_frag = [&](int stage, int q) -> u32x4 {
        return *(const u32x4*)(smem + fbase[stage][slot_g(q)] + slot_mi(q) * 32 * PITCH);
    };

    acc_t acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[i] = AC::zero();

    BStep ring[RD];
    u32x4 af[L];

    // ---- prologue: x of step 0, weights of steps 0 .. PD-1 ------------------------------------------------------------
    // x and weights of step 0 first; only the x tile of step 0 is waited for (counted), see gemm_wn_mma.hip
#pragma unroll
    for (int j = 0; j < PIECES; ++j) req_x(0, 0, j);
#pragma unroll
    for (int g = 0; g < NS; ++g) req_b(ring[0], 0, g);
#pragma unroll
    for (int st = 1; st < NST - 1; ++st)
#pragma unroll
        for (int j = 0; j < PIECES; ++j) req_x(st, st < nsteps ? st : nsteps - 1, j);
#pragma unroll
    for (int r = 1; r < PD; ++r)
#pragma unroll
        for (int g = 0; g < NS; ++g) req_b(ring[r], r < nsteps ? r : nsteps - 1, g);
    {
        constexpr int AFTER = (NST - 2) * PIECES + PD * NS;
        wait_vm<(AFTER < 63 ? AFTER : 63)>();
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < L; ++q) af[q] = read_frag(0, q);
    __builtin_amdgcn_sched_barrier(0);

    constexpr int NLB = NS, NL = NLB + PIECES, NQI = NQ - L;
    constexpr int RPS = (NL + NQI - 1) / NQI;
    auto do_step = [&](auto Jc, int step) {
        constexpr int J = decltype(Jc)::value;
        constexpr int stage = J % NST, stage_next = (J + 1) % NST, stage_fill = (J + NST - 1) % NST;
        const BStep& bc = ring[J];
        BStep& bl = ring[(J + PD) % RD];
        const int lstep = step + PD < nsteps ? step + PD : nsteps - 1;  // run-ahead repeats the last step (never consumed)
        const int xstep = step + NST - 1 < nsteps ? step + NST - 1 : nsteps - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int g = slot_g(q), mi = slot_mi(q);
            if constexpr (INT) acc[mi] = AC::mma(af[q % L], bc.w[g], acc[mi]);
            else if (q & 1) acc[mi] = AC::mma64(af[(q - 1) % L], af[q % L], bc.w[g - 1], bc.w[g], acc[mi]);
            if (q == NQI) {
                wait_vm<(NST - 2) * PIECES + (NST - 1) * NLB>();  // the x DMA of step + 1 has landed
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // refill the ring L slots ahead (fp8: both fragments of a pair after the pair's MFMA; NQI is even, so the
            // reads of the current stage are still all issued before the barrier slot)
#pragma unroll
            for (int r = (INT ? q : (q & 1 ? q - 1 : NQ)); r <= q; ++r) {
                if (r + L < NQ) af[r % L] = read_frag(stage, r + L);
                else af[r % L] = read_frag(stage_next, r + L - NQ);
            }
#pragma unroll
            for (int it = q * RPS; it < (q + 1) * RPS && it < NL; ++it) {
                if (it < PIECES) req_x(stage_fill, xstep, it);
                else req_b(bl, lstep, it - PIECES);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int s0 = 0; s0 < nsteps; s0 += RD) {
        do_step(std::integral_constant<int, 0>{}, s0);
        if (s0 + 1 >= nsteps) break;
        do_step(std::integral_constant<int, 1>{}, s0 + 1);
        if (s0 + 2 >= nsteps) break;
        do_step(std::integral_constant<int, 2>{}, s0 + 2);
        if (s0 + 3 >= nsteps) break;
        do_step(std::integral_constant<int, 3>{}, s0 + 3);
        if constexpr (RD > 4) {
            if (s0 + 4 >= nsteps) break;
            do_step(std::integral_constant<int, 4 % RD>{}, s0 + 4);
            if (s0 + 5 >= nsteps) break;
            do_step(std::integral_constant<int, 5 % RD>{}, s0 + 5);
        }
        if constexpr (RD > 6) {
            if (s0 + 6 >= nsteps) break;
            do_step(std::integral_constant<int, 6 % RD>{}, s0 + 6);
            if (s0 + 7 >= nsteps) break;
            do_step(std::integral_constant<int, 7 % RD>{}, s0 + 7);
        }
    }
    wait_vm<0>();

    a8_epilogue<DT, MI>(p, acc, smem, tid, lane, cg, kh, col, h, bid, slice, m0, nt);
}

// ---------------------------------------------------------------------------------------------------------------
// 8-wave kernel with BOTH operands through LDS (128- / 256-row tiles).  A lane's B operand is 16 bytes of ONE weight row,
// so loading it straight from memory makes every wave instruction touch 32 rows x 32 bytes — FP8 x FP8 16384^2 at M = 256
// streamed its 268 MB of weights at 2.4 TB/s that way (110 us; matrix pipes 63 % busy even at the slow 16-k instruction).
// Here the weight tile [128 rows][128 bytes] of a step travels like the x tile: 1-KiB LDS-DMA pieces (8 rows x 128
// contiguous bytes per wave instruction, full cache lines), XOR-swizzled through the source address, and the lanes pick
// their fragments with ds_read_b128.  Steps are 128 bytes of K (wave half: 64 = two 32-k slices), NST stages, no register
// ring; every request is an asm DMA with one counted wait per step.  fp8 consumes the two slices of a step with ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 per row block.
// ---------------------------------------------------------------------------------------------------------------
template <int DT, int MI, int NST>
__global__ __launch_bounds__(512, 2) void gemm_a8w8_lds_kernel(const GenericParams p) {
    using namespace async;
    using AC = A8Acc<DT>;
    typedef typename AC::T acc_t;
    constexpr bool INT = DT == GEMLITE_DT_INT8;
    constexpr int BM = 32 * MI, BN = 128, KSTEP = 128, KW = 64, PITCH = KSTEP;
    constexpr int STAGE = (BM + BN) * PITCH;              // x rows, then weight rows
    constexpr int PX = BM * PITCH / 1024 / 8, PW = BN * PITCH / 1024 / 8, PT = PX + PW;  // DMA pieces per wave and stage
    constexpr int NS = KW / 32, NQ = NS * MI, L = 4;
    constexpr int U = (NST % 2 == 0) ? NST : 2 * NST;     // steps per unrolled group: stage AND buffer parity static
    static_assert(MI >= 4 && PX >= 1 && NS == 2 && NQ >= 2 * L && NST >= 2, "128- / 256-row tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NST][STAGE], later the epilogue tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, kh = wave >> 2;
    const int col = lane & 31, h = lane >> 5;
    const int mtiles = (p.M + BM - 1) / BM;
    const int bid = blockIdx.x, slice = blockIdx.y;
    const int mt = bid % mtiles, nt = bid / mtiles;
    const int m0 = mt * BM;
    const int units = p.K / KSTEP;
    const int s_begin = (int)((int64_t)slice * units / p.splitk), s_end = (int)((int64_t)(slice + 1) * units / p.splitk);
    const int nsteps = s_end - s_begin;
    const int k_s0 = s_begin * KSTEP;

    const srd_t rsX = make_srd(p.x, (uint32_t)((int64_t)(p.M - 1) * p.stride_xm + p.K));
    const srd_t rsW = make_srd(p.w, (uint32_t)((int64_t)(p.N - 1) * p.stride_wn + p.K));
    // piece j of wave w covers LDS bytes [(w * P + j) * 1024, +1024) of its region: row = byte / 128, physical 16-byte slot
    // (byte % 128) / 16 holds the logical slot phys ^ ((row >> 1) & 7) (128-byte rows: key row >> 1, see gemm_wn_mma.hip)
    uint32_t xvoff[PX], wvoff[PW];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int byte = (wave * PX + j) * 1024 + lane * 16;
        const int r = byte / PITCH, phys = (byte % PITCH) / 16;
        const int logical = phys ^ ((r >> 1) & 7);
        xvoff[j] = m0 + r < p.M ? (uint32_t)((int64_t)(m0 + r) * p.stride_xm + k_s0 + logical * 16) : 0x80000000u;
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int byte = (wave * PW + j) * 1024 + lane * 16;
        const int r = byte / PITCH, phys = (byte % PITCH) / 16;
        const int logical = phys ^ ((r >> 1) & 7);
        wvoff[j] = (uint32_t)((int64_t)(nt * BN + r) * p.stride_wn + k_s0 + logical * 16);
    }
    const uint32_t ldsx = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (uint32_t)(wave * PX) * 1024u);
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (uint32_t)(BM * PITCH) + (uint32_t)(wave * PW) * 1024u);
    auto request = [&](int stage, int step, int j) __attribute__((always_inline)) {  // piece j of the step's PT pieces
        const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(step * KSTEP);
        if (j < PX) req_lds16(rsX, ldsx + (uint32_t)(stage * STAGE + j * 1024), xvoff[j], so);
        else req_lds16(rsW, ldsw + (uint32_t)(stage * STAGE + (j - PX) * 1024), wvoff[j - PX], so);
    };
    // fragments: A of (slice g, row block mi): row mi * 32 + col; B of slice g: weight row cg * 32 + col; both at byte
    // kh * 64 + g * 32 + h * 16 of the row
    int fa[NST][NS], fb[NST][NS];
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            const int slot = (kh * KW + g * 32 + h * 16) >> 4;
            const int off = ((slot ^ (col >> 1)) & 7) << 4;
            fa[st][g] = st * STAGE + col * PITCH + off;
            fb[st][g] = st * STAGE + (BM + cg * 32 + col) * PITCH + off;
        }
    // slot q -> (slice g, row block mi): int8 walks slice-major; fp8 pairs the two slices of a row block
    auto slot_g = [&](int q) { return INT ? q / MI : (q & 1); };
    auto slot_mi = [&](int q) { return INT ? q % MI : (q >> 1); };
    auto read_a = [&](int stage, int q) -> u32x4 { return *(const u32x4*)(smem + fa[stage][slot_g(q)] + slot_mi(q) * 32 * PITCH); };
    auto read_b = [&](int stage, int g) -> u32x4 { return *(const u32x4*)(smem + fb[stage][g]); };

    acc_t acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[i] = AC::zero();
    u32x4 af[L], bw[2][NS];

    // ---- prologue: stages 0 .. NST-2 requested, stage 0 waited for ----------------------------------------------------
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
#pragma unroll
        for (int j = 0; j < PT; ++j) request(st, st < nsteps ? st : nsteps - 1, j);
    wait_vm<(NST - 2) * PT>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int g = 0; g < NS; ++g) bw[0][g] = read_b(0, g);
#pragma unroll
    for (int q = 0; q < L; ++q) af[q] = read_a(0, q);
    __builtin_amdgcn_sched_barrier(0);

    constexpr int NQI = NQ - L;  // the barrier slot: every read of the current stage has been issued before it
    auto do_step = [&](auto Jc, int step) __attribute__((always_inline)) {
        constexpr int J = decltype(Jc)::value;
        constexpr int stage = J % NST, stage_next = (J + 1) % NST, stage_fill = (J + NST - 1) % NST, P = J & 1;
        const int xstep = step + NST - 1 < nsteps ? step + NST - 1 : nsteps - 1;  // past the end: repeat the last step (never consumed)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int g = slot_g(q), mi = slot_mi(q);
            if constexpr (INT) acc[mi] = AC::mma(af[q % L], bw[P][g], acc[mi]);
            else if (q & 1) acc[mi] = AC::mma64(af[(q - 1) % L], af[q % L], bw[P][0], bw[P][1], acc[mi]);
            if (q == NQI) {
                wait_vm<(NST - 2) * PT>();  // the DMAs of step + 1 have landed (those of later stages stay in flight)
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of the current stage are complete
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (q == NQI + 1) {  // the next step's B fragments, into the other buffer
#pragma unroll
                for (int gg = 0; gg < NS; ++gg) bw[P ^ 1][gg] = read_b(stage_next, gg);
            }
#pragma unroll
            for (int r = (INT ? q : (q & 1 ? q - 1 : NQ)); r <= q; ++r) {
                if (r + L < NQ) af[r % L] = read_a(stage, r + L);
                else af[r % L] = read_a(stage_next, r + L - NQ);
            }
            if (q < PT) request(stage_fill, xstep, q);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    static_assert(PT <= NQ, "one request per slot");
    auto chain = [&](auto self, auto Jc, int s0) -> void {
        constexpr int J = decltype(Jc)::value;
        do_step(std::integral_constant<int, J>{}, s0 + J);
        if constexpr (J + 1 < U) {
            if (s0 + J + 1 < nsteps) self(self, std::integral_constant<int, J + 1>{}, s0);
        }
    };
    for (int s0 = 0; s0 < nsteps; s0 += U) chain(chain, std::integral_constant<int, 0>{}, s0);
    wait_vm<0>();
    a8_epilogue<DT, MI>(p, acc, smem, tid, lane, cg, kh, col, h, bid, slice, m0, nt);
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: 64 x 64 tiles, K NOT split ("gemm_a8w8_sq_kernel").  At config 4 (int8, 4096^2, M = 256) the matrix pipes need 1.7 us
// and every 128-column tiling above lands on 20-24 us (profiles/r03/probe_a8w8_m256_tiles.log: MFMA busy 8 %): 64 tiles of
// 128 x 128 need four K slices to fill 256 CUs, and a launch is then prologue + a 4-slice combine through memory around a
// 2-us loop.  Nothing is dequantised here, so a small tile costs no arithmetic — only operand traffic: a 64 x 64 tile
// reads (64 + 64) bytes per k for 8192 int8 operations, i.e. its loop is bound by the CU's 64 B/clk vector-memory path (512 KB per
// block at K = 4096 = ~3.9 us), 4 row tiles x 64 column tiles = 256 blocks fill the chip WITHOUT splitting K, and the
// epilogue is a plain store.  8 waves: wave (rb, cb, kh) owns the 32 x 32 block (rb, cb) of the tile and half of every
// 256-byte K step; both operands travel as 1-KiB LDS-DMA pieces (4 rows x 256 contiguous bytes, full cache lines) into
// NST stages of [64 x rows | 64 weight rows] x 256 B, 16-byte slots XOR-swizzled with (row & 15) through the source address; one
// counted wait + one barrier per step, the DMAs of the stage just freed go out right behind the barrier.  The four blocks that
// share a weight column tile sit on ONE XCD (block b runs on XCD b % 8; speed only) so that three of them read it from L2.
// ---------------------------------------------------------------------------------------------------------------
// ILV (round 6): the step's four DMA requests are issued BETWEEN its MFMAs, behind the LDS reads of the step, and int8 alternates two
// accumulators — a step is one barrier-to-barrier chain per wave (all eight waves in lockstep), and with the requests in front of the reads
// (rounds 4-5) that chain was ~4 x 100 cycles of request issue + the LDS latency + four dependent MFMAs; the number of stages does not matter
// beyond three (profiles/r06/probe_a8w8_sq_stages.log: 13.2 / 13.3 / 13.7 us with 3 / 4 / 5), the length of the chain does
template <int DT, int NST, bool ILV = false>
__global__ __launch_bounds__(512, (NST <= 2 ? 2 : 1)) void gemm_a8w8_sq_kernel(const GenericParams p) {
    using namespace async;
    using AC = A8Acc<DT>;
    typedef typename AC::T acc_t;
    constexpr bool INT = DT == GEMLITE_DT_INT8;
    constexpr int BM = 64, BN = 64, KSTEP = 256, PITCH = KSTEP, KW = KSTEP / 2, NS = KW / 32;
    constexpr int STAGE = (BM + BN) * PITCH;               // x rows, then weight rows
    constexpr int PX = BM * PITCH / 1024 / 8, PW = BN * PITCH / 1024 / 8, PT = PX + PW;  // DMA pieces per wave and stage
    static_assert(PX == 2 && PW == 2 && NS == 4 && NST >= 2, "64 x 64 x 256 B");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NST][STAGE], later the epilogue tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = (wave >> 1) & 1, cb = wave & 1, kh = wave >> 2;
    const int col = lane & 31, h = lane >> 5;
    const int mtiles = (p.M + BM - 1) / BM, ntiles = p.N / BN;
    int mt, nt;
    {
        const int lin = blockIdx.x;
        if ((ntiles & 7) == 0) {  // the row tiles of one weight column tile on one XCD, back to back in its dispatch order
            const int xcd = lin & 7, idx = lin >> 3;
            mt = idx % mtiles;
            nt = (idx / mtiles) * 8 + xcd;
        } else {
            mt = lin % mtiles;
            nt = lin / mtiles;
        }
    }
    const int m0 = mt * BM;
    const int nsteps = p.K / KSTEP;

    const srd_t rsX = make_srd(p.x, (uint32_t)((int64_t)(p.M - 1) * p.stride_xm + p.K));
    const srd_t rsW = make_srd(p.w, (uint32_t)((int64_t)(p.N - 1) * p.stride_wn + p.K));
    // piece j of wave w covers LDS bytes [(w * 2 + j) * 1024, +1024) of its region: row = byte / 256, physical 16-byte slot
    // (byte % 256) / 16 holds the logical slot phys ^ (row & 15)
    uint32_t xvoff[PX], wvoff[PW];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int byte = (wave * PX + j) * 1024 + lane * 16;
        const int r = byte / PITCH, phys = (byte % PITCH) / 16;
        const int logical = phys ^ (r & 15);
        xvoff[j] = m0 + r < p.M ? (uint32_t)((int64_t)(m0 + r) * p.stride_xm + logical * 16) : 0x80000000u;  // rows >= M: zeros
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int byte = (wave * PW + j) * 1024 + lane * 16;
        const int r = byte / PITCH, phys = (byte % PITCH) / 16;
        const int logical = phys ^ (r & 15);
        wvoff[j] = (uint32_t)((int64_t)(nt * BN + r) * p.stride_wn + logical * 16);
    }
    const uint32_t ldsx = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (uint32_t)(wave * PX) * 1024u);
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (uint32_t)(BM * PITCH) + (uint32_t)(wave * PW) * 1024u);
    // K rotation (round 6): the row tiles that share a weight column tile start their K loops a quarter apart (mtiles = 4) — each of the sibling
    // CUs then pulls a DIFFERENT part of the tile from HBM and finds the rest in its XCD's L2, instead of all of them waiting on the same cold
    // lines in lockstep.  The order of the K steps changes per row tile: exact for int8; for fp8 the fp32 sums differ in the last bits between tiles.
    // (measured, one box: 13.14 -> 12.05 us at int8 4096^2 M = 256, M = 128 13.1 -> 11.1, M = 512 21.5 -> 18.0, 2048 x 8192 M = 256 19.7 -> 15.0;
    //  staggering the column tiles that share a row tile of x as well: nothing.  It pays only while the weight tiles an XCD works on at a time stay in
    //  its L2 — the planner sets bit 30 of flags by that rule (k_rotation_pays() below: 4096 x 14336 M = 256, 7 MB per XCD, 34.4 -> 40.2 us).
    //  profiles/r06/probe_a8w8_sq_k_rotation.log, probe_k_rotation_*.log; tuning[3] & 4194304 = never, for A/B runs)
    KOrder kord;
    kord.init(mt, mtiles, nsteps, p.flags);
    auto kof = [&](int step) __attribute__((always_inline)) { return kord.at(step); };
    auto request = [&](int stage, int step) __attribute__((always_inline)) {
        const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(kof(step) * KSTEP);
#pragma unroll
        for (int j = 0; j < PX; ++j) req_lds16(rsX, ldsx + (uint32_t)(stage * STAGE + j * 1024), xvoff[j], so);
#pragma unroll
        for (int j = 0; j < PW; ++j) req_lds16(rsW, ldsw + (uint32_t)(stage * STAGE + j * 1024), wvoff[j], so);
    };
    // fragments of slice g (32 k): A row rb * 32 + col, B weight row cb * 32 + col, both at byte kh * 128 + g * 32 + h * 16
    int fa[NS], fb[NS];
#pragma unroll
    for (int g = 0; g < NS; ++g) {
        const int slot = (kh * KW + g * 32 + h * 16) >> 4;
        const int ra = rb * 32 + col, rw = cb * 32 + col;
        fa[g] = ra * PITCH + ((slot ^ (ra & 15)) << 4);
        fb[g] = (BM + rw) * PITCH + ((slot ^ (rw & 15)) << 4);
    }
    acc_t acc = AC::zero();
    acc_t acc1 = AC::zero();  // (ILV, int8: the odd k slices)
    auto request_piece = [&](int j, int stage, uint32_t so) __attribute__((always_inline)) {
        if (j < PX) req_lds16(rsX, ldsx + (uint32_t)(stage * STAGE + j * 1024), xvoff[j], so);
        else req_lds16(rsW, ldsw + (uint32_t)(stage * STAGE + (j - PX) * 1024), wvoff[j - PX], so);
    };

#pragma unroll
    for (int st = 0; st < NST - 1; ++st) request(st, st < nsteps ? st : nsteps - 1);
    auto do_step = [&](auto Jc, int step) __attribute__((always_inline)) {
        constexpr int stage = decltype(Jc)::value, stage_fill = (stage + NST - 1) % NST;
        wait_vm<(NST - 2) * PT>();  // this step's pieces have landed (the later stages' stay in flight)
        __builtin_amdgcn_s_barrier();  // ... everybody's have, and everybody is done reading the stage refilled next
        asm volatile("" ::: "memory");
        if constexpr (ILV) {
            u32x4 a[NS], b[NS];
#pragma unroll
            for (int g = 0; g < NS; ++g) {
                a[g] = *(const u32x4*)(smem + stage * STAGE + fa[g]);
                b[g] = *(const u32x4*)(smem + stage * STAGE + fb[g]);
            }
            const int fstep = step + NST - 1 < nsteps ? step + NST - 1 : nsteps - 1;  // past the end: repeat the last step (never consumed)
            const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(kof(fstep) * KSTEP);
            if constexpr (INT) {
#pragma unroll
                for (int g = 0; g < NS; ++g) {
                    __builtin_amdgcn_sched_barrier(0);
                    request_piece(g, stage_fill, so);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g & 1) acc1 = AC::mma(a[g], b[g], acc1);
                    else acc = AC::mma(a[g], b[g], acc);
                }
            } else {
#pragma unroll
                for (int g = 0; g < NS; g += 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    request_piece(g, stage_fill, so);
                    request_piece(g + 1, stage_fill, so);
                    __builtin_amdgcn_sched_barrier(0);
                    acc = AC::mma64(a[g], a[g + 1], b[g], b[g + 1], acc);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            return;
        }
        request(stage_fill, step + NST - 1 < nsteps ? step + NST - 1 : nsteps - 1);  // past the end: repeat the last step (never consumed)
        u32x4 a[NS], b[NS];
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            a[g] = *(const u32x4*)(smem + stage * STAGE + fa[g]);
            b[g] = *(const u32x4*)(smem + stage * STAGE + fb[g]);
        }
        if constexpr (INT) {
#pragma unroll
            for (int g = 0; g < NS; ++g) acc = AC::mma(a[g], b[g], acc);
        } else {
#pragma unroll
            for (int g = 0; g < NS; g += 2) acc = AC::mma64(a[g], a[g + 1], b[g], b[g + 1], acc);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of the stage are complete before it reaches the next barrier
    };
    auto chain = [&](auto self, auto Jc, int s0) -> void {
        constexpr int J = decltype(Jc)::value;
        do_step(std::integral_constant<int, J>{}, s0 + J);
        if constexpr (J + 1 < NST) {
            if (s0 + J + 1 < nsteps) self(self, std::integral_constant<int, J + 1>{}, s0);
        }
    };
    for (int s0 = 0; s0 < nsteps; s0 += NST) chain(chain, std::integral_constant<int, 0>{}, s0);
    wait_vm<0>();
    if constexpr (ILV && INT) acc += acc1;
    __syncthreads();

    // ---- epilogue: add the two K halves (raw accumulator words: int32 stays exact), transpose through LDS, 16-byte output rows
    typedef typename std::conditional<INT, int, float>::type word_t;
    typedef word_t word4 __attribute__((ext_vector_type(4)));
    constexpr int C_PITCH = BN + 4;
    // (round 6: BOTH K halves drop their partial tile into LDS in output order and every thread adds the two while it reads its 16-byte row segment
    //  — one barrier; rounds 4-5 handed the second half to the first in fragment order, added, transposed: three barriers with half of the block
    //  idle.  Same addition, same order: bit-identical.)
    word_t* ct = (word_t*)smem;  // [2 K halves][64][C_PITCH]
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int r = rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        ct[(kh * 64 + r) * C_PITCH + cb * 32 + col] = acc[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = tid + 512 * i, r = u >> 4, c4 = (u & 15) * 4;
        const int m = m0 + r;
        if (m < p.M) {
            const word4 v0 = *(const word4*)(ct + r * C_PITCH + c4), v1 = *(const word4*)(ct + (64 + r) * C_PITCH + c4);
            const word4 v = v0 + v1;
            store_out4_any(p.epi, (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, m, (int64_t)nt * BN + c4);
        }
    }
}

// Round 5: 128 x 128 tiles, K NOT split ("gemm_a8w8_sq_kernel<128x128>") — the same structure for layers whose 128 x 128 tiles number
// about one per CU (FP8 x FP8 16384^2 at M = 256, BASELINE configs[4]: 2 x 128 = 256 tiles; int8 8192^2 at M = 512; 4096^2 at M = 1024).
// A wave owns a 64 x 64 block = 2 x 2 MFMA blocks (two A and two B fragments feed FOUR MFMAs: half the LDS reads per MFMA of the
// 64 x 64 tiles), 8 waves = 2 x 2 wave tiles x 2 K halves of every step, NST stages of [128 x rows | 128 weight rows] x KSTEP bytes with
// NST - 1 of them in flight.  Measured (profiles/r05/probe_a8w8_sq128.log): two 64-KB stages of 256-byte steps (ONE in flight: every step
// pays the fill latency) 105 us on configs[4] at M = 256; four 32-KB stages of 128-byte steps 95.8 us (five: the same) against 98.5 us of
// gemm_a8w8_lds_kernel<256x128> with two K slices — all of them at the operand-fill rate the CUs reach when a quarter .. half of the
// stream comes from HBM (scripts/ubench/ldsfill.hip: 23 GB/s per CU for bytes read once, 105 GB/s for L2-resident ones, and the two
// ADD: 1 MB of weights + 2 .. 3 MB of re-read operands per CU = 62 .. 72 us before any arithmetic); 10 - 15 % ahead of the K-sliced
// tiles on the one-round shapes above.
template <int DT, int KSTEP, int NST>
__global__ __launch_bounds__(512, 2) void gemm_a8w8_sq128_kernel(const GenericParams p) {
    using namespace async;
    using AC = A8Acc<DT>;
    typedef typename AC::T acc_t;
    constexpr bool INT = DT == GEMLITE_DT_INT8;
    constexpr int BM = 128, BN = 128, PITCH = KSTEP, KW = KSTEP / 2, NS = KW / 32;
    constexpr int STAGE = (BM + BN) * PITCH;               // x rows, then weight rows: 64 KB (256-byte steps) / 32 KB (128-byte steps)
    constexpr int PX = BM * PITCH / 1024 / 8, PW = BN * PITCH / 1024 / 8, PT = PX + PW;  // DMA pieces per wave and stage
    static_assert((KSTEP == 256 || KSTEP == 128) && NST >= 2 && NST * STAGE <= 160 * 1024, "two 64-KB or up to five 32-KB stages");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NST][STAGE], later the epilogue tiles
    // 16-byte slots XOR-swizzled through the source address: key row & 15 (256-byte rows) / (row >> 1) & 7 (128-byte rows)
    auto key = [](int r) { return KSTEP == 256 ? (r & 15) : ((r >> 1) & 7); };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = (wave >> 1) & 1, cb = wave & 1, kh = wave >> 2;  // 64 x 64 wave tile (rb, cb), K half kh
    const int col = lane & 31, h = lane >> 5;
    const int mtiles = (p.M + BM - 1) / BM, ntiles = p.N / BN;
    int mt, nt;
    {
        const int lin = blockIdx.x;
        if ((ntiles & 7) == 0) {  // the row tiles of one weight column tile on one XCD, back to back in its dispatch order
            const int xcd = lin & 7, idx = lin >> 3;
            mt = idx % mtiles;
            nt = (idx / mtiles) * 8 + xcd;
        } else {
            mt = lin % mtiles;
            nt = lin / mtiles;
        }
    }
    const int m0 = mt * BM;
    const int nsteps = p.K / KSTEP;

    const srd_t rsX = make_srd(p.x, (uint32_t)((int64_t)(p.M - 1) * p.stride_xm + p.K));
    const srd_t rsW = make_srd(p.w, (uint32_t)((int64_t)(p.N - 1) * p.stride_wn + p.K));
    // piece j of wave w covers LDS bytes [(w * P + j) * 1024, +1024) of its region: row = byte / PITCH, physical 16-byte slot
    // (byte % PITCH) / 16 holds the logical slot phys ^ key(row)
    uint32_t xvoff[PX], wvoff[PW];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int byte = (wave * PX + j) * 1024 + lane * 16;
        const int r = byte / PITCH, phys = (byte % PITCH) / 16;
        const int logical = phys ^ key(r);
        xvoff[j] = m0 + r < p.M ? (uint32_t)((int64_t)(m0 + r) * p.stride_xm + logical * 16) : 0x80000000u;  // rows >= M: zeros
    }
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int byte = (wave * PW + j) * 1024 + lane * 16;
        const int r = byte / PITCH, phys = (byte % PITCH) / 16;
        const int logical = phys ^ key(r);
        wvoff[j] = (uint32_t)((int64_t)(nt * BN + r) * p.stride_wn + logical * 16);
    }
    const uint32_t ldsx = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (uint32_t)(wave * PX) * 1024u);
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (uint32_t)(BM * PITCH) + (uint32_t)(wave * PW) * 1024u);
    // (K rotation between the row tiles of a column tile, which pays on the 64 x 64 tiles, LOSES here: FP8 16384^2 M = 256 94.5 -> 113 us, int8 8192^2
    //  M = 512 46.6 -> 51.2, 4096^2 M = 1024 27.5 -> 28.0 — the sibling runs half of a 2-MB weight tile ahead, far more than an XCD's L2 keeps, and the
    //  lockstep sharing of today is lost: profiles/r06/probe_k_rotation_sq128_slower.log)
    KOrder kord;
    kord.init(mt, mtiles, nsteps, p.flags);
    auto request = [&](int stage, int step) __attribute__((always_inline)) {
        const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane(kord.at(step) * KSTEP);
#pragma unroll
        for (int j = 0; j < PX; ++j) req_lds16(rsX, ldsx + (uint32_t)(stage * STAGE + j * 1024), xvoff[j], so);
#pragma unroll
        for (int j = 0; j < PW; ++j) req_lds16(rsW, ldsw + (uint32_t)(stage * STAGE + j * 1024), wvoff[j], so);
    };
    // fragments of slice g (32 k) of MFMA block mi / ni: A row rb * 64 + mi * 32 + col, B weight row cb * 64 + ni * 32 + col, both at byte
    // kh * KW + g * 32 + h * 16 (the swizzle key does not depend on mi / ni: one base per slice + an immediate)
    int fa[NS], fb[NS];
#pragma unroll
    for (int g = 0; g < NS; ++g) {
        const int slot = (kh * KW + g * 32 + h * 16) >> 4;
        const int ra = rb * 64 + col, rw = cb * 64 + col;
        fa[g] = ra * PITCH + ((slot ^ key(ra)) << 4);
        fb[g] = (BM + rw) * PITCH + ((slot ^ key(rw)) << 4);
    }
    acc_t acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = AC::zero();

    // NST - 1 stages in flight: steps 0 .. NST-2 requested here, step J + NST - 1 right behind the barrier of step J (into the stage step
    // J - 1 read: every wave's reads of it completed before that wave reached the barrier)
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) request(st, st < nsteps ? st : nsteps - 1);
    auto do_step = [&](auto Jc, int step) __attribute__((always_inline)) {
        constexpr int stage = decltype(Jc)::value % NST, stage_fill = (decltype(Jc)::value + NST - 1) % NST;
        wait_vm<(NST - 2) * PT>();     // this step's pieces have landed (those of the NST - 2 later steps stay in flight)
        __builtin_amdgcn_s_barrier();  // ... everybody's have, and everybody is done reading the stage refilled next
        asm volatile("" ::: "memory");
        request(stage_fill, step + NST - 1 < nsteps ? step + NST - 1 : nsteps - 1);  // past the end: repeat the last step (never consumed)
        u32x4 a[NS][2], b[NS][2];
#pragma unroll
        for (int g = 0; g < NS; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[g][i] = *(const u32x4*)(smem + stage * STAGE + fa[g] + i * 32 * PITCH);
                b[g][i] = *(const u32x4*)(smem + stage * STAGE + fb[g] + i * 32 * PITCH);
            }
        if constexpr (INT) {
#pragma unroll
            for (int g = 0; g < NS; ++g)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = AC::mma(a[g][mi], b[g][ni], acc[mi][ni]);
        } else {
#pragma unroll
            for (int g = 0; g < NS; g += 2)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = AC::mma64(a[g][mi], a[g + 1][mi], b[g][ni], b[g + 1][ni], acc[mi][ni]);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of the stage are complete before it reaches the next barrier
    };
    auto chain = [&](auto self, auto Jc, int s0) -> void {
        constexpr int J = decltype(Jc)::value;
        do_step(std::integral_constant<int, J>{}, s0 + J);
        if constexpr (J + 1 < NST) {
            if (s0 + J + 1 < nsteps) self(self, std::integral_constant<int, J + 1>{}, s0);
        }
    };
    for (int s0 = 0; s0 < nsteps; s0 += NST) chain(chain, std::integral_constant<int, 0>{}, s0);
    wait_vm<0>();
    __syncthreads();

    // ---- epilogue: add the two K halves (raw accumulator words: int32 stays exact), transpose through LDS, 16-byte output rows
    typedef typename std::conditional<INT, int, float>::type word_t;
    typedef word_t word4 __attribute__((ext_vector_type(4)));
    constexpr int C_PITCH = BN + 4;
    {
        acc_t* xch = (acc_t*)smem;  // [rb][cb][mi][ni][lane]
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if (kh == 1) xch[(((rb * 2 + cb) * 2 + mi) * 2 + ni) * 64 + lane] = acc[mi][ni];
            }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if (kh == 0) acc[mi][ni] += xch[(((rb * 2 + cb) * 2 + mi) * 2 + ni) * 64 + lane];
            }
        __syncthreads();
    }
    word_t* ct = (word_t*)smem;  // [128][C_PITCH]
    if (kh == 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = rb * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    ct[r * C_PITCH + cb * 64 + ni * 32 + col] = acc[mi][ni][e];
                }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int u = tid + 512 * i, r = u >> 5, c4 = (u & 31) * 4;
        const int m = m0 + r;
        if (m < p.M) {
            const word4 v = *(const word4*)(ct + r * C_PITCH + c4);
            store_out4_any(p.epi, (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, m, (int64_t)nt * BN + c4);
        }
    }
}

// tuning[0] = 10 forces it (tuning[2]: stage geometry); automatic: a8w8_sq128_pays() in api.hip
bool plan_gemm_a8w8_sq128(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp) {
    if (a.elements_per_sample != 1 || a.W_group_mode != 0 || a.M < 2) return false;
    if (a.w_dtype != a.input_dtype) return false;
    if (!(a.input_dtype == GEMLITE_DT_INT8 || a.input_dtype == GEMLITE_DT_FP8E4 || a.input_dtype == GEMLITE_DT_FP8E5)) return false;
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.stride_on != 1 || a.N % 128 != 0 || a.K % 256 != 0) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0 || a.stride_xm % 16 != 0 || a.stride_wn % 16 != 0) return false;
    if ((int64_t)a.M * a.stride_xm >= (1ll << 31) || (int64_t)a.N * a.stride_wn >= (1ll << 31)) return false;
    if (!(a.output_dtype == GEMLITE_DT_FP32 || a.output_dtype == GEMLITE_DT_FP16 || a.output_dtype == GEMLITE_DT_BF16)) return false;
    if ((a.channel_scale_mode == 1 || a.channel_scale_mode == 3) &&
        !(a.meta_dtype == GEMLITE_DT_FP32 || a.meta_dtype == GEMLITE_DT_FP16 || a.meta_dtype == GEMLITE_DT_BF16)) return false;
    if ((a.channel_scale_mode == 1 || a.channel_scale_mode == 3) && ((uintptr_t)a.scales % 16) != 0) return false;
    const int oal = a.output_dtype == GEMLITE_DT_FP32 ? 16 : 8;  // 4 outputs per store
    if (((uintptr_t)a.out % oal) != 0 || (a.stride_om * (oal / 4)) % oal != 0) return false;
    const int64_t tiles = (int64_t)(a.N / 128) * ((a.M + 127) / 128);
    if (tiles > 0x7FFFFFFF) return false;
    typedef void (*fn_t)(const GenericParams);
    // tuning[2]: 0 / 4 = four stages of 128-byte K steps (three in flight), 5 = five (two 64-KB stages of 256-byte steps: measured, removed)
    const int geo = a.tuning[0] == 10 ? a.tuning[2] : 0;
    if (!(geo == 0 || geo == 4 || geo == 5)) return false;
    fn_t fn;
    size_t lds;
#define GL_SQ128(KS, NSTG) (a.input_dtype == GEMLITE_DT_INT8 ? gemm_a8w8_sq128_kernel<GEMLITE_DT_INT8, KS, NSTG>                            \
                            : (a.input_dtype == GEMLITE_DT_FP8E4 ? gemm_a8w8_sq128_kernel<GEMLITE_DT_FP8E4, KS, NSTG>                      \
                                                                 : gemm_a8w8_sq128_kernel<GEMLITE_DT_FP8E5, KS, NSTG>))
    if (geo == 5) fn = GL_SQ128(128, 5), lds = (size_t)5 * 256 * 128;
    else fn = GL_SQ128(128, 4), lds = (size_t)4 * 256 * 128;
#undef GL_SQ128
    if (lds < (size_t)128 * 132 * 4) lds = (size_t)128 * 132 * 4;
    g.splitk = 1;
    // K order between the row tiles of a column tile (k_order(), gl_async.h): groups of ONE step for fp8 — FP8 16384^2 M = 256 94.7 -> 89.6 us, 8192^2 M = 512 45.1 -> 43.2 —
    // and the plain order for int8 (8192^2 M = 512 46.3 -> 47.5, 4096^2 M = 1024 27.5 -> 28.1 with it); larger groups and the whole-K rotation lose on these
    // 2-MB tiles (profiles/r06/probe_k_order_groups*.log, probe_k_rotation_sq128_slower.log).  tuning[3] bits 24 .. 27 force a group size, & 4194304 = plain.
    g.flags = a.tuning[3] & ~((1 << 30) | 0x0F000000);
    if (a.tuning[3] & 0x0F000000) g.flags |= (1 << 30) | (a.tuning[3] & 0x0F000000);
    else if (!(a.tuning[3] & 4194304) && a.input_dtype != GEMLITE_DT_INT8 && (a.M + 127) / 128 >= 2 && ((a.N / 128) & 7) == 0) g.flags |= (1 << 30) | (1 << 24);
    lp.fn = (const void*)fn;
    lp.name = "gemm_a8w8_sq_kernel<128x128>";
    lp.grid = dim3((unsigned)tiles, 1, 1);
    lp.block = dim3(512, 1, 1);
    lp.lds_bytes = lds;
    lp.ws_bytes = 0;
    lp.slab_bytes = 0;
    return true;
}

// K rotation of the unsplit 64 x 64 tiles (gemm_a8w8_sq_kernel, gemm_mx_sq_kernel): row tile mt of a weight column tile starts its K loop at step
// mt nsteps / mtiles.  The row tiles of a column tile run on one XCD (block map of the kernels; needs N / 64 % 8 == 0), so each sibling CU pulls a different
// part of the tile from HBM and the others find it in that XCD's L2 — IF it is still there: an XCD works on min(column tiles per XCD, CUs per XCD /
// row tiles) weight tiles at a time, and the rotation pays while those fit its 4-MB L2 (2 MB: -8 .. -24 %; 4 MB: even; 7 MB: +17 %).
bool k_rotation_pays(const gemlite_hip_forward_args& a, int64_t tile_bytes) {
    if (a.tuning[3] & 4194304) return false;
    const int64_t mtiles = (a.M + 63) / 64, ntiles = a.N / 64;
    if (mtiles < 2 || (ntiles & 7) != 0) return false;
    const int64_t cus_per_xcd = resident_block_limit() / 8 > 0 ? resident_block_limit() / 8 : 1;
    const int64_t at_a_time = cus_per_xcd / mtiles > 0 ? cus_per_xcd / mtiles : 1;
    const int64_t tiles_x = ntiles / 8 < at_a_time ? ntiles / 8 : at_a_time;
    return tiles_x * tile_bytes <= (4ll << 20);
}
// The K-order bits of GenericParams::flags for the unsplit 64 x 64 tiles (k_order(), gl_async.h): whole-K rotation while the XCD's weight tiles stay well
// inside its L2, else the GROUPED order with groups of one step — every tile leads on one step of each run of mtiles steps, so a line only has to survive
// mtiles - 1 steps (late round 6, profiles/r06/probe_k_order_groups*.log: int8 4096 x 14336 M = 256, 7 MB per XCD: 34.5 -> 31.0 us where the whole rotation
// cost +17 %; 4096 x 8192, 4 MB: whole 21.4 = none 21.6, grouped 20.1; MXFP8 4096 x 8192 28 .. 31 -> 25.4 whole -> 23.4 grouped; where the whole rotation
// pays it stays ahead: 4096^2 M = 256 12.0 vs 12.8, 2048 x 8192 15.5 vs 17.9).  tuning[3] bits 24 .. 27 force a group size (1 + log2), & 4194304 = plain order.
int k_order_flags(const gemlite_hip_forward_args& a, int64_t tile_bytes) {
    if (a.tuning[3] & 4194304) return 0;
    if (a.tuning[3] & 0x0F000000) return (1 << 30) | (a.tuning[3] & 0x0F000000);
    const int64_t mtiles = (a.M + 63) / 64, ntiles = a.N / 64;
    if (mtiles < 2 || (ntiles & 7) != 0) return 0;
    const int64_t cus_per_xcd = resident_block_limit() / 8 > 0 ? resident_block_limit() / 8 : 1;
    const int64_t at_a_time = cus_per_xcd / mtiles > 0 ? cus_per_xcd / mtiles : 1;
    const int64_t tiles_x = ntiles / 8 < at_a_time ? ntiles / 8 : at_a_time;
    return tiles_x * tile_bytes < (4ll << 20) ? (1 << 30) : ((1 << 30) | (1 << 24));
}

bool plan_gemm_a8w8_sq(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp) {
    if (a.elements_per_sample != 1 || a.W_group_mode != 0 || a.M < 2) return false;
    if (a.w_dtype != a.input_dtype) return false;
    if (!(a.input_dtype == GEMLITE_DT_INT8 || a.input_dtype == GEMLITE_DT_FP8E4 || a.input_dtype == GEMLITE_DT_FP8E5)) return false;
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.stride_on != 1 || a.N % 64 != 0 || a.K % 256 != 0) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0 || a.stride_xm % 16 != 0 || a.stride_wn % 16 != 0) return false;
    if ((int64_t)a.M * a.stride_xm >= (1ll << 31) || (int64_t)a.N * a.stride_wn >= (1ll << 31)) return false;
    if (!(a.output_dtype == GEMLITE_DT_FP32 || a.output_dtype == GEMLITE_DT_FP16 || a.output_dtype == GEMLITE_DT_BF16)) return false;
    if ((a.channel_scale_mode == 1 || a.channel_scale_mode == 3) &&
        !(a.meta_dtype == GEMLITE_DT_FP32 || a.meta_dtype == GEMLITE_DT_FP16 || a.meta_dtype == GEMLITE_DT_BF16)) return false;
    if ((a.channel_scale_mode == 1 || a.channel_scale_mode == 3) && ((uintptr_t)a.scales % 16) != 0) return false;
    const int oal = a.output_dtype == GEMLITE_DT_FP32 ? 16 : 8;  // 4 outputs per store
    if (((uintptr_t)a.out % oal) != 0 || (a.stride_om * (oal / 4)) % oal != 0) return false;
    const int64_t tiles = (int64_t)(a.N / 64) * ((a.M + 63) / 64);
    // tuning[0] = 5 forces this kernel.  Automatic: see the caller (api.hip) — 65 .. 256 rows whose 64 x 64 tiles fill the chip once
    typedef void (*fn_t)(const GenericParams);
    // tuning[2] = 2 / 3: two (64 KB of LDS: two blocks per CU) / three stages (development A/B); default four
    // (one round of tiles: 4 stages in flight per block; two rounds: 2 stages = 64 KB of LDS, two co-resident blocks per CU cover each
    //  other's prologue and epilogue — 4096^2 int8 M = 384: 24.4 us with 4 stages, 19.3 with 2; M = 256: 13.6 vs 17.9)
    const int nst = (a.tuning[0] == 5 && (a.tuning[2] >= 2 && a.tuning[2] <= 4)) ? a.tuning[2] : (tiles <= 256 ? 4 : 2);
    fn_t fn = nullptr;
    auto pick = [&](auto dt) -> fn_t {
        constexpr int DT = decltype(dt)::value;
        // (tuning[3] & 2097152: the round-4 order of a step — requests first — for A/B runs)
        if (!(a.tuning[3] & 2097152)) return nst == 2 ? gemm_a8w8_sq_kernel<DT, 2, true> : (nst == 3 ? gemm_a8w8_sq_kernel<DT, 3, true> : gemm_a8w8_sq_kernel<DT, 4, true>);
        return nst == 2 ? gemm_a8w8_sq_kernel<DT, 2> : (nst == 3 ? gemm_a8w8_sq_kernel<DT, 3> : gemm_a8w8_sq_kernel<DT, 4>);
    };
    fn = a.input_dtype == GEMLITE_DT_INT8 ? pick(std::integral_constant<int, GEMLITE_DT_INT8>{})
         : (a.input_dtype == GEMLITE_DT_FP8E4 ? pick(std::integral_constant<int, GEMLITE_DT_FP8E4>{}) : pick(std::integral_constant<int, GEMLITE_DT_FP8E5>{}));
    g.splitk = 1;
    g.flags = (a.tuning[3] & ~((1 << 30) | 0x0F000000)) | k_order_flags(a, (int64_t)64 * a.K);
    lp.fn = (const void*)fn;
    lp.name = "gemm_a8w8_sq_kernel<64x64>";
    lp.grid = dim3((unsigned)tiles, 1, 1);
    lp.block = dim3(512, 1, 1);
    lp.lds_bytes = (size_t)nst * (64 + 64) * 256;
    if (lp.lds_bytes < 2 * 64 * 68 * 4) lp.lds_bytes = 2 * 64 * 68 * 4;  // (the two K halves of the epilogue)
    lp.ws_bytes = 0;
    lp.slab_bytes = 0;
    return true;
}

typedef void (*a8_kernel_fn)(const GenericParams);
template <int DT>
static const void* a8_pick(int mi) {
    a8_kernel_fn f = nullptr;
    switch (mi) {
        case 8: f = gemm_a8w8_mma_kernel<DT, 8, 4, 2>; break;
        case 4: f = gemm_a8w8_mma_kernel<DT, 4, 6, 3>; break;
        case 2: f = gemm_a8w8_mma_kernel<DT, 2, 8, 4>; break;
        case 1: f = gemm_a8w8_mma_kernel<DT, 1, 8, 4>; break;
        default: break;
    }
    return (const void*)f;
}

template <int DT>
static const void* a8_pick_lds(int mi) {
    a8_kernel_fn f = nullptr;
    if (mi == 8) f = gemm_a8w8_lds_kernel<DT, 8, 3>;
    else if (mi == 4) f = gemm_a8w8_lds_kernel<DT, 4, 4>;
    return (const void*)f;
}

// 8-wave kernels: M >= 2 (tuning[0]: 2 = the 4-wave kernel of round 1; tuning[1] = K slices; tuning[2] = tile rows / 32;
// tuning[3] & 64: weights straight from memory also for the 128- / 256-row tiles, the A/B switch of the LDS-B variant)
bool plan_gemm_a8w8_mma(const gemlite_hip_forward_args& a, GenericParams& g, LaunchPlan& lp) {
    if (a.elements_per_sample != 1 || a.W_group_mode != 0 || a.M < 2) return false;
    if (a.w_dtype != a.input_dtype) return false;
    if (!(a.input_dtype == GEMLITE_DT_INT8 || a.input_dtype == GEMLITE_DT_FP8E4 || a.input_dtype == GEMLITE_DT_FP8E5)) return false;
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.stride_on != 1 || a.N % 128 != 0 || a.K % 256 != 0) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0 || a.stride_xm % 16 != 0 || a.stride_wn % 16 != 0) return false;
    if ((int64_t)a.M * a.stride_xm >= (1ll << 31) || (int64_t)a.N * a.stride_wn >= (1ll << 31)) return false;
    if (!(a.output_dtype == GEMLITE_DT_FP32 || a.output_dtype == GEMLITE_DT_FP16 || a.output_dtype == GEMLITE_DT_BF16)) return false;
    if ((a.channel_scale_mode == 1 || a.channel_scale_mode == 3) &&
        !(a.meta_dtype == GEMLITE_DT_FP32 || a.meta_dtype == GEMLITE_DT_FP16 || a.meta_dtype == GEMLITE_DT_BF16)) return false;
    if ((a.channel_scale_mode == 1 || a.channel_scale_mode == 3) && ((uintptr_t)a.scales % 16) != 0) return false;
    int units = (int)(a.K / 256);
    // Tile rows: nothing is dequantised here, so a small tile costs no extra arithmetic (only more weight re-reads from
    // L2) while every K slice costs slab traffic and a tail: take the TALLEST tile (<= the rows M fills) that still gives
    // >= 112 tiles, i.e. at most two K slices; below that, the smallest tile with as many slices as needed.
    const int cap = a.M > 128 ? 8 : (a.M > 64 ? 4 : (a.M > 32 ? 2 : 1));
    int mi = 1;
    for (int c = cap; c >= 1; c >>= 1) {
        if ((int64_t)(a.N / 128) * ((a.M + 32 * c - 1) / (32 * c)) >= 112) { mi = c; break; }
    }
    // From 65 rows on, the two tiles whose weights travel through LDS (measured 7-25 % faster than the direct loads at every
    // shape of profiles/r02/wide/probe_a8lds.log, and faster than the 64-row direct tile at M = 256): 256 rows when they
    // alone fill the chip, or with two K slices of a long K (16384); else 128 rows.
    if (cap >= 4) {
        const int64_t t8 = (int64_t)(a.N / 128) * ((a.M + 255) / 256);
        mi = (cap >= 8 && (t8 >= 224 || (2 * t8 >= 224 && a.K >= 16384))) ? 8 : 4;
    }
    if (a.tuning[2] == 1 || a.tuning[2] == 2 || a.tuning[2] == 4 || a.tuning[2] == 8) mi = a.tuning[2];
    const int bm = 32 * mi;
    const bool lds_b = mi >= 4 && !(a.tuning[3] & 64);  // both operands through LDS (128-byte K steps)
    if (lds_b) units = (int)(a.K / 128);
    const int64_t tiles = (int64_t)(a.N / 128) * ((a.M + bm - 1) / bm);
    int splitk = 0;
    if (a.tuning[1] > 0) {
        if (a.tuning[1] > units) return false;
        splitk = a.tuning[1];
    } else {
        // (the LDS-fed tiles hold one block per CU: a slice count that spills into a second round loses to the one before it while that one
        //  keeps half the chip busy — 8192 x 2048 M = 384, 192 tiles: two slices 30.1 us, one 19.6; 4096 x 14336 M = 384, 96 tiles: three 65.9,
        //  two 47.6; profiles/r06/scan_a8w8_*.log)
        for (int sk = 1; sk <= units && sk <= 32; ++sk) {
            if (units / sk < (lds_b ? 4 : 2)) continue;
            if (lds_b && splitk >= 1 && tiles * sk > gl::resident_block_limit() && tiles * splitk >= 128) break;
            splitk = sk;
            if (tiles * sk >= 224) break;
        }
        if (!splitk) splitk = 1;
    }
    if (splitk > 1 && tiles > MAX_SPLITK_COUNTERS) return false;
    if ((uint64_t)splitk * bm * 128 * 4 >= (1ull << 31)) return false;
    const void* fn = lds_b ? (a.input_dtype == GEMLITE_DT_INT8 ? a8_pick_lds<GEMLITE_DT_INT8>(mi)
                              : (a.input_dtype == GEMLITE_DT_FP8E4 ? a8_pick_lds<GEMLITE_DT_FP8E4>(mi) : a8_pick_lds<GEMLITE_DT_FP8E5>(mi)))
                           : (a.input_dtype == GEMLITE_DT_INT8 ? a8_pick<GEMLITE_DT_INT8>(mi)
                              : (a.input_dtype == GEMLITE_DT_FP8E4 ? a8_pick<GEMLITE_DT_FP8E4>(mi) : a8_pick<GEMLITE_DT_FP8E5>(mi)));
    if (!fn) return false;
    g.splitk = splitk;
    g.flags = a.tuning[3];
    lp.fn = fn;
    lp.name = mi == 8 ? "gemm_a8w8_mma_kernel<256x128>" : (mi == 4 ? "gemm_a8w8_mma_kernel<128x128>"
              : (mi == 2 ? "gemm_a8w8_mma_kernel<64x128>" : "gemm_a8w8_mma_kernel<32x128>"));
    if (lds_b) lp.name = mi == 8 ? "gemm_a8w8_lds_kernel<256x128>" : "gemm_a8w8_lds_kernel<128x128>";
    lp.grid = dim3((unsigned)tiles, splitk, 1);
    lp.block = dim3(512, 1, 1);
    const size_t stages = lds_b ? (size_t)(mi == 8 ? 3 : 4) * (bm + 128) * 128 : (size_t)(mi == 8 ? 2 : (mi == 4 ? 3 : 4)) * bm * 256;
    const size_t xch = (size_t)4 * mi * 64 * 64;
    const size_t c_b = (size_t)(bm < 128 ? bm : 128) * 132 * 4 + 16;
    lp.lds_bytes = stages > xch ? stages : xch;
    if (lp.lds_bytes < c_b) lp.lds_bytes = c_b;
    lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * bm * 128 * 4 : 0;
    lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

// M >= 32, unpacked 8-bit weights with the same dtype as the activations, no group metadata (channel / token scales
// live in the epilogue).  Smaller M stays with the streaming kmajor kernel (one wave per column).
bool plan_gemm_a8w8(const gemlite_hip_forward_args& a, LaunchPlan& lp) {
    // below ~32 rows the 128-column tiles leave most CUs without a block (N / 128 tiles, K not split) and the
    // streaming kernel is faster (4096^2, M = 16: 21.5 vs 33 us)
    if (a.elements_per_sample != 1 || a.W_group_mode != 0 || a.M < 32) return false;
    if (a.w_dtype != a.input_dtype) return false;
    if (!(a.input_dtype == GEMLITE_DT_INT8 || a.input_dtype == GEMLITE_DT_FP8E4 || a.input_dtype == GEMLITE_DT_FP8E5)) return false;
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.N % 128 != 0 || a.K % 128 != 0) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0 || a.stride_xm % 16 != 0 || a.stride_wn % 16 != 0) return false;
    if ((int64_t)a.M * a.stride_xm >= (1ll << 31) || (int64_t)a.N * a.stride_wn >= (1ll << 31)) return false;
    // the largest tile that still gives every CU a block (K is not split): 128x128, 64x128, else 64x64
    int mi = 1, ni = 1;
    if (((a.M + 127) / 128) * (a.N / 128) >= 256) { mi = 2; ni = 2; }
    else if (((a.M + 63) / 64) * (a.N / 128) >= 256) { mi = 1; ni = 2; }
    const int64_t tiles = ((a.M + 64 * mi - 1) / (64 * mi)) * (a.N / (64 * ni));
    if (tiles > 0x7FFFFFFF) return false;
    auto pick = [&](auto dt) -> const void* {
        constexpr int DT = decltype(dt)::value;
        if (mi == 2) return (const void*)gemm_a8w8_kernel<DT, 2, 2>;
        return ni == 2 ? (const void*)gemm_a8w8_kernel<DT, 1, 2> : (const void*)gemm_a8w8_kernel<DT, 1, 1>;
    };
    switch (a.input_dtype) {
        case GEMLITE_DT_INT8: lp.fn = pick(std::integral_constant<int, GEMLITE_DT_INT8>{}); break;
        case GEMLITE_DT_FP8E4: lp.fn = pick(std::integral_constant<int, GEMLITE_DT_FP8E4>{}); break;
        default: lp.fn = pick(std::integral_constant<int, GEMLITE_DT_FP8E5>{}); break;
    }
    lp.name = mi == 2 ? "gemm_a8w8_kernel<128x128>" : (ni == 2 ? "gemm_a8w8_kernel<64x128>" : "gemm_a8w8_kernel<64x64>");
    lp.grid = dim3((unsigned)tiles, 1, 1);
    lp.block = dim3(256, 1, 1);
    lp.lds_bytes = 0;
    lp.ws_bytes = 0;
    lp.slab_bytes = 0;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// Few rows (M <= 16): a bandwidth kernel.  Block = 16 output columns (16 K-contiguous weight rows) x all of K, 8 waves
// dealing the 64-k chunks round-robin; a chunk is ONE 16-row MFMA (v_mfma_i32_16x16x64_i8, or two
// v_mfma_f32_16x16x32_{fp8,bf8} on the 8-byte halves of the same registers) whose operands are one 16-byte load per lane
// each — weight row n0 + (lane & 15), x row lane & 15, both at k = chunk * 64 + (lane >> 4) * 16: every weight row is read
// as 64-byte segments, each byte once (the 128-column tiles of the 8-wave kernel read 32-byte segments and leave 7/8 of
// the chip without a block at this M: 4096^2 int8, M = 16: 15.6 us).  Rows >= M read zeros through the descriptor's range
// check.  The eight partial 16 x 16 tiles are added through LDS (int32: exact), then the reference's epilogue
// (channel_scale_mode 0..3) runs per output.
// ---------------------------------------------------------------------------------------------------------------------
// MT = row tiles of 16 (round 3: 17..64 rows used to fall to the 32-row tile of the 8-wave kernel — 17.4 us at 4096^2 int8, M = 32,
// against 6.7 us for 16 rows here): the weight fragment of a chunk is loaded once and multiplied with MT x fragments.
// FQ (round 4): the launch quantises the activations itself (gl_coopquant.h) — p.x / p.epi.scales_x are workspace copies that the
// blocks fill cooperatively; this block's first weight fragments are requested before that and arrive while it waits for the flags.
template <int DT, int MT, bool FQ>
__global__ __launch_bounds__(512) void a8w8_rows_kernel(const GenericParams p) {
    typedef typename std::conditional<DT == GEMLITE_DT_INT8, i32x4, f32x4>::type acc_t;
    __shared__ __attribute__((aligned(16))) uint32_t red[MT][8][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kg = lane >> 4;
    int tile = blockIdx.x;
    if constexpr (FQ) {  // the first gridDim.x - N / 16 blocks quantise rows and leave
        const int nprod = (int)gridDim.x - p.N / 16;
        if (tile < nprod) {
            cq::produce<DT>(p, nprod, (float*)&red[0][0][0][0]);
            cq::depart(p);
            return;
        }
        tile -= nprod;
    }
    const int64_t n0 = (int64_t)tile * 16;
    const int nchunks = (int)(p.K / 64);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)((int64_t)(p.N - 1) * p.stride_wn + p.K), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, (short)0, (int)((int64_t)(p.M - 1) * p.stride_xm + p.K), 0x00020000);
    const uint32_t wvoff = (uint32_t)((n0 + c) * p.stride_wn + kg * 16);
    uint32_t xvoff[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
        xvoff[t] = c + 16 * t < p.M ? (uint32_t)((int64_t)(c + 16 * t) * p.stride_xm + kg * 16) : 0x80000000u;  // rows >= M: zeros
    acc_t acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = acc_t{0, 0, 0, 0};
    constexpr int D = MT == 1 ? 8 : (MT == 2 ? 6 : 4);  // chunks in flight per wave ((1 + MT) x 16 bytes per lane each)
    u32x4 wb[D], xb[D][MT];
    const int mine = (nchunks - wave + 7) >> 3;  // chunks wave, wave + 8, ...
    auto load_w = [&](int slot, int i) __attribute__((always_inline)) {
        const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((wave + 8 * i) * 64);
        wb[slot] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wvoff, so, 0);
    };
    auto load_x = [&](int slot, int i) __attribute__((always_inline)) {
        const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((wave + 8 * i) * 64);
#pragma unroll
        for (int t = 0; t < MT; ++t) xb[slot][t] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xvoff[t], so, 0);
    };
    auto load = [&](int slot, int i) __attribute__((always_inline)) {
        load_w(slot, i);
        load_x(slot, i);
    };
    auto mma = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            if constexpr (DT == GEMLITE_DT_INT8) {
                acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, xb[slot][t]), __builtin_bit_cast(i32x4, wb[slot]), acc[t], 0, 0, 0);
            } else {
                const u32x4 a = xb[slot][t], b = wb[slot];
                const long a0 = (long)(((uint64_t)a[1] << 32) | a[0]), a1 = (long)(((uint64_t)a[3] << 32) | a[2]);
                const long b0 = (long)(((uint64_t)b[1] << 32) | b[0]), b1 = (long)(((uint64_t)b[3] << 32) | b[2]);
                if constexpr (DT == GEMLITE_DT_FP8E4) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, acc[t], 0, 0, 0);
                } else {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(a0, b0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(a1, b1, acc[t], 0, 0, 0);
                }
            }
        }
    };
    if constexpr (FQ) {
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < mine) load_w(j, j);
        cq::wait_rows<DT>(p, (float*)&red[0][0][0][0]);
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < mine) load_x(j, j);
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < mine) load(j, j);
    }
    // (the ring's requests sit behind `if (base + j + D < mine)`: hipcc retires each group of D chunks with one full vmcnt(0) at the
    //  loop header — the group's D x (1 + MT) KB per wave are requested together and 8 waves per CU interleave, which keeps >= 100 KB
    //  per CU in flight; an unconditional steady-state loop compiles to the same bulk-synchronous shape, tried in round 4)
    for (int base = 0; base < mine; base += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            if (base + j < mine) {
                mma(j);
                if (base + j + D < mine) load(j, base + j + D);
            }
        }
    }
    // one 16-byte store of the fragment (storing it element by element through bit_cast<uint32_t>(acc[r]) made hipcc 7.2
    // overwrite acc[1..3] with acc[0] after the loop — every output row read row 4 (m / 4); found with structured inputs)
#pragma unroll
    for (int t = 0; t < MT; ++t) *(acc_t*)&red[t][wave][lane][0] = acc[t];
    __syncthreads();
    for (int u = tid; u < MT * 256; u += 512) {
        const int t = u >> 8, l = u & 63, r = (u >> 6) & 3;
        const int m = 16 * t + 4 * (l >> 4) + r;  // C fragment of a 16 x 16 MFMA: column lane & 15, rows 4 (lane >> 4) + r
        float v;
        if constexpr (DT == GEMLITE_DT_INT8) {
            int sum = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += (int)red[t][w][l][r];
            v = (float)sum;
        } else {
            v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += __builtin_bit_cast(float, red[t][w][l][r]);
        }
        if (m < p.M) epilogue_store(p.epi, v, m, n0 + (l & 15));
    }
    if constexpr (FQ) cq::depart(p);
}

// 2 <= M <= 64 (M = 1 too when forced with tuning[0] = 4), same-dtype 8-bit operands, both K-contiguous.  17..64 rows: 2 / 4 row
// tiles per block while every block's re-read of x from L2 (M K bytes) stays below the 8-wave kernel's fixed cost — measured
// crossover in profiles/r03/probe_a8w8_rows_mt.log
bool plan_a8w8_rows(const gemlite_hip_forward_args& a, LaunchPlan& lp, bool fq) {
    if (a.elements_per_sample != 1 || a.W_group_mode != 0 || a.M > 64 || a.M < 1) return false;
    if (a.w_dtype != a.input_dtype) return false;
    if (!(a.input_dtype == GEMLITE_DT_INT8 || a.input_dtype == GEMLITE_DT_FP8E4 || a.input_dtype == GEMLITE_DT_FP8E5)) return false;
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.N % 16 != 0 || a.K % 64 != 0) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0 || a.stride_xm % 16 != 0 || a.stride_wn % 16 != 0) return false;
    if ((int64_t)a.M * a.stride_xm + a.K >= (1ll << 31) || (int64_t)a.N * a.stride_wn + a.K >= (1ll << 31)) return false;
    const int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
    // round 6: x through LDS in whole cache lines (gemm_w8_rows.hip; tuning[3] & 524288 keeps the register-fed kernel of round 3 below).  Measured
    // on 4096^2, 8192^2, 14336 x 4096, 4096 x 14336, int8 and fp8, `layer(x)` (profiles/r06/probe_w8_rows_lds_*.log):
    //   * layers whose 16-column blocks are ONE resident round (N <= 4096 on 256 CUs): ahead of the round-3 kernel from 2 rows (4096^2 M = 2 / 16 /
    //     32 / 64: 9.2 / 10.8 / 12.6 / 17.2 -> 8.6 / 9.7 / 11.7 / 13.7 us) and ahead of every tile kernel while the blocks' re-reads of x stay below
    //     256 MiB (4096 x 14336 M = 32 / 64 — 117 / 235 MB: 31.3 / 37.7 -> 25.7 / 31.9 us; the round-3 kernel's budget was 88 MB);
    //   * more blocks than CUs: the 64-KB form, two blocks per CU, from 4 rows (8192^2 M = 4 / 8 / 16: 20.5 / 22.1 / 23.8 (tiles) -> 19.9 / 20.6 /
    //     22.0); not on layers with 192 or more 64-column tiles, where the unsplit tiles take over from 5 rows (14336 x 4096 M = 8: 18.2 vs 19.8)
    //     and the round-3 kernel is ahead below (M = 2 / 4: 17.8 / 18.6 vs 18.7 / 19.4).
    if (!fq && !(a.tuning[3] & 524288) && a.K % 256 == 0) {
        const int64_t blocks = a.N / 16;
        const bool forced = a.tuning[0] == 4;
        bool lds;
        if (blocks <= resident_block_limit()) lds = (a.M >= 2 || forced) && (forced || (int64_t)a.M * a.K * blocks <= (256ll << 20));
        else lds = a.M <= 32 && (forced ? a.M >= 2 : (a.M >= 4 && a.M <= 16 && a.N / 64 < 192));
        if (lds && plan_w8_rows_lds(a, lp, w8_lds_mt(a.M), w8_lds_two(a))) return true;
    }
    // every block re-reads its M rows of x from L2: M K bytes x N / 16 blocks.  Up to ~90 MB of that the 16-column blocks beat the
    // 8-wave tiles (4096^2: 7.5 / 8.8 / 13.4 us at M = 17 / 32 / 64 against 16.9 / 17.4 / 18.3; 8192^2 M = 17: 21.3 vs 24.7), beyond
    // they lose (8192^2 M = 32: 28.1 vs 24.9; M = 64: 43.9 vs 27.7) — profiles/r03/probe_a8w8_rows_mt.log.  tuning[0] = 4 forces them.
    if (mt > 1 && a.tuning[0] != 4 && (int64_t)a.M * a.K * (a.N / 16) > (88ll << 20)) return false;
    // round 4 (profiles/r04/probe_rows_vs_tiles*.log): against the unsplit 64 x 64 tiles, which fill the chip from N / 64 >= 128 column
    // tiles even with one row tile, the crossover sits at M N K ~ 800 M (8192^2: M = 12), and at ~ 250 M from 192 column tiles
    // (14336 x 4096: M = 4; M = 16: 24.4 vs 19.1 us, M = 64: 30.8 vs 21.9)
    if (a.M >= 2 && a.tuning[0] != 4 && !fq && a.N % 64 == 0 && a.K % 256 == 0 && a.N / 64 >= 128 &&
        (int64_t)a.M * a.N * a.K > (a.N / 64 >= 192 ? 250000000ll : 800000000ll)) return false;
    auto pick = [&](auto dt) -> const void* {
        constexpr int DT = decltype(dt)::value;
        if (fq) return mt == 1 ? (const void*)a8w8_rows_kernel<DT, 1, true> : (mt == 2 ? (const void*)a8w8_rows_kernel<DT, 2, true> : (const void*)a8w8_rows_kernel<DT, 4, true>);
        return mt == 1 ? (const void*)a8w8_rows_kernel<DT, 1, false> : (mt == 2 ? (const void*)a8w8_rows_kernel<DT, 2, false> : (const void*)a8w8_rows_kernel<DT, 4, false>);
    };
    switch (a.input_dtype) {
        case GEMLITE_DT_INT8: lp.fn = pick(std::integral_constant<int, GEMLITE_DT_INT8>{}); break;
        case GEMLITE_DT_FP8E4: lp.fn = pick(std::integral_constant<int, GEMLITE_DT_FP8E4>{}); break;
        default: lp.fn = pick(std::integral_constant<int, GEMLITE_DT_FP8E5>{}); break;
    }
    lp.name = fq ? (mt == 1 ? "a8w8_rows_fq_kernel<16x16>" : (mt == 2 ? "a8w8_rows_fq_kernel<32x16>" : "a8w8_rows_fq_kernel<64x16>"))
                 : (mt == 1 ? "a8w8_rows_kernel<16x16>" : (mt == 2 ? "a8w8_rows_kernel<32x16>" : "a8w8_rows_kernel<64x16>"));
    lp.grid = dim3((unsigned)(a.N / 16 + (fq ? (a.M < 64 ? a.M : 64) : 0)), 1, 1);  // fq: one producer block per row in front
    lp.block = dim3(512, 1, 1);
    lp.lds_bytes = 0;
    lp.slab_bytes = fq ? cq::payload_bytes(a.M, a.K) : 0;  // the quantised rows and their scales
    lp.ws_bytes = fq ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// A16W8 (round 4): 8-bit weight-only layers — int8 / fp8 unpacked K-contiguous weights under fp16 / bf16 activations, no metadata or one
// scale per output channel (helper.py:88-171; the reference runs them through its GEMV / GEMM_SPLITK / GEMM kernels with W_nbits = 8).
// Rounds 1-3 had one kernel for them, kmajor_w8a16_kernel: a wave per column, 4 rows of x per pass, the weights re-streamed for every
// 4 rows — 9.2 us at 4096^2 M = 1, 36.6 us at M = 16 (profiles/r04/probe_processors.log).  This is a8w8_rows_kernel's shape with the
// weights converted in registers: block = 16 output columns x all of K, 8 waves dealing 64-k chunks, lane (c = lane & 15, q = lane >> 4)
// loads the 16 weight bytes k = 16 q .. 16 q + 15 of column c and, per 16-row tile, the 32 bytes of x that face them; a chunk is two
// v_mfma_f32_16x16x32_{f16,bf16} per row tile (k = 16 q + j and 16 q + 8 + j: any k assignment works as long as both operands use it).
// int8 -> fp16: 0x6400 | (b ^ 0x80) = 1024 + (b + 128) exactly, minus 1152 (packed); int8 -> bf16 and fp8 -> either: hardware converters.
// The channel scale multiplies the fp32 sum once (pre-scale: a per-column constant; post-scale: the reference's epilogue).  More than 64
// rows: 64-row tiles along grid.y (weights re-read from L2 per tile) until a tile kernel takes these layers.
// ---------------------------------------------------------------------------------------------------------------------
// WDT 103 / 104 (round 4): the block-scaled weight-only layers (A16W8_MXFP / A16W4_MXFP, helper.py:372-400) on the same kernel — fp8 e4m3 /
// fp4 e2m1 weights, K-contiguous, one e8m0 scale per 32 k ([K/32][N] bytes).  A lane's 16 k of a chunk lie inside ONE block, so it loads
// one scale byte per chunk and the hardware converters apply it (v_cvt_scalef32_pk_{f16,bf16}_{fp8,fp4}: exact, a power of two).
constexpr int A16_MXW8 = 103, A16_MXW4 = 104;
template <typename Tag, int WDT, int MT>
__global__ __launch_bounds__(512) void a16w8_rows_kernel(const GenericParams p) {
    using TR = F16Traits<Tag>;
    constexpr bool MXW = WDT == A16_MXW8 || WDT == A16_MXW4;
    constexpr int WB = WDT == A16_MXW4 ? 8 : 16;  // weight bytes per lane and 64-k chunk
    __shared__ __attribute__((aligned(16))) float red[MT][8][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    const int mbase = (int)blockIdx.y * (16 * MT);
    const int nchunks = p.K / 64;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)((int64_t)(p.N - 1) * p.stride_wn + p.K * WB / 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, (short)0, (int)(((int64_t)(p.M - 1) * p.stride_xm + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(MXW ? p.scales : p.w), (short)0, MXW ? (int)((int64_t)(p.K / 32 - 1) * p.stride_meta_g + (int64_t)(p.N - 1) * p.stride_meta_n + 1) : 4, 0x00020000);
    const uint32_t wvoff = (uint32_t)((n0 + c) * p.stride_wn + q * WB);
    const uint32_t svoff = MXW ? (uint32_t)((n0 + c) * p.stride_meta_n + (int64_t)(q >> 1) * p.stride_meta_g) : 0u;
    uint32_t xvoff[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = mbase + c + 16 * t;
        xvoff[t] = m < p.M ? (uint32_t)(((int64_t)m * p.stride_xm + q * 16) * 2) : 0x80000000u;  // rows >= M: zeros
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int D = MT == 1 ? 8 : (MT == 2 ? 4 : 2);  // chunks in flight per wave (16 + 32 MT bytes per lane each; 8 = all of K = 4096)
    u32x4 wb[D], xb[D][MT][2];
    uint32_t sb[D];  // MXW: this lane's block scale of the chunk (e8m0 byte)
    const int mine = (nchunks - wave + 7) >> 3;  // chunks wave, wave + 8, ...
    auto load = [&](int slot, int i) __attribute__((always_inline)) {
        const int ch = wave + 8 * i;
        if constexpr (WB == 16) {
            wb[slot] = __builtin_amdgcn_raw_buffer_load_b128(rsW, wvoff, (uint32_t)__builtin_amdgcn_readfirstlane(ch * 64), 0);
        } else {
            const u32x2 w2 = __builtin_amdgcn_raw_buffer_load_b64(rsW, wvoff, (uint32_t)__builtin_amdgcn_readfirstlane(ch * 32), 0);
            wb[slot] = (u32x4){w2[0], w2[1], 0u, 0u};
        }
        if constexpr (MXW) sb[slot] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rsS, svoff, (uint32_t)__builtin_amdgcn_readfirstlane(ch * 2 * (int)p.stride_meta_g), 0);
        const uint32_t xo = (uint32_t)__builtin_amdgcn_readfirstlane(ch * 128);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            xb[slot][t][0] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xvoff[t], xo, 0);
            xb[slot][t][1] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xvoff[t], xo + 16u, 0);
        }
    };
    // 8 weight bytes (two dwords) -> one B fragment: 8 values of the activation type (gl_w8cvt.h: shared with w8_rows_lds_kernel)
    auto convert = [&](uint32_t lo, uint32_t hi) __attribute__((always_inline)) -> u32x4 {
        if constexpr (!MXW) return w8_to_frag<Tag, WDT>(lo, hi);
        else return (u32x4){0u, 0u, 0u, 0u};
    };
    // block-scaled rows: 8 values (k = 8 h .. 8 h + 7 of the lane's 16) with the block scale applied by the converter
    auto convert_mx = [&](int slot, int h) __attribute__((always_inline)) -> u32x4 {
        const float sc = __builtin_bit_cast(float, sb[slot] << 23);  // e8m0 byte = the exponent field (0 -> 0.0: the quantisers floor at 2^-30)
        u32x4 f;
        if constexpr (WDT == A16_MXW8) {
            const uint32_t d0 = wb[slot][2 * h], d1 = wb[slot][2 * h + 1];
            if constexpr (TR::DT == GEMLITE_DT_FP16) {
                f[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(d0, sc, false));
                f[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(d0, sc, true));
                f[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(d1, sc, false));
                f[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(d1, sc, true));
            } else {
                f[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, sc, false));
                f[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d0, sc, true));
                f[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, sc, false));
                f[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d1, sc, true));
            }
        } else {
            const uint32_t d = wb[slot][h];  // 8 nibbles, k even in the low nibble
            if constexpr (TR::DT == GEMLITE_DT_FP16) {
                f[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(d, sc, 0));
                f[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(d, sc, 1));
                f[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(d, sc, 2));
                f[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(d, sc, 3));
            } else {
                f[0] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(d, sc, 0));
                f[1] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(d, sc, 1));
                f[2] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(d, sc, 2));
                f[3] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(d, sc, 3));
            }
        }
        return f;
    };
    auto mma = [&](int slot) __attribute__((always_inline)) {
        u32x4 b0, b1;
        if constexpr (MXW) {
            b0 = convert_mx(slot, 0);
            b1 = convert_mx(slot, 1);
        } else {
            b0 = convert(wb[slot][0], wb[slot][1]);
            b1 = convert(wb[slot][2], wb[slot][3]);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            if constexpr (TR::DT == GEMLITE_DT_FP16) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, xb[slot][t][0]), __builtin_bit_cast(h8_t, b0), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, xb[slot][t][1]), __builtin_bit_cast(h8_t, b1), acc[t], 0, 0, 0);
            } else {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_t, xb[slot][t][0]), __builtin_bit_cast(b8_t, b0), acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_t, xb[slot][t][1]), __builtin_bit_cast(b8_t, b1), acc[t], 0, 0, 0);
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
    for (int t = 0; t < MT; ++t) *(f32x4*)&red[t][wave][lane][0] = acc[t];
    __syncthreads();
    for (int u = tid; u < MT * 256; u += 512) {
        const int t = u >> 8, l = u & 63, r = (u >> 6) & 3;
        const int m = mbase + 16 * t + 4 * (l >> 4) + r;  // C fragment of a 16 x 16 MFMA: column lane & 15, rows 4 (lane >> 4) + r
        const int64_t n = n0 + (l & 15);
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[t][w][l][r];
        if (m < p.M) {
            const float sc = (!MXW && p.w_mode == 2) ? load_as_float(p.scales, n, p.meta_dt) : 1.f;  // per-channel pre-scale: once, on the sum
            epilogue_store(p.epi, v * sc, m, n);
        }
    }
}

// Where w8_rows_lds_kernel runs AHEAD of the 8-bit weight-only tile kernel (api.hip asks before its M N K > 850 M rule; measured `layer(x)`,
// profiles/r06/probe_w8_rows_lds_*.log, A16W8_INT8 / _FP8):
//   * one resident round of 16-column blocks (N <= 4096 on 256 CUs), 4 .. 64 rows, while the blocks' re-reads of x stay below 256 MiB:
//     4096^2 M = 64 (134 MB) 17.0 (tile) -> 12.7 us; 4096 x 14336 M = 16 / 32 (117 / 235 MB) 26.0 / 26.8 -> 20.2 / 25.3, M = 64 (470 MB) 29.6 vs 34.3;
//   * more blocks than CUs (the 64-KB form, two blocks per CU): up to 16 rows — 8192^2 M = 16: 24.4 -> 21.7, 14336 x 4096 M = 16: 23.0 -> 21.7;
//     M = 24: 25.3 (tile) vs 25.6.
bool a16w8_rows_lds_pays(const gemlite_hip_forward_args& a) {
    if ((a.tuning[3] & 524288) || a.K % 256 != 0 || a.N % 16 != 0 || a.M < 4 || a.M > 64) return false;
    const int64_t blocks = a.N / 16;
    if (blocks <= resident_block_limit()) return (int64_t)a.M * a.K * 2 * blocks <= (256ll << 20);
    return a.M <= 16;
}

// (block-scaled weight-only layers: input_dtype MXFP16 / MXBF16, W_nbits 8 (fp8 bytes) or 4 (two e2m1 codes per byte), group 32)
bool plan_a16w8_rows(const gemlite_hip_forward_args& a, LaunchPlan& lp) {
    const bool mx = a.input_dtype == GEMLITE_DT_MXFP16 || a.input_dtype == GEMLITE_DT_MXBF16;
    if (a.M < 1 || a.M > 65535 * 64) return false;
    if (mx) {
        if (a.group_size != 32 || !a.scales || !(a.W_nbits == 8 || a.W_nbits == 4)) return false;
        if ((int64_t)(a.K / 32) * a.stride_meta_g + (int64_t)a.N * a.stride_meta_n >= (1ll << 31)) return false;
    } else {
        if (a.elements_per_sample != 1) return false;
        if (!(a.input_dtype == GEMLITE_DT_FP16 || a.input_dtype == GEMLITE_DT_BF16)) return false;
        if (!(a.w_dtype == GEMLITE_DT_INT8 || a.w_dtype == GEMLITE_DT_FP8E4 || a.w_dtype == GEMLITE_DT_FP8E5)) return false;
    }
    if (a.stride_wk != 1 || a.stride_xk != 1 || a.N % 16 != 0 || a.K % 64 != 0) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0 || (a.stride_xm * 2) % 16 != 0 || a.stride_wn % 16 != 0) return false;
    if (((int64_t)a.M * a.stride_xm + a.K) * 2 >= (1ll << 31) || (int64_t)a.N * a.stride_wn + a.K >= (1ll << 31)) return false;
    // round 6: x through LDS in whole cache lines (gemm_w8_rows.hip; tuning[3] & 524288 keeps the register-fed kernel of round 4 below) from 4
    // rows — at 2 / 3 rows the two tie on one-round layers and the round-4 kernel leads on the others (a16w8_rows_lds_pays() has the numbers)
    if (!mx && !(a.tuning[3] & 524288) && a.K % 256 == 0 && (a.M >= 4 || (a.tuning[3] & 1048576)) && (a.N / 16 <= resident_block_limit() || a.M <= 32) &&
        plan_w8_rows_lds(a, lp, w8_lds_mt(a.M), w8_lds_two(a))) return true;
    const int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
    typedef void (*fn_t)(const GenericParams);
    fn_t fn = nullptr;
    auto pick = [&](auto tag, auto wdt) -> fn_t {
        using T = decltype(tag);
        constexpr int W = decltype(wdt)::value;
        return mt == 1 ? a16w8_rows_kernel<T, W, 1> : (mt == 2 ? a16w8_rows_kernel<T, W, 2> : a16w8_rows_kernel<T, W, 4>);
    };
    typedef std::integral_constant<int, GEMLITE_DT_INT8> I8;
    typedef std::integral_constant<int, GEMLITE_DT_FP8E4> F8;
    typedef std::integral_constant<int, GEMLITE_DT_FP8E5> B8;
    const bool f16 = a.input_dtype == GEMLITE_DT_FP16 || a.input_dtype == GEMLITE_DT_MXFP16;
    typedef std::integral_constant<int, A16_MXW8> M8;
    typedef std::integral_constant<int, A16_MXW4> M4;
    if (mx) fn = a.W_nbits == 8 ? (f16 ? pick(half_tag{}, M8{}) : pick(bf16_tag{}, M8{})) : (f16 ? pick(half_tag{}, M4{}) : pick(bf16_tag{}, M4{}));
    else if (a.w_dtype == GEMLITE_DT_INT8) fn = f16 ? pick(half_tag{}, I8{}) : pick(bf16_tag{}, I8{});
    else if (a.w_dtype == GEMLITE_DT_FP8E4) fn = f16 ? pick(half_tag{}, F8{}) : pick(bf16_tag{}, F8{});
    else fn = f16 ? pick(half_tag{}, B8{}) : pick(bf16_tag{}, B8{});
    lp.fn = (const void*)fn;
    static const char* names[3][3] = {{"a16w8_rows_kernel<16x16>", "a16w8_rows_kernel<32x16>", "a16w8_rows_kernel<64x16>"},
                                      {"a16w8_mxfp_rows_kernel<16x16>", "a16w8_mxfp_rows_kernel<32x16>", "a16w8_mxfp_rows_kernel<64x16>"},
                                      {"a16w4_mxfp_rows_kernel<16x16>", "a16w4_mxfp_rows_kernel<32x16>", "a16w4_mxfp_rows_kernel<64x16>"}};
    lp.name = names[mx ? (a.W_nbits == 8 ? 1 : 2) : 0][mt == 1 ? 0 : (mt == 2 ? 1 : 2)];
    lp.grid = dim3((unsigned)(a.N / 16), (unsigned)((a.M + 16 * mt - 1) / (16 * mt)), 1);
    lp.block = dim3(512, 1, 1);
    lp.lds_bytes = 0;
    lp.ws_bytes = 0;
    lp.slab_bytes = 0;
    return true;
}

}  // namespace gl
