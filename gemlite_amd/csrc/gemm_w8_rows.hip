// gemm_w8_rows.hip — 2 .. 64 activation rows x UNPACKED 8-bit weights (round 6): the x-through-LDS treatment of gemm_wn_rows.hip for the two
// 8-bit few-row families of BASELINE config 4 and their weight-only siblings:
//   * A8W8   (x and weights int8 / fp8 e4m3 / e5m2, helper.py:405-500 "A8W8_*_dynamic"; matmul of gemm_splitK_kernels.py:277-450 with W_nbits = 8)
//   * A16W8  (x fp16 / bf16, weights int8 / fp8, helper.py:88-171 "A16W8_*"), weights converted in registers (gl_w8cvt.h).
// Rounds 3-5 ran these on a8w8_rows_kernel / a16w8_rows_kernel (gemm_a8w8.hip): the same launch shape — N / 16 blocks of 16 output columns,
// 8 waves dealing K among themselves, K never split across blocks — but with the A fragments of x loaded straight into registers: one
// wave-level request = 16 rows x 64 bytes = sixteen HALF cache lines, and EVERY block re-reads all of x through its CU's L2 -> L1 path.  At
// 4096^2 that is what the launch spends its time on from 17 rows (A16W8 12.6 / 16.9 us, A8W8 8.8 / 13.4 us at M = 32 / 64),
// and it is exactly what gemm_w4_rows_kernel cured for packed 4-bit words in round 5.  Here, as there:
//   * x: LDS-DMA pieces of 1 KiB = whole 128-byte lines (RPI = 2 .. 8 rows per instruction), two buffers per wave, the 16-byte slot of
//     (row r, logical slot s) stored at slot s ^ f(r) through the SOURCE address so that the ds_read_b128 of an A fragment is conflict-free
//     (lane groups of ds_read_b128: MI355X_MICROARCH.md section LDS; f below; tests/test_host_cpu.py restates the layout in numpy);
//     piece i + 2 is requested when piece i has been consumed.
//   * weights: K-contiguous rows, so lane (j = lane & 15, kb = lane >> 4) loads the 16 bytes k = 64 b + 16 kb .. + 15 of column j directly in
//     MFMA layout — four non-temporal 16-byte loads per 256-k chunk, two chunks (8 KB per wave, 64 KB per CU) ahead.  No LDS turn-around.
//     16-bit x: the lane's 16 k are two MFMA k-blocks of 8 (k = 16 kb + 8 e + 0..7) — any assignment of k to MFMA slots is correct as
//     long as both operands use it, so the A fragments are the 16-byte slots 2 kb + e of the 64-k block.
//   * all requests are inline asm retired by hand-counted waits (gl_async.h; scripts/isa_asmloads.py audits the generated code).
// Numerics are those of the kernels it replaces: int8 exact in int32; fp8 / 16-bit: fp32 MFMA accumulation per wave over its chunks, the eight
// partial tiles added in wave order — the chunk -> wave deal differs from the round-4 kernels (256-k chunks instead of 64-k), so float results
// differ from theirs by summation order only.
#include <type_traits>

#include "gl_common.h"
#include "gl_async.h"
#include "gl_w8cvt.h"

namespace gl {

typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace rows8 {
constexpr int NW = 8;  // waves per block
template <int OFF>
__device__ __forceinline__ void gld128_nt(u32x4& dst, const char* base, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(dst) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void tie4(u32x4& v) { asm volatile("" : "+v"(v)); }

// geometry of the x pieces (also used by the host planner for the LDS size).  BUFCAP = bytes of one x buffer: 8192, or 4096 for the forms that
// run TWO blocks per CU (64 KB of LDS per block) on layers with more 16-column blocks than CUs — a block's prologue, K-part join and store then
// overlap the other block's stream instead of idling the CU between rounds (up to 32 rows: 64-byte row pieces would be half cache lines again)
template <int XB, int MT, int BUFCAP = 8192>
struct Geo {
    static constexpr int CAP = BUFCAP / (16 * MT) >= 512 ? 512 : (BUFCAP / (16 * MT) >= 256 ? 256 : 128);  // row bytes that keep a buffer <= BUFCAP
    static constexpr int ROWB = 256 * XB < CAP ? 256 * XB : CAP;  // bytes per row of a piece (a piece never exceeds the 256-k chunk)
    static constexpr int BPP = ROWB / (64 * XB);                  // 64-k blocks per piece
    static constexpr int NP = 4 / BPP;                            // pieces per chunk
    static constexpr int PPR = ROWB / 16;                         // 16-byte slots per row
    static constexpr int RPI = 1024 / ROWB;                       // rows per LDS-DMA instruction
    static constexpr int DPI = 16 * MT / RPI;                     // LDS-DMA instructions per piece
    static constexpr int XBUF = DPI * 1024;
    static constexpr int WAVE_LDS = 2 * XBUF;
    static_assert(BPP >= 1 && NP * BPP == 4 && DPI * RPI == 16 * MT && XBUF <= BUFCAP && MT * 1024 <= WAVE_LDS, "x piece geometry");
};
// swizzle of row j (0 .. 15 inside its 16-row tile): which slot bits the row flips.  256- / 512-byte rows: the row's low four bits (the sixteen
// rows of one ds_read_b128 lane group land on sixteen different slots of the 256-byte bank row).  128-byte rows (two rows per bank row): three
// bits; for 16-bit x the two k-blocks of a lane are ADJACENT slots (2 kb + e), and the lane groups pair kb with kb + 1 — rows 4 .. 11 get one
// more flip of bit 1 so that the halves of a group stay apart (checked exhaustively by tests/test_host_cpu.py::test_w8_rows_lds_layout)
template <int PPR, bool A16>
__device__ __host__ constexpr int fswz(int j) {
    return PPR >= 16 ? (j & 15) : (A16 ? (((j >> 1) & 7) ^ ((((j + 4) >> 3) & 1) << 1)) : ((j >> 1) & 7));
}
}  // namespace rows8

// XDT: GEMLITE_DT_FP16 / BF16 (weights of type WDT converted to it) or GEMLITE_DT_INT8 / FP8E4 / FP8E5 (= WDT).  MT = row tiles of 16.
// BUFCAP: see Geo (4096: two blocks per CU)
template <int XDT, int WDT, int MT, int BUFCAP = 8192>
__global__ __launch_bounds__(rows8::NW * 64, (BUFCAP == 4096 ? 2 : 1)) void w8_rows_lds_kernel(const GenericParams p) {
    using namespace rows8;
    constexpr bool A16 = XDT == GEMLITE_DT_FP16 || XDT == GEMLITE_DT_BF16;
    constexpr int XB = A16 ? 2 : 1;
    using G = Geo<XB, MT, BUFCAP>;
    constexpr int ROWB = G::ROWB, BPP = G::BPP, NP = G::NP, PPR = G::PPR, RPI = G::RPI, DPI = G::DPI, XBUF = G::XBUF, WAVE_LDS = G::WAVE_LDS;
    constexpr int NWM = 4, NS = 2;  // weight requests per chunk; chunks the weight requests run ahead
    typedef typename std::conditional<XDT == GEMLITE_DT_FP16, half_tag, bf16_tag>::type Tag;
    typedef typename std::conditional<XDT == GEMLITE_DT_INT8, i32x4, f32x4>::type acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, kb = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    const int mbase = (int)blockIdx.y * (16 * MT);
    int M = p.M - mbase;
    if (M > 16 * MT) M = 16 * MT;
    const int nch_total = p.K / 256;
    const int nchunks = (nch_total - wave + NW - 1) / NW;
    const int npieces = nchunks * NP;
    unsigned char* wl = smem + (size_t)wave * WAVE_LDS;

    // ---- x pieces
    const uint32_t sxmB = (uint32_t)p.stride_xm * (uint32_t)XB;
    const char* xb = (const char*)p.x + (size_t)mbase * sxmB;
    uint32_t xvo[DPI];
#pragma unroll
    for (int q = 0; q < DPI; ++q) {
        const int r = q * RPI + lane / PPR, pp = lane % PPR;
        xvo[q] = r < M ? (uint32_t)r * sxmB + (uint32_t)((pp ^ fswz<PPR, A16>(r & 15)) * 16) : 0x80000000u;  // rows >= M: beyond num_records -> zeros
    }
    const async::srd_t rsX = async::make_srd(xb, (uint32_t)(M - 1) * sxmB + (uint32_t)p.K * (uint32_t)XB);
    const uint32_t xlds = async::lds_addr_of(wl);
    // A fragment of (tile t, 64-k block bq of the piece, half e): 16 bytes at xbase[t] ^ (slot bits of (bq, e)) — the lane's own slot bits
    // (2 kb for 16-bit x, kb for 8-bit x) are XOR-ed with the row's swizzle once, here
    uint32_t xbase[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) xbase[t] = (uint32_t)((t * 16 + j) * ROWB + ((((A16 ? 2 : 1) * kb) ^ fswz<PPR, A16>(j)) << 4));
    auto issue_x = [&](int i, int par) {
        const uint32_t koff = (uint32_t)((((i / NP) * NW + wave) * 256) * XB + (i % NP) * ROWB);
#pragma unroll
        for (int q = 0; q < DPI; ++q) async::req_lds16(rsX, xlds + (uint32_t)(par * XBUF + q * 1024), xvo[q] + koff, 0u);
    };

    // ---- weights: chunk c of this wave = global chunk wave + 8 c; register w[b] = bytes k = 64 b + 16 kb .. + 15 of column n0 + j
    const char* wb = (const char*)p.w;
    const uint32_t wvo = (uint32_t)((n0 + j) * p.stride_wn + wave * 256 + kb * 16);
    struct WSet { u32x4 w[4]; };
    auto issue_w = [&](WSet& S, int ch) {
        const uint32_t wo = wvo + (uint32_t)ch * (uint32_t)(NW * 256);
        gld128_nt<0>(S.w[0], wb, wo);
        gld128_nt<64>(S.w[1], wb, wo);
        gld128_nt<128>(S.w[2], wb, wo);
        gld128_nt<192>(S.w[3], wb, wo);
    };
    auto wait_newer = [&](bool w_behind, bool x_behind) {
        if (w_behind) { if (x_behind) wait_vm<NWM + DPI>(); else wait_vm<NWM>(); }
        else          { if (x_behind) wait_vm<DPI>(); else wait_vm<0>(); }
    };

    acc_t acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (acc_t)(0);

    // Request queue of the wave, oldest first:  W(0) X(0) W(1) X(1) | end of piece i:  X(i + 2)  [W(c + 2) if i closed chunk c]
    // (gemm_wn_rows.hip has the reasoning: x first, a piece must not wait behind the HBM round trip of the weights requested with it)
    auto chunk = [&](WSet& S, int ch, auto cpar) {
        constexpr int CPAR = decltype(cpar)::value;
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
            const int i = ch * NP + pi;
            const int par = ((NP & 1) ? CPAR : 0) ^ (pi & 1);
            wait_newer(pi == 0 && ch + 1 < nchunks, i + 1 < npieces);
            if (pi == 0) {
#pragma unroll
                for (int b = 0; b < 4; ++b) tie4(S.w[b]);
            }
            const unsigned char* xbuf = wl + par * XBUF;
#pragma unroll
            for (int bq = 0; bq < BPP; ++bq) {
                const u32x4 w = S.w[pi * BPP + bq];
                if constexpr (A16) {
                    const u32x4 b0 = w8_to_frag<Tag, WDT>(w[0], w[1]), b1 = w8_to_frag<Tag, WDT>(w[2], w[3]);
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const u32x4 a0 = *(const u32x4*)(xbuf + (xbase[t] ^ (uint32_t)((8 * bq) << 4)));
                        const u32x4 a1 = *(const u32x4*)(xbuf + (xbase[t] ^ (uint32_t)((8 * bq + 1) << 4)));
                        if constexpr (XDT == GEMLITE_DT_FP16) {
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a0), __builtin_bit_cast(h8_t, b0), acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a1), __builtin_bit_cast(h8_t, b1), acc[t], 0, 0, 0);
                        } else {
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_t, a0), __builtin_bit_cast(b8_t, b0), acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_t, a1), __builtin_bit_cast(b8_t, b1), acc[t], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        const u32x4 a = *(const u32x4*)(xbuf + (xbase[t] ^ (uint32_t)((4 * bq) << 4)));
                        if constexpr (XDT == GEMLITE_DT_INT8) {
                            acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, w), acc[t], 0, 0, 0);
                        } else {  // the 16 bytes as two 8-byte k-blocks (the same split on both operands)
                            const long a0 = (long)(((uint64_t)a[1] << 32) | a[0]), a1 = (long)(((uint64_t)a[3] << 32) | a[2]);
                            const long b0 = (long)(((uint64_t)w[1] << 32) | w[0]), b1 = (long)(((uint64_t)w[3] << 32) | w[2]);
                            if constexpr (XDT == GEMLITE_DT_FP8E4) {
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, acc[t], 0, 0, 0);
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, acc[t], 0, 0, 0);
                            } else {
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(a0, b0, acc[t], 0, 0, 0);
                                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(a1, b1, acc[t], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            // the buffer is free once this wave's reads of it have returned; then the requests two pieces / two chunks ahead
            wait_lgkm0();
            if (i + 2 < npieces) issue_x(i + 2, par);
            if (pi == NP - 1 && ch + NS < nchunks) issue_w(S, ch + NS);
        }
    };

    WSet W[NS];
    if (nchunks > 0) issue_w(W[0], 0);
    if (npieces > 0) issue_x(0, 0);
    if (nchunks > 1) issue_w(W[1], 1);
    if (npieces > 1) issue_x(1, 1);
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ch += NS) {
        chunk(W[0], ch, std::integral_constant<int, 0>{});
        if (ch + 1 < nchunks) chunk(W[1], ch + 1, std::integral_constant<int, 1>{});
    }

    // ---- the 8 waves (disjoint K) meet in LDS: each wave's fragments at the start of its own region (its DS operations execute in order, its
    //      DMA requests have all landed); C layout of a 16 x 16 MFMA: column lane & 15, rows 4 (lane >> 4) + r
#pragma unroll
    for (int t = 0; t < MT; ++t) *(acc_t*)(wl + (size_t)(t * 64 + lane) * 16) = acc[t];
    __syncthreads();
    for (int u = tid; u < MT * 256; u += NW * 64) {
        const int t = u >> 8, l = u & 63, r = (u >> 6) & 3;
        const int m = 16 * t + 4 * (l >> 4) + r;
        const int64_t n = n0 + (l & 15);
        const size_t off = (size_t)(t * 64 + l) * 16 + (size_t)r * 4;
        float v;
        if constexpr (XDT == GEMLITE_DT_INT8) {
            int sum = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += *(const int*)(smem + (size_t)w * WAVE_LDS + off);
            v = (float)sum;
        } else {
            v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += *(const float*)(smem + (size_t)w * WAVE_LDS + off);
        }
        if (m < M) {
            if constexpr (A16) v *= (p.w_mode == 2) ? load_as_float(p.scales, n, p.meta_dt) : 1.f;  // per-channel pre-scale: once, on the sum
            epilogue_store(p.epi, v, mbase + m, n);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side: called by plan_a8w8_rows / plan_a16w8_rows (gemm_a8w8.hip) after THEIR checks of dtypes, strides and alignment; these add the
// kernel's own (whole 256-k chunks, 32-bit offsets with the padded rows' 0x80000000 beyond the descriptor) and pick the instantiation.
// ---------------------------------------------------------------------------------------------------------------------
typedef void (*rows8_fn)(const GenericParams);

template <int XDT, int WDT>
static rows8_fn rows8_pick_mt(int mt, bool two) {
    if (two) {  // 64 KB of LDS per block
        if (mt == 1) return w8_rows_lds_kernel<XDT, WDT, 1, 4096>;
        if (mt == 2) return w8_rows_lds_kernel<XDT, WDT, 2, 4096>;
        return nullptr;
    }
    switch (mt) {
        case 1: return w8_rows_lds_kernel<XDT, WDT, 1>;
        case 2: return w8_rows_lds_kernel<XDT, WDT, 2>;
        case 3: return w8_rows_lds_kernel<XDT, WDT, 3>;
        case 4: return w8_rows_lds_kernel<XDT, WDT, 4>;
        default: return nullptr;
    }
}
template <int XDT>
static rows8_fn rows8_pick_w(int wdt, int mt, bool two) {
    switch (wdt) {
        case GEMLITE_DT_INT8: return rows8_pick_mt<XDT, GEMLITE_DT_INT8>(mt, two);
        case GEMLITE_DT_FP8E4: return rows8_pick_mt<XDT, GEMLITE_DT_FP8E4>(mt, two);
        case GEMLITE_DT_FP8E5: return rows8_pick_mt<XDT, GEMLITE_DT_FP8E5>(mt, two);
        default: return nullptr;
    }
}

static size_t rows8_lds(bool a16, int mt, bool two) {
    using namespace rows8;
    int w;
    if (two) w = a16 ? (mt == 1 ? Geo<2, 1, 4096>::WAVE_LDS : Geo<2, 2, 4096>::WAVE_LDS) : (mt == 1 ? Geo<1, 1, 4096>::WAVE_LDS : Geo<1, 2, 4096>::WAVE_LDS);
    else w = a16 ? (mt == 1 ? Geo<2, 1>::WAVE_LDS : (mt == 2 ? Geo<2, 2>::WAVE_LDS : (mt == 3 ? Geo<2, 3>::WAVE_LDS : Geo<2, 4>::WAVE_LDS)))
                 : (mt == 1 ? Geo<1, 1>::WAVE_LDS : (mt == 2 ? Geo<1, 2>::WAVE_LDS : (mt == 3 ? Geo<1, 3>::WAVE_LDS : Geo<1, 4>::WAVE_LDS)));
    return (size_t)w * NW;
}

// mt = row tiles per block (more than 16 mt rows: blocks along grid.y); two = the 64-KB form (two blocks per CU; mt <= 2)
bool plan_w8_rows_lds(const gemlite_hip_forward_args& a, LaunchPlan& lp, int mt, bool two) {
    const bool a16 = a.input_dtype == GEMLITE_DT_FP16 || a.input_dtype == GEMLITE_DT_BF16;
    if (!a16 && a.input_dtype != a.w_dtype) return false;
    if (mt < 1 || mt > 4 || a.K % 256 != 0 || a.N % 16 != 0 || a.M < 1) return false;
    const int xb = a16 ? 2 : 1;
    if ((a.stride_xm * xb) % 16 != 0 || a.stride_wn % 16 != 0 || a.stride_xk != 1 || a.stride_wk != 1) return false;
    if (((uintptr_t)a.x | (uintptr_t)a.w_q) % 16 != 0) return false;
    // 32-bit byte offsets; a row block's x extent below 2^31 (the padded rows' offset 0x80000000 must lie beyond it)
    if (((int64_t)(16 * mt) * a.stride_xm + a.K) * xb >= (1ll << 31) || (int64_t)a.N * a.stride_wn + a.K >= (1ll << 31)) return false;
    const int64_t gy = (a.M + 16 * mt - 1) / (16 * mt);
    if (gy > 65535) return false;
    rows8_fn fn = nullptr;
    switch (a.input_dtype) {
        case GEMLITE_DT_FP16: fn = rows8_pick_w<GEMLITE_DT_FP16>(a.w_dtype, mt, two); break;
        case GEMLITE_DT_BF16: fn = rows8_pick_w<GEMLITE_DT_BF16>(a.w_dtype, mt, two); break;
        case GEMLITE_DT_INT8: fn = rows8_pick_mt<GEMLITE_DT_INT8, GEMLITE_DT_INT8>(mt, two); break;
        case GEMLITE_DT_FP8E4: fn = rows8_pick_mt<GEMLITE_DT_FP8E4, GEMLITE_DT_FP8E4>(mt, two); break;
        case GEMLITE_DT_FP8E5: fn = rows8_pick_mt<GEMLITE_DT_FP8E5, GEMLITE_DT_FP8E5>(mt, two); break;
        default: return false;
    }
    if (!fn) return false;
    static const char* names16[4] = {"a16w8_rows_lds_kernel<16x16>", "a16w8_rows_lds_kernel<32x16>", "a16w8_rows_lds_kernel<48x16>", "a16w8_rows_lds_kernel<64x16>"};
    static const char* names8[4] = {"a8w8_rows_lds_kernel<16x16>", "a8w8_rows_lds_kernel<32x16>", "a8w8_rows_lds_kernel<48x16>", "a8w8_rows_lds_kernel<64x16>"};
    lp.fn = (const void*)fn;
    static const char* names16h[2] = {"a16w8_rows_lds_kernel<16x16,2/cu>", "a16w8_rows_lds_kernel<32x16,2/cu>"};
    static const char* names8h[2] = {"a8w8_rows_lds_kernel<16x16,2/cu>", "a8w8_rows_lds_kernel<32x16,2/cu>"};
    lp.name = two ? (a16 ? names16h[mt - 1] : names8h[mt - 1]) : (a16 ? names16[mt - 1] : names8[mt - 1]);
    lp.grid = dim3((unsigned)(a.N / 16), (unsigned)gy, 1);
    lp.block = dim3(rows8::NW * 64, 1, 1);
    lp.lds_bytes = rows8_lds(a16, mt, two);
    lp.ws_bytes = 0;
    lp.slab_bytes = 0;
    return true;
}

}  // namespace gl
