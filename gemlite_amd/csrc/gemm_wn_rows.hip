// gemm_wn_rows.hip — 2 .. 64 activation rows x packed 4-bit (and, late in the round, 2-bit) weights (round 5): the batched-decode regime,
// shaped like the M = 1 decode kernel of gemv_decode.hip instead of like a GEMM tile.  Template forms: MT = 1 .. 4 row tiles of 16, SPG = group
// size class (128+ / 64 / 32), NT = 1 / 2 column tiles per block (4096 < N <= 8192), BITS = 4 / 2; epilogue channel scales in the 16-bit
// type or fp32 (BitNet).  Experiments that did NOT ship are in profiles/r05/probe_rows5_*_slower.log.
//
// Replaces gemm_splitK_INT_kernel (gemlite/triton_kernels/gemm_splitK_kernels.py:277-450) for the decode-batch sizes where
// rounds 2-4 paid either a cross-block K-slice combine (gemm_wn_direct.hip: 32- / 64-column tiles x 2 slices, 7.4 us at 4096^2 M = 16)
// or an LDS-staged 64 x 64 tile (gemm_wn_mma_kernel.inc: 12.3 / 12.9 us at M = 32 / 64) for a weight stream that the M = 1 kernel
// finishes in 4.7 us.  Same numerics as gemm_wn_direct.hip (raw integer codes through the matrix core, scale / zero applied once per
// quantisation group to the fp32 accumulators, the group sums of x from a second MFMA against a constant B fragment).
//
// Shape of a launch: N / 16 blocks (256 at N = 4096: one per CU, K is NEVER split across blocks), 8 waves; wave w owns the 256-k chunks
// w, w + 8, ... of K and all MT row tiles of 16 rows.
//   * weights: requested FIRST, two chunks ahead, exactly like the decode kernel — lane (g = lane >> 2, c = lane & 3) asks for packed rows
//     2g, 2g + 1 of the chunk x columns 4c .. 4c + 3 as two 16-byte non-temporal loads (64-byte row segments; tiles 2p, 2p + 1 share an
//     XCD).  The matrix core wants lane (j = lane & 15, kb = lane >> 4) to hold the word of column j, packed row 4 s + kb for k-step s:
//     the wave turns its 2 KB around through a PRIVATE LDS slot (two ds_write_b128, eight ds_read_b32, conflict-free, no barrier: DS
//     operations of one wave execute in order).  Dword loads in MFMA layout (gemv_mfma.hip, V = 1) cost 4x the memory instructions.
//   * x: every block reads all of x (M K 2 bytes through the CU's L2 -> L1 path), so HOW it is requested decides the kernel.  The first
//     version loaded the A fragment of lane (j, kb) — 16 bytes of row j — straight into registers: one wave-level request = 16 rows x 64
//     bytes = sixteen HALF cache lines, and measured 37 GB/s per CU (4096^2: M = 32 11.3 us, M = 64 18.4).  A timing experiment with the
//     same bytes requested as 8 rows x 128 bytes per instruction (wrong results) ran M = 32 in 7.6 and M = 64 in 10.4 us
//     (profiles/r05/probe_rows5_xline.log).  The MFMA operand layout cannot give that from registers (a request covers 16 rows whatever
//     the k order), so x goes through LDS: LDS-DMA pieces of 1 KiB = whole 128-byte lines (2 / 4 / 8 rows per instruction), two
//     8-KiB buffers per wave (16 MT rows x 256 / 128 / 64 k), the 16-byte slot of (row r, piece p) XOR-swizzled with f(r) through the
//     SOURCE address so that the ds_read_b128 of an A fragment is conflict-free; piece i + 2 is requested when piece i has been consumed.
//     The B fragment is built in NATURAL k order — t_lo = w & 0x0F0F0F0F, t_hi = (w >> 4) & 0x0F0F0F0F, pair p = v_perm(t_hi, t_lo) |
//     MAGIC2 = (OFF + q_2p, OFF + q_2p+1), 11 VALU per word — so the A fragment is x as it lies in memory: nothing is permuted or
//     pre-scaled per row tile.  OFF = 1024 (fp16) / 128 (bf16): exact, and OFF * sum(x) leaves with the zero-point term.
//   * scale / zero: one 2-byte load each per lane and FOUR groups (lane (j, kb) fetches group 4 l + kb, column j), distributed with
//     ds_bpermute.
//   * the 8 partial tiles meet in LDS (the wave's own buffer again), one barrier, packed 4-byte stores.
// All memory requests are inline asm retired by hand-counted waits (gl_async.h has the reasoning; scripts/isa_asmloads.py audits the
// generated code).  Scalar kernel arguments, the first 14 dwords preloaded into SGPRs (-amdgpu-kernarg-preload-count, see gemv_decode.hip).
#include <type_traits>

#include "gl_common.h"
#include "gl_async.h"

namespace gl {

namespace rows5 {

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

// modes: bits 0..3 W_group_mode | 4 zero_is_scalar | 5 pair the half-line tiles on one XCD | 7 channel scales (sp) in the epilogue | 8..15 log2(group)
constexpr uint32_t M_ZSCALAR = 16u, M_PAIR = 32u, M_POST32 = 64u, M_POST = 128u;  // (M_POST32: the epilogue's channel scales are fp32 — BitNet A16W158)
constexpr int NW = 8;                              // waves per block
constexpr int WSLOT_I1 = 272;                      // dword offset of the odd packed rows inside a wave's weight slot (16 dwords of padding: see below)
constexpr int WSLOT_BYTES = 2304;                  // >= (WSLOT_I1 + 256) * 4
constexpr int XBUF = 8192;                         // one x piece: 16 MT rows x 256 / 128 / 64 k
constexpr int WAVE_LDS = 2 * XBUF + WSLOT_BYTES;   // per wave: two x buffers + the weight slot

__device__ __forceinline__ void gld128_nt(u32x4& dst, const char* base, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void gld16(uint32_t& dst, const char* base, uint32_t voff) {
    asm volatile("global_load_ushort %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void tie4(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void tie1(uint32_t& v) { asm volatile("" : "+v"(v)); }

}  // namespace rows5

// MT = row tiles of 16 (M <= 16 MT); SPG = 32-k steps per quantisation group (4: groups of >= 128, 2: 64, 1: 32); NT = 16-column tiles
// per block (2: layers with 4096 < N <= 8192 — N / 32 blocks stay resident in ONE round, and an x piece read from LDS feeds both tiles)
// BITS = 4 or 2 (late round 5: 2-bit words — A16W2, BitNet A16W158 — through the same kernel): a packed row holds E = 32 / BITS k-values, a chunk
// of 32 packed rows is 256 / 512 k = 8 / 16 MFMA k-steps.  2-bit: lane (j, kb) of k-step s needs k = 32 s + 8 kb .. + 7 = HALF (kb & 1) of the word
// of packed row 2 s + (kb >> 1): sixteen ds_read_b32 per chunk instead of eight (two lanes share an address: broadcast), the 16-bit half is
// spread to eight nibbles (three shift-or + and pairs) and then treated like a 4-bit word — natural k order again, so groups of 32 work
template <typename Tag, int MT, int SPG, int NT = 1, int BITS = 4>
__global__ __launch_bounds__(rows5::NW * 64, 1) void gemm_w4_rows_kernel(const char* wb, const char* xb, const char* sp, const char* zp, uint16_t* out,
                                                                          uint32_t sw4, uint32_t mstride2, int nch_total, uint32_t modes,
                                                                          int M, uint32_t sxm2, uint32_t som) {
    using namespace rows5;
    using TR = F16Traits<Tag>;
    constexpr int CHUNK = 32, TC = 16, TCN = TC * NT, CSTRIDE = NW * CHUNK;  // packed rows per chunk (256 k), tile columns, block columns
    constexpr int E = 32 / BITS, KS = E;                      // k-values per packed row; MFMA k-steps (32 k) per chunk
    constexpr int KCH = 32 * E;                               // k per chunk: 256 (4-bit) / 512 (2-bit)
    constexpr int NG = KS / SPG;                              // quantisation groups per chunk (group sizes above the chunk repeat their row)
    constexpr int NML = (NG + 3) / 4;                         // metadata loads per chunk and kind (scales / zeros), each 4 groups x 16 columns
    constexpr int XK = MT == 1 ? 256 : (MT == 2 ? 128 : 64);  // k per x piece
    constexpr int NP = KCH / XK;                              // x pieces per chunk
    constexpr int SPP = XK / 32;                              // MFMA k-steps per piece
    constexpr int PPR = XK / 8;                               // 16-byte slots per row of a piece
    constexpr int RPI = 64 / PPR;                             // rows per LDS-DMA instruction (1 KiB)
    constexpr int DPI = 16 * MT / RPI;                        // LDS-DMA instructions per piece
    constexpr int NWM = NT * (2 + 2 * NML);                   // requests of one chunk's weights + metadata
    constexpr int NS = 2;                                     // weight register sets = chunks the weight requests run ahead.  (4 sets — 64 KB per CU in
                                                              // flight like the decode kernel — measured SLOWER: a wave's requests return in order, so every
                                                              // x piece then waits behind more HBM round trips; 4096 x 8192 M = 8: 9.4 -> 10.7 us)
    static_assert((BITS == 4 || BITS == 2) && NG >= 2 && DPI * 1024 <= XBUF && (WSLOT_I1 + 256) * 4 <= WSLOT_BYTES && MT * NT * 1024 <= XBUF, "LDS layout");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x;
    if (NT == 1 && (modes & M_PAIR)) {  // adjacent half-line tiles on one XCD (speed only; any mapping is correct)
        const int xcd = tile & 7, idx = tile >> 3;
        tile = (((idx >> 1) << 3) + xcd) * 2 + (idx & 1);
    }
    // more than 16 MT rows (shapes no tile kernel takes — groups of 32, N % 64 != 0 — at prefill sizes): blocks along grid.y, 16 MT rows each
    if (gridDim.y > 1) {
        const uint32_t m0 = blockIdx.y * (uint32_t)(16 * MT);
        xb += (size_t)m0 * sxm2;
        out += (size_t)m0 * som;
        M -= (int)m0;
    }
    if (M > 16 * MT) M = 16 * MT;
    const int c = lane & 3, g = lane >> 2;    // role in the weight request
    const int j = lane & 15, kb = lane >> 4;  // role in the MFMA: column j / row j of a tile, 8-k block kb of the 32-k step
    const int nchunks = (nch_total - wave + NW - 1) / NW;
    const int npieces = nchunks * NP;
    const int w_mode = (int)(modes & 15u), gs_shift = (int)((modes >> 8) & 255u);
    const bool need_s = w_mode >= 2, need_z = (w_mode == 1 || w_mode >= 3) && !(modes & M_ZSCALAR);

    unsigned char* wl = smem + (size_t)wave * WAVE_LDS;  // this wave's LDS: [x buffer 0][x buffer 1][weight slot]
    uint32_t* wslot = (uint32_t*)(wl + 2 * XBUF);
    // weight slot, write side: packed row 2g + i of the chunk at dword (i ? WSLOT_I1 : 0) + 16 g + 4 c (a fixed i makes 8 lanes = 128 contiguous
    // bytes); read side: row 4 s + kb = 2 (2 s + (kb >> 1)) + (kb & 1) -> dword (kb & 1) WSLOT_I1 + 32 s + 16 (kb >> 1) + j: the 32 lanes of a
    // ds_read_b32 half (kb = 0, 1 or 2, 3) land on banks j and 16 + j
    const int wr_off = g * 16 + c * 4;
    const int rd_off = BITS == 4 ? (kb & 1) * WSLOT_I1 + (kb >> 1) * 16 + j   // row 4 s + kb: + 32 s
                                 : (kb >> 1) * WSLOT_I1 + j;                  // 2-bit: row 2 s + (kb >> 1): + 16 s (kb = 2 m, 2 m + 1 read the same dword)
    const uint32_t half_sh = (uint32_t)(kb & 1) * 16u;                        // 2-bit: which half of that word
    const uint32_t wo0 = (uint32_t)(wave * CHUNK + g * 2) * sw4 + (uint32_t)(tile * TCN + c * 4) * 4u;  // (+ 64 bytes per further column tile)

    // ---- x pieces: LDS slot (row r, 16-byte slot p') of a buffer holds piece p = p' ^ f(r) of the row; DMA instruction q fills rows
    //      q RPI .. q RPI + RPI - 1 lane-linearly.  f: the low bits of r that separate the rows one ds_read_b128 lane group touches
    auto fswz = [](int r) { return PPR >= 16 ? (r & 15) : ((r >> 1) & 7); };
    uint32_t xvo[DPI];  // per-lane source byte offset of DMA instruction q (k offset of the piece added per request); rows >= M: out of range -> zeros
#pragma unroll
    for (int q = 0; q < DPI; ++q) {
        const int r = q * RPI + lane / PPR, pp = lane % PPR;
        xvo[q] = r < M ? (uint32_t)r * sxm2 + (uint32_t)((pp ^ fswz(r)) * 16) : 0x80000000u;
    }
    const async::srd_t rsX = async::make_srd(xb, (uint32_t)(M - 1) * sxm2 + (uint32_t)nch_total * (uint32_t)(KCH * 2));
    const uint32_t xlds = async::lds_addr_of(wl);
    uint32_t abase[MT];  // byte offset of this lane's A fragment (k-step 0 of a piece) inside a buffer; k-step s' = abase ^ (s' << 6)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int r = t * 16 + j;
        abase[t] = (uint32_t)(r * PPR * 16 + ((kb ^ fswz(r)) << 4));
    }
    // piece i of the wave = (chunk i / NP, part i % NP): its first k as a byte offset inside a row of x
    auto issue_x = [&](int i, int par) {
        const uint32_t koff = (uint32_t)(((i / NP) * CSTRIDE + wave * CHUNK) * E + (i % NP) * XK) * 2u;
#pragma unroll
        for (int q = 0; q < DPI; ++q) async::req_lds16(rsX, xlds + (uint32_t)(par * XBUF + q * 1024), xvo[q] + koff, 0u);
    };

    // ---- weights + metadata of a chunk: lane (j, kb) loads the scale and the zero of group 4 l + kb of the chunk, column j (uniform base +
    //      32-bit lane offset: no 64-bit address arithmetic between the requests; absent metadata is still "loaded" — from the weight buffer,
    //      always in bounds — so the loop stays branch-free)
    const char* sbase = need_s ? sp : wb;
    const char* zbase = need_z ? zp : wb;
    const uint32_t mcol = (uint32_t)(tile * TCN + j) * 2u;  // (+ 32 bytes per further column tile)
    struct WSet { u32x4 w0[NT], w1[NT]; uint32_t s[NT][NML], z[NT][NML]; };
    auto issue_w = [&](WSet& S, int ch) {
        const uint32_t wo = wo0 + (uint32_t)(ch * CSTRIDE) * sw4;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            gld128_nt(S.w0[n], wb, wo + (uint32_t)(n * 64));
            gld128_nt(S.w1[n], wb, wo + sw4 + (uint32_t)(n * 64));
        }
        const uint32_t k0 = (uint32_t)(ch * CSTRIDE + wave * CHUNK) * (uint32_t)E;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int l = 0; l < NML; ++l) {
                const int gq = 4 * l + kb < NG ? 4 * l + kb : NG - 1;
                const uint32_t mo = ((k0 + (uint32_t)(gq * 32 * SPG)) >> gs_shift) * mstride2 + mcol + (uint32_t)(n * 32);
                gld16(S.s[n][l], sbase, need_s ? mo : 0u);
                gld16(S.z[n][l], zbase, need_z ? mo : 0u);
            }
    };
    // everything but the newest `newer` requests of this wave has landed; newer is wave-uniform and one of four values
    auto wait_newer = [&](bool w_behind, bool x_behind) {
        if (w_behind) { if (x_behind) wait_vm<NWM + DPI>(); else wait_vm<NWM>(); }
        else          { if (x_behind) wait_vm<DPI>(); else wait_vm<0>(); }
    };

    f32x4 tot[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) tot[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float scalar_zero = (modes & M_ZSCALAR) ? (float)((const int32_t*)zp)[0] : 0.f;
    const float bz = (w_mode == 1 || w_mode == 3) ? -1.f : (w_mode == 4 ? 1.f : 0.f);
    const bool b_times_s = w_mode == 3;
    const u32x4 onesb = {TR::ONES2, TR::ONES2, TR::ONES2, TR::ONES2};

    // One chunk.  CPAR = chunk parity (which of the two register sets and — for one piece per chunk — which x buffer).  Request queue of the
    // wave, oldest first:
    //     W(0) X(0) W(1) X(1) | end of piece i:  X(i + 2)  [W(c + 2) if i closed chunk c]            (each only if it exists)
    // so behind X(i) sit [X(i + 1)] and [W(c + 1) if i opens chunk c] — x first: a wave's requests return in order, and a piece
    // must not wait behind the HBM round trip of the weights requested with it (4096 x 11008, M = 8: 12.7 -> 11.4 us).
    auto chunk = [&](WSet& S, int ch, auto cpar) {
        constexpr int CPAR = decltype(cpar)::value;
        uint32_t bw[NT][KS];
        f32x4 acc[MT][NT], ones[MT];
#pragma unroll
        for (int pi = 0; pi < NP; ++pi) {
            const int i = ch * NP + pi;
            const int par = ((NP & 1) ? CPAR : 0) ^ (pi & 1);
            wait_newer(pi == 0 && ch + 1 < nchunks, i + 1 < npieces);
            if (pi == 0) {  // the wave's 2 KB of weights: registers -> own LDS slot -> MFMA layout
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    tie4(S.w0[n]);
                    tie4(S.w1[n]);
#pragma unroll
                    for (int l = 0; l < NML; ++l) {
                        tie1(S.s[n][l]);
                        tie1(S.z[n][l]);
                    }
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {  // (one slot, tile after tile: the DS operations of a wave execute in order)
                    *(u32x4*)(wslot + wr_off) = S.w0[n];
                    *(u32x4*)(wslot + WSLOT_I1 + wr_off) = S.w1[n];
#pragma unroll
                    for (int s = 0; s < KS; ++s) bw[n][s] = wslot[rd_off + s * (BITS == 4 ? 32 : 16)];
                }
            }
            const unsigned char* xbuf = wl + par * XBUF;
#pragma unroll
            for (int sq = 0; sq < SPP; ++sq) {
                const int s = pi * SPP + sq;
                u32x4 bf[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    uint32_t wv = bw[n][s];
                    if constexpr (BITS == 2) {  // this lane's 16-bit half -> eight nibbles (q0 .. q7 in the low two bits of each)
                        wv = (wv >> half_sh) & 0xFFFFu;
                        wv = (wv | (wv << 8)) & 0x00FF00FFu;
                        wv = (wv | (wv << 4)) & 0x0F0F0F0Fu;
                        wv = (wv | (wv << 2)) & 0x33333333u;
                    }
                    const uint32_t t_lo = wv & 0x0F0F0F0Fu, t_hi = (wv >> 4) & 0x0F0F0F0Fu;
#pragma unroll
                    for (int pq = 0; pq < 4; ++pq) bf[n][pq] = __builtin_amdgcn_perm(t_hi, t_lo, 0x0C040C00u + (uint32_t)pq * 0x00010001u) | TR::MAGIC2;
                }
                const bool first = s % SPG == 0;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const u32x4 a = *(const u32x4*)(xbuf + (abase[t] ^ (uint32_t)(sq << 6)));
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[t][n] = mfma16<Tag>(a, bf[n], first ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[t][n]);
                    ones[t] = mfma16<Tag>(a, onesb, first ? (f32x4){0.f, 0.f, 0.f, 0.f} : ones[t]);
                }
                if ((s + 1) % SPG == 0) {  // end of a quantisation group: fold scale / zero into the totals
                    const int q = s / SPG, l = q >> 2, gq = q & 3;  // group q of the chunk: loaded by the lanes kb = gq of load l
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const uint32_t sraw = (uint32_t)__builtin_amdgcn_ds_bpermute((j + 16 * gq) * 4, (int)S.s[n][l]);
                        const uint32_t zraw = (uint32_t)__builtin_amdgcn_ds_bpermute((j + 16 * gq) * 4, (int)S.z[n][l]);
                        const float sv = need_s ? TR::to_float((uint16_t)sraw) : 1.f;
                        const float zv = need_z ? TR::to_float((uint16_t)zraw) : scalar_zero;
                        const float a = sv;
                        const float b = bz * zv * (b_times_s ? sv : 1.f) - a * TR::OFF;
#pragma unroll
                        for (int t = 0; t < MT; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) tot[t][n][r] += a * acc[t][n][r] + b * ones[t][r];
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
        if constexpr (NS == 4) {
            if (ch + 2 < nchunks) chunk(W[2], ch + 2, std::integral_constant<int, 0>{});
            if (ch + 3 < nchunks) chunk(W[3], ch + 3, std::integral_constant<int, 1>{});
        }
    }

    // ---- the 8 waves (disjoint K) meet in LDS: [MT x 16 rows][16 NT columns] fp32 at the start of each wave's region; C layout: column j,
    //      rows 4 kb + r ------------------------------------------------------------------------------------------------------------------
    float* part = (float*)wl;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(t * 16 + 4 * kb + r) * TCN + n * TC + j] = tot[t][n][r];
    __syncthreads();
    constexpr int CPR = TCN / 2;               // pairs of adjacent columns per row of the block
    constexpr int NPAIR = MT * 16 * CPR;
    for (int o = tid; o < NPAIR; o += NW * 64) {
        const int m = o / CPR, cp = o % CPR;
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float2 pv = *(const float2*)(smem + (size_t)w * WAVE_LDS + (size_t)(m * TCN + cp * 2) * 4);
            v0 += pv.x;
            v1 += pv.y;
        }
        if (modes & M_POST) {  // channel scales (channel_scale_mode 1): the kernel's 16-bit type, or fp32 (the BitNet processors' default)
            if (modes & M_POST32) {
                const float2 sw = *(const float2*)(sp + (size_t)(tile * TCN + cp * 2) * 4);
                v0 *= sw.x;
                v1 *= sw.y;
            } else {
                const uint32_t sw = *(const uint32_t*)(sp + (size_t)(tile * TCN + cp * 2) * 2);
                v0 *= TR::to_float((uint16_t)(sw & 0xFFFFu));
                v1 *= TR::to_float((uint16_t)(sw >> 16));
            }
        }
        if (m < M) *(uint32_t*)(out + (size_t)m * som + (size_t)(tile * TCN + cp * 2)) = (uint32_t)TR::from_float(v0) | ((uint32_t)TR::from_float(v1) << 16);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host-side planning.  tuning[0] = 9 forces this kernel (any M), tuning[3] & 65536 = never (the round-4 choice, A/B runs)
// ---------------------------------------------------------------------------------------------------------------------
typedef void (*rows5_fn)(const char*, const char*, const char*, const char*, uint16_t*, uint32_t, uint32_t, int, uint32_t, int, uint32_t, uint32_t);

template <typename Tag, int MT>
static rows5_fn rows5_pick_spg(int spg, int nt, int bits) {
    if (bits == 2) {  // 2-bit words: one column tile per block
        if (nt != 1) return nullptr;
        switch (spg) {  // (16 unrolled k-steps per chunk: groups of 64 fit 256 registers up to 32 rows, groups of 32 up to 16)
            case 4: return gemm_w4_rows_kernel<Tag, MT, 4, 1, 2>;
            case 2:
                if constexpr (MT <= 2) return gemm_w4_rows_kernel<Tag, MT, 2, 1, 2>;
                else return nullptr;
            case 1:
                if constexpr (MT <= 1) return gemm_w4_rows_kernel<Tag, MT, 1, 1, 2>;
                else return nullptr;
            default: return nullptr;
        }
    }
    if (nt == 2) {  // two column tiles per block: groups of >= 128 up to 64 rows, groups of 64 up to 32 rows (registers)
        if (spg == 4) return gemm_w4_rows_kernel<Tag, MT, 4, 2>;
        if constexpr (MT <= 2) {
            if (spg == 2) return gemm_w4_rows_kernel<Tag, MT, 2, 2>;
        }
        return nullptr;
    }
    switch (spg) {
        case 4: return gemm_w4_rows_kernel<Tag, MT, 4>;
        case 2: return gemm_w4_rows_kernel<Tag, MT, 2>;
        case 1:  // groups of 32: up to 32 rows per block (the 48- / 64-row forms need more than 256 registers); more rows go along grid.y
            if constexpr (MT <= 2) return gemm_w4_rows_kernel<Tag, MT, 1>;
            else return nullptr;
        default: return nullptr;
    }
}
template <typename Tag>
static rows5_fn rows5_pick(int mt, int spg, int nt, int bits) {
    switch (mt) {
        case 1: return rows5_pick_spg<Tag, 1>(spg, nt, bits);
        case 2: return rows5_pick_spg<Tag, 2>(spg, nt, bits);
        case 3: return rows5_pick_spg<Tag, 3>(spg, nt, bits);
        case 4: return rows5_pick_spg<Tag, 4>(spg, nt, bits);
        default: return nullptr;
    }
}

bool plan_gemm_wn_rows(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    if ((a.W_nbits != 4 && a.W_nbits != 2) || a.w_pack_bits != 32) return false;
    const int bits = a.W_nbits, epr = 32 / bits;  // k-values per packed row
    if (a.M < 1 || a.M > 65535 * 64) return false;  // above 64 rows: 64-row blocks along grid.y (the weights stream once per block row)
    if (a.input_dtype != GEMLITE_DT_FP16 && a.input_dtype != GEMLITE_DT_BF16) return false;
    if (a.output_dtype != a.input_dtype || a.stride_on != 1 || a.stride_xk != 1 || a.stride_wn != 1) return false;
    // channel scales (mode 1) of the kernel's 16-bit type ride in the epilogue — then nothing in the K loop reads scales (W_group_mode 0 / 1)
    const bool post_s = a.channel_scale_mode == 1;
    if (a.channel_scale_mode != 0 && !post_s) return false;
    const bool post32 = post_s && a.meta_dtype == GEMLITE_DT_FP32;
    if (post_s && (a.W_group_mode >= 2 || (a.meta_dtype != a.input_dtype && !post32) || !a.scales || ((uintptr_t)a.scales % (post32 ? 8 : 4)) != 0)) return false;
    const bool loop_s = a.W_group_mode >= 2, has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (loop_s && a.meta_dtype != a.input_dtype) return false;
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != a.input_dtype) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    if ((a.stride_xm * 2) % 16 != 0 || ((uintptr_t)a.x % 16) != 0) return false;        // 16-byte A fragments
    if (((uintptr_t)a.w_q % 16) != 0 || (a.stride_wk % 4) != 0) return false;            // 16-byte weight loads
    if (((uintptr_t)a.out % 4) != 0 || a.stride_om % 2 != 0) return false;               // packed pairs of outputs
    if (a.N % 16 != 0 || a.K % (32 * epr) != 0) return false;   // whole chunks of 32 packed rows: K % 256 (4-bit) / % 512 (2-bit)
    if (p.gs_shift < 5 || (p.gs_shift > 30 && p.gs_shift != 31)) return false;            // groups of 32, 64, 128, ... k (31: one group spans K)
    const int spg = p.gs_shift >= 7 ? 4 : (p.gs_shift == 6 ? 2 : 1);
    const int64_t rows = a.K / epr;
    // 32-bit byte offsets in the kernel; the x extent of a row block stays below 2^31 so that the 0x80000000 offset of the padded rows (>= M) is
    // beyond the descriptor's num_records and reads zeros
    if (rows * a.stride_wk * 4 + a.N * 4 >= (1ll << 32) || (64 * a.stride_xm + a.K) * 2 >= (1ll << 31) || 64 * a.stride_om * 2 >= (1ll << 32)) return false;
    if (p.gs_shift < 31 && ((a.K >> p.gs_shift) * p.stride_meta_g + a.N) * 2 >= (1ll << 32)) return false;
    // two column tiles per block (tuning[1] = 2 forces, 1 = never): up to 32 rows of a layer whose 16-column tiles do not fit one round
    // of resident blocks but whose 32-column blocks do (4096 < N <= 8192 on 256 CUs)
    const int64_t resident = resident_block_limit();
    int nt = 1;
    if (a.tuning[1] == 2 || (a.tuning[1] == 0 && a.N / 16 > resident && a.N / 32 <= resident)) nt = 2;
    if (nt == 2 && (a.N % 32 != 0 || spg == 1 || bits != 4 || a.M > (spg == 4 ? 64 : 32))) {
        if (a.tuning[1] == 2) return false;
        nt = 1;
    }
    if (a.tuning[1] < 0 || a.tuning[1] > 2) return false;
    // row tiles per block (registers); more rows: blocks along grid.y (the weights stream once per block row)
    const int mt_cap = bits == 4 ? (spg == 1 ? 2 : 4) : (spg == 4 ? 4 : (spg == 2 ? 2 : 1));
    const int mt = a.M > 16 * mt_cap ? mt_cap : (int)((a.M + 15) / 16);
    if ((a.M + 16 * mt - 1) / (16 * mt) > 65535) return false;
    if (a.tuning[2] != 0 && a.tuning[2] != 8) return false;
    const bool f16 = a.input_dtype == GEMLITE_DT_FP16;
    const rows5_fn fn = f16 ? rows5_pick<half_tag>(mt, spg, nt, bits) : rows5_pick<bf16_tag>(mt, spg, nt, bits);
    if (!fn) return false;
    const int nw = rows5::NW;
    const int tiles = (int)(a.N / (16 * nt));
    const bool need_s = loop_s, need_z = has_z && !a.zero_is_scalar;
    p.splitk = 1;
    p.rows_per_slice = (int)rows;
    lp.fn = (const void*)fn;
    static const char* names[4] = {"gemm_w4_rows_kernel<16x16>", "gemm_w4_rows_kernel<32x16>", "gemm_w4_rows_kernel<48x16>", "gemm_w4_rows_kernel<64x16>"};
    static const char* names2[4] = {"gemm_w4_rows_kernel<16x32>", "gemm_w4_rows_kernel<32x32>", "gemm_w4_rows_kernel<48x32>", "gemm_w4_rows_kernel<64x32>"};
    static const char* names_w2[4] = {"gemm_w2_rows_kernel<16x16>", "gemm_w2_rows_kernel<32x16>", "gemm_w2_rows_kernel<48x16>", "gemm_w2_rows_kernel<64x16>"};
    lp.name = bits == 2 ? names_w2[mt - 1] : (nt == 2 ? names2[mt - 1] : names[mt - 1]);
    lp.grid = dim3((unsigned)tiles, (unsigned)((a.M + 16 * mt - 1) / (16 * mt)), 1);
    lp.block = dim3(64 * nw, 1, 1);
    lp.lds_bytes = (size_t)rows5::WAVE_LDS * nw;
    lp.slab_bytes = 0;
    lp.ws_bytes = 0;
    lp.arg_kind = 2;
    lp.r5.w = (const char*)p.w;
    lp.r5.x = (const char*)p.x;
    lp.r5.s = (need_s || post_s) ? (const char*)p.scales : (const char*)p.w;
    lp.r5.z = (need_z || (has_z && a.zero_is_scalar)) ? (const char*)p.zeros : (const char*)p.w;
    lp.r5.out = (uint16_t*)p.epi.out;
    lp.r5.sw4 = (uint32_t)a.stride_wk * 4u;
    lp.r5.mstride2 = ((need_s || need_z) && p.gs_shift < 31) ? (uint32_t)p.stride_meta_g * 2u : 0u;
    lp.r5.nch_total = (int)(rows / 32);
    lp.r5.modes = (uint32_t)a.W_group_mode | ((has_z && a.zero_is_scalar) ? 16u : 0u) | ((nt == 1 && (tiles & 15) == 0) ? 32u : 0u) | (post_s ? 128u : 0u) | (post32 ? 64u : 0u) | ((uint32_t)p.gs_shift << 8);
    lp.r5.M = (int)a.M;
    lp.r5.sxm2 = (uint32_t)a.stride_xm * 2u;
    lp.r5.som = (uint32_t)a.stride_om;
    return true;
}

}  // namespace gl
