// gemm_wn_rows.hip — 2 .. 64 activation rows x packed 4-bit weights (round 5): the batched-decode regime, shaped like the M = 1
// decode kernel of gemv_decode.hip instead of like a GEMM tile.
//
// Replaces gemm_splitK_INT_kernel (gemlite/triton_kernels/gemm_splitK_kernels.py:277-450) for the decode-batch sizes where
// rounds 2-4 paid either a cross-block K-slice combine (gemm_wn_direct.hip: 32- / 64-column tiles x 2 slices, 7.3 us at 4096^2 M = 16)
// or an LDS-staged 64 x 64 tile (gemm_wn_mma_kernel.inc: 12.3 / 12.5 us at M = 32 / 64) for a weight stream that the M = 1 kernel
// finishes in 4.6 us.  Same numerics as gemm_wn_direct.hip (raw integer codes through the matrix core, scale / zero applied once per
// quantisation group to the fp32 accumulators, the group sums of x from a second MFMA against a constant B fragment).
//
// Shape of a launch: N / 16 blocks (256 at N = 4096: one per CU, K is NEVER split across blocks), NW = 16 (8 from 33 rows) waves; wave w
// owns the 256-k chunks w, w + NW, ... of K and all MT row tiles of 16 rows.
//   * weights: requested FIRST, exactly like the decode kernel — lane (g = lane >> 2, c = lane & 3) asks for packed rows 2g, 2g + 1 of
//     the chunk x columns 4c .. 4c + 3 as two 16-byte non-temporal loads (64-byte row segments; tiles 2p, 2p + 1 share an XCD).  The
//     matrix core wants lane (j = lane & 15, kb = lane >> 4) to hold the word of column j, packed row 4 s + kb for k-step s: the wave
//     turns its 2 KB around through a PRIVATE LDS slot (two ds_write_b128, eight ds_read_b32, conflict-free, no barrier: DS operations
//     of one wave execute in order).  Dword loads in MFMA layout (gemv_mfma.hip, V = 1) cost 4x the memory instructions, and the CU's
//     address path — 64 B per clock, ~16 clocks per wave-level instruction whatever its width — is this kernel's limit.
//   * x: the A fragment of lane (j, kb), row tile t, k-step s is the 16 bytes x[16 t + j][k0 + 32 s + 8 kb ..] straight from global
//     memory (L2-resident), no staging, no arithmetic: the B fragment is built in NATURAL k order — t_lo = w & 0x0F0F0F0F, t_hi =
//     (w >> 4) & 0x0F0F0F0F, pair p = v_perm(t_hi, t_lo) | MAGIC2 = (OFF + q_2p, OFF + q_2p+1), 11 VALU per word — so nothing is
//     permuted or pre-scaled per row tile (the direct kernel spends 4 v_perm (+ 2 v_pk_mul) per A fragment).  OFF = 1024 (fp16) / 128
//     (bf16): exact, and OFF * sum(x) leaves with the zero-point term.
//   * every block reads all of x: M K 2 bytes per block through the address path (256 KB at M = 32, K = 4096: 2 us at 64 B/clk) — the
//     planner's budget (plan_gemm_wn_rows) hands larger M N K to the tile kernels.
//   * scale / zero: ONE 2-byte load per lane and pair of groups (lane (j, kb) fetches {scale, zero}[kb & 1] of group 2 l + (kb >> 1),
//     column j), distributed with ds_bpermute.
//   * the NW partial tiles meet in LDS (the wave's own slot again), one barrier, packed 4-byte stores.
// Scalar kernel arguments, the first 14 dwords preloaded into SGPRs (-amdgpu-kernarg-preload-count, see gemv_decode.hip).
#include <type_traits>

#include "gl_common.h"

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

// modes: bits 0..3 W_group_mode | 4 zero_is_scalar | 5 pair the half-line tiles on one XCD | 8..15 log2(group)
constexpr uint32_t M_ZSCALAR = 16u, M_PAIR = 32u;
constexpr int WSLOT_I1 = 272;                 // dword offset of the odd packed rows inside a wave's slot (16 dwords of padding: see below)
constexpr int WSLOT_BYTES = (WSLOT_I1 + 256) * 4;  // 2112

// Memory requests from inline asm, retired by hand with counted waits (gl_async.h has the reasoning; loads return in issue order).  The
// first build left them to hipcc: its scheduler sank fourteen of the sixteen x requests of a chunk BETWEEN the MFMAs to save registers —
// one request in flight per wave, each answered with s_waitcnt vmcnt(0).  scripts/isa_asmloads.py audits the generated code (no
// destination register touched before the wait that covers it).
__device__ __forceinline__ void gld128_nt(u32x4& dst, const char* base, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(base) : "memory");
}
template <int OFF>
__device__ __forceinline__ void gld128(u32x4& dst, const char* base, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(OFF) : "memory");
}
__device__ __forceinline__ void gld16(uint32_t& dst, const char* vaddr) {
    asm volatile("global_load_ushort %0, %1, off" : "=v"(dst) : "v"(vaddr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void tie4(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void tie1(uint32_t& v) { asm volatile("" : "+v"(v)); }

}  // namespace rows5

// MT = row tiles of 16 (M <= 16 MT); SPG = 32-k steps per quantisation group (4: groups of >= 128, 2: 64, 1: 32); NW = waves per block
template <typename Tag, int MT, int SPG, int NW>
__global__ __launch_bounds__(NW * 64, 1) void gemm_w4_rows_kernel(const char* wb, const char* xb, const char* sp, const char* zp, uint16_t* out,
                                                                   uint32_t sw4, uint32_t mstride2, int nch_total, uint32_t modes,
                                                                   int M, uint32_t sxm2, uint32_t som) {
    using namespace rows5;
    using TR = F16Traits<Tag>;
    constexpr int CHUNK = 32, TC = 16, CSTRIDE = NW * CHUNK;  // packed rows per chunk (256 k), tile columns
    constexpr int NG = 8 / SPG;                               // quantisation groups per chunk (group sizes above 256 repeat their row)
    constexpr int NML = NG / 2;                               // metadata loads per chunk (each: 2 groups x {scale, zero} x 16 columns)
    constexpr int SLOT = MT * 1024 > 2304 ? MT * 1024 : 2304; // a wave's LDS slot: its 2 KB of weights, later its MT partial tiles
    static_assert(NG >= 2 && WSLOT_BYTES <= 2304, "slot layout");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x;
    if (modes & M_PAIR) {  // adjacent half-line tiles on one XCD (speed only; any mapping is correct)
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
    const int c = lane & 3, g = lane >> 2;    // role in the weight request
    const int j = lane & 15, kb = lane >> 4;  // role in the MFMA: column j / row j of a tile, 8-k block kb of the 32-k step
    const int nchunks = (nch_total - wave + NW - 1) / NW;
    const int w_mode = (int)(modes & 15u), gs_shift = (int)((modes >> 8) & 255u);
    const bool need_s = w_mode >= 2, need_z = (w_mode == 1 || w_mode >= 3) && !(modes & M_ZSCALAR);

    unsigned char* slot = smem + (size_t)wave * SLOT;
    uint32_t* wslot = (uint32_t*)slot;
    // write side: packed row 2g + i of the chunk at dword (i ? WSLOT_I1 : 0) + 16 g + 4 c (a fixed i makes 8 lanes = 128 contiguous bytes);
    // read side: row 4 s + kb = 2 (2 s + (kb >> 1)) + (kb & 1) -> dword (kb & 1) WSLOT_I1 + 32 s + 16 (kb >> 1) + j: the 32 lanes of a
    // ds_read_b32 half (kb = 0, 1 or 2, 3) land on banks j and 16 + j
    const int wr_off = g * 16 + c * 4;
    const int rd_off = (kb & 1) * WSLOT_I1 + (kb >> 1) * 16 + j;

    const uint32_t wo0 = (uint32_t)(wave * CHUNK + g * 2) * sw4 + (uint32_t)(tile * TC + c * 4) * 4u;
    uint32_t xo[MT];  // byte offset of this lane's A fragment of chunk 0, k-step 0; rows past M repeat the last one (never stored)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int m = t * 16 + j;
        xo[t] = (uint32_t)(m < M ? m : M - 1) * sxm2 + (uint32_t)(wave * CHUNK * 8 + kb * 8) * 2u;
    }
    // metadata: lane (j, kb) loads {scales, zeros}[kb & 1] of group 2 l + (kb >> 1) of the chunk, column j (absent metadata is still
    // "loaded" — from the weight buffer, always in bounds — so the loop stays branch-free)
    const bool meta_z = kb & 1;
    const char* mbase = meta_z ? (need_z ? zp : wb) : (need_s ? sp : wb);
    const bool meta_live = meta_z ? need_z : need_s;
    const uint32_t mcol = meta_live ? (uint32_t)(tile * TC + j) * 2u : 0u;
    const uint32_t mstr = meta_live ? mstride2 : 0u;

    f32x4 tot[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) tot[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float scalar_zero = (modes & M_ZSCALAR) ? (float)((const int32_t*)zp)[0] : 0.f;
    const float bz = (w_mode == 1 || w_mode == 3) ? -1.f : (w_mode == 4 ? 1.f : 0.f);
    const bool b_times_s = w_mode == 3;
    const u32x4 onesb = {TR::ONES2, TR::ONES2, TR::ONES2, TR::ONES2};
    // x of a chunk: 8 k-steps x MT row tiles, 16 bytes each, XS steps per batch (the g32 variants of the 48- / 64-row tiles keep a
    // half / a quarter of a chunk in registers at a time)
    constexpr int XS = (MT >= 3 && SPG == 1) ? (MT == 4 ? 2 : 4) : 8;
    constexpr int XB = XS * MT;  // x requests per batch

    // (one chunk per wave at K = 4096; longer K: the next chunk's requests leave when this one's arithmetic is done)
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
        // ---- requests, in the order the CU's in-order memory path should see them: weights, metadata, x ---------------------------
        u32x4 w0, w1;
        uint32_t mraw[NML];
        u32x4 xf[XS][MT];
        const uint32_t wo = wo0 + (uint32_t)(ch * CSTRIDE) * sw4;
        gld128_nt(w0, wb, wo);
        gld128_nt(w1, wb, wo + sw4);
        const uint32_t k0 = (uint32_t)(ch * CSTRIDE + wave * CHUNK) * 8u;
#pragma unroll
        for (int l = 0; l < NML; ++l) {
            const uint32_t kg = k0 + (uint32_t)((2 * l + (kb >> 1)) * 32 * SPG);
            gld16(mraw[l], mbase + ((kg >> gs_shift) * mstr + mcol));
        }
        uint32_t xv[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) xv[t] = xo[t] + (uint32_t)(ch * CSTRIDE) * 16u;
        auto request_x = [&](auto s0c) {
            constexpr int s0 = decltype(s0c)::value;
#pragma unroll
            for (int s = 0; s < XS; ++s)
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    // (immediate offsets: one template instantiation per k-step)
                    switch (s0 + s) {
                        case 0: gld128<0>(xf[s][t], xb, xv[t]); break;
                        case 1: gld128<64>(xf[s][t], xb, xv[t]); break;
                        case 2: gld128<128>(xf[s][t], xb, xv[t]); break;
                        case 3: gld128<192>(xf[s][t], xb, xv[t]); break;
                        case 4: gld128<256>(xf[s][t], xb, xv[t]); break;
                        case 5: gld128<320>(xf[s][t], xb, xv[t]); break;
                        case 6: gld128<384>(xf[s][t], xb, xv[t]); break;
                        default: gld128<448>(xf[s][t], xb, xv[t]); break;
                    }
                }
        };
        request_x(std::integral_constant<int, 0>{});
        // ---- the wave's 2 KB of weights: registers -> own LDS slot -> MFMA layout ------------------------------------------------
        wait_vm<NML + XB>();
        tie4(w0);
        tie4(w1);
        *(u32x4*)(wslot + wr_off) = w0;
        *(u32x4*)(wslot + WSLOT_I1 + wr_off) = w1;
        uint32_t bw[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) bw[s] = wslot[rd_off + s * 32];
        wait_vm<XB>();
#pragma unroll
        for (int l = 0; l < NML; ++l) tie1(mraw[l]);

        f32x4 acc[MT], ones[MT];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if constexpr (XS < 8) {  // the previous batch has been consumed (its last wait was vmcnt(0)): the next one reuses its registers
                if (s == XS) request_x(std::integral_constant<int, XS>{});
                if constexpr (XS < 4) {
                    if (s == 2 * XS) request_x(std::integral_constant<int, 2 * XS>{});
                    if (s == 3 * XS) request_x(std::integral_constant<int, 3 * XS>{});
                }
            }
            const uint32_t t_lo = bw[s] & 0x0F0F0F0Fu, t_hi = (bw[s] >> 4) & 0x0F0F0F0Fu;
            u32x4 bf;
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) bf[pq] = __builtin_amdgcn_perm(t_hi, t_lo, 0x0C040C00u + (uint32_t)pq * 0x00010001u) | TR::MAGIC2;
            const bool first = s % SPG == 0;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                // request (s % XS) MT + t of the batch: everything newer may still be in flight
                switch (XB - 1 - ((s % XS) * MT + t)) {
#define GL_R5_W(n) case n: wait_vm<n>(); break;
                    GL_R5_W(0) GL_R5_W(1) GL_R5_W(2) GL_R5_W(3) GL_R5_W(4) GL_R5_W(5) GL_R5_W(6) GL_R5_W(7) GL_R5_W(8) GL_R5_W(9) GL_R5_W(10)
                    GL_R5_W(11) GL_R5_W(12) GL_R5_W(13) GL_R5_W(14) GL_R5_W(15) GL_R5_W(16) GL_R5_W(17) GL_R5_W(18) GL_R5_W(19) GL_R5_W(20)
                    GL_R5_W(21) GL_R5_W(22) GL_R5_W(23) GL_R5_W(24) GL_R5_W(25) GL_R5_W(26) GL_R5_W(27) GL_R5_W(28) GL_R5_W(29) GL_R5_W(30)
                    default: wait_vm<31>(); break;
#undef GL_R5_W
                }
                tie4(xf[s % XS][t]);
                acc[t] = mfma16<Tag>(xf[s % XS][t], bf, first ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[t]);
                ones[t] = mfma16<Tag>(xf[s % XS][t], onesb, first ? (f32x4){0.f, 0.f, 0.f, 0.f} : ones[t]);
            }
            if ((s + 1) % SPG == 0) {  // end of a quantisation group: fold scale / zero into the totals
                const int q = s / SPG, l = q >> 1, gq = q & 1;
                const uint32_t sraw = (uint32_t)__builtin_amdgcn_ds_bpermute((j + 32 * gq) * 4, (int)mraw[l]);
                const uint32_t zraw = (uint32_t)__builtin_amdgcn_ds_bpermute((j + 32 * gq + 16) * 4, (int)mraw[l]);
                const float sv = need_s ? TR::to_float((uint16_t)sraw) : 1.f;
                const float zv = need_z ? TR::to_float((uint16_t)zraw) : scalar_zero;
                const float a = sv;
                const float b = bz * zv * (b_times_s ? sv : 1.f) - a * TR::OFF;
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tot[t][r] += a * acc[t][r] + b * ones[t][r];
            }
        }
    }

    // ---- the NW waves (disjoint K) meet in LDS: slot = [MT][16 rows][16 columns] fp32; C layout: column j, rows 4 kb + r ------------
    float* part = (float*)slot;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(t * 16 + 4 * kb + r) * TC + j] = tot[t][r];
    __syncthreads();
    constexpr int NPAIR = MT * 16 * (TC / 2);  // pairs of adjacent columns in the tile
    for (int o = tid; o < NPAIR; o += NW * 64) {
        const int m = o >> 3, cp = o & 7;
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float2 pv = *(const float2*)(smem + (size_t)w * SLOT + (size_t)(m * TC + cp * 2) * 4);
            v0 += pv.x;
            v1 += pv.y;
        }
        if (m < M) *(uint32_t*)(out + (size_t)m * som + (size_t)(tile * TC + cp * 2)) = (uint32_t)TR::from_float(v0) | ((uint32_t)TR::from_float(v1) << 16);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host-side planning.  tuning[0] = 9 forces this kernel (any M <= 64 it can take), tuning[2] = 8 / 16 its waves per block;
// tuning[3] & 65536 = never (the round-4 choice, A/B runs)
// ---------------------------------------------------------------------------------------------------------------------
typedef void (*rows5_fn)(const char*, const char*, const char*, const char*, uint16_t*, uint32_t, uint32_t, int, uint32_t, int, uint32_t, uint32_t);

// waves per block: 16 while a wave's chunk fits 128 registers (<= 32 rows with groups of >= 128, <= 16 rows else), 8 (256 registers) above;
// 8 waves also exist for the 16-wave shapes with groups of >= 128 (tuning[2] = 8, A/B runs)
template <typename Tag, int MT, int SPG>
static rows5_fn rows5_pick_nw(int nw) {
    constexpr bool W16 = MT == 1 || (MT == 2 && SPG == 4);
    if constexpr (W16) {
        if (nw == 0 || nw == 16) return gemm_w4_rows_kernel<Tag, MT, SPG, 16>;
        if constexpr (SPG == 4) { if (nw == 8) return gemm_w4_rows_kernel<Tag, MT, SPG, 8>; }
        return nullptr;
    } else {
        return (nw == 0 || nw == 8) ? gemm_w4_rows_kernel<Tag, MT, SPG, 8> : nullptr;
    }
}
template <typename Tag, int MT>
static rows5_fn rows5_pick_spg(int spg, int nw) {
    switch (spg) {
        case 4: return rows5_pick_nw<Tag, MT, 4>(nw);
        case 2: return rows5_pick_nw<Tag, MT, 2>(nw);
        case 1: return rows5_pick_nw<Tag, MT, 1>(nw);
        default: return nullptr;
    }
}
template <typename Tag>
static rows5_fn rows5_pick(int mt, int spg, int nw) {
    switch (mt) {
        case 1: return rows5_pick_spg<Tag, 1>(spg, nw);
        case 2: return rows5_pick_spg<Tag, 2>(spg, nw);
        case 3: return rows5_pick_spg<Tag, 3>(spg, nw);
        case 4: return rows5_pick_spg<Tag, 4>(spg, nw);
        default: return nullptr;
    }
}
static int rows5_waves(int mt, int spg, int want) { return want ? want : ((mt == 1 || (mt == 2 && spg == 4)) ? 16 : 8); }

bool plan_gemm_wn_rows(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    if (a.W_nbits != 4 || a.w_pack_bits != 32) return false;
    if (a.M < 1 || a.M > 65535 * 64) return false;  // above 64 rows: 64-row blocks along grid.y (the weights stream once per block row)
    if (a.input_dtype != GEMLITE_DT_FP16 && a.input_dtype != GEMLITE_DT_BF16) return false;
    if (a.output_dtype != a.input_dtype || a.channel_scale_mode != 0 || a.stride_on != 1 || a.stride_xk != 1 || a.stride_wn != 1) return false;
    const bool loop_s = a.W_group_mode >= 2, has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (loop_s && a.meta_dtype != a.input_dtype) return false;
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != a.input_dtype) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    if ((a.stride_xm * 2) % 16 != 0 || ((uintptr_t)a.x % 16) != 0) return false;        // 16-byte A fragments
    if (((uintptr_t)a.w_q % 16) != 0 || (a.stride_wk % 4) != 0) return false;            // 16-byte weight loads
    if (((uintptr_t)a.out % 4) != 0 || a.stride_om % 2 != 0) return false;               // packed pairs of outputs
    if (a.N % 16 != 0 || a.K % 256 != 0) return false;
    if (p.gs_shift < 5 || (p.gs_shift > 30 && p.gs_shift != 31)) return false;            // groups of 32, 64, 128, ... k (31: one group spans K)
    const int spg = p.gs_shift >= 7 ? 4 : (p.gs_shift == 6 ? 2 : 1);
    const int64_t rows = a.K / 8;
    // 32-bit byte offsets in the kernel
    if (rows * a.stride_wk * 4 + a.N * 4 >= (1ll << 32) || ((a.M < 64 ? a.M : 64) * a.stride_xm + a.K) * 2 >= (1ll << 32) || (a.M < 64 ? a.M : 64) * a.stride_om * 2 >= (1ll << 32)) return false;
    if (p.gs_shift < 31 && ((a.K >> p.gs_shift) * p.stride_meta_g + a.N) * 2 >= (1ll << 32)) return false;
    const int mt = a.M > 64 ? 4 : (int)((a.M + 15) / 16);
    if (a.tuning[2] != 0 && a.tuning[2] != 8 && a.tuning[2] != 16) return false;
    const bool f16 = a.input_dtype == GEMLITE_DT_FP16;
    const rows5_fn fn = f16 ? rows5_pick<half_tag>(mt, spg, a.tuning[2]) : rows5_pick<bf16_tag>(mt, spg, a.tuning[2]);
    if (!fn) return false;
    const int nw = rows5_waves(mt, spg, a.tuning[2]);
    const int tiles = (int)(a.N / 16);
    const bool need_s = loop_s, need_z = has_z && !a.zero_is_scalar;
    p.splitk = 1;
    p.rows_per_slice = (int)rows;
    lp.fn = (const void*)fn;
    static const char* names[4] = {"gemm_w4_rows_kernel<16x16>", "gemm_w4_rows_kernel<32x16>", "gemm_w4_rows_kernel<48x16>", "gemm_w4_rows_kernel<64x16>"};
    lp.name = names[mt - 1];
    lp.grid = dim3((unsigned)tiles, (unsigned)((a.M + 16 * mt - 1) / (16 * mt)), 1);
    lp.block = dim3(64 * nw, 1, 1);
    const size_t slot = (size_t)(mt * 1024 > 2304 ? mt * 1024 : 2304);
    lp.lds_bytes = slot * nw;
    lp.slab_bytes = 0;
    lp.ws_bytes = 0;
    lp.arg_kind = 2;
    lp.r5.w = (const char*)p.w;
    lp.r5.x = (const char*)p.x;
    lp.r5.s = need_s ? (const char*)p.scales : (const char*)p.w;
    lp.r5.z = (need_z || (has_z && a.zero_is_scalar)) ? (const char*)p.zeros : (const char*)p.w;
    lp.r5.out = (uint16_t*)p.epi.out;
    lp.r5.sw4 = (uint32_t)a.stride_wk * 4u;
    lp.r5.mstride2 = ((need_s || need_z) && p.gs_shift < 31) ? (uint32_t)p.stride_meta_g * 2u : 0u;
    lp.r5.nch_total = (int)(rows / 32);
    lp.r5.modes = (uint32_t)a.W_group_mode | ((has_z && a.zero_is_scalar) ? 16u : 0u) | (((tiles & 15) == 0) ? 32u : 0u) | ((uint32_t)p.gs_shift << 8);
    lp.r5.M = (int)a.M;
    lp.r5.sxm2 = (uint32_t)a.stride_xm * 2u;
    lp.r5.som = (uint32_t)a.stride_om;
    return true;
}

}  // namespace gl
