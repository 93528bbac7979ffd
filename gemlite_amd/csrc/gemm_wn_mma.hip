// gemm_wn_mma.hip — fused unpack + dequant + tiled MFMA GEMM for packed low-bit weights (4 / 2 / 1 / 8-bit words),
// the large-M (prefill) kernel.  Replaces gemm_INT_kernel (gemlite/triton_kernels/gemm_kernels.py:248-413) and, for
// shapes the few-row kernels do not take, gemm_splitK_INT_kernel (gemm_splitK_kernels.py:277-450).
//
// Design (CDNA4; numbers in DESIGN.md §3.3):
//   * Block = 8 waves = two waves per SIMD, tile (32*MI) x 128, K step KSTEP (128 or 256).  Wave (cg = wave & 3,
//     kh = wave >> 2) owns ALL rows x columns [32cg, 32cg+32) x the kh-th HALF of every K step: the two waves of a SIMD
//     work on different halves of K, so one wave's MFMAs cover the other's unpack arithmetic, LDS reads and memory
//     requests, and a 256-row tile (the shape that needs the least dequant work per MFMA: 19 VALU per 8 MFMAs) no
//     longer leaves the SIMD with a single in-order wave.  The two K halves are added through LDS in the epilogue.
//   * B: one packed int32 word holds 8 (4-bit) / 16 / 32 / 4 consecutive k of ONE column, i.e. 1 / 2 / 4 / half of a
//     lane's B fragment of v_mfma_f32_32x32x16 (lane = column l & 31, k-octet l >> 5).  The wave loads its words
//     straight from HBM / L2 with buffer loads (voffset = column, soffset = packed row: no VALU per load),
//     dequantises them in registers in natural k order and feeds MI MFMAs with each fragment.  Weights never touch LDS.
//   * A (x): the (32*MI) x KSTEP tile of every step goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), no
//     VGPR staging and no ds_write; the 16-byte slots of a row are XOR-swizzled with (row & 15) by permuting the
//     per-lane SOURCE address (the LDS image of a DMA is lane-linear), which makes the ds_read_b128 of 32 rows
//     conflict-free.  Two stages; everything a step needs is requested one step (x) / two steps (weights) ahead,
//     so ONE s_waitcnt vmcnt(0) + ONE s_barrier per step is all the synchronisation there is, and both sit a few
//     MFMA slots before the end of the step so that the first fragments of the next stage are read under MFMAs.
//   * K may be split over gridDim.y; slices are combined with the write-through slab + ticket protocol (gl_common.h).
#include <type_traits>

#include "gl_common.h"
#include "gl_async.h"

namespace gl {

namespace mma {

using namespace async;

constexpr int BN = 128;
constexpr int C_ROWS = 128;        // rows of the epilogue staging tile (one pass per 128 rows)
constexpr int C_PITCH = BN + 4;    // floats per row of that tile
constexpr int LOOKAHEAD = 4;       // A fragments requested ahead of their MFMA

template <typename Tag>
__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c);
template <>
__device__ __forceinline__ f32x16 mfma32<half_tag>(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mfma32<bf16_tag>(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
}

// ---- activation element type of the MFMA --------------------------------------------------------------------------------
// XDT = 0: x holds the 16-bit floats of Tag.  XDT = GEMLITE_DT_FP8E4 / GEMLITE_DT_INT8: 8-bit activations (the reference's
// A8Wn dynamic and BitNet-int8 processors, helper.py:502-615, 1006-1062): the dequantised weight is cast to the activation
// type, like the reference's `b.to(a.dtype)` before tl.dot (gemm_kernels.py:384), and the product runs on the fp8 / int8
// MFMA of the same 32x32x16 shape — a lane's fragment is still 8 consecutive k, now 8 bytes instead of 16.
template <typename Tag, int XDT>
struct XOps {
    static constexpr int ES = 2;
    typedef u32x4 frag_t;
    typedef f32x16 acc_t;
    static __device__ __forceinline__ f32x16 mfma(frag_t a, frag_t b, f32x16 c) { return mfma32<Tag>(a, b, c); }
};
template <typename Tag>
struct XOps<Tag, GEMLITE_DT_FP8E4> {
    static constexpr int ES = 1;
    typedef u32x2 frag_t;
    typedef f32x16 acc_t;
    static __device__ __forceinline__ f32x16 mfma(frag_t a, frag_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
    }
};
template <typename Tag>
struct XOps<Tag, GEMLITE_DT_INT8> {  // int32 accumulation (converted to fp32 once, after the K loop)
    static constexpr int ES = 1;
    typedef u32x2 frag_t;
    typedef i32x16 acc_t;
    static __device__ __forceinline__ i32x16 mfma(frag_t a, frag_t b, i32x16 c) {
        return __builtin_amdgcn_mfma_i32_32x32x16_i8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b), c, 0, 0, 0);
    }
};

// ---- geometry of one 64-k sub-block by bit width ------------------------------------------------------------------
// WPL words per lane, packed row of word i for lane half h: rb + RI(i) + HS * h, k offset (inside the sub-block) of
// MFMA slice u for lane half h: KO(u, h).
template <int NBITS>
struct Geo {
    static constexpr int E = 32 / NBITS;
    static constexpr int WPL = NBITS;                       // 4 -> 4 words, 2 -> 2, 1 -> 1, 8 -> 8
    static constexpr int ROWS = 64 / E;                     // packed rows per sub-block
    static constexpr int HS = NBITS == 8 ? 2 : 1;
    static __device__ __forceinline__ constexpr int row_of(int i) { return NBITS == 8 ? 4 * (i >> 1) + (i & 1) : (NBITS == 1 ? 0 : 2 * i); }
    static __device__ __forceinline__ constexpr int k_of(int u, int h) {
        return NBITS == 2 ? 16 * (2 * (u >> 1) + h) + 8 * (u & 1) : (NBITS == 1 ? 32 * h + 8 * u : 8 * (2 * u + h));
    }
};

// Block-scaled weights under 16-bit activations (A16W8 / A16W4 MXFP, helper.py:372-400): NBITS = 108 (fp8 e4m3) / 104 (e2m1
// codes, two per byte).  These rows are K-CONTIGUOUS per output column (core.py:363-398): lane half h owns the 32 k of ONE
// microscaling block of the 64-k sub-block — 32 / 16 consecutive bytes = WPL dwords, fetched as 16-byte pieces — and
// slice u is its k [8u, 8u + 8): k_of(u, h) = 32 h + 8 u (any k order works as long as A follows).  E = 1: "packed rows"
// are k itself.
constexpr int MXW8 = 108, MXW4 = 104;
template <>
struct Geo<MXW8> {
    static constexpr int E = 1, WPL = 8, ROWS = 64, HS = 0, WBYTES64 = 64;
    static __device__ __forceinline__ constexpr int row_of(int) { return 0; }
    static __device__ __forceinline__ constexpr int k_of(int u, int h) { return 32 * h + 8 * u; }
};
template <>
struct Geo<MXW4> {
    static constexpr int E = 1, WPL = 4, ROWS = 64, HS = 0, WBYTES64 = 32;
    static __device__ __forceinline__ constexpr int row_of(int) { return 0; }
    static __device__ __forceinline__ constexpr int k_of(int u, int h) { return 32 * h + 8 * u; }
};
// pair j (k = 2j, 2j + 1 of the slice) of a block-scaled fragment: hardware converters with the block scale applied
// (v_cvt_scalef32_pk_{bf16,f16}_{fp8,fp4}; operand order measured with scripts/ubench/probe_mx.hip)
template <typename Tag, int NBITS, int J>
__device__ __forceinline__ uint32_t mx_pair_c(const uint32_t* w, int u, float sc) {  // the selectors are immediates
    if constexpr (NBITS == MXW4) {
        if constexpr (std::is_same<Tag, bf16_tag>::value) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w[u], sc, J));
        else return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp4(w[u], sc, J));
    } else {
        const uint32_t src = w[2 * u + (J >> 1)];
        if constexpr (std::is_same<Tag, bf16_tag>::value) return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src, sc, (J & 1) != 0));
        else return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(src, sc, (J & 1) != 0));
    }
}
template <typename Tag, int NBITS>
__device__ __forceinline__ uint32_t mx_pair(const uint32_t* w, int u, int j, float sc) {
    switch (j) {
        case 0: return mx_pair_c<Tag, NBITS, 0>(w, u, sc);
        case 1: return mx_pair_c<Tag, NBITS, 1>(w, u, sc);
        case 2: return mx_pair_c<Tag, NBITS, 2>(w, u, sc);
        default: return mx_pair_c<Tag, NBITS, 3>(w, u, sc);
    }
}

// ---- integer codes of slice u as bytes: byte j of `ev` = code 2j, byte j of `od` = code 2j + 1 (natural k order) -----
template <int NBITS>
struct Extract;
template <>
struct Extract<4> {
    __device__ __forceinline__ static void run(const uint32_t (&w)[4], int u, uint32_t& ev, uint32_t& od) {
        ev = w[u] & 0x0F0F0F0Fu;
        od = (w[u] >> 4) & 0x0F0F0F0Fu;
    }
};
template <>
struct Extract<8> {
    __device__ __forceinline__ static void run(const uint32_t (&w)[8], int u, uint32_t& ev, uint32_t& od) {
        ev = __builtin_amdgcn_perm(w[2 * u + 1], w[2 * u], 0x06040200u);  // bytes 0, 2 of each word
        od = __builtin_amdgcn_perm(w[2 * u + 1], w[2 * u], 0x07050301u);  // bytes 1, 3
    }
};
template <>
struct Extract<2> {
    // m_a = (w >> 2a) & 0x03030303: byte b of m_a = code 4b + a.  Slice t = u & 1 of word u >> 1 holds codes 8t .. 8t+7.
    __device__ __forceinline__ static void run(const uint32_t (&w)[2], int u, uint32_t& ev, uint32_t& od) {
        const uint32_t x = w[u >> 1];
        const uint32_t m0 = x & 0x03030303u, m1 = (x >> 2) & 0x03030303u, m2 = (x >> 4) & 0x03030303u, m3 = (x >> 6) & 0x03030303u;
        const uint32_t sel = (u & 1) ? 0x07030602u : 0x05010400u;  // {m0.b[2t], m2.b[2t], m0.b[2t+1], m2.b[2t+1]}
        ev = __builtin_amdgcn_perm(m2, m0, sel);
        od = __builtin_amdgcn_perm(m3, m1, sel);
    }
};
template <>
struct Extract<1> {
    // m_a = (w >> a) & 0x01010101: byte b of m_a = bit 8b + a, i.e. code a of slice b
    __device__ __forceinline__ static void run(const uint32_t (&w)[1], int u, uint32_t& ev, uint32_t& od) {
        const uint32_t x = w[0];
        uint32_t m[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) m[a] = (x >> a) & 0x01010101u;
        const uint32_t s2 = 0x0C0C0400u + 0x00000101u * (uint32_t)u;  // {lo.b[u], hi.b[u], 0, 0}
        const uint32_t e01 = __builtin_amdgcn_perm(m[2], m[0], s2), e23 = __builtin_amdgcn_perm(m[6], m[4], s2);
        const uint32_t o01 = __builtin_amdgcn_perm(m[3], m[1], s2), o23 = __builtin_amdgcn_perm(m[7], m[5], s2);
        ev = e01 | (e23 << 16);
        od = o01 | (o23 << 16);
    }
};

// ---- codes -> one B fragment (8 dequantised 16-bit floats, natural k order) --------------------------------------------
template <typename Tag>
struct Convert;
// Integer codes 0 .. 15 read as e4m3 BYTES are linear (b < 8: subnormal b * 2^-9; 8 <= b < 16: (8 + (b - 8)) * 2^-9), so the
// block-scale converter of gfx950 turns two code bytes into two exact fp32 integers in ONE instruction (scale 2^9) where
// v_cvt_f32_ubyte<i> takes two.  HI selects bytes 2, 3 of the register (an immediate).
typedef float f2_t __attribute__((ext_vector_type(2)));
template <bool HI>
__device__ __forceinline__ f2_t codes_f32x2(uint32_t bytes) { return __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(bytes, 512.0f, HI); }
// The affine map stays two SCALAR v_fma_f32 per converter result: v_pk_fma_f32 beside MFMAs is an anti-lever (guide: +22
// cycles per instruction; measured here on one box, scripts/ab_mma.sh: cfgB 47.6 us with v_cvt_f32_ubyte, 44.7 with the
// converter + scalar fmas, 47.6 with the converter + packed fmas; prefill 251 / 250 / 264 us)
__device__ __forceinline__ f2_t affine2(f2_t q, float A, float B) { return (f2_t){__builtin_fmaf(q.x, A, B), __builtin_fmaf(q.y, A, B)}; }

template <>
struct Convert<bf16_tag> {
    float A, B;  // v = fma(q, A, B) in fp32, rounded once to bf16
    f2_t te, to; // LIN: the fp32 values of codes (4h, 4h + 2) and (4h + 1, 4h + 3) of the half-slice in flight
    __device__ __forceinline__ void set(float s, float z, float u13, float u4) {
        A = s;
        B = z * __builtin_fmaf(-u13, s, u4);
    }
    // pair j = codes 2j, 2j+1 of the slice: two v_cvt_f32_ubyte<j>, two v_fma_f32, one v_cvt_pk_bf16_f32
    __device__ __forceinline__ uint32_t pair(uint32_t ev, uint32_t od, int j) const {
        const float lo = (float)((ev >> (8 * j)) & 0xFFu), hi = (float)((od >> (8 * j)) & 0xFFu);
        const b2_t v = {(__bf16)__builtin_fmaf(lo, A, B), (__bf16)__builtin_fmaf(hi, A, B)};
        return __builtin_bit_cast(uint32_t, v);
    }
    // LIN (codes < 16): per half-slice two converter ops + four v_fma_f32, then one v_cvt_pk_bf16_f32 per pair: 19 VALU per
    // fragment with the extraction instead of 23
    template <bool LIN>
    __device__ __forceinline__ uint32_t put(uint32_t ev, uint32_t od, int j) {
        if constexpr (!LIN) return pair(ev, od, j);
        else {
            if ((j & 1) == 0) {
                te = affine2((j >> 1) ? codes_f32x2<true>(ev) : codes_f32x2<false>(ev), A, B);
                to = affine2((j >> 1) ? codes_f32x2<true>(od) : codes_f32x2<false>(od), A, B);
            }
            const b2_t v = {(__bf16)((j & 1) ? te.y : te.x), (__bf16)((j & 1) ? to.y : to.x)};
            return __builtin_bit_cast(uint32_t, v);
        }
    }
};
template <>
struct Convert<half_tag> {
    h2_t zsub2, s2, zadd2;  // v = fma(q - zsub, s, zadd) in fp16: rounds where the reference rounds (utils.py:73-87)
    __device__ __forceinline__ void set(float s, float z, float u13, float u4) {
        const _Float16 zs = (_Float16)(z * u13), za = (_Float16)(z * u4), sc = (_Float16)s;
        zsub2 = (h2_t){zs, zs};
        zadd2 = (h2_t){za, za};
        s2 = (h2_t){sc, sc};
    }
    __device__ __forceinline__ uint32_t pair(uint32_t ev, uint32_t od, int j) const {
        // {0, od.b[j], 0, ev.b[j]} | (1024, 1024): 1024 + q exactly, then q = that - 1024
        const uint32_t sel = 0x0C040C00u + 0x00010001u * (uint32_t)j;
        const uint32_t h = __builtin_amdgcn_perm(od, ev, sel) | 0x64006400u;
        const h2_t q = __builtin_bit_cast(h2_t, h) - (h2_t){(_Float16)1024.0f, (_Float16)1024.0f};
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(q - zsub2, s2, zadd2));
    }
    template <bool LIN>
    __device__ __forceinline__ uint32_t put(uint32_t ev, uint32_t od, int j) { return pair(ev, od, j); }
};

// ---- codes -> one B fragment in the activation type ----------------------------------------------------------------------
template <typename Tag, int XDT>
struct ConvertX {  // 16-bit activations: the converters above, pair j -> register j
    Convert<Tag> c;
    __device__ __forceinline__ void set(float s, float z, float u13, float u4) { c.set(s, z, u13, u4); }
    __device__ __forceinline__ void prep(uint32_t&, uint32_t&) const {}
    template <bool LIN>
    __device__ __forceinline__ void put(u32x4& out, uint32_t ev, uint32_t od, int j) { out[j] = c.template put<LIN>(ev, od, j); }
};
template <typename Tag>
struct ConvertX<Tag, GEMLITE_DT_FP8E4> {  // fp32 fma, one rounding to e4m3 (v_cvt_pk_fp8_f32): pair j -> half j & 1 of register j >> 1
    float A, B;
    f2_t te, to;
    __device__ __forceinline__ void set(float s, float z, float u13, float u4) {
        A = s;
        B = z * __builtin_fmaf(-u13, s, u4);
    }
    __device__ __forceinline__ void prep(uint32_t&, uint32_t&) const {}
    template <bool LIN>
    __device__ __forceinline__ void put(u32x2& out, uint32_t ev, uint32_t od, int j) {
        float a, b;
        if constexpr (LIN) {  // see Convert<bf16_tag>::put
            if ((j & 1) == 0) {
                te = affine2((j >> 1) ? codes_f32x2<true>(ev) : codes_f32x2<false>(ev), A, B);
                to = affine2((j >> 1) ? codes_f32x2<true>(od) : codes_f32x2<false>(od), A, B);
            }
            a = (j & 1) ? te.y : te.x;
            b = (j & 1) ? to.y : to.x;
        } else {
            const float lo = (float)((ev >> (8 * j)) & 0xFFu), hi = (float)((od >> (8 * j)) & 0xFFu);
            a = __builtin_fmaf(lo, A, B), b = __builtin_fmaf(hi, A, B);
        }
        const uint32_t prev = out[j >> 1];
        out[j >> 1] = (j & 1) ? (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)prev, true)
                              : (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)prev, false);
    }
};
template <typename Tag>
struct ConvertX<Tag, GEMLITE_DT_INT8> {  // integer codes minus an integer zero (W_group_mode 0 / 1 with a scalar zero): exact int8
    uint32_t zz;  // the zero in every byte
    __device__ __forceinline__ void set(float, float z, float u13, float) { zz = 0x01010101u * ((uint32_t)(int)(z * u13) & 0xFFu); }
    // bytewise q - z without borrows between bytes: ((q | 0x80) - z) ^ 0x80  (q <= 127 + z, z <= 127)
    __device__ __forceinline__ void prep(uint32_t& ev, uint32_t& od) const {
        ev = ((ev | 0x80808080u) - zz) ^ 0x80808080u;
        od = ((od | 0x80808080u) - zz) ^ 0x80808080u;
    }
    template <bool LIN>
    __device__ __forceinline__ void put(u32x2& out, uint32_t ev, uint32_t od, int j) const {
        if (j & 1) out[j >> 1] = __builtin_amdgcn_perm(od, ev, j == 1 ? 0x05010400u : 0x07030602u);  // {ev[2r], od[2r], ev[2r+1], od[2r+1]}
    }
};

// (in a __device__ function: a "v" constraint inside a lambda of the kernel body silently drops the kernel's host stub)
__device__ __forceinline__ void opaque2(uint32_t& a, uint32_t& b) { asm volatile("" : "+v"(a), "+v"(b)); }

// One LDS-DMA piece: 64 lanes x 16 bytes from buffer offset (voff per lane + soff) to LDS [dst, dst + 1024), lane-linear.
// (A __device__ function: the address-space cast inside a kernel-body lambda silently drops the kernel's host stub.)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, unsigned char* dst, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}

}  // namespace mma

// MI 32-row blocks per wave (tile rows = 32 * MI), KSTEP k per step (each wave: KSTEP / 2).
// EXP (development builds only, -DGL_MMA_EXPERIMENTS + tuning[3] >> 8): drop parts of the K loop to see what each costs —
// 1 barrier + counted wait, 2 dequant VALU, 4 A-fragment reads, 8 x DMA requests, 16 weight requests, 32 the scalar address
// arithmetic of the requests (constant offsets).  Results are wrong.
// KH = 2 (default): tile 32 MI x 128, wave (cg = wave & 3, kh = wave >> 2) owns 32 columns x one HALF of every K step.
// KH = 1 ("wide"): tile 32 MI x 256, wave cg = wave owns 32 columns x the whole step — every x byte DMA'd into LDS feeds twice
// the columns: the 128-column tiles move 576 B of operands per k for 2 * 256 * 128 flop, i.e. ~10 TB/s through the L2s at the
// prefill rate they reach, which is the measured ceiling of that path; the wide tile needs 640 B per k for twice the flop.
template <typename Tag, int NBITS, int MI, int KSTEP, int RD, int NST, int EXP = 0, int XDT = 0, int KH = 2>
__global__ __launch_bounds__(512, 2) void gemm_wn_mma_kernel(const WnParams p) {
    using namespace mma;
    using TR = F16Traits<Tag>;  // output / metadata type; also the activation type when XDT == 0
    using G = Geo<NBITS>;
    using XO = XOps<Tag, XDT>;
    typedef typename XO::frag_t frag_t;
    constexpr int ES = XO::ES;  // bytes per activation
    constexpr int BM = 32 * MI, KW = KSTEP / KH, SUB = KW / 64, WPL = G::WPL;
    constexpr int BN = 256 / KH, C_PITCH = BN + 4;  // (shadow the namespace's 128-column constants)
    static_assert(KH == 1 || KH == 2, "one or two K parts per step");
    constexpr bool MXW = NBITS == MXW8 || NBITS == MXW4;  // block-scaled K-contiguous weights (16-bit activations only)
    static_assert(!MXW || XDT == 0, "block-scaled weights on this kernel: 16-bit activations");
    constexpr int PITCH = KSTEP * ES, STAGE = BM * PITCH;  // bytes per row / per stage of x
    constexpr int SWZ = (PITCH / 16 < 16 ? PITCH / 16 : 16) - 1;  // XOR swizzle of the 16-byte slots inside a row (8 or 16 slots)
    // the swizzle key of row r: r itself for rows of >= 256 B (one row spans all 64 banks); 128-byte rows alternate between
    // the two bank halves, so rows r and r + 8 of one ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31})
    // would meet in the same slot AND the same half — key r >> 1 separates them (PMC on the first wide build:
    // SQ_LDS_BANK_CONFLICT = 49 % of SQ_LDS_IDX_ACTIVE with key r)
    constexpr int SWZ_SH = (PITCH == 128 && ES == 2) ? 1 : 0;
    constexpr int PIECES = STAGE / 1024 / 8;               // 1-KiB LDS-DMA pieces per wave and stage
    constexpr int NS = SUB * 4;                            // MFMA slices (k16) per wave and step
    constexpr int NQ = NS * MI;                            // MFMA slots per wave and step
    constexpr int L = LOOKAHEAD;
    constexpr int PD = RD - 2;  // weights are requested PD steps ahead (ring of RD steps: short steps need a deep ring —
                                // an HBM round trip under load is 2-3k cycles, a step of the 32-row tile 512)
    // NST LDS stages of x: the tile of step s + NST - 1 is requested during step s.  With two stages the DMA is waited for
    // in the very step that issued it, which only a long step (256-row tile: ~2k cycles) covers; the shorter steps of the
    // smaller tiles spent ~40 % of their time in that wait (profiles/r02 PMC: SQ_WAIT_ANY 42 % of wave cycles).
    static_assert(SUB >= 1 && PIECES >= 1 && NQ >= 2 * L && RD % NST == 0 && RD >= 4 && NST >= 2, "tile too small for the slot schedule");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NST][STAGE], later the epilogue tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = KH == 2 ? (wave & 3) : wave, kh = KH == 2 ? (wave >> 2) : 0;
    const int col = lane & 31, h = lane >> 5;
    const int mtiles = (p.M + BM - 1) / BM;
    // (tile, K slice) of this block.  Opt-in (tuning[3] & 8): block b runs on XCD b % 8 and every XCD has its own L2, so
    // giving all blocks of one XCD the SAME K slice keeps that slice's x rows (re-read by every column tile) in one L2.
    // Measured SLOWER (cfgB 53 vs 46 us, cfgA 21.2 vs 19.1 us: 32 CUs asking for the same lines at once), so the default
    // is the plain (x = tile, y = slice) grid.  Speed only; any map is correct.
    int bid = blockIdx.x, slice = blockIdx.y;
    {
        const int T = gridDim.x, S = gridDim.y;
        if ((p.flags & 8) && S > 1 && (8 % S) == 0 && ((T * S) & 7) == 0) {
            const int lin = blockIdx.x + T * blockIdx.y, xcd = lin & 7, idx = lin >> 3;
            slice = xcd % S;
            bid = idx * (8 / S) + xcd / S;
        }
    }
    const int mt = bid % mtiles, nt = bid / mtiles;  // M tiles fastest: neighbours share the weight tile in L2
    const int m0 = mt * BM;
    const int n = nt * BN + cg * 32 + col;  // this lane's column

    // K steps of this slice: the units are dealt as evenly as possible (rows_per_slice = ALL packed rows here), so any
    // step count works (K = 11008: 43 steps of 256)
    constexpr int STEP_ROWS = KSTEP / G::E;
    const int units = p.rows_per_slice / STEP_ROWS;
    const int s_begin = (int)((int64_t)slice * units / p.splitk), s_end = (int)((int64_t)(slice + 1) * units / p.splitk);
    const int nsteps = s_end - s_begin;
    const int row_s0 = s_begin * STEP_ROWS;  // first packed row of the slice
    const int k_s0 = row_s0 * G::E;

    // opt-in timeline (tuning[3] & 4): lane 0 of every wave of block (0, 0) stores s_memtime stamps behind the tickets
    const bool probe = (p.flags & 4) && p.counters && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0;
    unsigned long long* stamps = (unsigned long long*)(p.counters + MAX_SPLITK_COUNTERS) + wave * 16;
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if (probe) stamps[i] = __builtin_readcyclecounter();
    };
    stamp(0);

    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    const float scalar_zero = p.zero_is_scalar ? (float)((const int32_t*)p.zeros)[0] : 0.f;
    const float u13 = (p.w_mode == 1 || p.w_mode == 3) ? 1.f : 0.f, u4 = p.w_mode == 4 ? 1.f : 0.f;
    const int sw = (int)p.stride_wk;
    const int ms = (need_s || need_z) ? (int)p.stride_meta_g : 0;
    const int meta_rows = p.gs_shift >= 31 ? 1 : (p.K >> p.gs_shift);
    const int meta_bytes = ((meta_rows - 1) * ms + p.N) * 2;
    // buffer descriptors (rows >= M and absent metadata read zeros through the range check on the per-lane offset)
    const srd_t rsX = make_srd(p.x, (uint32_t)(((int64_t)(p.M - 1) * p.stride_xm + p.K) * ES));

    // ---- B stream --------------------------------------------------------------------------------------------------
    struct BStep { uint32_t w[SUB][WPL]; uint32_t s[SUB], z[SUB]; };
    const int wave_row0 = kh * (KW / G::E);  // first packed row of this wave's half inside a step
    constexpr int NREQ = MXW ? WPL / 4 + 1 : WPL + 2;  // requests per sub-block: MX = 16-byte pieces + one scale byte
    constexpr int NLB = SUB * NREQ;                    // weight / metadata requests per step and wave
    // request `it` (0 .. NLB-1) of step `step` (slice-relative) into ring slot b
    // (the weight / metadata requests are ordinary buffer loads the compiler tracks: it retires them with its own COUNTED
    //  vmcnt — it cannot see the asm DMAs, so its count can only over-wait, never under-wait — and it never copies a
    //  register whose load is still in flight, which it is free to do with the output of an asm load)
    uint32_t wvoff, mvoff;
    __amdgpu_buffer_rsrc_t brW, brS, brZ;
    if constexpr (MXW) {
        constexpr int WB = G::WBYTES64;  // weight bytes per 64 k
        wvoff = (uint32_t)((int64_t)n * p.stride_wn_b + (int64_t)(k_s0 + kh * KW) * WB / 64 + h * (WB / 2));
        mvoff = (uint32_t)((int64_t)n * p.stride_meta_n + (int64_t)((k_s0 + kh * KW) / 32 + h) * p.stride_meta_g);
        brW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, (short)0, (int)((int64_t)(p.N - 1) * p.stride_wn_b + (int64_t)p.K * WB / 64), 0x00020000);
        brS = __builtin_amdgcn_make_buffer_rsrc((void*)p.scales, (short)0,
                                                (int)((int64_t)(p.K / 32 - 1) * p.stride_meta_g + (int64_t)(p.N - 1) * p.stride_meta_n + 1), 0x00020000);
        brZ = brS;
    } else {
        wvoff = (uint32_t)(G::HS * h * sw + n) * 4u;
        mvoff = (uint32_t)n * 2u;
        brW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (int64_t)row_s0 * sw), (short)0, nsteps * STEP_ROWS * sw * 4, 0x00020000);
        brS = __builtin_amdgcn_make_buffer_rsrc((void*)(need_s ? p.scales : (const void*)p.w), (short)0, need_s ? meta_bytes : 4, 0x00020000);
        brZ = __builtin_amdgcn_make_buffer_rsrc((void*)(need_z ? p.zeros : (const void*)p.w), (short)0, need_z ? meta_bytes : 4, 0x00020000);
    }
    // (scalar offsets of the packed-word requests = step * bytes-per-step + a loop-invariant per request: one s_mul per step and
    //  one s_add per request instead of add / mul / shift each — the SIMD issues scalar and vector instructions of its two
    //  waves one after the other, 36 SALU per step were ~8 % of the loop)
    uint32_t wconst[SUB][WPL > 8 ? 1 : WPL];
    const uint32_t step_wbytes = MXW ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane(STEP_ROWS * sw * 4);
    if constexpr (!MXW) {
#pragma unroll
        for (int sb = 0; sb < SUB; ++sb)
#pragma unroll
            for (int i = 0; i < WPL; ++i)
                wconst[sb][i] = (uint32_t)__builtin_amdgcn_readfirstlane((wave_row0 + sb * G::ROWS + G::row_of(i)) * sw * 4);
    }
    const int kconst0 = k_s0 + kh * KW;
    const int ms2 = ms * 2;
    auto req_b = [&](BStep& b, int step, int it) __attribute__((always_inline)) {
        const int sb = it / NREQ, i = it % NREQ;
        if constexpr (MXW) {
            constexpr int WB = G::WBYTES64;
            if (i < WPL / 4) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                    brW, wvoff, (EXP & 32) ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((step * KSTEP + sb * 64) * WB / 64 + i * 16), 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) b.w[sb][4 * i + t] = v[t];
            } else {
                b.s[sb] = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(
                    brS, mvoff, (EXP & 32) ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane(((step * KSTEP + sb * 64) / 32) * (int)p.stride_meta_g), 0);
            }
        } else {
            if (i < WPL) {
                b.w[sb][i] = __builtin_amdgcn_raw_buffer_load_b32(brW, wvoff, (EXP & 32) ? 0u : (uint32_t)step * step_wbytes + wconst[sb][i], 0);
            } else {
                const uint32_t mo = (EXP & 32) ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane(((kconst0 + sb * 64 + step * KSTEP) >> p.gs_shift) * ms2);
                if (i == WPL) b.s[sb] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(brS, mvoff, mo, 0);
                else b.z[sb] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(brZ, mvoff, mo, 0);
            }
        }
    };
    // ---- A stream: LDS-DMA pieces.  Piece j of wave w covers LDS bytes [(w * PIECES + j) * 1024, +1024) of a stage;
    //      lane i's 16 bytes land at +16 i, i.e. row (byte / PITCH), physical slot (byte % PITCH) / 16, which holds the
    //      logical slot  phys ^ (row & SWZ)  of that row.
    uint32_t xvoff[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const int byte = (wave * PIECES + j) * 1024 + lane * 16;
        const int r = byte / PITCH, phys = (byte % PITCH) / 16;
        const int logical = phys ^ ((r >> SWZ_SH) & SWZ);
        xvoff[j] = m0 + r < p.M ? (uint32_t)(((int64_t)(m0 + r) * p.stride_xm + k_s0) * ES + logical * 16) : 0x80000000u;
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (uint32_t)(wave * PIECES) * 1024u);
    auto req_x = [&](int stage, int step, int j) __attribute__((always_inline)) {
        req_lds16(rsX, lds0 + (uint32_t)(stage * STAGE + j * 1024), xvoff[j], (EXP & 32) ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane(step * KSTEP * ES));
    };
    // A fragment of slot q = (slice g, row block mi): row mi*32 + col, k = kh*KW + (g/4)*64 + k_of(g%4, h)
    int fbase[NST][NS];
#pragma unroll
    for (int st = 0; st < NST; ++st)
#pragma unroll
        for (int g = 0; g < NS; ++g) {
            const int kb = (kh * KW + (g >> 2) * 64 + G::k_of(g & 3, h)) * ES;  // byte offset inside the row
            const int slot = kb >> 4;
            fbase[st][g] = st * STAGE + col * PITCH + (((slot & ~SWZ) | ((slot ^ (col >> SWZ_SH)) & SWZ)) << 4) + (kb & 15);
        }
    auto read_frag = [&](int stage, int q) __attribute__((always_inline)) -> frag_t {
        return *(const frag_t*)(smem + fbase[stage][q / MI] + (q % MI) * 32 * PITCH);
    };

    typename XO::acc_t accm[MI];  // fp32, or int32 for int8 activations
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) accm[i][e] = 0;

    ConvertX<Tag, XDT> cv;
    uint32_t ev = 0, od = 0;  // codes of the slice being dequantised
    // Dequantisation of one slice (-> 4 registers of a B fragment) is cut into pieces that hide behind the MI MFMAs of
    // the slice before it: piece 0 = (scale, zero) of the sub-block + code extraction, then the 4 pairs.
    float mx_sc = 0.f;  // block scale of the sub-block being dequantised (MX)
    auto deq_piece = [&](const BStep& b, int g, int mi, frag_t& out) __attribute__((always_inline)) {
        const int sb = g >> 2, u = g & 3;
        if constexpr (MXW) {
            // e8m0 byte -> fp32 2^(b - 127): the byte IS the exponent field (0 -> 0.0 instead of 2^-127: the quantisers
            // floor the scale at 2^-30)
            if (mi == 0 && u == 0) mx_sc = __builtin_bit_cast(float, b.s[sb] << 23);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((MI == 8 ? j + 1 : (j * MI) / 4) == mi) out[j] = mx_pair<Tag, NBITS>(b.w[sb], u, j, mx_sc);
        } else {
            if (mi == 0) {
                if (u == 0) {  // (scale, zero) of the sub-block; slices 1..3 reuse them
                    const float sc = need_s ? TR::to_float((uint16_t)b.s[sb]) : 1.f;
                    const float zr = need_z ? TR::to_float((uint16_t)b.z[sb]) : scalar_zero;
                    cv.set(sc, zr, u13, u4);
                }
                Extract<NBITS>::run(b.w[sb], u, ev, od);
                cv.prep(ev, od);
                // opaque to the optimiser: otherwise byte i becomes v_bfe_u32 + v_cvt_f32_ubyte0 instead of one v_cvt_f32_ubyte<i>
                opaque2(ev, od);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if ((MI == 8 ? j + 1 : (j * MI) / 4) == mi) cv.template put<(NBITS <= 4)>(out, ev, od, j);
        }
    };

    BStep ring[RD];
    frag_t af[L];
    frag_t bfrag[2];

    // ---- prologue: x of step 0, weights of steps 0 .. PD-1 ------------------------------------------------------------
    // request order = the order things are needed: x and weights of step 0 first.  Only the x tile of step 0 is waited
    // for (counted: everything issued after it may still be in flight); the compiler waits for the weights of step 0 where
    // they are first used.
#pragma unroll
    for (int j = 0; j < PIECES; ++j) req_x(0, 0, j);
#pragma unroll
    for (int it = 0; it < NLB; ++it) req_b(ring[0], 0, it);
#pragma unroll
    for (int st = 1; st < NST - 1; ++st)
#pragma unroll
        for (int j = 0; j < PIECES; ++j) req_x(st, st < nsteps ? st : nsteps - 1, j);
#pragma unroll
    for (int r = 1; r < PD; ++r)
#pragma unroll
        for (int it = 0; it < NLB; ++it) req_b(ring[r], r < nsteps ? r : nsteps - 1, it);
    {
        constexpr int AFTER = (NST - 2) * PIECES + PD * NLB;  // requests issued after the x tile of step 0
        wait_vm<(AFTER < 63 ? AFTER : 63)>();  // (the counter holds 63: with more issued behind it, the tile has landed anyway)
    }
    __builtin_amdgcn_s_barrier();
    stamp(1);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) deq_piece(ring[0], 0, mi, bfrag[0]);
#pragma unroll
    for (int q = 0; q < L; ++q) af[q] = read_frag(0, q);
    __builtin_amdgcn_sched_barrier(0);

    // One K step; J = step & 3 selects the ring slot / stage statically.  The step is written as NQ MFMA "slots"; slot q
    // issues MFMA q, one piece of the next slice's dequantisation, the A fragment needed L slots later and — in the first
    // slots — the step's memory requests: x of step + 1 FIRST (LDS-DMA), then the weights of step + 2.  The order is
    // pinned with sched_barrier (left alone, the machine scheduler pulls every ds_read back to just before its MFMA and
    // groups the requests).  Waits are counted: at slot NQ - L "all but the NLB newest requests" = this step's DMA has
    // landed (the weights just requested stay in flight for another step), then the block barrier: every wave's part of
    // the next stage is in LDS and every wave has finished reading the current stage.
    constexpr int NL = NLB + PIECES;
    constexpr int NQI = NQ - L;                       // request slots
    constexpr int RPS = (NL + NQI - 1) / NQI;         // requests per slot
    static_assert((NST - 2) * PIECES + (NST - 1) * NLB + NL <= 63, "vmcnt is a 6-bit counter");
    auto do_step = [&](auto Jc, int step) __attribute__((always_inline)) {
        constexpr int J = decltype(Jc)::value;
        constexpr int stage = J % NST, stage_next = (J + 1) % NST, stage_fill = (J + NST - 1) % NST;
        const BStep& bc = ring[J];
        const BStep& bn = ring[(J + 1) % RD];
        BStep& bl = ring[(J + PD) % RD];
        // Past the end of the slice the requests repeat the last step (never consumed): the SGPR offset of a buffer
        // access is not range-checked, so "out of range reads zeros" cannot be relied on; and the last step re-requests
        // its own x tile into the idle stage, which keeps the counted waits identical for every step.
        const int lstep = step + PD < nsteps ? step + PD : nsteps - 1;
        const int xstep = step + NST - 1 < nsteps ? step + NST - 1 : nsteps - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int g = q / MI, mi = q % MI;
            accm[mi] = XO::mfma(af[q % L], bfrag[g & 1], accm[mi]);
            if (q == NQI && !(EXP & 1)) {
                // everything but the requests issued after the DMA of step + 1: (NST - 2) later DMAs, (NST - 1) weight sets
                wait_vm<(NST - 2) * PIECES + (NST - 1) * NLB>();
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's reads of the current stage are complete
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if (!(EXP & 2)) {
                if (g + 1 < NS) deq_piece(bc, g + 1, mi, bfrag[(g + 1) & 1]);
                else deq_piece(bn, 0, mi, bfrag[(g + 1) & 1]);  // the next step's first slice
            }
            if (!(EXP & 4)) {
                if (q + L < NQ) af[q % L] = read_frag(stage, q + L);
                else af[q % L] = read_frag(stage_next, q + L - NQ);
            }
#pragma unroll
            for (int it = q * RPS; it < (q + 1) * RPS && it < NL; ++it) {
                if (it < PIECES) { if (!(EXP & 8)) req_x(stage_fill, xstep, it); }
                else if (!(EXP & 16)) req_b(bl, lstep, it - PIECES);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // unrolled by the ring depth: every ring / stage index is static.  Nested ifs, not breaks: the loop has ONE exit, so the
    // accumulators reach the epilogue through one set of registers (with a break per step the int32 accumulators of the
    // int8 variant got a copy per exit edge and spilled)
    auto chain = [&](auto self, auto Jc, int s0) -> void {
        constexpr int J = decltype(Jc)::value;
        do_step(std::integral_constant<int, J % RD>{}, s0 + J);
        if constexpr (J == 0) {
            if (s0 == 0) stamp(2);
        }
        if constexpr (J + 1 < RD) {
            if (s0 + J + 1 < nsteps) self(self, std::integral_constant<int, J + 1>{}, s0);
        }
    };
    // (A loop of whole RD-step groups without exits + a conditional tail was tried: hipcc's waitcnt pass then stops draining the
    //  request queue at the first step of every group — vmcnt(1..2) instead of vmcnt(18) — but the kernel time did not move:
    //  cfgB 44.36 vs 44.37 us, profiles/r02 lab notes.)
    for (int s0 = 0; s0 < nsteps; s0 += RD) chain(chain, std::integral_constant<int, 0>{}, s0);
    // retire every outstanding request (the last step's run-ahead DMA) before the LDS is reused
    wait_vm<0>();
    stamp(3);
    f32x16 acc[MI];  // int32 sums -> fp32 (exact below 2^24)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][e] = (float)accm[mi][e];

    // ---- epilogue 1: add the two K halves (waves 4..7 hand their accumulators to waves 0..3 through LDS) ------------
    __syncthreads();
    if constexpr (KH == 2) {
        float* xch = (float*)smem;  // [cg][mi][e4][lane][4]
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                float* a = xch + (((cg * MI + mi) * 4 + e4) * 64 + lane) * 4;
                if (kh == 1) *(f32x4*)a = (f32x4){acc[mi][4 * e4], acc[mi][4 * e4 + 1], acc[mi][4 * e4 + 2], acc[mi][4 * e4 + 3]};
            }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const f32x4 v = *(const f32x4*)(xch + (((cg * MI + mi) * 4 + e4) * 64 + lane) * 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[mi][4 * e4 + t] += v[t];
                }
        }
    }

    stamp(4);
    constexpr int NOUT = BM * BN;
    const int64_t ncol0 = (int64_t)nt * BN;
    constexpr int PASS_ROWS = BM < C_ROWS ? BM : C_ROWS;
    unsigned* flag = (unsigned*)(smem + PASS_ROWS * C_PITCH * 4);
    constexpr int MIH = (KH == 2 && MI >= 2) ? MI / 2 : MI;  // row blocks per wave in the combine
    bool split_rows = false;                    // combine done: the two K-half waves of a column group share the rows
    // ---- epilogue 2 (K split over blocks): the partial tile travels in FRAGMENT order — slabs are private to this
    //      kernel, so nothing is transposed: waves 0..3 store their registers as 16-byte write-through rows (1 KiB per
    //      wave instruction), and only the last block to arrive goes on: ALL 8 of its waves load the slices' words back
    //      into the same lanes (wave (cg, kh) takes row blocks [kh MIH, kh MIH + MIH)), add them in slice order (run-to-run
    //      deterministic) into the accumulator registers, and hand the sums to the output stage.
    // ---- epilogue 2x (K split over blocks, round 3): REDUCE-SCATTER between the co-resident slices of a tile.  With the slab +
    //      ticket protocol below every block writes its whole partial tile and the last block to arrive reads all of them back
    //      alone (cfgB, 256 x 128 x 4 slices: 4 us of slab stores + 1.5 us ticket + ~9 us of one CU reading 512 KB while its
    //      three peers have left).  Here slice s keeps the row blocks mi = s (mod S) and sends each of the others straight to
    //      its owner's inbox as write-through 16-byte rows in fragment order; every block bumps one arrival counter per peer,
    //      waits until its own counter shows S - 1, pulls the S - 1 partial copies of ITS row blocks into LDS with LDS-DMA (all
    //      pieces in flight at once, no registers), adds them in slice order (bit-identical to the ticket path) and writes its
    //      rows of the output.  (S - 1) / S of the tile leaves and enters every block, all blocks work in parallel, nobody
    //      waits for a ticket round trip.  The wait is a spin: the planner only picks this mode when ALL blocks of the launch
    //      fit on the device at once (tiles x S <= CUs, one block per CU), so every peer is running or about to be dispatched.
    if constexpr (KH == 2 && MI >= 2) {
        if (p.splitk > 1 && p.combine == 1) {
            const int S = p.splitk, LS = 31 - __builtin_clz((unsigned)S);  // S = 2^LS divides MI
            const int OWB = MI >> LS;                                      // row blocks this block owns
            float* inbox = p.slabs + (int64_t)bid * S * NOUT;             // [owner][from][j][cg][e4][lane] float4
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(inbox, (short)0, S * NOUT * 4, 0x00020000);
            if (kh == 0) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int o = mi & (S - 1), j = mi >> LS;
                    if (o != slice) {
                        const int unit = (((o * S + slice) * OWB + j) * 4 + cg) * 4;  // in 1-KiB pieces
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const f32x4 v = {acc[mi][4 * e4], acc[mi][4 * e4 + 1], acc[mi][4 * e4 + 2], acc[mi][4 * e4 + 3]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, (unit + e4) * 1024 + lane * 16, 0, 16);  // sc1
                        }
                    }
                }
            }
            stamp(5);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have left
            __syncthreads();
            unsigned* cnt = p.counters + (int64_t)bid * S;    // one arrival counter per owner
            if (tid < S && tid != slice) __hip_atomic_fetch_add(cnt + tid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(cnt + slice, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(S - 1)) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1u << 26)) __builtin_trap();  // seconds: a peer that never runs (see the planner's residency rule)
                }
                __hip_atomic_store(cnt + slice, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every peer has arrived: leave it zero
            }
            __syncthreads();
            stamp(6);
            // the S - 1 partial copies of the owned row blocks -> LDS behind the output staging tile, as 1-KiB DMA pieces
            const int ct_bytes = (OWB * 32 * C_PITCH * 4 + 1023) & ~1023;
            const int per_from = OWB * 16;  // pieces one peer sent: [j][cg][e4]
            const int npieces = (S - 1) * per_from;
            const srd_t rsI = make_srd(inbox + (int64_t)slice * NOUT, (uint32_t)NOUT * 4u);
            const uint32_t gaddr = lds_addr_of(smem) + (uint32_t)ct_bytes;
            for (int q = wave; q < npieces; q += 8) {
                const int fi = q / per_from, rem = q - fi * per_from, f = fi + (fi >= slice ? 1 : 0);
                req_lds16_sc1(rsI, gaddr + (uint32_t)q * 1024u, (uint32_t)lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((f * per_from + rem) * 1024));
            }
            wait_vm<0>();
            __syncthreads();
            float* ct = (float*)smem;  // [OWB * 32][C_PITCH]
            if (kh == 0) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if ((mi & (S - 1)) != slice) continue;
                    const int j = mi >> LS;
                    f32x16 sum;
#pragma unroll
                    for (int e = 0; e < 16; ++e) sum[e] = 0.f;
                    for (int f = 0; f < S; ++f) {  // slice order: bit-identical to the ticket path's sum
                        if (f == slice) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) sum[e] += acc[mi][e];
                        } else {
                            const int fi = f - (f > slice ? 1 : 0);
                            const unsigned char* src = smem + ct_bytes + (((fi * OWB + j) * 4 + cg) * 4) * 1024 + lane * 16;
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) {
                                const f32x4 v = *(const f32x4*)(src + e4 * 1024);
#pragma unroll
                                for (int t = 0; t < 4; ++t) sum[4 * e4 + t] += v[t];
                            }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int r = j * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        ct[r * C_PITCH + cg * 32 + col] = sum[e];
                    }
                }
            }
            __syncthreads();
            const bool typed_out_x = p.epi.out_dt == TR::DT && (p.epi.meta_dt == TR::DT || p.epi.c_mode == 0 || p.epi.c_mode == 2);
            const int64_t ncol0x = (int64_t)nt * BN;
            for (int u = tid; u < OWB * 32 * (BN / 4); u += 512) {
                const int r = u / (BN / 4), c4 = (u % (BN / 4)) * 4;
                const int m = m0 + ((((r >> 5) << LS) + slice) << 5) + (r & 31);
                if (m < p.M) {
                    const f32x4 v = *(const f32x4*)(ct + r * C_PITCH + c4);
                    if (typed_out_x) store_out4_t<Tag>(p.epi, v, m, ncol0x + c4);
                    else store_out4_any(p.epi, v, m, ncol0x + c4);
                }
            }
            stamp(7);
            return;
        }
    }
    if (p.splitk > 1) {
        float* slab = p.slabs + ((int64_t)bid * p.splitk) * NOUT;  // wave-uniform base of this tile's slabs
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, (short)0, p.splitk * NOUT * 4, 0x00020000);
        const int frag0 = (cg * MI * 4 * 64 + lane) * 4;  // float index of (mi = 0, e4 = 0) of this lane inside a slab
        if (kh == 0) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const f32x4 v = {acc[mi][4 * e4], acc[mi][4 * e4 + 1], acc[mi][4 * e4 + 2], acc[mi][4 * e4 + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                            (slice * NOUT + frag0 + (mi * 4 + e4) * 256) * 4, 0, 16);  // sc1
                }
        }
        stamp(5);
        const bool last = splitk_arrive_is_last(p.counters + bid, p.splitk, flag);
        stamp(6);
        if (!last) return;
        auto gather = [&](auto BASEc) {  // row blocks [BASE, BASE + MIH): sum of all slices, in place
            constexpr int BASE = decltype(BASEc)::value;
#pragma unroll
            for (int j = 0; j < MIH; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[BASE + j][e] = 0.f;
            constexpr int CH = MIH > 2 ? 2 : MIH;  // row blocks per round trip (registers: 16 CH words in flight)
            for (int s = 0; s < p.splitk; ++s) {
#pragma unroll
                for (int c = 0; c < MIH; c += CH) {
                    u32x4 t[CH][4];
#pragma unroll
                    for (int j = 0; j < CH; ++j)
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4)
                            t[j][e4] = __builtin_amdgcn_raw_buffer_load_b128(
                                rs, (s * NOUT + frag0 + ((BASE + c + j) * 4 + e4) * 256) * 4, 0, 16);
#pragma unroll
                    for (int j = 0; j < CH; ++j)
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            const f32x4 v = __builtin_bit_cast(f32x4, t[j][e4]);
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt) acc[BASE + c + j][4 * e4 + tt] += v[tt];
                        }
                }
            }
        };
        if (KH == 2 && MI >= 2) {
            split_rows = true;
            if (kh == 0) gather(std::integral_constant<int, 0>{});
            else gather(std::integral_constant<int, ((KH == 2 && MI >= 2) ? MIH : 0)>{});
        } else if (kh == 0) {
            gather(std::integral_constant<int, 0>{});
        }
        if (tid == 0) splitk_reset(p.counters + bid);
    }

    // ---- epilogue 3: the complete tile is transposed through LDS, 128 rows per pass, so that the output moves as 16-byte
    //      row segments; C fragment of a 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    float* ct = (float*)smem;  // [PASS_ROWS][C_PITCH]
    const bool typed_out = p.epi.out_dt == TR::DT && (p.epi.meta_dt == TR::DT || p.epi.c_mode == 0 || p.epi.c_mode == 2);
    constexpr int NPASS = BM / PASS_ROWS, MIP = PASS_ROWS / 32;
    constexpr int UNITS = (PASS_ROWS * BN / 4 + 511) / 512;  // float4 units per thread and pass
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        __syncthreads();  // the exchange buffer / the previous pass is no longer read
#pragma unroll
        for (int mi = 0; mi < MIP; ++mi) {
            const int blk = ps * MIP + mi;  // row block of the tile (static)
            const bool mine = split_rows ? ((blk >= MIH) == (kh == 1)) : (kh == 0);
            if (mine) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    ct[r * C_PITCH + cg * 32 + col] = acc[blk][e];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = tid + 512 * i, r = u / (BN / 4), c4 = (u % (BN / 4)) * 4;
            const int m = m0 + ps * PASS_ROWS + r;
            if (r < PASS_ROWS && m < p.M) {
                const f32x4 v = *(const f32x4*)(ct + r * C_PITCH + c4);
                if (typed_out) store_out4_t<Tag>(p.epi, v, m, ncol0 + c4);
                else store_out4_any(p.epi, v, m, ncol0 + c4);  // output / channel-scale dtype differs from Tag
            }
        }
    }
    stamp(7);
}

// ---------------------------------------------------------------------------------------------------------------
// host-side planning.  tuning[1]: 0 auto | n force split-K n;  tuning[2]: 0 auto | 1/2/4/8 force MI (tile rows / 32)
// ---------------------------------------------------------------------------------------------------------------
typedef void (*mma_kernel_fn)(const WnParams);
template <typename Tag, int NBITS, int XDT>
static const void* mma_pick_mi(int mi) {
    mma_kernel_fn f = nullptr;  // typed pointer first: a direct cast of the specialisation to void* does not instantiate the host stub
    switch (mi) {
        case 8: f = gemm_wn_mma_kernel<Tag, NBITS, 8, 128, 4, 2, 0, XDT>; break;
        case 4: f = gemm_wn_mma_kernel<Tag, NBITS, 4, 128, 6, 3, 0, XDT>; break;
        // 8 dwords per lane and sub-block (8-bit words, fp8 MX rows): a shallower ring keeps the registers in budget
        case 2: f = gemm_wn_mma_kernel<Tag, NBITS, 2, 256, 6, (mma::Geo<NBITS>::WPL == 8 ? 2 : 3), 0, XDT>; break;
        case 1: f = gemm_wn_mma_kernel<Tag, NBITS, 1, 256, (mma::Geo<NBITS>::WPL == 8 ? 6 : 8), (mma::Geo<NBITS>::WPL == 8 ? 2 : 4), 0, XDT>; break;
        default: break;
    }
    return (const void*)f;
}
template <typename Tag>
static const void* mma_pick(int nbits, int mi, int xdt) {
    if (xdt == GEMLITE_DT_FP8E4) {  // 8-bit activations: the bit widths the reference's A8Wn / BitNet processors produce
        if (nbits == 4) return mma_pick_mi<Tag, 4, GEMLITE_DT_FP8E4>(mi);
        if (nbits == 2) return mma_pick_mi<Tag, 2, GEMLITE_DT_FP8E4>(mi);
        return nullptr;
    }
    if (xdt == GEMLITE_DT_INT8) {
        if (nbits == 4) return mma_pick_mi<Tag, 4, GEMLITE_DT_INT8>(mi);
        if (nbits == 2) return mma_pick_mi<Tag, 2, GEMLITE_DT_INT8>(mi);
        return nullptr;
    }
    switch (nbits) {
        case 4: return mma_pick_mi<Tag, 4, 0>(mi);
        case 2: return mma_pick_mi<Tag, 2, 0>(mi);
        case 1: return mma_pick_mi<Tag, 1, 0>(mi);
        case 8: return mma_pick_mi<Tag, 8, 0>(mi);
        default: return nullptr;
    }
}

// wide tiles (32 MI x 256, KH = 1, 64-k steps): 16-bit activations x 4- / 2-bit words, 128 or 256 rows
template <typename Tag>
static const void* mma_pick_wide(int nbits, int mi) {
    mma_kernel_fn f = nullptr;
    if (nbits == 4) f = mi == 8 ? gemm_wn_mma_kernel<Tag, 4, 8, 64, 4, 2, 0, 0, 1> : gemm_wn_mma_kernel<Tag, 4, 4, 64, 6, 3, 0, 0, 1>;
    else if (nbits == 2) f = mi == 8 ? gemm_wn_mma_kernel<Tag, 2, 8, 64, 4, 2, 0, 0, 1> : gemm_wn_mma_kernel<Tag, 2, 4, 64, 6, 3, 0, 0, 1>;
    return (const void*)f;
}

// 16-bit activations x block-scaled weights (layer formats MXFP16 / MXBF16: A16W8_MXFP, A16W4_MXFP): the same kernel with the
// K-contiguous weight geometry (Geo<MXW8 / MXW4>) — the weights are converted by v_cvt_scalef32_pk_* with their block scale, 4
// VALU per fragment instead of 23.  `p` arrives with x / w / scales / epilogue / M / N / K / strides filled in by the caller.
// tuning[1] = K slices, tuning[2] = tile rows / 32.
bool plan_gemm_wn_mma_mx(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    const bool f16 = a.input_dtype == GEMLITE_DT_MXFP16;
    if (!f16 && a.input_dtype != GEMLITE_DT_MXBF16) return false;
    if (a.output_dtype != (f16 ? GEMLITE_DT_FP16 : GEMLITE_DT_BF16)) return false;  // typed epilogue
    const int nb = a.W_nbits == 8 ? mma::MXW8 : mma::MXW4;
    if (a.group_size != 32 || a.stride_wk != 1 || a.stride_xk != 1 || a.stride_on != 1 || a.N % mma::BN != 0 || a.K % 128 != 0) return false;
    if ((a.stride_xm * 2) % 16 != 0 || ((uintptr_t)a.x % 16) != 0 || ((uintptr_t)a.w_q % 16) != 0 || a.stride_wn % 16 != 0) return false;
    if (((uintptr_t)a.out % 8) != 0 || (a.stride_om * 2) % 8 != 0) return false;
    if ((int64_t)a.N * a.stride_wn >= (1ll << 31) || ((int64_t)a.M * a.stride_xm + a.K) * 2 >= (1ll << 31)) return false;
    if ((int64_t)(a.K / 32) * a.stride_meta_g + (int64_t)a.N * a.stride_meta_n >= (1ll << 31)) return false;
    auto kstep_of = [](int c) { return c >= 4 ? 128 : 256; };
    const int cap = a.M > 128 ? 8 : (a.M > 64 ? 4 : (a.M > 32 ? 2 : 1));
    // cheap conversion: the tallest tile that still gives >= 128 tiles (else >= 64, else the tallest M fills), then the
    // fewest K slices that give >= 224 blocks while a slice keeps >= 4 steps (the rule of the other non-4-bit widths)
    int mi = cap;
    for (int want = 128; want >= 64; want >>= 1) {
        int found = 0;
        for (int c = cap; c >= 2 && !found; c >>= 1)
            if (a.K % kstep_of(c) == 0 && (a.N / mma::BN) * ((a.M + 32 * c - 1) / (32 * c)) >= want) found = c;
        if (found) { mi = found; break; }
    }
    if (a.tuning[2] == 1 || a.tuning[2] == 2 || a.tuning[2] == 4 || a.tuning[2] == 8) mi = a.tuning[2];
    else if (a.tuning[2] != 0) return false;
    if (a.K % kstep_of(mi) != 0) {
        if (a.tuning[2] != 0) return false;
        mi = mi < 4 ? 4 : mi;
    }
    const int ks = kstep_of(mi), bm = 32 * mi;
    const int units = (int)(a.K / ks);
    const int64_t tiles = (int64_t)(a.N / mma::BN) * ((a.M + bm - 1) / bm);
    int splitk = 0;
    if (a.tuning[1] > 0) splitk = a.tuning[1];
    else {
        for (int sk = 1; sk <= units && sk <= 32; ++sk) {
            if (sk > 1 && units / sk < 4) continue;
            splitk = sk;
            if (tiles * sk >= 224) break;
        }
        if (!splitk) splitk = 1;
    }
    if (splitk > units) return false;
    if (splitk > 1 && tiles > MAX_SPLITK_COUNTERS) return false;
    if ((uint64_t)splitk * bm * mma::BN * 4 >= (1ull << 31)) return false;
    const void* fn = f16 ? (nb == mma::MXW8 ? mma_pick_mi<half_tag, mma::MXW8, 0>(mi) : mma_pick_mi<half_tag, mma::MXW4, 0>(mi))
                         : (nb == mma::MXW8 ? mma_pick_mi<bf16_tag, mma::MXW8, 0>(mi) : mma_pick_mi<bf16_tag, mma::MXW4, 0>(mi));
    if (!fn) return false;
    p.splitk = splitk;
    p.rows_per_slice = (int)a.K;  // E = 1: "packed rows" are k
    p.stride_wn_b = a.stride_wn;  // 1-byte elements
    p.stride_meta_n = a.stride_meta_n;
    p.stride_meta_g = a.stride_meta_g;
    p.w_mode = 2;
    p.gs_shift = 5;
    lp.fn = fn;
    static const char* names[2][4] = {
        {"gemm_a16w8_mxfp_kernel<32x128>", "gemm_a16w8_mxfp_kernel<64x128>", "gemm_a16w8_mxfp_kernel<128x128>", "gemm_a16w8_mxfp_kernel<256x128>"},
        {"gemm_a16w4_mxfp_kernel<32x128>", "gemm_a16w4_mxfp_kernel<64x128>", "gemm_a16w4_mxfp_kernel<128x128>", "gemm_a16w4_mxfp_kernel<256x128>"}};
    lp.name = names[nb == mma::MXW8 ? 0 : 1][mi == 1 ? 0 : (mi == 2 ? 1 : (mi == 4 ? 2 : 3))];
    lp.grid = dim3((unsigned)tiles, splitk, 1);
    lp.block = dim3(512, 1, 1);
    const bool w8 = nb == mma::MXW8;
    const int nst = mi == 8 ? 2 : (mi == 4 ? 3 : (w8 ? 2 : (mi == 2 ? 3 : 4)));  // LDS stages of x (mma_pick_mi)
    const size_t stages = (size_t)nst * bm * ks * 2;
    const size_t xch = (size_t)4 * mi * 4 * 64 * 16;
    const size_t c_b = (size_t)(bm < mma::C_ROWS ? bm : mma::C_ROWS) * mma::C_PITCH * 4 + 16;
    lp.lds_bytes = stages > xch ? stages : xch;
    if (lp.lds_bytes < c_b) lp.lds_bytes = c_b;
    lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * bm * mma::BN * 4 : 0;
    lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

bool plan_gemm_wn_mma(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    const int nbits = a.W_nbits;
    if (nbits != 4 && nbits != 2 && nbits != 1 && nbits != 8) return false;
    const int e = 32 / nbits;
    if (a.N % mma::BN != 0) return false;
    // Activation type: the 16-bit float of the kernel's Tag, or 8 bits (fp8 e4m3 / int8: A8Wn dynamic, BitNet int8) with a
    // 16-bit output.  Tag = the type of the metadata read in the K loop = the output type (any of fp16 / bf16 / fp32 output
    // and fp32 channel scales go through the untyped epilogue store).
    const bool x16 = a.input_dtype == GEMLITE_DT_FP16 || a.input_dtype == GEMLITE_DT_BF16;
    const int xdt = x16 ? 0 : a.input_dtype;
    if (!x16 && a.input_dtype != GEMLITE_DT_FP8E4 && a.input_dtype != GEMLITE_DT_INT8) return false;
    if (!x16 && a.output_dtype != GEMLITE_DT_FP16 && a.output_dtype != GEMLITE_DT_BF16) return false;
    const int tag_dt = x16 ? a.input_dtype : a.output_dtype;
    const bool loop_s = a.W_group_mode >= 2, post_s = a.channel_scale_mode == 1 || a.channel_scale_mode == 3;
    const bool has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (loop_s && a.meta_dtype != tag_dt) return false;
    if (post_s && !loop_s && a.meta_dtype != GEMLITE_DT_FP32 && a.meta_dtype != GEMLITE_DT_FP16 && a.meta_dtype != GEMLITE_DT_BF16) return false;
    if (post_s && ((uintptr_t)a.scales % (a.meta_dtype == GEMLITE_DT_FP32 ? 16 : 8)) != 0) return false;  // vector loads of the channel scales
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != tag_dt) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    if (xdt == GEMLITE_DT_INT8 && (a.W_group_mode >= 2 || (has_z && !a.zero_is_scalar))) return false;  // integer codes only
    const int es = x16 ? 2 : 1;
    if ((a.stride_xm * es) % 16 != 0 || ((uintptr_t)a.x % 16) != 0) return false;  // 16-byte LDS-DMA pieces
    const int oal = a.output_dtype == GEMLITE_DT_FP32 ? 16 : 8;  // 4 outputs per store
    if (((uintptr_t)a.out % oal) != 0 || (a.stride_om * (oal / 4)) % oal != 0) return false;
    if (p.group_size % 64 != 0) return false;  // one (scale, zero) pair per column and 64-k sub-block
    // Tile rows (32 MI) and K slices.  A dequantised fragment feeds MI MFMAs, so tall tiles need the least unpack arithmetic
    // per MFMA; short tiles and K slices fill the 256 CUs (one 8-wave block each) — but every K slice costs slab traffic
    // through memory (the XCD L2s are not coherent) plus a serial tail in the last block to arrive, and narrow row tiles
    // re-read x from L2 once per column tile.  The choice is the minimum of a block-time model fitted to 66 shapes x 24
    // (MI, slices) pairs measured with HBM-cold weights (scripts/make_tuning_table.py, profiles/r02/autotune_report.json;
    // mean regret against the measured best 0.9 %, worst 18 %; the rule it replaces: 13 % / 90 %):
    //   us = rounds * (P + steps * S + [slices > 1] * (Q + slices * R)) + 0.096 * slab_MB + 0.048 * x_through_L2_MB
    // with rounds = ceil(blocks / 256) and steps = the K steps of the longest slice.
    const int cap = a.M > 128 ? 8 : (a.M > 64 ? 4 : (a.M > 32 ? 2 : 1));
    auto kstep_of = [](int c) { return c >= 4 ? 128 : 256; };
    auto cost_us = [&](int c, int sk) -> double {
        static const double P[4] = {1.46, 1.21, 1.67, 2.82}, S[4] = {0.945, 1.061, 0.695, 1.247};
        static const double Q[4] = {1.78, 1.20, 1.51, 1.09}, R[4] = {0.0, 0.222, 0.283, 0.516};
        const int i = c == 1 ? 0 : (c == 2 ? 1 : (c == 4 ? 2 : 3));
        const int un = (int)(a.K / kstep_of(c));
        const int steps = (un + sk - 1) / sk;
        const int64_t mpad = (a.M + 32 * c - 1) / (32 * c) * (32 * c);
        const int64_t blocks = (a.N / mma::BN) * (mpad / (32 * c)) * sk;
        const double rounds = (double)((blocks + 255) / 256);
        const double slab_mb = sk > 1 ? (double)sk * mpad * a.N * 8e-6 : 0.0;
        const double x_mb = (double)(a.N / mma::BN) * mpad * a.K * 2e-6;
        return rounds * (P[i] + steps * S[i] + (sk > 1 ? Q[i] + sk * R[i] : 0.0)) + 0.0955 * slab_mb + 0.0484 * x_mb;
    };
    int mi = 0, splitk = 0;
    // wide tiles (32 MI x 256): tuning[2] = 16 + MI (20 / 24) forces them
    const bool wide_ok = x16 && (nbits == 4 || nbits == 2) && a.N % 256 == 0 && a.K % 64 == 0;
    bool wide = (a.tuning[2] == 20 || a.tuning[2] == 24);
    if (wide) {
        if (!wide_ok) return false;
        mi = a.tuning[2] - 16;
        const int un = (int)(a.K / 64);
        const int64_t tl = (int64_t)(a.N / 256) * ((a.M + 32 * mi - 1) / (32 * mi));
        if (a.tuning[1] > 0) splitk = a.tuning[1];
        else {
            for (int sk = 1; sk <= un && sk <= 16; ++sk) {
                if (sk > 1 && un / sk < 8) continue;
                splitk = sk;
                if (tl * sk >= 224) break;
            }
            if (!splitk) splitk = 1;
        }
    } else if (nbits != 4) {
        // other bit widths (not swept; their unpack arithmetic per weight differs): the rule of thumb the 4-bit model
        // replaced, measured on config 5 (A16W2 16384^2, M = 256: 256-row tiles x 2 slices, 137 us) — the tallest tile
        // (>= 64 rows) that still yields >= 128 tiles, else >= 64, else the tallest M fills; then the fewest K slices that
        // give >= 224 blocks while a slice keeps >= 4 steps.
        mi = cap;
        for (int want = 128; want >= 64; want >>= 1) {
            int found = 0;
            for (int c = cap; c >= 2 && !found; c >>= 1)
                if (a.K % kstep_of(c) == 0 && (a.N / mma::BN) * ((a.M + 32 * c - 1) / (32 * c)) >= want) found = c;
            if (found) { mi = found; break; }
        }
        if (a.tuning[2] == 1 || a.tuning[2] == 2 || a.tuning[2] == 4 || a.tuning[2] == 8) mi = a.tuning[2];
        else if (a.tuning[2] != 0) return false;
        if (a.K % kstep_of(mi) != 0) {
            if (a.K % 128 != 0 || a.tuning[2] != 0) return false;
            mi = mi < 4 ? 4 : mi;  // K = 128 * odd: only the 128-k-step variants apply
        }
        const int un = (int)(a.K / kstep_of(mi));
        const int64_t tl = (int64_t)(a.N / mma::BN) * ((a.M + 32 * mi - 1) / (32 * mi));
        if (a.tuning[1] > 0) splitk = a.tuning[1];
        else {
            for (int sk = 1; sk <= un && sk <= 32; ++sk) {
                if (un / sk < 4) continue;
                splitk = sk;
                if (tl * sk >= 224) break;
            }
            if (!splitk) splitk = 1;
        }
    } else {
        double best = 1e30;
        // second pass: K = 128 * odd (896, 640, ...) divides only the 128-k steps of the 128- / 256-row tiles — those then
        // apply whatever M is (rows >= M are never loaded nor stored)
        for (int pass = 0; pass < 2 && !mi; ++pass)
            for (int c = 1; c <= 8; c <<= 1) {
                if (a.tuning[2] != 0 ? c != a.tuning[2] : (pass == 0 ? c > cap : c != 4)) continue;
                if (a.K % kstep_of(c) != 0) continue;
                const int un = (int)(a.K / kstep_of(c));
                for (int sk = 1; sk <= un && sk <= 16; ++sk) {
                    if (a.tuning[1] > 0 ? sk != a.tuning[1] : (sk > 1 && un / sk < 2)) continue;
                    const double t = cost_us(c, sk);
                    if (t < best) { best = t; mi = c; splitk = sk; }
                }
            }
        if (!mi) return false;  // no tile variant divides K (K % 128 != 0), or an override that does not apply
    }
    // automatic choice of the wide tile: when 256 x 256 tiles alone fill the chip (prefill)
    if (!wide && wide_ok && a.tuning[1] == 0 && a.tuning[2] == 0 && (int64_t)(a.N / 256) * ((a.M + 255) / 256) >= 192) {
        wide = true;
        mi = 8;
        splitk = 1;
    }
    const int bn = wide ? 256 : mma::BN;
    const int ks = wide ? 64 : kstep_of(mi);
    const int bm = 32 * mi;
    const int rows = (int)(a.K / e), step_rows = ks / e;
    const int units = rows / step_rows;
    if (splitk > units) return false;
    // buffer descriptors: 32-bit byte offsets
    if ((int64_t)rows * a.stride_wk * 4 >= (1ll << 31) || ((int64_t)a.M * a.stride_xm + a.K) * es >= (1ll << 31)) return false;
    if (((int64_t)(a.K / (p.group_size > 0 ? p.group_size : a.K)) * p.stride_meta_g + a.N) * 2 >= (1ll << 31)) return false;
    const int64_t tiles = (int64_t)(a.N / bn) * ((a.M + bm - 1) / bm);
    if (splitk > 1 && tiles > MAX_SPLITK_COUNTERS) return false;
    if ((uint64_t)splitk * bm * bn * 4 >= (1ull << 31)) return false;  // slab buffer descriptor range
    const bool f16 = tag_dt == GEMLITE_DT_FP16;
    const void* fn = wide ? (f16 ? mma_pick_wide<half_tag>(nbits, mi) : mma_pick_wide<bf16_tag>(nbits, mi))
                          : (f16 ? mma_pick<half_tag>(nbits, mi, xdt) : mma_pick<bf16_tag>(nbits, mi, xdt));
#ifdef GL_MMA_EXPERIMENTS
    if (!wide && !f16 && xdt == 0 && nbits == 4 && (mi == 4 || mi == 8)) {
        mma_kernel_fn f = nullptr;
#define GL_EXP_CASE(E) case E: f = mi == 4 ? gemm_wn_mma_kernel<bf16_tag, 4, 4, 128, 6, 3, E> : gemm_wn_mma_kernel<bf16_tag, 4, 8, 128, 4, 2, E>; break;
        switch (a.tuning[3] >> 8) { GL_EXP_CASE(1) GL_EXP_CASE(2) GL_EXP_CASE(4) GL_EXP_CASE(8) GL_EXP_CASE(16) GL_EXP_CASE(3) GL_EXP_CASE(6) GL_EXP_CASE(7) GL_EXP_CASE(31) GL_EXP_CASE(32) GL_EXP_CASE(34) GL_EXP_CASE(63) default: break; }
#undef GL_EXP_CASE
        if (f) fn = (const void*)f;
    }
#endif
    if (!fn) return false;
    p.splitk = splitk;
    p.rows_per_slice = rows;  // ALL packed rows: the kernel derives each slice's step range itself
    // K-slice combine: reduce-scatter between the slices of a tile when they are certain to be co-resident (every block of the
    // launch fits on the device at once: <= one block per CU) and the slices divide the tile's row blocks; else slabs + ticket.
    // tuning[3] & 128 forces the ticket protocol (A/B runs, tests of both paths).
    p.combine = (!wide && splitk > 1 && (splitk & (splitk - 1)) == 0 && mi >= 2 && splitk <= mi && !(a.tuning[3] & 128) &&
                 tiles * splitk <= resident_block_limit()) ? 1 : 0;
    lp.fn = fn;
    static const char* names[4][4] = {
        {"gemm_w4_mma_kernel<32x128>", "gemm_w4_mma_kernel<64x128>", "gemm_w4_mma_kernel<128x128>", "gemm_w4_mma_kernel<256x128>"},
        {"gemm_w2_mma_kernel<32x128>", "gemm_w2_mma_kernel<64x128>", "gemm_w2_mma_kernel<128x128>", "gemm_w2_mma_kernel<256x128>"},
        {"gemm_w1_mma_kernel<32x128>", "gemm_w1_mma_kernel<64x128>", "gemm_w1_mma_kernel<128x128>", "gemm_w1_mma_kernel<256x128>"},
        {"gemm_w8_mma_kernel<32x128>", "gemm_w8_mma_kernel<64x128>", "gemm_w8_mma_kernel<128x128>", "gemm_w8_mma_kernel<256x128>"}};
    static const char* names8[2][4] = {
        {"gemm_a8w4_mma_kernel<32x128>", "gemm_a8w4_mma_kernel<64x128>", "gemm_a8w4_mma_kernel<128x128>", "gemm_a8w4_mma_kernel<256x128>"},
        {"gemm_a8w2_mma_kernel<32x128>", "gemm_a8w2_mma_kernel<64x128>", "gemm_a8w2_mma_kernel<128x128>", "gemm_a8w2_mma_kernel<256x128>"}};
    const int mix = mi == 1 ? 0 : (mi == 2 ? 1 : (mi == 4 ? 2 : 3));
    lp.name = xdt ? names8[nbits == 4 ? 0 : 1][mix] : names[nbits == 4 ? 0 : (nbits == 2 ? 1 : (nbits == 1 ? 2 : 3))][mix];
    static const char* names_wide[2][2] = {{"gemm_w4_mma_kernel<128x256>", "gemm_w4_mma_kernel<256x256>"},
                                           {"gemm_w2_mma_kernel<128x256>", "gemm_w2_mma_kernel<256x256>"}};
    if (wide) lp.name = names_wide[nbits == 4 ? 0 : 1][mi == 8 ? 1 : 0];
    lp.grid = dim3((unsigned)tiles, splitk, 1);
    lp.block = dim3(512, 1, 1);
    const int nst = mi == 8 ? 2 : (mi == 4 ? 3 : (nbits == 8 ? 2 : (mi == 2 ? 3 : 4)));  // LDS stages of x (mma_pick_mi)
    const size_t stages = (size_t)nst * bm * ks * es;
    const size_t xch = (size_t)4 * mi * 4 * 64 * 16;  // K-half exchange: [cg][mi][e4][lane] float4
    const size_t c_b = (size_t)(bm < mma::C_ROWS ? bm : mma::C_ROWS) * (bn + 4) * 4 + 16;
    lp.lds_bytes = stages > xch ? stages : xch;
    if (lp.lds_bytes < c_b) lp.lds_bytes = c_b;
    if (p.combine == 1) {  // output staging tile of the owned rows + the S - 1 inbound copies of them
        const size_t owb = (size_t)mi / splitk;
        const size_t ex = ((owb * 32 * (bn + 4) * 4 + 1023) & ~(size_t)1023) + (size_t)(splitk - 1) * owb * 16 * 1024;
        if (lp.lds_bytes < ex) lp.lds_bytes = ex;
    }
    lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * bm * bn * 4 : 0;
    lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

}  // namespace gl
