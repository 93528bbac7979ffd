// gemm_wn_tiled.hip — fused unpack + dequant + tiled MFMA GEMM for packed low-bit weights, large M (prefill).
// Replaces gemm_INT_kernel (gemlite/triton_kernels/gemm_kernels.py:248-413).
//
// Why this shape (CDNA4, measured costs in DESIGN.md): every VALU instruction costs ~4 cycles per wave and
// v_mfma_f32_32x32x16 32 cycles per SIMD, so a weight must be dequantised ONCE per block and reused by many
// MFMAs.  Block = 4 waves, tile 128 (M) x 128 (N); wave w owns all 128 rows x columns [32w, 32w+32):
//   * B: lane (col = lane&31, kb = lane>>5) of a 32x32x16 MFMA holds 8 consecutive k of ONE column = one packed
//     int32 word (4-bit).  The wave therefore loads its B fragments straight from HBM/L2 with one dword per lane
//     (2 packed rows x 128 contiguous bytes per instruction), dequantises them in registers exactly like the
//     reference (q exact -> fma / sub / mul in fp16, triton_kernels/utils.py:73-87; bf16: one fp32 fma rounded
//     once) and feeds 4 MFMAs (the 4 row blocks) with each fragment: no LDS traffic and no redundancy for B.
//   * A (x): 128 x 64 tile per K step through LDS, double buffered, stored pair-permuted (the k order the AND/OR
//     unpack produces) and XOR-swizzled on 16-byte slots so that ds_read_b128 of 32 rows is conflict-free.
//   * K is optionally split over gridDim.y (128-column tiles alone rarely fill 256 CUs at M = 256); slices are
//     combined with the write-through slab + ticket protocol (gl_common.h).
#include "gl_common.h"

namespace gl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// two dequantised weights (packed 16-bit floats) from one magic-number pair
template <typename Tag>
struct Deq2;
template <>
struct Deq2<half_tag> {
    // branch-free form of triton_kernels/utils.py:73-87:  w = fma(q - zsub, s, zadd)  with
    //   mode 0: (0, 1, 0) | 1: (z, 1, 0) | 2: (0, s, 0) | 3: (z, s, 0) | 4: (0, s, z')
    // q - 0 and t * 1 + 0 are exact, so every mode rounds exactly where the reference rounds.
    h2_t zsub2, s2, zadd2;
    __device__ __forceinline__ void set(float s, float z, int w_mode) {
        const float zs = (w_mode == 1 || w_mode == 3) ? z : 0.f;
        const float sc = w_mode >= 2 ? s : 1.f;
        const float za = w_mode == 4 ? z : 0.f;
        zsub2 = (h2_t){(_Float16)zs, (_Float16)zs};
        s2 = (h2_t){(_Float16)sc, (_Float16)sc};
        zadd2 = (h2_t){(_Float16)za, (_Float16)za};
    }
    __device__ __forceinline__ uint32_t apply(uint32_t h, int) const {
        h2_t q = __builtin_bit_cast(h2_t, h) - (h2_t){(_Float16)1024.0f, (_Float16)1024.0f};  // exact integer
        q = __builtin_elementwise_fma(q - zsub2, s2, zadd2);
        return __builtin_bit_cast(uint32_t, q);
    }
};
template <>
struct Deq2<bf16_tag> {
    float A, B;  // w = fma(128 + q, A, B), the 128 offset folded into B
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

constexpr int TBN = 128, TBK = 64;
constexpr int C_ROWS = 128;                 // rows of the epilogue staging tile (processed per 128-row half)
constexpr int C_PITCH = TBN + 4;            // floats per row of that tile (16-byte aligned, conflict-free)

// byte offset of the 16-byte slot (row r, slot s of 8) inside an A buffer: XOR swizzle on (r >> 1)
__device__ __forceinline__ int a_slot(int r, int s) { return r * (TBK * 2) + ((s ^ ((r >> 1) & 7)) << 4); }

// ---- epilogue shared by the tiled kernels.  C fragment of a 32x32 MFMA: col = lane & 31,
// row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
template <typename Tag, int MI>
__device__ __forceinline__ void tiled_epilogue(const WnParams& p, f32x16 (&acc)[MI], unsigned char* smem, int bid, int nt,
                                               int m0, int slice) {
    constexpr int TBM = 32 * MI;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kb = lane >> 5;
    // The tile is transposed through LDS (128 rows at a time) so that slabs and the output move as 16-byte row
    // segments (4-byte write-through stores are ~10x slower: MI355X_MICROARCH.md "stores of each flavour").
    float* ct = (float*)smem;  // [C_ROWS][C_PITCH]
    unsigned* flag = (unsigned*)(smem + C_ROWS * C_PITCH * 4);
    const int tile_lin = bid;
    constexpr int UNITS = C_ROWS * TBN / 4 / 256;  // float4 units per thread and 128-row half
    constexpr int NOUT = TBM * TBN;
    const int64_t ncol0 = (int64_t)nt * TBN;
    float* slab = p.slabs + ((int64_t)tile_lin * p.splitk) * NOUT;  // wave-uniform base of this tile's slabs
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, (short)0, p.splitk * NOUT * 4, 0x00020000);
#pragma unroll
    for (int half = 0; half < MI / 4; ++half) {
        __syncthreads();  // A buffers (or the previous half) are no longer read
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kb;
                ct[r * C_PITCH + wave * 32 + col] = acc[half * 4 + mi][e];
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = tid + 256 * i, r = u >> 5, c4 = (u & 31) * 4;
            const int m = m0 + half * C_ROWS + r;
            if (m < p.M) {
                const f32x4 v = *(const f32x4*)(ct + r * C_PITCH + c4);
                if (p.splitk == 1) store_out4_t<Tag>(p.epi, v, m, ncol0 + c4);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                            (slice * NOUT + (half * C_ROWS + r) * TBN + c4) * 4, 0, 16);  // sc1
            }
        }
    }
    if (p.splitk == 1) return;
    __syncthreads();
    if (!splitk_arrive_is_last(p.counters + tile_lin, p.splitk, flag)) return;
    // last arriver: slices outer, units inner -> every slice's 16-byte loads are in flight together.  (Requesting the
    // next slice before adding the current one was measured SLOWER — 34.3 vs 23.0 us at 4096 x 4096, M = 256 — the
    // 64 extra registers cost more than the overlapped round trip gains.)
    for (int half = 0; half < MI / 4; ++half) {
        f32x4 sum[UNITS];
#pragma unroll
        for (int i = 0; i < UNITS; ++i) sum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < p.splitk; ++s) {
            u32x4 t[UNITS];
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                const int u = tid + 256 * i, r = u >> 5, c4 = (u & 31) * 4;
                t[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (s * NOUT + (half * C_ROWS + r) * TBN + c4) * 4, 0, 16);
            }
#pragma unroll
            for (int i = 0; i < UNITS; ++i) sum[i] += __builtin_bit_cast(f32x4, t[i]);
        }
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = tid + 256 * i, r = u >> 5, c4 = (u & 31) * 4;
            const int m = m0 + half * C_ROWS + r;
            if (m < p.M) store_out4_t<Tag>(p.epi, sum[i], m, ncol0 + c4);
        }
    }
    if (tid == 0) splitk_reset(p.counters + tile_lin);
}

#ifdef GL_AB_KERNELS  // the one-step-ahead kernel of round 1: only built for A/B runs (make AB=1)
// MI = 32-row blocks per wave: tile = (32*MI) x 128, 4 waves, wave w owns all rows x columns [32w, 32w+32)
template <typename Tag, int MI>
__global__ __launch_bounds__(256, (MI <= 4 ? 2 : 1)) void gemm_w4_tiled_kernel(const WnParams p) {
    using TR = F16Traits<Tag>;
    constexpr int TBM = 32 * MI;
    constexpr int A_BUF_BYTES = TBM * TBK * 2;
    constexpr int SLOTS = TBM * 8 / 256;  // 16-byte A slots staged per thread and K step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][A_BUF_BYTES], later the C tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kb = lane >> 5;
    // tile order: consecutive blocks share the same column tile (B reuse in L2), M tiles fastest
    const int mtiles = (p.M + TBM - 1) / TBM;
    const int bid = blockIdx.x;
    const int mt = bid % mtiles, nt = bid / mtiles;
    const int slice = blockIdx.y;
    const int m0 = mt * TBM;
    const int n = nt * TBN + wave * 32 + col;  // this lane's column

    const int ksteps = p.rows_per_slice / (TBK / 8);       // K steps of 64 in this slice (rows_per_slice packed rows)
    const int row_s0 = slice * p.rows_per_slice;           // first packed row of the slice
    const int64_t k_s0 = (int64_t)row_s0 * 8;

    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    const uint16_t* sp = need_s ? (const uint16_t*)p.scales : (const uint16_t*)p.w;
    const uint16_t* zp = need_z ? (const uint16_t*)p.zeros : (const uint16_t*)p.w;
    const int64_t mstride = (need_s || need_z) ? p.stride_meta_g : 0;
    const float scalar_zero = p.zero_is_scalar ? (float)((const int32_t*)p.zeros)[0] : 0.f;

    // ---- B stream: 8 packed rows per K step; this lane needs rows 2*ks + kb (ks = 0..3) of its column ----------
    const uint32_t* wbase = p.w + (int64_t)(row_s0 + kb) * p.stride_wk + n;
    const int sw = (int)p.stride_wk, ms = (int)mstride;  // 32-bit index math in the loop
    const int kbase = (int)k_s0;
    struct BStep { uint32_t w[4]; uint16_t s, z; };
    auto load_b = [&](BStep& b, int step) {
        const uint32_t* wp = wbase + step * 8 * sw;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b.w[ks] = wp[2 * ks * sw];
        const int grp = group_of(kbase + step * TBK, p.gs_shift);
        b.s = sp[grp * ms + n];
        b.z = zp[grp * ms + n];
    };
    // ---- A stream: TBM rows x 64 k per step = TBM*8 16-byte slots, SLOTS per thread ---------------------------
    const uint16_t* xg = (const uint16_t*)p.x;
    struct AStep { u32x4 v[SLOTS]; };
    const uint16_t* xrow[SLOTS];  // this thread's staging rows (clamped: rows >= M are zeroed after the load)
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int u = tid + 256 * i, r = u >> 3, s = u & 7;
        const int rr = m0 + r < p.M ? m0 + r : p.M - 1;
        xrow[i] = xg + (int64_t)rr * p.stride_xm + k_s0 + s * 8;
    }
    auto load_a = [&](AStep& a, int step) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            a.v[i] = *(const u32x4*)(xrow[i] + step * TBK);
            if (m0 + ((tid + 256 * i) >> 3) >= p.M) a.v[i] = (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto put_a = [&](const AStep& a, int buf) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int u = tid + 256 * i, r = u >> 3, s = u & 7;
            // 8 halfs x0..x7 -> pairs (x0,x4)(x1,x5)(x2,x6)(x3,x7): the k order of the unpacked B fragment
            const uint32_t d0 = a.v[i][0], d1 = a.v[i][1], d2 = a.v[i][2], d3 = a.v[i][3];
            u32x4 o;
            o[0] = __builtin_amdgcn_perm(d2, d0, 0x05040100u);  // (x0, x4)
            o[1] = __builtin_amdgcn_perm(d2, d0, 0x07060302u);  // (x1, x5)
            o[2] = __builtin_amdgcn_perm(d3, d1, 0x05040100u);  // (x2, x6)
            o[3] = __builtin_amdgcn_perm(d3, d1, 0x07060302u);  // (x3, x7)
            *(u32x4*)(smem + buf * A_BUF_BYTES + a_slot(r, s)) = o;
        }
    };

    f32x16 acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    Deq2<Tag> dq;
    // A-fragment offsets: row = mi*32 + col, slot = ks*2 + kb; the swizzle term depends on col only
    int a_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_off[ks] = a_slot(col, ks * 2 + kb);
    auto compute = [&](const BStep& b, int buf) {
        const unsigned char* abase = smem + buf * A_BUF_BYTES;
        // the A fragments of k-step pair (0,1) are requested before the dequant VALU work, pair (2,3) before the
        // first MFMAs: LDS latency hides behind VALU / MFMA instead of stalling each MFMA
        u32x4 af[2][MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[0][mi] = *(const u32x4*)(abase + mi * 32 * (TBK * 2) + a_off[0]);
        float s = need_s ? TR::to_float(b.s) : 1.f;
        float z = need_z ? TR::to_float(b.z) : scalar_zero;
        dq.set(s, z, p.w_mode);
        u32x4 bfrag[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const uint32_t h = ((b.w[ks] >> (4 * dd)) & 0x000F000Fu) | TR::MAGIC2;
                bfrag[ks][dd] = dq.apply(h, p.w_mode);
            }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[(ks + 1) & 1][mi] = *(const u32x4*)(abase + mi * 32 * (TBK * 2) + a_off[ks + 1]);
            }
            if (p.flags & 1) __builtin_amdgcn_s_setprio(1);  // experiment (tuning[3] & 1): favour the MFMA-issuing wave
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = mfma32<Tag>(af[ks & 1][mi], bfrag[ks], acc[mi]);
            if (p.flags & 1) __builtin_amdgcn_s_setprio(0);
        }
    };

    BStep B0, B1;
    AStep A0;
    load_a(A0, 0);
    load_b(B0, 0);
    if (ksteps > 1) load_b(B1, 1);
    put_a(A0, 0);
    __syncthreads();
    {
        int st = 0;
        for (; st + 2 <= ksteps; st += 2) {
            load_a(A0, st + 1);
            compute(B0, 0);
            load_b(B0, st + 2 < ksteps ? st + 2 : ksteps - 1);
            put_a(A0, 1);
            __syncthreads();
            load_a(A0, st + 2 < ksteps ? st + 2 : ksteps - 1);
            compute(B1, 1);
            load_b(B1, st + 3 < ksteps ? st + 3 : ksteps - 1);
            put_a(A0, 0);
            __syncthreads();
        }
        if (ksteps & 1) compute(B0, 0);
    }

    tiled_epilogue<Tag, MI>(p, acc, smem, bid, nt, m0, slice);
}
#endif  // GL_AB_KERNELS

// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined tiled kernel (default).  Same tile (128 x 128), wave layout (wave w: all rows x 32 columns) and
// epilogue as gemm_w4_tiled_kernel<Tag, 4>; what changes is the K loop, which on one wave per SIMD spent ~75 % of
// its time with the matrix cores idle (MFMA busy 22 %, profiles/r01_pmc): dequant VALU, LDS waits and MFMAs ran one
// after the other, and every 64-k step waited for a memory round trip.
//   * One 64-k step = 16 MFMA "slots".  Slot m issues MFMA m of THIS step, the ds_read of the A fragment that is
//     needed four slots later, and a ~5-instruction slice of the dequantisation of the NEXT step's weights (two
//     B-fragment sets), so the matrix core executes (32 cycles per MFMA) while the VALU unpacks.  The order is
//     written out in the source and pinned with sched_barrier: left alone, the machine scheduler regroups it into
//     "all VALU, then all MFMA" and sinks / hoists the memory operations across the section.
//   * x goes global -> registers (requested at the top of step i for step i + 2) -> LDS (written in slots MI..2MI-1
//     of step i + 1) -> fragments; four LDS stages make one barrier per step sufficient.  (Requesting x four steps
//     ahead was measured slower: the extra 32 registers and in-flight loads cost more than the latency they hide.)
//   * weights + metadata of step i + 1 + PB are requested when step i + 1 has been dequantised (register ring of
//     PB = 4 steps = 4 KiB per wave in flight).
//   * every global access is a raw buffer load whose descriptor ends with this block's K slice (weights), the
//     metadata table or the M x K activations: the ring's run-ahead, rows >= M and the rounded-up trip count read
//     zeros without memory traffic and without branches — the compiler's vmcnt bookkeeping stays exact.
//   * dequant is branch-free: bf16 converts nibbles with v_cvt_f32_ubyteN in natural k order (no v_perm staging of
//     x, 19 instead of 27 VALU per packed word), fp16 keeps the AND/OR magic pairs; the W_group_mode switch is folded
//     into two kernel-uniform coefficients.
constexpr int PB = 4;        // K steps of weights in flight per wave; also the unroll factor of the K loop
constexpr int NSTAGE = 4;    // LDS stages of x

template <typename Tag>
struct DeqPipe;
template <>
struct DeqPipe<half_tag> {
    static constexpr bool PERM_A = true;  // fragments hold (k_d, k_{d+4}) pairs: x is staged pair-permuted
    h2_t zsub2, s2, zadd2;
    uint32_t w;
    // v = fma(q - z * u13, s, z * u4): u13 = 1 for modes 1, 3 (zero subtracted first), u4 = 1 for mode 4 (fma mode)
    __device__ __forceinline__ void set(float s, float z, float u13, float u4) {
        const _Float16 zs = (_Float16)(z * u13), za = (_Float16)(z * u4), sc = (_Float16)s;
        zsub2 = (h2_t){zs, zs};
        zadd2 = (h2_t){za, za};
        s2 = (h2_t){sc, sc};
    }
    h2_t q;
    __device__ __forceinline__ void word(uint32_t v) { w = v; }
    __device__ __forceinline__ void stage_a(int dd) {  // extract the pair, remove the magic offset
        const uint32_t h = ((w >> (4 * dd)) & 0x000F000Fu) | 0x64006400u;  // 1024 + q: exact
        q = __builtin_bit_cast(h2_t, h) - (h2_t){(_Float16)1024.0f, (_Float16)1024.0f};
    }
    __device__ __forceinline__ uint32_t stage_b() const {
        return __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(q - zsub2, s2, zadd2));
    }
};
template <>
struct DeqPipe<bf16_tag> {
    static constexpr bool PERM_A = false;  // natural k order
    float A, Bq;
    uint32_t t0, t1;
    // v = fma(q, s, z * (u4 - u13 * s)) in fp32, rounded once to bf16
    __device__ __forceinline__ void set(float s, float z, float u13, float u4) {
        A = s;
        Bq = z * __builtin_fmaf(-u13, s, u4);
    }
    __device__ __forceinline__ void word(uint32_t v) {  // even / odd nibbles as bytes
        t0 = v & 0x0F0F0F0Fu;
        t1 = (v >> 4) & 0x0F0F0F0Fu;
        // opaque to the optimiser: otherwise it rewrites byte i of t0 as v_bfe_u32(v, 8i, 4) + v_cvt_f32_ubyte0
        // (2 instructions per value) instead of one v_cvt_f32_ubyte<i> on the masked word
        asm volatile("" : "+v"(t0), "+v"(t1));
    }
    float lo, hi;
    __device__ __forceinline__ void stage_a(int dd) {  // two v_cvt_f32_ubyte<dd>
        lo = (float)((t0 >> (8 * dd)) & 0xFFu);
        hi = (float)((t1 >> (8 * dd)) & 0xFFu);
    }
    __device__ __forceinline__ uint32_t stage_b() const {  // two v_fma_f32 + v_cvt_pk_bf16_f32
        const b2_t v = {(__bf16)__builtin_fmaf(lo, A, Bq), (__bf16)__builtin_fmaf(hi, A, Bq)};
        return __builtin_bit_cast(uint32_t, v);
    }
};

// MI = 32-row blocks per wave: 4 (128-row tiles, two blocks per CU) or 8 (256-row tiles, one block per CU, 512
// registers).  A dequantised weight fragment feeds MI MFMAs, and one wave per SIMD can hide only ~5 other
// instructions behind a 32-cycle MFMA (MI355X_MICROARCH.md), so MI = 8 halves the VALU work per MFMA slot:
// 19 dequant instructions per packed word / 8 MFMAs + 1 ds_read per slot fits that budget, MI = 4 does not.
// EXP (development, tuning[3] >> 8): drop parts of the loop to see what each costs — 1 LDS writes, 2 barrier,
// 4 dequant VALU, 8 fragment reads, 16 global loads.  Results are wrong for EXP != 0.
template <typename Tag, int MI, int EXP = 0>
__global__ __launch_bounds__(256, (MI <= 4 ? 2 : 1)) void gemm_w4_pipe_kernel(const WnParams p) {
    using TR = F16Traits<Tag>;
    using DQ = DeqPipe<Tag>;
    constexpr int TBM = 32 * MI, A_BUF_BYTES = TBM * TBK * 2, SLOTS = TBM * 8 / 256;
    constexpr int NSLOT = 4 * MI;   // MFMAs per 64-k step
    constexpr int PS = MI / 4;      // MFMA slots per dequantised pair (16 pairs per step)
    static_assert(SLOTS == MI && PB == 4 && NSTAGE == 4 && (MI == 4 || MI == 8), "slot schedule below is written for these");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [NSTAGE][A_BUF_BYTES], later the C tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, kb = lane >> 5;
    const int mtiles = (p.M + TBM - 1) / TBM;
    const int bid = blockIdx.x;
    const int mt = bid % mtiles, nt = bid / mtiles;  // M tiles fastest: neighbours share the weight tile in L2
    const int slice = blockIdx.y;
    const int m0 = mt * TBM;
    const int n = nt * TBN + wave * 32 + col;  // this lane's column

    const int ksteps = p.rows_per_slice / (TBK / 8);
    const int row_s0 = slice * p.rows_per_slice;
    const int k_s0 = row_s0 * 8;

    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    const float scalar_zero = p.zero_is_scalar ? (float)((const int32_t*)p.zeros)[0] : 0.f;
    const float u13 = (p.w_mode == 1 || p.w_mode == 3) ? 1.f : 0.f, u4 = p.w_mode == 4 ? 1.f : 0.f;
    const int sw = (int)p.stride_wk;
    const int ms = (need_s || need_z) ? (int)p.stride_meta_g : 0;
    const int meta_rows = p.gs_shift >= 31 ? 1 : (p.K >> p.gs_shift);
    const int meta_bytes = ((meta_rows - 1) * ms + p.N) * 2;
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + (int64_t)row_s0 * sw), (short)0, p.rows_per_slice * sw * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(need_s ? p.scales : (const void*)p.w), (short)0, need_s ? meta_bytes : 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(need_z ? p.zeros : (const void*)p.w), (short)0, need_z ? meta_bytes : 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, (short)0, (int)(((int64_t)(p.M - 1) * p.stride_xm + p.K) * 2), 0x00020000);

    // ---- B stream: 8 packed rows per K step; this lane needs rows 2*ks + kb (ks = 0..3) of its column ----------
    uint32_t woff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) woff[ks] = (uint32_t)((kb + 2 * ks) * sw + n) * 4u;
    const uint32_t moff = (uint32_t)n * 2u;
    struct BStep { uint32_t w[4]; uint16_t s, z; };
    // (round 1 had two timing probes here, tuning[3] & 8 / & 16, that pushed the weight / activation offsets out of range — wrong
    //  results on request; removed in round 3: & 8 is the documented XCD-map switch of the 8-wave kernel, and this kernel is that
    //  kernel's fallback for K = 64 x odd)
    constexpr uint32_t bkill = 0u, akill = 0u;
    // piece 0..3: packed words, 4: scale, 5: zero (one load instruction each, so they can be placed one per MFMA gap)
    auto load_b1 = [&](BStep& b, int step, int piece) {
        if (piece < 4) {
            b.w[piece] = __builtin_amdgcn_raw_buffer_load_b32(rsW, woff[piece] + (uint32_t)(step * 8 * sw) * 4u + bkill, 0, 0);
        } else {
            const uint32_t mo = moff + (uint32_t)(((k_s0 + step * TBK) >> p.gs_shift) * ms) * 2u + bkill;
            if (piece == 4) b.s = __builtin_amdgcn_raw_buffer_load_b16(rsS, mo, 0, 0);
            else b.z = __builtin_amdgcn_raw_buffer_load_b16(rsZ, mo, 0, 0);
        }
    };
    auto load_b = [&](BStep& b, int step) {
#pragma unroll
        for (int piece = 0; piece < 6; ++piece) load_b1(b, step, piece);
    };
    // ---- A stream: 128 rows x 64 k per step = 1024 16-byte slots, 4 per thread ---------------------------------
    struct AStep { u32x4 v[SLOTS]; };
    uint32_t xoff[SLOTS];
    int a_wr[SLOTS];  // LDS byte offset (inside a stage) of this thread's staging slots
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int u = tid + 256 * i, r = u >> 3, sl = u & 7;
        xoff[i] = m0 + r < p.M ? (uint32_t)(((int64_t)(m0 + r) * p.stride_xm + k_s0 + sl * 8) * 2) : 0x80000000u;
        a_wr[i] = a_slot(r, sl);
    }
    auto load_a1 = [&](AStep& a, int step, int i) {
        const uint32_t so = (uint32_t)(step * TBK * 2) + (step >= ksteps ? 0x40000000u : akill);  // past the slice: zeros
        a.v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, xoff[i] + so, 0, 0);
    };
    auto load_a = [&](AStep& a, int step) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) load_a1(a, step, i);
    };
    auto put_a1 = [&](const AStep& a, int i, int stage) {
        u32x4 o = a.v[i];
        if constexpr (DQ::PERM_A) {  // 8 halfs x0..x7 -> pairs (x0,x4)(x1,x5)(x2,x6)(x3,x7)
            const uint32_t d0 = o[0], d1 = o[1], d2 = o[2], d3 = o[3];
            o[0] = __builtin_amdgcn_perm(d2, d0, 0x05040100u);
            o[1] = __builtin_amdgcn_perm(d2, d0, 0x07060302u);
            o[2] = __builtin_amdgcn_perm(d3, d1, 0x05040100u);
            o[3] = __builtin_amdgcn_perm(d3, d1, 0x07060302u);
        }
        *(u32x4*)(smem + stage * A_BUF_BYTES + a_wr[i]) = o;
    };

    f32x16 acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    int a_off[4];  // A-fragment offsets: row = mi*32 + col, slot = ks*2 + kb; the swizzle term depends on col only
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_off[ks] = a_slot(col, ks * 2 + kb);
    // one address register per (stage, k16 step): the row-block term fits the 16-bit immediate of ds_read, the stage
    // term does not (it cost one v_add_u32 per fragment read when left to the compiler)
    int fbase[NSTAGE][4];
#pragma unroll
    for (int st = 0; st < NSTAGE; ++st)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fbase[st][ks] = st * A_BUF_BYTES + a_off[ks];
            asm volatile("" : "+v"(fbase[st][ks]));
        }
    auto read_frag = [&](int stage, int ks, int mi) -> u32x4 {
        return *(const u32x4*)(smem + fbase[stage][ks] + mi * 32 * (TBK * 2));
    };
    DQ dq;
    auto deq_set = [&](const BStep& b) {
        const float s = need_s ? TR::to_float(b.s) : 1.f;
        const float z = need_z ? TR::to_float(b.z) : scalar_zero;
        dq.set(s, z, u13, u4);
    };

    BStep ring[PB];
    AStep RA[2];        // x of steps i + 1, i + 2 in flight / parked
    u32x4 bfrag[2][4];  // [step parity][k16 step]
    u32x4 af[2][MI];    // [k16 parity][row block]

    // ---- prologue: x of steps 0, 1 and weights of steps 0..3 requested; step 0 staged and dequantised ----------
    load_a(RA[0], 0);
    load_a(RA[1], 1);
#pragma unroll
    for (int j = 0; j < PB; ++j) load_b(ring[j], j);
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) put_a1(RA[0], i, 0);
    deq_set(ring[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        dq.word(ring[0].w[ks]);
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            dq.stage_a(dd);
            bfrag[0][ks][dd] = dq.stage_b();
        }
    }
    load_b(ring[0], PB);
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[0][mi] = read_frag(0, 0, mi);

    const int nsteps = (ksteps + PB - 1) / PB * PB;  // steps >= ksteps multiply zeros (x is killed, weights read as 0)
    for (int st = 0; st < nsteps; st += PB) {
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int step = st + j;
            BStep& bn = ring[(j + 1) & 3];  // weights of step + 1: dequantised during this step's MFMAs, then re-requested
            // Slot plan (MFMA m of this step, then what hides behind it).  Nothing is issued in bulk: a buffer load costs
            // ~40 issue cycles and a ds_write_b128 ~30-90 when they queue up, one per 32-cycle MFMA gap is free.
            //   gaps 0 .. MI-1            one x request each (step + 2 -> the register set written out one step ago)
            //   gaps WPOS(i)              the MI LDS writes of x of step + 1, spread over the first 3/4 of the step
            //   gap  3MI - 2              barrier (only the writes are waited for: lgkmcnt(reads issued since))
            //   gaps 3MI .. 4MI - 1       fragments of the next step's first k16 (next stage); weight requests
            //   every gap                 one fragment read; dequant of step + 1: stage_a / stage_b alternate (MI = 8)
            if constexpr (MI == 8) {
                constexpr int BAR = 3 * MI - 2;
                auto wpos = [](int i) { return 1 + (i * (3 * MI - 3)) / MI; };
    #pragma unroll
                for (int m = 0; m < NSLOT; ++m) {
                    const int ks = m / MI, mi = m % MI;
                    acc[mi] = mfma32<Tag>(af[ks & 1][mi], bfrag[j & 1][ks], acc[mi]);
                    // fragment needed MI slots from now (for ks == 3: the first k16 of the next step, next stage)
                    if (!(EXP & 8)) af[(ks + 1) & 1][mi] = ks < 3 ? read_frag(j, ks + 1, mi) : read_frag((j + 1) & 3, 0, mi);
                    if (m == 0) deq_set(bn);
                    if (!(EXP & 4)) {  // 16 pairs per step: one per PS slots, split over the PS slots
                        const int pi = m / PS, wd = pi >> 2, dd = pi & 3;
                        if (m % PS == 0) {
                            if (dd == 0) dq.word(bn.w[wd]);
                            dq.stage_a(dd);
                        }
                        if (m % PS == PS - 1) bfrag[(j + 1) & 1][wd][dd] = dq.stage_b();
                    }
                    if (m < MI && !(EXP & 16)) load_a1(RA[j & 1], step + 2, m);
    #pragma unroll
                    for (int i = 0; i < SLOTS; ++i)
                        if (m == wpos(i) && !(EXP & 1)) put_a1(RA[(j + 1) & 1], i, (j + 1) & 3);  // x of step + 1
                    if (m == BAR && !(EXP & 2)) {
                        // the last write was issued in slot wpos(MI-1); the fragment reads issued since may stay in flight
                        constexpr int READS_SINCE = BAR - (1 + ((MI - 1) * (3 * MI - 3)) / MI);
                        asm volatile("" ::: "memory");  // compiler-level fence only: no LDS access may move across
                        __builtin_amdgcn_s_waitcnt(0xC07F | (READS_SINCE << 8));
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    if (m >= NSLOT - 6 && !(EXP & 16)) load_b1(bn, step + 1 + PB, m - (NSLOT - 6));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // 128-row tiles run two blocks per CU: the second block's instructions fill the gaps, and grouping the
                // requests / LDS writes measured faster than spreading them (cfgB 47.0 vs 52.3 us)
                if (!(EXP & 16)) load_a(RA[j & 1], step + 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < NSLOT; ++m) {
                    const int ks = m / MI, mi = m % MI;
                    acc[mi] = mfma32<Tag>(af[ks & 1][mi], bfrag[j & 1][ks], acc[mi]);
                    af[(ks + 1) & 1][mi] = ks < 3 ? read_frag(j, ks + 1, mi) : read_frag((j + 1) & 3, 0, mi);
                    if (m == 0) deq_set(bn);
                    const int wd = m >> 2, dd = m & 3;
                    if (dd == 0) dq.word(bn.w[wd]);
                    dq.stage_a(dd);
                    bfrag[(j + 1) & 1][wd][dd] = dq.stage_b();
                    if (m >= MI && m < MI + SLOTS) put_a1(RA[(j + 1) & 1], m - MI, (j + 1) & 3);  // x of step + 1
                    if (m == MI + SLOTS - 1) __syncthreads();  // stage (j + 1) complete before slot 3 * MI reads it
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!(EXP & 16)) load_b(bn, step + 1 + PB);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    tiled_epilogue<Tag, MI>(p, acc, smem, bid, nt, m0, slice);
}

// tuning[1]: 0 auto | n force split-K n
bool plan_gemm_wn_tiled(const gemlite_hip_forward_args& a, WnParams& p, LaunchPlan& lp) {
    if (a.W_nbits != 4) return false;  // 2-/1-bit words span more than one MFMA fragment: streaming kernel
    if (a.N % TBN != 0 || a.K % TBK != 0) return false;
    if (a.output_dtype != a.input_dtype) return false;
    const bool uses_s = a.W_group_mode >= 2 || a.channel_scale_mode == 1 || a.channel_scale_mode == 3;
    const bool has_z = (a.W_group_mode == 1 || a.W_group_mode >= 3);
    if (uses_s && a.meta_dtype != a.input_dtype) return false;
    if (has_z && !a.zero_is_scalar && a.zeros_dtype != a.input_dtype) return false;
    if (has_z && a.zero_is_scalar && a.zeros_dtype != GEMLITE_DT_INT32) return false;
    if ((a.stride_xm * 2) % 16 != 0 || ((uintptr_t)a.x % 16) != 0) return false;
    if (p.group_size % TBK != 0) return false;  // one (scale, zero) pair per column and K step
    const int rows = (int)(a.K / 8), step_rows = TBK / 8;
    const int units = rows / step_rows;
    // buffer descriptors: 32-bit byte offsets (and a 2^30 "kill" bit on the activation offsets)
    if ((int64_t)rows * a.stride_wk * 4 >= (1ll << 31) || ((int64_t)a.M * a.stride_xm + a.K) * 2 >= (1ll << 30)) return false;
    if (((int64_t)(a.K / (p.group_size > 0 ? p.group_size : a.K)) * p.stride_meta_g + a.N) * 2 >= (1ll << 31)) return false;
    // tuning[2]: 0 auto = the deep-pipelined kernel | 4 = the one-step-ahead kernel it replaced (kept for A/B runs)
    //            8 = the software-pipelined kernel with 256-row tiles, one block per CU (default: 128-row tiles)
    const bool legacy = a.tuning[2] == 4;
#ifndef GL_AB_KERNELS
    // the one-step-ahead kernel and the 256-row variant lost their A/B runs in round 1 / 2 and are built only with `make AB=1`
    if (a.tuning[2] == 4 || a.tuning[2] == 8) return false;
#endif
    // (256-row tiles: 52.6 us per 4096-deep K at one block per CU vs 39.8 us for two 128-row blocks, and their
    //  split-K needs twice the slices for the same block count — opt-in until the loop is under the issue budget)
    const int mi = legacy ? 4 : (a.tuning[2] == 8 ? 8 : 4);
    const int tbm = 32 * mi;
    const int64_t tiles = (int64_t)(a.N / TBN) * ((a.M + tbm - 1) / tbm);
    auto ok = [&](int sk) { return sk >= 1 && units % sk == 0; };
    int splitk = 0;
    if (a.tuning[1] > 0) {
        if (!ok(a.tuning[1])) return false;
        splitk = a.tuning[1];
    } else {
        // Measured (profiles/r01_run31): every K slice costs ~1-2 us in the combine and a block needs >= 16 steps to
        // amortise its prologue, so: the fewest slices that give every CU a block, then one more doubling towards two
        // blocks per CU only while a slice keeps >= 2048 of K (4096^2: 4 slices / 256 blocks 19.9 us vs 8 / 512
        // 23.0 us; 8192^2: 4 / 512 49.4 us vs 2 / 256 61 us).  Slices of whole PB-step rounds are preferred.
        for (int pass = 0; pass < 2 && !splitk; ++pass)
            for (int sk = 1; sk <= units && sk <= 16; sk *= 2) {
                if (!ok(sk) || (pass == 0 && (units / sk) % PB != 0)) continue;
                splitk = sk;
                if (tiles * sk >= 256) break;
            }
        if (!splitk) return false;
        if (mi == 4 && tiles * splitk < 512 && ok(splitk * 2) && a.K / (splitk * 2) >= 2048) splitk *= 2;
    }
    if (splitk > 1 && tiles > MAX_SPLITK_COUNTERS) return false;
    if ((uint64_t)splitk * tbm * TBN * 4 >= (1ull << 31)) return false;  // buffer descriptor range
    p.splitk = splitk;
    p.rows_per_slice = rows / splitk;
    const bool f16 = a.input_dtype == GEMLITE_DT_FP16;
#ifdef GL_AB_KERNELS
    if (legacy) {
        lp.fn = f16 ? (const void*)gemm_w4_tiled_kernel<half_tag, 4> : (const void*)gemm_w4_tiled_kernel<bf16_tag, 4>;
        lp.name = "gemm_w4_tiled_kernel<legacy>";
    } else
#endif
    {
#ifdef GL_AB_KERNELS
        if (mi == 8) lp.fn = f16 ? (const void*)gemm_w4_pipe_kernel<half_tag, 8> : (const void*)gemm_w4_pipe_kernel<bf16_tag, 8>;
        else
#endif
        lp.fn = f16 ? (const void*)gemm_w4_pipe_kernel<half_tag, 4> : (const void*)gemm_w4_pipe_kernel<bf16_tag, 4>;
#ifdef GL_TILED_EXPERIMENTS
        if (mi == 8 && !f16) switch (a.tuning[3] >> 8) {
            case 1: lp.fn = (const void*)gemm_w4_pipe_kernel<bf16_tag, 8, 1>; break;
            case 2: lp.fn = (const void*)gemm_w4_pipe_kernel<bf16_tag, 8, 2>; break;
            case 3: lp.fn = (const void*)gemm_w4_pipe_kernel<bf16_tag, 8, 3>; break;
            case 4: lp.fn = (const void*)gemm_w4_pipe_kernel<bf16_tag, 8, 4>; break;
            case 8: lp.fn = (const void*)gemm_w4_pipe_kernel<bf16_tag, 8, 8>; break;
            case 16: lp.fn = (const void*)gemm_w4_pipe_kernel<bf16_tag, 8, 16>; break;
            case 19: lp.fn = (const void*)gemm_w4_pipe_kernel<bf16_tag, 8, 19>; break;
            case 31: lp.fn = (const void*)gemm_w4_pipe_kernel<bf16_tag, 8, 31>; break;
            default: break;
        }
#endif
        lp.name = mi == 8 ? "gemm_w4_tiled_kernel<256x128>" : "gemm_w4_tiled_kernel<128x128>";
    }
    lp.grid = dim3((unsigned)tiles, splitk, 1);
    lp.block = dim3(256, 1, 1);
    const size_t a_b = (size_t)(legacy ? 2 : NSTAGE) * tbm * TBK * 2, c_b = (size_t)C_ROWS * C_PITCH * 4 + 16;
    lp.lds_bytes = a_b > c_b ? a_b : c_b;
    lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * tbm * TBN * 4 : 0;
    lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

}  // namespace gl
