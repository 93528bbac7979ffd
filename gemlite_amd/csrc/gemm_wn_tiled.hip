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
    const int sw = (int)p.stride_wk, gsz = p.group_size, ms = (int)mstride;  // 32-bit index math in the loop
    const int kbase = (int)k_s0;
    struct BStep { uint32_t w[4]; uint16_t s, z; };
    auto load_b = [&](BStep& b, int step) {
        const uint32_t* wp = wbase + step * 8 * sw;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) b.w[ks] = wp[2 * ks * sw];
        const int grp = group_of(kbase + step * TBK, gsz, p.gs_shift);
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

    // ---- epilogue.  C fragment of a 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5).
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
    // last arriver: slices outer, units inner -> every slice's 16-byte loads are in flight together
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

// ---------------------------------------------------------------------------------------------------------------
// 8-wave variant: tile 256 (M) x 128 (N), waves (wm, wn) = (wave >> 2, wave & 3), each 128 rows x 32 columns.
// The two waves that share a column slice each dequantise HALF of the slice's B fragments (k16 steps 2*wm, 2*wm+1)
// and exchange them through LDS (fragment layout: one 16-byte slot per lane, conflict-free), so the dequant VALU
// work per flop is half of the 4-wave kernel's at the same 2-waves-per-SIMD occupancy.
// ---------------------------------------------------------------------------------------------------------------
constexpr int T8_BM = 256;
constexpr int T8_A_BYTES = T8_BM * TBK * 2;          // 32 KiB per stage
constexpr int T8_B_BYTES = 4 * 4 * 64 * 16;          // [wn][ks][lane] 16-byte fragments: 16 KiB per stage
constexpr int T8_STAGE = T8_A_BYTES + T8_B_BYTES;    // 48 KiB

template <typename Tag>
__global__ __launch_bounds__(512, 2) void gemm_w4_tiled8_kernel(const WnParams p) {
    using TR = F16Traits<Tag>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2][T8_STAGE], later the C tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int col = lane & 31, kb = lane >> 5;
    const int mtiles = (p.M + T8_BM - 1) / T8_BM;
    const int bid = blockIdx.x;
    const int mt = bid % mtiles, nt = bid / mtiles;
    const int slice = blockIdx.y;
    const int m0 = mt * T8_BM;
    const int n = nt * TBN + wn * 32 + col;  // this lane's column

    const int ksteps = p.rows_per_slice / (TBK / 8);
    const int row_s0 = slice * p.rows_per_slice;
    const int64_t k_s0 = (int64_t)row_s0 * 8;

    const bool need_s = p.w_mode >= 2, need_z = (p.w_mode == 1 || p.w_mode >= 3) && !p.zero_is_scalar;
    const uint16_t* sp = need_s ? (const uint16_t*)p.scales : (const uint16_t*)p.w;
    const uint16_t* zp = need_z ? (const uint16_t*)p.zeros : (const uint16_t*)p.w;
    const int64_t mstride = (need_s || need_z) ? p.stride_meta_g : 0;
    const float scalar_zero = p.zero_is_scalar ? (float)((const int32_t*)p.zeros)[0] : 0.f;

    // B: this wave dequantises k16 steps ks = 2*wm + {0, 1}: packed rows 2*ks + kb of its column
    const uint32_t* wbase = p.w + (int64_t)(row_s0 + 4 * wm + kb) * p.stride_wk + n;
    const int sw = (int)p.stride_wk, gsz = p.group_size, ms = (int)mstride, kbase = (int)k_s0;
    struct BStep { uint32_t w[2]; uint16_t s, z; };
    auto load_b = [&](BStep& b, int step) {
        const uint32_t* wp = wbase + step * 8 * sw;
        b.w[0] = wp[0];
        b.w[1] = wp[2 * sw];
        const int grp = group_of(kbase + step * TBK, gsz, p.gs_shift);
        b.s = sp[grp * ms + n];
        b.z = zp[grp * ms + n];
    };
    Deq2<Tag> dq;
    auto put_b = [&](const BStep& b, int buf) {  // dequantise 2 fragments -> LDS
        const float s = need_s ? TR::to_float(b.s) : 1.f;
        const float z = need_z ? TR::to_float(b.z) : scalar_zero;
        dq.set(s, z, p.w_mode);
        unsigned char* bb = smem + buf * T8_STAGE + T8_A_BYTES + ((wn * 4 + 2 * wm) * 64 + lane) * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            u32x4 f;
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) f[dd] = dq.apply(((b.w[q] >> (4 * dd)) & 0x000F000Fu) | TR::MAGIC2, p.w_mode);
            *(u32x4*)(bb + q * 64 * 16) = f;
        }
    };
    // A: 256 rows x 64 k per step = 2048 16-byte slots, 4 per thread
    const uint16_t* xg = (const uint16_t*)p.x;
    struct AStep { u32x4 v[4]; };
    const uint16_t* xrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = tid + 512 * i, r = u >> 3, s = u & 7;
        const int rr = m0 + r < p.M ? m0 + r : p.M - 1;
        xrow[i] = xg + (int64_t)rr * p.stride_xm + k_s0 + s * 8;
    }
    auto load_a = [&](AStep& a, int step) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a.v[i] = *(const u32x4*)(xrow[i] + step * TBK);
            if (m0 + ((tid + 512 * i) >> 3) >= p.M) a.v[i] = (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto put_a = [&](const AStep& a, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = tid + 512 * i, r = u >> 3, s = u & 7;
            const uint32_t d0 = a.v[i][0], d1 = a.v[i][1], d2 = a.v[i][2], d3 = a.v[i][3];
            u32x4 o;
            o[0] = __builtin_amdgcn_perm(d2, d0, 0x05040100u);
            o[1] = __builtin_amdgcn_perm(d2, d0, 0x07060302u);
            o[2] = __builtin_amdgcn_perm(d3, d1, 0x05040100u);
            o[3] = __builtin_amdgcn_perm(d3, d1, 0x07060302u);
            *(u32x4*)(smem + buf * T8_STAGE + a_slot(r, s)) = o;
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    int a_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_off[ks] = a_slot(wm * 128 + col, ks * 2 + kb);

    auto compute = [&](int buf) {
        const unsigned char* abase = smem + buf * T8_STAGE;
        const unsigned char* bbase = abase + T8_A_BYTES + (wn * 4 * 64 + lane) * 16;
        u32x4 bfrag[4], af[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bfrag[ks] = *(const u32x4*)(bbase + ks * 64 * 16);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[0][mi] = *(const u32x4*)(abase + mi * 32 * (TBK * 2) + a_off[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) af[(ks + 1) & 1][mi] = *(const u32x4*)(abase + mi * 32 * (TBK * 2) + a_off[ks + 1]);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[mi] = mfma32<Tag>(af[ks & 1][mi], bfrag[ks], acc[mi]);
        }
    };

    // prologue: stage 0 in LDS, B words of step 1 in registers
    BStep Bn;
    AStep An;
    load_a(An, 0);
    load_b(Bn, 0);
    put_a(An, 0);
    put_b(Bn, 0);
    if (ksteps > 1) { load_a(An, 1); load_b(Bn, 1); }
    __syncthreads();
    for (int st = 0; st < ksteps; ++st) {
        const int cur = st & 1;
        compute(cur);
        if (st + 1 < ksteps) {
            put_a(An, cur ^ 1);   // data of step st+1 (requested one iteration ago)
            put_b(Bn, cur ^ 1);
            if (st + 2 < ksteps) { load_a(An, st + 2); load_b(Bn, st + 2); }
        }
        __syncthreads();
    }

    // ---- epilogue: 128 rows (one wm half) at a time through LDS, 16-byte slabs / outputs ----------------------
    float* ct = (float*)smem;
    unsigned* flag = (unsigned*)(smem + C_ROWS * C_PITCH * 4);
    const int tile_lin = bid;
    constexpr int UNITS = C_ROWS * TBN / 4 / 512;
    constexpr int NOUT = T8_BM * TBN;
    const int64_t ncol0 = (int64_t)nt * TBN;
    float* slab = p.slabs + ((int64_t)tile_lin * p.splitk) * NOUT;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, (short)0, p.splitk * NOUT * 4, 0x00020000);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        if (wm == half) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * kb;
                    ct[r * C_PITCH + wn * 32 + col] = acc[mi][e];
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = tid + 512 * i, r = u >> 5, c4 = (u & 31) * 4;
            const int m = m0 + half * C_ROWS + r;
            if (m < p.M) {
                const f32x4 v = *(const f32x4*)(ct + r * C_PITCH + c4);
                if (p.splitk == 1) store_out4_t<Tag>(p.epi, v, m, ncol0 + c4);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                                            (slice * NOUT + (half * C_ROWS + r) * TBN + c4) * 4, 0, 16);
            }
        }
    }
    if (p.splitk == 1) return;
    __syncthreads();
    if (!splitk_arrive_is_last(p.counters + tile_lin, p.splitk, flag)) return;
    for (int half = 0; half < 2; ++half) {
        f32x4 sum[UNITS];
#pragma unroll
        for (int i = 0; i < UNITS; ++i) sum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < p.splitk; ++s) {
            u32x4 t[UNITS];
#pragma unroll
            for (int i = 0; i < UNITS; ++i) {
                const int u = tid + 512 * i, r = u >> 5, c4 = (u & 31) * 4;
                t[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (s * NOUT + (half * C_ROWS + r) * TBN + c4) * 4, 0, 16);
            }
#pragma unroll
            for (int i = 0; i < UNITS; ++i) sum[i] += __builtin_bit_cast(f32x4, t[i]);
        }
#pragma unroll
        for (int i = 0; i < UNITS; ++i) {
            const int u = tid + 512 * i, r = u >> 5, c4 = (u & 31) * 4;
            const int m = m0 + half * C_ROWS + r;
            if (m < p.M) store_out4_t<Tag>(p.epi, sum[i], m, ncol0 + c4);
        }
    }
    if (tid == 0) splitk_reset(p.counters + tile_lin);
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
    // tuning[2]: 0 auto | 4 / 8 = 32-row blocks per wave (128- / 256-row tiles).  256-row tiles halve the dequant and
    // staging work per flop but run at one wave per SIMD; measured slower than two 128-row blocks per CU
    // (profiles/r01_run12_bench_sweep.jsonl), so 128-row tiles are the default.
    //            16 = the 8-wave 256 x 128 kernel (B fragments shared through LDS)
    // (measured: 61 us vs 56 us for the 4-wave kernel at cfgB — the K loop is stall-bound, not VALU-bound — so opt-in only)
    const bool eight = a.tuning[2] == 16;
    int mi = a.tuning[2] == 8 ? 8 : 4;
    const int tbm = eight ? T8_BM : 32 * mi;
    const int64_t tiles = (int64_t)(a.N / TBN) * ((a.M + tbm - 1) / tbm);
    auto ok = [&](int sk) { return sk >= 1 && units % sk == 0; };
    int splitk = 0;
    if (a.tuning[1] > 0) {
        if (!ok(a.tuning[1])) return false;
        splitk = a.tuning[1];
    } else {
        for (int sk = 1; sk <= units && sk <= 16; sk *= 2) {
            if (!ok(sk)) continue;
            splitk = sk;
            if (tiles * sk >= (eight ? 256 : 512)) break;  // two waves per SIMD: one wave's VALU overlaps the other's MFMAs
        }
        if (!splitk) return false;
    }
    if (splitk > 1 && tiles > MAX_SPLITK_COUNTERS) return false;
    if ((uint64_t)splitk * tbm * TBN * 4 >= (1ull << 31)) return false;  // buffer descriptor range
    p.splitk = splitk;
    p.rows_per_slice = rows / splitk;
    const bool f16 = a.input_dtype == GEMLITE_DT_FP16;
    if (eight) {
        lp.fn = f16 ? (const void*)gemm_w4_tiled8_kernel<half_tag> : (const void*)gemm_w4_tiled8_kernel<bf16_tag>;
        lp.name = "gemm_w4_tiled8_kernel";
        lp.grid = dim3((unsigned)tiles, splitk, 1);
        lp.block = dim3(512, 1, 1);
        lp.lds_bytes = 2 * T8_STAGE;
        lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * tbm * TBN * 4 : 0;
        lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
        return true;
    }
    lp.fn = mi == 8 ? (f16 ? (const void*)gemm_w4_tiled_kernel<half_tag, 8> : (const void*)gemm_w4_tiled_kernel<bf16_tag, 8>)
                    : (f16 ? (const void*)gemm_w4_tiled_kernel<half_tag, 4> : (const void*)gemm_w4_tiled_kernel<bf16_tag, 4>);
    lp.name = "gemm_w4_tiled_kernel";
    lp.grid = dim3((unsigned)tiles, splitk, 1);
    lp.block = dim3(256, 1, 1);
    const size_t a_b = (size_t)2 * tbm * TBK * 2, c_b = (size_t)C_ROWS * C_PITCH * 4 + 16;
    lp.lds_bytes = a_b > c_b ? a_b : c_b;
    lp.slab_bytes = splitk > 1 ? (uint64_t)tiles * splitk * tbm * TBN * 4 : 0;
    lp.ws_bytes = splitk > 1 ? COUNTER_BYTES + lp.slab_bytes : 0;
    return true;
}

}  // namespace gl
