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

#include <type_traits>

namespace gl {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

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

}  // namespace gl
