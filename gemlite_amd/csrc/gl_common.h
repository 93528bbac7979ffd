// gl_common.h — shared device/host definitions for libgemlite_hip.so (gfx950 only).
//
// Data layout facts every kernel relies on (reference: gemlite/core.py:384-398,419-420,478-485;
// gemlite/bitpack.py:36-60):
//   W_q packed   : int32 [K/e, N], N-contiguous; word (j, n) holds k = j*e .. j*e+e-1 of column n,
//                  element i at bits [b*i, b*i+b)  (e = 32 / W_nbits)
//   scales/zeros : [K/group, N], N-contiguous (or [N] channel-wise, or a scalar zero)
//   x            : [M, K] row-major,  out : [M, N] row-major
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gemlite_hip.h"

namespace gl {

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

struct half_tag {};
struct bf16_tag {};

// ---------------------------------------------------------------------------------------------
// 16-bit float traits: the "magic number" unpack (q | MAGIC is the float OFF + q), packed dot2
// ---------------------------------------------------------------------------------------------
template <typename Tag>
struct F16Traits;

template <>
struct F16Traits<half_tag> {
    static constexpr uint32_t MAGIC2 = 0x64006400u;  // two fp16 1024.0
    static constexpr float OFF = 1024.0f;             // exact for q < 1024
    static constexpr int MAX_QBITS = 8;
    static constexpr uint32_t ONES2 = 0x3C003C00u;  // (1.0h, 1.0h)
    static constexpr int DT = GEMLITE_DT_FP16;
    typedef h8_t frag8;
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a), __builtin_bit_cast(h2_t, b), c, false);
    }
    static __device__ __forceinline__ float to_float(uint16_t v) {
        return (float)__builtin_bit_cast(_Float16, v);
    }
    static __device__ __forceinline__ uint16_t from_float(float f) {
        return __builtin_bit_cast(uint16_t, (_Float16)f);
    }
};

template <>
struct F16Traits<bf16_tag> {
    static constexpr uint32_t MAGIC2 = 0x43004300u;  // two bf16 128.0
    static constexpr float OFF = 128.0f;              // exact for q < 128
    static constexpr int MAX_QBITS = 4;
    static constexpr uint32_t ONES2 = 0x3F803F80u;
    static constexpr int DT = GEMLITE_DT_BF16;
    typedef b8_t frag8;
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2_t, a), __builtin_bit_cast(b2_t, b), c, false);
    }
    static __device__ __forceinline__ float to_float(uint16_t v) {
        return __builtin_bit_cast(float, (uint32_t)v << 16);
    }
    static __device__ __forceinline__ uint16_t from_float(float f) {
        return __builtin_bit_cast(uint16_t, (__bf16)f);
    }
};

// ---------------------------------------------------------------------------------------------
// generic scalar load / store by dtype code (metadata, epilogue). Uniform branches only.
// ---------------------------------------------------------------------------------------------
#define GL_HD __host__ __device__ __forceinline__

GL_HD float fp8e4m3_to_float(uint8_t v) {
    // OCP e4m3fn: bias 7, no inf, NaN = S.1111.111
    const uint32_t s = (uint32_t)(v & 0x80) << 24;
    const uint32_t em = v & 0x7F;
    if (em == 0x7F) return __builtin_bit_cast(float, s | 0x7FC00000u);
    const uint32_t e = em >> 3, m = em & 7;
    if (e == 0) {  // subnormal: m * 2^-9
        const float f = (float)m * 0.001953125f;
        return s ? -f : f;
    }
    return __builtin_bit_cast(float, s | ((e + 120u) << 23) | (m << 20));
}

GL_HD float fp8e5m2_to_float(uint8_t v) {
    // e5m2 is the top byte of an fp16
    const uint16_t h = (uint16_t)v << 8;
    return (float)__builtin_bit_cast(_Float16, h);
}

// round-to-nearest-even, saturating to +-448 like torch's .to(float8_e4m3fn) for finite inputs
GL_HD uint8_t float_to_fp8e4m3(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    const uint8_t s = (u >> 24) & 0x80;
    u &= 0x7FFFFFFFu;
    if (u > 0x7F800000u) return s | 0x7F;  // NaN
    const float a = __builtin_bit_cast(float, u);
    if (a > 464.0f) return s | 0x7F;  // torch: overflow -> NaN pattern (no inf in e4m3fn)
    if (a < 0.015625f) {               // below 2^-6: subnormal grid of 2^-9
        const float r = __builtin_rintf(a * 512.0f);
        return s | (uint8_t)r;          // r in 0..8 (8 == smallest normal 0x08)
    }
    // normal: keep 3 mantissa bits with RNE
    const uint32_t lsb = (u >> 20) & 1u;
    u += 0x7FFFFu + lsb;
    const uint32_t e = (u >> 23) - 120u, m = (u >> 20) & 7u;
    return s | (uint8_t)((e << 3) | m);
}

GL_HD uint8_t float_to_fp8e5m2(float f) {
    // direct fp32 -> e5m2 RNE (bias 15, 2 mantissa bits); overflow -> inf like torch
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    const uint8_t s = (u >> 24) & 0x80;
    u &= 0x7FFFFFFFu;
    if (u > 0x7F800000u) return s | 0x7F;  // NaN
    const float a = __builtin_bit_cast(float, u);
    if (a >= 61440.0f) return s | 0x7C;    // >= midpoint(57344, 65536) -> inf
    if (a < 6.103515625e-05f) {            // below 2^-14: subnormal grid of 2^-16
        const float r = __builtin_rintf(a * 65536.0f);
        return s | (uint8_t)r;
    }
    const uint32_t lsb = (u >> 21) & 1u;
    u += 0xFFFFFu + lsb;
    const uint32_t e = (u >> 23) - 112u, m = (u >> 21) & 3u;
    return s | (uint8_t)((e << 2) | m);
}

__device__ __forceinline__ float load_as_float(const void* p, int64_t i, int dt) {
    switch (dt) {
        case GEMLITE_DT_FP32: return ((const float*)p)[i];
        case GEMLITE_DT_FP16: return (float)((const _Float16*)p)[i];
        case GEMLITE_DT_BF16: return __builtin_bit_cast(float, (uint32_t)((const uint16_t*)p)[i] << 16);
        case GEMLITE_DT_INT8: return (float)((const int8_t*)p)[i];
        case GEMLITE_DT_UINT8: return (float)((const uint8_t*)p)[i];
        case GEMLITE_DT_INT32: return (float)((const int32_t*)p)[i];
        case GEMLITE_DT_FP8E4: return fp8e4m3_to_float(((const uint8_t*)p)[i]);
        case GEMLITE_DT_FP8E5: return fp8e5m2_to_float(((const uint8_t*)p)[i]);
        default: return 0.0f;
    }
}

__device__ __forceinline__ void store_from_float(void* p, int64_t i, int dt, float v) {
    switch (dt) {
        case GEMLITE_DT_FP32: ((float*)p)[i] = v; break;
        case GEMLITE_DT_FP16: ((_Float16*)p)[i] = (_Float16)v; break;
        case GEMLITE_DT_BF16: ((__bf16*)p)[i] = (__bf16)v; break;
        case GEMLITE_DT_INT32: ((int32_t*)p)[i] = (int32_t)__builtin_rintf(v); break;
        default: break;
    }
}

// 4 consecutive metadata values (one per owned column) as floats
__device__ __forceinline__ f32x4 load_meta4(const void* p, int64_t i, int dt) {
    f32x4 r;
    if (dt == GEMLITE_DT_FP16) {
        for (int j = 0; j < 4; ++j) r[j] = (float)((const _Float16*)p)[i + j];
    } else if (dt == GEMLITE_DT_BF16) {
        const u32x2 v = *(const u32x2*)((const uint16_t*)p + i);
        r[0] = __builtin_bit_cast(float, v[0] << 16); r[1] = __builtin_bit_cast(float, v[0] & 0xFFFF0000u);
        r[2] = __builtin_bit_cast(float, v[1] << 16); r[3] = __builtin_bit_cast(float, v[1] & 0xFFFF0000u);
    } else if (dt == GEMLITE_DT_FP32) {
        r = *(const f32x4*)((const float*)p + i);
    } else {
        for (int j = 0; j < 4; ++j) r[j] = load_as_float(p, i + j, dt);
    }
    return r;
}

// ---------------------------------------------------------------------------------------------
// epilogue: channel scaling (gemm_kernels.py:392-404), cast, store
// ---------------------------------------------------------------------------------------------
struct Epilogue {
    void* out;
    const void* scales_w;   // [N] channel scales (channel_scale_mode 1/3)
    const float* scales_x;  // [M] per-token scales (channel_scale_mode 2/3)
    int64_t stride_om, stride_on, stride_sx_m;
    int out_dt, meta_dt, c_mode;
};

__device__ __forceinline__ float epilogue_scale(const Epilogue& e, float v, int64_t m, int64_t n) {
    if (e.c_mode == 1) {
        v *= load_as_float(e.scales_w, n, e.meta_dt);
    } else if (e.c_mode == 2) {
        v *= e.scales_x[m * e.stride_sx_m];
    } else if (e.c_mode == 3) {
        v *= e.scales_x[m * e.stride_sx_m] * load_as_float(e.scales_w, n, e.meta_dt);
    }
    return v;
}

__device__ __forceinline__ void epilogue_store(const Epilogue& e, float v, int64_t m, int64_t n) {
    store_from_float(e.out, m * e.stride_om + n * e.stride_on, e.out_dt, epilogue_scale(e, v, m, n));
}

// typed (compile-time dtype) variants used by the specialised kernels: no runtime dtype switches in hot loops
template <typename Tag>
__device__ __forceinline__ f32x4 load4_t(const void* p, int64_t i) {  // 4 consecutive 16-bit floats -> fp32
    const u32x2 v = *(const u32x2*)((const uint16_t*)p + i);
    f32x4 r;
    const uint32_t v0 = v[0], v1 = v[1];
    if constexpr (F16Traits<Tag>::DT == GEMLITE_DT_FP16) {
        // NOTE: going through bit_cast<h2_t>(v[1]) here was miscompiled by hipcc 7.2 (second dword dropped,
        // r[2], r[3] left undefined); extract the halves as integers instead.
        r[0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v0 & 0xFFFFu));
        r[1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v0 >> 16));
        r[2] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v1 & 0xFFFFu));
        r[3] = (float)__builtin_bit_cast(_Float16, (uint16_t)(v1 >> 16));
    } else {
        r[0] = __builtin_bit_cast(float, v0 << 16); r[1] = __builtin_bit_cast(float, v0 & 0xFFFF0000u);
        r[2] = __builtin_bit_cast(float, v1 << 16); r[3] = __builtin_bit_cast(float, v1 & 0xFFFF0000u);
    }
    return r;
}

template <typename Tag>
__device__ __forceinline__ void store_out_t(const Epilogue& e, float v, int64_t m, int64_t n) {
    using TR = F16Traits<Tag>;
    // channel scales of the kernel's own 16-bit type, or (round 4: BitNet's fp32 scale, helper.py:173-208) any float type
    auto sw = [&]() -> float { return e.meta_dt == TR::DT ? TR::to_float(((const uint16_t*)e.scales_w)[n]) : load_as_float(e.scales_w, n, e.meta_dt); };
    if (e.c_mode == 1) {
        v *= sw();
    } else if (e.c_mode == 2) {
        v *= e.scales_x[m * e.stride_sx_m];
    } else if (e.c_mode == 3) {
        v *= e.scales_x[m * e.stride_sx_m] * sw();
    }
    ((uint16_t*)e.out)[m * e.stride_om + n] = TR::from_float(v);
}

// 4 consecutive outputs of one row: channel scaling, cast, one 8-byte store (n0 multiple of 4)
template <typename Tag>
__device__ __forceinline__ void store_out4_t(const Epilogue& e, f32x4 v, int64_t m, int64_t n0) {
    using TR = F16Traits<Tag>;
    if (e.c_mode == 1 || e.c_mode == 3) {
        const f32x4 sw = load4_t<Tag>(e.scales_w, n0);
        v *= sw;
    }
    if (e.c_mode == 2 || e.c_mode == 3) {
        const float sx = e.scales_x[m * e.stride_sx_m];
        v *= (f32x4){sx, sx, sx, sx};
    }
    u32x2 o;
    o[0] = (uint32_t)TR::from_float(v[0]) | ((uint32_t)TR::from_float(v[1]) << 16);
    o[1] = (uint32_t)TR::from_float(v[2]) | ((uint32_t)TR::from_float(v[3]) << 16);
    *(u32x2*)((uint16_t*)e.out + m * e.stride_om + n0) = o;
}

// 4 consecutive outputs of one row from fp32 values: channel scaling (scales of fp32 / fp16 / bf16: one uniform
// three-way branch, vector loads), cast, one vector store
__device__ __forceinline__ void store_out4_any(const Epilogue& e, f32x4 v, int64_t m, int64_t n0) {
    // same operation order as epilogue_scale(): mode 3 multiplies by the PRODUCT s_x[m] * s_w[n] (bit-identical outputs)
    f32x4 sw = {1.f, 1.f, 1.f, 1.f};
    if (e.c_mode == 1 || e.c_mode == 3) {
        if (e.meta_dt == GEMLITE_DT_FP32) sw = *(const f32x4*)((const float*)e.scales_w + n0);
        else if (e.meta_dt == GEMLITE_DT_FP16) sw = load4_t<half_tag>(e.scales_w, n0);
        else sw = load4_t<bf16_tag>(e.scales_w, n0);
    }
    if (e.c_mode == 2 || e.c_mode == 3) {
        const float sx = e.scales_x[m * e.stride_sx_m];
        sw = e.c_mode == 3 ? (f32x4){sx * sw[0], sx * sw[1], sx * sw[2], sx * sw[3]} : (f32x4){sx, sx, sx, sx};
    }
    if (e.c_mode != 0) v *= sw;
    if (e.out_dt == GEMLITE_DT_FP32) {
        *(f32x4*)((float*)e.out + m * e.stride_om + n0) = v;
        return;
    }
    u32x2 o;
    if (e.out_dt == GEMLITE_DT_FP16) {
        o[0] = (uint32_t)F16Traits<half_tag>::from_float(v[0]) | ((uint32_t)F16Traits<half_tag>::from_float(v[1]) << 16);
        o[1] = (uint32_t)F16Traits<half_tag>::from_float(v[2]) | ((uint32_t)F16Traits<half_tag>::from_float(v[3]) << 16);
    } else {
        o[0] = (uint32_t)F16Traits<bf16_tag>::from_float(v[0]) | ((uint32_t)F16Traits<bf16_tag>::from_float(v[1]) << 16);
        o[1] = (uint32_t)F16Traits<bf16_tag>::from_float(v[2]) | ((uint32_t)F16Traits<bf16_tag>::from_float(v[3]) << 16);
    }
    *(u32x2*)((uint16_t*)e.out + m * e.stride_om + n0) = o;
}


// dequant of an integer code q (as float) for W_group_mode (triton_kernels/utils.py:73-87),
// evaluated in fp32 on the stored scale / zero
__device__ __forceinline__ float dequant_f32(float q, float s, float z, int w_mode) {
    switch (w_mode) {
        case 1: return q - z;
        case 2: return q * s;
        case 3: return (q - z) * s;
        case 4: return __builtin_fmaf(q, s, z);
        default: return q;
    }
}

// (a, b) such that  sum_k x_k * dequant(q_k) = a * sum_k x_k q_k + b * sum_k x_k  inside one group
__device__ __forceinline__ void group_affine(float s, float z, int w_mode, float& a, float& b) {
    switch (w_mode) {
        case 1: a = 1.0f; b = -z; break;
        case 2: a = s; b = 0.0f; break;
        case 3: a = s; b = -z * s; break;
        case 4: a = s; b = z; break;
        default: a = 1.0f; b = 0.0f; break;
    }
}

// Workspace layout: [arrival counters: MAX_SPLITK_COUNTERS x u32 | slabs].  The counters sit at a FIXED
// place so that slab payload of one shape can never alias the counters of another; kernels leave them zero.
// The last 4096 words of the block belong to the opt-in timeline probes (tuning[3] & 4): never used as tickets.
constexpr int COUNTER_WORDS = 65536;
constexpr int PROBE_WORDS = 4096;
constexpr int MAX_SPLITK_COUNTERS = COUNTER_WORDS - PROBE_WORDS;
constexpr uint64_t COUNTER_BYTES = (uint64_t)COUNTER_WORDS * 4;

// ---------------------------------------------------------------------------------------------
// split-K hand-off words: write-through (sc1) stores / loads at agent scope
// (MI355X: per-XCD L2s are not coherent; see DESIGN.md "split-K combine")
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void slab_store(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float slab_load(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every storing wave drains its write-through stores, the block joins, one lane takes a ticket.
// Returns true in every thread of the LAST block to arrive for `counter`.
__device__ __forceinline__ bool splitk_arrive_is_last(unsigned* counter, unsigned nslices, unsigned* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        *lds_flag = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return *lds_flag == nslices - 1u;
}
__device__ __forceinline__ void splitk_reset(unsigned* counter) {
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// kernel parameter block shared by the packed-weight kernels
// ---------------------------------------------------------------------------------------------
struct WnParams {
    const void* x;
    const uint32_t* w;   // packed int32 words [K/e, N]
    const void* scales;
    const void* zeros;
    Epilogue epi;
    float* slabs;        // split-K partial sums (workspace)
    unsigned* counters;  // split-K arrival counters (workspace, zero between launches)
    int M, N, K;
    int group_size;      // K elements per metadata row; == K for channel-wise / none
    int w_mode;          // W_group_mode
    int meta_dt, zeros_dt;
    int zero_is_scalar;
    int splitk;          // K slices (gridDim.y)
    int rows_per_slice;  // packed rows per K slice
    int64_t stride_xm, stride_xk, stride_wk, stride_meta_g;
    int64_t stride_wn_b, stride_meta_n;  // block-scaled K-contiguous weights (gemm_wn_mma.hip MXW): bytes between weight rows / scale columns
    int combine;         // K-slice combine: 0 = slabs + arrival ticket (last block sums), 1 = reduce-scatter between co-resident
                         // slices (gemm_wn_mma.hip; needs every block of the launch resident at once)
    int flags;           // experiment switches forwarded from tuning[3] (kernel-specific)
    int gs_shift;        // log2(group_size); 31 when one metadata row spans all of K; -1 (not a power of two)
                         // sends the problem to the coverage kernel — an integer division per metadata load costs
                         // ~50 VALU instructions and a branch in kernels that have ~300 per chunk — except in the 8-wave tile kernel
                         // (round 6), whose metadata row index is wave-uniform: k / group = mulhi(k / 32, gs_magic) on the scalar unit
    uint32_t gs_magic;   // ceil(2^32 / (group_size / 32)) for a group size that is a multiple of 32 and not a power of two, else 0
};

__device__ __forceinline__ int group_of(int k, int gs_shift) { return k >> gs_shift; }
// the same for kernels that also take group sizes that are multiples of 32 and not a power of two (k a multiple of 32; gs_magic: see WnParams)
__device__ __forceinline__ int group_of(int k, int gs_shift, uint32_t gs_magic) {
    return gs_shift >= 0 ? (k >> gs_shift) : (int)__umulhi((uint32_t)k >> 5, gs_magic);
}

// Two-buffer software pipeline over n >= 1 units (chunks / pieces of K): unit i + 2 is requested as soon as unit i
// has been consumed, and nothing is requested twice — a clamped "re-request the last unit" tail would stream the
// whole weight matrix through L2 -> L1 up to three times when n == 2, which is the case for the 4096 x 4096 decode
// shape.  The steady-state loop has unconditional loads so the compiler's vmcnt bookkeeping stays exact; the
// last (up to three) units are peeled.
template <typename Buf, typename Load>
__device__ __forceinline__ void pipeline2_prime(int n, Buf& A, Buf& B, Load&& load) {
    load(A, 0);
    if (n > 1) load(B, 1);
}
template <typename Buf, typename Load, typename Compute>
__device__ __forceinline__ void pipeline2_run(int n, Buf& A, Buf& B, Load&& load, Compute&& compute) {
    int i = 0;
    for (; i + 4 <= n; i += 2) {
        // (round 3, scripts/isa_loops.py: hipcc 7.2 waits with s_waitcnt vmcnt(0) at the head of this loop in every gemv_wn_kernel
        // variant, i.e. both buffers are drained per iteration, and its scheduler makes up for it by hoisting the next requests above
        // the arithmetic.  Pinning the four phases with sched_barrier(0) and priming unconditionally removed neither the drain nor
        // its cause and took the hoisting away: 16384^2 M = 1 went 23.7 -> 27.9 us, 8192^2 10.4 -> 11.3 us.  Left to the compiler.)
        compute(A, i);
        load(A, i + 2);
        compute(B, i + 1);
        load(B, i + 3);
    }
    const int rem = n - i;  // 1, 2 or 3
    compute(A, i);
    if (rem >= 2) {
        if (rem == 3) load(A, i + 2);
        compute(B, i + 1);
        if (rem == 3) compute(A, i + 2);
    }
}

// parameter block of the coverage kernels (generic.hip)
struct GenericParams {
    const void* x;
    const void* w;
    const void* scales;
    const void* zeros;
    Epilogue epi;
    int M, N, K;
    int nbits, e, pack_bits;  // e == 1: unpacked
    int w_dt;                 // dtype of unpacked weights
    int x_dt;
    int group_size, w_mode, meta_dt, zeros_dt, zero_is_scalar;
    int int_acc;              // int8 x integer weights: exact int32 accumulation
    int64_t stride_xm, stride_xk, stride_wk, stride_wn, stride_meta_g, stride_meta_n;
    // split-K of the A8W8 MFMA kernel (workspace, same layout as WnParams)
    float* slabs;
    unsigned* counters;
    int splitk;
    int flags;
    // block-scaled (MX / NV) formats, gemm_mx.hip: element formats of x / w (MX_F16, MX_BF16, MX_FP8, MX_FP4), the
    // per-block scales of x (channel_scale_mode 4: e8m0 bytes [M_pad, K/32], or e4m3 bytes [M_pad, K/16] for NVFP4),
    // and the constant the accumulator is multiplied by at the end (NVFP4 meta scale squared, else 1)
    const void* sx_blocks;
    int64_t stride_sx_blk_m;
    int mx_x, mx_w, mx_scale_e4m3;
    float mx_post;
    // cooperative activation quantisation inside the launch (gl_coopquant.h): the caller's 16-bit activations, their row stride
    // (elements) and type.  `x` / `epi.scales_x` then point at the workspace copies (row stride K) that the blocks fill first;
    // `counters` [0, M) are the row flags, [M] the departure count.
    const void* cq_x;
    int64_t cq_stride_xm;
    int cq_xdt;
};
enum { MX_F16 = 1, MX_BF16 = 2, MX_FP8 = 3, MX_FP4 = 4 };

// host-side launch description produced by the dispatcher
// How many blocks of a one-block-per-CU kernel are resident at once on the current device (its CU count; 256 until a device
// has been seen).  Kernels whose blocks WAIT for each other (reduce-scatter combine) are only planned within this limit.
int resident_block_limit();

// scalar kernel arguments of gemv_w4_decode3_kernel (gemv_decode.hip): 14 dwords the command processor preloads into SGPRs
struct Decode3Args {
    const char *w, *x, *s, *z;
    uint16_t* out;
    uint32_t sw4, mstride2;
    int nch_total;
    uint32_t modes;
    unsigned* counters;
};

// scalar kernel arguments of gemm_w4_rows_kernel (gemm_wn_rows.hip): the same 14 preloaded dwords, then M and the row strides
struct Rows5Args {
    const char *w, *x, *s, *z;
    uint16_t* out;
    uint32_t sw4, mstride2;
    int nch_total;
    uint32_t modes;
    int M;
    uint32_t sxm2, som;
};

struct LaunchPlan {
    const void* fn;
    const char* name;
    dim3 grid, block;
    size_t lds_bytes;
    uint64_t ws_bytes;    // total workspace needed (COUNTER_BYTES + slab_bytes when K is split, else 0)
    uint64_t slab_bytes;
    int arg_kind;         // 0: the kernel takes its parameter struct by value | 1: the scalar arguments of `d3` | 2: those of `r5`
    Decode3Args d3;
    Rows5Args r5;
};

}  // namespace gl
