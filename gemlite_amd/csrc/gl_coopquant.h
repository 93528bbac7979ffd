// gl_coopquant.h — per-token activation quantisation done COOPERATIVELY inside the matmul launch (round 4).
//
// The reference runs scale_activations_per_token as its own launch in front of every dynamically quantised matmul
// (gemlite/core.py:155-175, quant_utils.py:268-347); on this part that is a 2.3-3 us kernel plus a 1.7 us launch boundary in front
// of a 7-14 us matmul.  At M = 1 every block re-quantises the one row itself (kmajor_fused_quant_kernel).  From 2 rows that would
// repeat M x K IEEE divisions in every one of ~256 blocks, so here the launch carries min(M, 64) extra PRODUCER blocks in front of
// the tile blocks: producer b quantises rows b, b + P, ... into the launch's workspace with write-through stores and raises one flag
// per row; every tile block requests its first weights, then waits for the M flags and only then touches the quantised rows — the
// chain (row load, amax, divide, write-through, flag, poll: ~3 us of dependent memory round trips) sits under the weight stream.
// (First form, measured: the tile blocks themselves quantised a row each before their weights were consumed — the chain then sat in
//  FRONT of a block's own weight wait and layer(x) was 3-5 us SLOWER than quantiser + matmul; profiles/r04/probe_fused_quant_v1.log.)
//
// Visibility (MI355X: per-CU L1 and per-XCD L2 are not coherent, cdna_hip_programming.md §6 G16): payload = agent-scope (sc1,
// write-through) stores, every storing wave drains (s_waitcnt vmcnt(0)), block barrier, ONE lane stores the flag; consumers poll
// the flags with agent-scope loads and read the payload with plain loads afterwards — no line of it can sit in a consumer's L1 / L2
// before the flag is up, because nobody reads the payload region earlier in the launch and the caches start a launch invalidated.
// No block waits for another one without bound: after STEAL_AFTER polls a waiting block quantises the first missing row ITSELF
// (same arithmetic, same bytes: a benign duplicate), so the launch finishes whatever the dispatch order, co-residency or CU mask.
// The flags are left zero for the next launch by the last block to leave (departure count), like the split-K tickets.
//
// Arithmetic: bit-identical to act_quant_per_token_kernel (generic.hip): s = max(amax / qmax, 1e-6) with IEEE division, x / s with
// IEEE division, clamp, floor(v + 0.5) for int8 (the reference's AMD rounding, quant_utils.py:259-266) / the hardware fp8 converters.
#pragma once
#include "gl_common.h"

namespace gl {
namespace cq {

constexpr int STEAL_AFTER = 24;   // polls (~0.6 us each) before a waiting block quantises a missing row itself
constexpr int TEST_NO_PRODUCE = 32768;  // GenericParams::flags (tuning[3]) bit, tests only: no block quantises its own rows — every row is stolen
constexpr int MAX_ROWS = 1024;    // flags live in the ticket words of the workspace: [0, M) row flags, [M] departures

// workspace bytes behind the counters: [M x K quantised bytes, padded to 256][M fp32 scales, padded to 256]
__host__ __device__ inline uint64_t xq_bytes(int64_t M, int64_t K) { return (uint64_t)((M * K + 255) & ~(int64_t)255); }
__host__ __device__ inline uint64_t payload_bytes(int64_t M, int64_t K) { return xq_bytes(M, K) + (uint64_t)((M * 4 + 255) & ~(int64_t)255); }

// One row by the whole block (512 threads; K % 8 == 0, 16-byte aligned rows).  `wmax`: 8 floats of LDS.
template <int QDT>
__device__ __forceinline__ void quantise_row(const GenericParams& p, int m, float* wmax) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t* row = (const uint16_t*)p.cq_x + (int64_t)m * p.cq_stride_xm;
    const bool f16 = p.cq_xdt == GEMLITE_DT_FP16;
    float amax = 0.f;
    for (int k = tid * 8; k < p.K; k += 512 * 8) {
        const u32x4 v = *(const u32x4*)(row + k);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint16_t hbits = (uint16_t)(v[e >> 1] >> (16 * (e & 1)));
            const float f = f16 ? F16Traits<half_tag>::to_float(hbits) : F16Traits<bf16_tag>::to_float(hbits);
            amax = fmaxf(amax, fabsf(f));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
    if (lane == 0) wmax[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])), fmaxf(fmaxf(wmax[4], wmax[5]), fmaxf(wmax[6], wmax[7])));
    constexpr float qmin = QDT == GEMLITE_DT_INT8 ? -128.f : (QDT == GEMLITE_DT_FP8E4 ? -448.f : -57344.f);
    constexpr float qmax = QDT == GEMLITE_DT_INT8 ? 127.f : (QDT == GEMLITE_DT_FP8E4 ? 448.f : 57344.f);
    const float sx = fmaxf(__fdiv_rn(amax, qmax), 1e-6f);
    uint8_t* yrow = (uint8_t*)p.x + (int64_t)m * p.K;  // the workspace copy the matmul reads (row stride K)
    for (int k = tid * 8; k < p.K; k += 512 * 8) {     // second pass: the row is in this CU's L1 / L2
        const u32x4 v = *(const u32x4*)(row + k);
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint16_t hbits = (uint16_t)(v[e >> 1] >> (16 * (e & 1)));
            const float f = f16 ? F16Traits<half_tag>::to_float(hbits) : F16Traits<bf16_tag>::to_float(hbits);
            t[e] = fminf(fmaxf(__fdiv_rn(f, sx), qmin), qmax);
        }
        uint32_t q[2] = {0u, 0u};
        if constexpr (QDT == GEMLITE_DT_INT8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e >> 2] |= (uint32_t)(uint8_t)(int8_t)floorf(t[e] + 0.5f) << (8 * (e & 3));
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int w = 0;
                if constexpr (QDT == GEMLITE_DT_FP8E4) {
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(t[4 * h], t[4 * h + 1], w, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(t[4 * h + 2], t[4 * h + 3], w, true);
                } else {
                    w = __builtin_amdgcn_cvt_pk_bf8_f32(t[4 * h], t[4 * h + 1], w, false);
                    w = __builtin_amdgcn_cvt_pk_bf8_f32(t[4 * h + 2], t[4 * h + 3], w, true);
                }
                q[h] = (uint32_t)w;
            }
        }
        __hip_atomic_store((unsigned long long*)(yrow + k), ((unsigned long long)q[1] << 32) | q[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) __hip_atomic_store((float*)p.epi.scales_x + m, sx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores have been acknowledged
    __syncthreads();
    if (tid == 0) __hip_atomic_store(p.counters + m, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Producer blocks (the first `nprod` blocks of the grid; they compute no tile): row b, b + nprod, ...  They are the first blocks the
// dispatcher places, hold no weight requests and leave as soon as their flags are up.
template <int QDT>
__device__ __forceinline__ void produce(const GenericParams& p, int nprod, float* lds) {
    if (p.flags & TEST_NO_PRODUCE) return;
    for (int m = blockIdx.x; m < p.M; m += nprod) quantise_row<QDT>(p, m, lds);
}

// Consumer blocks: wait until every row of the launch is up.  Wave 0 polls the flags (agent-scope loads, nothing else of the block
// runs); the other waves wait at the barrier.  After STEAL_AFTER polls without success the whole block quantises the first missing
// row itself and polling resumes.  `lds`: 16 words, free for the duration.  (Call it AFTER the block's first weight requests are out
// and BEFORE anything reads p.x / p.epi.scales_x.)
__device__ __forceinline__ int first_missing(const GenericParams& p) {  // wave 0, all 64 lanes
    int first = -1;
    for (int base = 0; base < p.M; base += 64) {
        const int m = base + (int)threadIdx.x;
        const unsigned f = m < p.M ? __hip_atomic_load(p.counters + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
        const unsigned long long down = __ballot(f == 0u);
        if (down != 0ull && first < 0) first = base + __builtin_ctzll(down);
    }
    return first;
}
template <int QDT>
__device__ __forceinline__ void wait_rows(const GenericParams& p, float* lds) {
    volatile int* pend = (volatile int*)(lds + 8);
    for (int round = 0;; ++round) {
        if (threadIdx.x < 64) {
            int miss = first_missing(p);
            for (int it = 1; miss >= 0 && it < STEAL_AFTER; ++it) {
                __builtin_amdgcn_s_sleep(2);
                miss = first_missing(p);
            }
            if (threadIdx.x == 0) pend[round & 1] = miss;
        }
        __syncthreads();
        const int miss = pend[round & 1];
        if (miss < 0) break;
        quantise_row<QDT>(p, miss, lds);  // (block-uniform: every thread read the same slot)
    }
}

// Last statement of EVERY block of the launch (producers included: a producer that was placed late — its rows long stolen — still
// raises its flags, and they must be down again before the next launch polls them): this block will not read the rows, the scales
// or the flags again and every flag it raised has been acknowledged.  The last block to leave zeroes the flags and the count.
__device__ __forceinline__ void depart(const GenericParams& p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a flag this block raised has been acknowledged before its departure counts)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.counters + p.M, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1u)
            for (int m = 0; m <= p.M; ++m) __hip_atomic_store(p.counters + m, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace cq
}  // namespace gl
