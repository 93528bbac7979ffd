// gl_async.h — memory requests that the compiler must not wait for, and the counted waits that retire them.
#pragma once
#include "gl_common.h"

namespace gl {
namespace async {

// ---- memory requests the compiler must NOT wait for.  hipcc answers any use of a tracked load with vmcnt(0) while an
// LDS-DMA is outstanding, which would drain the x / weight pipeline every step; these requests are therefore issued from
// inline asm and retired by hand with COUNTED s_waitcnt (cdna_hip_programming.md §5.7): loads return in issue order, so
// "vmcnt(n)" = everything but the newest n requests of this wave has landed.  A loaded register is handed to the compiler
// with tie() right after the wait that covers it.  s_nop 4: SALU-written SGPR / M0 -> VMEM wait states (nothing inside an
// asm string is padded by the compiler).
typedef uint32_t srd_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ srd_t make_srd(const void* base, uint32_t bytes) {
    const uint64_t b = (uint64_t)base;
    srd_t r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xFFFFu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void req_u32(uint32_t& dst, srd_t rs, uint32_t voff, uint32_t soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void req_u16(uint32_t& dst, srd_t rs, uint32_t voff, uint32_t soff) {
    asm volatile("s_nop 4\n\tbuffer_load_ushort %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// the same without the nop: soffset computed by SALU instructions (s_mul / s_add / s_mov) is interlocked; only a VALU-written SGPR
// (v_readfirstlane) needs the 5 wait states
__device__ __forceinline__ void req_u32n(uint32_t& dst, srd_t rs, uint32_t voff, uint32_t soff) {
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void req_u8(uint32_t& dst, srd_t rs, uint32_t voff, uint32_t soff) {
    asm volatile("s_nop 4\n\tbuffer_load_ubyte %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
typedef uint32_t u128_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void req_u128(u128_t& dst, srd_t rs, uint32_t voff, uint32_t soff) {
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// one LDS-DMA piece: 64 lanes x 16 bytes from buffer offset (voff per lane + soff) to LDS [lds_addr, +1024), lane-linear
// (m0 is a RESERVED register of the AMDGPU backend: hipcc never keeps a value in it across statements, it writes m0 right
//  before the few instructions that read it — none of which these kernels contain; `grep m0` on the generated code of the
//  three files that use this helper shows only the moves below.  The clobber is declared anyway; clang warns that clobbers of
//  reserved registers are advisory, hence the pragma.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void req_lds16(srd_t rs, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
    // m0 is declared clobbered, not saved / restored (two SALU + a 5-cycle nop per piece in the round-2 loops); one wait state
    // between the SALU write of M0 and the LDS-DMA that reads it
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 : : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
// the same reading around the (non-coherent) L1 / XCD L2: the consumer side of a write-through (sc1) hand-off
__device__ __forceinline__ void req_lds16_sc1(srd_t rs, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen sc1 lds"
                 : : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
// the same with 4 bytes per lane: 64 lanes x 4 bytes to LDS [lds_addr, +256)
__device__ __forceinline__ void req_lds4(srd_t rs, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %0, %1, %2 offen lds"
                 : : "v"(voff), "s"(rs), "s"(soff), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop
// K order of the row tiles that share a weight column tile (round 6; gemm_a8w8_sq_kernel has the reasoning).  flags bit 30 = on; bits 24 .. 27 = 0: row tile
// mt starts at step mt nsteps / mtiles (whole-K rotation); = 1 + log2(G): the K steps form groups of G, and inside every run of mtiles groups row tile mt takes
// them in the order mt, mt + 1, ... — each tile LEADS (pulls HBM-cold lines) on one group of the run and follows its siblings on the others, so the lines only have
// to survive mtiles - 1 groups in L2 instead of a whole rotation.  A tail of fewer than mtiles groups keeps the plain order.
// Everything per step is SCALAR arithmetic (the quotient by mtiles through a multiply-high with a constant set up once): the step offset feeds the soffset operand
// of inline-asm requests, and an SGPR written by v_readfirstlane right in front of them would need wait states nothing inserts.
struct KOrder {
    int mode;        // 0 plain, 1 whole-K rotation, 2 grouped
    int rot, nsteps; // whole rotation
    int gsh, P, mt, ngroups;
    uint32_t magic;  // ceil(2^32 / P)
    __device__ __forceinline__ void init(int mt_, int mtiles, int nsteps_, int flags) {
        mode = 0; rot = 0; nsteps = nsteps_; gsh = 0; P = mtiles; mt = mt_; ngroups = 0; magic = 0u;
        if (!(flags & (1 << 30)) || mtiles < 2) return;
        const int gm = (flags >> 24) & 15;
        if (gm == 0) {
            mode = 1;
            rot = __builtin_amdgcn_readfirstlane((mt_ * nsteps_) / mtiles);
        } else {
            mode = 2;
            gsh = gm - 1;
            ngroups = nsteps_ >> gsh;
            magic = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(((1ull << 32) + (uint64_t)mtiles - 1ull) / (uint64_t)mtiles));
        }
    }
    __device__ __forceinline__ int at(int step) const {
        if (mode == 0) return step;
        if (mode == 1) {
            const int k = step + rot;
            return k >= nsteps ? k - nsteps : k;
        }
        int g = step >> gsh;
        const int i = step & ((1 << gsh) - 1);
        const int base = (int)__umulhi((uint32_t)g, magic) * P;  // (g / P) * P: exact while g P < 2^32
        if (base + P <= ngroups) {
            int r = g - base + mt;
            if (r >= P) r -= P;
            g = base + r;
        }
        return (g << gsh) + i;
    }
};
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void tie(uint32_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ uint32_t lds_addr_of(const unsigned char* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
}


}  // namespace async
}  // namespace gl
