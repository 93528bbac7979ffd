// gemm_wn_mma_bf16.hip — the 8-wave MFMA tile kernel (gemm_wn_mma_kernel.inc) instantiated for bf16_tag: one translation unit per
// 16-bit type so that the two halves of the ~100 instantiations compile in parallel.
#include "gemm_wn_mma_kernel.inc"

namespace gl {
const void* mma_lookup_bf16(int kind, int nbits, int mi, int xdt, int xch) { return mma_lookup<bf16_tag>(kind, nbits, mi, xdt, xch); }
}  // namespace gl
