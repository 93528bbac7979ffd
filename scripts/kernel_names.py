"""Which kernel the C ABI picks for a grid of shapes (no GPU needed: gemlite_hip_kernel_name never launches)."""
import ctypes
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemlite_amd import _hip

lib = _hip.load()


def name(M, N, K, nbits=4, gs=128, dt=1, mt=-1, tuning=(0, 0, 0, 0)):
    a = _hip.ForwardArgs(); a.struct_size = ctypes.sizeof(a); a.matmul_type = mt
    a.x = a.w_q = a.out = a.scales = a.zeros = 0x10000
    a.M, a.N, a.K = M, N, K
    e = 32 // nbits
    a.W_nbits, a.group_size, a.unpack_mask, a.elements_per_sample, a.w_pack_bits, a.w_dtype = nbits, gs, 2 ** nbits - 1, e, 32, 6
    a.input_dtype = a.output_dtype = a.meta_dtype = a.zeros_dtype = dt
    a.W_group_mode = 4
    a.stride_xm, a.stride_xk, a.stride_wk, a.stride_wn, a.stride_om, a.stride_on = K, 1, N, 1, N, 1
    a.stride_meta_g, a.stride_meta_n = N, 1
    for i in range(4):
        a.tuning[i] = tuning[i]
    return lib.gemlite_hip_kernel_name(ctypes.byref(a)).decode()


if __name__ == "__main__":
    for nbits in (4, 2):
        for K, N in [(4096, 4096), (11008, 4096), (8960, 1536), (4096, 11008), (14336, 4096), (4096, 14336), (8192, 8192), (16384, 16384)]:
            print(f"W{nbits} K={K} N={N}: " + " | ".join(f"M={M}: {name(M, N, K, nbits)}" for M in (1, 2, 8, 32, 64, 256)))
