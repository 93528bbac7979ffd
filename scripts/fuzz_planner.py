"""CPU-only fuzz of the planner through the C ABI (gemlite_hip_query / kernel_name / workspace_bytes never launch): random layer
families x M x (N, K) x tuning.  Checks: no crash, a kernel name and a sane workspace for every accepted request, and — on
"realistic" shapes (N % 128 == 0, K % 256 == 0, both >= 1024) with default tuning — which requests still reach a coverage kernel.
    python scripts/fuzz_planner.py [seed] [iterations]"""
import collections, ctypes as C, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gemlite_amd import _hip  # noqa: E402
import test_host_cpu as T  # noqa: E402

FAMS = {"a16w4": dict(), "a16w2": dict(nbits=2), "a16w1": dict(nbits=1), "a16w8p": dict(nbits=8), "a16w4bf": dict(in_dt=2),
        "a16w8i": dict(nbits=8, e=1, in_dt=1, w_dtype=4, w_mode=2, gsK=1), "a16w8f": dict(nbits=8, e=1, in_dt=2, w_dtype=3, w_mode=0, c_mode=1, gsK=1),
        "a8w8i": dict(nbits=8, e=1, in_dt=4, w_mode=0, c_mode=3, out_dt=1, gsK=1), "a8w8f": dict(nbits=8, e=1, in_dt=3, w_mode=0, c_mode=3, out_dt=2, gsK=1),
        "a8w4": dict(in_dt=3, out_dt=2, meta_dt=2, zeros_dt=2, w_mode=4, c_mode=2), "a8w2": dict(nbits=2, in_dt=3, out_dt=1, meta_dt=1, zeros_dt=1, w_mode=3, c_mode=2),
        "bitnet8": dict(nbits=2, in_dt=4, out_dt=1, meta_dt=0, zeros_dt=6, zero_scalar=1, w_mode=1, c_mode=3, gsK=1),
        "bitnet16": dict(nbits=2, in_dt=1, meta_dt=0, zeros_dt=6, zero_scalar=1, w_mode=1, c_mode=1, gsK=1)}
MXF = {"mx88": (16, 8, 4, 32), "mx84": (16, 4, 2, 32), "mx84b": (16, 4, 4, 32), "mx44": (17, 4, 4, 32), "mx16w8": (15, 8, 0, 32), "mx16w4": (14, 4, 0, 32),
       "nv": (18, 4, 4, 16), "mx88pt": (16, 8, 2, 32)}
DIMS = [1, 2, 3, 4, 5, 8, 16, 17, 22, 23, 32, 33, 40, 48, 64, 65, 96, 128, 200, 256, 300, 384, 385, 512, 513, 700, 1024, 2048, 4096]
NK = [64, 128, 192, 256, 512, 1024, 1280, 1536, 2048, 2560, 3072, 4096, 4224, 5120, 8192, 8960, 11008, 13824, 14336, 16384, 28672]
TUN = [0, 1, 2, 3, 4, 5, 6, 7, 8, 32, 34]


def run(seed=0, iters=100000, lib=None):
    """-> (kernels chosen, {family: [(M, N, K) realistic default-tuning requests on a coverage kernel]})"""
    lib = lib or _hip.load()
    rnd = random.Random(seed)
    buf = (C.c_uint8 * 64)()
    ptr = C.addressof(buf) // 16 * 16 + 16
    cnt, cov = collections.Counter(), collections.defaultdict(list)
    for _ in range(iters):
        M, N, K = rnd.choice(DIMS), rnd.choice(NK), rnd.choice(NK)
        tun = tuple(rnd.choice(TUN) if rnd.random() < 0.15 else 0 for _ in range(3)) + (rnd.choice([0, 0, 0, 16, 64, 128, 1024, 2048, 4096, 16384]),)
        if rnd.random() < 0.5:
            tun = (0, 0, 0, 0)
        if rnd.random() < 0.6:
            f = rnd.choice(list(FAMS))
            kw = dict(FAMS[f])
            gsk = kw.pop("gsK", 0)
            gs = K if gsk else rnd.choice([32, 64, 128, 128, 256])
            if K % gs:
                continue
            a = T._args(M=M, N=N, K=K, gs=gs, tuning=tun, mt=rnd.choice([-1, -1, -1, 0, 1, 2, 3, 4]) if any(tun) else -1, **kw)
            if kw.get("c_mode", 0) in (2, 3):
                a.scales_x = 0x1000
            f += "" if gsk else "/g%d" % gs
        else:
            f = rnd.choice(list(MXF))
            in_dt, nbits, c_mode, group = MXF[f]
            if K % group:
                continue
            a = _hip.ForwardArgs()
            a.struct_size = C.sizeof(_hip.ForwardArgs)
            a.matmul_type = -1
            a.x = a.w_q = a.scales = a.out = a.scales_x = ptr
            a.M, a.N, a.K = M, N, K
            a.W_nbits, a.group_size, a.unpack_mask = nbits, group, 2 ** nbits - 1
            a.elements_per_sample = 1 if nbits == 8 else 2
            a.w_pack_bits = 0 if nbits == 8 else 8
            a.w_dtype = 3 if nbits == 8 else 5
            a.input_dtype, a.output_dtype, a.meta_dtype = in_dt, (1 if in_dt in (14, 18) else 2), 5
            a.channel_scale_mode, a.W_group_mode = c_mode, 0
            a.stride_xm, a.stride_xk = (K // 2 if in_dt in (17, 18) else K), 1
            a.stride_wk, a.stride_wn = 1, (K if nbits == 8 else K // 2)
            a.stride_om, a.stride_on = N, 1
            a.stride_meta_g, a.stride_meta_n = N, 1
            a.stride_sx_m = K // group if c_mode == 4 else 1
            for i in range(4):
                a.tuning[i] = tun[i]
        st = lib.gemlite_hip_query(C.byref(a))
        cnt["status %d" % st] += 1
        if st != 0:
            continue
        name = lib.gemlite_hip_kernel_name(C.byref(a)).decode()
        ws = lib.gemlite_hip_workspace_bytes(C.byref(a))
        assert name and ws < (1 << 36), (f, M, N, K, tun, name, ws)
        cnt[name.split("<")[0]] += 1
        if "generic" in name and not any(tun) and N % 128 == 0 and K % 256 == 0 and N >= 1024 and K >= 1024:
            cov[f].append((M, N, K))
    return cnt, cov


if __name__ == "__main__":
    cnt, cov = run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 100000)
    print(dict(cnt.most_common()))
    for f in sorted(cov):
        print("coverage kernel on realistic shapes:", f, len(cov[f]), sorted(set(m for m, _, _ in cov[f]))[:12])
