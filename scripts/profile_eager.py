"""where the host time of an eager layer(x) call goes (cProfile over 20000 calls, M = 1 A16W4 4096^2)"""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ["bench.py"]
import bench
from gemlite_amd import _hip
lib = _hip.load()
dev = torch.device("cuda", 0)
r = bench.Runner("a16w4_4096_m1", dev, lib, layers=4, use_graph=False)
lin, x = r.mods[0], r.x
for _ in range(200): lin(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20000): lin(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host us per call: %.2f" % ((t1 - t0) / 20000 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(20000): lin(x)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
