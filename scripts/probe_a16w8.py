import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gemlite_amd
from gemlite_amd import helper as H, core
from gemlite_amd.bench_utils import kernel_device_us
torch.manual_seed(0)
lins=[H.A16W8(device="cuda:0").from_weights((torch.randn(4096,4096)/30).half()) for _ in range(8)]
for M in (1, 8):
    x=(torch.randn(M,4096)/10).half().cuda(); i=[0]
    def run(t):
        def f():
            l=lins[i[0]%8]; i[0]+=1
            core._hip_matmul(x, l.W_q, l.scales, l.zeros, None, l.get_meta_args(), -1, t)
        return f
    print('A16W8 int8 4096^2 M', M, 'us', round(kernel_device_us(run(None), 32),2))
