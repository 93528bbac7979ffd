import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gemlite_amd, numpy as np
from gemlite_amd import GemLiteLinear, DType, helper, core
from gemlite_amd.bench_utils import kernel_device_us
from oracle import gemlite_oracle as O
lins=[]
for i in range(8):
    W_q,s,z = O.gen_data(4096,4096,4,128,seed=i)
    lin = GemLiteLinear(4,128,4096,4096,DType.FP16,DType.FP16); lin.pack(torch.from_numpy(W_q).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(z).cuda()); lins.append(lin)
for M in (24, 32, 33, 48, 64, 96, 128):
    x = torch.from_numpy(O.gen_x(M,4096,seed=M)).cuda()
    i=[0]
    def run(mt):
        def f():
            l=lins[i[0]%8]; i[0]+=1
            core._hip_matmul(x, l.W_q, l.scales, l.zeros, None, l.get_meta_args(), mt)
        return f
    print(M, 'auto', round(kernel_device_us(run(-1), 48),2), 'GEMM_SPLITK', round(kernel_device_us(run(3),48),2), 'GEMM', round(kernel_device_us(run(4),48),2))
print(helper.autotune_layer(lins[0], batch_sizes=(1,2,4,16,48,256), iters=30))
