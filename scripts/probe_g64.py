import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gemlite_amd, numpy as np
from gemlite_amd import GemLiteLinear, DType, core
from gemlite_amd.bench_utils import kernel_device_us
from oracle import gemlite_oracle as O
lins=[]
for i in range(8):
    W_q,s,z = O.gen_data(4096,4096,4,64,seed=i)
    lin = GemLiteLinear(4,64,4096,4096,DType.FP16,DType.FP16); lin.pack(torch.from_numpy(W_q).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(z).cuda()); lins.append(lin)
for M in (2, 8, 16):
    x = torch.from_numpy(O.gen_x(M,4096,seed=M)).cuda()
    i=[0]
    def run(t):
        def f():
            l=lins[i[0]%8]; i[0]+=1
            core._hip_matmul(x, l.W_q, l.scales, l.zeros, None, l.get_meta_args(), -1, t)
        return f
    print('gs64 M', M, 'direct(auto)', round(kernel_device_us(run((0,0,0,0)), 48),2), 'stream', round(kernel_device_us(run((0,0,1,0)),48),2))
