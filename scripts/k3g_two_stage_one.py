"""one configuration of K3g (order, batch, algo from the command line) a few times: for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K
k, B, algo = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dtype = torch.float32 if len(sys.argv) > 4 and sys.argv[4] == "f32" else torch.float64
p = int(sys.argv[5]) if len(sys.argv) > 5 else 6
dev = torch.device("cuda:0")
R = torch.randn(B, k, k, dtype=torch.float64, generator=torch.Generator().manual_seed(k))
T = (R + R.transpose(1, 2)).to(dtype).to(dev)
for _ in range(5):
    K.small_eigh_big(T, k, p, algo=algo)
torch.cuda.synchronize()
