"""K1w (MFMA wide-panel product) for the rocprofv3 PMC pass (MFMA busy)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xitorch_amd.kernels import dense_wide
dev = torch.device("cuda:0")
for (dt, B, N) in ((torch.float32, 4, 32768), (torch.float64, 8, 16384)):
    A = torch.empty(B, N, N, dtype=dt, device=dev).uniform_(-1, 1)
    X = torch.randn(B, 32, N, dtype=dt, device=dev)
    Y = torch.empty_like(X)
    for _ in range(3):
        dense_wide(A, X, out=Y)
    torch.cuda.synchronize()
    del A
    torch.cuda.empty_cache()
