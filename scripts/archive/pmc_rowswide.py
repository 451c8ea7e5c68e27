"""K1wr (row-orientation MFMA panel product) for the rocprofv3 PMC passes (MFMA busy, HBM traffic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xitorch_amd.kernels import dense_rows_wide
dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B, N = 16, 16384
A = torch.empty(B, N, N, dtype=torch.float64, device=dev).uniform_(-1, 1)
X = torch.randn(B, P, N, dtype=torch.float64, device=dev)
Y = torch.empty_like(X)
for _ in range(3):
    dense_rows_wide(A, X, out=Y)
torch.cuda.synchronize()
