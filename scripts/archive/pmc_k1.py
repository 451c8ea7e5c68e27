"""K1 alone at the config-2 shape, for the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) and kernel stats.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o k1 -- python scripts/pmc_k1.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xitorch_amd.kernels import dense_mm
dev = torch.device("cuda:0")
B = int(os.environ.get("K1_B", "64")); N = 16384; P = 6
A = torch.empty(B, N, N, dtype=torch.float64, device=dev).uniform_(-1, 1)
X = torch.randn(B, P, N, dtype=torch.float64, device=dev)
Y = torch.empty_like(X)
for _ in range(3):
    dense_mm(A, X, out=Y, trans=True)      # column-oriented variant (what symeig uses for Hermitian A)
for _ in range(3):
    dense_mm(A, X, out=Y, trans=False)     # row-sweep variant
torch.cuda.synchronize()
print("done", B, N, P)
