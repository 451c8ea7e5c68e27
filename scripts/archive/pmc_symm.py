import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xitorch_amd.kernels import dense_mm, dense_symm
dev = torch.device("cuda:0")
B, N = 16, 16384
P = int(os.environ.get("SYMM_P", "6"))
A = torch.empty(B, N, N, dtype=torch.float64, device=dev).uniform_(-1, 1)
for b in range(B):
    A[b] = A[b] + A[b].T.clone()
X = torch.randn(B, P, N, dtype=torch.float64, device=dev)
Y = torch.empty_like(X)
for _ in range(3):
    dense_symm(A, X, out=Y)
for _ in range(2):
    dense_mm(A, X, out=Y, trans=True)
torch.cuda.synchronize()
