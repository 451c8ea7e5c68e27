"""K1s (upper-triangle panel product) alone at the config-2 shape for the rocprofv3 PMC passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd.kernels import dense_symm
from xitorch_amd import synthetic
dev = torch.device("cuda:0")
B, N, P = 64, 16384, 6
A = torch.empty(B, N, N, dtype=torch.float64, device=dev)
synthetic.dense_symmetric(B, N, "S1", device=dev, out=A)
X = torch.randn(B, P, N, dtype=torch.float64, device=dev)
Y = torch.empty_like(X)
for _ in range(3):
    dense_symm(A, X, out=Y)
torch.cuda.synchronize()
