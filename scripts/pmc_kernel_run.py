"""A few standalone launches of one secondary config's dominant kernel at the config's shape, for the rocprofv3 PMC passes
(scripts/sessions/r05d.sh -> scripts/pmc_collect.py -> profiles/c3_pmc_traffic.json / c4_pmc_traffic.json).
    python scripts/pmc_kernel_run.py c3|c4"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic as syn
dev = torch.device("cuda:0")
which = sys.argv[1]
if which == "c3":
    B, N, hb = 256, 65536, 63
    band = syn.banded(B, N, hb=hb, device=dev)
    x = syn.banded_rhs_solution(B, N, device=dev)
    A = xa.BandedLinearOperator(band)
    with torch.no_grad():
        for _ in range(4):
            y = A.mm(x)
elif which == "c4":
    B, N = 64, 8192
    mat = syn.root_matrix(B, N, device=dev) * 2.0
    y0 = torch.randn(B, N, dtype=torch.float64, device=dev)
    op = xa.LinearOperator.m(mat, is_hermitian=False)
    with torch.no_grad():
        for _ in range(4):
            z = op.mv(y0)
else:
    raise SystemExit("c3 or c4")
torch.cuda.synchronize()
