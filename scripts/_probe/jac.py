import os, sys, json, time
sys.path.insert(0, "/root/repo")
import torch
import xitorch_amd as xa
from xitorch_amd import kernels as K
from xitorch_amd.linalg import symeig
from tests import cases
dev = torch.device("cuda:0")
m1 = cases.random_symmetric(512, -1.0, 1.0, 123).to(dev)
A = xa.LinearOperator.m(m1, is_hermitian=True)
tr = {}
for _ in range(2):
    tr = {}
    ev, X = symeig(A, neig=6, mode="lowest", method="davidson", min_eps=1e-8, trace=tr)
print({k: tr.get(k) for k in ("niter", "k3_fallbacks", "basis_size")})
def t_of(f, n=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for k in (6, 12, 18, 24, 48, 96, 120):
    g = torch.Generator().manual_seed(k)
    R = torch.randn(1, k, k, dtype=torch.float64, generator=g)
    T = (R + R.transpose(1, 2)).to(dev)
    rec = {"k": k, "jacobi_ms": round(t_of(lambda: K.small_eigh(T, k, 6)), 4)}
    if k >= 8:
        rec["tri_ms"] = round(t_of(lambda: K.small_eigh(T, k, 6, method="tri")), 4)
        rec["k3p_ms"] = round(t_of(lambda: K.small_eigh_big(T, k, 6, algo=3)), 4)
    print(json.dumps(rec))
