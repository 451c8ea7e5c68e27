"""BASELINE configs[0] (N = 512, batch 1, lowest 6, fp64: the reference's benchmarks_solve.py shape) through the native
Davidson, for rocprofv3 --kernel-trace --stats: where the 90 ms of a call go (GPU kernels vs host).  Prints wall times."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd.linalg import symeig
from tests import cases
dev = torch.device("cuda:0")
m1 = cases.random_symmetric(512, -1.0, 1.0, 123).to(dev)
A = xa.LinearOperator.m(m1, is_hermitian=True)
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ts = []
for i in range(nrep + 1):
    tr = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        ev, X = symeig(A, neig=6, mode="lowest", method="davidson", min_eps=1e-8, trace=tr)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(json.dumps({"ms": [round(t, 2) for t in ts], "niter": tr["niter"], "basis": tr["basis_size"], "groups": tr["groups"]}))
