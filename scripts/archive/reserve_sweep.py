"""In-process sweep of the CUs the panel-product stream leaves to the small kernels (davidson(reserve_cus=...)) on the
headline workload: same operators, same process (same physical placement), configurations interleaved."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic
from xitorch_amd.linalg.native_eig import davidson
dev = torch.device("cuda:0")
B, N, p = 64, 16384, 6
mat = torch.empty((B, N, N), dtype=torch.float64, device=dev)
synthetic.dense_symmetric(B, N, "S1", dtype=torch.float64, device=dev, out=mat)
A = xa.LinearOperator.m(mat, is_hermitian=True)
cfgs = [int(v) for v in sys.argv[1:]] or [64, 32, 16, 96, 0]
res = {c: [] for c in cfgs}
for rnd in range(4):
    for c in cfgs:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            davidson(A, p, "lowest", min_eps=1e-8, rng_device="device", reserve_cus=c)
        torch.cuda.synchronize()
        if rnd:
            res[c].append(round((time.perf_counter() - t0) * 1e3, 2))
print(json.dumps({"ms_per_call_by_reserved_cus": res}))
