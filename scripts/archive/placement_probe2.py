"""Which allocation carries the two timing modes of the headline call: the operator batch or the solver's workspace?
(a) the batch stays, the workspace is released and re-allocated between trials; (b) the reverse order of allocation."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic
from xitorch_amd.linalg.native_eig import davidson
dev = torch.device("cuda:0")
B, N, p = 64, 16384, 6


def timed(A, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            davidson(A, p, "lowest", min_eps=1e-8, rng_device="device")
        torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    return ts[1:]


mat = torch.empty((B, N, N), dtype=torch.float64, device=dev)
synthetic.dense_symmetric(B, N, "S1", dtype=torch.float64, device=dev, out=mat)
A = xa.LinearOperator.m(mat, is_hermitian=True)
for trial in range(6):
    print(json.dumps({"mode": "batch fixed, workspace re-allocated", "trial": trial, "ms": timed(A),
                      "reserved_GB": round(torch.cuda.memory_reserved() / 2**30, 2)}), flush=True)
    torch.cuda.empty_cache()
    if trial % 2 == 1:      # shift the next workspace allocation
        pad = torch.empty((trial + 1) * (256 << 20), dtype=torch.uint8, device=dev)
    else:
        pad = None
