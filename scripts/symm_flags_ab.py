"""K1s: non-temporal handling of the row / column partials (xk_dense_symm_set_flags: 0 plain, 1 nt stores, 3 nt stores +
nt loads in the fold) — same process, same buffers, interleaved: half-batch and whole-batch launches, pipelined call."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic, kernels as K
from xitorch_amd._capi import fn
from xitorch_amd.linalg.native_eig import davidson
dev = torch.device("cuda:0")
B, N, p = 64, 16384, 6
mat = torch.empty((B, N, N), dtype=torch.float64, device=dev)
synthetic.dense_symmetric(B, N, "S1", dtype=torch.float64, device=dev, out=mat)
X = torch.randn((B, p, N), dtype=torch.float64, device=dev); Y = torch.empty_like(X)
A = xa.LinearOperator.m(mat, is_hermitian=True)
h = B // 2
def ev_time(f, reps=4):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 3)
res = {}
ref = None
for rnd in range(3):
    for fl in (0, 1, 3):
        fn("xk_dense_symm_set_flags")(fl)
        r = res.setdefault(fl, {"half": [], "whole": [], "call": []})
        r["half"].append(ev_time(lambda: K.dense_symm(mat[:h], X[:h], out=Y[:h])))
        r["whole"].append(ev_time(lambda: K.dense_symm(mat, X, out=Y)))
        if ref is None: ref = Y.clone()
        assert torch.allclose(Y, ref, rtol=1e-12, atol=1e-9)
        ts = []
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad(): davidson(A, p, "lowest", min_eps=1e-8, rng_device="device")
            torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
        r["call"].append(ts[-1])
fn("xk_dense_symm_set_flags")(3)
print(json.dumps(res))
