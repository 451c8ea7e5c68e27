"""Does the time of the headline call depend on WHERE the operator batch lives?  One process, the 137 GB batch is
allocated, generated, timed and released several times (optionally behind a dummy allocation that shifts it)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic
from xitorch_amd.linalg.native_eig import davidson
dev = torch.device("cuda:0")
B, N, p = 64, 16384, 6
out = []
for trial, shift_mb in enumerate([0, 0, 0, 0, 0, 0]):
    dummy = torch.empty(shift_mb << 20, dtype=torch.uint8, device=dev) if shift_mb else None
    mat = torch.empty((B, N, N), dtype=torch.float64, device=dev)
    synthetic.dense_symmetric(B, N, "S1", dtype=torch.float64, device=dev, out=mat)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    from xitorch_amd import kernels as K
    X = torch.randn((B, p, N), dtype=torch.float64, device=dev); Y = torch.empty_like(X)
    K.dense_symm(mat, X, out=Y); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2):
        K.dense_symm(mat, X, out=Y)
    e1.record(); torch.cuda.synchronize()
    probe = e0.elapsed_time(e1) / 2
    del X, Y
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            davidson(A, p, "lowest", min_eps=1e-8, rng_device="device")
        torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    out.append({"trial": trial, "shift_MB": shift_mb, "ptr_mod_1GB_MB": (mat.data_ptr() % (1 << 30)) >> 20,
                "ptr_hex": hex(mat.data_ptr()), "k1s_probe_ms": round(probe, 3), "ms": ts[1:]})
    print(json.dumps(out[-1]), flush=True)
    del A, mat, dummy
    torch.cuda.empty_cache()
