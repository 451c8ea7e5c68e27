"""Per placement of the operator batch (released and re-allocated between trials): K1s alone on each half, on the
whole batch, the pipelined symeig call, the same call with the full-matrix kernel, and a bare read of each half."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic, kernels as K
from xitorch_amd.linalg.native_eig import davidson
dev = torch.device("cuda:0")
B, N, p = 64, 16384, 6


def ev_time(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 3)


def call_ms(A, **kw):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            davidson(A, p, "lowest", min_eps=1e-8, rng_device="device", **kw)
        torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    return ts[1:]


for trial in range(4):
    mat = torch.empty((B, N, N), dtype=torch.float64, device=dev)
    synthetic.dense_symmetric(B, N, "S1", dtype=torch.float64, device=dev, out=mat)
    X = torch.randn((B, p, N), dtype=torch.float64, device=dev); Y = torch.empty_like(X)
    h = B // 2
    out = {"trial": trial, "ptr": hex(mat.data_ptr())}
    out["k1s_first_half_ms"] = ev_time(lambda: K.dense_symm(mat[:h], X[:h], out=Y[:h]))
    out["k1s_second_half_ms"] = ev_time(lambda: K.dense_symm(mat[h:], X[h:], out=Y[h:]))
    out["k1s_whole_ms"] = ev_time(lambda: K.dense_symm(mat, X, out=Y))
    out["read_first_half_ms"] = ev_time(lambda: K.stream_read(mat[:h]))
    out["read_second_half_ms"] = ev_time(lambda: K.stream_read(mat[h:]))
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    out["call_ms"] = call_ms(A)
    out["call_one_group_ms"] = call_ms(A, overlap=False)
    A.symmetric_storage = False
    out["call_general_kernel_ms"] = call_ms(A)
    print(json.dumps(out), flush=True)
    del A, mat, X, Y
    torch.cuda.empty_cache()
