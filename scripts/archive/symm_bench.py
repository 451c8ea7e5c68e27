"""K1s harness: the symmetric-storage panel product is checked against the general K1 kernel on ragged /
odd shapes (lower triangle poisoned with NaN), then timed at bench size.

    python scripts/symm_bench.py [B]          # B = batch members at N = 16384 (default 16)
"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xitorch_amd.kernels import dense_mm, dense_symm

dev = torch.device("cuda:0")
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def symmetric(B, N, dtype, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    A = torch.empty(B, N, N, dtype=dtype, device=dev).uniform_(-1, 1, generator=g)
    for b in range(B):
        A[b] = A[b] + A[b].T.clone()
    return A


# ---- correctness of every variant (small, ragged, odd P, fp32) -------------------------------------
bad = 0
for (B, N, P, dtype) in [(2, 2048, 6, torch.float64), (3, 1536, 4, torch.float64), (2, 1000, 6, torch.float64),
                         (1, 512, 1, torch.float64), (2, 130, 3, torch.float64), (1, 3072, 5, torch.float64),
                         (2, 2, 2, torch.float64), (1, 2050, 6, torch.float64), (1, 1030, 2, torch.float64),
                         (2, 4096, 6, torch.float32), (1, 1100, 5, torch.float32), (1, 2052, 6, torch.float32),
                         (1, 5000, 6, torch.float64)]:
    A = symmetric(B, N, dtype, N + P)
    X = torch.randn(B, P, N, dtype=dtype, device=dev)
    ref = dense_mm(A, X, trans=True)
    Ap = torch.triu(A) + torch.tril(torch.full_like(A, float("nan")), -1)     # the lower triangle must not be read
    tol = (1e-13 if dtype == torch.float64 else 3e-6) * N ** 0.5
    Y = dense_symm(Ap, X)
    err = ((Y - ref).abs().max() / ref.abs().max()).item()
    ok = err < tol
    bad += (not ok)
    print(json.dumps({"check": [B, N, P, str(dtype)], "relerr": err, "ok": ok}), flush=True)
print(json.dumps({"checks_failed": bad}), flush=True)

# ---- timing at bench size -------------------------------------------------------------------------
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = 16384
A = symmetric(B, N, torch.float64, 1)
full = B * N * N * 8 / 1e6
for P in (6, 4, 1):
    X = torch.randn(B, P, N, dtype=torch.float64, device=dev)
    Y1 = dense_mm(A, X, trans=True)
    t1 = timeit(lambda: dense_mm(A, X, out=Y1, trans=True))
    row = {"B": B, "P": P, "general_ms": round(t1, 3), "general_GBps": round(full / t1, 1)}
    Y2 = dense_symm(A, X)
    err = ((Y1 - Y2).abs().max() / Y1.abs().max()).item()
    t2 = timeit(lambda: dense_symm(A, X, out=Y2))
    row["symm"] = {"ms": round(t2, 3), "triangle_GBps": round(full / 2 / t2, 1), "relerr": err}
    print(json.dumps(row), flush=True)
del A
if B >= 16:
    A = symmetric(8, 32768, torch.float32, 2)
    full = 8 * 32768 * 32768 * 4 / 1e6
    X = torch.randn(8, 6, 32768, dtype=torch.float32, device=dev)
    Y1 = dense_mm(A, X, trans=True)
    t1 = timeit(lambda: dense_mm(A, X, out=Y1, trans=True))
    row = {"fp32_B": 8, "N": 32768, "P": 6, "general_ms": round(t1, 3), "general_GBps": round(full / t1, 1)}
    Y2 = dense_symm(A, X)
    err = ((Y1 - Y2).abs().max() / Y1.abs().max()).item()
    t2 = timeit(lambda: dense_symm(A, X, out=Y2))
    row["symm"] = {"ms": round(t2, 3), "triangle_GBps": round(full / 2 / t2, 1), "relerr": err}
    print(json.dumps(row), flush=True)
