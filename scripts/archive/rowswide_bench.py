"""K1wr (row orientation, wide panels) vs the VALU rows kernel (ceil(P/8) passes), rocBLAS, and — for the solver
shape — BiCGStab with many right-hand sides.   python scripts/rowswide_bench.py"""
import os, sys, json, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import xitorch_amd as xa
from xitorch_amd.kernels import dense_mm
from xitorch_amd.linalg import native_krylov as nk
dev = torch.device("cuda:0")


def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (dt, B, N) in ((torch.float64, 16, 16384), (torch.float64, 8, 8192), (torch.float64, 3, 700), (torch.float32, 8, 16384)):
    A = torch.empty(B, N, N, dtype=dt, device=dev).uniform_(-1, 1)
    es = A.element_size()
    for P in (12, 16, 32, 48, 50):
        X = torch.randn(B, P, N, dtype=dt, device=dev)
        Y = torch.empty_like(X)
        tw = timeit(lambda: dense_mm(A, X, out=Y, trans=False))
        tv = timeit(lambda: dense_mm(A, X, out=Y, trans=False, wide=False))
        tb = timeit(lambda: torch.matmul(A, X.transpose(-2, -1)), 3)
        ref = torch.matmul(A, X.transpose(-2, -1)).transpose(-2, -1)
        err = (dense_mm(A, X, trans=False) - ref).abs().max().item()
        byt = B * N * N * es
        print(json.dumps({"kernel": "K1wr", "dtype": str(dt)[6:], "B": B, "N": N, "P": P, "k1wr_ms": tw, "valu_rows_ms": tv,
                          "rocblas_ms": tb, "GBps_one_pass_equivalent": byt / tw / 1e6,
                          "GBps_per_actual_pass": byt * ((P + 31) // 32) / tw / 1e6, "TFLOPs": 2 * B * N * N * P / tw / 1e9,
                          "speedup_vs_valu_rows": tv / tw, "speedup_vs_rocblas": tb / tw, "max_abs_err_vs_rocblas": err}),
              flush=True)
    del A
    torch.cuda.empty_cache()

# the solver shape of VERDICT r1 item 7: BiCGStab, 8 x 8192^2 fp64, 48 right-hand sides, non-Hermitian operator
g = torch.Generator(device=dev).manual_seed(5)
B, N, nc = 8, 8192, 48
A = torch.empty(B, N, N, dtype=torch.float64, device=dev).uniform_(-1, 1, generator=g) * (0.3 / N ** 0.5)
A += torch.eye(N, dtype=torch.float64, device=dev) * 2.0
Bm = torch.empty(B, N, nc, dtype=torch.float64, device=dev).uniform_(-1, 1, generator=g)
op = xa.LinearOperator.m(A, is_hermitian=False)
for rep in range(2):
    tr = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        Xs = nk.bicgstab(op, Bm, rtol=1e-10, atol=1e-12, posdef=True, trace=tr)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
res = (torch.matmul(A, Xs) - Bm).norm(dim=-2).max().item() / Bm.norm(dim=-2).max().item()
print(json.dumps({"solver": "bicgstab 8 x 8192^2 fp64, 48 rhs, non-Hermitian (no transposed copy)", "ms": t * 1e3, "niter": tr["niter"],
                  "napply": tr["napply"], "rel_resid": res, "mem_allocated_GB": torch.cuda.max_memory_allocated() / 1e9}), flush=True)
