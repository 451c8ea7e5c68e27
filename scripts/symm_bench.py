import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = 16384
A = torch.empty(B, N, N, dtype=torch.float64, device=dev).uniform_(-1, 1)
for b in range(B):
    A[b] = A[b] + A[b].T.clone()
for P in (6, 1, 4):
    X = torch.randn(B, P, N, dtype=torch.float64, device=dev)
    Y1 = dense_mm(A, X, trans=True); Y2 = dense_symm(A, X)
    err = ((Y1 - Y2).abs().max() / Y1.abs().max()).item()
    t1 = timeit(lambda: dense_mm(A, X, out=Y1, trans=True)); t2 = timeit(lambda: dense_symm(A, X, out=Y2))
    full = B * N * N * 8 / 1e6
    print(json.dumps({"B": B, "P": P, "relerr": err, "general_ms": t1, "symm_ms": t2, "speedup": t1 / t2,
                      "general_GBps": full / t1, "symm_fullmatrix_equiv_GBps": full / t2, "symm_actual_GBps": full / 2 / t2}), flush=True)
