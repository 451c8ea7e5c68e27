import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xitorch_amd.kernels import dense_mm
dev = torch.device("cuda:0")
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (dt, B, N) in ((torch.float32, 4, 32768), (torch.float64, 8, 16384)):
    A = torch.empty(B, N, N, dtype=dt, device=dev).uniform_(-1, 1)
    es = A.element_size()
    for P in (16, 32, 50):
        X = torch.randn(B, P, N, dtype=dt, device=dev)
        Y = torch.empty_like(X)
        tw = timeit(lambda: dense_mm(A, X, out=Y, trans=True))
        tv = timeit(lambda: dense_mm(A, X, out=Y, trans=True, wide=False))
        tb = timeit(lambda: torch.matmul(A.transpose(-2, -1), X.transpose(-2, -1)), 3)
        byt = B * N * N * es
        print(json.dumps({"dtype": str(dt)[6:], "B": B, "N": N, "P": P, "wide_ms": tw, "valu_ms": tv, "rocblas_ms": tb,
                          "wide_GBps_one_pass_equiv": byt / tw / 1e6, "wide_TFLOPs": 2 * B * N * N * P / tw / 1e9,
                          "speedup_vs_valu": tv / tw, "speedup_vs_rocblas": tb / tw}), flush=True)
    del A
    torch.cuda.empty_cache()
