"""Where the upper-triangle kernel K1s stops paying: small operators are latency-bound (two launches, one 1024-row tile
per operator) and the one-launch full-matrix kernel is faster although it reads twice the bytes.  Times both on the
same symmetric operators, p = 6 fp64 / fp32.  One JSON line per shape."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K, synthetic
dev = torch.device("cuda:0")


def t_of(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dtype in (torch.float64, torch.float32):
    for (B, N) in ((1, 512), (1, 1024), (1, 2048), (1, 4096), (1, 8192), (2, 2048), (4, 2048), (8, 1024), (8, 2048), (16, 1024),
                   (4, 4096), (32, 512)):
        A = synthetic.dense_symmetric(B, N, "S1", dtype=dtype, device=dev)
        X = torch.randn(B, 6, N, dtype=dtype, device=dev)
        Y = torch.empty_like(X)
        rec = {"dtype": str(dtype).split(".")[1], "B": B, "N": N, "MB": B * N * N * A.element_size() / 2 ** 20,
               "k1s_us": round(t_of(lambda: K.dense_symm(A, X, out=Y)), 1),
               "k1_cols_us": round(t_of(lambda: K.dense_mm(A, X, out=Y, trans=True)), 1),
               "k1_rows_us": round(t_of(lambda: K.dense_mm(A, X, out=Y, trans=False)), 1)}
        print(json.dumps(rec), flush=True)
