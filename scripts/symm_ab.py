"""K1s variant A/B alone on the GPU: 1 = per-lane rows + wave reductions (xk_symm.hip), 2 = LDS turn + MFMA row part
(xk_symm2.hip).   python scripts/symm_ab.py [B N P]  -> one JSON line per variant"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K
from xitorch_amd._capi import fn
dev = torch.device("cuda:0")
B, N, P = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 16384, 6)
dtype = torch.float32 if "f32" in sys.argv else torch.float64
torch.manual_seed(0)
A = torch.randn(B, N, N, dtype=dtype, device=dev)
for b in range(B):
    A[b] = torch.triu(A[b]) + torch.triu(A[b], 1).transpose(-2, -1)
X = torch.randn(B, P, N, dtype=dtype, device=dev)
ref = torch.matmul(X[:2].double(), A[:2].double())
Y = torch.empty_like(X)
es = A.element_size()
tri_bytes = B * (N * (N + 1) // 2) * es + 2 * B * P * N * es
for variant in (1, 2):
    fn("xk_dense_symm_set_variant")(variant)
    K.dense_symm(A, X, out=Y); torch.cuda.synchronize()
    err = ((Y[:2].double() - ref).abs().max() / ref.abs().max()).item()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        K.dense_symm(A, X, out=Y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"variant": variant, "B": B, "N": N, "P": P, "dtype": str(dtype), "ms": round(ms, 4),
                      "TBps_triangle": round(tri_bytes / ms / 1e9, 3), "frac": round(tri_bytes / ms / 1e9 / 8.0, 4),
                      "relerr": err}), flush=True)
fn("xk_dense_symm_set_variant")(1)
