"""K1sw (symmetric storage, wide panel, matrix cores) against K1w (full matrix, matrix cores) and K1s in ceil(P/6) passes
on the configs[4] shard shape: 8 x 32768^2 fp32 (one pipeline group), P = 16.  One JSON line."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K, synthetic
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
P = 16
A = torch.empty(B, N, N, dtype=torch.float32, device=dev)
synthetic.dense_symmetric(B, N, "S1:16", dtype=torch.float32, device=dev, out=A)
X = torch.randn(B, P, N, dtype=torch.float32, device=dev)
Y = torch.empty_like(X)


def t_of(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tri = B * N * (N + 1) // 2 * 4 + 2 * B * N * P * 4
full = B * N * N * 4 + 2 * B * N * P * 4
K.K1SW_OPTS = 0
t_sw = t_of(lambda: K.dense_symm_wide(A, X, out=Y))
Ysw = Y.clone()
K.K1SW_OPTS = 1
t_sw1 = t_of(lambda: K.dense_symm_wide(A, X, out=Y))
Ysw1 = Y.clone()
K.K1SW_OPTS = 3
t_sw3 = t_of(lambda: K.dense_symm_wide(A, X, out=Y))
t_w = t_of(lambda: K.dense_mm(A, X, out=Y, trans=True))
err = ((Ysw - Y).abs().max() / Y.abs().max()).item()
t_s = t_of(lambda: K.dense_symm(A, X, out=Y))
err1 = ((Ysw1 - Ysw).abs().max() / Ysw.abs().max()).item()
print(json.dumps({"B": B, "N": N, "P": P, "k1sw_ms": t_sw, "k1sw_coop_ms": t_sw1, "k1sw_coop_prio_ms": t_sw3, "rel_diff_coop_vs_k1sw": err1, "k1sw_TBps_triangle": tri / t_sw / 1e9, "k1sw_frac_triangle": tri / t_sw / 1e9 / 8.0,
                  "k1sw_TFLOPs": 2.0 * B * N * N * P / t_sw / 1e9, "k1w_ms": t_w, "k1w_frac_full": full / t_w / 1e9 / 8.0,
                  "k1s_3passes_ms": t_s, "rel_diff_k1sw_vs_k1w": err}))
