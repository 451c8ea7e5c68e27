import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd.kernels import small_eigh
dev = torch.device("cuda:0")
torch.manual_seed(0)
for k in (24, 54, 108, 128):
    B = 64
    R = torch.randn(B, k, k, dtype=torch.float64, device=dev)
    T = (R + R.transpose(-2, -1)) * 0.5 + torch.diag(torch.arange(k, dtype=torch.float64, device=dev)) * 3
    lam, Y, sw = small_eigh(T, k, 6); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): small_eigh(T, k, 6)
    e1.record(); torch.cuda.synchronize()
    ref = torch.linalg.eigvalsh(T)[:, :6]
    print(json.dumps({"k": k, "ms": e0.elapsed_time(e1) / 5, "sweeps": int(sw.max()), "err": (lam - ref).abs().max().item(),
                      "us_per_step": e0.elapsed_time(e1) / 5 * 1e3 / (int(sw.max()) * (k - 1) + 1)}), flush=True)
# a Davidson-like projected matrix: a few isolated low eigenvalues + a dense bulk
from xitorch_amd import synthetic
for k in (54, 108):
    B = 64
    D = synthetic.spectrum("S1", 4096, device=dev)
    Q, _ = torch.linalg.qr(torch.randn(B, 4096, k, dtype=torch.float64, device=dev))
    T = Q.transpose(-2, -1) @ (D[None, :, None] * Q)
    lam, Y, sw = small_eigh(T, k, 6); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): small_eigh(T, k, 6)
    e1.record(); torch.cuda.synchronize()
    ref, Yr = torch.linalg.eigh(T)
    res = (T @ Y.transpose(-2, -1) - Y.transpose(-2, -1) * lam.unsqueeze(-2)).abs().max().item()
    print(json.dumps({"davidson_like_k": k, "ms": e0.elapsed_time(e1) / 5, "sweeps": int(sw.max()),
                      "err": (lam - ref[:, :6]).abs().max().item(), "resid": res}), flush=True)
