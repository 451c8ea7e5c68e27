"""K3 (parallel Jacobi) vs K3t (tridiagonalisation + bisection + inverse iteration) across orders k, p = 6 wanted pairs.
    python scripts/eigh_bench.py -> one JSON line per (matrix family, k)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xitorch_amd import kernels as K, synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timed(fcn, reps=5):
    fcn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fcn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for B in ((4,) if os.environ.get('TRI_SWEEP') else (4, 64)):
    for fam in ("generic", "davidson"):
        for k in (6, 12, 18, 24, 36, 54, 78, 108, 112, 128):
            if fam == "generic":
                R = torch.randn(B, k, k, dtype=torch.float64, device=dev)
                T = (R + R.transpose(-2, -1)) * 0.5 + torch.diag(torch.arange(k, dtype=torch.float64, device=dev)) * 3
            else:
                D = synthetic.spectrum("S1", 4096, device=dev)
                Q, _ = torch.linalg.qr(torch.randn(B, 4096, k, dtype=torch.float64, device=dev))
                T = Q.transpose(-2, -1) @ (D[None, :, None] * Q)
            ref = torch.linalg.eigvalsh(T)[:, :6]
            out = {"B": B, "family": fam, "k": k}
            lam, Y, sw = K.small_eigh(T, k, 6, method="jacobi")
            out["jacobi_ms"] = timed(lambda: K.small_eigh(T, k, 6, method="jacobi"))
            out["jacobi_err"] = (lam - ref).abs().max().item()
            out["jacobi_sweeps"] = int(sw.max())
            if K.small_eigh_tri_ok(k, 6, torch.float64):
                lam, Y, info = K.small_eigh(T, k, 6, method="tri")
                out["tri_ms"] = timed(lambda: K.small_eigh(T, k, 6, method="tri"))
                out["tri_err"] = (lam - ref).abs().max().item()
                Yc = Y.transpose(-2, -1)
                out["tri_resid"] = (T @ Yc - Yc * lam.unsqueeze(-2)).abs().max().item()
                out["tri_flags"] = int(info.abs().max())
                if os.environ.get("TRI_SWEEP"):
                    from xitorch_amd._capi import fn
                    out["tri_ms_by_threads"] = {}
                    for nthr in (64, 128, 256, 512, 1024):
                        fn("xk_small_eigh_tri_set_threads")(nthr)
                        out["tri_ms_by_threads"][nthr] = round(timed(lambda: K.small_eigh(T, k, 6, method="tri")), 4)
                    fn("xk_small_eigh_tri_set_threads")(0)
            print(json.dumps(out), flush=True)
