"""Rayleigh-Ritz solvers at the orders configs[0] walks through (one matrix, p = 6, fp64): ms per call of K3t (<= 128),
K3g one launch per Householder step (algo 1) and the two-stage form (algo 2).  JSON lines."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K
dev = torch.device("cuda:0")


def t_of(f, n=10):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


batches = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1]
orders = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [48, 96, 128, 132, 160, 192, 224, 256, 288, 330]
P = int(sys.argv[3]) if len(sys.argv) > 3 else 6
DT = torch.float32 if len(sys.argv) > 4 and sys.argv[4] == "f32" else torch.float64
for B in batches:
    for k in orders:
        g = torch.Generator().manual_seed(k)
        R = torch.randn(B, k, k, dtype=torch.float64, generator=g)
        T = (R + R.transpose(1, 2)).to(DT).to(dev)
        rec = {"B": B, "k": k, "p": P, "dtype": str(DT)}
        ref = torch.linalg.eigvalsh(T.double())[:, :P]
        if K.small_eigh_tri_ok(k, P, T.dtype):
            lam, Y, info = K.small_eigh(T, k, P, method="tri")
            rec["tri_ms"] = round(t_of(lambda: K.small_eigh(T, k, P, method="tri")), 4)
            rec["tri_err"] = float((lam.double() - ref).abs().max() / ref.abs().max())
        if K.small_eigh_big_ok(k, P, T.dtype):
            for algo in (1, 2, 3):
                try:
                    lam, Y, info = K.small_eigh_big(T, k, P, algo=algo)
                except Exception:                       # noqa
                    continue
                rec["algo%d_ms" % algo] = round(t_of(lambda: K.small_eigh_big(T, k, P, algo=algo)), 4)
                rec["algo%d_err" % algo] = float((lam.double() - ref).abs().max() / ref.abs().max())
                rec["algo%d_info" % algo] = int(info.max())
        print(json.dumps(rec), flush=True)
