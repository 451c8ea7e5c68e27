"""S2 un-restarted (64 x 16384^2): per-iteration durations of the chain stages of group 0 against the panel product.
    python scripts/s2_timeline.py [reserve_cus]"""
import os, sys, json, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic as syn
from xitorch_amd.linalg.native_eig import davidson
dev = torch.device("cuda:0")
B, N, p = 64, 16384, 6
reserve = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mat = torch.empty((B, N, N), dtype=torch.float64, device=dev)
syn.dense_symmetric(B, N, "S2", device=dev, out=mat)
A = xa.LinearOperator.m(mat, is_hermitian=True)
for rep in range(2):
    tl, ev = [], []
    tr = {"timeline": tl, "k1_events": ev}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        davidson(A, p, "lowest", min_eps=1e-8, rng_device="device", max_niter=3000, reserve_cus=reserve, trace=tr)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
by = {}
for (g, lab, e0, e1) in tl:
    by.setdefault((g, lab), []).append(e0.elapsed_time(e1))
k1 = [a.elapsed_time(b) for (a, b, pc, nb) in ev]
out = {"wall_traced_ms": round(wall, 1), "niter": tr["niter"], "k1_total_ms": round(sum(k1), 1), "k1_launches": len(k1),
       "phase_total_ms": {"g%d_%s" % k: round(sum(v), 1) for k, v in by.items()}}
for lab in ("k3", "ritz", "orth", "extT"):
    v = by.get((0, lab), [])
    out["g0_%s_ms_every_8th_call" % lab] = [round(x, 2) for x in v[::8]]
out["k1_ms_every_16th_launch"] = [round(x, 2) for x in k1[::16]]
print(json.dumps(out))
