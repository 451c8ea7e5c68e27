"""Where the panel-product stream idles in a small-batch Davidson call (the strong-scaling shards): GPU-side time stamps
of every panel launch and chain stage of one call (`trace["timeline"]`), the gaps between consecutive panel launches and
the chain stage that ends each gap.   python scripts/timeline_gaps.py [B] [N] [p] [f64|f32] [min_eps] [reserve_cus] [groups]
(defaults: the configs[1] shards; `16 32768 16 f32 2e-3` = the configs[4] shard)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic, kernels as K
from xitorch_amd.linalg.native_eig import davidson
from xitorch_amd.linalg import native_eig as _ne
_ne.CHAIN_CUS = os.environ.get("XK_CHAIN_CUS", _ne.CHAIN_CUS)
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
p = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dtype = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.float64
min_eps = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-8
reserve = (sys.argv[6] if sys.argv[6] == "auto" else int(sys.argv[6])) if len(sys.argv) > 6 else "auto"
ngroups = (sys.argv[7] if sys.argv[7] == "auto" else int(sys.argv[7])) if len(sys.argv) > 7 else "auto"
mat = torch.empty((B, N, N), dtype=dtype, device=dev)
synthetic.dense_symmetric(B, N, "S1" if p <= 6 else "S1:%d" % p, dtype=dtype, device=dev, out=mat)
A = xa.LinearOperator.m(mat, is_hermitian=True)
K.prefill_timing_events(4000)
for rep in range(3):
    tl, ev = [], []
    tr = {"timeline": tl, "k1_events": ev}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        davidson(A, p, "lowest", min_eps=min_eps, rng_device="device", max_niter=60, reserve_cus=reserve, groups=ngroups,
                 trace=tr)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
base = min(tl, key=lambda t: 0)[2] if tl else None
base = tl[0][2]
rows = sorted(((g, lab, base.elapsed_time(e0), base.elapsed_time(e1)) for (g, lab, e0, e1) in tl), key=lambda r: r[2])
k1 = [r for r in rows if r[1] == "k1"]
other = [r for r in rows if r[1] != "k1"]
gaps = []
for a, b in zip(k1[:-1], k1[1:]):
    gap = b[2] - a[3]
    if gap > 0.005:
        # the chain stage of b's group that ended last before b started
        prev = [r for r in other if r[0] == b[0] and r[3] <= b[2] + 0.02]
        last = max(prev, key=lambda r: r[3]) if prev else None
        gaps.append({"after_launch_of_group": a[0], "before_launch_of_group": b[0], "gap_ms": round(gap, 3),
                     "last_stage": last[1] if last else None, "stage_end_to_launch_ms": round(b[2] - last[3], 3) if last else None})
k1_busy = sum(r[3] - r[2] for r in k1)
span = max(r[3] for r in rows) - min(r[2] for r in rows)
tot = {}
for r in other:
    tot.setdefault(r[1], [0.0, 0])
    tot[r[1]][0] += r[3] - r[2]; tot[r[1]][1] += 1
print(json.dumps({"B": B, "N": N, "p": p, "dtype": str(dtype), "reserve_cus": reserve, "groups": ngroups, "chain_cus": _ne.CHAIN_CUS, "wall_ms": round(wall, 2), "gpu_span_ms": round(span, 2), "k1_launches": len(k1), "k1_busy_ms": round(k1_busy, 2),
                  "k1_avg_ms": round(k1_busy / len(k1), 4), "gap_total_ms": round(sum(g["gap_ms"] for g in gaps), 2),
                  "first_k1_start_ms": round(k1[0][2], 3), "after_last_k1_ms": round(max(r[3] for r in rows) - k1[-1][3], 3),
                  "stage_totals_ms_calls": {k: [round(v[0], 2), v[1]] for k, v in tot.items()},
                  "gaps": gaps[:60]}))
