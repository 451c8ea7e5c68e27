"""Phase timeline of the native Davidson at small per-GPU batches (the strong-scaling shard of configs[1]).
    python scripts/timeline_small.py [B] [key=value ...]
-> per-phase totals, K3 per iteration, wall time; one group vs two.  `chain=kernels,calls` and / or `groups=2,3` run every
combination in THIS process (same physical placement of the operator batch); `overlap_only=1` skips the one-group run."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic
from xitorch_amd.linalg.native_eig import davidson
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
extra = {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    extra[k] = int(v) if v.lstrip("-").isdigit() else v
only = extra.pop("overlap_only", None)
variants = [dict(extra)]
if "," in str(extra.get("chain", "")):
    variants = [dict(v, chain=c) for v in variants for c in str(extra["chain"]).split(",")]
if "," in str(extra.get("orth_passes", "")):
    variants = [dict(v, orth_passes=(g if g == "auto" else int(g))) for v in variants for g in str(extra["orth_passes"]).split(",")]
if "," in str(extra.get("groups", "")):
    variants = [dict(v, groups=int(g)) for v in variants for g in str(extra["groups"]).split(",")]
N, p = 16384, 6
mat = torch.empty((B, N, N), dtype=torch.float64, device=dev)
synthetic.dense_symmetric(B, N, "S1", dtype=torch.float64, device=dev, out=mat)
A = xa.LinearOperator.m(mat, is_hermitian=True)
for opts in variants:
    for overlap in ((True,) if only else (False, True)):
        for rep in range(3):
            tl = []
            tr = {"timeline": tl}
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad():
                davidson(A, p, "lowest", min_eps=1e-8, rng_device="device", overlap=overlap, trace=tr, **opts)
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        tot = {}
        for (g, lab, e0, e1) in tl:
            tot.setdefault(lab, []).append(e0.elapsed_time(e1))
        # untraced wall time of the same call
        ws = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad():
                davidson(A, p, "lowest", min_eps=1e-8, rng_device="device", overlap=overlap, **opts)
            torch.cuda.synchronize(); ws.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"B": B, "opts": {k: str(v) for k, v in opts.items()}, "overlap": overlap,
                          "groups": tr["groups"], "niter": tr["niter"], "orth_two_pass_from": tr.get("orth_two_pass_from"), "wall_traced_ms": round(wall, 2),
                          "wall_ms": [round(w, 2) for w in ws],
                          "phase_total_ms": {k: round(sum(v), 2) for k, v in tot.items()},
                          "phase_calls": {k: len(v) for k, v in tot.items()},
                          "k3_ms_by_call": [round(x, 3) for x in tot.get("k3", [])][:40]}), flush=True)
