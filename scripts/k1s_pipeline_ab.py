"""K1s inside the eigensolver's two-group pipeline, A/B in ONE process on ONE resident operator batch (r05).
   python scripts/k1s_pipeline_ab.py [--batch 64] [--n 16384] [--steps 4] [--reps 2] name=opts:streams[:reserve] ...
name=opts:streams[:reserve[:early:kswitch]] — `opts` = low 16 bits of the K1s `opts` argument (include/xitorch_amd.h; 16 = resident launch, + run << 8), `streams` =
1 (both groups' panel products on one CU-masked stream) or 2 (one stream each), `reserve` = compute units left to the
other group's chain (default: the solver's auto).  Repetitions of all variants are interleaved; one JSON line per
variant: median ms per symeig call, completion periods of the panel launches (bench.py's definition), eigenvalue error.
Also: the same kernel forms alone on the idle GPU (half batch, whole batch)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from xitorch_amd import LinearOperator, synthetic, kernels as K  # noqa: E402
from xitorch_amd.linalg import symeig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--alone", action="store_true", help="also time the kernel forms alone on the idle GPU")
ap.add_argument("variants", nargs="*")
args = ap.parse_args()
dev = torch.device("cuda:0")
B, N, p = args.batch, args.n, 6
mat = torch.empty((B, N, N), dtype=torch.float64, device=dev)
synthetic.dense_symmetric(B, N, "S1", dtype=torch.float64, device=dev, out=mat)
A = LinearOperator.m(mat, is_hermitian=True)
exact = synthetic.spectrum("S1", N, torch.float64, dev)[:p]
tri = lambda nb: nb * N * (N + 1) // 2 * 8 + 2 * nb * N * p * 8


def parse(spec):
    name, rest = spec.split("=", 1)
    f = rest.split(":")
    return {"name": name, "opts": None if f[0] == "auto" else int(f[0]),
            "streams": ("auto" if f[1] == "auto" else int(f[1])) if len(f) > 1 else 1,
            "reserve": int(f[2]) if len(f) > 2 else "auto",
            "early": (int(f[3]), int(f[4])) if len(f) > 4 and int(f[4]) > 0 else None,   # (reserved CUs, basis width) of the early phase
            "pat": int(f[5]) if len(f) > 5 else 0,                       # which mask bits the panel streams give up (kernels.CU_MASK_PATTERN)
            "groups": int(f[6]) if len(f) > 6 else "auto"}               # batch groups of the pipeline


variants = [parse(v) for v in args.variants]
K.prefill_timing_events(2 * 48 * (args.steps + 1) * args.reps * max(1, len(variants)) + 64)


def call(v, events):
    K.K1S_OPTS = v["opts"]
    K.CU_MASK_PATTERN = v["pat"]
    tr = {"k1_events": events}
    with torch.no_grad():
        ev, _ = symeig(A, neig=p, mode="lowest", method="davidson", min_eps=1e-8, v_init="randn", rng_device="device",
                       max_niter=200, reserve_cus=v["reserve"], reserve_early=v["early"], groups=v["groups"],
                       k1_streams=("auto" if v["streams"] == "auto" else v["streams"] == 2), trace=tr)
    return ev, tr


res = {v["name"]: {"ms": [], "periods": [], "raw": [], "err": 0.0, "niter": None} for v in variants}
for v in variants:                       # warm-up of every form (streams, workspaces, basis storage)
    call(v, None)
torch.cuda.synchronize()
for rep in range(args.reps):
    for v in variants:
        events = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ev, tr = call(v, events)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        r = res[v["name"]]
        r["ms"].append(el / args.steps * 1e3)
        per, raw, nb = bench._k1_periods(events, p)
        r["periods"] += per
        r["raw"] += raw
        r["nb"] = nb
        r["err"] = max(r["err"], (ev.double() - exact).abs().max().item())
        r["niter"] = tr["niter"]
for v in variants:
    r = res[v["name"]]
    ms = sorted(r["ms"])[len(r["ms"]) // 2]
    per = r["periods"]
    avg = sum(per) / len(per)
    print(json.dumps({"variant": v["name"], "opts": v["opts"], "k1_streams": v["streams"], "reserve_cus": v["reserve"],
                      "reserve_early": v["early"], "cu_mask_pattern": v["pat"], "groups": v["groups"], "batch": B, "N": N, "ms_per_call_median": round(ms, 2), "ms_per_call_all": [round(t, 2) for t in r["ms"]],
                      "k1_launches": len(per), "k1_period_avg_ms": round(avg * 1e3, 4),
                      "k1_period_p10_p50_p90_ms": [round(bench._pct(per, q) * 1e3, 3) for q in (0.1, 0.5, 0.9)],
                      "k1_own_interval_avg_ms": round(sum(r["raw"]) / len(r["raw"]) * 1e3, 4),
                      "frac_of_8TBps": round(tri(r["nb"]) / avg / 8e12, 4), "k1_share_of_call": round(sum(per) / (sum(r["ms"]) * 1e-3 * args.steps), 4),
                      "niter": r["niter"], "max_eval_err": r["err"]}), flush=True)

if args.alone:
    forms = sorted(set(v["opts"] for v in variants if v["opts"] is not None) | {0})
    for nb in (B // 2, B):
        X = torch.randn((nb, p, N), dtype=torch.float64, device=dev)
        Y = torch.empty_like(X)
        ts = {o: [] for o in forms}
        for o in forms:
            K.dense_symm(mat[:nb], X, out=Y, opts=o)
        torch.cuda.synchronize()
        for rep in range(5):
            for o in forms:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    K.dense_symm(mat[:nb], X, out=Y, opts=o)
                e1.record()
                torch.cuda.synchronize()
                ts[o].append(e0.elapsed_time(e1) / 3)
        for o in forms:
            m = sorted(ts[o])[len(ts[o]) // 2]
            print(json.dumps({"alone": True, "opts": o, "operators": nb, "ms_tiles_plus_fold_median": round(m, 4),
                              "ms_all": [round(t, 3) for t in ts[o]], "frac_of_8TBps": round(tri(nb) / (m * 1e-3) / 8e12, 4)}),
                  flush=True)
