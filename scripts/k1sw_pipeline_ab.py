"""K1sw (fp32, 16-column block, symmetric storage) inside the eigensolver's two-group pipeline: resident launches and
per-group panel streams against the r04 form, one process, one resident operator batch (BASELINE configs[4] shard:
16 x 32768^2 fp32).   python scripts/k1sw_pipeline_ab.py [--batch 16] name=resident:streams[:reserve] ...
resident = 0 | 1 | auto, streams = 1 | 2 | auto."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from xitorch_amd import LinearOperator, synthetic, kernels as K  # noqa: E402
from xitorch_amd.linalg import symeig  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--n", type=int, default=32768)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("variants", nargs="*")
args = ap.parse_args()
dev = torch.device("cuda:0")
B, N, p = args.batch, args.n, 16
mat = torch.empty((B, N, N), dtype=torch.float32, device=dev)
synthetic.dense_symmetric(B, N, "S1:16", dtype=torch.float32, device=dev, out=mat)
A = LinearOperator.m(mat, is_hermitian=True)
exact = synthetic.spectrum("S1:16", N, device=dev)[:p]
tri = lambda nb: nb * N * (N + 1) // 2 * 4 + 2 * nb * N * p * 4


def parse(spec):
    name, rest = spec.split("=", 1)
    f = rest.split(":")
    tf = lambda v: "auto" if v == "auto" else bool(int(v))
    return {"name": name, "resident": tf(f[0]), "streams": "auto" if f[1] == "auto" else int(f[1]) == 2,
            "reserve": int(f[2]) if len(f) > 2 else "auto"}


variants = [parse(v) for v in args.variants]
K.prefill_timing_events(2 * 40 * (args.steps + 1) * args.reps * max(1, len(variants)) + 64)


def call(v, events):
    K.K1SW_RESIDENT = v["resident"]
    tr = {"k1_events": events}
    with torch.no_grad():
        ev, _ = symeig(A, neig=p, mode="lowest", method="davidson", min_eps=2e-3, rng_device="device", max_niter=60,
                       reserve_cus=v["reserve"], k1_streams=v["streams"], trace=tr)
    return ev, tr


res = {v["name"]: {"ms": [], "per": [], "raw": [], "err": 0.0} for v in variants}
for v in variants:
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
        r["per"] += per; r["raw"] += raw; r["nb"] = nb
        r["err"] = max(r["err"], (ev.double() - exact).abs().max().item())
        r["niter"], r["kernel"] = tr["niter"], tr.get("panel_kernel")
for v in variants:
    r = res[v["name"]]
    ms = sorted(r["ms"])[len(r["ms"]) // 2]
    avg = sum(r["per"]) / len(r["per"])
    print(json.dumps({"variant": v["name"], "resident": v["resident"], "k1_streams_two": v["streams"], "reserve_cus": v["reserve"],
                      "batch": B, "N": N, "P": p, "panel_kernel": r["kernel"], "ms_per_call_median": round(ms, 2),
                      "ms_per_call_all": [round(t, 2) for t in r["ms"]], "k1_period_avg_ms": round(avg * 1e3, 4),
                      "k1_period_p10_p50_p90_ms": [round(bench._pct(r["per"], q) * 1e3, 3) for q in (0.1, 0.5, 0.9)],
                      "k1_own_interval_avg_ms": round(sum(r["raw"]) / len(r["raw"]) * 1e3, 4),
                      "frac_of_8TBps": round(tri(r["nb"]) / avg / 8e12, 4), "TFLOPs": round(2.0 * r["nb"] * N * N * p / avg / 1e12, 2),
                      "niter": r["niter"], "max_eval_err": r["err"]}), flush=True)
