"""From a rocprofv3 kernel trace of the native Davidson: how much of the small-kernel chain between two panel products
of one batch group is kernel time and how much is gaps (launch latency, host round trips)?
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python scripts/timeline_small.py 8
    python scripts/chain_gaps.py OUT/.../*kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "xk::" in r["Kernel_Name"] or "at::native" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# per stream (Queue/Stream id): segments between consecutive dense_symm_tiles kernels
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
by_q = collections.defaultdict(list)
for r in rows:
    by_q[r[key]].append(r)
tiles = [r for r in rows if "dense_symm_tiles" in r["Kernel_Name"]]
print("streams:", {q: len(v) for q, v in by_q.items()}, "tile launches:", len(tiles))
# chain kernels = everything that is not the tile kernel; group them by stream and split at long idle gaps
for q, v in by_q.items():
    small = [r for r in v if "dense_symm_tiles" not in r["Kernel_Name"]]
    if len(small) < 50:
        continue
    busy = sum(r["e"] - r["s"] for r in small)
    # a 'chain' = run of kernels separated by < 300 us
    chains, cur = [], [small[0]]
    for a, b in zip(small, small[1:]):
        if b["s"] - a["e"] > 300000:
            chains.append(cur); cur = []
        cur.append(b)
    chains.append(cur)
    chains = [c for c in chains if len(c) >= 8]
    span = sum(c[-1]["e"] - c[0]["s"] for c in chains)
    kern = sum(r["e"] - r["s"] for c in chains for r in c)
    n = sum(len(c) for c in chains)
    print("stream %s: %d chains, %d kernels (%.1f per chain), kernel time %.2f ms, span %.2f ms, gaps %.2f ms (%.0f us per chain, %.1f us per kernel)"
          % (q, len(chains), n, n / max(len(chains), 1), kern / 1e6, span / 1e6, (span - kern) / 1e6,
             (span - kern) / 1e3 / max(len(chains), 1), (span - kern) / 1e3 / max(n, 1)))
