"""Completion periods of the panel-product launches from a rocprofv3 kernel trace (the same definition as bench.py's
`roofline.avg_launch_ms`: e_i - max(s_i, e_{i-1}) over the launches ordered by completion) next to their plain durations.
With the resident launches of the two batch groups on two streams a dispatch's own interval includes the time it waits
for the workgroup slots of the launch before it, so the trace's average duration agrees with
`roofline.launch_ms_own_interval_avg` and the period computed here with `roofline.avg_launch_ms`.
    python scripts/rocprof_k1_periods.py <kernel_trace.csv> <kernel-name substring> [out.json]"""
import csv, json, sys
path, pat = sys.argv[1], sys.argv[2]
iv = []
for r in csv.DictReader(open(path)):
    if pat in r["Kernel_Name"]:
        iv.append((int(r["Start_Timestamp"]) / 1e6, int(r["End_Timestamp"]) / 1e6))
iv.sort(key=lambda t: t[1])
own = [e - s for s, e in iv]
per, prev = [], None
for s, e in iv:
    per.append(e - (s if prev is None or s > prev else prev))
    prev = e
pct = lambda v, q: sorted(v)[min(len(v) - 1, int(round(q * (len(v) - 1))))]
rec = {"kernel": pat, "launches": len(iv), "own_interval_avg_ms": sum(own) / len(own), "completion_period_avg_ms": sum(per) / len(per),
       "completion_period_p10_p50_p90_ms": [pct(per, q) for q in (0.1, 0.5, 0.9)],
       "overlapping_launches": sum(1 for i in range(1, len(iv)) if iv[i][0] < iv[i - 1][1])}
print(json.dumps(rec))
if len(sys.argv) > 3:
    json.dump(rec, open(sys.argv[3], "w"), indent=1)
