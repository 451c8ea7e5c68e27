"""Trim a rocprofv3 *_kernel_stats.csv to a small, committable summary (kernel names shortened)."""
import csv, re, sys
src, dst = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
def short(n):
    n = re.sub(r"\(.*", "", n)          # drop the argument list
    n = re.sub(r"^void ", "", n)
    return n[:110]
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent", "min_us", "max_us"])
    for r in rows[:top]:
        w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6),
                    "%.1f" % (float(r["AverageNs"]) / 1e3), r["Percentage"], "%.1f" % (float(r["MinNs"]) / 1e3),
                    "%.1f" % (float(r["MaxNs"]) / 1e3)])
print("wrote", dst)
