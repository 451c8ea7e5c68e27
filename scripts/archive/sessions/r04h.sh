#!/bin/bash
# round 4, GPU session H: where configs[0]'s native Davidson call goes (kernel time vs host), K1s PMC re-stamp
cd "$(dirname "$0")/../.."
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
python scripts/c1_profile.py 5 > $O/c1_plain.json 2>$O/c1_plain.err; cat $O/c1_plain.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c1 -- python scripts/c1_profile.py 5 > $O/c1_under_rocprof.json 2>$O/prof_c1.err
cat $O/c1_under_rocprof.json
F=$(find $O/prof_c1 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/r04_c1_kernel_stats_summary.csv 30 && cut -c1-140 $O/r04_c1_kernel_stats_summary.csv
python - <<'P'
import csv,glob
f=glob.glob("gpurun_out/r04h/prof_c1/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print("total kernel ms over 6 calls:", sum(float(r["TotalDurationNs"]) for r in rows)/1e6, "launches:", sum(int(r["Calls"]) for r in rows))
P
rm -rf $O/prof_c1
bash scripts/pmc_traffic.sh $O/pmc_k1s 2>&1 | tail -3
