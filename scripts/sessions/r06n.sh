#!/bin/bash
# r06n: kernel stats of the un-restarted S2 run (basis to 582 vectors) on the r06 code
cd "$(dirname "$0")/../.."
O=gpurun_out/r06n; mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python scripts/bench_configs.py c2:S2:0 > $O/s2_under_rocprof.json 2>$O/prof_err.txt
KS=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python scripts/summarize_rocprof.py $KS $O/r06_s2_unrestarted_kernel_stats_summary.csv 40
rm -rf $O/prof
head -30 $O/r06_s2_unrestarted_kernel_stats_summary.csv
tail -1 $O/s2_under_rocprof.json
exit 0
