#!/bin/bash
# round 3, GPU session T: per-kernel totals of the un-restarted S2 run (64 x 16384^2, basis to 582 vectors)
cd "$(dirname "$0")/../.."
O=gpurun_out/r03t; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python scripts/bench_configs.py c2:S2:0 > $O/s2_line.jsonl 2>$O/prof.err
F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/r03_s2_unrestarted_kernel_stats_summary.csv 25 && cut -c1-130 $O/r03_s2_unrestarted_kernel_stats_summary.csv
rm -rf $O/prof
cut -c1-400 $O/s2_line.jsonl
