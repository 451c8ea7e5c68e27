#!/bin/bash
# r04za: kernel stats of the un-restarted S2 run with the two-stage K3g
cd "$(dirname "$0")/../.."
O=gpurun_out/r04za; mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python scripts/bench_configs.py c2:S2:0 > $O/out.jsonl 2>$O/err.txt
F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/s2_two_stage_kernel_stats.csv 25 && cut -c1-160 $O/s2_two_stage_kernel_stats.csv
rm -rf $O/prof
