#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for c in c4 c3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r02_$c -- python $R/scripts/bench_configs.py $c > $O/prof_r02_$c.log 2>&1
  f=$(find $O/prof_r02_$c -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python $R/scripts/summarize_rocprof.py "$f" $O/r02_${c}_kernel_stats_summary.csv 22
  head -16 $O/r02_${c}_kernel_stats_summary.csv | cut -c1-170
done
