#!/bin/bash
# r06i: kernel times of the persistent form by order (rocprofv3 kernel stats)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06i; mkdir -p $O
export TMPDIR=/tmp
for k in 64 128 192 256 330; do
  rm -rf $O/prof
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python scripts/k3g_two_stage_one.py $k 1 3 > /dev/null 2>$O/prof_err.txt
  KS=$(find $O/prof -name '*kernel_stats.csv' | head -1)
  python scripts/summarize_rocprof.py $KS $O/k3p_k${k}_b1_kernel_stats.csv 8 > /dev/null
  echo "== k=$k"; head -6 $O/k3p_k${k}_b1_kernel_stats.csv
done
rm -rf $O/prof
exit 0
