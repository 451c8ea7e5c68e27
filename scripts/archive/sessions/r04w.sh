#!/bin/bash
# r04w: per-kernel times of the two-stage K3g
cd "$(dirname "$0")/../.."
O=gpurun_out/r04w; mkdir -p $O
export TMPDIR=/tmp
for cfg in "582 32" "582 1"; do
  set -- $cfg
  rm -rf $O/prof
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python scripts/k3g_two_stage_one.py $1 $2 2 > /dev/null 2>$O/prof.err
  F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
  echo "== k=$1 B=$2"; [ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/k3g2_k$1_b$2.csv 10 && cut -c1-150 $O/k3g2_k$1_b$2.csv
done
rm -rf $O/prof
