#!/bin/bash
# r06g: where configs[0] (N = 512, one operator) spends its call on the current code: kernel stats, K3t phase stamps,
# the Rayleigh-Ritz solvers by order at batch 1
cd "$(dirname "$0")/../.."
O=gpurun_out/r06g; mkdir -p $O
export TMPDIR=/tmp
python scripts/c1_profile.py 5 2>&1 | tail -1 | tee $O/c1_wall.json
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python scripts/c1_profile.py 5 > $O/c1_under_rocprof.json 2>$O/prof_err.txt
KS=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python scripts/summarize_rocprof.py $KS $O/r06_c1_kernel_stats_summary.csv 40
rm -rf $O/prof
python scripts/tri_profile.py 2>&1 | tee $O/tri_profile.jsonl
python scripts/k3_small_batch.py 1 2>&1 | tee $O/k3_small_batch.jsonl
head -30 $O/r06_c1_kernel_stats_summary.csv
exit 0
