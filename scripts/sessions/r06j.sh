#!/bin/bash
# r06j: the shipped library after the K3p / final-kernel work: eigensolver tests, the three K3g forms by order and batch,
# configs[0] wall time and kernel stats
cd "$(dirname "$0")/../.."
O=gpurun_out/r06j; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_k1.py tests/test_gpu_exacteig.py tests/test_gpu_davidson.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests_tail.txt
python scripts/k3_small_batch.py 1,4,32 64,96,128,132,160,192,224,256,288,330,384,450,512 2>&1 | grep -v amdgpu.ids | tee $O/r06_k3p_orders.jsonl
python scripts/c1_profile.py 5 2>&1 | tail -1 | tee $O/c1_wall.json
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python scripts/c1_profile.py 5 > $O/c1_under_rocprof.json 2>$O/prof_err.txt
KS=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python scripts/summarize_rocprof.py $KS $O/r06_c1_kernel_stats_summary.csv 40
rm -rf $O/prof
head -24 $O/r06_c1_kernel_stats_summary.csv
exit 0
