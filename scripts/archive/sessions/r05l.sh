#!/bin/bash
# r05l: xk_panel_chol as one wave per member (LDS): parity, then the configs[4] shard and wide-block Davidson cases
cd "$(dirname "$0")/../.."
O=gpurun_out/r05l; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_davidson.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $O/tests.txt
for i in 1 2; do
timeout 600 python bench.py --config c5w --steps 6 --warmup 2 --no-general-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'c5w_ms_per_step': d['ms_per_step'], 'value': d['value'], 'k1sw_avg_launch_ms': d['roofline']['avg_launch_ms'], 'frac': d['roofline']['frac'], 'iters': d['config']['iterations_per_step']}))" | tee -a $O/c5w.jsonl
done
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --config c5w --steps 3 --warmup 1 --no-general-extra > /dev/null 2>$O/prof_err.txt
KS=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python scripts/summarize_rocprof.py $KS $O/r05_c5w_kernel_stats_summary.csv 30
rm -rf $O/prof
head -16 $O/r05_c5w_kernel_stats_summary.csv | cut -c1-130
