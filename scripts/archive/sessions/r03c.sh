#!/bin/bash
# round 3, GPU session C: one-call chain stages — correctness, small-batch timelines before / after, headline bench
cd "$(dirname "$0")/../.."
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_davidson.py tests/test_gpu_solve.py tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_sel.txt
tail -6 $O/pytest_sel.txt
for B in 8 16; do
  python scripts/timeline_small.py $B chain=kernels 2>$O/tl_kernels_$B.err | tee $O/tl_kernels_$B.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('kernels B=%d overlap=%s wall' % (d['B'], d['overlap']), d['wall_ms'], d['phase_total_ms'])"
  python scripts/timeline_small.py $B chain=calls 2>$O/tl_calls_$B.err | tee $O/tl_calls_$B.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('calls   B=%d overlap=%s wall' % (d['B'], d['overlap']), d['wall_ms'], d['phase_total_ms'])"
done
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-general-extra 2>$O/bench.err | tee $O/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench ms/step', round(d['ms_per_step'],2), 'k1s_ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],4), 'ok', d['check']['ok'])"
python bench.py --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-general-extra 2>$O/bench8.err | tee $O/bench8.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench b8 ms/step', round(d['ms_per_step'],2))"
rocprofv3 --kernel-trace --stats -d $O/prof8 -o b8 -- python bench.py --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-general-extra > $O/bench8_prof.json 2>$O/prof8.err
F=$(find $O/prof8 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/b8_kernel_stats_summary.csv 40 && cat $O/b8_kernel_stats_summary.csv | head -30
