#!/bin/bash
# round 3, GPU session D: chain A/B in one process at 64 / 16 / 8 operators, 3 groups, rocprof of the 8-operator shard
cd "$(dirname "$0")/../.."
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
show() { python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('B=%d %s overlap=%s groups=%s wall' % (d['B'], d['opts'], d['overlap'], d['groups']), d['wall_ms'], d['phase_total_ms'])"; }
python scripts/timeline_small.py 64 chain=kernels,calls overlap_only=1 2>$O/tl64.err | tee $O/tl64.jsonl | show
python scripts/timeline_small.py 8 chain=calls groups=2,3,4 overlap_only=1 2>$O/tl8g.err | tee $O/tl8g.jsonl | show
python scripts/timeline_small.py 16 chain=kernels,calls groups=2,3 overlap_only=1 2>$O/tl16g.err | tee $O/tl16g.jsonl | show
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof8 -- python bench.py --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-general-extra > $O/bench8_prof.json 2>$O/prof8.err
F=$(find $O/prof8 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/b8_kernel_stats_summary.csv 40 && head -30 $O/b8_kernel_stats_summary.csv
rm -rf $O/prof8
