#!/bin/bash
# round 3, GPU session E: full GPU suite on the one-pass / deeper-unroll chain + shard timings + headline
cd "$(dirname "$0")/../.."
O=gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_all.txt
tail -8 $O/pytest_all.txt
show() { python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('B=%d %s overlap=%s groups=%s wall' % (d['B'], d['opts'], d['overlap'], d['groups']), d['wall_ms'], d['phase_total_ms'])"; }
python scripts/timeline_small.py 8 orth_passes=1,2 overlap_only=1 2>$O/tl8.err | tee $O/tl8.jsonl | show
python scripts/timeline_small.py 64 overlap_only=1 2>$O/tl64.err | tee $O/tl64.jsonl | show
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-general-extra 2>$O/bench.err | tee $O/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench ms/step', round(d['ms_per_step'],2), 'k1s_ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],4), 'ok', d['check']['ok'])"
