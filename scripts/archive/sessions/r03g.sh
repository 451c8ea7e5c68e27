#!/bin/bash
# round 3, GPU session G: status read after the speculative orthonormalisation, wide blocks, GMRES line, PMC traffic
cd "$(dirname "$0")/../.."
O=gpurun_out/r03g; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_davidson.py -m gpu -q -x -k "wide" --tb=long 2>&1 | tail -60 > $O/pytest_wide.txt
tail -30 $O/pytest_wide.txt
python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest_all.txt
tail -8 $O/pytest_all.txt
show() { python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('B=%d %s overlap=%s groups=%s wall' % (d['B'], d['opts'], d['overlap'], d['groups']), d['wall_ms'], d['phase_total_ms'])"; }
python scripts/timeline_small.py 8 overlap_only=1 2>$O/tl8.err | tee $O/tl8.jsonl | show
python scripts/timeline_small.py 16 overlap_only=1 2>$O/tl16.err | tee $O/tl16.jsonl | show
python scripts/timeline_small.py 64 overlap_only=1 2>$O/tl64.err | tee $O/tl64.jsonl | show
python scripts/bench_configs.py c3g c3 2>$O/c3g.err | tee $O/c3g.jsonl
bash scripts/pmc_traffic.sh $O/pmc 2>&1 | tail -3
