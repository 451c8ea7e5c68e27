#!/bin/bash
# round 3, GPU session S: two re-orthogonalised passes as the default of the Davidson loop (one pass returned duplicated
# eigenpairs on wide blocks with mixed convergence): regression tests, full suite, what it costs (headline, shards)
cd "$(dirname "$0")/../.."
O=gpurun_out/r03s; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1
tail -4 $O/gputests.log
for b in 8 8 16; do python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-general-extra 2>/dev/null; done | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('shard', d['config']['global_batch'], round(d['ms_per_step'],2))"
python scripts/timeline_small.py 8 overlap_only=1 orth_passes=auto,1,2 2>/dev/null | cut -c1-330
python scripts/timeline_small.py 64 overlap_only=1 orth_passes=auto,1,2 2>/dev/null | cut -c1-330
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-general-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
