#!/bin/bash
# round 3, GPU session O: K3t with the streamed LU / unrolled solves, K3m in the full suite; shards and headline after it
cd "$(dirname "$0")/../.."
O=gpurun_out/r03o; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1
tail -4 $O/gputests.log
for b in 8 8 16; do python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-general-extra 2>/dev/null; done > $O/shards.jsonl
python -c "
import json
for l in open('gpurun_out/r03o/shards.jsonl'):
    d=json.loads(l); print('shard', d['config']['global_batch'], round(d['ms_per_step'],2))"
python scripts/timeline_small.py 8 overlap_only=1 2>/dev/null | cut -c1-900
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-general-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
timeout 300 python scripts/bench_configs.py c2:S2:0 2>/dev/null | cut -c1-700
