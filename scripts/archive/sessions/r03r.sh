#!/bin/bash
# round 3, GPU session R: the fused CholeskyQR kernel after its serial section moved to LDS (it had spilled 554 VGPRs
# once the shift argument was added: 35 -> 75 us); Davidson tests, per-kernel stats at 8 operators, shard times
cd "$(dirname "$0")/../.."
O=gpurun_out/r03r; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_davidson.py -m gpu -q -x 2>&1 | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof8 -- python bench.py --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-general-extra > /dev/null 2>$O/prof8.err
F=$(find $O/prof8 -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/r03_b8_kernel_stats_summary.csv 30 > /dev/null && grep "cholqr\|tridiag\|dense_symm_tiles\|ritz_res\|lincomb" $O/r03_b8_kernel_stats_summary.csv | cut -c1-110
rm -rf $O/prof8
for b in 8 8; do python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-general-extra 2>/dev/null; done | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('shard', d['config']['global_batch'], round(d['ms_per_step'],2))"
python scripts/timeline_small.py 8 overlap_only=1 2>/dev/null | cut -c1-330
