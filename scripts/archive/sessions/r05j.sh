#!/bin/bash
# r05j: what the two CU-mask patterns leave (XCD / CU probe); configs[4] shard under both; the bench line with the strided pattern
cd "$(dirname "$0")/../.."
O=gpurun_out/r05j; mkdir -p $O
timeout 300 python scripts/cu_mask_probe.py 2>/dev/null | tee $O/cu_mask_probe.jsonl
for pat in 0 1; do
  XITORCH_AMD_CU_MASK_PATTERN=$pat timeout 600 python bench.py --config c5w --steps 6 --warmup 2 --no-general-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'cu_mask_pattern': $pat, 'c5w_ms_per_step': d['ms_per_step'], 'k1sw_avg_launch_ms': d['roofline']['avg_launch_ms'], 'frac': d['roofline']['frac']}))" | tee -a $O/c5w_pattern.jsonl
done
for b in 8 16; do
for pat in 0 1; do
  XITORCH_AMD_CU_MASK_PATTERN=$pat timeout 600 python bench.py --batch $b --steps 10 --warmup 3 --no-configs --no-cpu-baseline --no-general-extra --no-standalone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'cu_mask_pattern': $pat, 'batch': $b, 'ms_per_step': d['ms_per_step'], 'k1_avg_launch_ms': d['roofline']['avg_launch_ms']}))" | tee -a $O/shards_pattern.jsonl
done
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2>$O/bench_err.txt; tail -2 $O/bench_err.txt; cut -c1-300 $O/bench_line.json
