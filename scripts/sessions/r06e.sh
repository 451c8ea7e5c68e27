#!/bin/bash
# r06e: configs[4] shard with the r06 K1sw form (202 registers x 2 waves per SIMD: ~100 registers per lane and 58 KB of LDS
# stay free on every unit) — how many compute units does the panel stream have to leave to the other group's chain?
cd "$(dirname "$0")/../.."
O=gpurun_out/r06e; mkdir -p $O
for R in 32 16 8 0 24 32; do
  timeout 300 python bench.py --config c5w --steps 6 --warmup 2 --no-general-extra --reserve-cus $R 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'reserve_cus': $R, 'c5w_ms_per_step': d['ms_per_step'], 'k1sw_avg_launch_ms': d['roofline']['avg_launch_ms'], 'frac': d['roofline']['frac'], 'iters': d['config']['iterations_per_step'], 'ok': d['check']['ok']}))" | tee -a $O/c5w_reserve.jsonl
done
