#!/bin/bash
# r04zg: K1s 512-row tiles (opts bit 2) against 1024-row tiles (bit 3) by batch size: parity, then bench --batch b
cd "$(dirname "$0")/../.."
O=gpurun_out/r04zg; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_k1.py -q -m gpu -k "dense_symm_vs_oracle or bit_reproducible" 2>&1 | tail -2
for b in 8 16 32 64; do for o in 8 4; do
  timeout 300 python bench.py --batch $b --k1s-opts $o --steps 6 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(json.dumps({'batch': $b, 'opts': $o, 'tile_rows': 512 if $o == 4 else 1024, 'ms': d['ms_per_step'], 'k1s_ms': d['roofline']['avg_launch_ms'], 'frac': d['roofline']['frac']}))" | tee -a $O/k1s_tile_rows.jsonl
done; done
