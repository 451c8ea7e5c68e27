#!/bin/bash
# r05c: the shipped launch-form choice (kernels.k1s_auto_opts) at 64 / 32 / 16 / 8 operators against the forced forms;
# reserve sweep of the 8-wave form; then the whole bench line with the configs block
cd "$(dirname "$0")/../.."
O=gpurun_out/r05c; mkdir -p $O
timeout 600 python scripts/k1s_pipeline_ab.py --batch 64 --steps 4 --reps 3 base=0:1 auto=auto:auto w8r64=48:2:64 w8r48=48:2:48 w8r32=48:2:32 w8r16=48:2:16 p2r32=16:2:32 \
   2>$O/ab64_err.txt | tee $O/ab_b64.jsonl | cut -c1-400
for b in 32 16 8; do
timeout 600 python scripts/k1s_pipeline_ab.py --batch $b --steps 6 --reps 3 base=0:1 auto=auto:auto p2=16:2 w8=48:2 2>$O/ab${b}_err.txt | tee $O/ab_b$b.jsonl | cut -c1-400
done
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_line.json 2>$O/bench_err.txt; tail -3 $O/bench_err.txt; cut -c1-1500 $O/bench_line.json
