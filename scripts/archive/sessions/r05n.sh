#!/bin/bash
# r05n: more batch groups now that the panel launches are resident and their tails overlap (2 / 3 / 4 groups)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05n; mkdir -p $O
timeout 900 python scripts/k1s_pipeline_ab.py --batch 64 --steps 4 --reps 3 g2=48:2:32:0:0:0:2 g3=48:2:32:0:0:0:3 g4=48:2:32:0:0:0:4 \
   2>$O/ab_err.txt | tee $O/ab_b64.jsonl | cut -c1-420
tail -2 $O/ab_err.txt
