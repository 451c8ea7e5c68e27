#!/bin/bash
# r05m: how few compute units the chain needs beside the resident panel launches (tail mask: reserve / 8 per XCD): 32 / 16 / 8 / 0
cd "$(dirname "$0")/../.."
O=gpurun_out/r05m; mkdir -p $O
timeout 900 python scripts/k1s_pipeline_ab.py --batch 64 --steps 4 --reps 3 r32=48:2:32 r0=48:2:0 r8=48:2:8 r16=48:2:16 \
   2>$O/ab_err.txt | tee $O/ab_b64.jsonl | cut -c1-420
timeout 900 python scripts/k1s_pipeline_ab.py --batch 32 --steps 6 --reps 3 r32=48:2:32 r0=48:2:0 r8=48:2:8 \
   2>>$O/ab_err.txt | tee $O/ab_b32.jsonl | cut -c1-420
tail -2 $O/ab_err.txt
