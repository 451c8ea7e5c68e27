#!/bin/bash
# r05i: which compute units the panel streams give up: the last bits of the linear CU mask (shipped) against every 8th / 4th bit
cd "$(dirname "$0")/../.."
O=gpurun_out/r05i; mkdir -p $O
timeout 900 python scripts/k1s_pipeline_ab.py --batch 64 --steps 4 --reps 3 tail32=48:2:32:0:0:0 strided32=48:2:32:0:0:1 tail64=48:2:64:0:0:0 strided64=48:2:64:0:0:1 strided16=48:2:16:0:0:1 \
   2>$O/ab_err.txt | tee $O/ab_b64.jsonl | cut -c1-420
tail -2 $O/ab_err.txt
