#!/bin/bash
# r05b: longer runs now that the launches' tails overlap; the 8-wave form (2048 x 2048 tiles, one workgroup per CU)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_k1.py -x -q -m gpu -k "resident" 2>&1 | tail -5 | tee $O/tests.txt
timeout 900 python scripts/k1s_pipeline_ab.py --steps 4 --reps 3 --alone \
   p2r32=16:2:32 L4=1040:2:32 L8=2064:2:32 L16=4112:2:32 w8=48:2:32 w8L2=560:2:32 w8L8=2096:2:32 w8L8r64=2096:2:64 w8L8r16=2096:2:16 \
   2>$O/ab_err.txt | tee $O/k1s_pipeline_ab.jsonl | cut -c1-500
tail -3 $O/ab_err.txt
