#!/bin/bash
# r05a: resident K1s launch — bit identity against the one-workgroup-per-run launch, then the two-group pipeline with
# one / two panel-product streams, all variants in one process on one resident operator batch
cd "$(dirname "$0")/../.."
O=gpurun_out/r05a; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_k1.py -x -q -m gpu -k "resident or reproducible" 2>&1 | tail -5 | tee $O/tests.txt
timeout 900 python scripts/k1s_pipeline_ab.py --steps 4 --reps 3 --alone \
   base=0:1 pers1=16:1 pers2=16:2 plain2=0:2 pers2L2=528:2 pers2r32=16:2:32 2>$O/ab_err.txt | tee $O/k1s_pipeline_ab.jsonl | cut -c1-600
tail -3 $O/ab_err.txt
