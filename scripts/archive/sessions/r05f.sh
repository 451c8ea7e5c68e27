#!/bin/bash
# r05f: K3g at fp64 orders 769 .. 1024 (tests), a wider-masked panel stream while the basis is small, reserve 24 / 40
cd "$(dirname "$0")/../.."
O=gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_k1.py tests/test_gpu_exacteig.py tests/test_gpu_davidson.py -x -q -m gpu -k "eigh_big or exacteig or 900_vectors or beyond_128 or wide_blocks" 2>&1 | tail -4 | tee $O/tests.txt
timeout 900 python scripts/k1s_pipeline_ab.py --batch 64 --steps 4 --reps 3 auto=auto:auto r24=48:2:24 r40=48:2:40 e16k30=48:2:32:16:30 e16k54=48:2:32:16:54 e16k78=48:2:32:16:78 e8k30=48:2:32:8:30 r16=48:2:16 \
   2>$O/ab_err.txt | tee $O/ab_b64.jsonl | cut -c1-420
tail -2 $O/ab_err.txt
