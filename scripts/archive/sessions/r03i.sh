#!/bin/bash
# round 3, GPU session I: K3g on the upper triangle only — correctness, timing against rocSOLVER, S2 un-restarted by cross-over
cd "$(dirname "$0")/../.."
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_k1.py tests/test_gpu_davidson.py -m gpu -q -x -k "big or beyond_128" --tb=short 2>&1 | tail -15 > $O/pytest_sel.txt
tail -6 $O/pytest_sel.txt
python scripts/_ab/k3g_time.py 2>&1 | tee $O/k3g_time.jsonl | tail -12
python scripts/_ab/s2_thresholds.py 128 448 768 2>&1 | tee $O/s2_thresholds.jsonl | tail -4
