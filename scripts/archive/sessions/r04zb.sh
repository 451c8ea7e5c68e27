#!/bin/bash
# r04zb: the suites that run through K3g, with the two-stage form as the automatic choice
cd "$(dirname "$0")/../.."
O=gpurun_out/r04zb; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_k1.py tests/test_gpu_davidson.py tests/test_gpu_exacteig.py tests/test_gpu_fuzz.py tests/test_gpu_guard.py -q -m gpu -x > $O/tests.txt 2>&1; echo "tests rc=$?"
tail -8 $O/tests.txt
