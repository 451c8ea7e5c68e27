#!/bin/bash
# round 4, GPU session K: configs[4] (16-column block) with K1sw inside symeig — bench line, fullsize test, K1 tests
cd "$(dirname "$0")/../.."
O=gpurun_out/r04k; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --config c5w --steps 3 --warmup 1 > $O/bench_c5w.json 2>$O/bench_c5w.err; echo "c5w rc=$?"; cut -c1-1900 $O/bench_c5w.json; tail -3 $O/bench_c5w.err
timeout 900 python -m pytest tests/test_gpu_backward_fullsize.py tests/test_gpu_k1.py tests/test_gpu_davidson.py -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -6 $O/tests.txt
