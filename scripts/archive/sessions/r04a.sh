#!/bin/bash
# round 4, GPU session A: a-posteriori guard — new tests, guard scan (thresholds), full suite, headline with the guard on
cd "$(dirname "$0")/../.."
O=gpurun_out/r04a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_guard.py tests/test_gpu_fuzz.py -q -m gpu --durations=12 -x > $O/tests_new.txt 2>&1; echo "new tests rc=$?" 
tail -30 $O/tests_new.txt
timeout 600 python scripts/guard_scan.py > $O/guard_scan.jsonl 2>$O/guard_scan.err; echo "scan rc=$?"; tail -3 $O/guard_scan.err
timeout 900 python -m pytest tests -q -m gpu -x --durations=10 --deselect tests/test_gpu_guard.py --deselect tests/test_gpu_fuzz.py > $O/tests_all.txt 2>&1; echo "suite rc=$?"
tail -18 $O/tests_all.txt
timeout 300 python bench.py --steps 5 --warmup 2 > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench.json
