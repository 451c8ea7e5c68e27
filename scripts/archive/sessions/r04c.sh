#!/bin/bash
# round 4, GPU session C: guard (whole-basis re-orthonormalisation on roll-back) + K1w with the 16-wide fp32 MFMA tile
cd "$(dirname "$0")/../.."
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_guard.py tests/test_gpu_fuzz.py "tests/test_gpu_k1.py::test_dense_wide_mfma_vs_oracle" -q -m gpu --durations=12 > $O/tests_new.txt 2>&1; echo "new tests rc=$?"
tail -40 $O/tests_new.txt
timeout 600 python scripts/guard_scan.py > $O/guard_scan.jsonl 2>$O/guard_scan.err; echo "scan rc=$?"; tail -3 $O/guard_scan.err
timeout 600 python scripts/bench_configs.py c5w c5 > $O/c5w.jsonl 2>$O/c5w.err; echo "c5w rc=$?"; cut -c1-900 $O/c5w.jsonl; tail -3 $O/c5w.err
timeout 900 python -m pytest tests -q -m gpu --durations=10 --deselect tests/test_gpu_guard.py --deselect tests/test_gpu_fuzz.py > $O/tests_all.txt 2>&1; echo "suite rc=$?"
tail -18 $O/tests_all.txt
