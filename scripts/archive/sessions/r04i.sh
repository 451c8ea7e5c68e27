#!/bin/bash
# round 4, GPU session I: K1sw (symmetric wide panel on the matrix cores) — parity, timing against K1w; K1s small-N crossover
cd "$(dirname "$0")/../.."
O=gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_k1.py -q -m gpu -k "symm_wide" > $O/tests_k1sw.txt 2>&1; echo "k1sw tests rc=$?"
tail -30 $O/tests_k1sw.txt
timeout 300 python scripts/k1sw_bench.py 8 > $O/k1sw_bench.jsonl 2>$O/k1sw_bench.err; cat $O/k1sw_bench.jsonl; tail -3 $O/k1sw_bench.err
timeout 300 python scripts/k1s_small_crossover.py > $O/k1s_crossover.jsonl 2>$O/k1s_crossover.err; cat $O/k1s_crossover.jsonl | cut -c1-200
