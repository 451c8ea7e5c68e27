#!/bin/bash
# r04r: K1sw workgroup-cooperative form (opts=1) against the shipped one-wave-per-tile form
mkdir -p gpurun_out/r04r
timeout 600 python -m pytest tests/test_gpu_k1.py -q -m gpu -k "symm_wide" > gpurun_out/r04r/test.log 2>&1
tail -5 gpurun_out/r04r/test.log
timeout 300 python scripts/k1sw_bench.py 8 > gpurun_out/r04r/k1sw_b8.json 2> gpurun_out/r04r/k1sw_b8.err
cat gpurun_out/r04r/k1sw_b8.json; tail -3 gpurun_out/r04r/k1sw_b8.err
timeout 300 python scripts/k1sw_bench.py 16 > gpurun_out/r04r/k1sw_b16.json 2> gpurun_out/r04r/k1sw_b16.err
cat gpurun_out/r04r/k1sw_b16.json; tail -3 gpurun_out/r04r/k1sw_b16.err
