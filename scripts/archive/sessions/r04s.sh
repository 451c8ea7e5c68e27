#!/bin/bash
# r04s: K1sw forms across sizes (opts 0 / 1 / 3)
mkdir -p gpurun_out/r04s
for cfg in "8 32768" "32 8192" "64 4096" "64 2048"; do
  set -- $cfg
  timeout 300 python scripts/k1sw_bench.py $1 $2 2>/dev/null | tee -a gpurun_out/r04s/k1sw_sizes.jsonl
done
