#!/bin/bash
# r04z: un-restarted S2 Davidson (64 x 16384^2 fp64) with K3g one-stage against two-stage inside the pipeline
cd "$(dirname "$0")/../.."
O=gpurun_out/r04z; mkdir -p $O
for algo in 1 0; do
  XK_K3G_ALGO=$algo timeout 900 python scripts/bench_configs.py c2:S2:0 2>$O/err_$algo.txt | tee -a $O/c2_S2_algo$algo.jsonl | cut -c1-700
  tail -2 $O/err_$algo.txt
done
