#!/bin/bash
# r05p: the un-restarted hard spectrum (S2, 64 x 16384^2, basis to 582) on the round's pipeline (resident panel launches, per-group streams)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05p; mkdir -p $O
timeout 900 python scripts/bench_configs.py c2:S2:0 2>$O/err.txt | tee $O/c2_S2.jsonl | cut -c1-900
tail -2 $O/err.txt
