#!/bin/bash
# round 4, GPU session L: configs[4] with K1sw — CUs left to the other group's chain (the panel kernel is issue-bound, not
# HBM-bound: it scales with the CUs it gets)
cd "$(dirname "$0")/../.."
O=gpurun_out/r04l; mkdir -p $O
export TMPDIR=/tmp
for r in 64 32 16 8; do
  timeout 300 python bench.py --config c5w --steps 3 --warmup 1 --reserve-cus $r > $O/bench_c5w_r$r.json 2>$O/err_$r.txt
  python - $r <<'P'
import json,sys
r=sys.argv[1]
d=json.load(open("gpurun_out/r04l/bench_c5w_r%s.json"%r))
print("reserve",r,"ms_per_step",round(d["ms_per_step"],2),"k1_ms",round(d["roofline"]["avg_launch_ms"],3),"eigpairs/s",round(d["value"],1),d["config"]["panel_kernel"],d["check"]["ok"])
P
done
