#!/bin/bash
# round 4, GPU session M: K1sw shipped form (fold with loads in flight, reserve_cus auto) — K1 tests, standalone timing,
# c5w / c5 bench lines, full suite, headline
cd "$(dirname "$0")/../.."
O=gpurun_out/r04m; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/k1sw_bench.py 8 > $O/k1sw_bench.jsonl 2>$O/k1sw_bench.err; cat $O/k1sw_bench.jsonl
for c in c5w c5; do
  timeout 300 python bench.py --config $c --steps 3 --warmup 1 > $O/bench_$c.json 2>$O/bench_$c.err
  python - $c <<'P'
import json,sys
c=sys.argv[1]
d=json.load(open("gpurun_out/r04m/bench_%s.json"%c))
print(c,"ms_per_step",round(d["ms_per_step"],2),"k1_ms",round(d["roofline"]["avg_launch_ms"],3),"frac",round(d["roofline"]["frac"],3),"eigpairs/s",round(d["value"],1),d["config"]["panel_kernel"],d["check"]["ok"])
P
done
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"; tail -6 $O/tests.txt
timeout 300 python bench.py --steps 5 --warmup 2 > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r04m/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["cpu_baseline"]["config1_n512_b1"].get("gpu_davidson_ms"), d["cpu_baseline"]["config1_n512_b1"].get("gpu_exacteig_ms"))
P
