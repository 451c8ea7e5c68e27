#!/bin/bash
# round 4, GPU session D: native exacteig (K3t / K3g behind symeig's default method) + suite
cd "$(dirname "$0")/../.."
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_exacteig.py -q -m gpu --durations=8 > $O/tests_exacteig.txt 2>&1; echo "exacteig rc=$?"
tail -30 $O/tests_exacteig.txt
timeout 900 python -m pytest tests -q -m gpu --durations=10 --deselect tests/test_gpu_exacteig.py > $O/tests_all.txt 2>&1; echo "suite rc=$?"
tail -14 $O/tests_all.txt
timeout 300 python bench.py --steps 3 --warmup 1 > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r04d/bench.json"))
print(json.dumps(d["cpu_baseline"].get("config1_n512_b1"))[:1500])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
P
