#!/bin/bash
# round 4, GPU session F: ABI without process-wide state (opts / wg / threads as arguments), device-side collectives
# (xk_comm_*), row-block sharded operator, forced single-rank RCCL group through bench.py; full suite; headline
cd "$(dirname "$0")/../.."
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_k1.py tests/test_gpu_solve.py -q -m gpu --durations=6 > $O/tests_a.txt 2>&1; echo "tests a rc=$?"
tail -25 $O/tests_a.txt
timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_gpu_dist.py --deselect tests/test_gpu_k1.py --deselect tests/test_gpu_solve.py > $O/tests_b.txt 2>&1; echo "tests b rc=$?"
tail -8 $O/tests_b.txt
XITORCH_BENCH_FORCE_PG=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-extra > $O/bench_forcepg.json 2>$O/bench_forcepg.err; echo "forced pg rc=$?"; cut -c1-300 $O/bench_forcepg.json; tail -3 $O/bench_forcepg.err
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r04f/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("avg_launch_ms"), d["step_ms"])
P
timeout 200 python bench.py --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-general-extra > $O/bench_b8.json 2>$O/bench_b8.err; python -c "
import json; d=json.load(open('gpurun_out/r04f/bench_b8.json')); print('b8', d['ms_per_step'], d['roofline']['frac'])"
