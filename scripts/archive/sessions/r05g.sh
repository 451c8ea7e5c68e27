#!/bin/bash
# r05g: validation of the refactored driver: GPU suite, smoke(), the bench line as the driver runs it (timed)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05g; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/gputests_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
/usr/bin/time -v -o $O/bench_time.txt timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2>$O/bench_err.txt; tail -2 $O/bench_err.txt; grep "Elapsed (wall" $O/bench_time.txt; cut -c1-400 $O/bench_line.json
XITORCH_BENCH_FORCE_PG=1 timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-configs --no-cpu-baseline --no-general-extra --no-standalone 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forced PG:', d['ms_per_step'], d['multi_gpu'])" | tee $O/forced_pg.txt
