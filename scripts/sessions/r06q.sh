#!/bin/bash
# r06q: closing record of round 6 (after the limit changes): GPU suite, smoke(), the bench line as the driver runs it, rocprofv3
# command (shortened) and the configs[4] shard under rocprofv3 (kernel trace + stats)
cd "$(dirname "$0")/../.."
O=gpurun_out/r06q; mkdir -p $O
export TMPDIR=/tmp
if [ "$1" != "nobtests" ]; then
  timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $O/gputests_tail.txt
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2>$O/bench_err.txt; echo "bench wall seconds: $SECONDS" | tee $O/bench_wall.txt; tail -1 $O/bench_err.txt
python - <<PY
import json
d = json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
keep = {k: v for k, v in r.items() if isinstance(v, (int, float)) and not isinstance(v, bool)}
print(json.dumps({"value": d["value"], "ms_per_step": d["ms_per_step"], "roofline_scalars": keep}, indent=0)[:3000])
PY
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-general-extra --no-configs --no-standalone > $O/bench_under_rocprof.json 2>$O/prof_err.txt
KS=$(find $O/prof -name '*kernel_stats.csv' | head -1); KT=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/summarize_rocprof.py $KS $O/r06_bench_kernel_stats_summary.csv 30
python scripts/rocprof_k1_periods.py $KT dense_symm_tiles $O/r06_bench_k1_periods_from_trace.json
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --config c5w --steps 3 --warmup 1 --no-general-extra > $O/c5w_under_rocprof.json 2>>$O/prof_err.txt
KS=$(find $O/prof -name '*kernel_stats.csv' | head -1); KT=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/summarize_rocprof.py $KS $O/r06_c5w_kernel_stats_summary.csv 30
python scripts/rocprof_k1_periods.py $KT dense_symm_wide8 $O/r06_c5w_k1sw_periods_from_trace.json
rm -rf $O/prof
