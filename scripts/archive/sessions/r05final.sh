#!/bin/bash
# r05final: the round's closing record on the last code: GPU suite, smoke(), bench line as the driver runs it, the same
# under rocprofv3 (kernel trace + stats), PMC re-check of the K1s record's hash
cd "$(dirname "$0")/../.."
O=gpurun_out/r05final; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee $O/gputests_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
SECONDS=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2>$O/bench_err.txt; echo "bench wall seconds: $SECONDS" | tee $O/bench_wall.txt; tail -1 $O/bench_err.txt; cut -c1-300 $O/bench_line.json
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-general-extra --no-configs --no-standalone > $O/bench_under_rocprof.json 2>$O/prof_err.txt
KS=$(find $O/prof -name '*kernel_stats.csv' | head -1); KT=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/summarize_rocprof.py $KS $O/r05_bench_kernel_stats_summary.csv 30
python scripts/rocprof_k1_periods.py $KT dense_symm_tiles $O/r05_bench_k1_periods_from_trace.json
rm -rf $O/prof
