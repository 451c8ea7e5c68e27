#!/bin/bash
# round 4, GPU session E: restarted GMRES(m) + bench lines with rooflines for configs[2..4] (bench.py --config), under
# rocprofv3 --kernel-trace --stats for c3 / c4 / c5w
cd "$(dirname "$0")/../.."
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_solve.py -q -m gpu --durations=5 > $O/tests_solve.txt 2>&1; echo "solve tests rc=$?"
tail -15 $O/tests_solve.txt
for c in c3 c3g c4 c5 c5w; do
  timeout 600 python bench.py --config $c --steps 3 --warmup 1 > $O/bench_$c.json 2>$O/bench_$c.err; echo "$c rc=$?"
  cut -c1-1800 $O/bench_$c.json; tail -2 $O/bench_$c.err
done
timeout 600 python bench.py --config c3g --gmres-restart 30 --steps 2 --warmup 1 --max-niter 200 > $O/bench_c3g_restart30.json 2>$O/bench_c3g_restart30.err; echo "c3g restart rc=$?"
cut -c1-1500 $O/bench_c3g_restart30.json; tail -2 $O/bench_c3g_restart30.err
for c in c3 c4 c5w; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python bench.py --config $c --steps 3 --warmup 1 > $O/bench_${c}_under_rocprof.json 2>$O/prof_$c.err
  F=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/r04_${c}_kernel_stats_summary.csv 20 && cut -c1-150 $O/r04_${c}_kernel_stats_summary.csv | head -12
  rm -rf $O/prof_$c
done
