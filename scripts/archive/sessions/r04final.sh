#!/bin/bash
# r04final: the round's closing record on the final code: full -m gpu suite, smoke, the bench line, the same command under
# rocprofv3 --kernel-trace --stats, configs[4] (K1sw cooperative form) under rocprofv3
cd "$(dirname "$0")/../.."
O=gpurun_out/r04final; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"
tail -4 $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $O/bench_line.json 2>$O/bench.err; echo "bench rc=$?"; cut -c1-600 $O/bench_line.json
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py > $O/bench_line_profiled.json 2>$O/prof.err
F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/r04_bench_kernel_stats_summary.csv 14 && cut -c1-140 $O/r04_bench_kernel_stats_summary.csv
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --config c5w > $O/c5w_line_profiled.json 2>$O/prof2.err
F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/r04_c5w_kernel_stats_summary.csv 10 && cut -c1-140 $O/r04_c5w_kernel_stats_summary.csv
rm -rf $O/prof
