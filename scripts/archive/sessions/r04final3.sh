#!/bin/bash
# r04final3: closing record on the last code (K1s tile-height template, two-stage K3g): -m gpu suite, smoke, bench line,
# the same command under rocprofv3 --kernel-trace --stats
cd "$(dirname "$0")/../.."
O=gpurun_out/r04final3; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"
grep -E "passed|failed" $O/tests.txt | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_line.json 2>$O/bench.err; echo "bench rc=$?"; cut -c1-260 $O/bench_line.json
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py > $O/bench_line_profiled.json 2>$O/prof.err
F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/r04_bench_kernel_stats_summary.csv 14 && cut -c1-120 $O/r04_bench_kernel_stats_summary.csv | head -6
rm -rf $O/prof
