#!/bin/bash
# r04final2: after the last kernel changes (QR pass fusion, order-614 case): full -m gpu suite, smoke, bench line, K3g sweep
cd "$(dirname "$0")/../.."
O=gpurun_out/r04final2; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?"
grep -E "passed|failed" $O/tests.txt | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_line.json 2>$O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench_line.json
timeout 600 python scripts/k3g_two_stage.py > $O/k3g_two_stage.jsonl 2>/dev/null; grep -c algo2_ms $O/k3g_two_stage.jsonl
