#!/bin/bash
# r04v: K3g two-stage form: parity, then timing against the one-stage form
cd "$(dirname "$0")/../.."
O=gpurun_out/r04v; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_k1.py -q -m gpu -k "two_stage" -x > $O/tests.txt 2>&1; echo "tests rc=$?"
tail -30 $O/tests.txt
timeout 600 python scripts/k3g_two_stage.py > $O/k3g_two_stage.jsonl 2>$O/k3g.err; tail -3 $O/k3g.err; cat $O/k3g_two_stage.jsonl
