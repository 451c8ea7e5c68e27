#!/bin/bash
# r05final2: PMC record of K1s re-stamped on the last kernel source; GPU suite summary line
cd "$(dirname "$0")/../.."
O=gpurun_out/r05final2; mkdir -p $O
export TMPDIR=/tmp
bash scripts/pmc_traffic.sh $O/pmc_k1s 2>&1 | tail -2
cp profiles/k1s_pmc_traffic.json $O/
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/gputests_tail.txt
timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-configs --no-cpu-baseline --no-general-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('traffic', d['roofline']['traffic'], d['ms_per_step'])"
