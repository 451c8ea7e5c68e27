#!/bin/bash
# r05r: S2 un-restarted with the shipped schedule (32 units below 132 basis vectors, 64 above) against 64 throughout
cd "$(dirname "$0")/../.."
O=gpurun_out/r05r; mkdir -p $O
for sch in auto 0:64 0:32,132:64 0:32,200:64; do
  if [ $sch = auto ]; then unset XK_RESERVE_SCHEDULE; else export XK_RESERVE_SCHEDULE=$sch; fi
  timeout 600 python scripts/bench_configs.py c2:S2:0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); d['reserve_schedule'] = '$sch'
print(json.dumps({k: d[k] for k in ('reserve_schedule', 'ms', 'niter', 'basis_size', 'panel_product_share_of_call', 'k1_ms_per_launch', 'max_eval_err_vs_closed_form', 'stop')}))" | tee -a $O/c2_S2_schedule.jsonl
done
