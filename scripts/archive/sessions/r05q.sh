#!/bin/bash
# r05q: the un-restarted hard spectrum (S2, basis to 582) by how many units the panel streams leave to the chains as the basis grows
cd "$(dirname "$0")/../.."
O=gpurun_out/r05q; mkdir -p $O
for sch in auto none 0:32,160:64,320:96 0:32,100:64,250:96 0:64 0:64,300:96 0:96; do
  if [ $sch = auto ]; then unset XK_RESERVE_SCHEDULE; else export XK_RESERVE_SCHEDULE=$sch; fi
  timeout 600 python scripts/bench_configs.py c2:S2:0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); d['reserve_schedule'] = '$sch'
print(json.dumps({k: d[k] for k in ('reserve_schedule', 'ms', 'niter', 'basis_size', 'panel_product_share_of_call', 'k1_ms_per_launch', 'max_eval_err_vs_closed_form', 'stop')}))" | tee -a $O/c2_S2_schedule.jsonl
done
unset XK_RESERVE_SCHEDULE
timeout 600 python scripts/k1s_pipeline_ab.py --batch 64 --steps 4 --reps 2 auto=auto:auto 2>/dev/null | cut -c1-300 | tee $O/headline_check.jsonl
