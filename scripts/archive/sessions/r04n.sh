#!/bin/bash
# round 4, GPU session N: GMRES(30) restarted on configs[2]'s systems until rtol = 1e-10; first-call cost of the
# un-restarted S2 run with the basis storage sized up front
cd "$(dirname "$0")/../.."
O=gpurun_out/r04n; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --config c3g --gmres-restart 30 --max-niter 300 --steps 2 --warmup 1 > $O/bench_c3g_restart30.json 2>$O/c3g.err; echo "c3g rc=$?"; cut -c1-1300 $O/bench_c3g_restart30.json; tail -2 $O/c3g.err
timeout 600 python scripts/bench_configs.py c2:S2:0 > $O/c2_S2.jsonl 2>$O/c2.err; timeout 600 python scripts/bench_configs.py c2cap:S2:600 >> $O/c2_S2.jsonl 2>>$O/c2.err; cut -c1-700 $O/c2_S2.jsonl; tail -2 $O/c2.err
