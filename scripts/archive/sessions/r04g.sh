#!/bin/bash
# round 4, GPU session G: PMC passes — K1w (c5w: FETCH / WRITE / MFMA busy) and the full-matrix K1 of the headline
# (`--k1 general`: FETCH / WRITE), each stamped with the hash of its source
cd "$(dirname "$0")/../.."
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
run_pmc () {   # name counters... -- command
  name=$1; shift; ctrs=""
  while [ "$1" != "--" ]; do ctrs="$ctrs $1"; shift; done; shift
  rm -rf $O/$name
  timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/$name -- "$@" > $O/$name.out 2>$O/$name.err || { echo "rocprofv3 $name failed"; tail -3 $O/$name.err; }
  find $O/$name -name '*counter_collection.csv' | head -1
}
C5W="python bench.py --config c5w --steps 2 --warmup 1"
F=$(run_pmc c5w_fetch FETCH_SIZE -- $C5W); W=$(run_pmc c5w_write WRITE_SIZE -- $C5W)
M=$(run_pmc c5w_mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- $C5W)
python scripts/pmc_collect.py dense_wide_cols 34393292800 xk_wide.hip,xk_common.h $O/r04_c5w_pmc.json FETCH=$F WRITE=$W MFMA=$M \
  "note=K1w (v_mfma_f32_16x16x4_f32) inside symeig, BASELINE configs[4] per-GPU shard 16 x 32768^2 fp32, 16-column block, 8 operators per launch; python bench.py --config c5w under rocprofv3 --pmc (separate passes)"
GEN="python bench.py --k1 general --steps 2 --warmup 1 --no-cpu-baseline --no-general-extra"
F=$(run_pmc k1_fetch FETCH_SIZE -- $GEN); W=$(run_pmc k1_write WRITE_SIZE -- $GEN)
python scripts/pmc_collect.py dense_rmm_cols 68770856960 xk_dense.hip,xk_common.h $O/k1_pmc_traffic.json FETCH=$F WRITE=$W \
  "note=full-matrix K1 (dense_rmm_cols<double,6>) inside the headline symeig call, half-batch launch of 32 operators (two-group pipeline): 32 x 16384^2 x 8 + 2 x 32 x 16384 x 6 x 8 bytes; python bench.py --k1 general under rocprofv3 --pmc (separate passes)"
rm -rf $O/c5w_fetch $O/c5w_write $O/c5w_mfma $O/k1_fetch $O/k1_write
cat $O/r04_c5w_pmc.json | head -30
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_general10.json 2>$O/bench.err; python -c "
import json; d=json.load(open('gpurun_out/r04g/bench_general10.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(json.dumps(d['general_k1'])[:700])"
