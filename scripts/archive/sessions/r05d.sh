#!/bin/bash
# r05d: PMC traffic of the new K1s form and of the c3 / c4 kernels; rocprofv3 kernel trace of the bench; the GPU suite;
# the bench line with everything in it
cd "$(dirname "$0")/../.."
O=gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
bash scripts/pmc_traffic.sh $O/pmc_k1s 2>&1 | tail -3
for cfg in c3 c4; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_$cfg/$C
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$cfg/$C -- python scripts/pmc_kernel_run.py $cfg > /dev/null 2>$O/pmc_${cfg}_$C.err || tail -3 $O/pmc_${cfg}_$C.err
  done
done
F=$(find $O/pmc_c3/FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_c3/WRITE_SIZE -name '*counter_collection.csv' | head -1)
python scripts/pmc_collect.py banded_mm_kernel $((256*127*65536*8 + 2*256*65536*8)) xk_krylov.hip,xk_common.h profiles/c3_pmc_traffic.json FETCH=$F WRITE=$W B=256 "note=banded apply, 256 x (bw 127, N 65536) fp64, one right-hand side (scripts/pmc_kernel_run.py c3)"
F=$(find $O/pmc_c4/FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $O/pmc_c4/WRITE_SIZE -name '*counter_collection.csv' | head -1)
python scripts/pmc_collect.py dense_mm_rows $((64*8192*8192*8 + 2*64*8192*8)) xk_dense.hip,xk_common.h profiles/c4_pmc_traffic.json FETCH=$F WRITE=$W B=64 "note=A_b y_b, 64 x 8192^2 fp64, one column (scripts/pmc_kernel_run.py c4)"
cp profiles/c3_pmc_traffic.json profiles/c4_pmc_traffic.json profiles/k1s_pmc_traffic.json $O/
rm -rf $O/pmc_c3 $O/pmc_c4
# rocprofv3 kernel trace + stats of the headline (timed region only)
rm -rf $O/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-general-extra --no-configs --no-standalone > $O/bench_under_rocprof.json 2>$O/prof_err.txt
KS=$(find $O/prof -name '*kernel_stats.csv' | head -1); KT=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/summarize_rocprof.py $KS $O/r05_bench_kernel_stats_summary.csv 30
python scripts/rocprof_k1_periods.py $KT dense_symm_tiles $O/r05_bench_k1_periods_from_trace.json
rm -rf $O/prof
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/gputests_tail.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2>$O/bench_err.txt; tail -2 $O/bench_err.txt; cut -c1-600 $O/bench_line.json
