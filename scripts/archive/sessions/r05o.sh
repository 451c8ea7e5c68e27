#!/bin/bash
# r05o: TRIAL BUILD (the -DXK_RMM_TARGET macro is not in the tree: see git history at e6844c9) — work-item size of the full-matrix column kernel K1 (dense_rmm_cols): 32 operators of order 16384 are 2048 workgroups
# of 32 MB on ~1536 resident slots = 1.33 rounds; trial builds with more row slabs (-DXK_RMM_TARGET = workgroups aimed at)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05o; mkdir -p $O scripts/_ab
CS=xitorch_amd/csrc
OBJS=$(ls $CS/build/*.hip.o | grep -v xk_dense.hip.o)
for t in 4096 8192 16384 32768; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $CS -DXK_RMM_TARGET=$t -c $CS/xk_dense.hip -o scripts/_ab/dense_t$t.o 2>$O/build_$t.err && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/_ab/libt$t.so $OBJS scripts/_ab/dense_t$t.o 2>>$O/build_$t.err ) &
done
wait
for t in 2048 4096 8192 16384 32768 2048; do
  if [ $t = 2048 ]; then LIBV=""; else LIBV=$PWD/scripts/_ab/libt$t.so; fi
  XITORCH_AMD_LIB=$LIBV timeout 300 python bench.py --k1 general --steps 4 --warmup 1 --no-configs --no-cpu-baseline --no-general-extra --no-standalone 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'rmm_target': $t, 'ms_per_step': d['ms_per_step'], 'k1_avg_launch_ms': d['roofline']['avg_launch_ms'], 'frac': d['roofline']['frac'], 'check': d['check']['ok'], 'max_eval_err': d['check']['max_eval_err_vs_exact']}))" | tee -a $O/rmm_target.jsonl
done
