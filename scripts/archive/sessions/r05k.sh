#!/bin/bash
# r05k: cache-policy bits of K1s' operator loads (trial builds -DXK_SYMM_AUX=n; 2 = nt ships): alone and in the pipeline,
# one process per build on the same box
cd "$(dirname "$0")/../.."
O=gpurun_out/r05k; mkdir -p $O scripts/_ab
CS=xitorch_amd/csrc
OBJS=$(ls $CS/build/*.hip.o | grep -v xk_symm.hip.o)
for aux in 0 1 3 16 17 18 19; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $CS -DXK_SYMM_AUX=$aux -c $CS/xk_symm.hip -o scripts/_ab/symm_aux$aux.o 2>$O/build_$aux.err && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/_ab/libaux$aux.so $OBJS scripts/_ab/symm_aux$aux.o 2>>$O/build_$aux.err ) &
done
wait
ls -la scripts/_ab/libaux*.so | awk '{print $5, $9}'
for aux in 2 0 1 3 16 17 18 19 2; do
  if [ $aux = 2 ]; then LIBV=""; else LIBV=$PWD/scripts/_ab/libaux$aux.so; fi
  XITORCH_AMD_LIB=$LIBV timeout 300 python scripts/k1s_pipeline_ab.py --batch 64 --steps 3 --reps 2 --alone shipped=auto:auto 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    d['aux'] = $aux
    if d.get('alone') and d['opts'] != 48: continue
    print(json.dumps(d))" | tee -a $O/cache_policy.jsonl | cut -c1-300
done
