#!/bin/bash
# K1s2 build configurations alone on the GPU and (WITH_BENCH=1) inside bench.py; run on the GPU box.
#   CFGS="448,4 512,8"  (tile rows, waves per block)   EXTRA="-D..."   NOTEST=1
cd "$(dirname "$0")/.."
CS=xitorch_amd/csrc
for V in ${CFGS:-448,4 512,8}; do
  TRH=${V%,*}; WAVES=${V#*,}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $CS -DXK_S2_TRH=$TRH -DXK_S2_WAVES=$WAVES $EXTRA -c $CS/xk_symm2.hip -o $CS/build/xk_symm2.hip.o 2>&1 | grep -E "error" 
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CS/libxitorch_amd.so $CS/build/*.o
  echo "== TRH=$TRH WAVES=$WAVES $EXTRA"
  [ -z "$NOTEST" ] && python -m pytest tests/test_gpu_k1.py -m gpu -q -k "symm" 2>&1 | tail -1
  python scripts/symm_ab.py $ABARGS 2>/dev/null | cut -c1-150
  [ -n "$WITH_BENCH" ] && XITORCH_AMD_K1S_VARIANT=2 python bench.py --steps 3 --no-cpu-baseline --no-general-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step', round(d['ms_per_step'],2), 'k1s_ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],4), 'ok', d['check']['ok'])"
done
