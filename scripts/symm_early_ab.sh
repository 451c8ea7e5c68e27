#!/bin/bash
# same-box A/B of xk_symm.hip build flags (run on the GPU box): EARLY=0 old order (LDS set-up + barrier, then loads)
cd "$(dirname "$0")/.."
CS=xitorch_amd/csrc
for ROUND in 1 2; do
for E in 0 1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $CS -DXK_SYMM_EARLY=$E -c $CS/xk_symm.hip -o $CS/build/xk_symm.hip.o 2>&1 | grep error
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CS/libxitorch_amd.so $CS/build/*.o
  echo "== EARLY=$E"
  python scripts/symm_ab.py 2>/dev/null | grep '"variant": 1' | cut -c1-140
  python bench.py --steps 3 --no-cpu-baseline --no-general-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step', round(d['ms_per_step'],2), 'k1s_ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],4), 'standalone', round(r['standalone_whole_batch_launch']['frac'],4))"
done
done
