#!/bin/bash
# A/B of K1s build variants inside the eigensolver pipeline (run on the GPU box: hipcc is in the image)
cd "$(dirname "$0")/.."
CS=xitorch_amd/csrc
for V in "2 2" "1 2" "1 3" "1 4" "2 3"; do
  set -- $V
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $CS -DXK_SYMM_NU=$1 -DXK_SYMM_WPE=$2 -c $CS/xk_symm.hip -o $CS/build/xk_symm.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CS/libxitorch_amd.so $CS/build/*.o
  python -m pytest tests/test_gpu_k1.py -m gpu -q -k "symm" 2>&1 | tail -1
  python bench.py --steps 3 --no-cpu-baseline --no-general-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('NU=$1 WPE=$2', 'ms/step', round(d['ms_per_step'],2), 'k1s_ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],4), 'ok', d['check']['ok'])"
done
