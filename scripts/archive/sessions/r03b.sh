#!/bin/bash
# round 3, GPU session B: native GMRES + tightened parity tests; K1s refill-granularity / barrier-cost variants
cd "$(dirname "$0")/../.."
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_solve.py tests/test_gpu_davidson.py tests/test_gpu_dist.py tests/test_gpu_complex.py -m gpu -q 2>&1 | tail -40 > $O/pytest_sel.txt
tail -25 $O/pytest_sel.txt
SRC=scripts/micro/xk_symm_r03_knobs.hip bash scripts/k1s_build_ab.sh $O "rf1:-DXK_SYMM_REFILL=1" "rf2:-DXK_SYMM_REFILL=2" "nobar:-DXK_SYMM_NOBAR" "rf2nobar:-DXK_SYMM_REFILL=2 -DXK_SYMM_NOBAR"
CS=xitorch_amd/csrc
for V in 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $CS -DXK_SYMM_REFILL=$V -c scripts/micro/xk_symm_r03_knobs.hip -o $CS/build/xk_symm.hip.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CS/libxitorch_amd.so $CS/build/*.o
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-general-extra 2>$O/bench_rf$V.err | tee $O/bench_rf$V.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('REFILL=$V ms/step', round(d['ms_per_step'],2), 'k1s_ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],4), 'ok', d['check']['ok'])"
done
