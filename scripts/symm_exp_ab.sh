#!/bin/bash
# same-box upper bounds for xk_symm.hip (WRONG results for EXP>0): EXP=1 no end barrier / row flush, EXP=2 no flush at all
cd "$(dirname "$0")/.."
CS=xitorch_amd/csrc
for E in 0 1 2 0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $CS -DXK_SYMM_EXP=$E -c $CS/xk_symm.hip -o $CS/build/xk_symm.hip.o 2>&1 | grep error
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CS/libxitorch_amd.so $CS/build/*.o
  echo "== EXP=$E"
  python scripts/symm_ab.py 2>/dev/null | grep '"variant": 1' | cut -c1-140
  python scripts/symm_ab.py 64 16384 6 2>/dev/null | grep '"variant": 1' | cut -c1-140
done
