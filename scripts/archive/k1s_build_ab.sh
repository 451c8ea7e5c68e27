#!/bin/bash
# Build stand-alone K1s libraries from csrc/xk_symm.hip with other -D knobs (on the GPU box: hipcc is in the image) and
# time them against the library and the round-2 kernel in one process; then the chosen ones inside the eigensolver.
#   usage: [SRC=scripts/micro/xk_symm_r03_knobs.hip] scripts/k1s_build_ab.sh OUTDIR "name:-Dflags" ...
# SRC defaults to the product kernel; scripts/micro/xk_symm_r03_knobs.hip is the round-3 kernel with the measurement knobs
# (-DXK_SYMM_REFILL=0|1|2, -DXK_SYMM_NOBAR) that were taken out of the product file after the A/B; the round-2 kernel
# (LDS float atomics, one tile per workgroup) is scripts/micro/xk_symm_r02.hip
cd "$(dirname "$0")/.."
O=$1; shift; mkdir -p $O scripts/_ab
CS=xitorch_amd/csrc
SRC=${SRC:-$CS/xk_symm.hip}
SPECS="lib=$CS/libxitorch_amd.so"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $CS scripts/micro/xk_symm_r02.hip -o scripts/_ab/libsymm_r02.so 2>/dev/null && SPECS="r02=scripts/_ab/libsymm_r02.so $SPECS"
for V in "$@"; do
  name=${V%%:*}; flags=${V#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $CS $flags $SRC -o scripts/_ab/libsymm_$name.so 2>$O/build_$name.err || { echo "build $name failed"; cat $O/build_$name.err | head; continue; }
  SPECS="$SPECS $name=scripts/_ab/libsymm_$name.so"
done
echo $SPECS
python scripts/k1s_ab.py 32 16384 6 $SPECS 2>$O/ab.err | tee $O/k1s_build_ab.jsonl
