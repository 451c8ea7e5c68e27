#!/bin/bash
# Build stand-alone K1s libraries from csrc/xk_symm.hip with other -D knobs (on the GPU box: hipcc is in the image) and
# time them against the library and the round-2 kernel in one process; then the chosen ones inside the eigensolver.
#   usage: scripts/k1s_build_ab.sh OUTDIR "name:-Dflags" ...
cd "$(dirname "$0")/.."
O=$1; shift; mkdir -p $O
CS=xitorch_amd/csrc
SPECS="lib=$CS/libxitorch_amd.so"
[ -f scripts/_ab/libsymm_r02.so ] && SPECS="r02=scripts/_ab/libsymm_r02.so $SPECS"
for V in "$@"; do
  name=${V%%:*}; flags=${V#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $CS $flags $CS/xk_symm.hip -o scripts/_ab/libsymm_$name.so 2>$O/build_$name.err || { echo "build $name failed"; cat $O/build_$name.err | head; continue; }
  SPECS="$SPECS $name=scripts/_ab/libsymm_$name.so"
done
echo $SPECS
python scripts/k1s_ab.py 32 16384 6 $SPECS 2>$O/ab.err | tee $O/k1s_build_ab.jsonl
