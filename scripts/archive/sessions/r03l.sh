#!/bin/bash
# round 3, GPU session L: FETCH_SIZE of K1s builds in one profiled process: r02, shipped, shipped without the diagonal
# blocks' fetch suppression — does the suppression reach the counters?
cd "$(dirname "$0")/../.."
O=gpurun_out/r03l; mkdir -p $O scripts/_ab
export TMPDIR=/tmp
CS=xitorch_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $CS scripts/micro/xk_symm_r02.hip -o scripts/_ab/libsymm_r02.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $CS scripts/micro/xk_symm_r03_knobs.hip -o scripts/_ab/libsymm_knobs.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $CS -DXK_SYMM_NOSUPP scripts/micro/xk_symm_r03_knobs.hip -o scripts/_ab/libsymm_nosupp.so
for V in r02 knobs nosupp; do
  rm -rf $O/p
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p -- python scripts/k1s_ab.py 32 16384 6 $V=scripts/_ab/libsymm_$V.so > /dev/null 2>$O/$V.err
  f=$(find $O/p -name '*counter_collection.csv' | head -1)
  python scripts/pmc_parse.py "$f" dense_symm_tiles | grep -E "FETCH|_dur" | sed "s/^/$V /" | tee -a $O/fetch_variants.txt
  rm -rf $O/p
done
