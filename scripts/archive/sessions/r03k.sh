#!/bin/bash
# round 3, GPU session K: FETCH_SIZE of the round-2 and the round-3 K1s tile kernels in ONE profiled process (same box,
# same operator batch) — is the 1.9 % more traffic of the new kernel real?
cd "$(dirname "$0")/../.."
O=gpurun_out/r03k; mkdir -p $O scripts/_ab
export TMPDIR=/tmp
CS=xitorch_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $CS scripts/micro/xk_symm_r02.hip -o scripts/_ab/libsymm_r02.so
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -- python scripts/k1s_ab.py 32 16384 6 r02=scripts/_ab/libsymm_r02.so lib=$CS/libxitorch_amd.so > /dev/null 2>$O/$C.err
  f=$(find $O/$C -name '*counter_collection.csv' | head -1)
  python - "$f" $C <<'PY' | tee -a $O/fetch_r02_vs_r03.txt
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if r["Counter_Name"] != sys.argv[2] or "dense_symm_tiles" not in n:
        continue
    key = "r03 (5 trailing ints)" if n.count("int") >= 7 else "r02 (3 trailing ints)"
    agg[key].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(sys.argv[2], k, "launches", len(v), "avg_KB_raw %.6g" % (sum(v) / len(v)), "min %.6g max %.6g" % (min(v), max(v)))
PY
  rm -rf $O/$C
done
