#!/bin/bash
# HBM traffic of the shipped K1s (one whole-batch launch): FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_k1s_$C -- python $R/scripts/pmc_k1s.py > /dev/null 2>&1
  f=$(find $O/pmc_k1s_$C -name '*counter_collection.csv' | head -1)
  echo "== $C"
  [ -n "$f" ] && { python $R/scripts/pmc_parse.py "$f" dense_symm_tiles; python $R/scripts/pmc_parse.py "$f" symm_fold; }
done
