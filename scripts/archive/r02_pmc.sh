#!/bin/bash
# PMC passes for K1wr (separate passes per counter group, kernel-trace only: the gpurun rule)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for P in 16 32; do
  for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
    tag=$(echo $C | tr ' ' '_' | cut -c1-24)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_rw_${P}_$tag -- python $R/scripts/archive/pmc_rowswide.py $P > /dev/null 2>&1
    f=$(find $O/pmc_rw_${P}_$tag -name '*counter_collection.csv' | head -1)
    echo "== P=$P counters: $C"
    [ -n "$f" ] && python $R/scripts/pmc_parse.py "$f" dense_rows_wide_kernel
  done
done
