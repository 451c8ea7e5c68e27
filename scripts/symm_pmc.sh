#!/bin/bash
# PMC passes for the K1s variants alone on the GPU (kernel-trace + one counter group per pass: the gpurun rule)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE WRITE_SIZE"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-28)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_symm_$tag -- python $R/scripts/symm_ab.py ${ABARGS:-32 16384 6} > /dev/null 2>&1
  f=$(find $O/pmc_symm_$tag -name '*counter_collection.csv' | head -1)
  echo "== counters: $C"
  if [ -n "$f" ]; then
    python $R/scripts/pmc_parse.py "$f" dense_symm_tiles
    python $R/scripts/pmc_parse.py "$f" dense_symm2_tiles
  fi
done
