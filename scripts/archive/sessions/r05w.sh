#!/bin/bash
# r05w: SQ / TCC counters of the shipped K1s form (resident, 8-wave tiles; one whole-batch launch of 64 operators), separate passes
cd "$(dirname "$0")/../.."
O=gpurun_out/r05w; mkdir -p $O
export TMPDIR=/tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1)); rm -rf $O/p$i
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python scripts/pmc_k1s.py > /dev/null 2>$O/p$i.err || tail -2 $O/p$i.err
  F=$(find $O/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && python scripts/pmc_parse.py $F dense_symm_tiles | tee -a $O/sq_counters.txt
  rm -rf $O/p$i
done
