#!/bin/bash
# r04zi: SQ counters of the chase kernel (order 582 x 32, fp64 and fp32): how busy are the vector ALU and the matrix pipe
# of the ONE CU a matrix gets
cd "$(dirname "$0")/../.."
O=gpurun_out/r04zi; mkdir -p $O
export TMPDIR=/tmp
for dt in f64 f32; do
  rm -rf $O/pmc
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc -- python scripts/k3g_two_stage_one.py 582 32 2 $dt > /dev/null 2>$O/pmc_$dt.err
  P=$(find $O/pmc -name "*counter_collection.csv" | head -1)
  echo "== $dt"; [ -n "$P" ] && python scripts/pmc_parse.py $P band_chase_kernel | tee $O/chase_pmc_$dt.txt
  rm -rf $O/pmc
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- python scripts/k3g_two_stage_one.py 582 32 2 $dt > /dev/null 2>>$O/pmc_$dt.err
  P=$(find $O/pmc -name "*counter_collection.csv" | head -1)
  [ -n "$P" ] && python scripts/pmc_parse.py $P band_chase_kernel | tee -a $O/chase_pmc_$dt.txt
done
rm -rf $O/pmc
