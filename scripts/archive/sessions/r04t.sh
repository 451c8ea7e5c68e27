#!/bin/bash
# r04t: K1sw cooperative form as the default: parity, configs[4] line, kernel stats and PMC of the new kernel
cd "$(dirname "$0")/../.."
O=gpurun_out/r04t; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_k1.py tests/test_gpu_backward_fullsize.py tests/test_gpu_fuzz.py -q -m gpu -k "wide or config5 or fuzz" > $O/tests.txt 2>&1; echo "tests rc=$?"
tail -4 $O/tests.txt
timeout 600 python bench.py --config c5w > $O/c5w.json 2>$O/c5w.err; cat $O/c5w.json | cut -c1-1500; tail -2 $O/c5w.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc -- python scripts/k1sw_bench.py 8 > /dev/null 2>$O/pmc.err
P=$(find $O/pmc -name "*counter_collection.csv" | head -1)
[ -n "$P" ] && python scripts/pmc_parse.py $P dense_symm_wide7 | tee $O/pmc_sq.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc2 -- python scripts/k1sw_bench.py 8 > /dev/null 2>$O/pmc2.err
P=$(find $O/pmc2 -name "*counter_collection.csv" | head -1)
[ -n "$P" ] && python scripts/pmc_parse.py $P dense_symm_wide7 | tee $O/pmc_grbm.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -- python scripts/k1sw_bench.py 8 > /dev/null 2>$O/$C.err
  P=$(find $O/$C -name "*counter_collection.csv" | head -1)
  [ -n "$P" ] && python scripts/pmc_parse.py $P dense_symm_wide7 | tee $O/pmc_$C.txt
done
rm -rf $O/pmc $O/pmc2 $O/FETCH_SIZE $O/WRITE_SIZE
