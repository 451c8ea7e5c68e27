#!/bin/bash
# r05e: K1sw resident launches in the pipeline (A/B), its PMC record, the 8 / 16 / 32-operator shards, a generic operator
cd "$(dirname "$0")/../.."
O=gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_k1.py -x -q -m gpu -k "wide" 2>&1 | tail -3 | tee $O/tests.txt
timeout 900 python scripts/k1sw_pipeline_ab.py base=0:1 res1=1:1 res2=1:2 auto=auto:auto res2r16=1:2:16 res2r48=1:2:48 2>$O/ab_err.txt | tee $O/k1sw_pipeline_ab.jsonl | cut -c1-420
tail -2 $O/ab_err.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -- python scripts/k1sw_bench.py 8 > /dev/null 2>$O/$C.err
done
rm -rf $O/MFMA
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/MFMA -- python scripts/k1sw_bench.py 8 > /dev/null 2>$O/MFMA.err
F=$(find $O/FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $O/WRITE_SIZE -name '*counter_collection.csv' | head -1); M=$(find $O/MFMA -name '*counter_collection.csv' | head -1)
python scripts/pmc_collect.py "dense_symm_wide7_kernel<1" 17213947904 xk_symmwide.hip,xk_common.h $O/k1sw_pmc_traffic.json FETCH=$F WRITE=$W MFMA=$M B=8 "note=K1sw cooperative form (opts = 3), 8 x 32768^2 fp32, P = 16, standalone (scripts/k1sw_bench.py 8); algorithmic bytes = upper triangles + panels in + out"
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE $O/MFMA
for b in 32 16 8; do
  timeout 600 python bench.py --batch $b --steps 10 --warmup 3 --no-configs --no-cpu-baseline --no-general-extra --no-standalone 2>/dev/null | tee -a $O/strong_scaling_shards.jsonl | cut -c1-200
done
timeout 600 python scripts/generic_operator_bench.py 2>$O/generic_err.txt | tee $O/generic_operator.json | cut -c1-800; tail -2 $O/generic_err.txt
