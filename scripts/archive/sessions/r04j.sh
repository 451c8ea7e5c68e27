#!/bin/bash
# round 4, GPU session J: K1sw versions (v2: fused interleaved compute, single register buffer; v3: one wave per SIMD, ring of 4 sub-tiles, two LDS tiles) — parity, timing, kernel split, PMC
cd "$(dirname "$0")/../.."
O=gpurun_out/r04j; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_k1.py -q -m gpu -k "symm_wide" > $O/tests_k1sw.txt 2>&1; echo "k1sw tests rc=$?"
tail -5 $O/tests_k1sw.txt
timeout 300 python scripts/k1sw_bench.py 8 > $O/k1sw_bench.jsonl 2>$O/k1sw_bench.err; cat $O/k1sw_bench.jsonl; tail -3 $O/k1sw_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python scripts/k1sw_bench.py 8 > /dev/null 2>$O/prof.err
F=$(find $O/prof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/k1sw_kernel_stats.csv 8 && cut -c1-120 $O/k1sw_kernel_stats.csv
rm -rf $O/prof
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc -- python scripts/k1sw_bench.py 8 > /dev/null 2>$O/pmc.err
P=$(find $O/pmc -name "*counter_collection.csv" | head -1)
[ -n "$P" ] && python scripts/pmc_parse.py $P dense_symm_wide_kernel
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc2 -- python scripts/k1sw_bench.py 8 > /dev/null 2>$O/pmc2.err
P=$(find $O/pmc2 -name "*counter_collection.csv" | head -1)
[ -n "$P" ] && python scripts/pmc_parse.py $P dense_symm_wide_kernel
rm -rf $O/pmc $O/pmc2
