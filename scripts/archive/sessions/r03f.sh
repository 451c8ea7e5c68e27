#!/bin/bash
# round 3, GPU session F: configs[4] with the MFMA panel inside symeig (+ PMC), wide blocks, library eigh probe, S2 un-restarted
cd "$(dirname "$0")/../.."
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests/test_gpu_davidson.py tests/test_gpu_backward_fullsize.py -m gpu -q -x -k "wide or config5 or one_gram or chain" 2>&1 | tail -12 > $O/pytest_sel.txt
tail -6 $O/pytest_sel.txt
python scripts/bench_configs.py c5 c5w 2>$O/c5.err | tee $O/c5.jsonl
python scripts/eigh_library_probe.py 2>$O/eighp.err | tee $O/eigh_library_probe.jsonl
timeout 300 python scripts/bench_configs.py c2:S2:0 2>$O/s2.err | tee $O/s2_unrestarted.jsonl
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_c5w -- python scripts/bench_configs.py c5w > /dev/null 2>$O/pmc_c5w.err
  f=$(find $O/pmc_c5w -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python scripts/pmc_parse.py "$f" dense_wide | tee $O/pmc_c5w_mfma.txt
done
rm -rf $O/pmc_c5w
