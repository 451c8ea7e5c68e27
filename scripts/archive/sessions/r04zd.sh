#!/bin/bash
# r04zd: HBM traffic + matrix-pipe occupancy of the shipped K1sw kernel from the PMC counters (separate passes), stamped
# with the hash of its source -> profiles/k1sw_pmc_traffic.json; then the configs[4] line that reads it
cd "$(dirname "$0")/../.."
O=gpurun_out/r04zd; mkdir -p $O
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -- python scripts/k1sw_bench.py 8 > /dev/null 2>$O/$C.err
done
rm -rf $O/MFMA
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/MFMA -- python scripts/k1sw_bench.py 8 > /dev/null 2>$O/MFMA.err
F=$(find $O/FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find $O/WRITE_SIZE -name '*counter_collection.csv' | head -1)
M=$(find $O/MFMA -name '*counter_collection.csv' | head -1)
python scripts/pmc_collect.py "dense_symm_wide7_kernel<1>" 17213947904 xk_symmwide.hip,xk_common.h $O/k1sw_pmc_traffic.json FETCH=$F WRITE=$W MFMA=$M B=8 "note=K1sw cooperative form (opts = 3), 8 x 32768^2 fp32, P = 16, standalone (scripts/k1sw_bench.py 8); algorithmic bytes = upper triangles + panels in + out"
cp $O/k1sw_pmc_traffic.json profiles/k1sw_pmc_traffic.json
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE $O/MFMA
timeout 600 python bench.py --config c5w > $O/c5w_line.json 2>$O/c5w.err; python -c "
import json; d=json.load(open('$O/c5w_line.json')); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline'].get('traffic'), d['roofline'].get('traffic_note'))"
