#!/bin/bash
# HBM traffic of one whole-batch K1s launch from the PMC counters (two separate passes, kernel-trace only) ->
# profiles/k1s_pmc_traffic.json, stamped with the hash of the kernel source.  Fails when the record it leaves does not
# match the source in the tree (bench.py reports `roofline.traffic` only from a matching record).
cd "$(dirname "$0")/.."
O=${1:-gpurun_out/pmc_k1s}; mkdir -p $O
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -- python scripts/pmc_k1s.py > /dev/null 2>$O/$C.err || { echo "rocprofv3 $C failed"; tail -3 $O/$C.err; exit 1; }
done
F=$(find $O/FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find $O/WRITE_SIZE -name '*counter_collection.csv' | head -1)
[ -n "$F" ] && [ -n "$W" ] || { echo "no counter files"; exit 1; }
python scripts/pmc_k1s_collect.py "$F" "$W" profiles/k1s_pmc_traffic.json || exit 1
cp profiles/k1s_pmc_traffic.json $O/k1s_pmc_traffic.json
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
python - <<'PY' || exit 1
import json, sys
sys.path.insert(0, "scripts")
from pmc_k1s_collect import kernel_source_hash
rec = json.load(open("profiles/k1s_pmc_traffic.json"))
assert rec["kernel_source_sha256"] == kernel_source_hash(), "stored PMC record does not match the kernel source"
print("PMC record matches the kernel source:", rec["kernel_source_sha256"][:16])
PY
