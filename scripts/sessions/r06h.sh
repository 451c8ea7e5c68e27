#!/bin/bash
# r06h: K3p (persistent register-resident tridiagonalisation): parity test, timing by order against the other forms
cd "$(dirname "$0")/../.."
O=gpurun_out/r06h; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_k1.py -x -q -m gpu -k "persistent" 2>&1 | tail -15 | tee $O/persist_test.txt
python scripts/k3_small_batch.py 1,4,32 64,96,128,132,160,192,224,256,288,330,384,450 2>&1 | tee $O/k3_small_batch.jsonl
python scripts/c1_profile.py 5 2>&1 | tail -1 | tee $O/c1_wall.json
exit 0
