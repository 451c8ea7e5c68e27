#!/bin/bash
# round 3, GPU session M: the tridiagonalisation of K3g spread over W workgroups per matrix (one launch per step)
cd "$(dirname "$0")/../.."
O=gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_k1.py -m gpu -q -x -k "big" --tb=short 2>&1 | tail -15 > $O/pytest_sel.txt
tail -8 $O/pytest_sel.txt
timeout 900 python scripts/k3m_sweep.py $1 2>$O/sweep.err | tee $O/k3m_sweep.jsonl | cut -c1-1500
tail -5 $O/sweep.err
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k3mprof -- python - <<'PY' > /dev/null 2>$GRAFT_REPO_ROOT/gpurun_out/r03m/prof.err
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from xitorch_amd import kernels as K
dev = torch.device("cuda:0")
R = torch.randn(32, 582, 582, dtype=torch.float64, device=dev); T = (R + R.transpose(-2, -1)).contiguous()
for _ in range(3): K.small_eigh_big(T, 582, 6)
torch.cuda.synchronize()
PY
cd $GRAFT_REPO_ROOT
F=$(find /tmp/k3mprof -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && python scripts/summarize_rocprof.py $F gpurun_out/r03m/k3m_582_kernel_stats.csv 12 && cat gpurun_out/r03m/k3m_582_kernel_stats.csv | cut -c1-200
