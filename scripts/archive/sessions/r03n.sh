#!/bin/bash
# round 3, GPU session N: per-launch durations of the K3m step kernels by batch, W and with parts of the kernel skipped
cd "$(dirname "$0")/../.."
O=gpurun_out/r03n; mkdir -p $O
export TMPDIR=/tmp
for cfg in "$@"; do
set -- $cfg
rm -rf /tmp/k3mprof
( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k3mprof -- python - $1 $2 $3 <<'PY' > /dev/null 2>>$GRAFT_REPO_ROOT/gpurun_out/r03n/prof.err
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from xitorch_amd import kernels as K, _capi
B, W, skip = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
_capi.fn("xk_small_eigh_big_tune")(0, W)
_capi.fn("xk_small_eigh_big_tune")(3, skip)
dev = torch.device("cuda:0")
R = torch.randn(B, 582, 582, dtype=torch.float64, device=dev); T = (R + R.transpose(-2, -1)).contiguous()
for _ in range(3): K.small_eigh_big(T, 582, 6)
torch.cuda.synchronize()
PY
)
F=$(find /tmp/k3mprof -name "*kernel_stats.csv" | head -1)
echo "== B=$1 W=$2 skip=$3"
[ -n "$F" ] && python scripts/summarize_rocprof.py $F $O/k3m_582_B$1_W$2_s$3.csv 8 >/dev/null && cut -c1-100 $O/k3m_582_B$1_W$2_s$3.csv | grep step | sort | head -5
done
