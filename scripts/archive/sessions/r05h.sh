#!/bin/bash
# r05h: TRIAL BUILD (kernel change not in the tree: git history of this session) — L2 touch-ahead of the K1s ring (opts bit 6): bit identity, alone and inside the pipeline
cd "$(dirname "$0")/../.."
O=gpurun_out/r05h; mkdir -p $O
python - <<'PY' 2>&1 | tail -3 | tee $O/identity.txt
import torch
from xitorch_amd import kernels as K
dev = torch.device("cuda:0")
for (B, N, P, dt) in ((3, 4096, 6, torch.float64), (2, 5000, 5, torch.float64), (2, 6144, 6, torch.float32), (1, 2050, 3, torch.float64)):
    g = torch.Generator().manual_seed(N)
    R = torch.randn(B, N, N, dtype=dt, generator=g)
    A = (R + R.transpose(-2, -1)).to(dev)
    X = torch.randn(B, P, N, dtype=dt, generator=g).to(dev)
    for base in (0, 16, 48, 8, 4):
        Y0 = K.dense_symm(A, X, opts=base).clone()
        Y1 = K.dense_symm(A, X, opts=base | 64)
        assert torch.equal(Y0, Y1), (B, N, P, dt, base)
print("touch-ahead: bit-identical")
PY
timeout 900 python scripts/k1s_pipeline_ab.py --batch 64 --steps 4 --reps 3 --alone auto=auto:auto pf=112:2:32 nopf=48:2:32 pf4=80:2:32 \
   2>$O/ab_err.txt | tee $O/ab_b64.jsonl | cut -c1-420
tail -2 $O/ab_err.txt
