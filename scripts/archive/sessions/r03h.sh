#!/bin/bash
# round 3, GPU session H: K3g (orders 129..768) correctness + timing against rocSOLVER, S2 un-restarted on it, restart fixes
cd "$(dirname "$0")/../.."
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_k1.py tests/test_gpu_davidson.py -m gpu -q -x -k "big or beyond_128 or restart or wide or one_gram" --tb=short 2>&1 | tail -40 > $O/pytest_sel.txt
tail -25 $O/pytest_sel.txt
timeout 300 python - <<'PY' 2>&1 | tee $O/k3g_vs_library.jsonl
import json, time, torch, sys
sys.path.insert(0, ".")
from xitorch_amd import kernels as K
dev = torch.device("cuda:0")
for B in (32, 4):
    for k in (192, 256, 384, 512, 640, 768):
        g = torch.Generator().manual_seed(k)
        R = torch.randn(B, k, k, dtype=torch.float64, generator=g).to(dev)
        T = (R + R.transpose(-2, -1)).contiguous()
        def t_of(f):
            f(); torch.cuda.synchronize(); ts = []
            for _ in range(3):
                t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return sorted(ts)[1] * 1e3
        ms_g = t_of(lambda: K.small_eigh_big(T, k, 6))
        ms_l = t_of(lambda: torch.linalg.eigh(T))
        lam, Y, info = K.small_eigh_big(T, k, 6)
        err = (lam - torch.linalg.eigvalsh(T)[:, :6]).abs().max().item()
        print(json.dumps({"B": B, "k": k, "p": 6, "k3g_ms": round(ms_g, 3), "library_eigh_ms": round(ms_l, 3), "max_eval_err": err, "flag": int(info.max())}), flush=True)
PY
timeout 300 python scripts/bench_configs.py c2:S2:0 2>$O/s2.err | tee $O/s2_unrestarted.jsonl
