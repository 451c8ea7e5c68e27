#!/bin/bash
# round 3, GPU session Q: K1s with 512-row tiles for small batches — correctness, then time per launch by batch and tile
# height alone on the GPU, then the shards and the headline
cd "$(dirname "$0")/../.."
O=gpurun_out/r03q; mkdir -p $O
export TMPDIR=/tmp
# (ran with a build of xk_symm.hip whose tile height was a template parameter chosen by batch: not kept)
python -m pytest tests/test_gpu_k1.py tests/test_gpu_davidson.py -m gpu -q -x -k "symm or pipeline or golden" 2>&1 | tail -4
python - <<'PY' | tee $O/r03_k1s_tile_height.jsonl
import json, sys, time, torch
sys.path.insert(0, ".")
from xitorch_amd import kernels as K, synthetic, _capi
dev = torch.device("cuda:0")
tune = _capi.fn("xk_dense_symm_tune")
N, P = 16384, 6
mat = torch.empty((32, N, N), dtype=torch.float64, device=dev)
synthetic.dense_symmetric(32, N, "S1", dtype=torch.float64, device=dev, out=mat)
X = torch.randn(32, P, N, dtype=torch.float64, device=dev)
for B in (2, 4, 8, 16, 32):
    rec = {"B": B, "N": N, "P": P}
    ref = None
    for trh in (1024, 512, 0):
        tune(2, trh)
        A, Xb = mat[:B], X[:B]
        for _ in range(3): Y = K.dense_symm(A, Xb)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for e0, e1 in ev:
            e0.record(); Y = K.dense_symm(A, Xb); e1.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        ms = ts[len(ts) // 2]
        tri = B * (N * (N + 1) // 2 * 8 + 2 * N * P * 8)
        rec["trh_%d" % trh] = {"ms": round(ms, 4), "frac_of_8TBps_triangle": round(tri / ms / 1e9 / 8000, 4)}
        if ref is None: ref = Y.clone()
        else: rec["trh_%d" % trh]["max_abs_diff_vs_1024"] = (Y - ref).abs().max().item()
    tune(2, 0)
    print(json.dumps(rec), flush=True)
PY
for b in 8 8 16 32; do python bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-general-extra 2>/dev/null; done > $O/shards.jsonl
python -c "
import json
for l in open('gpurun_out/r03q/shards.jsonl'):
    d=json.loads(l); print('shard', d['config']['global_batch'], round(d['ms_per_step'],2), round(d['roofline']['avg_launch_ms'],3))"
python scripts/timeline_small.py 8 overlap_only=1 2>/dev/null | cut -c1-400
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-general-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],2), round(d['roofline']['frac'],4))"
