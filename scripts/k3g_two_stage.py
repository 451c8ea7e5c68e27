"""K3g one launch per Householder step (algo 1) against the two-stage form (algo 2): ms per call, p = 6, fp64 (and fp32
where given), by order and batch.  JSON lines."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K
dev = torch.device("cuda:0")


def t_of(f, n=5):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for dtype in (torch.float64, torch.float32):
    for B in (32, 4, 1):
        for k in (192, 256, 330, 384, 512, 582, 600, 768):
            g = torch.Generator().manual_seed(k)
            R = torch.randn(B, k, k, dtype=torch.float64, generator=g)
            T = (R + R.transpose(1, 2)).to(dtype).to(dev)
            rec = {"dtype": str(dtype), "B": B, "k": k}
            ref = torch.linalg.eigvalsh(T.double())[:, :6]
            for algo in (1, 2):
                try:
                    lam, Y, info = K.small_eigh_big(T, k, 6, algo=algo)
                except Exception as e:                      # noqa
                    rec["algo%d" % algo] = "unsupported"
                    continue
                rec["algo%d_ms" % algo] = round(t_of(lambda: K.small_eigh_big(T, k, 6, algo=algo)), 3)
                rec["algo%d_err" % algo] = float((lam.double() - ref).abs().max() / ref.abs().max())
                rec["algo%d_info" % algo] = int(info.max())
            print(json.dumps(rec), flush=True)
