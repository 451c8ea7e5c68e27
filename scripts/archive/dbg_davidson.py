import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from xitorch_amd import MatrixLinearOperator, synthetic
from xitorch_amd.linalg.native_eig import davidson
dev = torch.device("cuda:0")
def run(B, N, vinit, max_niter=40):
    mat = torch.empty((B, N, N), dtype=torch.float64, device=dev)
    synthetic.dense_symmetric(B, N, "S1", device=dev, out=mat)
    A = MatrixLinearOperator(mat, True)
    tr = {}
    torch.cuda.synchronize(); t0 = time.time()
    try:
        ev, X = davidson(A, 6, "lowest", min_eps=1e-8, v_init="randn", rng_device=vinit, max_niter=max_niter, trace=tr)
        torch.cuda.synchronize()
        err = (ev - synthetic.spectrum("S1", N, device=dev)[:6]).abs().max().item()
    except Exception as e:
        err = repr(e)
    h = tr.get("resid_history", [])
    print("B=%d N=%d %s: niter=%s t=%.2fs err=%s hist=%s ... %s" % (B, N, vinit, tr.get("niter"), time.time() - t0, err,
          ["%.1e" % v for v in h[:4]], ["%.1e" % v for v in h[-3:]]), flush=True)
    del mat
    torch.cuda.empty_cache()
for (B, N) in [(64, 2048), (16, 8192)]:
    for vinit in ("cpu", "device"):
        run(B, N, vinit)
