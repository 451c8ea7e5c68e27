"""A-posteriori guard of the native Davidson (r04): what max|X^T X - I| healthy runs reach (the thresholds GUARD_GOOD /
GUARD_BAD sit above it) and what the guard does when ONE projection pass is forced on the configurations that
returned duplicated eigenpairs in round 3.  One JSON line per run.
    python scripts/guard_scan.py"""
import os, sys, json, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic
from xitorch_amd.linalg.native_eig import davidson
dev = torch.device("cuda:0")
warnings.simplefilter("ignore")
for dtype, eps in ((torch.float64, 1e-8), (torch.float32, 2e-3)):
    for spec in ("S1", "S2", "S3"):
        for (B, N) in ((2, 900), (2, 2048)):
            mat = synthetic.dense_symmetric(B, N, spec, dtype=dtype, device=dev)
            A = xa.LinearOperator.m(mat, is_hermitian=True)
            allev = torch.linalg.eigvalsh(mat.double())
            for p in (3, 6, 8, 10, 16):
                if spec != "S1" and p > 8:
                    continue
                for passes in ("auto", 1):
                    tr = {}
                    rec = {"dtype": str(dtype).split(".")[1], "spectrum": spec, "N": N, "neig": p, "passes": passes}
                    t0 = time.time()
                    try:
                        ev, X = davidson(A, p, "lowest", min_eps=eps, orth_passes=passes, trace=tr, max_niter=500)
                        torch.cuda.synchronize()
                        G = X.double().transpose(1, 2) @ X.double()
                        gh = tr["orth_guard_history"]
                        rec.update(niter=tr["niter"], stop=tr["stop_reason"], sec=round(time.time() - t0, 3),
                                   eval_err=(ev.double() - allev[:, :p]).abs().max().item(),
                                   orth_err=(G - torch.eye(p, device=dev, dtype=G.dtype)).abs().max().item(),
                                   guard_max=max(gh), guard_last=gh[-1], redo=tr["orth_redo"],
                                   two_pass_from=tr.get("orth_two_pass_from"), rerun=tr.get("orth_rerun"))
                    except Exception as e:                                          # noqa
                        rec["error"] = repr(e)[:200]
                    print(json.dumps(rec), flush=True)
