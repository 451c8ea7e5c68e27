"""Randomised small-size scan of the block Davidson's EXTENSION paths against dense references: thick restart,
diagonal preconditioner, both, an overlap operator M with restarts, two forced batch groups (orders 150 .. 1200, blocks
1 .. 10, both ends of the spectrum, diagonally dominant operators).   python scripts/solver_fuzz_extensions.py [seed]"""
import os, sys, json, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd.linalg.native_eig import davidson
dev = torch.device("cuda:0")
warnings.simplefilter("ignore")
g = torch.Generator().manual_seed(int(sys.argv[1]) if len(sys.argv) > 1 else 11)
def ri(lo, hi): return int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
for case in range(40):
    N, B, p = ri(150, 1200), ri(1, 3), ri(1, 10)
    mode = "lowest" if case % 2 else "uppest"
    opt = ["restart", "precond", "restart+precond", "M+restart", "groups"][case % 5]
    dtype = torch.float64
    # diagonally dominant symmetric matrix (a diagonal preconditioner makes sense), slowly converging without it
    dgl = torch.sort(torch.rand(N, dtype=torch.float64, generator=g) * 100.0)[0]
    R = torch.randn(B, N, N, dtype=torch.float64, generator=g) * 0.05
    mat = torch.diag(dgl) + (R + R.transpose(1, 2)) * 0.5
    mat = mat.to(dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    kw = {}
    Mop = None; Mm = None
    if "restart" in opt: kw["restart"] = max(3 * p, ri(4, 8) * p)
    if "precond" in opt: kw["precond"] = "diag"
    if opt.startswith("M"):
        R2 = torch.randn(B, N, N, dtype=torch.float64, generator=g) * (0.2 / N ** 0.5)
        Mm = (torch.eye(N, dtype=torch.float64) + R2 @ R2.transpose(1, 2)).to(dev); Mm = (Mm + Mm.transpose(1, 2)) * 0.5
        Mop = xa.LinearOperator.m(Mm, is_hermitian=True)
        p = min(p, 8)
    if opt == "groups": kw["overlap"] = True; kw["groups"] = 2 if B >= 2 else "auto"
    tr = {}
    rec = {"case": case, "N": N, "B": B, "neig": p, "mode": mode, "opt": opt, **{k: str(v) for k, v in kw.items()}}
    try:
        ev, X = davidson(A, p, mode, M=Mop, min_eps=1e-8, max_niter=1500, trace=tr, **kw)
        if Mm is not None:
            L = torch.linalg.cholesky(Mm); Li = torch.linalg.inv(L); ref = torch.linalg.eigvalsh(Li @ mat @ Li.transpose(1, 2))
        else:
            ref = torch.linalg.eigvalsh(mat)
        want = ref[:, :p] if mode == "lowest" else ref[:, -p:]
        err = (ev - want).abs().max().item() / ref.abs().max().item()
        G = X.transpose(1, 2) @ ((Mm @ X) if Mm is not None else X)
        orth = (G - torch.eye(p, dtype=torch.float64, device=dev)).abs().max().item()
        ok = err < 1e-9 and orth < 1e-8 and tr["stop_reason"] in ("converged", "full_basis")
        rec.update(niter=tr["niter"], stop=tr["stop_reason"], err=err, orth=orth, restarts=tr.get("restarts"))
        if not ok:
            bad += 1; print(json.dumps(rec), flush=True)
    except Exception as e:
        rec["error"] = repr(e)[:160]; bad += 1; print(json.dumps(rec), flush=True)
print(json.dumps({"summary": True, "cases": 40, "bad": bad}))
