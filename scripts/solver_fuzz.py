"""Randomised small-size scan of the native solvers against dense references (torch.linalg on the same device):
block Davidson (orders 40 .. 1500, blocks 1 .. 12, lowest / uppest, nguess > neig, fp64 / fp32, with and without an overlap
operator M, batch 1 .. 3, spectra with clusters and with mixed convergence) and cg / bicgstab / gmres (SPD and
non-symmetric, E shifts, several right-hand sides).  Prints one JSON line per failing case and a summary.
    python scripts/solver_fuzz.py [n_cases] [seed]"""
import os, sys, json, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd.linalg.native_eig import davidson
from xitorch_amd.linalg import native_krylov as nk
dev = torch.device("cuda:0")
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator().manual_seed(seed)
warnings.simplefilter("ignore")


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g))


def spectrum(kind, N):
    i = torch.arange(N, dtype=torch.float64)
    if kind == 0:      # separated lowest / uppermost few, dense middle (pairs inside the bulk converge late)
        d = 10.0 + 5.0 * i / N
        d[:5] = torch.tensor([1.0, 2.0, 3.0, 4.5, 6.0]); d[-4:] = torch.tensor([40.0, 45.0, 52.0, 60.0])
    elif kind == 1:    # smooth, slowly converging
        d = 1.0 + (i / N) ** 2 * 100.0
    elif kind == 2:    # clusters of (nearly) equal eigenvalues at both ends
        d = 20.0 + 10.0 * torch.rand(N, dtype=torch.float64, generator=g)
        d[:6] = torch.tensor([1.0, 1.0 + 1e-9, 1.0 + 2e-9, 2.0, 2.0, 3.0]); d[-3:] = torch.tensor([90.0, 90.0, 95.0])
    else:              # random
        d = torch.rand(N, dtype=torch.float64, generator=g) * 50.0
    return d


fails, slow, done = [], [], {"davidson": 0, "krylov": 0}
for case in range(ncases):
    # ---------------- Davidson
    N = [ri(40, 90), ri(100, 400), ri(401, 1500)][ri(0, 2)]
    B, p = ri(1, 3), ri(1, 12)
    if 3 * p > N:
        p = max(1, N // 4)
    nguess = p + (ri(0, 3) if ri(0, 2) == 0 else 0)
    mode = "lowest" if ri(0, 2) else "uppest"
    dtype = torch.float64 if ri(0, 4) else torch.float32
    kind = ri(0, 3)
    useM = ri(0, 5) == 0 and p <= 8 and nguess <= 8
    d = spectrum(kind, N)
    Q, _ = torch.linalg.qr(torch.randn(B, N, N, dtype=torch.float64, generator=g))
    mat = (Q * d) @ Q.transpose(1, 2)
    mat = ((mat + mat.transpose(1, 2)) * 0.5).to(dtype).to(dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    Mop, Mm = None, None
    if useM:
        R = torch.randn(B, N, N, dtype=torch.float64, generator=g) * (0.3 / N ** 0.5)
        Mm = (torch.eye(N, dtype=torch.float64) + R @ R.transpose(1, 2)).to(dtype).to(dev)
        Mm = (Mm + Mm.transpose(1, 2)) * 0.5
        Mop = xa.LinearOperator.m(Mm, is_hermitian=True)
    eps = 1e-8 if dtype == torch.float64 else 2e-3
    rec = {"solver": "davidson", "case": case, "N": N, "B": B, "neig": p, "nguess": nguess, "mode": mode,
           "dtype": str(dtype).split(".")[1], "spectrum": kind, "M": useM}
    try:
        tr = {}
        ev, X = davidson(A, p, mode, M=Mop, nguess=nguess, min_eps=eps, max_niter=600, trace=tr)
        md = mat.double()
        if useM:
            L = torch.linalg.cholesky(Mm.double())
            Li = torch.linalg.inv(L)
            ref = torch.linalg.eigvalsh(Li @ md @ Li.transpose(1, 2))
        else:
            ref = torch.linalg.eigvalsh(md)
        want = ref[:, :p] if mode == "lowest" else ref[:, -p:]
        scale = ref.abs().max().item()
        err = (ev.double() - want).abs().max().item() / scale
        Xd = X.double()
        G = Xd.transpose(1, 2) @ ((Mm.double() @ Xd) if useM else Xd)
        orth = (G - torch.eye(p, dtype=torch.float64, device=dev)).abs().max().item()
        tol_e, tol_o = (1e-9, 1e-8) if dtype == torch.float64 else (3e-4, 2e-3)
        rec.update(niter=tr["niter"], stop=tr["stop_reason"], eval_rel_err=err, orth_err=orth)
        if tr["stop_reason"] == "max_niter" and orth <= tol_o:
            rec["note"] = "not converged within max_niter (one vector per iteration on a slowly converging spectrum): no failure"
            slow.append(rec); print(json.dumps(rec), flush=True)
        elif not (err <= tol_e and orth <= tol_o and tr["stop_reason"] in ("converged", "full_basis")):
            fails.append(rec); print(json.dumps(rec), flush=True)
    except Exception as e:                                                   # noqa
        rec["error"] = repr(e)[:160]
        fails.append(rec); print(json.dumps(rec), flush=True)
    done["davidson"] += 1
    # ---------------- Krylov
    N = ri(30, 700)
    B, nc = ri(1, 3), ri(1, 5)
    sym = ri(0, 1) == 1
    R = torch.randn(B, N, N, dtype=torch.float64, generator=g) / N ** 0.5
    if sym:
        Am = R @ R.transpose(1, 2) + 0.5 * torch.eye(N, dtype=torch.float64)
    else:
        Am = 0.4 * R + 2.0 * torch.eye(N, dtype=torch.float64)
    Bm = torch.randn(B, N, nc, dtype=torch.float64, generator=g)
    useE = ri(0, 2) == 0
    E = (-torch.rand(B, nc, dtype=torch.float64, generator=g)) if useE else None    # negative shifts keep A - E definite
    Ad, Bd = Am.to(dev), Bm.to(dev)
    Aop = xa.LinearOperator.m(Ad, is_hermitian=sym)
    Ed = E.to(dev) if useE else None
    Xref = torch.empty_like(Bd)
    for c in range(nc):
        Ac = Ad - (Ed[:, c, None, None] * torch.eye(N, dtype=torch.float64, device=dev) if useE else 0.0)
        Xref[:, :, c] = torch.linalg.solve(Ac, Bd[:, :, c])
    for name in (("cg", "bicgstab", "gmres") if sym else ("bicgstab", "gmres")):
        rec = {"solver": name, "case": case, "N": N, "B": B, "ncols": nc, "sym": sym, "E": useE}
        try:
            fn_ = getattr(nk, name)
            kw = dict(rtol=1e-10, atol=1e-12, max_niter=N + 20)
            if name != "gmres":
                kw["posdef"] = True if sym else None
            Xs = fn_(Aop, Bd, E=Ed, **kw)
            err = ((Xs - Xref).norm() / Xref.norm()).item()
            rec["rel_err"] = err
            if not err <= 1e-6:
                fails.append(rec); print(json.dumps(rec), flush=True)
        except Exception as e:                                               # noqa
            rec["error"] = repr(e)[:160]
            fails.append(rec); print(json.dumps(rec), flush=True)
        done["krylov"] += 1
print(json.dumps({"summary": True, "seed": seed, "cases": done, "failures": len(fails), "not_converged_within_max_niter": len(slow)}))
