"""Randomised sweep of the host-memory drivers (xitorch_amd/linalg/host_krylov.py, host_eig.py) on CPU tensors: cg / bicgstab / gmres
over dtypes (fp64, fp32, complex128), operator batch shapes that broadcast against the right-hand sides, 1-3 columns, with and
without per-column shifts E and an overlap operator M, checked through the residual identity A X - M X E = B; davidson over
orders, batch shapes, 1-4 pairs, both ends of the spectrum, with and without M, checked against the dense method.  No GPU.
    python scripts/host_fuzz.py"""
import sys, warnings, itertools, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd.linalg import solve, symeig
torch.manual_seed(0)
fails, n = [], 0
def mk(shape, dtype, herm, g, spd=True):
    n_ = shape[-1]
    a = torch.rand(shape, dtype=dtype, generator=g) * 0.1 if not dtype.is_complex else \
        torch.complex(torch.rand(shape, dtype=torch.float64, generator=g), torch.rand(shape, dtype=torch.float64, generator=g)).to(dtype) * 0.1
    a = a + torch.eye(n_, dtype=dtype) * (1.0 if spd else 0.3)
    if herm:
        a = (a + a.transpose(-2, -1).conj()) * 0.5
    return a
for seed in range(60):
    g = torch.Generator().manual_seed(seed)
    dtype = [torch.float64, torch.float64, torch.complex128, torch.float32][seed % 4]
    n_ = [12, 30, 47][seed % 3]
    ashape = [(n_, n_), (2, n_, n_), (1, n_, n_), (3, 1, n_, n_)][(seed // 2) % 4]
    bbatch = [(), (2,), (3, 2)][(seed // 3) % 3]
    try:
        torch.broadcast_shapes(ashape[:-2], bbatch)
    except RuntimeError:
        bbatch = ()
    ncols = 1 + seed % 3
    for method in ("cg", "bicgstab", "gmres"):
        if method == "gmres" and dtype.is_complex:
            continue
        herm = method == "cg" or seed % 2 == 0
        useE = seed % 5 in (1, 2) 
        useM = useE and seed % 5 == 2
        if method == "gmres" and useE and ncols > 1:
            pass
        A = mk(ashape, dtype, herm, g)
        Bm = torch.rand((*bbatch, n_, ncols), dtype=torch.float64, generator=g).to(dtype)
        E = (torch.rand((*bbatch, ncols), dtype=torch.float64, generator=g) * 0.2).to(dtype) if useE else None
        M = mk((n_, n_), dtype, True, g) if useM else None
        tol = 1e-9 if dtype != torch.float32 else 1e-5
        n += 1
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("error", xa.ConvergenceWarning)
                X = solve(xa.LinearOperator.m(A, is_hermitian=herm), Bm, E=E, M=(xa.LinearOperator.m(M, True) if M is not None else None),
                          method=method, rtol=tol, atol=1e-14 if dtype != torch.float32 else 1e-8)
            AX = torch.matmul(A, X)
            if E is not None:
                MX = torch.matmul(M, X) if M is not None else X
                AX = AX - MX * E.unsqueeze(-2)
            err = ((AX - Bm).abs().max() / Bm.abs().max()).item()
            lim = 1e-6 if dtype != torch.float32 else 2e-3
            if not err < lim:
                fails.append((seed, method, str(dtype), ashape, bbatch, ncols, useE, useM, err))
        except Exception as e:
            fails.append((seed, method, str(dtype), ashape, bbatch, ncols, useE, useM, repr(e)[:200]))
# davidson
for seed in range(24):
    g = torch.Generator().manual_seed(100 + seed)
    n_ = [40, 64, 97][seed % 3]
    ashape = [(n_, n_), (2, n_, n_), (2, 1, n_, n_)][seed % 3]
    neig = 1 + seed % 4
    mode = ["lowest", "uppest"][seed % 2]
    dtype = torch.float64 if seed % 4 else torch.float32
    A = mk(ashape, dtype, True, g, spd=False) * torch.linspace(1, 2, n_, dtype=dtype)
    A = (A + A.transpose(-2, -1)) * 0.5
    useM = seed % 5 == 3
    M = mk((n_, n_), dtype, True, g) if useM else None
    n += 1
    try:
        ev, X = symeig(xa.LinearOperator.m(A, True), neig=neig, mode=mode, M=(xa.LinearOperator.m(M, True) if useM else None),
                       method="davidson", min_eps=1e-9 if dtype == torch.float64 else 1e-4)
        MX = torch.matmul(M, X) if useM else X
        R = torch.matmul(A, X) - MX * ev.unsqueeze(-2)
        ref, _ = symeig(xa.LinearOperator.m(A, True), neig=neig, mode=mode, M=(xa.LinearOperator.m(M, True) if useM else None))
        e1, e2 = R.abs().max().item(), (ev - ref).abs().max().item()
        lim = 1e-7 if dtype == torch.float64 else 5e-3
        if not (e1 < lim and e2 < lim):
            fails.append(("dav", seed, ashape, neig, mode, str(dtype), useM, e1, e2))
    except Exception as e:
        fails.append(("dav", seed, ashape, neig, mode, str(dtype), useM, repr(e)[:200]))
print(json.dumps({"cases": n, "failures": len(fails)}))
for f in fails[:30]:
    print(f)
