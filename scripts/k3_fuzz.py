"""randomised scan of the Rayleigh-Ritz solvers (K3t, K3g forms 1 / 2 / 3 and the library's choice) against LAPACK: orders
8 .. 520, 1 .. 256 pairs, fp64 / fp32, lowest / uppest, 1 .. 5 matrices, matrix kinds that stress the bisection (diagonal =
decoupled, identity = one cluster, zero, clusters, graded, huge / tiny scale, Ritz-like) — with the workspace poisoned.
Prints failing cases and a summary line.  GPU."""
import os, sys, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xitorch_amd import kernels as K
dev = torch.device("cuda:0")
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)


def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=g).item())


def make(kind, B, k):
    if kind == "random":
        R = torch.randn(B, k, k, dtype=torch.float64, generator=g)
        return R + R.transpose(1, 2)
    if kind == "diagonal":
        return torch.diag_embed(torch.randn(B, k, dtype=torch.float64, generator=g))
    if kind == "int_diagonal":                      # shifts of the multisection coincide with eigenvalues exactly
        return torch.diag_embed(torch.randint(-8, 9, (B, k), generator=g).double())
    if kind == "identity":
        return torch.eye(k, dtype=torch.float64).expand(B, k, k).clone() * 3.0
    if kind == "zero":
        return torch.zeros(B, k, k, dtype=torch.float64)
    Q, _ = torch.linalg.qr(torch.randn(B, k, k, dtype=torch.float64, generator=g))
    if kind == "clusters":
        d = torch.cat([torch.full((k // 3,), 1.0), torch.full((k // 3,), 1.0 + 1e-9), torch.linspace(2, 3, k - 2 * (k // 3))]).double()
    elif kind == "graded":
        d = torch.logspace(-12, 2, k, dtype=torch.float64)
    elif kind == "ritz":
        d = torch.cat([torch.arange(1.0, 9.0)[:min(8, k - 1)], 50.0 + 50.0 * torch.rand(k - min(8, k - 1), generator=g)]).double()
    elif kind == "block2":                          # two decoupled dense blocks
        T = torch.zeros(B, k, k, dtype=torch.float64)
        h = k // 2
        R1 = torch.randn(B, h, h, dtype=torch.float64, generator=g); R2 = torch.randn(B, k - h, k - h, dtype=torch.float64, generator=g)
        T[:, :h, :h] = R1 + R1.transpose(1, 2); T[:, h:, h:] = R2 + R2.transpose(1, 2)
        return T
    else:
        raise ValueError(kind)
    T = Q @ torch.diag_embed(d.expand(B, k)) @ Q.transpose(1, 2)
    return (T + T.transpose(1, 2)) * 0.5


kinds = ["random", "diagonal", "int_diagonal", "identity", "zero", "clusters", "graded", "ritz", "block2"]
nfail = 0
ran = {}
nflag = {}
for case in range(ncase):
    dtype = torch.float64 if ri(0, 2) else torch.float32
    k = ri(8, 520) if ri(0, 3) else ri(8, 140)
    p = min(k, ri(1, 256) if ri(0, 3) == 0 else ri(1, 16))
    B = ri(1, 5)
    uppest = bool(ri(0, 1))
    kind = kinds[ri(0, len(kinds) - 1)]
    scale = [1.0, 1.0, 1e-30 if dtype == torch.float32 else 1e-150, 1e30 if dtype == torch.float32 else 1e150][ri(0, 3)]
    Tm = make(kind, B, k) * scale
    ref = torch.linalg.eigvalsh(Tm)
    sl = slice(k - p, k) if uppest else slice(0, p)
    Td = torch.tril(Tm).to(dtype).to(dev)
    forms = []
    if K.small_eigh_tri_ok(k, p, dtype):
        forms.append(("tri", None))
    if K.small_eigh_big_ok(k, p, dtype):
        forms += [("big", 0), ("big", 1), ("big", 3)]
        if k >= 35 and (dtype == torch.float32 or k <= 614):
            forms.append(("big", 2))
    for name, algo in forms:
        K._workspace(1 << 22, dtype, dev).fill_(float("nan"))
        try:
            if name == "tri":
                lam, Y, info = K.small_eigh(Td, k, p, uppest=uppest, method="tri")
            else:
                lam, Y, info = K.small_eigh_big(Td, k, p, uppest=uppest, algo=algo)
        except Exception as e:                      # noqa
            if "UNSUPPORTED" in repr(e) or "-2" in repr(e):
                continue
            print(json.dumps({"case": case, "form": [name, algo], "error": repr(e)[:200]})); nfail += 1
            continue
        ran[(name, algo)] = ran.get((name, algo), 0) + 1
        lam, Y = lam.cpu().double(), Y.cpu().double()
        tol = 2e-12 if dtype == torch.float64 else 1e-4
        nrm = float(ref.abs().max())
        zero = nrm == 0.0
        nrm = nrm if nrm > 0 else 1.0                # (zero matrix: absolute errors)
        Tq = Tm.to(dtype).double()                   # what the kernel was given
        Yc = Y.transpose(1, 2)
        flagged = int(info.max()) != 0
        err = float((lam - ref[:, sl]).abs().max()) / nrm
        res = float((Tq @ Yc - Yc * lam.unsqueeze(1)).abs().max()) / nrm
        orth = float((Yc.transpose(1, 2) @ Yc - torch.eye(p, dtype=torch.float64)).abs().max())
        ok = flagged or (err < 10 * tol and res < 100 * tol and orth < 400 * tol and bool(torch.all(lam[:, 1:] >= lam[:, :-1])))
        if not math.isfinite(err) and not flagged:
            ok = False
        nflag[kind] = nflag.get(kind, 0) + (1 if flagged else 0)
        if not ok:
            nfail += 1
            print(json.dumps({"case": case, "form": [name, algo], "dtype": str(dtype), "B": B, "k": k, "p": p, "uppest": uppest,
                              "kind": kind, "scale": scale, "flagged": flagged, "err": err, "res": res, "orth": orth, "ok": ok}))
print(json.dumps({"cases": ncase, "failures": nfail, "calls_by_form": {str(kk): v for kk, v in ran.items()},
                  "flagged_calls_by_kind (redone on the library by the caller)": nflag}))
