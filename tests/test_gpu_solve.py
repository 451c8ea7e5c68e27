"""-m gpu: native HIP Krylov solvers + banded operator vs the oracle and the reference's golden outputs."""
import os
import warnings
import numpy as np
import pytest
import torch
from oracle import ops as oops, solve as osolve
from tests import cases
import xitorch_amd as xa
from xitorch_amd import kernels as K
from xitorch_amd.linalg import solve
from xitorch_amd.linalg import native_krylov as nk

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("B,N,hb,C,dtype", [(2, 1024, 63, 1, torch.float64), (3, 777, 5, 3, torch.float64),
                                            (1, 4096, 63, 8, torch.float64), (2, 1000, 20, 2, torch.float32),
                                            (2, 50, 63, 2, torch.float64), (1, 513, 1, 1, torch.float64)])
@pytest.mark.parametrize("trans", [False, True])
def test_banded_mm_vs_oracle(dev, B, N, hb, C, dtype, trans):
    g = torch.Generator().manual_seed(N + hb)
    band = torch.randn(B, 2 * hb + 1, N, dtype=dtype, generator=g)      # out-of-matrix entries hold garbage
    X = torch.randn(B, C, N, dtype=dtype, generator=g)
    op = oops.BandedOp(band.double())
    xo = X.double().transpose(-2, -1)
    ref = (op._rmm(xo) if trans else op._mm(xo)).transpose(-2, -1)
    Y = K.banded_mm(band.to(dev), X.to(dev), trans=trans).cpu().double()
    tol = 1e-13 if dtype == torch.float64 else 2e-6
    assert (Y - ref).abs().max().item() <= tol * (ref.abs().max().item() + 1) * 10


def _operators(case, A, M, dev):
    if case["op"] == "banded":
        return xa.BandedLinearOperator(A.to(dev), is_hermitian=False), None, oops.BandedOp(A)
    Aop = xa.LinearOperator.m(A.to(dev), is_hermitian=case["hermitian"])
    Mop = xa.LinearOperator.m(M.to(dev), is_hermitian=True) if M is not None else None
    return Aop, Mop, oops.DenseOp(A, case["hermitian"])


@pytest.mark.parametrize("case", cases.SOLVE_CASES, ids=[c["name"] for c in cases.SOLVE_CASES])
def test_krylov_vs_golden_and_oracle(dev, case):
    gold = np.load(os.path.join(GOLD, "solve_%s.npz" % case["name"]))
    A, B, E, M = cases.solve_inputs(case)
    Aop, Mop, oA = _operators(case, A, M, dev)
    fcn = getattr(nk, case["method"])
    pre = {k: xa.LinearOperator.m(P.to(dev), is_hermitian=True) for k, P in cases.solve_precond(case, A).items()} \
        if case["op"] != "banded" else {}
    tr = {}
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        X = fcn(Aop, B.to(dev), E.to(dev) if E is not None else None, Mop, trace=tr, **case["kwargs"], **pre)
    nconv = [w for w in wlist if issubclass(w.category, xa.ConvergenceWarning)]
    # a ConvergenceWarning is a failure, except in the case that pins the non-converging path (best iterate returned)
    assert bool(nconv) == bool(case.get("nonconv")), [str(w.message) for w in wlist]
    X = X.cpu()
    Xg, Xe = torch.from_numpy(gold["X"]), torch.from_numpy(gold["X_exact"])
    if case.get("gold_swapped"):
        # the reference's gmres returns its column-swapped work layout (ncols, *batch, n, 1) when E is given
        # (solve.py:349-432 never undoes it); the native method returns (*batch, n, ncols) like every other method
        Xg = Xg.squeeze(-1).movedim(0, -1)
    assert list(X.shape) == list(Xg.shape) == list(Xe.shape)
    kw = case["kwargs"]
    rtol, atol = kw.get("rtol", 1e-6), kw.get("atol", 1e-8)
    if case.get("nonconv"):
        # the best-residual iterate of the truncated iteration: same Krylov space, same minimiser as the reference's
        assert (X - Xg).norm().item() <= 1e-9 * float(gold["kappa"]) * Xg.norm().item()
        assert not tr["converged"] and tr["niter"] == int(gold["niter"])
        return
    # (1) the residual identity the reference tests assert (test_linop_fcns.py:467-468, 674-676)
    oM = oops.DenseOp(M, True) if M is not None else None
    AX = oA.mm(X)
    if E is not None:
        AX = AX - (oM.mm(X) if oM is not None else X) * E.unsqueeze(-2)
    if case["name"].startswith("cg_nonsym"):
        pass      # normal equations: the stopping test is on A^T(AX - B); checked through X below
    else:
        lim = torch.clamp(rtol * B.norm(dim=-2), min=atol)
        assert torch.all((AX - B).norm(dim=-2) <= lim * 1.001)
    # (2) against the reference's OWN output: both iterates satisfy |r| <= rtol |b|, so they lie within
    # 2 rtol kappa of each other, kappa = condition number of the iterated operator (from the fixture)
    kappa = float(gold["kappa"])
    assert (X - Xg).norm().item() <= 2.0 * rtol * kappa * Xg.norm().item(), \
        ((X - Xg).norm().item() / Xg.norm().item(), rtol, kappa)
    #     ... and the dense solution
    assert (X - Xe).norm().item() <= 2.0 * rtol * kappa * Xe.norm().item()
    # (3) same iteration path as the reference (identical algorithm, rounding-level differences only): the count
    # may move by one where a residual norm sits within rounding of the threshold
    assert abs(tr["niter"] - int(gold["niter"])) <= 1, (tr["niter"], int(gold["niter"]))
    assert tr["converged"]


def test_solve_frontend_defaults_and_backward(dev):
    # implicit (non-matrix) Hermitian operator -> default method cg; general -> bicgstab; gradients vs dense
    g = torch.Generator().manual_seed(11)
    n = 48
    R = torch.rand(2, n, n, dtype=torch.float64, generator=g)
    Asym = ((R + R.transpose(-2, -1)) * 0.05 + torch.eye(n, dtype=torch.float64)).to(dev)
    Agen = (0.1 * R + torch.eye(n, dtype=torch.float64)).to(dev)
    Bm = torch.rand(2, n, 3, dtype=torch.float64, generator=g).to(dev)

    class Wrapped(xa.LinearOperator):
        def __init__(self, mat, herm):
            super().__init__(mat.shape, is_hermitian=herm, dtype=mat.dtype, device=mat.device)
            self.mat = mat

        def _mv(self, x):
            return torch.matmul(self.mat, x.unsqueeze(-1)).squeeze(-1)

        def _rmv(self, x):
            return torch.matmul(self.mat.transpose(-2, -1), x.unsqueeze(-1)).squeeze(-1)

        def _getparamnames(self, prefix=""):
            return [prefix + "mat"]

    for mat, herm in ((Asym, True), (Agen, False)):
        m1 = mat.clone().requires_grad_()
        b1 = Bm.clone().requires_grad_()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")      # Hermitian + _rmv warning of the operator contract
            op = Wrapped(m1, herm)
        x = solve(op, b1, rtol=1e-12, atol=1e-14, posdef=True,
                  bck_options=dict(rtol=1e-12, atol=1e-14, posdef=True))
        xd = torch.linalg.solve(mat, Bm)
        assert torch.allclose(x, xd, rtol=1e-8, atol=1e-9)
        loss = (x ** 2).sum()
        gm, gb = torch.autograd.grad(loss, (m1, b1))
        m2 = mat.clone().requires_grad_()
        b2 = Bm.clone().requires_grad_()
        l2 = (torch.linalg.solve(m2, b2) ** 2).sum()
        gm2, gb2 = torch.autograd.grad(l2, (m2, b2))
        assert torch.allclose(gb, gb2, rtol=1e-6, atol=1e-8)
        assert torch.allclose(gm, gm2, rtol=1e-6, atol=1e-8)


def test_nonconvergence_warns_and_returns_best(dev):
    g = torch.Generator().manual_seed(2)
    n = 64
    R = torch.rand(1, n, n, dtype=torch.float64, generator=g)
    A = xa.LinearOperator.m((R + torch.eye(n)).to(dev), is_hermitian=False)
    B = torch.rand(1, n, 2, dtype=torch.float64, generator=g).to(dev)
    with pytest.warns(xa.ConvergenceWarning):
        X = nk.bicgstab(A, B, max_niter=2, rtol=1e-14, atol=1e-16, posdef=True)
    assert X.shape == (1, n, 2) and torch.isfinite(X).all()


def test_zero_rhs_shortcut(dev):
    A = xa.LinearOperator.m(torch.eye(8, dtype=torch.float64, device=dev) * 2, is_hermitian=True)
    X = nk.cg(A, torch.zeros(8, 2, dtype=torch.float64, device=dev))
    assert torch.all(X == 0)


def test_preconditioned_cg_and_bicgstab(dev):
    # precond / precond_l / precond_r are LinearOperators applied inside the native loops (solve.py:73,196-197)
    g = torch.Generator().manual_seed(31)
    n = 120
    R = torch.rand(2, n, n, dtype=torch.float64, generator=g)
    d = torch.linspace(1.0, 50.0, n, dtype=torch.float64)
    Asym = ((R + R.transpose(-2, -1)) * 0.05 + torch.diag(d)).to(dev)
    Agen = (0.1 * R + torch.diag(d)).to(dev)
    Bm = torch.rand(2, n, 2, dtype=torch.float64, generator=g).to(dev)
    Pinv = xa.LinearOperator.m(torch.diag(1.0 / d).to(dev), is_hermitian=True)       # Jacobi preconditioner
    tr0, tr1 = {}, {}
    x0 = nk.cg(xa.LinearOperator.m(Asym, True), Bm, rtol=1e-10, posdef=True, trace=tr0)
    x1 = nk.cg(xa.LinearOperator.m(Asym, True), Bm, rtol=1e-10, posdef=True, precond=Pinv, trace=tr1)
    ref = torch.linalg.solve(Asym, Bm)
    assert torch.allclose(x0, ref, rtol=1e-7, atol=1e-9) and torch.allclose(x1, ref, rtol=1e-7, atol=1e-9)
    assert tr1["niter"] < tr0["niter"]                    # the preconditioner must actually be used
    for kw in (dict(precond_r=Pinv), dict(precond_l=Pinv), dict(precond_l=Pinv, precond_r=Pinv)):
        x = nk.bicgstab(xa.LinearOperator.m(Agen, False), Bm, rtol=1e-10, posdef=True, **kw)
        assert torch.allclose(x, torch.linalg.solve(Agen, Bm), rtol=1e-7, atol=1e-9), kw
    with pytest.raises(TypeError):
        nk.cg(xa.LinearOperator.m(Asym, True), Bm, precond=torch.eye(n))


def test_many_rhs_nonhermitian_goes_through_the_mfma_kernel(dev):
    # 20 right-hand sides, general dense A: the panel product A X runs on K1wr (row orientation on the matrix cores,
    # LDS tile turn) — one pass over A, and NO transposed copy of the operator; results vs the dense solution
    from xitorch_amd.linalg._panel import PanelOperator
    g = torch.Generator().manual_seed(13)
    B, n, nc = 2, 256, 20
    R = torch.rand(B, n, n, dtype=torch.float64, generator=g)
    A = (0.1 * R + torch.diag(torch.linspace(1.0, 3.0, n, dtype=torch.float64))).to(dev)
    Bm = torch.rand(B, n, nc, dtype=torch.float64, generator=g).to(dev)
    op = PanelOperator(xa.LinearOperator.m(A, is_hermitian=False), [B], B, n)
    X = Bm.transpose(-2, -1).contiguous()                       # (B, nc, n) panel
    out = torch.empty_like(X)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    op.apply(X, out)
    torch.cuda.synchronize()
    assert getattr(op, "_matT", None) is None                   # no transposed copy any more
    assert torch.cuda.memory_allocated() - before < A.numel() * A.element_size()   # nothing operator-sized appeared
    ref = torch.matmul(A, Bm).transpose(-2, -1)
    assert (out - ref).abs().max().item() <= 1e-12 * ref.abs().max().item() * n ** 0.5
    op.apply(X, out, trans=True)                                # A^T X: K1w directly
    refT = torch.matmul(A.transpose(-2, -1), Bm).transpose(-2, -1)
    assert (out - refT).abs().max().item() <= 1e-12 * refT.abs().max().item() * n ** 0.5
    x = nk.bicgstab(xa.LinearOperator.m(A, is_hermitian=False), Bm, rtol=1e-11, atol=1e-13, posdef=True)
    assert torch.allclose(x, torch.linalg.solve(A, Bm), rtol=1e-8, atol=1e-10)
    # float32, ragged size: falls back to the generic kernels where K1w's tile constraints do not hold
    A32 = A[:, :250, :250].float().contiguous()
    B32 = Bm[:, :250, :16].float().contiguous()
    op32 = PanelOperator(xa.LinearOperator.m(A32, is_hermitian=False), [B], B, 250)
    X32 = B32.transpose(-2, -1).contiguous()
    o32 = torch.empty_like(X32)
    op32.apply(X32, o32)
    r32 = torch.matmul(A32.double(), B32.double()).transpose(-2, -1)
    assert (o32.double() - r32).abs().max().item() <= 3e-5 * r32.abs().max().item()


def test_gmres_device_state_semantics(dev):
    """Native GMRES (reference: solve.py:326-433): iterates checked on TRUE residuals; with resid_calc_every > 1 the
    same solution at fewer applies; one host read per iteration; many columns with per-column shifts E (which the
    reference cannot do: its swapped layout breaks for ncols > 1); float32; breakdown of a system whose Krylov space
    is exhausted early."""
    g = torch.Generator().manual_seed(5)
    n, nc = 96, 3
    R = torch.rand(2, n, n, dtype=torch.float64, generator=g)
    Amat = 0.1 * R + torch.eye(n, dtype=torch.float64)
    Bm = torch.rand(2, n, nc, dtype=torch.float64, generator=g)
    E = torch.rand(2, nc, dtype=torch.float64, generator=g) * 0.3
    A = xa.LinearOperator.m(Amat.to(dev), is_hermitian=False)
    ref = torch.linalg.solve(Amat, Bm)
    tr1, tr5 = {}, {}
    X1 = nk.gmres(A, Bm.to(dev), rtol=1e-10, atol=1e-12, posdef=True, trace=tr1).cpu()
    X5 = nk.gmres(A, Bm.to(dev), rtol=1e-10, atol=1e-12, posdef=True, resid_calc_every=5, trace=tr5).cpu()
    for X in (X1, X5):
        assert ((Amat @ X - Bm).norm(dim=-2) <= 1e-10 * Bm.norm(dim=-2) * 1.001).all()
        assert (X - ref).norm().item() <= 1e-8 * ref.norm().item()
    assert tr1["converged"] and tr5["converged"]
    assert tr5["arnoldi_steps"] in (tr1["arnoldi_steps"], tr1["arnoldi_steps"] + 1)
    assert tr5["napply"] < tr1["napply"]
    # one read of the status per iteration (+1 for the initial residual, +1 when the estimate triggers the check)
    assert tr1["host_syncs"] == tr1["arnoldi_steps"] + 1
    assert tr5["host_syncs"] <= tr5["arnoldi_steps"] + 3
    # per-column shifts with several columns
    XE = nk.gmres(A, Bm.to(dev), E.to(dev), rtol=1e-10, atol=1e-12, posdef=True).cpu()
    for c in range(nc):
        Ac = Amat - E[:, c].reshape(2, 1, 1) * torch.eye(n, dtype=torch.float64)
        xc = torch.linalg.solve(Ac, Bm[..., c:c + 1])
        assert (XE[..., c:c + 1] - xc).norm().item() <= 1e-8 * xc.norm().item()
    # float32
    A32 = xa.LinearOperator.m(Amat.float().to(dev), is_hermitian=False)
    X32 = nk.gmres(A32, Bm.float().to(dev), rtol=1e-4, atol=1e-6, posdef=True).cpu().double()
    assert (X32 - ref).norm().item() <= 1e-3 * ref.norm().item()
    # a system whose Krylov space is exhausted after one step (b is an eigenvector) next to generic ones: the
    # reference divides by h[k+1,k] = 0 there (NaN); here that system simply stays converged
    D = torch.diag(torch.linspace(1.0, 2.0, n, dtype=torch.float64)).unsqueeze(0).repeat(2, 1, 1)
    Bd = torch.rand(2, n, 1, dtype=torch.float64, generator=g)
    Bd[0] = 0.0
    Bd[0, 3, 0] = 1.0
    Xd = nk.gmres(xa.LinearOperator.m(D.to(dev), is_hermitian=False), Bd.to(dev), rtol=1e-10, atol=1e-12,
                  posdef=True).cpu()
    assert torch.isfinite(Xd).all()
    assert (D @ Xd - Bd).norm().item() <= 1e-9


def test_gmres_restarted_cycles_reach_a_tight_tolerance(dev):
    """(r04, extension) GMRES(m): cycles of m Arnoldi steps restarted from the true residual, the reference's stopping
    and best-iterate rules across the cycles (reference: un-restarted only, solve.py:384-389).  A restart length the run
    never reaches leaves the un-restarted path bit for bit."""
    from xitorch_amd.linalg import native_krylov as nk
    g = torch.Generator().manual_seed(21)
    Bn, N, nc = 2, 300, 2
    R = torch.randn(Bn, N, N, dtype=torch.float64, generator=g) / N ** 0.5
    Am = (0.9 * R + 2.0 * torch.eye(N, dtype=torch.float64)).to(dev)
    rhs = torch.randn(Bn, N, nc, dtype=torch.float64, generator=g).to(dev)
    A = xa.LinearOperator.m(Am, is_hermitian=False)
    Xref = torch.linalg.solve(Am, rhs)
    tr0, tr1, tr2 = {}, {}, {}
    X0 = nk.gmres(A, rhs, rtol=1e-10, atol=1e-12, max_niter=200, trace=tr0)
    assert tr0["converged"] and tr0["restarts"] == 0
    X1 = nk.gmres(A, rhs, rtol=1e-10, atol=1e-12, max_niter=2000, restart=8, trace=tr1)
    assert tr1["converged"] and tr1["restarts"] >= 2 and tr1["arnoldi_steps"] >= tr0["arnoldi_steps"]
    assert ((X1 - Xref).norm() / Xref.norm()).item() <= 1e-8
    # every cycle but the last has exactly 8 steps: restarts = floor((steps - 1) / 8)
    assert tr1["restarts"] == (tr1["arnoldi_steps"] - 1) // 8
    X2 = nk.gmres(A, rhs, rtol=1e-10, atol=1e-12, max_niter=200, restart=150, trace=tr2)
    assert tr2["restarts"] == 0 and tr2["arnoldi_steps"] == tr0["arnoldi_steps"] and torch.equal(X2, X0)
    # not converging within max_niter: warning + the best iterate over all cycles
    with pytest.warns(xa.ConvergenceWarning if hasattr(xa, "ConvergenceWarning") else Warning):
        tr3 = {}
        X3 = nk.gmres(A, rhs, rtol=1e-14, atol=1e-16, max_niter=20, restart=4, trace=tr3)
    assert not tr3["converged"] and tr3["restarts"] == 4
    r3 = (Am @ X3 - rhs).norm(dim=-2).max().item()
    assert abs(r3 - tr3["best_resid"]) <= 1e-9 * max(1.0, r3)
    with pytest.raises(Exception):
        nk.gmres(A, rhs, restart=0)


def test_gmres_default_max_niter_on_a_large_order(dev):
    """(r05) The reference's default ``max_niter=None`` means the operator's order (solve.py:366-367); an
    order beyond the 8192-vector cycle limit of the native least-squares kernel must not be refused up front — the run
    below converges in a few dozen steps.  Also through the ``solve(..., method="gmres")`` front end; an explicit
    restart length beyond the limit is refused."""
    from xitorch_amd.linalg import solve
    N, hb = 9000, 2
    g = torch.Generator().manual_seed(3)
    band = torch.zeros(1, 2 * hb + 1, N, dtype=torch.float64)
    band[:, hb] = 4.0 + torch.rand(1, N, dtype=torch.float64, generator=g)
    band[:, hb - 1] = -1.0
    band[:, hb + 1] = -0.8
    band[:, 0] = 0.1
    A = xa.BandedLinearOperator(band.to(dev))
    xs = torch.rand(1, N, 1, dtype=torch.float64, generator=g).to(dev)
    with torch.no_grad():
        rhs = A.mm(xs)
    tr = {}
    x = nk.gmres(A, rhs, rtol=1e-10, atol=1e-12, trace=tr)
    assert tr["converged"] and tr["arnoldi_steps"] < 200
    assert (x - xs).abs().max().item() < 1e-8
    x2 = solve(A, rhs, method="gmres", rtol=1e-10, atol=1e-12)
    assert (x2 - xs).abs().max().item() < 1e-8
    from xitorch_amd._capi import NativeLibraryError
    with pytest.raises(NativeLibraryError):
        nk.gmres(A, rhs, restart=9000, max_niter=20000)
