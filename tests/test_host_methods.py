"""not-gpu: the iterative methods on operators that live in HOST memory (device dispatch: the reference runs its methods on
whatever device the operator is on, xitorch/_impls/linalg/solve.py:69-433, _impls/linalg/symeig.py:100-227,
_impls/optimize/root/rootsolver.py) — `xitorch_amd/linalg/host_krylov.py`, `host_eig.py` and the host branch of the
Broyden model — against the reference's golden outputs (tests/golden/, written by make_golden.py from the live reference)
and the oracle.  The HIP path is covered by tests/test_gpu_*.py against the same goldens; that a device tensor never
reaches these drivers is asserted in tests/test_gpu_k1.py::test_device_operators_never_reach_the_host_drivers."""
import os
import re
import warnings
import numpy as np
import pytest
import torch
from oracle import ops as oops, rootfinder as oroot
from tests import cases
import xitorch_amd as xa
from xitorch_amd.linalg import native_krylov as nk, host_krylov, host_eig, solve, symeig
from xitorch_amd.linalg.native_eig import davidson
from xitorch_amd.optimize import native_root as nr, rootfinder
from xitorch_amd._capi import NativeLibraryError

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case", cases.SOLVE_CASES, ids=[c["name"] for c in cases.SOLVE_CASES])
def test_host_krylov_vs_reference_golden(case):
    gold = np.load(os.path.join(GOLD, "solve_%s.npz" % case["name"]))
    A, B, E, M = cases.solve_inputs(case)
    if case["op"] == "banded":
        Aop, Mop, oA = xa.BandedLinearOperator(A, is_hermitian=False), None, oops.BandedOp(A)
    else:
        Aop = xa.LinearOperator.m(A, is_hermitian=case["hermitian"])
        Mop = xa.LinearOperator.m(M, is_hermitian=True) if M is not None else None
        oA = oops.DenseOp(A, case["hermitian"])
    pre = {k: xa.LinearOperator.m(P, is_hermitian=True) for k, P in cases.solve_precond(case, A).items()} \
        if case["op"] != "banded" else {}
    before = dict(host_krylov.calls)
    tr = {}
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        X = getattr(nk, case["method"])(Aop, B, E, Mop, trace=tr, **case["kwargs"], **pre)
    assert host_krylov.calls[case["method"]] == before[case["method"]] + 1          # the host driver is what ran
    nconv = [w for w in wlist if issubclass(w.category, xa.ConvergenceWarning)]
    assert bool(nconv) == bool(case.get("nonconv")), [str(w.message) for w in wlist]
    Xg, Xe = torch.from_numpy(gold["X"]), torch.from_numpy(gold["X_exact"])
    if case.get("gold_swapped"):
        Xg = Xg.squeeze(-1).movedim(0, -1)        # the reference's gmres leaves its column-swapped layout with E (solve.py:432)
    assert list(X.shape) == list(Xg.shape) == list(Xe.shape)
    kw = case["kwargs"]
    rtol, kappa = kw.get("rtol", 1e-6), float(gold["kappa"])
    if case.get("nonconv"):
        assert (X - Xg).norm().item() <= 1e-9 * kappa * Xg.norm().item()
        assert not tr["converged"] and tr["niter"] == int(gold["niter"])
        return
    # the bar of the GPU tests (tests/test_gpu_solve.py): both iterates satisfy |r| <= rtol |b|, so they lie within
    # 2 rtol kappa of each other; same iteration path, the count may move by one where a residual norm sits within
    # rounding of the threshold (measured here: 18 of 19 cases equal, 1e-16 .. 2e-8 relative distance)
    assert (X - Xg).norm().item() <= 2.0 * rtol * kappa * Xg.norm().item(), (X - Xg).norm().item() / Xg.norm().item()
    assert (X - Xe).norm().item() <= 2.0 * rtol * kappa * Xe.norm().item()
    assert abs(tr["niter"] - int(gold["niter"])) <= 1, (tr["niter"], int(gold["niter"]))
    assert tr["converged"]


@pytest.mark.parametrize("case", cases.DAVIDSON_CASES + cases.DAVIDSON_CASES_F32,
                         ids=[c["name"] for c in cases.DAVIDSON_CASES + cases.DAVIDSON_CASES_F32])
def test_host_davidson_vs_reference_golden(case):
    gold = np.load(os.path.join(GOLD, "davidson_%s.npz" % case["name"]))
    mat, Mmat = cases.davidson_matrix(case), cases.davidson_M(case)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    Mop = xa.LinearOperator.m(Mmat, is_hermitian=True) if Mmat is not None else None
    n0 = host_eig.calls["davidson"]
    tr = {}
    evals, X = davidson(A, case["neig"], case["mode"], Mop, min_eps=case["min_eps"], v_init="randn", trace=tr)
    assert host_eig.calls["davidson"] == n0 + 1
    f32 = mat.dtype == torch.float32
    scale = max(1.0, float(np.abs(gold["evals"]).max()))
    assert evals.shape == gold["evals"].shape and bool(torch.all(evals[..., 1:] >= evals[..., :-1] - 1e-12))
    assert np.abs(evals.double().numpy() - gold["evals"]).max() <= (2e-6 if f32 else 1e-10) * scale
    MX = torch.matmul(Mmat, X) if Mmat is not None else X
    R = torch.matmul(mat, X) - MX * evals.unsqueeze(-2)
    assert R.abs().max().item() <= 10 * case["min_eps"]
    G = torch.matmul(X.transpose(-2, -1), MX)
    assert (G - torch.eye(G.shape[-1], dtype=G.dtype)).abs().max().item() < (1e-4 if f32 else 1e-9)
    assert abs(tr["niter"] - int(gold["niter"])) <= max(1, int(gold["niter"]) // 20), (tr["niter"], int(gold["niter"]))
    if not f32:
        sig = torch.linalg.svdvals(torch.matmul(torch.from_numpy(gold["X"]).transpose(-2, -1), MX))
        assert sig.min().item() >= 1.0 - 1e-8 and sig.max().item() <= 1.0 + 1e-8


@pytest.mark.parametrize("case", cases.ROOT_CASES, ids=[c["name"] for c in cases.ROOT_CASES])
def test_host_broyden_vs_reference_golden(case):
    gold = np.load(os.path.join(GOLD, "root_%s.npz" % case["name"]))
    fcn, y0, params = cases.root_inputs(case)
    tr = {}
    meth = case.get("method", "broyden1")
    y = getattr(nr, meth)(fcn, y0, params, trace=tr, **case["kwargs"])
    yg = torch.from_numpy(gold["y"])
    assert (y - yg).abs().max().item() <= 1e-10
    assert tr["nfev"] == int(gold["nfev"]) and tr["niter"] == int(gold["niter"])      # incl. Q1: the iterate BEFORE convergence
    yo = getattr(oroot, meth)(fcn, y0, params, **case["kwargs"])
    assert (y - yo).abs().max().item() <= 1e-10


def test_front_ends_on_host_tensors_with_gradients():
    """solve / symeig / rootfinder front ends with operators in host memory: forward through the host drivers, backward
    through the implicit-function formulas (the adjoint solves run on the host drivers too)."""
    g = torch.Generator().manual_seed(5)
    n = 24
    R = torch.rand(n, n, dtype=torch.float64, generator=g)
    amat = ((R + R.T) * 0.05 + torch.eye(n, dtype=torch.float64)).requires_grad_()
    bmat = torch.rand(2, n, 2, dtype=torch.float64, generator=g).requires_grad_()
    for method, kw in (("cg", dict(rtol=1e-10)), ("bicgstab", dict(rtol=1e-10)), ("gmres", dict(rtol=1e-10))):
        def f(a, b):
            return solve(xa.LinearOperator.m(a), b, method=method, bck_options=dict(method=method, **kw), **kw)
        x = f(amat, bmat)
        assert torch.allclose(torch.matmul(amat, x), bmat.expand_as(x), atol=1e-8)
        ga, gb = torch.autograd.grad(x.sum(), (amat, bmat))
        xd = torch.linalg.solve(amat, bmat)
        ga_d, gb_d = torch.autograd.grad(xd.sum(), (amat, bmat))
        assert torch.allclose(ga, ga_d, atol=1e-7) and torch.allclose(gb, gb_d, atol=1e-7), method
    ev, X = symeig(xa.LinearOperator.m((amat + amat.T) * 0.5, True), neig=3, mode="lowest", method="davidson", min_eps=1e-9)
    ev_d = torch.linalg.eigvalsh((amat + amat.T) * 0.5)[:3]
    assert torch.allclose(ev, ev_d, atol=1e-9)
    gd, = torch.autograd.grad(ev.sum(), amat, retain_graph=True)
    gd_d, = torch.autograd.grad(ev_d.sum(), amat)
    assert torch.allclose(gd, gd_d, atol=1e-6)

    def fcn(y, a):
        return torch.tanh(torch.matmul(a, y.unsqueeze(-1)).squeeze(-1) * 0.3 + 0.1) + y
    y0 = torch.zeros(2, n, dtype=torch.float64)
    y = rootfinder(fcn, y0, params=(amat,), method="broyden1", f_tol=1e-10)
    assert fcn(y, amat).abs().max().item() < 1e-7
    gy, = torch.autograd.grad(y.sum(), amat)
    assert torch.isfinite(gy).all()


def test_host_drivers_import_nothing_from_the_oracle_and_refuse_device_tensors():
    pkg = os.path.join(ROOT, "xitorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), os.path.join(dirpath, f)
    # the host problem set-up refuses anything that is not host memory (a meta tensor stands in for a device tensor)
    A = xa.LinearOperator.m(torch.eye(4, dtype=torch.float64))

    class Fake:
        device, dtype, shape, is_hermitian = torch.device("meta"), torch.float64, (4, 4), True
    with pytest.raises(NativeLibraryError):
        host_krylov._HostProblem(Fake(), torch.zeros(4, 1), None, None, [], True, False)
    with pytest.raises(NativeLibraryError):
        host_eig.davidson(Fake(), 1, "lowest")
    assert A.shape == (4, 4)
