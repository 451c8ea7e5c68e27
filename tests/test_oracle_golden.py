"""not-gpu: the oracle (CPU restatement) against the golden outputs of the REAL reference.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py, which imports the
reference from /root/reference and also asserts bit-equality with the oracle at generation time.
Here (no reference available) the oracle must still reproduce them.
"""
import os
import warnings
import numpy as np
import pytest
import torch
from oracle import ops as oops, symeig as osym, solve as osolve, rootfinder as oroot
from tests import cases

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FAST_DAVIDSON = [c for c in cases.DAVIDSON_CASES if c["n"] <= 600 or c["kind"] != "alarge"]


@pytest.mark.parametrize("case", FAST_DAVIDSON, ids=[c["name"] for c in FAST_DAVIDSON])
def test_oracle_davidson(case):
    gold = np.load(os.path.join(GOLD, "davidson_%s.npz" % case["name"]))
    mat = cases.davidson_matrix(case)
    Mmat = cases.davidson_M(case)
    oM = oops.DenseOp(Mmat, True) if Mmat is not None else None
    tr = {}
    ev, X = osym.davidson(oops.DenseOp(mat, True), case["neig"], case["mode"], oM, min_eps=case["min_eps"], trace=tr)
    scale = max(1.0, np.abs(gold["evals"]).max())
    assert np.abs(ev.numpy() - gold["evals"]).max() <= 1e-11 * scale
    assert np.abs(ev.numpy() - gold["evals_exact"]).max() <= 1e-10 * scale
    assert abs(tr["niter"] - int(gold["niter"])) <= 1       # bit-equal on the generating machine
    MX = torch.matmul(Mmat, X) if Mmat is not None else X
    assert (torch.matmul(mat, X) - MX * ev.unsqueeze(-2)).abs().max().item() <= 10 * case["min_eps"]


@pytest.mark.parametrize("case", cases.EXACTEIG_CASES, ids=[c["name"] for c in cases.EXACTEIG_CASES])
def test_oracle_exacteig(case):
    # the dense method at the reference benchmark's shapes (benchmarks_solve.py:37-59), reference outputs as fixtures
    gold = np.load(os.path.join(GOLD, "exacteig_%s.npz" % case["name"]))
    A, M = cases.exacteig_inputs(case)
    ev, X = osym.exacteig(oops.DenseOp(A, True), case["neig"], case["mode"], oops.DenseOp(M, True) if M is not None else None)
    assert np.abs(ev.numpy() - gold["evals"]).max() <= 1e-12
    MX = torch.matmul(M, X) if M is not None else X
    assert (torch.matmul(A, X) - MX * ev.unsqueeze(-2)).abs().max().item() <= 1e-11
    sig = torch.linalg.svdvals(torch.matmul(torch.from_numpy(gold["X"]).transpose(-2, -1), MX))
    assert sig.min().item() >= 1 - 1e-10 and sig.max().item() <= 1 + 1e-10


@pytest.mark.parametrize("case", cases.DAVIDSON_CASES_F32, ids=[c["name"] for c in cases.DAVIDSON_CASES_F32])
def test_oracle_davidson_fp32_mixed_convergence(case):
    # fp32 reference golden (r04): on the generating machine the oracle is bit-equal to the reference (make_golden.py
    # asserts torch.equal); here, with whatever BLAS threading the box has, within fp32 rounding of it
    gold = np.load(os.path.join(GOLD, "davidson_%s.npz" % case["name"]))
    mat = cases.davidson_matrix(case)
    assert mat.dtype == torch.float32
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)          # (several MKL threads: the fp32 torch.inverse of the reference's tallqr stalls)
    try:
        tr = {}
        ev, X = osym.davidson(oops.DenseOp(mat, True), case["neig"], case["mode"], None, min_eps=case["min_eps"], trace=tr)
    finally:
        torch.set_num_threads(nthreads)
    assert ev.dtype == torch.float32
    assert np.abs(ev.numpy() - gold["evals"]).max() <= 2e-4           # fp32: eps * |A| ~ 1e-5, resid^2 / gap ~ 1e-4
    assert np.abs(ev.numpy().astype(np.float64) - gold["evals_exact"]).max() <= 5e-4
    assert abs(tr["niter"] - int(gold["niter"])) <= 2
    assert (torch.matmul(mat, X) - X * ev.unsqueeze(-2)).abs().max().item() <= 10 * case["min_eps"]


@pytest.mark.parametrize("case", cases.SOLVE_CASES, ids=[c["name"] for c in cases.SOLVE_CASES])
def test_oracle_solve(case):
    gold = np.load(os.path.join(GOLD, "solve_%s.npz" % case["name"]))
    A, B, E, M = cases.solve_inputs(case)
    oA = oops.DenseOp(oops.BandedOp(A).fullmatrix(), False) if case["op"] == "banded" else \
        oops.DenseOp(A, case["hermitian"])
    oM = oops.DenseOp(M, True) if M is not None else None
    pre = {k: oops.DenseOp(P, True) for k, P in cases.solve_precond(case, A).items()} if case["op"] != "banded" else {}
    tr = {}
    with warnings.catch_warnings():
        # a ConvergenceWarning is a failure, except in the case that pins the non-converging path
        warnings.simplefilter("ignore" if case.get("nonconv") else "error")
        X = getattr(osolve, case["method"])(oA, B, E, oM, trace=tr, **case["kwargs"], **pre)
    assert bool(tr["converged"]) == bool(gold["converged"]) == (not case.get("nonconv"))
    assert np.abs(X.numpy() - gold["X"]).max() <= 1e-9 * max(1.0, np.abs(gold["X"]).max())
    assert abs(tr["niter"] - int(gold["niter"])) <= 1


@pytest.mark.parametrize("case", cases.ROOT_CASES, ids=[c["name"] for c in cases.ROOT_CASES])
def test_oracle_root(case):
    gold = np.load(os.path.join(GOLD, "root_%s.npz" % case["name"]))
    fcn, y0, params = cases.root_inputs(case)
    tr = {}
    y = getattr(oroot, case.get("method", "broyden1"))(fcn, y0, params, trace=tr, **case["kwargs"])
    assert np.abs(y.numpy() - gold["y"]).max() <= 1e-10
    assert tr["nfev"] == int(gold["nfev"])


def test_banded_op_matches_dense():
    from xitorch_amd import synthetic as syn
    band = syn.banded(2, 300, hb=7)
    op = oops.BandedOp(band)
    full = op.fullmatrix()
    x = torch.randn(2, 300, 3, dtype=torch.float64)
    assert torch.allclose(op._mm(x), torch.matmul(full, x))
    assert torch.allclose(op._rmm(x), torch.matmul(full.transpose(-2, -1), x))
