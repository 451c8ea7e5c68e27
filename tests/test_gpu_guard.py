"""-m gpu: the a-posteriori guard of the native block Davidson (r04).

The reference re-orthonormalises the WHOLE basis every iteration (tallqr of [V, t], xitorch/_utils/tensor.py:8-19,
_impls/linalg/symeig.py:207-223), so it cannot return two copies of one eigenpair.  This build orthonormalises only the
new panel; what stands in for the reference's guarantee is a check of every Rayleigh-Ritz block (max|X^T M X - I|,
`xk_ritz_guard`) read with the iteration's status: a block above GUARD_BAD is never returned, the run rolls back to the
last basis width whose block passed and continues with re-orthogonalised passes.  Tested here: the kernel against
torch, the status fold, the configurations that returned duplicated eigenpairs in round 3 with ONE projection pass
forced (now caught and repaired), the K3 fallback inside the loop, and the fall-back re-run.
"""
import warnings
import pytest
import torch
import xitorch_amd as xa
from xitorch_amd import kernels as K, synthetic
from xitorch_amd.linalg import native_eig
from xitorch_amd.linalg.native_eig import davidson

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("P,N", [(1, 77), (3, 1000), (6, 16384), (8, 4099), (12, 900), (20, 2048), (40, 333)])
def test_ritz_guard_kernel_vs_torch(dev, dtype, P, N):
    g = torch.Generator().manual_seed(P * 1000 + N)
    B = 3
    ld = (N + 7) // 8 * 8
    Q, _ = torch.linalg.qr(torch.randn(B, N, P, dtype=torch.float64, generator=g))
    Q = Q + 1e-3 * torch.randn(B, N, P, dtype=torch.float64, generator=g)            # visibly not orthonormal
    X = torch.zeros(B, P + 2, ld, dtype=dtype, device=dev)
    X[:, :P, :N] = Q.transpose(1, 2).to(dtype).to(dev)
    X[:, P:, :N] = 7.0                                                                # rows beyond P are not read
    Xd = X[:, :P, :N].double()
    want = (Xd @ Xd.transpose(1, 2) - torch.eye(P, dtype=torch.float64, device=dev)).abs().amax(dim=(1, 2))
    orth = torch.zeros(B, dtype=dtype, device=dev)
    K.ritz_guard(X, orth, P, ld)
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    assert (orth.double() - want).abs().max().item() <= tol * max(1.0, want.max().item())
    # accumulates a maximum: a second, better block does not lower it; a worse one raises it
    before = orth.clone()
    Xo = torch.zeros_like(X)
    Xo[:, :P, :N] = torch.linalg.qr(Q)[0].transpose(1, 2).to(dtype).to(dev)
    K.ritz_guard(Xo, orth, P, ld)
    assert torch.equal(orth, before)
    # with an overlap operator: <X_c, (M X)_d>
    Mx = torch.zeros_like(X)
    Mx[:, :P, :N] = (2.0 * Q + 0.1).transpose(1, 2).to(dtype).to(dev)
    orth2 = torch.zeros(B, dtype=dtype, device=dev)
    K.ritz_guard(X, orth2, P, ld, MX=Mx)
    raw = (Xd @ Mx[:, :P, :N].double().transpose(1, 2) - torch.eye(P, dtype=torch.float64, device=dev)).abs()
    # raw[c, d] = |<X_c, (M X)_d> - delta|; the one-kernel form (P <= 8) reads c <= d only (M is symmetric)
    want2 = (torch.triu(raw) if P <= 8 else raw).amax(dim=(1, 2))
    assert (orth2.double() - want2).abs().max().item() <= 10 * tol * max(1.0, want2.max().item())
    # non-finite entries count as an infinite deviation
    X[1, 0, 3] = float("nan")
    orth3 = torch.zeros(B, dtype=dtype, device=dev)
    K.ritz_guard(X, orth3, P, ld)
    assert torch.isinf(orth3[1]) and torch.isfinite(orth3[0])


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_group_status_folds_and_rezeroes_the_guard(dev, dtype):
    B = 70
    g = torch.Generator().manual_seed(5)
    rmax = torch.rand(B, dtype=dtype, generator=g).to(dev)
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    orth = torch.rand(B, dtype=dtype, generator=g).to(dev)
    want = float(orth.max().double())
    st = torch.full((5,), -1.0, dtype=torch.float64, device=dev)
    K.group_status(rmax, info, None, st, orth=orth)
    out = st.tolist()
    assert out[0] == float(rmax.max().double()) and out[4] == want and out[3] == -1.0
    assert float(orth.abs().max()) == 0.0
    orth[7] = float("nan")
    K.group_status(rmax, info, None, st, orth=orth)
    assert st.tolist()[4] == float("inf")
    with pytest.raises(Exception):
        K.group_status(rmax, info, None, torch.zeros(3, dtype=torch.float64, device=dev), orth=orth)


# the configurations of profiles/r03_orth_passes_scan.jsonl that returned duplicated eigenpairs (eigenvalue error 50,
# orthonormality error 1.0) or failed their panel Cholesky with ONE projection pass throughout
R03_FAILURES = [(900, 8), (900, 10), (900, 16), (2048, 8), (2048, 10), (2048, 16)]


@pytest.mark.parametrize("N,p", R03_FAILURES)
def test_guard_catches_the_r03_failures_with_one_pass_forced(dev, N, p):
    mat = synthetic.dense_symmetric(2, N, "S1", dtype=torch.float64, device=dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    exact = torch.linalg.eigvalsh(mat)[:, :p]
    tr = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, X = davidson(A, p, "lowest", min_eps=1e-8, orth_passes=1, trace=tr, max_niter=600)
    assert tr["stop_reason"] == "converged", tr["stop_reason"]
    assert len(tr["orth_redo"]) >= 1, "one pass throughout must trip the guard on this configuration"
    for r in tr["orth_redo"]:
        assert r["k_to"] < r["k_from"] and (r["guard"] > native_eig.GUARD_BAD[torch.float64] or r["chol_flag"] != 0)
    assert (ev - exact).abs().max().item() <= 1e-9
    G = X.transpose(1, 2) @ X
    assert (G - torch.eye(p, dtype=G.dtype, device=dev)).abs().max().item() <= 1e-8
    R = mat @ X - X * ev.unsqueeze(-2)
    assert R.abs().max().item() <= 1e-7
    # the default needs no repair on the same problem, and returns the same eigenvalues
    tr2 = {}
    ev2, _ = davidson(A, p, "lowest", min_eps=1e-8, trace=tr2, max_niter=600)
    assert tr2["orth_redo"] == [] and max(tr2["orth_guard_history"]) <= native_eig.GUARD_GOOD[torch.float64]
    assert (ev2 - exact).abs().max().item() <= 1e-9


def test_guard_in_fp32_one_pass_forced(dev):
    # fp32, mixed convergence, one pass forced: whatever happens to the basis, what is returned passes the guard
    mat = synthetic.dense_symmetric(2, 900, "S1", dtype=torch.float32, device=dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    exact = torch.linalg.eigvalsh(mat.double())[:, :8]
    tr = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, X = davidson(A, 8, "lowest", min_eps=2e-3, orth_passes=1, trace=tr, max_niter=400)
    assert tr["stop_reason"] == "converged"
    assert (ev.double() - exact).abs().max().item() <= 5e-4
    G = X.double().transpose(1, 2) @ X.double()
    assert (G - torch.eye(8, dtype=G.dtype, device=dev)).abs().max().item() <= native_eig.GUARD_BAD[torch.float32]


def test_returned_block_always_passes_the_guard_two_groups_and_M(dev):
    # guard values are part of the trace on every path: two pipelined groups (fused chain), overlap operator M
    # (kernel-by-kernel chain with M X), preconditioner, thick restart (guard of the kept block)
    N, B, p = 700, 4, 5
    mat = synthetic.dense_symmetric(B, N, "S1", dtype=torch.float64, device=dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    g = torch.Generator().manual_seed(3)
    R = torch.randn(B, N, N, dtype=torch.float64, generator=g) * (0.3 / N ** 0.5)
    Mm = (torch.eye(N, dtype=torch.float64) + R @ R.transpose(1, 2)).to(dev)
    Mm = (Mm + Mm.transpose(1, 2)) * 0.5
    Mop = xa.LinearOperator.m(Mm, is_hermitian=True)
    for kw in (dict(overlap=True), dict(M=Mop), dict(precond="diag"), dict(restart=4 * p), dict(chain="kernels"),
               dict(orth_passes=1, overlap=True)):
        tr = {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev, X = davidson(A, p, "lowest", min_eps=1e-8, trace=tr, max_niter=800, **kw)
        assert tr["stop_reason"] == "converged", (kw.keys(), tr["stop_reason"])
        assert len(tr["orth_guard_history"]) == tr["niter"]
        MX = Mm @ X if "M" in kw else X
        G = X.transpose(1, 2) @ MX
        assert (G - torch.eye(p, dtype=G.dtype, device=dev)).abs().max().item() <= 1e-8, list(kw.keys())
        if "M" in kw:
            L = torch.linalg.cholesky(Mm)
            Li = torch.linalg.inv(L)
            exact = torch.linalg.eigvalsh(Li @ mat @ Li.transpose(1, 2))[:, :p]
        else:
            exact = torch.linalg.eigvalsh(mat)[:, :p]
        assert (ev - exact).abs().max().item() <= 1e-9, list(kw.keys())


def test_k3_fallback_inside_davidson_clears_what_the_speculation_left(dev, monkeypatch):
    # a flagged K3t result (zero eigenvectors) feeds the speculative orthonormalisation, whose
    # CholeskyQR then sets the STICKY Cholesky flag; the repeated step is fine, but the driver used to raise "panel Gram
    # matrix is not positive definite" on the stale flag.  Force two flagged results and require a clean run.
    real = K.small_eigh
    calls = {"tri": 0}

    def flaky(T, k, p, uppest=False, max_sweeps=16, method="jacobi"):
        lam, Y, aux = real(T, k, p, uppest=uppest, max_sweeps=max_sweeps, method=method)
        if method == "tri":
            calls["tri"] += 1
            if calls["tri"] in (2, 5):
                Y = Y.clone()
                Y[0].zero_()                       # an annihilated iterate, as K3t's self-check reports it
                aux = aux.clone()
                aux[0] = 1
        return lam, Y, aux
    monkeypatch.setattr(K, "small_eigh", flaky)
    mat = synthetic.dense_symmetric(2, 600, "S1", dtype=torch.float64, device=dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    tr = {}
    ev, X = davidson(A, 6, "lowest", min_eps=1e-8, trace=tr)
    assert tr["k3_fallbacks"] == 2 and tr["stop_reason"] == "converged" and tr["orth_redo"] == []
    exact = torch.linalg.eigvalsh(mat)[:, :6]
    assert (ev - exact).abs().max().item() <= 1e-10
    # the same on the two-group pipeline
    calls["tri"] = 0
    tr = {}
    ev, X = davidson(A, 6, "lowest", min_eps=1e-8, trace=tr, overlap=True)
    assert tr["k3_fallbacks"] >= 1 and tr["stop_reason"] == "converged"
    assert (ev - exact).abs().max().item() <= 1e-10


def test_guard_failure_without_a_rollback_point_reruns_then_raises(dev, monkeypatch):
    # a guard that can never pass (threshold below rounding): no roll-back point ever exists, the wrapper repeats the
    # run with three passes and then raises — a block that fails the guard is never returned
    mat = synthetic.dense_symmetric(1, 300, "S1", dtype=torch.float64, device=dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    monkeypatch.setitem(native_eig.GUARD_GOOD, torch.float64, 1e-30)
    monkeypatch.setitem(native_eig.GUARD_BAD, torch.float64, 1e-29)
    with pytest.raises(RuntimeError, match="lost its orthonormality"):
        davidson(A, 4, "lowest", min_eps=1e-8)


def test_roll_back_on_the_last_allowed_iteration_returns_the_best_iterate_recorded_before(dev):
    """ `rollback()` keeps the best iterate: a guard failure on the LAST allowed iteration must return the block
    that was best before the void step — with its own eigenvalues and residual — like the reference returns its best
    iterate with a warning (symeig.py:196-200); nothing of the void step may leak into it."""
    N, p = 900, 8
    mat = synthetic.dense_symmetric(2, N, "S1", dtype=torch.float64, device=dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    tr = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        davidson(A, p, "lowest", min_eps=1e-8, orth_passes=1, trace=tr, max_niter=600)
    assert len(tr["orth_redo"]) >= 1
    it_void = tr["orth_redo"][0]["iter"]                     # the iteration whose step the guard voided
    # the same run, stopped right after that step (max_niter counts Rayleigh-Ritz steps) and one step earlier
    tr_a, tr_b = {}, {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev_a, X_a = davidson(A, p, "lowest", min_eps=1e-8, orth_passes=1, trace=tr_a, max_niter=it_void)
        ev_b, X_b = davidson(A, p, "lowest", min_eps=1e-8, orth_passes=1, trace=tr_b, max_niter=it_void - 1)
    assert tr_a["stop_reason"] != "converged" and tr_b["stop_reason"] != "converged"
    # what comes back after the void last step is the best block of the steps before it
    assert torch.equal(ev_a, ev_b) and torch.equal(X_a, X_b)
    G = X_a.transpose(1, 2) @ X_a
    assert (G - torch.eye(p, dtype=G.dtype, device=dev)).abs().max().item() <= native_eig.GUARD_BAD[torch.float64]
    R = mat @ X_a - X_a * ev_a.unsqueeze(-2)
    assert abs(R.abs().max().item() - tr_a["best_resid"]) <= 1e-12 * max(1.0, tr_a["best_resid"]) + 1e-10
