"""-m gpu: native HIP Davidson vs (a) the oracle on the same inputs, (b) the reference's golden outputs."""
import os
import numpy as np
import warnings
import pytest
import torch
from oracle import ops as oops, symeig as osym
from tests import cases
import xitorch_amd as xa
from xitorch_amd import kernels as K
from xitorch_amd.linalg import symeig
from xitorch_amd.linalg.native_eig import davidson

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _check_pairs(mat, evals, evecs, ref_evals, min_eps, exact_evals=None, Mmat=None):
    # eigenvalues: 1e-10 (north_star tolerance), ascending order identical
    scale = max(1.0, float(np.abs(ref_evals).max()))
    assert np.all(np.diff(evals, axis=-1) >= -1e-12), "eigenvalues must be ascending"
    assert np.abs(evals - ref_evals).max() <= 1e-10 * scale
    if exact_evals is not None:
        assert np.abs(evals - exact_evals).max() <= 1e-10 * scale
    # residual identity A X = X E (reference test style, test_linop_fcns.py:174-176)
    X = torch.from_numpy(evecs)
    MX = torch.matmul(Mmat, X) if Mmat is not None else X
    R = torch.matmul(mat, X) - MX * torch.from_numpy(evals).unsqueeze(-2)
    assert R.abs().max().item() <= 10 * min_eps
    # (M-)orthonormality
    G = torch.matmul(X.transpose(-2, -1), MX)
    assert (G - torch.eye(G.shape[-1], dtype=G.dtype)).abs().max().item() < 1e-9


@pytest.mark.parametrize("case", cases.DAVIDSON_CASES, ids=[c["name"] for c in cases.DAVIDSON_CASES])
def test_davidson_vs_golden_and_oracle(dev, case):
    gold = np.load(os.path.join(GOLD, "davidson_%s.npz" % case["name"]))
    mat = cases.davidson_matrix(case)
    Mmat = cases.davidson_M(case)
    A = xa.LinearOperator.m(mat.to(dev), is_hermitian=True)
    Mop = xa.LinearOperator.m(Mmat.to(dev), is_hermitian=True) if Mmat is not None else None
    tr = {}
    evals, evecs = davidson(A, case["neig"], case["mode"], Mop, min_eps=case["min_eps"], v_init="randn", trace=tr)
    evals, evecs = evals.cpu().numpy(), evecs.cpu().numpy()
    assert evals.shape == gold["evals"].shape
    _check_pairs(mat, evals, evecs, gold["evals"], case["min_eps"], gold["evals_exact"], Mmat)
    # same iteration path as the reference: same start block, same algorithm -> same count (one more or less only
    # where max|resid| sits within rounding of min_eps when the reference stops).  The mixed-convergence cases (60-75
    # iterations, the last 40 of them fed by the rounding-noise residuals of pairs that converged long ago) are
    # chaotic in the rounding: 5 % there
    slack = max(1, int(gold["niter"]) // 20)
    assert abs(tr["niter"] - int(gold["niter"])) <= slack, (tr["niter"], int(gold["niter"]))
    # subspace parity with the reference's own eigenvectors: sigma_min(X_ref^T M X) >= 1 - 1e-8, i.e. the
    # two invariant subspaces coincide to 1e-4 rad; signs / rotations inside the subspace are free (quirk Q15)
    Xr = torch.from_numpy(gold["X"])
    Xm = torch.from_numpy(evecs)
    MX = torch.matmul(Mmat, Xm) if Mmat is not None else Xm
    sig = torch.linalg.svdvals(torch.matmul(Xr.transpose(-2, -1), MX))
    assert sig.min().item() >= 1.0 - 1e-8 and sig.max().item() <= 1.0 + 1e-8, (sig.min().item(), sig.max().item())
    # ... and entrywise up to sign where the wanted eigenvalues are separated
    probe = [int(i) for i in gold["probe"]]
    sep_ok = np.abs(np.diff(gold["evals"], axis=-1)).min() > 1e-6 if gold["evals"].shape[-1] > 1 else True
    if sep_ok:
        assert np.abs(np.abs(evecs[..., probe, :]) - gold["absX_probe"]).max() < 1e-6
    # the live oracle on the same inputs agrees with the fixture (oracle is the checker)
    tr_o = {}
    oM = oops.DenseOp(Mmat, True) if Mmat is not None else None
    ev_o, _ = osym.davidson(oops.DenseOp(mat, True), case["neig"], case["mode"], oM, min_eps=case["min_eps"], trace=tr_o)
    assert np.abs(ev_o.numpy() - gold["evals"]).max() <= 1e-12 * max(1.0, np.abs(gold["evals"]).max())


@pytest.mark.parametrize("case", cases.DAVIDSON_CASES_F32, ids=[c["name"] for c in cases.DAVIDSON_CASES_F32])
def test_davidson_fp32_mixed_convergence_vs_reference_golden(dev, case):
    # (r04) the mixed-convergence regime in fp32 against the reference's own fp32 run (43 iterations; the six separated
    # eigenvalues of S1 converge early, the two wanted pairs in the dense part late).  Tolerances are fp32's and are
    # stated here: eigenvalues 5e-4 absolute on a spectrum of scale 100 (eps32 * |A| ~ 1e-5, resid^2 / gap ~ 1e-4, both
    # sides carry them), residual 10 * min_eps, subspace overlap 5e-3 (residual 2e-3 over a gap of 0.056 allows an
    # angle of 0.04 rad inside the dense part), orthonormality 1e-4, iteration count within 12 % (43 reference iterations;
    # the last ~20 are driven by fp32 rounding noise of pairs that converged long ago, so the count moves with the
    # summation order of the panel kernel: 44 on the upper-triangle kernel, 47 on the full-matrix one).
    gold = np.load(os.path.join(GOLD, "davidson_%s.npz" % case["name"]))
    mat = cases.davidson_matrix(case)
    assert mat.dtype == torch.float32
    A = xa.LinearOperator.m(mat.to(dev), is_hermitian=True)
    tr = {}
    evals, evecs = davidson(A, case["neig"], case["mode"], min_eps=case["min_eps"], v_init="randn", trace=tr)
    assert evals.dtype == torch.float32 and tr["stop_reason"] == "converged"
    ev, X = evals.cpu().double(), evecs.cpu().double()
    assert torch.all(ev[..., 1:] >= ev[..., :-1])
    assert (ev - torch.from_numpy(gold["evals"]).double()).abs().max().item() <= 5e-4
    assert (ev - torch.from_numpy(gold["evals_exact"])).abs().max().item() <= 5e-4
    R = torch.matmul(mat.double(), X) - X * ev.unsqueeze(-2)
    assert R.abs().max().item() <= 10 * case["min_eps"]
    G = torch.matmul(X.transpose(-2, -1), X)
    assert (G - torch.eye(G.shape[-1], dtype=G.dtype)).abs().max().item() <= 1e-4
    sig = torch.linalg.svdvals(torch.matmul(torch.from_numpy(gold["X"]).double().transpose(-2, -1), X))
    assert sig.min().item() >= 1.0 - 5e-3 and sig.max().item() <= 1.0 + 5e-3, (sig.min().item(), sig.max().item())
    assert abs(tr["niter"] - int(gold["niter"])) <= 5, (tr["niter"], int(gold["niter"]))
    assert tr["orth_redo"] == [] and max(tr["orth_guard_history"]) <= 2e-4, tr["orth_guard_history"]


def test_symeig_frontend_davidson_and_custom_operator(dev):
    # method="davidson" through the functional, on a user-defined implicit operator (ALarge of the reference tests)
    n = 400

    class ALarge(xa.LinearOperator):
        def __init__(self, shape, dtype, device):
            super().__init__(shape, is_hermitian=True, dtype=dtype, device=device)
            self.b = torch.arange(shape[-1], dtype=dtype, device=device).repeat(*shape[:-2], 1)

        def _mv(self, x):
            return x * self.b + 1e-3 * (torch.roll(x, 1, -1) + torch.roll(x, -1, -1))

        def _getparamnames(self, prefix=""):
            return [prefix + "b"]

    for shape in [(n, n), (2, n, n), (2, 3, n, n)]:
        A = ALarge(shape, torch.float64, dev)
        for mode in ("lowest", "uppermost"):
            evals, evecs = symeig(A, neig=2, mode=mode, method="davidson", min_eps=1e-8)
            assert list(evals.shape) == [*shape[:-2], 2] and list(evecs.shape) == [*shape[:-2], n, 2]
            AX = A.mm(evecs)
            assert torch.allclose(AX, evecs * evals.unsqueeze(-2), atol=1e-6)
            if mode == "lowest":
                assert evals.max().item() < 1.1 and evals.min().item() > -0.1
            else:
                assert evals.min().item() > n - 2.1


def test_davidson_generalized_M(dev):
    g = torch.Generator().manual_seed(3)
    n = 120
    R = torch.rand(2, n, n, dtype=torch.float64, generator=g)
    Amat = (R + R.transpose(-2, -1)) * 0.5 + torch.diag(torch.arange(n, dtype=torch.float64) * 0.5)
    R2 = torch.rand(2, n, n, dtype=torch.float64, generator=g)
    Mmat = 0.02 * (R2 + R2.transpose(-2, -1)) + torch.eye(n, dtype=torch.float64)
    ev_o, X_o = osym.exacteig(oops.DenseOp(Amat, True), 3, "lowest", oops.DenseOp(Mmat, True))
    A = xa.LinearOperator.m(Amat.to(dev), True)
    M = xa.LinearOperator.m(Mmat.to(dev), True)
    ev, X = davidson(A, 3, "lowest", M, min_eps=1e-9)
    assert torch.allclose(ev.cpu(), ev_o, atol=1e-9)
    Xc = X.cpu()
    assert torch.allclose(torch.matmul(Amat, Xc), torch.matmul(Mmat, Xc) * ev.cpu().unsqueeze(-2), atol=1e-7)


def test_device_operators_never_reach_the_host_drivers(dev, monkeypatch):
    """Device dispatch (r06): an operator in HOST memory is served by xitorch_amd/linalg/host_*.py, an operator on the HIP
    device by the HIP kernels and by nothing else — the host drivers' call counters do not move during device calls, a
    host driver refuses a device operator, and a device call FAILS (no silent detour) when the native library is gone."""
    from xitorch_amd.linalg import host_eig, host_krylov, native_krylov as nk
    from xitorch_amd.optimize import native_root as nr
    from xitorch_amd import _capi
    g = torch.Generator().manual_seed(3)
    n = 96
    R = torch.rand(2, n, n, dtype=torch.float64, generator=g)
    S = (R + R.transpose(-2, -1)) * 0.05 + torch.eye(n, dtype=torch.float64)
    Bm = torch.rand(2, n, 2, dtype=torch.float64, generator=g)
    before = (dict(host_krylov.calls), dict(host_eig.calls))
    Ad = xa.LinearOperator.m(S.to(dev), True)
    davidson(Ad, 3, "lowest", min_eps=1e-8)
    for meth in ("cg", "bicgstab", "gmres"):
        getattr(nk, meth)(Ad, Bm.to(dev), rtol=1e-9)
    nr.broyden1(lambda y: torch.matmul(S.to(dev), y.unsqueeze(-1)).squeeze(-1) - 1.0, torch.zeros(2, n, dtype=torch.float64,
                                                                                                 device=dev), alpha=-1.0)
    assert (dict(host_krylov.calls), dict(host_eig.calls)) == before
    with pytest.raises(_capi.NativeLibraryError):
        host_eig.davidson(Ad, 3, "lowest")
    with pytest.raises(_capi.NativeLibraryError):
        host_krylov.cg(Ad, Bm.to(dev))
    # host memory -> host drivers (same answers)
    Ah = xa.LinearOperator.m(S, True)
    n_host = host_eig.calls["davidson"]           # (the refused call above counted itself before it raised)
    ev_h, _ = davidson(Ah, 3, "lowest", min_eps=1e-8)
    ev_d, _ = davidson(Ad, 3, "lowest", min_eps=1e-8)
    assert host_eig.calls["davidson"] == n_host + 1 and (ev_h - ev_d.cpu()).abs().max().item() < 1e-10
    # no library -> device calls raise, whatever host drivers exist
    def gone(*a, **k):
        raise _capi.NativeLibraryError("libxitorch_amd.so not found (simulated)")
    monkeypatch.setattr(_capi, "fn", gone)
    monkeypatch.setattr(K, "fn", gone)
    monkeypatch.setattr(nk, "fn", gone)
    with pytest.raises(_capi.NativeLibraryError):
        nk.cg(Ad, Bm.to(dev), rtol=1e-9)
    with pytest.raises(_capi.NativeLibraryError):
        davidson(Ad, 3, "lowest", min_eps=1e-8)


@pytest.mark.parametrize("B,N", [(4, 512), (5, 384)])
def test_two_group_pipeline_matches_single_group(dev, B, N):
    # overlap=True: two batch groups on two HIP streams; must be the same computation as one group
    from xitorch_amd import synthetic
    mat = synthetic.dense_symmetric(B, N, "S1").to(dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    t1, t2 = {}, {}
    e1, X1 = davidson(A, 4, "lowest", min_eps=1e-9, overlap=False, trace=t1)
    e2, X2 = davidson(A, 4, "lowest", min_eps=1e-9, overlap=True, trace=t2)
    torch.cuda.synchronize()
    assert t2["groups"] == 2 and t1["groups"] == 1
    assert t1["niter"] == t2["niter"] and t1["napply"] == t2["napply"]
    # same iteration; bitwise equality is not expected (the contraction splits depend on the launch's batch size
    # and K1s accumulates row sums with LDS float atomics), rounding-level agreement is
    assert torch.allclose(e1, e2, rtol=0, atol=1e-12) and torch.allclose(X1.abs(), X2.abs(), rtol=0, atol=1e-9)
    assert max(abs(a - b) for a, b in zip(t1["resid_history"], t2["resid_history"])) < 1e-9
    # the general (non symmetric-storage) kernel path too
    Ag = xa.MatrixLinearOperator(mat, True, symmetric_storage=False)
    e3, _ = davidson(Ag, 4, "uppest", min_eps=1e-9, overlap=True)
    e4, _ = davidson(Ag, 4, "uppest", min_eps=1e-9, overlap=False)
    assert torch.allclose(e3, e4, rtol=0, atol=1e-11)


def test_davidson_diagonal_preconditioner(dev):
    # (extension) Davidson's diagonal correction: same eigenpairs as the reference iteration, in far fewer
    # iterations on a diagonally dominant operator started from unit vectors (v_init="eye", symeig.py:241);
    # the default (precond=None) is the reference's t = -resid
    B, N, neig = 2, 600, 4
    g = torch.Generator().manual_seed(5)
    R = torch.randn(B, N, N, dtype=torch.float64, generator=g) * 0.02
    A = (R + R.transpose(-2, -1)) * 0.5 + torch.diag(torch.linspace(1.0, 600.0, N, dtype=torch.float64))
    lam_all = torch.linalg.eigvalsh(A)
    lam_ref = lam_all[:, :neig]
    Aop = xa.LinearOperator.m(A.to(dev), is_hermitian=True)
    t0, t1, t2, t3 = {}, {}, {}, {}
    kw = dict(min_eps=1e-9, v_init="eye")
    ev0, _ = davidson(Aop, neig, "lowest", trace=t0, **kw)
    ev1, X1 = davidson(Aop, neig, "lowest", trace=t1, precond="diag", **kw)
    ev2, _ = davidson(Aop, neig, "lowest", trace=t2, precond=A.diagonal(dim1=-2, dim2=-1).to(dev), **kw)
    assert t0["stop_reason"] == t1["stop_reason"] == t2["stop_reason"] == "converged"
    for ev in (ev0, ev1, ev2):
        assert (ev.cpu() - lam_ref).abs().max().item() < 1e-10 * 600
    assert 3 * t1["niter"] < t0["niter"] and t2["niter"] == t1["niter"], (t0["niter"], t1["niter"], t2["niter"])
    Xc = X1.cpu()
    assert (torch.matmul(A, Xc) - Xc * ev1.cpu().unsqueeze(-2)).abs().max().item() < 1e-8
    # a LinearOperator preconditioner (fixed-shift inverse diagonal) is applied through its own .mm
    Kop = xa.LinearOperator.m(torch.diag_embed(1.0 / (A.diagonal(dim1=-2, dim2=-1) - 0.5)).to(dev), is_hermitian=True)
    ev3, _ = davidson(Aop, neig, "lowest", trace=t3, precond=Kop, **kw)
    assert (ev3.cpu() - lam_ref).abs().max().item() < 1e-10 * 600 and t3["niter"] < t0["niter"], t3["niter"]
    # uppermost pairs and the generalised problem go through the same correction (random start: correctness only)
    Md = torch.linspace(1.0, 2.0, N, dtype=torch.float64)
    Mop = xa.LinearOperator.m(torch.diag_embed(Md).expand(B, N, N).contiguous().to(dev), is_hermitian=True)
    evu, _ = davidson(Aop, neig, "uppest", min_eps=1e-8, precond="diag")
    assert (evu.cpu() - lam_all[:, -neig:]).abs().max().item() < 1e-10 * 600
    evm, _ = davidson(Aop, neig, "lowest", M=Mop, precond="diag", **kw)
    Li = torch.diag_embed(Md ** -0.5)
    ref_m = torch.linalg.eigvalsh(Li @ A @ Li)[:, :neig]
    assert (evm.cpu() - ref_m).abs().max().item() < 1e-9 * 600
    with pytest.raises(RuntimeError):
        davidson(Aop, neig, "lowest", precond="nope")
    # through the front-end
    evf, _ = xa.linalg.symeig(Aop, neig=neig, method="davidson", precond="diag", **kw)
    assert (evf.cpu() - lam_ref).abs().max().item() < 1e-10 * 600


@pytest.mark.parametrize("B,N,neig,mode", [(3, 333, 3, "lowest"), (2, 130, 5, "uppest"), (1, 77, 1, "lowest")])
def test_davidson_ragged_sizes_vs_oracle(dev, B, N, neig, mode):
    # N not a multiple of the vector width / pad unit: padded pitch, general panel kernel; same iterates as the oracle
    g = torch.Generator().manual_seed(N)
    R = torch.randn(B, N, N, dtype=torch.float64, generator=g)
    mat = (R + R.transpose(-2, -1)) * 0.05 + torch.diag(torch.linspace(0.0, 30.0, N, dtype=torch.float64))
    tr, tr_o = {}, {}
    ev, X = davidson(xa.LinearOperator.m(mat.to(dev), True), neig, mode, min_eps=1e-8, trace=tr)
    ev_o, X_o = osym.davidson(oops.DenseOp(mat, True), neig, mode, None, min_eps=1e-8, trace=tr_o)
    assert (ev.cpu() - ev_o).abs().max().item() < 1e-10 * 30
    assert abs(tr["niter"] - tr_o["niter"]) <= 2
    Xc = X.cpu()
    assert (torch.matmul(mat, Xc) - Xc * ev.cpu().unsqueeze(-2)).abs().max().item() < 1e-7
    exact = torch.linalg.eigvalsh(mat)
    exact = exact[:, :neig] if mode == "lowest" else exact[:, -neig:]
    assert (ev.cpu() - exact).abs().max().item() < 1e-10 * 30


def test_davidson_full_basis_stop_and_float32(dev):
    # (1) an unreachable tolerance: the basis grows until it is square, then the exact pairs are returned
    #     (stop rule `k == N`, symeig.py:196-203)
    g = torch.Generator().manual_seed(4)
    N = 20
    R = torch.randn(2, N, N, dtype=torch.float64, generator=g)
    mat = (R + R.transpose(-2, -1)) * 0.5
    tr = {}
    ev, X = davidson(xa.LinearOperator.m(mat.to(dev), True), 3, "lowest", min_eps=1e-300, trace=tr)
    assert tr["stop_reason"] == "full_basis" and tr["basis_size"] == N
    assert (ev.cpu() - torch.linalg.eigvalsh(mat)[:, :3]).abs().max().item() < 1e-10
    # (2) float32 operator: fp32 kernels end to end, eigenvalues to fp32 accuracy
    mat32 = (mat * 0.1 + torch.diag(torch.arange(N, dtype=torch.float64))).float()
    mat32 = (mat32 + mat32.transpose(-2, -1)) * 0.5
    ev32, X32 = davidson(xa.LinearOperator.m(mat32.to(dev), True), 2, "uppest", min_eps=1e-4)
    assert ev32.dtype == torch.float32
    assert (ev32.cpu().double() - torch.linalg.eigvalsh(mat32.double())[:, -2:]).abs().max().item() < 2e-4
    # (3) batch dims of rank 2 and neig == nguess == 1
    mat4 = mat.reshape(2, 1, N, N).expand(2, 2, N, N).contiguous()
    ev4, X4 = davidson(xa.LinearOperator.m(mat4.to(dev), True), 1, "lowest", min_eps=1e-9)
    assert list(ev4.shape) == [2, 2, 1] and list(X4.shape) == [2, 2, N, 1]
    assert (ev4.cpu()[:, 0, 0] - torch.linalg.eigvalsh(mat)[:, 0]).abs().max().item() < 1e-10


def test_two_group_pipeline_with_preconditioner_and_odd_batch(dev):
    # the diagonal of the operator is sliced per group; an odd batch splits 2 + 3
    B, N, neig = 5, 400, 3
    g = torch.Generator().manual_seed(9)
    R = torch.randn(B, N, N, dtype=torch.float64, generator=g) * 0.02
    A = (R + R.transpose(-2, -1)) * 0.5 + torch.diag(torch.linspace(1.0, 400.0, N, dtype=torch.float64))
    Aop = xa.LinearOperator.m(A.to(dev), is_hermitian=True)
    t1, t2 = {}, {}
    kw = dict(min_eps=1e-9, v_init="eye", precond="diag")
    ev1, X1 = davidson(Aop, neig, "lowest", overlap=False, trace=t1, **kw)
    ev2, X2 = davidson(Aop, neig, "lowest", overlap=True, trace=t2, **kw)
    assert t1["groups"] == 1 and t2["groups"] == 2 and t1["niter"] == t2["niter"]
    assert torch.allclose(ev1, ev2, rtol=0, atol=1e-11 * 400)
    ref = torch.linalg.eigvalsh(A)[:, :neig]
    assert (ev2.cpu() - ref).abs().max().item() < 1e-10 * 400
    Xc = X2.cpu()
    assert (torch.matmul(A, Xc) - Xc * ev2.cpu().unsqueeze(-2)).abs().max().item() < 1e-7


def test_two_group_pipeline_full_basis_exact_pairs(dev):
    # N == nguess + m * neig: with the two-group pipeline the FIRST group is expanded early inside the iteration, so
    # the "basis became square" test must look at a group that has not been expanded yet — otherwise the loop stops
    # one Rayleigh-Ritz short of the exact pairs (symeig.py:196-203 breaks only after the RR on the square basis)
    g = torch.Generator().manual_seed(14)
    N, neig, B = 24, 3, 4
    R = torch.randn(B, N, N, dtype=torch.float64, generator=g)
    mat = (R + R.transpose(-2, -1)) * 0.5
    exact = torch.linalg.eigvalsh(mat)[:, :neig]
    for overlap in (False, True):
        tr = {}
        ev, X = davidson(xa.LinearOperator.m(mat.to(dev), True), neig, "lowest", min_eps=1e-300, overlap=overlap,
                         trace=tr)
        assert tr["stop_reason"] == "full_basis" and tr["basis_size"] == N and tr["groups"] == (2 if overlap else 1)
        assert (ev.cpu() - exact).abs().max().item() < 1e-10, (overlap, (ev.cpu() - exact).abs().max().item())
        Xc = X.cpu()
        assert (torch.matmul(mat, Xc) - Xc * ev.cpu().unsqueeze(-2)).abs().max().item() < 1e-9


def test_rank_deficient_start_block_raises(dev):
    # the reference raises from torch.linalg.cholesky on a rank-deficient guess block (tallqr, tensor.py:16);
    # here the panel Cholesky flags the non-positive pivot and the host raises
    g = torch.Generator().manual_seed(15)
    N = 64
    R = torch.randn(2, N, N, dtype=torch.float64, generator=g)
    mat = ((R + R.transpose(-2, -1)) * 0.5).to(dev)
    V0 = torch.randn(2, N, 3, dtype=torch.float64, generator=g)
    V0[1, :, 2] = 0.0                                    # member 1: a zero column -> pivot exactly 0
    with pytest.raises(RuntimeError, match="positive definite"):
        davidson(xa.LinearOperator.m(mat, True), 3, "lowest", V0=V0.to(dev), max_niter=5)
    # the flag is sticky: a later factorisation of a healthy panel into the same info buffer (the second CholeskyQR
    # pass of `start`) must not erase it
    from xitorch_amd import kernels as K
    Gbad = torch.eye(3, dtype=torch.float64).repeat(2, 1, 1)
    Gbad[1, 1, 1] = -1.0
    W = torch.empty(2, 3, 3, dtype=torch.float64, device=dev)
    info = torch.zeros(2, dtype=torch.int32, device=dev)
    K.panel_chol(Gbad.to(dev), W, info, 3)
    assert info.tolist() == [0, 2]
    K.panel_chol(torch.eye(3, dtype=torch.float64).repeat(2, 1, 1).to(dev), W, info, 3)
    assert info.tolist() == [0, 2]
    assert torch.allclose(W.cpu(), torch.eye(3, dtype=torch.float64).repeat(2, 1, 1))


@pytest.mark.parametrize("withM", [False, True])
def test_panel_orthogonalisation_vs_reference_tallqr(dev, withM):
    """Row a5 directly: block Gram-Schmidt(x2) + panel CholeskyQR (native) against the reference's full CholeskyQR
    `tallqr(cat(V, t))` (oracle.symeig.tallqr == xitorch/_utils/tensor.py:8-19): same Q column by column up to
    rounding, orthonormal to working precision; then a NEARLY dependent residual block, where the reference's one-shot
    CholeskyQR of the whole basis loses orthogonality like cond^2 * eps — the native path must be no worse."""
    from xitorch_amd.linalg.native_eig import tallqr_extend
    g = torch.Generator().manual_seed(41)
    B, N, k, p = 3, 500, 18, 6
    Mmat = None
    if withM:
        R = torch.rand(B, N, N, dtype=torch.float64, generator=g)
        Mmat = 0.02 * (R + R.transpose(-2, -1)) + torch.eye(N, dtype=torch.float64)
    V0 = torch.randn(B, N, k, dtype=torch.float64, generator=g)
    V, _ = osym.tallqr(V0, torch.matmul(Mmat, V0) if withM else None)            # (M-)orthonormal basis
    t = torch.randn(B, N, p, dtype=torch.float64, generator=g) + 0.5 * V[..., :p]  # a block with components along V
    full = torch.cat([V, t], dim=-1)
    Qref, _ = osym.tallqr(full, torch.matmul(Mmat, full) if withM else None)
    Mop = xa.LinearOperator.m(Mmat.to(dev), True) if withM else None
    Q = tallqr_extend(V.to(dev), t.to(dev), M=Mop).cpu()
    assert torch.equal(Q[..., :k], V)                                             # the old basis is untouched
    assert (Q[..., k:] - Qref[..., k:]).abs().max().item() < 1e-11                # same vectors, same signs
    MQ = torch.matmul(Mmat, Q) if withM else Q
    eye = torch.eye(k + p, dtype=torch.float64)
    assert (torch.matmul(Q.transpose(-2, -1), MQ) - eye).abs().max().item() < 1e-13
    # nearly dependent block: column 1 = column 0 + 1e-6 * noise
    t2 = t.clone()
    t2[..., 1] = t2[..., 0] + 1e-6 * torch.randn(B, N, dtype=torch.float64, generator=g)
    full2 = torch.cat([V, t2], dim=-1)
    Qr2, _ = osym.tallqr(full2, torch.matmul(Mmat, full2) if withM else None)
    Q2 = tallqr_extend(V.to(dev), t2.to(dev), M=Mop).cpu()
    MQ2 = torch.matmul(Mmat, Q2) if withM else Q2
    MQr2 = torch.matmul(Mmat, Qr2) if withM else Qr2
    loss_native = (torch.matmul(Q2.transpose(-2, -1), MQ2) - eye).abs().max().item()
    loss_ref = (torch.matmul(Qr2.transpose(-2, -1), MQr2) - eye).abs().max().item()
    assert loss_native <= max(10.0 * loss_ref, 1e-12), (loss_native, loss_ref)
    # both span the same space: projector difference at the level of the lost orthogonality
    P1 = torch.matmul(Q2, MQ2.transpose(-2, -1))
    P2 = torch.matmul(Qr2, MQr2.transpose(-2, -1))
    assert (P1 - P2).abs().max().item() < 1e3 * max(loss_ref, loss_native, 1e-13)
    # an exactly dependent block is refused, like torch.linalg.cholesky does in the reference
    t3 = t.clone()
    t3[..., 2] = 0.0
    with pytest.raises(RuntimeError):
        tallqr_extend(V.to(dev), t3.to(dev), M=Mop)


@pytest.mark.parametrize("mode", ["lowest", "uppest"])
@pytest.mark.parametrize("overlap", [False, True])
def test_thick_restart_bounded_basis(dev, mode, overlap):
    """(extension, opt-in) restart=k_max: the basis never exceeds k_max vectors, the Rayleigh-Ritz matrix stays inside
    the LDS-resident eigensolver, and the converged pairs are those of the unrestarted (= reference) iteration:
    eigenvalues against the closed-form spectrum, residual below min_eps, orthonormal vectors."""
    from xitorch_amd import synthetic
    B, N, p = 4, 768, 4
    mat = synthetic.dense_symmetric(B, N, "S2")                       # sqrt(i+1): slow, 50+ reference iterations
    exact = synthetic.spectrum("S2", N)
    exact = exact[:p] if mode == "lowest" else exact[-p:]
    A = xa.LinearOperator.m(mat.to(dev), is_hermitian=True)
    t0, t1 = {}, {}
    ev0, _ = davidson(A, p, mode, min_eps=1e-8, trace=t0, overlap=False)
    ev1, X1 = davidson(A, p, mode, min_eps=1e-8, restart=6 * p, trace=t1, overlap=overlap, max_niter=2000)
    assert t1["stop_reason"] == "converged" and t1["restarts"] > 0 and t1["basis_size"] <= 6 * p
    assert t0["restarts"] == 0 and t0["basis_size"] > 6 * p            # the default never restarts
    assert (ev1.cpu() - exact).abs().max().item() < 1e-9 and (ev0.cpu() - exact).abs().max().item() < 1e-9
    Xc = X1.cpu()
    assert (torch.matmul(mat, Xc) - Xc * ev1.cpu().unsqueeze(-2)).abs().max().item() < 1e-8
    G = torch.matmul(Xc.transpose(-2, -1), Xc)
    assert (G - torch.eye(p, dtype=torch.float64)).abs().max().item() < 1e-9
    with pytest.raises(ValueError):
        davidson(A, p, mode, restart=2 * p)


def test_thick_restart_generalised_problem(dev):
    case = [c for c in cases.DAVIDSON_CASES if c["name"] == "genM_120_b2_lowest3"][0]
    mat, Mmat = cases.davidson_matrix(case), cases.davidson_M(case)
    gold = np.load(os.path.join(GOLD, "davidson_%s.npz" % case["name"]))
    A = xa.LinearOperator.m(mat.to(dev), True)
    Mop = xa.LinearOperator.m(Mmat.to(dev), True)
    tr = {}
    ev, X = davidson(A, 3, "lowest", M=Mop, min_eps=1e-8, restart=15, trace=tr, max_niter=2000)
    assert tr["restarts"] > 0 and tr["stop_reason"] == "converged"
    assert np.abs(ev.cpu().numpy() - gold["evals_exact"]).max() < 1e-9
    Xc = X.cpu()
    assert (torch.matmul(mat, Xc) - torch.matmul(Mmat, Xc) * ev.cpu().unsqueeze(-2)).abs().max().item() < 1e-7


def test_chain_calls_match_kernel_by_kernel(dev):
    """The one-C-call chain stages (xk_davidson_ritz / _orth / _extend_t, fused CholeskyQR of the panel) against the
    kernel-by-kernel host loop: same iteration count, eigenvalues and residual history to rounding."""
    from xitorch_amd import synthetic
    for (B, N, p, dtype, tol) in ((3, 1536, 6, torch.float64, 1e-12), (2, 1000, 4, torch.float64, 1e-12),
                                  (2, 2048, 6, torch.float32, 2e-5), (2, 900, 10, torch.float64, 1e-12)):
        mat = synthetic.dense_symmetric(B, N, "S1", dtype=dtype, device=dev)
        A = xa.LinearOperator.m(mat, is_hermitian=True)
        eps = 1e-8 if dtype == torch.float64 else 1e-3
        # one projection pass + CholeskyQR, and the re-orthogonalised order ([projection, CholeskyQR] twice, the first
        # CholeskyQR shifted), which the host loop mirrors with its own Gram shift
        for extra in ({"orth_passes": 1}, {"orth_passes": 2}):
            if p > 8:
                # (10 columns: converged pairs' noise residuals decide the path — one pass breaks down after ~30
                #  iterations, see test_default_orthonormalisation_survives_mixed_convergence, and with two any
                #  rounding-level difference between the two hosts' shift arithmetic, a kernel there and torch here,
                #  diverges within 16 iterations: the wide-panel kernels are compared over the first 14)
                extra = dict(extra, max_niter=14)
            out = {}
            for chain in ("calls", "kernels"):
                tr = {}
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    ev, X = davidson(A, p, "lowest", min_eps=eps, chain=chain, trace=tr, **extra)
                out[chain] = (ev.double().cpu(), tr)
            (e1, t1), (e2, t2) = out["calls"], out["kernels"]
            # (the shifted pass adds its shift in torch on one side and in the kernel on the other: the rounding-level
            #  difference may move a stopping test that sits at the threshold by one iteration; the residual histories
            #  must agree over the common part)
            slack = 1 if extra.get("orth_passes") == 2 else 0
            assert abs(t1["niter"] - t2["niter"]) <= slack, (extra, t1["niter"], t2["niter"])
            assert (e1 - e2).abs().max().item() <= tol * max(1.0, e2.abs().max().item())
            m = min(len(t1["resid_history"]), len(t2["resid_history"]))
            h1, h2 = np.array(t1["resid_history"][:m]), np.array(t2["resid_history"][:m])
            assert np.all(np.abs(h1 - h2) <= 1e-3 * np.abs(h2) + 10 * tol), (extra, np.abs(h1 - h2).max(), h1[-3:], h2[-3:])


@pytest.mark.parametrize("q,dtype", [(6, torch.float64), (20, torch.float64), (6, torch.float32)])
def test_reorthogonalised_passes_survive_nearly_dependent_panel(dev, q, dtype):
    """xk_davidson_orth with two passes = [projection, shifted CholeskyQR, projection, CholeskyQR]: a panel whose Gram
    matrix spans ~14 decades (what 20 residuals close to convergence look like: DESIGN 4) comes out orthonormal and
    orthogonal to the basis; the plain order [projection, projection, CholeskyQR] lost 1e-6 .. 1e-2 there.  fp32: 6
    decades."""
    from xitorch_amd.linalg._panel import pad_len
    B, N, k0 = 2, 1536, 40
    g = torch.Generator().manual_seed(q)
    ld = pad_len(N)
    Q0, _ = torch.linalg.qr(torch.randn(B, N, k0 + q, dtype=torch.float64, generator=g))
    basis, dirs = Q0[:, :, :k0], Q0[:, :, k0:]
    decades = 14.0 if dtype == torch.float64 else 6.0
    sv = torch.logspace(0, -decades / 2, q, dtype=torch.float64)               # singular values of the panel
    mix, _ = torch.linalg.qr(torch.randn(B, q, q, dtype=torch.float64, generator=g))
    panel = (dirs * sv) @ mix.transpose(-2, -1) + 1e-3 * basis @ torch.randn(B, k0, q, dtype=torch.float64, generator=g)
    V = torch.zeros(B, k0 + q, ld, dtype=dtype)
    V[:, :k0, :N] = basis.transpose(-2, -1).to(dtype)
    V[:, k0:, :N] = panel.transpose(-2, -1).to(dtype)
    Vd = V.to(dev)
    C = torch.empty(B * max(q, 8) * (k0 + q + 8), dtype=dtype, device=dev)
    W = torch.empty(B * q * q, dtype=dtype, device=dev)
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    K.davidson_orth(Vd, N, k0, q, C, W, info, passes=2)
    assert int(info.max()) == 0
    Qn = Vd[:, :, :N].double().cpu()
    if dtype == torch.float64 and q <= 32:
        # the same sequence restated in torch: [projection, CholeskyQR of G + shift, projection, CholeskyQR]
        Vb, t = V[:, :k0, :N].double(), V[:, k0:, :N].double().clone()
        sh = min(1e-3, 11.0 * (N * q + q * (q + 1)) * 1.1102230246251565e-16)
        for it in range(2):
            t = t - (t @ Vb.transpose(1, 2)) @ Vb
            Gt = t @ t.transpose(1, 2)
            if it == 0:
                Gt = Gt + sh * torch.diagonal(Gt, dim1=-2, dim2=-1).sum(-1)[:, None, None] * torch.eye(q, dtype=torch.float64)
            Rt = torch.linalg.cholesky(Gt, upper=True)
            t = torch.linalg.solve_triangular(Rt.transpose(1, 2), t, upper=False)
        assert (Qn[:, k0:] - t).abs().max().item() < 1e-7          # (cond ~1e7 panel: rounding amplified by it)
    G = Qn @ Qn.transpose(-2, -1)
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    assert (G - torch.eye(k0 + q, dtype=torch.float64)).abs().max().item() < tol
    # same span as the input panel (right-multiplications by triangular matrices only)
    Pn = Qn[:, k0:]
    resid = dirs.transpose(-2, -1) - (dirs.transpose(-2, -1) @ Pn.transpose(-2, -1)) @ Pn
    assert resid.abs().max().item() < (1e-4 if dtype == torch.float64 else 2e-1)


def test_fused_panel_cholqr_vs_separate_kernels(dev):
    # xk_davidson_orth's one-kernel CholeskyQR (q <= 8) and its wide-panel path against Gram + xk_panel_chol +
    # xk_panel_transform; orthonormality of the result; the sticky breakdown flag on a rank-deficient panel
    from xitorch_amd.linalg._panel import pad_len
    for (B, N, k0, q, dtype, tol) in ((3, 1000, 12, 6, torch.float64, 1e-13), (2, 2048, 0, 8, torch.float64, 1e-13),
                                      (2, 1536, 20, 12, torch.float64, 1e-13), (2, 777, 5, 3, torch.float32, 1e-5)):
        g = torch.Generator().manual_seed(N + q)
        ld = pad_len(N)
        V = torch.zeros(B, k0 + q, ld, dtype=dtype)
        Q0, _ = torch.linalg.qr(torch.randn(B, N, k0 + q, dtype=torch.float64, generator=g))
        V[:, :k0, :N] = Q0[:, :, :k0].transpose(-2, -1).to(dtype)
        V[:, k0:, :N] = torch.randn(B, q, N, dtype=torch.float64, generator=g).to(dtype)
        Vd = V.to(dev)
        C = torch.empty(B * max(q, 8) * (k0 + q + 8), dtype=dtype, device=dev)
        W = torch.empty(B * q * q, dtype=dtype, device=dev)
        info = torch.zeros(B, dtype=torch.int32, device=dev)
        K.davidson_orth(Vd, N, k0, q, C, W, info, passes=2)
        Qn = Vd[:, :, :N].double().cpu()
        G = Qn @ Qn.transpose(-2, -1)
        assert (G - torch.eye(k0 + q, dtype=torch.float64)).abs().max().item() < 50 * tol
        assert int(info.max()) == 0
        # the new rows span the same space as the input block projected off the basis
        t = V[:, k0:, :N].double()
        tp = t - (t @ Qn[:, :k0].transpose(-2, -1)) @ Qn[:, :k0]
        resid = tp - (tp @ Qn[:, k0:].transpose(-2, -1)) @ Qn[:, k0:]
        assert resid.abs().max().item() < 1e3 * tol * t.abs().max().item()
    # rank-deficient panel: flagged, not silently "orthonormalised"
    V = torch.zeros(1, 4, 512, dtype=torch.float64)
    V[0, :2] = torch.randn(2, 512, dtype=torch.float64)
    V[0, 2] = 0.0                               # a zero vector: the pivot is exactly 0
    V[0, 3] = torch.randn(512, dtype=torch.float64)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    K.davidson_orth(V.to(dev), 512, 0, 4, torch.empty(1024, dtype=torch.float64, device=dev),
                    torch.empty(16, dtype=torch.float64, device=dev), info, passes=0)
    assert int(info[0]) != 0


def _separated_ends_operator(dev, N=1536):
    """48 separated eigenvalues at each end of the spectrum (so that 40 wanted pairs converge long before the basis is
    square), the rest in a band in between; dense through a Householder similarity, symmetrised exactly."""
    d = torch.cat([torch.arange(1.0, 49.0, dtype=torch.float64),
                   200.0 + 100.0 * torch.arange(N - 96, dtype=torch.float64) / (N - 96),
                   400.0 + torch.arange(1.0, 49.0, dtype=torch.float64)]).to(dev)
    mats = []
    for b in range(2):
        w = torch.sin(0.37 * torch.arange(1, N + 1, dtype=torch.float64, device=dev) + 0.11 * b) + 1.5
        w = w / w.norm()
        Dw = d * w
        m = torch.diag(d) - 2.0 * torch.outer(w, Dw) - 2.0 * torch.outer(Dw, w) + 4.0 * (w @ Dw) * torch.outer(w, w)
        mats.append((m + m.T) * 0.5)
    mat = torch.stack(mats)
    return mat, torch.linalg.eigvalsh(mat)


@pytest.mark.parametrize("neig,nguess,mode,restart", [(40, None, "lowest", None), (36, 40, "uppest", None),
                                                      (20, None, "lowest", 100), (10, None, "lowest", 40), (16, None, "lowest", None)])
def test_wide_blocks_and_restart_beyond_16(dev, neig, nguess, mode, restart, monkeypatch):
    """No width cliff: neig / nguess > 32 go through the chunked panel orthonormalisation
    (the reference has no limit: symeig.py:100-140); thick restart with 16 < neig <= 32 keeps at least the wanted
    vectors; more than 16 wanted pairs (up to 64) stay on the native Rayleigh-Ritz solver (K3g) — no library eigh as long
    as the basis holds at most 768 vectors.  Against the dense eigendecomposition."""
    mat, lam_all = _separated_ends_operator(dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    tr = {}
    calls = []
    real_eigh = torch.linalg.eigh
    monkeypatch.setattr(torch.linalg, "eigh", lambda *a, **k: (calls.append(1), real_eigh(*a, **k))[1])
    ev, X = davidson(A, neig, mode, min_eps=1e-8, nguess=nguess, restart=restart, trace=tr)
    monkeypatch.undo()
    if tr["basis_size"] <= 768 and (restart is None or 2 * neig <= 64):
        assert calls == [] and tr["k3_fallbacks"] == 0, (len(calls), tr["k3_fallbacks"], tr["basis_size"])
    assert tr["stop_reason"] == "converged", tr["stop_reason"]
    assert ev.shape == (2, neig) and X.shape == (2, 1536, neig)
    want = lam_all[:, :neig] if mode == "lowest" else lam_all[:, -neig:]
    assert (ev - want).abs().max().item() <= 1e-9 * lam_all.abs().max().item()
    assert (mat @ X - X * ev.unsqueeze(-2)).abs().max().item() <= 1e-7
    G = X.transpose(-2, -1) @ X
    assert (G - torch.eye(neig, dtype=torch.float64, device=dev)).abs().max().item() <= 1e-9
    if restart is not None:
        assert tr["restarts"] >= 1


def test_one_gram_schmidt_pass_equals_two(dev):
    """On short, well-conditioned runs ONE projection pass of the new residual block (orth_passes=1, opt-in since the
    wide-block failure below) gives the iteration of the default two: same count, same eigenvalues, same residual
    history to rounding, and an orthonormal result — on the clustered and on the slowly converging spectrum, fp64 and
    fp32."""
    from xitorch_amd import synthetic
    for (kind, B, N, p, dtype, eps, tol) in (("S1", 2, 2048, 6, torch.float64, 1e-8, 1e-11),
                                             ("S2", 2, 1024, 4, torch.float64, 1e-8, 1e-11),
                                             ("S3", 2, 1024, 4, torch.float64, 1e-8, 1e-11),     # > 100 iterations
                                             ("S1", 2, 2048, 6, torch.float32, 1e-3, 2e-5)):
        mat = synthetic.dense_symmetric(B, N, kind, dtype=dtype, device=dev)
        A = xa.LinearOperator.m(mat, is_hermitian=True)
        res = {}
        for passes in (1, 2):
            tr = {}
            ev, X = davidson(A, p, "lowest", min_eps=eps, orth_passes=passes, trace=tr)
            res[passes] = (ev.double().cpu(), X.double().cpu(), tr)
        (e1, X1, t1), (e2, X2, t2) = res[1], res[2]
        assert abs(t1["niter"] - t2["niter"]) <= (0 if dtype == torch.float64 else 1), (kind, t1["niter"], t2["niter"])
        assert (e1 - e2).abs().max().item() <= tol * max(1.0, e2.abs().max().item())
        G = X1.transpose(-2, -1) @ X1
        assert (G - torch.eye(p, dtype=torch.float64)).abs().max().item() <= (1e-10 if dtype == torch.float64 else 1e-4)


def test_unrestarted_run_beyond_128_vectors_stays_native(dev, monkeypatch):
    """The reference-default (un-restarted) iteration on a slowly converging spectrum grows its basis far beyond the
    128 vectors of the LDS-resident eigensolvers (symeig.py:132-135, 174-175); from 129 to 768 vectors the
    Rayleigh-Ritz step runs on K3g — no library eigh on the way — and reproduces the closed-form eigenvalues."""
    from xitorch_amd import synthetic
    B, N, p = 2, 2048, 6
    mat = synthetic.dense_symmetric(B, N, "S2", dtype=torch.float64, device=dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    calls = []
    real_eigh = torch.linalg.eigh
    monkeypatch.setattr(torch.linalg, "eigh", lambda *a, **k: (calls.append(1), real_eigh(*a, **k))[1])
    tr = {}
    ev, X = davidson(A, p, "lowest", min_eps=1e-8, trace=tr)
    # (S2 at N = 2048: 53 iterations in the reference's probe, SURVEY 8d -> a basis of ~320 <= 448, K3g's range at B = 2)
    assert tr["stop_reason"] == "converged" and 128 < tr["basis_size"] <= 448, tr["basis_size"]
    assert not calls and tr["k3_fallbacks"] == 0
    exact = synthetic.spectrum("S2", N, device=dev)[:p]
    assert (ev - exact).abs().max().item() <= 1e-10
    assert (mat @ X - X * ev.unsqueeze(-2)).abs().max().item() <= 1e-7


def test_unrestarted_run_to_900_vectors_makes_no_library_eigh_call(dev, monkeypatch):
    """(r05) The reference's basis is unbounded until k == N (symeig.py:174,202).  An un-restarted run that
    is stopped by max_niter at a basis of 900+ vectors (fp64: beyond the 768 of r04's native range) makes no
    torch.linalg.eigh call, and its best iterate is a valid Ritz block: eigenvalues above the exact ones (Cauchy
    interlacing), decreasing residual."""
    from xitorch_amd import synthetic
    import warnings
    B, N, p = 2, 4096, 6
    mat = synthetic.dense_symmetric(B, N, "S3", dtype=torch.float64, device=dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    calls = []
    real_eigh = torch.linalg.eigh
    monkeypatch.setattr(torch.linalg, "eigh", lambda *a, **k: (calls.append(1), real_eigh(*a, **k))[1])
    tr = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, X = davidson(A, p, "lowest", min_eps=1e-12, max_niter=152, trace=tr)
    monkeypatch.undo()
    assert tr["basis_size"] >= 900, tr["basis_size"]
    assert not calls and tr["k3_fallbacks"] == 0, (len(calls), tr["k3_fallbacks"])
    exact = synthetic.spectrum("S3", N, device=dev)[:p]
    # (a 912-dimensional block Krylov space of a spectrum of width 4096 with unit gaps: far from converged, by design)
    assert (ev - exact).min().item() >= -1e-9 and (ev - exact).abs().max().item() <= 0.5
    G = X.transpose(-2, -1) @ X
    assert (G - torch.eye(p, dtype=torch.float64, device=dev)).abs().max().item() <= 1e-9
    hist = tr["resid_history"]
    assert hist[-1] < hist[5]
    R = mat @ X - X * ev.unsqueeze(-2)
    assert abs(R.abs().max().item() - tr["best_resid"]) <= 1e-6 * max(1.0, tr["best_resid"])


def test_unrestarted_run_to_1500_vectors_makes_no_library_eigh_call(dev, monkeypatch):
    """(r06) the same beyond 1024 vectors: K3g's one-launch-per-step form with 24 column slots serves the
    orders 1025 .. 1536 for 16 and more matrices per batch group (for fewer the library is faster there and stays:
    native_eig.K3G_MAX_K); a run of 16 operators stopped by max_niter at a basis of 1500+ vectors makes no
    torch.linalg.eigh call."""
    from xitorch_amd import synthetic
    import warnings
    B, N, p = 16, 4096, 8                     # (8 pairs: 6 reach 1e-12 at 1338 vectors, before the basis gets there)
    mat = synthetic.dense_symmetric(B, N, "S3", dtype=torch.float64, device=dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    calls = []
    real_eigh = torch.linalg.eigh
    monkeypatch.setattr(torch.linalg, "eigh", lambda *a, **k: (calls.append(1), real_eigh(*a, **k))[1])
    tr = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ev, X = davidson(A, p, "lowest", min_eps=1e-12, max_niter=190, trace=tr)
    monkeypatch.undo()
    assert tr["basis_size"] >= 1500, tr["basis_size"]
    assert not calls and tr["k3_fallbacks"] == 0, (len(calls), tr["k3_fallbacks"])
    exact = synthetic.spectrum("S3", N, device=dev)[:p]
    assert (ev - exact).min().item() >= -1e-9 and (ev - exact).abs().max().item() <= 0.5
    G = X.transpose(-2, -1) @ X
    assert (G - torch.eye(p, dtype=torch.float64, device=dev)).abs().max().item() <= 1e-9
    R = mat @ X - X * ev.unsqueeze(-2)
    assert abs(R.abs().max().item() - tr["best_resid"]) <= 1e-6 * max(1.0, tr["best_resid"])


@pytest.mark.parametrize("N,p", [(900, 8), (900, 10), (2048, 12)])
def test_default_orthonormalisation_survives_mixed_convergence(dev, N, p):
    """Regression (round 3): with some wanted pairs long converged and others (inside the dense part of the S1
    spectrum) still far, the residual block's condition number reaches 1e7 and more; ONE projection + CholeskyQR then
    lost the basis' orthogonality after ~30 iterations and the run stopped on DUPLICATED eigenpairs (eigenvalue error
    50, residual < min_eps).  The default ([projection, shifted CholeskyQR] + [projection, CholeskyQR]) must return
    the exact lowest pairs, orthonormal, in about the reference's number of iterations (oracle on the host)."""
    from xitorch_amd import synthetic
    from oracle import ops as oops, symeig as osym
    B = 2
    mat = synthetic.dense_symmetric(B, N, "S1", dtype=torch.float64, device=dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    exact = torch.linalg.eigvalsh(mat)[:, :p]
    tr = {}
    ev, X = davidson(A, p, "lowest", min_eps=1e-8, trace=tr)
    assert tr["stop_reason"] == "converged"
    assert tr["orth_adaptive"]
    if p <= 8:       # blocks of up to 8 start on one pass and must have been moved to two by the condition estimate
        assert all(t is not None and t < tr["niter"] - 5 for t in tr["orth_two_pass_from"]), tr["orth_two_pass_from"]
    assert (ev - exact).abs().max().item() <= 1e-10 * exact.abs().max().item()
    G = X.transpose(1, 2) @ X
    assert (G - torch.eye(p, dtype=torch.float64, device=dev)).abs().max().item() <= 1e-10
    assert (mat @ X - X * ev[:, None, :]).abs().max().item() <= 1e-7
    if N <= 900:
        tro = {}
        osym.davidson(oops.DenseOp(mat.cpu(), True), p, "lowest", min_eps=1e-8, trace=tro)
        assert abs(tr["niter"] - tro["niter"]) <= max(3, tro["niter"] // 10), (tr["niter"], tro["niter"])


@pytest.mark.parametrize("B,P,dtype", [(8, 16, torch.float32), (3, 32, torch.float64), (5, 9, torch.float64),
                                       (2, 1, torch.float64), (64, 12, torch.float32), (2, 31, torch.float32)])
def test_panel_chol_wave_per_member_matches_the_scalar_algorithm(dev, B, P, dtype):
    """(r05) xk_panel_chol is one wave per batch member with R in LDS (it was one thread per member with a scratch-memory
    array: 0.5 ms per call on the configs[4] chain).  Every entry is formed by the operations of the column-by-column
    scalar algorithm in the same order (tensor.py:15-17's cholesky + inverse restricted to the panel's Gram matrix):
    bit-identical to a scalar restatement, equal to torch's factor to rounding, first non-positive pivot flagged."""
    g = torch.Generator().manual_seed(100 * P + B)
    Q = torch.randn(B, P, P + 3, dtype=torch.float64, generator=g)
    G = (Q @ Q.transpose(-2, -1) + 0.1 * torch.eye(P, dtype=torch.float64)).to(dtype)
    G = G + 1e-3 * torch.randn(B, P, P, dtype=torch.float64, generator=g).to(dtype)       # not exactly symmetric
    W = torch.empty(B, P, P, dtype=dtype, device=dev)
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    K.panel_chol(G.to(dev), W, info, P)
    assert int(info.abs().max()) == 0
    # scalar restatement in the kernel's precision (numpy scalars round after every operation like the device does;
    # fp32 products are not contracted with the subtraction here, so compare fp32 to a few ulps and fp64 exactly where
    # the device code has no FMA contraction either: both to 64 ulps of the factor's scale)
    import numpy as np
    npdt = np.float64 if dtype == torch.float64 else np.float32
    Gn = G.numpy().astype(npdt)
    Wn = W.cpu().numpy()
    for b in range(B):
        R = np.zeros((P, P), dtype=npdt)
        for j in range(P):
            for i in range(j + 1):
                s = npdt(0.5) * (Gn[b, i, j] + Gn[b, j, i])
                for m in range(i):
                    s = npdt(s - R[m, i] * R[m, j])
                R[i, j] = np.sqrt(s) if i == j else npdt(s / R[i, i])
        Wref = np.linalg.inv(R.astype(np.float64))
        eps = np.finfo(npdt).eps
        assert np.abs(Wn[b] - Wref).max() <= 64 * eps * np.abs(Wref).max() * P, (b, np.abs(Wn[b] - Wref).max())
        assert np.abs(np.tril(Wn[b], -1)).max() == 0.0
        Gs = 0.5 * (Gn[b].astype(np.float64) + Gn[b].astype(np.float64).T)
        # W^T G W = I: the defining property of the inverse Cholesky factor
        assert np.abs(Wn[b].astype(np.float64).T @ Gs @ Wn[b].astype(np.float64) - np.eye(P)).max() <= 256 * eps * P
    # bit-reproducible, and a non-positive pivot is flagged at its index (sticky flag)
    W2 = torch.empty_like(W)
    K.panel_chol(G.to(dev), W2, info, P)
    assert torch.equal(W, W2)
    if P >= 3:
        Gb = G.clone()
        Gb[1, 2, :] = 0.0
        Gb[1, :, 2] = 0.0
        K.panel_chol(Gb.to(dev), W2, info, P)
        assert info.cpu().tolist()[1] == 3 and int(info.cpu()[0]) == 0


def test_pipeline_with_resident_launches_and_reserve_schedule_equals_one_group(dev, monkeypatch):
    """(r05) The two-group pipeline with RESIDENT panel launches on per-group CU-masked streams, and the panel stream
    changing with the basis width (`reserve_schedule`: more units for the chains once the basis is long), is the same
    iteration as the one-group run: iteration count, eigenvalues (closed form), residual history."""
    from xitorch_amd import synthetic
    B, N, p = 4, 2048, 6
    mat = synthetic.dense_symmetric(B, N, "S2", dtype=torch.float64, device=dev)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    assert A.symmetric_storage
    from xitorch_amd.linalg import _panel
    monkeypatch.setattr(_panel, "K1S_MIN_BYTES", 0.0)            # (the upper-triangle kernel also for this small batch)
    monkeypatch.setattr(K, "K1S_OPTS", K.K1S_PERSIST)            # resident launches whatever the launch size
    tr1, tr2 = {}, {}
    ev1, X1 = davidson(A, p, "lowest", min_eps=1e-8, overlap=False, trace=tr1)
    ev2, X2 = davidson(A, p, "lowest", min_eps=1e-8, overlap=True, k1_streams=True,
                       reserve_schedule=[(0, 32), (60, 64), (150, 96)], trace=tr2)
    assert tr2["groups"] == 2 and tr1["groups"] == 1 and tr2["panel_kernel"] == "K1s"
    assert tr1["niter"] == tr2["niter"] and tr2["basis_size"] > 150
    exact = synthetic.spectrum("S2", N, device=dev)[:p]
    assert (ev2 - exact).abs().max().item() <= 1e-10 and (ev1 - ev2).abs().max().item() <= 1e-11
    assert max(abs(a - b) for a, b in zip(tr1["resid_history"], tr2["resid_history"])) <= 1e-9
    assert (mat @ X2 - X2 * ev2.unsqueeze(-2)).abs().max().item() <= 1e-7
