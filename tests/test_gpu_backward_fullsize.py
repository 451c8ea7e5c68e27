"""-m gpu: (1) implicit backward of the native iterative paths vs dense autograd; (2) the BASELINE.json
full-size configs through size-independent properties (closed-form spectrum / manufactured solution)."""
import warnings
import pytest
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic as syn, kernels as K
from xitorch_amd.linalg import symeig, svd, solve
from xitorch_amd.linalg import native_krylov as nk

pytestmark = pytest.mark.gpu
f64 = torch.float64


def test_symeig_davidson_backward_matches_exacteig(dev):
    # symeig_torchfcn.backward: shifted multi-RHS solve (A - lam_i) g_i = -P b_i through the native CG
    g = torch.Generator().manual_seed(21)
    n = 60
    R = torch.rand(2, n, n, dtype=f64, generator=g)
    base = ((R + R.transpose(-2, -1)) * 0.5 + torch.diag(torch.arange(n, dtype=f64) * 2.0)).to(dev)
    wts = torch.arange(1.0, n + 1.0, dtype=f64, device=dev).unsqueeze(-1)

    def loss(mat, method, **kw):
        sym = (mat + mat.transpose(-2, -1)) * 0.5
        ev, X = symeig(xa.LinearOperator.m(sym, True), 3, "lowest", method=method, **kw)
        return (ev * torch.tensor([1.0, 2.0, 3.0], dtype=f64, device=dev)).sum() + (X.abs() * wts).sum()
    m1 = base.clone().requires_grad_()
    l1 = loss(m1, "davidson", min_eps=1e-10,
              bck_options=dict(method="cg", rtol=1e-12, atol=1e-14, posdef=False, max_niter=2000))
    g1, = torch.autograd.grad(l1, (m1,))
    m2 = base.clone().requires_grad_()
    g2, = torch.autograd.grad(loss(m2, "exacteig"), (m2,))
    assert torch.allclose(g1, g2, rtol=1e-6, atol=1e-7), (g1 - g2).abs().max().item()


def test_svd_davidson(dev):
    g = torch.Generator().manual_seed(22)
    a = torch.rand(2, 90, 40, dtype=f64, generator=g)
    s_ref = torch.linalg.svdvals(a)[..., :3]
    A = xa.LinearOperator.m(a.to(dev))
    with torch.no_grad():
        u, s, vh = svd(A, k=3, mode="uppest", method="davidson", min_eps=1e-9)
    assert torch.allclose(torch.flip(s.cpu(), dims=[-1]), s_ref, atol=1e-8)
    assert torch.allclose(A.mm(vh.transpose(-2, -1)), u * s.unsqueeze(-2), atol=1e-7)


def test_solve_with_many_columns_and_broadcast_operator(dev):
    # benchmarks_solve.py shape family: ncols = 50, one operator for a batch of right-hand sides
    g = torch.Generator().manual_seed(23)
    n = 100
    R = torch.rand(n, n, dtype=f64, generator=g)
    a = ((R + R.T) * 0.05 + torch.eye(n, dtype=f64)).to(dev)
    b = torch.rand(3, n, 50, dtype=f64, generator=g).to(dev)

    class Op(xa.LinearOperator):           # implicit operator -> default method cg
        def __init__(self, m):
            super().__init__(m.shape, is_hermitian=True, dtype=m.dtype, device=m.device)
            self.m_ = m

        def _mv(self, x):
            return torch.matmul(self.m_, x.unsqueeze(-1)).squeeze(-1)

        def _getparamnames(self, prefix=""):
            return [prefix + "m_"]
    with torch.no_grad():
        x = solve(Op(a), b, rtol=1e-10, atol=1e-12, posdef=True)
        x2 = nk.cg(xa.LinearOperator.m(a, True), b, rtol=1e-10, atol=1e-12, posdef=True)   # native K1 path
    ref = torch.linalg.solve(a, b)
    assert x.shape == (3, n, 50)
    assert torch.allclose(x, ref, rtol=1e-8, atol=1e-9) and torch.allclose(x2, ref, rtol=1e-8, atol=1e-9)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("storage", ["general", "symmetric"])
def test_fullsize_config2_symeig_properties(dev, storage):
    """BASELINE configs[1] at full size (64 x 16384^2 fp64 = 137 GB): eigenvalues against the exact closed-form
    spectrum, residual identity A X = X E, orthonormality — all size independent.  `general`: full-matrix panel
    kernel; `symmetric`: LinearOperator.m finds the storage exactly symmetric -> upper-triangle kernel (what
    bench.py runs).  Both go through the two-group pipeline (137 GB > 8 GiB)."""
    B, N, p = 64, 16384, 6
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip("needs ~140 GB of free HBM")
    mat = torch.empty((B, N, N), dtype=f64, device=dev)
    syn.dense_symmetric(B, N, "S1", device=dev, out=mat)
    A = xa.MatrixLinearOperator(mat, True) if storage == "general" else xa.LinearOperator.m(mat, is_hermitian=True)
    assert bool(getattr(A, "symmetric_storage", False)) == (storage == "symmetric")
    tr = {}
    with torch.no_grad():
        ev, X = symeig(A, neig=p, mode="lowest", method="davidson", min_eps=1e-8, rng_device="device", trace=tr)
    assert tr["groups"] == 2
    exact = syn.spectrum("S1", N, device=dev)[:p]
    assert (ev - exact).abs().max().item() <= 1e-10 * exact.abs().max().item()
    assert torch.all(ev[:, 1:] > ev[:, :-1])
    Xp = X.transpose(-2, -1).contiguous()                               # panel-major (B, p, N)
    AX = K.dense_mm(mat, Xp)                                            # row-sweep variant as an independent check
    assert (AX - Xp * ev.unsqueeze(-1)).abs().max().item() <= 1e-7
    G = torch.matmul(Xp, Xp.transpose(-2, -1))
    assert (G - torch.eye(p, dtype=f64, device=dev)).abs().max().item() < 1e-9
    assert tr["niter"] <= 25
    del mat
    torch.cuda.empty_cache()


@pytest.mark.timeout(600)
def test_fullsize_config3_bicgstab_properties(dev):
    """BASELINE configs[2] at full size (banded bw=127, N=65536, batch=256 fp64): manufactured solution."""
    B, N, hb = 256, 65536, 63
    band = syn.banded(B, N, hb=hb, device=dev)
    xs = syn.banded_rhs_solution(B, N, device=dev)
    A = xa.BandedLinearOperator(band)
    rhs = A.mm(xs)
    # the apply agrees with the plain-torch definition on a slice of the batch
    assert torch.allclose(rhs[:2], syn.banded_apply_reference(band[:2], xs[:2]), rtol=1e-12, atol=1e-12)
    tr = {}
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        x = nk.bicgstab(A, rhs, rtol=1e-10, atol=1e-12, posdef=True, trace=tr)
    assert tr["converged"]
    resid = (A.mm(x) - rhs).norm(dim=-2)
    assert torch.all(resid <= 1e-10 * rhs.norm(dim=-2) * 1.001)
    assert (x - xs).abs().max().item() < 1e-7
    # linearity of the operator at full size
    assert torch.allclose(A.mm(2.5 * xs), 2.5 * rhs, rtol=1e-12, atol=1e-12)


def test_no_device_memory_growth(dev):
    """Counterpart of the reference's test_memleak.py: repeated calls of the functionals must not accumulate
    device memory (workspaces are cached per stream, everything else is released)."""
    import gc
    g = torch.Generator().manual_seed(51)
    n = 96
    R = torch.rand(2, n, n, dtype=f64, generator=g)
    sym = ((R + R.transpose(-2, -1)) * 0.05 + torch.diag(torch.arange(n, dtype=f64))).to(dev)
    Bm = torch.rand(2, n, 2, dtype=f64, generator=g).to(dev)
    A = xa.LinearOperator.m(sym, True)

    def fcn(y, a):
        return torch.tanh(torch.einsum("bij,bj->bi", a, y) * 0.01 + 0.1) + y / 2.0

    def one_round():
        with torch.no_grad():
            symeig(A, neig=3, method="davidson", min_eps=1e-8)
            nk.cg(A, Bm, rtol=1e-9, posdef=True)
            nk.bicgstab(A, Bm, rtol=1e-9, posdef=True)
        m = sym.clone().requires_grad_()
        from xitorch_amd.optimize import rootfinder
        y = rootfinder(fcn, torch.zeros(2, n, dtype=f64, device=dev), params=(m,), alpha=-1.0, f_tol=1e-9)
        y.sum().backward()
    for _ in range(2):
        one_round()            # warm the caches (workspaces, cuBLAS-like handles)
    gc.collect()
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    for _ in range(5):
        one_round()
    gc.collect()
    torch.cuda.synchronize()
    after = torch.cuda.memory_allocated()
    assert after - before <= 1 << 20, (before, after)


@pytest.mark.timeout(600)
def test_fullsize_config5_shard_fp32_symeig(dev):
    """Per-GPU shard of BASELINE configs[4] (16 x 32768^2 fp32 = 68.7 GB): fp32 kernels end to end, eigenvalues
    against the exact closed-form spectrum at fp32 accuracy, residual identity."""
    B, N, p = 16, 32768, 6
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 80e9:
        pytest.skip("needs ~70 GB of free HBM")
    mat = torch.empty((B, N, N), dtype=torch.float32, device=dev)
    syn.dense_symmetric(B, N, "S1", dtype=torch.float32, device=dev, out=mat)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    tr = {}
    with torch.no_grad():
        ev, X = symeig(A, neig=p, mode="lowest", method="davidson", min_eps=2e-3, rng_device="device", max_niter=60,
                       trace=tr)
    assert ev.dtype == torch.float32 and tr["stop_reason"] == "converged" and tr["groups"] == 2
    exact = syn.spectrum("S1", N, device=dev)[:p]
    assert (ev.double() - exact).abs().max().item() <= 5e-4
    Xp = X.transpose(-2, -1).contiguous()
    AX = K.dense_mm(mat, Xp)
    assert (AX - Xp * ev.unsqueeze(-1)).abs().max().item() <= 2e-2
    del mat
    torch.cuda.empty_cache()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("storage,kernel", [("symmetric", "K1sw"), ("general", "K1w")])
def test_fullsize_config5_shard_fp32_wide_panel_on_the_matrix_cores(dev, storage, kernel):
    """BASELINE configs[4] AS STATED ("fp32, MFMA A@V panel"): the per-GPU shard 16 x 32768^2 fp32 with a 16-column
    eigen-block (neig = nguess = 16), so that every operator-panel product of the eigensolver runs on the matrix cores
    inside the two-group pipeline: K1sw (r04; exactly symmetric storage, upper triangle streamed once,
    v_mfma_f32_16x16x4_f32 for both y_I += A_IJ x_J and y_J += A_IJ^T x_I, xk_symmwide.hip) or, for an operator whose
    storage is only allclose-symmetric, K1w (full matrix, xk_wide.hip).
    Eigenvalues against the closed form at fp32 accuracy, residual identity, orthonormality."""
    B, N, p = 16, 32768, 16
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 80e9:
        pytest.skip("needs ~70 GB of free HBM")
    mat = torch.empty((B, N, N), dtype=torch.float32, device=dev)
    syn.dense_symmetric(B, N, "S1:16", dtype=torch.float32, device=dev, out=mat)
    A = xa.LinearOperator.m(mat, is_hermitian=True)
    if storage == "general":
        A.symmetric_storage = False
    tr = {}
    with torch.no_grad():
        ev, X = symeig(A, neig=p, mode="lowest", method="davidson", min_eps=2e-3, rng_device="device", max_niter=60,
                       trace=tr)
    assert ev.dtype == torch.float32 and tr["stop_reason"] == "converged"
    assert tr["panel_kernel"] == kernel, tr["panel_kernel"]
    exact = syn.spectrum("S1:16", N, device=dev)[:p]
    assert (ev.double() - exact).abs().max().item() <= 1e-3
    Xp = X.transpose(-2, -1).contiguous()
    AX = K.dense_mm(mat, Xp)
    assert (AX - Xp * ev.unsqueeze(-1)).abs().max().item() <= 3e-2
    G = torch.matmul(Xp, Xp.transpose(-2, -1))
    assert (G - torch.eye(p, device=dev)).abs().max().item() <= 1e-4
    del mat
    torch.cuda.empty_cache()


@pytest.mark.timeout(900)
def test_fullsize_config4_shard_rootfinder_backward(dev):
    """The per-GPU shard of BASELINE configs[3] (64 x 8192^2 fp64 = 34.4 GB of operators, f(y) = tanh(A y + 0.1) + y/2,
    batch 512 over 8 GPUs): the Broyden root, and the implicit gradient of sum(y) w.r.t. A (BiCGStab on the
    Jacobian + the operator-gradient kernel) checked against a finite difference along a random direction.
    Live at the peak: the operator, its gradient, the probe direction and the backward's own copies (~170 GB)."""
    import gc
    from xitorch_amd.optimize import rootfinder
    B, N = 64, 8192
    gc.collect()
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    assert free > 200e9, "only %.0f of %.0f GB free before the test: something upstream leaks device memory\n%s" % (
        free / 1e9, total / 1e9, torch.cuda.memory_summary(abbreviated=True))
    A = (syn.root_matrix(B, N, device=dev) * 2.0).requires_grad_()
    y0 = torch.zeros(B, N, dtype=f64, device=dev)

    def fcn(y, A_):
        return torch.tanh(xa.LinearOperator.m(A_, is_hermitian=False).mv(y) + 0.1) + y / 2.0
    kw = dict(method="broyden1", alpha=-1.0, max_rank=32, f_tol=1e-10)
    y = rootfinder(fcn, y0, params=(A,), bck_options=dict(method="bicgstab", posdef=True, rtol=1e-10), **kw)
    with torch.no_grad():
        assert fcn(y, A).abs().max().item() < 1e-8
    g, = torch.autograd.grad(y.sum(), (A,))
    del y
    assert torch.isfinite(g).all()
    # directional derivative: d/de sum(y(A + e D)) at e = 0 equals <g, D>; the operator is perturbed in place
    D = torch.empty_like(g).uniform_(-1.0, 1.0, generator=torch.Generator(device=dev).manual_seed(3)) / N
    an = sum((g[b] * D[b]).sum().item() for b in range(B))               # (torch.dot stops at 2^31 elements)
    del g
    eps = 1e-4
    with torch.no_grad():
        Ad = A.detach()
        Ad.add_(D, alpha=eps)
        yp = rootfinder(fcn, y0, params=(Ad,), **kw).sum().item()
        Ad.add_(D, alpha=-2.0 * eps)
        ym = rootfinder(fcn, y0, params=(Ad,), **kw).sum().item()
    fd = (yp - ym) / (2 * eps)
    assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (fd, an)
    del A, Ad, D
    torch.cuda.empty_cache()
