"""-m gpu: complex64 / complex128 through the native paths — the dense operator apply (real K1 kernels on the
interleaved storage), its gradients, the Krylov front-end with implicit backward, and the root finder.

The reference's own hot-path tests run these dtypes (xitorch/_tests/test_linop_fcns.py:474-524, 631-676;
_tests/test_optimize.py:118-155); the per-solver golden comparisons live in test_gpu_solve.py / test_gpu_root.py
(cases *_c128)."""
import warnings
import pytest
import torch
import xitorch_amd as xa
from xitorch_amd import kernels as K
from xitorch_amd.linalg import solve
from xitorch_amd.optimize import rootfinder

pytestmark = pytest.mark.gpu
c128, c64 = torch.complex128, torch.complex64


def _crand(g, *shape, dtype=c128):
    z = torch.complex(torch.randn(shape, dtype=torch.float64, generator=g),
                      torch.randn(shape, dtype=torch.float64, generator=g))
    return z.to(dtype)


@pytest.mark.parametrize("dtype,tol", [(c128, 1e-13), (c64, 3e-5)])
@pytest.mark.parametrize("B,M,N,P", [(2, 64, 128, 3), (1, 130, 70, 1), (3, 257, 257, 6), (2, 96, 512, 14)])
def test_complex_panel_product_vs_torch(dev, dtype, tol, B, M, N, P):
    g = torch.Generator().manual_seed(M + N)
    A = _crand(g, B, M, N, dtype=dtype)
    x = _crand(g, B, P, N, dtype=dtype)            # panel-major: A x
    xt = _crand(g, B, P, M, dtype=dtype)           # for the adjoint / transpose
    Ad, xd, xtd = A.to(dev), x.to(dev), xt.to(dev)
    A128, x128, xt128 = A.to(c128), x.to(c128), xt.to(c128)
    scale = float(N) ** 0.5 * 4
    ref = torch.matmul(x128, A128.transpose(-2, -1))
    assert (K.dense_mm_complex(Ad, xd).cpu().to(c128) - ref).abs().max().item() <= tol * scale
    refH = torch.matmul(xt128, A128.conj())
    assert (K.dense_mm_complex(Ad, xtd, adjoint=True).cpu().to(c128) - refH).abs().max().item() <= tol * scale
    refT = torch.matmul(xt128, A128)
    assert (K.dense_mm_complex(Ad, xtd, adjoint=True, conj_io=True).cpu().to(c128) - refT).abs().max().item() \
        <= tol * scale
    refC = torch.matmul(x128, A128.conj().transpose(-2, -1))
    assert (K.dense_mm_complex(Ad, xd, conj_io=True).cpu().to(c128) - refC).abs().max().item() <= tol * scale
    # one operator broadcast over the panel batch
    y1 = K.dense_mm_complex(Ad[:1], xd)
    assert (y1.cpu().to(c128) - torch.matmul(x128, A128[:1].transpose(-2, -1))).abs().max().item() <= tol * scale


def test_complex_operator_surface_and_views(dev):
    """MatrixLinearOperator on complex HIP tensors: mm / rmm / H (lazily conjugated transposed view) / fullmatrix
    agree with torch; nothing copies or resolves the operator"""
    g = torch.Generator().manual_seed(7)
    A = _crand(g, 2, 40, 40)
    X = _crand(g, 2, 40, 3)
    Ad, Xd = A.to(dev), X.to(dev)
    op = xa.LinearOperator.m(Ad)
    assert not op.is_hermitian
    assert torch.allclose(op.mm(Xd).cpu(), A @ X, atol=1e-12)
    assert torch.allclose(op.rmm(Xd).cpu(), A.transpose(-2, -1).conj() @ X, atol=1e-12)
    assert torch.allclose(op.mv(Xd[..., 0]).cpu(), (A @ X)[..., 0], atol=1e-12)
    opH = op.H
    assert opH.mat.data_ptr() == Ad.data_ptr()                       # a view, not a copy
    assert torch.allclose(opH.mm(Xd).cpu(), A.transpose(-2, -1).conj() @ X, atol=1e-12)
    assert torch.allclose(opH.rmm(Xd).cpu(), A @ X, atol=1e-12)
    # plain transposed view and plain conjugated view
    opT = xa.MatrixLinearOperator(Ad.transpose(-2, -1), False)
    assert torch.allclose(opT.mm(Xd).cpu(), A.transpose(-2, -1) @ X, atol=1e-12)
    opC = xa.MatrixLinearOperator(Ad.conj(), False)
    assert torch.allclose(opC.mm(Xd).cpu(), A.conj() @ X, atol=1e-12)
    assert torch.allclose(opC.rmm(Xd).cpu(), A.transpose(-2, -1) @ X, atol=1e-12)
    # Hermitian complex operator
    Hm = (A + A.transpose(-2, -1).conj()) * 0.5
    oph = xa.LinearOperator.m(Hm.to(dev))
    assert oph.is_hermitian and not oph.symmetric_storage
    assert torch.allclose(oph.mm(Xd).cpu(), Hm @ X, atol=1e-12)


def test_complex_apply_gradients_match_torch(dev):
    g = torch.Generator().manual_seed(8)
    n, r = 24, 2
    A0, x0 = _crand(g, 2, n, n), _crand(g, 2, n, r)
    w = _crand(g, 2, n, r).to(dev)
    for trans in (False, True):
        A1 = A0.to(dev).requires_grad_()
        x1 = x0.to(dev).requires_grad_()
        op = xa.MatrixLinearOperator(A1, False)
        y = op.rmm(x1) if trans else op.mm(x1)
        loss = (y * w.conj()).real.sum() + (y.abs() ** 2).sum()
        gA, gx = torch.autograd.grad(loss, (A1, x1))
        A2 = A0.to(dev).requires_grad_()
        x2 = x0.to(dev).requires_grad_()
        y2 = torch.matmul(A2.transpose(-2, -1).conj() if trans else A2, x2)
        loss2 = (y2 * w.conj()).real.sum() + (y2.abs() ** 2).sum()
        gA2, gx2 = torch.autograd.grad(loss2, (A2, x2))
        assert torch.allclose(gA, gA2, rtol=1e-10, atol=1e-11)
        assert torch.allclose(gx, gx2, rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("method", ["cg", "bicgstab"])
def test_complex_solve_frontend_with_backward(dev, method):
    """the reference's test_solve_A_methods / _AEM_methods shapes (Hermitian complex A, many right-hand sides, E, M),
    residual identity and gradients w.r.t. A and B against torch.linalg.solve + autograd"""
    g = torch.Generator().manual_seed(12345)
    na, nc = 60, 8
    crand = lambda *s: torch.complex(torch.rand(s, dtype=torch.float64, generator=g),
                                     torch.rand(s, dtype=torch.float64, generator=g))
    a0 = crand(na, na) * 0.1 + torch.eye(na, dtype=c128)
    a0 = (a0 + a0.transpose(-2, -1).conj()) * 0.5
    m0 = crand(na, na) * 0.05 + torch.eye(na, dtype=c128) * 0.5
    m0 = (m0 + m0.transpose(-2, -1).conj()) * 0.5
    b0 = crand(2, na, nc) + 0.1
    e0 = crand(nc) * 0.1
    if method == "cg":
        e0 = e0.real.to(c128)            # keep A - e M Hermitian: CG's territory (tight tolerance below)

    class Op(xa.LinearOperator):              # implicit operator: the front-end takes the iterative path
        def __init__(self, mat):
            super().__init__(mat.shape, is_hermitian=True, dtype=mat.dtype, device=mat.device)
            self.mat = mat

        def _mv(self, x):
            return torch.matmul(self.mat, x.unsqueeze(-1)).squeeze(-1)

        def _getparamnames(self, prefix=""):
            return [prefix + "mat"]
    opts = dict(rtol=1e-11, atol=1e-13, posdef=True)
    for use_em in (False, True):
        a = a0.to(dev).requires_grad_()
        b = b0.to(dev).requires_grad_()
        e = e0.to(dev).requires_grad_() if use_em else None
        Mop = xa.LinearOperator.m(m0.to(dev)) if use_em else None
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            # dense wrapped operator through the native panel path (forced method), and the implicit operator
            x = solve(xa.LinearOperator.m(a), b, E=e, M=Mop, method=method, bck_options=dict(method=method, **opts),
                      **opts)
            xi = solve(Op(a), b, E=e, M=Mop, method=method, **opts)
        res = a @ x - ((m0.to(dev) @ x) * e.unsqueeze(-2) if use_em else 0) - b
        assert res.abs().max().item() < 1e-9
        assert (x - xi).abs().max().item() < 1e-9
        wts = crand(2, na, nc).to(dev)
        loss = (x * wts.conj()).real.sum()
        ins = (a, b) + ((e,) if use_em else ())
        grads = torch.autograd.grad(loss, ins)
        # dense reference: column-wise (A - e_c M) x_c = b_c
        a2 = a0.to(dev).requires_grad_()
        b2 = b0.to(dev).requires_grad_()
        if use_em:
            e2 = e0.to(dev).requires_grad_()
            cols = [torch.linalg.solve(a2 - e2[c] * m0.to(dev), b2[..., c:c + 1]) for c in range(nc)]
            x2 = torch.cat(cols, dim=-1)
            ins2 = (a2, b2, e2)
        else:
            x2 = torch.linalg.solve(a2, b2)
            ins2 = (a2, b2)
        grads2 = torch.autograd.grad((x2 * wts.conj()).real.sum(), ins2)
        for ga, gb in zip(grads, grads2):
            assert torch.allclose(ga, gb, rtol=1e-7, atol=1e-8), (ga - gb).abs().max().item()


def test_complex_rootfinder_matches_cpu_reference_formula(dev):
    """complex unknowns are packed as [Re; Im] (rootsolver.py:52-73): root of a complex tanh system on the GPU, and its
    implicit gradient w.r.t. the complex parameter against the dense implicit-function formula"""
    g = torch.Generator().manual_seed(3)
    nb, n = 2, 12
    A0 = torch.complex(torch.rand(nb, n, n, dtype=torch.float64, generator=g),
                       torch.rand(nb, n, n, dtype=torch.float64, generator=g)) * (0.5 / n)

    def fcn(y, A):
        return torch.tanh(torch.einsum("bij,bj->bi", A, y) + (0.1 + 0.05j)) + y / 2.0
    A = A0.to(dev).requires_grad_()
    y0 = torch.zeros(nb, n, dtype=c128, device=dev)
    y = rootfinder(fcn, y0, params=(A,), method="broyden1", alpha=-1.0, f_tol=1e-11, x_tol=1e-11,
                   bck_options=dict(method="bicgstab", posdef=True, rtol=1e-12, atol=1e-14))
    assert y.dtype == c128 and fcn(y, A).abs().max().item() < 1e-9
    w = torch.complex(torch.rand(nb, n, dtype=torch.float64, generator=g),
                      torch.rand(nb, n, dtype=torch.float64, generator=g)).to(dev)
    gA, = torch.autograd.grad((y * w.conj()).real.sum(), (A,))
    # finite difference along a random complex direction
    D = torch.complex(torch.rand(nb, n, n, dtype=torch.float64, generator=g),
                      torch.rand(nb, n, n, dtype=torch.float64, generator=g)).to(dev) * 0.1
    eps = 1e-6
    with torch.no_grad():
        kw = dict(method="broyden1", alpha=-1.0, f_tol=1e-13, x_tol=1e-13)
        yp = rootfinder(fcn, y0, params=(A.detach() + eps * D,), **kw)
        ym = rootfinder(fcn, y0, params=(A.detach() - eps * D,), **kw)
    fd = (((yp - ym) / (2 * eps)) * w.conj()).real.sum().item()
    an = (gA.conj() * D).real.sum().item()          # <grad, D> in the real inner product of C^n ~ R^2n
    assert abs(fd - an) <= 1e-6 * max(1.0, abs(an)), (fd, an)
