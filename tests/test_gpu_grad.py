"""-m gpu: the operator-gradient kernels (xk_banded_grad, xk_dense_outer) and the implicit backward of
`solve` on the banded operator of BASELINE configs[2] — small sizes against dense autograd, the full size
(bw=127, N=65536, batch=256) through a directional finite difference.

Reference behaviour being matched: solve_torchfcn.backward (xitorch/linalg/solve.py:165-222): adjoint solve
with A.H, then `torch.autograd.grad(-A.mm(x), params, v)`."""
import warnings
import pytest
import torch
import xitorch_amd as xa
from xitorch_amd import synthetic as syn, kernels as K
from xitorch_amd.linalg import solve
from xitorch_amd.linop import banded_apply_torch

pytestmark = pytest.mark.gpu
f64 = torch.float64


def _band_to_dense(band):
    nd, n = band.shape[-2:]
    hb = nd // 2
    A = torch.zeros((*band.shape[:-2], n, n), dtype=band.dtype, device=band.device)
    for d in range(nd):
        off = d - hb
        lo, hi = max(0, -off), min(n, n - off)
        if hi > lo:
            idx = torch.arange(lo, hi, device=band.device)
            A[..., idx, idx + off] = band[..., d, lo:hi]
    return A


@pytest.mark.parametrize("dtype,tol", [(f64, 1e-13), (torch.float32, 2e-5)])
@pytest.mark.parametrize("B,N,hb,C", [(3, 1000, 5, 1), (2, 1537, 63, 3), (1, 515, 2, 11), (4, 64, 40, 8)])
def test_banded_grad_kernel_vs_torch(dev, dtype, tol, B, N, hb, C):
    g = torch.Generator().manual_seed(5 + N)
    U = torch.randn(B, C, N, dtype=f64, generator=g)
    W = torch.randn(B, C, N, dtype=f64, generator=g)
    nd = 2 * hb + 1
    ref = torch.zeros(B, nd, N, dtype=f64)
    for d in range(nd):
        off = d - hb
        lo, hi = max(0, -off), min(N, N - off)
        if hi > lo:
            ref[:, d, lo:hi] = (U[:, :, lo:hi] * W[:, :, lo + off:hi + off]).sum(1)
    Ud, Wd = U.to(dev, dtype), W.to(dev, dtype)
    out = K.banded_grad(Ud, Wd, nd)
    scale = ref.abs().max().item()
    assert (out.double().cpu() - ref).abs().max().item() <= tol * scale * C
    # accumulate = True adds into the output
    out2 = out.clone()
    K.banded_grad(Ud, Wd, nd, out=out2, accumulate=True)
    assert (out2 - 2 * out).abs().max().item() <= 4 * tol * scale * C


@pytest.mark.parametrize("dtype,tol", [(f64, 1e-13), (torch.float32, 2e-5)])
@pytest.mark.parametrize("B,M,N,C", [(2, 130, 1026, 1), (3, 65, 513, 5), (1, 700, 700, 12), (2, 64, 2048, 8)])
def test_dense_outer_kernel_vs_torch(dev, dtype, tol, B, M, N, C):
    g = torch.Generator().manual_seed(9 + N)
    U = torch.randn(B, C, M, dtype=f64, generator=g)
    W = torch.randn(B, C, N, dtype=f64, generator=g)
    ref = torch.matmul(U.transpose(1, 2), W)
    out = K.dense_outer(U.to(dev, dtype), W.to(dev, dtype))
    assert out.shape == (B, M, N)
    assert (out.double().cpu() - ref).abs().max().item() <= tol * ref.abs().max().item() * C
    out2 = out.clone()
    K.dense_outer(U.to(dev, dtype), W.to(dev, dtype), out=out2, accumulate=True)
    # (C > 8 runs as two accumulating passes, so the sums associate differently: rounding-level agreement)
    assert (out2 - 2 * out).abs().max().item() <= 4 * tol * ref.abs().max().item() * C


def test_dense_and_banded_apply_gradients_any_order(dev):
    """first and second derivatives of the native applies w.r.t. the operator storage and the operand,
    against the same expressions in plain torch (broadcast operator batch included)"""
    g = torch.Generator().manual_seed(31)
    n, r, hb = 48, 3, 4
    band = torch.randn(2 * hb + 1, n, dtype=f64, generator=g).to(dev).requires_grad_()       # one operator ...
    x = torch.randn(3, n, r, dtype=f64, generator=g).to(dev).requires_grad_()               # ... three panels
    w1 = torch.randn(3, n, r, dtype=f64, generator=g).to(dev)

    def f_native(b_, x_, trans):
        op = xa.BandedLinearOperator(b_)
        y = op.rmm(x_) if trans else op.mm(x_)
        return (torch.tanh(y) * w1).sum()

    def f_torch(b_, x_, trans):
        return (torch.tanh(banded_apply_torch(b_, x_, trans)) * w1).sum()
    for trans in (False, True):
        ga = torch.autograd.grad(f_native(band, x, trans), (band, x), create_graph=True)
        gb = torch.autograd.grad(f_torch(band, x, trans), (band, x), create_graph=True)
        for a, b in zip(ga, gb):
            assert torch.allclose(a, b, rtol=1e-10, atol=1e-11)
        # second order: differentiate a scalar of the first derivatives
        sa = (ga[0] ** 2).sum() + (ga[1] * w1).sum()
        sb = (gb[0] ** 2).sum() + (gb[1] * w1).sum()
        ha = torch.autograd.grad(sa, (band, x))
        hb_ = torch.autograd.grad(sb, (band, x))
        for a, b in zip(ha, hb_):
            assert torch.allclose(a, b, rtol=1e-9, atol=1e-10)
    # dense operator, broadcast over the panel batch
    mat = torch.randn(n, n, dtype=f64, generator=g).to(dev).requires_grad_()

    def d_native(m_, x_, trans):
        op = xa.MatrixLinearOperator(m_, False)
        y = op.rmm(x_) if trans else op.mm(x_)
        return (torch.tanh(y) * w1).sum()

    def d_torch(m_, x_, trans):
        return (torch.tanh(torch.matmul(m_.transpose(-2, -1) if trans else m_, x_)) * w1).sum()
    for trans in (False, True):
        ga = torch.autograd.grad(d_native(mat, x, trans), (mat, x), create_graph=True)
        gb = torch.autograd.grad(d_torch(mat, x, trans), (mat, x), create_graph=True)
        for a, b in zip(ga, gb):
            assert torch.allclose(a, b, rtol=1e-10, atol=1e-11)
        ha = torch.autograd.grad((ga[0] ** 2).sum() + (ga[1] * w1).sum(), (mat, x))
        hb_ = torch.autograd.grad((gb[0] ** 2).sum() + (gb[1] * w1).sum(), (mat, x))
        for a, b in zip(ha, hb_):
            assert torch.allclose(a, b, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("batch,ncols", [((2,), 1), ((), 3), ((2, 2), 2)])
def test_solve_banded_bicgstab_backward_vs_dense_autograd(dev, batch, ncols):
    """configs[2] shape family at small N: solve(BandedLinearOperator, B, method="bicgstab") and its implicit
    backward (gradients w.r.t. the band and the right-hand side) against torch.linalg.solve + autograd"""
    n, hb = 200, 7
    nb = 1
    for d in batch:
        nb *= d
    band0 = syn.banded(max(nb, 1), n, hb=hb).reshape(*batch, 2 * hb + 1, n).to(dev)
    g = torch.Generator().manual_seed(17)
    B0 = torch.randn(*batch, n, ncols, dtype=f64, generator=g).to(dev)
    wts = torch.randn(*batch, n, ncols, dtype=f64, generator=g).to(dev)
    band = band0.clone().requires_grad_()
    Bm = B0.clone().requires_grad_()
    opts = dict(rtol=1e-12, atol=1e-14, posdef=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        x = solve(xa.BandedLinearOperator(band), Bm, method="bicgstab", bck_options=dict(method="bicgstab", **opts),
                  **opts)
    gband, gB = torch.autograd.grad((x * wts).sum() + (x ** 2).sum(), (band, Bm))
    band2 = band0.clone().requires_grad_()
    B2 = B0.clone().requires_grad_()
    x2 = torch.linalg.solve(_band_to_dense(band2), B2)
    gband2, gB2 = torch.autograd.grad((x2 * wts).sum() + (x2 ** 2).sum(), (band2, B2))
    assert torch.allclose(x, x2, rtol=1e-9, atol=1e-10)
    assert torch.allclose(gB, gB2, rtol=1e-8, atol=1e-9)
    # entries of the DIA storage that fall outside the matrix have no effect: the dense route gives them a zero
    # gradient, and so must the kernel
    assert torch.allclose(gband, gband2, rtol=1e-8, atol=1e-9), (gband - gband2).abs().max().item()


@pytest.mark.timeout(900)
def test_fullsize_config3_bicgstab_implicit_backward(dev):
    """BASELINE configs[2] at full size WITH the implicit backward: banded bw=127, N=65536, batch=256 fp64.
    loss = sum(w * x(band, b)); its gradient w.r.t. the band (17 GB, written by xk_banded_grad after the adjoint
    BiCGStab solve with the transposed banded kernel) is checked along a random direction against a central
    finite difference of the forward solve, and the gradient w.r.t. b through the adjoint identity."""
    from xitorch_amd.linalg import native_krylov as nk
    B, N, hb = 256, 65536, 63
    band = syn.banded(B, N, hb=hb, device=dev).requires_grad_()
    xs = syn.banded_rhs_solution(B, N, device=dev)
    with torch.no_grad():
        rhs0 = xa.BandedLinearOperator(band.detach()).mm(xs)
    rhs = rhs0.clone().requires_grad_()
    gen = torch.Generator(device=dev).manual_seed(11)
    w = torch.empty_like(xs).uniform_(-1.0, 1.0, generator=gen)
    opts = dict(rtol=1e-12, atol=1e-14, posdef=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        x = solve(xa.BandedLinearOperator(band), rhs, method="bicgstab", bck_options=dict(method="bicgstab", **opts),
                  **opts)
        gband, grhs = torch.autograd.grad((x * w).sum(), (band, rhs))
    assert gband.shape == band.shape and torch.isfinite(gband).all()
    x = x.detach()
    # d loss / d rhs = v with A^T v = w  (solve.py:180-186): check the adjoint system's residual
    with torch.no_grad():
        Ad = xa.BandedLinearOperator(band.detach())
        assert (Ad.rmm(grhs) - w).norm(dim=-2).max().item() <= 1e-9 * w.norm(dim=-2).max().item()
        # d loss / d band[b,d,i] = -v[b,i] x[b,i+d-hb]  (entrywise identity on a few probes)
        for (b, d, i) in [(0, hb, 0), (3, 0, 70), (100, 2 * hb, 1000), (255, hb + 5, N - 6), (17, 10, 5)]:
            col = i + d - hb
            expect = -(grhs[b, i, 0] * x[b, col, 0]).item() if 0 <= col < N else 0.0
            assert abs(gband[b, d, i].item() - expect) <= 1e-9 * max(1.0, abs(expect))
        # directional derivative vs a central finite difference of the forward solve, along a band perturbation with
        # a random part and a part aligned with the gradient's sign pattern (so that <g, D> is far above the noise
        # floor of the two solves: |loss error| <= |w| |dx| ~ 1e-5, divided by 2 eps)
        D = torch.empty_like(gband).uniform_(-1.0, 1.0, generator=gen).add_(torch.sign(gband), alpha=0.2).mul_(0.05)
        an = sum((gband[b] * D[b]).sum().item() for b in range(B))          # (torch.dot stops at 2^31 elements)
        eps = 1e-3
        bd = band.detach()
        bd.add_(D, alpha=eps)
        lp = (nk.bicgstab(xa.BandedLinearOperator(bd), rhs0, **opts) * w).sum().item()
        bd.add_(D, alpha=-2 * eps)
        lm = (nk.bicgstab(xa.BandedLinearOperator(bd), rhs0, **opts) * w).sum().item()
        fd = (lp - lm) / (2 * eps)
    assert abs(an) > 1e3 and abs(fd - an) <= 1e-5 * abs(an), (fd, an)
