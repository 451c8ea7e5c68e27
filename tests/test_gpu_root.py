"""-m gpu: native Broyden root finder (+ implicit backward) vs the oracle and the reference's golden outputs."""
import os
import numpy as np
import pytest
import torch
from oracle import rootfinder as oroot
from tests import cases
import xitorch_amd as xa
from xitorch_amd.optimize import rootfinder
from xitorch_amd.optimize import native_root as nr

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("case", cases.ROOT_CASES, ids=[c["name"] for c in cases.ROOT_CASES])
def test_broyden1_vs_golden_and_oracle(dev, case):
    gold = np.load(os.path.join(GOLD, "root_%s.npz" % case["name"]))
    fcn, y0, params = cases.root_inputs(case)
    tr = {}
    meth = case.get("method", "broyden1")
    y = getattr(nr, meth)(fcn, y0.to(dev), tuple(p.to(dev) for p in params), trace=tr, **case["kwargs"])
    yg = torch.from_numpy(gold["y"])
    assert list(y.shape) == list(yg.shape)
    # same iterate as the reference (it returns the iterate BEFORE the converged one, quirk Q1)
    assert (y.cpu() - yg).abs().max().item() <= 1e-8
    assert tr["nfev"] == int(gold["nfev"]) and tr["niter"] == int(gold["niter"])
    assert abs(fcn(y.cpu(), *params).norm().item() - float(gold["fnorm"])) <= 1e-9
    # live oracle agrees too
    yo = getattr(oroot, meth)(fcn, y0, params, **case["kwargs"])
    assert (y.cpu() - yo).abs().max().item() <= 1e-8


def test_rootfinder_readme_gradients(dev):
    # README.md:16-32 flow of the reference: root, first and second derivative w.r.t. A
    gold = np.load(os.path.join(GOLD, "root_readme2.npz"))
    A = torch.tensor([[1.1, 0.4], [0.3, 0.8]], dtype=torch.float64, device=dev).requires_grad_()
    y0 = torch.zeros((2, 1), dtype=torch.float64, device=dev)
    y = rootfinder(cases.tanh_fcn, y0, params=(A,))
    assert (y.detach().cpu() - torch.from_numpy(gold["y"])).abs().max().item() < 1e-9
    g1, = torch.autograd.grad(y.sum(), (A,), create_graph=True)
    assert (g1.detach().cpu() - torch.from_numpy(gold["grad0"])).abs().max().item() < 1e-8
    g2, = torch.autograd.grad(g1.sum(), (A,))
    assert (g2.cpu() - torch.from_numpy(gold["gradgrad0"])).abs().max().item() < 1e-7


def test_rootfinder_batched_backward_bicgstab(dev):
    # config-4 shape family: per-batch dense A_b, y (B, N); implicit backward through the native bicgstab
    case = dict(kind="tanh", nbatch=3, n=40)
    fcn, y0, (A,) = cases.root_inputs(case)
    Ad = A.to(dev).requires_grad_()
    y = rootfinder(fcn, y0.to(dev), params=(Ad,), method="broyden1", alpha=-1.0, f_tol=1e-10, x_tol=1e-10,
                   bck_options=dict(method="bicgstab", posdef=True, rtol=1e-12, atol=1e-14))
    assert fcn(y, Ad).abs().max().item() < 1e-8
    gA, = torch.autograd.grad(y.sum(), (Ad,))
    # reference gradient by the implicit function theorem with dense algebra on the CPU
    yc = y.detach().cpu()
    Ac = A.clone().requires_grad_()
    f = fcn(yc, Ac)
    n = yc.numel()
    J = torch.autograd.functional.jacobian(lambda yy: fcn(yy, A), yc).reshape(n, n)
    g = torch.linalg.solve(J.T, -torch.ones(n, dtype=torch.float64)).reshape(yc.shape)
    gref, = torch.autograd.grad(f, (Ac,), grad_outputs=g)
    assert (gA.cpu() - gref).abs().max().item() < 1e-7 * max(1.0, gref.abs().max().item())


def test_other_methods_and_errors(dev):
    fcn, y0, (A,) = cases.root_inputs(dict(kind="tanh", nbatch=2, n=16))
    yd, Ad = y0.to(dev), A.to(dev)
    for method, kw in (("broyden2", dict(alpha=-1.0)), ("linearmixing", dict(alpha=-1.0, maxiter=400)),
                       ("newton", dict())):
        y = rootfinder(fcn, yd, params=(Ad,), method=method, f_tol=1e-9, **kw)
        assert fcn(y, Ad).abs().max().item() < 1e-6, method
    with pytest.raises(RuntimeError):
        rootfinder(fcn, yd, params=(Ad,), method="nonexistent")
    with pytest.raises(RuntimeError):
        nr.broyden1(fcn, y0, (A,), alpha=-1.0)          # CPU tensors: no fallback
    # a user plug-in method (callable) is used as is
    called = {}

    def mymethod(f, y0_, params, **opts):
        called["ok"] = True
        return nr.broyden1(f, y0_, params, alpha=-1.0, f_tol=1e-9)
    rootfinder(fcn, yd, params=(Ad,), method=mymethod)
    assert called["ok"]


def test_broyden_uv0_svd_and_rank_restart(dev):
    # uv0="svd": rank-1 SVD of the Jacobian (native davidson on the autograd operator) seeds the model
    fcn, y0, (A,) = cases.root_inputs(dict(kind="tanh", nbatch=2, n=10))
    tr = {}
    y = nr.broyden1(fcn, y0.to(dev), (A.to(dev),), alpha=-1.0, uv0="svd", f_tol=1e-9, trace=tr)
    assert fcn(y, A.to(dev)).abs().max().item() < 1e-7 and tr["converged"]
    # max_rank small: the history is dropped as a whole whenever it overflows (quirk Q3) and it still converges
    tr2 = {}
    y2 = nr.broyden1(fcn, y0.to(dev), (A.to(dev),), alpha=-1.0, max_rank=3, f_tol=1e-9, trace=tr2)
    assert fcn(y2, A.to(dev)).abs().max().item() < 1e-7 and tr2["rank"] <= 4


def test_equilibrium_and_minimize_native(dev):
    from xitorch_amd.optimize import equilibrium, minimize
    g = torch.Generator().manual_seed(41)
    n = 24
    A = (torch.rand(n, n, dtype=torch.float64, generator=g) * 0.1).to(dev).requires_grad_()
    y0 = torch.zeros(n, 1, dtype=torch.float64, device=dev)

    def fp(y, a):
        return torch.tanh(a @ y + 0.1)
    for method, kw in (("broyden1", dict(alpha=-1.0)), ("anderson_acc", dict())):
        y = equilibrium(fp, y0, params=(A,), method=method, f_tol=1e-10, x_tol=1e-10, **kw)
        assert (y - fp(y, A)).abs().max().item() < 1e-8, method
    gA, = torch.autograd.grad(y.sum(), (A,))
    assert torch.isfinite(gA).all() and gA.abs().max().item() > 0

    target = torch.linspace(-1, 1, n, dtype=torch.float64, device=dev).unsqueeze(-1)

    def energy(y, a):
        return 0.5 * ((y - target) ** 2).sum() + 0.25 * (y ** 4).sum() + (a.sum() * 0.0)
    ym = minimize(energy, y0, params=(A,), method="broyden1", alpha=-1.0, f_tol=1e-10, x_tol=1e-10)
    grad = (ym - target) + ym ** 3
    assert grad.abs().max().item() < 1e-8
