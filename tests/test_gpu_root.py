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
    # a variable in HOST memory is served by the host branch of the model (device dispatch, r06): same root
    yh = nr.broyden1(fcn, y0, (A,), alpha=-1.0, f_tol=1e-9)
    yg = nr.broyden1(fcn, yd, (Ad,), alpha=-1.0, f_tol=1e-9)
    assert (yh - yg.cpu()).abs().max().item() < 1e-9
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
    A = (torch.rand(n, n, dtype=torch.float64, generator=g) * 0.03).to(dev).requires_grad_()   # |A| < 1: a contraction
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


@pytest.mark.parametrize("case", cases.EXTRA_CASES, ids=[c["name"] for c in cases.EXTRA_CASES])
def test_extra_methods_vs_reference_goldens(dev, case):
    """anderson_acc / gd / adam / newton on the GPU against outputs of the REAL reference
    (equilibrium.py:9-134, minimizer.py:5-147, rootsolver.py:151-174; fixtures by make_golden.py::gen_extra)."""
    from tests.test_host_surface import _run_extra
    gold = np.load(os.path.join(GOLD, "extra_%s.npz" % case["name"]))
    y, nfev = _run_extra(case, dev)
    yg = torch.from_numpy(gold["y"])
    meth = case["method"]
    # the same iteration on another device: identical evaluation counts for the Newton / Anderson iterations (their
    # stopping tests sit orders of magnitude away from rounding), within 2 for the minimisers (|df| < f_rtol |f|)
    if meth in ("newton", "anderson_acc"):
        assert nfev == int(gold["nfev"]), (nfev, int(gold["nfev"]))
        assert (y.cpu() - yg).abs().max().item() <= 1e-9
    else:
        assert abs(nfev - int(gold["nfev"])) <= 2, (nfev, int(gold["nfev"]))
        assert (y.cpu() - yg).abs().max().item() <= 1e-6
    fcn, y0, params = cases.extra_inputs(case)
    out = fcn(y.cpu(), *params)
    quality = {"anderson_acc": lambda: (out - y.cpu()).norm(), "newton": lambda: out.norm()}.get(meth, lambda: out[1].norm())()
    assert quality.item() <= 2.0 * float(gold["quality"]) + 1e-12


def test_extra_methods_through_the_functionals(dev):
    """equilibrium(method="anderson_acc") / minimize(method="gd"|"adam") / rootfinder(method="newton") front-ends
    with their implicit backward on the GPU (optimize/rootfinder.py:104-288 of the reference)."""
    from xitorch_amd.optimize import equilibrium, minimize
    case = dict(method="anderson_acc", nbatch=2, n=24)
    fcn, y0, (A,) = cases.extra_inputs(case)
    Ad = A.to(dev).requires_grad_()
    y = equilibrium(fcn, y0.to(dev), params=(Ad,), method="anderson_acc", f_tol=1e-10, x_tol=1e-10, maxiter=200,
                    bck_options=dict(method="bicgstab", posdef=True, rtol=1e-12, atol=1e-14))
    assert (fcn(y, Ad) - y).abs().max().item() < 1e-8
    g, = torch.autograd.grad(y.sum(), (Ad,))
    # implicit-function-theorem gradient with dense algebra on the CPU
    yc, Ac = y.detach().cpu(), A.clone().requires_grad_()
    n = yc.numel()
    J = torch.autograd.functional.jacobian(lambda yy: (cases.fixed_point_fcn(yy, Ac.detach()) - yy).reshape(-1), yc)
    lam = torch.linalg.solve(J.reshape(n, n).T, -torch.ones(n, dtype=torch.float64))
    gref, = torch.autograd.grad((cases.fixed_point_fcn(yc, Ac).reshape(-1) * lam).sum(), (Ac,))
    assert torch.allclose(g.cpu(), gref, rtol=1e-6, atol=1e-8)
    # newton through rootfinder
    fr, y0r, (Ar,) = cases.root_inputs(dict(kind="tanh", nbatch=2, n=16))
    yn = rootfinder(fr, y0r.to(dev), params=(Ar.to(dev),), method="newton", f_tol=1e-11, x_tol=1e-11)
    assert fr(yn, Ar.to(dev)).abs().max().item() < 1e-9
    # minimize with gd: objective value function (autograd supplies the gradient)
    fobj = lambda yy, A_: cases.quartic_objective(yy, A_)[0]
    fq, y0q, (Aq,) = cases.extra_inputs(dict(method="gd", nbatch=2, n=16))
    ym = minimize(fobj, y0q.to(dev), params=(Aq.to(dev),), method="gd", step=5e-2, gamma=0.8, maxiter=800,
                  f_rtol=1e-13, x_rtol=1e-11)
    assert cases.quartic_objective(ym.cpu(), Aq)[1].abs().max().item() < 1e-3


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-13), (torch.float32, 3e-6)])
@pytest.mark.parametrize("L", [1, 7, 4096, 100003, 3 * 1024 * 1024 + 8])
def test_fused_blas1_kernels_vs_torch(dev, dtype, tol, L):
    """xk_vec_dots / xk_broyden_axpy (the driver's norms, dots and the low-rank apply / update) against torch,
    incl. lengths that are not a multiple of the vector width (scalar path) and many-block reductions"""
    from xitorch_amd import kernels as K
    g = torch.Generator().manual_seed(L % 1000 + 3)
    vecs = [torch.randn(L, dtype=torch.float64, generator=g).to(dev, dtype) for _ in range(5)]
    for npairs in (1, 2, 3, 4):
        pairs = [(vecs[i], vecs[i + 1]) for i in range(npairs)]
        pairs[0] = (vecs[0], vecs[0])                       # a norm: both operands the same tensor
        out = K.vec_dots(pairs)
        assert out.dtype == torch.float64 and out.shape == (npairs,)
        for i, (a, b) in enumerate(pairs):
            ref = torch.dot(a.double().cpu(), b.double().cpu()).item()
            scale = (a.double().norm() * b.double().norm()).item() + 1e-300
            assert abs(out[i].item() - ref) <= tol * scale * 4, (npairs, i)
    out2 = K.vec_dots([(vecs[1], vecs[2])])                 # deterministic: same bits on a second launch
    assert torch.equal(out2, K.vec_dots([(vecs[1], vecs[2])]))
    k = 5
    V = torch.randn(1, 8, (L + 7) // 8 * 8, dtype=torch.float64, generator=g).to(dev, dtype)
    coef = torch.randn(k, dtype=torch.float64, generator=g).to(dev, dtype)
    scale = torch.rand(k, dtype=torch.float64, generator=g).to(dev, dtype) + 0.5
    out = torch.empty(L, dtype=dtype, device=dev)
    K.broyden_axpy(out, vecs[0], 0.7, vecs[1], -1.3, V=V[0, :, :L] if L % 8 == 0 else V[0], coef=coef, scale=scale,
                   k=k, gamma=-0.5)
    ref = 0.7 * vecs[0].double() - 1.3 * vecs[1].double() - 0.5 * torch.einsum(
        "n,nl->l", (coef * scale).double(), V[0, :k, :L].double())
    assert (out.double() - ref).abs().max().item() <= tol * 50
    K.broyden_axpy(out, vecs[2], 2.0)                       # plain scaling, no low-rank part
    assert torch.allclose(out, 2.0 * vecs[2])


def test_broyden_one_sync_per_iteration_and_restart_parity(dev):
    """the fused driver: identical iterates to the oracle on a rank-restart case, and the host reads the device once
    per function evaluation (plus the set-up) — not ~8 times per outer iteration like rootsolver.py:96-143"""
    case = dict(kind="tanh", nbatch=3, n=96)
    fcn, y0, (A,) = cases.root_inputs(case)
    kw = dict(alpha=-1.0, max_rank=4, f_tol=1e-9)
    tr, tro = {}, {}
    import xitorch_amd.optimize.native_root as nrm
    calls = {"n": 0}
    orig = nrm._Reduce.dots

    def counting(self, pairs):
        calls["n"] += 1
        return orig(self, pairs)
    nrm._Reduce.dots = counting
    try:
        y = nr.broyden1(fcn, y0.to(dev), (A.to(dev),), trace=tr, **kw)
    finally:
        nrm._Reduce.dots = orig
    yo = oroot.broyden1(fcn, y0, (A,), trace=tro, **kw)
    assert tr["nfev"] == tro["nfev"] and tr["niter"] == tro["niter"]
    assert (y.cpu() - yo).abs().max().item() <= 1e-8
    assert calls["n"] <= tr["nfev"] + 1, (calls["n"], tr["nfev"])
