"""-m gpu: the dense method `exacteig` (the reference's DEFAULT for symeig, symeig.py:118-124, _impls/linalg/symeig.py:11-44)
on the native HIP eigensolvers (r04): the reference's own benchmark shapes — n in {100, 350, 700}, neig = 10
(benchmarks/benchmarks_solve.py:37-59) — both ends of the spectrum, with an overlap operator, batched; against golden
outputs of the reference, without any `torch.linalg.eigh` call in the forward pass; gradients (first and second order)
against the library path."""
import os
import numpy as np
import pytest
import torch
import xitorch_amd as xa
from xitorch_amd.linalg import symeig
from xitorch_amd.linalg import native_eig
from tests import cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("case", cases.EXACTEIG_CASES, ids=[c["name"] for c in cases.EXACTEIG_CASES])
@pytest.mark.parametrize("method", [None, "exacteig"])
def test_native_exacteig_vs_reference_golden(dev, case, method, monkeypatch):
    gold = np.load(os.path.join(GOLD, "exacteig_%s.npz" % case["name"]))
    A, M = cases.exacteig_inputs(case)
    Aop = xa.LinearOperator.m(A.to(dev), is_hermitian=True)
    Mop = xa.LinearOperator.m(M.to(dev), is_hermitian=True) if M is not None else None

    def no_library(*a, **k):
        raise AssertionError("torch.linalg.eigh called: the native dense eigensolver must serve this shape")
    monkeypatch.setattr(torch.linalg, "eigh", no_library)
    with torch.no_grad():
        ev, X = symeig(Aop, neig=case["neig"], mode=case["mode"], M=Mop, method=method)
    monkeypatch.undo()
    assert ev.shape == gold["evals"].shape and X.shape == gold["X"].shape
    ev, X = ev.cpu(), X.cpu()
    # eigenvalues: 1e-10 (north_star tolerance), ascending in both modes (index ordering)
    assert torch.all(ev[..., 1:] >= ev[..., :-1])
    assert np.abs(ev.numpy() - gold["evals"]).max() <= 1e-10
    MX = torch.matmul(M, X) if M is not None else X
    assert (torch.matmul(A, X) - MX * ev.unsqueeze(-2)).abs().max().item() <= 1e-10
    G = torch.matmul(X.transpose(-2, -1), MX)
    assert (G - torch.eye(G.shape[-1], dtype=G.dtype)).abs().max().item() <= 1e-10
    # same invariant subspace as the reference's eigenvectors (signs are free, quirk Q15)
    sig = torch.linalg.svdvals(torch.matmul(torch.from_numpy(gold["X"]).transpose(-2, -1), MX))
    assert sig.min().item() >= 1 - 1e-8 and sig.max().item() <= 1 + 1e-8


@pytest.mark.parametrize("dtype,n,p,mode", [(torch.float32, 300, 10, "lowest"), (torch.float64, 768, 64, "uppest"),
                                            (torch.float64, 64, 16, "lowest"), (torch.float64, 9, 3, "uppest")])
def test_native_partial_eigh_sizes_and_dtypes(dev, dtype, n, p, mode):
    g = torch.Generator().manual_seed(n)
    A = torch.randn(2, n, n, dtype=torch.float64, generator=g)
    A = ((A + A.transpose(1, 2)) * 0.5).to(dtype).to(dev)
    lam, X = native_eig.native_partial_eigh(A, p, mode)
    ref = torch.linalg.eigvalsh(A.double())
    want = ref[:, :p] if mode == "lowest" else ref[:, -p:]
    tol = 1e-11 * n if dtype == torch.float64 else 2e-4
    assert (lam.double() - want).abs().max().item() <= tol
    R = A.double() @ X.double() - X.double() * lam.double().unsqueeze(-2)
    assert R.abs().max().item() <= tol * 10


def test_native_exacteig_falls_back_where_it_must(dev, monkeypatch):
    # orders beyond 1024, complex Hermitian and CPU tensors take torch.linalg.eigh like the reference; fp64 orders
    # 769 .. 1024 are native since r05 (no library call in the forward pass)
    g = torch.Generator().manual_seed(1)
    A = torch.randn(1100, 1100, dtype=torch.float64, generator=g)
    A = (A + A.T) * 0.5
    ev, _ = symeig(xa.LinearOperator.m(A.to(dev), is_hermitian=True), neig=3)
    assert (ev.cpu() - torch.linalg.eigvalsh(A)[:3]).abs().max().item() <= 1e-10
    for n in (800, 1000):
        calls = []
        real_eigh = torch.linalg.eigh
        monkeypatch.setattr(torch.linalg, "eigh", lambda *a, **k: (calls.append(1), real_eigh(*a, **k))[1])
        with torch.no_grad():
            ev, X = symeig(xa.LinearOperator.m(A[:n, :n].contiguous().to(dev), is_hermitian=True), neig=3)
        monkeypatch.undo()
        assert not calls, "torch.linalg.eigh called at order %d" % n
        assert (ev.cpu() - torch.linalg.eigvalsh(A[:n, :n])[:3]).abs().max().item() <= 1e-10
        R = A[:n, :n].to(dev) @ X - X * ev.unsqueeze(-2)
        assert R.abs().max().item() <= 1e-9
    C = torch.randn(40, 40, dtype=torch.complex128, generator=g)
    C = (C + C.conj().T) * 0.5
    ev, _ = symeig(xa.LinearOperator.m(C.to(dev), is_hermitian=True), neig=3)
    assert (ev.cpu() - torch.linalg.eigvalsh(C)[:3]).abs().max().item() <= 1e-10
    ev, _ = symeig(xa.LinearOperator.m(A[:50, :50].contiguous(), is_hermitian=True), neig=3)
    assert not ev.is_cuda


def test_native_exacteig_gradients_match_the_library_path(dev, monkeypatch):
    # first and second derivatives through the native forward (its backward completes the eigenbasis with the
    # differentiable library eigh) against the all-library path = the reference's degen_symeig
    n, p = 24, 4
    g = torch.Generator().manual_seed(7)
    base = torch.randn(n, n, dtype=torch.float64, generator=g)
    base = ((base + base.T) * 0.5 + torch.diag(torch.arange(n, dtype=torch.float64))).to(dev)
    w = torch.randn(n, p, dtype=torch.float64, generator=g).to(dev)

    def loss(mat, mode):
        ev, X = symeig(xa.LinearOperator.m((mat + mat.T) * 0.5, is_hermitian=True), neig=p, mode=mode)
        return (ev * torch.arange(1, p + 1, device=dev)).sum() + ((X * X) * w).sum()      # sign-invariant in X

    for mode in ("lowest", "uppest"):
        grads = {}
        for path in ("native", "library"):
            if path == "library":
                monkeypatch.setattr(native_eig, "_native_dense_ok", lambda *a, **k: False)
            m = base.clone().requires_grad_()
            g1, = torch.autograd.grad(loss(m, mode), (m,), create_graph=True)
            g2, = torch.autograd.grad((g1 * g1).sum(), (m,))
            grads[path] = (g1.detach(), g2)
            monkeypatch.undo()
        assert (grads["native"][0] - grads["library"][0]).abs().max().item() <= 1e-9
        assert (grads["native"][1] - grads["library"][1]).abs().max().item() <= 1e-7 * max(1.0, grads["library"][1].abs().max().item())
