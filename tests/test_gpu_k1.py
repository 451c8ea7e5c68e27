"""-m gpu: the K1 HIP kernel (through the C ABI) against the oracle's dense operator on the same inputs."""
import pytest
import torch
from oracle import ops as oops
from xitorch_amd import kernels as K
from xitorch_amd import LinearOperator

pytestmark = pytest.mark.gpu

CASES = [
    # B, M, N, P, dtype, trans
    (2, 256, 256, 6, torch.float64, False), (3, 100, 130, 1, torch.float64, False),
    (2, 77, 201, 5, torch.float64, False), (1, 512, 512, 11, torch.float64, False),
    (2, 1000, 1000, 2, torch.float64, False), (2, 256, 384, 6, torch.float32, False),
    (2, 99, 131, 3, torch.float32, False), (2, 256, 384, 6, torch.float64, True),
    (3, 100, 131, 4, torch.float64, True), (2, 300, 200, 7, torch.float32, True),
    (1, 2048, 2048, 6, torch.float64, True), (4, 1, 64, 2, torch.float64, False),
    (1, 64, 2, 3, torch.float64, True),
]


@pytest.mark.parametrize("B,M,N,P,dtype,trans", CASES)
def test_dense_mm_vs_oracle(dev, B, M, N, P, dtype, trans):
    g = torch.Generator().manual_seed(B * 1000 + M + N + P)
    A = torch.randn(B, M, N, dtype=dtype, generator=g)
    X = torch.randn(B, P, M if trans else N, dtype=dtype, generator=g)
    op = oops.DenseOp(A.double())
    xo = X.double().transpose(-2, -1)
    ref = (op._rmm(xo) if trans else op._mm(xo)).transpose(-2, -1)         # oracle, fp64
    Y = K.dense_mm(A.to(dev), X.to(dev), trans=trans).cpu().double()
    tol = 1e-13 if dtype == torch.float64 else 3e-6        # fp64: summation-order differences only
    scale = ref.abs().max().item() + 1e-300
    assert (Y - ref).abs().max().item() / scale < tol * max(1.0, (N if not trans else M) ** 0.5)


def test_operator_surface_uses_native_kernel(dev):
    # LinearOperator.m(mat).mm/.mv/.rmm with broadcasting + extra leading dims, vs torch on CPU
    g = torch.Generator().manual_seed(5)
    mat = torch.randn(3, 40, 56, dtype=torch.float64, generator=g)
    op = LinearOperator.m(mat.to(dev))
    for xshape in [(56, 4), (3, 56, 2), (2, 3, 56, 5), (1, 56, 3)]:
        x = torch.randn(*xshape, dtype=torch.float64, generator=g)
        y = op.mm(x.to(dev)).cpu()
        assert torch.allclose(y, torch.matmul(mat, x), rtol=1e-12, atol=1e-12), xshape
    for xshape in [(40, 4), (3, 40, 2), (2, 3, 40, 5)]:
        x = torch.randn(*xshape, dtype=torch.float64, generator=g)
        y = op.rmm(x.to(dev)).cpu()
        assert torch.allclose(y, torch.matmul(mat.transpose(-2, -1), x), rtol=1e-12, atol=1e-12), xshape
    v = torch.randn(3, 56, dtype=torch.float64, generator=g)
    assert torch.allclose(op.mv(v.to(dev)).cpu(), torch.matmul(mat, v.unsqueeze(-1)).squeeze(-1), rtol=1e-12, atol=1e-12)
    # unbatched operator, batched operand -> batch folds into the panel width
    op1 = LinearOperator.m(mat[0].to(dev))
    x = torch.randn(5, 56, 3, dtype=torch.float64, generator=g)
    assert torch.allclose(op1.mm(x.to(dev)).cpu(), torch.matmul(mat[0], x), rtol=1e-12, atol=1e-12)


def test_dense_mm_autograd(dev):
    g = torch.Generator().manual_seed(6)
    mat = torch.randn(2, 12, 12, dtype=torch.float64, generator=g).to(dev).requires_grad_()
    x = torch.randn(2, 12, 3, dtype=torch.float64, generator=g).to(dev).requires_grad_()

    def f(m, xx):
        return LinearOperator.m(m, is_hermitian=False).mm(xx)
    assert torch.autograd.gradcheck(f, (mat, x))
    assert torch.autograd.gradgradcheck(f, (mat, x))
