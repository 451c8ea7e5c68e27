"""-m "not gpu": the reference's own front-end cases for symeig / lsymeig / svd on the dense methods, real and complex,
with first- and second-order gradient checks — xitorch/_tests/test_linop_fcns.py:52-127 (test_lsymeig_A / _AM),
:178-292 (degenerate eigenvalues, with and without M), :294-342 (the degeneracy requirement warning), :345-380 (svd).
Same constructions and seeds as there; these run on CPU through linalg/symeig.py (the a6 row of DESIGN.md §0)."""
import warnings
import pytest
import torch
from torch.autograd import gradcheck, gradgradcheck
import xitorch_amd as xa
from xitorch_amd import LinearOperator
from xitorch_amd.linalg import symeig, lsymeig, svd
from xitorch_amd.debug import enable_debug
from xitorch_amd._util import MathWarning

SEED = 12345
DTYPES = [torch.float64, torch.complex128]
METHODS = ["exacteig", "custom_exacteig"]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("shape", [(4, 4), (2, 3, 4, 4)])
def test_lsymeig_A(dtype, method, shape):
    torch.manual_seed(SEED)
    mat1 = torch.rand(shape, dtype=dtype)
    mat1 = (mat1 + mat1.transpose(-2, -1).conj()).requires_grad_()
    linop1 = LinearOperator.m(mat1, True)
    for neig in [2, shape[-1]]:
        eigvals, eigvecs = lsymeig(linop1, neig=neig, method=method)
        assert list(eigvecs.shape) == [*linop1.shape[:-1], neig]
        assert list(eigvals.shape) == [*linop1.shape[:-2], neig]
        ax = linop1.mm(eigvecs)
        xe = torch.matmul(eigvecs, torch.diag_embed(eigvals.to(eigvecs.dtype), dim1=-2, dim2=-1))
        assert torch.allclose(ax, xe)
    if len(shape) == 2:         # gradient checks on the small case only (cost)
        neig = shape[-1]

        def fcn(amat):
            amat = (amat + amat.transpose(-2, -1).conj()) * 0.5
            ev, evec = lsymeig(LinearOperator.m(amat, is_hermitian=True), neig=neig, method=method)
            return ev, evec.abs()
        gradcheck(fcn, (mat1,))
        gradgradcheck(fcn, (mat1,))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("ashape,mshape", [((3, 3), (3, 3)), ((2, 3, 3), (2, 1, 3, 3))])
def test_lsymeig_AM(dtype, method, ashape, mshape):
    torch.manual_seed(SEED)
    mata = torch.rand(ashape, dtype=dtype)
    matm = torch.rand(mshape, dtype=dtype) + torch.eye(mshape[-1], dtype=dtype)
    mata = (mata + mata.transpose(-2, -1).conj()).requires_grad_()
    matm = (matm + matm.transpose(-2, -1).conj()).requires_grad_()
    linopa = LinearOperator.m(mata, is_hermitian=True)
    linopm = LinearOperator.m(matm, is_hermitian=True)
    na = ashape[-1]
    bshape = list(torch.broadcast_shapes(ashape[:-2], mshape[:-2]))
    for neig in [2, na]:
        eigvals, eigvecs = lsymeig(linopa, M=linopm, neig=neig, method=method)
        assert list(eigvals.shape) == [*bshape, neig]
        assert list(eigvecs.shape) == [*bshape, na, neig]
        ax = linopa.mm(eigvecs)
        mxe = linopm.mm(torch.matmul(eigvecs, torch.diag_embed(eigvals.to(eigvecs.dtype), dim1=-2, dim2=-1)))
        assert torch.allclose(ax, mxe)
    if len(ashape) == 2:

        def fcn(amat, mmat):
            amat = (amat + amat.transpose(-2, -1).conj()) * 0.5
            mmat = (mmat + mmat.transpose(-2, -1).conj()) * 0.5
            ev, evec = lsymeig(LinearOperator.m(amat, is_hermitian=True), M=LinearOperator.m(mmat, is_hermitian=True),
                               neig=na, method=method)
            return ev, evec.abs()
        gradcheck(fcn, (mata, matm))
        gradgradcheck(fcn, (mata, matm))


def _levels(dtype, shift=0.0):
    """three distinct levels; the operators below repeat the middle and the top one (spectrum 1, 2, 2, 3, 3 + shift)"""
    torch.manual_seed(SEED)
    lv = torch.tensor([1.0, 2.0, 3.0], dtype=dtype) + shift
    return (lv.real if torch.is_complex(lv) else lv).requires_grad_()


def _operator_with_repeated_levels(levels, frame_src, dtype):
    """Q^H diag(l0, l1, l1, l2, l2) Q with Q from the QR factorisation of `frame_src`"""
    Q, _ = torch.linalg.qr(frame_src)
    spectrum = torch.cat((levels[:2], levels[1:2], levels[2:], levels[2:])).to(dtype)
    return Q.transpose(-2, -1).conj() @ torch.diag_embed(spectrum) @ Q


def _subspace_functional(W, U):
    """depends on span(U) only through U U^H: invariant under rotations inside a degenerate pair"""
    return torch.einsum("rc,rc->", W @ U, U.conj())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("eivaloffset", [0, -4])
def test_symeig_A_degenerate(dtype, method, eivaloffset):
    """the gradient propagates stably when the loss does not depend on which degenerate eigenvectors were picked"""
    n = 5
    levels = _levels(dtype, eivaloffset)
    frame = torch.randn((n, n), dtype=dtype).requires_grad_()
    W = torch.randn((n, n), dtype=dtype).requires_grad_()

    def loss(levels, frame, W):
        A = _operator_with_repeated_levels(levels, frame, dtype)
        _, vecs = symeig(LinearOperator.m(A, is_hermitian=True), neig=3, method=method,
                         bck_options={"method": "exactsolve"})
        return _subspace_functional(W, vecs[:, 1:3])          # the degenerate pair
    gradcheck(loss, (levels, frame, W))
    gradgradcheck(loss, (levels, frame, W))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("method", METHODS)
def test_symeig_AM_degenerate(dtype, method):
    n = 5
    levels = _levels(dtype)
    frame = torch.randn((n, n), dtype=dtype)
    overlap_src = torch.rand((n, n), dtype=dtype)
    W = torch.randn((n, n), dtype=dtype).requires_grad_()

    def loss(levels, frame, overlap_src, W):
        A = _operator_with_repeated_levels(levels, frame, dtype)
        R, _ = torch.linalg.qr(overlap_src)
        M = R.transpose(-2, -1).conj() @ R
        _, vecs = symeig(LinearOperator.m(A, is_hermitian=True), M=LinearOperator.m(M, is_hermitian=True), neig=3,
                         method=method, bck_options={"method": "exactsolve"})
        return _subspace_functional(W, vecs[:, 1:3])
    gradcheck(loss, (levels, frame, overlap_src, W))
    gradgradcheck(loss, (levels, frame, overlap_src, W))


def test_symeig_A_degenerate_requirement_not_satisfied_warns():
    """a loss that DOES depend on the choice inside the degenerate subspace: one MathWarning in debug mode"""
    n = 5
    levels = _levels(torch.float64)
    frame = torch.randn((n, n), dtype=torch.float64).requires_grad_()

    def loss(levels, frame):
        A = _operator_with_repeated_levels(levels, frame, torch.float64)
        _, vecs = symeig(LinearOperator.m(A), neig=3, method="custom_exacteig", bck_options={"method": "exactsolve"})
        return torch.sum(vecs[:, :3] ** 4)                     # columns 1, 2 span the degenerate pair
    with warnings.catch_warnings(record=True) as caught, enable_debug():
        warnings.simplefilter("always")
        loss(levels, frame).backward()
    caught = [x for x in caught if x.category is MathWarning]
    assert len(caught) == 1
    msg = str(caught[0].message).lower()
    assert "degener" in msg and "loss function" in msg and "incorrect" in msg


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("shape", [(4, 3), (2, 1, 3, 4)])
def test_svd_A(dtype, method, shape):
    torch.manual_seed(SEED)
    mat1 = torch.rand(shape, dtype=dtype).requires_grad_()
    linop1 = LinearOperator.m(mat1, is_hermitian=False)
    min_mn = min(shape[-1], shape[-2])
    for k in [min_mn]:
        u, s, vh = svd(linop1, k=k, method=method)
        assert list(u.shape) == [*shape[:-1], k]
        assert list(s.shape) == [*shape[:-2], k]
        assert list(vh.shape) == [*shape[:-2], k, shape[-1]]
        keye = torch.eye(k, dtype=dtype).expand(*shape[:-2], k, k)
        assert torch.allclose(u.transpose(-2, -1).conj() @ u, keye)
        assert torch.allclose(vh @ vh.transpose(-2, -1).conj(), keye)
        assert torch.allclose(mat1, u @ torch.diag_embed(s.to(u.dtype)) @ vh)
    if len(shape) == 2:

        def fcn(amat):
            u, s, vh = svd(LinearOperator.m(amat, is_hermitian=False), k=min_mn, method=method)
            return u.abs(), s, vh.abs()
        gradcheck(fcn, (mat1,))
        gradgradcheck(fcn, (mat1,))
