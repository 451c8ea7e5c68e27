"""oracle.ops — minimal CPU operator stand-ins (TEST INFRASTRUCTURE ONLY).

The reference's solvers touch the operator only through `.mm/.rmm/.shape/
.dtype/.device/.is_hermitian` (xitorch/_core/linop.py:238-345).  These small
classes provide exactly that for the oracle loops, independent of the product
package.
"""
import torch


class _Op:
    def __init__(self, shape, dtype, is_hermitian):
        self.shape = tuple(shape)
        self.dtype = dtype
        self.device = torch.device("cpu")
        self.is_hermitian = is_hermitian
        self.n_apply = 0  # number of mm/rmm calls (for traffic accounting)

    def mm(self, x):
        self.n_apply += 1
        return self._mm(x)

    def rmm(self, x):
        self.n_apply += 1
        if self.is_hermitian:  # linop.py:326-327
            return self._mm(x)
        return self._rmm(x)


class DenseOp(_Op):
    """MatrixLinearOperator (linop.py:676-708): mm = torch.matmul(mat, x)."""

    def __init__(self, mat, is_hermitian=False):
        super().__init__(mat.shape, mat.dtype, is_hermitian)
        self.mat = mat

    def _mm(self, x):
        return torch.matmul(self.mat, x)

    def _rmm(self, x):
        return torch.matmul(self.mat.transpose(-2, -1).conj(), x)

    def fullmatrix(self):
        return self.mat


class BandedOp(_Op):
    """Banded operator in DIA storage: band (*B, 2*hb+1, N), band[..., d, i] = A[i, i + d - hb].

    There is no banded class in the reference; a user would write it as a
    LinearOperator with a `_mv` (cf. the circulant `ALarge` test operator,
    xitorch/_tests/test_linop_fcns.py:129-150).  This is the dense-equivalent
    definition used by configs[2] of BASELINE.json.
    """

    def __init__(self, band, is_hermitian=False):
        nd, n = band.shape[-2:]
        assert nd % 2 == 1
        super().__init__((*band.shape[:-2], n, n), band.dtype, is_hermitian)
        self.band = band
        self.hb = nd // 2

    def _apply(self, x, trans):
        # x: (..., N, c)
        n = self.shape[-1]
        hb = self.hb
        y = torch.zeros(torch.broadcast_shapes(self.band.shape[:-2], x.shape[:-2]) + x.shape[-2:],
                        dtype=x.dtype)
        for d in range(2 * hb + 1):
            off = d - hb
            lo, hi = max(0, -off), min(n, n - off)  # rows i with 0 <= i+off < n
            if hi <= lo:
                continue
            coef = self.band[..., d, lo:hi].unsqueeze(-1)
            if not trans:
                y[..., lo:hi, :] += coef * x[..., lo + off:hi + off, :]
            else:
                y[..., lo + off:hi + off, :] += coef.conj() * x[..., lo:hi, :]
        return y

    def _mm(self, x):
        return self._apply(x, False)

    def _rmm(self, x):
        return self._apply(x, True)

    def fullmatrix(self):
        n = self.shape[-1]
        return self._apply(torch.eye(n, dtype=self.dtype), False)


class FuncOp(_Op):
    """Operator given by callables mm(x) / rmm(x) on (..., N, c) tensors."""

    def __init__(self, shape, dtype, mm, rmm=None, is_hermitian=False):
        super().__init__(shape, dtype, is_hermitian)
        self._mm = mm
        self._rmm = rmm if rmm is not None else mm
