"""Thin typed wrappers over the C ABI (one Python function per entry point).

All tensors are device tensors; "panel" arguments are PANEL-MAJOR: shape
(B, P, N) with unit stride along N (the reference's Fortran-order (B, N, P)
view, xitorch/_utils/tensor.py:21-32, seen as its transpose).
"""
import torch
from xitorch_amd import _capi
from xitorch_amd._capi import ptr, stream_ptr, check, suffix, fn, require_device

__all__ = ["dense_mm"]


def _panel_strides(X):
    # X: (B, P, N) with stride(-1) == 1
    if X.dim() != 3 or (X.shape[-1] > 1 and X.stride(-1) != 1):
        raise _capi.NativeLibraryError("panel must be (B, P, N) with unit stride along N, got shape %s stride %s"
                                       % (tuple(X.shape), X.stride()))
    return X.stride(1), X.stride(0)


_ws_cache = {}


def _workspace(nelem, dtype, device):
    key = (dtype, device)
    w = _ws_cache.get(key)
    if w is None or w.numel() < nelem:
        w = torch.empty(max(nelem, 1), dtype=dtype, device=device)
        _ws_cache[key] = w
    return w


def dense_mm(A, X, out=None, trans=False, rows_hint=0, stagger=1):
    """Y[b,c,:] = A[b] @ X[b,c,:]  (trans=False)   or   A[b]^T @ X[b,c,:]  (trans=True).

    A: (B, M, N) or (M, N) (broadcast over the panel batch), unit stride along N.
    X: (B, P, N) (trans=False) / (B, P, M) (trans=True), panel-major.
    Returns Y: (B, P, M) / (B, P, N), panel-major.
    Replaces torch.matmul in MatrixLinearOperator._mm/_rmm (xitorch/_core/linop.py:695-702).
    """
    require_device(A, "operator matrix")
    require_device(X, "panel")
    if A.dtype != X.dtype:
        raise _capi.NativeLibraryError("dtype mismatch %s vs %s" % (A.dtype, X.dtype))
    B, P = X.shape[0], X.shape[1]
    if A.dim() == 2:
        M, N = A.shape
        lda, sA = A.stride(0), 0
    else:
        if A.shape[0] != B and A.shape[0] != 1:
            raise _capi.NativeLibraryError("operator batch %d does not match panel batch %d" % (A.shape[0], B))
        M, N = A.shape[1], A.shape[2]
        lda, sA = A.stride(1), (A.stride(0) if A.shape[0] != 1 else 0)
    if N > 1 and A.stride(-1) != 1:
        raise _capi.NativeLibraryError("operator matrix must have unit stride along its last dim")
    nin, nout = (M, N) if trans else (N, M)
    if X.shape[2] != nin:
        raise _capi.NativeLibraryError("panel length %d != operator dim %d" % (X.shape[2], nin))
    ldx, sX = _panel_strides(X)
    if out is None:
        out = torch.empty((B, P, nout), dtype=X.dtype, device=X.device)
    ldy, sY = _panel_strides(out)
    ws, ws_n = None, 0
    if trans:
        ws_n = fn("xk_dense_mm_workspace_elems")(B, M, N, P, 1)
        ws = _workspace(ws_n, X.dtype, X.device)
    rc = fn("xk_dense_mm_" + suffix(X.dtype))(
        ptr(A), ptr(X), ptr(out), ptr(ws), ws_n, B, M, N, P, lda, sA, ldx, sX, ldy, sY,
        1 if trans else 0, rows_hint, stagger, stream_ptr())
    check(rc, "xk_dense_mm")
    return out
