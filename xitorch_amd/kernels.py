"""Thin typed wrappers over the C ABI (one Python function per entry point).

All tensors are device tensors; "panel" arguments are PANEL-MAJOR: shape
(B, P, N) with unit stride along N (the reference's Fortran-order (B, N, P)
view, xitorch/_utils/tensor.py:21-32, seen as its transpose).
"""
import os
import torch
from xitorch_amd import _capi
from xitorch_amd._capi import ptr, stream_ptr, check, suffix, fn, require_device
from ctypes import c_void_p as c_void_p_

__all__ = ["dense_mm"]


def _panel_strides(X):
    # X: (B, P, N) with stride(-1) == 1
    if X.dim() != 3 or (X.shape[-1] > 1 and X.stride(-1) != 1):
        raise _capi.NativeLibraryError("panel must be (B, P, N) with unit stride along N, got shape %s stride %s"
                                       % (tuple(X.shape), X.stride()))
    return X.stride(1), X.stride(0)


_ws_cache = {}


def _workspace(nelem, dtype, device):
    # one scratch buffer per (dtype, device, stream): launches on different streams may run concurrently
    key = (dtype, device, torch.cuda.current_stream().cuda_stream)
    w = _ws_cache.get(key)
    if w is None or w.numel() < nelem:
        # geometric growth: the Rayleigh-Ritz workspace of an un-restarted run grows with every iteration, and each
        # new size would be a fresh hipMalloc (milliseconds, and a device-wide stall) in a process that has not cached it
        grown = 0 if w is None else w.numel() + w.numel() // 2
        w = None
        _ws_cache.pop(key, None)
        w = torch.empty(max(nelem, grown, 1), dtype=dtype, device=device)
        _ws_cache[key] = w
    return w


def dense_mm(A, X, out=None, trans=False, rows_hint=0, stagger=1, wide=True):
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
    if not trans and wide and P >= WIDE_MIN_P and _rows_wide_ok(A, X, N, lda, sA, ldx, sX):
        # many columns in the ROW orientation: 64-row sub-tiles turned through LDS, 32 columns per pass (K1wr)
        return dense_rows_wide(A, X, out=out)
    if trans and wide and P >= WIDE_MIN_P and _wide_ok(A, N, lda, sA):
        # many columns: one MFMA pass per 32 columns instead of one VALU pass per 8/16
        for c0 in range(0, P, 32):
            pc = min(32, P - c0)
            dense_wide(A, X[:, c0:c0 + pc], out=out[:, c0:c0 + pc])
        return out
    ws, ws_n = None, 0
    ws_n = fn("xk_dense_mm_workspace_elems")(B, M, N, P, 1 if trans else 0)
    if ws_n > 0:
        ws = _workspace(ws_n, X.dtype, X.device)
    rc = fn("xk_dense_mm_" + suffix(X.dtype))(
        ptr(A), ptr(X), ptr(out), ptr(ws), ws_n, B, M, N, P, lda, sA, ldx, sX, ldy, sY,
        1 if trans else 0, rows_hint, stagger, stream_ptr())
    check(rc, "xk_dense_mm")
    return out


# --------------------------------------------------------------------------- basis kernels
def _cstrides(C, a_dim, c_dim):
    """(batch, a, c) strides of a 3-D coefficient tensor whose `a` index is dim a_dim and `c` index dim c_dim."""
    return C.stride(0), C.stride(a_dim), C.stride(c_dim)


def lincomb(V, C, out, k, P, coef_layout="ac", alpha=1.0, beta=0.0):
    """out[b,c,:] = beta*out[b,c,:] + alpha * sum_{a<k} C[b,a,c] * V[b,a,:]

    V: (B, >=k, Npad) panel-major padded basis; out: (B, >=P, Npad) panel-major.
    C: 3-D coefficients; coef_layout "ac" means C[b,a,c], "ca" means C[b,c,a] (e.g. a Gram block
    straight out of dense_mm).  Replaces the Ritz rotations `V @ Y` (symeig.py:178,181) and the
    projections of the block Gram-Schmidt step that stands in for tallqr (_utils/tensor.py:15-18).
    """
    require_device(V, "basis")
    B, N = V.shape[0], V.shape[2]
    ldv, sV = V.stride(1), V.stride(0)
    ldo, sO = out.stride(1), out.stride(0)
    if coef_layout == "ac":
        sC, sCa, sCc = _cstrides(C, 1, 2)
    else:
        sC, sCa, sCc = _cstrides(C, 2, 1)
    rc = fn("xk_lincomb_" + suffix(V.dtype))(ptr(V), ptr(C), ptr(out), B, k, N, P, ldv, sV, sC, sCa, sCc,
                                              ldo, sO, float(alpha), float(beta), stream_ptr())
    check(rc, "xk_lincomb")
    return out


def group_status(rmax, info, flag, status, orth=None):
    """status (3 doubles on the device) = {max rmax (NaN-propagating), max info, max flag or 0}: one launch instead of
    three reductions and three converting copies (the host reads it once per Davidson step, symeig.py:190-197).
    orth (B,), optional: the guard values of `ritz_guard`; status[4] = their maximum (status needs 5 doubles), orth is
    re-zeroed."""
    if orth is not None and status.numel() < 5:
        raise _capi.NativeLibraryError("group_status: status needs 5 doubles when orth is given")
    rc = fn("xk_group_status_" + suffix(rmax.dtype))(ptr(rmax), ptr(info), ptr(flag) if flag is not None else None,
                                                     ptr(orth) if orth is not None else None,
                                                     ptr(status), rmax.shape[0], stream_ptr())
    check(rc, "xk_group_status")


def ritz_guard(X, orth, P, N, MX=None):
    """orth[b] = max(orth[b], max_{c,d} |<X_c, (M X)_d> - delta_cd|) over the first P rows of the panel-major Ritz
    block X (B, >=P, ld) (MX: the panel M X, default X itself).  The a-posteriori check of a Rayleigh-Ritz block: the
    reference's eigenvectors are orthonormal by construction (whole-basis CholeskyQR every iteration,
    _utils/tensor.py:8-19, symeig.py:207-223); here a block that is not is rolled back by the driver."""
    require_device(X, "Ritz block")
    B = X.shape[0]
    M_ = MX if MX is not None else X
    gs = _guard_scratch(B, P, X.dtype, X.device) if P > 8 else None
    ws, nws = (None, 0)
    if P > 8:
        nws = fn("xk_dense_mm_workspace_elems")(B, P, N, P, 0)
        ws = _workspace(max(nws, 1), X.dtype, X.device)
    rc = fn("xk_ritz_guard_" + suffix(X.dtype))(ptr(X), ptr(M_), ptr(orth), B, N, P, X.stride(1), X.stride(0),
                                                 M_.stride(1), M_.stride(0), ptr(gs), (gs.numel() if gs is not None else 0),
                                                 ptr(ws), nws, stream_ptr())
    check(rc, "xk_ritz_guard")


_gs_cache = {}


def _guard_scratch(B, P, dtype, device):
    key = (dtype, device, torch.cuda.current_stream().cuda_stream)
    g = _gs_cache.get(key)
    if g is None or g.numel() < B * P * P:
        g = _gs_cache[key] = torch.empty(B * P * P, dtype=dtype, device=device)
    return g


def _check_ritz_shapes(V, AV, Y, lam, X, Tn, k, P):
    """the kernels take raw pointers: a coefficient block narrower than P would be read past its end"""
    B = V.shape[0]
    if Y.dim() != 3 or Y.shape[0] != B or Y.shape[1] < k or Y.shape[2] < P or lam.shape[0] != B or lam.shape[-1] < P:
        raise _capi.NativeLibraryError("ritz_residual: Y must be (B, >=k, >=P) and lam (B, >=P); got %s / %s for "
                                       "k = %d, P = %d" % (tuple(Y.shape), tuple(lam.shape), k, P))
    if V.shape[1] < k or AV.shape[1] < k or X.shape[1] < P or Tn.shape[1] < P:
        raise _capi.NativeLibraryError("ritz_residual: basis / output panels are smaller than (k, P) = (%d, %d)" % (k, P))


def ritz_residual(V, AV, Y, lam, X, Tn, rmax, k, P):
    """Fused K4/K5 (symeig.py:178-188): X = Y^T V, AX = Y^T AV, Tn = -(AX - lam X), rmax[b] = max|AX - lam X|.

    Y: (B, k, >=P) eigenvector coefficients (any strides), lam: (B, >=P) with unit stride along P,
    rmax: (B,) zero-initialised by the caller.
    """
    B, N = V.shape[0], V.shape[2]
    if lam.stride(-1) != 1 and P > 1:
        raise _capi.NativeLibraryError("lam must have unit stride along its last dim")
    _check_ritz_shapes(V, AV, Y, lam, X, Tn, k, P)
    rc = fn("xk_ritz_residual_" + suffix(V.dtype))(
        ptr(V), ptr(AV), ptr(Y), ptr(lam), ptr(X), ptr(Tn), ptr(rmax), B, k, N, P,
        V.stride(1), V.stride(0), AV.stride(1), AV.stride(0), Y.stride(0), Y.stride(1), Y.stride(2),
        lam.stride(0), X.stride(1), X.stride(0), Tn.stride(1), Tn.stride(0), stream_ptr())
    check(rc, "xk_ritz_residual")


def diag_precond(Tn, d, lam, P, m=None, floor=None):
    """Davidson's diagonal correction in place: Tn[b,c,:] /= (d[b,:] - lam[b,c] * m[b,:]) with the magnitude of
    the denominator kept above ``floor`` (extension: the reference's davidson takes t = -resid, symeig.py:206-207).
    Tn: (B, >=P, ld) panel, d / m: (B or 1, N), lam: (B, >=P)."""
    B, N = Tn.shape[0], d.shape[-1]
    if floor is None:
        floor = float(torch.finfo(Tn.dtype).eps) ** 0.5
    d2 = d.reshape(-1, N)
    m2 = m.reshape(-1, N) if m is not None else None
    for t in (d2, m2):
        if t is not None and (t.stride(-1) != 1 or t.dtype != Tn.dtype or t.shape[0] not in (1, B)):
            raise _capi.NativeLibraryError("diag_precond: diagonals must be (B or 1, N), unit stride, dtype of the panel")
    sD = d2.stride(0) if d2.shape[0] == B and B > 1 else (0 if d2.shape[0] == 1 else d2.stride(0))
    sM = 0 if m2 is None or m2.shape[0] == 1 else m2.stride(0)
    rc = fn("xk_diag_precond_" + suffix(Tn.dtype))(ptr(Tn), ptr(d2), ptr(m2) if m2 is not None else None,
                                                   ptr(lam), B, N, P, Tn.stride(1), Tn.stride(0), sD, sM,
                                                   lam.stride(0), float(floor), stream_ptr())
    check(rc, "xk_diag_precond")


def panel_chol(G, W, info, P):
    """W[b] = R^-1 with G[b] = R^T R (upper R); info[b] != 0 flags a non-positive pivot.
    G: (B, >=P, >=P) with unit stride along its last dim.  CholeskyQR step of tallqr (tensor.py:16-17)."""
    B = G.shape[0]
    rc = fn("xk_panel_chol_" + suffix(G.dtype))(ptr(G), ptr(W), ptr(info), B, P, G.stride(1), G.stride(0),
                                                 stream_ptr())
    check(rc, "xk_panel_chol")


def panel_transform(Tp, W, P):
    """In place Tp[b,c,:] <- sum_{a<=c} W[b,a,c] Tp[b,a,:]  (tensor.py:18, Q = V R^-1 for the new panel only)."""
    B, N = Tp.shape[0], Tp.shape[2]
    rc = fn("xk_panel_transform_" + suffix(Tp.dtype))(ptr(Tp), ptr(W), B, P, N, Tp.stride(1), Tp.stride(0),
                                                      stream_ptr())
    check(rc, "xk_panel_transform")


# --------------------------------------------------------------------------- Davidson chain (one C call per stage)
def davidson_ritz(V, AV, Y, lam, X, Tn, rmax, info, flag, status, k, P, cond=None, orth=None):
    """ritz_residual + (guard of the Ritz block) + group status in one C call (xk_davidson_ritz): X = Y^T V,
    Tn = -(Y^T AV - lam X), status = {max|resid| (NaN-propagating), max info, max flag[, max cond[, max orth]]}; rmax
    (and cond, orth) are left zeroed for the next step (symeig.py:178-197).  orth (B,): switches the a-posteriori guard
    on (see `ritz_guard`)."""
    B, N = V.shape[0], V.shape[2]
    if lam.stride(-1) != 1 and P > 1:
        raise _capi.NativeLibraryError("lam must have unit stride along its last dim")
    _check_ritz_shapes(V, AV, Y, lam, X, Tn, k, P)
    gs, ws, nws = None, None, 0
    if orth is not None:
        if status.numel() < 5:
            raise _capi.NativeLibraryError("davidson_ritz: status needs 5 doubles when orth is given")
        if P > 8:
            gs = _guard_scratch(B, P, V.dtype, V.device)
            nws = fn("xk_dense_mm_workspace_elems")(B, P, N, P, 0)
            ws = _workspace(max(nws, 1), V.dtype, V.device)
    rc = fn("xk_davidson_ritz_" + suffix(V.dtype))(
        ptr(V), ptr(AV), ptr(Y), ptr(lam), ptr(X), ptr(Tn), ptr(rmax), ptr(info), ptr(flag) if flag is not None else None,
        ptr(cond) if cond is not None else None, ptr(orth) if orth is not None else None, ptr(status), B, k, N, P,
        V.stride(1), V.stride(0), AV.stride(1), AV.stride(0), Y.stride(0), Y.stride(1),
        Y.stride(2), lam.stride(0), X.stride(1), X.stride(0), Tn.stride(1), Tn.stride(0),
        ptr(gs), (gs.numel() if gs is not None else 0), ptr(ws), nws, stream_ptr())
    check(rc, "xk_davidson_ritz")


def davidson_ws(B, cap, N, q, dtype, device):
    """split-contraction workspace of the Gram products inside xk_davidson_orth / _extend_t for bases up to `cap`"""
    nws = fn("xk_dense_mm_workspace_elems")(B, cap, N, q, 0)
    return _workspace(max(nws, 1), dtype, device), nws


def davidson_orth(V, N, k0, q, C, W, info, passes=2, cond=None):
    """Rows [k0, k0+q) of the basis V (B, cap, ld) are orthogonalised against rows [0, k0) (`passes` rounds of block
    Gram-Schmidt) and orthonormalised among themselves (CholeskyQR) — tallqr of [V, t] restricted to the new block
    (_utils/tensor.py:8-19, symeig.py:207-220) — in one C call.  C: scratch of >= B*q*max(k0,q) elements, W: B*q*q.
    passes >= 2: [projection, CholeskyQR] per pass, the first CholeskyQR shifted.  cond (B,), optional, q <= 8: receives
    max(cond, squared pivot ratio of the raw panel)."""
    B = V.shape[0]
    ws, nws = davidson_ws(B, V.shape[1], N, q, V.dtype, V.device)
    rc = fn("xk_davidson_orth_" + suffix(V.dtype))(ptr(V), B, N, k0, q, V.stride(1), V.stride(0), ptr(C), ptr(W),
                                                    ptr(info), ptr(cond) if cond is not None else None, ptr(ws), nws,
                                                    int(passes), stream_ptr())
    check(rc, "xk_davidson_orth")


def davidson_extend_t(V, AV, Tm, Tn, N, k0, q):
    """T[:, k0:k0+q, :k0+q] = <V_a, (AV)_{k0+c}> and the mirrored block T[:, :k0, k0:k0+q] (symeig.py:170 restricted to
    the new rows / columns) in one C call.  Tn: scratch of >= B*q*(k0+q) elements."""
    B = V.shape[0]
    ws, nws = davidson_ws(B, V.shape[1], N, q, V.dtype, V.device)
    rc = fn("xk_davidson_extend_t_" + suffix(V.dtype))(ptr(V), ptr(AV), ptr(Tm), ptr(Tn), B, N, k0, q, V.stride(1),
                                                        V.stride(0), AV.stride(1), AV.stride(0), Tm.stride(1),
                                                        Tm.stride(0), ptr(ws), nws, stream_ptr())
    check(rc, "xk_davidson_extend_t")


# --------------------------------------------------------------------------- K3 small eigensolver
SMALL_EIGH_MAX_K = 128
SMALL_EIGH_MAX_P = 16


SMALL_EIGH_TRI_MIN_K = 16        # below this order the Jacobi kernel's fixed costs are lower


def small_eigh_tri_ok(k, p, dtype):
    """does the tridiagonalisation kernel (K3t) serve order k with p wanted pairs?  (LDS-resident: <= 160 KiB)"""
    if k > SMALL_EIGH_MAX_K or p > SMALL_EIGH_MAX_P or p > k:
        return False
    esize = 8 if dtype == torch.float64 else 4
    return fn("xk_small_eigh_tri_lds_bytes")(k, p, esize) <= 160 * 1024


SMALL_EIGH_BIG_MAX_K = 1536          # (r06: was 1024; fp64 beyond 614 and fp32 where the band does not fit: one launch per
                                     #  Householder step, 24 column slots per lane beyond order 1024)
SMALL_EIGH_BIG_MAX_P = 256          # (r06: was 64)
# launch shape of K3g's step kernels handed to every call (0 = the library's measured defaults).  Module attributes of
# the PYTHON layer, for measurement scripts; the C ABI itself has no state
K3G_WG = 0
K3G_THREADS = 0
K3G_ALGO = int(os.environ.get("XITORCH_K3G_ALGO", "0"))   # (the environment variable is a measurement knob)
                      # 0: the library's choice; -1: the r05 choice; 1: one launch per Householder step; 2: two-stage (band + bulge chasing);
                      # 3: persistent register-resident kernel (r06; step launches first beyond order 256 / 384)


def small_eigh_big_ok(k, p, dtype):
    """does the global-memory tridiagonalisation kernel (K3g) serve order k with p wanted pairs?"""
    if k < 8 or k > SMALL_EIGH_BIG_MAX_K or p > SMALL_EIGH_BIG_MAX_P or p > k:
        return False
    return fn("xk_small_eigh_big_batch")(k, p, 8 if dtype == torch.float64 else 4) > 0


def small_eigh_big(T, k, p, uppest=False, wg=None, threads=None, algo=None):
    """K3g: lowest / uppermost p eigenpairs of the symmetric (B, k, k) matrices T[:, :k, :k] (lower triangle read) for
    orders beyond the LDS-resident kernels (129 .. 1536) or more than 16 wanted pairs (p <= 256, k >= 8): lam (B, p) ascending, Y (B, p, k), failure flags (B,) int32
    (nonzero -> redo with the library).  Replaces torch.linalg.eigh + _take_eigpairs (symeig.py:174-175) on the large
    bases of an un-restarted run."""
    require_device(T, "projected matrix")
    B = T.shape[0]
    if T.stride(2) != 1:
        raise _capi.NativeLibraryError("T must have unit stride along its last dim")
    lam = torch.empty((B, p), dtype=T.dtype, device=T.device)
    Y = torch.empty((B, p, k), dtype=T.dtype, device=T.device)
    info = torch.empty((B,), dtype=torch.int32, device=T.device)
    wg = K3G_WG if wg is None else int(wg)
    threads = K3G_THREADS if threads is None else int(threads)
    nws = fn("xk_small_eigh_big_workspace_elems")(B, k, wg)
    ws = _workspace(nws, T.dtype, T.device)
    algo = K3G_ALGO if algo is None else int(algo)
    if algo < 0:
        # measurement knob: the r05 choice (two-stage from order 192 on where its band fits the LDS, else step launches)
        algo = 2 if k >= 192 and (k <= 614 or T.dtype == torch.float32) else 1
    rc = fn("xk_small_eigh_big_" + suffix(T.dtype))(ptr(T), ptr(lam), ptr(Y), ptr(ws), nws, ptr(info), B, k, p,
                                                     1 if uppest else 0, T.stride(1), T.stride(0), wg, threads,
                                                     algo, stream_ptr())
    check(rc, "xk_small_eigh_big")
    return lam, Y, info


def small_eigh(T, k, p, uppest=False, max_sweeps=16, method="jacobi", threads=0, profile=None):
    """Lowest / uppermost `p` eigenpairs of the symmetric (B, k, k) matrices T[:, :k, :k] (lower
    triangle is read).  Returns lam (B, p) ascending, Y (B, p, k) with Y[b, c] the c-th eigenvector, and an int32
    (B,) tensor: Jacobi sweeps (method "jacobi") or the failure flags of the self-check (method "tri": K3t,
    Householder tridiagonalisation + bisection + inverse iteration; nonzero -> redo that call with "jacobi").
    Native replacement of `torch.linalg.eigh` + `_take_eigpairs` (symeig.py:174-175)."""
    require_device(T, "projected matrix")
    B = T.shape[0]
    if T.stride(2) != 1:
        raise _capi.NativeLibraryError("T must have unit stride along its last dim")
    lam = torch.empty((B, p), dtype=T.dtype, device=T.device)
    Y = torch.empty((B, p, k), dtype=T.dtype, device=T.device)
    aux = torch.empty((B,), dtype=torch.int32, device=T.device)
    if method == "tri":
        rc = fn("xk_small_eigh_tri_" + suffix(T.dtype))(ptr(T), ptr(lam), ptr(Y), ptr(aux), B, k, p, 1 if uppest else 0,
                                                         T.stride(1), T.stride(0), int(threads), ptr(profile),
                                                         stream_ptr())
        check(rc, "xk_small_eigh_tri")
        return lam, Y, aux
    nws = fn("xk_small_eigh_workspace_elems")(B, k, max_sweeps)
    ws = _workspace(nws, T.dtype, T.device)
    rc = fn("xk_small_eigh_" + suffix(T.dtype))(ptr(T), ptr(lam), ptr(Y), ptr(ws), nws, ptr(aux), B, k, p,
                                                 1 if uppest else 0, max_sweeps, T.stride(1), T.stride(0),
                                                 stream_ptr())
    check(rc, "xk_small_eigh")
    return lam, Y, aux


# --------------------------------------------------------------------------- banded operator
def banded_mm(band, X, out=None, trans=False):
    """Y[b,c,:] = A_b X[b,c,:] for DIA-stored banded operators: band (B or 1, 2*hb+1, N),
    band[b,d,i] = A_b[i, i+d-hb].  X, Y panel-major (B, C, N)."""
    require_device(band, "band")
    require_device(X, "panel")
    B, C, N = X.shape
    nd = band.shape[-2]
    if band.shape[-1] != N or nd % 2 != 1:
        raise _capi.NativeLibraryError("band shape %s does not match panel length %d" % (tuple(band.shape), N))
    if not band.is_contiguous():
        raise _capi.NativeLibraryError("band must be contiguous")
    sBand = band.stride(0) if (band.dim() == 3 and band.shape[0] != 1) else 0
    ldx, sX = _panel_strides(X)
    if out is None:
        out = torch.empty((B, C, N), dtype=X.dtype, device=X.device)
    ldy, sY = _panel_strides(out)
    rc = fn("xk_banded_mm_" + suffix(X.dtype))(ptr(band), ptr(X), ptr(out), B, N, nd // 2, C, sBand, ldx, sX,
                                                ldy, sY, 1 if trans else 0, stream_ptr())
    check(rc, "xk_banded_mm")
    return out


def banded_grad(U, W, nd, out=None, accumulate=False):
    """G[b,d,i] (+)= sum_c U[b,c,i] * W[b,c,i+d-hb]  — the DIA band gradient of the banded apply
    (y = A x: U = grad_y, W = x; y = A^T x: U = x, W = grad_y).  U, W panel-major (B, C, N); returns (B, nd, N)."""
    require_device(U, "panel")
    require_device(W, "panel")
    B, C, N = U.shape
    if W.shape != U.shape or U.dtype != W.dtype or nd % 2 != 1:
        raise _capi.NativeLibraryError("banded_grad: panels must agree (got %s / %s), nd odd" % (tuple(U.shape), tuple(W.shape)))
    ldu, sU = _panel_strides(U)
    ldw, sW = _panel_strides(W)
    if out is None:
        out = torch.empty((B, nd, N), dtype=U.dtype, device=U.device)
        accumulate = False
    if not out.is_contiguous():
        raise _capi.NativeLibraryError("banded_grad: output must be contiguous")
    rc = fn("xk_banded_grad_" + suffix(U.dtype))(ptr(U), ptr(W), ptr(out), B, N, nd // 2, C, ldu, sU, ldw, sW,
                                                  out.stride(0), 1 if accumulate else 0, stream_ptr())
    check(rc, "xk_banded_grad")
    return out


def dense_outer(U, W, out=None, accumulate=False):
    """G[b,i,j] (+)= sum_c U[b,c,i] * W[b,c,j]  — the dense-operator gradient (outer product of two panels).
    U (B, C, M), W (B, C, N) panel-major; returns (B, M, N) row-major."""
    require_device(U, "panel")
    require_device(W, "panel")
    B, C, M = U.shape
    N = W.shape[2]
    if W.shape[0] != B or W.shape[1] != C or U.dtype != W.dtype:
        raise _capi.NativeLibraryError("dense_outer: panels must agree (got %s / %s)" % (tuple(U.shape), tuple(W.shape)))
    ldu, sU = _panel_strides(U)
    ldw, sW = _panel_strides(W)
    if out is None:
        out = torch.empty((B, M, N), dtype=U.dtype, device=U.device)
        accumulate = False
    if out.stride(2) != 1 and N > 1:
        raise _capi.NativeLibraryError("dense_outer: output must have unit stride along its last dim")
    rc = fn("xk_dense_outer_" + suffix(U.dtype))(ptr(U), ptr(W), ptr(out), B, M, N, C, ldu, sU, ldw, sW,
                                                  out.stride(1), out.stride(0), 1 if accumulate else 0, stream_ptr())
    check(rc, "xk_dense_outer")
    return out


# --------------------------------------------------------------------------- fused BLAS-1 of the Broyden driver
_vd_scratch = {}


def vec_dots(pairs, out=None):
    """out[i] = <a_i, b_i> for up to 4 pairs of equal-length contiguous device vectors, one streaming pass and a
    deterministic device-side fold.  Returns a float64 device tensor of len(pairs) entries — NO host sync; read it
    with a single `.tolist()` when the values are needed (rootsolver.py:100,113,286-290,375-380 take one sync each)."""
    a0 = pairs[0][0]
    require_device(a0, "vector")
    L = a0.numel()
    np_ = len(pairs)
    if not 1 <= np_ <= 4:
        raise _capi.NativeLibraryError("vec_dots takes 1..4 pairs")
    flat = []
    for (a, b) in pairs:
        if a.numel() != L or b.numel() != L or a.dtype != a0.dtype or b.dtype != a0.dtype or \
                not a.is_contiguous() or not b.is_contiguous():
            raise _capi.NativeLibraryError("vec_dots: vectors must be contiguous, of one length and dtype")
        flat += [a, b]
    while len(flat) < 8:
        flat.append(None)
    key = (a0.device, torch.cuda.current_stream().cuda_stream)
    scratch = _vd_scratch.get(key)
    nws = fn("xk_vec_dots_workspace_elems")()
    if scratch is None:
        scratch = torch.empty(nws, dtype=torch.float64, device=a0.device)
        _vd_scratch[key] = scratch
    if out is None:
        out = torch.empty(np_, dtype=torch.float64, device=a0.device)
    rc = fn("xk_vec_dots_" + suffix(a0.dtype))(*[ptr(t) for t in flat], np_, L, ptr(scratch), nws, ptr(out), stream_ptr())
    check(rc, "xk_vec_dots")
    return out


def broyden_axpy(out, u0=None, g0=0.0, u1=None, g1=0.0, V=None, coef=None, scale=None, k=0, gamma=1.0):
    """out = g0*u0 + g1*u1 + gamma * sum_{n<k} coef[n]*scale[n] * V[n]   (flat length-L vectors; V (>=k, ldv) rows).
    The low-rank apply and the rank-1 update of the Broyden inverse-Jacobian model (_jacobian.py:112-119,172-182)."""
    require_device(out, "vector")
    L = out.numel()
    ldv = V.stride(-2) if (V is not None and k > 0) else 0
    rc = fn("xk_broyden_axpy_" + suffix(out.dtype))(ptr(out), ptr(u0), float(g0), ptr(u1), float(g1),
                                                     ptr(V) if k > 0 else ptr(None), ldv,
                                                     ptr(coef) if k > 0 else ptr(None), ptr(scale), int(k), float(gamma),
                                                     L, stream_ptr())
    check(rc, "xk_broyden_axpy")
    return out


# --------------------------------------------------------------------------- complex operators (real embedding)
def _as_real_matrix(A):
    """zero-copy real view (.., M, 2N) of a complex matrix (.., M, N): row i = (Re A_i0, Im A_i0, Re A_i1, ...)"""
    Ar = torch.view_as_real(A)
    return Ar.reshape(*A.shape[:-1], 2 * A.shape[-1])


def dense_mm_complex(A, X, adjoint=False, conj_io=False, out=None):
    """Complex operator-panel product on the REAL kernels (K1): the interleaved storage of A (B, M, N) complex is
    a real (B, M, 2N) matrix, and

        A x       : Re y = A~ . real(conj x),   Im y = A~ . real(i conj x)      -> xk_dense_mm(trans=0), 2P columns
        A^H x     : Z = A~^T [Re x; Im x];  Re y_j = Z[re, 2j] + Z[im, 2j+1],  Im y_j = Z[im, 2j] - Z[re, 2j+1]
                                                                                 -> xk_dense_mm(trans=1), 2P columns

    so the operator is streamed ONCE per product, exactly like the real case, and never copied or split.
    conj_io conjugates input and output (serves A^T x = conj(A^H conj x) and conj(A) x = conj(A conj x), i.e.
    transposed / conjugated *views* of the stored matrix).  X: (B, P, n_in) complex panel-major; returns
    (B, P, n_out) complex.  Replaces torch.matmul(mat, x) of MatrixLinearOperator for complex dtypes
    (xitorch/_core/linop.py:692-702; reference tests: _tests/test_linop_fcns.py:474-524)."""
    require_device(A, "operator matrix")
    require_device(X, "panel")
    if A.is_conj() or X.is_conj():
        raise _capi.NativeLibraryError("dense_mm_complex takes resolved tensors (conjugate views are expressed "
                                       "through adjoint/conj_io)")
    B, P = X.shape[0], X.shape[1]
    Ar = _as_real_matrix(A)
    M, N = A.shape[-2], A.shape[-1]
    rdtype = Ar.dtype
    if conj_io:
        X = X.conj().resolve_conj()
    if not adjoint:
        xc = X.conj().resolve_conj()
        U = torch.empty((B, 2 * P, 2 * N), dtype=rdtype, device=X.device)
        Uv = U.view(B, 2 * P, N, 2)
        Uv[:, :P, :, 0] = xc.real
        Uv[:, :P, :, 1] = xc.imag          # real(conj x)   = (xr, -xi)
        Uv[:, P:, :, 0] = X.imag
        Uv[:, P:, :, 1] = X.real           # real(i conj x) = (xi,  xr)
        Y = dense_mm(Ar, U, trans=False)                                  # (B, 2P, M)
        res = torch.complex(Y[:, :P], Y[:, P:])
    else:
        U = torch.cat([X.real, X.imag], dim=1).contiguous()                # (B, 2P, M)
        Z = dense_mm(Ar, U, trans=True).view(B, 2 * P, N, 2)               # (B, 2P, 2N)
        res = torch.complex(Z[:, :P, :, 0] + Z[:, P:, :, 1], Z[:, P:, :, 0] - Z[:, :P, :, 1])
    if conj_io:
        res = res.conj().resolve_conj()
    if out is not None:
        out.copy_(res)
        return out
    return res


def dense_outer_complex(U, W):
    """G[b,i,j] = sum_c U[b,c,i] conj(W[b,c,j])  (complex panels) through the real streaming kernel xk_dense_outer:
    in the interleaved storage G~[i, 2j] = sum Ur Wr + Ui Wi, G~[i, 2j+1] = sum Ui Wr - Ur Wi — a real outer product
    with 2C columns.  The operator gradient gy x^H of the complex dense apply."""
    B, C, M = U.shape
    N = W.shape[2]
    Ur = torch.cat([U.real, U.imag], dim=1).contiguous()                   # (B, 2C, M)
    Wr = torch.empty((B, 2 * C, 2 * N), dtype=Ur.dtype, device=U.device)
    Wv = Wr.view(B, 2 * C, N, 2)
    Wv[:, :C, :, 0] = W.real
    Wv[:, :C, :, 1] = -W.imag            # rows paired with Re U:  (Wr, -Wi)
    Wv[:, C:, :, 0] = W.imag
    Wv[:, C:, :, 1] = W.real             # rows paired with Im U:  (Wi,  Wr)
    G = dense_outer(Ur, Wr)                                                # (B, M, 2N) real
    return torch.view_as_complex(G.view(B, M, N, 2))


# --------------------------------------------------------------------------- measurement utility
def stream_read(t):
    """Enqueue one read-only pass over the storage of the contiguous HIP tensor `t` (rows = its last dimension; the
    panel kernels' tile walk, 16 B/lane non-temporal loads, no arithmetic) on the current stream.  Timed by the
    caller: what this very buffer streams at when nothing is computed (bench.py ``roofline.stream_read``)."""
    if not t.is_cuda or not t.is_contiguous() or t.dim() < 1:
        raise _capi.NativeLibraryError("stream_read needs a contiguous HIP tensor")
    pitch = t.shape[-1] * t.element_size()
    if pitch % 16:
        raise _capi.NativeLibraryError("stream_read: the row length must be a multiple of 16 bytes")
    nbytes = t.numel() * t.element_size()
    rc = fn("xk_stream_read")(ptr(t), nbytes, pitch, None, stream_ptr())
    check(rc, "xk_stream_read")
    return nbytes


# --------------------------------------------------------------------------- CU-masked stream
_MASKED_STREAMS = {}
# which bits of the linear CU mask a masked stream gives up (include/xitorch_amd.h): 0 (shipped) = the last `reserve` bits,
# 1 = every (units / reserve)-th bit.  The driver deals the linear mask out over the 8 XCDs bit by bit (measured with
# xk_probe_xcc, profiles/r05_cu_mask_probe.jsonl): the tail pattern takes reserve / 8 units from EVERY XCD (32 reserved: 28
# of 32 left on each); the strided pattern would take them all from one XCD — and a mask that empties an XCD is not
# honoured at all (every unit stays usable: "strided 32 / 64" in profiles/r05_cu_mask_pattern.jsonl are runs WITHOUT a
# reservation: 208.0 ms per configs[1] call against 209.7 with the tail pattern, but 33.8 vs 30.8 ms at 8 operators,
# 59.4 vs 55.9 at 16, and 109.4 vs 105.6 for the configs[4] shard)
CU_MASK_PATTERN = int(os.environ.get("XITORCH_AMD_CU_MASK_PATTERN", "0"))


def masked_stream(device, reserve_cus=64, slot=0, only_reserved=False):
    """A process-lifetime HIP stream on `device` that leaves `reserve_cus` compute units unused
    (hipExtStreamCreateWithCUMask), wrapped as a torch stream.  The HBM-bound panel product runs at full
    speed on 192-224 CUs; the CUs it leaves free serve the latency-bound kernels of the other batch half.
    A stream with a CU mask owns its hardware queue, so `reserve_cus=0` streams (distinct `slot`s) are also how
    the two batch groups get queues that never share a barrier packet — torch's pooled streams are multiplexed
    onto a few hardware queues, and a group whose stream lands behind the other group's "wait for the panel
    product" barrier is serialised with it."""
    import ctypes
    device = torch.device(device)
    # only_reserved (r06): the complement — a stream that may use ONLY the `reserve_cus` units a plain masked stream leaves
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(reserve_cus), int(slot),
           2 if only_reserved else int(CU_MASK_PATTERN))
    if key not in _MASKED_STREAMS:
        out = ctypes.c_void_p()
        rc = fn("xk_stream_create_cu_masked_pattern")(key[0], key[1], key[3], ctypes.byref(out))
        check(rc, "xk_stream_create_cu_masked")
        if not _MASKED_STREAMS:
            import atexit
            atexit.register(_destroy_masked_streams)
        _MASKED_STREAMS[key] = torch.cuda.ExternalStream(out.value, device=torch.device("cuda", key[0]))
        total = torch.cuda.get_device_properties(key[0]).multi_processor_count
        _STREAM_CUS[(key[0], out.value)] = max(1, key[1] if key[3] == 2 else total - key[1])
    return _MASKED_STREAMS[key]


_STREAM_CUS = {}             # (device index, stream handle) -> compute units the stream's mask leaves it


def stream_cus(stream):
    """compute units a stream may use: what its CU mask leaves (streams made by `masked_stream`), else all of them"""
    dev = stream.device.index if stream.device.index is not None else torch.cuda.current_device()
    n = _STREAM_CUS.get((dev, stream.cuda_stream))
    return n if n is not None else torch.cuda.get_device_properties(dev).multi_processor_count


def _destroy_masked_streams():
    """The streams live as long as the process; hand them back before the HIP runtime (and any profiler layered
    on it) tears down — rocprofv3 otherwise crashes at exit on streams it still tracks."""
    for key, st in list(_MASKED_STREAMS.items()):
        try:
            st.synchronize()
            fn("xk_stream_destroy")(c_void_p_(st.cuda_stream))
        except Exception:
            pass
    _MASKED_STREAMS.clear()


# --------------------------------------------------------------------------- K1s symmetric storage
K1S_OPTS = None       # None = the shipped choice per launch (`k1s_auto_opts`); an int = low 16 bits of the `opts` of the K1s
                      # entry points handed to every call (measurement scripts, tests; include/xitorch_amd.h)
K1S_PERSIST = 16      # opts bit 4: resident workgroups taking runs from a queue (two per compute unit of the stream)
K1S_WIDE8 = 32        # opts bit 5 (fp64): 8-wave workgroups on 2048 x 2048 tiles, one per compute unit


def k1s_auto_opts(B, N, dtype, cus, pipelined=False, lda=None):
    """The form of one K1s launch over B operators of order N on `cus` compute units (r05, profiles/r05_k1s_*):
      * resident launch (workgroups take runs from a queue) from two rounds of workgroups on — below that the launch is
        one wave of workgroups either way; inside the eigensolver's two-group pipeline this is what lets the other
        group's launch, on its own stream, move into the slots the tail frees (217.6 -> 211.9 ms per configs[1] call);
      * fp64: 8-wave workgroups on 2048 x 2048 tiles (half the partial-sum bytes written and folded) once the launch has
        enough of them: a tile is 32 MB, so alone on the GPU it needs ~8 per compute unit to beat the 4-wave form
        (64 x 16384^2: 10.91 vs 11.31 ms; 32 x 16384^2: 5.55 vs 5.45), inside the pipeline — tails overlapped — about 2
        (32 operators per launch: 216.7 -> 209.9 ms per call).  The 8-wave form addresses 2048 rows through one buffer
        descriptor: `lda` (elements; N when not given) must keep 2048 * lda * 8 bytes below 2 GiB (xk_symm.hip's
        launch check) — a wider operator keeps the 4-wave form, whose descriptor spans 1024 rows."""
    if B <= 0 or N <= 0:
        return 0
    f64 = dtype == torch.float64
    slab = 1024 if f64 else 2048
    ns = (N + slab - 1) // slab
    runs4 = B * sum(ns - (i * 1024) // slab for i in range((N + 1023) // 1024))
    # (inside the pipeline a launch of 8 operators of order 16384 — 1088 runs, 1.4 ms — is shorter than the other group's
    #  chain: nothing to overlap, the resident form costs 0.5 %; 16 operators: 112.0 -> 108.9 ms per call)
    o = K1S_PERSIST if runs4 >= (8 if pipelined else 4) * cus else 0
    if f64 and (o & K1S_PERSIST):
        n8 = (N + 2047) // 2048
        # (pipelined, 16 operators per launch = 576 tiles on 192-224 units: 108.8 -> 107.6 ms per call)
        fits = 2048 * int(N if lda is None else lda) * 8 <= 0x7fffffe0
        if fits and B * n8 * (n8 + 1) // 2 >= (2 if pipelined else 8) * cus:
            o |= K1S_WIDE8
    return o


def _k1s_opts(stream, opts=None, shape=None, pipelined=False, lda=None):
    """`opts` of one K1s launch on `stream`: the forced flag bits (argument or module) or the shipped choice for the
    launch's `shape` = (B, N, dtype), plus — for the resident form — the workgroup count (bits 16..27) that fills the
    compute units `stream` may use, two workgroups each."""
    o = K1S_OPTS if opts is None else int(opts)
    if o is None:
        o = k1s_auto_opts(shape[0], shape[1], shape[2], stream_cus(stream), pipelined, lda) if shape is not None else 0
    if (o & K1S_PERSIST) and not (o >> 16):
        o |= min(0xfff, 2 * stream_cus(stream)) << 16
    return o


def dense_symm(A, X, out=None, opts=None):
    """Y[b,c,:] = A_b X[b,c,:] for EXACTLY symmetric A (B or 1, N, N): only the upper triangle is read.
    The caller guarantees A == A^T bit for bit.  X, Y panel-major (B, P, N)."""
    require_device(A, "operator matrix")
    require_device(X, "panel")
    B, P, N = X.shape
    if A.dim() == 2:
        lda, sA = A.stride(0), 0
    else:
        lda, sA = A.stride(1), (A.stride(0) if A.shape[0] != 1 else 0)
    if A.shape[-1] != N or A.shape[-2] != N or (N > 1 and A.stride(-1) != 1):
        raise _capi.NativeLibraryError("symmetric operator must be (.., %d, %d) with unit stride" % (N, N))
    ldx, sX = _panel_strides(X)
    if out is None:
        out = torch.empty((B, P, N), dtype=X.dtype, device=X.device)
    ldy, sY = _panel_strides(out)
    esize = 8 if X.dtype == torch.float64 else 4
    nws = fn("xk_dense_symm_workspace_elems")(B, N, P, esize)
    ws = _workspace(nws, X.dtype, X.device)
    rc = fn("xk_dense_symm_" + suffix(X.dtype))(ptr(A), ptr(X), ptr(out), ptr(ws), nws, B, N, P, lda, sA,
                                                 ldx, sX, ldy, sY,
                                                 _k1s_opts(torch.cuda.current_stream(), opts, (B, N, X.dtype), lda=lda),
                                                 stream_ptr())
    check(rc, "xk_dense_symm")
    return out


# --------------------------------------------------------------------------- K1sw symmetric storage, wide panels (MFMA)
SYMM_WIDE_MIN_P, SYMM_WIDE_MAX_P = 9, 16
SYMM_WIDE_MIN_N = 1024        # below this the tiles are too few to fill the chip: K1w / K1s serve
K1SW_OPTS = 9                 # bit 0: workgroup-cooperative forms; + bit 3 (shipped, r06): column part from the load registers, ring of
                              # four blocks, 2 waves per SIMD; + bit 1 (r05, 3 waves per SIMD): s_setprio around the MFMA block; 0: one wave
                              # per tile (include/xitorch_amd.h)
K1SW_PERSIST = 4              # bit 2 (with bit 0): resident workgroups taking super-tiles from a queue, three per compute unit
K1SW_RESIDENT = False         # False (shipped): one workgroup per super-tile.  True / "auto" (inside the two-group pipeline, from
                              # 4 rounds of workgroups): resident launches + one panel stream per group — measured on the
                              # configs[4] shard (profiles/r05_k1sw_pipeline_ab.jsonl): the resident kernel is 3.7 % slower
                              # per launch (4.32 vs 4.17 ms on one stream: issue-bound, 672 workgroups starting in lock step),
                              # two streams win that back (4.20) and the call stays 106.6 -> 109.6 ms: not shipped


def _k1sw_opts(stream, B, N, pipelined=False):
    o = int(K1SW_OPTS)
    if not (o & 1) or (o & 8):
        return o
    cus = stream_cus(stream)
    tr = 1024 if N >= 8192 else (512 if N >= 2048 else 256)
    tiles = B * (N // 512 + 1) * (N // 512 + 2) // 2 * 512 // tr
    want = K1SW_RESIDENT
    if want == "auto":
        want = pipelined and tiles >= 4 * 3 * cus
    if want:
        o |= K1SW_PERSIST | (min(0xfff, 3 * cus) << 16)
    return o


def symm_wide_ok(A, X):
    """does K1sw (upper triangle streamed once, both products on the matrix cores) serve this operator / panel?"""
    N, P = X.shape[2], X.shape[1]
    if A.dtype != torch.float32 or X.dtype != torch.float32 or not (SYMM_WIDE_MIN_P <= P <= SYMM_WIDE_MAX_P):
        return False
    lda = A.stride(-2)
    sA = A.stride(0) if (A.dim() == 3 and A.shape[0] != 1) else 0
    return (N >= SYMM_WIDE_MIN_N and N % 64 == 0 and lda % 4 == 0 and sA % 4 == 0 and A.data_ptr() % 16 == 0
            and X.stride(1) % 2 == 0 and X.stride(0) % 2 == 0 and X.data_ptr() % 8 == 0 and 64 * lda * 4 < 2 ** 31 - 32)


def _symm_wide_args(A, X, out):
    B, P, N = X.shape
    if A.dim() == 2:
        lda, sA = A.stride(0), 0
    else:
        lda, sA = A.stride(1), (A.stride(0) if A.shape[0] != 1 else 0)
    if A.shape[-1] != N or A.shape[-2] != N or (N > 1 and A.stride(-1) != 1):
        raise _capi.NativeLibraryError("symmetric operator must be (.., %d, %d) with unit stride" % (N, N))
    ldx, sX = _panel_strides(X)
    ldy, sY = _panel_strides(out)
    nws = fn("xk_dense_symm_wide_workspace_elems")(B, N)
    return B, P, N, lda, sA, ldx, sX, ldy, sY, nws


def dense_symm_wide(A, X, out=None):
    """Y[b,c,:] = A_b X[b,c,:] for EXACTLY symmetric fp32 A (B or 1, N, N) and 9 .. 16 panel columns: the upper triangle
    is streamed once, both y_I += A_IJ x_J and y_J += A_IJ^T x_I run on the matrix cores (K1sw).  The caller guarantees
    A == A^T bit for bit.  X, Y panel-major (B, P, N)."""
    require_device(A, "operator matrix")
    require_device(X, "panel")
    if out is None:
        out = torch.empty_like(X)
    B, P, N, lda, sA, ldx, sX, ldy, sY, nws = _symm_wide_args(A, X, out)
    ws = _workspace(nws, X.dtype, X.device)
    rc = fn("xk_dense_symm_wide_f32")(ptr(A), ptr(X), ptr(out), ptr(ws), nws, B, N, P, lda, sA, ldx, sX, ldy, sY,
                                      _k1sw_opts(torch.cuda.current_stream(), B, N), stream_ptr())
    check(rc, "xk_dense_symm_wide")
    return out


def dense_symm_wide_split(A, X, out, tiles_stream, timed=False):
    """K1sw with its two launches on two streams, like `dense_symm_split`: the tile kernel on `tiles_stream`, the fold
    back on the current stream.  Returns the timing events around the tile kernel when `timed`."""
    require_device(A, "operator matrix")
    require_device(X, "panel")
    B, P, N, lda, sA, ldx, sX, ldy, sY, nws = _symm_wide_args(A, X, out)
    cur = torch.cuda.current_stream()
    ws = _workspace(nws, X.dtype, X.device)                 # keyed by the CURRENT (group) stream
    ready, done = sync_events(cur)
    ready.record(cur)
    e0 = e1 = None
    wopts = _k1sw_opts(tiles_stream, B, N, pipelined=True)
    with torch.cuda.stream(tiles_stream):
        tiles_stream.wait_event(ready)
        if timed:
            e0, e1 = timing_event_pair()
            e0.record(tiles_stream)
        rc = fn("xk_dense_symm_wide_tiles_f32")(ptr(A), ptr(X), ptr(ws), nws, B, N, P, lda, sA, ldx, sX, wopts,
                                                stream_ptr())
        check(rc, "xk_dense_symm_wide_tiles")
        if timed:
            e1.record(tiles_stream)
        done.record(tiles_stream)
    cur.wait_event(done)
    rc = fn("xk_dense_symm_wide_fold_f32")(ptr(out), ptr(ws), nws, B, N, P, ldy, sY, wopts, stream_ptr())
    check(rc, "xk_dense_symm_wide_fold")
    return e0, e1


# --------------------------------------------------------------------------- events without per-launch creation
# Creating a HIP event costs host time (and every few hundred of them the runtime grows a pool: a ~30 ms stall seen
# once per process in the benchmark's timed region).  Cross-stream ordering therefore re-records two cached events per
# issuing stream, and timing events come from a pool that measurement code can pre-fill.
_SYNC_EVENTS = {}
_TIMING_POOL = {}            # device index -> pre-created timing events (events belong to the device they were made on)


def sync_events(stream):
    """(ready, done): two cached non-timing events owned by `stream` (the stream that issues the work).  A wait
    captures the record that precedes it, so re-recording them launch after launch is safe."""
    key = (stream.device.index, stream.cuda_stream)
    ev = _SYNC_EVENTS.get(key)
    if ev is None:
        ev = _SYNC_EVENTS[key] = (torch.cuda.Event(), torch.cuda.Event())
    return ev


def timing_event_pair():
    pool = _TIMING_POOL.get(torch.cuda.current_device())
    if pool is not None and len(pool) >= 2:
        return pool.pop(), pool.pop()
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def prefill_timing_events(n):
    """Create (and record once, which is what instantiates them) `n` timing events ahead of a timed region."""
    st = torch.cuda.current_stream()
    fresh = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for e in fresh:
        e.record(st)
    st.synchronize()
    _TIMING_POOL.setdefault(torch.cuda.current_device(), []).extend(fresh)


def dense_symm_split(A, X, out, tiles_stream, timed=False):
    """K1s with its two launches on two streams (P <= 6): the tile kernel on `tiles_stream` (after everything
    queued so far on the current stream), the fold back on the current stream once the tiles are done.  The
    partial-sum workspace belongs to the current stream, so callers on different streams never share one.
    Returns the (start, end) timing events recorded around the tile kernel when `timed`, else (None, None)."""
    require_device(A, "operator matrix")
    require_device(X, "panel")
    B, P, N = X.shape
    if P > 6:
        raise _capi.NativeLibraryError("dense_symm_split serves panels of at most 6 columns")
    if A.dim() == 2:
        lda, sA = A.stride(0), 0
    else:
        lda, sA = A.stride(1), (A.stride(0) if A.shape[0] != 1 else 0)
    ldx, sX = _panel_strides(X)
    ldy, sY = _panel_strides(out)
    esize = 8 if X.dtype == torch.float64 else 4
    nws = fn("xk_dense_symm_workspace_elems")(B, N, P, esize)
    cur = torch.cuda.current_stream()
    ws = _workspace(nws, X.dtype, X.device)                 # keyed by the CURRENT (group) stream
    ready, done = sync_events(cur)                          # re-recorded on every launch: no event is created here
    ready.record(cur)
    sfx = suffix(X.dtype)
    kopts = _k1s_opts(tiles_stream, None, (B, N, X.dtype), pipelined=True, lda=lda)
    e0 = e1 = None
    with torch.cuda.stream(tiles_stream):
        tiles_stream.wait_event(ready)
        if timed:
            e0, e1 = timing_event_pair()
            e0.record(tiles_stream)
        rc = fn("xk_dense_symm_tiles_" + sfx)(ptr(A), ptr(X), ptr(ws), nws, B, N, P, lda, sA, ldx, sX,
                                              kopts, stream_ptr())
        check(rc, "xk_dense_symm_tiles")
        if timed:
            e1.record(tiles_stream)
        done.record(tiles_stream)
    cur.wait_event(done)
    rc = fn("xk_dense_symm_fold_" + sfx)(ptr(out), ptr(ws), nws, B, N, P, ldy, sY, kopts, stream_ptr())
    check(rc, "xk_dense_symm_fold")
    return e0, e1


# --------------------------------------------------------------------------- K1w wide panels (MFMA)
WIDE_MIN_P = 12


def _rows_wide_ok(A, X, N, lda, sA, ldx, sX):
    vn = 2 if A.dtype == torch.float64 else 4
    return N % vn == 0 and lda % vn == 0 and sA % vn == 0 and ldx % vn == 0 and sX % vn == 0 and \
        A.data_ptr() % 16 == 0 and X.data_ptr() % 16 == 0


def dense_rows_wide(A, X, out=None):
    """Y[b,c,:] = A_b X[b,c,:] (row orientation) for any number of panel columns, 32 per pass over A (K1wr: LDS
    tile transpose, lane <-> row, scalar panel loads; no transposed copy of the operator).
    A (B or 1, M, N); X panel-major (B, P, N); returns panel-major (B, P, M)."""
    require_device(A, "operator matrix")
    require_device(X, "panel")
    B, P, N = X.shape
    if A.dim() == 2:
        M = A.shape[0]
        lda, sA = A.stride(0), 0
    else:
        M = A.shape[1]
        lda, sA = A.stride(1), (A.stride(0) if A.shape[0] != 1 else 0)
    if A.shape[-1] != N:
        raise _capi.NativeLibraryError("panel length %d != operator columns %d" % (N, A.shape[-1]))
    ldx, sX = _panel_strides(X)
    if out is None:
        out = torch.empty((B, P, M), dtype=X.dtype, device=X.device)
    ldy, sY = _panel_strides(out)
    esize = 8 if X.dtype == torch.float64 else 4
    nws = fn("xk_dense_rows_wide_workspace_elems")(B, M, N, P, esize)
    ws = _workspace(nws, X.dtype, X.device) if nws > 0 else None
    rc = fn("xk_dense_rows_wide_" + suffix(X.dtype))(ptr(A), ptr(X), ptr(out), ptr(ws), nws, B, M, N, P, lda, sA,
                                                      ldx, sX, ldy, sY, stream_ptr())
    check(rc, "xk_dense_rows_wide")
    return out


def _wide_ok(A, N, lda, sA):
    vn = 2 if A.dtype == torch.float64 else 4
    wcols = 32 if A.dtype == torch.float64 else 128
    return N % wcols == 0 and lda % vn == 0 and sA % vn == 0 and A.data_ptr() % 16 == 0


def dense_wide(A, X, out=None):
    """Y[b,c,:] = A_b^T X[b,c,:] for up to 32 panel columns in ONE pass over A (MFMA).
    A (B or 1, M, N); X panel-major (B, P, M); returns panel-major (B, P, N)."""
    require_device(A, "operator matrix")
    require_device(X, "panel")
    B, P, M = X.shape
    if A.dim() == 2:
        lda, sA = A.stride(0), 0
        N = A.shape[1]
    else:
        lda, sA = A.stride(1), (A.stride(0) if A.shape[0] != 1 else 0)
        N = A.shape[2]
    if A.shape[-2] != M:
        raise _capi.NativeLibraryError("panel length %d != operator rows %d" % (M, A.shape[-2]))
    esize = 8 if X.dtype == torch.float64 else 4
    PP = fn("xk_dense_wide_padded_width")(P, esize)
    Xrm = torch.zeros((B, M, PP), dtype=X.dtype, device=X.device)      # row-major, zero-padded to whole tiles
    Xrm[:, :, :P].copy_(X.transpose(1, 2))
    if out is None:
        out = torch.empty((B, P, N), dtype=X.dtype, device=X.device)
    ldy, sY = _panel_strides(out)
    nws = fn("xk_dense_wide_workspace_elems")(B, M, N, P, esize)
    ws = _workspace(nws, X.dtype, X.device)
    rc = fn("xk_dense_wide_" + suffix(X.dtype))(ptr(A), ptr(Xrm), ptr(out), ptr(ws), nws, B, M, N, P, lda, sA,
                                                 Xrm.stride(1), Xrm.stride(0), ldy, sY, stream_ptr())
    check(rc, "xk_dense_wide")
    return out
