"""oracle.symeig — CPU restatement of the reference block-Davidson eigensolver.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
xitorch/_impls/linalg/symeig.py:100-264 and xitorch/_utils/tensor.py:8-32 op by
op, so that on CPU it reproduces the reference bit-for-bit.
"""
import torch


def tallqr(V, MV=None):
    """CholeskyQR of a tall panel (reference: _utils/tensor.py:8-19).

    G = V^T (M V);  R = chol(G^H)^H;  Q = V R^-1.  Returns (Q, R).
    """
    if MV is None:
        MV = V
    G = torch.matmul(V.transpose(-2, -1), MV)
    R = torch.linalg.cholesky(G.transpose(-2, -1).conj()).transpose(-2, -1).conj()
    Q = torch.matmul(V, torch.inverse(R))
    return Q, R


def to_fortran_order(V):
    """Column-major copy of the trailing 2 dims (reference: _utils/tensor.py:21-32)."""
    if V.is_contiguous():
        return V.transpose(-2, -1).contiguous().transpose(-2, -1)
    if V.transpose(-2, -1).is_contiguous():
        return V
    raise RuntimeError("Only the last two dimensions can be made Fortran order.")


def initial_v(kind, dtype, batch_dims, na, nguess, M=None):
    """Start block (reference: _set_initial_v, symeig.py:229-253).

    Re-seeds the GLOBAL torch RNG with 12421 (quirk Q5) and orthonormalises with tallqr.
    """
    torch.manual_seed(12421)
    if kind == "eye":
        nb = 1
        for d in batch_dims:
            nb *= d
        V = torch.eye(na, nguess, dtype=dtype).unsqueeze(0).repeat(nb, 1, 1).reshape(*batch_dims, na, nguess)
    elif kind == "randn":
        V = torch.randn((*batch_dims, na, nguess), dtype=dtype)
    elif kind in ("rand", "random"):
        V = torch.rand((*batch_dims, na, nguess), dtype=dtype)
    else:
        raise ValueError("Unknown v_init type: %s" % kind)
    if M is not None:
        V, _ = tallqr(V, MV=M.mm(V))
    else:
        V, _ = tallqr(V)
    return V


def take_eigpairs(evals, evecs, neig, mode):
    """First / last `neig` of an ascending eigh result (reference: symeig.py:255-264)."""
    if mode == "lowest":
        return evals[..., :neig], evecs[..., :neig]
    return evals[..., -neig:], evecs[..., -neig:]


def davidson(A, neig, mode="lowest", M=None, max_niter=1000, nguess=None, v_init="randn",
             max_addition=None, min_eps=1e-6, verbose=False, V0=None, trace=None, **unused):
    """Un-restarted block Davidson (reference: davidson, symeig.py:100-227).

    Extra (oracle-only) arguments: `V0` an already orthonormal start block replacing the
    RNG draw (for device-independent parity runs); `trace` a dict that receives
    `niter`, `napply`, `resid_history`, `basis_size`.
    """
    if nguess is None:
        nguess = neig
    na = A.shape[-1]
    if M is None:
        bdims = list(A.shape[:-2])
    else:
        bdims = list(torch.broadcast_shapes(tuple(A.shape[:-2]), tuple(M.shape[:-2])))
    dtype = A.dtype

    if V0 is None:
        V = initial_v(v_init.lower(), dtype, bdims, na, nguess, M=M)
    else:
        V = V0
        nguess = V.shape[-1]

    best_resid = float("inf")
    best_evals = best_evecs = None
    history = []
    AV = A.mm(V)                                             # symeig.py:163
    napply = 1
    niter = 0
    for it in range(max_niter):
        niter = it + 1
        T = torch.matmul(V.transpose(-2, -1), AV)            # :170
        lam, Y = torch.linalg.eigh(T)                        # :174
        lam, Y = take_eigpairs(lam, Y, neig, mode)           # :175
        X = torch.matmul(V, Y)                               # :178
        AX = torch.matmul(AV, Y)                             # :181
        LX = lam.unsqueeze(-2) * X                           # :182
        if M is not None:
            LX = M.mm(LX)                                    # :184
        resid = AX - LX                                      # :185
        max_resid = resid.abs().max()                        # :188 (global over batch and columns)
        history.append(float(max_resid))
        if verbose:
            print("oracle davidson iter %3d (basis %d): resid %.3e" % (it + 1, nguess, float(max_resid)))
        if max_resid < best_resid:                           # :196-199 best-so-far
            best_resid = max_resid
            best_evals, best_evecs = lam, X
        if max_resid < min_eps:                              # :200
            break
        if AV.shape[-1] == AV.shape[-2]:                     # :202 basis is square
            break
        t = to_fortran_order(-resid)                         # :207-210 (no preconditioner)
        Vnew = torch.cat((V, t), dim=-1)                     # :211
        if Vnew.shape[-1] > Vnew.shape[-2]:
            Vnew = Vnew[..., :Vnew.shape[-2]]
        nadd = Vnew.shape[-1] - V.shape[-1]
        nguess = nguess + nadd
        if M is not None:                                    # :216-220 full CholeskyQR of the basis
            V, _ = tallqr(Vnew, MV=M.mm(Vnew))
        else:
            V, _ = tallqr(Vnew)
        AVnew = to_fortran_order(A.mm(V[..., -nadd:]))       # :221-222
        napply += 1
        AV = torch.cat((AV, AVnew), dim=-1)                  # :223
    if trace is not None:
        trace.update(niter=niter, napply=napply, resid_history=history, basis_size=int(V.shape[-1]),
                     best_resid=float(best_resid))
    return best_evals, best_evecs


def exacteig(A, neig, mode="lowest", M=None):
    """Dense reference answer (reference: exacteig, symeig.py:11-44)."""
    Amat = A.fullmatrix()
    if M is None:
        lam, Y = torch.linalg.eigh(Amat)
        return take_eigpairs(lam, Y, neig, mode)
    L = torch.linalg.cholesky(M.fullmatrix())
    Linv = torch.inverse(L)
    LinvT = Linv.transpose(-2, -1).conj()
    A2 = torch.matmul(Linv, torch.matmul(Amat, LinvT))
    lam, Y = torch.linalg.eigh(A2)
    lam, Y = take_eigpairs(lam, Y, neig, mode)
    return lam, torch.matmul(LinvT, Y)
