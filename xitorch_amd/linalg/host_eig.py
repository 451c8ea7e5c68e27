"""Block Davidson for operators that live in HOST memory.

Device dispatch, not a fallback (see host_krylov.py): the reference's `davidson` runs on whatever device the operator is
on (xitorch/_impls/linalg/symeig.py:100-227, :149); an operator on a HIP device is served by native_eig.py's HIP kernels
and nothing else, an operator whose tensors are in host memory by this file.  The iteration is native_eig.py's — start
block from the reference's seed, Rayleigh-Ritz on the whole basis, ONE global stopping test `max|A X - M X diag(lam)| <
min_eps`, best block returned, expansion by the negated residual block, the basis grows until it spans the space — and so
is the restructuring: only the NEW block is orthonormalised (two passes of block Gram-Schmidt against the basis + a QR of
the block; the reference re-factorises the whole basis by CholeskyQR every iteration, `tallqr([V, t])`, which yields the
same Q in exact arithmetic), and `T = V^H A V` is extended by its new rows instead of being recomputed.  Each HIP kernel
call of the native driver is the torch expression it computes.  Nothing here imports `oracle/`.
"""
import torch
from xitorch_amd._capi import NativeLibraryError
from xitorch_amd._util import bcast_shape
from xitorch_amd.dist import allreduce_max_

__all__ = ["davidson"]

calls = {"davidson": 0}


def _H(x):
    return x.transpose(-2, -1).conj()


def _orthonormalise(W, V, MV, M):
    """W (.., N, q) -> an (M-)orthonormal block, (M-)orthogonal to the basis V (V is None: the start block).
    Two rounds of [project out the basis, Householder QR of the block, and — with an overlap operator — a CholeskyQR of
    the now well-conditioned block in the M inner product].  A rank-deficient START block raises, like the native
    kernels; an expansion block whose residual columns are rounding noise (pairs that converged long ago) gets whatever
    orthonormal completion the QR supplies, which is harmless."""
    for _ in range(2):
        if V is not None:
            W = W - torch.matmul(V, torch.matmul(_H(MV), W))          # V (V^H M W): MV = M V, M Hermitian
        W, R = torch.linalg.qr(W)
        if V is None:
            d = torch.diagonal(R, dim1=-2, dim2=-1).abs()
            if bool((d.min(dim=-1)[0] <= 1e-13 * d.max(dim=-1)[0]).any()):
                raise RuntimeError("davidson: the start block is rank deficient (linearly dependent start vectors)")
        if M is not None:
            G = torch.matmul(_H(W), M.mm(W))
            L, info = torch.linalg.cholesky_ex((G + _H(G)) * 0.5)
            if bool((info != 0).any()):
                raise RuntimeError("davidson: the overlap operator M is not positive definite on the new block")
            W = torch.linalg.solve_triangular(_H(L), W, upper=True, left=False)
    return W


def davidson(A, neig, mode, M=None, max_niter=1000, nguess=None, v_init="randn", max_addition=None, min_eps=1e-6,
             verbose=False, V0=None, process_group=None, trace=None, rng_device="cpu", precond=None, restart=None,
             **unused):
    """Options as the reference's `davidson` plus `V0` (start block), `process_group` (batch-sharded ranks decide on
    the all-reduced residual) and `trace`, as in `native_eig.davidson`; the HIP driver's scheduling knobs are accepted
    and have no meaning here; `precond=` / `restart=` (extensions of the HIP driver) are not available."""
    calls["davidson"] += 1
    dev = torch.device(A.device)
    if dev.type != "cpu":
        raise NativeLibraryError("host_eig serves operators in host memory only (operator is on %s): device operators "
                                 "run on the HIP kernels" % dev)
    if precond is not None or restart is not None:
        raise NativeLibraryError("precond= / restart= are extensions of the HIP davidson; not available for an "
                                 "operator in host memory")
    from xitorch_amd.linalg.native_eig import _initial_block, _shard_of_global_batch
    N = A.shape[-1]
    if nguess is None:
        nguess = neig
    bdims = list(A.shape[:-2]) if M is None else list(bcast_shape(A.shape[:-2], M.shape[:-2]))
    dtype = A.dtype
    B = 1
    for d in bdims:
        B *= d
    distributed = process_group is not None and torch.distributed.get_world_size(process_group) > 1
    shard = None
    if distributed and V0 is None and v_init.lower() in ("randn", "rand", "random"):
        shard = _shard_of_global_batch(B, dev, process_group)
    V = _initial_block(v_init, V0, bdims, B, N, nguess, dtype, dev, "cpu", shard)      # (B, nguess, N), panel-major
    V = V.transpose(-2, -1).reshape(*bdims, N, V.shape[-2])
    V = _orthonormalise(V, None, None, M)
    MV = M.mm(V) if M is not None else V
    AV = A.mm(V)
    T = torch.matmul(_H(V), AV)
    napply = 1
    best_resid, best = float("inf"), None
    history = []
    niter = 0
    for it in range(max_niter):
        niter = it + 1
        k = V.shape[-1]
        lam, Y = torch.linalg.eigh((T + _H(T)) * 0.5)
        if mode == "lowest":
            lam, Y = lam[..., :neig], Y[..., :neig]
        else:
            lam, Y = lam[..., -neig:], Y[..., -neig:]
        X = torch.matmul(V, Y)
        R = torch.matmul(AV, Y) - torch.matmul(MV, Y) * lam.unsqueeze(-2)
        mx = R.abs().max().double().reshape(1)
        max_resid = float(allreduce_max_(mx, process_group if distributed else None).item())
        history.append(max_resid)
        if verbose:
            print("Iter %3d (guess size: %d): resid: %.3e" % (it + 1, k, max_resid))
        if max_resid < best_resid:
            best_resid, best = max_resid, (lam, X)
        if max_resid < min_eps or k == N:
            break
        nadd = min(R.shape[-1], N - k)
        W = _orthonormalise(-R[..., :nadd], V, MV, M)
        AW = A.mm(W)
        napply += 1
        Tcol = torch.matmul(_H(V), AW)                                 # (.., k, nadd)
        Tnew = torch.matmul(_H(W), AW)                                 # (.., nadd, nadd)
        T = torch.cat((torch.cat((T, Tcol), dim=-1), torch.cat((_H(Tcol), Tnew), dim=-1)), dim=-2)
        V = torch.cat((V, W), dim=-1)
        AV = torch.cat((AV, AW), dim=-1)
        MV = torch.cat((MV, M.mm(W)), dim=-1) if M is not None else V
    if trace is not None:
        trace.update(niter=niter, napply=napply, resid_history=history, basis_size=int(V.shape[-1]),
                     best_resid=float(best_resid), groups=1, panel_kernel="host")
    return best
