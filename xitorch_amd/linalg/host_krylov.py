"""cg / bicgstab / gmres for operators that live in HOST memory.

Device dispatch, not a fallback: the reference runs its Krylov methods on whatever device the operator's tensors are
on (xitorch/_impls/linalg/solve.py:69-433).  Here an operator on a HIP device is served by the HIP kernels of
native_krylov.py and by nothing else (a missing library raises `NativeLibraryError`; nothing in this file is ever
reached with a device tensor — `_HostProblem` refuses one), and an operator whose tensors are in host memory is served
by this file: the SAME host drivers as native_krylov.py — every (batch member, column) pair is one system, all systems
advance in lock step, one stopping test per iteration over all of them, best iterate returned, true residual refreshed
every `resid_calc_every` iterations, exact zeros in denominators replaced by `eps`, normal equations when the problem is
not positive definite — with each fused kernel call replaced by the torch expression it computes.  Vectors keep the
caller's layout `(*batch, N, ncols)`; per-system scalars are `(*batch, 1, ncols)` tensors; the per-column shift `E` is a
broadcast factor (the reference moves the columns to a leading axis instead, solve.py:575-604).

Nothing here imports `oracle/` (test infrastructure); tests/test_host_methods.py checks this file against the reference's
golden vectors and the reference's own test-suite runs on it through scripts/reference_suite_shim.py.
"""
import warnings
import torch
from xitorch_amd._capi import NativeLibraryError
from xitorch_amd._util import bcast_shape, ConvergenceWarning
from xitorch_amd.dist import allreduce_max_, all_ranks_agree_true

__all__ = ["cg", "bicgstab", "gmres", "scipy_gmres"]

calls = {"cg": 0, "bicgstab": 0, "gmres": 0, "scipy_gmres": 0}        # how often each host driver ran (tests assert 0 on the GPU path)


def _nonzero(d, eps):
    """denominators: exact zeros become eps (reference: _safedenom, solve.py:437-439)"""
    return torch.where(d == 0, torch.full_like(d, eps), d)


def _coldot(a, b):
    """per-system inner product sum conj(a) b over the vector axis -> (*batch, 1, ncols)"""
    return (a.conj() * b).sum(dim=-2, keepdim=True)


def _colnorm(a):
    return torch.linalg.vector_norm(a, dim=-2, keepdim=True)


class _HostProblem:
    """apply(X) = A X - (M X) E per column, or the normal equations of it; rhs accordingly
    (reference: _setup_linear_problem, solve.py:560-643)."""

    def __init__(self, A, B, E, M, bdims, posdef, need_hermit):
        dev = torch.device(A.device)
        if dev.type != "cpu":
            raise NativeLibraryError("host_krylov serves operators in host memory only (operator is on %s): device "
                                     "operators run on the HIP kernels" % dev)
        self.A, self.M = A, (M if E is not None else None)
        self.dtype = A.dtype
        self.bdims = list(bdims)
        self.N, self.nc = A.shape[-1], B.shape[-1]
        self.shape = (*self.bdims, self.N, self.nc)
        self.E = None if E is None else E.to(self.dtype).unsqueeze(-2)            # (*BE, 1, ncols)
        self.napply = 0
        rhs = B.to(self.dtype).expand(*self.shape)
        hermit = A.is_hermitian and (M is None or M.is_hermitian)
        if need_hermit and not hermit:
            posdef = False                                                        # solve.py:607-612
        if posdef is None:
            posdef = self._posdef_heuristic()
        self.normal = not posdef
        self.rhs = self._apply1(rhs, adjoint=True) if self.normal else rhs        # A -> A^H A, B -> A^H B (:637-643)

    def _apply1(self, X, adjoint=False):
        self.napply += 1
        if adjoint:
            Y = self.A.rmm(X)
        else:
            Y = self.A.mm(X)
        if self.E is not None:
            if self.M is None:
                Z = X
            else:
                Z = self.M.rmm(X) if adjoint else self.M.mm(X)
            Y = Y - Z * (self.E.conj() if adjoint else self.E)
        return Y.expand(*self.shape) if Y.shape != torch.Size(self.shape) else Y

    def apply(self, X):
        if not self.normal:
            return self._apply1(X)
        return self._apply1(self._apply1(X), adjoint=True)

    def _posdef_heuristic(self):
        # solve.py:617-634 + :645-663: <= 10 power iterations from an (unseeded) random start; the estimate is a NORM,
        # so the answer is "positive definite unless the operator is zero"
        x0 = torch.randn(self.shape, dtype=self.dtype)
        x0 = x0 / _colnorm(x0)

        def last_norm(fcn, x):
            prev = None
            for i in range(10):
                x = fcn(x)
                xn = _colnorm(x)
                if i > 0 and bool(torch.all(torch.abs(prev - xn) <= 1e-3 * xn + 1e-6)):
                    break
                prev = xn
                if i < 9:
                    x = x / xn
            return xn
        big = last_norm(self._apply1, x0)
        neg = big <= 0
        if bool(torch.all(neg)):
            return False
        offset = torch.clamp(big, min=0.0)
        mostneg = last_norm(lambda x: self._apply1(x) - offset * x, x0)
        return bool(torch.all(torch.logical_or(-mostneg <= offset, neg)).item())


def _batchdims(A, B, E, M):
    shapes = [A.shape[:-2], B.shape[:-2]]
    if E is not None:
        shapes.append(E.shape[:-1])
        if M is not None:
            shapes.append(M.shape[:-2])
    return bcast_shape(*shapes)


def _zero_solution(A, B, bdims):
    return torch.zeros((*bdims, A.shape[-1], B.shape[-1]), dtype=A.dtype, device=A.device)


def _stop_of(prob, rtol, atol):
    bn = _colnorm(prob.rhs)
    return torch.max(rtol * bn, atol * torch.ones_like(bn))


def _status(rnorm, stop, process_group):
    """(largest residual norm, number of systems above their threshold), over all systems and all ranks"""
    st = torch.stack([rnorm.max().double(), (rnorm >= stop).sum().double()])
    allreduce_max_(st, process_group)
    mx, nbad = st.tolist()
    return mx, nbad


def _precond(P):
    if P is None:
        return None
    from xitorch_amd.linop import LinearOperator
    if not isinstance(P, LinearOperator):
        raise TypeError("precond can only be LinearOperator or None")
    return P.mm


def cg(A, B, E=None, M=None, posdef=None, precond=None, max_niter=None, rtol=1e-6, atol=1e-8, eps=1e-12,
       resid_calc_every=10, verbose=False, process_group=None, trace=None, **unused):
    """Preconditioned conjugate gradients in host memory (reference: cg, solve.py:69-190); options as
    `native_krylov.cg`."""
    calls["cg"] += 1
    if max_niter is None:
        max_niter = int(1.5 * A.shape[-1])
    bdims = _batchdims(A, B, E, M)
    if all_ranks_agree_true(torch.allclose(B, B * 0, rtol=rtol, atol=atol), B.device, process_group):
        return _zero_solution(A, B, bdims)
    prob = _HostProblem(A, B, E, M, bdims, posdef, need_hermit=True)
    pre = _precond(precond)
    stop = _stop_of(prob, rtol, atol)
    x = torch.zeros(prob.shape, dtype=prob.dtype)
    r = prob.rhs                                                    # x0 = 0
    z = pre(r) if pre is not None else r
    p = z
    rz = _coldot(r, z)
    best, _ = _status(_colnorm(r), stop, process_group)
    xbest = x
    converged, niter = False, 0
    for k in range(1, max_niter + 1):
        niter = k
        Ap = prob.apply(p)
        alpha = rz / _nonzero(_coldot(p, Ap), eps)
        x = x + alpha * p
        if resid_calc_every != 0 and k % resid_calc_every == 0:     # solve.py:148-149
            r = prob.rhs - prob.apply(x)
        else:
            r = r - alpha * Ap
        mx, nbad = _status(_colnorm(r), stop, process_group)
        if mx < best:
            best, xbest = mx, x
        if verbose and (k < 10 or k % 10 == 0):
            print("%4d: |dy|=%.3e" % (k, mx))
        if nbad == 0:
            converged = True
            break
        z = pre(r) if pre is not None else r
        rz_new = _coldot(r, z)
        p = z + (rz_new / _nonzero(rz, eps)) * p
        rz = rz_new
    if trace is not None:
        trace.update(niter=niter, napply=prob.napply, converged=converged, best_resid=best)
    if not converged:
        warnings.warn(ConvergenceWarning("Convergence is not achieved after %d iterations. "
                                         "Max norm of best resid: %.3e" % (max_niter, best)))
    return xbest


def bicgstab(A, B, E=None, M=None, posdef=None, precond_l=None, precond_r=None, max_niter=None, rtol=1e-6, atol=1e-8,
             eps=1e-12, verbose=False, resid_calc_every=10, process_group=None, trace=None, **unused):
    """Stabilised bi-conjugate gradients in host memory (reference: bicgstab, solve.py:192-324); options as
    `native_krylov.bicgstab`."""
    calls["bicgstab"] += 1
    if max_niter is None:
        max_niter = int(1.5 * B.shape[-2])
    bdims = _batchdims(A, B, E, M)
    if all_ranks_agree_true(torch.allclose(B, B * 0, rtol=rtol, atol=atol), B.device, process_group):
        return _zero_solution(A, B, bdims)
    prob = _HostProblem(A, B, E, M, bdims, posdef, need_hermit=False)
    pl, pr = _precond(precond_l), _precond(precond_r)
    stop = _stop_of(prob, rtol, atol)
    x = torch.zeros(prob.shape, dtype=prob.dtype)
    r = prob.rhs
    r0 = r
    rho_old = _coldot(r0, r)
    omega = torch.ones_like(rho_old)
    alpha = torch.ones_like(rho_old)
    p = torch.zeros_like(r)
    v = torch.zeros_like(r)
    best, _ = _status(_colnorm(r), stop, process_group)
    xbest = x
    converged, niter = False, 0
    for k in range(1, max_niter + 1):
        niter = k
        rho = _coldot(r0, r)
        omega = _nonzero(omega, eps)                                # (the reference patches omega itself, :274)
        beta = rho / _nonzero(rho_old, eps) * (alpha / omega)
        p = r + beta * (p - omega * v)
        y = pr(p) if pr is not None else p
        v = prob.apply(y)
        alpha = rho / _nonzero(_coldot(r0, v), eps)
        s = r - alpha * v
        z = pr(s) if pr is not None else s
        t = prob.apply(z)
        Kt = pl(t) if pl is not None else t
        Ks = pl(s) if pl is not None else s
        omega = _coldot(Kt, Ks) / _nonzero(_coldot(Kt, Kt), eps)
        x = x + alpha * y + omega * z
        if resid_calc_every != 0 and k % resid_calc_every == 0:     # solve.py:290-291
            r = prob.rhs - prob.apply(x)
        else:
            r = s - omega * t
        mx, nbad = _status(_colnorm(r), stop, process_group)
        if mx < best:
            best, xbest = mx, x
        if verbose and (k < 10 or k % 10 == 0):
            print("%4d: |dy|=%.3e" % (k, mx))
        if nbad == 0:
            converged = True
            break
        rho_old = rho
    if trace is not None:
        trace.update(niter=niter, napply=prob.napply, converged=converged, best_resid=best)
    if not converged:
        warnings.warn(ConvergenceWarning("Convergence is not achieved after %d iterations. "
                                         "Max norm of resid: %.3e" % (max_niter, best)))
    return xbest


def gmres(A, B, E=None, M=None, posdef=None, max_niter=None, rtol=1e-6, atol=1e-8, eps=1e-12, resid_calc_every=1,
          restart=None, process_group=None, trace=None, **unused):
    """Un-restarted GMRES in host memory (reference: gmres, solve.py:326-433; real operators only, like the
    reference's): the same iterates, stopping rule and return value as `native_krylov.gmres` — after k Arnoldi steps
    the iterate minimises the residual over the k-dimensional Krylov space, the TRUE residual decides convergence and
    which iterate is the best one, at most min(nr, max_niter) - 1 Krylov vectors contribute.  The Hessenberg matrices
    of all systems are one batched tensor; the small least-squares problem is `torch.linalg.lstsq`, as in the
    reference (:403).  `restart=m` (extension): GMRES(m), as in the native driver."""
    calls["gmres"] += 1
    nr = A.shape[-1]
    if A.dtype.is_complex:
        raise NativeLibraryError("xitorch_amd gmres supports real operators only, like the reference's gmres")
    if max_niter is None:
        max_niter = int(nr)
    bdims = _batchdims(A, B, E, M)
    if all_ranks_agree_true(torch.allclose(B, B * 0, rtol=rtol, atol=atol), B.device, process_group):
        return _zero_solution(A, B, bdims)
    prob = _HostProblem(A, B, E, M, bdims, posdef, need_hermit=False)
    stop = _stop_of(prob, rtol, atol)
    msteps = min(nr, max_niter) - 1
    if restart is not None:
        restart = int(restart)
        if restart < 1:
            raise ValueError("gmres: restart must be a positive number of Arnoldi steps, got %d" % restart)
        msteps = max_niter - 1 if max_niter > 1 else 0
    mcyc = msteps if restart is None else min(restart, max(msteps, 1))
    every = max(1, int(resid_calc_every))
    xbase = torch.zeros(prob.shape, dtype=prob.dtype)
    r = prob.rhs
    beta = _colnorm(r)                                              # (*batch, 1, nc)
    best = float(allreduce_max_(beta.max().double().reshape(1), process_group).item())
    xbest = xbase
    converged, nsteps, ncycles = False, 0, 0
    Q = [r / _nonzero(beta, eps)]
    H = torch.zeros((*prob.bdims, prob.nc, mcyc + 1, max(mcyc, 1)), dtype=prob.dtype)
    cyc0 = 0
    for k in range(msteps):
        j = k - cyc0
        nsteps = k + 1
        w = prob.apply(Q[j])
        for i in range(j + 1):                                      # modified Gram-Schmidt (solve.py:391-393)
            hij = _coldot(Q[i], w)
            H[..., i, j] = hij.squeeze(-2)
            w = w - hij * Q[i]
        hn = _colnorm(w)
        H[..., j + 1, j] = hn.squeeze(-2)
        Q.append(w / _nonzero(hn, eps))
        cycle_end = restart is not None and j + 1 == mcyc
        if not ((k + 1) % every == 0 or k == msteps - 1 or cycle_end):
            continue
        # x = x_base + Q y with y = argmin |beta e1 - H y|  (:403-410), then the true residual (:414)
        g = torch.zeros((*prob.bdims, prob.nc, j + 2, 1), dtype=prob.dtype)
        g[..., 0, 0] = beta.squeeze(-2)
        ycoef = torch.linalg.lstsq(H[..., :j + 2, :j + 1], g)[0]    # (*batch, nc, j+1, 1)
        x = xbase
        for i in range(j + 1):
            x = x + Q[i] * ycoef[..., i, 0].unsqueeze(-2)
        rtrue = prob.rhs - prob.apply(x)
        mx, nbad = _status(_colnorm(rtrue), stop, process_group)
        if mx < best:                                               # solve.py:417-421
            best, xbest = mx, x
        if nbad == 0:                                               # :423-425
            converged = True
            break
        if cycle_end and k < msteps - 1:
            xbase, r = x, rtrue
            beta = _colnorm(r)
            Q = [r / _nonzero(beta, eps)]
            H.zero_()
            cyc0 = k + 1
            ncycles += 1
    if trace is not None:
        trace.update(niter=nsteps + 1, napply=prob.napply, converged=converged, best_resid=best, arnoldi_steps=nsteps,
                     restarts=ncycles)
    if not converged:
        warnings.warn(ConvergenceWarning("Convergence is not achieved after %d iterations. "
                                         "Max norm of resid: %.3e" % (max_niter, best)))
    return xbest


def scipy_gmres(A, B, E=None, M=None, min_eps=1e-9, max_niter=None, **unused):
    """The reference's `method="scipy_gmres"` (wrap_gmres, solve.py:14-66): SciPy's restarted GMRES, system by system,
    for an UNBATCHED real operator and a right-hand side with one batch dimension, `A X = B` only.  The operator is
    handed to SciPy through `LinearOperator.scipy_linalg_op()`; the tolerance goes in under the name the installed SciPy
    understands (`rtol` from 1.12 on, `tol` before — the reference passes `tol`, which SciPy >= 1.14 rejects)."""
    import inspect
    import numpy as np
    from scipy.sparse.linalg import gmres as _sp_gmres
    calls["scipy_gmres"] += 1
    if len(A.shape) != 2 or len(B.shape) != 3:
        raise RuntimeError("Currently only works for batched B (1 batch dim), but unbatched A")
    if torch.is_complex(B):
        raise RuntimeError("complex is not supported in gmres")
    if E is not None or M is not None:
        raise RuntimeError("GMRES can only do AX=B")
    nbatch, na, ncols = B.shape
    if max_niter is None:
        max_niter = 2 * na
    tolname = "rtol" if "rtol" in inspect.signature(_sp_gmres).parameters else "tol"
    op = A.scipy_linalg_op()
    rhs = B.detach().transpose(-2, -1).cpu().numpy()
    out = np.empty_like(rhs)
    for i in range(nbatch):
        for j in range(ncols):
            x, info = _sp_gmres(op, rhs[i, j], atol=1e-12, maxiter=max_niter, **{tolname: min_eps})
            if info > 0:
                warnings.warn(ConvergenceWarning("The GMRES iteration does not converge to the desired value "
                                                 "(%.3e) after %d iterations" % (min_eps, info)))
            out[i, j] = x
    return torch.as_tensor(out, dtype=B.dtype, device=B.device).transpose(-2, -1)
