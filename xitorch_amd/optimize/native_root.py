"""Native quasi-Newton root solvers (Broyden-1/2, linear mixing, Newton) on the flattened variable.

Drop-in for the reference's method functions (xitorch/_impls/optimize/root/rootsolver.py:15-256,
_jacobian.py:51-232): same names, options (``alpha``, ``uv0``, ``max_rank``, ``maxiter``,
``f_tol/f_rtol/x_tol/x_rtol``, ``line_search``, ``verbose``, ``custom_terminator``), same Armijo
backtracking, same AND-termination, same quirks — the iterate *before* the converged one is
returned (Q1), SciPy's sign convention of ``alpha`` (Q2), "restart" drops the whole history when
the rank exceeds ``max_rank`` (Q3), the whole batch is one flat system (Q4).

What runs natively: the inverse-Jacobian model G = alpha*I + sum_n c_n d_n^T.  The reference keeps
c_n, d_n as Python lists and applies G with a Python loop of rank x (torch.dot + axpy)
(_jacobian.py:172-182).  Here C and D are two growing (rank, L) device buffers and

    G v   = alpha v + C^T (D v)      ->  one multi-dot  (xk_dense_mm, split-contraction path)
    G^T v = alpha v + D^T (C v)          + one multi-axpy (xk_lincomb)

so an apply is two streaming passes over 2*rank*L elements regardless of the rank.  With a
``process_group`` the flat vector is sharded over the ranks (batch sharding) and every inner
product / norm is completed by ONE small all-reduce(SUM) (RCCL) — the rank-vector of the
multi-dot travels as a single message.
"""
import warnings
import torch
from xitorch_amd import kernels as K
from xitorch_amd._capi import NativeLibraryError
from xitorch_amd._util import ConvergenceWarning
from xitorch_amd.linalg._panel import pad_len
from xitorch_amd.dist import allreduce_sum_

__all__ = ["broyden1", "broyden2", "linearmixing", "newton"]


# ------------------------------------------------------------------------------ reductions
class _Reduce:
    """Global inner products of (possibly sharded) flat vectors."""

    def __init__(self, group):
        self.group = group

    def _sum(self, t):
        return allreduce_sum_(t, self.group)

    def dot(self, a, b):
        return self._sum(torch.dot(a, b).reshape(1))[0]

    def norm(self, a):
        if self.group is None:
            return a.norm()
        return torch.sqrt(self._sum(torch.dot(a, a).reshape(1))[0])

    def total_numel(self, a):
        if self.group is None:
            return a.numel()
        t = torch.tensor([float(a.numel())], dtype=torch.float64, device=a.device)
        return int(self._sum(t).item())


# ------------------------------------------------------------------------------ low-rank model
class _LowRank:
    """G = alpha*I + sum_n c_n d_n^T in two growing device buffers (reference: LowRankMatrix /
    FullRankMatrix, _jacobian.py:156-222)."""

    def __init__(self, alpha, uv0, L, dtype, device, red, total_L):
        if device.type != "cuda":
            raise NativeLibraryError("xitorch_amd Broyden runs on a HIP device only (variable is on %s); "
                                     "there is no CPU fallback" % device)
        if dtype not in (torch.float64, torch.float32):
            raise NativeLibraryError("xitorch_amd Broyden supports float64/float32, got %s" % dtype)
        self.alpha = float(alpha)
        self.L, self.Lp = L, pad_len(L)
        self.total_L = total_L
        self.dtype, self.device, self.red = dtype, device, red
        self.cap, self.rank = 8, 0
        self.C = torch.zeros((1, self.cap, self.Lp), dtype=dtype, device=device)
        self.D = torch.zeros((1, self.cap, self.Lp), dtype=dtype, device=device)
        self.dense = None
        self._buf = torch.zeros((1, 1, self.Lp), dtype=dtype, device=device)
        if uv0 is not None:
            self.append(uv0[0], uv0[1])

    def _grow(self):
        new = self.cap * 2
        for name in ("C", "D"):
            old = getattr(self, name)
            buf = torch.zeros((1, new, self.Lp), dtype=self.dtype, device=self.device)
            buf[:, :self.cap].copy_(old)
            setattr(self, name, buf)
        self.cap = new

    def _apply(self, first, second, v):
        # alpha v + second^T (first v)
        if self.dense is not None:
            raise AssertionError
        L = self.L
        vin = self._buf
        vin[0, 0, :L].copy_(v)
        out = torch.zeros((1, 1, self.Lp), dtype=self.dtype, device=self.device)
        out[0, 0, :L].copy_(v)
        if self.rank == 0:
            return out[0, 0, :L] * self.alpha
        coef = K.dense_mm(first[:, :self.rank, :L], vin[:, :, :L])          # (1, 1, rank): <first_n, v>
        if self.red.group is not None:
            self.red._sum(coef)
        K.lincomb(second, coef, out, self.rank, 1, coef_layout="ca", alpha=1.0, beta=self.alpha)
        return out[0, 0, :L]

    def mv(self, v):
        if self.dense is not None:
            return torch.matmul(self.dense, v)
        return self._apply(self.D, self.C, v)

    def rmv(self, v):
        if self.dense is not None:
            return torch.matmul(self.dense.T, v)
        return self._apply(self.C, self.D, v)

    def append(self, c, d):
        if self.dense is not None:                                    # FullRankMatrix.append
            self.dense += torch.outer(c, d)
            return
        if self.rank == self.cap:
            self._grow()
        self.C[0, self.rank, :self.L].copy_(c)
        self.D[0, self.rank, :self.L].copy_(d)
        self.rank += 1
        if self.rank >= self.total_L and self.red.group is None:       # _jacobian.py:187-188
            n = self.L
            mat = torch.eye(n, dtype=self.dtype, device=self.device) * self.alpha
            mat += torch.matmul(self.C[0, :self.rank, :n].T, self.D[0, :self.rank, :n])
            self.dense = mat

    def reduce(self, max_rank):
        # "restart": forget everything once the rank EXCEEDS max_rank (checked before appending)
        if self.dense is None and self.rank > max_rank:
            self.C[:, :self.rank].zero_()
            self.D[:, :self.rank].zero_()
            self.rank = 0


class _BroydenFirst:
    """reference: BroydenFirst, _jacobian.py:51-119."""

    def __init__(self, alpha=None, uv0=None, max_rank=None):
        self.alpha, self.uv0, self.max_rank = alpha, uv0, max_rank

    def setup(self, x0, y0, func, red):
        self.red = red
        self.x_prev, self.y_prev = x0, y0
        if self.max_rank is None:
            self.max_rank = float("inf")
        if self.alpha is None:                                        # _jacobian.py:76-82 (Q2)
            ny0 = float(red.norm(y0))
            self.alpha = 0.5 * max(float(red.norm(x0)), 1.0) / ny0 if ny0 else 1.0
        if isinstance(self.uv0, str) and self.uv0 == "svd":
            self.uv0 = _svd_uv0(func, x0)
        self.Gm = _LowRank(-float(self.alpha), self.uv0, x0.numel(), x0.dtype, x0.device, red,
                           red.total_numel(x0))

    def solve(self, v, tol=0):
        return self.Gm.mv(v)

    def update(self, x, y):
        dy = y - self.y_prev
        dx = x - self.x_prev
        self._update(dx, dy)
        self.y_prev, self.x_prev = y, x

    def _update(self, dx, dy):
        self.Gm.reduce(self.max_rank)
        v = self.Gm.rmv(dx)
        c = dx - self.Gm.mv(dy)
        d = v / self.red.dot(dy, v)
        self.Gm.append(c, d)

    @property
    def rank(self):
        return self.Gm.rank


class _BroydenSecond(_BroydenFirst):
    """reference: BroydenSecond, _jacobian.py:121-137."""

    def _update(self, dx, dy):
        self.Gm.reduce(self.max_rank)
        c = dx - self.Gm.mv(dy)
        dyn = self.red.norm(dy)
        self.Gm.append(c, dy / (dyn * dyn))


class _LinearMixing:
    """reference: LinearMixing, _jacobian.py:139-154."""
    rank = 0

    def __init__(self, alpha=None):
        self.alpha = -1.0 if alpha is None else alpha

    def setup(self, x0, y0, func, red):
        pass

    def solve(self, v, tol=0):
        return -v * self.alpha

    def update(self, x, y):
        pass


class _NewtonJacobian:
    """reference: NewtonJacobian, _jacobian.py:27-49 — the exact Jacobian as a LinearOperator."""
    rank = 0

    def __init__(self, solver_method="exactsolve", solver_kwargs=None):
        self.solver_method = solver_method
        self.solver_kwargs = solver_kwargs if solver_kwargs is not None else {}

    def setup(self, x0, y0, func, red):
        self.x, self.func = x0, func

    def solve(self, v, tol=0):
        from xitorch_amd.linalg import solve
        from xitorch_amd.grad import jac
        J = jac(self.func, (self.x.clone().requires_grad_(),), idxs=0)
        return solve(J, v[..., None], method=self.solver_method, **self.solver_kwargs)[..., 0]

    def update(self, x, y):
        self.x = x


def _svd_uv0(func, x0):
    # reference: _get_svd_uv0, _jacobian.py:224-232
    from xitorch_amd.linalg import svd
    from xitorch_amd.grad import jac
    fjac = jac(func, (x0.clone().requires_grad_(),), idxs=[0])[0]
    u, s, vh = svd(fjac, k=1, mode="lowest", method="davidson", min_eps=1e-3)
    sinv_sqrt = 1.0 / torch.sqrt(torch.clamp(s, min=0.1))
    return (sinv_sqrt * vh.squeeze(-2), sinv_sqrt * u.squeeze(-1))


# ------------------------------------------------------------------------------ line search
def _armijo(phi, phi0, derphi0, c1=1e-4, alpha0=1.0, amin=0.0, max_niter=20):
    """Backtracking with quadratic then cubic interpolation (reference: _scalar_search_armijo,
    rootsolver.py:312-357).  Scalars are host floats."""
    phi_a0 = phi(alpha0)
    if phi_a0 <= phi0 + c1 * alpha0 * derphi0:
        return alpha0, phi_a0
    alpha1 = -(derphi0) * alpha0 ** 2 / 2.0 / (phi_a0 - phi0 - derphi0 * alpha0)
    phi_a1 = phi(alpha1)
    if phi_a1 <= phi0 + c1 * alpha1 * derphi0:
        return alpha1, phi_a1
    niter = 0
    alpha2, phi_a2 = alpha1, phi_a1
    while alpha1 > amin and niter < max_niter:
        factor = alpha0 ** 2 * alpha1 ** 2 * (alpha1 - alpha0)
        a = alpha0 ** 2 * (phi_a1 - phi0 - derphi0 * alpha1) - alpha1 ** 2 * (phi_a0 - phi0 - derphi0 * alpha0)
        a = a / factor
        b = -alpha0 ** 3 * (phi_a1 - phi0 - derphi0 * alpha1) + alpha1 ** 3 * (phi_a0 - phi0 - derphi0 * alpha0)
        b = b / factor
        alpha2 = (-b + abs(b ** 2 - 3 * a * derphi0) ** 0.5) / (3.0 * a)
        phi_a2 = phi(alpha2)
        if phi_a2 <= phi0 + c1 * alpha2 * derphi0:
            return alpha2, phi_a2
        if (alpha1 - alpha2) > alpha1 / 2.0 or (1 - alpha2 / alpha1) < 0.96:
            alpha2 = alpha1 / 2.0
        alpha0, alpha1, phi_a0, phi_a1 = alpha1, alpha2, phi_a1, phi_a2
        niter += 1
    if niter == max_niter:
        return alpha2, phi_a2
    return None, phi_a1


def _line_search(func, x, y, dx, red, smin=1e-2):
    """reference: _nonline_line_search, rootsolver.py:272-310."""
    state = {"s": 0.0, "y": y, "phi": float(red.dot(y, y))}

    def phi(s):
        if s == state["s"]:
            return state["phi"]
        v = func(x + s * dx)
        p = float(red.dot(v, v))
        state.update(s=s, y=v, phi=p)
        return p

    s, _ = _armijo(phi, state["phi"], -state["phi"], amin=smin)
    if s is None:
        s = 1.0
    xn = x + s * dx
    yn = state["y"] if s == state["s"] else func(xn)
    return s, xn, yn, float(red.norm(yn))


class _Termination:
    """reference: TerminationCondition, rootsolver.py:359-380."""

    def __init__(self, f_tol, f_rtol, f0_norm, x_tol, x_rtol, red):
        self.f_tol = 1e-6 if f_tol is None else f_tol
        self.f_rtol = float("inf") if f_rtol is None else f_rtol
        self.x_tol = 1e-6 if x_tol is None else x_tol
        self.x_rtol = float("inf") if x_rtol is None else x_rtol
        self.f0_norm, self.red = f0_norm, red

    def check(self, x, y, dx):
        xn, yn, dxn = float(self.red.norm(x)), float(self.red.norm(y)), float(self.red.norm(dx))
        return (dxn < self.x_tol) and (dxn < self.x_rtol * xn) and (yn < self.f_tol) and \
            (yn < self.f_rtol * self.f0_norm)


# ------------------------------------------------------------------------------ driver
def _nonlin_solver(fcn, x0, params, jacobian, maxiter=None, f_tol=None, f_rtol=None, x_tol=None,
                   x_rtol=None, line_search=True, verbose=False, custom_terminator=None,
                   process_group=None, trace=None, **unused):
    """
    Keyword arguments
    -----------------
    maxiter: int or None
        Maximum number of iterations, or ``100*(numel+1)`` if None.
    f_tol: float or None
        The absolute tolerance of the norm of the output ``f``.
    f_rtol: float or None
        The relative tolerance of the norm of the output ``f``.
    x_tol: float or None
        The absolute tolerance of the norm of the input ``x``.
    x_rtol: float or None
        The relative tolerance of the norm of the input ``x``.
    line_search: bool or str
        Options to perform line search. If ``True``, it is set to ``"armijo"``.
    verbose: bool
        Options for verbosity
    process_group: torch.distributed group or None
        (extension) the flat variable is sharded over the group's ranks; all inner products and
        norms are completed by an all-reduce(SUM)
    """
    red = _Reduce(process_group)
    if maxiter is None:
        maxiter = 100 * (red.total_numel(x0) + 1)
    if line_search is True:
        line_search = "armijo"
    elif line_search is False:
        line_search = None
    if torch.is_complex(x0):
        raise NativeLibraryError("complex variables are not supported by the native root solvers")
    xshape = x0.shape
    func = lambda x: fcn(x.reshape(xshape), *params).reshape(-1)
    nfev = [0]

    def cfunc(x):
        nfev[0] += 1
        return func(x)

    x = x0.reshape(-1)
    y = cfunc(x)
    y_norm = float(red.norm(y))
    stop_cond = custom_terminator if custom_terminator is not None else \
        _Termination(f_tol, f_rtol, y_norm, x_tol, x_rtol, red)
    if y_norm == 0:
        return x.reshape(xshape)
    jacobian.setup(x, y, cfunc, red)

    gamma, eta_max, eta_threshold, eta = 0.9, 0.9999, 0.1, 1e-3
    converge = False
    best_ynorm, best_x, best_dxnorm, best_iter = y_norm, x, float(red.norm(x)), 0
    niter = 0
    for i in range(maxiter):
        niter = i + 1
        dx = -jacobian.solve(y, tol=min(eta, eta * y_norm))
        dx_norm = float(red.norm(dx))
        if dx_norm == 0:
            raise ValueError("Jacobian inversion yielded zero vector. "
                             "This indicates a bug in the Jacobian approximation.")
        if line_search:
            s, xnew, ynew, y_norm_new = _line_search(cfunc, x, y, dx, red)
        else:
            s = 1.0
            xnew = x + dx
            ynew = cfunc(xnew)
            y_norm_new = float(red.norm(ynew))
        if y_norm_new < best_ynorm:
            best_x, best_dxnorm, best_ynorm, best_iter = xnew, dx_norm, y_norm_new, i + 1
        jacobian.update(xnew.clone(), ynew)
        to_stop = stop_cond.check(xnew, ynew, dx)
        if verbose and (i < 10 or i % 10 == 0 or to_stop):
            print("%6d: |dx|=%.3e, |f|=%.3e" % (i, dx_norm, y_norm))
        if to_stop:
            converge = True
            break                  # NB: leaves x at the PREVIOUS iterate, like the reference (Q1)
        eta_A = float(gamma * (y_norm_new / y_norm) ** 2)
        gamma_eta2 = gamma * eta * eta
        eta = min(eta_max, eta_A) if gamma_eta2 < eta_threshold else min(eta_max, max(eta_A, gamma_eta2))
        y_norm, x, y = y_norm_new, xnew, ynew
    if trace is not None:
        trace.update(niter=niter, nfev=nfev[0], converged=converge, rank=getattr(jacobian, "rank", None),
                     best_ynorm=best_ynorm)
    if not converge:
        warnings.warn(ConvergenceWarning("The rootfinder does not converge after %d iterations. "
                                         "Best |dx|=%.3e, |f|=%.3e at iter %d"
                                         % (maxiter, best_dxnorm, best_ynorm, best_iter)))
        x = best_x
    return x.reshape(xshape)


def newton(fcn, x0, params=(), *, solver_method="exactsolve", solver_kwargs=None, **kwargs):
    """
    Solve the root finder using the Newton method, ``x_{n+1} = x_n - J^{-1}(x_n) f(x_n)``.

    Keyword arguments
    -----------------
    solver_method: str
        The method of :func:`xitorch_amd.linalg.solve` used to apply the inverse Jacobian.
    solver_kwargs: dict or None
        Its keyword arguments.
    """
    return _nonlin_solver(fcn, x0, params, _NewtonJacobian(solver_method, solver_kwargs), **kwargs)


def broyden1(fcn, x0, params=(), *, alpha=None, uv0=None, max_rank=None, **kwargs):
    """
    Solve the root finder or linear equation using the first Broyden method (van der Rotten 2003),
    with the inverse-Jacobian model applied by HIP kernels.

    Keyword arguments
    -----------------
    alpha: float or None
        The initial guess of inverse Jacobian is ``- alpha * I + u v^T``.
    uv0: tuple of tensors or str or None
        ``u`` and ``v`` above; ``"svd"`` takes them from a rank-1 SVD of the Jacobian; None: zeros.
    max_rank: int or None
        The maximum rank of inverse Jacobian approximation. If ``None``, it is ``inf``.
    """
    return _nonlin_solver(fcn, x0, params, _BroydenFirst(alpha=alpha, uv0=uv0, max_rank=max_rank), **kwargs)


def broyden2(fcn, x0, params=(), *, alpha=None, uv0=None, max_rank=None, **kwargs):
    """
    Solve the root finder or linear equation using the second Broyden method.

    Keyword arguments
    -----------------
    alpha, uv0, max_rank
        As for :func:`broyden1`.
    """
    return _nonlin_solver(fcn, x0, params, _BroydenSecond(alpha=alpha, uv0=uv0, max_rank=max_rank), **kwargs)


def linearmixing(fcn, x0, params=(), *, alpha=None, **kwargs):
    """
    Solve the root finding problem by approximating the inverse of Jacobian to be a constant scalar.

    Keyword arguments
    -----------------
    alpha: float or None
        The initial guess of inverse Jacobian is ``-alpha * I``.
    """
    return _nonlin_solver(fcn, x0, params, _LinearMixing(alpha=alpha), **kwargs)


for _f in (newton, broyden1, broyden2, linearmixing):
    _f.__doc__ += _nonlin_solver.__doc__
