"""oracle.rootfinder — CPU restatement of the reference Broyden root solver.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
xitorch/_impls/optimize/root/rootsolver.py:15-380 (quasi-Newton driver, Armijo
search, termination test) and _jacobian.py:51-222 (Broyden-1 inverse-Jacobian
model, Broyden-2 and linear-mixing models, low-rank storage), real dtypes only.
"""
import warnings
import torch


class OracleConvergenceWarning(Warning):
    pass


class LowRank:
    """G = alpha*I + sum_n c_n d_n^T as Python lists (reference: LowRankMatrix, _jacobian.py:156-199)."""

    def __init__(self, alpha, uv0=None):
        self.alpha = alpha
        self.cs, self.ds = ([], []) if uv0 is None else ([uv0[0]], [uv0[1]])
        self.dense = None

    def mv(self, v):
        if self.dense is not None:
            return torch.matmul(self.dense, v)
        res = self.alpha * v
        for c, d in zip(self.cs, self.ds):
            res += c * torch.dot(d, v)
        return res

    def rmv(self, v):
        if self.dense is not None:
            return torch.matmul(self.dense.T, v)
        res = self.alpha * v
        for c, d in zip(self.cs, self.ds):
            res += d * torch.dot(c, v)
        return res

    def append(self, c, d):
        if self.dense is not None:                                   # FullRankMatrix.append, :216-218
            self.dense += torch.outer(c, d)
            return
        self.cs.append(c)
        self.ds.append(d)
        if len(self.cs) >= c.numel():                                # :187-188 -> dense L x L
            n = c.numel()
            mat = torch.eye(n, dtype=c.dtype) * self.alpha
            for ci, di in zip(self.cs, self.ds):
                mat += torch.outer(ci, di)
            self.dense = mat

    def reduce(self, max_rank):
        # "restart" policy: drop the WHOLE history once it exceeds max_rank (quirk Q3, :191-199)
        if self.dense is None and len(self.cs) > max_rank:
            del self.cs[:]
            del self.ds[:]


class BroydenFirst:
    """reference: BroydenFirst, _jacobian.py:51-119."""

    def __init__(self, alpha=None, uv0=None, max_rank=None):
        self.alpha, self.uv0, self.max_rank = alpha, uv0, max_rank

    def setup(self, x0, y0, func):
        self.x_prev, self.y_prev = x0, y0
        if self.max_rank is None:
            self.max_rank = float("inf")
        if self.alpha is None:                                       # :76-82 (SciPy sign, quirk Q2)
            ny0 = torch.norm(y0)
            ones = torch.ones_like(ny0)
            self.alpha = 0.5 * torch.max(torch.norm(x0), ones) / ny0 if ny0 else ones
        self.Gm = LowRank(-self.alpha, self.uv0)

    def solve(self, v, tol=0):
        return self.Gm.mv(v)

    def update(self, x, y):
        dy = y - self.y_prev
        dx = x - self.x_prev
        self.Gm.reduce(self.max_rank)                                # checked BEFORE appending
        v = self.Gm.rmv(dx)
        c = dx - self.Gm.mv(dy)
        d = v / torch.dot(dy, v)
        self.Gm.append(c, d)
        self.y_prev, self.x_prev = y, x

    @property
    def rank(self):
        return len(self.Gm.cs)


def _armijo(phi, phi0, derphi0, c1=1e-4, alpha0=1, amin=0, max_niter=20):
    """reference: _scalar_search_armijo, rootsolver.py:312-357."""
    phi_a0 = phi(alpha0)
    if phi_a0 <= phi0 + c1 * alpha0 * derphi0:
        return alpha0, phi_a0
    alpha1 = -(derphi0) * alpha0**2 / 2.0 / (phi_a0 - phi0 - derphi0 * alpha0)
    phi_a1 = phi(alpha1)
    if phi_a1 <= phi0 + c1 * alpha1 * derphi0:
        return alpha1, phi_a1
    niter = 0
    while alpha1 > amin and niter < max_niter:
        factor = alpha0**2 * alpha1**2 * (alpha1 - alpha0)
        a = alpha0**2 * (phi_a1 - phi0 - derphi0 * alpha1) - alpha1**2 * (phi_a0 - phi0 - derphi0 * alpha0)
        a = a / factor
        b = -alpha0**3 * (phi_a1 - phi0 - derphi0 * alpha1) + alpha1**3 * (phi_a0 - phi0 - derphi0 * alpha0)
        b = b / factor
        alpha2 = (-b + torch.sqrt(torch.abs(b**2 - 3 * a * derphi0))) / (3.0 * a)
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


def _line_search(func, x, y, dx, smin=1e-2, counter=None):
    """reference: _nonline_line_search, rootsolver.py:272-310."""
    cache_s, cache_y, cache_phi = [0], [y], [y.norm()**2]

    def phi(s):
        if s == cache_s[0]:
            return cache_phi[0]
        v = func(x + s * dx)
        if counter is not None:
            counter[0] += 1
        p = torch.dot(v.reshape(-1), v.reshape(-1))
        cache_s[0], cache_phi[0], cache_y[0] = s, p, v
        return p

    s, _ = _armijo(phi, cache_phi[0], -cache_phi[0], amin=smin)
    if s is None:
        s = 1.0
    xn = x + s * dx
    if s == cache_s[0]:
        yn = cache_y[0]
    else:
        yn = func(xn)
        if counter is not None:
            counter[0] += 1
    return s, xn, yn, yn.norm()


def nonlin_solve(fcn, x0, params, jacobian, maxiter=None, f_tol=None, f_rtol=None, x_tol=None,
                 x_rtol=None, line_search=True, verbose=False, trace=None, **unused):
    """Quasi-Newton driver on the flattened variable (reference: _nonlin_solver, rootsolver.py:15-149)."""
    if maxiter is None:
        maxiter = 100 * (x0.numel() + 1)
    xshape = x0.shape
    # complex unknowns: real vector [Re x; Im x] of twice the length (rootsolver.py:52-73)
    if torch.is_complex(x0):
        ravel = lambda t: torch.cat((t.real, t.imag), dim=0).reshape(-1)

        def pack(v):
            n = len(v) // 2
            return (v[:n] + 1j * v[n:]).reshape(xshape)
    else:
        ravel = lambda t: t.reshape(-1)
        pack = lambda v: v.reshape(xshape)
    func = lambda x: ravel(fcn(pack(x), *params))
    nfev = [1]
    x = ravel(x0)
    y = func(x)
    y_norm = y.norm()
    f_tol = 1e-6 if f_tol is None else f_tol
    f_rtol = float("inf") if f_rtol is None else f_rtol
    x_tol = 1e-6 if x_tol is None else x_tol
    x_rtol = float("inf") if x_rtol is None else x_rtol
    f0_norm = y_norm
    if y_norm == 0:
        return x.reshape(xshape)
    jacobian.setup(x, y, func)
    gamma, eta_max, eta_thresh, eta = 0.9, 0.9999, 0.1, 1e-3
    converged = False
    best_ynorm, best_x, best_dx, best_it = y_norm, x, x.norm(), 0
    niter = 0
    for i in range(maxiter):
        niter = i + 1
        dx = -jacobian.solve(y, tol=min(eta, eta * y_norm))
        dx_norm = dx.norm()
        if dx_norm == 0:
            raise ValueError("Jacobian inversion yielded zero vector.")
        if line_search:
            s, xnew, ynew, y_norm_new = _line_search(func, x, y, dx, counter=nfev)
        else:
            xnew = x + dx
            ynew = func(xnew)
            nfev[0] += 1
            y_norm_new = ynew.norm()
        if y_norm_new < best_ynorm:
            best_x, best_dx, best_ynorm, best_it = xnew, dx_norm, y_norm_new, i + 1
        jacobian.update(xnew.clone(), ynew)
        stop = (dx_norm < x_tol) and (dx_norm < x_rtol * xnew.norm()) and \
               (ynew.norm() < f_tol) and (ynew.norm() < f_rtol * f0_norm)   # :359-380 AND-termination
        if stop:
            converged = True
            break                                                     # NB: breaks BEFORE x = xnew (quirk Q1)
        eta_A = float(gamma * (y_norm_new / y_norm)**2)
        g2 = gamma * eta * eta
        eta = min(eta_max, eta_A) if g2 < eta_thresh else min(eta_max, max(eta_A, g2))
        y_norm, x, y = y_norm_new, xnew, ynew
    if trace is not None:
        trace.update(niter=niter, nfev=nfev[0], converged=converged, rank=getattr(jacobian, "rank", None),
                     best_ynorm=float(best_ynorm))
    if not converged:
        warnings.warn(OracleConvergenceWarning("rootfinder: no convergence after %d iterations" % maxiter))
        x = best_x
    return pack(x)


class BroydenSecond(BroydenFirst):
    """reference: BroydenSecond, _jacobian.py:120-137 ("bad" Broyden: v = dy, d = dy / |dy|^2)."""

    def update(self, x, y):
        dy = y - self.y_prev
        dx = x - self.x_prev
        dynorm = dy.norm()
        self.Gm.reduce(self.max_rank)
        c = dx - self.Gm.mv(dy)
        d = dy / (dynorm * dynorm)
        self.Gm.append(c, d)
        self.y_prev, self.x_prev = y, x


class LinearMixing:
    """reference: LinearMixing, _jacobian.py:139-154: the inverse Jacobian is the constant -alpha*I."""

    def __init__(self, alpha=None):
        self.alpha = -1.0 if alpha is None else alpha

    def setup(self, x0, y0, func):
        pass

    def solve(self, v, tol=0):
        return -v * self.alpha

    def update(self, x, y):
        pass


def broyden1(fcn, x0, params=(), alpha=None, uv0=None, max_rank=None, **kwargs):
    """reference: broyden1, rootsolver.py:176-206."""
    return nonlin_solve(fcn, x0, params, BroydenFirst(alpha=alpha, uv0=uv0, max_rank=max_rank), **kwargs)


def broyden2(fcn, x0, params=(), alpha=None, uv0=None, max_rank=None, **kwargs):
    """reference: broyden2, rootsolver.py:209-238."""
    return nonlin_solve(fcn, x0, params, BroydenSecond(alpha=alpha, uv0=uv0, max_rank=max_rank), **kwargs)


def linearmixing(fcn, x0, params=(), alpha=None, **kwargs):
    """reference: linearmixing, rootsolver.py:241-256."""
    return nonlin_solve(fcn, x0, params, LinearMixing(alpha=alpha), **kwargs)
