"""Fixed-point and minimisation methods that share the root-finder's implicit backward:
Anderson acceleration, gradient descent with momentum, Adam.

"Next" rows of the scope: same names, options and stopping rules as the reference
(xitorch/_impls/optimize/equilibrium.py:9-134, minimizer.py:5-208).  They are short host loops of
element-wise updates on the caller's device (the user function dominates), so they run on plain torch
ops; the operator-level work of their backward pass goes through the native `solve`.
"""
import warnings
import torch
from xitorch_amd._util import ConvergenceWarning
from xitorch_amd.optimize.native_root import _Termination, _Reduce

__all__ = ["anderson_acc", "gd", "adam"]


def anderson_acc(fcn, x0, params, feat_ndims=1, msize=5, beta=1.0, lmbda=1e-4, maxiter=None, f_tol=None,
                 f_rtol=None, x_tol=None, x_rtol=None, custom_terminator=None, verbose=False):
    """
    Solve the equilibrium (or fixed-point iteration) problem using Anderson acceleration
    (Walker & Ni, SIAM J. Numer. Anal. 49, 1715).

    Keyword arguments
    -----------------
    feat_ndims: int
        The number of dimensions at the end that describe the features (i.e. non-batch dimensions)
    msize: int
        The maximum number of previous iterations we should save for the algorithm
    beta: float
        The damped or overcompensated parameters
    lmbda: float
        Small number to ensure invertability of the matrix
    maxiter: int or None
        Maximum number of iterations, or inf if it is set to None.
    f_tol, f_rtol, x_tol, x_rtol: float or None
        Tolerances on ``f - x`` and on the step (absolute / relative)
    verbose: bool
        Options for verbosity
    """
    featshape = x0.shape[x0.dim() - feat_ndims:]
    bshape = x0.shape[:x0.dim() - feat_ndims]
    nfeat = 1
    for d in featshape:
        nfeat *= d
    dtype, device = x0.dtype, x0.device
    if maxiter is None:
        maxiter = 100 * (nfeat + 1)
    flat = lambda x: x.reshape(*bshape, -1)
    unflat = lambda x: x.reshape(*bshape, *featshape)
    g = lambda xn: flat(fcn(unflat(xn), *params))

    xn = flat(x0)
    fn = g(xn)
    xs = torch.zeros((*bshape, msize, nfeat), dtype=dtype, device=device)
    fs = torch.zeros((*bshape, msize, nfeat), dtype=dtype, device=device)
    xs[..., 0, :], fs[..., 0, :] = xn, fn
    xn = fn
    fn = g(xn)
    xs[..., 1, :], fs[..., 1, :] = xn, fn
    H = torch.zeros((*bshape, msize + 1, msize + 1), dtype=dtype, device=device)
    rhs = torch.zeros((*bshape, msize + 1, 1), dtype=dtype, device=device)
    H[..., 0, 1:] = 1.0
    H[..., 1:, 0] = 1.0
    rhs[..., 0, :] = 1.0
    dev0 = float((fn - xn).norm())
    stop = custom_terminator if custom_terminator is not None else \
        _Termination(f_tol, f_rtol, dev0, x_tol, x_rtol, _Reduce(None))
    if dev0 == 0:
        return x0
    converged = False
    for k in range(2, maxiter):
        n = min(k, msize)
        G = fs[..., :n, :] - xs[..., :n, :]
        H[..., 1:n + 1, 1:n + 1] = torch.einsum("...nf,...mf->...nm", G, G) + \
            lmbda * torch.eye(n, dtype=dtype, device=device)
        alpha = torch.linalg.solve(H[..., :n + 1, :n + 1], rhs[..., :n + 1, :])[..., 1:n + 1, 0]
        xnew = torch.einsum("...n,...nf->...f", alpha, fs[..., :n, :]) * beta + \
            torch.einsum("...n,...nf->...f", alpha, xs[..., :n, :]) * (1 - beta)
        fnew = g(xnew)
        xs[..., k % msize, :], fs[..., k % msize, :] = xnew, fnew
        done = stop.check(xnew, fnew - xnew, xnew - xn)
        if verbose and (k < 10 or k % 10 == 0 or done):
            print("%6d: |dx|=%.3e, |f-x|=%.3e" % (k, (xnew - xn).norm(), (fnew - xnew).norm()))
        xn = xnew
        if done:
            converged = True
            break
    if not converged:
        warnings.warn(ConvergenceWarning("The rootfinder does not converge after %d iterations." % maxiter))
    return unflat(xn)


class _MinStop:
    """OR-termination + best-x tracking of the minimisers (reference: minimizer.py:149-208)."""

    def __init__(self, f_tol, f_rtol, x_tol, x_rtol, verbose):
        self.f_tol, self.f_rtol, self.x_tol, self.x_rtol, self.verbose = f_tol, f_rtol, x_tol, x_rtol, verbose
        self.ever, self.max_i = False, -1
        self.best = (float("inf"), None, float("inf"), float("inf"))     # f, x, |dx|, |df|

    def to_stop(self, i, xnext, x, f, fprev):
        xnorm = float(x.detach().norm())
        dxnorm = float((x - xnext).detach().norm())
        fabs, fval = float(f.detach().abs()), float(f.detach())
        df = float((fprev - f).detach().abs())
        conv = dxnorm < self.x_tol or dxnorm < self.x_rtol * xnorm or df < self.f_tol or df < self.f_rtol * fabs
        if self.verbose:
            if i == 0:
                print("   #:             f |        dx,        df")
            if conv:
                print("Finish with convergence")
            if i == 0 or ((i + 1) % 10) == 0 or conv:
                print("%4d: %.6e | %.3e, %.3e" % (i + 1, fval, dxnorm, df))
        res = i > 0 and conv
        self.ever = self.ever or res
        self.max_i = max(self.max_i, i)
        if fval < self.best[0]:
            self.best = (fval, x, dxnorm, df)
        return res

    def get_best_x(self, x):
        if not self.ever and self.max_i > -1:
            warnings.warn("The minimizer does not converge after %d iterations. Best |dx|=%.4e, |df|=%.4e, f=%.4e"
                          % (self.max_i, self.best[2], self.best[3], self.best[0]))
            return self.best[1]
        return x


def gd(fcn, x0, params, step=1e-3, gamma=0.9, maxiter=1000, f_tol=0.0, f_rtol=1e-8, x_tol=0.0, x_rtol=1e-8,
       verbose=False, **unused):
    r"""
    Vanilla gradient descent with momentum (OR stopping criteria):
    :math:`v_{t+1} = \gamma v_t - \eta \nabla f(x_t)`, :math:`x_{t+1} = x_t + v_{t+1}`.

    Keyword arguments
    -----------------
    step: float
        The step size :math:`\eta`
    gamma: float
        The momentum factor
    maxiter: int
        Maximum number of iterations.
    f_tol, f_rtol, x_tol, x_rtol: float
        Absolute / relative tolerances on the change of ``f`` and of ``x``
    """
    x = x0.clone()
    stop = _MinStop(f_tol, f_rtol, x_tol, x_rtol, verbose)
    fprev = torch.tensor(0.0, dtype=x0.dtype, device=x0.device)
    v = torch.zeros_like(x)
    for i in range(maxiter):
        f, dfdx = fcn(x, *params)
        v = (gamma * v - step * dfdx).detach()
        xprev = x.detach()
        x = (xprev + v).detach()
        if stop.to_stop(i, x, xprev, f, fprev):
            break
        fprev = f
    return stop.get_best_x(x)


def adam(fcn, x0, params, step=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, maxiter=1000, f_tol=0.0, f_rtol=1e-8,
         x_tol=0.0, x_rtol=1e-8, verbose=False, **unused):
    r"""
    Adam optimizer (Kingma & Ba 2015), OR stopping criteria.

    Keyword arguments
    -----------------
    step: float
        The step size
    beta1, beta2: float
        Exponential decay rates of the first / second moment estimates
    eps: float
        Small number to prevent division by 0.
    maxiter: int
        Maximum number of iterations.
    f_tol, f_rtol, x_tol, x_rtol: float
        Absolute / relative tolerances on the change of ``f`` and of ``x``
    """
    x = x0.clone()
    stop = _MinStop(f_tol, f_rtol, x_tol, x_rtol, verbose)
    fprev = torch.tensor(0.0, dtype=x0.dtype, device=x0.device)
    m, v = torch.zeros_like(x), torch.zeros_like(x)
    b1t, b2t = beta1, beta2
    for i in range(maxiter):
        f, dfdx = fcn(x, *params)
        f, dfdx = f.detach(), dfdx.detach()
        m = beta1 * m + (1 - beta1) * dfdx
        v = beta2 * v + (1 - beta2) * dfdx ** 2
        mhat, vhat = m / (1 - b1t), v / (1 - b2t)
        b1t, b2t = b1t * beta1, b2t * beta2
        xprev = x.detach()
        x = (xprev - step * mhat / (vhat ** 0.5 + eps)).detach()
        if stop.to_stop(i, x, xprev, f, fprev):
            break
        fprev = f
    return stop.get_best_x(x)
