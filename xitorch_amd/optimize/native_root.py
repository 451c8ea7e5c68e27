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

so an apply is two streaming passes over 2*rank*L elements regardless of the rank; the rank-1
update writes v = G^T dx and c = dx - G dy straight into the new rows of the buffers, and the
division d = v / <dy, v> (_jacobian.py:118) is a per-term scalar applied to the multi-dot's
coefficients, never a pass over L.  The driver's norms and dot products (rootsolver.py:100,113,
286-290,375-380) come from the fused reduction xk_vec_dots: an outer iteration whose first trial
step is accepted reads {|f|^2, |dx|^2, |x|^2} with ONE host sync (the reference: about eight).
With a ``process_group`` the flat vector is sharded over the ranks (batch sharding) and every
group of inner products is completed by ONE small all-reduce(SUM) (RCCL) — the rank-vector of the
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
def _native_vec(t):
    return t.is_cuda and t.dtype in (torch.float64, torch.float32)


class _Reduce:
    """Global inner products of (possibly sharded) flat vectors.  Device vectors go through the fused HIP
    reduction (several products per pass, result left on the device); host vectors — the driver itself is
    device-agnostic and also serves CPU callers of newton / linearmixing — through torch."""

    def __init__(self, group):
        self.group = group

    def _sum(self, t):
        return allreduce_sum_(t, self.group)

    def dots_dev(self, pairs):
        """device tensor (float64) of <a_i, b_i>, all-reduced over the group; no host sync"""
        a0 = pairs[0][0]
        if _native_vec(a0):
            t = K.vec_dots([(a.reshape(-1), b.reshape(-1)) for a, b in pairs])
        else:
            t = torch.stack([torch.dot(a.reshape(-1), b.reshape(-1)).double() for a, b in pairs])
        return self._sum(t)

    def dots(self, pairs):
        """the same as host floats: ONE sync for all of them"""
        return self.dots_dev(pairs).tolist()

    def dot(self, a, b):
        return self.dots_dev([(a, b)])[0]

    def norm(self, a):
        return torch.sqrt(self.dots_dev([(a, a)])[0])

    def total_numel(self, a):
        if self.group is None:
            return a.numel()
        t = torch.tensor([float(a.numel())], dtype=torch.float64, device=a.device)
        return int(self._sum(t).item())


def _axpy(u0, g0, u1=None, g1=0.0):
    """g0*u0 + g1*u1 as a new flat vector (native kernel on the device)."""
    if _native_vec(u0) and u0.is_contiguous() and (u1 is None or u1.is_contiguous()):
        return K.broyden_axpy(torch.empty_like(u0), u0, g0, u1, g1)
    return u0 * g0 if u1 is None else u0 * g0 + u1 * g1


# ------------------------------------------------------------------------------ low-rank model
class _LowRank:
    """G = alpha*I + sum_n c_n d_n^T with d_n = v_n * inv_n, in two growing device buffers (rows c_n / v_n) and
    a device vector of the per-term scalars inv_n (reference: LowRankMatrix / FullRankMatrix,
    _jacobian.py:156-222, where c_n, d_n are Python lists and every apply loops over them)."""

    def __init__(self, alpha, uv0, L, dtype, device, red, total_L):
        # A variable on a HIP device is served by the HIP kernels and by nothing else (a missing library raises).  A
        # variable that lives in HOST memory is applied with torch ops on the host, like the reference runs on
        # whatever device its tensors are on (_jacobian.py:156-222): device dispatch, not a fallback.
        self.native = device.type == "cuda"
        if dtype not in (torch.float64, torch.float32):
            raise NativeLibraryError("xitorch_amd Broyden supports float64/float32, got %s" % dtype)
        self.alpha = float(alpha)
        self.L, self.Lp = L, pad_len(L)
        self.total_L = total_L
        self.dtype, self.device, self.red = dtype, device, red
        self.cap, self.rank = 8, 0
        self.C = torch.zeros((1, self.cap, self.Lp), dtype=dtype, device=device)
        self.D = torch.zeros((1, self.cap, self.Lp), dtype=dtype, device=device)
        self.dinv = torch.ones((self.cap,), dtype=dtype, device=device)
        self.dense = None
        if uv0 is not None:
            self.append_rows(uv0[0], uv0[1], None)

    def _grow(self):
        new = self.cap * 2
        for name in ("C", "D"):
            old = getattr(self, name)
            buf = torch.zeros((1, new, self.Lp), dtype=self.dtype, device=self.device)
            buf[:, :self.cap].copy_(old)
            setattr(self, name, buf)
        dn = torch.ones((new,), dtype=self.dtype, device=self.device)
        dn[:self.cap].copy_(self.dinv)
        self.dinv, self.cap = dn, new

    def _vec(self, v):
        """(1, 1, L) view of a flat vector for the multi-dot (contiguous, 16 B aligned rows)"""
        v = v.reshape(-1)
        if not v.is_contiguous():
            v = v.contiguous()
        return v.reshape(1, 1, -1)

    def _coef(self, rows, v):
        """coefficients <rows_n, v> for n < rank, all-reduced over the shards: one multi-dot (K1, split-contraction)"""
        if self.native:
            coef = K.dense_mm(rows[:, :self.rank, :self.L], self._vec(v))          # (1, 1, rank)
        else:
            coef = torch.mv(rows[0, :self.rank, :self.L], v.reshape(-1)).reshape(1, 1, -1)
        if self.red.group is not None:
            self.red._sum(coef)
        return coef

    def apply_into(self, out, v, transpose=False, g_v=None, extra=None, g_extra=0.0, gamma=1.0):
        """out = g_v * v + g_extra * extra + gamma * (low-rank part of G (or G^T)) v.
        G v = alpha v + sum_n c_n inv_n <v_n, v>;  G^T v = alpha v + sum_n v_n inv_n <c_n, v>."""
        first, second = (self.C, self.D) if transpose else (self.D, self.C)
        v = v.reshape(-1).contiguous()            # the update kernel reads it with unit stride
        if extra is not None:
            extra = extra.reshape(-1).contiguous()
        if not self.native:
            return self._apply_host(out, v, first, second, g_v, extra, g_extra, gamma)
        if self.rank == 0:
            return K.broyden_axpy(out, v, g_v, extra, g_extra)
        coef = self._coef(first, v)
        return K.broyden_axpy(out, v, g_v, extra, g_extra, V=second[0], coef=coef, scale=self.dinv, k=self.rank,
                              gamma=gamma)

    def _apply_host(self, out, v, first, second, g_v, extra, g_extra, gamma):
        """the same update on host memory: out = g_v v + g_extra extra + gamma sum_n second_n inv_n <first_n, v>"""
        res = v * g_v
        if extra is not None:
            res = res + extra * g_extra
        if self.rank > 0:
            w = self._coef(first, v).reshape(-1) * self.dinv[:self.rank]
            res = res + gamma * torch.mv(second[0, :self.rank, :self.L].T, w)
        out.copy_(res)
        return out

    def mv(self, v, sign=1.0):
        if self.dense is not None:
            return torch.matmul(self.dense, v) * sign
        out = torch.empty(self.L, dtype=self.dtype, device=self.device)
        return self.apply_into(out, v, False, g_v=sign * self.alpha, gamma=sign)

    def rmv(self, v):
        if self.dense is not None:
            return torch.matmul(self.dense.T, v)
        out = torch.empty(self.L, dtype=self.dtype, device=self.device)
        return self.apply_into(out, v, True, g_v=self.alpha)

    def new_rows(self):
        """views of the next free rows of C and D (growing the buffers when full)"""
        if self.rank == self.cap:
            self._grow()
        return self.C[0, self.rank, :self.L], self.D[0, self.rank, :self.L]

    def commit(self, inv):
        """the rows handed out by new_rows() are filled: make them term number `rank`; inv: device scalar or None (= 1)"""
        if inv is None:
            self.dinv[self.rank] = 1.0
        else:
            self.dinv[self.rank] = inv.to(self.dtype)
        self.rank += 1
        if self.rank >= self.total_L and self.red.group is None:       # _jacobian.py:187-188: dense from here on
            n = self.L
            mat = torch.eye(n, dtype=self.dtype, device=self.device) * self.alpha
            mat += torch.matmul(self.C[0, :self.rank, :n].T, self.D[0, :self.rank, :n] * self.dinv[:self.rank, None])
            self.dense = mat

    def append_rows(self, c, v, inv):
        if self.dense is not None:                                    # FullRankMatrix.append
            self.dense += torch.outer(c, v if inv is None else v * inv.to(self.dtype))
            return
        crow, drow = self.new_rows()
        crow.copy_(c)
        drow.copy_(v)
        self.commit(inv)

    def reduce(self, max_rank):
        # "restart": forget everything once the rank EXCEEDS max_rank (checked before appending)
        if self.dense is None and self.rank > max_rank:
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
            y2, x2 = red.dots([(y0, y0), (x0, x0)])
            ny0 = y2 ** 0.5
            self.alpha = 0.5 * max(x2 ** 0.5, 1.0) / ny0 if ny0 else 1.0
        if isinstance(self.uv0, str) and self.uv0 == "svd":
            self.uv0 = _svd_uv0(func, x0)
        self.Gm = _LowRank(-float(self.alpha), self.uv0, x0.numel(), x0.dtype, x0.device, red,
                           red.total_numel(x0))

    def solve(self, v, tol=0):
        return self.Gm.mv(v)

    def neg_solve(self, v, tol=0):
        """-G v in one pass (the driver's dx = -jacobian.solve(y), rootsolver.py:98)"""
        return self.Gm.mv(v, sign=-1.0)

    def update(self, x, y):
        dy = _axpy(y, 1.0, self.y_prev, -1.0)
        dx = _axpy(x, 1.0, self.x_prev, -1.0)
        self._update(dx, dy)
        self.y_prev, self.x_prev = y, x

    def _update(self, dx, dy):
        Gm = self.Gm
        Gm.reduce(self.max_rank)
        if Gm.dense is not None:
            v = Gm.rmv(dx)
            Gm.append_rows(dx - Gm.mv(dy), v, 1.0 / self.red.dot(dy, v))
            return
        crow, vrow = Gm.new_rows()
        Gm.apply_into(vrow, dx, transpose=True, g_v=Gm.alpha)                                  # v = G^T dx
        Gm.apply_into(crow, dy, transpose=False, g_v=-Gm.alpha, extra=dx, g_extra=1.0, gamma=-1.0)   # c = dx - G dy
        Gm.commit(1.0 / self.red.dot(dy, vrow))                                                # d = v / <dy, v>

    @property
    def rank(self):
        return self.Gm.rank


class _BroydenSecond(_BroydenFirst):
    """reference: BroydenSecond, _jacobian.py:121-137."""

    def _update(self, dx, dy):
        Gm = self.Gm
        Gm.reduce(self.max_rank)
        inv = 1.0 / self.red.dot(dy, dy)                              # d = dy / |dy|^2
        if Gm.dense is not None:
            Gm.append_rows(dx - Gm.mv(dy), dy, inv)
            return
        crow, vrow = Gm.new_rows()
        Gm.apply_into(crow, dy, transpose=False, g_v=-Gm.alpha, extra=dx, g_extra=1.0, gamma=-1.0)   # c = dx - G dy
        vrow.copy_(dy)
        Gm.commit(inv)


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
def _cubic_backtrack(s_old, f_old, s_new, f_new, f0, slope0):
    """Minimiser of the cubic that interpolates f(0) = f0, f'(0) = slope0, f(s_old) = f_old, f(s_new) = f_new
    (Nocedal & Wright, Numerical Optimization, eq. 3.59; the expression order follows SciPy's
    `scalar_search_armijo`, from which the reference's line search derives, so that the trial steps — and with
    them the number of function evaluations — are bit-identical to the reference's)."""
    d_old = f_old - f0 - slope0 * s_old
    d_new = f_new - f0 - slope0 * s_new
    scale = s_old ** 2 * s_new ** 2 * (s_new - s_old)
    c3 = (s_old ** 2 * d_new - s_new ** 2 * d_old) / scale
    c2 = (-s_old ** 3 * d_new + s_new ** 3 * d_old) / scale
    return (-c2 + abs(c2 ** 2 - 3 * c3 * slope0) ** 0.5) / (3.0 * c3)


def _armijo(phi, phi0, slope0, c1=1e-4, first_step=1.0, smallest=0.0, max_cubic=20):
    """Armijo backtracking on the scalar function ``phi``: accept the first trial step s with
    phi(s) <= phi0 + c1 s slope0.  Trials: the full step, the minimiser of the interpolating quadratic, then
    minimisers of interpolating cubics, each safeguarded to lie in [s/2 .. 0.96 s] of its predecessor
    (behaviour of the reference's `_scalar_search_armijo`, rootsolver.py:312-357, itself SciPy's
    scalar_search_armijo).  Returns (step or None, phi at the last trial).  Scalars are host floats."""
    def acceptable(s, val):
        return val <= phi0 + c1 * s * slope0

    s_old, f_old = first_step, phi(first_step)
    if acceptable(s_old, f_old):
        return s_old, f_old
    s_new = -slope0 * s_old ** 2 / 2.0 / (f_old - phi0 - slope0 * s_old)      # quadratic through f0, slope0, f_old
    f_new = phi(s_new)
    if acceptable(s_new, f_new):
        return s_new, f_new
    trial, f_trial = s_new, f_new
    for _ in range(max_cubic):
        if not s_new > smallest:
            return None, f_new              # the step shrank below the floor: give up (caller takes the full step)
        trial = _cubic_backtrack(s_old, f_old, s_new, f_new, phi0, slope0)
        f_trial = phi(trial)
        if acceptable(trial, f_trial):
            return trial, f_trial
        if (s_new - trial) > s_new / 2.0 or (1 - trial / s_new) < 0.96:
            trial = s_new / 2.0             # too far from / too close to the previous trial: halve instead
        s_old, f_old, s_new, f_new = s_new, f_new, trial, f_trial
    return trial, f_trial                   # out of cubic trials: hand back the last one


def _line_search(func, x, y, dx, red, phi0, smin=1e-2):
    """Armijo search along dx on phi(s) = |f(x + s dx)|^2 (reference: _nonline_line_search, rootsolver.py:272-310).
    Every trial costs one function evaluation and ONE host sync: the fused reduction delivers |f|^2 together with
    |x + s dx|^2 and (first trial only) |dx|^2, which is everything the termination test needs afterwards.
    Returns (s, xnew, ynew, {"y2", "x2", "dx2"})."""
    state = {"s": 0.0, "y": y, "x": x, "phi": phi0, "x2": None, "dx2": None}

    def phi(s):
        if s == state["s"]:
            return state["phi"]
        xt = _axpy(x, 1.0, dx, s)
        v = func(xt)
        if state["dx2"] is None:
            p, x2, dx2 = red.dots([(v, v), (xt, xt), (dx, dx)])
            state["dx2"] = dx2
            if dx2 == 0:
                # |dx|^2 arrives with the first trial's reduction: raise here, one function evaluation after the
                # reference would (rootsolver.py:100 tests it before the search), not after a full backtracking
                raise ValueError("Jacobian inversion yielded zero vector. "
                                 "This indicates a bug in the Jacobian approximation.")
        else:
            p, x2 = red.dots([(v, v), (xt, xt)])
        state.update(s=s, y=v, x=xt, phi=p, x2=x2)
        return p

    s, _ = _armijo(phi, phi0, -phi0, smallest=smin)
    if s is None:
        s = 1.0
    if s != state["s"]:                     # the accepted step is not the last one evaluated: evaluate it
        phi(s)
    return s, state["x"], state["y"], {"y2": state["phi"], "x2": state["x2"], "dx2": state["dx2"]}


class _Termination:
    """reference: TerminationCondition, rootsolver.py:359-380."""

    def __init__(self, f_tol, f_rtol, f0_norm, x_tol, x_rtol, red):
        self.f_tol = 1e-6 if f_tol is None else f_tol
        self.f_rtol = float("inf") if f_rtol is None else f_rtol
        self.x_tol = 1e-6 if x_tol is None else x_tol
        self.x_rtol = float("inf") if x_rtol is None else x_rtol
        self.f0_norm, self.red = f0_norm, red

    def check_norms(self, xn, yn, dxn):
        return (dxn < self.x_tol) and (dxn < self.x_rtol * xn) and (yn < self.f_tol) and \
            (yn < self.f_rtol * self.f0_norm)

    def check(self, x, y, dx):
        x2, y2, dx2 = self.red.dots([(x, x), (y, y), (dx, dx)])
        return self.check_norms(x2 ** 0.5, y2 ** 0.5, dx2 ** 0.5)


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
    xshape = x0.shape
    if torch.is_complex(x0):
        # a complex unknown is solved as the real vector [Re x; Im x] of twice the length, exactly like the
        # reference (rootsolver.py:52-73): real parts first, then imaginary parts
        def ravel(t):
            return torch.cat((t.real, t.imag), dim=0).reshape(-1).contiguous()

        def pack(v):
            n = v.numel() // 2
            return torch.complex(v[:n], v[n:]).reshape(xshape)
    else:
        def ravel(t):
            # unit stride: the fused reductions / updates take raw pointers (a strided 1-D view such as M[:, 0]
            # stays strided under reshape)
            return t.reshape(-1).contiguous()

        def pack(v):
            return v.reshape(xshape)
    xdtype = ravel(x0).dtype

    def func(x):
        out = ravel(fcn(pack(x), *params))
        return out if out.dtype == xdtype else out.to(xdtype)      # (torch would promote in the reference's dots)
    nfev = [0]

    def cfunc(x):
        nfev[0] += 1
        return func(x)

    x = ravel(x0)
    y = cfunc(x)
    y2, x2 = red.dots([(y, y), (x, x)])
    y_norm = y2 ** 0.5
    stop_cond = custom_terminator if custom_terminator is not None else \
        _Termination(f_tol, f_rtol, y_norm, x_tol, x_rtol, red)
    if y_norm == 0:
        return pack(x)
    jacobian.setup(x, y, cfunc, red)
    neg_solve = getattr(jacobian, "neg_solve", None)

    gamma, eta_max, eta_threshold, eta = 0.9, 0.9999, 0.1, 1e-3
    converge = False
    best_ynorm, best_x, best_dxnorm, best_iter = y_norm, x, x2 ** 0.5, 0
    niter = 0
    for i in range(maxiter):
        niter = i + 1
        tol = min(eta, eta * y_norm)
        dx = neg_solve(y, tol=tol) if neg_solve is not None else -jacobian.solve(y, tol=tol)
        if line_search:
            s, xnew, ynew, st = _line_search(cfunc, x, y, dx, red, y_norm * y_norm)
        else:
            s = 1.0
            xnew = _axpy(x, 1.0, dx, 1.0)
            ynew = cfunc(xnew)
            st = dict(zip(("y2", "x2", "dx2"), red.dots([(ynew, ynew), (xnew, xnew), (dx, dx)])))
        # the norms of this iteration, all from the one reduction that the accepted trial step produced
        y_norm_new, x_norm_new, dx_norm = st["y2"] ** 0.5, st["x2"] ** 0.5, st["dx2"] ** 0.5
        if dx_norm == 0:
            raise ValueError("Jacobian inversion yielded zero vector. "
                             "This indicates a bug in the Jacobian approximation.")
        if y_norm_new < best_ynorm:
            best_x, best_dxnorm, best_ynorm, best_iter = xnew, dx_norm, y_norm_new, i + 1
        jacobian.update(xnew, ynew)
        if isinstance(stop_cond, _Termination):
            to_stop = stop_cond.check_norms(x_norm_new, y_norm_new, dx_norm)
        else:
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
    return pack(x)


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
