"""Linear solvers: dense `exactsolve` and the native (HIP) Krylov methods cg / bicgstab / gmres.

Drop-in for the method functions of the reference (xitorch/_impls/linalg/solve.py): same names,
signatures ``f(A, B, E, M, **options) -> X``, options, stopping rule
(``all(|r_col| < max(rtol |b_col|, atol))``), best-iterate return (quirk Q10), true-residual
refresh every ``resid_calc_every`` iterations, ``_safedenom`` patching (Q9), normal-equation
fallback for ``posdef=False`` and ``ConvergenceWarning`` on non-convergence.

Every (batch member, column) pair is one *system*; all S = Bt*ncols systems advance in lock step.
Vectors live in padded (Bt, ncols, ld) panels (== (S, ld) arrays); per-system scalars never leave
the device; an iteration is a fixed sequence of fused kernels of libxitorch_amd.so
(xk_bicg_*, xk_cg_*, xk_kry_*) around the operator applies (xk_dense_mm / xk_banded_mm for the
native operators, the operator's own ``.mm`` otherwise) and ONE host sync (the stopping test).
With per-column shifts ``E`` the reference moves the columns to a leading axis (solve.py:575-604,
Q11); here the shift is simply a per-system scalar.
"""
import warnings
import torch
from xitorch_amd import kernels as K
from xitorch_amd._capi import NativeLibraryError, fn, ptr, stream_ptr, check, suffix
from xitorch_amd._util import bcast_shape, pad_shapes, ConvergenceWarning
from xitorch_amd.linalg._panel import PanelOperator, pad_len, to_panel, from_panel
from xitorch_amd.dist import allreduce_max_, all_ranks_agree_true

__all__ = ["exactsolve", "custom_exactsolve", "cg", "bicgstab", "gmres", "scipy_gmres", "broyden1_solve", "get_batchdims"]


def _in_host_memory(A):
    """Device dispatch (the reference runs a method on whatever device the operator lives on): operators in HOST memory
    are served by host_krylov.py; everything else — any device tensor — by the HIP kernels below and by nothing else:
    `_Problem` raises for a device it has no kernels for, `_capi.lib()` raises when the library is missing."""
    return torch.device(A.device).type == "cpu"


def get_batchdims(A, B, E, M):
    """Broadcast batch shape of the solution (reference: _get_batchdims, solve.py:540-549)."""
    shapes = [A.shape[:-2], B.shape[:-2]]
    if E is not None:
        shapes.append(E.shape[:-1])
        if M is not None:
            shapes.append(M.shape[:-2])
    return bcast_shape(*shapes)


# ------------------------------------------------------------------------------- dense
def exactsolve(A, B, E, M):
    """Solve by building the full matrices (reference: exactsolve, solve.py:481-512)."""
    if E is None:
        return torch.linalg.solve(A.fullmatrix(), B)
    if M is None:
        return _solve_ABE(A.fullmatrix(), B, E)
    L = torch.linalg.cholesky(M.fullmatrix())
    Linv = torch.inverse(L)
    LinvH = Linv.transpose(-2, -1).conj()
    A2 = torch.matmul(Linv, A.mm(LinvH))
    X2 = _solve_ABE(A2, torch.matmul(Linv, B), E)
    return torch.matmul(LinvH, X2)


def _solve_ABE(A, B, E):
    """Column c solves (A - E_c I) x_c = b_c; columns become a leading batch axis
    (reference: _solve_ABE, solve.py:514-537, including the one jittered retry)."""
    na = A.shape[-1]
    BA, BB, BE = pad_shapes(A.shape[:-2], B.shape[:-2], E.shape[:-1])
    Ec = E.reshape(1, *BE, E.shape[-1]).transpose(0, -1)                 # (ncols, *BE, 1)
    Bc = B.reshape(1, *BB, *B.shape[-2:]).transpose(0, -1)               # (ncols, *BB, na, 1)
    AE = A - torch.diag_embed(Ec.repeat_interleave(repeats=na, dim=-1), dim1=-2, dim2=-1)
    try:
        r = torch.linalg.solve(AE, Bc)
    except torch._C._LinAlgError:
        eps = torch.finfo(A.dtype).eps
        jitter = 10 * eps * torch.max(AE.reshape(*AE.shape[:-2], -1), dim=-1)[0][..., None, None]
        AE = AE + torch.eye(na, dtype=A.dtype, device=A.device) * jitter
        r = torch.linalg.solve(AE, Bc)
    return r.transpose(0, -1).squeeze(0)


def custom_exactsolve(A, B, E=None, M=None, **options):
    return exactsolve(A, B, E, M)


# ------------------------------------------------------------------------------- problem set-up
class _Problem:
    """The linear problem in panel layout: apply(X) = A X - E * (M X), optionally the normal
    equations (reference: _setup_linear_problem, solve.py:560-643)."""

    def __init__(self, A, B, E, M, bdims, posdef, need_hermit):
        dev = torch.device(A.device)
        if dev.type != "cuda":
            raise NativeLibraryError("xitorch_amd native solvers run on a HIP device only (operator is on %s)" % dev)
        if A.dtype not in (torch.float64, torch.float32, torch.complex128, torch.complex64):
            raise NativeLibraryError("xitorch_amd native solvers support float64/float32/complex128/complex64, "
                                     "got %s" % A.dtype)
        self.dtype, self.device = A.dtype, dev
        self.cplx = A.dtype.is_complex
        # element type of the kernels' scalars / partials, and complex elements per 16 B vector (block sizing)
        self.rdtype = {torch.complex128: torch.float64, torch.complex64: torch.float32}.get(A.dtype, A.dtype)
        self.vec_elems = {torch.float64: 2, torch.float32: 4, torch.complex128: 1, torch.complex64: 2}[A.dtype]
        self.bdims = list(bdims)
        self.N = A.shape[-1]
        self.nc = B.shape[-1]
        self.Bt = 1
        for d in self.bdims:
            self.Bt *= d
        self.ld = pad_len(self.N)
        self.S = self.Bt * self.nc
        self.opA = PanelOperator(A, self.bdims, self.Bt, self.N)
        self.opM = PanelOperator(M, self.bdims, self.Bt, self.N) if (M is not None and E is not None) else None
        self.E = None
        if E is not None:
            self.E = E.to(self.dtype).expand(*self.bdims, self.nc).reshape(self.Bt, self.nc).resolve_conj().contiguous()
        self._tmp = None
        self._tmp2 = None
        self._scratchP = None
        rhs = to_panel(B.to(self.dtype), self.bdims, self.Bt, self.N)

        hermit = A.is_hermitian and (M is None or M.is_hermitian)
        if need_hermit and not hermit:
            posdef = False                                            # solve.py:607-612
        if posdef is None:
            posdef = self._posdef_heuristic()
        self.normal = not posdef
        if self.normal:                                               # A -> A^H A, B -> A^H B (solve.py:637-643)
            rhs2 = self.new()
            self._apply1(rhs, rhs2, trans=True)
            rhs = rhs2
        self.rhs = rhs

    def new(self, n=None):
        return torch.zeros((self.Bt, self.nc if n is None else n, self.ld), dtype=self.dtype, device=self.device)

    def _shift_operand(self, X, trans):
        """Z of the shift term  - E * Z:  M X (or M^H X), or X itself when there is no M."""
        if self.opM is None:
            return X
        if self._tmp2 is None or self._tmp2.shape != X.shape:
            self._tmp2 = torch.zeros_like(X)
        self.opM.apply(X, self._tmp2, trans=trans)
        return self._tmp2

    def nblk(self):
        vn = self.vec_elems
        return max(1, min(fn("xk_kry_max_partials")(), (self.N + 256 * vn * 4 - 1) // (256 * vn * 4)))

    def _shift_now(self, out, Z):
        """out -= E * Z in place through xk_kry_dots' shift stage (its dot product goes to a scratch partial)."""
        if self._scratchP is None:
            self._scratchP = torch.zeros((self.S, 64, 2 if self.cplx else 1), dtype=self.rdtype, device=self.device)
        extra = (0,) if self.cplx else ()
        check(fn("xk_kry_dots_" + suffix(self.dtype))(ptr(out), ptr(out), ptr(None), ptr(None), ptr(Z), ptr(self.E),
                                                      ptr(self._scratchP), ptr(None), self.S, self.N, self.ld,
                                                      self.nblk(), *extra, stream_ptr()), "xk_kry_dots")

    def _apply1(self, X, out, trans=False, defer=False):
        """out = A X - E * (M X)  (solve.py:590-604).  defer=True leaves the shift to the caller's next
        xk_kry_dots launch, which applies it while it streams `out` anyway: returns the operand Z (or None)."""
        self.opA.apply(X, out, trans=trans)
        if self.E is None:
            return None
        Z = self._shift_operand(X, trans)
        if defer:
            return Z
        self._shift_now(out, Z)
        return None

    def apply(self, X, out, defer=False):
        """out = (A - E M) X, or the normal-equation operator.  With defer=True the return value, when not None,
        is the pending shift operand: pass it as `shift=` to the `_Kry.dots` call that follows."""
        if not self.normal:
            return self._apply1(X, out, defer=defer)
        if self._tmp is None or self._tmp.shape != X.shape:
            self._tmp = torch.zeros_like(X)
        self._apply1(X, self._tmp)
        self._apply1(self._tmp, out, trans=True)
        return None

    @property
    def napply(self):
        return self.opA.napply

    def _posdef_heuristic(self):
        # reference solve.py:617-634 + _get_largest_eival (:645-663): power iterations from an
        # (unseeded) random start; `largest` is a NORM, so this is "posdef unless the operator is 0" (Q8)
        x0 = torch.randn((self.Bt, self.nc, self.ld), dtype=self.dtype, device=self.device)
        x0[:, :, self.N:] = 0
        x0 = x0 / x0.norm(dim=-1, keepdim=True)

        def largest(fcn, x):
            prev = None
            for i in range(10):
                y = self.new()
                x = fcn(x, y)
                xn = x.norm(dim=-1, keepdim=True)
                if i > 0 and bool(torch.all(torch.abs(prev - xn) <= 1e-3 * xn + 1e-6)):
                    break
                prev = xn
                if i < 9:
                    x = x / xn
            return xn
        def op(x, y):
            self._apply1(x, y)
            return y
        big = largest(op, x0)
        neg = big <= 0
        if bool(torch.all(neg)):
            return False
        offset = torch.clamp(big, min=0.0)
        mostneg = largest(lambda x, y: op(x, y).sub_(offset * x), x0)
        return bool(torch.all(torch.logical_or(-mostneg <= offset, neg)).item())

    def solution(self, Xp):
        return from_panel(Xp, self.bdims, self.N)


class _Kry:
    """ctypes plumbing of the fused Krylov kernels on (S, ld) arrays (real or interleaved complex)."""

    def __init__(self, prob):
        self.S, self.N, self.ld = prob.S, prob.N, prob.ld
        self.sfx = suffix(prob.dtype)                      # f64 / f32 / c128 / c64: the vector kernels
        self.rsfx = suffix(prob.rdtype)                    # the (real) status kernel
        self.cplx = prob.cplx
        self.nblk = prob.nblk()
        self.dtype, self.rdtype, self.device = prob.dtype, prob.rdtype, prob.device
        self.E = prob.E
        self.status = torch.zeros((2,), dtype=torch.float64, device=prob.device)
        self.rnorm = torch.zeros((prob.S,), dtype=prob.rdtype, device=prob.device)

    def partial(self):
        """block partials of an inner product: (S, 64) real, or (S, 64, 2) for complex products"""
        shape = (self.S, 64, 2) if self.cplx else (self.S, 64)
        return torch.zeros(shape, dtype=self.rdtype, device=self.device)

    def partial_real(self):
        """block partials of |r|^2 (always real, consumed by xk_kry_status)"""
        return torch.zeros((self.S, 64), dtype=self.rdtype, device=self.device)

    def scalar(self, val=0.0):
        if not self.cplx:
            return torch.full((self.S,), val, dtype=self.dtype, device=self.device)
        t = torch.zeros((self.S, 2), dtype=self.rdtype, device=self.device)
        t[:, 0] = val
        return t

    def _c(self, name, *args):
        check(fn("xk_%s_%s" % (name, self.sfx))(*args, stream_ptr()), "xk_" + name)

    def dots(self, x1, y1, P1, x2=None, y2=None, P2=None, shift=None, conj1=False):
        """partials of <x1,y1> (and <x2,y2>), <x,y> = sum conj(x) y; shift = Z: first y1 -= E * Z in the same pass
        (the `- M X E` term of the operator, solve.py:590-595, folded into the reduction that follows every apply);
        conj1 (complex only): P1 <- <y1,x1> instead."""
        extra = ((1 if conj1 else 0),) if self.cplx else ()
        self._c("kry_dots", ptr(x1), ptr(y1), ptr(x2), ptr(y2), ptr(shift), ptr(self.E if shift is not None else None),
                ptr(P1), ptr(P2), self.S, self.N, self.ld, self.nblk, *extra)

    def bicg_p(self, r, p, v, Prho, rho_old, alpha, omega, rho_store, eps, first):
        self._c("bicg_p", ptr(r), ptr(p), ptr(v), ptr(Prho), ptr(rho_old), ptr(alpha), ptr(omega),
                ptr(rho_store), self.S, self.N, self.ld, self.nblk, float(eps), 1 if first else 0)

    def bicg_s(self, r, v, s, rho, Pr0v, alpha, eps):
        self._c("bicg_s", ptr(r), ptr(v), ptr(s), ptr(rho), ptr(Pr0v), ptr(alpha), self.S, self.N, self.ld,
                self.nblk, float(eps))

    def bicg_final(self, x, xout, yd, zd, s, t, r, r0, alpha, Pts, Ptt, omega, Prr, Prho, eps, skip_r):
        self._c("bicg_final", ptr(x), ptr(xout), ptr(yd), ptr(zd), ptr(s), ptr(t), ptr(r), ptr(r0), ptr(alpha),
                ptr(Pts), ptr(Ptt), ptr(omega), ptr(Prr), ptr(Prho), self.S, self.N, self.ld, self.nblk,
                float(eps), 1 if skip_r else 0)

    def resid(self, b, y, r, r0, Prr, Prho, init=False):
        """r = b - y with partials |r|^2 -> Prr (real) and <r0, r> -> Prho; init=True: only the partials of the
        vector `b` itself (y is taken as zero: the kernel is run with y = a zero panel once)."""
        if init:
            if getattr(self, "_zero", None) is None or self._zero.shape != b.shape:
                self._zero = torch.zeros_like(b)
            self._scr = torch.empty_like(b)
            self._c("kry_resid", ptr(b), ptr(self._zero), ptr(self._scr), ptr(r0), ptr(Prr), ptr(Prho), self.S,
                    self.N, self.ld, self.nblk)
            self._scr = None
            return
        self._c("kry_resid", ptr(b), ptr(y), ptr(r), ptr(r0), ptr(Prr), ptr(Prho), self.S, self.N, self.ld,
                self.nblk)

    def cg_update(self, x, xout, p, Ap, r, Prz, PpAp, Prr, eps, skip_r):
        self._c("cg_update", ptr(x), ptr(xout), ptr(p), ptr(Ap), ptr(r), ptr(Prz), ptr(PpAp), ptr(Prr),
                self.S, self.N, self.ld, self.nblk, float(eps), 1 if skip_r else 0)

    def cg_p(self, z, p, Prz_new, Prz_old, eps):
        self._c("cg_p", ptr(z), ptr(p), ptr(Prz_new), ptr(Prz_old), self.S, self.N, self.ld, self.nblk,
                float(eps))

    def check_status(self, Prr, stop, process_group=None):
        """-> (max residual norm over all systems, number of unconverged systems): the one host sync."""
        check(fn("xk_kry_status_" + self.rsfx)(ptr(Prr), ptr(stop), ptr(self.rnorm), ptr(self.status), self.S,
                                               self.nblk, stream_ptr()), "xk_kry_status")
        # MAX over the ranks of both entries: max residual, and "someone is unconverged" (count > 0)
        allreduce_max_(self.status, process_group)
        mx, nbad = self.status.tolist()
        return mx, nbad


def _zeros_like_solution(A, B, bdims):
    return torch.zeros((*bdims, A.shape[-1], B.shape[-1]), dtype=A.dtype, device=A.device)


def _stop_vector(prob, rtol, atol):
    bnorm = prob.rhs.norm(dim=-1).reshape(-1)
    return torch.max(rtol * bnorm, atol * torch.ones_like(bnorm)).contiguous()


def _precond(P, prob):
    if P is None:
        return None
    from xitorch_amd.linop import LinearOperator
    if not isinstance(P, LinearOperator):
        raise TypeError("precond can only be LinearOperator or None")
    return PanelOperator(P, prob.bdims, prob.Bt, prob.N)


class _XRing:
    """Three rotating solution buffers: current, next, and whichever holds the best iterate —
    best-iterate tracking (solve.py:157-160, 300-303) without copying."""

    def __init__(self, prob):
        self.bufs = [prob.new() for _ in range(3)]
        self.cur, self.best = 0, 0

    def next_index(self):
        for i in range(3):
            if i != self.cur and i != self.best:
                return i
        raise AssertionError


# ------------------------------------------------------------------------------- BiCGStab
def bicgstab(A, B, E=None, M=None, posdef=None, precond_l=None, precond_r=None, max_niter=None,
             rtol=1e-6, atol=1e-8, eps=1e-12, verbose=False, resid_calc_every=10, process_group=None,
             trace=None, **unused):
    r"""
    Solve the linear equations using the stabilized Biconjugate-Gradient method on HIP kernels.

    Keyword arguments
    -----------------
    posdef: bool or None
        Whether :math:`\mathbf{AX-MXE}` is positive definite for all columns and batches; ``None``
        runs the reference's power-iteration heuristic; ``False`` solves the normal equations
    precond_l, precond_r: LinearOperator or None
        Left / right preconditioners
    max_niter: int or None
        Maximum number of iterations (default ``int(1.5 * A.shape[-1])``)
    rtol, atol: float
        Relative / absolute tolerance of the stopping condition w.r.t. the norm of B
    eps: float
        Replacement of exact zeros in denominators
    resid_calc_every: int
        Recompute the true residual with this period (0: never)
    verbose: bool
        Print the progress
    process_group: torch.distributed group or None
        (extension) batch-sharded multi-GPU run: the stopping test is all-reduced over the group
    """
    if _in_host_memory(A):
        from xitorch_amd.linalg import host_krylov
        return host_krylov.bicgstab(A, B, E, M, posdef=posdef, precond_l=precond_l, precond_r=precond_r,
                                    max_niter=max_niter, rtol=rtol, atol=atol, eps=eps, verbose=verbose,
                                    resid_calc_every=resid_calc_every, process_group=process_group, trace=trace)
    nr, ncols = B.shape[-2:]
    if max_niter is None:
        max_niter = int(1.5 * nr)
    bdims = get_batchdims(A, B, E, M)
    # (sharded runs: every rank takes this shortcut or none does — the loop below contains collectives)
    if all_ranks_agree_true(torch.allclose(B, B * 0, rtol=rtol, atol=atol), B.device, process_group):
        return _zeros_like_solution(A, B, bdims)
    prob = _Problem(A, B, E, M, bdims, posdef, need_hermit=False)
    if trace is not None and trace.get("k1_events") is not None:
        prob.opA.events = trace["k1_events"]          # measurement: HIP events around every operator apply
    kr = _Kry(prob)
    pl, pr = _precond(precond_l, prob), _precond(precond_r, prob)
    stop = _stop_vector(prob, rtol, atol)

    r = prob.rhs.clone()                      # x0 = 0  ->  r = B - A 0 = B (solve.py:262)
    r0 = r.clone()
    p, v, s, t = prob.new(), prob.new(), prob.new(), prob.new()
    y = prob.new() if pr is not None else p
    z = prob.new() if pr is not None else s
    Kt = prob.new() if pl is not None else t
    Ks = prob.new() if pl is not None else s
    tmp = prob.new()
    ring = _XRing(prob)
    Prho, Pr0v, Pts, Ptt = (kr.partial() for _ in range(4))
    Prr = kr.partial_real()
    rho = [kr.scalar(), kr.scalar()]
    alpha, omega = kr.scalar(1.0), kr.scalar(1.0)

    kr.resid(r, None, None, r0, Prr, Prho, init=True)              # |r|^2 and <r0, r> of the initial residual
    best, _ = kr.check_status(Prr, stop, process_group)
    converged = False
    niter = 0
    for k in range(1, max_niter + 1):
        niter = k
        kr.bicg_p(r, p, v, Prho, rho[(k + 1) % 2], alpha, omega, rho[k % 2], eps, first=(k == 1))
        if pr is not None:
            pr.apply(p, y)
        sh = prob.apply(y, v, defer=True)
        kr.dots(r0, v, Pr0v, shift=sh)
        kr.bicg_s(r, v, s, rho[k % 2], Pr0v, alpha, eps)
        if pr is not None:
            pr.apply(s, z)
        sh = prob.apply(z, t, defer=(pl is None))
        if pl is not None:
            pl.apply(t, Kt)
            pl.apply(s, Ks)
        kr.dots(Ks, Kt, Pts, Kt, Kt, Ptt, shift=sh, conj1=True)    # <t, s> (solve.py:286) and <t, t>
        refresh = resid_calc_every != 0 and k % resid_calc_every == 0
        nxt = ring.next_index()
        kr.bicg_final(ring.bufs[ring.cur], ring.bufs[nxt], y, z, s, t, r, r0, alpha, Pts, Ptt, omega, Prr, Prho,
                      eps, skip_r=refresh)
        ring.cur = nxt
        if refresh:                                                # solve.py:290-291
            prob.apply(ring.bufs[nxt], tmp)
            kr.resid(prob.rhs, tmp, r, r0, Prr, Prho)
        mx, nbad = kr.check_status(Prr, stop, process_group)
        if mx < best:
            best, ring.best = mx, nxt
        if verbose and (k < 10 or k % 10 == 0):
            print("%4d: |dy|=%.3e" % (k, mx))
        if nbad == 0:
            converged = True
            break
    if trace is not None:
        trace.update(niter=niter, napply=prob.napply, converged=converged, best_resid=best)
    if not converged:
        warnings.warn(ConvergenceWarning("Convergence is not achieved after %d iterations. "
                                         "Max norm of resid: %.3e" % (max_niter, best)))
    return prob.solution(ring.bufs[ring.best])


# ------------------------------------------------------------------------------- CG
def cg(A, B, E=None, M=None, posdef=None, precond=None, max_niter=None, rtol=1e-6, atol=1e-8, eps=1e-12,
       resid_calc_every=10, verbose=False, process_group=None, trace=None, **unused):
    r"""
    Solve the linear equations using the Conjugate-Gradient (CG) method on HIP kernels.

    Keyword arguments
    -----------------
    posdef: bool or None
        Whether :math:`\mathbf{AX-MXE}` is positive definite for all columns and batches; ``None``
        runs the reference's power-iteration heuristic; ``False`` (forced for non-Hermitian
        operators) solves the normal equations
    precond: LinearOperator or None
        Preconditioner
    max_niter: int or None
        Maximum number of iterations (default ``int(1.5 * A.shape[-1])``)
    rtol, atol: float
        Relative / absolute tolerance of the stopping condition w.r.t. the norm of B
    eps: float
        Replacement of exact zeros in denominators
    resid_calc_every: int
        Recompute the true residual with this period (0: never)
    verbose: bool
        Print the progress
    process_group: torch.distributed group or None
        (extension) batch-sharded multi-GPU run: the stopping test is all-reduced over the group
    """
    if _in_host_memory(A):
        from xitorch_amd.linalg import host_krylov
        return host_krylov.cg(A, B, E, M, posdef=posdef, precond=precond, max_niter=max_niter, rtol=rtol, atol=atol,
                              eps=eps, resid_calc_every=resid_calc_every, verbose=verbose,
                              process_group=process_group, trace=trace)
    nr = A.shape[-1]
    ncols = B.shape[-1]
    if max_niter is None:
        max_niter = int(1.5 * nr)
    bdims = get_batchdims(A, B, E, M)
    # (sharded runs: every rank takes this shortcut or none does — the loop below contains collectives)
    if all_ranks_agree_true(torch.allclose(B, B * 0, rtol=rtol, atol=atol), B.device, process_group):
        return _zeros_like_solution(A, B, bdims)
    prob = _Problem(A, B, E, M, bdims, posdef, need_hermit=True)
    if trace is not None and trace.get("k1_events") is not None:
        prob.opA.events = trace["k1_events"]          # measurement: HIP events around every operator apply
    kr = _Kry(prob)
    pre = _precond(precond, prob)
    stop = _stop_vector(prob, rtol, atol)

    r = prob.rhs.clone()
    z = prob.new() if pre is not None else r
    if pre is not None:
        pre.apply(r, z)
    p = z.clone()
    Ap, tmp = prob.new(), prob.new()
    ring = _XRing(prob)
    Prz = [kr.partial(), kr.partial()]
    PpAp, Prr = kr.partial(), kr.partial_real()
    kr.resid(r, None, None, None, Prr, None, init=True)            # |r|^2 of the initial residual
    kr.dots(r, z, Prz[0])
    best, _ = kr.check_status(Prr, stop, process_group)
    converged = False
    cur = 0
    niter = 0
    for k in range(1, max_niter + 1):
        niter = k
        sh = prob.apply(p, Ap, defer=True)
        kr.dots(p, Ap, PpAp, shift=sh)
        refresh = resid_calc_every != 0 and k % resid_calc_every == 0
        nxt = ring.next_index()
        kr.cg_update(ring.bufs[ring.cur], ring.bufs[nxt], p, Ap, r, Prz[cur], PpAp, Prr, eps, skip_r=refresh)
        ring.cur = nxt
        if refresh:                                                # solve.py:148-149
            prob.apply(ring.bufs[nxt], tmp)
            kr.resid(prob.rhs, tmp, r, None, Prr, None)
        mx, nbad = kr.check_status(Prr, stop, process_group)
        if mx < best:
            best, ring.best = mx, nxt
        if verbose and (k < 10 or k % 10 == 0):
            print("%4d: |dy|=%.3e" % (k, mx))
        if nbad == 0:
            converged = True
            break
        if pre is not None:
            pre.apply(r, z)
        kr.dots(r, z, Prz[1 - cur])
        kr.cg_p(z, p, Prz[1 - cur], Prz[cur], eps)
        cur = 1 - cur
    if trace is not None:
        trace.update(niter=niter, napply=prob.napply, converged=converged, best_resid=best)
    if not converged:
        warnings.warn(ConvergenceWarning("Convergence is not achieved after %d iterations. "
                                         "Max norm of best resid: %.3e" % (max_niter, best)))
    return prob.solution(ring.bufs[ring.best])


# ------------------------------------------------------------------------------- GMRES
class _GmresState:
    """Per-system Hessenberg / Givens state on the device (float64 whatever the vector dtype), growing with the basis
    (the reference preallocates max_niter vectors and a (max_niter+1) x max_niter Hessenberg, solve.py:384-386, Q12)."""

    def __init__(self, S, cap, device):
        self.S, self.cap, self.device = S, cap, device
        z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=device)
        self.R, self.cs, self.sn, self.g = z(S, cap + 1, cap), z(S, cap), z(S, cap), z(S, cap + 1)

    def grow(self, need, limit):
        if need <= self.cap:
            return
        new = min(limit, max(need, 2 * self.cap))
        z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=self.device)
        R, cs, sn, g = z(self.S, new + 1, new), z(self.S, new), z(self.S, new), z(self.S, new + 1)
        R[:, :self.cap + 1, :self.cap] = self.R
        cs[:, :self.cap], sn[:, :self.cap], g[:, :self.cap + 1] = self.cs, self.sn, self.g
        self.R, self.cs, self.sn, self.g, self.cap = R, cs, sn, g, new


def gmres(A, B, E=None, M=None, posdef=None, max_niter=None, rtol=1e-6, atol=1e-8, eps=1e-12,
          resid_calc_every=1, restart=None, process_group=None, trace=None, **unused):
    r"""
    Solve the linear equations using the Generalised minimal residual method on HIP kernels
    (reference: gmres, xitorch/_impls/linalg/solve.py:326-433; real operators only, like the reference's).

    Same iterates, same stopping rule and same return value as the reference: un-restarted GMRES from ``x0 = 0``;
    after ``k`` Arnoldi steps the iterate ``x_k`` minimises the residual over the ``k``-dimensional Krylov space; the
    TRUE residual ``B - (A x_k - M x_k E)`` decides convergence (``all(|r_col| < max(rtol |b_col|, atol))``) and
    which iterate is the best one (smallest maximum residual norm over all batches and columns, solve.py:414-425);
    the best iterate is returned, with a ``ConvergenceWarning`` when the tolerance was not reached.  Like the
    reference's loop (``for k in range(min(nr, max_niter))`` solving with the first ``k`` columns, :389,403-410) at most
    ``min(nr, max_niter) - 1`` Krylov vectors contribute.

    How it runs here: the Krylov basis of every system is kept panel-major on the device and grows by one vector per
    iteration; the new direction is orthogonalised by classical Gram-Schmidt applied twice (two Gram products on
    the K1 kernel, ``xk_lincomb`` / ``xk_gmres_finish`` — the reference's modified Gram-Schmidt, :391-393, is a
    sequential chain of k dot / axpy pairs); the Hessenberg column, its Givens rotations and the rotated right-hand
    side live on the device (``xk_gmres_step``), so the least-squares problem the reference hands to
    ``torch.linalg.lstsq`` every iteration (:403) is one back substitution (``xk_gmres_solve``); its residual
    ``|g[k+1]|`` equals the true residual norm in exact arithmetic.  One host read per iteration.

    Keyword arguments
    -----------------
    posdef: bool or None
        As for :func:`bicgstab`
    max_niter: int or None
        Maximum number of iterations (default ``A.shape[-1]``)
    rtol, atol: float
        Relative / absolute tolerance of the stopping condition w.r.t. the norm of B
    eps: float
        Replacement of exact zeros in denominators
    resid_calc_every: int
        (extension) ``1`` (default): the iterate and its true residual are formed every iteration, exactly like the
        reference.  ``n > 1``: only every n-th iteration, on the last one, and whenever the least-squares estimate
        says that every system has converged (one more operator apply and one pass over the basis saved per skipped
        iteration; stopping decisions are still taken on true residuals only, best-iterate tracking sees the
        checked iterates).
    restart: int or None
        (extension) ``None`` (default): un-restarted, like the reference, whose Krylov basis grows until ``max_niter``
        (solve.py:384-389: it preallocates ``max_niter`` vectors).  An integer ``m``: GMRES(m) — after ``m`` Arnoldi steps
        the iterate and its true residual are formed, the basis is dropped and the next cycle starts from that residual
        (memory ``m + 1`` vectors per system instead of ``max_niter``; lets systems that need more steps than fit
        reach a tight ``rtol``).  Stopping rule and best-iterate rule are unchanged and run across the cycles; ``max_niter``
        bounds the total number of steps.
    process_group: torch.distributed group or None
        (extension) batch-sharded multi-GPU run: the stopping test is all-reduced over the group

    With ``E`` the reference returns its column-swapped work layout ``(ncols, *batch, nr, 1)`` without undoing the
    swap (:432, and fails for more than one column); this function returns ``(*batch, nr, ncols)`` like every other
    method.
    """
    if _in_host_memory(A):
        from xitorch_amd.linalg import host_krylov
        return host_krylov.gmres(A, B, E, M, posdef=posdef, max_niter=max_niter, rtol=rtol, atol=atol, eps=eps,
                                 resid_calc_every=resid_calc_every, restart=restart, process_group=process_group,
                                 trace=trace)
    nr, ncols = A.shape[-1], B.shape[-1]
    if A.dtype.is_complex:
        # the reference's own gmres is real-only as well (its tests xfail complex input, test_linop_fcns.py:479-481)
        raise NativeLibraryError("xitorch_amd gmres supports real operators only, like the reference's gmres")
    if max_niter is None:
        max_niter = int(nr)
    bdims = get_batchdims(A, B, E, M)
    # (sharded runs: every rank takes this shortcut or none does — the loop below contains collectives)
    if all_ranks_agree_true(torch.allclose(B, B * 0, rtol=rtol, atol=atol), B.device, process_group):
        return _zeros_like_solution(A, B, bdims)
    prob = _Problem(A, B, E, M, bdims, posdef, need_hermit=False)
    if trace is not None and trace.get("k1_events") is not None:
        prob.opA.events = trace["k1_events"]          # measurement: HIP events around every operator apply
    kr = _Kry(prob)
    S, N, ld = prob.S, prob.N, prob.ld
    dtype, dev = prob.dtype, prob.device
    sfx = suffix(dtype)
    stop = _stop_vector(prob, rtol, atol)
    every = max(1, int(resid_calc_every))
    msteps = min(nr, max_niter) - 1           # Arnoldi steps whose column enters an iterate (solve.py:389,403)
    if restart is not None:
        restart = int(restart)
        if restart < 1:
            raise ValueError("gmres: restart must be a positive number of Arnoldi steps, got %d" % restart)
        msteps = max_niter - 1 if max_niter > 1 else 0      # (restarted: the total is not bounded by the order)
    mcyc = msteps if restart is None else min(restart, max(msteps, 1))     # Arnoldi steps per cycle
    lazy_limit = None
    if mcyc > 8192:
        # xk_gmres_solve keeps the least-squares solution of a system in LDS: at most 8192 basis vectors per cycle (the
        # dense Hessenberg of such a run is 8 S k^2 bytes long before that).  An explicit restart length beyond it is
        # refused here; the un-restarted default (max_niter = None -> the operator's order, solve.py:389) only fails if
        # a run really gets that far — almost every run converges in a few dozen steps whatever the order
        if restart is not None:
            raise NativeLibraryError("xitorch_amd gmres: restart=%d exceeds the supported cycle length (8192 basis "
                                     "vectors per system)" % restart)
        lazy_limit, mcyc = 8192, 8192
    rhs = prob.rhs.reshape(S, ld)
    beta = rhs.norm(dim=-1)                                                      # (S,)
    best = float(allreduce_max_(beta.max().double().reshape(1), process_group).item())      # solve.py:380-381
    xbufs = [torch.zeros((S, 1, ld), dtype=dtype, device=dev) for _ in range(2)]
    best_i, cur_i = 0, 1                      # xbufs[0] = x0 = 0 is the best iterate so far (:382)
    converged = False
    nsteps, nsync, ncycles = 0, 1, 0
    if msteps > 0:
        cap = min(mcyc + 1, 32)
        x_base = None                         # restarted cycles: the iterate the current cycle corrects
        Q = torch.zeros((S, cap, ld), dtype=dtype, device=dev)
        Q[:, 0] = rhs / torch.where(beta == 0, torch.full_like(beta, eps), beta).unsqueeze(-1)   # :385, _safedenom
        st = _GmresState(S, cap - 1, dev)
        st.g[:, 0] = beta.double()
        inv_hn = torch.zeros((S,), dtype=dtype, device=dev)
        Pest, Ptrue = kr.partial_real(), kr.partial_real()
        ycoef = torch.zeros((S, 1, cap), dtype=dtype, device=dev)
        status4 = torch.zeros((4,), dtype=torch.float64, device=dev)
        tmp = prob.new()
        rtrue = prob.new()

        def status_of(Prr, nblk, slot):
            check(fn("xk_kry_status_" + kr.rsfx)(ptr(Prr), ptr(stop), ptr(kr.rnorm), ptr(status4[2 * slot:]), S, nblk,
                                                 stream_ptr()), "xk_kry_status")

        def true_residual(kd):
            """x = Q y with R y = g (the reference's lstsq solution, :403-410), r = B - A x (:414): partials -> Ptrue"""
            check(fn("xk_gmres_solve_" + sfx)(ptr(st.R), ptr(st.g), ptr(ycoef), ycoef.stride(0), S, kd, st.cap,
                                              stream_ptr()), "xk_gmres_solve")
            x = xbufs[cur_i]
            K.lincomb(Q, ycoef, x, kd, 1, coef_layout="ca", alpha=1.0, beta=0.0)
            if x_base is not None:
                x.add_(x_base)
            prob.apply(x.reshape(prob.Bt, prob.nc, ld), tmp)
            kr.resid(prob.rhs, tmp, rtrue, None, Ptrue, None)
            status_of(Ptrue, kr.nblk, 1)

        cyc0 = 0                              # global index of the current cycle's first Arnoldi step
        for k in range(msteps):
            j = k - cyc0                                              # step index inside the cycle
            if lazy_limit is not None and j >= lazy_limit:
                raise NativeLibraryError("xitorch_amd gmres: the un-restarted Krylov basis reached %d vectors per system "
                                         "without converging (best residual %.3e); pass restart=m (GMRES(m)) or a "
                                         "max_niter <= %d" % (lazy_limit, best, lazy_limit + 1))
            nsteps = k + 1
            if j + 2 > cap:                                           # grow the basis storage
                newcap = min(mcyc + 1, 2 * cap)
                Qn = torch.zeros((S, newcap, ld), dtype=dtype, device=dev)
                Qn[:, :cap].copy_(Q)
                Q, cap = Qn, newcap
                ycoef = torch.zeros((S, 1, cap), dtype=dtype, device=dev)
                st.grow(cap - 1, mcyc)
            # w = A q_j (solve.py:390) straight into basis row j+1; with a shift E the fused shift kernel wants
            # contiguous (S, ld) arrays, so q_j / w pass through two contiguous buffers (O(N) copies)
            if prob.E is None:
                prob.apply(Q[:, j].reshape(prob.Bt, prob.nc, ld), Q[:, j + 1].reshape(prob.Bt, prob.nc, ld))
            else:
                rtrue.reshape(S, ld).copy_(Q[:, j])
                prob.apply(rtrue, tmp)
                Q[:, j + 1].copy_(tmp.reshape(S, ld))
            wrow = Q[:, j + 1:j + 2]
            c1 = K.dense_mm(Q[:, :j + 1, :N], wrow[:, :, :N])                      # (S, 1, j+1): <q_i, w>
            K.lincomb(Q, c1, wrow, j + 1, 1, coef_layout="ca", alpha=-1.0, beta=1.0)
            c2n = K.dense_mm(Q[:, :j + 2, :N], wrow[:, :, :N])                     # second pass; last entry |w1|^2
            check(fn("xk_gmres_step_" + sfx)(ptr(c1), c1.stride(0), ptr(c2n), c2n.stride(0), j, st.cap, ptr(st.R),
                                             ptr(st.cs), ptr(st.sn), ptr(st.g), ptr(inv_hn), ptr(Pest), S,
                                             stream_ptr()), "xk_gmres_step")
            check(fn("xk_gmres_finish_" + sfx)(ptr(Q), ptr(c2n), c2n.stride(0), ptr(inv_hn), S, N, j, Q.stride(1),
                                               Q.stride(0), stream_ptr()), "xk_gmres_finish")
            status_of(Pest, 1, 0)
            cycle_end = restart is not None and j + 1 == mcyc
            checked = (k + 1) % every == 0 or k == msteps - 1 or cycle_end
            if checked:
                true_residual(j + 1)
            allreduce_max_(status4, process_group)
            est_mx, est_bad, tr_mx, tr_bad = status4.tolist()                      # the iteration's host read
            nsync += 1
            if not checked and est_bad == 0:
                # the least-squares residual says every system is done: decide on the true residual, like the
                # reference does every iteration; from here on every iterate is checked
                true_residual(j + 1)
                allreduce_max_(status4, process_group)
                est_mx, est_bad, tr_mx, tr_bad = status4.tolist()
                nsync += 1
                checked, every = True, 1
            just = cur_i                                                            # buffer of the iterate just formed
            if checked:
                if tr_mx < best:                                                    # solve.py:417-421
                    best = tr_mx
                    best_i, cur_i = cur_i, best_i
                if tr_bad == 0:                                                     # :423-425
                    converged = True
                    break
            if cycle_end and k < msteps - 1:
                # GMRES(m): the next cycle corrects the iterate just formed, starting from its TRUE residual
                if x_base is None:
                    x_base = torch.zeros((S, 1, ld), dtype=dtype, device=dev)
                x_base.copy_(xbufs[just])
                rcur = rtrue.reshape(S, ld)
                beta = rcur.norm(dim=-1)
                Q[:, 0] = rcur / torch.where(beta == 0, torch.full_like(beta, eps), beta).unsqueeze(-1)
                for t in (st.R, st.cs, st.sn, st.g):
                    t.zero_()
                st.g[:, 0] = beta.double()
                cyc0 = k + 1
                ncycles += 1
    if trace is not None:
        # niter counts like the reference's loop: the pass that tests x_k is pass k + 1
        trace.update(niter=nsteps + 1, napply=prob.napply, converged=converged, best_resid=best, arnoldi_steps=nsteps,
                     host_syncs=nsync, restarts=ncycles)
    if not converged:
        warnings.warn(ConvergenceWarning("Convergence is not achieved after %d iterations. "
                                         "Max norm of resid: %.3e" % (max_niter, best)))
    return prob.solution(xbufs[best_i].reshape(prob.Bt, prob.nc, ld))


def scipy_gmres(A, B, E=None, M=None, min_eps=1e-9, max_niter=None, **unused):
    """The reference's `method="scipy_gmres"` (wrap_gmres, solve.py:14-66): SciPy's GMRES on the host, one system at a
    time, with the operator applied wherever it lives (the reference copies every iterate between the host and the
    operator's device as well)."""
    from xitorch_amd.linalg import host_krylov
    return host_krylov.scipy_gmres(A, B, E, M, min_eps=min_eps, max_niter=max_niter)


# ------------------------------------------------------------------------------- root-finder based
def broyden1_solve(A, B, E=None, M=None, **options):
    """Solve ``A X - M X E = B`` as a root-finding problem with Broyden's first method
    (reference: broyden1_solve/_rootfinder_solve, solve.py:447-478)."""
    from xitorch_amd.optimize.native_root import broyden1
    nr = A.shape[-1]
    ncols = B.shape[-1]

    def residual(xi):
        x = xi.reshape(*xi.shape[:-1], nr, ncols)
        y = A.mm(x) - B
        if E is not None:
            MX = M.mm(x) if M is not None else x
            y = y - MX * E.unsqueeze(-2)
        return y.reshape(*xi.shape[:-1], -1)

    bdims = get_batchdims(A, B, E, M)
    x0 = torch.zeros((*bdims, nr * ncols), dtype=A.dtype, device=A.device)
    x = broyden1(residual, x0, **options)
    return x.reshape(*x.shape[:-1], nr, ncols)
