"""oracle.solve — CPU restatement of the reference Krylov solvers (CG, BiCGStab, GMRES).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows
xitorch/_impls/linalg/solve.py:69-433 and the helpers at :437-445, :540-663.
Solves  A X = B  or  A X - M X E = B  for X of shape (*batch, nr, ncols).
"""
import warnings
import torch


class OracleConvergenceWarning(Warning):
    pass


def _safedenom(r, eps):
    # reference solve.py:437-439 — patches exact zeros IN PLACE (quirk Q9)
    r[r == 0] = eps
    return r


def _dot(r, z):
    # reference solve.py:441-445 — column-wise <r, z>, shape (*, 1, nc)
    return torch.einsum("...rc,...rc->...c", r.conj(), z).unsqueeze(-2)


def _pad_shapes(*shapes):
    n = max(len(s) for s in shapes)
    return [[1] * (n - len(s)) + list(s) for s in shapes]


def batchdims(A, B, E, M):
    # reference solve.py:540-549
    shapes = [tuple(A.shape[:-2]), tuple(B.shape[:-2])]
    if E is not None:
        shapes.append(tuple(E.shape[:-1]))
        if M is not None:
            shapes.append(tuple(M.shape[:-2]))
    return list(torch.broadcast_shapes(*shapes))


def _largest_eival(Afcn, x):
    # reference solve.py:645-663: <=10 power iterations, returns the last NORM (>= 0, quirk Q8)
    niter, rtol, atol = 10, 1e-3, 1e-6
    prev = None
    for i in range(niter):
        x = Afcn(x)
        xnorm = x.norm(dim=-2, keepdim=True)
        if i > 0:
            if torch.all(torch.abs(prev - xnorm) <= rtol * xnorm + atol):
                break
        prev = xnorm
        if i < niter - 1:
            x = x / xnorm
    return xnorm


def setup_linear_problem(A, B, E, M, bdims, posdef, need_hermit):
    """reference solve.py:560-643.  Returns (A_fcn, AT_fcn, B2, col_swapped)."""
    if E is None:
        A_fcn = lambda x: A.mm(x)
        AT_fcn = lambda x: A.rmm(x)
        B_new = B
        col_swapped = False
    else:
        # move the columns to a new leading axis: every column has its own shift (quirk Q11)
        if M is None:
            BAs, BBs, BEs = _pad_shapes(A.shape[:-2], B.shape[:-2], E.shape[:-1])
        else:
            BAs, BBs, BEs, BMs = _pad_shapes(A.shape[:-2], B.shape[:-2], E.shape[:-1], M.shape[:-2])
        E = E.reshape(*BEs, *E.shape[-1:])
        E_new = E.unsqueeze(0).transpose(-1, 0).unsqueeze(-1)      # (ncols, *BE, 1, 1)
        B = B.reshape(*BBs, *B.shape[-2:])
        B_new = B.unsqueeze(0).transpose(-1, 0)                    # (ncols, *BB, nr, 1)

        def A_fcn(x):
            Ax = A.mm(x)
            Mx = M.mm(x) if M is not None else x
            return Ax - Mx * E_new

        def AT_fcn(x):
            ATx = A.rmm(x)
            MTx = M.rmm(x) if M is not None else x
            return ATx - MTx * E_new

        col_swapped = True

    if need_hermit:
        is_hermit = A.is_hermitian and (M is None or M.is_hermitian)
        if not is_hermit:
            posdef = False                                         # :607-612 -> normal equations

    if posdef is None:                                             # :617-634 (unseeded randn, quirk Q8)
        nr, ncols = B.shape[-2:]
        x0shape = (ncols, *bdims, nr, 1) if col_swapped else (*bdims, nr, ncols)
        x0 = torch.randn(x0shape, dtype=A.dtype)
        x0 = x0 / x0.norm(dim=-2, keepdim=True)
        largest = _largest_eival(A_fcn, x0)
        neg = largest <= 0
        if torch.all(neg):
            posdef = False
        else:
            offset = torch.clamp(largest, min=0.0)
            A_fcn2 = lambda x: A_fcn(x) - offset * x
            mostneg = _largest_eival(A_fcn2, x0)
            posdef = bool(torch.all(torch.logical_or(-mostneg <= offset, neg)).item())

    if posdef:
        return A_fcn, AT_fcn, B_new, col_swapped
    A2 = lambda x: AT_fcn(A_fcn(x))                                # :637-643
    return A2, A2, AT_fcn(B_new), col_swapped


def _finish(x, col_swapped):
    if col_swapped:
        x = x.transpose(0, -1).squeeze(0)
    return x


def cg(A, B, E=None, M=None, posdef=None, precond=None, max_niter=None, rtol=1e-6, atol=1e-8,
       eps=1e-12, resid_calc_every=10, verbose=False, trace=None, **unused):
    """Preconditioned CG (reference: cg, solve.py:69-190)."""
    nr = A.shape[-1]
    ncols = B.shape[-1]
    if max_niter is None:
        max_niter = int(1.5 * nr)
    bdims = batchdims(A, B, E, M)
    if torch.allclose(B, B * 0, rtol=rtol, atol=atol):             # :117
        return torch.zeros((*bdims, nr, ncols), dtype=A.dtype)
    pre = (lambda x: precond.mm(x)) if precond is not None else (lambda x: x)
    A_fcn, _, B2, swapped = setup_linear_problem(A, B, E, M, bdims, posdef, True)
    B_norm = B2.norm(dim=-2, keepdim=True)
    stop = torch.max(rtol * B_norm, atol * torch.ones_like(B_norm))
    x0shape = (ncols, *bdims, nr, 1) if swapped else (*bdims, nr, ncols)
    xk = torch.zeros(x0shape, dtype=A.dtype)
    rk = B2 - A_fcn(xk)
    zk = pre(rk)
    pk = zk
    rkzk = _dot(rk, zk)
    converged = False
    best = rk.norm(dim=-2).max().item()
    best_x = xk
    niter = 0
    for k in range(1, max_niter + 1):
        niter = k
        Apk = A_fcn(pk)
        alpha = rkzk / _safedenom(_dot(pk, Apk), eps)
        xk1 = xk + alpha * pk
        if resid_calc_every != 0 and k % resid_calc_every == 0:    # :148-151 true-residual refresh
            rk1 = B2 - A_fcn(xk1)
        else:
            rk1 = rk - alpha * Apk
        rnorm = rk1.norm(dim=-2, keepdim=True)
        mx = rnorm.max().item()
        if mx < best:                                              # :157-160 best iterate (quirk Q10)
            best, best_x = mx, xk1
        if torch.all(rnorm < stop):
            converged = True
            break
        zk1 = pre(rk1)
        rkzk1 = _dot(rk1, zk1)
        beta = rkzk1 / _safedenom(rkzk, eps)
        pk = zk1 + beta * pk
        xk, rk, rkzk = xk1, rk1, rkzk1
    if trace is not None:
        trace.update(niter=niter, converged=converged, best_resid=best)
    if not converged:
        warnings.warn(OracleConvergenceWarning("cg: no convergence after %d iterations (best resid %.3e)"
                                               % (max_niter, best)))
    return _finish(best_x, swapped)


def bicgstab(A, B, E=None, M=None, posdef=None, precond_l=None, precond_r=None, max_niter=None,
             rtol=1e-6, atol=1e-8, eps=1e-12, verbose=False, resid_calc_every=10, trace=None, **unused):
    """BiCGStab (reference: bicgstab, solve.py:192-324)."""
    nr, ncols = B.shape[-2:]
    if max_niter is None:
        max_niter = int(1.5 * nr)
    bdims = batchdims(A, B, E, M)
    if torch.allclose(B, B * 0, rtol=rtol, atol=atol):
        return torch.zeros((*bdims, nr, ncols), dtype=A.dtype)
    pl = (lambda x: precond_l.mm(x)) if precond_l is not None else (lambda x: x)
    pr = (lambda x: precond_r.mm(x)) if precond_r is not None else (lambda x: x)
    A_fcn, _, B2, swapped = setup_linear_problem(A, B, E, M, bdims, posdef, False)
    B_norm = B2.norm(dim=-2, keepdim=True)
    stop = torch.max(rtol * B_norm, atol * torch.ones_like(B_norm))
    x0shape = (ncols, *bdims, nr, 1) if swapped else (*bdims, nr, ncols)
    xk = torch.zeros(x0shape, dtype=A.dtype)
    rk = B2 - A_fcn(xk)
    r0hat = rk
    rho_k = _dot(r0hat, rk)
    omega_k = torch.tensor(1.0, dtype=A.dtype)
    alpha = 1.0
    vk = 0.0
    pk = 0.0
    converged = False
    best = rk.norm(dim=-2).max()
    best_x = xk
    niter = 0
    for k in range(1, max_niter + 1):
        niter = k
        rho_new = _dot(r0hat, rk)                                  # :273
        omega_den = _safedenom(omega_k, eps)                       # :274 (in place)
        beta = rho_new / _safedenom(rho_k, eps) * (alpha / omega_den)
        pk = rk + beta * (pk - omega_k * vk)
        y = pr(pk)
        vk = A_fcn(y)
        alpha = rho_new / _safedenom(_dot(r0hat, vk), eps)
        h = xk + alpha * y
        s = rk - alpha * vk
        z = pr(s)
        t = A_fcn(z)
        Kt = pl(t)
        omega_k = _dot(Kt, pl(s)) / _safedenom(_dot(Kt, Kt), eps)
        xk = h + omega_k * z
        if resid_calc_every != 0 and k % resid_calc_every == 0:    # :290-293
            rk = B2 - A_fcn(xk)
        else:
            rk = s - omega_k * t
        rnorm = rk.norm(dim=-2, keepdim=True)
        mx = rnorm.max().item()
        if mx < best:
            best, best_x = mx, xk
        if torch.all(rnorm < stop):
            converged = True
            break
        rho_k = rho_new
    if trace is not None:
        trace.update(niter=niter, converged=converged, best_resid=float(best))
    if not converged:
        warnings.warn(OracleConvergenceWarning("bicgstab: no convergence after %d iterations (best resid %.3e)"
                                               % (max_niter, float(best))))
    return _finish(best_x, swapped)


def gmres(A, B, E=None, M=None, posdef=None, max_niter=None, rtol=1e-6, atol=1e-8, eps=1e-12,
          trace=None, **unused):
    """Un-restarted GMRES with MGS Arnoldi and a least-squares solve per step
    (reference: gmres, solve.py:326-433; note: returns WITHOUT undoing the column swap, as the
    reference does — `E` is therefore only meaningful through the callers' own handling)."""
    converged = False
    nr, ncols = A.shape[-1], B.shape[-1]
    if max_niter is None:
        max_niter = int(nr)
    bdims = batchdims(A, B, E, M)
    if torch.allclose(B, B * 0, rtol=rtol, atol=atol):
        return torch.zeros((*bdims, nr, ncols), dtype=A.dtype)
    A_fcn, _, B2, swapped = setup_linear_problem(A, B, E, M, bdims, posdef, False)
    B_norm = B2.norm(dim=-2, keepdim=True)
    stop = torch.max(rtol * B_norm, atol * torch.ones_like(B_norm))
    x0shape = (ncols, *bdims, nr, 1) if swapped else (*bdims, nr, ncols)
    x0 = torch.zeros(x0shape, dtype=A.dtype)
    r = B2 - A_fcn(x0)
    best = r.norm(dim=-2, keepdim=True).max().item()
    best_res = x0
    q = torch.empty([max_niter] + list(r.shape), dtype=A.dtype)     # :384 (quirk Q12)
    q[0] = r / _safedenom(r.norm(dim=-2, keepdim=True), eps)
    h = torch.zeros((*bdims, ncols, max_niter + 1, max_niter), dtype=A.dtype)
    h = h.reshape((-1, ncols, max_niter + 1, max_niter))
    niter = 0
    for k in range(min(nr, max_niter)):
        niter = k + 1
        y = A_fcn(q[k])
        for j in range(k + 1):                                      # MGS, :391-393
            h[..., j, k] = _dot(q[j], y).reshape(-1, ncols)
            y = y - h[..., j, k].reshape(*bdims, 1, ncols) * q[j]
        h[..., k + 1, k] = torch.linalg.norm(y, dim=-2)
        if torch.any(h[..., k + 1, k]) != 0 and k != max_niter - 1:
            q[k + 1] = y.reshape(-1, nr, ncols) / h[..., k + 1, k].reshape(-1, 1, ncols)
            q[k + 1] = q[k + 1].reshape(*bdims, nr, ncols)
        b = torch.zeros((*bdims, ncols, k + 1), dtype=A.dtype).reshape(-1, ncols, k + 1)
        b[..., 0] = torch.linalg.norm(r, dim=-2)
        rk = torch.linalg.lstsq(h[..., :k + 1, :k], b)[0]           # :403
        res = None
        for i in range(k):                                          # :407-410
            term = q[i] * rk[..., i].reshape(*bdims, 1, ncols) + x0
            res = term if res is None else res + term
        if res is not None:
            resid = B2 - A_fcn(res)                                 # :414 true residual every step
            rnorm = resid.norm(dim=-2, keepdim=True)
            mx = rnorm.max().item()
            if mx < best:
                best, best_res = mx, res
            if torch.all(rnorm < stop):
                converged = True
                break
    if trace is not None:
        trace.update(niter=niter, converged=converged, best_resid=best)
    if not converged:
        warnings.warn(OracleConvergenceWarning("gmres: no convergence after %d iterations (best resid %.3e)"
                                               % (max_niter, best)))
    return best_res


def exactsolve(A, B, E=None, M=None):
    """Dense answer (reference: exactsolve/_solve_ABE, solve.py:481-537)."""
    Amat = A.fullmatrix()
    if E is None:
        return torch.linalg.solve(Amat, B)
    n = Amat.shape[-1]
    Mmat = M.fullmatrix() if M is not None else torch.eye(n, dtype=Amat.dtype)
    cols = []
    for c in range(B.shape[-1]):
        AE = Amat - Mmat * E[..., c].unsqueeze(-1).unsqueeze(-1)
        cols.append(torch.linalg.solve(AE, B[..., c:c + 1]))
    return torch.cat(cols, dim=-1)
