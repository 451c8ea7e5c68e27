"""symeig / lsymeig / usymeig / svd — functional front-ends of the eigen path.

Same signatures, defaults, argument checks and implicit-differentiation backward as the
reference (xitorch/linalg/symeig.py:17-461); the forward method table points at the native
HIP eigensolver (xitorch_amd/linalg/native_eig.py) and the backward solve goes through
`xitorch_amd.linalg.solve` (native Krylov kernels).
"""
import warnings
import torch
from xitorch_amd.linop import LinearOperator  # noqa: F401  (re-exported for type references in user code)
from xitorch_amd.linalg.solve import solve
from xitorch_amd.linalg.native_eig import davidson, exacteig
from xitorch_amd.debug import is_debug_enabled
from xitorch_amd._util import assert_runtime, merge_options, null_context, get_method, pop_keys, MathWarning

__all__ = ["lsymeig", "usymeig", "symeig", "svd"]


def lsymeig(A, neig=None, M=None, bck_options={}, method=None, **fwd_options):
    return symeig(A, neig, "lowest", M, method=method, bck_options=bck_options, **fwd_options)


def usymeig(A, neig=None, M=None, bck_options={}, method=None, **fwd_options):
    return symeig(A, neig, "uppest", M, method=method, bck_options=bck_options, **fwd_options)


def symeig(A, neig=None, mode="lowest", M=None, bck_options={}, method=None, **fwd_options):
    r"""
    Obtain ``neig`` lowest (or uppermost) eigenvalues and eigenvectors of a Hermitian linear
    operator, :math:`\mathbf{AX = MXE}`.

    Arguments
    ---------
    A: LinearOperator
        Hermitian operator of shape ``(*BA, q, q)``
    neig: int or None
        Number of eigenpairs (``None``: all)
    mode: str
        ``"lowest"`` or ``"uppermost"``/``"uppest"``
    M: LinearOperator or None
        Hermitian right-hand-side operator ``(*BM, q, q)`` (``None``: identity)
    bck_options: dict
        Options of :func:`solve` for the backward pass, plus ``degen_atol`` / ``degen_rtol``
        (thresholds below which two eigenvalues are treated as degenerate; defaults
        ``eps**0.6`` / ``eps**0.4``)
    method: str or callable or None
        ``"exacteig"`` (default, dense), ``"davidson"`` (native HIP block Davidson), or a callable
        ``f(A, neig, mode, M, **fwd_options) -> (evals, evecs)``
    **fwd_options
        Method-specific options

    Returns
    -------
    (eigenvalues ``(*BAM, neig)``, eigenvectors ``(*BAM, q, neig)``), eigenvalues ascending.
    """
    assert_runtime(A.is_hermitian, "The linear operator A must be Hermitian")
    assert_runtime(not torch.is_grad_enabled() or A.is_getparamnames_implemented,
                   "The _getparamnames(self, prefix) of linear operator A must be "
                   "implemented if using symeig with grad enabled")
    if M is not None:
        assert_runtime(M.is_hermitian, "The linear operator M must be Hermitian")
        assert_runtime(M.shape[-1] == A.shape[-1], "The shape of A & M must match (A: %s, M: %s)" % (A.shape, M.shape))
        assert_runtime(not torch.is_grad_enabled() or M.is_getparamnames_implemented,
                       "The _getparamnames(self, prefix) of linear operator M must be "
                       "implemented if using symeig with grad enabled")
    mode = mode.lower()
    if mode == "uppermost":
        mode = "uppest"
    if method is None:
        method = "exacteig"          # the reference's default for every operator kind (quirk Q7)
    if neig is None:
        neig = A.shape[-1]
    if is_debug_enabled():
        A.check()
        if M is not None:
            M.check()
    if method == "exacteig":
        return exacteig(A, neig, mode, M)
    fwd_options["method"] = method
    params = A.getlinopparams()
    mparams = M.getlinopparams() if M is not None else []
    return _SymeigFunction.apply(A, neig, mode, M, fwd_options, bck_options, len(params), *params, *mparams)


def svd(A, k=None, mode="uppest", bck_options={}, method=None, **fwd_options):
    r"""
    Singular value decomposition :math:`\mathbf{A} = \mathbf{U\Sigma V}^H` of an operator
    ``(*BA, m, n)`` through ``symeig`` of ``A^H A`` or ``A A^H`` (whichever is smaller).
    Returns ``(u (*BA,m,k), s (*BA,k), vh (*BA,k,n))``.  (reference: symeig.py:146-250)
    """
    if is_debug_enabled():
        A.check()
    m, n = A.shape[-2], A.shape[-1]
    if m < n:
        AA = A.matmul(A.H, is_hermitian=True)
    else:
        AA = A.H.matmul(A, is_hermitian=True)
    evals, evecs = symeig(AA, k, mode, bck_options=bck_options, method=method, **fwd_options)
    s = torch.sqrt(torch.clamp(evals, min=0.0))
    sdiv = torch.clamp(s, min=1e-12).unsqueeze(-2)
    if m < n:
        u = evecs
        v = A.rmm(u) / sdiv
    else:
        v = evecs
        u = A.mm(v) / sdiv
    return u, s, v.transpose(-2, -1).conj()


def _custom_exacteig(A, neig, mode, M=None, **options):
    return exacteig(A, neig, mode, M)


_SYMEIG_METHODS = {"davidson": davidson, "custom_exacteig": _custom_exacteig}


class _SymeigFunction(torch.autograd.Function):
    """Forward: run the chosen eigensolver without a graph.  Backward: implicit differentiation of
    the (possibly degenerate) partial eigendecomposition (reference: symeig.py:252-402,
    arXiv:2011.04366): one shifted multi-RHS solve ``(A - lam_i M) g_i = -P b_i`` plus a VJP
    through ``A.mm`` (and ``M.mm``)."""

    @staticmethod
    def forward(ctx, A, neig, mode, M, fwd_options, bck_options, na, *amparams):
        params, mparams = amparams[:na], amparams[na:]
        config = merge_options({}, fwd_options)
        ctx.bck_config = merge_options({"degen_atol": None, "degen_rtol": None}, bck_options)
        ctx.bck_alg_config = pop_keys(ctx.bck_config, ["degen_atol", "degen_rtol"])
        method = config.pop("method")
        with A.uselinopparams(*params), (M.uselinopparams(*mparams) if M is not None else null_context()):
            fcn = get_method("symeig", _SYMEIG_METHODS, method)
            evals, evecs = fcn(A, neig, mode, M, **config)
        ctx.save_for_backward(evals, evecs, *amparams)
        ctx.na, ctx.A, ctx.M = na, A, M
        return evals, evecs

    @staticmethod
    def backward(ctx, grad_evals, grad_evecs):
        evals, evecs = ctx.saved_tensors[:2]
        amparams = ctx.saved_tensors[2:]
        na, A, M = ctx.na, ctx.A, ctx.M
        params, mparams = amparams[:na], amparams[na:]
        atol, rtol = ctx.bck_alg_config["degen_atol"], ctx.bck_alg_config["degen_rtol"]
        eps = torch.finfo(evals.dtype).eps
        atol = eps ** 0.6 if atol is None else atol
        rtol = eps ** 0.4 if rtol is None else rtol

        idx_degen = None
        if atol > 0 or rtol > 0:
            dmap, isdegen = _check_degen(evals, atol, rtol)
            if isdegen:
                idx_degen = dmap

        # connect A.mm(evecs) to fresh leaf copies of the operator parameters
        with torch.enable_grad():
            params = [p.clone().requires_grad_() for p in params]
            with A.uselinopparams(*params):
                loss = A.mm(evecs)

        if is_debug_enabled() and idx_degen is not None:
            xtg = torch.matmul(evecs.transpose(-2, -1).conj(), grad_evecs)
            req = idx_degen * (xtg - xtg.transpose(-2, -1).conj())
            tol = xtg.abs().max() * grad_evecs.shape[-2] * torch.finfo(grad_evecs.dtype).eps
            if not torch.all(torch.abs(req) <= tol):
                warnings.warn(MathWarning(
                    "Degeneracy appears but the loss function seem to depend strongly on the eigenvector. "
                    "The gradient might be incorrect.\nEigenvalues:\n%s\nDegenerate map:\n%s\n"
                    "Requirements (should be all 0s):\n%s" % (str(evals), str(idx_degen), str(req))))

        gevalsA = grad_evals.unsqueeze(-2) * evecs
        with (M.uselinopparams(*mparams) if M is not None else null_context()):
            Bmat = _ortho(grad_evecs, evecs, D=idx_degen, M=M, mright=False)
            shift = evals + 1e-14 if torch.is_complex(Bmat) else evals
            with A.uselinopparams(*params):
                gevecs = solve(A, -Bmat, shift, M, bck_options=ctx.bck_config, **ctx.bck_config)
            gevecsA = _ortho(gevecs, evecs, D=None, M=M, mright=True)

        gaccumA = gevalsA + gevecsA
        grad_params = torch.autograd.grad(outputs=(loss,), inputs=params, grad_outputs=(gaccumA,),
                                          create_graph=torch.is_grad_enabled())
        grad_mparams = []
        if M is not None:
            with torch.enable_grad():
                mparams = [p.clone().requires_grad_() for p in mparams]
                with M.uselinopparams(*mparams):
                    mloss = M.mm(evecs)
            ev = evals.unsqueeze(-2)
            par = (-0.5 * torch.einsum("...ae,...ae->...e", grad_evecs, evecs.conj())).unsqueeze(-2) * evecs
            gaccumM = -gevalsA * ev - gevecsA * ev + par
            grad_mparams = torch.autograd.grad(outputs=(mloss,), inputs=mparams, grad_outputs=(gaccumM,),
                                               create_graph=torch.is_grad_enabled())
        return (None, None, None, None, None, None, None, *grad_params, *grad_mparams)


def _check_degen(evals, degen_atol, degen_rtol):
    # (*, neig, neig) 0/1 map of eigenvalue pairs closer than atol + rtol*|lam|  (symeig.py:404-414)
    diff = torch.abs(evals.unsqueeze(-2) - evals.unsqueeze(-1))
    thresh = degen_atol + degen_rtol * torch.abs(evals).unsqueeze(-1)
    dmap = (diff < thresh).to(evals.dtype)
    return dmap, bool(torch.sum(dmap) > torch.numel(evals))


def _ortho(A, B, *, D=None, M=None, mright=False):
    """Remove from every column of ``A`` its component along the matching column of ``B`` (or along
    the whole degenerate block given by the map ``D``), in the ``M`` inner product
    (reference: symeig.py:416-448)."""
    if D is None:
        Bc = B.conj()
        dot = lambda X: torch.einsum("...rc,...rc->...c", X, Bc).unsqueeze(-2)
        if M is None:
            return A - dot(A) * B
        if mright:
            return A - dot(M.mm(A)) * B
        return A - M.mm(dot(A) * B)
    BH = B.transpose(-2, -1).conj()
    if M is None:
        return A - torch.matmul(B, D * torch.matmul(BH, A))
    if mright:
        return A - torch.matmul(B, D * torch.matmul(BH, M.mm(A)))
    return A - M.mm(torch.matmul(B, D * torch.matmul(BH, A)))
