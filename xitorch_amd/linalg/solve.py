"""solve — functional front-end of the linear-equation path  A X = B  /  A X - M X E = B.

Same signature, defaults, argument checks and adjoint backward as the reference
(xitorch/linalg/solve.py:13-243).  The iterative methods ("cg", "bicgstab", "gmres",
"broyden1") are the native HIP implementations of xitorch_amd/linalg/native_krylov.py;
"exactsolve" is the dense `torch.linalg.solve` path.
"""
import warnings
import torch
from xitorch_amd.linop import MatrixLinearOperator
from xitorch_amd.debug import is_debug_enabled
from xitorch_amd._util import assert_runtime, merge_options, null_context, get_method

__all__ = ["solve"]


def solve(A, B, E=None, M=None, bck_options={}, method=None, **fwd_options):
    r"""
    Solve :math:`\mathbf{AX=B}` or :math:`\mathbf{AX-MXE=B}` (``E`` a diagonal matrix given as
    ``(*BE, ncols)``) for ``X`` of shape ``(..., na, ncols)``.

    Arguments
    ---------
    A: LinearOperator ``(*BA, na, na)``
    B: torch.Tensor ``(*BB, na, ncols)``
    E: torch.Tensor ``(*BE, ncols)`` or None
    M: LinearOperator ``(*BM, na, na)`` (Hermitian) or None — ignored when ``E`` is None
    bck_options: dict
        Options of the solver used in the backward pass
    method: str or callable or None
        ``None`` picks ``"exactsolve"`` for dense matrices / ``na <= 5``, else ``"cg"`` for Hermitian
        and ``"bicgstab"`` for general operators.  A callable ``f(A, B, E, M, **fwd_options) -> X``
        plugs in a user method.
    **fwd_options
        Method-specific options
    """
    assert_runtime(A.shape[-1] == A.shape[-2], "The linear operator A must have a square shape")
    assert_runtime(A.shape[-1] == B.shape[-2], "Mismatch shape of A & B (A: %s, B: %s)" % (A.shape, B.shape))
    assert_runtime(not torch.is_grad_enabled() or A.is_getparamnames_implemented,
                   "The _getparamnames(self, prefix) of linear operator A must be "
                   "implemented if using solve with grad enabled")
    if M is not None:
        assert_runtime(M.shape[-1] == M.shape[-2], "The linear operator M must have a square shape")
        assert_runtime(M.shape[-1] == A.shape[-1], "The shape of A & M must match (A: %s, M: %s)" % (A.shape, M.shape))
        assert_runtime(M.is_hermitian, "The linear operator M must be a Hermitian matrix")
        assert_runtime(not torch.is_grad_enabled() or M.is_getparamnames_implemented,
                       "The _getparamnames(self, prefix) of linear operator M must be "
                       "implemented if using solve with grad enabled")
    if E is not None:
        assert_runtime(E.shape[-1] == B.shape[-1],
                       "The last dimension of E & B must match (E: %s, B: %s)" % (E.shape, B.shape))
    if E is None and M is not None:
        warnings.warn("M is supplied but will be ignored because E is not supplied")

    if is_debug_enabled():
        A.check()
        if M is not None:
            M.check()

    if method is None:
        dense = isinstance(A, MatrixLinearOperator) and (M is None or isinstance(M, MatrixLinearOperator))
        if dense or A.shape[-1] <= 5:
            method = "exactsolve"
        else:
            method = "cg" if (A.is_hermitian and (M is None or M.is_hermitian)) else "bicgstab"

    if method == "exactsolve":
        from xitorch_amd.linalg.native_krylov import exactsolve
        return exactsolve(A, B, E, M)
    params = A.getlinopparams()
    mparams = M.getlinopparams() if M is not None else []
    return _SolveFunction.apply(A, B, E, M, method, fwd_options, bck_options, len(params), *params, *mparams)


class _SolveFunction(torch.autograd.Function):
    """Forward: the chosen solver, graph-free.  Backward: one adjoint solve
    ``(A - E M)^H v = grad_X`` and VJPs through ``A.mm`` / ``M.mm`` (reference: solve.py:118-222)."""

    @staticmethod
    def forward(ctx, A, B, E, M, method, fwd_options, bck_options, na, *all_params):
        from xitorch_amd.linalg import native_krylov as nk
        params, mparams = all_params[:na], all_params[na:]
        config = merge_options({}, fwd_options)
        ctx.bck_config = merge_options({}, bck_options)
        from xitorch_amd.dist import all_ranks_agree_true
        # sharded runs (process_group in the options): the zero-rhs shortcut must be collective, see dist.py
        if all_ranks_agree_true(bool(torch.all(B == 0)), B.device, config.get("process_group")):
            dims = (*nk.get_batchdims(A, B, E, M), *B.shape[-2:])
            x = torch.zeros(dims, dtype=B.dtype, device=B.device)
        else:
            with A.uselinopparams(*params), (M.uselinopparams(*mparams) if M is not None else null_context()):
                methods = {
                    "custom_exactsolve": nk.custom_exactsolve,
                    "broyden1": nk.broyden1_solve,
                    "cg": nk.cg,
                    "bicgstab": nk.bicgstab,
                    "gmres": nk.gmres,
                    "scipy_gmres": nk.scipy_gmres,
                }
                x = get_method("solve", methods, method)(A, B, E, M, **config)
        ctx.e_is_none = E is None
        ctx.A, ctx.M, ctx.na = A, M, na
        if ctx.e_is_none:
            ctx.save_for_backward(x, *all_params)
        else:
            ctx.save_for_backward(x, E, *all_params)
        return x

    @staticmethod
    def backward(ctx, grad_x):
        x = ctx.saved_tensors[0]
        first = 1 if ctx.e_is_none else 2
        all_params = ctx.saved_tensors[first:]
        params, mparams = all_params[:ctx.na], all_params[ctx.na:]
        E = None if ctx.e_is_none else ctx.saved_tensors[1]
        A, M = ctx.A, ctx.M

        with A.uselinopparams(*params), (M.uselinopparams(*mparams) if M is not None else null_context()):
            v = solve(A.H, grad_x, E.conj() if E is not None else None, M.H if M is not None else None,
                      bck_options=ctx.bck_config, **ctx.bck_config)
        grad_B = v

        with torch.enable_grad():
            params = [p.clone().requires_grad_() for p in params]
            with A.uselinopparams(*params):
                loss = -A.mm(x)
        grad_params = torch.autograd.grad((loss,), params, grad_outputs=(v,),
                                          create_graph=torch.is_grad_enabled(), allow_unused=True)

        grad_E = None
        if E is not None:
            if M is None:
                Mx = x
            else:
                with M.uselinopparams(*mparams):
                    Mx = M.mm(x)
            grad_E = torch.einsum("...rc,...rc->...c", v, Mx.conj())

        grad_mparams = []
        if M is not None and E is not None:
            with torch.enable_grad():
                mparams = [p.clone().requires_grad_() for p in mparams]
                with M.uselinopparams(*mparams):
                    mloss = M.mm(x * E.unsqueeze(-2))
            grad_mparams = torch.autograd.grad((mloss,), mparams, grad_outputs=(v,),
                                               create_graph=torch.is_grad_enabled(), allow_unused=True)
        return (None, grad_B, grad_E, None, None, None, None, None, *grad_params, *grad_mparams)
