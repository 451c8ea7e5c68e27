"""rootfinder — functional front-end of the root-finding path, with the implicit-function backward.

Same signature, defaults and backward as the reference (xitorch/optimize/rootfinder.py:35-102,
290-366): ``y`` with ``f(y, *params) = 0``; backward solves ``(df/dy)^H g = -grad_y`` through
``xitorch_amd.linalg.solve`` on the autograd Jacobian operator and pulls ``g`` back through ``f``.
The forward methods are the native ones of xitorch_amd/optimize/native_root.py.
"""
import torch
from xitorch_amd._util import ParamSplitter, get_method
from xitorch_amd.purefn import get_pure_function
from xitorch_amd.linalg.solve import solve
from xitorch_amd.grad.jachess import jac
from xitorch_amd.debug import is_debug_enabled
from xitorch_amd.editable import EditableModule
from xitorch_amd.optimize.native_root import newton, broyden1, broyden2, linearmixing

__all__ = ["rootfinder"]

_RF_METHODS = {"newton": newton, "broyden1": broyden1, "broyden2": broyden2, "linearmixing": linearmixing}


def rootfinder(fcn, y0, params=[], bck_options={}, method=None, **fwd_options):
    r"""
    Solve :math:`\mathbf{0} = \mathbf{f}(\mathbf{y}, \theta)` for :math:`\mathbf{y}`.

    Arguments
    ---------
    fcn : callable
        The function :math:`\mathbf{f}` with output tensor of the shape of ``y0``
    y0 : torch.Tensor
        Initial guess of the solution
    params : list
        Other parameters of ``fcn``
    bck_options : dict
        Options of :func:`xitorch_amd.linalg.solve` for the backward pass
    method : str or callable or None
        ``"broyden1"`` (default), ``"broyden2"``, ``"linearmixing"``, ``"newton"``, or a callable
        ``f(fcn, y0, params, **fwd_options) -> y``
    **fwd_options
        Method-specific options

    Example
    -------
    >>> def func1(y, A):
    ...     return torch.tanh(A @ y + 0.1) + y / 2.0
    >>> A = torch.tensor([[1.1, 0.4], [0.3, 0.8]], device="cuda").requires_grad_()
    >>> yroot = rootfinder(func1, torch.zeros((2, 1), device="cuda"), params=(A,))
    """
    if is_debug_enabled():
        import inspect
        if inspect.ismethod(fcn) and isinstance(fcn.__self__, EditableModule):
            fcn.__self__.assertparams(fcn, y0, *params)
    pfunc = get_pure_function(fcn)
    fwd_options["method"] = "broyden1" if method is None else method
    return _RootFinder.apply(pfunc, y0, pfunc, "rootfinder", fwd_options, bck_options, len(params), *params,
                             *pfunc.objparams())


class _RootFinder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fcn, y0, fwd_fcn, alg_type, options, bck_options, nparams, *allparams):
        config = options
        ctx.bck_options = bck_options
        params, objparams = allparams[:nparams], allparams[nparams:]
        with fwd_fcn.useobjparams(objparams):
            method = config.pop("method")
            y = get_method(alg_type, _RF_METHODS, method)(fwd_fcn, y0, params, **config)
        ctx.fcn = fcn
        ctx.nparams = nparams
        ctx.param_sep = ParamSplitter(allparams)
        ctx.save_for_backward(y, *ctx.param_sep.get_tensor_params())
        return y

    @staticmethod
    def backward(ctx, grad_yout):
        sep = ctx.param_sep
        yout = ctx.saved_tensors[0]
        tensor_params = ctx.saved_tensors[1:]
        allparams = sep.reconstruct_params(tensor_params)
        n = ctx.nparams
        params, objparams = allparams[:n], allparams[n:]
        fcn = ctx.fcn
        with fcn.useobjparams(objparams):
            jac_dfdy = jac(fcn, params=(yout, *params), idxs=[0])[0]
            g = solve(A=jac_dfdy.H, B=-grad_yout.reshape(-1, 1), bck_options=ctx.bck_options, **ctx.bck_options)
            g = g.reshape(grad_yout.shape)
            with torch.enable_grad():
                copies = [p.clone().requires_grad_() for p in tensor_params]
                allcopy = sep.reconstruct_params(copies)
                with fcn.useobjparams(allcopy[n:]):
                    yfcn = fcn(yout, *allcopy[:n])
            grads = torch.autograd.grad(yfcn, copies, grad_outputs=g, create_graph=torch.is_grad_enabled(),
                                        allow_unused=True)
            grad_params = sep.reconstruct_params(grads, [None] * sep.nnontensors())
        return (None, None, None, None, None, None, None, *grad_params)
