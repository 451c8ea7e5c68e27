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
from xitorch_amd.purefn import make_sibling
from xitorch_amd.optimize.native_root import newton, broyden1, broyden2, linearmixing
from xitorch_amd.optimize.extra import anderson_acc, gd, adam

__all__ = ["rootfinder", "equilibrium", "minimize"]

_RF_METHODS = {"newton": newton, "broyden1": broyden1, "broyden2": broyden2, "linearmixing": linearmixing}
_EQUIL_METHODS = {"anderson_acc": anderson_acc}      # fixed-point-only methods (besides all root-finder ones)
_OPT_METHODS = {"gd": gd, "adam": adam}
_METHOD_TABLES = {"rootfinder": _RF_METHODS, "equilibrium": _EQUIL_METHODS, "minimizer": _OPT_METHODS}


def _debug_check(fcn, args):
    if is_debug_enabled():
        import inspect
        if inspect.ismethod(fcn) and isinstance(fcn.__self__, EditableModule):
            fcn.__self__.assertparams(fcn, *args)


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
    _debug_check(fcn, (y0, *params))
    pfunc = get_pure_function(fcn)
    fwd_options["method"] = "broyden1" if method is None else method
    return _RootFinder.apply(pfunc, y0, pfunc, "rootfinder", fwd_options, bck_options, len(params), *params,
                             *pfunc.objparams())


def equilibrium(fcn, y0, params=[], bck_options={}, method=None, **fwd_options):
    r"""
    Solve the equilibrium (fixed-point) equation :math:`\mathbf{y} = \mathbf{f}(\mathbf{y}, \theta)`, i.e. the
    root of :math:`\mathbf{y} - \mathbf{f}(\mathbf{y}, \theta)` (reference: optimize/rootfinder.py:104-184).

    ``method``: any root-finder method (default ``"broyden1"``) or ``"anderson_acc"``; arguments as for
    :func:`rootfinder`.
    """
    _debug_check(fcn, (y0, *params))
    pfunc = get_pure_function(fcn)

    @make_sibling(pfunc)
    def deviation(y, *p):
        return y - pfunc(y, *p)

    method = "broyden1" if method is None else method
    fwd_options["method"] = method
    fixed_point_method = isinstance(method, str) and method.lower() in _EQUIL_METHODS
    fwd_fcn = pfunc if fixed_point_method else deviation
    alg_type = "equilibrium" if fixed_point_method else "rootfinder"
    return _RootFinder.apply(deviation, y0, fwd_fcn, alg_type, fwd_options, bck_options, len(params), *params,
                             *pfunc.objparams())


def minimize(fcn, y0, params=[], bck_options={}, method=None, **fwd_options):
    r"""
    Solve the unbounded minimisation :math:`\mathbf{y^*} = \arg\min_\mathbf{y} f(\mathbf{y}, \theta)` of a
    scalar function (reference: optimize/rootfinder.py:186-288).  The solution is differentiable through the
    stationarity condition :math:`\nabla_y f = 0`.

    ``method``: ``"broyden1"`` (default) or any root-finder method applied to the gradient, or the
    minimisers ``"gd"`` / ``"adam"``; arguments as for :func:`rootfinder`.
    """
    _debug_check(fcn, (y0, *params))
    pfunc = get_pure_function(fcn)
    method = "broyden1" if method is None else method
    fwd_options["method"] = method
    opt_method = not (isinstance(method, str) and method.lower() in _RF_METHODS)

    @make_sibling(pfunc)
    def value_and_grad(y, *p):
        with torch.enable_grad():
            y1 = y.clone().requires_grad_()
            z = pfunc(y1, *p)
        gy, = torch.autograd.grad(z, (y1,), retain_graph=True, create_graph=torch.is_grad_enabled())
        return z, gy

    @make_sibling(value_and_grad)
    def grad_only(y, *p):
        return value_and_grad(y, *p)[1]

    fwd_fcn = value_and_grad if opt_method else grad_only
    alg_type = "minimizer" if opt_method else "rootfinder"
    return _RootFinder.apply(grad_only, y0, fwd_fcn, alg_type, fwd_options, bck_options, len(params), *params,
                             *pfunc.objparams())


class _RootFinder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fcn, y0, fwd_fcn, alg_type, options, bck_options, nparams, *allparams):
        config = options
        ctx.bck_options = bck_options
        params, objparams = allparams[:nparams], allparams[nparams:]
        with fwd_fcn.useobjparams(objparams):
            method = config.pop("method")
            y = get_method(alg_type, _METHOD_TABLES[alg_type], method)(fwd_fcn, y0, params, **config)
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
