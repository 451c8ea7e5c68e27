"""jac / hess — Jacobian and Hessian of a function as LinearOperators built on torch.autograd.

Same behaviour as the reference (xitorch/grad/jachess.py:11-224): ``jac(fcn, params, idxs)`` returns
operator(s) of shape ``(nout, nin)`` whose ``mv`` is a forward-mode product obtained with the
double-backward trick and whose ``rmv`` is a vector-Jacobian product.  The user function is
arbitrary Python, so this stays autograd (K17 in SURVEY.md); it is the operator of the implicit
backward of ``rootfinder``, where each apply costs one pass through ``fcn``.
"""
import warnings
import torch
from xitorch_amd.linop import LinearOperator
from xitorch_amd.purefn import get_pure_function, make_sibling
from xitorch_amd._util import ParamSplitter, assert_type

__all__ = ["jac", "hess"]


def _normalise_idxs(idxs, params):
    if idxs is None:
        idxs = [i for i, t in enumerate(params) if isinstance(t, torch.Tensor) and t.requires_grad]
    elif isinstance(idxs, int):
        idxs = [idxs]
    for p in idxs:
        assert_type(isinstance(params[p], torch.Tensor) and params[p].requires_grad,
                    "The %d-th element (0-based) must be a tensor which requires grad" % p)
    return idxs


def jac(fcn, params, idxs=None):
    """LinearOperator(s) acting as the Jacobian of ``fcn`` w.r.t. ``params[idxs]``; shape
    ``(numel(out), numel(params[idx]))``.  An int ``idxs`` returns one operator, otherwise a list."""
    idxs_list = _normalise_idxs(idxs, params)
    pfcn = get_pure_function(fcn)
    res = [_Jac(pfcn, params, idx) for idx in idxs_list]
    return res[0] if isinstance(idxs, int) else res


def hess(fcn, params, idxs=None):
    """LinearOperator(s) acting as the Hessian of the scalar function ``fcn`` w.r.t. ``params[idxs]``."""
    idxs_list = _normalise_idxs(idxs, params)
    pfcn = get_pure_function(fcn)

    def gradient_of(idx):
        @make_sibling(pfcn)
        def grad_fcn(*p):
            with torch.enable_grad():
                z = pfcn(*p)
            g, = torch.autograd.grad(z, (p[idx],), retain_graph=True, create_graph=torch.is_grad_enabled())
            return g
        return grad_fcn

    res = []
    for idx in idxs_list:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")      # Hermitian operator that also implements _rmv
            res.append(_Jac(gradient_of(idx), params, idx, is_hermitian=True))
    return res[0] if isinstance(idxs, int) else res


class _Jac(LinearOperator):
    def __init__(self, fcn, params, idx, is_hermitian=False):
        yparam = params[idx]
        with torch.enable_grad():
            yout = fcn(*params)
            v = torch.ones_like(yout).to(yout.device).requires_grad_()
            dfdy, = torch.autograd.grad(yout, (yparam,), grad_outputs=v, create_graph=True)
        self.inshape, self.outshape = yparam.shape, yout.shape
        self.nin, self.nout = torch.numel(yparam), torch.numel(yout)
        super().__init__(shape=(self.nout, self.nin), is_hermitian=is_hermitian, dtype=yparam.dtype,
                         device=yparam.device)
        self.fcn = fcn
        self.yparam = yparam
        self.params = list(params)
        self.objparams = fcn.objparams()
        self.yout, self.v, self.dfdy, self.idx = yout, v, dfdy, idx
        self.param_sep = ParamSplitter(params)
        self.params_tensor = self.param_sep.get_tensor_params()
        self._ids = ([id(p) for p in self.params_tensor], [id(p) for p in self.objparams])

    def _getparamnames(self, prefix=""):
        return [prefix + "yparam"] + \
            [prefix + "params_tensor[%d]" % i for i in range(len(self.params_tensor))] + \
            [prefix + "objparams[%d]" % i for i in range(len(self.objparams))]

    def _unchanged(self):
        return ([id(p) for p in self.params_tensor], [id(p) for p in self.objparams]) == self._ids

    def _reevaluate(self, need_double):
        # the operator's tensors were swapped (uselinopparams): rebuild the graph with the new ones
        with torch.enable_grad(), self.fcn.useobjparams(self.objparams):
            self.params = self.param_sep.reconstruct_params(self.params_tensor)
            yparam = self.params[self.idx]
            yout = self.fcn(*self.params)
            if not need_double:
                return yout, yparam, None, None
            v = torch.ones_like(yout).to(yout.device).requires_grad_()
            dfdy, = torch.autograd.grad(yout, (yparam,), grad_outputs=v, create_graph=True)
            return yout, yparam, v, dfdy

    def _mv(self, gy):
        # J g = d/dv [ (dy/dx)^T v ] . g   (double-backward trick)
        if self._unchanged():
            v, dfdy = self.v, self.dfdy
        else:
            _, _, v, dfdy = self._reevaluate(True)
        rows = gy.reshape(-1, self.nin)
        res = _batched_grad(dfdy, v, rows, self.inshape).reshape(*gy.shape[:-1], self.nout)
        return _connect(_connect(res, self.params_tensor), self.objparams)

    def _rmv(self, gout):
        if self._unchanged():
            yout, yparam = self.yout, self.yparam
        else:
            yout, yparam, _, _ = self._reevaluate(False)
        rows = gout.reshape(-1, self.nout)
        res = _batched_grad(yout, yparam, rows, self.outshape).reshape(*gout.shape[:-1], self.nin)
        return _connect(_connect(res, self.params_tensor), self.objparams)


def _batched_grad(out, inp, rows, shape):
    """rows[i] -> d<out, rows[i]>/d inp for every row: ONE vmapped backward pass (``is_grads_batched``)
    instead of the reference's Python loop over the vectors (jachess.py:163-170,189-196); graphs that
    cannot be vmapped (or a single row) take the loop."""
    n = rows.shape[0]
    create = torch.is_grad_enabled()
    if n > 1:
        try:
            o, = torch.autograd.grad(out, (inp,), grad_outputs=rows.reshape(n, *shape), retain_graph=True,
                                     create_graph=create, is_grads_batched=True)
            return o.reshape(n, -1)
        except RuntimeError:
            pass
    outs = []
    for i in range(n):
        o, = torch.autograd.grad(out, (inp,), grad_outputs=rows[i].reshape(shape), retain_graph=True,
                                 create_graph=create)
        outs.append(o.reshape(1, -1))
    return torch.cat(outs, dim=0)


def _connect(out, params):
    # keep every parameter attached to the graph even when df/dy does not depend on it
    return out + sum(p.reshape(-1)[0] * 0 for p in params)
