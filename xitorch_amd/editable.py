"""EditableModule — objects whose tensors can be swapped out so that their methods act
as pure functions (needed by the implicit-function backward passes).

Same contract as the reference class (xitorch/_core/editable_module.py:14-240):
subclasses implement ``getparamnames(methodname, prefix="")`` returning the
dotted/bracketed attribute paths of every tensor that influences
``methodname``; the base class derives get/set of those tensors, the
unique-by-identity view used by ``LinearOperator.uselinopparams`` and a
debugging checker.  Pure host Python — nothing to accelerate.
"""
import copy
import inspect
import warnings
from abc import abstractmethod
import torch
from xitorch_amd._util import GetSetParamsError, get_attr, set_attr, del_attr, UniqueMap

__all__ = ["EditableModule"]

_FLOAT_TYPES = (torch.float32, torch.float64, torch.float16)


class EditableModule(object):
    # ---------------------------------------------------------------- names
    @abstractmethod
    def getparamnames(self, methodname, prefix=""):
        """List the attribute paths (with ``prefix`` prepended) of the tensors affecting
        ``methodname``; raise ``KeyError`` for an unknown method name."""
        pass

    def cached_getparamnames(self, methodname, refresh=False):
        cache = self.__dict__.setdefault("_paramnames_", {})
        if refresh or methodname not in cache:
            cache[methodname] = self.getparamnames(methodname)
        return cache[methodname]

    # ---------------------------------------------------------------- all params
    def getparams(self, methodname):
        return [get_attr(self, nm) for nm in self.cached_getparamnames(methodname)]

    def setparams(self, methodname, *params):
        """Assign ``params`` (possibly more than needed) to the named attributes; returns
        ``len(params)`` like the reference (editable_module.py:26-39)."""
        for nm, val in zip(self.cached_getparamnames(methodname), params):
            try:
                set_attr(self, nm, val)
            except TypeError:
                # e.g. replacing a torch.nn.Parameter slot by a plain tensor
                del_attr(self, nm)
                set_attr(self, nm, val)
        return len(params)

    # ---------------------------------------------------------------- unique params
    def _umap(self, methodname, allparams=None):
        maps = self.__dict__.setdefault("_unique_maps_", {})
        if methodname not in maps:
            if allparams is None:
                allparams = self.getparams(methodname)
            maps[methodname] = UniqueMap(allparams)
        return maps[methodname]

    def getuniqueparams(self, methodname, onlyleaves=False):
        allp = self.getparams(methodname)
        um = self._umap(methodname, allp)
        uniq = um.unique(allp)
        if onlyleaves:
            uniq = [p for p in uniq if p.is_leaf]
        return uniq

    def setuniqueparams(self, methodname, *uniqueparams):
        um = self._umap(methodname)
        # positions not covered by the given unique params stay None, like the reference
        slots = list(uniqueparams) + [None] * (len(um.first) - len(uniqueparams))
        return self.setparams(methodname, *um.expand(slots))

    # ---------------------------------------------------------------- debugging
    def assertparams(self, method, *args, **kwargs):
        """Check ``getparamnames`` of ``method`` against what the method really uses
        (reference: editable_module.py:177-240).  Warns about missing / excess names, raises
        ``GetSetParamsError`` if the method mutates the object's float tensors."""
        if not inspect.ismethod(method):
            raise TypeError("The input method must be a method")
        if method.__self__ is not self:
            raise RuntimeError("The method does not belong to the same instance")
        clsname = self.__class__.__name__
        mname = method.__name__

        # 1. the method must preserve the object's float tensors
        before, names0 = _collect_tensors(self)
        snap = [t.clone() for t in before]
        method(*args, **kwargs)
        after, names1 = _collect_tensors(self)
        head = "The method %s.%s does not preserve the object's float tensors: \n" % (clsname, mname)
        if len(snap) != len(after):
            raise GetSetParamsError(head + "The number of parameters changed:\n"
                                    "* number of object's parameters before: %d\n"
                                    "* number of object's parameters after : %d\n" % (len(snap), len(after)))
        for nm, t0, t1 in zip(names0, snap, after):
            if t0.shape != t1.shape:
                raise GetSetParamsError(head + "The shape of %s changed\n* (before) %s.shape: %s\n"
                                        "* (after ) %s.shape: %s\n" % (nm, nm, t0.shape, nm, t1.shape))
            if not torch.allclose(t0, t1):
                raise GetSetParamsError(head + "The value of %s changed\n* (before) %s: %s\n* (after ) %s: %s\n"
                                        % (nm, nm, t0, nm, t1))

        # 2. which tensors does the method actually depend on? swap in fresh leaves and backprop
        tensors, names = _collect_tensors(self)
        leaves = [t.clone().detach().requires_grad_() for t in tensors]
        _assign_tensors(self, list(leaves))
        try:
            out = method(*args, **kwargs)
            if not isinstance(out, torch.Tensor):
                raise RuntimeError("The method to be asserted must have a tensor output")
            grads = torch.autograd.grad(out.sum(), leaves, retain_graph=True, allow_unused=True)
        finally:
            _assign_tensors(self, list(tensors))
        used_ids = {id(t) for t, g in zip(tensors, grads) if g is not None}
        used_names = [nm for nm, g in zip(names, grads) if g is not None]

        user_names = self.getparamnames(mname)
        user_params = [get_attr(self, nm) for nm in user_names]
        for nm, p in zip(user_names, user_params):
            if not isinstance(p, torch.Tensor) or p.dtype not in _FLOAT_TYPES:
                raise GetSetParamsError("Parameter %s is a non-floating point tensor" % nm)
        user_ids = {id(p) for p in user_params}

        missing = [nm for nm, t in zip(names, tensors) if id(t) in used_ids and id(t) not in user_ids]
        # keep the order of first use
        missing = [nm for nm in used_names if nm in missing]
        if missing:
            warnings.warn("getparams for %s.%s does not include: %s" % (clsname, mname, ", ".join(missing)),
                          stacklevel=2)
        excess = [nm for nm, p in zip(user_names, user_params) if id(p) not in used_ids]
        if excess:
            warnings.warn("getparams for %s.%s has excess parameters: %s" % (clsname, mname, ", ".join(excess)),
                          stacklevel=2)
        print('"%s" method check done' % mname)


# -------------------------------------------------------------------- object traversal
def _is_float_tensor(x):
    return isinstance(x, torch.Tensor) and x.dtype in _FLOAT_TYPES


def _visit(obj, prefix, fn, depth, seen):
    """Depth-first walk over attributes / items reachable from ``obj``; calls
    ``fn(container, key, name, tensor)`` for every float tensor."""
    if isinstance(obj, torch.nn.Module):
        groups = [(obj._parameters, False), (obj._modules, False)]
    elif hasattr(obj, "__dict__"):
        groups = [(obj.__dict__, False)]
    elif isinstance(obj, dict):
        groups = [(obj, True)]
    elif isinstance(obj, (list, tuple)):
        groups = [(obj, True)]
    elif hasattr(obj, "__iter__") and not isinstance(obj, (str, bytes)):
        return
    else:
        raise RuntimeError("The object must be iterable or keyable")
    for cont, bracket in groups:
        items = cont.items() if isinstance(cont, dict) or hasattr(cont, "items") else enumerate(cont)
        for key, val in list(items):
            name = "%s[%r]" % (prefix, key) if bracket and not isinstance(key, int) else \
                ("%s[%d]" % (prefix, key) if bracket else "%s%s" % (prefix, key))
            if _is_float_tensor(val):
                fn(cont, key, name, val)
                continue
            if isinstance(val, (str, bytes)) or val is None:
                continue
            if hasattr(val, "__dict__") or isinstance(val, (dict, list, tuple)):
                if id(val) in seen:
                    continue
                seen.add(id(val))
                if depth <= 0:
                    raise RecursionError("Maximum number of recursion reached")
                sub = name + "." if hasattr(val, "__dict__") and not isinstance(val, (dict, list, tuple)) else name
                _visit(val, sub, fn, depth - 1, seen)


def _collect_tensors(obj, prefix="", max_depth=20):
    tensors, names = [], []

    def grab(cont, key, name, t):
        tensors.append(t)
        names.append(name)
    _visit(obj, prefix, grab, max_depth, set())
    return tensors, names


def _assign_tensors(obj, new_tensors, max_depth=20):
    queue = copy.copy(new_tensors)

    def put(cont, key, name, t):
        if isinstance(cont, tuple):
            return
        cont[key] = queue.pop(0)
    _visit(obj, "", put, max_depth, set())
