"""Pure-function wrappers: expose the hidden tensors of a bound method so that
``torch.autograd.Function`` can take them as explicit inputs.

Same surface as the reference (xitorch/_core/pure_function.py): ``get_pure_function``
accepts a plain function, a method of an ``EditableModule`` or of a
``torch.nn.Module`` (or an already wrapped function); the wrapper offers
``objparams()``, ``useobjparams(list)`` and ``make_sibling``.  Host Python only.
"""
import inspect
from contextlib import contextmanager
import torch
from xitorch_amd._util import UniqueMap, set_attr, del_attr
from xitorch_amd.editable import EditableModule

__all__ = ["get_pure_function", "make_sibling", "PureFunction"]


class PureFunction(object):
    """Callable whose object-held tensors (``objparams``) can be temporarily replaced."""

    def __init__(self, fcntocall):
        self._fcn = fcntocall
        self._locked = False
        self._all = self._read_obj_params()
        self._umap = UniqueMap(self._all)
        self._current = self._umap.unique()
        self._stack = []   # (previous objparams, was_identical)

    def __call__(self, *params):
        return self._fcn(*params)

    # subclasses: how to read / write the underlying object's tensors
    def _read_obj_params(self):
        return []

    def _write_obj_params(self, allobjparams):
        pass

    # kept under the reference's private names too: sibling wrappers delegate through them
    def _get_all_obj_params_init(self):
        return self._read_obj_params()

    def _set_all_obj_params(self, allobjparams):
        self._write_obj_params(allobjparams)

    def objparams(self):
        return self._current

    def set_objparams(self, objparams):
        same = len(objparams) == len(self._current) and \
            all(a is b for a, b in zip(objparams, self._current))
        self._stack.append((self._current, same))
        if not same:
            self._write_obj_params(self._umap.expand(list(objparams)))
            self._current = list(objparams)

    def restore_objparams(self):
        old, same = self._stack.pop()
        if not same:
            self._write_obj_params(self._umap.expand(list(old)))
            self._current = old

    @contextmanager
    def useobjparams(self, objparams):
        if self._locked:
            raise RuntimeError("The state change is disabled")
        self.set_objparams(objparams)
        try:
            yield
        finally:
            self.restore_objparams()

    @contextmanager
    def disable_state_change(self):
        prev, self._locked = self._locked, True
        try:
            yield
        finally:
            self._locked = prev


class _PlainFunction(PureFunction):
    pass


class _EditableMethod(PureFunction):
    def __init__(self, obj, method):
        self.obj, self.method = obj, method
        super().__init__(method)

    def _read_obj_params(self):
        return list(self.obj.getparams(self.method.__name__))

    def _write_obj_params(self, allobjparams):
        self.obj.setparams(self.method.__name__, *allobjparams)


class _ModuleMethod(PureFunction):
    def __init__(self, obj, method):
        self.obj, self.method = obj, method
        super().__init__(method)

    def _read_obj_params(self):
        named = list(self.obj.named_parameters())
        self.names = [n for n, _ in named]
        return [p for _, p in named]

    def _write_obj_params(self, allobjparams):
        for name, p in zip(self.names, allobjparams):
            del_attr(self.obj, name)   # the slot may hold an nn.Parameter
            set_attr(self.obj, name, p)


class _Sibling(PureFunction):
    """A different callable sharing the state of one or several wrapped functions."""

    def __init__(self, fcns, fcntocall):
        self.pfuncs = [get_pure_function(f) for f in fcns]
        super().__init__(fcntocall)

    def _read_obj_params(self):
        out, self._cuts = [], [0]
        for pf in self.pfuncs:
            out = out + list(pf._get_all_obj_params_init())
            self._cuts.append(len(out))
        return out

    def _write_obj_params(self, allobjparams):
        for i, pf in enumerate(self.pfuncs):
            pf._set_all_obj_params(allobjparams[self._cuts[i]:self._cuts[i + 1]])


_ERR = ("The input function must be a function, a method of torch.nn.Module, a method of "
        "xitorch.EditableModule, or a sibling method")


def get_pure_function(fcn):
    """Wrap ``fcn`` into a :class:`PureFunction` (reference: pure_function.py:161-203)."""
    if isinstance(fcn, PureFunction):
        return fcn
    if inspect.isfunction(fcn) or isinstance(fcn, torch.jit.ScriptFunction):
        return _PlainFunction(fcn)
    if inspect.ismethod(fcn) or hasattr(fcn, "__call__"):
        if inspect.ismethod(fcn):
            obj = fcn.__self__
        else:
            obj, fcn = fcn, fcn.__call__
        if isinstance(obj, EditableModule):
            return _EditableMethod(obj, fcn)
        if isinstance(obj, torch.nn.Module):
            return _ModuleMethod(obj, fcn)
    raise RuntimeError(_ERR)


def make_sibling(*pfuncs):
    """Decorator: the decorated callable shares (and switches) the object state of ``pfuncs``
    (reference: pure_function.py:205-219)."""
    if len(pfuncs) == 0:
        raise TypeError("At least 1 function is required as the argument")
    return lambda fcn: _Sibling(pfuncs, fcntocall=fcn)
