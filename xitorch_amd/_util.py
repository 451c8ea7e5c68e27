"""Small host-side helpers shared by the functional front-ends.

Mirrors the behaviour (names, error types and messages) of the reference's
xitorch/_utils/{misc,bcast,attr,exceptions,assertfuncs,unique}.py, rewritten
for this package.
"""
import ast
import contextlib
import re
import torch

__all__ = ["ConvergenceWarning", "MathWarning", "GetSetParamsError", "get_method", "merge_options",
           "pop_keys", "bcast_shape", "pad_shapes", "get_attr", "set_attr", "del_attr",
           "ParamSplitter", "assert_runtime", "assert_type", "null_context", "UniqueMap"]


# ---------------------------------------------------------------- warnings / errors
class GetSetParamsError(Exception):
    """Raised by EditableModule.assertparams (reference: _utils/exceptions.py:6-7)."""


class ConvergenceWarning(Warning):
    """An iterative algorithm stopped without reaching its tolerance (reference: exceptions.py:9-13).
    The best iterate is still returned — non-convergence is never an exception."""


class MathWarning(Warning):
    """A mathematical precondition is violated, e.g. degenerate eigenvectors in a
    gradient (reference: exceptions.py:15-19)."""


def assert_runtime(cond, msg=""):
    if not cond:
        raise RuntimeError(msg)


def assert_type(cond, msg=""):
    if not cond:
        raise TypeError(msg)


@contextlib.contextmanager
def null_context():
    yield None


# ---------------------------------------------------------------- method plug-in
def get_method(algname, methods, method):
    """Resolve `method=` (reference contract: _utils/misc.py:21-39).

    A string is looked up case-insensitively in `methods`; any callable is
    returned as is (the user plug-in hook); unknown names raise
    RuntimeError("Unknown <alg> method: <name>").
    """
    if isinstance(method, str):
        key = method.lower()
        if key in methods:
            return methods[key]
        raise RuntimeError("Unknown %s method: %s" % (algname, method))
    if callable(method):
        return method
    if method is None:
        raise AssertionError("internal error: the default %s method was not set" % algname)
    raise TypeError("Invalid method type: %s. Only str and callable are accepted." % type(method))


def merge_options(defaults, given):
    out = dict(defaults)
    out.update(given)
    return out


def pop_keys(dct, keys):
    return {k: dct.pop(k) for k in keys}


# ---------------------------------------------------------------- broadcasting
def pad_shapes(*shapes):
    n = max(len(s) for s in shapes)
    return [[1] * (n - len(s)) + list(s) for s in shapes]


def bcast_shape(*shapes):
    return list(torch.broadcast_shapes(*[tuple(s) for s in shapes]))


# ---------------------------------------------------------------- attribute paths  "a.b[0]['k'].c"
_TOKEN = re.compile(r"\[[^\]]+\]|[^.\[\]]+")


def _walk(obj, tokens):
    for tok in tokens:
        obj = _step_get(obj, tok)
    return obj


def _key(tok):
    return ast.literal_eval(tok[1:-1])


def _step_get(obj, tok):
    if tok[0] == "[":
        if not isinstance(obj, (dict, list, tuple)):
            raise TypeError("The parameter with [] must be either a dictionary or a list. Got type: %s" % type(obj))
        return obj[_key(tok)]
    return getattr(obj, tok)


def get_attr(obj, name):
    return _walk(obj, _TOKEN.findall(name))


def set_attr(obj, name, val):
    toks = _TOKEN.findall(name)
    parent = _walk(obj, toks[:-1])
    last = toks[-1]
    if last[0] == "[":
        if not isinstance(parent, (dict, list)):
            raise TypeError("The parameter with [] must be either a dictionary or a list. Got type: %s" % type(parent))
        parent[_key(last)] = val
    else:
        setattr(parent, last, val)


def del_attr(obj, name):
    toks = _TOKEN.findall(name)
    parent = _walk(obj, toks[:-1])
    last = toks[-1]
    if last[0] == "[":
        k = _key(last)
        if isinstance(parent, list):
            parent[k] = None        # keep the length
        else:
            del parent[k]
    else:
        delattr(parent, last)


# ---------------------------------------------------------------- tensor / non-tensor split
class ParamSplitter:
    """Split a parameter list into differentiable tensors and everything else, and put them
    back together (reference: TensorNonTensorSeparator, _utils/misc.py:45-95)."""

    def __init__(self, params, varonly=True):
        self.n = len(params)
        self.t_idx, self.t_val, self.o_idx, self.o_val = [], [], [], []
        for i, p in enumerate(params):
            if isinstance(p, torch.Tensor) and (p.requires_grad or not varonly):
                self.t_idx.append(i)
                self.t_val.append(p)
            else:
                self.o_idx.append(i)
                self.o_val.append(p)

    def get_tensor_params(self):
        return self.t_val

    def ntensors(self):
        return len(self.t_idx)

    def nnontensors(self):
        return len(self.o_idx)

    def reconstruct_params(self, tensors, others=None):
        others = self.o_val if others is None else others
        if len(tensors) + len(others) != self.n:
            raise ValueError("The total length of tensor and nontensor params do not match with the "
                             "expected length: %d instead of %d" % (len(tensors) + len(others), self.n))
        out = [None] * self.n
        for i, p in zip(self.o_idx, others):
            out[i] = p
        for i, p in zip(self.t_idx, tensors):
            out[i] = p
        return out


class UniqueMap:
    """Identity-based de-duplication of a list of objects (reference: Uniquifier, _utils/unique.py)."""

    def __init__(self, objs):
        self.n = len(objs)
        seen = {}
        self.first = []          # index of the first occurrence of every distinct object
        self.slot = []           # for every position: which distinct object it is
        for i, o in enumerate(objs):
            k = id(o)
            if k not in seen:
                seen[k] = len(self.first)
                self.first.append(i)
            self.slot.append(seen[k])
        self.uniques = [objs[i] for i in self.first]

    def unique(self, objs=None):
        if objs is None:
            return self.uniques
        assert_runtime(len(objs) == self.n, "The allobjs must have %d elements" % self.n)
        return [objs[i] for i in self.first]

    def expand(self, uniq):
        assert_runtime(len(uniq) == len(self.first), "The uniqueobjs must have %d elements" % len(self.first))
        return [uniq[s] for s in self.slot]
