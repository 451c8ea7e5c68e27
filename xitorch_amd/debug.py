"""Process-global debug flag (reference: xitorch/debug/modes.py).  When enabled the
functional front-ends run ``LinearOperator.check()`` / ``assertparams`` on their inputs."""
from contextlib import contextmanager

__all__ = ["is_debug_enabled", "set_debug_mode", "enable_debug", "disable_debug"]

_state = {"debug": False}


def set_debug_mode(mode):
    _state["debug"] = bool(mode)


def is_debug_enabled():
    return _state["debug"]


@contextmanager
def enable_debug():
    prev = _state["debug"]
    _state["debug"] = True
    try:
        yield
    finally:
        _state["debug"] = prev


@contextmanager
def disable_debug():
    prev = _state["debug"]
    _state["debug"] = False
    try:
        yield
    finally:
        _state["debug"] = prev
