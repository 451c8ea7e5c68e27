"""Drop-in audit (this container only: it needs /root/reference): run the REFERENCE's own test files against
xitorch_amd through a module-name shim — a throw-away package `xitorch` under a temp directory whose modules re-export
xitorch_amd's (`xitorch.linalg.symeig` -> `xitorch_amd.linalg.symeig`, `xitorch._core.editable_module` ->
`xitorch_amd.editable`, ...) and whose `xitorch._tests` points at /root/reference/xitorch/_tests in place (nothing of
the reference is copied anywhere).

    python scripts/reference_suite_shim.py            # the CPU-runnable part of the hot path's surface
    python scripts/reference_suite_shim.py -k solve   # extra pytest arguments are passed through

What is expected to fail here: every test that drives an ITERATIVE method (davidson, cg, bicgstab, gmres, broyden1, ...)
on CPU tensors — this package has no CPU path for them and raises NativeLibraryError (by contract); those cases are
mirrored on the GPU by tests/test_gpu_*.py against goldens the reference produced (tests/golden/make_golden.py).
Default selection: everything else in test_linop, test_linop_fcns, test_jac, test_editable_module,
test_pure_function, test_debug (r02: 337 passed)."""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = "/root/reference/xitorch/_tests"

SHIM = {
    "__init__.py": "from xitorch_amd import *\nfrom xitorch_amd import LinearOperator, EditableModule\n"
                   "from . import linalg, optimize, grad, debug\n",
    "linalg/__init__.py": "from xitorch_amd.linalg import *\nfrom xitorch_amd.linalg import solve, symeig, lsymeig, usymeig, svd\n",
    "linalg/symeig.py": "from xitorch_amd.linalg.symeig import *\nfrom xitorch_amd.linalg.symeig import lsymeig, usymeig, symeig, svd\n",
    "linalg/solve.py": "from xitorch_amd.linalg.solve import *\nfrom xitorch_amd.linalg.solve import solve\n",
    "optimize/__init__.py": "from xitorch_amd.optimize import *\nfrom xitorch_amd.optimize import rootfinder, equilibrium, minimize\n",
    "grad/__init__.py": "from xitorch_amd.grad import *\n",
    "grad/jachess.py": "from xitorch_amd.grad.jachess import *\nfrom xitorch_amd.grad.jachess import jac, hess\n",
    "_core/__init__.py": "",
    "_core/editable_module.py": "from xitorch_amd.editable import *\nfrom xitorch_amd.editable import EditableModule\n",
    "_core/pure_function.py": "from xitorch_amd.purefn import *\nfrom xitorch_amd.purefn import get_pure_function, PureFunction\n",
    "_utils/__init__.py": "",
    "_utils/bcast.py": "from xitorch_amd._util import bcast_shape\n\n\ndef get_bcasted_dims(*shapes):\n    return list(bcast_shape(*shapes))\n",
    "_utils/exceptions.py": "from xitorch_amd._util import MathWarning, GetSetParamsError, ConvergenceWarning\n",
    "debug/__init__.py": "from xitorch_amd.debug import *\n",
    "debug/modes.py": "from xitorch_amd.debug import enable_debug, disable_debug, is_debug_enabled, set_debug_mode\n",
    "_tests/__init__.py": "__path__ = [%r]\n" % REF_TESTS,
}
DEFAULT = ["test_linop.py", "test_linop_fcns.py", "test_jac.py", "test_editable_module.py", "test_pure_function.py",
           "test_debug.py"]
ITERATIVE = "not large_methods and not _methods"      # the iterative-method cases (HIP only here)


def main():
    if not os.path.isdir(REF_TESTS):
        sys.exit("needs %s (this audit runs in the build container only)" % REF_TESTS)
    with tempfile.TemporaryDirectory() as tmp:
        root = os.path.join(tmp, "xitorch")
        for rel, txt in SHIM.items():
            path = os.path.join(root, rel)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as fh:
                fh.write(txt)
        # the reference tree is read-only for us: no bytecode caches next to its test files
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([tmp, REPO, os.environ.get("PYTHONPATH", "")]),
                   PYTHONDONTWRITEBYTECODE="1")
        extra = sys.argv[1:]
        cmd = [sys.executable, "-B", "-m", "pytest", "-q", "--no-header", "-p", "no:cacheprovider",
               "--rootdir", tmp, "--import-mode=importlib"]
        cmd += [os.path.join(REF_TESTS, f) for f in DEFAULT]
        cmd += extra if extra else ["-k", ITERATIVE]
        sys.exit(subprocess.call(cmd, cwd=tmp, env=env))


if __name__ == "__main__":
    main()
