"""xitorch_amd — MI355X-native iterative linear algebra behind the xitorch operator API.

Drop-in surface (same names as the reference package `xitorch`):
    LinearOperator, EditableModule, get_pure_function, make_sibling,
    ConvergenceWarning, MathWarning, GetSetParamsError, debug mode helpers,
    xitorch_amd.linalg.{symeig, lsymeig, usymeig, svd, solve},
    xitorch_amd.optimize.rootfinder, xitorch_amd.grad.{jac, hess}
"""
from xitorch_amd.editable import EditableModule
from xitorch_amd.purefn import get_pure_function, make_sibling, PureFunction
from xitorch_amd.linop import LinearOperator, MatrixLinearOperator, BandedLinearOperator, RowShardedMatrixLinearOperator
from xitorch_amd.debug import is_debug_enabled, set_debug_mode, enable_debug, disable_debug
from xitorch_amd._util import ConvergenceWarning, MathWarning, GetSetParamsError

__version__ = "0.1.0"
