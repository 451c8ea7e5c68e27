from xitorch_amd.linalg.solve import solve
from xitorch_amd.linalg.symeig import symeig, lsymeig, usymeig, svd

__all__ = ["solve", "symeig", "lsymeig", "usymeig", "svd"]
