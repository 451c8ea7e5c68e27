"""A user-defined (`_mv` only) operator through the native Davidson: the reference's `ALarge` test operator
(xitorch/_tests/test_linop_fcns.py:133-153: diag(arange) + 1e-3 (shift + shift^T)) at N = 1e6, batch 2.  What a generic
operator costs: its panel product is the user's torch ops on the strided Fortran-order view (`LinearOperator.mm` ->
`_columns_through`: the columns go to the front, every vector contiguous) plus ONE copy of the result into the panel.
One JSON line: ms per iteration, share of the user's `_mv`, of the result copy, of the native chain."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import xitorch_amd as xa
from xitorch_amd.linalg import symeig
from xitorch_amd.linalg._panel import PanelOperator
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
shape = (2, N, N)


class ALarge(xa.LinearOperator):
    def __init__(self, shape, dtype, device):
        super().__init__(shape, is_hermitian=True, dtype=dtype, device=device)
        self.b = torch.arange(shape[-1], dtype=dtype, device=device).repeat(*shape[:-2], 1)

    def _mv(self, x):
        xb = x * self.b
        xsmall = x * 1e-3
        return xb + torch.roll(xsmall, shifts=1, dims=-1) + torch.roll(xsmall, shifts=-1, dims=-1)

    def _getparamnames(self, prefix=""):
        return [prefix + "b"]


A = ALarge(shape, torch.float64, dev)
niter = 24
for _ in range(2):
    tr = {}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev, X = symeig(A, neig=2, mode="lowest", method="davidson", min_eps=1e-8, max_niter=niter, trace=tr)
        torch.cuda.synchronize(); t_call = time.perf_counter() - t0
# the panel product alone: user ops + the copy into the panel
op = PanelOperator(A, [2], 2, N)
ld = (N + 7) // 8 * 8
Xp = torch.randn(2, 2, ld, dtype=torch.float64, device=dev)
out = torch.empty_like(Xp)


def t_of(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_apply = t_of(lambda: op.apply(Xp, out))
xv = Xp[:, :, :N].transpose(-2, -1)
t_user = t_of(lambda: A.mm(xv))
y = A.mm(xv)
t_copy = t_of(lambda: out[:, :, :N].copy_(y.transpose(-2, -1)))
by = 2 * 2 * N * 8
print(json.dumps({"operator": "ALarge (user _mv: 2 multiplies, 2 rolls, 2 adds)", "N": N, "batch": 2, "neig": 2,
                  "davidson_iterations": tr["niter"], "ms_per_call": t_call * 1e3, "ms_per_iteration": t_call * 1e3 / tr["niter"],
                  "panel_apply_ms": t_apply, "user_mm_ms": t_user, "copy_into_panel_ms": t_copy,
                  "panel_bytes_in_plus_out": 2 * by, "apply_GBps_on_panel_bytes": 2 * by / t_apply / 1e6,
                  "share_of_apply_in_iteration": t_apply / (t_call * 1e3 / tr["niter"]),
                  "eigenvalues_batch0": ev[0].tolist()}))
