"""oracle — CPU restatement of the reference (xitorch v0.5.1-dev) hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `xitorch_amd/` imports this package;
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may use it, and only as the checker / the timed CPU baseline.

What it is: the reference is pure Python on PyTorch-CPU, so the most faithful
restatement of its algorithm is the same sequence of ATen CPU ops, written
from scratch here, function by function, each citing the reference file:line
it follows.  Because it issues the same ATen kernels in the same order, it
reproduces the reference bit-for-bit on CPU (iteration counts included).

Pinning: `tests/golden/make_golden.py` imports the real reference from
/root/reference (only in the build container), runs both on identical inputs
and commits the reference outputs as fixtures under `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks the oracle against those fixtures on
every run.  The reference itself never travels to the GPU box.

Modules
  oracle.ops        operator stand-ins (dense / banded / callable)  -> linop.py
  oracle.symeig     tallqr, initial guess, block Davidson           -> _impls/linalg/symeig.py
  oracle.solve      cg, bicgstab, gmres + problem set-up            -> _impls/linalg/solve.py
  oracle.rootfinder quasi-Newton driver, Broyden-1 / Broyden-2 / linear-mixing models, Armijo -> _impls/optimize/root/*.py
"""
