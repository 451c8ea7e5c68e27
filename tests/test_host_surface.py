"""not-gpu: host logic — the operator / EditableModule / method plug-in contracts, the C ABI symbol
table, the backward formulas (exact methods on CPU tensors, gradcheck) and the closed-form inputs.
Modelled on the reference's xitorch/_tests/test_linop.py, test_linop_fcns.py, test_optimize.py."""
import ctypes
import os
import warnings
import pytest
import torch
import xitorch_amd as xa
from xitorch_amd import LinearOperator, EditableModule, synthetic, _capi
from xitorch_amd.linalg import symeig, lsymeig, usymeig, svd, solve
from xitorch_amd.optimize import rootfinder
from xitorch_amd.grad import jac, hess
from xitorch_amd._util import get_method, get_attr, set_attr, del_attr
from oracle import rootfinder as oroot
from tests import cases

f64 = torch.float64


# ----------------------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    L = _capi.lib()
    assert isinstance(L, ctypes.CDLL)
    syms = _capi.header_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert _capi.fn("xk_abi_version")() >= 1
    assert _capi.fn("xk_kry_max_partials")() == 64
    assert _capi.fn("xk_dense_mm_workspace_elems")(4, 16, 16, 2, 0) == 0
    # every declared entry point has a ctypes signature (argument conversion is never left to defaults)
    untyped = [s for s in syms if getattr(L, s).argtypes is None]
    assert not untyped, untyped


def test_kernels_refuse_host_tensors_and_methods_dispatch_on_the_device():
    """The C-ABI wrappers (kernels.py) take device memory only — handing them a host tensor raises, there is no detour;
    the METHOD functions dispatch on the operator's device like the reference (r06): host memory -> linalg/host_*.py."""
    A = LinearOperator.m(torch.diag(torch.arange(1.0, 9.0, dtype=f64)), True)
    from xitorch_amd.linalg.native_eig import davidson
    from xitorch_amd.linalg import native_krylov as nk, host_eig, host_krylov
    from xitorch_amd import kernels as K
    with pytest.raises(RuntimeError):
        K.dense_mm(torch.eye(4, dtype=f64), torch.ones(1, 1, 4, dtype=f64))
    with pytest.raises(RuntimeError):
        K.dense_symm(torch.eye(4, dtype=f64), torch.ones(1, 1, 4, dtype=f64))
    with pytest.raises(RuntimeError):
        K.vec_dots([(torch.ones(4, dtype=f64), torch.ones(4, dtype=f64))])
    n0, n1 = host_eig.calls["davidson"], host_krylov.calls["cg"]
    ev, _ = davidson(A, 2, "lowest")
    assert torch.allclose(ev, torch.tensor([1.0, 2.0], dtype=f64), atol=1e-8)
    x = nk.cg(A, torch.ones(8, 1, dtype=f64))
    assert torch.allclose(x.squeeze(-1), 1.0 / torch.arange(1.0, 9.0, dtype=f64), atol=1e-6)
    ev2, _ = symeig(A, 2, method="davidson")
    assert torch.allclose(ev2, ev)
    assert host_eig.calls["davidson"] == n0 + 2 and host_krylov.calls["cg"] == n1 + 1


# ----------------------------------------------------------------------------- operator contract
class _Mv(LinearOperator):
    def __init__(self, mat, is_hermitian=False):
        super().__init__(shape=mat.shape, is_hermitian=is_hermitian, dtype=mat.dtype, device=mat.device)
        self.mat = mat

    def _mv(self, x):
        return torch.matmul(self.mat, x.unsqueeze(-1)).squeeze(-1)

    def _getparamnames(self, prefix=""):
        return [prefix + "mat"]


def test_linop_requires_mv_and_init():
    class NoMv(LinearOperator):
        def __init__(self):
            super().__init__(shape=(2, 2))
    with pytest.raises(RuntimeError):
        NoMv()

    class NoInit(LinearOperator):
        def __init__(self):
            pass

        def _mv(self, x):
            return x
    with pytest.raises(RuntimeError, match="must be executed first"):
        NoInit().mv(torch.ones(2))
    with pytest.raises(RuntimeError):
        _Mv(torch.ones(3))                      # shape needs >= 2 dims
    with pytest.raises(RuntimeError):
        _Mv(torch.ones(2, 3), is_hermitian=True)  # Hermitian must be square


def test_mm_rmm_fallbacks_match_matmul():
    g = torch.Generator().manual_seed(0)
    mat = torch.randn(2, 3, 4, 5, dtype=f64, generator=g)
    op = _Mv(mat)
    for xs in [(5, 2), (3, 5, 2), (2, 3, 5, 6), (7, 2, 3, 5, 1)]:
        x = torch.randn(*xs, dtype=f64, generator=g)
        assert torch.allclose(op.mm(x), torch.matmul(mat, x))
    for xs in [(4, 2), (2, 3, 4, 3)]:
        x = torch.randn(*xs, dtype=f64, generator=g)
        assert torch.allclose(op.rmm(x), torch.matmul(mat.transpose(-2, -1), x))   # rmv via the adjoint trick
    v = torch.randn(2, 3, 5, dtype=f64, generator=g)
    assert torch.allclose(op.rmv(op.mv(v)), torch.matmul(mat.transpose(-2, -1), torch.matmul(mat, v.unsqueeze(-1))).squeeze(-1))
    assert torch.allclose(op.fullmatrix(), mat)
    with pytest.raises(RuntimeError, match="Cannot operate .mm"):
        op.mm(torch.ones(4, 2, dtype=f64))
    with pytest.raises(RuntimeError, match="Cannot operate .mv"):
        op.mv(torch.ones(4, dtype=f64))
    with pytest.raises(RuntimeError, match="Cannot operate .rmv"):
        op.rmv(torch.ones(5, dtype=f64))


def test_matrix_operator_and_algebra():
    g = torch.Generator().manual_seed(1)
    a = torch.randn(3, 4, 4, dtype=f64, generator=g)
    b = torch.randn(4, 4, dtype=f64, generator=g)
    A, Bm = LinearOperator.m(a), LinearOperator.m(b)
    assert not A.is_hermitian
    assert LinearOperator.m(a + a.transpose(-2, -1)).is_hermitian
    with pytest.raises(RuntimeError, match="indicated to be hermitian"):
        LinearOperator.m(a, is_hermitian=True)
    x = torch.randn(3, 4, 2, dtype=f64, generator=g)
    assert torch.allclose((A + Bm).mm(x), torch.matmul(a + b, x))
    assert torch.allclose((A - Bm).mm(x), torch.matmul(a - b, x))
    assert torch.allclose((A * 2.5).mm(x), torch.matmul(a * 2.5, x))
    assert torch.allclose((2 * A).mm(x), torch.matmul(a * 2, x))
    assert torch.allclose(A.matmul(Bm).mm(x), torch.matmul(a @ b, x))
    assert torch.allclose(A.H.mm(x), torch.matmul(a.transpose(-2, -1), x))
    with pytest.raises(TypeError):
        A * "a"
    # composed (non-matrix) operators
    Ai, Bi = _Mv(a), _Mv(b)
    assert torch.allclose((Ai + Bi).mm(x), torch.matmul(a + b, x))
    assert torch.allclose((Ai - Bi).mm(x), torch.matmul(a - b, x))
    assert torch.allclose((Ai * 3).mm(x), torch.matmul(a * 3, x))
    assert torch.allclose(Ai.matmul(Bi).mm(x), torch.matmul(a @ b, x))
    assert "MatrixLinearOperator with shape (3, 4, 4)" in repr(A)
    assert "AddLinearOperator" in repr(Ai + Bi) and "MatmulLinearOperator" in repr(Ai.matmul(Bi))
    with pytest.raises(RuntimeError, match="_rmv"):
        Ai.H.mv(torch.ones(4, dtype=f64))
    Bm.check(warn=False)       # (like the reference, check() stacks inputs: use an unbatched operator)
    xa.BandedLinearOperator(synthetic.banded(1, 40, hb=3)[0]).check(warn=False)


def test_getparamnames_needed_only_with_grad():
    class NoNames(LinearOperator):
        def __init__(self, mat):
            super().__init__(shape=mat.shape, is_hermitian=True, dtype=mat.dtype)
            self.mat = mat

        def _mv(self, x):
            return torch.matmul(self.mat, x.unsqueeze(-1)).squeeze(-1)
    mat = torch.eye(3, dtype=f64) * torch.tensor([1.0, 2.0, 3.0], dtype=f64)
    op = NoNames(mat)
    with torch.no_grad():
        ev, _ = symeig(op, 2)
        assert torch.allclose(ev, torch.tensor([1.0, 2.0], dtype=f64))
    with pytest.raises(RuntimeError, match="_getparamnames"):
        symeig(op, 2)
    with pytest.raises(RuntimeError, match="_getparamnames"):
        solve(op, torch.ones(3, 1, dtype=f64))


# ----------------------------------------------------------------------------- EditableModule / misc
def test_editable_module_and_attr_paths():
    class Mod(EditableModule):
        def __init__(self, a):
            self.a = a
            self.lst = [a * 2, {"k": a * 3}]

        def f(self, x):
            return self.a * x + self.lst[1]["k"]

        def getparamnames(self, methodname, prefix=""):
            if methodname == "f":
                return [prefix + "a", prefix + "lst[1]['k']", prefix + "a"]
            raise KeyError(methodname)
    a = torch.tensor(2.0, dtype=f64)
    m = Mod(a)
    assert len(m.getparams("f")) == 3 and len(m.getuniqueparams("f")) == 2
    new = [torch.tensor(5.0, dtype=f64), torch.tensor(7.0, dtype=f64)]
    m.setuniqueparams("f", *new)
    assert m.a is new[0] and m.lst[1]["k"] is new[1]
    assert get_attr(m, "lst[1]['k']") is new[1]
    set_attr(m, "lst[0]", 1)
    assert m.lst[0] == 1
    del_attr(m, "lst[0]")
    assert m.lst[0] is None and len(m.lst) == 2
    with pytest.raises(KeyError):
        m.getparams("g")
    m2 = Mod(torch.tensor(2.0, dtype=f64))
    m2.assertparams(m2.f, torch.tensor(1.0, dtype=f64))


def test_get_method_contract():
    table = {"a": lambda: 1}
    assert get_method("x", table, "A")() == 1
    fn = lambda: 2
    assert get_method("x", table, fn) is fn
    with pytest.raises(RuntimeError, match="Unknown x method: b"):
        get_method("x", table, "b")
    with pytest.raises(TypeError):
        get_method("x", table, 3)


# ----------------------------------------------------------------------------- functionals (exact methods, CPU)
def test_symeig_exact_modes_and_gradcheck():
    g = torch.Generator().manual_seed(3)
    r = torch.rand(2, 4, 4, dtype=f64, generator=g)

    def f(mat, neig, mode):
        sym = (mat + mat.transpose(-2, -1)) * 0.5
        ev, X = symeig(LinearOperator.m(sym, True), neig, mode)
        return ev, X.abs()
    mat = (r + torch.diag(torch.arange(4, dtype=f64)) * 2).requires_grad_()
    ev, X = f(mat, 2, "lowest")
    evu, _ = f(mat, 2, "uppermost")
    full = torch.linalg.eigvalsh((mat + mat.transpose(-2, -1)) * 0.5)
    assert torch.allclose(ev, full[..., :2]) and torch.allclose(evu, full[..., -2:])
    assert torch.autograd.gradcheck(lambda m: f(m, 2, "lowest"), (mat,))
    assert torch.autograd.gradgradcheck(lambda m: f(m, 2, "lowest")[0], (mat,))
    e1, _ = lsymeig(LinearOperator.m(torch.eye(3, dtype=f64), True), 1)
    e2, _ = usymeig(LinearOperator.m(torch.eye(3, dtype=f64), True), 1)
    assert torch.allclose(e1, e2)


def test_symeig_backward_through_plugin_method_matches_exact():
    # the implicit backward of the iterative path (symeig_torchfcn + shifted solve), exercised on CPU through
    # the documented plug-in hook: method=<callable> (here a thin dense eigh) -> same gradients as exacteig
    g = torch.Generator().manual_seed(4)
    r = torch.rand(5, 5, dtype=f64, generator=g)
    base = ((r + r.T) * 0.5 + torch.diag(torch.arange(5, dtype=f64))).requires_grad_()

    def plugin(A, neig, mode, M=None, **kw):
        ev, X = torch.linalg.eigh(A.fullmatrix())
        return ev[..., :neig], X[..., :neig]

    def loss(mat, method):
        ev, X = symeig(LinearOperator.m((mat + mat.T) * 0.5, True), 2, "lowest", method=method)
        return (ev * torch.tensor([1.0, 2.0], dtype=f64)).sum() + (X.abs() ** 2 * torch.arange(1.0, 6.0, dtype=f64).unsqueeze(-1)).sum()
    g1, = torch.autograd.grad(loss(base, plugin), (base,))
    g2, = torch.autograd.grad(loss(base, "exacteig"), (base,))
    assert torch.allclose(g1, g2, rtol=1e-7, atol=1e-9)


def test_svd_exact():
    g = torch.Generator().manual_seed(5)
    a = torch.rand(2, 5, 3, dtype=f64, generator=g)
    u, s, vh = svd(LinearOperator.m(a), k=3)
    assert torch.allclose(torch.matmul(u * s.unsqueeze(-2), vh), a, atol=1e-10)
    assert torch.allclose(torch.matmul(u.transpose(-2, -1), u), torch.eye(3, dtype=f64).expand(2, 3, 3), atol=1e-10)


def test_solve_exact_AE_AEM_and_gradcheck():
    g = torch.Generator().manual_seed(6)
    n = 4
    a = (torch.rand(2, n, n, dtype=f64, generator=g) * 0.3 + torch.eye(n, dtype=f64)).requires_grad_()
    b = torch.rand(2, n, 3, dtype=f64, generator=g).requires_grad_()
    e = (torch.rand(2, 3, dtype=f64, generator=g) * 0.1).requires_grad_()
    mm = torch.rand(2, n, n, dtype=f64, generator=g)
    m = (0.05 * (mm + mm.transpose(-2, -1)) + torch.eye(n, dtype=f64)).requires_grad_()

    def f(a_, b_, e_, m_):
        msym = (m_ + m_.transpose(-2, -1)) * 0.5
        return solve(LinearOperator.m(a_), b_, e_, LinearOperator.m(msym, True))
    x = f(a, b, e, m)
    msym = (m + m.transpose(-2, -1)) * 0.5
    assert torch.allclose(torch.matmul(a, x) - torch.matmul(msym, x) * e.unsqueeze(-2), b, atol=1e-10)
    assert torch.autograd.gradcheck(f, (a, b, e, m))
    with pytest.warns(UserWarning, match="ignored"):
        solve(LinearOperator.m(a.detach()), b.detach(), None, LinearOperator.m(msym.detach(), True))
    with pytest.raises(RuntimeError, match="square"):
        solve(LinearOperator.m(torch.ones(2, 3, dtype=f64)), torch.ones(2, 1, dtype=f64))


def test_solve_backward_through_plugin_method():
    # solve_torchfcn forward/backward with a user callable (dense solve) on an implicit operator
    g = torch.Generator().manual_seed(7)
    n = 6
    a = (torch.rand(n, n, dtype=f64, generator=g) * 0.2 + torch.eye(n, dtype=f64)).requires_grad_()
    b = torch.rand(n, 2, dtype=f64, generator=g).requires_grad_()

    def dense_method(A, B, E=None, M=None, **kw):
        return torch.linalg.solve(A.fullmatrix(), B)

    class Op(_Mv):
        def _rmv(self, x):
            return torch.matmul(self.mat.transpose(-2, -1), x.unsqueeze(-1)).squeeze(-1)

    def f(a_, b_):
        return solve(Op(a_), b_, method=dense_method, bck_options={"method": dense_method})
    assert torch.autograd.gradcheck(f, (a, b))
    assert torch.autograd.gradgradcheck(f, (a, b))


def test_rootfinder_backward_with_plugin_forward():
    # forward through a plug-in callable (the oracle's Broyden as a user method), backward = product code
    A = torch.tensor([[1.1, 0.4], [0.3, 0.8]], dtype=f64).requires_grad_()
    y0 = torch.zeros((2, 1), dtype=f64)

    def user_method(fcn, y0_, params, **kw):
        return oroot.broyden1(fcn, y0_, params, **kw)

    def f(a):
        return rootfinder(cases.tanh_fcn, y0, params=(a,), method=user_method, f_tol=1e-12, x_tol=1e-12)
    y = f(A)
    assert cases.tanh_fcn(y, A).abs().max().item() < 1e-9
    assert torch.autograd.gradcheck(f, (A,), atol=1e-6)
    assert torch.autograd.gradgradcheck(f, (A,), atol=1e-5)


def test_jac_hess_operators():
    g = torch.Generator().manual_seed(8)
    a = torch.rand(3, 3, dtype=f64, generator=g).requires_grad_()
    y = torch.rand(3, dtype=f64, generator=g).requires_grad_()

    def fcn(y_, a_):
        return torch.tanh(a_ @ y_) + y_ ** 2
    J = jac(fcn, (y, a), idxs=0)
    Jd = torch.autograd.functional.jacobian(lambda yy: fcn(yy, a), y)
    v = torch.rand(2, 3, dtype=f64, generator=g)
    assert torch.allclose(J.mv(v), torch.matmul(Jd, v.unsqueeze(-1)).squeeze(-1))
    assert torch.allclose(J.rmv(v), torch.matmul(Jd.T, v.unsqueeze(-1)).squeeze(-1))
    assert torch.allclose(J.fullmatrix(), Jd)
    H = hess(lambda y_, a_: fcn(y_, a_).sum(), (y, a), idxs=0)
    Hd = torch.autograd.functional.hessian(lambda yy: fcn(yy, a).sum(), y)
    assert torch.allclose(H.fullmatrix(), Hd) and H.is_hermitian
    with pytest.raises(TypeError):
        jac(fcn, (y.detach(), a), idxs=0)


# ----------------------------------------------------------------------------- synthetic inputs
def test_synthetic_operators_have_the_advertised_spectrum():
    for kind in ("S1", "S2", "S3"):
        mat = synthetic.dense_symmetric(2, 96, kind)
        assert torch.equal(mat, mat.transpose(-2, -1))
        ev = torch.linalg.eigvalsh(mat)
        exact = torch.sort(synthetic.spectrum(kind, 96))[0]
        assert torch.allclose(ev, exact.expand(2, 96), atol=1e-10)
    off = synthetic.dense_symmetric(1, 64, "S1", batch_offset=3)
    assert torch.equal(off[0], synthetic.dense_symmetric(4, 64, "S1")[3])
    band = synthetic.banded(2, 50, hb=4)
    assert band.shape == (2, 9, 50) and band[0, 0, 0] == 0 and band[0, 8, 49] == 0


# ----------------------------------------------------------------------------- equilibrium / minimize (next rows)
def test_equilibrium_and_minimize_on_cpu_methods():
    from xitorch_amd.optimize import equilibrium, minimize
    A = torch.tensor([[0.3, 0.1], [0.05, 0.2]], dtype=f64).requires_grad_()
    y0 = torch.zeros((2, 1), dtype=f64)

    def fp(y, a):                      # contraction: y = tanh(a y + 0.1)
        return torch.tanh(a @ y + 0.1)
    y = equilibrium(fp, y0, params=(A,), method="anderson_acc", f_tol=1e-12, x_tol=1e-12)
    assert torch.allclose(y, fp(y, A), atol=1e-10)
    g, = torch.autograd.grad(y.sum(), (A,))
    # implicit-function gradient by dense algebra
    yd = y.detach()
    J = torch.autograd.functional.jacobian(lambda yy: yy - fp(yy, A.detach()), yd).reshape(2, 2)
    lam = torch.linalg.solve(J.T, -torch.ones(2, dtype=f64)).reshape(2, 1)
    Ac = A.detach().clone().requires_grad_()
    gref, = torch.autograd.grad(yd - fp(yd, Ac), (Ac,), grad_outputs=lam)
    assert torch.allclose(g, gref, atol=1e-8)
    y2 = equilibrium(fp, y0, params=(A,), method="linearmixing", alpha=-1.0, f_tol=1e-12, x_tol=1e-12)   # root-finder method
    assert torch.allclose(y2, y, atol=1e-9)

    def quad(y, a):
        return ((y - 1.5) ** 2).sum() + 0.5 * (a * y).sum() ** 2 * 0.0 + (a.sum() * 0.0)
    for method, kw in (("gd", dict(step=0.1, gamma=0.5, maxiter=400, x_rtol=1e-14, f_rtol=1e-16)),
                       ("adam", dict(step=0.05, maxiter=3000, x_rtol=1e-12, f_rtol=1e-16))):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ym = minimize(quad, y0, params=(A,), method=method, **kw)
        assert torch.allclose(ym, torch.full_like(ym, 1.5), atol=1e-4), method
    with pytest.raises(RuntimeError, match="Unknown"):
        minimize(quad, y0, params=(A,), method="nope")


def test_header_is_plain_c_and_usable_from_c(tmp_path):
    # the boundary is a C ABI: the header must compile as C99 and a C program must be able to bind the library
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = _capi.HEADER_PATH
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    src = tmp_path / "bind.c"
    src.write_text(r'''
#include <stdio.h>
#include <dlfcn.h>
#include "xitorch_amd.h"
typedef int (*abi_fn)(void);
typedef long (*ws_fn)(int, int, int, int);
int main(int argc, char** argv) {
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
  abi_fn abi = (abi_fn)dlsym(h, "xk_abi_version");
  ws_fn ws = (ws_fn)dlsym(h, "xk_dense_symm_workspace_elems");
  if (!abi || !ws) return 3;
  /* prototypes from the header and the exported symbols agree in arity/types at least for these two */
  printf("%d %ld\n", abi(), ws(2, 2048, 6, 8));
  (void)argc;
  return 0;
}
''')
    exe = tmp_path / "bind"
    subprocess.check_call([gcc, "-std=gnu99", "-I", os.path.dirname(hdr), str(src), "-o", str(exe), "-ldl"])
    out = subprocess.check_output([str(exe), _capi.LIB_PATH]).decode().split()
    assert int(out[0]) >= 1
    # (NS + NT) slots of P x N per batch member: 2048 / 1024 column slabs + 2048 / 512 row tiles (sized for the
    # 512-row tiles small fp64 launches use), + 16 elements for the run queue of the resident launch (r05)
    assert int(out[1]) == 2 * (2 + 4) * 6 * 2048 + 16


def test_symmetric_storage_promise_follows_the_tensor():
    """the exact-symmetry promise (upper-triangle kernel) and the checked-symmetry status (transposed
    kernel) belong to ONE tensor; swapping `mat` (uselinopparams / setuniqueparams) or writing into it drops them."""
    import xitorch_amd as xa
    g = torch.Generator().manual_seed(3)
    R = torch.rand(2, 6, 6, dtype=torch.float64, generator=g)
    sym = (R + R.transpose(-2, -1)) * 0.5
    A = xa.LinearOperator.m(sym.clone())
    assert A.is_hermitian and A.symmetric_storage and A.hermitian_verified
    other = R.clone()                                     # not symmetric at all
    with A.uselinopparams(other):
        assert not A.symmetric_storage and not A.hermitian_verified
        assert torch.allclose(A.mm(R), torch.matmul(other, R))            # the reference's mat @ x
    assert A.symmetric_storage and A.hermitian_verified    # the original tensor is back
    A.mat.add_(1.0)                                       # in-place write: version counter moves
    assert not A.symmetric_storage
    # a directly constructed operator is taken at its word for is_hermitian, never for the storage
    A2 = xa.MatrixLinearOperator(sym.clone(), True)
    assert A2.is_hermitian and not A2.symmetric_storage and not A2.hermitian_verified
    A2.symmetric_storage = True
    assert A2.symmetric_storage and A2.hermitian_verified
    # allclose-symmetric but not bitwise: checked, not exact
    near = sym.clone()
    near[0, 1, 2] += 1e-12
    A3 = xa.LinearOperator.m(near)
    assert A3.is_hermitian and A3.hermitian_verified and not A3.symmetric_storage


def _run_extra(case, device):
    import xitorch_amd.optimize.extra as xextra
    from xitorch_amd.optimize import native_root as xroot
    from tests import cases as _cases
    fcn, y0, params = _cases.extra_inputs(case)
    fn = {"anderson_acc": xextra.anderson_acc, "gd": xextra.gd, "adam": xextra.adam, "newton": xroot.newton}[case["method"]]
    n = [0]

    def cfcn(y, *p):
        n[0] += 1
        return fcn(y, *p)
    with warnings.catch_warnings():
        warnings.filterwarnings("error", message=".*converge.*")
        y = fn(cfcn, y0.to(device), [p.to(device) for p in params], **case["kwargs"])
    return y.detach(), n[0]


def test_extra_methods_match_reference_goldens_on_cpu():
    """anderson_acc / gd / adam / newton are short host loops of torch ops: on CPU tensors they must
    reproduce the reference's outputs (tests/golden/extra_*.npz, written by make_golden.py after a bit-for-bit
    comparison with xitorch/_impls/optimize/{equilibrium,minimizer}.py and root/rootsolver.py:151-174)."""
    import os
    import numpy as np
    from tests import cases as _cases
    gold_dir = os.path.join(os.path.dirname(__file__), "golden")
    for case in _cases.EXTRA_CASES:
        gold = np.load(os.path.join(gold_dir, "extra_%s.npz" % case["name"]))
        y, nfev = _run_extra(case, "cpu")
        assert nfev == int(gold["nfev"]), (case["name"], nfev, int(gold["nfev"]))
        assert np.abs(y.detach().numpy() - gold["y"]).max() <= 1e-12, case["name"]


def test_davidson_orthonormalisation_pass_policy():
    """The adaptive choice of projection passes (native_eig._Group.current_passes / note_condition), without a device:
    two passes for the first two panels, one while the reported squared condition estimate stays below the limit, two
    for the rest of the run from the first panel above it; fixed counts, wide panels and the non-fused paths ignore the
    estimate."""
    from types import SimpleNamespace
    from xitorch_amd.linalg.native_eig import _Group

    def group(**kw):
        g = SimpleNamespace(adaptive=True, passes_now=2, two_pass_from=None, orth_passes=2, fast=True, opM=None,
                            precond=None, dtype=torch.float64, ONE_PASS_MAX_COND2=_Group.ONE_PASS_MAX_COND2)
        g.__dict__.update(kw)
        return g
    g = group()
    seen = []
    for it, cond2 in enumerate([0.0, 1.0, 30.0, 2e3, 8e3, 3e5, 10.0, 1.0]):
        seen.append(_Group.current_passes(g, 6))          # the panel enqueued before this iteration's status is read
        _Group.note_condition(g, cond2, it)
    assert seen == [2, 2, 1, 1, 1, 1, 2, 2] and g.two_pass_from == 5
    # fp32: the limit is 1e2
    g = group(dtype=torch.float32)
    for it, cond2 in enumerate([0.0, 1.0, 5e2]):
        _Group.note_condition(g, cond2, it)
    assert g.passes_now == 2 and g.two_pass_from == 2
    # NaN counts as "above"
    g = group(passes_now=1)
    _Group.note_condition(g, float("nan"), 4)
    assert g.passes_now == 2
    # wide panels, an overlap operator, a fixed count
    g = group(passes_now=1)
    assert _Group.current_passes(g, 10) == 2
    assert _Group.current_passes(group(passes_now=1, opM=object()), 6) == 2
    assert _Group.current_passes(group(adaptive=False, orth_passes=1), 6) == 1
    assert _Group.current_passes(group(adaptive=False, orth_passes=3), 6) == 3


def test_rayleigh_ritz_solver_limits_by_order_and_precision():
    """Host-side queries of K3g (no device needed): which orders the one-launch-per-step form and the two-stage form of
    r04 serve — both precisions to 1024 (r05: fp64 beyond 768 on the one-launch-per-step form with 16 column slots; the
    two-stage form to 614 in fp64, to 1024 in fp32: its band + bulges must fit 160 KB of LDS) — and that the workspace
    query covers the two-stage blocks
    (torch.linalg.eigh of the whole T in the reference: xitorch/_impls/linalg/symeig.py:174-175)."""
    batch = _capi.fn("xk_small_eigh_big_batch")
    ws = _capi.fn("xk_small_eigh_big_workspace_elems")
    for k in (8, 129, 614, 615, 768, 769, 1024, 1025, 1536):
        assert batch(k, 6, 8) > 0, k
    assert batch(1537, 6, 8) == 0
    for k in (8, 600, 768, 769, 1024, 1536):
        assert batch(k, 6, 4) > 0, k
    assert batch(1537, 6, 4) == 0 and batch(7, 6, 4) == 0
    assert batch(300, 257, 8) == 0 and batch(300, 256, 8) > 0         # at most 256 wanted pairs (r06; 64 before)
    # work copy + hand-over blocks of the one-stage form + the two-stage form's V / T / R / W / Z / reflector blocks
    k, B = 582, 32
    assert ws(B, k, 0) > B * k * k * 3
    assert ws(B, 24, 0) >= B * 24 * 24                                # below order 35 there is no two-stage form
    assert ws(2 * B, k, 0) == 2 * ws(B, k, 0) or ws(2 * B, k, 0) > ws(B, k, 0)


def test_k1s_form_choice_respects_the_descriptor_limit_of_the_8_wave_tiles():
    """Host-side choice of the K1s launch form (kernels.k1s_auto_opts, no device needed): the 8-wave form addresses 2048
    rows through ONE buffer descriptor, so 2048 * lda * 8 bytes must stay below 2 GiB (xk_symm.hip returns
    XK_ERR_UNSUPPORTED beyond) — one exactly symmetric fp64 operator of order 131072 fits in HBM and must keep the
    4-wave form instead of failing."""
    from xitorch_amd import kernels as K
    o = K.k1s_auto_opts(64, 16384, torch.float64, 256)
    assert (o & K.K1S_PERSIST) and (o & K.K1S_WIDE8)                   # the headline's form is unchanged
    for n in (131072, 150016, 188416):
        o = K.k1s_auto_opts(1, n, torch.float64, 256)
        assert (o & K.K1S_PERSIST) and not (o & K.K1S_WIDE8), n
    assert K.k1s_auto_opts(1, 131064, torch.float64, 256) & K.K1S_WIDE8   # just below the limit: 8-wave tiles
    # a padded leading dimension counts, not the order
    assert not (K.k1s_auto_opts(4, 65536, torch.float64, 256, lda=131072) & K.K1S_WIDE8)
    assert K.k1s_auto_opts(4, 65536, torch.float64, 256, lda=65536) & K.K1S_WIDE8
    # fp32 never takes the 8-wave form
    assert not (K.k1s_auto_opts(64, 16384, torch.float32, 256) & K.K1S_WIDE8)


def test_sharded_start_block_is_the_slice_of_the_global_draw():
    """ On a batch-sharded run every rank draws the start block of the WHOLE batch from the
    reference's seed (symeig.py:236-246) and keeps its members: member b of an N-GPU run starts from the vectors
    member b of the one-GPU run starts from."""
    from xitorch_amd.linalg.native_eig import _initial_block
    cpu = torch.device("cpu")
    full = _initial_block("randn", None, [5], 5, 40, 3, torch.float64, cpu, "cpu")            # (5, 3, 40)
    for off, b in ((0, 2), (2, 2), (4, 1)):
        part = _initial_block("randn", None, [b], b, 40, 3, torch.float64, cpu, "cpu", shard=(off, 5))
        assert torch.equal(part, full[off:off + b])
    # unsharded / single-rank arguments leave the draw as it was
    assert torch.equal(_initial_block("randn", None, [5], 5, 40, 3, torch.float64, cpu, "cpu", shard=(0, 5)), full)
    r = _initial_block("rand", None, [2], 2, 40, 3, torch.float64, cpu, "cpu", shard=(3, 5))
    assert torch.equal(r, _initial_block("rand", None, [5], 5, 40, 3, torch.float64, cpu, "cpu")[3:5])
