"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference); the fixtures it
writes are plain data (inputs are regenerated from closed forms / fixed CPU
seeds, only reference OUTPUTS are stored).  It also cross-checks the oracle
against the reference on every case (bit-for-bit on CPU) and refuses to write
a fixture if they disagree.

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys
import warnings
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import xitorch  # noqa: E402  (the reference)
from xitorch._impls.linalg import symeig as ref_symeig  # noqa: E402
from xitorch._impls.linalg import solve as ref_solve  # noqa: E402
from xitorch._impls.optimize.root import rootsolver as ref_root  # noqa: E402
import xitorch.linalg  # noqa: E402
import xitorch.optimize  # noqa: E402

import oracle.ops as oops  # noqa: E402
import oracle.symeig as osym  # noqa: E402
import oracle.solve as osolve  # noqa: E402
import oracle.rootfinder as oroot  # noqa: E402
from tests import cases  # noqa: E402

torch.set_num_threads(1)   # bit-reproducible reductions
f64 = torch.float64


def exact(a, b, what):
    if not torch.equal(a, b):
        err = (a - b).abs().max().item()
        raise SystemExit("oracle != reference for %s (max abs diff %.3e)" % (what, err))


def close(a, b, what, rel=1e-13):
    """gmres only: both sides call torch.linalg.lstsq every iteration (solve.py:403), whose CPU kernel is not
    bit-reproducible between two calls on identical inputs (the last bit follows the alignment of its work buffers:
    probed, the SAME function gives different last bits in two calls of one process).  The oracle is therefore pinned
    to 1e-13 of the reference there instead of bit for bit; every other method stays torch.equal."""
    err = (a - b).abs().max().item()
    if not err <= rel * max(1.0, a.abs().max().item()):
        raise SystemExit("oracle != reference for %s (max abs diff %.3e)" % (what, err))


class CountingRefOp(xitorch.LinearOperator):
    """Reference-side wrapper that counts applies (to pin iteration counts)."""

    def __init__(self, op):
        super().__init__(shape=op.shape, is_hermitian=op.is_hermitian, dtype=op.dtype, device=op.device)
        self.op = op
        self.n = 0

    def _mv(self, x):
        self.n += 1
        return self.op.mv(x)

    def _mm(self, x):
        self.n += 1
        return self.op.mm(x)

    def _rmv(self, x):
        self.n += 1
        return self.op.rmv(x)

    def _rmm(self, x):
        self.n += 1
        return self.op.rmm(x)

    def _getparamnames(self, prefix=""):
        return []


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in out.items()})


# --------------------------------------------------------------------------- symeig / exacteig (dense)
def gen_exacteig():
    from xitorch._utils.tensor import create_random_square_matrix
    for case in cases.EXACTEIG_CASES:
        A, M = cases.exacteig_inputs(case)
        if not case["batch"]:
            # the input generator restates the reference's benchmark generator: same matrix
            Aref = create_random_square_matrix(case["n"], is_hermitian=True, min_eival=case["minmax"][0],
                                               max_eival=case["minmax"][1], seed=123)
            exact(A, Aref.to(A.dtype), "exacteig %s input matrix" % case["name"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rA = xitorch.LinearOperator.m(A, is_hermitian=True)
            rM = xitorch.LinearOperator.m(M, is_hermitian=True) if M is not None else None
        ev_r, X_r = ref_symeig.exacteig(rA, case["neig"], case["mode"], rM)
        ev_o, X_o = osym.exacteig(oops.DenseOp(A, True), case["neig"], case["mode"],
                                  oops.DenseOp(M, True) if M is not None else None)
        exact(ev_r, ev_o, "exacteig %s evals" % case["name"])
        exact(X_r, X_o, "exacteig %s evecs" % case["name"])
        save("exacteig_" + case["name"], evals=ev_r, X=X_r)


# --------------------------------------------------------------------------- symeig / davidson
def gen_davidson(only=None):
    for case in cases.DAVIDSON_CASES + cases.DAVIDSON_CASES_F32:
        name = case["name"]
        if only and name not in only:
            continue
        mat = cases.davidson_matrix(case)
        Mmat = cases.davidson_M(case)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rop = CountingRefOp(xitorch.LinearOperator.m(mat, is_hermitian=True))
            rM = xitorch.LinearOperator.m(Mmat, is_hermitian=True) if Mmat is not None else None
        kw = dict(max_niter=case.get("max_niter", 1000), min_eps=case["min_eps"], v_init=case.get("v_init", "randn"))
        ev_r, X_r = ref_symeig.davidson(rop, case["neig"], case["mode"], rM, **kw)
        tr = {}
        oop = oops.DenseOp(mat, is_hermitian=True)
        oM = oops.DenseOp(Mmat, is_hermitian=True) if Mmat is not None else None
        ev_o, X_o = osym.davidson(oop, case["neig"], case["mode"], oM, trace=tr, **kw)
        exact(ev_r, ev_o, name + " evals")
        exact(X_r, X_o, name + " evecs")
        assert tr["napply"] == rop.n, (tr["napply"], rop.n)
        if mat.dtype == torch.float32:       # the exact spectrum of the fp32 operator, in double
            ev_all = torch.linalg.eigvalsh(mat.double())
            ev_x = ev_all[..., :case["neig"]] if case["mode"] == "lowest" else ev_all[..., -case["neig"]:]
        else:
            ev_x, _ = ref_symeig.exacteig(xitorch.LinearOperator.m(mat, is_hermitian=True), case["neig"], case["mode"], rM)
        MX = torch.matmul(Mmat, X_r) if Mmat is not None else X_r
        resid = (torch.matmul(mat, X_r) - MX * ev_r.unsqueeze(-2)).abs().max()
        # store evecs only through a sign-free, small summary: |X|^T at a few probe rows
        probe = cases.probe_rows(mat.shape[-1])
        # (round 3) the reference's eigenvectors themselves: the GPU tests compare subspaces,
        # sigma_min(X_ref^T M X) >= 1 - 1e-8; signs are never compared (quirk Q15)
        save("davidson_" + name, evals=ev_r, evals_exact=ev_x, napply=rop.n, niter=tr["niter"],
             max_resid=resid, absX_probe=X_r.abs()[..., probe, :], probe=probe, X=X_r)


# --------------------------------------------------------------------------- solve
def gen_solve():
    for case in cases.SOLVE_CASES:
        name = case["name"]
        A, B, E, M = cases.solve_inputs(case)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if case["op"] == "banded":
                oA = oops.BandedOp(A, is_hermitian=False)
                rA = CountingRefOp(xitorch.LinearOperator.m(oA.fullmatrix(), is_hermitian=False))
                # the oracle must act like the dense matrix for bit-equality: use the same dense op
                oA = oops.DenseOp(oA.fullmatrix(), is_hermitian=False)
            else:
                rA = CountingRefOp(xitorch.LinearOperator.m(A, is_hermitian=case["hermitian"]))
                oA = oops.DenseOp(A, is_hermitian=case["hermitian"])
            rM = xitorch.LinearOperator.m(M, is_hermitian=True) if M is not None else None
            oM = oops.DenseOp(M, is_hermitian=True) if M is not None else None
        kw = dict(case["kwargs"])
        pre = cases.solve_precond(case, A) if case["op"] != "banded" else {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            kw_r = dict(kw, **{k: xitorch.LinearOperator.m(P, is_hermitian=True) for k, P in pre.items()})
            kw_o = dict(kw, **{k: oops.DenseOp(P, is_hermitian=True) for k, P in pre.items()})
            fr = getattr(ref_solve, case["method"])
            fo = getattr(osolve, case["method"])
            X_r = fr(rA, B, E, rM, **kw_r)
            tr = {}
            X_o = fo(oA, B, E, oM, trace=tr, **kw_o)
        (close if case["method"] == "gmres" else exact)(X_r, X_o, name)
        assert oA.n_apply == rA.n, (name, oA.n_apply, rA.n)
        assert bool(tr["converged"]) != bool(case.get("nonconv")), (name, tr)
        X_x = osolve.exactsolve(oA, B, E, oM)
        if case.get("gold_swapped"):
            # the reference's gmres hands back its column-swapped work layout (ncols, *batch, n, 1)
            assert X_r.shape == (B.shape[-1], *B.shape[:-2], B.shape[-2], 1), X_r.shape
        # (round 3) kappa: condition number of the iterated operator, the factor between the residual tolerance
        # and the solution error the GPU tests allow against X
        X_store = X_r
        if case["method"] == "gmres":
            # the reference's own output moves in its last bits from run to run here (lstsq, see close()): the
            # fixture keeps 12 decimals, so that regenerating it gives the same file (barring a value that sits on
            # a rounding boundary); the GPU tests compare at 1e-8 * kappa
            X_store = torch.round(X_r * 1e12) / 1e12
        save("solve_" + name, X=X_store, X_exact=X_x, napply=rA.n, niter=tr["niter"], converged=tr["converged"],
             kappa=cases.solve_kappa(case, A, E, M))


# --------------------------------------------------------------------------- rootfinder
def gen_root():
    for case in cases.ROOT_CASES:
        name = case["name"]
        fcn, y0, params = cases.root_inputs(case)
        kw = dict(case["kwargs"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            nfev_r = [0]

            def cfcn(y, *p):
                nfev_r[0] += 1
                return fcn(y, *p)
            meth = case.get("method", "broyden1")
            y_r = getattr(ref_root, meth)(cfcn, y0, params, **kw)
            tr = {}
            y_o = getattr(oroot, meth)(fcn, y0, params, trace=tr, **kw)
        exact(y_r, y_o, name)
        assert tr["nfev"] == nfev_r[0], (name, tr["nfev"], nfev_r[0])
        out = dict(y=y_r, nfev=nfev_r[0], niter=tr["niter"], fnorm=fcn(y_r, *params).norm())
        if case.get("grad"):
            # full functional with implicit backward through the reference (README.md:16-32 flow)
            ps = [p.clone().requires_grad_() for p in params]
            yr = xitorch.optimize.rootfinder(fcn, y0, params=ps, method="broyden1", **kw)
            g1 = torch.autograd.grad(yr.sum(), ps, create_graph=True)
            g2 = torch.autograd.grad(sum(g.sum() for g in g1), ps)
            for i, (a, b) in enumerate(zip(g1, g2)):
                out["grad%d" % i] = a.detach()
                out["gradgrad%d" % i] = b.detach()
        save("root_" + name, **out)


# --------------------------------------------------------------------------- anderson_acc / gd / adam / newton
def gen_extra():
    """SURVEY 8f.2 methods.  These are short host loops of torch ops, so the product implementation itself
    (xitorch_amd.optimize.extra / native_root.newton) runs on CPU tensors and is compared BIT FOR BIT with the
    reference here; the fixtures then pin the same functions running on the GPU (tests/test_gpu_root.py)."""
    from xitorch._impls.optimize import equilibrium as ref_eq, minimizer as ref_min
    from xitorch_amd.optimize import extra as xextra, native_root as xroot
    for case in cases.EXTRA_CASES:
        name, meth = case["name"], case["method"]
        fcn, y0, params = cases.extra_inputs(case)
        kw = dict(case["kwargs"])
        ref_fn = {"anderson_acc": ref_eq.anderson_acc, "gd": ref_min.gd, "adam": ref_min.adam,
                  "newton": ref_root.newton}[meth]
        own_fn = {"anderson_acc": xextra.anderson_acc, "gd": xextra.gd, "adam": xextra.adam,
                  "newton": xroot.newton}[meth]
        counts = []
        outs = []
        for fn_ in (ref_fn, own_fn):
            n = [0]

            def cfcn(y, *p):
                n[0] += 1
                return fcn(y, *p)
            with warnings.catch_warnings(record=True) as wlist:
                warnings.simplefilter("always")
                outs.append(fn_(cfcn, y0, list(params), **kw))
            # the cases must converge: a convergence warning is a bad case
            assert not [w for w in wlist if "converge" in str(w.message)], (name, [str(w.message) for w in wlist])
            counts.append(n[0])
        exact(outs[0], outs[1], name)
        assert counts[0] == counts[1], (name, counts)
        y = outs[0]
        if meth == "anderson_acc":
            quality = (fcn(y, *params) - y).norm()
        elif meth == "newton":
            quality = fcn(y, *params).norm()
        else:
            quality = fcn(y, *params)[1].norm()
        save("extra_" + name, y=y, nfev=counts[0], quality=quality)


if __name__ == "__main__":
    which = sys.argv[1:] or ["davidson", "exacteig", "solve", "root", "extra"]
    if "exacteig" in which:
        gen_exacteig()
    only = [w.split(":", 1)[1] for w in which if w.startswith("davidson:")]      # e.g. davidson:s1_900_b2_lowest8_f32
    if "davidson" in which or only:
        gen_davidson(only or None)
    if "solve" in which:
        gen_solve()
    if "root" in which:
        gen_root()
    if "extra" in which:
        gen_extra()
    print("golden fixtures written; reference =", xitorch.__version__, "torch", torch.__version__)
