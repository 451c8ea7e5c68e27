"""Shared definitions of the parity cases: names, sizes, options and INPUT generators.

Inputs come from closed forms (xitorch_amd.synthetic) or from the CPU torch
generator with a fixed seed, always generated on CPU in float64 and then moved
to the device under test, so that the oracle, the reference (when the golden
fixtures were made) and the HIP path see the very same numbers.
Nothing here imports the reference.
"""
import math
import torch
from xitorch_amd import synthetic as syn

f64 = torch.float64


def probe_rows(n):
    return [0, 1, n // 3, n // 2, n - 2, n - 1]


# ------------------------------------------------------------------ eigensolver cases
DAVIDSON_CASES = [
    # the reference's own large-operator test (xitorch/_tests/test_linop_fcns.py:129-176)
    dict(name="alarge1000_lowest", kind="alarge", n=1000, batch=(), neig=2, mode="lowest", min_eps=1e-8),
    dict(name="alarge1000_uppest", kind="alarge", n=1000, batch=(), neig=2, mode="uppest", min_eps=1e-8),
    dict(name="alarge600_b2_lowest", kind="alarge", n=600, batch=(2,), neig=2, mode="lowest", min_eps=1e-8),
    # closed-form dense spectra of the benchmark
    dict(name="s1_512_b2_lowest6", kind="S1", n=512, batch=(2,), neig=6, mode="lowest", min_eps=1e-8),
    dict(name="s2_256_b3_uppest4", kind="S2", n=256, batch=(3,), neig=4, mode="uppest", min_eps=1e-8),
    dict(name="s1_1024_b1_lowest6", kind="S1", n=1024, batch=(1,), neig=6, mode="lowest", min_eps=1e-8),
    # mixed convergence (r03): six separated eigenvalues converge within ~20 iterations, the rest of the wanted pairs
    # sit in the dense part of S1 and take 60-75 — the regime in which restricting the reference's full-basis
    # CholeskyQR to the new block needs a second orthonormalisation pass
    dict(name="s1_900_b2_lowest10", kind="S1", n=900, batch=(2,), neig=10, mode="lowest", min_eps=1e-8),
    dict(name="s1_900_b2_lowest8", kind="S1", n=900, batch=(2,), neig=8, mode="lowest", min_eps=1e-8),
    # config 1 of BASELINE.json: benchmarks_solve.py shape family, N=512, lowest 6
    dict(name="c1_rand512_lowest6", kind="randsym", n=512, batch=(), neig=6, mode="lowest", min_eps=1e-8),
    # generalised problem A x = lam M x (M-orthonormal basis, symeig.py:183-185,216-218).  (v_init="eye" is not
    # pinned: on these operators the reference's own CholeskyQR of the basis fails — its residual blocks become
    # linearly dependent — so there is no reference output to compare with.)
    dict(name="genM_120_b2_lowest3", kind="gen", n=120, batch=(2,), neig=3, mode="lowest", min_eps=1e-8, M=True),
    dict(name="genM_90_b2_uppest2", kind="gen", n=90, batch=(2,), neig=2, mode="uppest", min_eps=1e-8, M=True),
]


# fp32, mixed convergence (r04): the six separated eigenvalues of S1 converge early, the two wanted pairs inside the
# dense part late (43 reference iterations) — the regime of the r03 duplicate-eigenpair defect, in the precision of
# BASELINE configs[4].  (The reference's own fp32 run needs torch.set_num_threads(1), which make_golden.py sets: with
# several MKL threads its torch.inverse stalls in SLASWP at a basis of 152 vectors in this image.)
DAVIDSON_CASES_F32 = [
    dict(name="s1_900_b2_lowest8_f32", kind="S1", n=900, batch=(2,), neig=8, mode="lowest", min_eps=2e-3,
         dtype="float32"),
]


# dense `exacteig` at the shapes of the reference's only benchmark harness (benchmarks/benchmarks_solve.py:37-59:
# symeig(A, neig=10, mode="lowest") with method=None -> exacteig, n in {100, 350, 700}, create_random_square_matrix with
# seed 123), both ends of the spectrum, with and without an overlap operator, and a batched case
EXACTEIG_CASES = [
    dict(name="n100_lowest10", n=100, batch=(), neig=10, mode="lowest", minmax=(-1.0, 1.0)),
    dict(name="n350_lowest10", n=350, batch=(), neig=10, mode="lowest", minmax=(0.0, 1.0)),
    dict(name="n700_lowest10", n=700, batch=(), neig=10, mode="lowest", minmax=(-1.0, 1.0)),
    dict(name="n700_uppest10", n=700, batch=(), neig=10, mode="uppest", minmax=(0.2, 1.0)),
    dict(name="n100_uppest10_M", n=100, batch=(), neig=10, mode="uppest", minmax=(-1.0, 1.0), M=True),
    dict(name="n350_lowest10_M", n=350, batch=(), neig=10, mode="lowest", minmax=(0.5, 1.0), M=True),
    dict(name="n700_lowest10_M", n=700, batch=(), neig=10, mode="lowest", minmax=(-1.0, 1.0), M=True),
    dict(name="n120_b3_lowest4", n=120, batch=(3,), neig=4, mode="lowest", minmax=(-1.0, 1.0)),
]


def exacteig_inputs(case):
    """(A, M) of an EXACTEIG case: the benchmark's matrix (batched cases: one seed per member), M SPD or None"""
    n, batch = case["n"], tuple(case["batch"])
    nb = 1
    for d in batch:
        nb *= d
    mats = [random_symmetric(n, case["minmax"][0], case["minmax"][1], 123 + b) for b in range(nb)]
    A = torch.stack(mats).reshape(*batch, n, n)
    M = None
    if case.get("M"):
        g = torch.Generator().manual_seed(91 + n)
        R2 = torch.rand((*batch, n, n), dtype=f64, generator=g)
        M = 0.02 * (R2 + R2.transpose(-2, -1)) + torch.eye(n, dtype=f64)
    return A, M


from xitorch_amd.synthetic import random_symmetric  # noqa: E402,F401  (one definition: bench.py uses it too)


def davidson_matrix(case):
    if case.get("dtype") == "float32":
        return davidson_matrix({k: v for k, v in case.items() if k != "dtype"}).to(torch.float32)
    n, kind, batch = case["n"], case["kind"], tuple(case["batch"])
    if kind == "alarge":
        nb = 1
        for d in batch:
            nb *= d
        mats = []
        for b in range(nb):
            m = torch.diag(torch.arange(n, dtype=f64) * (1.0 + 0.1 * b))
            eye = torch.eye(n, dtype=f64)
            m = m + 1e-3 * (torch.roll(eye, 1, 0) + torch.roll(eye, -1, 0))
            mats.append(m)
        return torch.stack(mats).reshape(*batch, n, n)
    if kind in ("S1", "S2", "S3"):
        return syn.dense_symmetric(batch[0], n, kind)
    if kind == "randsym":
        return random_symmetric(n, -1.0, 1.0, 123)
    if kind == "gen":
        g = torch.Generator().manual_seed(3 + n)
        R = torch.rand((*batch, n, n), dtype=f64, generator=g)
        return (R + R.transpose(-2, -1)) * 0.5 + torch.diag(torch.arange(n, dtype=f64) * 0.5)
    raise ValueError(kind)


def davidson_M(case):
    """SPD overlap matrix of the generalised cases (None otherwise)."""
    if not case.get("M"):
        return None
    n, batch = case["n"], tuple(case["batch"])
    g = torch.Generator().manual_seed(77 + n)
    R2 = torch.rand((*batch, n, n), dtype=f64, generator=g)
    return 0.02 * (R2 + R2.transpose(-2, -1)) + torch.eye(n, dtype=f64)


# ------------------------------------------------------------------ linear-solver cases
SOLVE_CASES = [
    dict(name="cg_sym100", method="cg", op="dense", hermitian=True, n=100, batch=(2,), ncols=3,
         kwargs=dict(rtol=1e-8, posdef=True)),
    dict(name="cg_nonsym60_normal_eq", method="cg", op="dense", hermitian=False, n=60, batch=(2,), ncols=2,
         kwargs=dict(rtol=1e-8, posdef=True)),
    dict(name="bicgstab_nonsym100", method="bicgstab", op="dense", hermitian=False, n=100, batch=(2,), ncols=3,
         kwargs=dict(rtol=1e-8, posdef=True)),
    dict(name="bicgstab_banded1024", method="bicgstab", op="banded", hermitian=False, n=1024, batch=(2,),
         ncols=1, kwargs=dict(rtol=1e-10, atol=1e-12, posdef=True)),
    dict(name="gmres_nonsym60", method="gmres", op="dense", hermitian=False, n=60, batch=(2,), ncols=2,
         kwargs=dict(rtol=1e-8, posdef=True, max_niter=60)),
    # gmres with a shift: the reference handles one column only (its column-swapped layout breaks for more) and
    # returns that layout (ncols, *batch, n, 1) without undoing the swap (solve.py:349-432): `gold_swapped`
    dict(name="gmres_nonsym_AE", method="gmres", op="dense", hermitian=False, n=50, batch=(2,), ncols=1, E=True,
         gold_swapped=True, kwargs=dict(rtol=1e-8, posdef=True, max_niter=50)),
    dict(name="gmres_nonsym_AEM", method="gmres", op="dense", hermitian=False, n=50, batch=(2,), ncols=1, E=True,
         M=True, gold_swapped=True, kwargs=dict(rtol=1e-8, posdef=True, max_niter=50)),
    # not converging within max_niter: ConvergenceWarning + the best-residual iterate is returned (:417-432)
    dict(name="gmres_nonconv60", method="gmres", op="dense", hermitian=False, n=60, batch=(2,), ncols=2,
         nonconv=True, kwargs=dict(rtol=1e-12, posdef=True, max_niter=6)),
    dict(name="cg_sym_AEM", method="cg", op="dense", hermitian=True, n=80, batch=(2,), ncols=3, E=True, M=True,
         kwargs=dict(rtol=1e-8, posdef=True)),
    dict(name="bicgstab_nonsym_AE", method="bicgstab", op="dense", hermitian=False, n=80, batch=(2,), ncols=3,
         E=True, kwargs=dict(rtol=1e-8, posdef=True)),
    # preconditioned loops (solve.py:73,121-122,136,170 and :196-197,247-249,277,283): Jacobi preconditioner of a
    # matrix with a spread diagonal
    dict(name="cg_sym120_jacobi", method="cg", op="dense", hermitian=True, n=120, batch=(2,), ncols=2, spread=True,
         precond=("precond",), kwargs=dict(rtol=1e-10, posdef=True)),
    dict(name="bicgstab_nonsym120_jacobi_r", method="bicgstab", op="dense", hermitian=False, n=120, batch=(2,),
         ncols=2, spread=True, precond=("precond_r",), kwargs=dict(rtol=1e-10, posdef=True)),
    dict(name="bicgstab_nonsym120_jacobi_l", method="bicgstab", op="dense", hermitian=False, n=120, batch=(2,),
         ncols=2, spread=True, precond=("precond_l",), kwargs=dict(rtol=1e-10, posdef=True)),
    # complex128 operators: the reference's own solver tests run cg / bicgstab on complex Hermitian matrices, with and
    # without E, M (xitorch/_tests/test_linop_fcns.py:474-524, 631-676); conjugated inner products (solve.py:441-445)
    dict(name="cg_herm100_c128", method="cg", op="dense", hermitian=True, n=100, batch=(2,), ncols=3, cplx=True,
         kwargs=dict(rtol=1e-8, posdef=True)),
    dict(name="bicgstab_herm100_c128", method="bicgstab", op="dense", hermitian=True, n=100, batch=(2,), ncols=3,
         cplx=True, kwargs=dict(rtol=1e-8, posdef=True)),
    dict(name="bicgstab_nonherm80_c128", method="bicgstab", op="dense", hermitian=False, n=80, batch=(2,), ncols=2,
         cplx=True, kwargs=dict(rtol=1e-9, posdef=True)),
    dict(name="cg_nonsym60_normal_eq_c128", method="cg", op="dense", hermitian=False, n=60, batch=(2,), ncols=2,
         cplx=True, kwargs=dict(rtol=1e-8, posdef=True)),
    dict(name="cg_herm_AEM_c128", method="cg", op="dense", hermitian=True, n=80, batch=(2,), ncols=3, E=True, M=True,
         cplx=True, kwargs=dict(rtol=1e-8, posdef=True)),
    dict(name="bicgstab_herm_AEM_c128", method="bicgstab", op="dense", hermitian=True, n=80, batch=(2,), ncols=3,
         E=True, M=True, cplx=True, kwargs=dict(rtol=1e-8, posdef=True)),
]


def solve_inputs(case):
    n, batch, nc = case["n"], tuple(case["batch"]), case["ncols"]
    g = torch.Generator().manual_seed(12345)
    if case["op"] == "banded":
        A = syn.banded(batch[0], n, hb=63)
        xs = syn.banded_rhs_solution(batch[0], n)
        B = syn.banded_apply_reference(A, xs)
    elif case.get("cplx"):
        c128 = torch.complex128
        crand = lambda *shape: torch.complex(torch.rand(shape, dtype=f64, generator=g),
                                             torch.rand(shape, dtype=f64, generator=g))
        A = 0.1 * crand(*batch, n, n) + torch.eye(n, dtype=c128)
        if case["hermitian"]:
            A = (A + A.transpose(-2, -1).conj()) * 0.5
        B = crand(*batch, n, nc) + 0.1
        E = M = None
        if case.get("E"):
            E = crand(*batch, nc) * 0.1
        if case.get("M"):
            M = crand(*batch, n, n) * 0.05 + torch.eye(n, dtype=c128) * 0.5
            M = (M + M.transpose(-2, -1).conj()) * 0.5
        return A, B, E, M
    else:
        R = torch.rand((*batch, n, n), dtype=f64, generator=g)
        A = 0.1 * R + (torch.diag(torch.linspace(1.0, 50.0, n, dtype=f64)) if case.get("spread")
                       else torch.eye(n, dtype=f64))
        if case["hermitian"]:
            A = (A + A.transpose(-2, -1)) * 0.5
        B = torch.rand((*batch, n, nc), dtype=f64, generator=g)
    E = M = None
    if case.get("E"):
        E = torch.rand((*batch, nc), dtype=f64, generator=g) * 0.3
    if case.get("M"):
        R2 = torch.rand((*batch, n, n), dtype=f64, generator=g)
        M = 0.05 * (R2 + R2.transpose(-2, -1)) * 0.5 + torch.eye(n, dtype=f64)
    return A, B, E, M


def solve_kappa(case, A, E, M):
    """2-norm condition number of the operator the Krylov loop works on — A - E_c M per column (worst over batch
    and columns), its normal-equation form A^H A when the loop needs a Hermitian operator and A is not
    (solve.py:607-612,637-643) — the factor between a residual tolerance and the solution error."""
    if case["op"] == "banded":
        from oracle import ops as oops
        A = oops.BandedOp(A).fullmatrix()
    n = A.shape[-1]
    worst = 0.0
    ncols = E.shape[-1] if E is not None else 1
    for c in range(ncols):
        op = A
        if E is not None:
            Mm = M if M is not None else torch.eye(n, dtype=A.dtype)
            op = A - E[..., c].unsqueeze(-1).unsqueeze(-1) * Mm
        if case["method"] == "cg" and not case["hermitian"]:
            op = op.transpose(-2, -1).conj() @ op
        worst = max(worst, float(torch.linalg.cond(op).max()))
    return worst


def solve_precond(case, A):
    """kwarg name -> dense matrix diag(A)^-1 for the cases that run with a (Jacobi) preconditioner."""
    names = case.get("precond", ())
    if not names:
        return {}
    P = torch.diag_embed(1.0 / A.diagonal(dim1=-2, dim2=-1))
    return {k: P for k in names}


# ------------------------------------------------------------------ root-finder cases
def tanh_fcn(y, A):
    # README.md:16-20 example: f(y) = tanh(A y + 0.1) + y/2
    return torch.tanh(A @ y + 0.1) + y / 2.0


def tanh_fcn_batched(y, A):
    # config 4: per-batch dense A_b, y: (B, N)
    return torch.tanh(torch.einsum("bij,bj->bi", A, y) + 0.1) + y / 2.0


def ctanh_fcn_batched(y, A):
    # complex variant: f(y) = tanh(A_b y + 0.1 + 0.05i) + y/2 with complex A_b, y
    return torch.tanh(torch.einsum("bij,bj->bi", A, y) + (0.1 + 0.05j)) + y / 2.0


ROOT_CASES = [
    dict(name="readme2", kind="readme", grad=True, kwargs=dict()),
    dict(name="tanh_b4_n64", kind="tanh", nbatch=4, n=64, kwargs=dict(alpha=-1.0, max_rank=None, f_tol=1e-8)),
    dict(name="tanh_b3_n96_rank8", kind="tanh", nbatch=3, n=96, kwargs=dict(alpha=-1.0, max_rank=8, f_tol=1e-8)),
    # the other quasi-Newton models of the reference (rootsolver.py:209-256, _jacobian.py:120-154)
    dict(name="tanh_b3_n64_broyden2", kind="tanh", nbatch=3, n=64, method="broyden2",
         kwargs=dict(alpha=-1.0, max_rank=None, f_tol=1e-8)),
    dict(name="tanh_b2_n48_linearmixing", kind="tanh", nbatch=2, n=48, method="linearmixing",
         kwargs=dict(alpha=-1.0, f_tol=1e-8, maxiter=400)),
    # complex unknowns: solved as [Re; Im] of twice the length (rootsolver.py:52-73; the reference tests run
    # rootfinder on complex128, _tests/test_optimize.py:118-155, 312-344)
    dict(name="ctanh_b3_n32_c128", kind="ctanh", nbatch=3, n=32, kwargs=dict(alpha=-1.0, max_rank=None, f_tol=1e-9)),
]


def root_inputs(case):
    if case["kind"] == "readme":
        A = torch.tensor([[1.1, 0.4], [0.3, 0.8]], dtype=f64)
        return tanh_fcn, torch.zeros((2, 1), dtype=f64), (A,)
    A = syn.root_matrix(case["nbatch"], case["n"]) * 2.0   # eigenvalues in (0, 1]
    y0 = torch.zeros((case["nbatch"], case["n"]), dtype=f64)
    if case["kind"] == "ctanh":
        Ac = torch.complex(A, 0.3 * A.flip(-1))
        return ctanh_fcn_batched, torch.complex(y0, y0), (Ac,)
    return tanh_fcn_batched, y0, (A,)


# ------------------------------------------------------------------ "next" optimiser methods
def fixed_point_fcn(y, A):
    # equilibrium problem y = f(y): a contraction built on the config-4 matrices
    return torch.tanh(torch.einsum("bij,bj->bi", A, y) * 0.4 + 0.1)


def quartic_objective(y, A):
    # minimisation problem for gd / adam: f(y) = sum_b [ 1/2 y^T A_b y + 1/4 |y|^4 - <c, y> ], returns (f, df/dy)
    c = torch.linspace(-1.0, 1.0, y.shape[-1], dtype=y.dtype, device=y.device)
    Ay = torch.einsum("bij,bj->bi", A, y)
    f = 0.5 * (y * Ay).sum() + 0.25 * (y ** 4).sum() - (c * y).sum()
    g = 0.5 * (Ay + torch.einsum("bji,bj->bi", A, y)) + y ** 3 - c
    return f, g


EXTRA_CASES = [
    # reference: xitorch/_impls/optimize/equilibrium.py:9-134
    dict(name="anderson_b3_n40", method="anderson_acc", nbatch=3, n=40,
         kwargs=dict(msize=5, beta=1.0, lmbda=1e-4, f_tol=1e-10, x_tol=1e-10, maxiter=200)),
    dict(name="anderson_b2_n24_damped", method="anderson_acc", nbatch=2, n=24,
         kwargs=dict(msize=3, beta=0.7, lmbda=1e-6, f_tol=1e-9, x_tol=1e-9, maxiter=300)),
    # reference: xitorch/_impls/optimize/minimizer.py:5-147
    dict(name="gd_b2_n32", method="gd", nbatch=2, n=32,
         kwargs=dict(step=5e-2, gamma=0.8, maxiter=600, f_rtol=1e-12, x_rtol=1e-10)),
    dict(name="adam_b2_n32", method="adam", nbatch=2, n=32,
         kwargs=dict(step=5e-2, beta1=0.9, beta2=0.99, maxiter=500, f_rtol=1e-12, x_rtol=1e-9)),
    # reference: rootsolver.py:151-174 + NewtonJacobian (_jacobian.py:27-49)
    dict(name="newton_b3_n20", method="newton", nbatch=3, n=20, kwargs=dict(f_tol=1e-10, x_tol=1e-10, maxiter=50)),
]


def extra_inputs(case):
    A = syn.root_matrix(case["nbatch"], case["n"]) * 2.0          # symmetric, eigenvalues in (0, 1]
    y0 = torch.zeros((case["nbatch"], case["n"]), dtype=f64)
    if case["method"] == "anderson_acc":
        return fixed_point_fcn, y0, (A,)
    if case["method"] in ("gd", "adam"):
        return quartic_objective, y0 + 0.1, (A,)
    return tanh_fcn_batched, y0, (A,)
