"""-m gpu: a small randomised sweep of the block Davidson against the dense eigendecomposition — block widths 1 .. 12,
both ends of the spectrum, nguess > neig, fp64 / fp32, spectra whose wanted pairs converge at very different rates (the
case in which round 3's one-pass orthonormalisation returned duplicated eigenpairs: DESIGN 4).  The full-size version
of this sweep is scripts/solver_fuzz.py (profiles/r03_solver_fuzz.jsonl)."""
import warnings
import pytest
import torch
import xitorch_amd as xa
from xitorch_amd.linalg.native_eig import davidson

pytestmark = pytest.mark.gpu


def _spectrum(kind, N, g):
    i = torch.arange(N, dtype=torch.float64)
    if kind == 0:      # separated ends, dense middle: pairs inside the bulk converge late
        d = 10.0 + 5.0 * i / N
        d[:5] = torch.tensor([1.0, 2.0, 3.0, 4.5, 6.0])
        d[-4:] = torch.tensor([40.0, 45.0, 52.0, 60.0])
    elif kind == 1:    # clusters of (nearly) equal eigenvalues at both ends
        d = 20.0 + 10.0 * torch.rand(N, dtype=torch.float64, generator=g)
        d[:6] = torch.tensor([1.0, 1.0 + 1e-9, 1.0 + 2e-9, 2.0, 2.0, 3.0])
        d[-3:] = torch.tensor([90.0, 90.0, 95.0])
    else:
        d = torch.rand(N, dtype=torch.float64, generator=g) * 50.0
    return d


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_davidson_random_cases_vs_dense(dev, seed):
    g = torch.Generator().manual_seed(1000 + seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))
    for case in range(6):
        N, B, p = ri(60, 700), ri(1, 2), ri(1, 12)
        p = min(p, max(1, N // 5))
        nguess = p + (ri(1, 3) if case % 3 == 0 else 0)
        mode = "lowest" if case % 2 else "uppest"
        dtype = torch.float32 if case == 5 else torch.float64
        d = _spectrum(case % 3, N, g)
        Q, _ = torch.linalg.qr(torch.randn(B, N, N, dtype=torch.float64, generator=g))
        mat = (Q * d) @ Q.transpose(1, 2)
        mat = ((mat + mat.transpose(1, 2)) * 0.5).to(dtype).to(dev)
        A = xa.LinearOperator.m(mat, is_hermitian=True)
        tr = {}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ev, X = davidson(A, p, mode, nguess=nguess, min_eps=1e-8 if dtype == torch.float64 else 2e-3,
                             max_niter=500, trace=tr)
        ref = torch.linalg.eigvalsh(mat.double())
        want = ref[:, :p] if mode == "lowest" else ref[:, -p:]
        tag = (seed, case, N, B, p, nguess, mode, str(dtype), tr["niter"], tr["stop_reason"])
        tol_e, tol_o = (1e-9, 1e-8) if dtype == torch.float64 else (3e-4, 2e-3)
        assert tr["stop_reason"] in ("converged", "full_basis"), tag
        assert (ev.double() - want).abs().max().item() <= tol_e * ref.abs().max().item(), tag
        G = X.double().transpose(1, 2) @ X.double()
        assert (G - torch.eye(p, dtype=torch.float64, device=dev)).abs().max().item() <= tol_o, tag
