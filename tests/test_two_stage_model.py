"""CPU check of the index arithmetic behind K3g's two-stage form (csrc/xk_eigh_band.hip): the numpy model in
scripts/two_stage_proto.py — panel QR + two-sided block update to a band of NB sub-diagonals, bulge chasing in the
kernel's pipelined order (sweep s + 1 three steps behind sweep s, the pairs of a tick on disjoint rows), vectors back
through both stages — against numpy.linalg.eigh (the role of torch.linalg.eigh in xitorch/_impls/linalg/symeig.py:174)."""
import os
import sys
import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
import two_stage_proto as proto                          # noqa: E402


@pytest.mark.parametrize("n,NB", [(35, 16), (67, 16), (100, 16), (41, 8)])
def test_two_stage_model_matches_eigh(n, NB):
    rng = np.random.default_rng(n)
    R = rng.standard_normal((n, n))
    A = R + R.T
    Bd, panels = proto.stage1(A, NB)
    i, j = np.indices((n, n))
    assert np.abs(Bd[np.abs(i - j) > NB]).max(initial=0) < 1e-12
    Tt, refl, nsteps = proto.chase(Bd, NB, pipeline=True)          # asserts the disjointness of every tick itself
    Ts, _, _ = proto.chase(Bd, NB, pipeline=False)
    assert np.abs(Tt - Ts).max() < 1e-12
    assert np.abs(Tt[np.abs(i - j) > 1]).max(initial=0) < 1e-11
    lam, Zt = np.linalg.eigh(Tt)
    Y = proto.back(Zt[:, :5], refl, nsteps, panels, n, NB)
    assert np.abs(lam - np.linalg.eigvalsh(A)).max() < 1e-10
    assert np.abs(A @ Y - Y * lam[:5]).max() < 1e-10
    assert np.abs(Y.T @ Y - np.eye(5)).max() < 1e-12
