"""numpy model of K3g's two-stage tridiagonalisation (dense -> band of NB sub-diagonals by block reflectors, band ->
tridiagonal by bulge chasing with the sweep pipeline of the kernel, eigenvectors back through both): pins the index
arithmetic of csrc/xk_eigh_band.hip before any HIP is involved.  `python scripts/two_stage_proto.py` checks it against
numpy.linalg.eigh for a few orders."""
import numpy as np


def house(x):
    """v (v[0] = 1), tau, beta with (I - tau v v^T) x = beta e1 (LAPACK dlarfg without rescaling)"""
    x0 = x[0]
    sigma = float(np.dot(x[1:], x[1:]))
    v = x.copy()
    v[0] = 1.0
    if sigma == 0.0:
        return v * 0 + np.eye(len(x))[0], 0.0, x0
    beta = -np.copysign(np.sqrt(x0 * x0 + sigma), x0)
    tau = (beta - x0) / beta
    v[1:] = x[1:] / (x0 - beta)
    return v, tau, beta


def stage1(A, NB):
    """A = Q1 Bd Q1^T, Bd with NB sub-diagonals.  Returns Bd (full storage) and the panels (r0, V, T)."""
    A = A.copy()
    n = A.shape[0]
    panels = []
    j = 0
    while n - (j + 1) * NB >= 2:
        c0, r0 = j * NB, (j + 1) * NB
        m = n - r0
        P = A[r0:, c0:c0 + NB].copy()
        nref = min(NB, m - 1)
        V = np.zeros((m, NB))
        tau = np.zeros(NB)
        for c in range(nref):
            v, t, beta = house(P[c:, c])
            V[c:, c] = v
            tau[c] = t
            w = v @ P[c:, c + 1:]
            P[c:, c + 1:] -= t * np.outer(v, w)
            P[c, c] = beta
            P[c + 1:, c] = 0
        T = np.zeros((NB, NB))
        for c in range(NB):
            T[c, c] = tau[c]
            if c:
                T[:c, c] = -tau[c] * (T[:c, :c] @ (V[:, :c].T @ V[:, c]))
        A22 = A[r0:, r0:]
        W = A22 @ V
        G = V.T @ W
        M2 = T.T @ G @ T
        Z = W @ T - 0.5 * V @ M2
        A[r0:, r0:] = A22 - V @ Z.T - Z @ V.T
        R = np.triu(P[:NB, :]) if m >= NB else np.vstack([np.triu(P), np.zeros((0, NB))])
        A[r0:, c0:c0 + NB] = 0
        A[r0:r0 + R.shape[0], c0:c0 + NB] = R
        A[c0:c0 + NB, r0:] = A[r0:, c0:c0 + NB].T
        panels.append((r0, V, T))
        j += 1
    return A, panels


def chase(Bd, NB, pipeline=True):
    """band (NB sub-diagonals, full storage) -> tridiagonal; reflectors (s, t) -> (start row, v, tau).  With `pipeline`
    the steps run in the kernel's tick order (tick = 3 s + t, all pairs of a tick 'at once')."""
    A = Bd.copy()
    n = A.shape[0]
    refl = {}

    def nsteps(s):
        return (n - 3 - s) // NB + 1 if s <= n - 3 else 0

    def step(s, t):
        lo = s + 1 + t * NB
        hi = min(s + (t + 1) * NB, n - 1)
        I = slice(lo, hi + 1)
        col = s if t == 0 else s + 1 + (t - 1) * NB
        v, tau, beta = house(A[I, col].copy())
        refl[(s, t)] = (lo, v, tau)
        if tau == 0.0:
            return
        H = np.eye(hi - lo + 1) - tau * np.outer(v, v)
        # left block: columns [col .. lo - 1]
        A[I, col:lo] = H @ A[I, col:lo]
        A[col:lo, I] = A[I, col:lo].T
        A[I, I] = H @ A[I, I] @ H
        hi2 = min(hi + NB, n - 1)
        if hi2 > hi:
            J = slice(hi + 1, hi2 + 1)
            A[J, I] = A[J, I] @ H
            A[I, J] = A[J, I].T

    if not pipeline:
        for s in range(n - 2):
            for t in range(nsteps(s)):
                step(s, t)
    else:
        maxtick = max([3 * s + nsteps(s) for s in range(max(n - 2, 0))] + [0])
        for tick in range(maxtick):
            pairs = [(s, tick - 3 * s) for s in range(n - 2) if 0 <= tick - 3 * s < nsteps(s)]
            # the pairs of one tick must touch disjoint ROW ranges of the lower triangle (rows of windows t and t + 1)
            ranges = sorted((s + 1 + t * NB, min(s + (t + 2) * NB, n - 1)) for s, t in pairs)
            for a, b in zip(ranges, ranges[1:]):
                assert a[1] < b[0], (tick, ranges)
            for s, t in pairs:
                step(s, t)
    return A, refl, nsteps


def back(z, refl, nsteps, panels, n, NB):
    y = z.copy()
    for s in range(n - 3, -1, -1):
        for t in range(nsteps(s)):
            lo, v, tau = refl[(s, t)]
            if tau != 0.0:
                seg = y[lo:lo + len(v)]
                seg -= tau * np.outer(v, v @ seg)
    for r0, V, T in reversed(panels):
        seg = y[r0:]
        seg -= V @ (T @ (V.T @ seg))
    return y


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n, NB in [(40, 8), (67, 16), (130, 16), (97, 8), (33, 16), (18, 16), (35, 16)]:
        R = rng.standard_normal((n, n))
        A = R + R.T
        Bd, panels = stage1(A, NB)
        i, j = np.indices((n, n))
        assert np.abs(Bd[np.abs(i - j) > NB]).max(initial=0) < 1e-12, "not a band"
        assert np.abs(np.linalg.eigvalsh(Bd) - np.linalg.eigvalsh(A)).max() < 1e-10
        Tt, refl, nsteps = chase(Bd, NB, pipeline=True)
        Ts, _, _ = chase(Bd, NB, pipeline=False)
        assert np.abs(Tt - Ts).max() < 1e-12, "pipeline order changes the result"
        assert np.abs(Tt[np.abs(i - j) > 1]).max(initial=0) < 1e-11, "not tridiagonal"
        lam, Zt = np.linalg.eigh(Tt)
        Y = back(Zt[:, :6], refl, nsteps, panels, n, NB)
        lam_ref = np.linalg.eigvalsh(A)
        assert np.abs(lam - lam_ref).max() < 1e-10
        res = np.abs(A @ Y - Y * lam[:6]).max()
        assert res < 1e-10, res
        print("n=%d NB=%d ok: panels=%d reflectors=%d residual=%.1e" % (n, NB, len(panels), len(refl), res))
