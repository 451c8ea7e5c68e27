"""Closed-form (RNG-free) synthetic inputs for the benchmark configs.

Both boxes — the CPU build container and the GPU box — generate bit-comparable
data from these formulas, so no input fixtures have to be stored.  Pure torch;
works on any device.  The 137 GB config-2 operator is built batch member by
batch member so temporaries stay at one N x N matrix.
"""
import math
import torch

__all__ = ["spectrum", "householder_w", "dense_symmetric", "start_block", "banded",
           "banded_rhs_solution", "banded_apply_reference", "root_matrix"]


def spectrum(kind, n, dtype=torch.float64, device="cpu"):
    """Exact eigenvalues D_i, i = 0..n-1, of the synthetic dense operators.

    S1 "clustered": 6 isolated low eigenvalues 1..6 then a band [50, 100]   (primary, eigpairs/s)
    S1:m   the same with m isolated low eigenvalues 1..m ("S1:16": a 16-column eigen-block, configs[4]'s MFMA panel)
    S2: sqrt(i+1)                                                           (stress)
    S3: i + 0.5 sin(i)                                                      (slow convergence)
    """
    i = torch.arange(n, dtype=dtype, device=device)
    kind = kind.upper()
    if kind == "S1" or kind.startswith("S1:"):
        m = 6 if kind == "S1" else int(kind[3:])
        d = 50.0 + 50.0 * (i - float(m)) / max(n - m - 1, 1)
        return torch.where(i < m, i + 1.0, d)
    if kind == "S2":
        return torch.sqrt(i + 1.0)
    if kind == "S3":
        return i + 0.5 * torch.sin(i)
    raise ValueError("unknown spectrum %s" % kind)


def householder_w(b, n, dtype=torch.float64, device="cpu"):
    i = torch.arange(n, dtype=dtype, device=device)
    return torch.sin(0.37 * (i + 1.0) + 0.11 * b) + 1.5


def dense_symmetric(nbatch, n, kind="S1", dtype=torch.float64, device="cpu", out=None, batch_offset=0,
                    scale=1.0):
    """A_b = H_b diag(D) H_b with H_b = I - 2 w_b w_b^T / |w_b|^2  (dense, exactly symmetric).

    Elementwise: A = diag(D) - beta (w u^T + u w^T) + gamma w w^T,  u = D*w,
    beta = 2/|w|^2, gamma = 4 (w.u)/|w|^4.  Eigenvalues are exactly `spectrum(kind, n)`.
    `batch_offset` shifts b (rank r of a sharded run generates members [offset, offset+nbatch)).
    Generated in float64 and cast, so the f32 operator is the rounding of the f64 one.
    """
    if out is None:
        out = torch.empty((nbatch, n, n), dtype=dtype, device=device)
    D = spectrum(kind, n, torch.float64, device)
    for b in range(nbatch):
        w = householder_w(b + batch_offset, n, torch.float64, device)
        u = D * w
        ww = torch.dot(w, w)
        beta = 2.0 / ww
        gamma = 4.0 * torch.dot(w, u) / (ww * ww)
        S = torch.outer(w, u)
        S = S + S.transpose(0, 1)              # exactly symmetric
        Ab = torch.outer(w, w)
        Ab.mul_(gamma).sub_(S.mul_(beta))      # gamma*w w^T - beta*(w u^T + u w^T)
        del S
        Ab.diagonal().add_(D)
        if scale != 1.0:
            Ab.mul_(scale)
        out[b].copy_(Ab)
        del Ab
    return out


def start_block(nbatch, n, p, dtype=torch.float64, device="cpu"):
    """Deterministic start block V0[b,i,j] = cos(0.1 (i+1)(j+1) + 0.05 b), NOT yet orthonormal.

    Returned PANEL-MAJOR (nbatch, p, n); `.transpose(-2,-1)` gives the reference's (n, p) view.
    """
    i = torch.arange(n, dtype=torch.float64, device=device)
    j = torch.arange(p, dtype=torch.float64, device=device)
    b = torch.arange(nbatch, dtype=torch.float64, device=device)
    V = torch.cos(0.1 * (i[None, None, :] + 1.0) * (j[None, :, None] + 1.0) + 0.05 * b[:, None, None])
    return V.to(dtype)


def banded(nbatch, n, hb=63, dtype=torch.float64, device="cpu", batch_offset=0):
    """Non-symmetric banded operator in DIA storage, band[b, d, i] = A_b[i, i + d - hb].

    band[b,d,i] = 0.05 cos(0.013 i + 0.7 d + 0.3 b) (d != hb);  band[b,hb,i] = 2 + 0.5 sin(0.001 i + b).
    Entries that fall outside the matrix are zeroed.
    """
    i = torch.arange(n, dtype=torch.float64, device=device)
    d = torch.arange(2 * hb + 1, dtype=torch.float64, device=device)
    b = torch.arange(nbatch, dtype=torch.float64, device=device) + batch_offset
    band = 0.05 * torch.cos(0.013 * i[None, None, :] + 0.7 * d[None, :, None] + 0.3 * b[:, None, None])
    band[:, hb, :] = 2.0 + 0.5 * torch.sin(0.001 * i[None, :] + b[:, None])
    col = i[None, :] + (d[:, None] - hb)
    band = band * ((col >= 0) & (col < n)).to(band.dtype)[None]
    return band.to(dtype)


def banded_rhs_solution(nbatch, n, dtype=torch.float64, device="cpu", batch_offset=0):
    """x*[b,i] = sin(0.01 i + b), shape (nbatch, n, 1)."""
    i = torch.arange(n, dtype=torch.float64, device=device)
    b = torch.arange(nbatch, dtype=torch.float64, device=device) + batch_offset
    return torch.sin(0.01 * i[None, :] + b[:, None]).unsqueeze(-1).to(dtype)


def banded_apply_reference(band, x):
    """Plain-torch banded apply used only to manufacture right-hand sides B = A x* (any device)."""
    nd, n = band.shape[-2:]
    hb = nd // 2
    y = torch.zeros_like(x)
    for d in range(nd):
        off = d - hb
        lo, hi = max(0, -off), min(n, n - off)
        if hi > lo:
            y[..., lo:hi, :] += band[..., d, lo:hi].unsqueeze(-1) * x[..., lo + off:hi + off, :]
    return y


def root_matrix(nbatch, n, dtype=torch.float64, device="cpu", batch_offset=0):
    """Per-batch dense matrix of the root problem f(y) = tanh(A_b y + 0.1) + y/2  (config 4):
    A_b = (0.5 / sqrt(n)) * dense_symmetric(kind="S2"), i.e. eigenvalues in (0, 0.5]."""
    return dense_symmetric(nbatch, n, "S2", dtype, device, batch_offset=batch_offset,
                           scale=0.5 / math.sqrt(n))


def random_symmetric(n, min_eival, max_eival, seed):
    """One dense symmetric fp64 matrix with a prescribed linspace spectrum in a seeded random orthogonal basis — the
    shape of the reference's test / benchmark matrices (xitorch/_utils/tensor.py:46-76, hermitian branch;
    benchmarks/benchmarks_solve.py:37-59).  BASELINE configs[0] in bench.py and the golden cases use it."""
    ev = torch.linspace(min_eival, max_eival, n, dtype=torch.float64)
    g = torch.Generator().manual_seed(seed)
    q, _ = torch.linalg.qr(torch.randn((n, n), dtype=torch.float64, generator=g))
    mat = q.transpose(-2, -1) @ torch.diag_embed(ev) @ q
    return (mat + mat.transpose(-2, -1)) * 0.5
