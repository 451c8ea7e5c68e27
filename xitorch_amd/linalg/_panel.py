"""Panel-major operator application shared by the native eigensolver and Krylov loops.

A *panel* is a padded device array (Bt, p, ld): Bt batch members, p vectors each, every vector a
contiguous length-N run (ld >= N, pads zero) — the reference's Fortran-order (.., N, p) view
(xitorch/_utils/tensor.py:21-32) seen as its transpose.  `PanelOperator.apply` computes
out[b, c, :N] = A_b x[b, c, :N] (or A_b^H) without any layout copies for the native operators:

  MatrixLinearOperator  -> xk_dense_mm   (K1; column-oriented variant when the matrix is symmetric)
  BandedLinearOperator  -> xk_banded_mm
  anything else         -> the operator's own .mm/.rmm on the (.., N, p) strided view
"""
import torch
from xitorch_amd import kernels as K

__all__ = ["PanelOperator", "pad_len", "to_panel", "from_panel"]

K1S_MIN_BYTES = 1.5e9      # operator storage from which the upper-triangle kernel K1s beats the full-matrix kernels


def pad_len(n):
    return (n + 7) // 8 * 8     # elements: keeps every vector 64 B aligned for f64 and f32


def to_panel(X, bdims, Bt, N):
    """(*batch, N, p) (broadcastable to bdims) -> zero-padded panel (Bt, p, ld)."""
    p = X.shape[-1]
    out = torch.zeros((Bt, p, pad_len(N)), dtype=X.dtype, device=X.device)
    out[:, :, :N].copy_(X.expand(*bdims, N, p).reshape(Bt, N, p).transpose(-2, -1))
    return out


def from_panel(P, bdims, N):
    """panel (Bt, p, ld) -> (*bdims, N, p) (a strided view, Fortran order like the reference's)."""
    return P[:, :, :N].transpose(-2, -1).reshape(*bdims, N, P.shape[1])


class PanelOperator:
    def __init__(self, A, bdims, Bt, N):
        from xitorch_amd.linop import MatrixLinearOperator, BandedLinearOperator
        self.A, self.bdims, self.Bt, self.N = A, list(bdims), Bt, N
        self.kind = "generic"
        self.symm = False
        self.symm_narrow = False
        self.napply = 0
        self.last_kernel = None     # which panel kernel served the last native apply (K1s / K1w / K1wr / K1 / banded)
        self.events = None          # when a list: (start, end, p) HIP events around every native launch
        self.hermitian = bool(getattr(A, "is_hermitian", False))
        nA = 1
        for d in A.shape[:-2]:
            nA *= d
        native_t = lambda t: t.is_cuda and t.dtype in (torch.float64, torch.float32)
        native_c = lambda t: t.is_cuda and t.dtype in (torch.complex128, torch.complex64)
        self.cplx, self.cj = False, False
        if isinstance(A, MatrixLinearOperator) and (native_t(A.mat) or native_c(A.mat)) and (nA == Bt or nA == 1):
            mat = A.mat
            flip = False
            if native_c(mat):
                # complex operator: applied by the real K1 kernels on its interleaved storage (K.dense_mm_complex);
                # lazily conjugated / transposed views (A.H = mat^T.conj()) only set orientation flags
                self.cplx = True
                if mat.is_conj():
                    mat, self.cj = mat.conj(), True
            if mat.dim() >= 2 and mat.stride(-1) != 1 and mat.stride(-2) == 1:
                mat, flip = mat.transpose(-2, -1), True           # a transposed view (e.g. A.H)
            if mat.is_contiguous() or mat.dim() == 2 and mat.stride(-1) == 1:
                self.kind, self.flip = "dense", flip
                self.herm_verified = bool(getattr(A, "hermitian_verified", False))
                self.mat = mat.reshape(nA, *mat.shape[-2:]) if mat.dim() > 2 else mat
                vn = 2 if mat.dtype == torch.float64 else 4
                # exactly symmetric storage: stream the upper triangle only (K1s)
                self.symm = bool(getattr(A, "symmetric_storage", False)) and N % vn == 0 and \
                    self.mat.stride(-2) % vn == 0 and not self.cplx
                # ... where that pays: K1s walks 1024-row tiles, one workgroup each, and has a floor of ~235 us per
                # launch (two launches) however small the operator; the one-launch full-matrix kernels are faster up to
                # ~1.5 GB of operator storage although they read twice the bytes (scripts/k1s_small_crossover.py,
                # profiles/r04_k1s_small_crossover.jsonl: 1 x 512^2 fp64 127 vs 11 us, 8 x 2048^2 236 vs 44, 4 x 4096^2
                # 238 vs 95)
                self.symm_narrow = self.symm and self.mat.numel() * self.mat.element_size() >= K1S_MIN_BYTES
        elif isinstance(A, BandedLinearOperator) and native_t(A.band) and (nA == Bt or nA == 1) \
                and A.band.is_contiguous():
            self.kind = "banded"
            self.band = A.band.reshape(nA, *A.band.shape[-2:])

    def diagonal(self):
        """diag(A) as a contiguous (nA, N) array (native operators only)."""
        if self.kind == "dense":
            mat = self.mat if self.mat.dim() == 3 else self.mat.unsqueeze(0)
            return mat.diagonal(dim1=-2, dim2=-1).contiguous()
        if self.kind == "banded":
            hb = (self.band.shape[-2] - 1) // 2
            return self.band[:, hb, :].contiguous()
        raise K._capi.NativeLibraryError("the diagonal of a generic LinearOperator is not available: pass it "
                                         "explicitly (precond=<tensor (*batch, N)>) or use a LinearOperator")

    def apply(self, X, out, trans=False):
        """out[:, :, :N] = A X  (trans: A^H X).  X, out: (Bt, p, ld)."""
        self.napply += 1
        N = self.N
        if self.hermitian:
            trans = False
        if self.events is not None and self.kind != "generic":
            e0, e1 = K.timing_event_pair()
            e0.record()
            self._native(X, out, trans)
            e1.record()
            self.events.append((e0, e1, X.shape[1], X.shape[0]))
            return out
        if self.kind != "generic":
            return self._native(X, out, trans)
        p = X.shape[1]
        x = X[:, :, :N].transpose(-2, -1).reshape(*self.bdims, N, p)
        y = self.A.rmm(x) if trans else self.A.mm(x)
        out[:, :, :N].copy_(y.expand(*self.bdims, N, p).reshape(self.Bt, N, p).transpose(-2, -1))
        return out

    def apply_on(self, X, out, k1_stream):
        """out = A X with the panel product running on `k1_stream` (the two-group pipeline's CU-masked stream)
        and the result ordered into the current stream.  For the symmetric-storage kernel only the tile kernel
        goes to `k1_stream`; its small fold runs on the current stream, off the panel-product critical path."""
        N = self.N
        if self.kind == "dense" and self.symm and self.symm_narrow and X.shape[1] <= 6 and not self.flip:
            self.napply += 1
            self.last_kernel = "K1s"
            e0, e1 = K.dense_symm_split(self.mat, X[:, :, :N], out[:, :, :N], k1_stream,
                                        timed=self.events is not None)
            if self.events is not None:
                self.events.append((e0, e1, X.shape[1], X.shape[0]))
            return out
        if self.kind == "dense" and self.symm and not self.flip and K.symm_wide_ok(self.mat, X[:, :, :N]):
            self.napply += 1
            self.last_kernel = "K1sw"
            e0, e1 = K.dense_symm_wide_split(self.mat, X[:, :, :N], out[:, :, :N], k1_stream,
                                             timed=self.events is not None)
            if self.events is not None:
                self.events.append((e0, e1, X.shape[1], X.shape[0]))
            return out
        cur = torch.cuda.current_stream()
        ready, done = K.sync_events(cur)            # cached per issuing stream, re-recorded on every launch
        ready.record(cur)
        with torch.cuda.stream(k1_stream):
            k1_stream.wait_event(ready)
            self.apply(X, out)
            done.record(k1_stream)
        cur.wait_event(done)
        return out

    def _native(self, X, out, trans):
        N = self.N
        if self.kind == "dense" and self.cplx:
            # stored matrix S, operator = S / S^T / conj(S) / S^H by (flip, cj); trans asks for the operator's adjoint
            K.dense_mm_complex(self.mat if self.mat.dim() == 3 else self.mat.unsqueeze(0), X[:, :, :N],
                               adjoint=(self.flip != trans), conj_io=(self.flip != self.cj), out=out[:, :, :N])
            return out
        if self.kind == "dense" and self.symm and not self.flip and K.symm_wide_ok(self.mat, X[:, :, :N]):
            # exactly symmetric fp32 storage, 9 .. 16 columns: the triangle once, both products on the matrix cores
            self.last_kernel = "K1sw"
            K.dense_symm_wide(self.mat, X[:, :, :N], out=out[:, :, :N])
        elif self.kind == "dense" and self.symm and self.symm_narrow and X.shape[1] < K.WIDE_MIN_P:
            self.last_kernel = "K1s"
            K.dense_symm(self.mat, X[:, :, :N], out=out[:, :, :N])
        elif self.kind == "dense":
            t = (trans != self.flip)
            # a real symmetric matrix equals its transpose: the column-oriented K1 variant (lanes own
            # output columns, panel values are wave-uniform scalars) measured 6.8 vs 6.3 TB/s at p = 6
            # (only for matrices whose symmetry was verified by LinearOperator.m — otherwise the product stays the
            # reference's mat @ x)
            if self.hermitian and not self.flip and self.herm_verified:
                t = True
            # (many columns in the ROW orientation — A X, non-Hermitian A — go to K1wr inside dense_mm: one pass over
            # the operator per 32 columns, no transposed copy)
            Xn = X[:, :, :N]
            wide = Xn.shape[1] >= K.WIDE_MIN_P
            if t:
                mat3 = self.mat
                self.last_kernel = "K1w" if wide and K._wide_ok(mat3, mat3.shape[-1], mat3.stride(-2),
                                                                mat3.stride(0) if mat3.dim() == 3 else 0) else "K1"
            else:
                self.last_kernel = "K1wr" if wide else "K1"
            K.dense_mm(self.mat, Xn, out=out[:, :, :N], trans=t)
        else:
            self.last_kernel = "banded"
            K.banded_mm(self.band, X[:, :, :N], out=out[:, :, :N], trans=trans)
        return out
