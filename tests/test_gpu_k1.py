"""-m gpu: the K1 HIP kernel (through the C ABI) against the oracle's dense operator on the same inputs."""
import pytest
import torch
from oracle import ops as oops
from xitorch_amd import kernels as K, _capi
from xitorch_amd import LinearOperator

pytestmark = pytest.mark.gpu

CASES = [
    # B, M, N, P, dtype, trans
    (2, 256, 256, 6, torch.float64, False), (3, 100, 130, 1, torch.float64, False),
    (2, 77, 201, 5, torch.float64, False), (1, 512, 512, 11, torch.float64, False),
    (2, 1000, 1000, 2, torch.float64, False), (2, 256, 384, 6, torch.float32, False),
    (2, 99, 131, 3, torch.float32, False), (2, 256, 384, 6, torch.float64, True),
    (3, 100, 131, 4, torch.float64, True), (2, 300, 200, 7, torch.float32, True),
    (1, 2048, 2048, 6, torch.float64, True), (4, 1, 64, 2, torch.float64, False),
    (1, 64, 2, 3, torch.float64, True), (2, 300, 256, 16, torch.float64, True), (1, 200, 130, 12, torch.float64, True),
    (2, 128, 128, 29, torch.float64, True), (1, 256, 256, 50, torch.float32, True), (1, 96, 96, 50, torch.float64, False),
]


@pytest.mark.parametrize("B,M,N,P,dtype,trans", CASES)
def test_dense_mm_vs_oracle(dev, B, M, N, P, dtype, trans):
    g = torch.Generator().manual_seed(B * 1000 + M + N + P)
    A = torch.randn(B, M, N, dtype=dtype, generator=g)
    X = torch.randn(B, P, M if trans else N, dtype=dtype, generator=g)
    op = oops.DenseOp(A.double())
    xo = X.double().transpose(-2, -1)
    ref = (op._rmm(xo) if trans else op._mm(xo)).transpose(-2, -1)         # oracle, fp64
    Y = K.dense_mm(A.to(dev), X.to(dev), trans=trans).cpu().double()
    tol = 1e-13 if dtype == torch.float64 else 3e-6        # fp64: summation-order differences only
    scale = ref.abs().max().item() + 1e-300
    assert (Y - ref).abs().max().item() / scale < tol * max(1.0, (N if not trans else M) ** 0.5)


def test_operator_surface_uses_native_kernel(dev):
    # LinearOperator.m(mat).mm/.mv/.rmm with broadcasting + extra leading dims, vs torch on CPU
    g = torch.Generator().manual_seed(5)
    mat = torch.randn(3, 40, 56, dtype=torch.float64, generator=g)
    op = LinearOperator.m(mat.to(dev))
    for xshape in [(56, 4), (3, 56, 2), (2, 3, 56, 5), (1, 56, 3)]:
        x = torch.randn(*xshape, dtype=torch.float64, generator=g)
        y = op.mm(x.to(dev)).cpu()
        assert torch.allclose(y, torch.matmul(mat, x), rtol=1e-12, atol=1e-12), xshape
    for xshape in [(40, 4), (3, 40, 2), (2, 3, 40, 5)]:
        x = torch.randn(*xshape, dtype=torch.float64, generator=g)
        y = op.rmm(x.to(dev)).cpu()
        assert torch.allclose(y, torch.matmul(mat.transpose(-2, -1), x), rtol=1e-12, atol=1e-12), xshape
    v = torch.randn(3, 56, dtype=torch.float64, generator=g)
    assert torch.allclose(op.mv(v.to(dev)).cpu(), torch.matmul(mat, v.unsqueeze(-1)).squeeze(-1), rtol=1e-12, atol=1e-12)
    # unbatched operator, batched operand -> batch folds into the panel width
    op1 = LinearOperator.m(mat[0].to(dev))
    x = torch.randn(5, 56, 3, dtype=torch.float64, generator=g)
    assert torch.allclose(op1.mm(x.to(dev)).cpu(), torch.matmul(mat[0], x), rtol=1e-12, atol=1e-12)


def test_dense_mm_autograd(dev):
    g = torch.Generator().manual_seed(6)
    mat = torch.randn(2, 12, 12, dtype=torch.float64, generator=g).to(dev).requires_grad_()
    x = torch.randn(2, 12, 3, dtype=torch.float64, generator=g).to(dev).requires_grad_()

    def f(m, xx):
        return LinearOperator.m(m, is_hermitian=False).mm(xx)
    assert torch.autograd.gradcheck(f, (mat, x))
    assert torch.autograd.gradgradcheck(f, (mat, x))


@pytest.mark.parametrize("B,M,N,P", [(3, 20, 8192, 6), (2, 54, 16384, 6), (64, 12, 4096, 6), (1, 7, 32768, 1)])
def test_gram_split_path_vs_oracle(dev, B, M, N, P):
    # skinny row sweeps (basis Gram blocks G = V W^T) take the split-contraction path
    g = torch.Generator().manual_seed(B + M + N)
    V = torch.randn(B, M, N, dtype=torch.float64, generator=g)
    W = torch.randn(B, P, N, dtype=torch.float64, generator=g)
    ref = torch.matmul(W, V.transpose(-2, -1))            # (B, P, M)
    G = K.dense_mm(V.to(dev), W.to(dev)).cpu()
    assert (G - ref).abs().max().item() < 1e-11 * N ** 0.5


@pytest.mark.parametrize("B,k,p,uppest,dtype", [(3, 12, 6, False, torch.float64), (2, 37, 4, True, torch.float64),
                                                 (5, 108, 6, False, torch.float64), (2, 128, 16, True, torch.float64),
                                                 (64, 54, 6, False, torch.float64), (2, 33, 3, False, torch.float32),
                                                 (1, 1, 1, False, torch.float64), (2, 7, 7, False, torch.float64)])
def test_small_eigh_vs_oracle(dev, B, k, p, uppest, dtype):
    g = torch.Generator().manual_seed(k * 7 + p)
    R = torch.randn(B, k, k, dtype=torch.float64, generator=g)
    T = (R + R.transpose(-2, -1)) * 0.5 + torch.diag(torch.arange(k, dtype=torch.float64))
    # embed in a larger buffer with garbage in the strict upper triangle: only the lower one may be read
    cap = k + 5
    Tbuf = torch.full((B, cap, cap), 777.0, dtype=torch.float64)
    Tbuf[:, :k, :k] = torch.tril(T) + torch.triu(torch.full((k, k), 99.0, dtype=torch.float64), 1)
    lam_ref, Y_ref = torch.linalg.eigh(T)                 # CPU LAPACK = what the oracle calls (symeig.py:174)
    sl = slice(k - p, k) if uppest else slice(0, p)
    lam, Y, sweeps = K.small_eigh(Tbuf.to(dev).to(dtype), k, p, uppest=uppest)
    lam, Y = lam.cpu().double(), Y.cpu().double()
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    scale = lam_ref.abs().max().item()
    assert (lam - lam_ref[:, sl]).abs().max().item() < tol * scale * 10
    assert torch.all(lam[:, 1:] >= lam[:, :-1])
    # eigenvector residual and orthonormality
    Yc = Y.transpose(-2, -1)                              # (B, k, p)
    res = torch.matmul(T, Yc) - Yc * lam.unsqueeze(-2)
    assert res.abs().max().item() < tol * scale * 50
    G = torch.matmul(Yc.transpose(-2, -1), Yc)
    assert (G - torch.eye(p, dtype=torch.float64)).abs().max().item() < tol * 100
    assert int(sweeps.max()) < 16


@pytest.fixture
def symm_run():
    """slabs per workgroup run of K1s (bits 8..15 of the entry points' `opts` argument, handed through the Python
    layer's K1S_OPTS); restored afterwards"""
    prev = K.K1S_OPTS

    def select(v, tile=0):
        # bit 2 / 3: force 512- / 1024-row tiles; forced opts = the one-workgroup-per-run launch (the resident and 8-wave
        # forms have their own test below)
        K.K1S_OPTS = (int(v) << 8) | {0: 0, 512: 4, 1024: 8}[tile]
    yield select
    K.K1S_OPTS = prev


@pytest.mark.parametrize("B,N,P,dtype", [(2, 2048, 6, torch.float64), (3, 1536, 4, torch.float64),
                                         (2, 1000, 6, torch.float64), (1, 512, 1, torch.float64),
                                         (2, 130, 3, torch.float64), (1, 3072, 7, torch.float64),
                                         (2, 2, 2, torch.float64), (2, 4096, 6, torch.float32),
                                         (1, 1100, 5, torch.float32), (1, 5000, 6, torch.float64)])
@pytest.mark.parametrize("run,tile", [(1, 1024), (2, 1024), (3, 1024), (1, 512), (2, 512), (3, 0)])
def test_dense_symm_vs_oracle(dev, B, N, P, dtype, run, tile, symm_run):
    # symmetric-storage K1s (upper triangle only) against the oracle's full dense product, for workgroup runs of 1, 2
    # and 3 column slabs (the row accumulator lives across a run; ragged last runs included) and for both tile heights
    # (512 rows: the small-launch form of fp64, r04; 0 = the library's choice)
    symm_run(run, tile)
    g = torch.Generator().manual_seed(N + P)
    R = torch.randn(B, N, N, dtype=dtype, generator=g)
    A = R + R.transpose(-2, -1)                                   # exactly symmetric
    assert torch.equal(A, A.transpose(-2, -1))
    X = torch.randn(B, P, N, dtype=dtype, generator=g)
    ref = oops.DenseOp(A.double(), True)._mm(X.double().transpose(-2, -1)).transpose(-2, -1)
    Y = K.dense_symm(A.to(dev), X.to(dev)).cpu().double()
    tol = 1e-13 if dtype == torch.float64 else 3e-6
    assert (Y - ref).abs().max().item() / ref.abs().max().item() < tol * N ** 0.5
    # the lower triangle must never be read: poison it
    Ap = torch.triu(A) + torch.tril(torch.full_like(A, float("nan")), -1)
    Y2 = K.dense_symm(Ap.to(dev), X.to(dev)).cpu().double()
    assert torch.equal(Y2, Y)                       # same launch shape -> bit-identical (fixed summation order)


@pytest.mark.parametrize("B,N,P,dtype", [(3, 4096, 6, torch.float64), (2, 5000, 5, torch.float64),
                                         (2, 6144, 6, torch.float32), (1, 8192, 3, torch.float64)])
def test_dense_symm_is_bit_reproducible(dev, B, N, P, dtype):
    # the row sums of the four waves of a workgroup meet in LDS in a fixed order (phase rotation + barriers) and the
    # partial slots are folded in a fixed order: repeated launches must agree bit for bit, also while another stream
    # keeps the GPU busy (different arrival order of the workgroups)
    g = torch.Generator().manual_seed(7 * N + P)
    R = torch.randn(B, N, N, dtype=dtype, generator=g)
    A = (R + R.transpose(-2, -1)).to(dev)
    X = torch.randn(B, P, N, dtype=dtype, generator=g).to(dev)
    Y0 = K.dense_symm(A, X).clone()
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=dev)
    for rep in range(6):
        if rep % 2:
            with torch.cuda.stream(side):
                for _ in range(4):
                    junk = junk @ junk * 1e-3
        Y = K.dense_symm(A, X)
        assert torch.equal(Y, Y0), "repetition %d differs" % rep
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,N,P,dtype", [(3, 4096, 6, torch.float64), (2, 5000, 5, torch.float64), (2, 130, 3, torch.float64),
                                         (2, 6144, 6, torch.float32), (1, 3072, 7, torch.float64), (2, 2, 2, torch.float64)])
@pytest.mark.parametrize("slots,run,tile", [(0, 1, 1024), (1, 1, 1024), (3, 2, 512), (8, 3, 0), (0, 1, 2048), (1, 2, 2048),
                                            (3, 8, 2048), (8, 1, 2048), (3, 1, 1024), (0, 3, 0)])
def test_dense_symm_resident_launch_is_bit_identical(dev, B, N, P, dtype, slots, run, tile):
    # the resident form of K1s (opts bit 4: `slots` workgroups — 0 = two per compute unit — take the runs from a queue)
    # must give the bits of the one-workgroup-per-run launch whatever workgroup serves which run and however many
    # runs one workgroup walks through (1 slot: all of them, in order; the queue word is reset by every launch)
    g = torch.Generator().manual_seed(11 * N + P)
    R = torch.randn(B, N, N, dtype=dtype, generator=g)
    A = (R + R.transpose(-2, -1)).to(dev)
    X = torch.randn(B, P, N, dtype=dtype, generator=g).to(dev)
    # (tile 2048: the 8-wave form of fp64 — opts bit 5: 2048 x 2048 tiles, one workgroup per compute unit)
    low = (int(run) << 8) | {0: 0, 512: 4, 1024: 8, 2048: 32}[tile]
    Y0 = K.dense_symm(A, X, opts=low).clone()
    ref = oops.DenseOp(A.cpu().double(), True)._mm(X.cpu().double().transpose(-2, -1)).transpose(-2, -1)
    tol = 1e-13 if dtype == torch.float64 else 3e-6
    assert (Y0.cpu().double() - ref).abs().max().item() / ref.abs().max().item() < tol * N ** 0.5
    for rep in range(3):
        Y = K.dense_symm(A, X, opts=low | K.K1S_PERSIST | (slots << 16))
        assert torch.equal(Y, Y0), "resident launch differs (repetition %d)" % rep
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,M,N,P,dtype", [(2, 512, 256, 32, torch.float32), (1, 300, 128, 17, torch.float32),
                                           (2, 256, 384, 12, torch.float32), (2, 512, 64, 16, torch.float64),
                                           (1, 130, 96, 32, torch.float64), (2, 77, 32, 25, torch.float64),
                                           (1, 256, 256, 50, torch.float64), (1, 1024, 1024, 50, torch.float32),
                                           # r04: fp32 panels of 9 .. 16 columns take the 16-wide MFMA tile
                                           # (v_mfma_f32_16x16x4_f32): no padded output columns at P = 16
                                           (2, 512, 256, 9, torch.float32), (2, 512, 256, 16, torch.float32),
                                           (3, 1030, 384, 16, torch.float32), (2, 256, 128, 17, torch.float32),
                                           (2, 256, 64, 9, torch.float64), (2, 256, 64, 17, torch.float64),
                                           (1, 2048, 2048, 16, torch.float32), (1, 2048, 2048, 40, torch.float32)])
def test_dense_wide_mfma_vs_oracle(dev, B, M, N, P, dtype):
    # K1w: Y = A^T X for many panel columns on the matrix cores (asymmetric A catches transposed fragments)
    g = torch.Generator().manual_seed(M + N + P)
    A = torch.randn(B, M, N, dtype=dtype, generator=g)
    X = torch.randn(B, P, M, dtype=dtype, generator=g)
    ref = oops.DenseOp(A.double())._rmm(X.double().transpose(-2, -1)).transpose(-2, -1)
    Y = K.dense_mm(A.to(dev), X.to(dev), trans=True).cpu().double()          # dispatches to K1w for P >= 12
    Yv = K.dense_mm(A.to(dev), X.to(dev), trans=True, wide=False).cpu().double()
    tol = 1e-13 if dtype == torch.float64 else 3e-6
    scale = ref.abs().max().item()
    assert (Y - ref).abs().max().item() / scale < tol * M ** 0.5
    assert (Yv - ref).abs().max().item() / scale < tol * M ** 0.5
    if P <= 32:
        Yw = K.dense_wide(A.to(dev), X.to(dev)).cpu().double()
        if P >= K.WIDE_MIN_P:
            assert torch.equal(Yw, Y)                   # (the dispatcher took this very kernel)
        else:
            assert (Yw - ref).abs().max().item() / scale < tol * M ** 0.5
        assert torch.equal(K.dense_wide(A.to(dev), X.to(dev)).cpu().double(), Yw)
    # fixed accumulation order: bit-identical from launch to launch
    assert torch.equal(K.dense_mm(A.to(dev), X.to(dev), trans=True).cpu().double(), Y)


def test_exact_symmetry_detection_large_batched(dev):
    # LinearOperator.m on a big batched matrix takes the member-by-member scan; it must tell exact symmetry
    # (-> upper-triangle kernel) from allclose-only symmetry (-> full-matrix kernel) and from non-symmetry
    from xitorch_amd import synthetic
    mat = synthetic.dense_symmetric(5, 2048, "S2", device=dev)              # 21M elements > 2^24
    A = LinearOperator.m(mat)
    assert A.is_hermitian and A.symmetric_storage
    m2 = mat.clone()
    m2[3, 5, 9] += 1e-13                                                     # allclose, not bit-exact
    A2 = LinearOperator.m(m2, is_hermitian=True)
    assert A2.is_hermitian and not A2.symmetric_storage
    m3 = mat.clone()
    m3[4, 7, 11] += 1.0
    assert not LinearOperator.m(m3).is_hermitian
    with pytest.raises(RuntimeError, match="indicated to be hermitian"):
        LinearOperator.m(m3, is_hermitian=True)
    # both kernels give the same product on the exactly symmetric operator
    x = torch.randn(5, 2048, 3, dtype=torch.float64, device=dev)
    from xitorch_amd.linalg._panel import PanelOperator, to_panel
    Xp = to_panel(x, [5], 5, 2048)
    y1 = PanelOperator(A, [5], 5, 2048).apply(Xp, torch.zeros_like(Xp))
    y2 = PanelOperator(A2, [5], 5, 2048).apply(Xp, torch.zeros_like(Xp))
    assert PanelOperator(A, [5], 5, 2048).symm and not PanelOperator(A2, [5], 5, 2048).symm
    assert torch.allclose(y1, y2, rtol=1e-12, atol=1e-9)


def _tri_case(kind, B, k, g):
    """projected matrices for the tridiagonalisation kernel: generic, Davidson-like (isolated low eigenvalues +
    a dense bulk), tight clusters, exact multiplicities, already-tridiagonal / diagonal input (zero reflectors)"""
    f64 = torch.float64
    if kind == "generic":
        R = torch.randn(B, k, k, dtype=f64, generator=g)
        return (R + R.transpose(-2, -1)) * 0.5 + torch.diag(torch.arange(k, dtype=f64))
    Q, _ = torch.linalg.qr(torch.randn(B, k, k, dtype=f64, generator=g))
    if kind == "davidson":
        d = torch.cat([torch.arange(1.0, 7.0, dtype=f64), 50.0 + 50.0 * torch.rand(max(k - 6, 0), dtype=f64, generator=g)])[:k]
    elif kind == "cluster":
        d = torch.linspace(1.0, 90.0, k, dtype=f64)
        d[:min(4, k)] = 1.0 + 1e-9 * torch.arange(min(4, k), dtype=f64)       # four eigenvalues within 3e-9
    elif kind == "multiple":
        d = torch.linspace(3.0, 40.0, k, dtype=f64)
        d[:min(3, k)] = 2.0                                                   # an exactly triple eigenvalue
        d[-2:] = 77.0
    elif kind == "diagonal":
        return torch.diag_embed(torch.rand(B, k, dtype=f64, generator=g) * 10.0 - 3.0)
    elif kind == "tridiagonal":
        T = torch.diag_embed(torch.rand(B, k, dtype=f64, generator=g) * 10.0)
        e = torch.rand(B, k - 1, dtype=f64, generator=g)
        return T + torch.diag_embed(e, offset=1) + torch.diag_embed(e, offset=-1)
    T = Q @ torch.diag_embed(d.expand(B, k)) @ Q.transpose(-2, -1)
    return (T + T.transpose(-2, -1)) * 0.5


@pytest.mark.parametrize("kind", ["generic", "davidson", "cluster", "multiple", "diagonal", "tridiagonal"])
@pytest.mark.parametrize("B,k,p,uppest,dtype", [(3, 12, 6, False, torch.float64), (2, 37, 4, True, torch.float64),
                                                 (5, 108, 6, False, torch.float64), (2, 112, 10, True, torch.float64),
                                                 (64, 54, 6, False, torch.float64), (2, 33, 3, False, torch.float32),
                                                 (1, 1, 1, False, torch.float64), (2, 2, 2, True, torch.float64),
                                                 (2, 3, 1, False, torch.float64), (2, 7, 7, False, torch.float64),
                                                 (3, 65, 6, True, torch.float64), (2, 128, 2, False, torch.float64)])
def test_small_eigh_tridiagonalisation_kernel(dev, kind, B, k, p, uppest, dtype):
    """K3t (Householder tridiagonalisation + bisection + inverse iteration) against CPU LAPACK eigh — what the oracle
    calls at symeig.py:174 — : eigenvalues, residuals, orthonormality, ascending order, self-check flags"""
    if kind == "tridiagonal" and k < 2:
        pytest.skip("needs an off-diagonal")
    if not K.small_eigh_tri_ok(k, p, dtype):
        pytest.skip("beyond the LDS capacity of the kernel for this (k, p)")
    g = torch.Generator().manual_seed(k * 11 + p)
    T = _tri_case(kind, B, k, g)
    cap = k + 3
    Tbuf = torch.full((B, cap, cap), 777.0, dtype=torch.float64)
    Tbuf[:, :k, :k] = torch.tril(T) + torch.triu(torch.full((k, k), 99.0, dtype=torch.float64), 1)
    Td = Tbuf.to(dev).to(dtype)
    Tq = torch.tril(Td[:, :k, :k]).cpu().double()
    Tq = Tq + torch.tril(Tq, -1).transpose(-2, -1)                            # the matrix the kernel really sees
    lam_ref = torch.linalg.eigvalsh(Tq)
    sl = slice(k - p, k) if uppest else slice(0, p)
    lam, Y, info = K.small_eigh(Td, k, p, uppest=uppest, method="tri")
    assert info.cpu().abs().max().item() == 0, info
    lam, Y = lam.cpu().double(), Y.cpu().double()
    tol = 1e-13 if dtype == torch.float64 else 5e-6
    scale = max(lam_ref.abs().max().item(), 1.0)
    assert (lam - lam_ref[:, sl]).abs().max().item() <= tol * scale * 20
    assert torch.all(lam[:, 1:] >= lam[:, :-1])
    Yc = Y.transpose(-2, -1)
    res = torch.matmul(Tq, Yc) - Yc * lam.unsqueeze(-2)
    assert res.abs().max().item() <= tol * scale * 200
    G = torch.matmul(Yc.transpose(-2, -1), Yc)
    assert (G - torch.eye(p, dtype=torch.float64)).abs().max().item() <= tol * 2000
    # same invariant subspace as LAPACK's vectors (clusters / multiplicities: compare projectors)
    _, Yr = torch.linalg.eigh(Tq)
    Yr = Yr[..., sl]
    gapped = True
    if p < k:
        edge = (lam_ref[:, p] - lam_ref[:, p - 1]) if not uppest else (lam_ref[:, k - p] - lam_ref[:, k - p - 1])
        gapped = bool((edge > 1e-6 * scale).all())
    if gapped:
        P1, P2 = Yc @ Yc.transpose(-2, -1), Yr @ Yr.transpose(-2, -1)
        assert (P1 - P2).abs().max().item() <= (1e-8 if dtype == torch.float64 else 1e-2)


def test_small_eigh_tri_flags_garbage(dev):
    """a matrix containing NaN cannot pass the kernel's self-check: the flag is what sends the call to the fallback"""
    T = torch.eye(20, dtype=torch.float64).repeat(2, 1, 1)
    T[1, 5, 3] = float("nan")                      # lower triangle: the part that is read
    lam, Y, info = K.small_eigh(T.to(dev), 20, 3, method="tri")
    assert info.cpu().tolist()[1] != 0 and info.cpu().tolist()[0] == 0


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-13), (torch.float32, 3e-6)])
@pytest.mark.parametrize("B,M,N,P", [(2, 256, 512, 16), (3, 700, 700, 50), (1, 130, 1026, 12), (2, 64, 32, 32),
                                     (8, 1024, 2048, 48), (2, 65, 8192, 17), (1, 2048, 96, 33), (4, 300, 4100, 13)])
def test_dense_rows_wide_vs_torch(dev, dtype, tol, B, M, N, P):
    """K1wr: A X in the row orientation for wide panels (LDS-transposed tiles, scalar panel loads), incl. ragged row
    counts, column counts that are not a multiple of the 32/64-column sub-tile, split contractions and the
    benchmarks_solve.py shape (n = 700, ncols = 50); also through dense_mm's dispatch and with a broadcast operator"""
    g = torch.Generator().manual_seed(M + N + P)
    A = torch.randn(B, M, N, dtype=torch.float64, generator=g)
    X = torch.randn(B, P, N, dtype=torch.float64, generator=g)
    ref = torch.matmul(X, A.transpose(-2, -1))                        # (B, P, M)
    Ad, Xd = A.to(dev, dtype), X.to(dev, dtype)
    vn = 2 if dtype == torch.float64 else 4
    if N % vn:
        pytest.skip("needs N multiple of the vector width")
    Y = K.dense_rows_wide(Ad, Xd)
    scale = float(N) ** 0.5 * 4
    assert (Y.double().cpu() - ref).abs().max().item() <= tol * scale
    Y2 = K.dense_mm(Ad, Xd, trans=False)                               # the dispatcher picks K1wr for P >= 12
    assert torch.equal(Y2, Y)
    Y3 = K.dense_rows_wide(Ad[:1], Xd)                                 # one operator for the whole panel batch
    ref3 = torch.matmul(X, A[:1].transpose(-2, -1))
    assert (Y3.double().cpu() - ref3).abs().max().item() <= tol * scale
    # deterministic (fixed-order fold of the split contraction)
    assert torch.equal(K.dense_rows_wide(Ad, Xd), Y)


@pytest.mark.parametrize("shape,dtype", [((3, 1000, 1000), torch.float64), ((1, 5, 16388), torch.float32),
                                         ((2, 2048, 2048), torch.float64), ((7, 12), torch.float32)])
def test_stream_read_utility_runs_on_ragged_shapes(dev, shape, dtype):
    # the measurement utility behind bench.py's roofline.stream_read: every byte once, nothing written; rows and row
    # lengths that are not multiples of its 1024 x 8 KB tiles must stay inside the buffer (a guard region is checked)
    n = 1
    for d in shape:
        n *= d
    guard = 4096
    buf = torch.full((n + 2 * guard,), 7.0, dtype=dtype, device=dev)
    t = buf[guard:guard + n].view(shape)
    if (t.data_ptr() % 16) or (shape[-1] * t.element_size()) % 16:
        pytest.skip("alignment")
    before = buf.clone()
    nbytes = K.stream_read(t)
    torch.cuda.synchronize()
    assert nbytes == n * t.element_size()
    assert torch.equal(buf, before)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("B", [1, 5, 64, 200])
def test_group_status_kernel(dev, dtype, B):
    # {max|resid| (NaN-propagating), max Cholesky flag, max K3t flag} of a batch group in one launch
    g = torch.Generator().manual_seed(B)
    rmax = torch.rand(B, dtype=dtype, generator=g).to(dev)
    info = torch.randint(0, 3, (B,), dtype=torch.int32, generator=g).to(dev)
    flag = torch.randint(0, 5, (B,), dtype=torch.int32, generator=g).to(dev)
    st = torch.full((3,), -1.0, dtype=torch.float64, device=dev)
    K.group_status(rmax, info, flag, st)
    assert st.tolist() == [float(rmax.max().double()), float(info.max()), float(flag.max())]
    K.group_status(rmax, info, None, st)
    assert st.tolist()[2] == 0.0
    rmax[B // 2] = float("nan")
    K.group_status(rmax, info, flag, st)
    out = st.tolist()
    assert out[0] != out[0] and out[1] == float(info.max())


@pytest.mark.parametrize("B,k,p,uppest,dtype", [(3, 130, 6, False, torch.float64), (2, 200, 6, True, torch.float64),
                                                (2, 333, 4, False, torch.float64), (2, 512, 6, False, torch.float64),
                                                (1, 600, 6, False, torch.float64), (1, 768, 6, True, torch.float64),
                                                (2, 300, 12, False, torch.float64), (2, 256, 16, False, torch.float32),
                                                (2, 400, 6, False, torch.float32), (2, 129, 1, False, torch.float64),
                                                (2, 100, 20, False, torch.float64), (2, 300, 40, True, torch.float64),
                                                (1, 200, 64, False, torch.float64), (2, 24, 17, False, torch.float64),
                                                (1, 90, 33, False, torch.float32),
                                                # r05: fp64 orders 769 .. 1024 (one launch per Householder step with
                                                # 16 column slots, 256-thread workgroups)
                                                (2, 800, 6, False, torch.float64), (1, 1024, 6, True, torch.float64),
                                                (1, 900, 20, False, torch.float64), (1, 769, 3, False, torch.float64),
                                                # r06: orders 1025 .. 1536 (24 column slots)
                                                (1, 1200, 6, False, torch.float64), (1, 1536, 4, True, torch.float64),
                                                (1, 1500, 6, False, torch.float32), (2, 1025, 3, False, torch.float64)])
def test_small_eigh_big_vs_lapack(dev, B, k, p, uppest, dtype):
    """K3g in its one-launch-per-Householder-step form (algo = 1; orders 129 .. 768, matrix in global memory) against
    LAPACK: eigenvalues, residual, orthonormality, on
    matrices shaped like a Davidson T (a few separated eigenvalues below a dense band) and on random ones; the
    matrix is handed over inside a larger allocation (ldt > k), only its lower triangle holds the data."""
    assert K.small_eigh_big_ok(k, p, dtype)
    g = torch.Generator().manual_seed(k + p)
    cap = k + 7
    for kind in ("ritz", "random"):
        if kind == "ritz":
            Q, _ = torch.linalg.qr(torch.randn(B, k, k, dtype=torch.float64, generator=g))
            d = torch.cat([torch.arange(1.0, 9.0, dtype=torch.float64),
                           50.0 + 50.0 * torch.arange(k - 8, dtype=torch.float64) / (k - 8)])
            Tm = Q @ torch.diag_embed(d.expand(B, k)) @ Q.transpose(-2, -1)
        else:
            R = torch.randn(B, k, k, dtype=torch.float64, generator=g)
            Tm = R + R.transpose(-2, -1)
        Tm = (Tm + Tm.transpose(-2, -1)) * 0.5
        lam_ref = torch.linalg.eigvalsh(Tm)
        buf = torch.full((B, cap, cap), float("nan"), dtype=dtype)
        buf[:, :k, :k] = torch.tril(Tm).to(dtype) + torch.triu(torch.full((k, k), float("nan"), dtype=dtype), 1)
        # the tridiagonalisation is spread over W workgroups per matrix, one launch per Householder step (automatic W,
        # an odd W, 8 with 256-thread workgroups, 16): same answers, each bit-reproducible
        if True:
            for W, threads in ((0, 512), (3, 512), (8, 256), (16, 512)):
                tag = (kind, W, threads)
                dbuf = buf.to(dev)
                lam, Y, info = K.small_eigh_big(dbuf, k, p, uppest=uppest, wg=W, threads=threads, algo=1)
                lam2, Y2, _ = K.small_eigh_big(dbuf, k, p, uppest=uppest, wg=W, threads=threads, algo=1)
                assert torch.equal(lam, lam2) and torch.equal(Y, Y2), tag
                assert int(info.max()) == 0, tag
                lam, Y = lam.cpu().double(), Y.cpu().double()
                sl = slice(k - p, k) if uppest else slice(0, p)
                tol = 1e-12 if dtype == torch.float64 else 3e-5
                scale = lam_ref.abs().max().item()
                assert (lam - lam_ref[:, sl]).abs().max().item() < tol * scale * 10, tag
                assert torch.all(lam[:, 1:] >= lam[:, :-1])
                Yc = Y.transpose(-2, -1)
                res = torch.matmul(Tm, Yc) - Yc * lam.unsqueeze(-2)
                assert res.abs().max().item() < tol * scale * 100, tag
                G = torch.matmul(Yc.transpose(-2, -1), Yc)
                assert (G - torch.eye(p, dtype=torch.float64)).abs().max().item() < tol * 200, tag


@pytest.mark.parametrize("B,k,p,uppest,dtype", [(2, 8, 3, False, torch.float64), (2, 35, 4, False, torch.float64),
                                                (3, 130, 6, False, torch.float64), (2, 200, 6, True, torch.float64),
                                                (2, 256, 6, False, torch.float64), (2, 257, 6, False, torch.float64),
                                                (2, 333, 4, False, torch.float64), (1, 400, 6, True, torch.float64),
                                                (2, 128, 16, False, torch.float32), (2, 384, 6, False, torch.float32),
                                                (1, 500, 6, False, torch.float32), (33, 150, 6, False, torch.float64),
                                                (2, 192, 40, True, torch.float64), (1, 64, 6, False, torch.float64),
                                                # r06: up to 256 wanted pairs (was 64)
                                                (1, 300, 100, False, torch.float64), (2, 200, 200, False, torch.float64),
                                                (1, 420, 256, True, torch.float32)])
def test_small_eigh_big_persistent_vs_lapack(dev, B, k, p, uppest, dtype):
    """K3g in its persistent form (r06, algo = 3: the trailing block of order <= 256 (fp64) / 384 (fp32) in the registers
    of ONE workgroup per matrix, one launch for the whole Householder reduction; larger orders start with step launches
    and hand over) against LAPACK like the other forms: eigenvalues, residual, orthonormality, bit-reproducible; orders
    around the register limit, the hand-over with several workgroup counts of the step launches."""
    assert K.small_eigh_big_ok(k, p, dtype)
    g = torch.Generator().manual_seed(k + p)
    cap = k + 5
    limit = 256 if dtype == torch.float64 else 384
    for kind in ("ritz", "random"):
        if kind == "ritz":
            Q, _ = torch.linalg.qr(torch.randn(B, k, k, dtype=torch.float64, generator=g))
            d = torch.cat([torch.arange(1.0, 9.0, dtype=torch.float64)[:min(8, k - 1)],
                           50.0 + 50.0 * torch.arange(k - min(8, k - 1), dtype=torch.float64) / (k - 8 if k > 8 else 1)])
            Tm = Q @ torch.diag_embed(d.expand(B, k)) @ Q.transpose(-2, -1)
        else:
            R = torch.randn(B, k, k, dtype=torch.float64, generator=g)
            Tm = R + R.transpose(-2, -1)
        Tm = (Tm + Tm.transpose(-2, -1)) * 0.5
        lam_ref = torch.linalg.eigvalsh(Tm)
        buf = torch.full((B, cap, cap), float("nan"), dtype=dtype)
        buf[:, :k, :k] = torch.tril(Tm).to(dtype) + torch.triu(torch.full((k, k), float("nan"), dtype=dtype), 1)
        dbuf = buf.to(dev)
        for W in ((0,) if k <= limit else (0, 3, 8)):
            tag = (kind, W)
            # the workspace (work copy of the matrix, hand-over blocks) starts out as NaN: nothing may depend on what an
            # earlier call left there (the persistent kernel writes only the reflectors into its work copy)
            K._workspace(1 << 22, dtype, dev).fill_(float("nan"))
            lam, Y, info = K.small_eigh_big(dbuf, k, p, uppest=uppest, wg=W, algo=3)
            lam2, Y2, _ = K.small_eigh_big(dbuf, k, p, uppest=uppest, wg=W, algo=3)
            assert torch.equal(lam, lam2) and torch.equal(Y, Y2), tag
            assert int(info.max()) == 0, tag
            lam, Y = lam.cpu().double(), Y.cpu().double()
            sl = slice(k - p, k) if uppest else slice(0, p)
            tol = 1e-12 if dtype == torch.float64 else 3e-5
            scale = lam_ref.abs().max().item()
            assert (lam - lam_ref[:, sl]).abs().max().item() < tol * scale * 10, tag
            assert torch.all(lam[:, 1:] >= lam[:, :-1])
            Yc = Y.transpose(-2, -1)
            res = torch.matmul(Tm, Yc) - Yc * lam.unsqueeze(-2)
            assert res.abs().max().item() < tol * scale * 100, tag
            G = torch.matmul(Yc.transpose(-2, -1), Yc)
            assert (G - torch.eye(p, dtype=torch.float64)).abs().max().item() < tol * 200, tag


@pytest.mark.parametrize("kind", ["diagonal", "int_diagonal", "identity", "zero", "block2", "tiny", "huge"])
@pytest.mark.parametrize("k,p,dtype", [(40, 6, torch.float64), (150, 6, torch.float64), (300, 9, torch.float64),
                                       (96, 16, torch.float32)])
def test_small_eigh_edge_matrices(dev, kind, k, p, dtype):
    """Matrices that stress the bisection of the Rayleigh-Ritz solvers (r06: product-form Sturm counts narrow the bracket,
    the ratio form confirms it): decoupled (diagonal: every e = 0; integer diagonal: the multisection's shifts hit
    eigenvalues exactly; two dense blocks), one cluster (identity), the zero matrix — all must come out right, unflagged,
    eigenvalues ascending; and scales at which the Householder reduction / e^2 leave the floating-point range (1e-150 /
    1e150 in fp64, 1e-30 / 1e30 in fp32) must be FLAGGED (the caller repeats those on the library, which rescales) —
    never returned wrong.  Every form: K3t where it fits, K3g forms 0 / 1 / 2 / 3."""
    g = torch.Generator().manual_seed(k + p)
    B = 2
    scale = 1.0
    if kind == "diagonal":
        Tm = torch.diag_embed(torch.randn(B, k, dtype=torch.float64, generator=g))
    elif kind == "int_diagonal":
        Tm = torch.diag_embed(torch.randint(-8, 9, (B, k), generator=g).double())
    elif kind == "identity":
        Tm = 3.0 * torch.eye(k, dtype=torch.float64).expand(B, k, k).clone()
    elif kind == "zero":
        Tm = torch.zeros(B, k, k, dtype=torch.float64)
    else:
        h = k // 2
        Tm = torch.zeros(B, k, k, dtype=torch.float64)
        R1 = torch.randn(B, h, h, dtype=torch.float64, generator=g)
        R2 = torch.randn(B, k - h, k - h, dtype=torch.float64, generator=g)
        Tm[:, :h, :h] = R1 + R1.transpose(1, 2)
        Tm[:, h:, h:] = R2 + R2.transpose(1, 2)
        if kind == "tiny":
            scale = 1e-150 if dtype == torch.float64 else 1e-30
        elif kind == "huge":
            scale = 1e150 if dtype == torch.float64 else 1e30
        Tm = Tm * scale
    ref = torch.linalg.eigvalsh(Tm)[:, :p]
    Tq = Tm.to(dtype).double()
    Td = torch.tril(Tm).to(dtype).to(dev)
    forms = [("big", a) for a in (0, 1, 3)] + ([("big", 2)] if k >= 35 else [])
    if K.small_eigh_tri_ok(k, p, dtype):
        forms.append(("tri", None))
    tol = 1e-12 if dtype == torch.float64 else 3e-5
    for name, algo in forms:
        K._workspace(1 << 22, dtype, dev).fill_(float("nan"))
        if name == "tri":
            lam, Y, info = K.small_eigh(Td, k, p, method="tri")
        else:
            lam, Y, info = K.small_eigh_big(Td, k, p, algo=algo)
        tag = (kind, name, algo)
        if kind in ("tiny", "huge"):
            assert int(info.min()) != 0, tag                   # flagged: never a wrong answer
            continue
        assert int(info.max()) == 0, tag
        lam, Y = lam.cpu().double(), Y.cpu().double()
        nrm = max(float(ref.abs().max()), 1.0)
        assert torch.all(lam[:, 1:] >= lam[:, :-1]), tag
        assert (lam - ref).abs().max().item() < 10 * tol * nrm, tag
        Yc = Y.transpose(1, 2)
        assert (Tq @ Yc - Yc * lam.unsqueeze(1)).abs().max().item() < 100 * tol * nrm, tag
        assert (Yc.transpose(1, 2) @ Yc - torch.eye(p, dtype=torch.float64)).abs().max().item() < 200 * tol, tag


@pytest.mark.parametrize("B,k,p,uppest,dtype", [(2, 35, 4, False, torch.float64), (3, 130, 6, False, torch.float64),
                                                (2, 200, 6, True, torch.float64), (2, 333, 4, False, torch.float64),
                                                (2, 512, 6, False, torch.float64), (1, 582, 6, False, torch.float64),
                                                (2, 600, 12, True, torch.float64), (1, 614, 5, False, torch.float64), (2, 256, 16, False, torch.float32),
                                                (2, 401, 6, False, torch.float32), (1, 768, 6, True, torch.float32),
                                                (1, 1000, 8, False, torch.float32),
                                                (2, 300, 40, True, torch.float64), (33, 257, 6, False, torch.float64)])
def test_small_eigh_big_two_stage_vs_lapack(dev, B, k, p, uppest, dtype):
    """K3g in its two-stage form (r04: dense -> band of 16 sub-diagonals by block reflectors, band -> tridiagonal by
    bulge chasing in LDS, vectors back through both stages) against LAPACK, like the one-stage test: eigenvalues,
    residual, orthonormality, bit-reproducible; orders that are no multiple of the panel / strip, the largest fp64 order
    whose band fits the LDS, more matrices than one wave of workgroups."""
    assert K.small_eigh_big_ok(k, p, dtype)
    g = torch.Generator().manual_seed(k + p)
    cap = k + 5
    for kind in ("ritz", "random"):
        if kind == "ritz":
            Q, _ = torch.linalg.qr(torch.randn(B, k, k, dtype=torch.float64, generator=g))
            d = torch.cat([torch.arange(1.0, 9.0, dtype=torch.float64),
                           50.0 + 50.0 * torch.arange(k - 8, dtype=torch.float64) / (k - 8)])
            Tm = Q @ torch.diag_embed(d.expand(B, k)) @ Q.transpose(-2, -1)
        else:
            R = torch.randn(B, k, k, dtype=torch.float64, generator=g)
            Tm = R + R.transpose(-2, -1)
        Tm = (Tm + Tm.transpose(-2, -1)) * 0.5
        lam_ref = torch.linalg.eigvalsh(Tm)
        buf = torch.full((B, cap, cap), float("nan"), dtype=dtype)
        buf[:, :k, :k] = torch.tril(Tm).to(dtype) + torch.triu(torch.full((k, k), float("nan"), dtype=dtype), 1)
        dbuf = buf.to(dev)
        lam, Y, info = K.small_eigh_big(dbuf, k, p, uppest=uppest, algo=2)
        lam2, Y2, _ = K.small_eigh_big(dbuf, k, p, uppest=uppest, algo=2)
        assert torch.equal(lam, lam2) and torch.equal(Y, Y2), kind
        assert int(info.max()) == 0, kind
        lam, Y = lam.cpu().double(), Y.cpu().double()
        sl = slice(k - p, k) if uppest else slice(0, p)
        tol = 1e-12 if dtype == torch.float64 else 3e-5
        scale = lam_ref.abs().max().item()
        assert (lam - lam_ref[:, sl]).abs().max().item() < tol * scale * 10, kind
        Yc = Y.transpose(-2, -1)
        res = torch.matmul(Tm, Yc) - Yc * lam.unsqueeze(-2)
        assert res.abs().max().item() < tol * scale * 100, kind
        G = torch.matmul(Yc.transpose(-2, -1), Yc)
        assert (G - torch.eye(p, dtype=torch.float64)).abs().max().item() < tol * 200, kind


@pytest.mark.parametrize("B,N,P", [(2, 1024, 16), (1, 2048, 9), (3, 1088, 12), (1, 4096, 16), (2, 2304, 13),
                                   (1, 8192, 16), (2, 1472, 16)])
@pytest.mark.parametrize("form", [0, 1, 3, 9])
def test_dense_symm_wide_mfma_vs_oracle(dev, B, N, P, form, monkeypatch):
    """K1sw (r04): exactly symmetric fp32 storage, 9 .. 16 panel columns: the upper triangle is streamed once and both
    y_I += A_IJ x_J and y_J += A_IJ^T x_I run on the matrix cores (torch.matmul(mat, x), linop.py:695-696, inside the
    eigensolver of BASELINE configs[4]).  Against the oracle's operator; orders that are not multiples of the 256-column
    strip or the 512-row tile; only the triangle (plus the 64 x 64 diagonal blocks) may be read; bit-reproducible.
    form 1 / 3: the workgroup-cooperative kernel (opts bit 0 of the C ABI; bit 1 = wave priority); form 9 (r06, bits 0 + 3):
    the column part straight from the load registers + a ring of four 16 x 64 blocks in flight per wave."""
    monkeypatch.setattr(K, "K1SW_OPTS", form)
    g = torch.Generator().manual_seed(N + P)
    R = torch.randn(B, N, N, dtype=torch.float32, generator=g)
    A = (R + R.transpose(1, 2)).contiguous()
    assert torch.equal(A, A.transpose(1, 2))
    X = torch.randn(B, P, N, dtype=torch.float32, generator=g)
    ref = oops.DenseOp(A.double())._mm(X.double().transpose(-2, -1)).transpose(-2, -1)
    Ad, Xd = A.to(dev), X.to(dev)
    assert K.symm_wide_ok(Ad, Xd)
    Y = K.dense_symm_wide(Ad, Xd)
    scale = ref.abs().max().item()
    assert (Y.cpu().double() - ref).abs().max().item() / scale < 3e-6 * N ** 0.5
    assert torch.equal(K.dense_symm_wide(Ad, Xd), Y)
    # everything strictly below the diagonal and outside the 64 x 64 diagonal blocks is never touched
    i = torch.arange(N)
    untouched = (i[:, None] > i[None, :]) & ((i[:, None] // 64) != (i[None, :] // 64))
    Ap = Ad.clone()
    Ap[:, untouched.to(dev)] = float("nan")
    assert torch.equal(K.dense_symm_wide(Ap, Xd), Y)
    # one operator for the whole panel batch
    Y1 = K.dense_symm_wide(Ad[:1], Xd)
    ref1 = torch.matmul(A[:1].double(), X.double().transpose(-2, -1)).transpose(-2, -1)
    assert (Y1.cpu().double() - ref1).abs().max().item() / scale < 3e-6 * N ** 0.5
    # the eigensolver's operator wrapper picks this kernel for such panels, also in its split (two-stream) form
    from xitorch_amd.linalg._panel import PanelOperator
    op = PanelOperator(LinearOperator.m(Ad, is_hermitian=True), [B], B, N)
    ld = (N + 7) // 8 * 8
    Xp = torch.zeros(B, P, ld, dtype=torch.float32, device=dev)
    Xp[:, :, :N] = Xd
    out = torch.zeros_like(Xp)
    op.apply(Xp, out)
    assert op.last_kernel == "K1sw" and torch.equal(out[:, :, :N], Y)
    side = torch.cuda.Stream(device=dev)
    out2 = torch.zeros_like(Xp)
    op.apply_on(Xp, out2, side)
    torch.cuda.synchronize()
    assert torch.equal(out2[:, :, :N], Y)


@pytest.mark.parametrize("B,N,P", [(2, 1152, 9), (1, 2624, 16), (3, 1024, 11)])
def test_dense_symm_wide_r06_form_padded_storage_and_forms_agree(dev, B, N, P, monkeypatch):
    """(r06) the shipped K1sw form (opts 9): operators with a padded leading dimension and a batch stride that is not
    N * lda give the bits of the contiguous copy (the band descriptor spans 64 rows of `lda`), two launches are
    `torch.equal`, and the three cooperative forms agree to fp32 rounding (different summation orders)."""
    g = torch.Generator().manual_seed(7 * N + P)
    R = torch.randn(B, N, N, dtype=torch.float32, generator=g)
    A = (R + R.transpose(1, 2)).contiguous().to(dev)
    X = torch.randn(B, P, N, dtype=torch.float32, generator=g).to(dev)
    big = torch.full((B, N + 3, N + 64), float("nan"), dtype=torch.float32, device=dev)
    big[:, :N, :N] = A
    Apad = big[:, :N, :N]
    assert Apad.stride(-2) == N + 64 and not Apad.is_contiguous() and K.symm_wide_ok(Apad, X)
    outs = {}
    for form in (1, 3, 9):
        monkeypatch.setattr(K, "K1SW_OPTS", form)
        outs[form] = K.dense_symm_wide(A, X).clone()
        assert torch.equal(K.dense_symm_wide(A, X), outs[form])
        assert torch.equal(K.dense_symm_wide(Apad, X), outs[form]), form
    ref = torch.matmul(X.double().cpu(), A.double().cpu())
    scale = ref.abs().max().item()
    for form in (1, 3, 9):
        assert (outs[form].cpu().double() - ref).abs().max().item() <= 3e-6 * N ** 0.5 * scale
    assert (outs[9] - outs[3]).abs().max().item() <= 2e-6 * N ** 0.5 * scale


@pytest.mark.parametrize("B,N,P", [(2, 2048, 16), (1, 4096, 9), (3, 1088, 12)])
def test_dense_symm_wide_resident_launch_is_bit_identical(dev, B, N, P, monkeypatch):
    """(r05) the resident form of the cooperative K1sw launch (opts bit 2: workgroups take the super-tiles from a queue,
    three per compute unit) gives the bits of the one-workgroup-per-super-tile launch, repeatedly (the queue word is
    reset by every launch)."""
    g = torch.Generator().manual_seed(13 * N + P)
    R = torch.randn(B, N, N, dtype=torch.float32, generator=g)
    A = (R + R.transpose(-2, -1)).to(dev)
    X = torch.randn(B, P, N, dtype=torch.float32, generator=g).to(dev)
    monkeypatch.setattr(K, "K1SW_OPTS", 3)
    monkeypatch.setattr(K, "K1SW_RESIDENT", False)
    Y0 = K.dense_symm_wide(A, X).clone()
    ref = torch.matmul(X.double().cpu(), A.double().cpu())
    assert (Y0.cpu().double() - ref).abs().max().item() <= 3e-6 * N ** 0.5 * ref.abs().max().item()
    monkeypatch.setattr(K, "K1SW_RESIDENT", True)
    for rep in range(3):
        assert torch.equal(K.dense_symm_wide(A, X), Y0), "resident K1sw launch differs (repetition %d)" % rep
    torch.cuda.synchronize()
