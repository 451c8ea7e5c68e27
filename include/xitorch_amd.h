/* xitorch_amd — C ABI of the MI355X (gfx950) native hot path.
 *
 * One shared library, libxitorch_amd.so, built by `__graft_entry__.build()`
 * with `hipcc --offload-arch=gfx950 -shared -fPIC`.  The reference (xitorch,
 * pure Python) has no FFI; these entry points sit *under* its three Python
 * plug-in contracts (operator / method / functional) and are
 * what a reference-side binding would call (INTEGRATION.md shows the ctypes
 * stubs).  Each function names the reference code it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch types.
 *  - every function returns int: 0 ok, <0 argument/shape error
 *    (XK_ERR_ARG=-1, XK_ERR_UNSUPPORTED=-2), >0 a hipError_t passed through.
 *  - all pointers are DEVICE pointers, borrowed for the duration of the
 *    stream-ordered work; nothing is allocated on behalf of the caller — a
 *    `*_workspace_elems` query precedes calls that need scratch.
 *  - every launch goes to the `stream` argument (a hipStream_t passed as
 *    void*; NULL = the default stream).  No hidden global state: nothing in
 *    the library is process-wide and writable — launch shapes and measurement
 *    switches are ARGUMENTS (0 = the shipped default), the only objects with a
 *    lifetime are the ones the caller creates and hands back (CU-masked streams,
 *    RCCL communicators).  Two threads / two devices of one process share nothing.
 *  - PANEL-MAJOR vectors: a block of P vectors of length N is stored (P, N),
 *    pitch `ld*` between vectors, `s*` between batch members.  This is the
 *    reference's "Fortran order" (N, P) view (_utils/tensor.py:21-32).
 *  - `_f64` / `_f32` suffix = element type.
 */
#ifndef XITORCH_AMD_H
#define XITORCH_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define XK_OK 0
#define XK_ERR_ARG (-1)
#define XK_ERR_UNSUPPORTED (-2)

/* library/ABI version (bumped when a signature changes) */
int xk_abi_version(void);

/* ---- CU-masked stream (no reference counterpart: the reference runs everything on one stream) ----
 * Creates a HIP stream whose kernels may use all but `reserve_cus` compute units of `device`
 * (hipExtStreamCreateWithCUMask).  The eigensolver launches the HBM-bound panel product on it so that the
 * latency-bound small kernels of the other half of the batch find free CUs (linalg.symeig davidson,
 * option overlap).  The caller owns the stream (xk_stream_destroy). */
/* measurement utility: read bytes/pitch_bytes rows of pitch_bytes (16 B multiples, 16 B aligned) of device memory once
 * in the panel kernels' tile walk, no arithmetic; `scratch`: >= 4 KiB of device memory or NULL.  Timed by the caller
 * (bench.py: the practical streaming ceiling of the operator batch) */
int xk_stream_read(const void* src, long bytes, long pitch_bytes, void* scratch, void* stream);
int xk_stream_create_cu_masked(int device, int reserve_cus, void** stream_out);
/* the same with the bits to clear chosen by `pattern`: 0 = the last reserve_cus bits of the linear mask (what
 * xk_stream_create_cu_masked does), 1 = every (units / reserve_cus)-th bit */
int xk_stream_create_cu_masked_pattern(int device, int reserve_cus, int pattern, void** stream_out);
/* measurement utility: `workgroups` one-wave workgroups on `stream`, each spinning `spin_cycles`; hist16[x] = workgroups
 * that ran on XCD x, units128[8 x + w] = bit set of the (SE, SH, CU) ids seen there (device memory, zeroed by the call) */
int xk_probe_xcc(unsigned* hist16, unsigned* units128, int workgroups, int spin_cycles, void* stream);
int xk_stream_destroy(void* stream);

/* ---- K1: batched dense operator-panel product --------------------------------
 * trans=0:  Y[b,c,i] = sum_j A[b,i,j] X[b,c,j]     i<M, j<N
 * trans=1:  Y[b,c,j] = sum_i A[b,i,j] X[b,c,i]
 * Replaces MatrixLinearOperator._mv/_mm/_rmv/_rmm (xitorch/_core/linop.py:692-702)
 * on the eigensolver panel product (xitorch/_impls/linalg/symeig.py:163,221) and
 * the Krylov operator apply (xitorch/_impls/linalg/solve.py:571-572).  Passing a
 * basis (B,k,N) as A computes Gram / Rayleigh blocks (symeig.py:170,
 * _utils/tensor.py:15).
 * A: (B,M,N) row-major, row pitch lda, batch pitch sA (sA=0 broadcasts one operator).
 * ws: scratch for trans=1 of xk_dense_mm_workspace_elems() elements (may be NULL for trans=0).
 * rows_hint: 0 = auto (tuning knob: rows per wave 4/8/16); stagger: 1 = de-phase row sweeps. */
long xk_dense_mm_workspace_elems(int B, int M, int N, int P, int trans);
int xk_dense_mm_f64(const double* A, const double* X, double* Y, double* ws, long ws_elems,
                    int B, int M, int N, int P, long lda, long sA, long ldx, long sX,
                    long ldy, long sY, int trans, int rows_hint, int stagger, void* stream);
int xk_dense_mm_f32(const float* A, const float* X, float* Y, float* ws, long ws_elems,
                    int B, int M, int N, int P, long lda, long sA, long ldx, long sX,
                    long ldy, long sY, int trans, int rows_hint, int stagger, void* stream);

/* ---- K1s: the same product for EXACTLY symmetric storage, reading only the upper triangle ------
 * Y[b,c,:] = A_b X[b,c,:] with A_b == A_b^T bit for bit (the caller's promise): every tile on or
 * above the diagonal is streamed once and feeds both y_I += A_IJ x_J and y_J += A_IJ^T x_I, i.e.
 * about half the HBM traffic of xk_dense_mm.  Used for the eigensolver panel product
 * (xitorch/_impls/linalg/symeig.py:163,221; symeig requires a Hermitian operator, linalg/symeig.py:103).
 * N must be a multiple of the 16 B vector width; ws: xk_dense_symm_workspace_elems(B,N,P,sizeof(T)). */
long xk_dense_symm_workspace_elems(int B, int N, int P, int elem_size);
/* Results are run-to-run bit-identical: the four waves of a workgroup add into the LDS row accumulator in a fixed
 * order (phase rotation with barriers), partial slots are folded in a fixed order.
 * opts (results never depend on the launch shape chosen through it, only — for bits 5, 8..15 and 2/3 — on the fixed
 * summation order of the partial slots): bit 0 plain instead of non-temporal stores of the row / column partials, bit 1
 * plain instead of non-temporal loads in the fold, bits 8..15 column slabs per workgroup run (0 = 1; row partials per row
 * tile = ceil(slabs / run), at most 64); bit 2: 512-row tiles (fp64), bit 3: 1024-row tiles — without either the library
 * picks 512 rows for fp64 launches of fewer than 2200 workgroups (<= 16 operators of order 16384: +1 %);
 * bit 4 (round 5): RESIDENT launch — bits 16..27 workgroups (0 = two per compute unit of the device) take the runs from a
 * queue (the last 64 bytes of `ws`, reset by the call in stream order) until it is empty: bit-identical to the
 * one-workgroup-per-run launch; a second resident launch on another stream moves into the slots this one's tail frees
 * (the eigensolver's two batch groups: 217.6 -> 211.9 ms per BASELINE configs[1] call);
 * bit 5 (round 5, fp64 only): 8-wave workgroups on 2048 x 2048 tiles, one workgroup per compute unit (with bit 4: half
 * the bits-16..27 count) — half the partial-sum bytes; worth it from ~8 tiles per compute unit alone on the device, ~2
 * inside the pipeline (-> 206.7 ms).  The Python host picks bits 4 / 5 per launch (xitorch_amd.kernels.k1s_auto_opts). */
int xk_dense_symm_f64(const double* A, const double* X, double* Y, double* ws, long ws_elems, int B,
                      int N, int P, long lda, long sA, long ldx, long sX, long ldy, long sY, int opts, void* stream);
int xk_dense_symm_f32(const float* A, const float* X, float* Y, float* ws, long ws_elems, int B, int N,
                      int P, long lda, long sA, long ldx, long sX, long ldy, long sY, int opts, void* stream);
/* The two halves of the call above as separate launches (P <= 6): the tile kernel leaves per-slab / per-row-tile
 * partial sums in `ws`, the fold adds them into Y.  The eigensolver's two-group pipeline runs the tile kernels of
 * both groups back to back on one (CU-masked) stream and each fold on its group's own stream, off that
 * critical path; `ws` must stay untouched between the two calls, and both take the same `opts`. */
int xk_dense_symm_tiles_f64(const double* A, const double* X, double* ws, long ws_elems, int B, int N, int P,
                            long lda, long sA, long ldx, long sX, int opts, void* stream);
int xk_dense_symm_tiles_f32(const float* A, const float* X, float* ws, long ws_elems, int B, int N, int P,
                            long lda, long sA, long ldx, long sX, int opts, void* stream);
int xk_dense_symm_fold_f64(double* Y, const double* ws, long ws_elems, int B, int N, int P, long ldy, long sY,
                           int opts, void* stream);
int xk_dense_symm_fold_f32(float* Y, const float* ws, long ws_elems, int B, int N, int P, long ldy, long sY,
                           int opts, void* stream);

/* ---- K1sw: the same product (exactly symmetric storage, upper triangle streamed once) for WIDE panels on the matrix
 * cores: 9 <= P <= 16 columns, fp32 — BASELINE configs[4]'s 16-column eigen-block (xitorch/_core/linop.py:695-696 inside
 * _impls/linalg/symeig.py:163,221).  Every 64-row x 128-byte sub-tile on or above the diagonal is loaded once, turned
 * through LDS per wave and feeds y_I += A_IJ x_J and y_J += A_IJ^T x_I through v_mfma_f32_16x16x4_f32; row / column sums
 * leave as partials in `ws` (one wave = one 512-row x 256-column tile, no atomics, no block barriers) and a fold adds the
 * slots that exist in fixed order: run-to-run bit-identical.  N must be a multiple of 64; the 64 x 64 diagonal blocks are
 * read whole.  ws: xk_dense_symm_wide_workspace_elems(B, N).  _tiles / _fold: the two launches separately (the
 * eigensolver's two-group pipeline), `ws` untouched in between.  opts: 0 = one wave per tile (above); 1 = the
 * workgroup-cooperative form (three waves per SIMD, four 128-column strips per workgroup sharing their row sums
 * through LDS; + 2 = raised wave priority around its MFMA block; 3 is what the Python host passes: 3.58 ms against 3.97 ms
 * for 8 x 32768^2, profiles/r04_k1sw_forms.jsonl); + 4 (round 5, with 1): resident launch — bits 16..27 workgroups (0 =
 * three per compute unit) take the super-tiles from a queue in the last 64 bytes of `ws` (reset by the call), bit-identical
 * to opts 1 / 3; + 8 (round 6, with 1, not with 2 / 4; 9 is what the Python host passes): the column part straight from the
 * load registers (a load = 4 rows x 256 B in the column part's B-operand layout), the row part through ds_write_b128 /
 * ds_read_b128 one block behind, a ring of four 16 x 64 blocks = 16 KB in flight per wave, two waves per SIMD — same
 * work split, partial slots and fold as opts 1: 3.62 -> 2.94 ms for 8 x 32768^2 (profiles/r06_k1sw_forms.json).
 * _tiles and _fold of one product take the same opts. */
long xk_dense_symm_wide_workspace_elems(int B, int N);
int xk_dense_symm_wide_f32(const float* A, const float* X, float* Y, float* ws, long ws_elems, int B, int N, int P,
                           long lda, long sA, long ldx, long sX, long ldy, long sY, int opts, void* stream);
int xk_dense_symm_wide_tiles_f32(const float* A, const float* X, float* ws, long ws_elems, int B, int N, int P,
                                 long lda, long sA, long ldx, long sX, int opts, void* stream);
int xk_dense_symm_wide_fold_f32(float* Y, const float* ws, long ws_elems, int B, int N, int P, long ldy, long sY,
                                int opts, void* stream);

/* ---- K1w: wide panels on the matrix cores (MFMA) --------------------------------------------
 * Y[b,c,n] = sum_i A[b,i,n] Xrm[b,i,c]  (= A^T X; = A X for a Hermitian operator), c < P <= 32, in ONE
 * pass over A (v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64, exact FMA chains).  Xrm is ROW-major
 * (B, M, PP) with PP = xk_dense_wide_padded_width(P) zero-padded columns; Y panel-major (B, P, N).
 * Requires N % 128 == 0 (fp32) / N % 32 == 0 (fp64).  Many-column torch.matmul(mat, x) of
 * MatrixLinearOperator._mm/_rmm (xitorch/_core/linop.py:695-702), e.g. solve with ncols = 50
 * (benchmarks/benchmarks_solve.py:11-15). */
long xk_dense_wide_workspace_elems(int B, int M, int N, int P, int elem_size);
int xk_dense_wide_padded_width(int P, int elem_size);
int xk_dense_wide_f32(const float* A, const float* Xrm, float* Y, float* ws, long ws_elems, int B, int M,
                      int N, int P, long lda, long sA, long ldxr, long sXr, long ldy, long sY, void* stream);
int xk_dense_wide_f64(const double* A, const double* Xrm, double* Y, double* ws, long ws_elems, int B, int M,
                      int N, int P, long lda, long sA, long ldxr, long sXr, long ldy, long sY, void* stream);

/* ---- K1wr: wide panels in the ROW orientation (no transposed copy, no MFMA) -----------------------
 * Y[b,c,i] = sum_j A[b,i,j] X[b,c,j], any P (32 columns per pass over A): `torch.matmul(mat, x)` of a
 * non-Hermitian MatrixLinearOperator with many right-hand sides (xitorch/_core/linop.py:695-696;
 * benchmarks/benchmarks_solve.py:11-15; the BiCGStab / GMRES applies of a multi-RHS solve,
 * _impls/linalg/solve.py:278,284).  Coalesced 64-row sub-tiles are turned through LDS so that every lane walks
 * along its own row with wave-uniform (scalar) panel values: no cross-lane reduction.  X, Y panel-major;
 * requires N, lda, ldx multiples of the 16 B vector width (else XK_ERR_UNSUPPORTED: use xk_dense_mm).
 * ws: xk_dense_rows_wide_workspace_elems() elements (0 when the contraction is not split). */
long xk_dense_rows_wide_workspace_elems(int B, int M, int N, int P, int elem_size);
int xk_dense_rows_wide_f64(const double* A, const double* X, double* Y, double* ws, long ws_elems, int B, int M,
                           int N, int P, long lda, long sA, long ldx, long sX, long ldy, long sY, void* stream);
int xk_dense_rows_wide_f32(const float* A, const float* X, float* Y, float* ws, long ws_elems, int B, int M, int N,
                           int P, long lda, long sA, long ldx, long sX, long ldy, long sY, void* stream);

/* ---- basis maintenance of the block eigensolver (K2/K4/K5/K6) ------------------------------
 * Panels must be PADDED: pitch a multiple of 16 B and >= N rounded up to 16 B, pads zero.
 *
 * xk_lincomb:  Out[b,c,:] = beta*Out[b,c,:] + alpha * sum_{a<k} C[b,a,c] V[b,a,:]
 *   C element (b,a,c) at C[b*sC + a*sCa + c*sCc].  Replaces the Ritz rotations `V @ Y`
 *   (xitorch/_impls/linalg/symeig.py:178,181) and the projection step standing in for the full
 *   CholeskyQR `tallqr` (xitorch/_utils/tensor.py:15-18). */
int xk_lincomb_f64(const double* V, const double* C, double* Out, int B, int k, int N, int P,
                   long ldv, long sV, long sC, long sCa, long sCc, long ldo, long sO,
                   double alpha, double beta, void* stream);
int xk_lincomb_f32(const float* V, const float* C, float* Out, int B, int k, int N, int P,
                   long ldv, long sV, long sC, long sCa, long sCc, long ldo, long sO,
                   double alpha, double beta, void* stream);
/* xk_ritz_residual (fused K4/K5, symeig.py:178-188,207):
 *   X[b,c,:] = sum_a Y[b,a,c] V[b,a,:];  AX likewise from AV;  Tn[b,c,:] = -(AX - lam[b,c] X);
 *   rmax[b] = max(rmax[b], max |AX - lam X|)  (caller zeroes rmax; NaN is reported as +inf). */
int xk_ritz_residual_f64(const double* V, const double* AV, const double* Y, const double* lam,
                         double* X, double* Tn, double* rmax, int B, int k, int N, int P,
                         long ldv, long sV, long ldav, long sAV, long sY, long sYa, long sYc,
                         long sLam, long ldx, long sX, long ldt, long sT, void* stream);
int xk_ritz_residual_f32(const float* V, const float* AV, const float* Y, const float* lam,
                         float* X, float* Tn, float* rmax, int B, int k, int N, int P,
                         long ldv, long sV, long ldav, long sAV, long sY, long sYa, long sYc,
                         long sLam, long ldx, long sX, long ldt, long sT, void* stream);
/* xk_panel_chol: G[b] (P x P, pitch ldg) = R^T R, W[b] = R^-1 compact (B,P,P) row-major; info[b] =
 *   1+index of the first non-positive pivot, 0 if none (tensor.py:16-17; P <= 32). */
int xk_panel_chol_f64(const double* G, double* W, int* info, int B, int P, long ldg, long sG, void* stream);
int xk_panel_chol_f32(const float* G, float* W, int* info, int B, int P, long ldg, long sG, void* stream);
/* xk_panel_transform: in place Tp[b,c,:] <- sum_{a<=c} W[b,a,c] Tp[b,a,:]  (tensor.py:18). */
int xk_panel_transform_f64(double* Tp, const double* W, int B, int P, int N, long ldt, long sT, void* stream);
int xk_panel_transform_f32(float* Tp, const float* W, int B, int P, int N, long ldt, long sT, void* stream);

/* ---- Davidson diagonal preconditioner (extension; the reference's davidson has none, `t = -resid`,
 * xitorch/_impls/linalg/symeig.py:206-207) ------------------------------------------------------
 * t[b,c,n] /= (d[b,n] - lam[b,c]*m[b,n]) with |denominator| >= floor (sign kept); m == NULL: identity.
 * t: (B,P,ld) panel, d/m: (B,N) with batch strides sD/sM (0 broadcasts), lam: (B,>=P) unit stride. */
int xk_diag_precond_f64(double* t, const double* d, const double* m, const double* lam, int B, int N, int P,
                        long ldt, long sT, long sD, long sM, long sLam, double floor_, void* stream);
int xk_diag_precond_f32(float* t, const float* d, const float* m, const float* lam, int B, int N, int P,
                        long ldt, long sT, long sD, long sM, long sLam, double floor_, void* stream);

/* ---- group status of a Davidson step in one launch (native_eig._Group.small; the reference reads
 * `resid.abs().max()` on the host, symeig.py:190-197): status[0] = max_b rmax[b] (NaN if any is NaN),
 * status[1] = max_b info[b] (panel Cholesky flags), status[2] = max_b flag[b] (K3t self-check; flag may be NULL);
 * orth (B, may be NULL): the a-posteriori guard values left by xk_ritz_guard: status[4] = max_b orth[b] (NaN counts as
 * infinite), orth is re-zeroed; status then has 5 doubles (status[3], the chain's condition estimate, is not written) */
int xk_group_status_f64(const double* rmax, const int* info, const int* flag, double* orth, double* status, int B,
                        void* stream);
int xk_group_status_f32(const float* rmax, const int* info, const int* flag, float* orth, double* status, int B,
                        void* stream);

/* ---- K3: batched small symmetric eigensolver (LDS-resident parallel Jacobi) -----------------
 * Lowest (uppest=0) / uppermost (uppest=1) p eigenpairs of B symmetric k x k matrices (lower
 * triangle read, pitch ldt, batch pitch sT), eigenvalues ascending: lam (B,p), Y (B,p,k) with
 * Y[b,c,:] the c-th eigenvector.  Replaces torch.linalg.eigh + _take_eigpairs
 * (symeig.py:174-175, 255-264).  k <= 128, p <= 16.  ws: xk_small_eigh_workspace_elems elements. */
long xk_small_eigh_workspace_elems(int B, int k, int max_sweeps);
int xk_small_eigh_f64(const double* T, double* lam, double* Y, double* ws, long ws_elems, int* sweeps,
                      int B, int k, int p, int uppest, int max_sweeps, long ldt, long sT, void* stream);
int xk_small_eigh_f32(const float* T, float* lam, float* Y, float* ws, long ws_elems, int* sweeps,
                      int B, int k, int p, int uppest, int max_sweeps, long ldt, long sT, void* stream);

/* ---- K3t: the same p eigenpairs by Householder tridiagonalisation + bisection + inverse iteration ------
 * (the LAPACK dsyevx route), one workgroup per matrix, LDS-resident: the O(k^3) work is ONE tridiagonalisation
 * instead of ~8 Jacobi sweeps over the whole matrix, which is what the Davidson loop — interested in p << k
 * pairs only (symeig.py:174-175, 255-264) — needs.  Same outputs as xk_small_eigh_*; info[b] != 0 flags a
 * batch member whose result failed the kernel's residual / orthogonality check (caller re-runs on the Jacobi
 * kernel).  Returns XK_ERR_UNSUPPORTED when xk_small_eigh_tri_lds_bytes(k, p, elem_size) exceeds 160 KiB.
 * threads: workgroup size, a multiple of 64 in [64, 1024], 0 = the measured default (512); profile: NULL, or a device
 * buffer of >= 8 int64 that receives the cycle counter at the phase boundaries of block 0 (measurements). */
long xk_small_eigh_tri_lds_bytes(int k, int p, int elem_size);
int xk_small_eigh_tri_f64(const double* T, double* lam, double* Y, int* info, int B, int k, int p, int uppest,
                          long ldt, long sT, int threads, long long* profile, void* stream);
int xk_small_eigh_tri_f32(const float* T, float* lam, float* Y, int* info, int B, int k, int p, int uppest, long ldt,
                          long sT, int threads, long long* profile, void* stream);

/* ---- banded operator (DIA storage) -------------------------------------------------------------
 * band (B, 2*hb+1, N), band[b,d,i] = A_b[i, i+d-hb]; entries outside the matrix are ignored.
 * trans=0: Y[b,c,i] = sum_d band[b,d,i] X[b,c,i+d-hb];  trans=1: the transposed operator.
 * sBand = 0 broadcasts one operator over the batch.  Operator of BASELINE configs[2]; in the
 * reference a user writes it as a custom `_mv` (cf. ALarge, xitorch/_tests/test_linop_fcns.py:129-150). */
int xk_banded_mm_f64(const double* band, const double* X, double* Y, int B, int N, int hb, int C,
                     long sBand, long ldx, long sX, long ldy, long sY, int trans, void* stream);
int xk_banded_mm_f32(const float* band, const float* X, float* Y, int B, int N, int hb, int C,
                     long sBand, long ldx, long sX, long ldy, long sY, int trans, void* stream);

/* ---- operator gradients of the implicit backward passes (streaming writes) -----------------------
 * The backward of solve / symeig / rootfinder ends with a VJP through the operator apply
 * (`loss = -A.mm(x)`; `autograd.grad(loss, params, v)`: xitorch/linalg/solve.py:188-195,
 * linalg/symeig.py:374-379, optimize/rootfinder.py:352-362); in the reference that is torch's matmul / slice
 * backward.  U, W panel-major (B, C, N), pitches ld*, batch pitches s*.
 * xk_banded_grad:  G[b,d,i] (+)= sum_c U[b,c,i] W[b,c,i+d-hb]   (DIA band gradient; trans apply: swap U and W)
 * xk_dense_outer:  G[b,i,j] (+)= sum_c U[b,c,i] W[b,c,j]        (i < M, j < N, row pitch ldg)
 * accumulate = 1 adds into G. */
int xk_banded_grad_f64(const double* U, const double* W, double* G, int B, int N, int hb, int C, long ldu,
                       long sU, long ldw, long sW, long sG, int accumulate, void* stream);
int xk_banded_grad_f32(const float* U, const float* W, float* G, int B, int N, int hb, int C, long ldu, long sU,
                       long ldw, long sW, long sG, int accumulate, void* stream);
int xk_dense_outer_f64(const double* U, const double* W, double* G, int B, int M, int N, int C, long ldu,
                       long sU, long ldw, long sW, long ldg, long sG, int accumulate, void* stream);
int xk_dense_outer_f32(const float* U, const float* W, float* G, int B, int M, int N, int C, long ldu, long sU,
                       long ldw, long sW, long ldg, long sG, int accumulate, void* stream);

/* ---- fused Krylov-loop kernels (K7-K9, K11; xitorch/_impls/linalg/solve.py:143-180, 272-314) ----
 * Vectors are (S, ld) arrays: S systems (batch member x column), pitch ld (16 B multiple, pads 0).
 * P* are partial-sum buffers (S, xk_kry_max_partials()); nblk <= that many blocks per system.
 * Per-system scalars (rho, alpha, omega) are device arrays of S elements.  eps = the reference's
 * `_safedenom` replacement of exact zeros (solve.py:437-439). */
int xk_kry_max_partials(void);
/* P1[s] <- partials of <x1,y1>, P2[s] (optional) <- <x2,y2>; if E != NULL first y1 -= E[s]*shiftz */
int xk_kry_dots_f64(const double* x1, double* y1, const double* x2, const double* y2, const double* shiftz,
                    const double* E, double* P1, double* P2, int S, int N, long ld, int nblk, void* stream);
int xk_kry_dots_f32(const float* x1, float* y1, const float* x2, const float* y2, const float* shiftz,
                    const float* E, float* P1, float* P2, int S, int N, long ld, int nblk, void* stream);
/* p = r + beta (p - omega v), beta = rho_new/safe(rho_old) * alpha/safe(omega); rho_store <- rho_new
 * (rho_old and rho_store must be different arrays); first=1: p = r  (solve.py:273-276) */
int xk_bicg_p_f64(const double* r, double* p, const double* v, const double* Prho_new, const double* rho_old,
                  const double* alpha, const double* omega, double* rho_store, int S, int N, long ld,
                  int nblk, double eps, int first, void* stream);
int xk_bicg_p_f32(const float* r, float* p, const float* v, const float* Prho_new, const float* rho_old,
                  const float* alpha, const float* omega, float* rho_store, int S, int N, long ld,
                  int nblk, double eps, int first, void* stream);
/* s = r - alpha v, alpha = rho / safe(<r0,v>)  (solve.py:279,282) */
int xk_bicg_s_f64(const double* r, const double* v, double* s, const double* rho, const double* Pr0v,
                  double* alpha_store, int S, int N, long ld, int nblk, double eps, void* stream);
int xk_bicg_s_f32(const float* r, const float* v, float* s, const float* rho, const float* Pr0v,
                  float* alpha_store, int S, int N, long ld, int nblk, double eps, void* stream);
/* omega = <Kt,Ks>/safe(<Kt,Kt>); xout = x + alpha*yd + omega*zd; unless skip_r: r = s - omega t and
 * Prr <- |r|^2, Prho <- <r0,r>  (solve.py:286-297) */
int xk_bicg_final_f64(const double* x, double* xout, const double* yd, const double* zd, const double* s,
                      const double* t, double* r, const double* r0, const double* alpha, const double* Pts,
                      const double* Ptt, double* omega_store, double* Prr, double* Prho, int S, int N,
                      long ld, int nblk, double eps, int skip_r, void* stream);
int xk_bicg_final_f32(const float* x, float* xout, const float* yd, const float* zd, const float* s,
                      const float* t, float* r, const float* r0, const float* alpha, const float* Pts,
                      const float* Ptt, float* omega_store, float* Prr, float* Prho, int S, int N,
                      long ld, int nblk, double eps, int skip_r, void* stream);
/* r = b - y; Prr <- |r|^2; Prho (optional) <- <r0,r> (or |r|^2 when r0 is NULL)  (solve.py:148-149, 290-291) */
int xk_kry_resid_f64(const double* b, const double* y, double* r, const double* r0, double* Prr,
                     double* Prho, int S, int N, long ld, int nblk, void* stream);
int xk_kry_resid_f32(const float* b, const float* y, float* r, const float* r0, float* Prr, float* Prho,
                     int S, int N, long ld, int nblk, void* stream);
/* alpha = <r,z>/safe(<p,Ap>); xout = x + alpha p; unless skip_r: r -= alpha Ap, Prr <- |r|^2  (solve.py:144-155) */
int xk_cg_update_f64(const double* x, double* xout, const double* p, const double* Ap, double* r,
                     const double* Prz, const double* PpAp, double* Prr, int S, int N, long ld, int nblk,
                     double eps, int skip_r, void* stream);
int xk_cg_update_f32(const float* x, float* xout, const float* p, const float* Ap, float* r, const float* Prz,
                     const float* PpAp, float* Prr, int S, int N, long ld, int nblk, double eps, int skip_r,
                     void* stream);
/* p = z + beta p, beta = <r,z>_new / safe(<r,z>_old)  (solve.py:171-173) */
int xk_cg_p_f64(const double* z, double* p, const double* Prz_new, const double* Prz_old, int S, int N,
                long ld, int nblk, double eps, void* stream);
int xk_cg_p_f32(const float* z, float* p, const float* Prz_new, const float* Prz_old, int S, int N, long ld,
                int nblk, double eps, void* stream);
/* rnorm[s] = sqrt(sum Prr[s,:]); status[0] = max_s rnorm (NaN -> +inf), status[1] = #{s: !(rnorm < stop[s])}
 * (the stopping test of solve.py:157,166,301,310 — read back once per iteration) */
int xk_kry_status_f64(const double* Prr, const double* stop, double* rnorm, double* status, int S, int nblk,
                      void* stream);
int xk_kry_status_f32(const float* Prr, const float* stop, float* rnorm, double* status, int S, int nblk,
                      void* stream);

/* ---- K3g: the same p wanted eigenpairs for Rayleigh-Ritz matrices of order 129 .. 1536 (xk_eigh_big.hip) ---------
 * torch.linalg.eigh + _take_eigpairs (symeig.py:174-175, 255-264) once the un-restarted basis (symeig.py:132-135) has
 * outgrown the LDS-resident kernels: Householder tridiagonalisation of the upper triangle of a work copy in global
 * memory, k - 1 launches (one per step, look-ahead form) over several workgroups per matrix; then bisection / inverse
 * iteration / self-check in LDS like K3t and the back-transformation from the reflectors parked in the work copy, one
 * workgroup per matrix.  The whole sequence is enqueued on `stream` by one call.  Only the lower triangle of T is read.
 * ws: xk_small_eigh_big_workspace_elems(B, k, wg) elements (work copies + the steps' hand-over blocks).
 * wg: workgroups per matrix of the step kernels, 0 = automatic (by batch and order), 1 .. 32; threads: 0 = 512, or 256.
 * lam (B, p) ascending, Y (B, p, k) eigenvectors, info[b] != 0 -> redo that call on the library solver.
 * xk_small_eigh_big_batch(k, p, elem_size): shifts factorised at a time (> 0) when the problem fits the 160 KiB of
 * LDS, 0 when it does not.  8 <= k <= 1536 (r06: beyond 1024 with 24 column slots per lane; r05: beyond 768 with 16 column slots per lane in 256-thread workgroups, or the two-stage form below where its band fits: fp32), p <= 256 (r06; 64 before) (also the solver for MORE THAN 16 wanted pairs at any order: wide
 * eigen-blocks, thick restarts; the batch of vectors in work lives in LDS, finished ones in the rows of Y).
 * algo: 1 = the form above; 2 = the TWO-STAGE form (xk_eigh_band.hip: dense -> band of 16 sub-diagonals by block
 * reflectors, two launches per 16 columns; band -> tridiagonal by bulge chasing in LDS, one workgroup per matrix, sweeps
 * pipelined three steps apart; eigenvectors back through both stages, one workgroup per vector), XK_ERR_UNSUPPORTED when
 * the band of order k does not fit the LDS (fp64: k <= 614); 3 = the PERSISTENT form (r06, xk_eigh_persist.hip: the whole
 * Householder reduction in ONE launch of one workgroup per matrix, the trailing block of order <= 256 (fp64) / 384 (fp32)
 * resident in the registers of its eight waves, one barrier per step; larger orders start with k - 256 / k - 384 step
 * launches of form 1 and hand over); 0 = the measured choice (form 3 up to 64 .. 128 orders beyond its register limit,
 * by batch; the two-stage form beyond where it fits; else form 1).  Every form is bit-reproducible; the forms differ in
 * rounding. */
int xk_small_eigh_big_batch(int k, int p, int elem_size);
long xk_small_eigh_big_workspace_elems(int B, int k, int wg);
int xk_small_eigh_big_f64(const double* T, double* lam, double* Y, double* ws, long ws_elems, int* info, int B, int k,
                          int p, int uppest, long ldt, long sT, int wg, int threads, int algo, void* stream);
int xk_small_eigh_big_f32(const float* T, float* lam, float* Y, float* ws, long ws_elems, int* info, int B, int k,
                          int p, int uppest, long ldt, long sT, int wg, int threads, int algo, void* stream);

/* ---- Davidson chain: one C call per stage of an iteration (xitorch/_impls/linalg/symeig.py:160-223) -------------
 * The stages between two operator-panel products are two to eight small launches each; issued from C++ they cost a
 * few microseconds of host time instead of an interpreter round trip per launch (what bounds small per-GPU batches).
 * xk_davidson_ritz: xk_ritz_residual + the group status {max_b rmax (NaN-propagating), max_b info, max_b flag} as
 *   three doubles (+ a fourth, max_b cond, when cond is given; cond is re-zeroed; + a fifth, max_b orth, when orth is
 *   given: xk_ritz_guard of the Ritz block X just formed, folded and re-zeroed); rmax is left zeroed for the next
 *   step (symeig.py:178-197).  flag, cond and orth may be NULL.  Gs: scratch >= B*P*P and ws:
 *   xk_dense_mm_workspace_elems(B, P, N, P, 0), both only read when orth is given and P > 8.
 * xk_ritz_guard: the a-posteriori check that stands where the reference's whole-basis CholeskyQR makes a wrong
 *   answer impossible (tallqr of [V, t] every iteration, _utils/tensor.py:8-19, symeig.py:207-223): orth[b] =
 *   max(orth[b], max_{c,d} |<X_c, (M X)_d> - delta_cd|) over the P rows of the Ritz block X (B, P, ldx) and M X
 *   (B, P, ldm; NULL: X itself).  Two copies of one eigenpair in a block give 1, a healthy block rounding level; the
 *   Davidson driver never returns (and never restarts from) a block above its threshold.  P <= 8: one kernel, the
 *   panels are read once; wider: Gram on K1 into Gs (>= B*P*P) + one wave per member.
 * xk_davidson_orth: rows [k0, k0+q) of the basis V (B, cap, ldv) against rows [0, k0): block Gram-Schmidt
 *   (C[b,c,a] = <V_a, t_c>, t_c -= sum_a C V_a) and CholeskyQR of the q rows — for q <= 8 in ONE kernel (Gram,
 *   Cholesky, inverse, transform; one workgroup per batch member) — i.e. tallqr of [V, t] restricted to the new block
 *   (_utils/tensor.py:8-19, symeig.py:207-220).  passes = 0: CholeskyQR only; 1: projection, CholeskyQR; >= 2:
 *   [projection, CholeskyQR] per pass with the FIRST CholeskyQR shifted (Gram + 11 (N q + q (q + 1)) u trace I): the
 *   order that keeps nearly dependent panels (Gram spectrum over 15 decades) positive definite and orthogonal to V.
 *   cond (B, may be NULL; q <= 8 only): cond[b] = max(cond[b], (largest / smallest pivot)^2 of the raw panel's
 *   CholeskyQR) — the squared condition estimate by which the Davidson driver decides when ONE pass stops being safe.  C: scratch >= B*q*max(k0,q), W: scratch B*q*q, info[b] sticky
 *   index+1 of a non-positive pivot, ws: xk_dense_mm_workspace_elems(B, cap, N, q, 0).  Any q: panels wider than
 *   32 are taken 32 rows at a time, each chunk against everything before it (twice), then among itself.
 * xk_davidson_extend_t: Tn[b,c,a] = <V_a, (AV)_{k0+c}> for a < k0+q, written to T[b, k0+c, a] and mirrored to
 *   T[b, a, k0+c] (symeig.py:170 restricted to the new rows / columns).  Tn: scratch >= B*q*(k0+q). */
int xk_davidson_ritz_f64(const double* V, const double* AV, const double* Y, const double* lam, double* X, double* Tn,
                         double* rmax, const int* info, const int* flag, double* cond, double* orth, double* status,
                         int B, int k, int N, int P, long ldv, long sV, long ldav, long sAV, long sY, long sYa,
                         long sYc, long sLam, long ldx, long sX, long ldt, long sT, double* Gs, long gs_elems,
                         double* ws, long ws_elems, void* stream);
int xk_davidson_ritz_f32(const float* V, const float* AV, const float* Y, const float* lam, float* X, float* Tn,
                         float* rmax, const int* info, const int* flag, float* cond, float* orth, double* status, int B,
                         int k, int N, int P, long ldv, long sV, long ldav, long sAV, long sY, long sYa, long sYc,
                         long sLam, long ldx, long sX, long ldt, long sT, float* Gs, long gs_elems, float* ws,
                         long ws_elems, void* stream);
int xk_ritz_guard_f64(const double* X, const double* MX, double* orth, int B, int N, int P, long ldx, long sX,
                      long ldm, long sM, double* Gs, long gs_elems, double* ws, long ws_elems, void* stream);
int xk_ritz_guard_f32(const float* X, const float* MX, float* orth, int B, int N, int P, long ldx, long sX, long ldm,
                      long sM, float* Gs, long gs_elems, float* ws, long ws_elems, void* stream);
int xk_davidson_orth_f64(double* V, int B, int N, int k0, int q, long ldv, long sV, double* C, double* W, int* info,
                         double* cond, double* ws, long ws_elems, int passes, void* stream);
int xk_davidson_orth_f32(float* V, int B, int N, int k0, int q, long ldv, long sV, float* C, float* W, int* info,
                         float* cond, float* ws, long ws_elems, int passes, void* stream);
int xk_davidson_extend_t_f64(const double* V, const double* AV, double* T, double* Tn, int B, int N, int k0, int q,
                             long ldv, long sV, long ldav, long sAV, long ldt, long sT, double* ws, long ws_elems,
                             void* stream);
int xk_davidson_extend_t_f32(const float* V, const float* AV, float* T, float* Tn, int B, int N, int k0, int q,
                             long ldv, long sV, long ldav, long sAV, long ldt, long sT, float* ws, long ws_elems,
                             void* stream);

/* ---- GMRES: per-system Hessenberg / Givens state on the device (xitorch/_impls/linalg/solve.py:326-433) ----------
 * The reference fills one Hessenberg column per iteration by modified Gram-Schmidt (:390-394) and solves the
 * (k+1) x k least-squares problem from scratch with torch.linalg.lstsq (:403) on every pass of its Python loop.
 * Here the S = batch x columns systems keep their state on the device, always in double:
 *   R  (S, cap+1, cap) row-major per system: the rotated (triangularised) Hessenberg, R[i][j] meaningful for i <= j
 *   cs, sn (S, cap): Givens rotations;  g (S, cap+1): rotated right-hand side, g[:,0] = |r0| set by the caller.
 * xk_gmres_step: column k from the two Gram passes of the CGS2 orthogonalisation — c1[s, 0..k] = <q_j, w>,
 *   c2n[s, 0..k] = <q_j, w1> with w1 = w - Q c1 and c2n[s, k+1] = <w1, w1> (strides sc1 / sc2 between systems) —
 *   h[j,k] = c1 + c2, h[k+1,k] = sqrt(<w1,w1> - |c2|^2); replays rotations 0..k-1, creates rotation k, updates g;
 *   inv_hn[s] = 1 / h[k+1,k] (0 on breakdown); est2[s * xk_kry_max_partials()] = g[k+1]^2, i.e. a one-partial |r|^2
 *   array for xk_kry_status: the least-squares residual norm, equal to :414-415's explicit one in exact arithmetic.
 * xk_gmres_finish: basis row k+1 (holding w1) <- (w1 - sum_{j<=k} c2n[j] q_j) * inv_hn   (:392,396-398).
 * xk_gmres_solve: y[s, 0..kd) = R^-1 g (back substitution; zero pivots give 0) — what :403 returns; kd <= 8192. */
int xk_gmres_step_f64(const double* c1, long sc1, const double* c2n, long sc2, int k, int cap, double* R, double* cs,
                      double* sn, double* g, double* inv_hn, double* est2, int S, void* stream);
int xk_gmres_step_f32(const float* c1, long sc1, const float* c2n, long sc2, int k, int cap, double* R, double* cs,
                      double* sn, double* g, float* inv_hn, float* est2, int S, void* stream);
int xk_gmres_finish_f64(double* Q, const double* c2n, long sc2, const double* inv_hn, int S, int N, int k, long ldq,
                        long sQ, void* stream);
int xk_gmres_finish_f32(float* Q, const float* c2n, long sc2, const float* inv_hn, int S, int N, int k, long ldq,
                        long sQ, void* stream);
int xk_gmres_solve_f64(const double* R, const double* g, double* y, long sy, int S, int kd, int cap, void* stream);
int xk_gmres_solve_f32(const double* R, const double* g, float* y, long sy, int S, int kd, int cap, void* stream);

/* ---- the same fused Krylov kernels for COMPLEX systems (complex64 = _c64, complex128 = _c128) ----------
 * The reference runs cg / bicgstab on complex operators with conjugated inner products
 * (xitorch/_impls/linalg/solve.py:441-445; _tests/test_linop_fcns.py:474-524, 631-676).  Pointers address
 * INTERLEAVED (re, im) storage (torch.view_as_real of a complex panel); N and ld count complex elements; the
 * per-system scalars (rho, alpha, omega, E) are (S, 2) arrays; partials of complex products are
 * (S, xk_kry_max_partials(), 2); |r|^2 partials (Prr) stay real (S, xk_kry_max_partials()) and are consumed by
 * xk_kry_status_f32/_f64.  <x, y> = sum conj(x) y; xk_kry_dots_c*: conj1 = 1 stores conj(<x1,y1>) = <y1,x1>
 * in P1 (BiCGStab's omega = <t, s> / <t, t>, solve.py:286, with the shift applied to y1 = t). */
int xk_kry_dots_c128(const double* x1, double* y1, const double* x2, const double* y2, const double* shiftz,
                     const double* E, double* P1, double* P2, int S, int N, long ld, int nblk, int conj1, void* stream);
int xk_kry_dots_c64(const float* x1, float* y1, const float* x2, const float* y2, const float* shiftz, const float* E,
                    float* P1, float* P2, int S, int N, long ld, int nblk, int conj1, void* stream);
int xk_bicg_p_c128(const double* r, double* p, const double* v, const double* Prho_new, const double* rho_old,
                   const double* alpha, const double* omega, double* rho_store, int S, int N, long ld, int nblk,
                   double eps, int first, void* stream);
int xk_bicg_p_c64(const float* r, float* p, const float* v, const float* Prho_new, const float* rho_old,
                  const float* alpha, const float* omega, float* rho_store, int S, int N, long ld, int nblk, double eps,
                  int first, void* stream);
int xk_bicg_s_c128(const double* r, const double* v, double* s, const double* rho, const double* Pr0v,
                   double* alpha_store, int S, int N, long ld, int nblk, double eps, void* stream);
int xk_bicg_s_c64(const float* r, const float* v, float* s, const float* rho, const float* Pr0v, float* alpha_store,
                  int S, int N, long ld, int nblk, double eps, void* stream);
int xk_bicg_final_c128(const double* x, double* xout, const double* yd, const double* zd, const double* s,
                       const double* t, double* r, const double* r0, const double* alpha, const double* Pts,
                       const double* Ptt, double* omega_store, double* Prr, double* Prho, int S, int N, long ld,
                       int nblk, double eps, int skip_r, void* stream);
int xk_bicg_final_c64(const float* x, float* xout, const float* yd, const float* zd, const float* s, const float* t,
                      float* r, const float* r0, const float* alpha, const float* Pts, const float* Ptt,
                      float* omega_store, float* Prr, float* Prho, int S, int N, long ld, int nblk, double eps,
                      int skip_r, void* stream);
int xk_kry_resid_c128(const double* b, const double* y, double* r, const double* r0, double* Prr, double* Prho,
                      int S, int N, long ld, int nblk, void* stream);
int xk_kry_resid_c64(const float* b, const float* y, float* r, const float* r0, float* Prr, float* Prho, int S, int N,
                     long ld, int nblk, void* stream);
int xk_cg_update_c128(const double* x, double* xout, const double* p, const double* Ap, double* r, const double* Prz,
                      const double* PpAp, double* Prr, int S, int N, long ld, int nblk, double eps, int skip_r,
                      void* stream);
int xk_cg_update_c64(const float* x, float* xout, const float* p, const float* Ap, float* r, const float* Prz,
                     const float* PpAp, float* Prr, int S, int N, long ld, int nblk, double eps, int skip_r,
                     void* stream);
int xk_cg_p_c128(const double* z, double* p, const double* Prz_new, const double* Prz_old, int S, int N, long ld,
                 int nblk, double eps, void* stream);
int xk_cg_p_c64(const float* z, float* p, const float* Prz_new, const float* Prz_old, int S, int N, long ld, int nblk,
                double eps, void* stream);

/* ---- fused BLAS-1 of the quasi-Newton (Broyden) driver -------------------------------------------------
 * The reference's _nonlin_solver / LowRankMatrix (xitorch/_impls/optimize/root/rootsolver.py:96-143,
 * _jacobian.py:99-119,172-189) run torch.dot / .norm() / axpy chains on one flat length-L vector with a host sync
 * after almost every one.
 * xk_vec_dots: out[i] = <a_i, b_i>, i < npairs <= 4, in one streaming pass + a fixed-order device fold (double
 *   results; `partials`: scratch of xk_vec_dots_workspace_elems() doubles).  Replaces the y.norm(), dx.norm(),
 *   x.norm(), torch.dot calls of rootsolver.py:100,113,286-290,375-380 — the driver reads them with ONE sync.
 * xk_broyden_axpy: out = g0*u0 + g1*u1 + gamma * sum_{n<k} coef[n]*scale[n] * V[n*ldv + :]  (u0/u1/scale may be
 *   NULL; out must not alias V).  Replaces LowRankMatrix.mv/rmv's Python loop of axpys (_jacobian.py:172-182) and
 *   the update formulas of BroydenFirst/Second (_jacobian.py:112-119,132-137); `scale[n]` carries 1/<dy_n, v_n>. */
long xk_vec_dots_workspace_elems(void);
int xk_vec_dots_f64(const double* a0, const double* b0, const double* a1, const double* b1, const double* a2,
                    const double* b2, const double* a3, const double* b3, int npairs, long L, double* partials,
                    long npart, double* out, void* stream);
int xk_vec_dots_f32(const float* a0, const float* b0, const float* a1, const float* b1, const float* a2,
                    const float* b2, const float* a3, const float* b3, int npairs, long L, double* partials,
                    long npart, double* out, void* stream);
int xk_broyden_axpy_f64(double* out, const double* u0, double g0, const double* u1, double g1, const double* V,
                        long ldv, const double* coef, const double* scale, int k, double gamma, long L, void* stream);
int xk_broyden_axpy_f32(float* out, const float* u0, double g0, const float* u1, double g1, const float* V, long ldv,
                        const float* coef, const float* scale, int k, double gamma, long L, void* stream);

/* ---- device-side collectives of the sharded solvers (RCCL over xGMI) -------------------------
 * The per-iteration exchanges that reproduce the reference's GLOBAL decisions under batch sharding — MAX of
 * {max|resid|, flags} (xitorch/_impls/linalg/symeig.py:188-203), MAX of {max residual norm, unconverged flag}
 * (_impls/linalg/solve.py:157,166,301,310), SUM of the Broyden inner products (_impls/optimize/root/_jacobian.py:172-182) —
 * as in-place all-reduces enqueued on the caller's stream, right behind the kernel that wrote the few doubles: the
 * host's one status read per iteration then returns reduced values.  librccl is looked up at the first call (the copy
 * the process already maps first); without it every entry point returns -2.  No RCCL type crosses the ABI.
 * xk_comm_available: 1 when a librccl was found.
 * xk_comm_unique_id: fills 128 opaque bytes on ONE rank; the caller distributes them (any side channel).
 * xk_comm_init_rank: this process' communicator of `nranks` ranks on HIP device `device` (collective over the ranks).
 * xk_comm_init_all: ndev communicators of one process, one per device devs[i] (devs == NULL: 0 .. ndev-1).
 * xk_comm_size: ranks / this rank of a communicator.   xk_comm_destroy: releases it (NULL: no-op).
 * xk_allreduce_f64 / _f32: buf[0..n) <- reduction over the ranks, op 0 = SUM, 1 = MAX, 2 = MIN, in place, on `stream`.
 *   One communicator serves one stream at a time.  Return codes: 0 ok, < 0 argument / unsupported, 1000 + ncclResult_t. */
int xk_comm_available(void);
int xk_comm_unique_id(void* id128);
int xk_comm_init_rank(const void* id128, int nranks, int rank, int device, void** comm);
int xk_comm_init_all(int ndev, const int* devs, void** comms);
int xk_comm_size(void* comm, int* nranks, int* rank);
int xk_comm_destroy(void* comm);
int xk_allreduce_f64(void* comm, double* buf, long n, int op, void* stream);
int xk_allreduce_f32(void* comm, float* buf, long n, int op, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XITORCH_AMD_H */
