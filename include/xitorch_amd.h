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
 *    void*; NULL = the default stream).  No hidden global state.
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

#ifdef __cplusplus
}
#endif
#endif /* XITORCH_AMD_H */
