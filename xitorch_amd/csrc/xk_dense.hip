// xitorch_amd :: K1 — batched dense operator-panel product for gfx950.
//
//   Y[b, c, i] = sum_j A[b, i, j] * X[b, c, j]          (trans = 0,  "mm")
//   Y[b, c, j] = sum_i A[b, i, j] * X[b, c, i]          (trans = 1,  "rmm")
//
// Replaces the reference's MatrixLinearOperator._mv/_mm/_rmv/_rmm
// (xitorch/_core/linop.py:692-702, `torch.matmul(self.mat, x)`) as called
// from the eigensolver panel product (xitorch/_impls/linalg/symeig.py:163,221)
// and from every Krylov operator apply (xitorch/_impls/linalg/solve.py:571-572).
// The same kernel computes panel Gram/Rayleigh blocks G = V W^T
// (symeig.py:170, _utils/tensor.py:15) by passing the basis as "A".
//
// Layout: A is row-major (B, M, N) with row pitch lda and batch pitch sA.
// Panels are PANEL-MAJOR: X is (B, P, N) — every panel column is a contiguous
// length-N vector (this is exactly the reference's "Fortran order" (B,N,P)
// view, _utils/tensor.py:21-32) — so all loads are unit-stride.
//
// Roofline: HBM.  Algorithmic bytes per call = B*M*N*s + B*P*(M+N)*s
// (SURVEY.md §8d).  Arithmetic intensity P/4 flop/B in fp64 — far below the
// ridge, so the design goal is: touch A exactly once with 16 B/lane streaming
// loads, keep the panel in L2/L1, never round-trip partial sums through HBM.
//
// Mapping (trans=0): one wave owns R consecutive rows of one batch member and
// sweeps the full row length; lane l reads 16 B of each of the R rows per
// step, the matching 16 B of each of the P panel columns (L2-resident, shared
// by all waves working on that batch member), and accumulates R*P partial
// sums in registers.  A transposing wave reduction folds the 64 lanes at the
// end.  Sweeps start at a wave-dependent column offset so that the power-of-2
// row pitch does not line every wave up on the same HBM channels.
#include "xk_common.h"

namespace xk {

template <typename T, int P, int R>
__global__ __launch_bounds__(256) void dense_mm_rows(
    const T* __restrict__ A, const T* __restrict__ X, T* __restrict__ Y,
    int M, int N, long lda, long sA, long ldx, long sX, long ldy, long sY,
    int row_groups_per_batch, int stagger, int steps_per_split, long sSplit) {
  // blockIdx.y = split of the contraction range (Gram blocks of a skinny basis: few long rows);
  // split s accumulates steps [s*steps_per_split, (s+1)*steps_per_split) into Y + s*sSplit.
  typedef typename Vec16<T>::type V;
  constexpr int VN = Vec16<T>::n;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int b = blockIdx.x / row_groups_per_batch;
  const int rg = (blockIdx.x - b * row_groups_per_batch) * 4 + wave;
  const int row0 = rg * R;
  if (row0 >= M) return;

  const T* Ab = A + (long)b * sA;
  const T* Xb = X + (long)b * sX;

  const T* arow[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int row = row0 + r;
    row = row < M ? row : M - 1;  // clamp: duplicate work, masked at the store
    arow[r] = Ab + (long)row * lda;
  }

  T acc[R * P];
#pragma unroll
  for (int i = 0; i < R * P; ++i) acc[i] = T(0);

  const int step_cols = 64 * VN;
  const int nsteps_all = (N + step_cols - 1) / step_cols;
  const int step_lo = blockIdx.y * steps_per_split;
  int nsteps = nsteps_all - step_lo;
  nsteps = nsteps < steps_per_split ? nsteps : steps_per_split;
  if (nsteps < 0) nsteps = 0;
  int t0 = (stagger && nsteps > 0) ? (int)(((unsigned)rg * 29u + (unsigned)b * 13u) % (unsigned)nsteps) : 0;

  for (int it = 0; it < nsteps; ++it) {
    int t = it + t0;
    t = t >= nsteps ? t - nsteps : t;
    const int j = ((step_lo + t) * 64 + lane) * VN;
    if (j < N) {  // N % VN == 0 is guaranteed by the launcher
      V xv[P];
#pragma unroll
      for (int c = 0; c < P; ++c) xv[c] = *reinterpret_cast<const V*>(Xb + (long)c * ldx + j);
      V av[R];
#pragma unroll
      for (int r = 0; r < R; ++r) av[r] = ld_stream(reinterpret_cast<const V*>(arow[r] + j));
      // all R row loads must be in flight before the first FMA: left alone, hipcc sinks every load next to its
      // use to save registers (one 1 KB request in flight per wave, `s_waitcnt vmcnt(0)` after each)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < P; ++c)
#pragma unroll
          for (int v = 0; v < VN; ++v) acc[r * P + c] += av[r][v] * xv[c][v];
    }
  }

  wave_reduce_scatter<T, R * P>(acc, lane);
  if (wave_rs_is_writer<R * P>(lane)) {
    T* Yb = Y + (long)blockIdx.y * sSplit + (long)b * sY;
#pragma unroll
    for (int i = 0; i < WaveRsCount<R * P>::value; ++i) {
      const int idx = wave_rs_orig_index<R * P>(i, lane);
      const int r = idx / P, c = idx - r * P;
      if (row0 + r < M) Yb[(long)c * ldy + row0 + r] = acc[i];
    }
  }
}

// fold the split partials: Y[b,c,i] = sum_s W[s,b,c,i]   (fixed order -> deterministic)
template <typename T>
__global__ __launch_bounds__(256) void fold_splits(const T* __restrict__ W, T* __restrict__ Y, int M, int P,
                                                    int nsplit, long ldy, long sY, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over B*P*M
  if (idx >= total) return;
  const long per_b = (long)P * M;
  const long b = idx / per_b;
  const long rem = idx - b * per_b;
  const int c = (int)(rem / M);
  const int i = (int)(rem - (long)c * M);
  T s = T(0);
  for (int k = 0; k < nsplit; ++k) s += W[(long)k * total + idx];
  Y[b * sY + (long)c * ldy + i] = s;
}

// Scalar-load fallback for shapes the 16 B path cannot take (N % VN != 0 or
// unaligned pitches).  Same mapping, 1 element per lane per step.
template <typename T, int P, int R>
__global__ __launch_bounds__(256) void dense_mm_rows_scalar(
    const T* __restrict__ A, const T* __restrict__ X, T* __restrict__ Y,
    int M, int N, long lda, long sA, long ldx, long sX, long ldy, long sY,
    int row_groups_per_batch) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int b = blockIdx.x / row_groups_per_batch;
  const int rg = (blockIdx.x - b * row_groups_per_batch) * 4 + wave;
  const int row0 = rg * R;
  if (row0 >= M) return;
  const T* Ab = A + (long)b * sA;
  const T* Xb = X + (long)b * sX;
  T acc[R * P];
#pragma unroll
  for (int i = 0; i < R * P; ++i) acc[i] = T(0);
  for (int j = lane; j < N; j += 64) {
    T xv[P];
#pragma unroll
    for (int c = 0; c < P; ++c) xv[c] = Xb[(long)c * ldx + j];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int row = row0 + r;
      row = row < M ? row : M - 1;
      const T a = Ab[(long)row * lda + j];
#pragma unroll
      for (int c = 0; c < P; ++c) acc[r * P + c] += a * xv[c];
    }
  }
  wave_reduce_scatter<T, R * P>(acc, lane);
  if (wave_rs_is_writer<R * P>(lane)) {
    T* Yb = Y + (long)b * sY;
#pragma unroll
    for (int i = 0; i < WaveRsCount<R * P>::value; ++i) {
      const int idx = wave_rs_orig_index<R * P>(i, lane);
      const int r = idx / P, c = idx - r * P;
      if (row0 + r < M) Yb[(long)c * ldy + row0 + r] = acc[i];
    }
  }
}

// ---------------------------------------------------------------------------
// trans = 1:  Y[b, c, j] = sum_i A[b, i, j] X[b, c, i]
// Lane l owns VN consecutive output columns j and walks down a slab of rows;
// every A load is the same coalesced 16 B/lane stream as above, X[c, i] is
// wave-uniform (scalar loads).  The row range is split into `nslab` slabs so
// the grid fills the chip; slab partials go to a workspace (B, nslab, P, N)
// and a second tiny kernel folds them in a fixed order (deterministic, no
// atomics).
// ---------------------------------------------------------------------------
template <typename T, int P>
__global__ __launch_bounds__(256) void dense_rmm_cols(
    const T* __restrict__ A, const T* __restrict__ X, T* __restrict__ W,
    int M, int N, long lda, long sA, long ldx, long sX,
    int col_tiles, int nslab, int rows_per_slab) {
  typedef typename Vec16<T>::type V;
  constexpr int VN = Vec16<T>::n;
  int bid = blockIdx.x;
  const int ct = bid % col_tiles; bid /= col_tiles;
  const int slab = bid % nslab;
  const int b = bid / nslab;
  const int j = (ct * 256 + threadIdx.x) * VN;
  if (j >= N) return;
  const T* Ab = A + (long)b * sA;
  const T* Xb = X + (long)b * sX;
  const int i0 = slab * rows_per_slab;
  int i1 = i0 + rows_per_slab; i1 = i1 < M ? i1 : M;
  V acc[P];
#pragma unroll
  for (int c = 0; c < P; ++c)
#pragma unroll
    for (int v = 0; v < VN; ++v) acc[c][v] = T(0);
  int i = i0;
  // 4 rows per step and no scheduling fence: this kernel runs 8 waves per SIMD, and an 8-row step with the
  // loads fenced together measured 4-9 % SLOWER (6.2 vs 6.8 TB/s at P = 6) — unlike dense_mm_rows above.  The
  // K1s recipe (buffer descriptor + ring of 8 rows refilled after use) is equal alone (6.73 vs 6.79 TB/s) and
  // 3 % slower inside the eigensolver's pipeline (410 vs 397 ms per call), and spills SGPRs at P = 8.
  for (; i + 4 <= i1; i += 4) {
    V a0 = ld_stream(reinterpret_cast<const V*>(Ab + (long)(i + 0) * lda + j));
    V a1 = ld_stream(reinterpret_cast<const V*>(Ab + (long)(i + 1) * lda + j));
    V a2 = ld_stream(reinterpret_cast<const V*>(Ab + (long)(i + 2) * lda + j));
    V a3 = ld_stream(reinterpret_cast<const V*>(Ab + (long)(i + 3) * lda + j));
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const T x0 = Xb[(long)c * ldx + i + 0], x1 = Xb[(long)c * ldx + i + 1];
      const T x2 = Xb[(long)c * ldx + i + 2], x3 = Xb[(long)c * ldx + i + 3];
#pragma unroll
      for (int v = 0; v < VN; ++v) {
        acc[c][v] += a0[v] * x0;
        acc[c][v] += a1[v] * x1;
        acc[c][v] += a2[v] * x2;
        acc[c][v] += a3[v] * x3;
      }
    }
  }
  for (; i < i1; ++i) {
    V a0 = ld_stream(reinterpret_cast<const V*>(Ab + (long)i * lda + j));
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const T x0 = Xb[(long)c * ldx + i];
#pragma unroll
      for (int v = 0; v < VN; ++v) acc[c][v] += a0[v] * x0;
    }
  }
  T* Wb = W + (((long)b * nslab + slab) * P) * (long)N;
#pragma unroll
  for (int c = 0; c < P; ++c) *reinterpret_cast<V*>(Wb + (long)c * N + j) = acc[c];
}

template <typename T>
__global__ __launch_bounds__(256) void fold_slabs(
    const T* __restrict__ W, T* __restrict__ Y, int N, int P, int nslab, long ldy, long sY,
    long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // over B*P*N
  if (idx >= total) return;
  const long per_b = (long)P * N;
  const long b = idx / per_b;
  const long rem = idx - b * per_b;
  const int c = (int)(rem / N);
  const int j = (int)(rem - (long)c * N);
  T s = T(0);
  for (int k = 0; k < nslab; ++k) s += W[(((long)b * nslab + k) * P + c) * (long)N + j];
  Y[b * sY + (long)c * ldy + j] = s;
}

// scalar fallback for trans=1 (any N / alignment): one thread per output column
template <typename T, int P>
__global__ __launch_bounds__(256) void dense_rmm_cols_scalar(
    const T* __restrict__ A, const T* __restrict__ X, T* __restrict__ Y,
    int M, int N, long lda, long sA, long ldx, long sX, long ldy, long sY, int col_tiles) {
  const int ct = blockIdx.x % col_tiles;
  const int b = blockIdx.x / col_tiles;
  const int j = ct * 256 + threadIdx.x;
  if (j >= N) return;
  const T* Ab = A + (long)b * sA;
  const T* Xb = X + (long)b * sX;
  T acc[P];
#pragma unroll
  for (int c = 0; c < P; ++c) acc[c] = T(0);
  for (int i = 0; i < M; ++i) {
    const T a = Ab[(long)i * lda + j];
#pragma unroll
    for (int c = 0; c < P; ++c) acc[c] += a * Xb[(long)c * ldx + i];
  }
#pragma unroll
  for (int c = 0; c < P; ++c) Y[(long)b * sY + (long)c * ldy + j] = acc[c];
}

// ---------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------
template <typename T> static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// number of contraction splits for a (B, M, N) row-sweep: only skinny problems (few waves, long
// rows) are split, so that the chip is filled; K1 proper (M = N) is never split.
template <typename T>
static int choose_nsplit(int B, int M, int N, int R) {
  constexpr int VN = Vec16<T>::n;
  const long waves = (long)B * ((M + R - 1) / R);
  const int nsteps = (N + 64 * VN - 1) / (64 * VN);
  if (waves >= 2048 || nsteps < 16) return 1;
  long ns = (4096 + waves - 1) / waves;
  const long max_ns = nsteps / 8 > 0 ? nsteps / 8 : 1;   // >= 8 steps per split
  if (ns > max_ns) ns = max_ns;
  if (ns > 64) ns = 64;
  return ns < 1 ? 1 : (int)ns;
}

template <typename T, int P, int R>
static int launch_rows(const T* A, const T* X, T* Y, T* ws, long ws_elems, int B, int M, int N, long lda,
                       long sA, long ldx, long sX, long ldy, long sY, int stagger, hipStream_t st) {
  constexpr int VN = Vec16<T>::n;
  const int rgpb = (M + 4 * R - 1) / (4 * R);
  const long nblk = (long)B * rgpb;
  if (nblk <= 0) return XK_OK;
  if (nblk > 0x7fffffffL) return XK_ERR_UNSUPPORTED;
  const bool vec_ok = (N % VN == 0) && (lda % VN == 0) && (sA % VN == 0) && (ldx % VN == 0) &&
                      (sX % VN == 0) && aligned16<T>(A) && aligned16<T>(X);
  if (vec_ok) {
    const int nsteps = (N + 64 * VN - 1) / (64 * VN);
    int nsplit = choose_nsplit<T>(B, M, N, R);
    const long total = (long)B * P * M;
    if (nsplit > 1 && (ws == nullptr || ws_elems < (long)nsplit * total)) nsplit = 1;
    if (nsplit <= 1) {
      hipLaunchKernelGGL((dense_mm_rows<T, P, R>), dim3((unsigned)nblk), dim3(256), 0, st, A, X, Y, M, N,
                         lda, sA, ldx, sX, ldy, sY, rgpb, stagger, nsteps, 0L);
    } else {
      const int sps = (nsteps + nsplit - 1) / nsplit;
      nsplit = (nsteps + sps - 1) / sps;
      // partials are compact (nsplit, B, P, M)
      hipLaunchKernelGGL((dense_mm_rows<T, P, R>), dim3((unsigned)nblk, (unsigned)nsplit), dim3(256), 0, st,
                         A, X, ws, M, N, lda, sA, ldx, sX, (long)M, (long)P * M, rgpb, stagger, sps, total);
      XK_LAUNCH_CHECK();
      hipLaunchKernelGGL((fold_splits<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws, Y, M,
                         P, nsplit, ldy, sY, total);
    }
  } else
    hipLaunchKernelGGL((dense_mm_rows_scalar<T, P, R>), dim3((unsigned)nblk), dim3(256), 0, st, A, X,
                       Y, M, N, lda, sA, ldx, sX, ldy, sY, rgpb);
  XK_LAUNCH_CHECK();
  return XK_OK;
}

// rows-per-wave by panel width: keep R*P accumulators <= 64 per lane
template <typename T, int P> struct RowsFor {
  static constexpr int value = P <= 2 ? 16 : (P <= 4 ? 12 : (P <= 8 ? 8 : 4));
};

template <typename T, int P>
static int mm_rows_p(const T* A, const T* X, T* Y, T* ws, long wsn, int B, int M, int N, long lda, long sA,
                     long ldx, long sX, long ldy, long sY, int rows_hint, int stagger, hipStream_t st) {
  // rows_hint lets the tuning harness override R; 0 = default
  switch (rows_hint) {
    case 4: return launch_rows<T, P, 4>(A, X, Y, ws, wsn, B, M, N, lda, sA, ldx, sX, ldy, sY, stagger, st);
    case 8: return launch_rows<T, P, 8>(A, X, Y, ws, wsn, B, M, N, lda, sA, ldx, sX, ldy, sY, stagger, st);
    case 16:
      if (P <= 4)
        return launch_rows<T, P, 16>(A, X, Y, ws, wsn, B, M, N, lda, sA, ldx, sX, ldy, sY, stagger, st);
      return XK_ERR_UNSUPPORTED;
    default:
      return launch_rows<T, P, RowsFor<T, P>::value>(A, X, Y, ws, wsn, B, M, N, lda, sA, ldx, sX, ldy, sY,
                                                     stagger, st);
  }
}

template <typename T>
static int mm_rows(const T* A, const T* X, T* Y, T* ws, long wsn, int B, int M, int N, int P, long lda,
                   long sA, long ldx, long sX, long ldy, long sY, int rows_hint, int stagger,
                   hipStream_t st) {
  // panels wider than 8 are processed in column blocks of <= 8 (A is re-read
  // once per block; the MFMA wide-panel kernel takes over for P >= 16).
  int c0 = 0;
  while (c0 < P) {
    const int pc = (P - c0) >= 8 ? 8 : (P - c0);
    const T* Xc = X + (long)c0 * ldx;
    T* Yc = Y + (long)c0 * ldy;
    int rc;
    switch (pc) {
#define XK_CASE(PP)                                                                           \
  case PP:                                                                                    \
    rc = mm_rows_p<T, PP>(A, Xc, Yc, ws, wsn, B, M, N, lda, sA, ldx, sX, ldy, sY, rows_hint, stagger, st); \
    break;
      XK_CASE(1) XK_CASE(2) XK_CASE(3) XK_CASE(4) XK_CASE(5) XK_CASE(6) XK_CASE(7) XK_CASE(8)
#undef XK_CASE
      default: rc = XK_ERR_UNSUPPORTED;
    }
    if (rc != XK_OK) return rc;
    c0 += pc;
  }
  return XK_OK;
}

template <typename T, int P>
static int rmm_cols_p(const T* A, const T* X, T* Y, T* ws, long ws_elems, int B, int M, int N,
                      long lda, long sA, long ldx, long sX, long ldy, long sY, hipStream_t st) {
  constexpr int VN = Vec16<T>::n;
  const bool vec_ok = (N % VN == 0) && (lda % VN == 0) && (sA % VN == 0) && aligned16<T>(A) &&
                      aligned16<T>(ws) && ws != nullptr;
  if (!vec_ok) {
    const int ct = (N + 255) / 256;
    hipLaunchKernelGGL((dense_rmm_cols_scalar<T, P>), dim3((unsigned)((long)B * ct)), dim3(256), 0, st,
                       A, X, Y, M, N, lda, sA, ldx, sX, ldy, sY, ct);
    XK_LAUNCH_CHECK();
    return XK_OK;
  }
  const int ct = (N + 256 * VN - 1) / (256 * VN);
  // enough slabs to put >= ~8 blocks on every CU, but never below 64 rows/slab
  int nslab = (2048 + B * ct - 1) / (B * ct);
  int max_slab = (M + 63) / 64;
  if (nslab > max_slab) nslab = max_slab;
  if (nslab < 1) nslab = 1;
  while ((long)B * nslab * P * (long)N > ws_elems && nslab > 1) --nslab;
  if ((long)B * nslab * P * (long)N > ws_elems) return XK_ERR_ARG;
  const int rps = (M + nslab - 1) / nslab;
  hipLaunchKernelGGL((dense_rmm_cols<T, P>), dim3((unsigned)((long)B * nslab * ct)), dim3(256), 0, st,
                     A, X, ws, M, N, lda, sA, ldx, sX, ct, nslab, rps);
  XK_LAUNCH_CHECK();
  const long tot = (long)B * P * N;
  hipLaunchKernelGGL((fold_slabs<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, ws, Y,
                     N, P, nslab, ldy, sY, tot);
  XK_LAUNCH_CHECK();
  return XK_OK;
}

template <typename T>
static int rmm_cols(const T* A, const T* X, T* Y, T* ws, long ws_elems, int B, int M, int N, int P,
                    long lda, long sA, long ldx, long sX, long ldy, long sY, hipStream_t st) {
  int c0 = 0;
  while (c0 < P) {
    // the column-oriented kernel keeps P*VN accumulators per lane and takes its panel values from scalar
    // loads, so a 16-wide block still streams A at the HBM rate: wide panels (many right-hand sides) cost
    // ceil(P/16) passes over the operator instead of ceil(P/8)
    if (P - c0 >= 12) {
      const int pw = (P - c0) >= 16 ? 16 : 12;
      const T* Xw = X + (long)c0 * ldx;
      T* Yw = Y + (long)c0 * ldy;
      const int rcw = pw == 16 ? rmm_cols_p<T, 16>(A, Xw, Yw, ws, ws_elems, B, M, N, lda, sA, ldx, sX, ldy, sY, st)
                               : rmm_cols_p<T, 12>(A, Xw, Yw, ws, ws_elems, B, M, N, lda, sA, ldx, sX, ldy, sY, st);
      if (rcw != XK_OK) return rcw;
      c0 += pw;
      continue;
    }
    const int pc = (P - c0) >= 8 ? 8 : (P - c0);
    const T* Xc = X + (long)c0 * ldx;
    T* Yc = Y + (long)c0 * ldy;
    int rc;
    switch (pc) {
#define XK_CASE(PP)                                                                         \
  case PP:                                                                                  \
    rc = rmm_cols_p<T, PP>(A, Xc, Yc, ws, ws_elems, B, M, N, lda, sA, ldx, sX, ldy, sY, st); \
    break;
      XK_CASE(1) XK_CASE(2) XK_CASE(3) XK_CASE(4) XK_CASE(5) XK_CASE(6) XK_CASE(7) XK_CASE(8)
#undef XK_CASE
      default: rc = XK_ERR_UNSUPPORTED;
    }
    if (rc != XK_OK) return rc;
    c0 += pc;
  }
  return XK_OK;
}

}  // namespace xk

// ---------------------------------------------------------------------------
// C ABI (declared in include/xitorch_amd.h)
// ---------------------------------------------------------------------------
extern "C" {

long xk_dense_mm_workspace_elems(int B, int M, int N, int P, int trans) {
  if (!trans) {
    // split partials of skinny row sweeps (Gram blocks); upper bound over the R choices
    const int pc = P > 8 ? 8 : P;
    int ns = 1;
    for (int R = 4; R <= 16; R += 4) {
      const int n1 = xk::choose_nsplit<double>(B, M, N, R), n2 = xk::choose_nsplit<float>(B, M, N, R);
      ns = n1 > ns ? n1 : ns;
      ns = n2 > ns ? n2 : ns;
    }
    return ns > 1 ? (long)ns * B * pc * M : 0;
  }
  // trans=1 slab partials: at most ceil(2048/(B*ct)) slabs of (P<=16, N) per batch member
  const int pc = P > 16 ? 16 : P;
  const int ct = (N + 511) / 512;
  long nslab = (2048 + (long)B * ct - 1) / ((long)B * ct);
  long max_slab = (M + 63) / 64;
  if (nslab > max_slab) nslab = max_slab;
  if (nslab < 1) nslab = 1;
  return (long)B * nslab * pc * (long)N;
}

int xk_dense_mm_f64(const double* A, const double* X, double* Y, double* ws, long ws_elems, int B,
                    int M, int N, int P, long lda, long sA, long ldx, long sX, long ldy, long sY,
                    int trans, int rows_hint, int stagger, void* stream) {
  if (B < 0 || M < 0 || N < 0 || P < 0) return XK_ERR_ARG;
  if (B == 0 || P == 0 || M == 0 || N == 0) return XK_OK;
  hipStream_t st = (hipStream_t)stream;
  if (!trans)
    return xk::mm_rows<double>(A, X, Y, ws, ws_elems, B, M, N, P, lda, sA, ldx, sX, ldy, sY, rows_hint, stagger, st);
  return xk::rmm_cols<double>(A, X, Y, ws, ws_elems, B, M, N, P, lda, sA, ldx, sX, ldy, sY, st);
}

int xk_dense_mm_f32(const float* A, const float* X, float* Y, float* ws, long ws_elems, int B, int M,
                    int N, int P, long lda, long sA, long ldx, long sX, long ldy, long sY, int trans,
                    int rows_hint, int stagger, void* stream) {
  if (B < 0 || M < 0 || N < 0 || P < 0) return XK_ERR_ARG;
  if (B == 0 || P == 0 || M == 0 || N == 0) return XK_OK;
  hipStream_t st = (hipStream_t)stream;
  if (!trans)
    return xk::mm_rows<float>(A, X, Y, ws, ws_elems, B, M, N, P, lda, sA, ldx, sX, ldy, sY, rows_hint, stagger, st);
  return xk::rmm_cols<float>(A, X, Y, ws, ws_elems, B, M, N, P, lda, sA, ldx, sX, ldy, sY, st);
}

}  // extern "C"
