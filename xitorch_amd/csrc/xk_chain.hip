// xitorch_amd :: the per-iteration chain of the block Davidson eigensolver as a handful of C calls.
//
// Between two operator-panel products an iteration of `davidson` (xitorch/_impls/linalg/symeig.py:160-223) does
//   Rayleigh-Ritz rotation + residual + stopping norm            (symeig.py:178-197)
//   orthonormalisation of the new block against the basis         (:207-220, tallqr of [V, t], _utils/tensor.py:8-19)
//   extension of T = V^T A V by the new rows / columns            (:163-170 recomputes it whole)
// Each of these is two to eight small launches.  Issued one by one from the Python host loop they cost more host
// time than GPU time once the batch per GPU is small (8 operators of order 16384: ~20 launches x ~15 us of
// interpreter + ctypes per group and iteration against 0.8 ms of panel product), which is what bounds strong scaling.
// The entry points below enqueue a whole stage from C++ (a launch is ~3 us here) on the caller's stream:
//
//   xk_davidson_ritz      ritz_residual + group status (which re-zeroes the per-member max for the next round)
//   xk_davidson_orth      block Gram-Schmidt of the new panel against the basis (one pass: Gram, projection, CholeskyQR;
//                         more: [Gram, projection, CholeskyQR] per pass, the first CholeskyQR shifted), the
//                         CholeskyQR of the panel in ONE kernel (Gram, Cholesky, inverse, transform; one workgroup per
//                         batch member, the p x N panel is read twice and written once) — panels up to 8 vectors;
//                         wider ones take the separate Gram / xk_panel_chol / xk_panel_transform kernels
//   xk_davidson_extend_t  Gram block of the new A V panel against the basis + scatter into both triangles of T
//
// Nothing here changes the arithmetic of the stages: same kernels / same formulas as the separate entry points
// (the fused CholeskyQR sums its Gram entries per thread, then per wave, then across the 16 waves in fixed order).
#include "xk_common.h"

extern "C" {
long xk_dense_mm_workspace_elems(int B, int M, int N, int P, int trans);
int xk_dense_mm_f64(const double*, const double*, double*, double*, long, int, int, int, int, long, long, long, long,
                    long, long, int, int, int, void*);
int xk_dense_mm_f32(const float*, const float*, float*, float*, long, int, int, int, int, long, long, long, long, long,
                    long, int, int, int, void*);
int xk_lincomb_f64(const double*, const double*, double*, int, int, int, int, long, long, long, long, long, long, long,
                   double, double, void*);
int xk_lincomb_f32(const float*, const float*, float*, int, int, int, int, long, long, long, long, long, long, long,
                   double, double, void*);
int xk_ritz_residual_f64(const double*, const double*, const double*, const double*, double*, double*, double*, int,
                         int, int, int, long, long, long, long, long, long, long, long, long, long, long, long, void*);
int xk_ritz_residual_f32(const float*, const float*, const float*, const float*, float*, float*, float*, int, int, int,
                         int, long, long, long, long, long, long, long, long, long, long, long, long, void*);
int xk_panel_chol_f64(const double*, double*, int*, int, int, long, long, void*);
int xk_panel_chol_f32(const float*, float*, int*, int, int, long, long, void*);
int xk_panel_transform_f64(double*, const double*, int, int, int, long, long, void*);
int xk_panel_transform_f32(float*, const float*, int, int, int, long, long, void*);
}

namespace xk {

static inline int dense_mm(const double* A, const double* X, double* Y, double* ws, long wsn, int B, int M, int N,
                           int P, long lda, long sA, long ldx, long sX, long ldy, long sY, void* st) {
  return xk_dense_mm_f64(A, X, Y, ws, wsn, B, M, N, P, lda, sA, ldx, sX, ldy, sY, 0, 0, 1, st);
}
static inline int dense_mm(const float* A, const float* X, float* Y, float* ws, long wsn, int B, int M, int N, int P,
                           long lda, long sA, long ldx, long sX, long ldy, long sY, void* st) {
  return xk_dense_mm_f32(A, X, Y, ws, wsn, B, M, N, P, lda, sA, ldx, sX, ldy, sY, 0, 0, 1, st);
}
static inline int lincomb_c(const double* V, const double* C, double* O, int B, int k, int N, int P, long ldv, long sV,
                            long sC, long sCa, long sCc, long ldo, long sO, double al, double be, void* st) {
  return xk_lincomb_f64(V, C, O, B, k, N, P, ldv, sV, sC, sCa, sCc, ldo, sO, al, be, st);
}
static inline int lincomb_c(const float* V, const float* C, float* O, int B, int k, int N, int P, long ldv, long sV,
                            long sC, long sCa, long sCc, long ldo, long sO, double al, double be, void* st) {
  return xk_lincomb_f32(V, C, O, B, k, N, P, ldv, sV, sC, sCa, sCc, ldo, sO, al, be, st);
}
static inline int ritz_c(const double* V, const double* AV, const double* Y, const double* lam, double* X, double* Tn,
                         double* rmax, int B, int k, int N, int P, long a, long b, long c, long d, long e, long f,
                         long g, long h, long i, long j, long l, long m, void* st) {
  return xk_ritz_residual_f64(V, AV, Y, lam, X, Tn, rmax, B, k, N, P, a, b, c, d, e, f, g, h, i, j, l, m, st);
}
static inline int ritz_c(const float* V, const float* AV, const float* Y, const float* lam, float* X, float* Tn,
                         float* rmax, int B, int k, int N, int P, long a, long b, long c, long d, long e, long f, long g,
                         long h, long i, long j, long l, long m, void* st) {
  return xk_ritz_residual_f32(V, AV, Y, lam, X, Tn, rmax, B, k, N, P, a, b, c, d, e, f, g, h, i, j, l, m, st);
}
static inline int chol_c(const double* G, double* W, int* info, int B, int P, long ldg, long sG, void* st) {
  return xk_panel_chol_f64(G, W, info, B, P, ldg, sG, st);
}
static inline int chol_c(const float* G, float* W, int* info, int B, int P, long ldg, long sG, void* st) {
  return xk_panel_chol_f32(G, W, info, B, P, ldg, sG, st);
}
static inline int transform_c(double* Tp, const double* W, int B, int P, int N, long ldt, long sT, void* st) {
  return xk_panel_transform_f64(Tp, W, B, P, N, ldt, sT, st);
}
static inline int transform_c(float* Tp, const float* W, int B, int P, int N, long ldt, long sT, void* st) {
  return xk_panel_transform_f32(Tp, W, B, P, N, ldt, sT, st);
}

// a value every lane of the wave holds identically, moved to scalar registers (the 21 inverse-factor entries of the
// fused CholeskyQR otherwise sit in VGPRs across its panel loop: with them the fp32 kernel spilled at 128 VGPRs)
__device__ __forceinline__ float chain_uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ double chain_uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// status = {max_b rmax[b] (NaN if any is NaN), max_b info[b], max_b flag[b], max_b cond[b] (only with cond),
// max_b orth[b] (only with orth: the a-posteriori guard, NaN counts as infinite)} as doubles, cond <- 0, orth <- 0,
// then rmax <- 0 for the next
// Rayleigh-Ritz step (the residual kernel folds into it with an order-independent atomic max).  One wave.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void group_status_rezero_kernel(T* __restrict__ rmax, const int* __restrict__ info,
                                                                 const int* __restrict__ flag, T* __restrict__ cond,
                                                                 T* __restrict__ orth, double* __restrict__ status,
                                                                 int B) {
  const int lane = threadIdx.x;
  double m = 0.0, cm = 0.0, om_ = 0.0;
  int nan = 0, i1 = 0, i2 = 0;
  bool first = true;
  for (int b = lane; b < B; b += 64) {
    const double v = (double)rmax[b];
    rmax[b] = T(0);
    if (cond) {                                              // worst pivot ratio of the panels orthonormalised since the
      const double cv = (double)cond[b];                     // last status (NaN counts as infinite), re-zeroed
      cond[b] = T(0);
      cm = (cv != cv) ? __builtin_inf() : (cv > cm ? cv : cm);
    }
    if (orth) {                                              // worst |X^T M X - I| of the Ritz blocks checked since
      const double ov = (double)orth[b];                     // the last status, re-zeroed
      orth[b] = T(0);
      om_ = (ov != ov) ? __builtin_inf() : (ov > om_ ? ov : om_);
    }
    nan |= (v != v);
    m = first ? v : (v > m ? v : m);
    const int a = info[b];
    i1 = first ? a : (a > i1 ? a : i1);
    if (flag) {
      const int f = flag[b];
      i2 = first ? f : (f > i2 ? f : i2);
    }
    first = false;
  }
  double mm = first ? -__builtin_inf() : m;
  int a1 = first ? -2147483647 - 1 : i1, a2 = first ? -2147483647 - 1 : i2;
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) {
    const double om = __shfl_xor(mm, sft, 64), oc = __shfl_xor(cm, sft, 64), oo = __shfl_xor(om_, sft, 64);
    const int o1 = __shfl_xor(a1, sft, 64), o2 = __shfl_xor(a2, sft, 64), on = __shfl_xor(nan, sft, 64);
    mm = om > mm ? om : mm;
    cm = oc > cm ? oc : cm;
    om_ = oo > om_ ? oo : om_;
    a1 = o1 > a1 ? o1 : a1;
    a2 = o2 > a2 ? o2 : a2;
    nan |= on;
  }
  if (lane == 0) {
    status[0] = nan ? __builtin_nan("") : mm;
    status[1] = (double)a1;
    status[2] = flag ? (double)a2 : 0.0;
    if (cond) status[3] = cm;
    if (orth) status[4] = om_;
  }
}

// ---------------------------------------------------------------------------------------------
// A-posteriori guard of a Rayleigh-Ritz block: orth[b] = max(orth[b], max_{c,d} |<X_c, (M X)_d> - delta_cd|) for the P
// rows of the panel X (MX == X when there is no overlap operator).  The reference re-orthonormalises the WHOLE basis
// every iteration (tallqr of [V, t], _utils/tensor.py:8-19, symeig.py:207-223), so the Ritz vectors V y it returns are
// orthonormal by construction; this build orthonormalises only the new panel, and a basis that has lost its
// orthogonality shows up here as a Ritz block whose Gram matrix is not the identity (two copies of one eigenpair: an
// off-diagonal entry of 1) — the driver rolls such a step back instead of returning it.  One 1024-thread workgroup
// per batch member, the panel is read once; sums per thread -> wave -> 16 waves in fixed order.  P <= 8.
// ---------------------------------------------------------------------------------------------
template <typename T, int P>
__global__ __launch_bounds__(1024) void ritz_guard_kernel(const T* __restrict__ X, const T* __restrict__ MX,
                                                          T* __restrict__ orth, int N, long ldx, long sX, long ldm,
                                                          long sM) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  constexpr int NG = P * (P + 1) / 2;
  __shared__ T part[16][NG];
  const int b = blockIdx.x;
  const T* Xb = X + (long)b * sX;
  const T* Mb = MX + (long)b * sM;
  const bool same = (MX == X);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T g[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) g[i] = T(0);
  for (int j = tid * VN; j < N; j += 1024 * VN) {
    VT x[P], m[P];
#pragma unroll
    for (int c = 0; c < P; ++c) x[c] = *reinterpret_cast<const VT*>(Xb + (long)c * ldx + j);
    if (same) {
#pragma unroll
      for (int c = 0; c < P; ++c) m[c] = x[c];
    } else {
#pragma unroll
      for (int c = 0; c < P; ++c) m[c] = *reinterpret_cast<const VT*>(Mb + (long)c * ldm + j);
    }
    int i = 0;
#pragma unroll
    for (int c = 0; c < P; ++c)
#pragma unroll
      for (int d = c; d < P; ++d, ++i)
#pragma unroll
        for (int v = 0; v < VN; ++v) g[i] += x[c][v] * m[d][v];
  }
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const T s = wave_sum(g[i]);
    if (lane == 0) part[wave][i] = s;
  }
  __syncthreads();
  if (tid == 0) {
    T dev = T(0);
    bool bad = false;
    int i = 0;
    for (int c = 0; c < P; ++c)
      for (int d = c; d < P; ++d, ++i) {
        T s = T(0);
        for (int w = 0; w < 16; ++w) s += part[w][i];          // fixed order
        const T e = fabs(s - (c == d ? T(1) : T(0)));
        if (e != e) bad = true;
        dev = e > dev ? e : dev;
      }
    if (bad) dev = T(INFINITY);
    const T old = orth[b];
    orth[b] = (dev > old || old != old) ? dev : old;
  }
}

// the same from a Gram block computed on K1 (panels wider than 8): G (B, P, P) compact, one wave per member
template <typename T>
__global__ __launch_bounds__(64) void gram_guard_kernel(const T* __restrict__ G, T* __restrict__ orth, int P) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const T* Gb = G + (long)b * P * P;
  T dev = T(0);
  int bad = 0;
  for (int i = lane; i < P * P; i += 64) {
    const int c = i / P, d = i - c * P;
    const T e = fabs(Gb[i] - (c == d ? T(1) : T(0)));
    bad |= (e != e);
    dev = e > dev ? e : dev;
  }
  dev = wave_max(dev);
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) bad |= __shfl_xor(bad, sft, 64);
  if (lane == 0) {
    if (bad) dev = T(INFINITY);
    const T old = orth[b];
    orth[b] = (dev > old || old != old) ? dev : old;
  }
}

// ---------------------------------------------------------------------------------------------
// CholeskyQR of a P-row panel in one kernel (tallqr restricted to the new block, _utils/tensor.py:15-18):
//   G = t t^T (P x P, symmetrised like xk_panel_chol), G = R^T R, W = R^-1, t <- W^T t   (row c <- sum_{a<=c} W[a,c] t_a)
// One 1024-thread workgroup per batch member; info[b] = index+1 of the first non-positive pivot (sticky, like
// xk_panel_chol).  P <= 8.  shift_rel > 0: the first step of shifted CholeskyQR (Fukaya et al.): the Gram matrix gets
// shift_rel * trace(G) on its diagonal, which makes the factorisation safe for panels whose condition number squared
// exceeds 1 / eps; the result is then only well-conditioned, not orthonormal — a plain pass follows.
// ---------------------------------------------------------------------------------------------
template <typename T, int P, bool SHIFT>
__global__ __launch_bounds__(1024) void panel_cholqr_kernel(T* __restrict__ Tp, int* __restrict__ info, int N,
                                                            long ldt, long sT, T shift_rel, T* __restrict__ cond) {
  // (N here is the panel length rounded up to whole 16 B vectors: the pad elements are zero by the panel contract)
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  constexpr int NG = P * (P + 1) / 2;
  __shared__ T part[16][NG];
  __shared__ T Wsh[P][P];
  __shared__ T G[P][P], R[P][P], col[P];
  const int b = blockIdx.x;
  T* Tb = Tp + (long)b * sT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T g[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) g[i] = T(0);
  for (int j = tid * VN; j < N; j += 1024 * VN) {
    VT t[P];
#pragma unroll
    for (int c = 0; c < P; ++c) t[c] = *reinterpret_cast<const VT*>(Tb + (long)c * ldt + j);
    int i = 0;
#pragma unroll
    for (int c = 0; c < P; ++c)
#pragma unroll
      for (int d = c; d < P; ++d, ++i)
#pragma unroll
        for (int v = 0; v < VN; ++v) g[i] += t[c][v] * t[d][v];
  }
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const T s = wave_sum(g[i]);
    if (lane == 0) part[wave][i] = s;
  }
  __syncthreads();
  if (!SHIFT && tid == 0) {
    // plain CholeskyQR (every iteration of the default path): the 6 x 6 factorisation unrolled in registers
    T Gr[P][P], Rr[P][P];
    int i = 0;
    for (int c = 0; c < P; ++c)
      for (int d = c; d < P; ++d, ++i) {
        T s = T(0);
        for (int w = 0; w < 16; ++w) s += part[w][i];          // fixed order
        Gr[c][d] = s;
        Gr[d][c] = s;
      }
    int bad = 0;
    T pmax = T(0), pmin = T(0);                                // largest / smallest squared pivot
    for (int r = 0; r < P; ++r)
      for (int c = 0; c < P; ++c) Rr[r][c] = T(0);
    for (int j = 0; j < P; ++j)
      for (int r = 0; r <= j; ++r) {
        T s = Gr[r][j];
        for (int m = 0; m < r; ++m) s -= Rr[m][r] * Rr[m][j];
        if (r == j) {
          if (!(s > T(0))) { if (!bad) bad = j + 1; s = T(1); }
          pmax = j == 0 ? s : (s > pmax ? s : pmax);
          pmin = j == 0 ? s : (s < pmin ? s : pmin);
          Rr[j][j] = sqrt(s);
        } else {
          Rr[r][j] = s / Rr[r][r];
        }
      }
    // W = R^-1 (upper triangular), column by column: R W = I
    for (int c = 0; c < P; ++c) {
      T colr[P];
      for (int r = P - 1; r >= 0; --r) {
        T s = (r == c) ? T(1) : T(0);
        for (int m = r + 1; m <= c; ++m) s -= Rr[r][m] * colr[m];
        colr[r] = (r <= c) ? s / Rr[r][r] : T(0);
      }
      for (int r = 0; r < P; ++r) Wsh[r][c] = colr[r];
    }
    if (bad) info[b] = bad;
    if (cond) {                                                // (pivot ratio)^2 ~ condition number of the panel, squared
      const T ratio = bad ? T(INFINITY) : pmax / pmin;
      const T old = cond[b];
      cond[b] = (ratio > old || ratio != ratio) ? ratio : old;
    }
  }
  if (SHIFT && tid == 0) {
    // shifted variant (re-orthogonalised passes only): G, R, col in LDS and rolled loops — with the shift in the
    // register version above the 1024-thread kernel's 128-VGPR budget spilled 554 registers at P = 6 (35 -> 75 us for
    // EVERY call); this one costs 58 us where it runs
    int i = 0;
#pragma unroll 1
    for (int c = 0; c < P; ++c)
#pragma unroll 1
      for (int d = c; d < P; ++d, ++i) {
        T s = T(0);
#pragma unroll 1
        for (int w = 0; w < 16; ++w) s += part[w][i];          // fixed order
        G[c][d] = s;
        G[d][c] = s;
      }
    if (shift_rel > T(0)) {                                    // shifted CholeskyQR: G + (shift_rel trace G) I
      T tr = T(0);
#pragma unroll 1
      for (int c = 0; c < P; ++c) tr += G[c][c];
#pragma unroll 1
      for (int c = 0; c < P; ++c) G[c][c] += shift_rel * tr;
    }
    int bad = 0;
    T pmax = T(0), pmin = T(0);
#pragma unroll 1
    for (int r = 0; r < P; ++r)
#pragma unroll 1
      for (int c = 0; c < P; ++c) R[r][c] = T(0);
#pragma unroll 1
    for (int j = 0; j < P; ++j)
#pragma unroll 1
      for (int r = 0; r <= j; ++r) {
        T s = G[r][j];
#pragma unroll 1
        for (int m = 0; m < r; ++m) s -= R[m][r] * R[m][j];
        if (r == j) {
          if (!(s > T(0))) { if (!bad) bad = j + 1; s = T(1); }
          pmax = j == 0 ? s : (s > pmax ? s : pmax);
          pmin = j == 0 ? s : (s < pmin ? s : pmin);
          R[j][j] = sqrt(s);
        } else {
          R[r][j] = s / R[r][r];
        }
      }
    // W = R^-1 (upper triangular), column by column: R W = I
#pragma unroll 1
    for (int c = 0; c < P; ++c) {
#pragma unroll 1
      for (int r = P - 1; r >= 0; --r) {
        T s = (r == c) ? T(1) : T(0);
#pragma unroll 1
        for (int m = r + 1; m <= c; ++m) s -= R[r][m] * col[m];
        col[r] = (r <= c) ? s / R[r][r] : T(0);
      }
#pragma unroll 1
      for (int r = 0; r < P; ++r) Wsh[r][c] = col[r];
    }
    if (bad) info[b] = bad;
    if (cond) {
      const T ratio = bad ? T(INFINITY) : pmax / pmin;
      const T old = cond[b];
      cond[b] = (ratio > old || ratio != ratio) ? ratio : old;
    }
  }
  __syncthreads();
  for (int j = tid * VN; j < N; j += 1024 * VN) {
    VT t[P];
#pragma unroll
    for (int c = 0; c < P; ++c) t[c] = *reinterpret_cast<const VT*>(Tb + (long)c * ldt + j);
#pragma unroll
    for (int c = P - 1; c >= 0; --c) {
      VT acc;
#pragma unroll
      for (int v = 0; v < VN; ++v) acc[v] = T(0);
#pragma unroll
      for (int a = 0; a <= c; ++a) {
        const T w = chain_uniform(Wsh[a][c]);                   // the same for every lane: lives in SGPRs
#pragma unroll
        for (int v = 0; v < VN; ++v) acc[v] += w * t[a][v];
      }
      *reinterpret_cast<VT*>(Tb + (long)c * ldt + j) = acc;
    }
  }
}

// T[b, k0+c, a] = Tn[b, c, a] (a < k0+q) and its mirror T[b, a, k0+c] = Tn[b, c, a] (a < k0)
template <typename T>
__global__ __launch_bounds__(256) void t_scatter_kernel(const T* __restrict__ Tn, T* __restrict__ Tm, int k0, int q,
                                                        long ldt, long sT, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int kq = k0 + q;
  const long per_b = (long)q * kq;
  const long b = idx / per_b;
  const long rem = idx - b * per_b;
  const int c = (int)(rem / kq), a = (int)(rem - (long)c * kq);
  const T v = Tn[idx];
  T* Tb = Tm + b * sT;
  Tb[(long)(k0 + c) * ldt + a] = v;
  if (a < k0) Tb[(long)a * ldt + k0 + c] = v;
}

// G[b] += shift_rel * trace(G[b]) * I for the (B, q, q) Gram blocks of the wide-panel path
template <typename T>
__global__ __launch_bounds__(64) void gram_shift_kernel(T* __restrict__ G, int B, int q, T shift_rel) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  T* Gb = G + (long)b * q * q;
  T tr = T(0);
  for (int c = 0; c < q; ++c) tr += Gb[(long)c * q + c];
  for (int c = 0; c < q; ++c) Gb[(long)c * q + c] += shift_rel * tr;
}

template <typename T>
static int cholqr_fused(T* Tp, int* info, int B, int P, int N, long ldt, long sT, T shift_rel, T* cond,
                        hipStream_t st) {
  switch (P) {
#define XK_CASE(PP)                                                                                        \
  case PP:                                                                                                 \
    if (shift_rel > T(0))                                                                                  \
      hipLaunchKernelGGL((panel_cholqr_kernel<T, PP, true>), dim3(B), dim3(1024), 0, st, Tp, info, N, ldt, \
                         sT, shift_rel, cond);                                                             \
    else                                                                                                   \
      hipLaunchKernelGGL((panel_cholqr_kernel<T, PP, false>), dim3(B), dim3(1024), 0, st, Tp, info, N,     \
                         ldt, sT, shift_rel, cond);                                                        \
    break;
    XK_CASE(1) XK_CASE(2) XK_CASE(3) XK_CASE(4) XK_CASE(5) XK_CASE(6) XK_CASE(7) XK_CASE(8)
#undef XK_CASE
    default: return XK_ERR_UNSUPPORTED;
  }
  XK_LAUNCH_CHECK();
  return XK_OK;
}

// CholeskyQR of panel rows [k0, k0 + q) (q <= 32), optionally shifted
template <typename T>
static int panel_cholqr(T* V, int B, int N, int k0, int q, long ldv, long sV, T* C, T* W, int* info, T* ws,
                        long ws_elems, T shift_rel, T* cond, void* stream) {
  constexpr int VN = Vec16<T>::n;
  T* panel = V + (long)k0 * ldv;
  if (q <= 8)
    return cholqr_fused<T>(panel, info, B, q, (N + VN - 1) / VN * VN, ldv, sV, shift_rel, cond, (hipStream_t)stream);
  // wider panels: Gram on K1, Cholesky + inverse per member, transform
  T* G = C;                                               // (B, q, q) fits: the caller sizes C for q * max(k0, q)
  int rc = dense_mm(panel, panel, G, ws, ws_elems, B, q, N, q, ldv, sV, ldv, sV, (long)q, (long)q * q, stream);
  if (rc != XK_OK) return rc;
  if (shift_rel > T(0)) {
    hipLaunchKernelGGL((gram_shift_kernel<T>), dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, G, B, q, shift_rel);
    XK_LAUNCH_CHECK();
  }
  rc = chol_c(G, W, info, B, q, (long)q, (long)q * q, stream);
  if (rc != XK_OK) return rc;
  return transform_c(panel, W, B, q, (int)ldv, ldv, sV, stream);
}

// One pass (the default of the un-restarted iteration: a Ritz residual block is orthogonal to the basis up to
// rounding and its Gram matrix is benign): projection, CholeskyQR.  Two or more: block Gram-Schmidt with
// re-orthogonalisation in the order that survives ill-conditioned panels — [projection, CholeskyQR] per pass, the
// first CholeskyQR shifted: a panel of nearly dependent residuals (wide blocks close to convergence: Gram spectrum
// over 15 decades measured) otherwise loses (a) positive definiteness of its Gram matrix in floating point and (b)
// through R^-1 the orthogonality against the basis that the projections before it had established.
template <typename T>
static int davidson_orth_block(T* V, int B, int N, int k0, int q, long ldv, long sV, T* C, T* W, int* info, T* ws,
                               long ws_elems, int passes, T* cond, void* stream) {
  T* panel = V + (long)k0 * ldv;
  const double u = sizeof(T) == 8 ? 1.1102230246251565e-16 : 5.9604644775390625e-08;
  double sh = 11.0 * ((double)N * q + (double)q * (q + 1)) * u;
  if (sh > 1e-3) sh = 1e-3;
  const int rounds = passes >= 2 ? passes : 1;
  int rc;
  for (int it = 0; it < rounds; ++it) {
    if (k0 > 0 && passes >= 1) {
      // C[b,c,a] = <V_a, t_c> (a < k0), then t_c -= sum_a C[b,c,a] V_a   (tensor.py:15-18 restricted to the new block)
      rc = dense_mm(V, panel, C, ws, ws_elems, B, k0, N, q, ldv, sV, ldv, sV, (long)k0, (long)q * k0, stream);
      if (rc != XK_OK) return rc;
      rc = lincomb_c(V, C, panel, B, k0, (int)ldv, q, ldv, sV, (long)q * k0, 1L, (long)k0, ldv, sV, -1.0, 1.0, stream);
      if (rc != XK_OK) return rc;
    }
    const T shift = (rounds >= 2 && it == 0) ? (T)sh : T(0);
    // (the pivot ratio is reported for the raw panel only: the first CholeskyQR of the call)
    rc = panel_cholqr<T>(V, B, N, k0, q, ldv, sV, C, W, info, ws, ws_elems, shift, it == 0 ? cond : (T*)nullptr, stream);
    if (rc != XK_OK) return rc;
  }
  return XK_OK;
}

// Panels wider than the 32 columns of the per-member Cholesky kernel are taken 32 rows at a time: every chunk is
// orthogonalised (twice: its vectors are not nearly orthogonal to the earlier chunks of the same panel, unlike a
// residual block against the basis) against everything before it, then among itself — block Gram-Schmidt with
// CholeskyQR inside the blocks, the same Q as one CholeskyQR of the whole panel in exact arithmetic.  The reference has
// no width limit (tallqr, _utils/tensor.py:8-19; symeig.py:100-140 for any neig / nguess).
template <typename T>
static int davidson_orth(T* V, int B, int N, int k0, int q, long ldv, long sV, T* C, T* W, int* info, T* ws,
                         long ws_elems, int passes, T* cond, void* stream) {
  constexpr int VN = Vec16<T>::n;
  if ((ldv % VN) || (sV % VN) || ((uintptr_t)V & 15) || ldv < (long)((N + VN - 1) / VN) * VN) return XK_ERR_UNSUPPORTED;
  for (int off = 0; off < q; off += 32) {
    const int qc = q - off < 32 ? q - off : 32;
    const int np = off == 0 ? passes : (passes > 2 ? passes : 2);
    const int rc = davidson_orth_block<T>(V, B, N, k0 + off, qc, ldv, sV, C, W, info, ws, ws_elems, np, cond, stream);
    if (rc != XK_OK) return rc;
  }
  return XK_OK;
}

// orth[b] = max(orth[b], max|X^T (MX) - I|) for the P-row panels X, MX (MX == X: plain Gram).  Gs: scratch of B*P*P
// elements and ws: xk_dense_mm_workspace_elems(B, P, N, P, 0), both only used for P > 8.
template <typename T>
static int ritz_guard(const T* X, const T* MX, T* orth, int B, int N, int P, long ldx, long sX, long ldm, long sM,
                      T* Gs, long gs_elems, T* ws, long ws_elems, void* stream) {
  constexpr int VN = Vec16<T>::n;
  const bool vec = !(ldx % VN) && !(sX % VN) && !(ldm % VN) && !(sM % VN) && !((uintptr_t)X & 15) &&
                   !((uintptr_t)MX & 15) && ldx >= (long)((N + VN - 1) / VN) * VN &&
                   ldm >= (long)((N + VN - 1) / VN) * VN;
  if (P <= 8 && vec) {
    const int Nv = (N + VN - 1) / VN * VN;
    switch (P) {
#define XK_CASE(PP)                                                                                            \
  case PP:                                                                                                     \
    hipLaunchKernelGGL((ritz_guard_kernel<T, PP>), dim3(B), dim3(1024), 0, (hipStream_t)stream, X, MX, orth,   \
                       Nv, ldx, sX, ldm, sM);                                                                  \
    break;
      XK_CASE(1) XK_CASE(2) XK_CASE(3) XK_CASE(4) XK_CASE(5) XK_CASE(6) XK_CASE(7) XK_CASE(8)
#undef XK_CASE
    }
    XK_LAUNCH_CHECK();
    return XK_OK;
  }
  if (!Gs || gs_elems < (long)B * P * P) return XK_ERR_ARG;
  int rc = dense_mm(X, MX, Gs, ws, ws_elems, B, P, N, P, ldx, sX, ldm, sM, (long)P, (long)P * P, stream);
  if (rc != XK_OK) return rc;
  hipLaunchKernelGGL((gram_guard_kernel<T>), dim3(B), dim3(64), 0, (hipStream_t)stream, Gs, orth, P);
  XK_LAUNCH_CHECK();
  return XK_OK;
}

template <typename T>
static int davidson_extend_t(const T* V, const T* AV, T* Tm, T* Tn, int B, int N, int k0, int q, long ldv, long sV,
                             long ldav, long sAV, long ldt, long sT, T* ws, long ws_elems, void* stream) {
  const int kq = k0 + q;
  // Tn[b, c, a] = <V_a, (A V)_{k0+c}>, a < k0+q   (symeig.py:170 restricted to the new rows)
  int rc = dense_mm(V, AV + (long)k0 * ldav, Tn, ws, ws_elems, B, kq, N, q, ldv, sV, ldav, sAV, (long)kq, (long)q * kq,
                    stream);
  if (rc != XK_OK) return rc;
  const long total = (long)B * q * kq;
  hipLaunchKernelGGL((t_scatter_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     Tn, Tm, k0, q, ldt, sT, total);
  XK_LAUNCH_CHECK();
  return XK_OK;
}

}  // namespace xk

extern "C" {

#define XK_DEFINE_CHAIN(SUF, T)                                                                                    \
  int xk_davidson_ritz_##SUF(const T* V, const T* AV, const T* Y, const T* lam, T* X, T* Tn, T* rmax,              \
                             const int* info, const int* flag, T* cond, T* orth, double* status, int B, int k,     \
                             int N, int P,                                                                         \
                             long ldv, long sV, long ldav, long sAV, long sY, long sYa, long sYc, long sLam,       \
                             long ldx, long sX, long ldt, long sT, T* Gs, long gs_elems, T* ws, long ws_elems,     \
                             void* stream) {                                                                       \
    if (B <= 0 || k <= 0 || N <= 0 || P <= 0 || !rmax || !info || !status) return XK_ERR_ARG;                      \
    int rc = xk::ritz_c(V, AV, Y, lam, X, Tn, rmax, B, k, N, P, ldv, sV, ldav, sAV, sY, sYa, sYc, sLam, ldx, sX,   \
                        ldt, sT, stream);                                                                          \
    if (rc != XK_OK) return rc;                                                                                    \
    if (orth) {                                                                                                    \
      rc = xk::ritz_guard<T>(X, X, orth, B, N, P, ldx, sX, ldx, sX, Gs, gs_elems, ws, ws_elems, stream);           \
      if (rc != XK_OK) return rc;                                                                                  \
    }                                                                                                              \
    hipLaunchKernelGGL((xk::group_status_rezero_kernel<T>), dim3(1), dim3(64), 0, (hipStream_t)stream, rmax, info,  \
                       flag, cond, orth, status, B);                                                               \
    XK_LAUNCH_CHECK();                                                                                             \
    return XK_OK;                                                                                                  \
  }                                                                                                                \
  int xk_ritz_guard_##SUF(const T* X, const T* MX, T* orth, int B, int N, int P, long ldx, long sX, long ldm,      \
                          long sM, T* Gs, long gs_elems, T* ws, long ws_elems, void* stream) {                     \
    if (B < 0 || N <= 0 || P <= 0 || !X || !orth) return XK_ERR_ARG;                                               \
    if (B == 0) return XK_OK;                                                                                      \
    if (!MX) { MX = X; ldm = ldx; sM = sX; }                                                                       \
    return xk::ritz_guard<T>(X, MX, orth, B, N, P, ldx, sX, ldm, sM, Gs, gs_elems, ws, ws_elems, stream);          \
  }                                                                                                                \
  int xk_davidson_orth_##SUF(T* V, int B, int N, int k0, int q, long ldv, long sV, T* C, T* W, int* info,          \
                             T* cond, T* ws, long ws_elems, int passes, void* stream) {                            \
    if (B < 0 || N <= 0 || k0 < 0 || q <= 0 || passes < 0) return XK_ERR_ARG;                            \
    if (B == 0) return XK_OK;                                                                                      \
    return xk::davidson_orth<T>(V, B, N, k0, q, ldv, sV, C, W, info, ws, ws_elems, passes, cond, stream);          \
  }                                                                                                                \
  int xk_davidson_extend_t_##SUF(const T* V, const T* AV, T* Tm, T* Tn, int B, int N, int k0, int q, long ldv,     \
                                 long sV, long ldav, long sAV, long ldt, long sT, T* ws, long ws_elems,            \
                                 void* stream) {                                                                   \
    if (B < 0 || N <= 0 || k0 < 0 || q <= 0) return XK_ERR_ARG;                                                    \
    if (B == 0) return XK_OK;                                                                                      \
    return xk::davidson_extend_t<T>(V, AV, Tm, Tn, B, N, k0, q, ldv, sV, ldav, sAV, ldt, sT, ws, ws_elems,         \
                                    stream);                                                                       \
  }

XK_DEFINE_CHAIN(f64, double)
XK_DEFINE_CHAIN(f32, float)

}  // extern "C"
