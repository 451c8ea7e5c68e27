// xitorch_amd :: basis-maintenance kernels of the block eigensolver (K4/K5/K6).
//
// The reference recomputes, every Davidson iteration, T = V^T AV over the whole
// basis, the Ritz rotations V Y / AV Y, the residual and a full CholeskyQR of
// [V, t] (xitorch/_impls/linalg/symeig.py:170-223, _utils/tensor.py:8-19).
// Here the basis is stored PANEL-MAJOR, (B, k, N) with every basis vector
// contiguous, rows are appended in place, and each step is one fused pass:
//
//   xk_lincomb        Out[b,c,:] = beta*Out[b,c,:] + alpha * sum_a C[b,a,c] V[b,a,:]
//   xk_ritz_residual  X = Y^T V, AX = Y^T AV, t = -(AX - lam*X), rmax[b] = max|AX - lam*X|
//   xk_panel_chol     upper Cholesky of the p x p panel Gram + its inverse (CholeskyQR step)
//   xk_panel_transform in-place t <- W^T t with W upper triangular
//   xk_colnorm2 / xk_fill helpers
//
// All of them are HBM-bound streams over (k or p) x N panels: 16 B/lane
// coalesced loads, coefficients are wave-uniform (scalar loads), no atomics
// except the order-independent max.
#include "xk_common.h"

namespace xk {

template <typename T, int P>
__global__ __launch_bounds__(256) void lincomb_kernel(
    const T* __restrict__ V, const T* __restrict__ C, T* __restrict__ Out,
    int k, int N, long ldv, long sV, long sC, long sCa, long sCc, long ldo, long sO,
    T alpha, T beta, int col_tiles) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  const int b = blockIdx.x / col_tiles;
  const int ct = blockIdx.x - b * col_tiles;
  const int j = (ct * 256 + threadIdx.x) * VN;
  if (j >= N) return;
  const T* Vb = V + (long)b * sV + j;
  const T* Cb = C + (long)b * sC;
  VT acc[P];
#pragma unroll
  for (int c = 0; c < P; ++c)
#pragma unroll
    for (int v = 0; v < VN; ++v) acc[c][v] = T(0);
  int a = 0;
  // eight basis rows per trip: with few batch members the launch has only a few waves per CU, and what bounds it is
  // the number of loads each of them keeps in flight (4 ops x k = 57 x N = 16384: 52 us with four, r03)
  for (; a + 8 <= k; a += 8) {
    VT v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const VT*>(Vb + (long)(a + u) * ldv);
#pragma unroll
    for (int c = 0; c < P; ++c) {
      T cc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cc[u] = Cb[(long)(a + u) * sCa + c * sCc];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < VN; ++e) acc[c][e] += cc[u] * v[u][e];
    }
  }
  for (; a + 4 <= k; a += 4) {
    VT v0 = *reinterpret_cast<const VT*>(Vb + (long)(a + 0) * ldv);
    VT v1 = *reinterpret_cast<const VT*>(Vb + (long)(a + 1) * ldv);
    VT v2 = *reinterpret_cast<const VT*>(Vb + (long)(a + 2) * ldv);
    VT v3 = *reinterpret_cast<const VT*>(Vb + (long)(a + 3) * ldv);
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const T c0 = Cb[(long)(a + 0) * sCa + c * sCc], c1 = Cb[(long)(a + 1) * sCa + c * sCc];
      const T c2 = Cb[(long)(a + 2) * sCa + c * sCc], c3 = Cb[(long)(a + 3) * sCa + c * sCc];
#pragma unroll
      for (int v = 0; v < VN; ++v) {
        acc[c][v] += c0 * v0[v];
        acc[c][v] += c1 * v1[v];
        acc[c][v] += c2 * v2[v];
        acc[c][v] += c3 * v3[v];
      }
    }
  }
  for (; a < k; ++a) {
    VT v0 = *reinterpret_cast<const VT*>(Vb + (long)a * ldv);
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const T c0 = Cb[(long)a * sCa + c * sCc];
#pragma unroll
      for (int v = 0; v < VN; ++v) acc[c][v] += c0 * v0[v];
    }
  }
  T* Ob = Out + (long)b * sO + j;
#pragma unroll
  for (int c = 0; c < P; ++c) {
    VT o;
    if (beta != T(0)) {
      o = *reinterpret_cast<const VT*>(Ob + (long)c * ldo);
#pragma unroll
      for (int v = 0; v < VN; ++v) o[v] = beta * o[v] + alpha * acc[c][v];
    } else {
#pragma unroll
      for (int v = 0; v < VN; ++v) o[v] = alpha * acc[c][v];
    }
    *reinterpret_cast<VT*>(Ob + (long)c * ldo) = o;
  }
}

// fused Ritz rotation + residual + per-batch max-norm
template <typename T, int P>
__global__ __launch_bounds__(256) void ritz_residual_kernel(
    const T* __restrict__ V, const T* __restrict__ AV, const T* __restrict__ Y,
    const T* __restrict__ lam, T* __restrict__ X, T* __restrict__ Tn, T* __restrict__ rmax,
    int k, int N, long ldv, long sV, long ldav, long sAV, long sY, long sYa, long sYc, long sLam,
    long ldx, long sX, long ldt, long sT, int col_tiles) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  __shared__ T red[4];
  const int b = blockIdx.x / col_tiles;
  const int ct = blockIdx.x - b * col_tiles;
  const int j = (ct * 256 + threadIdx.x) * VN;
  const bool active = j < N;
  T local_max = T(0);
  if (active) {
    const T* Vb = V + (long)b * sV + j;
    const T* AVb = AV + (long)b * sAV + j;
    const T* Yb = Y + (long)b * sY;
    VT ax[P], xx[P];
#pragma unroll
    for (int c = 0; c < P; ++c)
#pragma unroll
      for (int v = 0; v < VN; ++v) { ax[c][v] = T(0); xx[c][v] = T(0); }
    int a = 0;
    // four basis rows (eight 16 B loads) per trip, see lincomb_kernel
    for (; a + 4 <= k; a += 4) {
      VT v[4], w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = *reinterpret_cast<const VT*>(Vb + (long)(a + u) * ldv);
        w[u] = *reinterpret_cast<const VT*>(AVb + (long)(a + u) * ldav);
      }
#pragma unroll
      for (int c = 0; c < P; ++c) {
        T yy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) yy[u] = Yb[(long)(a + u) * sYa + c * sYc];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < VN; ++e) {
            xx[c][e] += yy[u] * v[u][e];
            ax[c][e] += yy[u] * w[u][e];
          }
      }
    }
    for (; a + 2 <= k; a += 2) {
      VT v0 = *reinterpret_cast<const VT*>(Vb + (long)a * ldv);
      VT v1 = *reinterpret_cast<const VT*>(Vb + (long)(a + 1) * ldv);
      VT w0 = *reinterpret_cast<const VT*>(AVb + (long)a * ldav);
      VT w1 = *reinterpret_cast<const VT*>(AVb + (long)(a + 1) * ldav);
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const T y0 = Yb[(long)a * sYa + c * sYc], y1 = Yb[(long)(a + 1) * sYa + c * sYc];
#pragma unroll
        for (int v = 0; v < VN; ++v) {
          xx[c][v] += y0 * v0[v];
          xx[c][v] += y1 * v1[v];
          ax[c][v] += y0 * w0[v];
          ax[c][v] += y1 * w1[v];
        }
      }
    }
    for (; a < k; ++a) {
      VT v0 = *reinterpret_cast<const VT*>(Vb + (long)a * ldv);
      VT w0 = *reinterpret_cast<const VT*>(AVb + (long)a * ldav);
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const T y0 = Yb[(long)a * sYa + c * sYc];
#pragma unroll
        for (int v = 0; v < VN; ++v) { xx[c][v] += y0 * v0[v]; ax[c][v] += y0 * w0[v]; }
      }
    }
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const T l = lam[(long)b * sLam + c];
      VT t;
#pragma unroll
      for (int v = 0; v < VN; ++v) {
        const T r = ax[c][v] - l * xx[c][v];
        t[v] = -r;
        const T ar = r < T(0) ? -r : r;
        local_max = ar > local_max ? ar : local_max;
      }
      *reinterpret_cast<VT*>(X + (long)b * sX + (long)c * ldx + j) = xx[c];
      *reinterpret_cast<VT*>(Tn + (long)b * sT + (long)c * ldt + j) = t;
    }
  }
  local_max = wave_max(local_max);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local_max;
  __syncthreads();
  if (threadIdx.x == 0) {
    T m = red[0];
    m = red[1] > m ? red[1] : m;
    m = red[2] > m ? red[2] : m;
    m = red[3] > m ? red[3] : m;
    // NaN never compares greater: make it visible as +inf so the host loop cannot "converge" on it
    if (m != m) m = T(INFINITY);
    atomic_max_nonneg(rmax + b, m);
  }
}

// One WAVE per batch member: G = R^T R (upper R), W = R^-1.  P <= 32.
// info[b] = index+1 of the first non-positive pivot (0 = ok) — mirrors the
// reference, where torch.linalg.cholesky raises on a rank-deficient panel.
// (Until round 5 this was one THREAD per member with a 32 x 32 local array: dynamically indexed, so it lived in scratch
// memory — 0.5 ms per call for eight 16 x 16 matrices, 2 ms of every half-iteration of the configs[4] chain.)  Lane j owns
// column j of R (in LDS) and walks down its rows in lock step with the other lanes; every entry is formed by exactly the
// operations, in exactly the order, of the column-by-column scalar algorithm
//     s = (G_ij + G_ji) / 2 - sum_{m < i} R_mi R_mj  (m ascending);   R_jj = sqrt(s);   R_ij = s / R_ii
// so the result is bit-identical to the scalar form.  The inverse: lane c solves R w = e_c by back substitution
// (i descending, inner sum m ascending), again the scalar order.
template <typename T>
__global__ __launch_bounds__(64) void panel_chol_kernel(const T* __restrict__ G, T* __restrict__ W, int* __restrict__ info,
                                                        int B, int P, long ldg, long sG, long sW) {
  __shared__ T R[32][33];
  __shared__ T C[32][33];                   // C[i][c] = entry i of column c of the inverse
  const int b = blockIdx.x;
  const int j = threadIdx.x;                // column of this lane
  if (b >= B) return;
  const T* Gb = G + (long)b * sG;
  const bool own = j < P;
  int bad = 0;
  for (int i = 0; i < P; ++i) {
    T s = T(0);
    if (own && j >= i) {
      // symmetrised entry: G is mathematically symmetric, average the two computed halves
      s = T(0.5) * (Gb[(long)i * ldg + j] + Gb[(long)j * ldg + i]);
      for (int m = 0; m < i; ++m) s -= R[m][i] * R[m][j];
      if (j == i) {
        if (!(s > T(0))) { bad = i + 1; s = T(1); }
        R[i][i] = sqrt(s);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (own && j > i) R[i][j] = s / R[i][i];
    if (own && j < i) R[i][j] = T(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // W = R^-1 (upper triangular), column by column: R W = I
  T* Wb = W + (long)b * sW;
  if (own) {
    const int c = j;
    for (int i = P - 1; i >= 0; --i) {
      T s = (i == c) ? T(1) : T(0);
      for (int m = i + 1; m <= c; ++m) s -= R[i][m] * C[m][c];
      C[i][c] = (i <= c) ? s / R[i][i] : T(0);
    }
    for (int i = 0; i < P; ++i) Wb[(long)i * P + c] = C[i][c];
  }
  // sticky: a later pass over the same panel (CholeskyQR2) must not erase an earlier breakdown; the host
  // allocates info zeroed and raises as soon as it reads a non-zero flag.  The FIRST non-positive pivot is reported:
  // lane j can only have flagged pivot j, so the smallest flagged lane is it.
  const unsigned long long flagged = __ballot(bad != 0);
  if (flagged && j == (int)(__ffsll((long long)flagged) - 1)) info[b] = bad;
}

// in-place t[c,:] <- sum_{a<=c} W[a,c] t[a,:], W upper triangular (B,P,P) row-major.
// Processing c from high to low keeps the still-needed rows untouched.
template <typename T>
__global__ __launch_bounds__(256) void panel_transform_kernel(
    T* __restrict__ Tp, const T* __restrict__ W, int P, int N, long ldt, long sT, long sW, int col_tiles) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  const int b = blockIdx.x / col_tiles;
  const int ct = blockIdx.x - b * col_tiles;
  const int j = (ct * 256 + threadIdx.x) * VN;
  if (j >= N) return;
  T* Tb = Tp + (long)b * sT + j;
  const T* Wb = W + (long)b * sW;
  for (int c = P - 1; c >= 0; --c) {
    VT acc;
#pragma unroll
    for (int v = 0; v < VN; ++v) acc[v] = T(0);
    for (int a = 0; a <= c; ++a) {
      const T w = Wb[(long)a * P + c];
      VT t = *reinterpret_cast<const VT*>(Tb + (long)a * ldt);
#pragma unroll
      for (int v = 0; v < VN; ++v) acc[v] += w * t[v];
    }
    *reinterpret_cast<VT*>(Tb + (long)c * ldt) = acc;
  }
}

// Panels handed to these kernels must be PADDED: pitch ld a multiple of the 16 B vector width and
// >= N rounded up to it, batch pitch likewise, base 16 B aligned, pad elements zero.  Lanes then
// always move whole 16 B vectors; the (zero) pad lanes compute zeros.
template <typename T>
static bool vec_ok(int N, long ld, long s, const void* p) {
  constexpr int VN = Vec16<T>::n;
  const long npad = ((long)N + VN - 1) / VN * VN;
  return (ld % VN == 0) && (ld >= npad || ld == 0) && (s % VN == 0) && (((uintptr_t)p & 15) == 0);
}

template <typename T>
static int lincomb(const T* V, const T* C, T* Out, int B, int k, int N, int P, long ldv, long sV,
                   long sC, long sCa, long sCc, long ldo, long sO, T alpha, T beta, hipStream_t st) {
  constexpr int VN = Vec16<T>::n;
  if (!vec_ok<T>(N, ldv, sV, V) || !vec_ok<T>(N, ldo, sO, Out)) return XK_ERR_UNSUPPORTED;
  const int ct = (N + 256 * VN - 1) / (256 * VN);
  int c0 = 0;
  while (c0 < P) {
    const int pc = (P - c0) >= 8 ? 8 : (P - c0);
    const T* Cc = C + (long)c0 * sCc;
    T* Oc = Out + (long)c0 * ldo;
    switch (pc) {
#define XK_CASE(PP)                                                                                   \
  case PP:                                                                                            \
    hipLaunchKernelGGL((lincomb_kernel<T, PP>), dim3((unsigned)((long)B * ct)), dim3(256), 0, st, V, \
                       Cc, Oc, k, N, ldv, sV, sC, sCa, sCc, ldo, sO, alpha, beta, ct);                \
    break;
      XK_CASE(1) XK_CASE(2) XK_CASE(3) XK_CASE(4) XK_CASE(5) XK_CASE(6) XK_CASE(7) XK_CASE(8)
#undef XK_CASE
    }
    XK_LAUNCH_CHECK();
    c0 += pc;
  }
  return XK_OK;
}

template <typename T>
static int ritz_residual(const T* V, const T* AV, const T* Y, const T* lam, T* X, T* Tn, T* rmax, int B,
                         int k, int N, int P, long ldv, long sV, long ldav, long sAV, long sY, long sYa,
                         long sYc, long sLam, long ldx, long sX, long ldt, long sT, hipStream_t st) {
  constexpr int VN = Vec16<T>::n;
  if (!vec_ok<T>(N, ldv, sV, V) || !vec_ok<T>(N, ldav, sAV, AV) || !vec_ok<T>(N, ldx, sX, X) ||
      !vec_ok<T>(N, ldt, sT, Tn))
    return XK_ERR_UNSUPPORTED;
  const int ct = (N + 256 * VN - 1) / (256 * VN);
  int c0 = 0;
  while (c0 < P) {
    const int pc = (P - c0) >= 8 ? 8 : (P - c0);
    switch (pc) {
#define XK_CASE(PP)                                                                                    \
  case PP:                                                                                             \
    hipLaunchKernelGGL((ritz_residual_kernel<T, PP>), dim3((unsigned)((long)B * ct)), dim3(256), 0,   \
                       st, V, AV, Y + (long)c0 * sYc, lam + c0, X + (long)c0 * ldx, Tn + (long)c0 * ldt, \
                       rmax, k, N, ldv, sV, ldav, sAV, sY, sYa, sYc, sLam, ldx, sX, ldt, sT, ct);       \
    break;
      XK_CASE(1) XK_CASE(2) XK_CASE(3) XK_CASE(4) XK_CASE(5) XK_CASE(6) XK_CASE(7) XK_CASE(8)
#undef XK_CASE
    }
    XK_LAUNCH_CHECK();
    c0 += pc;
  }
  return XK_OK;
}


// Davidson's diagonal correction of the new search directions (the reference has none, symeig.py:206-207):
//   t[b,c,n] <- t[b,c,n] / (d[b,n] - lam[b,c] * m[b,n]),   |denominator| kept >= floor (sign preserved)
// d = diag(A), m = diag(M) (nullptr: identity).  One thread per element, coalesced along n.
template <typename T>
__global__ __launch_bounds__(256) void diag_precond_kernel(T* __restrict__ Tn, const T* __restrict__ d,
                                                            const T* __restrict__ m, const T* __restrict__ lam,
                                                            int N, int P, long ldt, long sT, long sD, long sM,
                                                            long sLam, T floor_, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long per_b = (long)P * N;
  const long b = idx / per_b;
  const long rem = idx - b * per_b;
  const int c = (int)(rem / N);
  const int n = (int)(rem - (long)c * N);
  const T mm = m ? m[b * sM + n] : T(1);
  T den = d[b * sD + n] - lam[b * sLam + c] * mm;
  if (fabs(den) < floor_) den = den < T(0) ? -floor_ : floor_;
  T* t = Tn + b * sT + (long)c * ldt + n;
  *t = *t / den;
}

// ---------------------------------------------------------------------------------------------
// Group status in one launch: status = {max_b rmax[b] (NaN if any is NaN), max_b info[b], max_b flag[b]} as doubles;
// with orth (the a-posteriori guard values of xk_ritz_guard, NaN counts as infinite) also status[4] = max_b orth[b],
// orth <- 0 (status[3] is the condition estimate of the fused chain, untouched here).
// Replaces three torch reductions + three converting element copies per Rayleigh-Ritz step (eight tiny launches on
// the critical chain of a small batch group).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void group_status_kernel(const T* __restrict__ rmax, const int* __restrict__ info,
                                                          const int* __restrict__ flag, T* __restrict__ orth,
                                                          double* __restrict__ status, int B) {
  const int lane = threadIdx.x;
  double m = 0.0, om_ = 0.0;
  int nan = 0, i1 = 0, i2 = 0;
  bool first = true;
  for (int b = lane; b < B; b += 64) {
    const double v = (double)rmax[b];
    nan |= (v != v);
    m = first ? v : (v > m ? v : m);
    const int a = info[b];
    i1 = first ? a : (a > i1 ? a : i1);
    if (flag) {
      const int f = flag[b];
      i2 = first ? f : (f > i2 ? f : i2);
    }
    if (orth) {
      const double ov = (double)orth[b];
      orth[b] = T(0);
      om_ = (ov != ov) ? __builtin_inf() : (ov > om_ ? ov : om_);
    }
    first = false;
  }
  // lanes without an element must not contribute: fold with explicit validity
  double mm = first ? -__builtin_inf() : m;
  int a1 = first ? -2147483647 - 1 : i1, a2 = first ? -2147483647 - 1 : i2;
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) {
    const double om = __shfl_xor(mm, sft, 64), oo = __shfl_xor(om_, sft, 64);
    const int o1 = __shfl_xor(a1, sft, 64), o2 = __shfl_xor(a2, sft, 64), on = __shfl_xor(nan, sft, 64);
    mm = om > mm ? om : mm;
    om_ = oo > om_ ? oo : om_;
    a1 = o1 > a1 ? o1 : a1;
    a2 = o2 > a2 ? o2 : a2;
    nan |= on;
  }
  if (lane == 0) {
    status[0] = nan ? __builtin_nan("") : mm;
    status[1] = (double)a1;
    status[2] = flag ? (double)a2 : 0.0;
    if (orth) status[4] = om_;
  }
}

}  // namespace xk

extern "C" {

#define XK_DEFINE_BASIS(SUF, T)                                                                          \
  int xk_lincomb_##SUF(const T* V, const T* C, T* Out, int B, int k, int N, int P, long ldv, long sV,    \
                       long sC, long sCa, long sCc, long ldo, long sO, double alpha, double beta,        \
                       void* stream) {                                                                   \
    if (B < 0 || k < 0 || N < 0 || P < 0) return XK_ERR_ARG;                                             \
    if (B == 0 || N == 0 || P == 0) return XK_OK;                                                        \
    return xk::lincomb<T>(V, C, Out, B, k, N, P, ldv, sV, sC, sCa, sCc, ldo, sO, (T)alpha, (T)beta,      \
                          (hipStream_t)stream);                                                          \
  }                                                                                                      \
  int xk_ritz_residual_##SUF(const T* V, const T* AV, const T* Y, const T* lam, T* X, T* Tn, T* rmax,    \
                             int B, int k, int N, int P, long ldv, long sV, long ldav, long sAV,         \
                             long sY, long sYa, long sYc, long sLam, long ldx, long sX, long ldt,        \
                             long sT, void* stream) {                                                    \
    if (B < 0 || k < 0 || N < 0 || P < 0) return XK_ERR_ARG;                                             \
    if (B == 0 || N == 0 || P == 0) return XK_OK;                                                        \
    return xk::ritz_residual<T>(V, AV, Y, lam, X, Tn, rmax, B, k, N, P, ldv, sV, ldav, sAV, sY, sYa,     \
                                sYc, sLam, ldx, sX, ldt, sT, (hipStream_t)stream);                       \
  }                                                                                                      \
  int xk_panel_chol_##SUF(const T* G, T* W, int* info, int B, int P, long ldg, long sG, void* stream) {  \
    if (B < 0 || P < 0 || P > 32) return XK_ERR_ARG;                                                     \
    if (B == 0 || P == 0) return XK_OK;                                                                  \
    hipLaunchKernelGGL((xk::panel_chol_kernel<T>), dim3((unsigned)B), dim3(64), 0,                       \
                       (hipStream_t)stream, G, W, info, B, P, ldg, sG, (long)P * P);                     \
    XK_LAUNCH_CHECK();                                                                                   \
    return XK_OK;                                                                                        \
  }                                                                                                      \
  int xk_panel_transform_##SUF(T* Tp, const T* W, int B, int P, int N, long ldt, long sT,                \
                               void* stream) {                                                           \
    if (B < 0 || P < 0 || N < 0) return XK_ERR_ARG;                                                      \
    if (B == 0 || P == 0 || N == 0) return XK_OK;                                                        \
    if (!xk::vec_ok<T>(N, ldt, sT, Tp)) return XK_ERR_UNSUPPORTED;                                       \
    constexpr int VN = xk::Vec16<T>::n;                                                                  \
    const int ct = (N + 256 * VN - 1) / (256 * VN);                                                      \
    hipLaunchKernelGGL((xk::panel_transform_kernel<T>), dim3((unsigned)((long)B * ct)), dim3(256), 0,    \
                       (hipStream_t)stream, Tp, W, P, N, ldt, sT, (long)P * P, ct);                      \
    XK_LAUNCH_CHECK();                                                                                   \
    return XK_OK;                                                                                        \
  }

XK_DEFINE_BASIS(f64, double)
XK_DEFINE_BASIS(f32, float)

#define XK_DEFINE_STATUS(SUF, T)                                                                            \
  int xk_group_status_##SUF(const T* rmax, const int* info, const int* flag, T* orth, double* status, int B, \
                            void* stream) {                                                                 \
    if (B <= 0 || !rmax || !info || !status) return XK_ERR_ARG;                                             \
    hipLaunchKernelGGL((xk::group_status_kernel<T>), dim3(1), dim3(64), 0, (hipStream_t)stream, rmax, info, \
                       flag, orth, status, B);                                                              \
    XK_LAUNCH_CHECK();                                                                                      \
    return XK_OK;                                                                                           \
  }
XK_DEFINE_STATUS(f64, double)
XK_DEFINE_STATUS(f32, float)

#define XK_DEFINE_PRECOND(SUF, T)                                                                          \
  int xk_diag_precond_##SUF(T* Tn, const T* d, const T* m, const T* lam, int B, int N, int P, long ldt,    \
                            long sT, long sD, long sM, long sLam, double floor_, void* stream) {           \
    if (B < 0 || N < 0 || P < 0 || !(floor_ > 0)) return XK_ERR_ARG;                                       \
    if (B == 0 || N == 0 || P == 0) return XK_OK;                                                          \
    const long total = (long)B * P * N;                                                                    \
    hipLaunchKernelGGL((xk::diag_precond_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,  \
                       (hipStream_t)stream, Tn, d, m, lam, N, P, ldt, sT, sD, sM, sLam, (T)floor_, total); \
    XK_LAUNCH_CHECK();                                                                                     \
    return XK_OK;                                                                                          \
  }
XK_DEFINE_PRECOND(f64, double)
XK_DEFINE_PRECOND(f32, float)

}  // extern "C"
