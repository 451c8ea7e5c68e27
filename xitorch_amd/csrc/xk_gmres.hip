// xitorch_amd :: device-side Hessenberg / Givens state of the native GMRES (xitorch/_impls/linalg/solve.py:326-433).
//
// The reference keeps h (B*c, m+1, m) on the host side of its Python loop, fills one column per iteration by modified
// Gram-Schmidt (:390-393) and solves the (k+1) x k least-squares problem from scratch with torch.linalg.lstsq every
// iteration (:403).  Here every (batch member, column) pair is one *system*; the S systems advance in lock step and
// their small per-system state never leaves the device:
//
//   xk_gmres_step    new Hessenberg column from the two Gram passes of the CGS2 orthogonalisation, previous Givens
//                    rotations replayed on it, new rotation, updated rotated right-hand side g, residual estimate
//                    |g[k+1]| (the least-squares residual of the reference's lstsq, in exact arithmetic the norm of
//                    its explicit residual :414-415), 1 / h[k+1,k] for the normalisation of the next basis vector
//   xk_gmres_finish  q[k+1] = (w - Q c2) / h[k+1,k]: second projection and normalisation (:392,396-398) in one pass
//   xk_gmres_solve   back substitution R y = g of the triangularised system (what lstsq returns, :403)
//
// State layout (double whatever the vector type): R (S, cap+1, cap) row-major per system (row i, column j: the rotated
// Hessenberg; only i <= j is meaningful), cs / sn (S, cap), g (S, cap+1).  One thread per system in xk_gmres_step
// (the rotation replay is a sequential recurrence of k steps), one wave per system in xk_gmres_solve (row i's dot
// product over j > i is lane-parallel, y lives in LDS).
#include "xk_common.h"

namespace xk {

constexpr int GMRES_PART = 64;     // pitch of the |r|^2 partial array shared with xk_kry_status

template <typename T>
__global__ __launch_bounds__(64) void gmres_step_kernel(
    const T* __restrict__ c1, long sc1, const T* __restrict__ c2n, long sc2, int k, int cap,
    double* __restrict__ R, double* __restrict__ cs, double* __restrict__ sn, double* __restrict__ g,
    T* __restrict__ inv_hn, T* __restrict__ est2, int S) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= S) return;
  const T* a1 = c1 + (long)s * sc1;
  const T* a2 = c2n + (long)s * sc2;
  double* Rs = R + (long)s * (cap + 1) * cap;
  double* css = cs + (long)s * cap;
  double* sns = sn + (long)s * cap;
  double* gs = g + (long)s * (cap + 1);
  // h[k+1,k] = |w - Q c1 - Q c2|: the Gram pass 2 ran over [Q; w1] with w1 = w - Q c1, so its last entry is |w1|^2
  // and |w1 - Q c2|^2 = |w1|^2 - |c2|^2 for orthonormal Q (c2 is rounding-sized next to w1: no cancellation)
  double n2 = (double)a2[k + 1];
  for (int j = 0; j <= k; ++j) { const double c = (double)a2[j]; n2 -= c * c; }
  const double hn = n2 > 0.0 ? sqrt(n2) : 0.0;
  // column k of the Hessenberg, h[j,k] = c1[j] + c2[j], with the previous rotations applied on the fly
  double prev = (double)a1[0] + (double)a2[0];
  for (int j = 0; j < k; ++j) {
    const double nxt = (double)a1[j + 1] + (double)a2[j + 1];
    const double c = css[j], t = sns[j];
    Rs[(long)j * cap + k] = c * prev + t * nxt;
    prev = -t * prev + c * nxt;
  }
  const double a = prev, b = hn;
  const double den = sqrt(a * a + b * b);
  const double c = den > 0.0 ? a / den : 1.0;
  const double t = den > 0.0 ? b / den : 0.0;
  css[k] = c;
  sns[k] = t;
  Rs[(long)k * cap + k] = c * a + t * b;
  const double gk = gs[k];
  gs[k] = c * gk;
  const double gn = -t * gk;
  gs[k + 1] = gn;
  inv_hn[s] = hn > 0.0 ? (T)(1.0 / hn) : T(0);
  est2[(long)s * GMRES_PART] = (T)(gn * gn);
}

// q[k+1] = (w1 - sum_{j<=k} c2[j] q[j]) * inv_hn, in place in basis row k+1 (which holds w1).  Lane owns 16 B.
template <typename T>
__global__ __launch_bounds__(256) void gmres_finish_kernel(T* __restrict__ Q, const T* __restrict__ c2n, long sc2,
                                                           const T* __restrict__ inv_hn, int N, int k, long ldq,
                                                           long sQ, int col_tiles) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  const int s = blockIdx.x / col_tiles;
  const int ct = blockIdx.x - s * col_tiles;
  const int j0 = (ct * 256 + threadIdx.x) * VN;
  if (j0 >= N) return;
  T* Qs = Q + (long)s * sQ + j0;
  const T* cc = c2n + (long)s * sc2;
  VT acc;
#pragma unroll
  for (int v = 0; v < VN; ++v) acc[v] = T(0);
  int j = 0;
  for (; j + 4 <= k + 1; j += 4) {
    const VT q0 = *reinterpret_cast<const VT*>(Qs + (long)(j + 0) * ldq);
    const VT q1 = *reinterpret_cast<const VT*>(Qs + (long)(j + 1) * ldq);
    const VT q2 = *reinterpret_cast<const VT*>(Qs + (long)(j + 2) * ldq);
    const VT q3 = *reinterpret_cast<const VT*>(Qs + (long)(j + 3) * ldq);
    const T c0 = cc[j], c1 = cc[j + 1], c2 = cc[j + 2], c3 = cc[j + 3];
#pragma unroll
    for (int v = 0; v < VN; ++v) {
      acc[v] += c0 * q0[v];
      acc[v] += c1 * q1[v];
      acc[v] += c2 * q2[v];
      acc[v] += c3 * q3[v];
    }
  }
  for (; j <= k; ++j) {
    const VT q0 = *reinterpret_cast<const VT*>(Qs + (long)j * ldq);
    const T c0 = cc[j];
#pragma unroll
    for (int v = 0; v < VN; ++v) acc[v] += c0 * q0[v];
  }
  const T sc = inv_hn[s];
  VT w = *reinterpret_cast<const VT*>(Qs + (long)(k + 1) * ldq);
#pragma unroll
  for (int v = 0; v < VN; ++v) w[v] = (w[v] - acc[v]) * sc;
  *reinterpret_cast<VT*>(Qs + (long)(k + 1) * ldq) = w;
}

// back substitution R y = g (upper triangular kd x kd), one wave per system; a zero pivot (breakdown: the Krylov
// space of that system is exhausted) gives y_i = 0
template <typename T>
__global__ __launch_bounds__(64) void gmres_solve_kernel(const double* __restrict__ R, const double* __restrict__ g,
                                                         T* __restrict__ y, long sy, int kd, int cap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* ys = reinterpret_cast<double*>(smem);
  const int s = blockIdx.x;
  const int lane = threadIdx.x;
  const double* Rs = R + (long)s * (cap + 1) * cap;
  const double* gs = g + (long)s * (cap + 1);
  for (int i = kd - 1; i >= 0; --i) {
    double part = 0.0;
    for (int j = i + 1 + lane; j < kd; j += 64) part += Rs[(long)i * cap + j] * ys[j];
    const double tot = wave_sum(part);
    if (lane == 0) {
      const double d = Rs[(long)i * cap + i];
      ys[i] = d != 0.0 ? (gs[i] - tot) / d : 0.0;
    }
    __syncthreads();
  }
  for (int j = lane; j < kd; j += 64) y[(long)s * sy + j] = (T)ys[j];
}

}  // namespace xk

extern "C" {

#define XK_DEFINE_GMRES(SUF, T)                                                                                  \
  int xk_gmres_step_##SUF(const T* c1, long sc1, const T* c2n, long sc2, int k, int cap, double* R, double* cs,   \
                          double* sn, double* g, T* inv_hn, T* est2, int S, void* stream) {                      \
    if (S < 0 || k < 0 || cap <= 0 || k >= cap) return XK_ERR_ARG;                                               \
    if (S == 0) return XK_OK;                                                                                    \
    hipLaunchKernelGGL((xk::gmres_step_kernel<T>), dim3((S + 63) / 64), dim3(64), 0, (hipStream_t)stream, c1,    \
                       sc1, c2n, sc2, k, cap, R, cs, sn, g, inv_hn, est2, S);                                    \
    XK_LAUNCH_CHECK();                                                                                           \
    return XK_OK;                                                                                                \
  }                                                                                                              \
  int xk_gmres_finish_##SUF(T* Q, const T* c2n, long sc2, const T* inv_hn, int S, int N, int k, long ldq,        \
                            long sQ, void* stream) {                                                             \
    if (S < 0 || N < 0 || k < 0) return XK_ERR_ARG;                                                              \
    if (S == 0 || N == 0) return XK_OK;                                                                          \
    constexpr int VN = xk::Vec16<T>::n;                                                                          \
    if ((ldq % VN) || (sQ % VN) || ((uintptr_t)Q & 15) || ldq < (long)((N + VN - 1) / VN) * VN)                   \
      return XK_ERR_UNSUPPORTED;                                                                                 \
    const int ct = (N + 256 * VN - 1) / (256 * VN);                                                              \
    hipLaunchKernelGGL((xk::gmres_finish_kernel<T>), dim3((unsigned)((long)S * ct)), dim3(256), 0,               \
                       (hipStream_t)stream, Q, c2n, sc2, inv_hn, N, k, ldq, sQ, ct);                             \
    XK_LAUNCH_CHECK();                                                                                           \
    return XK_OK;                                                                                                \
  }                                                                                                              \
  int xk_gmres_solve_##SUF(const double* R, const double* g, T* y, long sy, int S, int kd, int cap,              \
                           void* stream) {                                                                       \
    if (S < 0 || kd < 0 || cap <= 0 || kd > cap) return XK_ERR_ARG;                                              \
    if (S == 0 || kd == 0) return XK_OK;                                                                         \
    if (kd > 8192) return XK_ERR_UNSUPPORTED;     /* y of one system lives in LDS (64 KiB) */                    \
    hipLaunchKernelGGL((xk::gmres_solve_kernel<T>), dim3(S), dim3(64), (size_t)kd * sizeof(double),              \
                       (hipStream_t)stream, R, g, y, sy, kd, cap);                                               \
    XK_LAUNCH_CHECK();                                                                                           \
    return XK_OK;                                                                                                \
  }

XK_DEFINE_GMRES(f64, double)
XK_DEFINE_GMRES(f32, float)

}  // extern "C"
