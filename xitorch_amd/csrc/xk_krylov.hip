// xitorch_amd :: Krylov-loop kernels (K7-K9, K11) and the banded operator apply.
//
// The reference's CG / BiCGStab bodies (xitorch/_impls/linalg/solve.py:143-180, 272-314) are
// chains of torch ops on (B, N, c) tensors with 2 host syncs per iteration.  Here every system
// (batch member x column; S = B*c of them) is one contiguous length-N vector of a padded
// (S, ld) array, all per-system scalars (rho, alpha, omega, beta, <.,.>) stay on the device, and
// each iteration is a handful of fused streaming passes:
//
//   xk_banded_mm      y = A x for a DIA-stored banded operator (LDS halo tile, coalesced band)
//   xk_kry_dots       up to two column-wise dot products (+ optional in-place shift
//                     y -= E_s * z, the `- M X E` term of solve.py:590-595), block partials
//   xk_bicg_p         p = r + beta (p - omega v),  beta from the partials        (solve.py:273-276)
//   xk_bicg_s         s = r - alpha v                                            (solve.py:279,282)
//   xk_bicg_final     x' = x + alpha y + omega z ; r = s - omega t ; |r|^2, <r0,r> (solve.py:286-297)
//   xk_cg_update      x' = x + alpha p ; r -= alpha Ap ; |r|^2                   (solve.py:144-155)
//   xk_cg_p           p = z + beta p                                             (solve.py:171-173)
//   xk_kry_resid      r = b - y ; |r|^2, <r0,r>   (true-residual refresh, solve.py:148-149, 290-291)
//   xk_kry_status     residual norms + global max + number of unconverged systems
//
// Reductions are two-stage and deterministic: producers write one partial per block,
// consumers re-reduce the <= 64 partials of their system in a fixed order.  `_safedenom`
// (solve.py:437-439: exact zeros become eps) is applied wherever the reference applies it.
#include "xk_common.h"

namespace xk {

constexpr int KRY_MAX_PART = 64;   // partial sums per system

template <typename T>
__device__ __forceinline__ T safedenom(T v, T eps) { return v == T(0) ? eps : v; }

// sum of the `nblk` partials of system s (all threads of the block get the value)
template <typename T>
__device__ __forceinline__ T reduce_partials(const T* __restrict__ part, int s, int nblk, T* sh) {
  if (threadIdx.x < 64) {
    T v = (int)threadIdx.x < nblk ? part[(long)s * KRY_MAX_PART + threadIdx.x] : T(0);
    v = wave_sum(v);
    if (threadIdx.x == 0) *sh = v;
  }
  __syncthreads();
  const T r = *sh;
  __syncthreads();
  return r;
}

template <typename T>
__device__ __forceinline__ void block_store_partial(T v, T* __restrict__ part, int s, int blk, T* sh4) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) part[(long)s * KRY_MAX_PART + blk] = (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
  __syncthreads();
}

// each block handles the contiguous element range [lo, hi) of system s
__device__ __forceinline__ void block_range(int N, int nblk, int blk, int vn, int& lo, int& hi) {
  const int chunks = (N + vn - 1) / vn;                 // in 16 B vectors
  const int per = (chunks + nblk - 1) / nblk;
  lo = blk * per * vn;
  hi = lo + per * vn;
  const int npad = chunks * vn;
  if (hi > npad) hi = npad;
  if (lo > npad) lo = npad;
}

#define XK_KRY_PROLOGUE                                    \
  typedef typename Vec16<T>::type VT;                      \
  constexpr int VN = Vec16<T>::n;                          \
  const int s = blockIdx.x / nblk;                         \
  const int blk = blockIdx.x - s * nblk;                   \
  int lo, hi;                                              \
  block_range(N, nblk, blk, VN, lo, hi);                   \
  const long base = (long)s * ld;

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void kry_dots_kernel(
    const T* __restrict__ x1, T* __restrict__ y1, const T* __restrict__ x2, const T* __restrict__ y2,
    const T* __restrict__ shiftz, const T* __restrict__ E, T* __restrict__ P1, T* __restrict__ P2,
    int N, long ld, int nblk, int y1_is_x1) {
  __shared__ T sh4[4];
  XK_KRY_PROLOGUE
  const T e = (E != nullptr) ? E[s] : T(0);
  T a1 = T(0), a2 = T(0);
  for (int j = lo + threadIdx.x * VN; j < hi; j += 256 * VN) {
    VT yv = *reinterpret_cast<const VT*>(y1 + base + j);
    if (E != nullptr) {
      VT zv = *reinterpret_cast<const VT*>(shiftz + base + j);
#pragma unroll
      for (int v = 0; v < VN; ++v) yv[v] -= e * zv[v];
      *reinterpret_cast<VT*>(y1 + base + j) = yv;
    }
    VT xv = y1_is_x1 ? yv : *reinterpret_cast<const VT*>(x1 + base + j);
#pragma unroll
    for (int v = 0; v < VN; ++v) a1 += xv[v] * yv[v];
    if (P2 != nullptr) {
      VT x2v = (x2 == y1) ? yv : *reinterpret_cast<const VT*>(x2 + base + j);
      VT y2v = (y2 == y1) ? yv : *reinterpret_cast<const VT*>(y2 + base + j);
#pragma unroll
      for (int v = 0; v < VN; ++v) a2 += x2v[v] * y2v[v];
    }
  }
  block_store_partial(a1, P1, s, blk, sh4);
  if (P2 != nullptr) block_store_partial(a2, P2, s, blk, sh4);
}

// p = r + beta (p - omega v);  beta = rho_new / safe(rho_old) * (alpha / safe(omega))
// first == 1: first iteration (p = v = 0, rho_old = <r0,r>, alpha = omega = 1) -> p = r
template <typename T>
__global__ __launch_bounds__(256) void bicg_p_kernel(
    const T* __restrict__ r, T* __restrict__ p, const T* __restrict__ v, const T* __restrict__ Prho_new,
    const T* __restrict__ rho_old, const T* __restrict__ alpha, const T* __restrict__ omega,
    T* __restrict__ rho_store, int N, long ld, int nblk, T eps, int first) {
  __shared__ T sh;
  XK_KRY_PROLOGUE
  const T rho_new = reduce_partials(Prho_new, s, nblk, &sh);
  T beta = T(0), om = T(0);
  if (!first) {
    om = safedenom(omega[s], eps);
    beta = rho_new / safedenom(rho_old[s], eps) * (alpha[s] / om);
  } else {
    // reference first pass: rho_k = rho_knew, alpha = omega = 1, p = v = 0  -> beta irrelevant, p = r
    beta = T(0);
  }
  for (int j = lo + threadIdx.x * VN; j < hi; j += 256 * VN) {
    VT rv = *reinterpret_cast<const VT*>(r + base + j);
    VT o = rv;
    if (!first) {
      VT pv = *reinterpret_cast<const VT*>(p + base + j);
      VT vv = *reinterpret_cast<const VT*>(v + base + j);
#pragma unroll
      for (int q = 0; q < VN; ++q) o[q] = rv[q] + beta * (pv[q] - om * vv[q]);
    }
    *reinterpret_cast<VT*>(p + base + j) = o;
  }
  if (blk == 0 && threadIdx.x == 0) rho_store[s] = rho_new;
}

// s = r - alpha v;  alpha = rho / safe(<r0, v>)
template <typename T>
__global__ __launch_bounds__(256) void bicg_s_kernel(
    const T* __restrict__ r, const T* __restrict__ v, T* __restrict__ sv, const T* __restrict__ rho,
    const T* __restrict__ Pr0v, T* __restrict__ alpha_store, int N, long ld, int nblk, T eps) {
  __shared__ T sh;
  XK_KRY_PROLOGUE
  const T r0v = reduce_partials(Pr0v, s, nblk, &sh);
  const T alpha = rho[s] / safedenom(r0v, eps);
  for (int j = lo + threadIdx.x * VN; j < hi; j += 256 * VN) {
    VT rv = *reinterpret_cast<const VT*>(r + base + j);
    VT vv = *reinterpret_cast<const VT*>(v + base + j);
    VT o;
#pragma unroll
    for (int q = 0; q < VN; ++q) o[q] = rv[q] - alpha * vv[q];
    *reinterpret_cast<VT*>(sv + base + j) = o;
  }
  if (blk == 0 && threadIdx.x == 0) alpha_store[s] = alpha;
}

// omega = <Kt,Ks>/safe(<Kt,Kt>);  x' = x + alpha*yd + omega*zd;  r = s - omega t (unless skip_r)
// partials: |r|^2 and <r0, r>
template <typename T>
__global__ __launch_bounds__(256) void bicg_final_kernel(
    const T* __restrict__ x, T* __restrict__ xout, const T* __restrict__ yd, const T* __restrict__ zd,
    const T* __restrict__ sv, const T* __restrict__ t, T* __restrict__ r, const T* __restrict__ r0,
    const T* __restrict__ alpha, const T* __restrict__ Pts, const T* __restrict__ Ptt,
    T* __restrict__ omega_store, T* __restrict__ Prr, T* __restrict__ Prho, int N, long ld, int nblk,
    T eps, int skip_r) {
  __shared__ T sh;
  __shared__ T sh4[4];
  XK_KRY_PROLOGUE
  const T ts = reduce_partials(Pts, s, nblk, &sh);
  const T tt = reduce_partials(Ptt, s, nblk, &sh);
  const T omega = ts / safedenom(tt, eps);
  const T al = alpha[s];
  T arr = T(0), arho = T(0);
  for (int j = lo + threadIdx.x * VN; j < hi; j += 256 * VN) {
    VT xv = *reinterpret_cast<const VT*>(x + base + j);
    VT yv = *reinterpret_cast<const VT*>(yd + base + j);
    VT zv = *reinterpret_cast<const VT*>(zd + base + j);
    VT o;
#pragma unroll
    for (int q = 0; q < VN; ++q) o[q] = (xv[q] + al * yv[q]) + omega * zv[q];
    *reinterpret_cast<VT*>(xout + base + j) = o;
    if (!skip_r) {
      VT s2 = (sv == zd) ? zv : *reinterpret_cast<const VT*>(sv + base + j);
      VT tv = *reinterpret_cast<const VT*>(t + base + j);
      VT r0v = *reinterpret_cast<const VT*>(r0 + base + j);
      VT rn;
#pragma unroll
      for (int q = 0; q < VN; ++q) {
        rn[q] = s2[q] - omega * tv[q];
        arr += rn[q] * rn[q];
        arho += r0v[q] * rn[q];
      }
      *reinterpret_cast<VT*>(r + base + j) = rn;
    }
  }
  if (!skip_r) {
    block_store_partial(arr, Prr, s, blk, sh4);
    block_store_partial(arho, Prho, s, blk, sh4);
  }
  if (blk == 0 && threadIdx.x == 0) omega_store[s] = omega;
}

// r = b - y ; partials |r|^2, <r0, r> (r0 may be null -> only |r|^2; rz_same: second partial = |r|^2)
template <typename T>
__global__ __launch_bounds__(256) void kry_resid_kernel(
    const T* __restrict__ b, const T* __restrict__ y, T* __restrict__ r, const T* __restrict__ r0,
    T* __restrict__ Prr, T* __restrict__ Prho, int N, long ld, int nblk) {
  __shared__ T sh4[4];
  XK_KRY_PROLOGUE
  T arr = T(0), arho = T(0);
  for (int j = lo + threadIdx.x * VN; j < hi; j += 256 * VN) {
    VT bv = *reinterpret_cast<const VT*>(b + base + j);
    VT yv = *reinterpret_cast<const VT*>(y + base + j);
    VT rn;
#pragma unroll
    for (int q = 0; q < VN; ++q) { rn[q] = bv[q] - yv[q]; arr += rn[q] * rn[q]; }
    if (r0 != nullptr) {
      VT r0v = *reinterpret_cast<const VT*>(r0 + base + j);
#pragma unroll
      for (int q = 0; q < VN; ++q) arho += r0v[q] * rn[q];
    }
    *reinterpret_cast<VT*>(r + base + j) = rn;
  }
  block_store_partial(arr, Prr, s, blk, sh4);
  if (Prho != nullptr) block_store_partial(r0 != nullptr ? arho : arr, Prho, s, blk, sh4);
}

// alpha = rz / safe(<p,Ap>);  x' = x + alpha p;  r = r - alpha Ap (unless skip_r);  partial |r|^2
template <typename T>
__global__ __launch_bounds__(256) void cg_update_kernel(
    const T* __restrict__ x, T* __restrict__ xout, const T* __restrict__ p, const T* __restrict__ Ap,
    T* __restrict__ r, const T* __restrict__ Prz, const T* __restrict__ PpAp, T* __restrict__ Prr,
    int N, long ld, int nblk, T eps, int skip_r) {
  __shared__ T sh;
  __shared__ T sh4[4];
  XK_KRY_PROLOGUE
  const T rz = reduce_partials(Prz, s, nblk, &sh);
  const T pap = reduce_partials(PpAp, s, nblk, &sh);
  const T alpha = rz / safedenom(pap, eps);
  T arr = T(0);
  for (int j = lo + threadIdx.x * VN; j < hi; j += 256 * VN) {
    VT xv = *reinterpret_cast<const VT*>(x + base + j);
    VT pv = *reinterpret_cast<const VT*>(p + base + j);
    VT o;
#pragma unroll
    for (int q = 0; q < VN; ++q) o[q] = xv[q] + alpha * pv[q];
    *reinterpret_cast<VT*>(xout + base + j) = o;
    if (!skip_r) {
      VT rv = *reinterpret_cast<const VT*>(r + base + j);
      VT av = *reinterpret_cast<const VT*>(Ap + base + j);
#pragma unroll
      for (int q = 0; q < VN; ++q) { rv[q] -= alpha * av[q]; arr += rv[q] * rv[q]; }
      *reinterpret_cast<VT*>(r + base + j) = rv;
    }
  }
  if (!skip_r) block_store_partial(arr, Prr, s, blk, sh4);
}

// p = z + beta p;  beta = rz_new / safe(rz_old)
template <typename T>
__global__ __launch_bounds__(256) void cg_p_kernel(
    const T* __restrict__ z, T* __restrict__ p, const T* __restrict__ Prz_new, const T* __restrict__ Prz_old,
    int N, long ld, int nblk, T eps) {
  __shared__ T sh;
  XK_KRY_PROLOGUE
  const T rzn = reduce_partials(Prz_new, s, nblk, &sh);
  const T rzo = reduce_partials(Prz_old, s, nblk, &sh);
  const T beta = rzn / safedenom(rzo, eps);
  for (int j = lo + threadIdx.x * VN; j < hi; j += 256 * VN) {
    VT zv = *reinterpret_cast<const VT*>(z + base + j);
    VT pv = *reinterpret_cast<const VT*>(p + base + j);
#pragma unroll
    for (int q = 0; q < VN; ++q) pv[q] = zv[q] + beta * pv[q];
    *reinterpret_cast<VT*>(p + base + j) = pv;
  }
}

// rnorm[s] = sqrt(sum partials); status[0] = max_s rnorm, status[1] = #{s : !(rnorm < stop[s])}
template <typename T>
__global__ __launch_bounds__(256) void kry_status_kernel(
    const T* __restrict__ Prr, const T* __restrict__ stop, T* __restrict__ rnorm, double* __restrict__ status,
    int S, int nblk) {
  __shared__ double smax[4];
  __shared__ double scnt[4];
  double mx = 0.0, cnt = 0.0;
  for (int s = threadIdx.x; s < S; s += 256) {
    T acc = T(0);
    for (int i = 0; i < nblk; ++i) acc += Prr[(long)s * KRY_MAX_PART + i];
    const T rn = sqrt(acc);
    rnorm[s] = rn;
    const double d = (double)rn;
    if (d != d) { mx = INFINITY; cnt += 1.0; }
    else {
      mx = d > mx ? d : mx;
      if (!(rn < stop[s])) cnt += 1.0;
    }
  }
  mx = wave_max(mx);
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) { smax[threadIdx.x >> 6] = mx; scnt[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = smax[0];
    for (int i = 1; i < 4; ++i) m = smax[i] > m ? smax[i] : m;
    status[0] = m;
    status[1] = scnt[0] + scnt[1] + scnt[2] + scnt[3];
  }
}

// ---------------------------------------------------------------------------------------------
// Complex systems (complex64 / complex128): the same fused passes on INTERLEAVED (re, im) storage — what
// torch.view_as_real of a complex panel is.  T is the underlying real type; N, ld count COMPLEX elements; a
// 16 B vector holds CV = VN/2 complex numbers.  Inner products are conj(x).y (the reference's `_dot`,
// xitorch/_impls/linalg/solve.py:441-445), per-system scalars (rho, alpha, omega, E) are complex pairs, the
// partial sums of complex products are stored as pairs ((S, KRY_MAX_PART, 2)); |r|^2 partials stay REAL in the
// (S, KRY_MAX_PART) layout, so xk_kry_status serves both families.  `_safedenom` (solve.py:437-439) replaces an
// exact complex zero by eps + 0i.
// ---------------------------------------------------------------------------------------------
template <typename T> struct cx { T re, im; };
template <typename T> __device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename T> __device__ __forceinline__ cx<T> cdiv(cx<T> a, cx<T> b) {
  const T d = b.re * b.re + b.im * b.im;
  return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
template <typename T> __device__ __forceinline__ cx<T> csafe(cx<T> v, T eps) {
  return (v.re == T(0) && v.im == T(0)) ? cx<T>{eps, T(0)} : v;
}
template <typename T> __device__ __forceinline__ cx<T> cload(const T* p, int s) { return {p[2 * (long)s], p[2 * (long)s + 1]}; }

template <typename T>
__device__ __forceinline__ cx<T> reduce_partials_c(const T* __restrict__ part, int s, int nblk, T* sh2) {
  if (threadIdx.x < 64) {
    T vr = (int)threadIdx.x < nblk ? part[((long)s * KRY_MAX_PART + threadIdx.x) * 2] : T(0);
    T vi = (int)threadIdx.x < nblk ? part[((long)s * KRY_MAX_PART + threadIdx.x) * 2 + 1] : T(0);
    vr = wave_sum(vr);
    vi = wave_sum(vi);
    if (threadIdx.x == 0) { sh2[0] = vr; sh2[1] = vi; }
  }
  __syncthreads();
  const cx<T> r = {sh2[0], sh2[1]};
  __syncthreads();
  return r;
}

template <typename T>
__device__ __forceinline__ void block_store_partial_c(cx<T> v, T* __restrict__ part, int s, int blk, T* sh8) {
  const T vr = wave_sum(v.re), vi = wave_sum(v.im);
  if ((threadIdx.x & 63) == 0) { sh8[(threadIdx.x >> 6) * 2] = vr; sh8[(threadIdx.x >> 6) * 2 + 1] = vi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((long)s * KRY_MAX_PART + blk) * 2] = (sh8[0] + sh8[2]) + (sh8[4] + sh8[6]);
    part[((long)s * KRY_MAX_PART + blk) * 2 + 1] = (sh8[1] + sh8[3]) + (sh8[5] + sh8[7]);
  }
  __syncthreads();
}

// complex prologue: [lo, hi) in COMPLEX elements, `base` in real elements
#define XK_KRYC_PROLOGUE                                   \
  typedef typename Vec16<T>::type VT;                      \
  constexpr int VN = Vec16<T>::n;                          \
  constexpr int CV = VN / 2;                               \
  const int s = blockIdx.x / nblk;                         \
  const int blk = blockIdx.x - s * nblk;                   \
  int lo, hi;                                              \
  block_range(N, nblk, blk, CV, lo, hi);                   \
  const long base = (long)s * ld * 2;
#define XK_CLOOP for (int j = lo + threadIdx.x * CV; j < hi; j += 256 * CV)
#define XK_LDV(ptr) (*reinterpret_cast<const VT*>((ptr) + base + 2 * (long)j))
#define XK_STV(ptr, val) (*reinterpret_cast<VT*>((ptr) + base + 2 * (long)j) = (val))

template <typename T>
__global__ __launch_bounds__(256) void kry_dots_c_kernel(
    const T* __restrict__ x1, T* __restrict__ y1, const T* __restrict__ x2, const T* __restrict__ y2,
    const T* __restrict__ shiftz, const T* __restrict__ E, T* __restrict__ P1, T* __restrict__ P2,
    int N, long ld, int nblk, int y1_is_x1, int conj1) {
  __shared__ T sh8[8];
  XK_KRYC_PROLOGUE
  const cx<T> e = (E != nullptr) ? cload(E, s) : cx<T>{T(0), T(0)};
  cx<T> a1 = {T(0), T(0)}, a2 = {T(0), T(0)};
  XK_CLOOP {
    VT yv = XK_LDV(y1);
    if (E != nullptr) {
      const VT zv = XK_LDV(shiftz);
#pragma unroll
      for (int q = 0; q < CV; ++q) {
        const cx<T> ez = cmul(e, cx<T>{zv[2 * q], zv[2 * q + 1]});
        yv[2 * q] -= ez.re;
        yv[2 * q + 1] -= ez.im;
      }
      XK_STV(y1, yv);
    }
    const VT xv = y1_is_x1 ? yv : XK_LDV(x1);
#pragma unroll
    for (int q = 0; q < CV; ++q) {          // conj(x) * y
      a1.re += xv[2 * q] * yv[2 * q] + xv[2 * q + 1] * yv[2 * q + 1];
      a1.im += xv[2 * q] * yv[2 * q + 1] - xv[2 * q + 1] * yv[2 * q];
    }
    if (P2 != nullptr) {
      const VT x2v = (x2 == y1) ? yv : XK_LDV(x2);
      const VT y2v = (y2 == y1) ? yv : XK_LDV(y2);
#pragma unroll
      for (int q = 0; q < CV; ++q) {
        a2.re += x2v[2 * q] * y2v[2 * q] + x2v[2 * q + 1] * y2v[2 * q + 1];
        a2.im += x2v[2 * q] * y2v[2 * q + 1] - x2v[2 * q + 1] * y2v[2 * q];
      }
    }
  }
  if (conj1) a1.im = -a1.im;               // P1 <- conj(y1).x1 instead of conj(x1).y1
  block_store_partial_c(a1, P1, s, blk, sh8);
  if (P2 != nullptr) block_store_partial_c(a2, P2, s, blk, sh8);
}

template <typename T>
__global__ __launch_bounds__(256) void bicg_p_c_kernel(
    const T* __restrict__ r, T* __restrict__ p, const T* __restrict__ v, const T* __restrict__ Prho_new,
    const T* __restrict__ rho_old, const T* __restrict__ alpha, const T* __restrict__ omega,
    T* __restrict__ rho_store, int N, long ld, int nblk, T eps, int first) {
  __shared__ T sh2[2];
  XK_KRYC_PROLOGUE
  const cx<T> rho_new = reduce_partials_c(Prho_new, s, nblk, sh2);
  cx<T> beta = {T(0), T(0)}, om = {T(0), T(0)};
  if (!first) {
    om = csafe(cload(omega, s), eps);
    beta = cmul(cdiv(rho_new, csafe(cload(rho_old, s), eps)), cdiv(cload(alpha, s), om));
  }
  XK_CLOOP {
    const VT rv = XK_LDV(r);
    VT o = rv;
    if (!first) {
      const VT pv = XK_LDV(p);
      const VT vv = XK_LDV(v);
#pragma unroll
      for (int q = 0; q < CV; ++q) {
        const cx<T> ov = cmul(om, cx<T>{vv[2 * q], vv[2 * q + 1]});
        const cx<T> t = cmul(beta, cx<T>{pv[2 * q] - ov.re, pv[2 * q + 1] - ov.im});
        o[2 * q] = rv[2 * q] + t.re;
        o[2 * q + 1] = rv[2 * q + 1] + t.im;
      }
    }
    XK_STV(p, o);
  }
  if (blk == 0 && threadIdx.x == 0) { rho_store[2 * (long)s] = rho_new.re; rho_store[2 * (long)s + 1] = rho_new.im; }
}

template <typename T>
__global__ __launch_bounds__(256) void bicg_s_c_kernel(
    const T* __restrict__ r, const T* __restrict__ v, T* __restrict__ sv, const T* __restrict__ rho,
    const T* __restrict__ Pr0v, T* __restrict__ alpha_store, int N, long ld, int nblk, T eps) {
  __shared__ T sh2[2];
  XK_KRYC_PROLOGUE
  const cx<T> r0v = reduce_partials_c(Pr0v, s, nblk, sh2);
  const cx<T> al = cdiv(cload(rho, s), csafe(r0v, eps));
  XK_CLOOP {
    const VT rv = XK_LDV(r);
    const VT vv = XK_LDV(v);
    VT o;
#pragma unroll
    for (int q = 0; q < CV; ++q) {
      const cx<T> av = cmul(al, cx<T>{vv[2 * q], vv[2 * q + 1]});
      o[2 * q] = rv[2 * q] - av.re;
      o[2 * q + 1] = rv[2 * q + 1] - av.im;
    }
    XK_STV(sv, o);
  }
  if (blk == 0 && threadIdx.x == 0) { alpha_store[2 * (long)s] = al.re; alpha_store[2 * (long)s + 1] = al.im; }
}

template <typename T>
__global__ __launch_bounds__(256) void bicg_final_c_kernel(
    const T* __restrict__ x, T* __restrict__ xout, const T* __restrict__ yd, const T* __restrict__ zd,
    const T* __restrict__ sv, const T* __restrict__ t, T* __restrict__ r, const T* __restrict__ r0,
    const T* __restrict__ alpha, const T* __restrict__ Pts, const T* __restrict__ Ptt,
    T* __restrict__ omega_store, T* __restrict__ Prr, T* __restrict__ Prho, int N, long ld, int nblk,
    T eps, int skip_r) {
  __shared__ T sh2[2];
  __shared__ T sh8[8];
  __shared__ T sh4[4];
  XK_KRYC_PROLOGUE
  const cx<T> ts = reduce_partials_c(Pts, s, nblk, sh2);
  const cx<T> tt = reduce_partials_c(Ptt, s, nblk, sh2);
  const cx<T> omega = cdiv(ts, csafe(tt, eps));
  const cx<T> al = cload(alpha, s);
  T arr = T(0);
  cx<T> arho = {T(0), T(0)};
  XK_CLOOP {
    const VT xv = XK_LDV(x);
    const VT yv = XK_LDV(yd);
    const VT zv = XK_LDV(zd);
    VT o;
#pragma unroll
    for (int q = 0; q < CV; ++q) {
      const cx<T> ay = cmul(al, cx<T>{yv[2 * q], yv[2 * q + 1]});
      const cx<T> oz = cmul(omega, cx<T>{zv[2 * q], zv[2 * q + 1]});
      o[2 * q] = (xv[2 * q] + ay.re) + oz.re;
      o[2 * q + 1] = (xv[2 * q + 1] + ay.im) + oz.im;
    }
    XK_STV(xout, o);
    if (!skip_r) {
      const VT s2 = (sv == zd) ? zv : XK_LDV(sv);
      const VT tv = XK_LDV(t);
      const VT r0v = XK_LDV(r0);
      VT rn;
#pragma unroll
      for (int q = 0; q < CV; ++q) {
        const cx<T> ot = cmul(omega, cx<T>{tv[2 * q], tv[2 * q + 1]});
        rn[2 * q] = s2[2 * q] - ot.re;
        rn[2 * q + 1] = s2[2 * q + 1] - ot.im;
        arr += rn[2 * q] * rn[2 * q] + rn[2 * q + 1] * rn[2 * q + 1];
        arho.re += r0v[2 * q] * rn[2 * q] + r0v[2 * q + 1] * rn[2 * q + 1];
        arho.im += r0v[2 * q] * rn[2 * q + 1] - r0v[2 * q + 1] * rn[2 * q];
      }
      XK_STV(r, rn);
    }
  }
  if (!skip_r) {
    block_store_partial(arr, Prr, s, blk, sh4);
    block_store_partial_c(arho, Prho, s, blk, sh8);
  }
  if (blk == 0 && threadIdx.x == 0) { omega_store[2 * (long)s] = omega.re; omega_store[2 * (long)s + 1] = omega.im; }
}

template <typename T>
__global__ __launch_bounds__(256) void kry_resid_c_kernel(
    const T* __restrict__ b, const T* __restrict__ y, T* __restrict__ r, const T* __restrict__ r0,
    T* __restrict__ Prr, T* __restrict__ Prho, int N, long ld, int nblk) {
  __shared__ T sh8[8];
  __shared__ T sh4[4];
  XK_KRYC_PROLOGUE
  T arr = T(0);
  cx<T> arho = {T(0), T(0)};
  XK_CLOOP {
    const VT bv = XK_LDV(b);
    const VT yv = XK_LDV(y);
    VT rn;
#pragma unroll
    for (int q = 0; q < VN; ++q) { rn[q] = bv[q] - yv[q]; arr += rn[q] * rn[q]; }
    if (r0 != nullptr) {
      const VT r0v = XK_LDV(r0);
#pragma unroll
      for (int q = 0; q < CV; ++q) {
        arho.re += r0v[2 * q] * rn[2 * q] + r0v[2 * q + 1] * rn[2 * q + 1];
        arho.im += r0v[2 * q] * rn[2 * q + 1] - r0v[2 * q + 1] * rn[2 * q];
      }
    }
    XK_STV(r, rn);
  }
  block_store_partial(arr, Prr, s, blk, sh4);
  if (Prho != nullptr) block_store_partial_c(r0 != nullptr ? arho : cx<T>{arr, T(0)}, Prho, s, blk, sh8);
}

template <typename T>
__global__ __launch_bounds__(256) void cg_update_c_kernel(
    const T* __restrict__ x, T* __restrict__ xout, const T* __restrict__ p, const T* __restrict__ Ap,
    T* __restrict__ r, const T* __restrict__ Prz, const T* __restrict__ PpAp, T* __restrict__ Prr,
    int N, long ld, int nblk, T eps, int skip_r) {
  __shared__ T sh2[2];
  __shared__ T sh4[4];
  XK_KRYC_PROLOGUE
  const cx<T> rz = reduce_partials_c(Prz, s, nblk, sh2);
  const cx<T> pap = reduce_partials_c(PpAp, s, nblk, sh2);
  const cx<T> al = cdiv(rz, csafe(pap, eps));
  T arr = T(0);
  XK_CLOOP {
    const VT xv = XK_LDV(x);
    const VT pv = XK_LDV(p);
    VT o;
#pragma unroll
    for (int q = 0; q < CV; ++q) {
      const cx<T> ap = cmul(al, cx<T>{pv[2 * q], pv[2 * q + 1]});
      o[2 * q] = xv[2 * q] + ap.re;
      o[2 * q + 1] = xv[2 * q + 1] + ap.im;
    }
    XK_STV(xout, o);
    if (!skip_r) {
      VT rv = XK_LDV(r);
      const VT av = XK_LDV(Ap);
#pragma unroll
      for (int q = 0; q < CV; ++q) {
        const cx<T> aa = cmul(al, cx<T>{av[2 * q], av[2 * q + 1]});
        rv[2 * q] -= aa.re;
        rv[2 * q + 1] -= aa.im;
        arr += rv[2 * q] * rv[2 * q] + rv[2 * q + 1] * rv[2 * q + 1];
      }
      XK_STV(r, rv);
    }
  }
  if (!skip_r) block_store_partial(arr, Prr, s, blk, sh4);
}

template <typename T>
__global__ __launch_bounds__(256) void cg_p_c_kernel(
    const T* __restrict__ z, T* __restrict__ p, const T* __restrict__ Prz_new, const T* __restrict__ Prz_old,
    int N, long ld, int nblk, T eps) {
  __shared__ T sh2[2];
  XK_KRYC_PROLOGUE
  const cx<T> rzn = reduce_partials_c(Prz_new, s, nblk, sh2);
  const cx<T> rzo = reduce_partials_c(Prz_old, s, nblk, sh2);
  const cx<T> beta = cdiv(rzn, csafe(rzo, eps));
  XK_CLOOP {
    const VT zv = XK_LDV(z);
    VT pv = XK_LDV(p);
#pragma unroll
    for (int q = 0; q < CV; ++q) {
      const cx<T> bp = cmul(beta, cx<T>{pv[2 * q], pv[2 * q + 1]});
      pv[2 * q] = zv[2 * q] + bp.re;
      pv[2 * q + 1] = zv[2 * q + 1] + bp.im;
    }
    XK_STV(p, pv);
  }
}

// ---------------------------------------------------------------------------------------------
// Banded operator, DIA storage band[b, d, i] = A_b[i, i + d - hb]   (nd = 2*hb+1 diagonals)
//   trans=0: y[b,c,i] = sum_d band[b,d,i] * x[b,c,i+d-hb]
//   trans=1: y[b,c,j] = sum_d band[b,d,j-(d-hb)] * x[b,c,j-(d-hb)]
// One block = ROWS consecutive rows of one batch member for all C <= 8 columns: the x tile (rows
// + halo) is staged once in LDS, the band streams through registers with coalesced loads
// (consecutive lanes -> consecutive i of one diagonal).  Entries whose column falls outside the
// matrix are masked, so the storage there may hold anything.
// ---------------------------------------------------------------------------------------------
template <typename T, int C, bool VEC>
__global__ __launch_bounds__(256) void banded_mm_kernel(
    const T* __restrict__ band, const T* __restrict__ X, T* __restrict__ Y, int N, int hb, long sBand,
    long ldx, long sX, long ldy, long sY, int row_tiles, int trans) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;            // consecutive rows per thread (one 16 B band load)
  constexpr int ROWS = 256 * VN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* xs = reinterpret_cast<T*>(smem);       // C x (ROWS + 2*hb)
  const int b = blockIdx.x / row_tiles;
  const int rt = blockIdx.x - b * row_tiles;
  const int i0 = rt * ROWS;
  const int tw = ROWS + 2 * hb;
  const T* Xb = X + (long)b * sX;
  for (int idx = threadIdx.x; idx < C * tw; idx += 256) {
    const int c = idx / tw, l = idx - c * tw;
    const int g = i0 - hb + l;
    xs[idx] = (g >= 0 && g < N) ? Xb[(long)c * ldx + g] : T(0);
  }
  __syncthreads();
  const int li0 = threadIdx.x * VN;
  const int r0 = i0 + li0;                   // first row owned by this thread
  if (r0 >= N) return;
  const T* Bb = band + (long)b * sBand;
  const int nd = 2 * hb + 1;
  T acc[C][VN];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int q = 0; q < VN; ++q) acc[c][q] = T(0);
  if (!trans) {
#pragma unroll 4
    for (int d = 0; d < nd; ++d) {
      const int off = d - hb;
      T bv[VN];
      if (VEC) {
        const VT v = __builtin_nontemporal_load(reinterpret_cast<const VT*>(Bb + (long)d * N + r0));
#pragma unroll
        for (int q = 0; q < VN; ++q) bv[q] = v[q];
      } else {
#pragma unroll
        for (int q = 0; q < VN; ++q) bv[q] = (r0 + q < N) ? Bb[(long)d * N + r0 + q] : T(0);
      }
#pragma unroll
      for (int q = 0; q < VN; ++q) {
        const int col = r0 + q + off;
        if (col < 0 || col >= N) bv[q] = T(0);     // outside the matrix: ignored, whatever is stored
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c][q] += bv[q] * xs[c * tw + li0 + q + d];
      }
    }
  } else {
#pragma unroll 4
    for (int d = 0; d < nd; ++d) {
      const int off = d - hb;
#pragma unroll
      for (int q = 0; q < VN; ++q) {
        const int j = r0 + q;
        const int i = j - off;                 // source row of A^T's entry
        T bv = T(0);
        if (j < N && i >= 0 && i < N) bv = Bb[(long)d * N + i];
        // x[i] sits at tile position i - (i0 - hb) = li0 + q + 2*hb - d
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c][q] += bv * xs[c * tw + li0 + q + 2 * hb - d];
      }
    }
  }
  T* Yb = Y + (long)b * sY;
#pragma unroll
  for (int q = 0; q < VN; ++q) {
    if (r0 + q < N) {
#pragma unroll
      for (int c = 0; c < C; ++c) Yb[(long)c * ldy + r0 + q] = acc[c][q];
    }
  }
}

template <typename T, int C>
static int banded_launch(const T* band, const T* Xc, T* Yc, int B, int N, int hb, long sBand, long ldx,
                         long sX, long ldy, long sY, int trans, hipStream_t st) {
  constexpr int VN = Vec16<T>::n;
  constexpr int ROWS = 256 * VN;
  const int row_tiles = (N + ROWS - 1) / ROWS;
  const size_t lds = (size_t)C * (ROWS + 2 * hb) * sizeof(T);
  if (lds > 160 * 1024) return XK_ERR_UNSUPPORTED;
  const bool vec = (N % VN == 0) && (sBand % VN == 0) && (((uintptr_t)band & 15) == 0);
  const dim3 grid((unsigned)((long)B * row_tiles));
  if (vec) {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)banded_mm_kernel<T, C, true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((banded_mm_kernel<T, C, true>), grid, dim3(256), lds, st, band, Xc, Yc, N, hb, sBand,
                       ldx, sX, ldy, sY, row_tiles, trans);
  } else {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)banded_mm_kernel<T, C, false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((banded_mm_kernel<T, C, false>), grid, dim3(256), lds, st, band, Xc, Yc, N, hb, sBand,
                       ldx, sX, ldy, sY, row_tiles, trans);
  }
  XK_LAUNCH_CHECK();
  return XK_OK;
}

template <typename T>
static int banded_mm(const T* band, const T* X, T* Y, int B, int N, int hb, int C, long sBand, long ldx,
                     long sX, long ldy, long sY, int trans, hipStream_t st) {
  int c0 = 0;
  while (c0 < C) {
    const int pc = (C - c0) >= 8 ? 8 : (C - c0);
    const T* Xc = X + (long)c0 * ldx;
    T* Yc = Y + (long)c0 * ldy;
    int rc = XK_ERR_UNSUPPORTED;
    switch (pc) {
#define XK_CASE(CC) \
  case CC: rc = banded_launch<T, CC>(band, Xc, Yc, B, N, hb, sBand, ldx, sX, ldy, sY, trans, st); break;
      XK_CASE(1) XK_CASE(2) XK_CASE(3) XK_CASE(4) XK_CASE(5) XK_CASE(6) XK_CASE(7) XK_CASE(8)
#undef XK_CASE
    }
    if (rc != XK_OK) return rc;
    c0 += pc;
  }
  return XK_OK;
}

}  // namespace xk

extern "C" {

int xk_kry_max_partials(void) { return xk::KRY_MAX_PART; }

#define XK_GRID(S, nblk) dim3((unsigned)((long)(S) * (nblk))), dim3(256), 0, (hipStream_t)stream
#define XK_CHECK_KRY                                                     \
  if (S < 0 || N < 0 || nblk < 1 || nblk > xk::KRY_MAX_PART) return XK_ERR_ARG; \
  if (S == 0 || N == 0) return XK_OK;

#define XK_DEFINE_KRY(SUF, T)                                                                             \
  int xk_banded_mm_##SUF(const T* band, const T* X, T* Y, int B, int N, int hb, int C, long sBand,        \
                         long ldx, long sX, long ldy, long sY, int trans, void* stream) {                 \
    if (B < 0 || N < 0 || hb < 0 || C < 0) return XK_ERR_ARG;                                             \
    if (B == 0 || N == 0 || C == 0) return XK_OK;                                                         \
    return xk::banded_mm<T>(band, X, Y, B, N, hb, C, sBand, ldx, sX, ldy, sY, trans, (hipStream_t)stream); \
  }                                                                                                       \
  int xk_kry_dots_##SUF(const T* x1, T* y1, const T* x2, const T* y2, const T* shiftz, const T* E, T* P1, \
                        T* P2, int S, int N, long ld, int nblk, void* stream) {                           \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::kry_dots_kernel<T>), XK_GRID(S, nblk), x1, y1, x2, y2, shiftz, E, P1, P2, N,   \
                       ld, nblk, (x1 == y1) ? 1 : 0);                                                     \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_bicg_p_##SUF(const T* r, T* p, const T* v, const T* Prho_new, const T* rho_old, const T* alpha,  \
                      const T* omega, T* rho_store, int S, int N, long ld, int nblk, double eps,          \
                      int first, void* stream) {                                                          \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::bicg_p_kernel<T>), XK_GRID(S, nblk), r, p, v, Prho_new, rho_old, alpha,        \
                       omega, rho_store, N, ld, nblk, (T)eps, first);                                     \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_bicg_s_##SUF(const T* r, const T* v, T* sv, const T* rho, const T* Pr0v, T* alpha_store, int S,   \
                      int N, long ld, int nblk, double eps, void* stream) {                               \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::bicg_s_kernel<T>), XK_GRID(S, nblk), r, v, sv, rho, Pr0v, alpha_store, N, ld,  \
                       nblk, (T)eps);                                                                     \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_bicg_final_##SUF(const T* x, T* xout, const T* yd, const T* zd, const T* sv, const T* t, T* r,    \
                          const T* r0, const T* alpha, const T* Pts, const T* Ptt, T* omega_store,        \
                          T* Prr, T* Prho, int S, int N, long ld, int nblk, double eps, int skip_r,       \
                          void* stream) {                                                                 \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::bicg_final_kernel<T>), XK_GRID(S, nblk), x, xout, yd, zd, sv, t, r, r0,        \
                       alpha, Pts, Ptt, omega_store, Prr, Prho, N, ld, nblk, (T)eps, skip_r);             \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_kry_resid_##SUF(const T* b, const T* y, T* r, const T* r0, T* Prr, T* Prho, int S, int N,         \
                         long ld, int nblk, void* stream) {                                               \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::kry_resid_kernel<T>), XK_GRID(S, nblk), b, y, r, r0, Prr, Prho, N, ld, nblk);  \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_cg_update_##SUF(const T* x, T* xout, const T* p, const T* Ap, T* r, const T* Prz, const T* PpAp,  \
                         T* Prr, int S, int N, long ld, int nblk, double eps, int skip_r, void* stream) { \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::cg_update_kernel<T>), XK_GRID(S, nblk), x, xout, p, Ap, r, Prz, PpAp, Prr, N,  \
                       ld, nblk, (T)eps, skip_r);                                                         \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_cg_p_##SUF(const T* z, T* p, const T* Prz_new, const T* Prz_old, int S, int N, long ld,           \
                    int nblk, double eps, void* stream) {                                                 \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::cg_p_kernel<T>), XK_GRID(S, nblk), z, p, Prz_new, Prz_old, N, ld, nblk,        \
                       (T)eps);                                                                           \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_kry_status_##SUF(const T* Prr, const T* stop, T* rnorm, double* status, int S, int nblk,          \
                          void* stream) {                                                                 \
    if (S < 0 || nblk < 1 || nblk > xk::KRY_MAX_PART) return XK_ERR_ARG;                                   \
    hipLaunchKernelGGL((xk::kry_status_kernel<T>), dim3(1), dim3(256), 0, (hipStream_t)stream, Prr, stop,  \
                       rnorm, status, S, nblk);                                                           \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }

XK_DEFINE_KRY(f64, double)
XK_DEFINE_KRY(f32, float)

// complex families: T* arguments point at interleaved (re, im) storage; N, ld in complex elements; eps real
#define XK_DEFINE_KRYC(SUF, T)                                                                            \
  int xk_kry_dots_##SUF(const T* x1, T* y1, const T* x2, const T* y2, const T* shiftz, const T* E, T* P1, \
                        T* P2, int S, int N, long ld, int nblk, int conj1, void* stream) {                \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::kry_dots_c_kernel<T>), XK_GRID(S, nblk), x1, y1, x2, y2, shiftz, E, P1, P2, N, \
                       ld, nblk, (x1 == y1) ? 1 : 0, conj1);                                              \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_bicg_p_##SUF(const T* r, T* p, const T* v, const T* Prho_new, const T* rho_old, const T* alpha,  \
                      const T* omega, T* rho_store, int S, int N, long ld, int nblk, double eps,          \
                      int first, void* stream) {                                                          \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::bicg_p_c_kernel<T>), XK_GRID(S, nblk), r, p, v, Prho_new, rho_old, alpha,      \
                       omega, rho_store, N, ld, nblk, (T)eps, first);                                     \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_bicg_s_##SUF(const T* r, const T* v, T* sv, const T* rho, const T* Pr0v, T* alpha_store, int S,   \
                      int N, long ld, int nblk, double eps, void* stream) {                               \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::bicg_s_c_kernel<T>), XK_GRID(S, nblk), r, v, sv, rho, Pr0v, alpha_store, N,    \
                       ld, nblk, (T)eps);                                                                 \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_bicg_final_##SUF(const T* x, T* xout, const T* yd, const T* zd, const T* sv, const T* t, T* r,    \
                          const T* r0, const T* alpha, const T* Pts, const T* Ptt, T* omega_store,        \
                          T* Prr, T* Prho, int S, int N, long ld, int nblk, double eps, int skip_r,       \
                          void* stream) {                                                                 \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::bicg_final_c_kernel<T>), XK_GRID(S, nblk), x, xout, yd, zd, sv, t, r, r0,      \
                       alpha, Pts, Ptt, omega_store, Prr, Prho, N, ld, nblk, (T)eps, skip_r);             \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_kry_resid_##SUF(const T* b, const T* y, T* r, const T* r0, T* Prr, T* Prho, int S, int N,         \
                         long ld, int nblk, void* stream) {                                               \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::kry_resid_c_kernel<T>), XK_GRID(S, nblk), b, y, r, r0, Prr, Prho, N, ld,       \
                       nblk);                                                                             \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_cg_update_##SUF(const T* x, T* xout, const T* p, const T* Ap, T* r, const T* Prz, const T* PpAp,  \
                         T* Prr, int S, int N, long ld, int nblk, double eps, int skip_r, void* stream) { \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::cg_update_c_kernel<T>), XK_GRID(S, nblk), x, xout, p, Ap, r, Prz, PpAp, Prr,   \
                       N, ld, nblk, (T)eps, skip_r);                                                      \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }                                                                                                       \
  int xk_cg_p_##SUF(const T* z, T* p, const T* Prz_new, const T* Prz_old, int S, int N, long ld,           \
                    int nblk, double eps, void* stream) {                                                 \
    XK_CHECK_KRY                                                                                          \
    hipLaunchKernelGGL((xk::cg_p_c_kernel<T>), XK_GRID(S, nblk), z, p, Prz_new, Prz_old, N, ld, nblk,      \
                       (T)eps);                                                                           \
    XK_LAUNCH_CHECK();                                                                                    \
    return XK_OK;                                                                                         \
  }

XK_DEFINE_KRYC(c128, double)
XK_DEFINE_KRYC(c64, float)

}  // extern "C"
