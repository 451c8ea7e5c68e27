// xitorch_amd :: fused BLAS-1 of the quasi-Newton (Broyden) driver.
//
// The reference's driver (xitorch/_impls/optimize/root/rootsolver.py:96-143) and inverse-Jacobian model
// (_jacobian.py:99-119, 172-189) are chains of torch.dot / .norm() / axpy on ONE flat length-L vector (the whole
// batch is one system, quirk Q4), with ~8 host syncs per outer iteration.  Here:
//
//   xk_vec_dots       up to 4 inner products <a_i, b_i> of length-L vectors in ONE streaming pass, finished on
//                     the device in a fixed order (deterministic) into a small double array: the driver reads
//                     {|f|^2, |dx|^2, |x|^2} of a line-search trial with one host sync (wavefront-shuffle
//                     reductions, no atomics)
//   xk_broyden_axpy   out = g0*u0 + g1*u1 + gamma * sum_n (coef[n] * scale[n]) * V[n]
//                     the low-rank apply  G v = alpha v + C^T (D v)  and both halves of the rank-1 update
//                     (v = G^T dx,  c = dx - G dy) written straight into the new rows of the C / D buffers;
//                     `scale` holds the per-term 1/<dy, v> so that d_n = v_n / <dy_n, v_n> never costs a pass
//
// The multi-dot D v (rank x L contraction) is xk_dense_mm with the buffer as the "matrix" (split-contraction path).
// All kernels are HBM-bound streams with 16 B/lane accesses.
#include "xk_common.h"

namespace xk {

constexpr int VD_MAX_PAIRS = 4;
constexpr int VD_MAX_BLOCKS = 1024;

template <typename T>
struct VecPairs {
  const T* a[VD_MAX_PAIRS];
  const T* b[VD_MAX_PAIRS];
};

// stage 1: block partial sums (double accumulation of the wave/block folds; the per-lane sums run in T)
template <typename T, int NP, bool VEC>
__global__ __launch_bounds__(256) void vec_dots_kernel(VecPairs<T> pr, long L, long per_block,
                                                       double* __restrict__ partials) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  __shared__ double sh[4][NP];
  const long lo = (long)blockIdx.x * per_block;
  long hi = lo + per_block;
  hi = hi < L ? hi : L;
  T acc[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) acc[i] = T(0);
  if (VEC) {
    for (long j = lo + (long)threadIdx.x * VN; j + VN <= hi; j += 256L * VN) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const VT av = *reinterpret_cast<const VT*>(pr.a[i] + j);
        const VT bv = (pr.b[i] == pr.a[i]) ? av : *reinterpret_cast<const VT*>(pr.b[i] + j);
#pragma unroll
        for (int v = 0; v < VN; ++v) acc[i] += av[v] * bv[v];
      }
    }
  } else {
    for (long j = lo + threadIdx.x; j < hi; j += 256) {
#pragma unroll
      for (int i = 0; i < NP; ++i) acc[i] += pr.a[i][j] * pr.b[i][j];
    }
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const double w = wave_sum((double)acc[i]);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][i] = w;
  }
  __syncthreads();
  if (threadIdx.x < NP) {
    const int i = threadIdx.x;
    partials[(long)blockIdx.x * VD_MAX_PAIRS + i] = (sh[0][i] + sh[1][i]) + (sh[2][i] + sh[3][i]);
  }
}

// stage 2: fixed-order fold of the block partials (one wave per pair)
__global__ __launch_bounds__(256) void vec_dots_finish(const double* __restrict__ partials, int nblk, int np,
                                                       double* __restrict__ out) {
  const int i = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (i >= np) return;
  double s = 0.0;
  for (int k = lane; k < nblk; k += 64) s += partials[(long)k * VD_MAX_PAIRS + i];
  s = wave_sum(s);
  if (lane == 0) out[i] = s;
}

// out[j] = g0*u0[j] + g1*u1[j] + gamma * sum_n coef[n]*scale[n] * V[n*ldv + j]
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void broyden_axpy_kernel(
    T* __restrict__ out, const T* __restrict__ u0, T g0, const T* __restrict__ u1, T g1,
    const T* __restrict__ V, long ldv, const T* __restrict__ coef, const T* __restrict__ scale, int k,
    T gamma, long L) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  const long j = ((long)blockIdx.x * 256 + threadIdx.x) * (VEC ? VN : 1);
  if (j >= L) return;
  if (VEC) {
    VT acc;
#pragma unroll
    for (int v = 0; v < VN; ++v) acc[v] = T(0);
    int n = 0;
    for (; n + 4 <= k; n += 4) {
      const VT v0 = *reinterpret_cast<const VT*>(V + (long)(n + 0) * ldv + j);
      const VT v1 = *reinterpret_cast<const VT*>(V + (long)(n + 1) * ldv + j);
      const VT v2 = *reinterpret_cast<const VT*>(V + (long)(n + 2) * ldv + j);
      const VT v3 = *reinterpret_cast<const VT*>(V + (long)(n + 3) * ldv + j);
      T c0 = coef[n], c1 = coef[n + 1], c2 = coef[n + 2], c3 = coef[n + 3];      // wave-uniform scalar loads
      if (scale != nullptr) { c0 *= scale[n]; c1 *= scale[n + 1]; c2 *= scale[n + 2]; c3 *= scale[n + 3]; }
#pragma unroll
      for (int v = 0; v < VN; ++v) {
        acc[v] += c0 * v0[v];
        acc[v] += c1 * v1[v];
        acc[v] += c2 * v2[v];
        acc[v] += c3 * v3[v];
      }
    }
    for (; n < k; ++n) {
      const VT v0 = *reinterpret_cast<const VT*>(V + (long)n * ldv + j);
      T c0 = coef[n];
      if (scale != nullptr) c0 *= scale[n];
#pragma unroll
      for (int v = 0; v < VN; ++v) acc[v] += c0 * v0[v];
    }
    VT o;
#pragma unroll
    for (int v = 0; v < VN; ++v) o[v] = gamma * acc[v];
    if (u0 != nullptr) {
      const VT a = *reinterpret_cast<const VT*>(u0 + j);
#pragma unroll
      for (int v = 0; v < VN; ++v) o[v] += g0 * a[v];
    }
    if (u1 != nullptr) {
      const VT a = *reinterpret_cast<const VT*>(u1 + j);
#pragma unroll
      for (int v = 0; v < VN; ++v) o[v] += g1 * a[v];
    }
    *reinterpret_cast<VT*>(out + j) = o;
  } else {
    T acc = T(0);
    for (int n = 0; n < k; ++n) {
      T c0 = coef[n];
      if (scale != nullptr) c0 *= scale[n];
      acc += c0 * V[(long)n * ldv + j];
    }
    T o = gamma * acc;
    if (u0 != nullptr) o += g0 * u0[j];
    if (u1 != nullptr) o += g1 * u1[j];
    out[j] = o;
  }
}

template <typename T>
static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <typename T>
static int vec_dots(const T* const* a, const T* const* b, int np, long L, double* partials, long npart,
                    double* out, hipStream_t st) {
  constexpr int VN = Vec16<T>::n;
  if (np < 1 || np > VD_MAX_PAIRS || L < 0) return XK_ERR_ARG;
  VecPairs<T> pr;
  bool vec = (L % VN) == 0;
  for (int i = 0; i < VD_MAX_PAIRS; ++i) {
    pr.a[i] = a[i < np ? i : 0];
    pr.b[i] = b[i < np ? i : 0];
    if (i < np) vec = vec && aligned16<T>(a[i]) && aligned16<T>(b[i]);
  }
  // blocks of >= 8192 elements, a multiple of the vector width, at most VD_MAX_BLOCKS of them
  long per = (L + VD_MAX_BLOCKS - 1) / VD_MAX_BLOCKS;
  if (per < 8192) per = 8192;
  per = (per + 255) / 256 * 256;
  int nblk = (int)((L + per - 1) / per);
  if (nblk < 1) nblk = 1;
  if ((long)nblk * VD_MAX_PAIRS > npart) return XK_ERR_ARG;
#define XK_VD(NP)                                                                                         \
  case NP:                                                                                                \
    if (vec)                                                                                              \
      hipLaunchKernelGGL((vec_dots_kernel<T, NP, true>), dim3(nblk), dim3(256), 0, st, pr, L, per, partials); \
    else                                                                                                  \
      hipLaunchKernelGGL((vec_dots_kernel<T, NP, false>), dim3(nblk), dim3(256), 0, st, pr, L, per, partials); \
    break;
  switch (np) { XK_VD(1) XK_VD(2) XK_VD(3) XK_VD(4) }
#undef XK_VD
  XK_LAUNCH_CHECK();
  hipLaunchKernelGGL(vec_dots_finish, dim3(1), dim3(256), 0, st, partials, nblk, np, out);
  XK_LAUNCH_CHECK();
  return XK_OK;
}

template <typename T>
static int broyden_axpy(T* out, const T* u0, double g0, const T* u1, double g1, const T* V, long ldv,
                        const T* coef, const T* scale, int k, double gamma, long L, hipStream_t st) {
  constexpr int VN = Vec16<T>::n;
  if (L < 0 || k < 0 || (k > 0 && (V == nullptr || coef == nullptr))) return XK_ERR_ARG;
  if (L == 0) return XK_OK;
  bool vec = (L % VN) == 0 && aligned16<T>(out) && (u0 == nullptr || aligned16<T>(u0)) &&
             (u1 == nullptr || aligned16<T>(u1)) && (k == 0 || (aligned16<T>(V) && ldv % VN == 0));
  if (vec) {
    const long nthr = L / VN;
    hipLaunchKernelGGL((broyden_axpy_kernel<T, true>), dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, out,
                       u0, (T)g0, u1, (T)g1, V, ldv, coef, scale, k, (T)gamma, L);
  } else {
    hipLaunchKernelGGL((broyden_axpy_kernel<T, false>), dim3((unsigned)((L + 255) / 256)), dim3(256), 0, st, out,
                       u0, (T)g0, u1, (T)g1, V, ldv, coef, scale, k, (T)gamma, L);
  }
  XK_LAUNCH_CHECK();
  return XK_OK;
}

}  // namespace xk

extern "C" {

long xk_vec_dots_workspace_elems(void) { return (long)xk::VD_MAX_BLOCKS * xk::VD_MAX_PAIRS; }

#define XK_DEFINE_BROYDEN(SUF, T)                                                                           \
  int xk_vec_dots_##SUF(const T* a0, const T* b0, const T* a1, const T* b1, const T* a2, const T* b2,        \
                        const T* a3, const T* b3, int npairs, long L, double* partials, long npart,          \
                        double* out, void* stream) {                                                         \
    const T* a[4] = {a0, a1, a2, a3};                                                                        \
    const T* b[4] = {b0, b1, b2, b3};                                                                        \
    return xk::vec_dots<T>(a, b, npairs, L, partials, npart, out, (hipStream_t)stream);                      \
  }                                                                                                          \
  int xk_broyden_axpy_##SUF(T* out, const T* u0, double g0, const T* u1, double g1, const T* V, long ldv,    \
                            const T* coef, const T* scale, int k, double gamma, long L, void* stream) {      \
    return xk::broyden_axpy<T>(out, u0, g0, u1, g1, V, ldv, coef, scale, k, gamma, L, (hipStream_t)stream);  \
  }

XK_DEFINE_BROYDEN(f64, double)
XK_DEFINE_BROYDEN(f32, float)

}  // extern "C"
