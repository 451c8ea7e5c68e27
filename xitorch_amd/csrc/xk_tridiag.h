// xitorch_amd :: bisection on a symmetric tridiagonal matrix (d, e) held in LDS, shared by the final kernels of the small
// eigensolvers (xk_eigh_tri.hip: K3t, xk_eigh_big.hip: K3g / K3p).  Replaces the dstebz stage of torch.linalg.eigh on the
// projected matrix (xitorch/_impls/linalg/symeig.py:174-175) for the p wanted eigenvalues.
//
// One wave per wanted eigenvalue, 64 shifts per round (one per lane).  What a round costs is the DEPENDENT chain of the
// Sturm count, n steps long:
//   ratio form (dstebz):   q_i = d_i - s - e_{i-1}^2 / q_{i-1}        reciprocal + 2 Newton steps + 2 fma  ~ 150 cycles / step
//   product form:          p_i = (d_i - s) p_{i-1} - e_{i-1}^2 p_{i-2}    3 instructions, one fma on the chain
// (r06, profiles/r06_k3_final_phases.json: the bisection was a third of the final kernel.)  The product form needs
// scaling (every 8 steps both live values are divided by the power of two of the larger) and is not provably monotone in
// the shift (and an exact zero or a decoupled matrix gives it wrong counts), so it only NARROWS the bracket, to 4 eps |T|;
// the bracket is then widened by 4 eps |T| on both sides and the last round(s) run the ratio form with the end points of
// the bracket among the shifts: if they do not confirm the bracket (count(lo) < target <= count(hi)), the search
// restarts from the Gershgorin interval in the ratio form.
#pragma once
#include "xk_common.h"

namespace xk {

__device__ __forceinline__ double tri_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float tri_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  r = fmaf(fmaf(-x, r, 1.0f), r, r);
  return r;
}
__device__ __forceinline__ double tri_scale_down(double x, int e) { return ldexp(x, -e); }
__device__ __forceinline__ float tri_scale_down(float x, int e) { return ldexpf(x, -e); }
__device__ __forceinline__ int tri_exponent(double x) { return __builtin_amdgcn_frexp_exp(x); }
__device__ __forceinline__ int tri_exponent(float x) { return __builtin_amdgcn_frexp_expf(x); }

// Is the scale of the matrix (|T| = its Gershgorin bound) inside the range in which neither the Householder reduction
// (alpha^2 + sigma) nor the Sturm counts (e^2) nor the pivot floor leave the floating-point range?  Outside it the solver
// flags its result and the caller repeats the call on the library, which rescales (dsyevx's rmin / rmax): fp64
// 2^-450 .. 2^450 (1e-135 .. 1e135), fp32 2^-50 .. 2^50 (1e-15 .. 1e15).  A zero matrix is in range.
__device__ __forceinline__ bool tri_scale_in_range(double tnorm) {
  if (tnorm == 0.0) return true;
  const int ex = __builtin_amdgcn_frexp_exp(tnorm);
  return ex >= -450 && ex <= 450;
}
__device__ __forceinline__ bool tri_scale_in_range(float tnorm) {
  if (tnorm == 0.0f) return true;
  const int ex = __builtin_amdgcn_frexp_expf(tnorm);
  return ex >= -50 && ex <= 50;
}

// ratio form; the operands of eight steps are requested from LDS before their chain, the next eight while it runs
template <typename T>
__device__ __forceinline__ int tri_sturm_ratio(const T* __restrict__ dd, const T* __restrict__ e2, int n, T sigma,
                                               T pivmin) {
  constexpr int U = 8;
  T q = dd[0] - sigma;
  if (fabs(q) < pivmin) q = -pivmin;
  int cnt = q < T(0) ? 1 : 0;
  int i = 1;
  T dn[U], en[U];
  if (i + U <= n) {
#pragma unroll
    for (int u = 0; u < U; ++u) { dn[u] = dd[i + u]; en[u] = e2[i + u - 1]; }
  }
  for (; i + U <= n; i += U) {
    T dc[U], ec[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { dc[u] = dn[u] - sigma; ec[u] = en[u]; }
    if (i + 2 * U <= n) {
#pragma unroll
      for (int u = 0; u < U; ++u) { dn[u] = dd[i + U + u]; en[u] = e2[i + U + u - 1]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      q = dc[u] - ec[u] * tri_rcp(q);
      if (fabs(q) < pivmin) q = -pivmin;
      cnt += q < T(0) ? 1 : 0;
    }
  }
  for (; i < n; ++i) {
    q = dd[i] - sigma - e2[i - 1] * tri_rcp(q);
    if (fabs(q) < pivmin) q = -pivmin;
    cnt += q < T(0) ? 1 : 0;
  }
  return cnt;
}

__device__ __forceinline__ int tri_signword(double x) { return __double2hiint(x); }
__device__ __forceinline__ int tri_signword(float x) { return __float_as_int(x); }

// product form on the matrix scaled by sc = 1 / |T| (sc2 = sc^2): the number of sign changes of p_-1 = 1, p_0, .., p_{n-1}.
// Three vector instructions per step: e^2 p_{i-2}, the fma, and v_alignbit shifting the sign bit into a history word whose
// transitions are counted once per eight steps.  An exact zero counts as positive and a decoupled matrix (e = 0) can
// pin the sequence at zero: such counts are wrong by construction, which is why the caller never trusts this form for the
// final bracket.
template <typename T>
__device__ __forceinline__ int tri_sturm_product(const T* __restrict__ dd, const T* __restrict__ e2, int n, T sigma, T sc,
                                                 T sc2) {
  constexpr int U = 8;
  T pm = T(1);                                              // p_{i-2}
  T pc = (dd[0] - sigma) * sc;                              // p_{i-1}
  unsigned hist = (unsigned)tri_signword(pc) >> 31;         // bit 0 = sign of the newest value (p_-1 > 0 behind it)
  int cnt = (int)hist;
  int i = 1;
  T dn[U], en[U];
  if (i + U <= n) {
#pragma unroll
    for (int u = 0; u < U; ++u) { dn[u] = dd[i + u]; en[u] = e2[i + u - 1]; }
  }
  for (; i + U <= n; i += U) {
    T dc[U], ec[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { dc[u] = (dn[u] - sigma) * sc; ec[u] = en[u] * sc2; }
    if (i + 2 * U <= n) {
#pragma unroll
      for (int u = 0; u < U; ++u) { dn[u] = dd[i + U + u]; en[u] = e2[i + U + u - 1]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const T pn = fma(dc[u], pc, -(ec[u] * pm));
      hist = __builtin_amdgcn_alignbit(hist, (unsigned)tri_signword(pn), 31);      // (hist << 1) | sign
      pm = pc; pc = pn;
    }
    cnt += __builtin_popcount((hist ^ (hist >> 1)) & 0xffu);                        // the eight new transitions
    // |d - s| sc <= 2 and e^2 sc^2 <= 4: eight steps grow a value by < 6^8 and shrink it by > eps^8 — far inside the range
    const int ex = tri_exponent(fmax(fabs(pc), fabs(pm)));
    pc = tri_scale_down(pc, ex);
    pm = tri_scale_down(pm, ex);
  }
  for (; i < n; ++i) {
    const T pn = fma((dd[i] - sigma) * sc, pc, -(e2[i - 1] * sc2 * pm));
    hist = __builtin_amdgcn_alignbit(hist, (unsigned)tri_signword(pn), 31);
    cnt += (int)((hist ^ (hist >> 1)) & 1u);
    pm = pc; pc = pn;
  }
  return cnt;
}

// eigenvalue number `target` (1-based, ascending) of (d, e) inside the Gershgorin interval [gl, gu]; all lanes of the wave
// call it together and get the same value
template <typename T>
__device__ __forceinline__ T tri_bisect_wave(const T* __restrict__ dd, const T* __restrict__ e2, int n, int target, T gl,
                                             T gu, T tnorm, T pivmin, T eps, int lane) {
  const T slack = T(2) * eps * tnorm * n + T(2) * pivmin;
  T lo = gl - slack, hi = gu + slack;
  const T sc = tnorm > T(0) && tnorm < T(INFINITY) ? T(1) / tnorm : T(1);
  const T sc2 = sc * sc;
  const T coarse = T(4) * eps * tnorm;
  // ---- product form down to 4 eps |T|
  bool narrowed = false;
  for (int round = 0; round < 16; ++round) {
    const T width = hi - lo;
    if (!(width > coarse)) break;
    const T sig = lo + width * (T(lane + 1) / T(65));
    const int c = tri_sturm_product(dd, e2, n, sig, sc, sc2);
    const unsigned long long ge = __ballot(c >= target);
    const int f = ge ? __ffsll((long long)ge) - 1 : 64;
    const T sig_f = __shfl(sig, f < 64 ? f : 63, 64);
    const T sig_fm = __shfl(sig, f > 0 ? f - 1 : 0, 64);
    const T nlo = f > 0 ? sig_fm : lo;
    const T nhi = f < 64 ? sig_f : hi;
    if (!(nhi > nlo)) break;
    lo = nlo; hi = nhi;
    narrowed = true;
  }
  if (narrowed) { lo -= T(4) * eps * tnorm; hi += T(4) * eps * tnorm; }
  // ---- ratio form: lane l sits at lo + width l / 63 in the first round (the end points confirm the bracket)
  bool confirm = narrowed;
  for (int round = 0; round < 32; ++round) {
    const T width = hi - lo;
    if (!confirm && !(width > T(2) * eps * fmax(fabs(lo), fabs(hi)) + T(2) * pivmin)) break;
    const T sig = confirm ? (lane == 63 ? hi : lo + width * (T(lane) / T(63))) : lo + width * (T(lane + 1) / T(65));
    const int c = tri_sturm_ratio(dd, e2, n, sig, pivmin);
    const unsigned long long ge = __ballot(c >= target);
    const int f = ge ? __ffsll((long long)ge) - 1 : 64;
    if (confirm) {
      confirm = false;
      if (f == 0 || f == 64) {                              // count(lo) >= target or count(hi) < target: not a bracket
        lo = gl - slack; hi = gu + slack;
        continue;
      }
    }
    const T sig_f = __shfl(sig, f < 64 ? f : 63, 64);
    const T sig_fm = __shfl(sig, f > 0 ? f - 1 : 0, 64);
    const T nlo = f > 0 ? sig_fm : lo;
    const T nhi = f < 64 ? sig_f : hi;
    if (!(nhi > nlo)) break;
    lo = nlo; hi = nhi;
  }
  return T(0.5) * (lo + hi);
}

}  // namespace xk
