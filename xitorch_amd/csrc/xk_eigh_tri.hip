// xitorch_amd :: K3t — the p wanted eigenpairs of the small Rayleigh–Ritz matrix by Householder
// tridiagonalisation + bisection + inverse iteration (the LAPACK dsyevx route: dsytd2 / dstebz / dstein / dormtr),
// one workgroup per batch member, everything resident in LDS.
//
// Why a second small eigensolver: the parallel Jacobi kernel (xk_eigh.hip) diagonalises the WHOLE k x k matrix —
// ~8 sweeps x (k-1) steps x 2 barriers, every step moving the full matrix through LDS — although the Davidson
// loop (xitorch/_impls/linalg/symeig.py:174-175: torch.linalg.eigh + _take_eigpairs) only looks at p << k pairs.
// Measured on the strong-scaling shard of BASELINE configs[1] (8 operators per GPU, N = 16384): Jacobi 16.1 ms of a
// 50.3 ms call (2.2 ms per launch at k = 108), i.e. more than half of what is not the operator-panel product.
// Here the O(k^3) part is ONE tridiagonalisation (k-2 Householder steps, 3 barriers each, the trailing block
// shrinking), and everything that depends on p is O(k p) or O(k^2 p / threads):
//
//   1. tridiagonalise:   T = Q (d, e) Q^T,  Q = H_0 ... H_{k-3};  reflector j is kept in column j of the LDS copy
//   2. bisection:        one wave per wanted eigenvalue, 64-section of the Sturm count (each lane one shift),
//                        ~10 rounds to the last bit                                           (dstebz)
//   3. inverse iteration: lane j factorises (d, e) - lam_j I by LU with partial pivoting and solves 3 times,
//                        all p systems in SIMD; modified Gram–Schmidt inside clusters, vectors normalised (dstein)
//   4. back-transform:   y_j = H_0 ... H_{k-3} z_j, one wave per vector                        (dormtr)
//   5. check:            max |(d,e) z - lam z| and the orthonormality of the z_j; anything suspicious sets info[b]
//                        and the caller re-runs that call on the Jacobi kernel
//
// Eigenvalues ascending, lowest / uppermost p selected exactly like `_take_eigpairs` (symeig.py:255-264).
#include "xk_common.h"
#include "xk_tridiag.h"

namespace xk {

template <typename T> struct EpsT;
template <> struct EpsT<double> { static constexpr double eps = 2.220446049250313e-16; static constexpr double tiny = 2.2250738585072014e-308; };
template <> struct EpsT<float> { static constexpr float eps = 1.1920929e-07f; static constexpr float tiny = 1.17549435e-38f; };

constexpr int TRI_MAXP = 16;

// wave-uniform lane index -> v_readlane (scalar result, no LDS crossbar trip like a variable-index shuffle)
__device__ __forceinline__ double readlane_t(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane_t(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
// reciprocal to ~1-2 ulp: hardware estimate + Newton steps (a full IEEE division costs ~4x as much on the
// sequential critical paths of the Sturm count and of the triangular solves)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float fast_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  r = fmaf(fmaf(-x, r, 1.0f), r, r);
  return r;
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <typename T>
__global__ __launch_bounds__(1024) void tridiag_eigh_kernel(
    const T* __restrict__ Tin, T* __restrict__ lam_out, T* __restrict__ Y_out, int* __restrict__ info_out,
    int n, int p, int uppest, long ldt, long sT, long long* __restrict__ dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ld = n | 1;                               // odd pitch: column walks are conflict-free
  T* S = reinterpret_cast<T*>(smem);                  // n x ld   (full symmetric copy; column j <- reflector j)
  T* vv = S + (long)n * ld;                           // n  current Householder vector (rows <= j: 0)
  T* ww = vv + n;                                     // n  w = tau A v
  T* dd = ww + n;                                     // n  diagonal of the tridiagonal matrix
  T* ee = dd + n;                                     // n  sub-diagonal (ee[i] = (i+1, i))
  T* e2 = ee + n;                                     // n  squares
  T* tau = e2 + n;                                    // n
  T* red = tau + n;                                   // 16 scratch scalars
  T* lamv = red + 16;                                 // TRI_MAXP eigenvalues
  T* Z = lamv + TRI_MAXP;                             // p x n eigenvectors of (d, e)
  T* lu = Z + (long)p * n;                            // 5 x n x p: dl, d, du, du2, swap flag (index [a][i][j])
  const int b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, nw = nt >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform (scalar row indices)
  const T* Tb = Tin + (long)b * sT;
  const T eps = EpsT<T>::eps;

  // ---- load: lower triangle (eigh's UPLO = 'L') mirrored ---------------------------------------------
  for (int idx = tid; idx < n * n; idx += nt) {
    const int i = idx / n, j = idx - i * n;
    S[i * ld + j] = (i >= j) ? Tb[(long)i * ldt + j] : Tb[(long)j * ldt + i];
  }
  __syncthreads();

#define XK_TRI_STAMP(slot) if (dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) dbg[slot] = (long long)__builtin_readcyclecounter();
  XK_TRI_STAMP(0)
  // ---- 1. Householder tridiagonalisation (dsytd2, lower) ----------------------------------------------
  // Two barriers per step.  Every wave computes the reflector of column j itself (same data, same operations ->
  // the same numbers), keeps it in registers (lane <-> rows/columns j+1+lane, j+1+lane+64) and
  //   (i)  accumulates, for ITS rows i, the column form of the product  w_c += S[i][c] v_i  (S is symmetric):
  //        no cross-lane reduction per row, the per-wave partial rows go to LDS;               [barrier]
  //   (ii) sums the partials for its own columns, forms K = tau/2 w.v and q = w - K v in registers and applies
  //        the rank-2 update to its rows (v_i, q_i of a row come from the owning lane by a shuffle).   [barrier]
  T* part = lu;                                       // nw x n partial products (the LU area is idle until step 3)
  for (int j = 0; j + 2 < n; ++j) {
    const int c0 = j + 1 + lane, c1 = c0 + 64;        // this lane's rows == columns of the trailing block
    const T x0 = c0 < n ? S[c0 * ld + j] : T(0);
    const T x1 = c1 < n ? S[c1 * ld + j] : T(0);
    const T sigma = wave_sum_dpp((lane > 0 ? x0 * x0 : T(0)) + x1 * x1);
    const T alpha = readlane_t(x0, 0);
    T tj = T(0), scale = T(0), beta = alpha;
    if (!(sigma == T(0))) {                           // (a NaN column must poison the result, not be skipped)
      const T nrm = sqrt(alpha * alpha + sigma);
      beta = alpha >= T(0) ? -nrm : nrm;
      tj = (beta - alpha) * fast_rcp(beta);           // (every wave repeats this: keep it off the divider)
      scale = fast_rcp(alpha - beta);
    }
    const T v0 = lane == 0 ? T(1) : x0 * scale;       // v over rows j+1.. (v[j+1] = 1)
    const T v1 = x1 * scale;
    if (tj != T(0)) {                                 // (wave-uniform and identical in every wave)
      T a0 = T(0), a1 = T(0), b0 = T(0), b1 = T(0);
      int i = j + 1 + wave;
      for (; i + nw < n; i += 2 * nw) {               // two rows per trip: independent chains
        const int r0 = i - j - 1, r1 = r0 + nw;
        const T vi0 = readlane_t(r0 < 64 ? v0 : v1, r0 & 63);
        const T vi1 = readlane_t(r1 < 64 ? v0 : v1, r1 & 63);
        const T s00 = c0 < n ? S[i * ld + c0] : T(0), s01 = c1 < n ? S[i * ld + c1] : T(0);
        const T s10 = c0 < n ? S[(i + nw) * ld + c0] : T(0), s11 = c1 < n ? S[(i + nw) * ld + c1] : T(0);
        a0 += s00 * vi0; a1 += s01 * vi0;
        b0 += s10 * vi1; b1 += s11 * vi1;
      }
      if (i < n) {
        const int r0 = i - j - 1;
        const T vi0 = readlane_t(r0 < 64 ? v0 : v1, r0 & 63);
        a0 += (c0 < n ? S[i * ld + c0] : T(0)) * vi0;
        a1 += (c1 < n ? S[i * ld + c1] : T(0)) * vi0;
      }
      if (c0 < n) part[wave * n + c0] = a0 + b0;
      if (c1 < n) part[wave * n + c1] = a1 + b1;
    }
    __syncthreads();
    if (tj != T(0)) {
      T w0 = T(0), w1 = T(0);
      for (int ww_ = 0; ww_ < nw; ++ww_) {
        if (c0 < n) w0 += part[ww_ * n + c0];
        if (c1 < n) w1 += part[ww_ * n + c1];
      }
      w0 *= tj; w1 *= tj;
      const T K = T(0.5) * tj * wave_sum_dpp(w0 * v0 + w1 * v1);
      const T q0 = w0 - K * v0, q1 = w1 - K * v1;
      int i = j + 1 + wave;
      for (; i + nw < n; i += 2 * nw) {               // two rows per trip: their LDS round trips overlap
        const int r0 = i - j - 1, r1 = r0 + nw;
        const T via = readlane_t(r0 < 64 ? v0 : v1, r0 & 63), qia = readlane_t(r0 < 64 ? q0 : q1, r0 & 63);
        const T vib = readlane_t(r1 < 64 ? v0 : v1, r1 & 63), qib = readlane_t(r1 < 64 ? q0 : q1, r1 & 63);
        T sa0 = T(0), sa1 = T(0), sb0 = T(0), sb1 = T(0);
        if (c0 < n) { sa0 = S[i * ld + c0]; sb0 = S[(i + nw) * ld + c0]; }
        if (c1 < n) { sa1 = S[i * ld + c1]; sb1 = S[(i + nw) * ld + c1]; }
        if (c0 < n) { S[i * ld + c0] = sa0 - (via * q0 + qia * v0); S[(i + nw) * ld + c0] = sb0 - (vib * q0 + qib * v0); }
        if (c1 < n) { S[i * ld + c1] = sa1 - (via * q1 + qia * v1); S[(i + nw) * ld + c1] = sb1 - (vib * q1 + qib * v1); }
      }
      if (i < n) {
        const int r0 = i - j - 1;
        const T vi = readlane_t(r0 < 64 ? v0 : v1, r0 & 63);
        const T qi = readlane_t(r0 < 64 ? q0 : q1, r0 & 63);
        if (c0 < n) S[i * ld + c0] -= vi * q0 + qi * v0;
        if (c1 < n) S[i * ld + c1] -= vi * q1 + qi * v1;
      }
    }
    if (wave == 0) {
      // column j is no longer read by anybody: park the reflector there for the back-transformation
      if (lane > 0 && c0 < n) S[c0 * ld + j] = v0;
      if (c1 < n) S[c1 * ld + j] = v1;
      if (lane == 0) { tau[j] = tj; ee[j] = beta; dd[j] = S[j * ld + j]; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (n >= 2) { dd[n - 2] = S[(n - 2) * ld + (n - 2)]; ee[n - 2] = S[(n - 1) * ld + (n - 2)]; }
    dd[n - 1] = S[(n - 1) * ld + (n - 1)];
    if (n >= 1) ee[n - 1] = T(0);
  }
  __syncthreads();
  if (tid < n) e2[tid] = ee[tid] * ee[tid];
  __syncthreads();

  XK_TRI_STAMP(1)
  // ---- 2. bisection: wave w -> wanted eigenvalue number w (ascending) ---------------------------------
  // Gershgorin interval and scale (every wave computes them itself)
  T gl = T(INFINITY), gu = T(-INFINITY), emax = T(0);
  for (int i = lane; i < n; i += 64) {
    const T r = (i > 0 ? fabs(ee[i - 1]) : T(0)) + (i < n - 1 ? fabs(ee[i]) : T(0));
    gl = fmin(gl, dd[i] - r);
    gu = fmax(gu, dd[i] + r);
    emax = fmax(emax, e2[i]);
  }
  gl = -wave_max(-gl);
  gu = wave_max(gu);
  emax = wave_max(emax);
  const T tnorm = fmax(fabs(gl), fabs(gu));
  const T pivmin = EpsT<T>::tiny * fmax(T(1), emax);
  for (int w = wave; w < p; w += nw) {
    const int target = (uppest ? n - p + w : w) + 1;       // smallest sigma with count(sigma) >= target
    const T lamw = tri_bisect_wave<T>(dd, e2, n, target, gl, gu, tnorm, pivmin, eps, lane);   // (xk_tridiag.h)
    if (lane == 0) lamv[w] = lamw;
  }
  __syncthreads();

  XK_TRI_STAMP(2)
  if (tid == 0) red[15] = T(0);                              // "an iterate was annihilated" flag of step 3
  // (each wanted eigenvalue was bracketed on its own: inside a cluster two results may sit an ulp out of order)
  __syncthreads();
  if (tid == 0)
    for (int j = 1; j < p; ++j) lamv[j] = fmax(lamv[j], lamv[j - 1]);
  __syncthreads();
  // ---- 3. inverse iteration (dstein): lane j of wave 0 owns eigenvalue j ---------------------------------
  // coincident eigenvalues get distinct shifts so that their factorizations (and iterates) differ
  const T pfloor = eps * tnorm + pivmin;                     // floor of a pivot's magnitude
  if (tid < p) {
    const int j = tid;
    // (dstein: xj = xjm + pertol when eigenvalues coincide to working precision)
    T shift = lamv[j];
    for (int q = j - 1; q >= 0; --q) {
      if (lamv[j] - lamv[q] < T(10) * eps * tnorm) shift += T(10) * eps * tnorm; else break;
    }
    T* dl = lu + ((long)0 * n) * p + j;
    T* dg = lu + ((long)1 * n) * p + j;
    T* du = lu + ((long)2 * n) * p + j;
    T* du2 = lu + ((long)3 * n) * p + j;
    T* sw = lu + ((long)4 * n) * p + j;
#define AT(arr, i) (arr)[(long)(i) * p]
    // LU with partial pivoting (dgttrf), streamed: row i of the factorisation lives in (dcur, ucur), only the
    // never-modified (d, e) are read, every factor is written once (the sequential loops of this phase cost LDS round
    // trips, not arithmetic); dg holds the RECIPROCAL pivots (one reciprocal here instead of three divisions later)
    T dcur = dd[0] - shift, ucur = n > 1 ? ee[0] : T(0);
    for (int i = 0; i + 1 < n; ++i) {
      const T li = ee[i];
      const T dn = dd[i + 1] - shift;
      const T un = (i + 2 < n) ? ee[i + 1] : T(0);
      if (fabs(dcur) >= fabs(li)) {
        if (fabs(dcur) < pfloor) dcur = dcur < T(0) ? -pfloor : pfloor;
        const T inv = fast_rcp(dcur);
        const T fact = li * inv;
        AT(dl, i) = fact; AT(dg, i) = inv; AT(du, i) = ucur; AT(du2, i) = T(0); AT(sw, i) = T(0);
        dcur = dn - fact * ucur;
        ucur = un;
      } else {
        const T inv = fast_rcp(li);
        const T fact = dcur * inv;
        AT(dl, i) = fact; AT(dg, i) = inv; AT(du, i) = dn; AT(du2, i) = un; AT(sw, i) = T(1);
        dcur = ucur - fact * dn;
        ucur = -fact * un;
      }
    }
    if (fabs(dcur) < pfloor) dcur = dcur < T(0) ? -pfloor : pfloor;
    AT(dg, n - 1) = fast_rcp(dcur);
    // start vector: deterministic pseudo-random in (-1, 1)
    T* z = Z + (long)j * n;
    for (int i = 0; i < n; ++i) {
      const unsigned h = hash32((unsigned)(i * 131 + j * 7919 + 12345));
      z[i] = T((int)(h & 0xffffff) - 0x800000) / T(0x800000);
    }
  }
  __syncthreads();
  for (int it = 0; it < 3; ++it) {
    if (tid < p) {
      const int j = tid;
      T* dl = lu + ((long)0 * n) * p + j;
      T* dg = lu + ((long)1 * n) * p + j;
      T* du = lu + ((long)2 * n) * p + j;
      T* du2 = lu + ((long)3 * n) * p + j;
      T* sw = lu + ((long)4 * n) * p + j;
      T* z = Z + (long)j * n;
      // forward substitution with the recorded row interchanges (running value carried in a register); the operands
      // of 8 steps are fetched before their dependent chain
      constexpr int SU = 8;
      T cur = z[0];
      int i = 0;
      for (; i + SU <= n - 1; i += SU) {
        T nx[SU], l[SU], s_[SU], out[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) { nx[u] = z[i + 1 + u]; l[u] = AT(dl, i + u); s_[u] = AT(sw, i + u); }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          if (s_[u] == T(0)) { out[u] = cur; cur = nx[u] - l[u] * cur; }
          else { out[u] = nx[u]; cur = cur - l[u] * nx[u]; }
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) z[i + u] = out[u];
      }
      for (; i + 1 < n; ++i) {
        const T nxt = z[i + 1];
        const T l = AT(dl, i);
        if (AT(sw, i) == T(0)) { z[i] = cur; cur = nxt - l * cur; }
        else { z[i] = nxt; cur = cur - l * nxt; }
      }
      // back substitution
      T zp1 = cur * AT(dg, n - 1), zp2 = T(0);
      z[n - 1] = zp1;
      if (n > 1) {
        const T t = (z[n - 2] - AT(du, n - 2) * zp1) * AT(dg, n - 2);
        z[n - 2] = t;
        zp2 = zp1; zp1 = t;
      }
      i = n - 3;
      for (; i - (SU - 1) >= 0; i -= SU) {
        T zz[SU], a_[SU], b_[SU], g_[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) { zz[u] = z[i - u]; a_[u] = AT(du, i - u); b_[u] = AT(du2, i - u); g_[u] = AT(dg, i - u); }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const T t = (zz[u] - a_[u] * zp1 - b_[u] * zp2) * g_[u];
          zz[u] = t;
          zp2 = zp1; zp1 = t;
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) z[i - u] = zz[u];
      }
      for (; i >= 0; --i) {
        const T t = (z[i] - AT(du, i) * zp1 - AT(du2, i) * zp2) * AT(dg, i);
        z[i] = t;
        zp2 = zp1; zp1 = t;
      }
    }
#undef AT
    __syncthreads();
    // modified Gram–Schmidt + 2-norm normalisation: wave 0, lanes over the vector
    if (wave == 0) {
      for (int j = 0; j < p; ++j) {
        T* zj = Z + (long)j * n;
        // scale to unit max-norm first (keeps the sums below and the next solve in range)
        T mx = T(0);
        for (int i = lane; i < n; i += 64) mx = fmax(mx, fabs(zj[i]));
        mx = wave_max(mx);
        const T sc = (mx > T(0) && mx < T(INFINITY)) ? T(1) / mx : T(1);
        for (int i = lane; i < n; i += 64) zj[i] *= sc;
        // (dstein re-orthogonalises inside clusters only; with p <= 16 vectors orthogonalising against ALL
        // previous ones costs nothing and removes the eps |T| / gap cross-talk of nearby eigenvalues as well)
        for (int q = j - 1; q >= 0; --q) {
          const T* zq = Z + (long)q * n;
          T dp = T(0);
          for (int i = lane; i < n; i += 64) dp += zq[i] * zj[i];
          dp = wave_sum_dpp(dp);
          for (int i = lane; i < n; i += 64) zj[i] -= dp * zq[i];
        }
        T nn = T(0);
        for (int i = lane; i < n; i += 64) nn += zj[i] * zj[i];
        nn = wave_sum_dpp(nn);
        const T inv = nn > T(0) ? rsqrt(nn) : T(0);
        // a vector annihilated by the Gram-Schmidt step (or non-finite) would pass the residual / overlap checks
        // below as a zero vector: remember it (red[15] is read with the self-check)
        if (lane == 0 && !(nn > T(0) && nn < T(INFINITY))) red[15] = T(1);
        for (int i = lane; i < n; i += 64) zj[i] *= inv;
      }
    }
    __syncthreads();
  }

  XK_TRI_STAMP(3)
  // ---- 5. checks on the tridiagonal level (the reduction itself is backward stable) ----------------------
  if (wave == 0) {
    T worst = T(0);
    int nonfinite = 0;                               // fmax / comparisons drop NaN: track it explicitly
    for (int j = 0; j < p; ++j) {
      const T* zj = Z + (long)j * n;
      const T lam = lamv[j];
      T r = T(0);
      for (int i = lane; i < n; i += 64) {
        T t = (dd[i] - lam) * zj[i];
        if (i > 0) t += ee[i - 1] * zj[i - 1];
        if (i < n - 1) t += ee[i] * zj[i + 1];
        if (!(fabs(t) < T(INFINITY))) nonfinite = 1;
        r = fmax(r, fabs(t));
      }
      r = wave_max(r);
      worst = fmax(worst, r);
      // orthogonality against the previous vector (neighbouring eigenvalues are the risky ones)
      if (j > 0) {
        const T* zq = Z + (long)(j - 1) * n;
        T dp = T(0);
        for (int i = lane; i < n; i += 64) dp += zq[i] * zj[i];
        dp = fabs(wave_sum_dpp(dp));
        if (!(dp < T(INFINITY))) nonfinite = 1;
        worst = fmax(worst, dp * tnorm);
      }
    }
    nonfinite = __any(nonfinite) ? 1 : 0;
    if (!(tnorm < T(INFINITY))) nonfinite = 1;
    if (!tri_scale_in_range(tnorm)) nonfinite = 1;          // (xk_tridiag.h: outside the range the reduction is safe in)
    if (red[15] != T(0)) nonfinite = 1;                      // zero / non-finite iterate in the last normalisation
    if (lane == 0) {
      const T tol = T(100) * eps * tnorm + T(8) * pivmin;
      info_out[b] = (worst <= tol && !nonfinite) ? 0 : 1;
    }
  }

  __syncthreads();
  XK_TRI_STAMP(4)
  // ---- 4. back-transformation y = H_0 ... H_{n-3} z, one wave per vector --------------------------------
  for (int j = wave; j < p; j += nw) {
    T* zj = Z + (long)j * n;
    // lane owns rows lane, lane + 64
    T y0 = lane < n ? zj[lane] : T(0);
    T y1 = lane + 64 < n ? zj[lane + 64] : T(0);
    for (int r = n - 3; r >= 0; --r) {
      const T tr = tau[r];
      if (tr == T(0)) continue;
      // v_r: rows <= r are 0, row r+1 is 1, rows > r+1 in S[:, r]
      const T v0 = (lane >= n || lane <= r) ? T(0) : (lane == r + 1 ? T(1) : S[lane * ld + r]);
      const int i1 = lane + 64;
      const T v1 = (i1 >= n || i1 <= r) ? T(0) : (i1 == r + 1 ? T(1) : S[i1 * ld + r]);
      const T dp = wave_sum_dpp(v0 * y0 + v1 * y1);
      y0 -= tr * dp * v0;
      y1 -= tr * dp * v1;
    }
    T* Yb = Y_out + ((long)b * p + j) * n;
    if (lane < n) Yb[lane] = y0;
    if (lane + 64 < n) Yb[lane + 64] = y1;
    if (lane == 0) lam_out[(long)b * p + j] = lamv[j];
  }
  __syncthreads();
  XK_TRI_STAMP(5)
}

}  // namespace xk

extern "C" {

/* LDS bytes the kernel needs for order k and p wanted pairs (elem_size 8 / 4); the caller compares with 160 KiB */
long xk_small_eigh_tri_lds_bytes(int k, int p, int elem_size) {
  const long n = k, ld = n | 1;
  const long scratch = 5L * n * p > 16L * n ? 5L * n * p : 16L * n;   /* LU factors, aliased by the 16 x n partial rows */
  const long elems = n * ld + 6 * n + 16 + xk::TRI_MAXP + (long)p * n + scratch;
  return elems * elem_size + 64;
}

#define XK_DEFINE_EIGH_TRI(SUF, T)                                                                          \
  int xk_small_eigh_tri_##SUF(const T* Tin, T* lam, T* Y, int* info, int B, int k, int p, int uppest,        \
                              long ldt, long sT, int threads, long long* profile, void* stream) {            \
    if (B < 0 || k < 1 || p < 1 || p > k || k > 128 || p > xk::TRI_MAXP) return XK_ERR_ARG;                  \
    if (threads != 0 && (threads < 64 || threads > 1024 || threads % 64)) return XK_ERR_ARG;                 \
    if (B == 0) return XK_OK;                                                                                \
    const long lds = xk_small_eigh_tri_lds_bytes(k, p, (int)sizeof(T));                                      \
    if (lds > 160 * 1024) return XK_ERR_UNSUPPORTED;                                                         \
    hipError_t e = hipFuncSetAttribute((const void*)xk::tridiag_eigh_kernel<T>,                              \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                \
    if (e != hipSuccess) return (int)e;                                                                      \
    /* threads: 512 (8 waves) measured best at every order: 64 / 128 / 256 -> 2.9x / 1.7x / 1.2x slower (the    */  \
    /* LDS-latency chains of a step want other waves to hide behind), 1024 -> 1.1x slower (barrier skew and the  */  \
    /* reflector arithmetic that every wave repeats)                                                              */  \
    const int nthr = threads > 0 ? threads : 512;                                                            \
    hipLaunchKernelGGL((xk::tridiag_eigh_kernel<T>), dim3(B), dim3(nthr), (size_t)lds, (hipStream_t)stream,  \
                       Tin, lam, Y, info, k, p, uppest, ldt, sT, profile);                                   \
    XK_LAUNCH_CHECK();                                                                                       \
    return XK_OK;                                                                                            \
  }

XK_DEFINE_EIGH_TRI(f64, double)
XK_DEFINE_EIGH_TRI(f32, float)

}  // extern "C"
