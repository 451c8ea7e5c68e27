// xitorch_amd :: K3g — the p wanted eigenpairs of Rayleigh–Ritz matrices of order 129 .. 768, the MATRIX in global
// memory, the tridiagonalisation spread over several workgroups per matrix with one launch per Householder step.
// (From order 192 on the tridiagonalisation and the back-transformation come from the two-stage form of
// xk_eigh_band.hip where its band fits the LDS; this file's final kernel serves both.)
//
// The un-restarted Davidson iteration of the reference (xitorch/_impls/linalg/symeig.py:132-135: the basis grows by
// neig vectors per iteration until convergence, :174-175: torch.linalg.eigh of the full T every iteration) reaches
// bases of 300-800 vectors on slowly converging spectra.  The LDS-resident kernels (xk_eigh.hip, xk_eigh_tri.hip)
// stop at order 128; beyond that round 2 fell back to torch.linalg.eigh -> rocSOLVER: 8.7 / 14.7 / 22 ms for the 32
// matrices of a batch group at order 256 / 384 / 512, against 6.4 ms of operator-panel product to hide under, i.e.
// two thirds of the call on the S2 spectrum.
//
// Same route as K3t (LAPACK's dsyevx: dsytd2 / dstebz / dstein / dormtr):
//   1. tridiagonalise:   tridiag_step_kernel, k - 1 launches over W workgroups per matrix (look-ahead Householder on
//                        the upper triangle of a work copy; described at the kernel)
//   then ONE workgroup per matrix (tridiag_eigh_big_kernel), everything in LDS:
//   2. bisection (64 shifts per round and wave), 3. inverse iteration (in batches of pb shifts: the LU factors of
//   5 n pb elements are what limits LDS), 5. self-check, 4. back-transformation y = H_0 ... H_{k-3} z from the reflectors
//   parked in the rows of the work copy (one wave per vector, next reflector prefetched).
// The result is checked like K3t's (residual on the tridiagonal level, orthogonality, non-finite, annihilated iterate);
// a flagged member makes the caller repeat the step on the library solver.
//
// Measured (fp64, p = 6; profiles/r03_k3m_sweep.jsonl), order 192 / 256 / 384 / 512 / 582 / 640 / 768:
//   32 matrices:  1.7 / 2.5 / 5.1 / 8.5 / 11.0 / 13.2 / 18.6 ms   (rocSOLVER eigh: 5.4 / 8.6 / 14.6 / 22.1 / 27.2 / 32.7 / 43.8)
//    4 matrices:  1.7 / 2.4 / 3.9 / 6.1 /  7.9 /  9.3 / 12.7 ms   (rocSOLVER eigh: 4.3 / 6.5 /  9.6 / 12.7 / 14.1 / 16.2 / 20.2)
// at order 582, 32 matrices: step launches 8.4 ms (5.4 / 8.8 / 15.8 / 25.3 us each while 2 / 4 / 8 / 12 column slots of
// 64 are live: ~4.6 us of launch + row j + 1 + reflector, ~3 us of partial sums and reflector j, the rest the row sweep,
// which waits one trip to the Infinity Cache per two rows of a wave), bisection 0.4, inverse iteration 0.6, back-
// transformation 0.4.  Round 3's first version did all of it in one workgroup per matrix (three passes over the
// trailing block per step through ONE CU's L2 path): 3.0 / 5.4 / 13.6 / 28.7 / 44.8 / 56.8 / 91 ms, whatever the batch;
// sweeping only the upper triangle in that form was slower (latency-, not byte-bound), here it is what the launches need
// (8 -> 7.4 ms at 582).  Inside the Davidson pipeline the step kernels share the chip with the panel stream (64 CUs
// left): 15.6 ms at order 582 there (scripts/s2_timeline.py).
#include "xk_common.h"
#include "xk_tridiag.h"

namespace xk {

template <typename T> struct BigEps;
template <> struct BigEps<double> { static constexpr double eps = 2.220446049250313e-16; static constexpr double tiny = 2.2250738585072014e-308; };
template <> struct BigEps<float> { static constexpr float eps = 1.1920929e-07f; static constexpr float tiny = 1.17549435e-38f; };

constexpr int BIG_MAXK = 1536;                            // (r06: was 1024; one launch per Householder step with 24 column slots)
constexpr int BIG_MAXP = 256;                             // (r06: was 64; the LU batches and the finished vectors' rows in Y scale with p)

__device__ __forceinline__ double big_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float big_readlane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ double big_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float big_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  r = fmaf(fmaf(-x, r, 1.0f), r, r);
  return r;
}
// value of element r (0-based, wave-uniform) of a vector distributed as slot[t] of lane l <-> element l + 64 t
template <typename T, int NT>
__device__ __forceinline__ T dist_get(const T (&v)[NT], int r) {
  const int t = r >> 6, l = r & 63;
  T out = T(0);
#pragma unroll
  for (int u = 0; u < NT; ++u)
    if (u == t) out = big_readlane(v[u], l);        // t is wave-uniform: a scalar branch per slot
  return out;
}

__device__ __forceinline__ unsigned big_hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <typename T, int NT>
__global__ __launch_bounds__(512) void tridiag_eigh_big_kernel(
    T* __restrict__ Sws, const T* __restrict__ aux, long aux_stride, T* __restrict__ lam_out, T* __restrict__ Y_out,
    int* __restrict__ info_out, int n, int p, int pb, int uppest, int stop_after, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* dd = reinterpret_cast<T*>(smem);                 // n  diagonal of the tridiagonal matrix
  T* ee = dd + n;                                     // n  sub-diagonal
  T* e2 = ee + n;                                     // n  squares
  T* tau = e2 + n;                                    // n
  T* red = tau + n;                                   // 16 scratch scalars
  T* lamv = red + 16;                                 // BIG_MAXP eigenvalues
  T* Zb = lamv + BIG_MAXP;                            // pb x n: the eigenvectors of (d, e) of the batch in work; finished
                                                      // batches wait in the rows of Y_out (global), where step 4 turns them
  T* lu = Zb + (long)pb * n;                          // 5 x n x pb LU factors
  const int b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, nw = nt >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  T* S = Sws + (long)b * n * n;
  T* Yg = Y_out + (long)b * p * n;                    // (B, p, n): row j = eigenvector j, first of (d, e), then of T
  const T eps = BigEps<T>::eps;

  {
    // the tridiagonalisation was done by the step kernels (tridiag_step_kernel below): (d, e, tau) wait in the aux block,
    // the reflectors are parked in the rows of S (row j, columns > j + 1)
    // (mode 1, the two-stage form of xk_eigh_band.hip: (d, e) are complete in the aux block and the vectors go back
    // through both stages in a kernel of their own)
    const T* ab = aux + (long)b * aux_stride;
    for (int i = tid; i < n; i += nt) {
      T d = ab[i], e = ab[n + i], tv = mode == 1 ? T(0) : ab[2 * n + i];
      if (mode == 0) {                                  // (mode 2, the persistent kernel: d, e, tau are complete)
        if (i == n - 1) { d = S[(long)(n - 1) * n + (n - 1)]; e = T(0); tv = T(0); }
        if (i == n - 2) { e = ab[3 * n + (long)((n - 2) & 1) * n + (n - 1)]; tv = T(0); }
      }
      if (i == n - 1) e = T(0);
      dd[i] = d; ee[i] = e; tau[i] = tv; e2[i] = e * e;
    }
    __syncthreads();
  }

#ifdef XK_FINAL_DBG
  long long tph[6]; int iph = 0; tph[iph++] = __builtin_readcyclecounter();
#define XK_FSTAMP() { __syncthreads(); tph[iph++] = __builtin_readcyclecounter(); }
  long long tLU = 0, tSOL = 0, tGS = 0, ti0 = 0;
#define XK_ISTAMP(acc) { const long long t1_ = __builtin_readcyclecounter(); acc += t1_ - ti0; ti0 = t1_; }
#else
#define XK_FSTAMP()
#define XK_ISTAMP(acc)
#endif
  // ---- 2. bisection: wave w -> wanted eigenvalue number w (ascending) ---------------------------------
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
  const T pivmin = BigEps<T>::tiny * fmax(T(1), emax);
  for (int w = wave; w < p; w += nw) {
    const int target = (uppest ? n - p + w : w) + 1;
    const T lamw = tri_bisect_wave<T>(dd, e2, n, target, gl, gu, tnorm, pivmin, eps, lane);   // (xk_tridiag.h)
    if (lane == 0) lamv[w] = lamw;
  }
  if (tid == 0) red[15] = T(0);                       // "an iterate was annihilated / non-finite" flag of step 3
  // (each wanted eigenvalue was bracketed on its own: inside a cluster two results may sit an ulp out of order)
  __syncthreads();
  if (tid == 0)
    for (int j = 1; j < p; ++j) lamv[j] = fmax(lamv[j], lamv[j - 1]);
  __syncthreads();
  if (stop_after == 2) return;                        // (measurement hook: results are wrong by construction)
  XK_FSTAMP()
#ifdef XK_FINAL_DBG
  ti0 = __builtin_readcyclecounter();
#endif

  // ---- 3. inverse iteration (dstein), vectors in order, pb shifts factorised at a time -----------------
  // One thread per shift runs the sequential recurrences; what they cost is LDS round trips, so (a) the LU streams:
  // row i of the factorisation lives in registers and only the never-modified (d, e) are read, (b) the two solve
  // sweeps fetch 8 steps' operands before their dependent chain, (c) scaling / orthogonalisation are wave-parallel.
  const T pfloor = eps * tnorm + pivmin;
#define AT(arr, i) (arr)[(long)(i) * pb]
  constexpr int SU = 8;
  for (int j0 = 0; j0 < p; j0 += pb) {
    const int nb = p - j0 < pb ? p - j0 : pb;
    if (tid < nb) {
      const int jl = tid, j = j0 + tid;
      T shift = lamv[j];
      for (int q = j - 1; q >= 0; --q) {
        if (lamv[j] - lamv[q] < T(10) * eps * tnorm) shift += T(10) * eps * tnorm; else break;
      }
      T* dl = lu + ((long)0 * n) * pb + jl;           // multipliers
      T* dg = lu + ((long)1 * n) * pb + jl;           // RECIPROCAL pivots
      T* du = lu + ((long)2 * n) * pb + jl;
      T* du2 = lu + ((long)3 * n) * pb + jl;
      T* sw = lu + ((long)4 * n) * pb + jl;           // 1 = rows i, i + 1 were swapped
      T dcur = dd[0] - shift, ucur = n > 1 ? ee[0] : T(0);
      // LU with partial pivoting (dgttrf), row i in (dcur, ucur); the (d, e) of eight rows are fetched before their chain
      // (branch-free: the lanes of the shifts take different pivots, and a divergent branch runs both sides)
      auto lu_row = [&](int i, T li, T dn, T un) {
        const bool keep = fabs(dcur) >= fabs(li);       // no interchange
        T piv = keep ? dcur : li;
        if (keep && fabs(piv) < pfloor) piv = piv < T(0) ? -pfloor : pfloor;
        const T inv = big_rcp(piv);
        const T fact = (keep ? li : dcur) * inv;
        const T up = keep ? ucur : dn;                  // row i of U: (1 / inv, up, up2)
        AT(dl, i) = fact; AT(dg, i) = inv; AT(du, i) = up; AT(du2, i) = keep ? T(0) : un; AT(sw, i) = keep ? T(0) : T(1);
        dcur = (keep ? dn : ucur) - fact * up;
        ucur = keep ? un : -fact * un;
      };
      {
        constexpr int LU = 8;
        int i = 0;
        for (; i + LU + 1 < n; i += LU) {                 // rows i .. i + 7 read e[i .. i + 8], d[i + 1 .. i + 8]
          T ev[LU + 1], dv[LU];
#pragma unroll
          for (int u = 0; u <= LU; ++u) ev[u] = ee[i + u];
#pragma unroll
          for (int u = 0; u < LU; ++u) dv[u] = dd[i + 1 + u] - shift;
#pragma unroll
          for (int u = 0; u < LU; ++u) lu_row(i + u, ev[u], dv[u], ev[u + 1]);
        }
        for (; i + 1 < n; ++i) lu_row(i, ee[i], dd[i + 1] - shift, (i + 2 < n) ? ee[i + 1] : T(0));
      }
      if (fabs(dcur) < pfloor) dcur = dcur < T(0) ? -pfloor : pfloor;
      AT(dg, n - 1) = big_rcp(dcur);
      T* z = Zb + (long)jl * n;
      for (int i = 0; i < n; ++i) {
        const unsigned h = big_hash((unsigned)(i * 131 + j * 7919 + 12345));
        z[i] = T((int)(h & 0xffffff) - 0x800000) / T(0x800000);
      }
    }
    __syncthreads();
    XK_ISTAMP(tLU)
    for (int it = 0; it < 3; ++it) {
      if (tid < nb) {
        const int jl = tid;
        const T* dl = lu + ((long)0 * n) * pb + jl;
        const T* dg = lu + ((long)1 * n) * pb + jl;
        const T* du = lu + ((long)2 * n) * pb + jl;
        const T* du2 = lu + ((long)3 * n) * pb + jl;
        const T* sw = lu + ((long)4 * n) * pb + jl;
        T* z = Zb + (long)jl * n;
        T cur = z[0];
        int i = 0;
        for (; i + SU <= n - 1; i += SU) {            // forward: L^-1 with the row interchanges
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
        T zp1 = cur * AT(dg, n - 1), zp2 = T(0);      // backward: U^-1 (two super-diagonals)
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
      __syncthreads();
      XK_ISTAMP(tSOL)
      // scaling by the largest entry, modified Gram–Schmidt against ALL earlier vectors (finished batches and this
      // batch) and normalisation: wave 0, vectors in order
      if (wave == 0) {
        // (the vector stays in registers, lane l <-> elements l + 64 t: one LDS read and one write per vector instead of
        // one pass through LDS per operation)
        for (int j = j0; j < j0 + nb; ++j) {
          T* zj = Zb + (long)(j - j0) * n;
          T zr[NT];
          T mx = T(0);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int i = lane + 64 * t;
            zr[t] = i < n ? zj[i] : T(0);
            mx = fmax(mx, fabs(zr[t]));
          }
          mx = wave_max(mx);
          const T sc = (mx > T(0) && mx < T(INFINITY)) ? T(1) / mx : T(1);
#pragma unroll
          for (int t = 0; t < NT; ++t) zr[t] *= sc;
          for (int q = j - 1; q >= 0; --q) {
            const T* zq = q >= j0 ? Zb + (long)(q - j0) * n : Yg + (long)q * n;
            T qv[NT];
            T dp = T(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const int i = lane + 64 * t;
              qv[t] = i < n ? zq[i] : T(0);
              dp += qv[t] * zr[t];
            }
            dp = wave_sum_dpp(dp);
#pragma unroll
            for (int t = 0; t < NT; ++t) zr[t] -= dp * qv[t];
          }
          T nn = T(0);
#pragma unroll
          for (int t = 0; t < NT; ++t) nn += zr[t] * zr[t];
          nn = wave_sum_dpp(nn);
          const T inv = nn > T(0) ? rsqrt(nn) : T(0);
          if (lane == 0 && it == 2 && !(nn > T(0) && nn < T(INFINITY))) red[15] = T(1);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int i = lane + 64 * t;
            if (i < n) zj[i] = zr[t] * inv;
          }
        }
      }
      __syncthreads();
      XK_ISTAMP(tGS)
    }
    for (int idx = tid; idx < nb * n; idx += nt) Yg[(long)j0 * n + idx] = Zb[idx];     // rows j0 .. j0 + nb - 1
    __syncthreads();
  }
#undef AT
  if (stop_after == 3) return;
  XK_FSTAMP()

  // ---- 5. checks on the tridiagonal level ----------------------------------------------------------------
  if (wave == 0) {
    T worst = T(0);
    int nonfinite = 0;
    for (int j = 0; j < p; ++j) {
      const T* zj = Yg + (long)j * n;
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
      if (j > 0) {
        const T* zq = Yg + (long)(j - 1) * n;
        T dp = T(0);
        for (int i = lane; i < n; i += 64) dp += zq[i] * zj[i];
        dp = fabs(wave_sum_dpp(dp));
        if (!(dp < T(INFINITY))) nonfinite = 1;
        worst = fmax(worst, dp * tnorm);
      }
    }
    nonfinite = __any(nonfinite) ? 1 : 0;
    if (!(tnorm < T(INFINITY))) nonfinite = 1;
    if (!tri_scale_in_range(tnorm)) nonfinite = 1;        // (xk_tridiag.h: outside the range the reduction is safe in)
    if (red[15] != T(0)) nonfinite = 1;
    if (lane == 0) {
      const T tol = T(100) * eps * tnorm * T(n > 128 ? 4 : 1) + T(8) * pivmin;
      info_out[b] = (worst <= tol && !nonfinite) ? 0 : 1;
    }
  }
  __syncthreads();

  if (stop_after == 5) return;
  XK_FSTAMP()
  if (mode == 1) {                                    // the vectors of (d, e) stay in the rows of Y_out
    for (int j = tid; j < p; j += nt) lam_out[(long)b * p + j] = lamv[j];
    return;
  }
  // ---- 4. back-transformation y = H_0 ... H_{n-3} z, one wave per vector, next reflector prefetched -------
  for (int j = wave; j < p; j += nw) {
    const T* zj = Yg + (long)j * n;
    T y[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int i = lane + 64 * t;
      y[t] = i < n ? zj[i] : T(0);
    }
    // the reflectors of the next DEPTH steps are in flight while one is applied (each is a trip to L2).  The loads are
    // UNCONDITIONAL (clamped addresses, the mask applied where the value is used): a load under a lane mask makes the
    // compiler wait for vmcnt(0) at every use, i.e. for the reflector just requested — 1500 cycles per reflector
    // whatever the depth (profiles/r06_k3_final_phases.json)
    constexpr int DEPTH = NT <= 8 ? 4 : 2;                  // (registers: DEPTH x NT values per lane)
    T vq[DEPTH][NT];
    int ic[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ic[t] = lane + 64 * t < n ? lane + 64 * t : n - 1;
    auto fetch = [&](int rr, T (&dst)[NT]) {
      const T* row = S + (long)(rr < 0 ? 0 : rr) * n;
#pragma unroll
      for (int t = 0; t < NT; ++t) dst[t] = row[ic[t]];
    };
    int r = n - 3;
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) fetch(r - u, vq[u]);
    // Two reflectors per trip: the dots of both with y and their mutual dot go through the wave reduction TOGETHER
    // (d_b = v_b.y - tau_a (v_a.y) (v_a.v_b)), so the dependent chain  dot -> reduction -> update  is paid once per pair
    for (; r >= 0; r -= DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; u += 2) {
        const int ra = r - u, rb = ra - 1;
        // v_r: rows <= r are 0, row r+1 is 1, rows > r+1 parked in row r.  Column slots entirely at or below row r are
        // skipped (uniform), slots entirely above row r+1 and inside the matrix need no mask
        // (a reflector with tau = 0 — none at all past row 0 — is skipped as a whole: its row of S may hold anything)
        const T ta = ra >= 0 ? tau[ra] : T(0);
        const T tb = rb >= 0 ? tau[rb] : T(0);
        const bool la = ta != T(0), lb = tb != T(0);
        T va[NT], vb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          va[t] = T(0); vb[t] = T(0);
          const int i = lane + 64 * t;
          if (la && 64 * t + 63 > ra) {
            if (64 * t > ra + 1 && 64 * t + 63 < n) va[t] = vq[u][t];
            else va[t] = (i == ra + 1) ? T(1) : ((i > ra + 1 && i < n) ? vq[u][t] : T(0));
          }
          if (lb && 64 * t + 63 > rb) {
            if (64 * t > rb + 1 && 64 * t + 63 < n) vb[t] = vq[u + 1][t];
            else vb[t] = (i == rb + 1) ? T(1) : ((i > rb + 1 && i < n) ? vq[u + 1][t] : T(0));
          }
        }
        fetch(ra - DEPTH, vq[u]);
        fetch(rb - DEPTH, vq[u + 1]);
        T da = T(0), db = T(0), cc = T(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (64 * t + 63 > rb) {                            // (slots dead for b are dead for a)
            da += va[t] * y[t];
            db += vb[t] * y[t];
            cc += va[t] * vb[t];
          }
        }
        T red3[4] = {da, db, cc, T(0)};
        wave_reduce_scatter<T, 4>(red3, lane);               // lane groups by bits 5, 4 hold the total of index 0 .. 3
        const T tot = red3[0];
        da = big_readlane(tot, 0);                           // index = bit5 + 2 bit4 of the lane
        db = big_readlane(tot, 32);
        cc = big_readlane(tot, 16);
        const T fa = ta * da;
        const T fb = tb * (db - fa * cc);
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (64 * t + 63 > rb) y[t] -= fa * va[t] + fb * vb[t];
      }
    }
    T* Yb = Y_out + ((long)b * p + j) * n;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int i = lane + 64 * t;
      if (i < n) Yb[i] = y[t];
    }
    if (lane == 0) lam_out[(long)b * p + j] = lamv[j];
  }
#ifdef XK_FINAL_DBG
  XK_FSTAMP()
  if (b == 0 && tid == 0)
    printf("final n=%d p=%d pb=%d cycles: bisect %lld inviter %lld (LU %lld solves %lld GS %lld) check %lld backtr %lld\n", n, p, pb,
           tph[1] - tph[0], tph[2] - tph[1], tLU, tSOL, tGS, tph[3] - tph[2], tph[4] - tph[3]);
#endif
}


// ---- K3m: the tridiagonalisation spread over W workgroups per matrix, ONE KERNEL PER HOUSEHOLDER STEP ----------
// (the kernel boundary is the only cross-workgroup hand-off: no spinning, safe beside CU-masked streams).
// Look-ahead form: launch j applies the rank-2 update of step j to the trailing block and, in the same pass over it,
// accumulates the product of the UPDATED block with the reflector of step j + 1 — one read and one write of the
// trailing block per step instead of two reads and a write.  Every workgroup recomputes the small pieces (reflector j
// from row j, the sum of the partial products of the previous launch, row j + 1 and reflector j + 1) in the same
// order, so all of them hold bit-identical values; rows j + 2 .. n - 1 are dealt round-robin over the W x nw waves.
// Only the upper triangle of the trailing block is read and written: a row contributes the column form of the product
// for c >= i and, through one wave reduction, its own element of the row form for c > i.
// aux block per matrix: d[n], e[n], tau[n], X[2][n] (row j + 1 after update j, double-buffered: launch j reads X[j & 1]
// and writes X[(j + 1) & 1]), R[2][n] (row-form sums), P[2][W][n] (column-form partials), same double buffering.
// FIRST (j = -1): no update, the rows come from the lower triangle of T (eigh's UPLO = 'L') and are stored to S.

// rows i0, i0 + GW, ... (RPT of them) of the trailing block, columns j2 + lane + 64 t >= the row, into registers
template <typename T, int NT, int RPT, bool FIRST>
__device__ __forceinline__ void step_load_rows(T (&sr)[RPT][NT], const T* __restrict__ S, const T* __restrict__ Tb,
                                               long ldt, int n, int j2, int i0, int GW, int lane) {
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    const int i = i0 + u * GW;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = j2 + lane + 64 * t;
      T val = T(0);
      if (c >= i && c < n) {                          // (upper triangle of the trailing block only; i < n follows)
        if (FIRST) val = Tb[(long)c * ldt + i];       // element (i, c) = (c, i) of eigh's lower triangle
        else val = S[(long)i * n + c];
      }
      sr[u][t] = val;
    }
  }
}

// LB: threads per workgroup the instantiation is built for.  16 column slots (orders 769 .. 1024, r05) keep 16 values per
// lane in nine arrays (~330 registers in fp64): built for 256 threads = one wave per SIMD, which may use up to 512.
template <typename T, int NT, bool FIRST, int LB = 512>
__global__ __launch_bounds__(LB) void tridiag_step_kernel(
    const T* __restrict__ Tin, T* __restrict__ Sws, T* __restrict__ aux, long aux_stride, int n, int j, int W,
    long ldt, long sT, int skip) {
  constexpr int RPT = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* wL = reinterpret_cast<T*>(smem);                 // n  A v_j (sum of the partial products of launch j - 1, unscaled)
  T* vL = wL + n;                                     // n  reflector j by absolute column
  T* red = vL + n;                                    // 8  scalars
  T* partL = red + 8;                                 // nw x n
  const int b = blockIdx.y, wg = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lane = tid & 63, nw = nt >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* Tb = Tin + (long)b * sT;
  T* S = Sws + (long)b * n * n;
  T* ab = aux + (long)b * aux_stride;
  T* dd = ab; T* ee = ab + n; T* tau = ab + 2 * n;
  const T* Xcur = ab + 3 * n + (long)(j & 1) * n;
  T* Xnext = ab + 3 * n + (long)((j + 1) & 1) * n;
  const T* Rcur = ab + 5 * n + (long)(j & 1) * n;
  T* Rnext = ab + 5 * n + (long)((j + 1) & 1) * n;
  const T* Pcur = ab + 7 * n + (long)(j & 1) * W * n;
  T* Pnext = ab + 7 * n + (long)((j + 1) & 1) * W * n;
  const int j1 = j + 1, j2 = j + 2;                   // row handled redundantly, first row / column of the next block

  // every global load that does not depend on this launch's reflector is issued before anything else: row j + 1 and
  // the first rows of this wave (one trip to the Infinity Cache is what a small step costs)
  const int GW = W * nw;
  int i0 = j2 + wg * nw + wave;
  if (skip & 1) i0 = n;                               // (measurement hook: no row sweep)
  T s1[NT], sr[RPT][NT];
  T sdiag = T(0);
  if (!FIRST) {
    sdiag = S[(long)j1 * n + j1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = j2 + lane + 64 * t;
      s1[t] = c < n ? S[(long)j1 * n + c] : T(0);
    }
  }
  step_load_rows<T, NT, RPT, FIRST>(sr, S, Tb, ldt, n, j2, i0, GW, lane);

  T tj = T(0), K = T(0);
  if (!FIRST && !(skip & 2)) {
    // ---- A. reflector j (wave 0) and the summed partial products (all threads) --------------------------------
    for (int c = j1 + tid; c < n; c += nt) {          // (all W + 1 loads in flight at once: a dependent loop over g
      T sacc = Rcur[c];                               //  costs W round trips to the Infinity Cache per step)
      for (int g0 = 0; g0 < W; g0 += 8) {
        T pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) pv[u] = g0 + u < W ? Pcur[(long)(g0 + u) * n + c] : T(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) sacc += pv[u];
      }
      wL[c] = sacc;
    }
    if (wave == 0) {
      T ss = T(0);
      T x[NT + 1];
#pragma unroll
      for (int t = 0; t < NT + 1; ++t) {
        const int c = j1 + lane + 64 * t;
        x[t] = c < n ? Xcur[c] : T(0);
        if (!(t == 0 && lane == 0)) ss += x[t] * x[t];
      }
      const T sigma = wave_sum_dpp(ss);
      const T alpha = big_readlane(x[0], 0);
      T tjw = T(0), scale = T(0), beta = alpha;
      if (!(sigma == T(0))) {                         // (a NaN row must poison the result, not be skipped)
        const T nrm = sqrt(alpha * alpha + sigma);
        beta = alpha >= T(0) ? -nrm : nrm;
        tjw = (beta - alpha) * big_rcp(beta);
        scale = big_rcp(alpha - beta);
      }
#pragma unroll
      for (int t = 0; t < NT + 1; ++t) {
        const int c = j1 + lane + 64 * t;
        const T vv = (t == 0 && lane == 0) ? T(1) : x[t] * scale;
        if (c < n) {
          vL[c] = vv;
          if (wg == 0 && c > j1) S[(long)j * n + c] = vv;      // parked: nobody reads row j of S any more
        }
      }
      if (lane == 0) {
        red[0] = tjw;
        if (wg == 0) { tau[j] = tjw; ee[j] = beta; }
      }
    }
    __syncthreads();
    tj = red[0];
    T wv = T(0);
    for (int c = j1 + lane; c < n; c += 64) wv += wL[c] * vL[c];
    K = T(0.5) * tj * tj * wave_sum_dpp(wv);          // w = tj A v;  K = tj/2 w.v
  }

  // ---- B. row j + 1 after update j, reflector j + 1 (every wave, identical) -------------------------------------
  T q2[NT], v2[NT], vN[NT];
  T q0 = T(0);
  {
    T a[NT];
    T dnext;
    if (FIRST) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = j2 + lane + 64 * t;
        a[t] = c < n ? Tb[(long)c * ldt] : T(0);
        q2[t] = T(0); v2[t] = T(0);
      }
      dnext = Tb[0];
    } else {
      q0 = tj * wL[j1] - K;                           // v_j(j + 1) = 1
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = j2 + lane + 64 * t;
        const T vv = c < n ? vL[c] : T(0);
        const T ww = c < n ? wL[c] : T(0);
        v2[t] = vv;
        q2[t] = tj * ww - K * vv;
        a[t] = s1[t] - (q2[t] + q0 * vv);
      }
      dnext = sdiag - T(2) * q0;
    }
    T ss = T(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (!(t == 0 && lane == 0)) ss += a[t] * a[t];
    const T sigma = wave_sum_dpp(ss);
    const T alpha = big_readlane(a[0], 0);
    T scale = T(0);
    if (!(sigma == T(0))) {
      const T nrm = sqrt(alpha * alpha + sigma);
      const T beta = alpha >= T(0) ? -nrm : nrm;
      scale = big_rcp(alpha - beta);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) vN[t] = (t == 0 && lane == 0) ? T(1) : a[t] * scale;
    if (wg == 0 && wave == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int c = j2 + lane + 64 * t;
        if (c < n) Xnext[c] = a[t];
      }
      if (lane == 0) dd[j1] = dnext;
    }
  }

  // ---- C. rows j + 2 .. n - 1: update, store, accumulate the product with reflector j + 1 ------------------------
  T acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = T(0);
  for (; i0 < n; i0 += RPT * GW) {
    T nx[RPT][NT];
    step_load_rows<T, NT, RPT, FIRST>(nx, S, Tb, ldt, n, j2, i0 + RPT * GW, GW, lane);     // next trip, before the stores
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
      const int i = i0 + u * GW;
      if (i < n) {
        const T vNi = dist_get<T, NT>(vN, i - j2);
        T vi = T(0), qi = T(0);
        if (!FIRST) { vi = vL[i]; qi = tj * wL[i] - K * vi; }
        const int tmin = (i - j2) >> 6;               // first column slot that reaches the diagonal (wave-uniform)
        T rs = T(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t >= tmin) {
            const int c = j2 + lane + 64 * t;
            const bool on = c >= i && c < n;
            T an = FIRST ? sr[u][t] : sr[u][t] - (vi * q2[t] + qi * v2[t]);
            an = on ? an : T(0);
            if (on) S[(long)i * n + c] = an;
            acc[t] += an * vNi;                       // column form: (A v)(c) += a(i, c) v(i), c >= i
            rs += c > i ? an * vN[t] : T(0);          // row form:    (A v)(i) += a(i, c) v(c), c > i
          }
        }
        rs = wave_sum_dpp(rs);
        if (lane == 0) Rnext[i] = rs;                 // row i belongs to exactly one wave of the grid: complete
      }
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) sr[u][t] = nx[u][t];
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = j2 + lane + 64 * t;
    if (c < n) partL[wave * n + c] = acc[t];
  }
  if (skip & 4) return;
  __syncthreads();
  for (int c = j2 + tid; c < n; c += nt) {
    T sacc = T(0);
    for (int ww_ = 0; ww_ < nw; ++ww_) sacc += partL[ww_ * n + c];
    Pnext[(long)wg * n + c] = sacc;
  }
}

static long step_lds_elems(long n, long nw) { return 2 * n + 8 + nw * n; }

// LDS elements of the final kernel for order n, p wanted pairs, LU batches of pb shifts
static long big_lds_elems(long n, long p, long pb) { (void)p; return 4 * n + 16 + BIG_MAXP + 6L * n * pb; }

}  // namespace xk

extern "C" int xk_small_eigh_big_batch(int k, int p, int elem_size);

// No process-wide state: the launch shape (workgroups per matrix of the step kernels, their threads) arrives as
// arguments of the entry point (0 = the measured defaults).  The two measurement switches that give WRONG results by
// construction — "leave the final kernel after phase n", "skip parts of the step kernel" — exist only in a -DXK_DEBUG
// build (scripts/k3m_sweep.py builds its own copy); the shipped library passes 0 for both.
#ifdef XK_DEBUG
static int g_big_stop = 0;
static int g_big_skip = 0;
extern "C" int xk_debug_small_eigh_big(int what, int value) {
  int old = -1000;
  if (what == 2) { old = g_big_stop; g_big_stop = value; }
  if (what == 3) { old = g_big_skip; g_big_skip = value; }
  return old;
}
#define XK_BIG_STOP g_big_stop
#define XK_BIG_SKIP g_big_skip
#else
#define XK_BIG_STOP 0
#define XK_BIG_SKIP 0
#endif

#ifndef XK_PERSIST_EXTRA
#define XK_PERSIST_EXTRA 128          // orders up to this far beyond the register-resident limit start with step launches
#endif
namespace xk {
// the two-stage form (xk_eigh_band.hip)
long band_ws_elems(int k);
bool band_supported(int k, int elem_size);
template <typename T>
int band_tridiag(const T* Tin, T* S, T* aux, long aux_stride, T* bws, int B, int k, long ldt, long sT, hipStream_t st);
template <typename T>
int band_back(T* Y, T* bws, int B, int k, int p, hipStream_t st);
// the persistent one-workgroup form (xk_eigh_persist.hip)
int persist_max_order(int elem_size);
int persist_base(int k, int elem_size);
template <typename T>
int persist_tridiag(const T* Tin, T* S, T* aux, long aux_stride, int B, int k, int base, int W, long ldt, long sT,
                    hipStream_t st);

// which form serves order k (algo: 0 = the measured choice, 1 = one launch per Householder step, 2 = two-stage,
// 3 = step launches down to the order the persistent kernel holds in registers, then that kernel)
static bool big_persist(int B, int k, int elem_size, int algo) {
  if (algo == 3) return true;
  if (algo != 0) return false;
  // measured (profiles/r06_k3p_orders.jsonl; fp64, 1 / 4 / 32 matrices): ahead of both other forms up to the register limit
  // (256 / 384) at every batch; beyond it, where the first steps are step launches, up to +128 for a few matrices (order
  // 384: 3.31 vs 3.39 ms two-stage at 1 matrix, 4.19 vs 3.76 at 32) and up to +64 for many (order 330 x 32: 3.06 vs 3.08)
  return k <= persist_max_order(elem_size) + (B <= 8 ? XK_PERSIST_EXTRA : XK_PERSIST_EXTRA / 2);
}
static bool big_two_stage(int k, int elem_size, int algo) {
  if (algo == 1 || algo == 3 || !band_supported(k, elem_size)) return false;
  // measured (profiles/r04_k3g_two_stage.jsonl): ahead from order 192 on for 1 .. 32 matrices, fp64 and fp32
  return algo == 2 || k >= 192;
}

static int big_pick_w(int B, int k, int wg) {
  if (wg > 0) return wg;
  // measured (scripts/k3m_sweep.py, s2_k3_variants.py): alone on the chip 8 workgroups per matrix are fastest for 32
  // matrices; beside the panel stream of the Davidson pipeline, which leaves 64 CUs to everything else, 4 are (128
  // workgroups = two rounds on those CUs instead of four) — the pipeline is where this kernel runs
  int w = 1;
  while (w < 8 && (long)B * (w * 2) <= 128) w *= 2;
  if (k <= 256 && w > 4) w = 4;
  return w;
}

template <typename T, int NT>
static void big_launch_step(const T* Tin, T* S, T* aux, long aux_stride, int B, int k, int j, int W, long ldt, long sT,
                            int nt, hipStream_t st) {
  constexpr int LB = NT > 12 ? 256 : 512;
  if (nt > LB) nt = LB;
  const size_t lds = (size_t)step_lds_elems(k, nt / 64) * sizeof(T);
  if (lds > 65536) {
    // dynamic LDS beyond 64 KB is an opt-in (fp64 orders from 819 on): once per instantiation and process
    static bool opted[2] = {false, false};
    if (!opted[j < 0]) {
      if (j < 0)
        (void)hipFuncSetAttribute((const void*)tridiag_step_kernel<T, NT, true, LB>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      else
        (void)hipFuncSetAttribute((const void*)tridiag_step_kernel<T, NT, false, LB>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      opted[j < 0] = true;
    }
  }
  if (j < 0)
    hipLaunchKernelGGL((tridiag_step_kernel<T, NT, true, LB>), dim3(W, B), dim3(nt), lds, st, Tin, S, aux, aux_stride, k,
                       j, W, ldt, sT, XK_BIG_SKIP);
  else
    hipLaunchKernelGGL((tridiag_step_kernel<T, NT, false, LB>), dim3(W, B), dim3(nt), lds, st, Tin, S, aux, aux_stride, k,
                       j, W, ldt, sT, XK_BIG_SKIP);
}

template <typename T, int NT>
static int big_launch_final(T* ws, const T* aux, long aux_stride, T* lam, T* Y, int* info, int B, int k, int p, int pb,
                            int uppest, long lds, int mode, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute((const void*)tridiag_eigh_big_kernel<T, NT>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((tridiag_eigh_big_kernel<T, NT>), dim3(B), dim3(512), (size_t)lds, st, ws, aux, aux_stride, lam, Y,
                     info, k, p, pb, uppest, XK_BIG_STOP, mode);
  return XK_OK;
}

// the final kernel with the fewest 64-column slots that hold order k (its back-transformation is priced in slots)
template <typename T>
static int big_launch_final_nt(T* ws, const T* aux, long aux_stride, T* lam, T* Y, int* info, int B, int k, int p, int pb,
                               int uppest, long lds, int mode, hipStream_t st) {
  if (k <= 128) return big_launch_final<T, 2>(ws, aux, aux_stride, lam, Y, info, B, k, p, pb, uppest, lds, mode, st);
  if (k <= 256) return big_launch_final<T, 4>(ws, aux, aux_stride, lam, Y, info, B, k, p, pb, uppest, lds, mode, st);
  if (k <= 384) return big_launch_final<T, 6>(ws, aux, aux_stride, lam, Y, info, B, k, p, pb, uppest, lds, mode, st);
  if (k <= 512) return big_launch_final<T, 8>(ws, aux, aux_stride, lam, Y, info, B, k, p, pb, uppest, lds, mode, st);
  if (k <= 768) return big_launch_final<T, 12>(ws, aux, aux_stride, lam, Y, info, B, k, p, pb, uppest, lds, mode, st);
  if (k <= 1024) return big_launch_final<T, 16>(ws, aux, aux_stride, lam, Y, info, B, k, p, pb, uppest, lds, mode, st);
  return big_launch_final<T, 24>(ws, aux, aux_stride, lam, Y, info, B, k, p, pb, uppest, lds, mode, st);
}

template <typename T>
static int big_run(const T* Tin, T* lam, T* Y, T* ws, long ws_elems, int* info, int B, int k, int p, int uppest,
                   long ldt, long sT, int wg, int nt, int algo, hipStream_t st) {
  const int pb = xk_small_eigh_big_batch(k, p, (int)sizeof(T));
  if (pb == 0) return XK_ERR_UNSUPPORTED;
  if (algo == 2 && !band_supported(k, (int)sizeof(T))) return XK_ERR_UNSUPPORTED;
  const long lds = big_lds_elems(k, p, pb) * (long)sizeof(T) + 64;
  const int W = big_pick_w(B, k, wg);
  const long aux_stride = (long)k * (7 + 2 * W);
  if (ws == nullptr || ws_elems < (long)B * k * k + (long)B * aux_stride) return XK_ERR_ARG;
  T* aux = ws + (long)B * k * k;
  if (big_persist(B, k, (int)sizeof(T), algo)) {
    const int base = persist_base(k, (int)sizeof(T));
    for (int j = -1; j <= base - 2; ++j) {
      const int m2 = k - (j + 2);
      if (m2 <= 128) big_launch_step<T, 2>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
      else if (m2 <= 256) big_launch_step<T, 4>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
      else if (m2 <= 512) big_launch_step<T, 8>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
      else if (m2 <= 768) big_launch_step<T, 12>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
      else if (m2 <= 1024) big_launch_step<T, 16>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
      else big_launch_step<T, 24>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
    }
    int rc = persist_tridiag<T>(Tin, ws, aux, aux_stride, B, k, base, W, ldt, sT, st);
    if (rc != XK_OK) return rc;
    rc = big_launch_final_nt<T>(ws, aux, aux_stride, lam, Y, info, B, k, p, pb, uppest, lds, 2, st);
    if (rc != XK_OK) return rc;
    XK_LAUNCH_CHECK();
    return XK_OK;
  }
  if (big_two_stage(k, (int)sizeof(T), algo)) {
    if (ws_elems < (long)B * k * k + (long)B * aux_stride + (long)B * band_ws_elems(k)) return XK_ERR_ARG;
    T* bws = aux + (long)B * aux_stride;
    int rc = band_tridiag<T>(Tin, ws, aux, aux_stride, bws, B, k, ldt, sT, st);
    if (rc != XK_OK) return rc;
    rc = big_launch_final_nt<T>(ws, aux, aux_stride, lam, Y, info, B, k, p, pb, uppest, lds, 1, st);
    if (rc != XK_OK) return rc;
    XK_LAUNCH_CHECK();
    return band_back<T>(Y, bws, B, k, p, st);
  }
  for (int j = -1; j <= k - 3; ++j) {                     // (column slots of 64 that are still live: 2 / 4 / 8 / 12)
    const int m2 = k - (j + 2);
    if (m2 <= 128) big_launch_step<T, 2>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
    else if (m2 <= 256) big_launch_step<T, 4>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
    else if (m2 <= 512) big_launch_step<T, 8>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
    else if (m2 <= 768) big_launch_step<T, 12>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
    else if (m2 <= 1024) big_launch_step<T, 16>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
    else big_launch_step<T, 24>(Tin, ws, aux, aux_stride, B, k, j, W, ldt, sT, nt, st);
  }
  const int rc = big_launch_final_nt<T>(ws, aux, aux_stride, lam, Y, info, B, k, p, pb, uppest, lds, 0, st);
  if (rc != XK_OK) return rc;
  XK_LAUNCH_CHECK();
  return XK_OK;
}
}  // namespace xk

extern "C" {

/* the LU batch size (shifts factorised at a time) the kernel would use for order k, p pairs, or 0 when it does not
 * fit the 160 KiB of LDS at all (elem_size 8 / 4) */
int xk_small_eigh_big_batch(int k, int p, int elem_size) {
  // orders 769 .. 1024 (r05): the two-stage form where its band fits the LDS (fp32), else one launch per Householder step
  // with 16 column slots
  if (k < 8 || p < 1 || p > xk::BIG_MAXP || p > k || k > xk::BIG_MAXK) return 0;
  for (int pb = p; pb >= 1; --pb)
    if (xk::big_lds_elems(k, p, pb) * elem_size + 64 <= 160 * 1024) return pb;
  return 0;
}

long xk_small_eigh_big_workspace_elems(int B, int k, int wg) {
  const int W = xk::big_pick_w(B, k, wg);
  long e = (long)B * k * k + (long)B * k * (7 + 2 * W);
  if (xk::band_supported(k, 4)) e += (long)B * xk::band_ws_elems(k);       // (the two-stage form's blocks)
  return e;
}

#define XK_DEFINE_EIGH_BIG(SUF, T)                                                                            \
  int xk_small_eigh_big_##SUF(const T* Tin, T* lam, T* Y, T* ws, long ws_elems, int* info, int B, int k,      \
                              int p, int uppest, long ldt, long sT, int wg, int threads, int algo,            \
                              void* stream) {                                                                 \
    if (B < 0 || k < 8 || k > xk::BIG_MAXK || p < 1 || p > k || p > xk::BIG_MAXP) return XK_ERR_ARG;                  \
    if (wg < 0 || wg > 32 || (threads != 0 && threads != 256 && threads != 512)) return XK_ERR_ARG;           \
    if (algo < 0 || algo > 3) return XK_ERR_ARG;                                                              \
    if (B == 0) return XK_OK;                                                                                 \
    return xk::big_run<T>(Tin, lam, Y, ws, ws_elems, info, B, k, p, uppest, ldt, sT, wg,                      \
                          threads ? threads : 512, algo, (hipStream_t)stream);                                \
  }

XK_DEFINE_EIGH_BIG(f64, double)
XK_DEFINE_EIGH_BIG(f32, float)

}  // extern "C"
