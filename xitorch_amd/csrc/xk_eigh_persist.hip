// xitorch_amd :: K3p — the Householder tridiagonalisation of a Rayleigh–Ritz matrix in ONE launch of ONE workgroup per
// matrix, the trailing block resident in the REGISTERS of the workgroup's eight waves.
//
// Why: the reference calls torch.linalg.eigh on the whole projected matrix every iteration
// (xitorch/_impls/linalg/symeig.py:170-175) and never restarts (:132-135), so one small operator (BASELINE configs[0]:
// N = 512, benchmarks/benchmarks_solve.py:37-59) walks its basis up to 330 vectors: 55 dense eigenproblems of growing
// order.  K3g's one-launch-per-Householder-step form costs 5.7-8 us per step (a dependent kernel boundary + a trip to
// L2 each), its two-stage form 20 panels x 4 launches + a bulge chase; the LDS-resident K3t stops at order 128 and pays
// ~130 cycles of LDS latency several times per step.  The whole upper triangle of an fp64 matrix of order 256 is
// 41 k padded elements = 328 KB: it does not fit the LDS (160 KB) but it fits the 512 KB register file of one CU.
//
// Layout (all indices relative to `base`, the first row / column still alive when the kernel starts):
//   wave w (of 8) owns rows i = w + 8 u; lane l of column slot t holds column c = l + 64 t; only c >= i is stored, so row
//   u lives in slots t >= u >> 3 — a STATIC register index: A[g][r][t], u = 8 g + r, t >= g  (NT = 4: 80 values per lane).
// One step = the look-ahead step of xk_eigh_big.hip's tridiag_step_kernel with the kernel boundary replaced by two
// __syncthreads:
//   A. the four OLDER waves (one per SIMD; same data, same operations -> the same bits in each) fold the eight waves'
//      partial products of the previous step into w = A v_j, form K = tau/2 w.v and q = tau w - K v,
//   B. update row j + 1 (published by its owner in the previous step) and build reflector j + 1 from it; q and the
//      reflector go to LDS; barrier; the younger four waves fetch them (measured: this part is one dependent chain —
//      its four SIMD partners taken away, it takes as long: profiles/r06_k3_final_phases.json)
//   C. every wave, own rows i >= j + 2: rank-2 update in registers and, in the same pass, the product of the UPDATED row
//      with reflector j + 1 — column form into per-lane accumulators, row form through a transposing wave reduction of
//      eight rows at a time; the row's three scalars (v_i, q_i, v'_i) are LDS broadcasts; the owner of row j + 2 publishes
//      it for the next step
//   D. partial column sums to LDS (double-buffered), barrier.
// Orders beyond 64 NT columns: the first `base` steps are K3g step launches (matrix in global memory), the kernel takes
// over their hand-over blocks (row X, row-form sums R, column partials P[W]) when the trailing block fits.
// Outputs in the layout xk_eigh_big.hip's final kernel reads in mode 2: d, e, tau complete in the aux block, reflector r
// parked in row r of the work copy S (columns > r + 1).
#include "xk_common.h"
#include <type_traits>

namespace xk {

__device__ __forceinline__ double per_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float per_readlane(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ double per_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float per_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  r = fmaf(fmaf(-x, r, 1.0f), r, r);
  return r;
}
// element r (relative index, wave-uniform) of a vector distributed as slot[t] of lane l <-> element l + 64 t
template <typename T, int NT>
__device__ __forceinline__ T per_get(const T (&v)[NT], int r) {
  const int t = r >> 6, l = r & 63;
  T out = T(0);
#pragma unroll
  for (int u = 0; u < NT; ++u)
    if (u == t) out = per_readlane(v[u], l);
  return out;
}
// Householder reflector of x = (alpha, rest), sigma = |rest|^2: (I - tau v v^T) x = beta e1, v = (1, rest * scale)
template <typename T>
__device__ __forceinline__ void per_house(T alpha, T sigma, T& tau, T& beta, T& scale) {
  tau = T(0); beta = alpha; scale = T(0);
  if (!(sigma == T(0))) {                                   // (a NaN row must poison the result, not be skipped)
    const T nrm = sqrt(alpha * alpha + sigma);
    beta = alpha >= T(0) ? -nrm : nrm;
    tau = (beta - alpha) * per_rcp(beta);
    scale = per_rcp(alpha - beta);
  }
}

constexpr int PER_NW = 8;                                   // waves per workgroup = row stride of a wave

template <typename T, int NT>
__global__ __launch_bounds__(512) void tridiag_persist_kernel(
    const T* __restrict__ Tin, T* __restrict__ Sws, T* __restrict__ aux, long aux_stride, int n, int base, int W,
    long ldt, long sT) {
  constexpr int NW = PER_NW, NC = 64 * NT;
  __shared__ T Srow[2][NC];                                 // row j + 1 as its owner left it (relative columns)
  __shared__ T Rl[2][NC];                                   // row-form sums of the product with the next reflector
  __shared__ T Pl[2][NW][NC];                               // column-form partial sums per wave
  __shared__ T Ql[NC];                                      // q of the step, by relative column (row scalars are read from here)
  __shared__ T VNl[2][NC];                                  // reflector j (cur) and j + 1 (nxt), by relative column
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* Tb = Tin + (long)b * sT;
  T* S = Sws + (long)b * n * n;
  T* ab = aux + (long)b * aux_stride;
  T* dd = ab; T* ee = ab + n; T* tau = ab + 2 * n;
  const bool first = base == 0;
  const int jstart = base - 1;

  // ---- the wave's rows into registers: upper triangle, from eigh's lower triangle of T (first) or from the work copy S
  T A[NT][8][NT];
#pragma unroll
  for (int g = 0; g < NT; ++g)
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        T val = T(0);
        if (t >= g) {
          const int i = base + wave + 8 * (8 * g + r), c = base + lane + 64 * t;
          if (i < n && c < n && c >= i) val = first ? Tb[(long)c * ldt + i] : S[(long)i * n + c];
        }
        A[g][r][t] = val;
      }

  T v[NT], vN[NT];                                          // reflector j / j + 1 by relative column
  T tj = T(0);
#pragma unroll
  for (int t = 0; t < NT; ++t) { v[t] = T(0); vN[t] = T(0); }
  if (first) {
    for (int idx = tid; idx < NC; idx += 512) Srow[0][idx] = idx < n ? Tb[(long)idx * ldt] : T(0);
  } else {
    // the hand-over blocks of the last step launch j = jstart - 1 (xk_eigh_big.hip): row X, sums R, partials P[W]
    const int j = jstart, j1 = base;
    const T* Xcur = ab + 3 * n + (long)(j & 1) * n;
    const T* Rcur = ab + 5 * n + (long)(j & 1) * n;
    const T* Pcur = ab + 7 * n + (long)(j & 1) * W * n;
    for (int idx = tid; idx < NC; idx += 512) {
      const int c = base + idx;
      T rsum = T(0), psum = T(0), srow = T(0);
      if (c < n) {
        rsum = Rcur[c];
        for (int g = 0; g < W; ++g) psum += Pcur[(long)g * n + c];
        srow = S[(long)j1 * n + c];
      }
      Rl[0][idx] = rsum;
      Pl[0][0][idx] = psum;
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) Pl[0][w2][idx] = T(0);
      Srow[0][idx] = srow;
    }
    // reflector j from row j (columns >= j + 1), every wave the same numbers
    T x[NT];
    T ss = T(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = base + lane + 64 * t;
      x[t] = c < n ? Xcur[c] : T(0);
      if (!(t == 0 && lane == 0)) ss += x[t] * x[t];
    }
    const T sigma = wave_sum_dpp(ss);
    const T alpha = per_readlane(x[0], 0);
    T beta, scale;
    per_house(alpha, sigma, tj, beta, scale);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int c = base + lane + 64 * t;
      v[t] = (t == 0 && lane == 0) ? T(1) : x[t] * scale;
      if (wave == 0 && c < n && c > j1) S[(long)j * n + c] = v[t];
    }
    if (tid == 0) { tau[j] = tj; ee[j] = beta; }
  }
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) VNl[0][lane + 64 * t] = v[t];
  }
  __syncthreads();

  int cur = 0;
#ifdef XK_PERSIST_DBG
  long long tA = 0, tB = 0, tC = 0, tD = 0, t0 = __builtin_readcyclecounter(), tstart = t0;
#define XK_PSTAMP(acc) { const long long t1_ = __builtin_readcyclecounter(); acc += t1_ - t0; t0 = t1_; }
#else
#define XK_PSTAMP(acc)
#endif
  for (int j = jstart; j <= n - 3; ++j) {
    const int nxt = cur ^ 1;
    const int r1 = j + 1 - base, r2 = j + 2 - base;         // relative indices of row j + 1 and of the first live row
    const int nrel = n - base;
    // ---- A. w = A v_j from the partial sums, K, q -----------------------------------------------------------------
    // (A and B on the four OLDER waves only, one per SIMD — the same numbers in each; the younger four would only compete
    //  for the same vector ALUs: they wait at the barrier and fetch q and the new reflector from LDS)
    T q[NT];
    T q0 = T(0);
#pragma unroll
    for (int t = 0; t < NT; ++t) q[t] = T(0);
    T tauN = T(0), betaN = T(0);
    if (wave < 4) {
    if (j >= 0) {
      T wsum[NT];
      T dot = T(0);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        wsum[t] = T(0);
        if (64 * (t + 1) > r1) {                            // (slots left of row j + 1 are dead: uniform)
          const int rc = lane + 64 * t;
          T pv[NW];
          T s = Rl[cur][rc];
#pragma unroll
          for (int w2 = 0; w2 < NW; ++w2) pv[w2] = Pl[cur][w2][rc];
#pragma unroll
          for (int w2 = 0; w2 < NW; ++w2) s += pv[w2];
          wsum[t] = (rc >= r1 && rc < nrel) ? s : T(0);
          dot += wsum[t] * v[t];
        }
      }
      const T K = T(0.5) * tj * tj * wave_sum_dpp(dot);     // w = tj A v;  K = tj/2 w.v
#pragma unroll
      for (int t = 0; t < NT; ++t) q[t] = tj * wsum[t] - K * v[t];
      q0 = per_get<T, NT>(q, r1);                           // v_j(j + 1) = 1
    }
    XK_PSTAMP(tA)
    // ---- B. row j + 1 after update j, reflector j + 1 ----------------------------------------------------------------
    {
      T s1[NT], a[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int rc = lane + 64 * t;
        s1[t] = (64 * (t + 1) > r1 && rc >= r1) ? Srow[cur][rc] : T(0);
      }
      const T sdiag = per_get<T, NT>(s1, r1);
      T ss = T(0);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int rc = lane + 64 * t;
        a[t] = (rc >= r2 && rc < nrel) ? s1[t] - (q[t] + q0 * v[t]) : T(0);
        ss += rc > r2 ? a[t] * a[t] : T(0);
      }
      const T dnext = sdiag - T(2) * q0;
      const T sigma = wave_sum_dpp(ss);
      const T alpha = per_get<T, NT>(a, r2);
      T scaleN;
      per_house(alpha, sigma, tauN, betaN, scaleN);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int rc = lane + 64 * t;
        vN[t] = rc == r2 ? T(1) : (rc > r2 ? a[t] * scaleN : T(0));
        if (wave == 0 && rc > r2 && rc < nrel) S[(long)(j + 1) * n + base + rc] = vN[t];     // parked for the back-transformation
      }
      if (tid == 0) { dd[j + 1] = dnext; ee[j + 1] = betaN; tau[j + 1] = tauN; }
      if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (64 * (t + 1) > r1) { Ql[lane + 64 * t] = q[t]; VNl[nxt][lane + 64 * t] = vN[t]; }
        }
      }
    }
    }
    __syncthreads();
    if (wave >= 4) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (64 * (t + 1) > r1) { q[t] = Ql[lane + 64 * t]; vN[t] = VNl[nxt][lane + 64 * t]; }
      }
    }
    XK_PSTAMP(tB)
    // ---- C. own rows i >= j + 2: update, product with reflector j + 1 ---------------------------------------------------
    T acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = T(0);
#pragma unroll
    for (int g = 0; g < NT; ++g) {
      const int ib = wave + 64 * g;                         // relative rows ib + 8 r of this group
      // rows ib + 8 r >= r2 are alive: r >= rmin, entered through ONE scalar jump per group (rows past the matrix hold
      // zeros and meet zero reflector entries: they cost instructions in the last group only and change nothing)
      const int rmin = r2 > ib ? (r2 - ib + 7) >> 3 : 0;
      if (rmin > 7 || ib >= nrel) continue;                 // (whole group dead or beyond the matrix: uniform)
      T rs[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) rs[r] = T(0);
      auto row = [&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int ln = wave + 8 * r;                        // lane of column i inside slot g
        // (the row's three scalars as LDS broadcasts: six v_readlane less on the vector ALU, which is what bounds the loop)
        const T vi = VNl[cur][ib + 8 * r], qi = Ql[ib + 8 * r], vNi = VNl[nxt][ib + 8 * r];
        T racc = T(0);
        if (lane >= ln) {                                   // the diagonal slot: columns left of the diagonal are not stored
          T an = fma(-vi, q[g], A[g][r][g]);
          an = fma(-qi, v[g], an);
          A[g][r][g] = an;
          acc[g] = fma(an, vNi, acc[g]);                    // column form: (A v)(c) += a(i, c) v(i), c >= i
          if (lane > ln) racc = an * vN[g];                 // row form:    (A v)(i) += a(i, c) v(c), c > i
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t > g) {
            T an = fma(-vi, q[t], A[g][r][t]);
            an = fma(-qi, v[t], an);
            A[g][r][t] = an;
            acc[t] = fma(an, vNi, acc[t]);
            racc = fma(an, vN[t], racc);
          }
        }
        rs[r] = racc;
        if (ib + 8 * r == r2) {                             // the next step's row j + 1
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (t >= g) Srow[nxt][lane + 64 * t] = A[g][r][t];
        }
      };
      switch (rmin) {
        case 0: row(std::integral_constant<int, 0>{}); [[fallthrough]];
        case 1: row(std::integral_constant<int, 1>{}); [[fallthrough]];
        case 2: row(std::integral_constant<int, 2>{}); [[fallthrough]];
        case 3: row(std::integral_constant<int, 3>{}); [[fallthrough]];
        case 4: row(std::integral_constant<int, 4>{}); [[fallthrough]];
        case 5: row(std::integral_constant<int, 5>{}); [[fallthrough]];
        case 6: row(std::integral_constant<int, 6>{}); [[fallthrough]];
        default: row(std::integral_constant<int, 7>{});
      }
      wave_reduce_scatter<T, 8>(rs, lane);
      if (wave_rs_is_writer<8>(lane)) Rl[nxt][ib + 8 * wave_rs_orig_index<8>(0, lane)] = rs[0];
    }
    XK_PSTAMP(tC)
    // ---- D. partial column sums, one barrier per step ------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (64 * (t + 1) > r2) Pl[nxt][wave][lane + 64 * t] = acc[t];
    __syncthreads();
    XK_PSTAMP(tD)
#pragma unroll
    for (int t = 0; t < NT; ++t) v[t] = vN[t];
    tj = tauN;
    cur = nxt;
  }
#ifdef XK_PERSIST_DBG
  if (b == 0 && (tid == 0 || tid == 448))
    printf("K3p n=%d NT=%d wave=%d cycles: A %lld B %lld C %lld D %lld total %lld (setup %lld)\n", n, NT, wave, tA, tB, tC, tD,
           (long long)__builtin_readcyclecounter() - tstart, tstart);
#endif
  // ---- the last diagonal element sits in the registers of the owner of row n - 1 ------------------------------------------
  {
    const int rel = n - 1 - base;
    if (wave == (rel & 7)) {
      const int u = rel >> 3;
#pragma unroll
      for (int g = 0; g < NT; ++g)
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (8 * g + r == u && lane == (rel & 63)) dd[n - 1] = A[g][r][g];
    }
    if (tid == 0) { ee[n - 1] = T(0); tau[n - 1] = T(0); }
  }
}

// largest trailing order the kernel holds in registers (fp64: 4 column slots = 80 values per lane; fp32: 6 = 168)
int persist_max_order(int elem_size) { return elem_size == 8 ? 256 : 384; }

template <typename T, int NT>
static void persist_launch_nt(const T* Tin, T* S, T* aux, long aux_stride, int B, int k, int base, int W, long ldt,
                              long sT, hipStream_t st) {
  hipLaunchKernelGGL((tridiag_persist_kernel<T, NT>), dim3(B), dim3(512), 0, st, Tin, S, aux, aux_stride, k, base, W,
                     ldt, sT);
}

// the first row / column the kernel takes over at order k (0: the whole reduction in the one launch)
int persist_base(int k, int elem_size) {
  const int m = persist_max_order(elem_size);
  return k > m ? k - m : 0;
}

template <typename T>
int persist_tridiag(const T* Tin, T* S, T* aux, long aux_stride, int B, int k, int base, int W, long ldt, long sT,
                    hipStream_t st) {
  const int m = k - base;
  if (m > persist_max_order((int)sizeof(T)) || m < 3) return XK_ERR_UNSUPPORTED;
  if constexpr (sizeof(T) == 8) {
    if (m <= 128) persist_launch_nt<T, 2>(Tin, S, aux, aux_stride, B, k, base, W, ldt, sT, st);
    else if (m <= 192) persist_launch_nt<T, 3>(Tin, S, aux, aux_stride, B, k, base, W, ldt, sT, st);
    else persist_launch_nt<T, 4>(Tin, S, aux, aux_stride, B, k, base, W, ldt, sT, st);
  } else {
    if (m <= 128) persist_launch_nt<T, 2>(Tin, S, aux, aux_stride, B, k, base, W, ldt, sT, st);
    else if (m <= 256) persist_launch_nt<T, 4>(Tin, S, aux, aux_stride, B, k, base, W, ldt, sT, st);
    else persist_launch_nt<T, 6>(Tin, S, aux, aux_stride, B, k, base, W, ldt, sT, st);
  }
  XK_LAUNCH_CHECK();
  return XK_OK;
}

template int persist_tridiag<double>(const double*, double*, double*, long, int, int, int, int, long, long, hipStream_t);
template int persist_tridiag<float>(const float*, float*, float*, long, int, int, int, int, long, long, hipStream_t);

}  // namespace xk
