// xitorch_amd :: K3g, two-stage form — the tridiagonalisation of Rayleigh–Ritz matrices of order ~200 .. 1024 WITHOUT one
// launch per Householder step.
//
// The un-restarted Davidson iteration of the reference (xitorch/_impls/linalg/symeig.py:132-135, :174-175: eigh of the
// whole T every iteration) reaches bases of 300-800 vectors; xk_eigh_big.hip's one-stage reduction needs k - 1 launches,
// each a read + write of the trailing block (8.4 of 11.0 ms at order 582, 32 matrices).  Here (LAPACK's dsytrd_sy2sb /
// dsytrd_sb2st route, written for one chip):
//   1. dense -> band of NB = 16 sub-diagonals, one PANEL of 16 columns at a time, four launches per panel:
//        band_qr_kernel      Householder QR of the panel in LDS, one workgroup per matrix: V, T, R
//        band_w_kernel       W = A22 V for the workgroup's 64 rows and its piece of G = V^T W, streamed
//        band_z_kernel       Z = W T - 1/2 V (T^T G T)
//        band_update_kernel  A22 -= V Z^T + Z V^T on 64 x 64 tiles, streamed (V_c, Z_c by scalar loads)
//      (the trailing block is read twice and written once per 16 columns instead of once each per column)
//   2. band -> tridiagonal by bulge chasing, ONE workgroup per matrix, the band (2 NB diagonals with the bulges) in LDS:
//      sweep s removes column s; its step t works on rows s + 1 + 16 t .. s + 16 (t + 2), so sweep s + 1 may run three
//      steps behind sweep s: wave w owns sweeps w, w + 16, ... and waits on an LDS counter of the sweep before its own
//      (band_chase_kernel; the index arithmetic is pinned by scripts/two_stage_proto.py)
//   3. the wanted eigenpairs of the tridiagonal matrix: xk_eigh_big.hip's final kernel (bisection, inverse iteration,
//      checks), which leaves the vectors of (d, e) in the rows of Y
//   4. back through both stages (band_back_kernel, one workgroup per vector): the chase reflectors sweep by sweep in
//      reverse (the steps of a sweep touch disjoint rows), then the panels' block reflectors I - V T V^T in reverse.
// Nothing waits ACROSS workgroups: kernel boundaries are the only hand-offs between them (safe beside CU-masked streams);
// the waves of the chase kernel's one workgroup per matrix wait on each other through LDS counters.
#include "xk_common.h"

namespace xk {

constexpr int BAND_NB = 16;
constexpr int BAND_LP = BAND_NB + 1;             // LDS pitch of 16-column row blocks (conflict-free column walks)
constexpr int BAND_LD = 2 * BAND_NB;             // diagonals kept per column of the band in the chase
constexpr int BAND_STRIP = 64;                   // rows / columns of the trailing block per workgroup

__device__ __forceinline__ double band_rl(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float band_rl(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ double band_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float band_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  r = fmaf(fmaf(-x, r, 1.0f), r, r);
  return r;
}
// sum over each 16-lane row, every lane of the row gets the (bit-identical) total; DPP only
template <typename T>
__device__ __forceinline__ T row16_sum(T v) {
  v += lane_partner<8>(v);
  v += lane_partner<4>(v);
  v += lane_partner<2>(v);
  v += lane_partner<1>(v);
  return v;
}
// Householder reflector of x = (alpha, rest), sigma = |rest|^2: (I - tau v v^T) x = beta e1, v = (1, rest * scale)
template <typename T>
__device__ __forceinline__ void band_house(T alpha, T sigma, T& tau, T& beta, T& scale) {
  tau = T(0); beta = alpha; scale = T(0);
  if (!(sigma == T(0))) {                                   // (a NaN must poison the result, not be skipped)
    const T nrm = sqrt(alpha * alpha + sigma);
    beta = alpha >= T(0) ? -nrm : nrm;
    tau = (beta - alpha) * band_rcp(beta);
    scale = band_rcp(alpha - beta);
  }
}

// number of panels of order n: panel j eliminates below the band in columns 16 j .. 16 j + 15 while >= 2 rows lie below
__host__ __device__ inline int band_npanels(int n) {
  int np = 0;
  while (n - (np + 1) * BAND_NB >= 2) ++np;
  return np;
}
// steps of chase sweep s (window t starts at row s + 1 + 16 t and needs two rows)
__host__ __device__ inline int band_nsteps(int n, int s) { return s <= n - 3 ? (n - 3 - s) / BAND_NB + 1 : 0; }

// ---- 0. work copy: S = the symmetric matrix in FULL storage, from the lower triangle of T (eigh's UPLO = 'L') ----------
template <typename T>
__global__ __launch_bounds__(256) void band_copy_kernel(const T* __restrict__ Tin, T* __restrict__ S, int n, long ldt,
                                                        long sT, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long nn = (long)n * n;
  const long b = idx / nn;
  const long rem = idx - b * nn;
  const int i = (int)(rem / n), c = (int)(rem - (long)i * n);
  const int hi = i > c ? i : c, lo = i > c ? c : i;
  S[idx] = Tin[b * sT + (long)hi * ldt + lo];
}

// ---- 1a. panel j: Householder QR of the 16 columns below the band, ONE workgroup per matrix ------------------------------
// Vg[b][j][row][16] (absolute rows, explicit unit diagonal / zeros), Tg[b][j][16][16], Rg[b][j][16][16] (rows of R: the
// band entries A[r0 + i][c0 + c], i <= c).
// Everything here is latency of one workgroup (a lone wave issues an instruction every 5-8 cycles; a barrier costs a few
// hundred): one pass and one barrier pair per column — it applies reflector c and gathers, for column c + 1, the norm,
// v^T (columns to the right) and v^T v_a (columns to the left, for T) together; the T factor from those dots after the
// loop, by one wave, beside the pass that turns the panel into V.
template <typename T>
__global__ __launch_bounds__(1024) void band_qr_kernel(const T* __restrict__ Sws, T* __restrict__ Vg,
                                                       T* __restrict__ Tg, T* __restrict__ Rg, int n, int j, int np) {
  constexpr int NB = BAND_NB, LP = BAND_LP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int c0 = j * NB, r0 = c0 + NB, m = n - r0;
  const int nref = NB < m - 1 ? NB : m - 1;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int RG = 64;                              // row groups (1024 threads)
  T* Pl = reinterpret_cast<T*>(smem);                 // m x LP: the panel, then V
  T* red = Pl + (long)m * LP;                         // RG x NB partial dots
  T* scal = red + RG * NB;                            // NB: v_c = x_c * scal[c] below the diagonal
  T* tw = scal + NB;                                  // NB: tau * (v^T column)
  T* tauL = tw + NB;                                  // NB
  T* Tl = tauL + NB;                                  // NB x NB: v_a^T v_c above the diagonal, then T
  T* Rl = Tl + NB * NB;                               // NB x NB
  const int b = blockIdx.x;
  const T* S = Sws + (long)b * n * n;

  for (int idx = tid; idx < m * NB; idx += nt) {
    const int i = idx / NB, cc = idx - i * NB;
    Pl[i * LP + cc] = S[(long)(r0 + i) * n + c0 + cc];
  }
  for (int idx = tid; idx < NB * NB; idx += nt) { Tl[idx] = T(0); Rl[idx] = T(0); }
  if (tid < NB) { scal[tid] = T(0); tauL[tid] = T(0); }
  __syncthreads();

  const int cc = tid % NB, rg = tid / NB;
  // dots of column c (rows below its diagonal) with all 16 columns in one pass: the norm (cc == c), v^T column
  // (cc > c) and v_a^T v_c of the T factor (cc < c).  For column 0 here; for column c + 1 in the same pass that applies
  // reflector c (every thread repeats the update of column c + 1 for its rows: a read of Pl[i][c + 1] and the write
  // of the thread that owns that column are the same two instructions of ONE wave — rows are dealt by tid / 16 — so the
  // read comes first): two barriers per column, not three
  if (nref > 0) {
    T p0 = T(0), p1 = T(0);
    int i = 1 + rg;
    for (; i + RG < m; i += 2 * RG) {
      const T a0 = Pl[i * LP], b0 = Pl[i * LP + cc];
      const T a1 = Pl[(i + RG) * LP], b1 = Pl[(i + RG) * LP + cc];
      p0 += a0 * b0;
      p1 += a1 * b1;
    }
    if (i < m) p0 += Pl[i * LP] * Pl[i * LP + cc];
    red[rg * NB + cc] = p0 + p1;
  }
  for (int c = 0; c < nref; ++c) {
    __syncthreads();
    if (wave == 0) {
      const int c2 = lane & (NB - 1), qt = lane >> 4;
      T rv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) rv[u] = red[(qt * 16 + u) * NB + c2];   // 16 row groups per quarter
      T sacc = T(0);
#pragma unroll
      for (int u = 0; u < 16; u += 4) sacc += (rv[u] + rv[u + 1]) + (rv[u + 2] + rv[u + 3]);
      sacc = swap_add32(sacc, sacc);
      const T dot = swap_add16(sacc, sacc);             // every lane: the dot for its column c2
      const T sigma = band_rl(dot, c);
      const T prow = Pl[c * LP + c2];                   // row c of the panel
      const T alpha = band_rl(prow, c);
      T tau, beta, scale;
      band_house(alpha, sigma, tau, beta, scale);
      if (lane < NB) {
        if (lane > c) tw[lane] = tau * (prow + scale * dot);
        if (lane < c) Tl[lane * NB + c] = (prow + scale * dot) * scal[lane];   // v_a^T v_c (v_a(c) = prow * scal[a])
        if (lane == c) { scal[c] = scale; tauL[c] = tau; Rl[c * NB + c] = beta; }
      }
    }
    __syncthreads();
    {
      const bool nextc = c + 1 < nref;
      const int cn = c + 1 < NB ? c + 1 : NB - 1;
      const bool mine = cc > c;
      const T twc = mine ? tw[cc] : T(0);
      const T twn = nextc ? tw[cn] : T(0);
      const T sc = scal[c];
      T p0 = T(0), p1 = T(0);
      int i = c + rg;
      for (; i + RG < m; i += 2 * RG) {
        const T a0 = Pl[i * LP + c], a1 = Pl[(i + RG) * LP + c];
        const T b0 = Pl[i * LP + cc], b1 = Pl[(i + RG) * LP + cc];
        const T d0 = Pl[i * LP + cn], d1 = Pl[(i + RG) * LP + cn];
        const T v0 = i == c ? T(1) : a0 * sc, v1 = a1 * sc;
        const T nb0 = b0 - v0 * twc, nb1 = b1 - v1 * twc;     // (twc = 0 for the columns that stay)
        const T nd0 = d0 - v0 * twn, nd1 = d1 - v1 * twn;
        if (mine) { Pl[i * LP + cc] = nb0; Pl[(i + RG) * LP + cc] = nb1; }
        if (i > c + 1) p0 += nd0 * nb0;
        p1 += nd1 * nb1;                                      // i + RG > c + 1 always
      }
      if (i < m) {
        const T a0 = Pl[i * LP + c], b0 = Pl[i * LP + cc], d0 = Pl[i * LP + cn];
        const T v0 = i == c ? T(1) : a0 * sc;
        const T nb0 = b0 - v0 * twc, nd0 = d0 - v0 * twn;
        if (mine) Pl[i * LP + cc] = nb0;
        if (i > c + 1) p0 += nd0 * nb0;
      }
      if (nextc) red[rg * NB + cc] = p0 + p1;
    }
  }
  __syncthreads();
  // R out, V out: unit diagonal, zeros above, scaled below (columns without a reflector: zero); wave 0 meanwhile turns the
  // dots v_a^T v_c into T: column c = -tau_c T[0:c, 0:c] (V^T v_c), lane a keeps row a of T in registers
  T* Vb = Vg + ((long)b * np + j) * n * NB;
  if (wave == 0) {
    T trow[NB];
#pragma unroll
    for (int a2 = 0; a2 < NB; ++a2) trow[a2] = T(0);
    const int a = lane & (NB - 1);
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      const T tc = tauL[c];
      const T zc = a < c ? Tl[a * NB + c] : T(0);
      T t = T(0);
#pragma unroll
      for (int a2 = 0; a2 < NB; ++a2) {
        if (a2 < c) {
          const T za = band_rl(zc, a2);
          t += trow[a2] * za;                               // trow[a2] = T[a][a2], zero for a2 < a
        }
      }
      trow[c] = a == c ? tc : (a < c ? -tc * t : T(0));
    }
    if (lane < NB) {
      T* Tb = Tg + ((long)b * np + j) * NB * NB;
#pragma unroll
      for (int c = 0; c < NB; ++c) Tb[lane * NB + c] = trow[c];
    }
  } else {
    const int nt2 = nt - 64, tid2 = tid - 64;
    const int cc2 = tid2 % NB, rg2 = tid2 / NB, RG2 = nt2 / NB;
    const T sc = scal[cc2];
    const bool has = cc2 < nref;
    for (int i = rg2; i < m; i += RG2) {
      const T val = Pl[i * LP + cc2];
      T v;
      if (i < cc2) { Rl[i * NB + cc2] = val; v = T(0); }
      else if (i == cc2) { if (!has) Rl[cc2 * NB + cc2] = val; v = has ? T(1) : T(0); }
      else v = has ? val * sc : T(0);
      Vb[(long)(r0 + i) * NB + cc2] = v;
    }
  }
  __syncthreads();
  T* Rb = Rg + ((long)b * np + j) * NB * NB;
  for (int idx = tid; idx < NB * NB; idx += nt) Rb[idx] = Rl[idx];
}

// ---- 1a'. panel j: W = A22 V for the 64 rows of strip q and its piece of G = V^T W: a stream -------------------------------
// lane = row i of the strip (A22 symmetric: the walk goes down column r0 + i), a wave takes a slice of the rows c, V_c is
// wave-uniform (scalar loads); the slices' partial sums meet in LDS
template <typename T>
__global__ __launch_bounds__(512) void band_w_kernel(const T* __restrict__ Sws, const T* __restrict__ Vg,
                                                     T* __restrict__ Wg, T* __restrict__ Gp, int n, int j, int np,
                                                     int gs) {
  constexpr int NB = BAND_NB, LP = BAND_LP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Wp = reinterpret_cast<T*>(smem);                 // 8 x 64 x LP
  const int r0 = (j + 1) * NB, m = n - r0;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, nw = nt >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = blockIdx.x, b = blockIdx.y;
  const T* S = Sws + (long)b * n * n;
  const T* Vb = Vg + ((long)b * np + j) * n * NB + (long)r0 * NB;
  const int i = q * BAND_STRIP + lane;
  const bool iv = i < m;
  T acc[NB];
#pragma unroll
  for (int a = 0; a < NB; ++a) acc[a] = T(0);
  {
    const int cw = (m + nw - 1) / nw;
    const int cb = __builtin_amdgcn_readfirstlane(wave * cw);
    const int ce = __builtin_amdgcn_readfirstlane(cb + cw < m ? cb + cw : m);
    const T* Sc = S + (long)r0 * n + r0 + (iv ? i : 0);
    constexpr int U = 8;
    int c = cb;
    for (; c + U <= ce; c += U) {
      T av[U];
#pragma unroll
      for (int u = 0; u < U; ++u) av[u] = Sc[(long)(c + u) * n];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const T* vc = Vb + (long)(c + u) * NB;
#pragma unroll
        for (int a = 0; a < NB; ++a) acc[a] += av[u] * vc[a];
      }
    }
    for (; c < ce; ++c) {
      const T av = Sc[(long)c * n];
      const T* vc = Vb + (long)c * NB;
#pragma unroll
      for (int a = 0; a < NB; ++a) acc[a] += av * vc[a];
    }
  }
#pragma unroll
  for (int a = 0; a < NB; ++a) Wp[(wave * BAND_STRIP + lane) * LP + a] = iv ? acc[a] : T(0);
  __syncthreads();
  for (int idx = tid; idx < BAND_STRIP * NB; idx += nt) {     // slices added in fixed order: bit-reproducible
    const int ii = idx / NB, c2 = idx - ii * NB;
    T sacc = Wp[ii * LP + c2];
    for (int w2 = 1; w2 < nw; ++w2) sacc += Wp[(w2 * BAND_STRIP + ii) * LP + c2];
    Wp[ii * LP + c2] = sacc;
    const int row = q * BAND_STRIP + ii;
    if (row < m) Wg[(long)b * n * NB + (long)(r0 + row) * NB + c2] = sacc;
  }
  __syncthreads();
  if (tid < NB * NB) {
    const int a = tid / NB, c2 = tid - a * NB;
    T g0 = T(0), g1 = T(0);
    const int rmax = m - q * BAND_STRIP < BAND_STRIP ? m - q * BAND_STRIP : BAND_STRIP;
    const T* Vs = Vb + (long)q * BAND_STRIP * NB;
    int ii = 0;
    for (; ii + 2 <= rmax; ii += 2) {
      g0 += Vs[(long)ii * NB + a] * Wp[ii * LP + c2];
      g1 += Vs[(long)(ii + 1) * NB + a] * Wp[(ii + 1) * LP + c2];
    }
    if (ii < rmax) g0 += Vs[(long)ii * NB + a] * Wp[ii * LP + c2];
    Gp[((long)b * gs + q) * NB * NB + tid] = g0 + g1;
  }
}

// ---- 1b. panel j: Z = W T - V (1/2 T^T G T) for the rows of strip q ------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void band_z_kernel(const T* __restrict__ Vg, const T* __restrict__ Tg,
                                                     const T* __restrict__ Wg, const T* __restrict__ Gp,
                                                     T* __restrict__ Zg, int n, int j, int np, int gs, int nstrips) {
  constexpr int NB = BAND_NB;
  __shared__ T Gl[NB * NB], Tl[NB * NB], tmp[NB * NB], M2[NB * NB];
  const int r0 = (j + 1) * NB, m = n - r0;
  const int tid = threadIdx.x;
  const int q = blockIdx.x, b = blockIdx.y;
  const T* Vb = Vg + ((long)b * np + j) * n * NB;
  const T* Wb = Wg + (long)b * n * NB;
  T* Zb = Zg + (long)b * n * NB;
  // this thread's row of W and V first: the loads fly while G and T are folded
  const int row = q * BAND_STRIP + (tid >> 2), cq = (tid & 3) * 4;
  const bool rv = row < m;
  T wr[NB], vr[NB];
#pragma unroll
  for (int a = 0; a < NB; ++a) {
    wr[a] = rv ? Wb[(long)(r0 + row) * NB + a] : T(0);
    vr[a] = rv ? Vb[(long)(r0 + row) * NB + a] : T(0);
  }
  {
    T g = T(0);
    for (int s = 0; s < nstrips; ++s) g += Gp[((long)b * gs + s) * NB * NB + tid];
    Gl[tid] = g;
    Tl[tid] = Tg[((long)b * np + j) * NB * NB + tid];
  }
  __syncthreads();
  {
    const int a = tid / NB, c = tid - a * NB;
    T t = T(0);
#pragma unroll
    for (int k = 0; k < NB; ++k) t += Gl[a * NB + k] * Tl[k * NB + c];
    tmp[tid] = t;
  }
  __syncthreads();
  {
    const int a = tid / NB, c = tid - a * NB;
    T t = T(0);
#pragma unroll
    for (int k = 0; k < NB; ++k) t += Tl[k * NB + a] * tmp[k * NB + c];
    M2[tid] = T(0.5) * t;                                   // 1/2 T^T G T
  }
  __syncthreads();
  if (rv) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      T z = T(0);
#pragma unroll
      for (int a = 0; a < NB; ++a) z += wr[a] * Tl[a * NB + cq + u] - vr[a] * M2[a * NB + cq + u];
      Zb[(long)(r0 + row) * NB + cq + u] = z;
    }
  }
}

// ---- 1c. panel j: A22 -= V Z^T + Z V^T on a tile of the trailing block: columns of strip q, 64 * groups rows -------------
// lane = column i of the strip (its V_i, Z_i in registers), each wave `groups` times eight rows c, the eight loads of a
// trip in flight at once; V_c, Z_c are wave-uniform (scalar loads): no LDS, no barriers
template <typename T>
__global__ __launch_bounds__(512) void band_update_kernel(T* __restrict__ Sws, const T* __restrict__ Vg,
                                                          const T* __restrict__ Zg, int n, int j, int np, int groups) {
  constexpr int NB = BAND_NB;
  const int r0 = (j + 1) * NB, m = n - r0;
  const int tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = blockIdx.x, pch = blockIdx.y, b = blockIdx.z;
  T* S = Sws + (long)b * n * n;
  const T* Vb = Vg + ((long)b * np + j) * n * NB + (long)r0 * NB;
  const T* Zb = Zg + (long)b * n * NB + (long)r0 * NB;
  const int i = q * BAND_STRIP + lane;
  const bool iv = i < m;
  constexpr int U = 8;
  T* Sc = S + (long)r0 * n + r0 + (iv ? i : 0);
  int cb = __builtin_amdgcn_readfirstlane((pch * groups * nw + wave) * U);      // trips of a wave: nw * U rows apart
  T av[U];
#pragma unroll
  for (int u = 0; u < U; ++u) av[u] = (iv && cb + u < m) ? Sc[(long)(cb + u) * n] : T(0);
  T vi[NB], zi[NB];
#pragma unroll
  for (int a = 0; a < NB; ++a) {
    vi[a] = iv ? Vb[(long)i * NB + a] : T(0);
    zi[a] = iv ? Zb[(long)i * NB + a] : T(0);
  }
  for (int gidx = 0; gidx < groups; ++gidx) {
    const int cn = cb + nw * U;
    T an[U];
#pragma unroll
    for (int u = 0; u < U; ++u) an[u] = (iv && gidx + 1 < groups && cn + u < m) ? Sc[(long)(cn + u) * n] : T(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = cb + u;
      if (c < m) {                                          // (uniform)
        const T* vc = Vb + (long)c * NB;
        const T* zc = Zb + (long)c * NB;
        T s = T(0);
#pragma unroll
        for (int a = 0; a < NB; ++a) s += vi[a] * zc[a] + zi[a] * vc[a];
        if (iv) Sc[(long)c * n] = av[u] - s;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) av[u] = an[u];
    cb = cn;
  }
}

// ---- 2. band -> tridiagonal: one workgroup per matrix, the band in LDS, sweeps pipelined three steps apart ---------------
// Bd[c * LD + d] = A[c + d][c], d = 0 .. 2 NB - 1.  Reflector (s, t): Cv[(s * TS + t) * NB + i], Ct[s * TS + t].
// Wave w owns sweeps w, w + nw, ...; before step t of sweep s it waits (LDS counter, s_sleep — all waves of a workgroup
// are resident, so this cannot deadlock; nothing waits ACROSS workgroups) until sweep s - 1 has finished step t + 2: the
// waves run at their own pace, no block barrier per tick.
// the 16 x 16 x 4 matrix instruction of each precision and the row its accumulator register r holds in lane (., g)
template <typename T> struct BandMma;
template <> struct BandMma<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int g, int r) { return g + 4 * r; }
};
template <> struct BandMma<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int g, int r) { return 4 * g + r; }
};

// One chase step by one wave.  The three 16 x 16 blocks it touches sit in registers in the accumulator layout of the
// matrix instruction (lane & 15 = column, register r = row rw[r]); a product "v^T block" or "block v" is then FOUR matrix
// instructions with v broadcast in the other operand (K slot g of instruction r = index rw[r], the same in both operands)
// instead of sixteen DPP reduction stages.  A lone wave issues an instruction every 5-8 cycles, so the step is priced in
// INSTRUCTIONS: the band is padded with BAND_PAD zero columns (no window is ever clipped: no masks, no divergent
// branches — a reflector leaves zero rows zero) and every LDS address is "window origin lo * LD + a lane constant"
// (element (r, c) lives at c (LD - 1) + r; a step moves the window by NB in both), the constants computed once per wave.
constexpr int BAND_PAD = BAND_NB - 2;

template <typename T>
struct ChaseOffs {
  int x, x0, xr[4], xr0[4], e[4], a[4], am[4], e2[4], src[4];
  bool low[4], first[4];
};

template <typename T>
__device__ __forceinline__ ChaseOffs<T> band_chase_offsets(int lane) {
  constexpr int NB = BAND_NB, LD = BAND_LD;
  typedef BandMma<T> MM;
  const int i16 = lane & 15, g = lane >> 4;
  ChaseOffs<T> o;
  o.x = -NB * (LD - 1) + i16;                               // column lo - NB, row lo + i16
  o.x0 = -(LD - 1) + i16;                                   // column lo - 1 (t = 0: column s)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rw = MM::row(g, r);
    o.xr[r] = -NB * (LD - 1) + rw;
    o.xr0[r] = -(LD - 1) + rw;
    o.e[r] = (i16 - NB) * (LD - 1) + rw;                    // block to the left: column lo - NB + i16, row lo + rw
    o.a[r] = i16 * (LD - 1) + rw;                           // diagonal block (rw, i16), stored when rw >= i16 ...
    o.am[r] = rw * (LD - 1) + i16;                          // ... else its mirror image
    o.e2[r] = rw * (LD - 1) + NB + i16;                     // block below, transposed: row lo + NB + i16, column lo + rw
    o.src[r] = 16 * g + rw;                                 // a lane of this row that holds column rw
    o.low[r] = rw >= i16;
    o.first[r] = rw == 0;
  }
  return o;
}

template <typename T, typename FLAG>
__device__ __forceinline__ void band_chase_step(T* __restrict__ Bw, const ChaseOffs<T>& o, T* __restrict__ cv,
                                                T* __restrict__ ct, bool t0, int lane, FLAG* flag, int half_done) {
  typedef BandMma<T> MM;
  typedef typename MM::acc_t acc_t;
  const int i16 = lane & 15;
  // every LDS operand is requested before anything is computed
  const T x = Bw[t0 ? o.x0 : o.x];
  T xr[4], e[4], a[4], e2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    xr[r] = Bw[t0 ? o.xr0[r] : o.xr[r]];
    e[r] = Bw[t0 ? o.a[r] : o.e[r]];                        // (t = 0: no block to the left; any valid address)
    a[r] = Bw[o.low[r] ? o.a[r] : o.am[r]];
    e2[r] = Bw[o.e2[r]];
  }
  // ---- the reflector from the column before the window, rows of the window
  const T sigma = band_rl(row16_sum(i16 >= 1 ? x * x : T(0)), 0);
  const T alpha = band_rl(x, 0);
  T tau, beta, scale;
  band_house(alpha, sigma, tau, beta, scale);
  const T v = i16 == 0 ? T(1) : x * scale;                   // v by lane & 15
  T vr[4];                                                  // v by register row
#pragma unroll
  for (int r = 0; r < 4; ++r) vr[r] = o.first[r] ? T(1) : xr[r] * scale;
  if (lane < BAND_NB) cv[lane] = v;
  if (lane == 0) *ct = tau;
  if (tau == T(0)) return;
  acc_t w = {T(0), T(0), T(0), T(0)}, pc = w, u = w;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    w = MM::mma(vr[r], e[r], w);                            // v^T (block to the left), this lane's column
  }
  // (the other two chains are ISSUED after the left block is stored and signalled: a wave issues its matrix instructions in
  // order, 65 cycles apiece in fp64, and the sweep behind waits for that signal)
  __builtin_amdgcn_sched_barrier(0);
  // ---- from the left on the block to the left (its first column becomes (beta, 0, ...) exactly).  This is the only
  // part of the step the sweep behind waits for: it is signalled as soon as these stores have landed
  if (t0) {
    if (lane < BAND_NB) Bw[o.x0] = lane == 0 ? beta : T(0);
  } else {
    const T tw = tau * w[0];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T ne = e[r] - vr[r] * tw;
      if (i16 == 0) ne = o.first[r] ? beta : T(0);
      Bw[o.e[r]] = ne;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) *flag = half_done;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    pc = MM::mma(vr[r], a[r], pc);                          // (D v) of this lane's column, in every register
    u = MM::mma(vr[r], e2[r], u);                           // (E v) of this lane's row of the block below
  }
  // ---- both sides on the diagonal block: p = tau D v, K = tau/2 p.v, q = p - K v, D -= v q^T + q v^T
  const T K = T(0.5) * tau * tau * row16_sum(pc[0] * v);
  const T qc = tau * pc[0] - K * v;
  // (D v) of the register rows: from the lanes that hold it by column (a fourth chain of the fp64 matrix instruction,
  // 64 cycles each on this chip, costs more than four exchanges through the LDS crossbar)
  T pr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) pr[r] = __shfl(pc[0], o.src[r], 64);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const T qr = tau * pr[r] - K * vr[r];
    const T nv = a[r] - (vr[r] * qc + qr * v);
    if (o.low[r]) Bw[o.a[r]] = nv;
  }
  // ---- from the right on the block below: the next bulge
  const T tu = tau * u[0];
#pragma unroll
  for (int r = 0; r < 4; ++r) Bw[o.e2[r]] = e2[r] - tu * vr[r];
}

template <typename T>
__global__ __launch_bounds__(1024) void band_chase_kernel(const T* __restrict__ Sws, const T* __restrict__ Rg,
                                                          T* __restrict__ Cv, T* __restrict__ Ct, T* __restrict__ aux,
                                                          long aux_stride, int n, int np, int TS) {
  constexpr int NB = BAND_NB, LD = BAND_LD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Bd = reinterpret_cast<T*>(smem);                 // (n + BAND_PAD) x LD
  const int npad = n + BAND_PAD;
  // n counters: finished steps of sweep s.  An explicit LDS pointer: through a generic `volatile int*` every poll was a
  // FLAT load with system scope and every update waited for vmcnt(0), i.e. for the step's global reflector stores
  typedef __attribute__((address_space(3))) int lds_int;
  volatile lds_int* prog = (volatile lds_int*)(Bd + (long)npad * LD);
  const int b = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, nw = nt >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* S = Sws + (long)b * n * n;
  const T* Rb = Rg + (long)b * np * NB * NB;
  T* Cvb = Cv + (long)b * (long)n * TS * NB;
  T* Ctb = Ct + (long)b * (long)n * TS;
  for (int idx = tid; idx < npad * LD; idx += nt) {
    const int c = idx / LD, d = idx - c * LD;
    const int r = c + d;
    T val = T(0);
    if (d <= NB && r < n) {
      const int jp = c / NB;
      const int r0 = (jp + 1) * NB;
      if (jp < np && r >= r0) val = Rb[((long)jp * NB + (r - r0)) * NB + (c - jp * NB)];   // R: i = r - r0 <= c - c0
      else val = S[(long)r * n + c];
    }
    Bd[idx] = val;
  }
  for (int i = tid; i < n; i += nt) prog[i] = 0;
  __syncthreads();
  const ChaseOffs<T> offs = band_chase_offsets<T>(lane);
  for (int s = wave; s <= n - 3; s += nw) {
    const int ns = band_nsteps(n, s);
    const int nprev = s > 0 ? band_nsteps(n, s - 1) : 0;
    T* Bw = Bd + (long)(s + 1) * LD;
    T* cv = Cvb + (long)s * TS * NB;
    T* ct = Ctb + (long)s * TS;
    for (int t = 0; t < ns; ++t) {
      if (s > 0) {
        // the sweep before must have finished step t + 1 and the left part of step t + 2 (counter in half steps), or all
        // of its steps
        const int need = t + 3 <= nprev ? 2 * (t + 2) + 1 : 2 * nprev;
        while (prog[s - 1] < need) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
      band_chase_step<T>(Bw, offs, cv, ct, t == 0, lane, prog + s, 2 * t + 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this step's LDS writes have landed
      if (lane == 0) prog[s] = 2 * t + 2;
      Bw += NB * LD;
      cv += NB;
      ct += 1;
    }
  }
  __syncthreads();
  T* ab = aux + (long)b * aux_stride;
  for (int i = tid; i < n; i += nt) {
    ab[i] = Bd[i * LD];
    ab[n + i] = i < n - 1 ? Bd[i * LD + 1] : T(0);
  }
}

// ---- 4. eigenvectors of the tridiagonal matrix -> eigenvectors of T: one workgroup per vector ----------------------------
template <typename T>
__global__ __launch_bounds__(1024) void band_back_kernel(T* __restrict__ Y, const T* __restrict__ Cv,
                                                         const T* __restrict__ Ct, const T* __restrict__ Vg,
                                                         const T* __restrict__ Tg, int n, int p, int np, int TS) {
  constexpr int NB = BAND_NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* y = reinterpret_cast<T*>(smem);                  // n
  T* red = y + n;                                     // 64 x NB
  T* g16 = red + 64 * NB;                             // NB
  T* h16 = g16 + NB;                                  // NB
  const int jv = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, nt = blockDim.x;
  T* Yb = Y + ((long)b * p + jv) * n;
  const T* Cvb = Cv + (long)b * (long)n * TS * NB;
  const T* Ctb = Ct + (long)b * (long)n * TS;
  for (int i = tid; i < n; i += nt) y[i] = Yb[i];
  __syncthreads();
  // chase reflectors, sweeps in reverse; the steps of a sweep touch disjoint rows.  The reflectors of the next DEPTH
  // sweeps are in flight while one is applied (each is a trip to L2).
  {
    const int t = tid >> 4, i16 = tid & 15;
    constexpr int DEPTH = 4;
    T vq[DEPTH], tq[DEPTH];
    auto fetch = [&](int s, T& vv, T& tt) {
      vv = T(0); tt = T(0);
      if (s >= 0 && t < band_nsteps(n, s)) {
        vv = Cvb[((long)s * TS + t) * NB + i16];
        tt = Ctb[(long)s * TS + t];
      }
    };
    int s = n - 3;
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) fetch(s - u, vq[u], tq[u]);
    for (; s >= 0; s -= DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int su = s - u;
        const T vv = vq[u], tt = tq[u];
        fetch(su - DEPTH, vq[u], tq[u]);
        if (su >= 0) {
          const int idx = su + 1 + t * NB + i16;
          const bool on = t < band_nsteps(n, su) && idx < n;
          const T yv = on ? y[idx] : T(0);
          const T dot = row16_sum(vv * yv);
          if (on && tt != T(0)) y[idx] = yv - tt * dot * vv;
        }
        __syncthreads();
      }
    }
  }
  // panels in reverse: y[r0:] -= V (T (V^T y[r0:]))
  {
    const int a = tid % NB, rg = tid / NB;                  // 64 row groups at 1024 threads
    const int RG = nt / NB;
    for (int j = np - 1; j >= 0; --j) {
      const int r0 = (j + 1) * NB, m = n - r0;
      const T* Vb = Vg + ((long)b * np + j) * n * NB + (long)r0 * NB;
      const T* Tb = Tg + ((long)b * np + j) * NB * NB;
      T part = T(0);
      for (int i = rg; i < m; i += RG) part += Vb[(long)i * NB + a] * y[r0 + i];
      red[rg * NB + a] = part;
      __syncthreads();
      if (tid < NB) {
        T gsum = T(0);
        for (int r = 0; r < RG; ++r) gsum += red[r * NB + tid];
        g16[tid] = gsum;
      }
      __syncthreads();
      if (tid < NB) {
        T h = T(0);
        for (int k = 0; k < NB; ++k) h += Tb[tid * NB + k] * g16[k];
        h16[tid] = h;
      }
      __syncthreads();
      for (int i = tid; i < m; i += nt) {
        T s = T(0);
#pragma unroll
        for (int k = 0; k < NB; ++k) s += Vb[(long)i * NB + k] * h16[k];
        y[r0 + i] -= s;
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n; i += nt) Yb[i] = y[i];
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// workspace per matrix, in elements, after the k x k work copy and the aux block of the one-stage form
long band_ws_elems(int k) {
  const long np = band_npanels(k) > 0 ? band_npanels(k) : 1;
  const long TS = band_nsteps(k, 0) > 0 ? band_nsteps(k, 0) : 1;
  const long gs = (k + BAND_STRIP - 1) / BAND_STRIP;
  return np * k * BAND_NB            // Vg
         + 2 * np * BAND_NB * BAND_NB  // Tg, Rg
         + 2 * (long)k * BAND_NB       // Wg, Zg
         + gs * BAND_NB * BAND_NB      // Gp
         + (long)k * TS * BAND_NB      // Cv
         + (long)k * TS;               // Ct
}

// does the chase fit the LDS of one CU at this order?
bool band_supported(int k, int elem_size) {
  if (k < 2 * BAND_NB + 3 || k > 1024) return false;
  const long chase = (long)(k + BAND_NB) * BAND_LD * elem_size + (long)k * 4 + 64;
  const long panel = ((long)(k - BAND_NB) * BAND_LP + 64 * BAND_NB + 3 * BAND_NB + 2 * BAND_NB * BAND_NB) * elem_size + 64;
  return chase <= 160 * 1024 && panel <= 160 * 1024;
}

template <typename T>
struct BandPtrs {
  T *Vg, *Tg, *Rg, *Wg, *Zg, *Gp, *Cv, *Ct;
  int np, TS, gs;
};

template <typename T>
static BandPtrs<T> band_layout(T* base, int B, int k) {
  BandPtrs<T> P;
  P.np = band_npanels(k);
  P.TS = band_nsteps(k, 0) > 0 ? band_nsteps(k, 0) : 1;
  P.gs = (k + BAND_STRIP - 1) / BAND_STRIP;
  const long npa = P.np > 0 ? P.np : 1;
  T* p = base;
  P.Vg = p; p += (long)B * npa * k * BAND_NB;
  P.Tg = p; p += (long)B * npa * BAND_NB * BAND_NB;
  P.Rg = p; p += (long)B * npa * BAND_NB * BAND_NB;
  P.Wg = p; p += (long)B * k * BAND_NB;
  P.Zg = p; p += (long)B * k * BAND_NB;
  P.Gp = p; p += (long)B * P.gs * BAND_NB * BAND_NB;
  P.Cv = p; p += (long)B * k * P.TS * BAND_NB;
  P.Ct = p;
  return P;
}

// stages 1 and 2: S (B x k x k at `S`) from the lower triangle of Tin, (d, e) into aux[b][0 .. 2k)
template <typename T>
int band_tridiag(const T* Tin, T* S, T* aux, long aux_stride, T* bws, int B, int k, long ldt, long sT, hipStream_t st) {
  const BandPtrs<T> P = band_layout<T>(bws, B, k);
  const long total = (long)B * k * k;
  hipLaunchKernelGGL(band_copy_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Tin, S, k, ldt, sT,
                     total);
  // (set on every call: cheap, per device, and the library keeps no state of its own)
  {
    hipError_t e = hipFuncSetAttribute((const void*)band_qr_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)band_w_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)band_chase_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)band_back_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  for (int j = 0; j < P.np; ++j) {
    const int m = k - (j + 1) * BAND_NB;
    const int nstrips = (m + BAND_STRIP - 1) / BAND_STRIP;
    const size_t lds = ((size_t)m * BAND_LP + 64 * BAND_NB + 3 * BAND_NB + 2 * BAND_NB * BAND_NB) * sizeof(T) + 64;
    hipLaunchKernelGGL(band_qr_kernel<T>, dim3(B), dim3(1024), lds, st, S, P.Vg, P.Tg, P.Rg, k, j, P.np);
    hipLaunchKernelGGL(band_w_kernel<T>, dim3(nstrips, B), dim3(512), 8 * BAND_STRIP * BAND_LP * sizeof(T), st, S, P.Vg,
                       P.Wg, P.Gp, k, j, P.np, P.gs);
    hipLaunchKernelGGL(band_z_kernel<T>, dim3(nstrips, B), dim3(256), 0, st, P.Vg, P.Tg, P.Wg, P.Gp, P.Zg, k, j, P.np,
                       P.gs, nstrips);
    // tiles of 64 columns x 64 * groups rows: enough workgroups to fill the chip, not so many that each one's own
    // V_i / Z_i loads weigh as much as its tile
    int groups = 1;
    while (groups < 8 && (long)B * nstrips * ((m + 64 * groups - 1) / (64 * groups)) > 768) groups *= 2;
    const int nchunks = (m + 64 * groups - 1) / (64 * groups);
    hipLaunchKernelGGL(band_update_kernel<T>, dim3(nstrips, nchunks, B), dim3(512), 0, st, S, P.Vg, P.Zg, k, j, P.np,
                       groups);
  }
  const size_t lds = (size_t)(k + BAND_PAD) * BAND_LD * sizeof(T) + (size_t)k * 4 + 64;
  hipLaunchKernelGGL(band_chase_kernel<T>, dim3(B), dim3(1024), lds, st, S, P.Rg, P.Cv, P.Ct, aux, aux_stride, k, P.np,
                     P.TS);
  XK_LAUNCH_CHECK();
  return XK_OK;
}

// stage 4: rows of Y (B, p, k) hold eigenvectors of (d, e); on return, of T
template <typename T>
int band_back(T* Y, T* bws, int B, int k, int p, hipStream_t st) {
  const BandPtrs<T> P = band_layout<T>(bws, B, k);
  const size_t lds = ((size_t)k + 64 * BAND_NB + 2 * BAND_NB) * sizeof(T) + 64;
  hipLaunchKernelGGL(band_back_kernel<T>, dim3(p, B), dim3(1024), lds, st, Y, P.Cv, P.Ct, P.Vg, P.Tg, k, p, P.np, P.TS);
  XK_LAUNCH_CHECK();
  return XK_OK;
}

template int band_tridiag<double>(const double*, double*, double*, long, double*, int, int, long, long, hipStream_t);
template int band_tridiag<float>(const float*, float*, float*, long, float*, int, int, long, long, hipStream_t);
template int band_back<double>(double*, double*, int, int, int, hipStream_t);
template int band_back<float>(float*, float*, int, int, int, hipStream_t);

}  // namespace xk
