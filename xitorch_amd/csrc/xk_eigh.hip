// xitorch_amd :: K3 — batched small symmetric eigensolver for the Rayleigh–Ritz matrix T = V^T A V.
//
// Replaces `torch.linalg.eigh(T)` + `_take_eigpairs` of the reference Davidson loop
// (xitorch/_impls/linalg/symeig.py:174-175).  The library path (rocSOLVER syevd, one Householder
// reflector per launch) costs ~6 ms per call for 64 matrices of order ~100 — 20 % of the whole
// eigensolve — so T is diagonalised here by a parallel cyclic two-sided Jacobi method, one
// workgroup (1024 threads) per batch member, with the matrix resident in LDS:
//
//   * round-robin ("tournament") ordering: each of the k-1 steps of a sweep applies k/2 disjoint
//     rotations; the 2x2 blocks T[{p1,q1},{p2,q2}] <- G1^T [..] G2 are independent, so a step is
//     one parameter phase + one update phase (2 barriers);
//   * LDS: k2 x (k2+1) doubles (odd pitch => column walks are conflict-free), k <= 128 -> 132 KB
//     of the 160 KB/CU;
//   * only the `p` wanted eigenvectors are formed: the rotations (c,s) are logged to a global
//     scratch in order, and replayed BACKWARDS on the p unit vectors of the selected eigenvalues
//     (y = J_1 J_2 ... J_m e_i), so no k x k eigenvector matrix is ever stored;
//   * eigenvalues are selected lowest / uppermost and returned ascending — the index ordering of
//     `_take_eigpairs` (symeig.py:255-264).
//
// Accuracy: Jacobi is backward stable with small relative errors; convergence test
// off(T)^2 <= (eps*k)^2 * ||T||_F^2 or a sweep without rotations.
#include "xk_common.h"

namespace xk {

template <typename T> struct Eps;
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };
template <> struct Eps<float> { static constexpr float v = 1.1920929e-07f; };

template <typename T>
__device__ __forceinline__ T block_sum_1024(T v, T* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  T s = T(0);
  const int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

__device__ __forceinline__ void rr_pair(int I, int r, int k2, int& p, int& q) {
  // round-robin tournament on k2 (even) players: player k2-1 stays, the others rotate
  const int n1 = k2 - 1;
  if (I == 0) { p = n1; q = r; }
  else {
    p = r + I; if (p >= n1) p -= n1;
    q = r - I; if (q < 0) q += n1;
  }
}

template <typename T>
__global__ __launch_bounds__(1024) void jacobi_eigh_kernel(
    const T* __restrict__ Tin, T* __restrict__ lam_out, T* __restrict__ Y_out, T* __restrict__ rotlog,
    int* __restrict__ sweeps_out, int k, int p, int uppest, int max_sweeps, long ldt, long sT,
    long log_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int k2 = (k + 1) & ~1;
  const int m = k2 >> 1;
  const int ld = k2 + 1;
  T* S = reinterpret_cast<T*>(smem);                 // k2 x ld
  T* cs = S + (long)k2 * ld;                          // 2*m
  T* red = cs + 2 * m;                                // 16
  T* rowabs = red + 16;                               // k2: sum_{l != j} |S_jl|
  T* rowsq = rowabs + k2;                             // k2: sum_{l != j} S_jl^2
  int* sel = reinterpret_cast<int*>(rowsq + k2);      // p selected indices (ascending eigenvalue)
  int* flag = sel + 32;                               // k2 wanted flags + 1 decision slot
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int nt = blockDim.x;
  const T* Tb = Tin + (long)b * sT;
  T* logb = rotlog + (long)b * log_stride;

  // load the LOWER triangle (eigh's default UPLO='L') and mirror it
  for (int idx = tid; idx < k2 * k2; idx += nt) {
    const int i = idx / k2, j = idx - i * k2;
    T v = T(0);
    if (i < k && j < k) v = (i >= j) ? Tb[(long)i * ldt + j] : Tb[(long)j * ldt + i];
    S[i * ld + j] = v;
  }
  __syncthreads();
  T nrm = T(0);
  for (int idx = tid; idx < k2 * k2; idx += nt) {
    const int i = idx / k2, j = idx - i * k2;
    const T v = S[i * ld + j];
    nrm += v * v;
  }
  nrm = block_sum_1024(nrm, red);
  const T tol2 = (Eps<T>::v * k) * (Eps<T>::v * k) * nrm;

  // the 2x2 blocks (I, J) a thread updates are the same in every step: decode them once
  constexpr int MAXBLK = 4;                      // m*m <= 64*64 = 4096 blocks over 1024 threads
  int blkI[MAXBLK], blkJ[MAXBLK];
#pragma unroll
  for (int u = 0; u < MAXBLK; ++u) {
    const int idx = tid + u * nt;
    if (idx < m * m) { blkI[u] = idx / m; blkJ[u] = idx - blkI[u] * m; }
    else { blkI[u] = -1; blkJ[u] = 0; }
  }
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    // convergence test on the off-diagonal mass
    T off = T(0);
    for (int idx = tid; idx < k2 * k2; idx += nt) {
      const int i = idx / k2, j = idx - i * k2;
      if (i != j) { const T v = S[i * ld + j]; off += v * v; }
    }
    off = block_sum_1024(off, red);
    if (!(off > tol2)) break;
    // ---- early exit: only the p wanted eigenpairs have to be resolved ------------------------------
    // If the rows of the p extreme diagonal entries are decoupled from everything else (their
    // off-diagonal mass is at round-off level) they ARE eigenpairs, whatever mixing is left inside the
    // unwanted block — provided no eigenvalue of that block can still cross into the wanted range, which
    // the Gershgorin row bounds of the unwanted rows certify.  Clustered bulks otherwise cost Jacobi many
    // extra sweeps that the eigensolver never looks at.
    if (sweep > 0) {
      if (tid < k) {
        const T di = S[tid * ld + tid];
        int rank = 0;
        T ra = T(0), rq = T(0);
        for (int j = 0; j < k; ++j) {
          const T v = S[tid * ld + j];
          const T dj = S[j * ld + j];
          rank += (dj < di || (dj == di && j < tid)) ? 1 : 0;
          if (j != tid) { ra += fabs(v); rq += v * v; }
        }
        rowabs[tid] = ra;
        rowsq[tid] = rq;
        flag[tid] = uppest ? (rank >= k - p) : (rank < p);
      }
      __syncthreads();
      if (tid == 0) {
        T offsel = T(0), edge_w = uppest ? T(INFINITY) : T(-INFINITY), edge_u = uppest ? T(-INFINITY) : T(INFINITY);
        for (int j = 0; j < k; ++j) {
          const T dj = S[j * ld + j];
          if (flag[j]) {
            offsel += rowsq[j];
            edge_w = uppest ? (dj < edge_w ? dj : edge_w) : (dj > edge_w ? dj : edge_w);
          } else {
            const T g = uppest ? dj + rowabs[j] : dj - rowabs[j];      // Gershgorin bound of an unwanted row
            edge_u = uppest ? (g > edge_u ? g : edge_u) : (g < edge_u ? g : edge_u);
          }
        }
        const bool separated = (p == k) || (uppest ? (edge_u < edge_w) : (edge_u > edge_w));
        flag[k2] = (!(offsel > tol2) && separated) ? 1 : 0;
      }
      __syncthreads();
      const int done = flag[k2];
      __syncthreads();
      if (done) break;
    }
    for (int r = 0; r < k2 - 1; ++r) {
      if (tid < m) {
        int pp, qq;
        rr_pair(tid, r, k2, pp, qq);
        const T apq = S[pp * ld + qq];
        const T app = S[pp * ld + pp], aqq = S[qq * ld + qq];
        T c = T(1), s = T(0);
        // skip rotations that cannot change anything:  |apq| <= 0.01 eps sqrt(|app aqq|)   (compared squared)
        const T thr2 = (Eps<T>::v * T(0.01)) * (Eps<T>::v * T(0.01)) * fabs(app * aqq);
        if (apq * apq > thr2 && apq != T(0)) {
          // the annihilating rotation without divisions: with d = aqq - app, e = 2 apq, h = hypot(d, e),
          // w = |d| + h:  tan = sign(d) e / w,  cos = sqrt(w / 2h) = w r,  sin = sign(d) e r,  r = rsqrt(2 h w)
          // (the dependent sqrt/div chain of the textbook formula is what a Jacobi step waits on)
          const T d = aqq - app, e = T(2) * apq;
          const T h = sqrt(d * d + e * e);
          const T w = fabs(d) + h;
          const T r = rsqrt(T(2) * h * w);
          c = w * r;
          s = (d >= T(0) ? e : -e) * r;
        }
        cs[2 * tid] = c;
        cs[2 * tid + 1] = s;
        T* lg = logb + ((long)(sweep * (k2 - 1) + r) * m + tid) * 2;
        lg[0] = c;
        lg[1] = s;
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < MAXBLK; ++u) {
        if (blkI[u] >= 0) {
          const int I = blkI[u], J = blkJ[u];
          int p1, q1, p2, q2;
          rr_pair(I, r, k2, p1, q1);
          rr_pair(J, r, k2, p2, q2);
          const T c1 = cs[2 * I], s1 = cs[2 * I + 1], c2 = cs[2 * J], s2 = cs[2 * J + 1];
          const T a = S[p1 * ld + p2], bb = S[p1 * ld + q2], cc = S[q1 * ld + p2], d = S[q1 * ld + q2];
          // rows: G1^T * [[a,bb],[cc,d]]
          const T r1a = c1 * a - s1 * cc, r1b = c1 * bb - s1 * d;
          const T r2a = s1 * a + c1 * cc, r2b = s1 * bb + c1 * d;
          // cols: * G2
          T na = r1a * c2 - r1b * s2, nb = r1a * s2 + r1b * c2;
          T nc = r2a * c2 - r2b * s2, nd = r2a * s2 + r2b * c2;
          if (I == J && (c1 != T(1))) { nb = T(0); nc = T(0); }   // the annihilated pair, exactly
          S[p1 * ld + p2] = na; S[p1 * ld + q2] = nb; S[q1 * ld + p2] = nc; S[q1 * ld + q2] = nd;
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0) sweeps_out[b] = sweep;

  // ---- select the wanted eigenvalues (ascending) ---------------------------------------------
  // rank of d_i among the k real diagonal entries (ties broken by index)
  __syncthreads();
  if (tid < k) {
    const T di = S[tid * ld + tid];
    int rank = 0;
    for (int j = 0; j < k; ++j) {
      const T dj = S[j * ld + j];
      rank += (dj < di || (dj == di && j < tid)) ? 1 : 0;
    }
    const int pos = uppest ? rank - (k - p) : rank;
    if (pos >= 0 && pos < p) {
      sel[pos] = tid;
      lam_out[(long)b * p + pos] = di;
    }
  }
  __syncthreads();

  // ---- replay the rotations backwards on the p unit vectors ----------------------------------
  // Yv[c][i], c < p, i < k2, stored in S (no longer needed)
  T* Yv = S;   // pitch ld; p <= k2 rows fit in the matrix area
  const int nwork = p * m;
  int wc = -1, wI = 0;
  if (tid < nwork) { wc = tid / m; wI = tid - wc * m; }
  for (int idx = tid; idx < p * ld; idx += nt) Yv[idx] = T(0);
  __syncthreads();
  if (tid < p) Yv[tid * ld + sel[tid]] = T(1);
  __syncthreads();
  const int total_steps = sweep * (k2 - 1);
  for (int st = total_steps - 1; st >= 0; st -= 4) {
    // prefetch up to 4 steps of (c,s) for this thread's pair to overlap the global-load latency
    T pc[4], ps[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s2 = st - u;
      if (wc >= 0 && s2 >= 0) {
        const T* lg = logb + ((long)s2 * m + wI) * 2;
        pc[u] = lg[0];
        ps[u] = lg[1];
      } else { pc[u] = T(1); ps[u] = T(0); }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s2 = st - u;
      if (s2 >= 0) {
        if (wc >= 0) {
          const int r = s2 % (k2 - 1);
          int pp, qq;
          rr_pair(wI, r, k2, pp, qq);
          const T yp = Yv[wc * ld + pp], yq = Yv[wc * ld + qq];
          // y <- J y with J[p,p]=c, J[p,q]=s, J[q,p]=-s, J[q,q]=c
          Yv[wc * ld + pp] = pc[u] * yp + ps[u] * yq;
          Yv[wc * ld + qq] = -ps[u] * yp + pc[u] * yq;
        }
        __syncthreads();
      }
    }
  }
  for (int idx = tid; idx < p * k; idx += nt) {
    const int c = idx / k, i = idx - c * k;
    Y_out[((long)b * p + c) * k + i] = Yv[c * ld + i];
  }
}

}  // namespace xk

extern "C" {

// scratch (elements) for the rotation log of one call
long xk_small_eigh_workspace_elems(int B, int k, int max_sweeps) {
  const long k2 = (k + 1) & ~1;
  return (long)B * max_sweeps * (k2 - 1) * (k2 / 2) * 2;
}

#define XK_DEFINE_EIGH(SUF, T)                                                                         \
  int xk_small_eigh_##SUF(const T* Tin, T* lam, T* Y, T* ws, long ws_elems, int* sweeps, int B, int k, \
                          int p, int uppest, int max_sweeps, long ldt, long sT, void* stream) {        \
    if (B < 0 || k < 1 || p < 1 || p > k || k > 128 || p > 16) return XK_ERR_ARG;                     \
    if (B == 0) return XK_OK;                                                                          \
    const long k2 = (k + 1) & ~1;                                                                      \
    const long per = (long)max_sweeps * (k2 - 1) * (k2 / 2) * 2;                                       \
    if (ws_elems < per * B) return XK_ERR_ARG;                                                         \
    const size_t lds = (size_t)(k2 * (k2 + 1) + 3 * k2 + 16) * sizeof(T) + (size_t)(k2 + 40) * sizeof(int);    \
    hipError_t e = hipFuncSetAttribute((const void*)xk::jacobi_eigh_kernel<T>,                         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
    if (e != hipSuccess) return (int)e;                                                                \
    /* one thread per 2x2 block while that fits: fewer waves per barrier for small bases */            \
    const long nblk = (k2 / 2) * (k2 / 2);                                                             \
    int nthr = (int)((nblk + 255) / 256 * 256);                                                        \
    nthr = nthr > 1024 ? 1024 : (nthr < 256 ? 256 : nthr);                                             \
    hipLaunchKernelGGL((xk::jacobi_eigh_kernel<T>), dim3(B), dim3(nthr), lds, (hipStream_t)stream,     \
                       Tin, lam, Y, ws, sweeps, k, p, uppest, max_sweeps, ldt, sT, per);               \
    XK_LAUNCH_CHECK();                                                                                 \
    return XK_OK;                                                                                      \
  }

XK_DEFINE_EIGH(f64, double)
XK_DEFINE_EIGH(f32, float)

}  // extern "C"
