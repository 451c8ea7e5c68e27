// xitorch_amd :: operator-gradient kernels of the implicit backward passes.
//
// The backward of `solve` / `symeig` / `rootfinder` ends with a vector-Jacobian product through the
// operator apply (`loss = -A.mm(x)` then `autograd.grad(loss, params, v)`, xitorch/linalg/solve.py:188-195,
// linalg/symeig.py:374-379, optimize/rootfinder.py:352-362).  For the two native operators that VJP is a
// pure streaming WRITE of the operator-sized gradient:
//
//   xk_banded_grad   G[b,d,i]  (+)= sum_c U[b,c,i] * W[b,c,i+d-hb]      (DIA band gradient; BASELINE configs[2])
//   xk_dense_outer   G[b,i,j]  (+)= sum_c U[b,c,i] * W[b,c,j]           (dense operator gradient; configs[3])
//
// Both are HBM-write-bound: bytes = B*(2hb+1)*N*s resp. B*M*N*s (+ the two panels once).  Panels are
// panel-major (B, C, N) like everywhere else.  C <= 8 per pass; the host loops with accumulate = 1.
#include "xk_common.h"

namespace xk {

// ---------------------------------------------------------------------------------------------
// Band gradient.  One block = ROWS = 256*VN consecutive rows i of one batch member; the W tile
// (rows + halo of hb on both sides) is staged once in LDS exactly like the x tile of xk_banded_mm,
// the U values of the thread's VN rows sit in registers, and every diagonal d is one coalesced
// 16 B non-temporal store per lane.  Entries whose column i+d-hb falls outside the matrix get 0.
// ---------------------------------------------------------------------------------------------
template <typename T, int C, bool VEC>
__global__ __launch_bounds__(256) void banded_grad_kernel(
    const T* __restrict__ U, const T* __restrict__ W, T* __restrict__ G, int N, int hb, long ldu, long sU,
    long ldw, long sW, long sG, int row_tiles, int accumulate) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  constexpr int ROWS = 256 * VN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* ws = reinterpret_cast<T*>(smem);       // C x (ROWS + 2*hb)
  const int b = blockIdx.x / row_tiles;
  const int rt = blockIdx.x - b * row_tiles;
  const int i0 = rt * ROWS;
  const int tw = ROWS + 2 * hb;
  const T* Wb = W + (long)b * sW;
  for (int idx = threadIdx.x; idx < C * tw; idx += 256) {
    const int c = idx / tw, l = idx - c * tw;
    const int g = i0 - hb + l;
    ws[idx] = (g >= 0 && g < N) ? Wb[(long)c * ldw + g] : T(0);      // zero halo == masked band entries
  }
  __syncthreads();
  const int li0 = threadIdx.x * VN;
  const int r0 = i0 + li0;
  if (r0 >= N) return;
  const T* Ub = U + (long)b * sU;
  T u[C][VN];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int q = 0; q < VN; ++q) u[c][q] = (r0 + q < N) ? Ub[(long)c * ldu + r0 + q] : T(0);
  T* Gb = G + (long)b * sG;
  const int nd = 2 * hb + 1;
#pragma unroll 4
  for (int d = 0; d < nd; ++d) {
    T o[VN];
#pragma unroll
    for (int q = 0; q < VN; ++q) {
      T a = T(0);
#pragma unroll
      for (int c = 0; c < C; ++c) a += u[c][q] * ws[c * tw + li0 + q + d];
      o[q] = a;
    }
    T* dst = Gb + (long)d * N + r0;
    if (VEC) {
      VT v;
      if (accumulate) {
        v = *reinterpret_cast<const VT*>(dst);
#pragma unroll
        for (int q = 0; q < VN; ++q) v[q] += o[q];
      } else {
#pragma unroll
        for (int q = 0; q < VN; ++q) v[q] = o[q];
      }
      __builtin_nontemporal_store(v, reinterpret_cast<VT*>(dst));
    } else {
#pragma unroll
      for (int q = 0; q < VN; ++q)
        if (r0 + q < N) dst[q] = accumulate ? dst[q] + o[q] : o[q];
    }
  }
}

template <typename T, int C>
static int banded_grad_launch(const T* U, const T* W, T* G, int B, int N, int hb, long ldu, long sU, long ldw,
                              long sW, long sG, int accumulate, hipStream_t st) {
  constexpr int VN = Vec16<T>::n;
  constexpr int ROWS = 256 * VN;
  const int row_tiles = (N + ROWS - 1) / ROWS;
  const size_t lds = (size_t)C * (ROWS + 2 * hb) * sizeof(T);
  if (lds > 160 * 1024) return XK_ERR_UNSUPPORTED;
  const bool vec = (N % VN == 0) && (sG % VN == 0) && (((uintptr_t)G & 15) == 0);
  const dim3 grid((unsigned)((long)B * row_tiles));
#define XK_BG_LAUNCH(VECF)                                                                                  \
  {                                                                                                         \
    if (lds > 64 * 1024) {                                                                                  \
      hipError_t e = hipFuncSetAttribute((const void*)banded_grad_kernel<T, C, VECF>,                       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
      if (e != hipSuccess) return (int)e;                                                                   \
    }                                                                                                       \
    hipLaunchKernelGGL((banded_grad_kernel<T, C, VECF>), grid, dim3(256), lds, st, U, W, G, N, hb, ldu, sU, \
                       ldw, sW, sG, row_tiles, accumulate);                                                 \
  }
  if (vec) XK_BG_LAUNCH(true) else XK_BG_LAUNCH(false)
#undef XK_BG_LAUNCH
  XK_LAUNCH_CHECK();
  return XK_OK;
}

template <typename T>
static int banded_grad(const T* U, const T* W, T* G, int B, int N, int hb, int C, long ldu, long sU, long ldw,
                       long sW, long sG, int accumulate, hipStream_t st) {
  int c0 = 0;
  while (c0 < C) {
    const int pc = (C - c0) >= 8 ? 8 : (C - c0);
    const T* Uc = U + (long)c0 * ldu;
    const T* Wc = W + (long)c0 * ldw;
    const int acc = (accumulate || c0 > 0) ? 1 : 0;
    int rc = XK_ERR_UNSUPPORTED;
    switch (pc) {
#define XK_CASE(CC) \
  case CC: rc = banded_grad_launch<T, CC>(Uc, Wc, G, B, N, hb, ldu, sU, ldw, sW, sG, acc, st); break;
      XK_CASE(1) XK_CASE(2) XK_CASE(3) XK_CASE(4) XK_CASE(5) XK_CASE(6) XK_CASE(7) XK_CASE(8)
#undef XK_CASE
    }
    if (rc != XK_OK) return rc;
    c0 += pc;
  }
  return XK_OK;
}

// ---------------------------------------------------------------------------------------------
// Dense outer product  G[b,i,j] (+)= sum_c U[b,c,i] W[b,c,j].
// Block = 256 threads = one slab of 256*VN columns x RT rows of one batch member.  A lane keeps the
// C panel values of its VN columns in registers for the whole slab (read once), the U values of a
// row are wave-uniform scalar loads, and every row is one contiguous 16 B/lane non-temporal store:
// the kernel is a pure B*M*N*s write stream.
// ---------------------------------------------------------------------------------------------
constexpr int OUTER_RT = 64;     // rows per block

template <typename T, int C, bool VEC>
__global__ __launch_bounds__(256) void dense_outer_kernel(
    const T* __restrict__ U, const T* __restrict__ W, T* __restrict__ G, int M, int N, long ldu, long sU,
    long ldw, long sW, long ldg, long sG, int col_slabs, int row_tiles, int accumulate) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  constexpr int SLAB = 256 * VN;
  // blockIdx.x -> (batch member, row tile, column slab); column slab fastest so that consecutive blocks
  // (round-robin over the XCDs) write disjoint 4-8 KB runs of the same rows
  int rem = blockIdx.x;
  const int cs = rem % col_slabs; rem /= col_slabs;
  const int rt = rem % row_tiles;
  const int b = __builtin_amdgcn_readfirstlane(rem / row_tiles);
  const int j0 = cs * SLAB + threadIdx.x * VN;
  const int i0 = __builtin_amdgcn_readfirstlane(rt * OUTER_RT);
  if (j0 >= N) return;
  const T* Wb = W + (long)b * sW;
  const T* Ub = U + (long)b * sU;
  T w[C][VN];
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int q = 0; q < VN; ++q) w[c][q] = (j0 + q < N) ? Wb[(long)c * ldw + j0 + q] : T(0);
  T* Gb = G + (long)b * sG;
  const int i_end = (i0 + OUTER_RT < M) ? i0 + OUTER_RT : M;
#pragma unroll 4
  for (int i = i0; i < i_end; ++i) {
    T o[VN];
#pragma unroll
    for (int q = 0; q < VN; ++q) o[q] = T(0);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const T uc = Ub[(long)c * ldu + i];            // wave-uniform: scalar load
#pragma unroll
      for (int q = 0; q < VN; ++q) o[q] += uc * w[c][q];
    }
    T* dst = Gb + (long)i * ldg + j0;
    if (VEC) {
      VT v;
      if (accumulate) {
        v = *reinterpret_cast<const VT*>(dst);
#pragma unroll
        for (int q = 0; q < VN; ++q) v[q] += o[q];
      } else {
#pragma unroll
        for (int q = 0; q < VN; ++q) v[q] = o[q];
      }
      __builtin_nontemporal_store(v, reinterpret_cast<VT*>(dst));
    } else {
#pragma unroll
      for (int q = 0; q < VN; ++q)
        if (j0 + q < N) dst[q] = accumulate ? dst[q] + o[q] : o[q];
    }
  }
}

template <typename T, int C>
static int dense_outer_launch(const T* U, const T* W, T* G, int B, int M, int N, long ldu, long sU, long ldw,
                              long sW, long ldg, long sG, int accumulate, hipStream_t st) {
  constexpr int VN = Vec16<T>::n;
  constexpr int SLAB = 256 * VN;
  const int col_slabs = (N + SLAB - 1) / SLAB;
  const int row_tiles = (M + OUTER_RT - 1) / OUTER_RT;
  const long nblk = (long)B * col_slabs * row_tiles;
  if (nblk > 0x7fffffffL) return XK_ERR_UNSUPPORTED;
  const bool vec = (N % VN == 0) && (ldg % VN == 0) && (sG % VN == 0) && (((uintptr_t)G & 15) == 0);
  if (vec)
    hipLaunchKernelGGL((dense_outer_kernel<T, C, true>), dim3((unsigned)nblk), dim3(256), 0, st, U, W, G, M, N,
                       ldu, sU, ldw, sW, ldg, sG, col_slabs, row_tiles, accumulate);
  else
    hipLaunchKernelGGL((dense_outer_kernel<T, C, false>), dim3((unsigned)nblk), dim3(256), 0, st, U, W, G, M, N,
                       ldu, sU, ldw, sW, ldg, sG, col_slabs, row_tiles, accumulate);
  XK_LAUNCH_CHECK();
  return XK_OK;
}

template <typename T>
static int dense_outer(const T* U, const T* W, T* G, int B, int M, int N, int C, long ldu, long sU, long ldw,
                       long sW, long ldg, long sG, int accumulate, hipStream_t st) {
  int c0 = 0;
  while (c0 < C) {
    const int pc = (C - c0) >= 8 ? 8 : (C - c0);
    const T* Uc = U + (long)c0 * ldu;
    const T* Wc = W + (long)c0 * ldw;
    const int acc = (accumulate || c0 > 0) ? 1 : 0;
    int rc = XK_ERR_UNSUPPORTED;
    switch (pc) {
#define XK_CASE(CC) \
  case CC: rc = dense_outer_launch<T, CC>(Uc, Wc, G, B, M, N, ldu, sU, ldw, sW, ldg, sG, acc, st); break;
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

#define XK_DEFINE_GRAD(SUF, T)                                                                             \
  int xk_banded_grad_##SUF(const T* U, const T* W, T* G, int B, int N, int hb, int C, long ldu, long sU,   \
                           long ldw, long sW, long sG, int accumulate, void* stream) {                     \
    if (B < 0 || N < 0 || hb < 0 || C < 0) return XK_ERR_ARG;                                              \
    if (B == 0 || N == 0) return XK_OK;                                                                    \
    if (C == 0) {                                                                                          \
      if (accumulate) return XK_OK;                                                                        \
      for (int b = 0; b < B; ++b) {                                                                        \
        hipError_t e = hipMemsetAsync(G + (long)b * sG, 0, sizeof(T) * (size_t)(2 * hb + 1) * N,           \
                                      (hipStream_t)stream);                                                \
        if (e != hipSuccess) return (int)e;                                                                \
      }                                                                                                    \
      return XK_OK;                                                                                        \
    }                                                                                                      \
    return xk::banded_grad<T>(U, W, G, B, N, hb, C, ldu, sU, ldw, sW, sG, accumulate,                      \
                              (hipStream_t)stream);                                                        \
  }                                                                                                        \
  int xk_dense_outer_##SUF(const T* U, const T* W, T* G, int B, int M, int N, int C, long ldu, long sU,    \
                           long ldw, long sW, long ldg, long sG, int accumulate, void* stream) {           \
    if (B < 0 || M < 0 || N < 0 || C < 1) return XK_ERR_ARG;                                               \
    if (B == 0 || M == 0 || N == 0) return XK_OK;                                                          \
    return xk::dense_outer<T>(U, W, G, B, M, N, C, ldu, sU, ldw, sW, ldg, sG, accumulate,                  \
                              (hipStream_t)stream);                                                        \
  }

XK_DEFINE_GRAD(f64, double)
XK_DEFINE_GRAD(f32, float)

}  // extern "C"
