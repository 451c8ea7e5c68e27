// xitorch_amd :: K1w — WIDE operator-panel product on the matrix cores (MFMA).
//
//   Y[b, c, n] = sum_i A[b, i, n] * X[b, i, c]        c < P <= 32   (i.e. Y = A^T X; = A X for Hermitian A)
//
// The VALU kernels of xk_dense.hip keep P <= 8 (16) panel columns per pass; a solve with many
// right-hand sides (benchmarks/benchmarks_solve.py: ncols = 50) or an eigensolve with a wide block then
// re-streams the operator ceil(P/8) times.  Here one pass serves up to 32 columns.  At that width the
// arithmetic intensity (2P/s flop per byte: 16 flop/B for fp32, P = 32) is beyond what fp32 VALU code
// sustains while streaming (the guide measures ~52 TF for VALU vs 122-147 TF for MFMA f32), so the
// contraction runs on v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64 — exact fp32/fp64 FMA chains, same
// numerics as the VALU kernels.  Measured at P = 32: fp32 5.6-5.8 TB/s one-pass-equivalent = 93 TFLOP/s with the
// matrix cores 85 % busy at the ~1.5 GHz the chip clocks to under fp32 MFMA load (compute/power-bound; rocBLAS
// 1.08x faster); fp64 5.2-5.6 TB/s with 69 % MFMA busy (balanced); P = 16 fp64 6.3-6.6 TB/s (HBM-bound).
//
// Operand mapping (column orientation, all global loads coalesced, no LDS):
//   MFMA "M" = operator columns n, "K" = operator rows i (the contraction), "N" = panel columns c.
//   A operand: lane l holds A[i0 + (l >> SH)][n0 + VN*(l & MSK) + q]: a 16 B vector load per lane gives
//     VN consecutive columns = the A operands of VN MFMAs (q = 0..VN-1); one load instruction covers
//     K rows x (VN*(MSK+1)) columns in contiguous 128/256 B runs.
//   B operand: X is given ROW-major (B, M, PP) (PP = padded panel width), so lane l reads
//     X[i0 + (l >> SH)][c0 + (l & MSK)]: contiguous 128 B runs.
//   D: per wave VN tiles of (MSK+1) x (MSK+1) accumulators over the whole row slab; slab partials go to
//     the same (B, nslab, P, N) workspace layout as dense_rmm_cols and are folded by fold_slabs.
#include "xk_common.h"

namespace xk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<float> {
  static constexpr int TM = 32, TK = 2, SH = 5, MSK = 31, NACC = 16;
  typedef f32x16 acc_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  // D element `r` of lane l: row m(r, l), column c = l & 31
  static __device__ __forceinline__ int drow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};
// fp32, panels of 9 .. 16 columns (BASELINE configs[4]: a 16-column eigen-block): v_mfma_f32_16x16x4_f32.  With the
// 32-wide form above half of every MFMA's output columns are padding at P = 16 — the matrix pipe was 0.69 busy for
// 27 % of its peak and the launch issue-bound at 0.65 of the HBM roofline (profiles/r03_c5w_mfma_pmc.json); the 16-wide
// form issues half the matrix-core cycles for the same useful flops, so the launch is HBM-bound again.
struct Mfma16f {
  static constexpr int TM = 16, TK = 4, SH = 4, MSK = 15, NACC = 4;
  typedef f32x4 acc_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // D element `r` of lane l: row m = 4 * (l >> 4) + r, column c = l & 15
  static __device__ __forceinline__ int drow(int r, int lane) { return 4 * (lane >> 4) + r; }
};
template <> struct Mfma<double> {
  static constexpr int TM = 16, TK = 4, SH = 4, MSK = 15, NACC = 4;
  typedef f64x4 acc_t;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int r, int lane) { return (lane >> 4) + 4 * r; }
};

// Streaming loads go through buffer descriptors (wave-uniform base of the slab, per-lane offset computed once,
// scalar row-step offset): left to flat addressing the compiler rebuilt a 64-bit address with five integer
// multiplies per load and — to save registers — waited for every load before issuing the next (one 1 KB
// request in flight per wave).  aux = 2: non-temporal (the operator), aux = 0 for the L2-resident panel.
typedef __amdgpu_buffer_rsrc_t WRsrc;
typedef unsigned int wu4 __attribute__((ext_vector_type(4)));
typedef unsigned int wu2 __attribute__((ext_vector_type(2)));

template <typename T>
__device__ __forceinline__ WRsrc wide_rsrc(const T* base, long bytes) {
  const uint64_t v = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  void* b = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  const uint32_t nrec = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xffffffffL ? 0xffffffffL : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(b, (short)0, (int)nrec, 0x00020000);
}
__device__ __forceinline__ f4 wide_ld_a(float, const WRsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 2));
}
__device__ __forceinline__ d2 wide_ld_a(double, const WRsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 2));
}
__device__ __forceinline__ float wide_ld_x(float, const WRsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ double wide_ld_x(double, const WRsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}

// NT = number of panel-column tiles of width TM handled per wave (P <= NT*TM)
template <typename T, int NT, typename MM>
__global__ __launch_bounds__(256) void dense_wide_cols(
    const T* __restrict__ A, const T* __restrict__ Xrm, T* __restrict__ W, int M, int N, long lda, long sA,
    long ldxr, long sXr, int P, int col_tiles, int nslab, int rows_per_slab) {
  typedef typename Vec16<T>::type VT;
  typedef typename MM::acc_t acc_t;
  constexpr int VN = Vec16<T>::n;
  constexpr int WCOLS = VN * MM::TM;            // operator columns per wave (128 fp32 / 32 fp64)
  int bid = blockIdx.x;
  const int ct = bid % col_tiles; bid /= col_tiles;
  const int slab = __builtin_amdgcn_readfirstlane(bid % nslab);
  const int b = __builtin_amdgcn_readfirstlane(bid / nslab);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = __builtin_amdgcn_readfirstlane((ct * 4 + wave) * WCOLS);
  if (n0 >= N) return;                           // whole wave out of range (N % WCOLS handled by the launcher)
  const int kk = lane >> MM::SH;                 // which of the TK rows of a step this lane feeds
  const int mm = lane & MM::MSK;
  const int i0 = slab * rows_per_slab;
  int i1 = i0 + rows_per_slab; i1 = i1 < M ? i1 : M;
  const int nrows = i1 - i0;
  if (nrows <= 0) return;
  // descriptors: the slab's rows of this wave's column window / of the row-major panel
  const unsigned ldab = (unsigned)(lda * (long)sizeof(T)), ldxb = (unsigned)(ldxr * (long)sizeof(T));
  const WRsrc ra = wide_rsrc(A + (long)b * sA + (long)i0 * lda + n0,
                             ((long)(nrows - 1) * lda + (N - n0)) * (long)sizeof(T));
  const WRsrc rx = wide_rsrc(Xrm + (long)b * sXr + (long)i0 * ldxr, (long)nrows * ldxr * (long)sizeof(T));
  const unsigned aoff = (unsigned)kk * ldab + (unsigned)(VN * mm) * (unsigned)sizeof(T);
  unsigned xoff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) xoff[t] = (unsigned)kk * ldxb + (unsigned)(mm + t * MM::TM) * (unsigned)sizeof(T);
  acc_t acc[NT][VN];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < VN; ++q)
#pragma unroll
      for (int r = 0; r < MM::NACC; ++r) acc[t][q][r] = T(0);
  constexpr int U = 8;                           // ring of K-steps in flight (8 KB of A per wave)
  constexpr int BLK = U * MM::TK;                // rows per ring revolution
  const int nfull = (nrows / BLK) * BLK;
  if (nfull > 0) {
    VT av[U];
    T bv[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned r0 = (unsigned)(u * MM::TK);
      av[u] = wide_ld_a(T(0), ra, aoff, r0 * ldab);
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[u][t] = wide_ld_x(T(0), rx, xoff[t], r0 * ldxb);
    }
    const int last = nfull - BLK;                // first row of the last full block (ring refills clamp to it)
    for (int i = 0; i < nfull; i += BLK) {
      int nxt = i + BLK;
      nxt = nxt < last ? nxt : last;
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < VN; ++q) acc[t][q] = MM::mma(av[u][q], bv[u][t], acc[t][q]);
        // refill the step just consumed with the same step of the next block
        const unsigned r0 = (unsigned)(nxt + u * MM::TK);
        av[u] = wide_ld_a(T(0), ra, aoff, r0 * ldab);
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[u][t] = wide_ld_x(T(0), rx, xoff[t], r0 * ldxb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  for (int i = nfull; i < nrows; i += MM::TK) {  // tail steps: rows past the slab end contribute zeros
    const bool ok = i + kk < nrows;
    // a lane whose row lies past the end reads the panel through an out-of-range offset (the hardware returns 0)
    // and the operator from the slab's last row (valid memory): its products vanish
    const unsigned av_off = ok ? aoff + (unsigned)i * ldab
                               : (unsigned)(nrows - 1) * ldab + (unsigned)(VN * mm) * (unsigned)sizeof(T);
    const VT a1 = wide_ld_a(T(0), ra, av_off, 0u);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const T b1 = wide_ld_x(T(0), rx, ok ? xoff[t] + (unsigned)i * ldxb : 0x7ffffff0u, 0u);
#pragma unroll
      for (int q = 0; q < VN; ++q) acc[t][q] = MM::mma(a1[q], b1, acc[t][q]);
    }
  }
  // slab partial: W[b][slab][c][n], c < P, n = n0 + VN*m + q
  T* Wb = W + (((long)b * nslab + slab) * P) * (long)N;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = t * MM::TM + mm;
    if (c < P) {
#pragma unroll
      for (int q = 0; q < VN; ++q)
#pragma unroll
        for (int r = 0; r < MM::NACC; ++r) {
          const int n = n0 + VN * MM::drow(r, lane) + q;
          Wb[(long)c * N + n] = acc[t][q][r];
        }
    }
  }
}

// the 16-wide tile of an element type (fp64 has only the one)
template <typename T> struct Narrow { typedef Mfma<T> type; };
template <> struct Narrow<float> { typedef Mfma16f type; };

template <typename T>
__global__ __launch_bounds__(256) void wide_fold(const T* __restrict__ W, T* __restrict__ Y, int N, int P,
                                                  int nslab, long ldy, long sY, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over B*P*N
  if (idx >= total) return;
  const long per_b = (long)P * N;
  const long b = idx / per_b;
  const long rem = idx - b * per_b;
  const int c = (int)(rem / N);
  const int n = (int)(rem - (long)c * N);
  T s = T(0);
  for (int k = 0; k < nslab; ++k) s += W[(((long)b * nslab + k) * P + c) * (long)N + n];
  Y[b * sY + (long)c * ldy + n] = s;
}

// MFMA tile width serving P panel columns: 16 for fp64 and for fp32 panels of at most 16 columns, else 32
template <typename T>
static int wide_tile(int P) {
  return (sizeof(T) == 8 || P <= 16) ? 16 : 32;
}

template <typename T>
static int wide_nslab(int B, int M, int N, int P) {
  const int WC = Vec16<T>::n * wide_tile<T>(P) * 4;   // columns per block
  const int ct = (N + WC - 1) / WC;
  int nslab = (2048 + B * ct - 1) / (B * ct);
  const int max_slab = (M + 127) / 128;
  if (nslab > max_slab) nslab = max_slab;
  return nslab < 1 ? 1 : nslab;
}

template <typename T>
static int wide_cols(const T* A, const T* Xrm, T* Y, T* ws, long ws_elems, int B, int M, int N, int P, long lda,
                     long sA, long ldxr, long sXr, long ldy, long sY, hipStream_t st) {
  constexpr int VN = Vec16<T>::n;
  if (P > 32 || P < 1) return XK_ERR_ARG;
  const int TM = wide_tile<T>(P), TK = (sizeof(T) == 8 || TM == 16) ? 4 : 2;
  const int WCOLS = VN * TM;
  if ((N % WCOLS) || (lda % VN) || (sA % VN) || ((uintptr_t)A & 15)) return XK_ERR_UNSUPPORTED;
  const int NT = (P + TM - 1) / TM;                   // fp32: 1; fp64: 1 or 2
  if (ldxr < (long)NT * TM) return XK_ERR_ARG;        // X must be padded to whole tiles
  const int ct = (N + 4 * WCOLS - 1) / (4 * WCOLS);
  int nslab = wide_nslab<T>(B, M, N, P);
  while ((long)B * nslab * P * (long)N > ws_elems && nslab > 1) --nslab;
  if ((long)B * nslab * P * (long)N > ws_elems) return XK_ERR_ARG;
  int rps = (M + nslab - 1) / nslab;
  rps = (rps + TK - 1) / TK * TK;                     // whole MFMA K-steps per slab
  nslab = (M + rps - 1) / rps;
  const dim3 grid((unsigned)((long)B * nslab * ct));
  if (sizeof(T) == 4 && TM == 16)
    hipLaunchKernelGGL((dense_wide_cols<T, 1, typename Narrow<T>::type>), grid, dim3(256), 0, st, A, Xrm, ws, M, N,
                       lda, sA, ldxr, sXr, P, ct, nslab, rps);
  else if (NT == 1)
    hipLaunchKernelGGL((dense_wide_cols<T, 1, Mfma<T>>), grid, dim3(256), 0, st, A, Xrm, ws, M, N, lda, sA, ldxr, sXr,
                       P, ct, nslab, rps);
  else
    hipLaunchKernelGGL((dense_wide_cols<T, 2, Mfma<T>>), grid, dim3(256), 0, st, A, Xrm, ws, M, N, lda, sA, ldxr, sXr,
                       P, ct, nslab, rps);
  XK_LAUNCH_CHECK();
  const long tot = (long)B * P * N;
  hipLaunchKernelGGL((wide_fold<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, ws, Y, N, P, nslab, ldy,
                     sY, tot);
  XK_LAUNCH_CHECK();
  return XK_OK;
}

}  // namespace xk

extern "C" {

long xk_dense_wide_workspace_elems(int B, int M, int N, int P, int elem_size) {
  const int ns = elem_size == 8 ? xk::wide_nslab<double>(B, M, N, P) : xk::wide_nslab<float>(B, M, N, P);
  return (long)B * ns * P * (long)N;
}

// padded panel width the row-major X must have: whole MFMA tiles (fp64 and fp32 up to 16 columns: 16-wide tiles;
// wider fp32 panels: 32)
int xk_dense_wide_padded_width(int P, int elem_size) {
  const int tm = (elem_size == 8 || P <= 16) ? 16 : 32;
  return (P + tm - 1) / tm * tm;
}

int xk_dense_wide_f32(const float* A, const float* Xrm, float* Y, float* ws, long ws_elems, int B, int M, int N,
                      int P, long lda, long sA, long ldxr, long sXr, long ldy, long sY, void* stream) {
  if (B < 0 || M < 0 || N < 0) return XK_ERR_ARG;
  if (B == 0 || M == 0 || N == 0) return XK_OK;
  return xk::wide_cols<float>(A, Xrm, Y, ws, ws_elems, B, M, N, P, lda, sA, ldxr, sXr, ldy, sY, (hipStream_t)stream);
}
int xk_dense_wide_f64(const double* A, const double* Xrm, double* Y, double* ws, long ws_elems, int B, int M, int N,
                      int P, long lda, long sA, long ldxr, long sXr, long ldy, long sY, void* stream) {
  if (B < 0 || M < 0 || N < 0) return XK_ERR_ARG;
  if (B == 0 || M == 0 || N == 0) return XK_OK;
  return xk::wide_cols<double>(A, Xrm, Y, ws, ws_elems, B, M, N, P, lda, sA, ldxr, sXr, ldy, sY, (hipStream_t)stream);
}

}  // extern "C"
