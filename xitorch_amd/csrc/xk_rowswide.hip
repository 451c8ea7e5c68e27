// xitorch_amd :: K1wr — operator-panel product in the ROW orientation for wide panels,
//
//        Y[b, c, i] = sum_j A[b, i, j] * X[b, c, j]          c < P,  12 <= P (any width, 32 columns per pass)
//
// i.e. `torch.matmul(mat, x)` of a non-Hermitian MatrixLinearOperator with many right-hand sides
// (xitorch/_core/linop.py:695-696; benchmarks/benchmarks_solve.py:11-15: ncols = 50, and every BiCGStab / GMRES
// apply of a multi-RHS solve, _impls/linalg/solve.py:278,284).  Round 1 served this shape by ceil(P/8) passes of
// the VALU rows kernel, or by a transposed COPY of the operator so that the MFMA kernel K1w (column orientation)
// could be used.  Neither is needed:
//
// Mapping.  The contraction runs over the operator's columns, which is also the direction of its contiguous
// storage, so a coalesced load hands a lane two CONSECUTIVE-column elements of one row — the wrong shape both for
// MFMA operands (which want 16 rows x 4 columns per instruction) and for per-lane accumulation.  The tile is
// therefore turned through LDS, per wave and without block barriers, and read back in the MFMA operand layouts:
//
//   load   : a wave streams a 64-row x 16-column sub-tile (fp64; 32 columns fp32) with 8 coalesced 16 B/lane
//            non-temporal loads (each instruction = 8 rows x one 128 B line), plus the matching 16 columns of the
//            panel (PC x 128 B)
//   turn   : ds_write_b128 into a [row][column] LDS tile with an odd 16 B pitch (the panel sub-tile goes in
//            transposed, [column][panel column])
//   compute: v_mfma_f64_16x16x4 / v_mfma_f32_16x16x4 with M = 16 operator rows, K = 4 operator columns, N = 16
//            panel columns; both operands are read from LDS directly in the MFMA lane layouts (conflict-free
//            ds_read_b64), accumulators (4 row blocks x 1-2 panel blocks) stay in AGPRs over the whole row:
//            no cross-lane reduction, results leave once per row.
//   History of this phase (fp64, 16 x 16384^2): scalar-cache panel loads + per-lane VALU FMAs: 1.0-3.6 TB/s (SMEM
//   returns out of order -> lgkmcnt(0) after every batch of four); panel broadcast from LDS + VALU FMAs: 3.9 TB/s
//   at P = 16, 2.4 at P = 32 (LDS-bound: one 4-cycle broadcast read per two FMAs); MFMA: see DESIGN.md.
//
// The contraction can be split over blockIdx.y (few long rows: Krylov solves with a small batch) with a fixed-
// order fold of the partial panels (deterministic).
#include "xk_common.h"

namespace xk {

constexpr int RW_ROWS = 64;                    // rows per wave (lane <-> row in the compute phase)
constexpr int RW_SEG_BYTES = 128;              // contiguous bytes per row and sub-tile (one cache line)
constexpr int RW_PITCH_BYTES = RW_SEG_BYTES + 16;   // 36 dwords: 16 consecutive rows hit 16 distinct 16 B bank groups
constexpr int RW_TILE_LDS = RW_ROWS * RW_PITCH_BYTES;   // 9216 B: the operator sub-tile of one wave
// + the panel sub-tile of the same columns, stored [column][panel column] (the MFMA B-operand order)
__host__ __device__ constexpr int rw_wave_lds(int pc, int esize) {
  return RW_TILE_LDS + (RW_SEG_BYTES / esize) * (pc * esize + 16);
}

typedef __amdgpu_buffer_rsrc_t RwRsrc;
typedef unsigned int rwu4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ RwRsrc rw_rsrc(const T* base, long bytes) {
  const uint64_t v = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  void* b = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  const uint32_t nrec = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xffffffffL ? 0xffffffffL : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(b, (short)0, (int)nrec, 0x00020000);
}

typedef float rw_f32x4 __attribute__((ext_vector_type(4)));
typedef double rw_f64x4 __attribute__((ext_vector_type(4)));

// v_mfma_{f32,f64}_16x16x4: D[m][n] += sum_k A[m][k] B[k][n], exact FMA chains.
//   A operand: lane l holds A[m = l & 15][k = l >> 4];  B operand: lane l holds B[k = l >> 4][n = l & 15];
//   D: lane l holds column n = l & 15 and four rows drow(r, l), r < 4.
template <typename T> struct RwMfma;
template <> struct RwMfma<double> {
  typedef rw_f64x4 acc_t;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int r, int lane) { return (lane >> 4) + 4 * r; }
};
template <> struct RwMfma<float> {
  typedef rw_f32x4 acc_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int r, int lane) { return 4 * (lane >> 4) + r; }
};

// NCB = 16-column panel blocks per pass (PC = 16 * NCB accumulated columns); the host loops over passes.
// Compute phase on the matrix cores: M = 16 operator rows, K = 4 operator columns, N = 16 panel columns; both
// operands come out of the LDS tiles in exactly the MFMA layouts (that is what the LDS turn is for):
//   A operand  <- tile[rowblock*16 + (l & 15)][4*kk + (l >> 4)]         (ds_read_b64 / b32, conflict-free)
//   B operand  <- xt[4*kk + (l >> 4)][16*cb + (l & 15)]                  (panel sub-tile stored [column j][c])
// A 64-row wave keeps 4 x NCB accumulator tiles (16 / 32 values per lane); per 128 B of every row it issues
// (SEG/4) * (4 + NCB) LDS reads and (SEG/4) * 4 * NCB MFMAs — 8x fewer LDS cycles per flop than broadcasting the
// panel values to per-lane VALU FMAs (measured first: 3.9 TB/s at P = 16, 2.4 at P = 32, LDS-bound).
template <typename T, int NCB>
__global__ __launch_bounds__(256) void dense_rows_wide_kernel(
    const T* __restrict__ A, const T* __restrict__ X, T* __restrict__ Y, int M, int N, int pc, long lda, long sA,
    long ldx, long sX, long ldy, long sY, int row_blocks, int cols_per_split, long sSplit) {
  typedef typename Vec16<T>::type VT;
  typedef RwMfma<T> MM;
  typedef typename MM::acc_t acc_t;
  constexpr int VN = Vec16<T>::n;
  constexpr int PC = 16 * NCB;
  constexpr int SEG = RW_SEG_BYTES / (int)sizeof(T);          // columns per sub-tile (16 fp64 / 32 fp32)
  constexpr int LPR = RW_SEG_BYTES / 16;                      // lanes per row segment (8)
  constexpr int RPL = 64 / LPR;                               // rows per load instruction (8)
  constexpr int NLD = RW_ROWS / RPL;                          // load instructions per sub-tile (8)
  constexpr int XPITCH = PC * (int)sizeof(T) + 16;            // bytes per column j of the panel sub-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x / row_blocks);
  const int rb = __builtin_amdgcn_readfirstlane(blockIdx.x - b * row_blocks);
  const int row0 = (rb * 4 + wave) * RW_ROWS;
  if (row0 >= M) return;
  char* tile = smem + wave * rw_wave_lds(PC, (int)sizeof(T));
  char* xt = tile + RW_TILE_LDS;                              // panel sub-tile [j][c]
  const int c_lo = blockIdx.y * cols_per_split;
  int c_hi = c_lo + cols_per_split;
  c_hi = c_hi < N ? c_hi : N;
  const T* Ab = A + (long)b * sA;
  const T* Xb = X + (long)b * sX;
  // rows of this wave: row0 .. row0+63, clamped (duplicates are masked at the store)
  const int nrows = (M - row0) < RW_ROWS ? (M - row0) : RW_ROWS;
  const RwRsrc ra = rw_rsrc(Ab + (long)row0 * lda, ((long)(nrows - 1) * lda + N) * (long)sizeof(T));
  const int lrow = lane / LPR, lcol = lane - lrow * LPR;      // load phase: RPL rows x LPR lanes
  unsigned ld_row_off[NLD];
#pragma unroll
  for (int t = 0; t < NLD; ++t) {
    int r = t * RPL + lrow;
    r = r < nrows ? r : nrows - 1;
    ld_row_off[t] = (unsigned)r * (unsigned)(lda * (long)sizeof(T));
  }
  const unsigned st_off = (unsigned)lrow * RW_PITCH_BYTES + (unsigned)lcol * 16u;   // + t*RPL*PITCH per load
  // panel rows c = t*RPL + lrow (columns past the panel's width re-read its last column: never stored)
  constexpr int NLX = PC / RPL;
  long x_row_off[NLX];
#pragma unroll
  for (int t = 0; t < NLX; ++t) {
    int c = t * RPL + lrow;
    c = c < pc ? c : pc - 1;
    x_row_off[t] = (long)c * ldx;
  }
  acc_t acc[4][NCB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][cb][r] = T(0);
  const int mm = lane & 15, kq = lane >> 4;                   // MFMA operand coordinates of this lane

  for (int c0 = c_lo; c0 < c_hi; c0 += SEG) {
    // ---- load the 64 x SEG operator sub-tile (columns past the end read as zeros through the offset poison) and
    //      the PC x SEG panel sub-tile, park both in LDS ------------------------------------------------------
    const int colv = c0 + lcol * VN;
    const bool colok = colv < c_hi;
    VT a[NLD], xv[NLX];
#pragma unroll
    for (int t = 0; t < NLD; ++t) {
      const unsigned voff = colok ? (unsigned)colv * (unsigned)sizeof(T) + ld_row_off[t] : 0x7ffffff0u;
      a[t] = __builtin_bit_cast(VT, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)voff, 0, 2));
    }
#pragma unroll
    for (int t = 0; t < NLX; ++t) {
      if (colok) xv[t] = *reinterpret_cast<const VT*>(Xb + x_row_off[t] + colv);
      else {
#pragma unroll
        for (int q = 0; q < VN; ++q) xv[t][q] = T(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);          // everything in flight before the first store waits on a load
#pragma unroll
    for (int t = 0; t < NLX; ++t)
#pragma unroll
      for (int q = 0; q < VN; ++q)              // transposed: [column j][panel column c]
        *reinterpret_cast<T*>(xt + (unsigned)(lcol * VN + q) * XPITCH + (unsigned)(t * RPL + lrow) * sizeof(T)) =
            xv[t][q];
#pragma unroll
    for (int t = 0; t < NLD; ++t)
      *reinterpret_cast<VT*>(tile + st_off + (unsigned)(t * RPL) * RW_PITCH_BYTES) = a[t];
    // LDS writes of this wave must land before its reads in the operand layouts (same wave: program order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0)
    // ---- compute on the matrix cores ------------------------------------------------------------------
#pragma unroll
    for (int kk = 0; kk < SEG / 4; ++kk) {
      T bop[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        bop[cb] = *reinterpret_cast<const T*>(xt + (unsigned)(kk * 4 + kq) * XPITCH +
                                              (unsigned)(cb * 16 + mm) * sizeof(T));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const T aop = *reinterpret_cast<const T*>(tile + (unsigned)(i * 16 + mm) * RW_PITCH_BYTES +
                                                  (unsigned)(kk * 4 + kq) * sizeof(T));
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[i][cb] = MM::mma(aop, bop[cb], acc[i][cb]);
      }
    }
    // the next sub-tile overwrites the LDS tiles: all reads of this one have been consumed by the MFMAs above
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  T* Yb = Y + (long)blockIdx.y * sSplit + (long)b * sY + row0;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int c = cb * 16 + mm;
    if (c < pc) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lr = i * 16 + MM::drow(r, lane);
          if (lr < nrows) Yb[(long)c * ldy + lr] = acc[i][cb][r];
        }
    }
  }
}

// Y[b,c,i] = sum_s W[s,b,c,i]  (fixed order)
template <typename T>
__global__ __launch_bounds__(256) void rows_wide_fold(const T* __restrict__ W, T* __restrict__ Y, int M, int P,
                                                       int nsplit, long ldy, long sY, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over B*P*M
  if (idx >= total) return;
  const long per_b = (long)P * M;
  const long b = idx / per_b;
  const long rem = idx - b * per_b;
  const int c = (int)(rem / M);
  const int i = (int)(rem - (long)c * M);
  T s = T(0);
  for (int k = 0; k < nsplit; ++k) s += W[(long)k * total + idx];
  Y[b * sY + (long)c * ldy + i] = s;
}

static int rows_wide_nsplit(int B, int M, int N, int elem_size) {
  // enough workgroups to fill 256 CUs x 2 resident blocks a few times over; every split at least 4 sub-tiles long
  const long blocks = (long)B * ((M + 255) / 256);
  const int seg = RW_SEG_BYTES / elem_size;
  long want = (2048 + blocks - 1) / blocks;
  const long maxsplit = (N + 4L * seg - 1) / (4L * seg);
  if (want > maxsplit) want = maxsplit;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  return (int)want;
}

template <typename T>
static int rows_wide(const T* A, const T* X, T* Y, T* ws, long ws_elems, int B, int M, int N, int P, long lda,
                     long sA, long ldx, long sX, long ldy, long sY, hipStream_t st) {
  constexpr int VN = Vec16<T>::n;
  if ((N % VN) || (lda % VN) || (sA % VN) || ((uintptr_t)A & 15)) return XK_ERR_UNSUPPORTED;
  if ((ldx % VN) || (sX % VN) || ((uintptr_t)X & 15)) return XK_ERR_UNSUPPORTED;      // 16 B scalar loads of the panel
  const int row_blocks = (M + 255) / 256;
  const int seg = RW_SEG_BYTES / (int)sizeof(T);
  const int nsplit = rows_wide_nsplit(B, M, N, (int)sizeof(T));
  int cps = (N + nsplit - 1) / nsplit;
  cps = (cps + seg - 1) / seg * seg;                        // whole sub-tiles per split
  const int nsp = (N + cps - 1) / cps;
  int c0 = 0;
  while (c0 < P) {
    const int rem = P - c0;
    const int pcap = rem > 16 ? 32 : 16;
    const int pc = rem < pcap ? rem : pcap;
    T* out = Y + (long)c0 * ldy;
    long ldo = ldy, so = sY, ssplit = 0;
    if (nsp > 1) {
      const long total = (long)B * pc * M;
      if (ws == nullptr || ws_elems < total * nsp) return XK_ERR_ARG;
      out = ws; ldo = M; so = (long)pc * M; ssplit = total;
    }
    const dim3 grid((unsigned)((long)B * row_blocks), (unsigned)nsp);
    const T* Xc = X + (long)c0 * ldx;
#define XK_RW_LAUNCH(NCB)                                                                                     \
  {                                                                                                           \
    const size_t lds = 4 * (size_t)rw_wave_lds(16 * NCB, (int)sizeof(T));                                     \
    hipError_t e = hipFuncSetAttribute((const void*)dense_rows_wide_kernel<T, NCB>,                           \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                 \
    if (e != hipSuccess) return (int)e;                                                                       \
    hipLaunchKernelGGL((dense_rows_wide_kernel<T, NCB>), grid, dim3(256), lds, st, A, Xc, out, M, N, pc, lda,  \
                       sA, ldx, sX, ldo, so, row_blocks, cps, ssplit);                                        \
  }
    if (pcap == 32) XK_RW_LAUNCH(2) else XK_RW_LAUNCH(1)
#undef XK_RW_LAUNCH
    XK_LAUNCH_CHECK();
    if (nsp > 1) {
      const long total = (long)B * pc * M;
      hipLaunchKernelGGL((rows_wide_fold<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws,
                         Y + (long)c0 * ldy, M, pc, nsp, ldy, sY, total);
      XK_LAUNCH_CHECK();
    }
    c0 += pc;
  }
  return XK_OK;
}

}  // namespace xk

extern "C" {

long xk_dense_rows_wide_workspace_elems(int B, int M, int N, int P, int elem_size) {
  const int nsplit = xk::rows_wide_nsplit(B, M, N, elem_size);
  if (nsplit <= 1) return 0;
  const long pc = P > 32 ? 32 : P;
  return (long)nsplit * B * pc * M;
}

#define XK_DEFINE_ROWSWIDE(SUF, T)                                                                          \
  int xk_dense_rows_wide_##SUF(const T* A, const T* X, T* Y, T* ws, long ws_elems, int B, int M, int N,      \
                               int P, long lda, long sA, long ldx, long sX, long ldy, long sY,               \
                               void* stream) {                                                               \
    if (B < 0 || M < 0 || N < 0 || P < 0) return XK_ERR_ARG;                                                 \
    if (B == 0 || M == 0 || P == 0) return XK_OK;                                                            \
    if (N == 0) return XK_ERR_UNSUPPORTED;                                                                   \
    return xk::rows_wide<T>(A, X, Y, ws, ws_elems, B, M, N, P, lda, sA, ldx, sX, ldy, sY,                    \
                            (hipStream_t)stream);                                                            \
  }

XK_DEFINE_ROWSWIDE(f64, double)
XK_DEFINE_ROWSWIDE(f32, float)

}  // extern "C"
