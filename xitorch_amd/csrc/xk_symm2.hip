// xitorch_amd :: K1s2 — the symmetric-storage operator-panel product (see xk_symm.hip for the contract:
//        Y[b,c,:] = A_b X[b,c,:],  A_b = A_b^T exactly, only the upper triangle is read)
// with the ROW part on the matrix cores and the COLUMN part on the vector ALU, and no cross-lane reduction in
// the streaming loop.
//
// Why.  In xk_symm.hip a lane holds consecutive columns of ONE row (the coalesced load's natural layout), which
// makes the column part  y_J += A_IJ^T x_I  lane-local but forces a 64-lane reduction per row for the row part
// y_I += A_IJ x_J  (r01 PMC: 38 % of the wave time waits behind those reductions; 0.69-0.76 of the HBM roofline).
// Here every 64-row x 128 B sub-tile is turned through a per-wave LDS tile (K1wr's mechanism, xk_rowswide.hip)
// and read back with   lane l  <->  row i = l & 15 (of a 16-row block),  16 B chunk q = l >> 4 (+4):
//   * row part    : v_mfma_{f64,f32}_16x16x4 with M = 16 rows, K = 4 columns (one per chunk q — the contraction
//                   index may be permuted freely as long as both operands use the same permutation), N = 16
//                   panel columns (P of them non-zero).  The B operand — x_J at the lane's columns — is loop-
//                   invariant for the whole sweep of a column strip; the 16x16 result block of a sub-tile is
//                   complete for this strip and goes into an LDS row accumulator (ds_add, P of 16 lanes).
//   * column part : the lane's SEG/4 matrix values of row i times x_I[i][c] (3 ds_read_b128 from a [row][c]
//                   copy of the panel's tile rows) into per-lane accumulators acc_col[SEG/4][P] that live for the
//                   whole strip; the sum over the 16 row-lanes is taken ONCE per strip (DPP row rotations).
// Per 8 KB sub-tile and wave: 8 coalesced nt loads, 8 ds_write_b128, 8+12 ds_read_b128, 16 MFMA (fp64; 32 fp32),
// 96 (192) FMAs, 16 ds_add.
//
// Tiles: S2_TRH rows x 8 KB of columns (1024 fp64 / 2048 fp32), one block per tile; wave w sweeps the 128 B column
// strips w, w+W, ... downwards over the rows STRICTLY ABOVE the strip's diagonal block (rows past that point are
// replaced by zeros in the load itself, through an out-of-range offset), so the streaming loop carries no masks.
// The loads run two sub-tiles (16 KB per wave) ahead of the compute and the prefetch continues across strip
// boundaries.  The SEG x SEG diagonal blocks (0.2 % of the data) are done afterwards, one synchronous masked step
// per diagonal strip.  Partials and the fold are as in xk_symm.hip (rowP[J][c][i], colP[I][c][j]).
//
// Status (r02, fp64, P = 6, 32 x 16384^2, same box): 6.2 ms alone vs 6.0 ms for xk_symm.hip (5.5 vs 5.7 TB/s), 7.2 vs
// 6.15 ms on the 192 CUs the eigensolver's two-group pipeline gives the panel product — NOT the default
// (xk_dense_symm_set_variant(2) / XITORCH_AMD_K1S_VARIANT=2 selects it; tests run both).  What the experiments behind
// it established (scripts/micro/stream_patterns.hip, scripts/symm_pmc.sh):
//   * every walk of the tile (rows of 1-2 KB, 64 x 128 B sub-tiles down or along the rows) streams at 6.9-7.1 TB/s
//     with 8 waves x 16 KB in flight per CU when nothing is computed; the LDS turn costs nothing; loading straight
//     in the MFMA operand layout (16 rows x 64 B per instruction) costs 16 % (5.8 TB/s);
//   * a vector-memory instruction under a branch inside the streaming loop makes the compiler wait with vmcnt(0)
//     on every step (a full drain of the prefetch): the loop below issues exactly 8 loads per step, poisoned
//     offsets instead of branches — this alone was worth 7 %;
//   * the kernel is bound per CU, not by HBM: v_mfma_f64_16x16x4_f64 occupies the matrix pipe for ~120 cycles
//     (K1wr's plateau at 32 of them per 8 KB says the same), so the 16 per sub-tile — of whose 16 output columns
//     only P = 6 are used — cost 5x the 96 vector FMAs they replace: MFMA ~60 % busy at 5.5 TB/s with two
//     dependent chains and two waves per SIMD.  The matrix cores only pay from P >= 12 (K1w / K1wr).
#include "xk_common.h"

namespace xk {

#ifndef XK_S2_TRH
#define XK_S2_TRH 512
#endif
#ifndef XK_S2_WAVES
#define XK_S2_WAVES 8
#endif
constexpr int S2_TRH = XK_S2_TRH;              // rows per tile
constexpr int S2_SLAB_BYTES = 8192;            // bytes of one tile row
constexpr int S2_ROWS = 64;                    // rows per sub-tile
constexpr int S2_SEG_BYTES = 128;              // bytes per row and sub-tile (one cache line)
constexpr int S2_PITCH = S2_SEG_BYTES + 16;    // LDS pitch of the turned tile (odd multiple of 16 B)
constexpr int S2_TILE_LDS = S2_ROWS * S2_PITCH;
constexpr int S2_WAVES = XK_S2_WAVES;
constexpr int S2_THREADS = 64 * S2_WAVES;
constexpr unsigned S2_POISON = 0x7ffffff0u;    // byte offset beyond every descriptor: the load returns zeros

// pitch (bytes) of the [row][c] panel copy: an odd number of 16 B units -> 16 consecutive rows are conflict-free
__host__ __device__ constexpr int s2_xi_pitch(int p, int esize) {
  return (((p * esize + 15) / 16) | 1) * 16;
}
__host__ __device__ constexpr size_t s2_lds_bytes(int p, int esize) {
  return (size_t)S2_TRH * p * esize + (size_t)S2_TRH * s2_xi_pitch(p, esize) + (size_t)S2_WAVES * S2_TILE_LDS;
}

typedef __amdgpu_buffer_rsrc_t S2Rsrc;
typedef float s2_f32x4 __attribute__((ext_vector_type(4)));
typedef double s2_f64x4 __attribute__((ext_vector_type(4)));

template <typename T> struct S2Mfma;
template <> struct S2Mfma<double> {
  typedef s2_f64x4 acc_t;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int r, int lane) { return (lane >> 4) + 4 * r; }
};
template <> struct S2Mfma<float> {
  typedef s2_f32x4 acc_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int drow(int r, int lane) { return 4 * (lane >> 4) + r; }
};

// sum over the 16 lanes of a DPP row (every lane of the row gets the total): row_ror 8 / 4 / 2 / 1
template <int CTRL>
__device__ __forceinline__ double s2_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float s2_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <typename T>
__device__ __forceinline__ T s2_row16_sum(T v) {
  v += s2_dpp<0x128>(v);      // row_ror:8
  v += s2_dpp<0x124>(v);      // row_ror:4
  v += s2_dpp<0x122>(v);      // row_ror:2
  v += s2_dpp<0x121>(v);      // row_ror:1
  return v;
}

// U (1 or 2) 16-row blocks rb0 .. rb0+U-1 of the wave's LDS tile: U independent MFMA accumulation chains (row
// part), the FMAs of the column part issue underneath them.  lr0 / r0: tile-local / global row of the LDS tile's
// first row.  MASK: the block holds diagonal elements (strictly-lower ones dropped, the diagonal counted once).
template <typename T, int P, int U, bool MASK>
__device__ __forceinline__ void s2_group(const char* tile, const char* xI, T* rowacc, int rb0, int lr0, int r0,
                                         int j0, const T (&bJ)[S2_SEG_BYTES / (int)sizeof(T) / 4],
                                         T (&acc_col)[S2_SEG_BYTES / (int)sizeof(T) / 4][P], int lane) {
  typedef typename Vec16<T>::type VT;
  typedef S2Mfma<T> MM;
  typedef typename MM::acc_t acc_t;
  constexpr int VN = Vec16<T>::n;
  constexpr int NT4 = S2_SEG_BYTES / (int)sizeof(T) / 4;      // values per lane and 16-row block (4 fp64 / 8 fp32)
  constexpr int XP = s2_xi_pitch(P, (int)sizeof(T));
  constexpr int XV = (P * (int)sizeof(T) + 15) / 16;           // 16 B reads per panel row
  const int mi = lane & 15, q = lane >> 4;
  T a[U][NT4], xi[U][XV * VN];
  acc_t D[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int rb = rb0 + u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const VT v = *reinterpret_cast<const VT*>(tile + (unsigned)(rb * 16 + mi) * S2_PITCH +
                                                (unsigned)(q + 4 * h) * 16u);
#pragma unroll
      for (int e = 0; e < VN; ++e) a[u][h * VN + e] = v[e];
    }
#pragma unroll
    for (int w = 0; w < XV; ++w) {
      const VT v = *reinterpret_cast<const VT*>(xI + (unsigned)(lr0 + rb * 16 + mi) * XP + (unsigned)w * 16u);
#pragma unroll
      for (int e = 0; e < VN; ++e) xi[u][w * VN + e] = v[e];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) D[u][r] = T(0);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < VN; ++e) {
      const int t = h * VN + e;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        T ar = a[u][t];
        if (MASK) {
          const int row = r0 + (rb0 + u) * 16 + mi;
          const int col = j0 + (q + 4 * h) * VN + e;
          const T v = a[u][t];
          ar = col >= row ? v : T(0);
          a[u][t] = col > row ? v : T(0);
        }
        D[u] = MM::mma(ar, bJ[t], D[u]);
      }
    }
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int t = 0; t < NT4; ++t)
#pragma unroll
      for (int c = 0; c < P; ++c) acc_col[t][c] += a[u][t] * xi[u][c];
  // the column sums must be complete here: without the pin they are sunk below the (exec-masked) accumulator
  // update, behind the MFMA results, and the operands of the next group are live on top of these
#pragma unroll
  for (int t = 0; t < NT4; ++t)
#pragma unroll
    for (int c = 0; c < P; ++c) asm volatile("" : "+v"(acc_col[t][c]));
  // the row accumulator always has S2_TRH rows: rows past a ragged tile collect values that are never flushed
  if (mi < P) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = lr0 + (rb0 + u) * 16 + MM::drow(r, lane);
        __hip_atomic_fetch_add(&rowacc[lr * P + mi], D[u][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// iterator over the strips of one wave that have rows strictly above their diagonal block (all scalar)
struct S2Strip {
  int s, j0, main_rows, nsub, sub;
};
template <typename T>
__device__ __forceinline__ bool s2_next_strip(S2Strip& it, int col0, int row0, int tile_rows, int N) {
  constexpr int SEG = S2_SEG_BYTES / (int)sizeof(T);
  constexpr int NSTRIP = S2_SLAB_BYTES / S2_SEG_BYTES;
  for (;;) {
    it.s += S2_WAVES;
    if (it.s >= NSTRIP) return false;
    it.j0 = col0 + it.s * SEG;
    if (it.j0 >= N) {
      it.s = NSTRIP;
      return false;
    }
    const int top = it.j0 < row0 + tile_rows ? it.j0 : row0 + tile_rows;
    it.main_rows = top - row0;
    if (it.main_rows > 0) {
      it.nsub = (it.main_rows + S2_ROWS - 1) / S2_ROWS;
      it.sub = 0;
      return true;
    }
  }
}

// the 8 coalesced loads of one sub-tile (8 rows x 128 B each, non-temporal).  Everything is in the per-lane offset
// (the scalar offset of a raw buffer load is not range-checked): rows past a ragged tile and columns past the matrix
// fall outside the descriptor and read as zeros; in the last sub-tile of a strip the rows from the diagonal block
// on get the poison offset as well.
template <typename T>
__device__ __forceinline__ void s2_issue(typename Vec16<T>::type (&an)[8], const S2Rsrc rs,
                                         const unsigned (&rowpart)[8], unsigned colpart, int sub, int main_rows,
                                         unsigned ldab, int lrow) {
  typedef typename Vec16<T>::type VT;
  const unsigned cb = colpart + (unsigned)(sub * S2_ROWS) * ldab;
  const int left = main_rows - sub * S2_ROWS;          // rows of this sub-tile above the diagonal block (scalar)
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const unsigned off = (lrow < left - t * 8) ? rowpart[t] + cb : S2_POISON;
    an[t] = __builtin_bit_cast(VT, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 2));
  }
}

// one element through a buffer descriptor (an out-of-range offset reads as zero)
template <typename T> __device__ __forceinline__ T s2_ld_elem(const S2Rsrc r, unsigned off);
template <> __device__ __forceinline__ double s2_ld_elem<double>(const S2Rsrc r, unsigned off) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
}
template <> __device__ __forceinline__ float s2_ld_elem<float>(const S2Rsrc r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}

template <typename T, int P>
__global__ __launch_bounds__(S2_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void dense_symm2_tiles(
    const T* __restrict__ A, const T* __restrict__ X, T* __restrict__ rowP, T* __restrict__ colP, int ntiles,
    int N, long lda, long sA, long ldx, long sX, int NS, int NT) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  constexpr int SEG = S2_SEG_BYTES / (int)sizeof(T);           // columns per strip
  constexpr int SLAB = S2_SLAB_BYTES / (int)sizeof(T);         // columns per tile
  constexpr int NSTRIP = SLAB / SEG;                           // 64
  constexpr int NT4 = SEG / 4;
  constexpr int NLD = 8;                                       // load instructions per sub-tile (8 rows each)
  constexpr int XP = s2_xi_pitch(P, (int)sizeof(T));
  static_assert(S2_TRH % SEG == 0, "tile rows: whole diagonal blocks");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* rowacc = reinterpret_cast<T*>(smem);                                  // S2_TRH x P
  char* xI = smem + (size_t)S2_TRH * P * sizeof(T);                        // S2_TRH x XP bytes
  int b = blockIdx.x / ntiles;
  int I = 0, J = 0;
  {
    int rem = blockIdx.x - b * ntiles;
    for (;; ++I) {
      const int jmin = (I * S2_TRH) / SLAB;
      const int cnt = NS - jmin;
      if (rem < cnt) { J = jmin + rem; break; }
      rem -= cnt;
    }
  }
  b = __builtin_amdgcn_readfirstlane(b);
  I = __builtin_amdgcn_readfirstlane(I);
  J = __builtin_amdgcn_readfirstlane(J);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* tile = xI + (size_t)S2_TRH * XP + (size_t)wave * S2_TILE_LDS;
  const int row0 = I * S2_TRH;
  const int col0 = J * SLAB;
  const int tile_rows = (row0 + S2_TRH <= N ? S2_TRH : N - row0);
  const T* Ab = A + (long)b * sA;
  const T* Xb = X + (long)b * sX;
  const unsigned ldab = (unsigned)(lda * (long)sizeof(T));
  S2Rsrc rs;
  {
    const uint64_t v = reinterpret_cast<uint64_t>(Ab + (long)row0 * lda);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    void* base = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    // (S2_TRH + 64 rows of a 2^20-column fp64 matrix stay below the poison offset)
    const long bytes = ((long)(tile_rows - 1) * lda + N) * (long)sizeof(T);
    const uint32_t nrec = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0x7fffffe0L ? 0x7fffffe0L : bytes));
    rs = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)nrec, 0x00020000);
  }
  const int lrow = lane >> 3, lcol = lane & 7;                // load phase: 8 rows x 8 lanes of 16 B
  const unsigned st_off = (unsigned)lrow * S2_PITCH + (unsigned)lcol * 16u;
  const int mi = lane & 15, q = lane >> 4;
  T* cp = colP + (((long)b * NT + I) * P) * (long)N;
  unsigned rowpart[NLD];
#pragma unroll
  for (int t = 0; t < NLD; ++t) rowpart[t] = (unsigned)(t * 8 + lrow) * ldab;

  // ---- main sweep ----------------------------------------------------------------------------------------------
  // The loads run two sub-tile steps ahead of the compute, across strip boundaries.  The steady loop must not
  // contain a vector-memory instruction under a branch: the compiler's s_waitcnt insertion falls back to vmcnt(0)
  // — a full drain of the prefetch on every step — as soon as the order of the outstanding loads differs between
  // paths.  Hence: every step issues exactly its 8 loads (offsets poisoned when there is nothing to fetch), every
  // strip runs an EVEN number of steps (an odd one gets a step of zeros), the panel values of the next strip are
  // requested at the start of the current one, and the stores of a strip's column sums sit between the loops.
  S2Strip pr, co, nx;                   // producer (loads), consumer (compute), the consumer's next strip
  pr.s = wave - S2_WAVES;
  bool pvalid = s2_next_strip<T>(pr, col0, row0, tile_rows, N);
  co = pr;
  bool cvalid = pvalid;
  unsigned pcol = S2_POISON;
  int prows = 0, pnst = 0;              // rows above the diagonal block / steps (even) of the producer's strip
  VT an0[NLD], an1[NLD];
  T bJ[NT4], bJn[NT4], acc_col[NT4][P];
  S2Rsrc rx;                            // the panel through a descriptor: absent values are poisoned offsets, not branches
  {
    const uint64_t v = reinterpret_cast<uint64_t>(Xb);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    void* base = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
    const long bytes = ((long)(P - 1) * ldx + N) * (long)sizeof(T);
    const uint32_t nrec = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0x7fffffe0L ? 0x7fffffe0L : bytes));
    rx = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)nrec, 0x00020000);
  }
  const unsigned xrow = mi < P ? (unsigned)mi * (unsigned)(ldx * (long)sizeof(T)) : S2_POISON;
  // panel values X[mi][j0 + lane's columns] of strip j0 (valid == false: zeros)
#define XK_S2_LOAD_BJ(DST, J0, VALID)                                                                       \
  _Pragma("unroll") for (int h = 0; h < 2; ++h) _Pragma("unroll") for (int e = 0; e < VN; ++e) {            \
    const int col = (J0) + (q + 4 * h) * VN + e;                                                            \
    const unsigned off = ((VALID) && mi < P && col < N) ? xrow + (unsigned)col * (unsigned)sizeof(T)        \
                                                        : S2_POISON;                                        \
    DST[h * VN + e] = s2_ld_elem<T>(rx, off);                                                               \
  }
#define XK_S2_PENTER()                                                                                    \
  {                                                                                                       \
    pcol = (pvalid && (pr.j0 + lcol * VN) < N) ? (unsigned)(pr.j0 + lcol * VN) * (unsigned)sizeof(T)      \
                                               : S2_POISON;                                               \
    prows = pvalid ? pr.main_rows : 0;                                                                    \
    pnst = pvalid ? ((pr.nsub + 1) & ~1) : 0x40000000;                                                    \
  }
#define XK_S2_PRODUCE(BUF)                                                                \
  {                                                                                       \
    s2_issue<T>(BUF, rs, rowpart, pcol, pr.sub, prows, ldab, lrow);                       \
    if (++pr.sub == pnst) {                                                               \
      pvalid = s2_next_strip<T>(pr, col0, row0, tile_rows, N);                            \
      XK_S2_PENTER()                                                                      \
    }                                                                                     \
  }
  XK_S2_LOAD_BJ(bJn, co.j0, cvalid)
  XK_S2_PENTER()
  XK_S2_PRODUCE(an0)
  XK_S2_PRODUCE(an1)
  // the block's LDS set-up runs UNDER the first 16 KB of loads per wave (they do not depend on it): zero row
  // accumulator, [row][c] copy of the panel's tile rows (zeros past the matrix)
  for (int idx = threadIdx.x; idx < S2_TRH * P; idx += S2_THREADS) rowacc[idx] = T(0);
  for (int idx = threadIdx.x; idx < S2_TRH * P; idx += S2_THREADS) {
    const int c = idx / S2_TRH, r = idx - c * S2_TRH;
    const T v = r < tile_rows ? Xb[(long)c * ldx + row0 + r] : T(0);
    *reinterpret_cast<T*>(xI + (unsigned)r * XP + (unsigned)c * sizeof(T)) = v;
  }
  __syncthreads();
#define XK_S2_STEP(BUF, SUB)                                                                                   \
  {                                                                                                            \
    /* the previous sub-tile's LDS reads are consumed (MFMA / FMA operands) before these writes are issued */  \
    _Pragma("unroll") for (int t = 0; t < NLD; ++t)                                                            \
        *reinterpret_cast<VT*>(tile + st_off + (unsigned)(t * 8) * S2_PITCH) = BUF[t];                         \
    XK_S2_PRODUCE(BUF)                                                                                         \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                     \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): the tile is in LDS (same wave: program order) */        \
    {                                                                                                          \
      int lr0 = (SUB) * S2_ROWS; /* (the step of zeros of an odd strip lands on the tile's last rows) */       \
      lr0 = lr0 < S2_TRH - S2_ROWS ? lr0 : S2_TRH - S2_ROWS;                                                   \
      s2_group<T, P, 2, false>(tile, xI, rowacc, 0, lr0, row0 + lr0, co.j0, bJ, acc_col, lane);                \
      s2_group<T, P, 2, false>(tile, xI, rowacc, 2, lr0, row0 + lr0, co.j0, bJ, acc_col, lane);                \
    }                                                                                                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                     \
  }
  while (cvalid) {
    // strip start: the prefetched panel values become the MFMA B operand, those of the next strip are requested
#pragma unroll
    for (int t = 0; t < NT4; ++t) bJ[t] = bJn[t];
    nx = co;
    const bool nvalid = s2_next_strip<T>(nx, col0, row0, tile_rows, N);
    XK_S2_LOAD_BJ(bJn, nx.j0, nvalid)
#pragma unroll
    for (int t = 0; t < NT4; ++t)
#pragma unroll
      for (int c = 0; c < P; ++c) acc_col[t][c] = T(0);
    const int nst = (co.nsub + 1) & ~1;
    for (int sub = 0; sub < nst; sub += 2) {
      XK_S2_STEP(an0, sub)
      XK_S2_STEP(an1, sub + 1)
    }
    // strip end: column sums over the 16 row-lanes, lane mi == c stores panel column c
#pragma unroll
    for (int t = 0; t < NT4; ++t) {
      const int col = co.j0 + (q + 4 * (t / VN)) * VN + (t % VN);
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const T sum = s2_row16_sum(acc_col[t][c]);
        if (mi == c && col < N) cp[(long)c * N + col] = sum;
      }
    }
    co = nx;
    cvalid = nvalid;
  }
#undef XK_S2_STEP
#undef XK_S2_PRODUCE
#undef XK_S2_PENTER

  // ---- diagonal blocks of this wave's strips (rows j0 .. j0+SEG-1): one synchronous masked step each ---------------
  for (int s = wave; s < NSTRIP; s += S2_WAVES) {
    const int j0 = col0 + s * SEG;
    if (j0 >= N) break;
    if (j0 < row0 || j0 >= row0 + tile_rows) continue;
    const int lr0 = j0 - row0;
    const bool colok = (j0 + lcol * VN) < N;
    const unsigned dcol = (unsigned)(j0 + lcol * VN) * (unsigned)sizeof(T) + (unsigned)lr0 * ldab;
    VT ad[SEG / 8];
#pragma unroll
    for (int t = 0; t < SEG / 8; ++t)
      ad[t] = __builtin_bit_cast(
          VT, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(colok ? rowpart[t] + dcol : S2_POISON), 0, 2));
    XK_S2_LOAD_BJ(bJ, j0, true)
#pragma unroll
    for (int t = 0; t < NT4; ++t)
#pragma unroll
      for (int c = 0; c < P; ++c) acc_col[t][c] = T(0);
#pragma unroll
    for (int t = 0; t < SEG / 8; ++t) *reinterpret_cast<VT*>(tile + st_off + (unsigned)(t * 8) * S2_PITCH) = ad[t];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    s2_group<T, P, SEG / 16, true>(tile, xI, rowacc, 0, lr0, j0, j0, bJ, acc_col, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool fresh = (lr0 == 0);       // no rows above the diagonal block: the main sweep wrote nothing here
#pragma unroll
    for (int t = 0; t < NT4; ++t) {
      const int col = j0 + (q + 4 * (t / VN)) * VN + (t % VN);
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const T sum = s2_row16_sum(acc_col[t][c]);
        if (mi == c && col < N) {
          T* dst = cp + (long)c * N + col;
          *dst = fresh ? sum : *dst + sum;       // the same lane stored the main sweep's sum: program order
        }
      }
    }
  }
#undef XK_S2_LOAD_BJ
  __syncthreads();
  T* rp = rowP + (((long)b * NS + J) * P) * (long)N;
  for (int idx = threadIdx.x; idx < tile_rows * P; idx += S2_THREADS) {
    const int c = idx / tile_rows, lr = idx - c * tile_rows;
    __builtin_nontemporal_store(rowacc[lr * P + c], &rp[(long)c * N + row0 + lr]);   // written once, read once
  }
}

// y[c][n] = sum over the column slabs J that own row tile n/TRH of rowP[J][c][n]
//         + sum over the row tiles I whose first row is <= the first column of n's strip of colP[I][c][n]
template <typename T>
__global__ __launch_bounds__(256) void symm2_fold(const T* __restrict__ rowP, const T* __restrict__ colP,
                                                   T* __restrict__ Y, int N, int P, int NS, int NT, long ldy,
                                                   long sY, long total) {
  constexpr int SEG = S2_SEG_BYTES / (int)sizeof(T);
  constexpr int SLAB = S2_SLAB_BYTES / (int)sizeof(T);
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over B*P*N
  if (idx >= total) return;
  const long per_b = (long)P * N;
  const long b = idx / per_b;
  const long rem = idx - b * per_b;
  const int c = (int)(rem / N);
  const int n = (int)(rem - (long)c * N);
  const int It = n / S2_TRH;
  const int Jfirst = (It * S2_TRH) / SLAB;
  T s = T(0);
  for (int J = Jfirst; J < NS; ++J) s += __builtin_nontemporal_load(&rowP[(((long)b * NS + J) * P + c) * (long)N + n]);
  const int Imax = ((n / SEG) * SEG) / S2_TRH;
  for (int I = 0; I <= Imax && I < NT; ++I)
    s += __builtin_nontemporal_load(&colP[(((long)b * NT + I) * P + c) * (long)N + n]);
  Y[b * sY + (long)c * ldy + n] = s;
}

template <typename T>
static long symm2_ws_elems(int B, int N, int P) {
  constexpr int SLAB = S2_SLAB_BYTES / (int)sizeof(T);
  const long NS = (N + SLAB - 1) / SLAB, NT = (N + S2_TRH - 1) / S2_TRH;
  const long pc = P > 6 ? 6 : P;
  return (long)B * (NS + NT) * pc * N;
}

template <typename T, int P>
static int symm2_launch_tiles(const T* A, const T* X, T* rowP, T* colP, int B, int nt, int N, long lda, long sA,
                              long ldx, long sX, int NS, int NT, hipStream_t st) {
  const size_t lds = s2_lds_bytes(P, (int)sizeof(T));
  hipError_t e = hipFuncSetAttribute((const void*)dense_symm2_tiles<T, P>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((dense_symm2_tiles<T, P>), dim3((unsigned)((long)B * nt)), dim3(S2_THREADS), lds, st, A, X,
                     rowP, colP, nt, N, lda, sA, ldx, sX, NS, NT);
  return XK_OK;
}

// phase: 0 = tiles + fold, 1 = tiles only, 2 = fold only (see xk_symm.hip)
template <typename T>
int symm2_launch(const T* A, const T* X, T* Y, T* ws, long ws_elems, int B, int N, int P, long lda, long sA,
                 long ldx, long sX, long ldy, long sY, void* stream, int phase) {
  constexpr int VN = Vec16<T>::n;
  constexpr int SLAB = S2_SLAB_BYTES / (int)sizeof(T);
  if ((N % VN) || (lda % VN) || (sA % VN) || (ldx % VN) || (sX % VN) || ((uintptr_t)A & 15) ||
      ((uintptr_t)X & 15) || ((uintptr_t)ws & 15))
    return XK_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int NS = (N + SLAB - 1) / SLAB, NT = (N + S2_TRH - 1) / S2_TRH;
  int nt = 0;
  for (int I = 0; I < NT; ++I) nt += NS - (I * S2_TRH) / SLAB;
  int c0 = 0;
  while (c0 < P) {
    const int pc = (P - c0) >= 6 ? 6 : (P - c0);
    const long nrow = (long)B * NS * pc * N, ncol = (long)B * NT * pc * N;
    if (ws_elems < nrow + ncol) return XK_ERR_ARG;
    T* rowP = ws;
    T* colP = ws + nrow;
    const T* Xc = X + (long)c0 * ldx;
    if (phase != 2) {
      int rc = XK_OK;
      switch (pc) {
        case 1: rc = symm2_launch_tiles<T, 1>(A, Xc, rowP, colP, B, nt, N, lda, sA, ldx, sX, NS, NT, st); break;
        case 2: rc = symm2_launch_tiles<T, 2>(A, Xc, rowP, colP, B, nt, N, lda, sA, ldx, sX, NS, NT, st); break;
        case 3: rc = symm2_launch_tiles<T, 3>(A, Xc, rowP, colP, B, nt, N, lda, sA, ldx, sX, NS, NT, st); break;
        case 4: rc = symm2_launch_tiles<T, 4>(A, Xc, rowP, colP, B, nt, N, lda, sA, ldx, sX, NS, NT, st); break;
        case 5: rc = symm2_launch_tiles<T, 5>(A, Xc, rowP, colP, B, nt, N, lda, sA, ldx, sX, NS, NT, st); break;
        default: rc = symm2_launch_tiles<T, 6>(A, Xc, rowP, colP, B, nt, N, lda, sA, ldx, sX, NS, NT, st); break;
      }
      if (rc != XK_OK) return rc;
      XK_LAUNCH_CHECK();
    }
    if (phase != 1) {
      const long total = (long)B * pc * N;
      hipLaunchKernelGGL((symm2_fold<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, rowP, colP,
                         Y + (long)c0 * ldy, N, pc, NS, NT, ldy, sY, total);
      XK_LAUNCH_CHECK();
    }
    c0 += pc;
  }
  return XK_OK;
}

template int symm2_launch<double>(const double*, const double*, double*, double*, long, int, int, int, long, long,
                                  long, long, long, long, void*, int);
template int symm2_launch<float>(const float*, const float*, float*, float*, long, int, int, int, long, long, long,
                                 long, long, long, void*, int);

long symm2_workspace_elems(int B, int N, int P, int elem_size) {
  return elem_size == 8 ? symm2_ws_elems<double>(B, N, P) : symm2_ws_elems<float>(B, N, P);
}

}  // namespace xk
