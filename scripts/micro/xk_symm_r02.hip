// xitorch_amd :: K1s — operator-panel product for EXACTLY symmetric dense storage, reading only the
// upper triangle:   Y[b,c,:] = A_b X[b,c,:],  A_b = A_b^T.
//
// The general K1 (xk_dense.hip) streams all N^2 elements per panel product.  A symmetric matrix
// carries every off-diagonal value twice, so here each tile on/above the diagonal is read ONCE
// and used for both of its contributions
//        y_I += A_IJ x_J      (row part)          y_J += A_IJ^T x_I     (column part)
// which halves the HBM traffic of the eigensolver's panel product (symeig operators are always
// Hermitian: xitorch/linalg/symeig.py:103).  Opt-in: the caller asserts exact symmetry of the
// storage (MatrixLinearOperator(..., symmetric_storage=True)); for merely "allclose" symmetric
// input (LinearOperator.m's check, linop.py:97-105) the general kernel keeps the reference's
// full-matrix semantics.
//
// Mapping.  Tiles of TRH=1024 rows x 1024 columns (fp64; 2048 columns fp32).  One 256-thread block
// per tile; the 4 waves own 4 x 256 columns (lane: two 16 B vectors, so each load instruction is a
// contiguous 1 KB and the cross-lane reduction is amortised over twice the data), and walk down the
// tile's rows in chunks of 8 (16 buffer loads = 16 KB in flight per wave; descriptor + one loop-invariant
// lane offset + scalar row offset, no 64-bit vector address arithmetic):
//   * column part: per-lane register accumulators acc_col[P][VN] over the whole tile (panel
//     values x_I are wave-uniform scalar loads);
//   * row part: per-lane products a[r]*x_J (x_J held in registers for the tile), folded across
//     the 64 lanes by an eager transposing tree (half-exchange swaps first), then added into an LDS
//     accumulator rowacc[1024][P] (ds_add_f64; 48 KB for P=6);
//   * tile results go to partial buffers  rowP[J][c][i]  /  colP[I][c][j]  (one slot per column slab /
//     row tile) and a fold kernel adds, for every output element, exactly the slots that exist:
//        y[c][n] = sum_{J >= 2*(n>>10)} rowP[J][c][n] + sum_{I <= n>>10} colP[I][c][n].
//   Tiles crossing the diagonal mask the strictly-lower elements (and count the diagonal once); each of
//   their waves stops at its own last column.  Lanes past the last column of a ragged matrix read through
//   an out-of-range offset (hardware returns zeros).
//
// Register budget (fp64, P=6): 203 VGPRs -> 2 waves per SIMD; the panel/accumulator registers (96) cannot be
// shared between waves, so the third wave (<=168 VGPRs) is out of reach and the chunk depth is what keeps
// enough bytes in flight.
//
// What was measured (fp64, P = 6, N = 16384, half batch of 32 per launch):
//   alone on the GPU — this kernel 5.60-5.76 ms; the same with the chunk's 16 loads issued up front instead of
//     the rolling ring 5.58-5.74; the first version (4-row chunks, per-lane 64-bit addresses, generic
//     reduction, 188 VGPRs) 5.53-5.70;
//   inside the eigensolver's two-group pipeline, i.e. sharing HBM and CUs with the other group's small kernels
//     — ring 6.03-6.12 ms (222-225 ms per symeig call), loads-up-front 6.32-6.41 (232-235), first version
//     6.78 (248.5).  The ring keeps 12-16 KB per wave in flight at all times and is what holds the request
//     stream up under contention, so it is the shipped form although it wins nothing in isolation.
//   tile orders other than row-tile-major (XCD-contiguous runs, batch-fastest, short-tiles-first, slab-major,
//     member pairs interleaved): 0-7 % slower; balancing the diagonal tiles across waves: 1 % slower.
//
// Traffic per launch: B*N^2*s/2 (+2 % for the crossing tiles) + 2 * B*(NS+NT)*P*N*s of partials
// (2.5 %) — vs B*N^2*s for the general kernel.
#include "xk_common.h"

namespace xk {

constexpr int SYMM_TRH = 1024;   // rows per tile

#ifndef XK_SYMM_NU
#define XK_SYMM_NU 2
#endif
#ifndef XK_SYMM_WPE
#define XK_SYMM_WPE 2
#endif
#ifndef XK_SYMM_EARLY
#define XK_SYMM_EARLY 1
#endif
constexpr int SYMM_NU = XK_SYMM_NU;      // 16 B vectors per lane per row: a wave spans NU x 64 x VN columns

// The operator tile is read through a buffer descriptor (base = first row of the tile, wave-uniform):
// every load is  descriptor + per-lane column offset (one VGPR, loop-invariant) + scalar row offset,
// so the streaming loop carries no 64-bit VGPR address arithmetic.  aux = 2: non-temporal.
typedef __amdgpu_buffer_rsrc_t TileRsrc;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <typename VT>
__device__ __forceinline__ VT ld_tile(const TileRsrc rsrc, unsigned lane_off, unsigned row_off) {
  const u4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)row_off, 2);
  return __builtin_bit_cast(VT, raw);
}

template <typename T>
__device__ __forceinline__ TileRsrc make_tile_rsrc(const T* tile_base, long bytes) {
  const uint64_t v = reinterpret_cast<uint64_t>(tile_base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  void* base = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  const uint32_t nrec = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xffffffffL ? 0xffffffffL : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)nrec, 0x00020000);
}

// ---------------------------------------------------------------------------------------------
// 8-row chunk (16 loads = 16 KB in flight per wave).  At 188+ VGPRs the kernel runs 2 waves per
// SIMD whatever the chunk size, so the only way to keep more bytes in flight is a deeper chunk; the
// row sums are folded EAGERLY so that the 48 partial sums never coexist:
//   rows (2h, 2h+1)  -> half-exchange over lane bit 5      (6 values per row pair)
//   row pairs        -> half-exchange over lane bit 4      (6 values per 4 rows)
//   the two 4-groups -> select + xor-8 shuffle             (6 values per 8 rows)
//   panel columns    -> (even P) select + xor-4 shuffle, then xor-2 / xor-1 butterflies
// afterwards lane l holds, for row r = 4*bit3 + 2*bit4 + bit5 of the chunk, the complete sums of
// columns c = 2w + bit2 (w < P/2); lanes with bits 1,0 clear add them into the LDS accumulator.
// ---------------------------------------------------------------------------------------------
constexpr int SYMM_R = 8;       // rows per chunk

template <typename T, int P, bool CROSSING, bool TAIL>
__device__ __forceinline__ void symm_chunk8(
    typename Vec16<T>::type (&a)[SYMM_R][SYMM_NU], const TileRsrc Ab, const T* __restrict__ Xb, unsigned lda,
    long ldx, int N, int i0, int i_end,
    const int (&jj)[SYMM_NU], const unsigned (&joff)[SYMM_NU], int row_tile0,
    typename Vec16<T>::type (&acc_col)[SYMM_NU][P], const typename Vec16<T>::type (&xJ)[SYMM_NU][P],
    T* rowacc, int lane) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  constexpr int R = SYMM_R, NU = SYMM_NU;
  const int i_last = i_end - 1;
  T L2[2][P];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    T L1[2][P];
    // the panel values of a row group are fetched when the group starts (not all 8 rows up front: 96 SGPRs)
    if (g > 0) asm volatile("" ::: "memory");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      T xi[P][2];
#pragma unroll
      for (int c = 0; c < P; ++c)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          int row = i0 + 4 * g + 2 * h + q;
          if (TAIL) row = row < i_last ? row : i_last;
          xi[c][q] = Xb[(long)c * ldx + row];
        }
      T s[2][P];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = 4 * g + 2 * h + q;
        const int row = i0 + r;
#pragma unroll
        for (int c = 0; c < P; ++c) s[q][c] = T(0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          VT ar = a[r][u], ac = a[r][u];
          if (CROSSING) {
#pragma unroll
            for (int v = 0; v < VN; ++v) {
              if (jj[u] + v < row) { ar[v] = T(0); ac[v] = T(0); }
              if (jj[u] + v == row) ac[v] = T(0);
            }
          }
          if (TAIL) {
            if (row >= i_end) {
#pragma unroll
              for (int v = 0; v < VN; ++v) { ar[v] = T(0); ac[v] = T(0); }
            }
          }
#pragma unroll
          for (int c = 0; c < P; ++c)
#pragma unroll
            for (int v = 0; v < VN; ++v) {
              acc_col[u][c][v] += ac[v] * xi[c][q];
              s[q][c] += ar[v] * xJ[u][c][v];
            }
        }
      }
#pragma unroll
      for (int c = 0; c < P; ++c) L1[h][c] = swap_add32(s[0][c], s[1][c]);
      // the column sums of this row pair must be complete here: without the pin the optimiser sinks all
      // of them below the reduction, which keeps the whole 8-row chunk of matrix data live until then
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int c = 0; c < P; ++c) asm volatile("" : "+v"(acc_col[u][c]));
      // rolling prefetch: the two rows just consumed are refilled with the rows 8 further down, so the wave
      // always has ~6 row pairs of loads in flight while it computes (rows past the end re-read the last one)
      if (!TAIL) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          int row = i0 + R + 4 * g + 2 * h + q;
          row = row < i_last ? row : i_last;
#pragma unroll
          for (int u = 0; u < NU; ++u)
            // (crossing tiles: a lane whose columns all lie strictly below the row fetches nothing — its values
            //  would be masked to zero anyway; the out-of-range offset returns the zeros without the traffic)
            a[4 * g + 2 * h + q][u] = ld_tile<VT>(Ab, (!CROSSING || jj[u] + VN - 1 >= row) ? joff[u] : 0x7ffffff0u,
                                                  (unsigned)(row - row_tile0) * lda);
        }
      }
      __builtin_amdgcn_sched_barrier(0);      // keep the row pairs in program order (bounded live ranges)
    }
#pragma unroll
    for (int c = 0; c < P; ++c) L2[g][c] = swap_add16(L1[0][c], L1[1][c]);
    __builtin_amdgcn_sched_barrier(0);
  }
  T L3[P];
  {
    const bool hi = (lane & 8) != 0;
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const T keep = hi ? L2[1][c] : L2[0][c];
      const T send = hi ? L2[0][c] : L2[1][c];
      L3[c] = keep + lane_partner<8>(send);
    }
  }
  const int r = ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 5) & 1);
  const int lrow = i0 + r - row_tile0;
  const bool rowok = !TAIL || (i0 + r < i_end);
  if (P % 2 == 0) {
    constexpr int PH = P / 2 > 0 ? P / 2 : 1;
    T L4[PH];
    const bool hi = (lane & 4) != 0;
#pragma unroll
    for (int w = 0; w < P / 2; ++w) {
      const T keep = hi ? L3[2 * w + 1] : L3[2 * w];
      const T send = hi ? L3[2 * w] : L3[2 * w + 1];
      L4[w] = keep + lane_partner<4>(send);
    }
#pragma unroll
    for (int w = 0; w < P / 2; ++w) {
      L4[w] += lane_partner<2>(L4[w]);
      L4[w] += lane_partner<1>(L4[w]);
    }
    if ((lane & 3) == 0 && rowok) {
#pragma unroll
      for (int w = 0; w < P / 2; ++w)
        __hip_atomic_fetch_add(&rowacc[lrow * P + 2 * w + (hi ? 1 : 0)], L4[w], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  } else {
#pragma unroll
    for (int c = 0; c < P; ++c) {
      L3[c] += lane_partner<4>(L3[c]);
      L3[c] += lane_partner<2>(L3[c]);
      L3[c] += lane_partner<1>(L3[c]);
    }
    if ((lane & 7) == 0 && rowok) {
#pragma unroll
      for (int c = 0; c < P; ++c)
        __hip_atomic_fetch_add(&rowacc[lrow * P + c], L3[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

template <typename T, int P, bool CROSSING>
__device__ __forceinline__ void symm_tile_rows(
    const TileRsrc Ab, const T* __restrict__ Xb, unsigned lda, long ldx, int N, int i_begin, int i_end,
    const int (&jj)[SYMM_NU], const unsigned (&joff)[SYMM_NU], int row_tile0,
    typename Vec16<T>::type (&acc_col)[SYMM_NU][P], const typename Vec16<T>::type (&xJ)[SYMM_NU][P],
    T* rowacc, int lane) {
  typedef typename Vec16<T>::type VT;
  const bool any = i_begin < i_end;          // (wave-uniform) a wave right of a crossing tile's diagonal has no rows
  const int full_end = i_begin + ((i_end - i_begin) / SYMM_R) * SYMM_R;
  const int i_last = i_end - 1;
  VT a[SYMM_R][SYMM_NU];                     // ring of 8 rows, refilled pair by pair inside the chunks
  // (unconditional: a branch around the fill would leave the compiler without the order of the outstanding loads at
  // the loop head and turn every wait of the ring into vmcnt(0); a wave without rows re-reads the tile's first row)
#pragma unroll
  for (int r = 0; r < SYMM_R; ++r) {
    int row = i_begin + r;
    row = row < i_last ? row : i_last;
    row = row > row_tile0 ? row : row_tile0;
#pragma unroll
    for (int u = 0; u < SYMM_NU; ++u)
      a[r][u] = ld_tile<VT>(Ab, (!CROSSING || jj[u] + Vec16<T>::n - 1 >= row) ? joff[u] : 0x7ffffff0u,
                            (unsigned)(row - row_tile0) * lda);
  }
  // the block's LDS set-up runs UNDER the first 16 KB of loads (they do not depend on it): every wave passes
  // here exactly once, whichever of the two instantiations it took
#if XK_SYMM_EARLY
  for (int idx = threadIdx.x; idx < SYMM_TRH * P; idx += 256) rowacc[idx] = T(0);
  __syncthreads();
#endif
  if (!any) return;
  for (int i0 = i_begin; i0 < full_end; i0 += SYMM_R)
    symm_chunk8<T, P, CROSSING, false>(a, Ab, Xb, lda, ldx, N, i0, i_end, jj, joff, row_tile0, acc_col, xJ, rowacc, lane);
  if (full_end < i_end)
    symm_chunk8<T, P, CROSSING, true>(a, Ab, Xb, lda, ldx, N, full_end, i_end, jj, joff, row_tile0, acc_col, xJ, rowacc, lane);
}

template <typename T, int P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(XK_SYMM_WPE))) void dense_symm_tiles(
    const T* __restrict__ A, const T* __restrict__ X, T* __restrict__ rowP, T* __restrict__ colP, int ntiles,
    int N, long lda, long sA, long ldx, long sX, int NS, int NT, int flags) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  constexpr int NU = SYMM_NU;
  constexpr int WCOLS = NU * 64 * VN;          // columns per wave
  constexpr int SLAB = 4 * WCOLS;              // columns per block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* rowacc = reinterpret_cast<T*>(smem);                   // SYMM_TRH x P
  // tile list in row-tile-major order: row tile I owns the column slabs J >= (I*TRH)/SLAB
  int b = blockIdx.x / ntiles;
  int I = 0, J = 0;
  {
    int rem = blockIdx.x - b * ntiles;
    for (;; ++I) {
      const int jmin = (I * SYMM_TRH) / SLAB;
      const int cnt = NS - jmin;
      if (rem < cnt) { J = jmin + rem; break; }
      rem -= cnt;
    }
  }
  // the integer divisions above run on the vector ALU: pin their (wave-uniform) results in SGPRs so that
  // everything derived from them (row pointers, loop bounds, panel addresses) is scalar arithmetic
  b = __builtin_amdgcn_readfirstlane(b);
  I = __builtin_amdgcn_readfirstlane(I);
  J = __builtin_amdgcn_readfirstlane(J);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: scalar row pointers
  const int row0 = I * SYMM_TRH;
  const int col0 = J * SLAB;
  int jj[NU];
  unsigned joff[NU];
  bool colok[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    jj[u] = col0 + wave * WCOLS + u * 64 * VN + lane * VN;      // each load instruction: 1 KB contiguous
    colok[u] = jj[u] < N;
    // lanes past the last column get a byte offset beyond the descriptor's range: the hardware bounds
    // check returns zeros for them (no branch, no select), so they add nothing to either sum
    joff[u] = colok[u] ? (unsigned)jj[u] * (unsigned)sizeof(T) : 0x7ffffff0u;
  }
  const T* Ab = A + (long)b * sA;
  const T* Xb = X + (long)b * sX;
#if !XK_SYMM_EARLY      /* A/B only: LDS set-up and barrier before the first load is issued */
  for (int idx = threadIdx.x; idx < SYMM_TRH * P; idx += 256) rowacc[idx] = T(0);
  __syncthreads();
#endif
  // rows of this tile that can hold an element on/above the diagonal: i <= last column of the slab
  int i_end = row0 + SYMM_TRH;
  const int col_last = col0 + SLAB - 1;
  if (i_end > col_last + 1) i_end = col_last + 1;
  if (i_end > N) i_end = N;
  const bool crossing = (i_end - 1 >= col0);   // some row index reaches the first column: mask needed
  const int tile_rows = (row0 + SYMM_TRH <= N ? SYMM_TRH : N - row0);
  const unsigned ldab = (unsigned)(lda * (long)sizeof(T));
  const TileRsrc tile = make_tile_rsrc(Ab + (long)row0 * lda, (long)tile_rows * lda * (long)sizeof(T));
  VT acc_col[NU][P], xJ[NU][P];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int c = 0; c < P; ++c) {
#pragma unroll
      for (int v = 0; v < VN; ++v) acc_col[u][c][v] = T(0);
      if (colok[u]) {
        xJ[u][c] = *reinterpret_cast<const VT*>(Xb + (long)c * ldx + jj[u]);
      } else {
#pragma unroll
        for (int v = 0; v < VN; ++v) xJ[u][c][v] = T(0);
      }
    }
  if (crossing) {
    // rows below this WAVE's last column hold only strictly-lower elements for it: stop there
    int w_end = col0 + (wave + 1) * WCOLS;
    w_end = w_end < i_end ? w_end : i_end;
    symm_tile_rows<T, P, true>(tile, Xb, ldab, ldx, N, row0, w_end, jj, joff, row0, acc_col, xJ, rowacc, lane);
  }
  else
    symm_tile_rows<T, P, false>(tile, Xb, ldab, ldx, N, row0, i_end, jj, joff, row0, acc_col, xJ, rowacc, lane);
#if defined(XK_SYMM_EXP) && XK_SYMM_EXP >= 1      /* experiment (wrong results): no end barrier, no row flush */
  if (rowacc[threadIdx.x] == T(12345.678)) rowP[threadIdx.x] = T(1);
#else
  __syncthreads();
  // flush: row partial slot J (rows of this tile), column partial slot I (columns of this slab)
  T* rp = rowP + (((long)b * NS + J) * P) * (long)N;
  const int nrows = (row0 + SYMM_TRH <= N ? SYMM_TRH : N - row0);
  if (flags & 1) {
    for (int idx = threadIdx.x; idx < nrows * P; idx += 256) {
      const int c = idx / nrows, lr = idx - c * nrows;
      __builtin_nontemporal_store(rowacc[lr * P + c], &rp[(long)c * N + row0 + lr]);
    }
  } else {
    for (int idx = threadIdx.x; idx < nrows * P; idx += 256) {
      const int c = idx / nrows, lr = idx - c * nrows;
      rp[(long)c * N + row0 + lr] = rowacc[lr * P + c];
    }
  }
#endif
  T* cp = colP + (((long)b * NT + I) * P) * (long)N;
#if defined(XK_SYMM_EXP) && XK_SYMM_EXP >= 2      /* experiment: no column flush either (all sums kept alive) */
  T chk = T(0);
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int c = 0; c < P; ++c)
#pragma unroll
      for (int v = 0; v < VN; ++v) chk += acc_col[u][c][v];
  const bool doflush = (chk == T(12345.678));
#else
  const bool doflush = true;
#endif
#pragma unroll
  for (int u = 0; u < NU; ++u)
    if (colok[u] && doflush) {
      if (flags & 1) {
#pragma unroll
        for (int c = 0; c < P; ++c) __builtin_nontemporal_store(acc_col[u][c], reinterpret_cast<VT*>(cp + (long)c * N + jj[u]));
      } else {
#pragma unroll
        for (int c = 0; c < P; ++c) *reinterpret_cast<VT*>(cp + (long)c * N + jj[u]) = acc_col[u][c];
      }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void symm_fold(const T* __restrict__ rowP, const T* __restrict__ colP,
                                                  T* __restrict__ Y, int N, int P, int NS, int NT, int slab,
                                                  long ldy, long sY, long total, int flags) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over B*P*N
  if (idx >= total) return;
  const long per_b = (long)P * N;
  const long b = idx / per_b;
  const long rem = idx - b * per_b;
  const int c = (int)(rem / N);
  const int n = (int)(rem - (long)c * N);
  const int It = n / SYMM_TRH;                 // row tile of n
  const int Jfirst = (It * SYMM_TRH) / slab;   // first column slab that owns a tile with row tile It
  T s = T(0);
  const int Imax = ((n / slab) * slab + slab - 1) / SYMM_TRH;   // row tiles I with I*TRH <= last column of n's slab
  if (flags & 2) {                                              // the partials are read exactly once
    for (int J = Jfirst; J < NS; ++J) s += __builtin_nontemporal_load(&rowP[(((long)b * NS + J) * P + c) * (long)N + n]);
    for (int I = 0; I <= Imax && I < NT; ++I)
      s += __builtin_nontemporal_load(&colP[(((long)b * NT + I) * P + c) * (long)N + n]);
  } else {
    for (int J = Jfirst; J < NS; ++J) s += rowP[(((long)b * NS + J) * P + c) * (long)N + n];
    for (int I = 0; I <= Imax && I < NT; ++I) s += colP[(((long)b * NT + I) * P + c) * (long)N + n];
  }
  Y[b * sY + (long)c * ldy + n] = s;
}


// bit 0: the row / column partials leave with non-temporal stores, bit 1: the fold reads them with non-temporal
// loads (they are written once and read once, 6 ms apart: keeping them out of L2's way is worth 1.2 % of the call,
// same-process A/B scripts/symm_flags_ab.py); 0 restores plain stores / loads for that A/B
static int g_symm_flags = 3;
static int g_symm_variant = 1;     // 1: per-lane rows + wave reductions (this file), 2: LDS turn + MFMA row part

}  // namespace xk

extern "C" {

// which implementation serves xk_dense_symm_* (both read only the upper triangle and share the workspace
// contract); returns the previous setting.  Kept for A/B measurements (bench.py --k1s-variant).
// A/B switch for the non-temporal handling of the partials (see g_symm_flags); returns the previous value
int xk_dense_symm_set_flags(int f) {
  const int old = xk::g_symm_flags;
  xk::g_symm_flags = f;
  return old;
}

int xk_dense_symm_set_variant(int v) {
  const int old = xk::g_symm_variant;
  if (v == 1 || v == 2) xk::g_symm_variant = v;
  return old;
}

// workspace (elements): row partials (B, NS, P, N) + column partials (B, NT, P, N), the larger of the variants
long xk_dense_symm_workspace_elems(int B, int N, int P, int elem_size) {
  const int vn = 16 / elem_size;
  const long slab = 256L * vn * xk::SYMM_NU;
  const long NS = (N + slab - 1) / slab, NT = (N + xk::SYMM_TRH - 1) / xk::SYMM_TRH;
  const long pc = P > 6 ? 6 : P;
  const long v1 = (long)B * (NS + NT) * pc * N;
  return v1;
}

#define XK_DEFINE_SYMM(SUF, T)                                                                              \
  static int symm_launch_##SUF(const T* A, const T* X, T* Y, T* ws, long ws_elems, int B, int N, int P,     \
                               long lda, long sA, long ldx, long sX, long ldy, long sY, void* stream,       \
                               int phase) {                                                                 \
    if (B < 0 || N < 0 || P < 0) return XK_ERR_ARG;                                                         \
    if (B == 0 || N == 0 || P == 0) return XK_OK;                                                           \
    if (phase != 0 && P > 6) return XK_ERR_UNSUPPORTED;   /* split phases: one column chunk only */         \
    constexpr int VN = xk::Vec16<T>::n;                                                                     \
    constexpr int SLAB = 256 * VN * xk::SYMM_NU;                                                            \
    if ((N % VN) || (lda % VN) || (sA % VN) || (ldx % VN) || (sX % VN) || ((uintptr_t)A & 15) ||             \
        ((uintptr_t)X & 15) || ((uintptr_t)ws & 15))                                                        \
      return XK_ERR_UNSUPPORTED;                                                                            \
    hipStream_t st = (hipStream_t)stream;                                                                   \
    const int NS = (N + SLAB - 1) / SLAB, NT = (N + xk::SYMM_TRH - 1) / xk::SYMM_TRH;                       \
    int nt = 0;                                                                                             \
    for (int I = 0; I < NT; ++I) nt += NS - (I * xk::SYMM_TRH) / SLAB;                                      \
    int c0 = 0;                                                                                             \
    while (c0 < P) {                                                                                        \
      const int pc = (P - c0) >= 6 ? 6 : (P - c0);                                                          \
      const long nrow = (long)B * NS * pc * N, ncol = (long)B * NT * pc * N;                                \
      if (ws_elems < nrow + ncol) return XK_ERR_ARG;                                                        \
      T* rowP = ws;                                                                                         \
      T* colP = ws + nrow;                                                                                  \
      const size_t lds = (size_t)xk::SYMM_TRH * pc * sizeof(T);                                             \
      const dim3 grid((unsigned)((long)B * nt));                                                            \
      const T* Xc = X + (long)c0 * ldx;                                                                     \
      if (phase != 2) {                                                                                     \
        switch (pc) {                                                                                       \
          XK_SYMM_CASE(1) XK_SYMM_CASE(2) XK_SYMM_CASE(3) XK_SYMM_CASE(4) XK_SYMM_CASE(5) XK_SYMM_CASE(6)   \
        }                                                                                                   \
        XK_LAUNCH_CHECK();                                                                                  \
      }                                                                                                     \
      if (phase != 1) {                                                                                     \
        const long total = (long)B * pc * N;                                                                \
        hipLaunchKernelGGL((xk::symm_fold<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,     \
                           rowP, colP, Y + (long)c0 * ldy, N, pc, NS, NT, SLAB, ldy, sY, total,             \
                           xk::g_symm_flags);                                                               \
        XK_LAUNCH_CHECK();                                                                                  \
      }                                                                                                     \
      c0 += pc;                                                                                             \
    }                                                                                                       \
    return XK_OK;                                                                                           \
  }                                                                                                         \
  int xk_dense_symm_##SUF(const T* A, const T* X, T* Y, T* ws, long ws_elems, int B, int N, int P, long lda, \
                          long sA, long ldx, long sX, long ldy, long sY, void* stream) {                    \
    return symm_launch_##SUF(A, X, Y, ws, ws_elems, B, N, P, lda, sA, ldx, sX, ldy, sY, stream, 0);         \
  }                                                                                                         \
  int xk_dense_symm_tiles_##SUF(const T* A, const T* X, T* ws, long ws_elems, int B, int N, int P,          \
                                long lda, long sA, long ldx, long sX, void* stream) {                       \
    return symm_launch_##SUF(A, X, (T*)nullptr, ws, ws_elems, B, N, P, lda, sA, ldx, sX, 0, 0, stream, 1);  \
  }                                                                                                         \
  int xk_dense_symm_fold_##SUF(T* Y, const T* ws, long ws_elems, int B, int N, int P, long ldy, long sY,    \
                               void* stream) {                                                              \
    /* the fold never touches A or X: alignment-checked placeholders */                                     \
    return symm_launch_##SUF((const T*)ws, (const T*)ws, Y, (T*)ws, ws_elems, B, N, P, N, 0, N, 0, ldy, sY, \
                             stream, 2);                                                                    \
  }

#define XK_SYMM_CASE(PP)                                                                                  \
  case PP:                                                                                                \
    hipLaunchKernelGGL((xk::dense_symm_tiles<TT, PP>), grid, dim3(256), lds, st, A, Xc, rowP, colP, nt,   \
                       N, lda, sA, ldx, sX, NS, NT, xk::g_symm_flags);                                    \
    break;

#define TT double
XK_DEFINE_SYMM(f64, double)
#undef TT
#define TT float
XK_DEFINE_SYMM(f32, float)
#undef TT

}  // extern "C"
