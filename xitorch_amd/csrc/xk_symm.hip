// xitorch_amd :: K1s — operator-panel product for EXACTLY symmetric dense storage, reading only the
// upper triangle:   Y[b,c,:] = A_b X[b,c,:],  A_b = A_b^T.
//
// The general K1 (xk_dense.hip) streams all N^2 elements per panel product.  A symmetric matrix
// carries every off-diagonal value twice, so here each tile on/above the diagonal is read ONCE
// and used for both of its contributions
//        y_I += A_IJ x_J      (row part)          y_J += A_IJ^T x_I     (column part)
// which halves the HBM traffic of the eigensolver's panel product (symeig operators are always
// Hermitian: xitorch/linalg/symeig.py:103; the product itself: _impls/linalg/symeig.py:163,221).
// Opt-in: the caller asserts exact symmetry of the storage (LinearOperator.m verifies it bit for
// bit, linop.py:97-105 in the reference only checks allclose); anything else takes the general kernel.
//
// Mapping (round 3).  Tiles of TRH = 1024 rows x SLAB columns (1024 fp64 / 2048 fp32) on/above the
// diagonal.  One 256-thread block owns a RUN of up to L consecutive tiles of one row tile (same rows,
// adjacent column slabs); its 4 waves own 4 x WCOLS columns of the current slab (lane: two 16 B vectors,
// so every load instruction is a contiguous 1 KB) and walk down rows in chunks of 8 through a ring of
// 8 rows per wave (16 buffer loads = 16 KB always in flight: descriptor + loop-invariant lane offset +
// scalar row/column offset, no 64-bit vector address arithmetic).  The ring never drains inside a run:
// the last chunk of a row range refills it with the first rows of the wave's next range, which may
// belong to the next tile.
//   * column part: per-lane register accumulators acc_col[NU][P] over the tile (panel values x_I are
//     wave-uniform scalar loads), flushed once per tile to the column partials colP[I][c][j];
//   * row part: per-lane products a[r]*x_J, folded across the 64 lanes by an eager transposing tree
//     (half-exchange swaps, then DPP partner exchanges), then added into an LDS accumulator
//     rowacc[1024][P] that lives for the whole run: one row partial rowP[slot][c][i] per RUN.
//   * DETERMINISTIC accumulation (round 3): the four waves never add into the same accumulator rows at
//     the same time.  A tile is processed in four phases separated by block barriers; in phase q wave w
//     works on row quarter (w + q) mod 4, so every accumulator entry receives its four waves'
//     contributions in a fixed order (phase order) and every wave's own contributions in program
//     order.  Two runs of the same launch give bit-identical results (round 2 let the LDS float atomics
//     of the four waves race).  `s_barrier` does not drain vector memory (the ring stays in flight).
//   * a fold kernel adds, for every output element, exactly the partial slots that exist, in fixed order:
//        y[c][n] = sum_{slot < runs(n>>10)} rowP[slot][c][n] + sum_{I <= Imax(n)} colP[I][c][n].
//   Quarters that reach the diagonal mask the strictly-lower elements (and count the diagonal once);
//   a wave skips quarters that lie entirely below its columns.  Lanes past the last column of a ragged
//   matrix read through an out-of-range offset (the hardware bounds check returns zeros).
//
// Register budget (fp64, P=6): ~200 VGPRs -> 2 waves per SIMD; the panel/accumulator registers (96)
// cannot be shared between waves, so a third wave (<=168 VGPRs) is out of reach and the ring depth is
// what keeps enough bytes in flight.
//
// Measured (round 3, fp64, P = 6, 32 x 16384^2 per launch, same process as the round-2 kernel, profiles/r03_k1s_ab.jsonl):
//   alone on the GPU 5.68-5.80 ms (6.05-5.93 TB/s = 0.757-0.741 of 8 TB/s), the round-2 kernel 5.69-5.82: the fixed
//   accumulation order costs nothing.  Builds without the phase barriers: 5.676 vs 5.684 ms (the barriers are free);
//   ring slots re-issued per row or per 16 B vector instead of per row pair: 5.66 / 5.67 (nothing); runs of L = 2 slabs:
//   equal alone, 4.6 % slower inside the eigensolver's pipeline; L = 4 / 8: 8 / 29 % slower (the launch has only ~8-11
//   tiles per resident workgroup, so longer work items lose more in the tail than the saved row partials return).
//   L = 1 is the default; the knob stays for measurements (bits 8..15 of the entry points' `opts`).
//
// Traffic per launch: B*N^2*s/2 (+2 % for the diagonal tiles) + B*(NT + NS/L)*P*N*s of partials written
// and read once — vs B*N^2*s for the general kernel.
#include "xk_common.h"

namespace xk {

constexpr int SYMM_TRH = 1024;   // rows per tile (template parameter TRH of the kernel: 1024, or 512 for small launches)
// rows per quarter (phase granularity of the deterministic accumulation): TRH / 4
constexpr long SYMM_SMALL_LAUNCH = 2200;   // workgroups (of 1024 rows) below which an fp64 launch uses 512-row tiles: +0.7 .. 1.3 % up to 16 operators of order 16384 per launch, -5 % at 32 (profiles/r04_k1s_tile_rows.jsonl)
constexpr int SYMM_NU = 2;       // 16 B vectors per lane per row: a wave spans NU x 64 x VN columns
constexpr int SYMM_R = 8;        // rows per chunk == ring depth
constexpr long SYMM_QUEUE_ELEMS = 16;   // workspace elements kept for the run queue of the resident launch (>= 64 B, at the end)

// The operator rows of a run are read through ONE buffer descriptor (base = first row of the row tile,
// wave-uniform): every load is  descriptor + per-lane column offset (one VGPR, loop-invariant) + scalar
// (row, slab) offset.  aux = 2: non-temporal (the operator is touched once per product).
typedef __amdgpu_buffer_rsrc_t TileRsrc;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr unsigned SYMM_OOR = 0x7ffffff0u;     // beyond any descriptor range: returns zeros, moves no data

// cache-policy bits of the operator loads (gfx942+: bit 0 sc0, bit 1 nt, bit 4 sc1).  2 = non-temporal, device-default scope:
// what ships; the other combinations were measured as trial builds (-DXK_SYMM_AUX=n, profiles/r05_k1s_cache_policy.jsonl)
#ifndef XK_SYMM_AUX
#define XK_SYMM_AUX 2
#endif
template <typename VT>
__device__ __forceinline__ VT ld_tile(const TileRsrc rsrc, unsigned lane_off, unsigned s_off) {
  const u4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)s_off, XK_SYMM_AUX);
  return __builtin_bit_cast(VT, raw);
}

template <typename T>
__device__ __forceinline__ TileRsrc make_tile_rsrc(const T* tile_base, long bytes) {
  const uint64_t v = reinterpret_cast<uint64_t>(tile_base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  void* base = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  const uint32_t nrec = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0x7fffffe0L ? 0x7fffffe0L : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)nrec, 0x00020000);
}

// Where the ring is refilled from while a chunk is consumed (everything wave-uniform but `loff`):
// the chunk one ring depth ahead in the wave's sequence — usually the next 8 rows of the same range,
// at the end of a range the first rows of the wave's next range (next phase, or next tile of the run).
struct SymmNext {
  int row;          // first row of that chunk (absolute)
  int last;         // last row of its range: rows past it re-read this one (masked when consumed)
  int col0;         // first column of its tile
  int diag;         // its range reaches the diagonal: lanes strictly below the row fetch nothing
};

// ---------------------------------------------------------------------------------------------
// 8-row chunk.  The row sums are folded EAGERLY so that the 48 partial sums never coexist:
//   rows (2h, 2h+1)  -> half-exchange over lane bit 5      (6 values per row pair)
//   row pairs        -> half-exchange over lane bit 4      (6 values per 4 rows)
//   the two 4-groups -> select + xor-8 exchange            (6 values per 8 rows)
//   panel columns    -> (even P) select + xor-4 exchange, then xor-2 / xor-1 butterflies
// afterwards lane l holds, for row r = 4*bit3 + 2*bit4 + bit5 of the chunk, the complete sums of
// columns c = 2w + bit2 (w < P/2); lanes with bits 1,0 clear add them into the LDS accumulator
// (ds_add_f64/f32: a single wave owns these accumulator rows during the phase, see above).
// DIAG: the chunk's rows reach the wave's columns (mask strictly-lower elements, diagonal once).
// TAIL: the chunk is the ragged end of its range (rows >= i_end are masked).
// ---------------------------------------------------------------------------------------------
template <typename T, int P, bool DIAG, bool TAIL>
__device__ __forceinline__ void symm_chunk8(
    typename Vec16<T>::type (&a)[SYMM_R][SYMM_NU], const TileRsrc Ab, const T* __restrict__ Xb, unsigned ldab,
    long ldx, int i0, int i_end, int row_tile0, int col0, int N, const int (&jj)[SYMM_NU], const SymmNext nx,
    typename Vec16<T>::type (&acc_col)[SYMM_NU][P], const typename Vec16<T>::type (&xJ)[SYMM_NU][P],
    T* rowacc, int lane) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  constexpr int R = SYMM_R, NU = SYMM_NU;
  const int i_last = i_end - 1;
  const unsigned ncoloff = (unsigned)nx.col0 * (unsigned)sizeof(T);
  // per-lane byte offset inside a row of the successor's tile; lanes past the last column of a ragged matrix get
  // an offset beyond the descriptor's range (the hardware bounds check returns zeros: no branch, no select later)
  unsigned nloff[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u)
    nloff[u] = (jj[u] - col0 + nx.col0 < N) ? (unsigned)(jj[u] - col0) * (unsigned)sizeof(T) : SYMM_OOR;
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
          if (DIAG) {
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
      // rolling prefetch: the two rows just consumed are refilled with the rows one ring depth ahead in the
      // wave's sequence, so the wave always has ~6 row pairs of loads in flight while it computes.  Issued on
      // every path (a range without successor refills through the out-of-range offset: no data moves), so the
      // compiler's vmcnt bookkeeping stays exact.
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        int row = nx.row + 4 * g + 2 * h + q;
        row = row < nx.last ? row : nx.last;
        const unsigned soff = (unsigned)(row - row_tile0) * ldab + ncoloff;
        // (diagonal ranges: a lane whose columns all lie strictly below the row fetches nothing — its values
        //  would be masked to zero anyway; the out-of-range offset returns the zeros without the traffic)
        const int thr = nx.diag ? row - nx.col0 - (VN - 1) + col0 : -0x40000000;
#pragma unroll
        for (int u = 0; u < NU; ++u)
          a[4 * g + 2 * h + q][u] = ld_tile<VT>(Ab, (!DIAG || jj[u] >= thr) ? nloff[u] : SYMM_OOR, soff);
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

// The geometry of one run as seen by one wave (all wave-uniform).
struct SymmRun {
  int row0;         // first row of the row tile
  int tile_end;     // one past its last row (ragged last tile: N)
  int J0;           // first column slab of the run
  int ntile;        // slabs in the run
  int N;
  int wave;
  int sig;          // the wave's quarter in phase q is q ^ sig (a bijection wave -> quarter in every phase)
};

// Rows [rb, re) wave `wave` handles in phase q of tile j of the run, and whether they reach the diagonal.
template <typename T, int TRH, int NW>
__device__ __forceinline__ bool symm_range(const SymmRun& run, int j, int q, int& rb, int& re, int& col0, int& diag) {
  constexpr int SYMM_QR = TRH / NW;
  constexpr int VN = Vec16<T>::n;
  constexpr int WCOLS = SYMM_NU * 64 * VN, SLAB = NW * WCOLS;
  col0 = (run.J0 + j) * SLAB;
  const int wc0 = col0 + run.wave * WCOLS, wc1 = wc0 + WCOLS;
  // rows that can hold an element on/above the diagonal for this wave: row <= its last column
  int wend = run.tile_end < wc1 ? run.tile_end : wc1;
  if (wc0 >= run.N) wend = run.row0;                       // ragged last slab: the wave has no columns
  const int k = q ^ run.sig;
  rb = run.row0 + k * SYMM_QR;
  re = rb + SYMM_QR;
  re = re < wend ? re : wend;
  diag = (re - 1 >= wc0) ? 1 : 0;
  return rb < re;
}

// first non-empty range after (j, q) in the wave's sequence; nx.row < 0 when there is none
template <typename T, int TRH, int NW>
__device__ __forceinline__ SymmNext symm_next_range(const SymmRun& run, int j, int q) {
  SymmNext nx;
  nx.row = -1; nx.last = 0; nx.col0 = 0; nx.diag = 0;
#pragma unroll 1
  for (int s = 0; s < 2 * NW; ++s) {
    if (++q == NW) { q = 0; ++j; }
    if (j >= run.ntile) break;
    int rb, re, c0, dg;
    if (symm_range<T, TRH, NW>(run, j, q, rb, re, c0, dg)) {
      nx.row = rb; nx.last = re - 1; nx.col0 = c0; nx.diag = dg;
      break;
    }
  }
  return nx;
}

template <typename T, int P, bool DIAG>
__device__ __forceinline__ void symm_rows(
    typename Vec16<T>::type (&a)[SYMM_R][SYMM_NU], const TileRsrc Ab, const T* __restrict__ Xb, unsigned ldab,
    long ldx, int rb, int re, int row_tile0, int col0, int N, const int (&jj)[SYMM_NU], const SymmNext after,
    typename Vec16<T>::type (&acc_col)[SYMM_NU][P], const typename Vec16<T>::type (&xJ)[SYMM_NU][P],
    T* rowacc, int lane) {
  const int nfull = (re - rb) / SYMM_R;
  const bool tail = ((re - rb) % SYMM_R) != 0;
  int i0 = rb;
  for (int c = 0; c < nfull; ++c, i0 += SYMM_R) {
    // the successor of a chunk lies in this same range, except for the last chunk of the range: there the ring
    // moves on to the wave's next range (scalar selects; the chunk's loads are the same on both paths)
    const bool last = !tail && c == nfull - 1;
    SymmNext nx;
    nx.row = last ? after.row : i0 + SYMM_R;
    nx.last = last ? after.last : re - 1;
    nx.col0 = last ? after.col0 : col0;
    nx.diag = last ? after.diag : (DIAG ? 1 : 0);
    symm_chunk8<T, P, DIAG, false>(a, Ab, Xb, ldab, ldx, i0, re, row_tile0, col0, N, jj, nx, acc_col, xJ, rowacc,
                                   lane);
  }
  if (tail)
    symm_chunk8<T, P, DIAG, true>(a, Ab, Xb, ldab, ldx, i0, re, row_tile0, col0, N, jj, after, acc_col, xJ, rowacc,
                                  lane);
}

// per-tile set-up of one wave: absolute columns of its lanes, zeroed column sums, the panel values x_J of its columns
template <typename T, int P>
__device__ __forceinline__ void symm_tile_setup(const T* __restrict__ Xb, int ldx, int col0, int N,
                                                const int (&lanecol)[SYMM_NU], int (&jj)[SYMM_NU],
                                                typename Vec16<T>::type (&acc_col)[SYMM_NU][P],
                                                typename Vec16<T>::type (&xJ)[SYMM_NU][P]) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
#pragma unroll
  for (int u = 0; u < SYMM_NU; ++u) {
    jj[u] = col0 + lanecol[u];
#pragma unroll
    for (int c = 0; c < P; ++c) {
#pragma unroll
      for (int v = 0; v < VN; ++v) acc_col[u][c][v] = T(0);
      if (jj[u] < N) {
        xJ[u][c] = *reinterpret_cast<const VT*>(Xb + (long)c * ldx + jj[u]);
      } else {
#pragma unroll
        for (int v = 0; v < VN; ++v) xJ[u][c][v] = T(0);
      }
    }
  }
  // Drain here, once per tile: the x_J loads above (and the previous tile's column-partial stores) are YOUNGER than
  // the ring loads already in flight for this tile's first rows.  Left pending, the wait for x_J at the head of the
  // chunk loop would be vmcnt(0) on every iteration (the compiler's counts are per program point, it cannot peel
  // the first one) and drain the whole ring once per chunk.  vmcnt(0), expcnt/lgkmcnt untouched.
  __builtin_amdgcn_s_waitcnt(0x0f70);
}

// PERSIST (round 5): the launch is `gridDim.x` resident workgroups (two per compute unit of the stream's CU mask) that
// take runs from a queue in global memory (one atomic per run, fetched one run ahead so that its latency lies under
// the current run) until it is empty, instead of one workgroup per run.  Which workgroup serves a run does not enter
// the result: partial slots are indexed by the run, the order inside a run is fixed (bit-identical to the
// one-workgroup-per-run launch).  What the resident form is for: a second launch on ANOTHER stream (the other batch
// group's panel product) cannot place a workgroup before this launch's workgroups retire, i.e. before its queue is
// empty — the older launch keeps the whole machine, the younger one fills the slots its tail frees.  Two
// one-workgroup-per-run launches on two streams would share the slots evenly and finish together, which is the one
// thing the two-group pipeline must not do.
//
// NW (round 5): waves per workgroup = row ranges ("quarters" in the comments: NW of them) per tile = phases per tile.
// 4 waves: 1024 (fp64) / 2048 (fp32) columns per workgroup, two workgroups per compute unit.  8 waves (fp64, TRH = 2048):
// ONE workgroup per compute unit owns 2048 x 2048 tiles — per tile the same two partials (row sums, column sums: TRH + SLAB
// values per panel column) for four times the elements, i.e. half the partial bytes written here and read by the fold;
// the row accumulator is 96 KB of LDS.
template <typename T, int P, int TRH, bool PERSIST, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2))) void dense_symm_tiles(
    const T* __restrict__ A, const T* __restrict__ X, T* __restrict__ rowP, T* __restrict__ colP, int nruns,
    int N, long lda, long sA, long ldx, long sX, int NS, int NT, int NSL, int L, int flags,
    unsigned* __restrict__ queue, int nitems) {
  typedef typename Vec16<T>::type VT;
  constexpr int VN = Vec16<T>::n;
  constexpr int NU = SYMM_NU;
  constexpr int WCOLS = NU * 64 * VN;          // columns per wave
  constexpr int SLAB = NW * WCOLS;             // columns per tile
  constexpr int SYMM_TRH = TRH, SYMM_QR = TRH / NW;
  static_assert(NW == 4 || WCOLS == TRH / NW, "more than four waves: a wave's columns span exactly one row range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* rowacc = reinterpret_cast<T*>(smem);                   // SYMM_TRH x P
  __shared__ int s_next[2];            // two slots, written alternately: a slot is rewritten two runs after it was read
  int item = blockIdx.x, par = 0;
  if (PERSIST) {
    if (threadIdx.x == 0) s_next[1] = (int)atomicAdd(queue, 1u);
    __syncthreads();
    item = __builtin_amdgcn_readfirstlane(s_next[1]);
  }
#pragma unroll 1
  for (;;) {
  if (PERSIST) {
    if (item >= nitems) break;
    if (threadIdx.x == 0) s_next[par] = (int)atomicAdd(queue, 1u);   // the run after this one: read at the end of the body
  }
  // run list in row-tile-major order: row tile I owns the slabs J >= (I*TRH)/SLAB, cut into runs of L
  int b = item / nruns;
  int I = 0, slot = 0, jmin = 0, cnt = 0;
  {
    int rem = item - b * nruns;
    for (;; ++I) {
      jmin = (I * SYMM_TRH) / SLAB;
      cnt = NS - jmin;
      const int nr = (cnt + L - 1) / L;
      if (rem < nr) { slot = rem; break; }
      rem -= nr;
    }
  }
  // the integer divisions above run on the vector ALU: pin their (wave-uniform) results in SGPRs so that
  // everything derived from them (row offsets, loop bounds, panel addresses) is scalar arithmetic
  b = __builtin_amdgcn_readfirstlane(b);
  I = __builtin_amdgcn_readfirstlane(I);
  slot = __builtin_amdgcn_readfirstlane(slot);
  jmin = __builtin_amdgcn_readfirstlane(jmin);
  cnt = __builtin_amdgcn_readfirstlane(cnt);
  const int lane = threadIdx.x & 63;
  SymmRun run;
  run.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform
  run.row0 = I * SYMM_TRH;
  run.tile_end = run.row0 + SYMM_TRH <= N ? run.row0 + SYMM_TRH : N;
  run.J0 = jmin + slot * L;
  run.ntile = (cnt - slot * L) < L ? (cnt - slot * L) : L;
  run.N = N;
  // phase -> quarter map q ^ sig: sig = wave when a wave's columns span one quarter of rows (fp64); with two quarters
  // per wave (fp32) waves 1 and 2 swap, so that both quarters of a wave's diagonal block come in phases 0 and 1
  run.sig = (WCOLS == SYMM_QR) ? run.wave : ((run.wave & 1) << 1 | (run.wave >> 1));   // (the second form: NW == 4 only)
  const int wave = run.wave;
  int lanecol[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) lanecol[u] = wave * WCOLS + u * 64 * VN + lane * VN;   // 1 KB contiguous per load
  const T* Ab = A + (long)b * sA;
  const T* Xb = X + (long)b * sX;
  const unsigned ldab = (unsigned)(lda * (long)sizeof(T));
  const TileRsrc tile = make_tile_rsrc(Ab + (long)run.row0 * lda,
                                       (long)(run.tile_end - run.row0) * lda * (long)sizeof(T));
  // ---- ring fill: the first rows of the wave's first range (unconditional: a wave without any range fills
  // through the out-of-range offset) — issued BEFORE the LDS set-up so that the set-up runs under the loads
  VT a[SYMM_R][NU];
  {
    SymmNext first = symm_next_range<T, TRH, NW>(run, 0, -1);
    const int ncols = first.row >= 0 ? N : 0;          // no range at all: every lane fills through the OOR offset
    if (first.row < 0) { first.row = run.row0; first.last = run.row0; }
    unsigned floff[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u)
      floff[u] = (first.col0 + lanecol[u] < ncols) ? (unsigned)lanecol[u] * (unsigned)sizeof(T) : SYMM_OOR;
#pragma unroll
    for (int r = 0; r < SYMM_R; ++r) {
      int row = first.row + r;
      row = row < first.last ? row : first.last;
      const unsigned soff = (unsigned)(row - run.row0) * ldab + (unsigned)first.col0 * (unsigned)sizeof(T);
      const int thr = first.diag ? row - first.col0 - (VN - 1) : -0x40000000;
#pragma unroll
      for (int u = 0; u < NU; ++u) a[r][u] = ld_tile<VT>(tile, lanecol[u] >= thr ? floff[u] : SYMM_OOR, soff);
    }
  }
  for (int idx = threadIdx.x; idx < SYMM_TRH * P; idx += NW * 64) rowacc[idx] = T(0);
  __syncthreads();

  VT acc_col[NU][P], xJ[NU][P];
  int jj[NU];
  int col0 = run.J0 * SLAB;
  symm_tile_setup<T, P>(Xb, (int)ldx, col0, N, lanecol, jj, acc_col, xJ);
  int q = 0;
  if (slot == 0) {
    // The first tile of a row tile's first run reaches the diagonal.  With the phase -> quarter map above every
    // range that needs the diagonal masks lies in the first WCOLS/QR phases of that tile: they run here, on the
    // masked instantiation (all four waves, whatever their own range needs), so the loop below is mask-free.
#pragma unroll 1
    for (; q < WCOLS / SYMM_QR; ++q) {
      int rb, re, c0, dg;
      if (symm_range<T, TRH, NW>(run, 0, q, rb, re, c0, dg)) {
        SymmNext after = symm_next_range<T, TRH, NW>(run, 0, q);
        if (after.row < 0) { after.row = run.row0; after.last = run.row0; after.col0 = N; }
        symm_rows<T, P, true>(a, tile, Xb, ldab, ldx, rb, re, run.row0, col0, N, jj, after, acc_col, xJ,
                              rowacc, lane);
      }
      __syncthreads();
    }
  }
  int j = 0;
#pragma unroll 1
  for (;;) {
#pragma unroll 1
    for (; q < NW; ++q) {
      int rb, re, c0, dg;
      if (symm_range<T, TRH, NW>(run, j, q, rb, re, c0, dg)) {
        SymmNext after = symm_next_range<T, TRH, NW>(run, j, q);
        // no successor (end of the run for this wave): refill through the out-of-range offset (a tile starting at
        // column N: every lane is beyond the last column), so that every chunk issues the same loads
        if (after.row < 0) { after.row = run.row0; after.last = run.row0; after.col0 = N; }
        symm_rows<T, P, false>(a, tile, Xb, ldab, ldx, rb, re, run.row0, col0, N, jj, after, acc_col, xJ,
                               rowacc, lane);
      }
      __syncthreads();        // s_waitcnt lgkmcnt(0) + s_barrier: the ring's loads stay in flight
    }
    // column partial slot I (columns of this slab)
    T* cp = colP + (((long)b * NT + I) * P) * (long)N;
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (jj[u] < N) {
        if (flags & 1) {
#pragma unroll
          for (int c = 0; c < P; ++c) __builtin_nontemporal_store(acc_col[u][c], reinterpret_cast<VT*>(cp + (long)c * N + jj[u]));
        } else {
#pragma unroll
          for (int c = 0; c < P; ++c) *reinterpret_cast<VT*>(cp + (long)c * N + jj[u]) = acc_col[u][c];
        }
      }
    if (++j >= run.ntile) break;
    q = 0;
    col0 = (run.J0 + j) * SLAB;
    symm_tile_setup<T, P>(Xb, (int)ldx, col0, N, lanecol, jj, acc_col, xJ);
  }
  {
    // the run's row sums of quarter 3 ^ sig are complete: the other three waves added theirs in the earlier
    // phases (barriers), this wave just added the last ones (LDS operations of one wave execute in order).
    // Row partial slot `slot` of this row tile.
    const int k3 = (NW - 1) ^ run.sig;
    const int fb = run.row0 + k3 * SYMM_QR;
    int fe = fb + SYMM_QR;
    fe = fe < run.tile_end ? fe : run.tile_end;
    const int nr = fe - fb;
    T* rp = rowP + (((long)b * NSL + slot) * P) * (long)N;
    for (int idx = lane; idx < nr * P; idx += 64) {
      const int c = idx / nr, lr = idx - c * nr;
      const T v = rowacc[(fb - run.row0 + lr) * P + c];
      if (flags & 1) __builtin_nontemporal_store(v, &rp[(long)c * N + fb + lr]);
      else rp[(long)c * N + fb + lr] = v;
    }
  }
  if (!PERSIST) break;
  __syncthreads();                      // every wave has read its quarter of the row accumulator; s_next is visible
  item = __builtin_amdgcn_readfirstlane(s_next[par]);
  par ^= 1;
  }
}

// the 8-wave form (fp64 only) keeps TRH x P row sums in LDS: 96 KB at P = 6, beyond the 64 KB a kernel gets without asking
template <typename T, int PP, bool PERS>
static int symm_launch_wide8(dim3 grid, size_t lds, hipStream_t st, const T* A, const T* Xc, T* rowP, T* colP,
                             int nruns, int N, long lda, long sA, long ldx, long sX, int NS, int NT, int NSL, int L,
                             int fl, unsigned* queue, int nitems) {
  if constexpr (sizeof(T) == 8) {
    auto kfn = dense_symm_tiles<T, PP, 2048, PERS, 8>;
    if (lds > 65536) {
      hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (ae != hipSuccess) return (int)ae;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, st, A, Xc, rowP, colP, nruns, N, lda, sA, ldx, sX, NS, NT, NSL, L,
                       fl, queue, nitems);
    return XK_OK;
  } else {
    return XK_ERR_UNSUPPORTED;
  }
}

// compute units of the current device (cached per device): the resident launch's default is two workgroups per unit
static int symm_device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

template <typename T>
__global__ __launch_bounds__(256) void symm_fold(const T* __restrict__ rowP, const T* __restrict__ colP,
                                                  T* __restrict__ Y, int N, int P, int NS, int NT, int NSL, int L,
                                                  int slab, long ldy, long sY, long total, int flags, int trh) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over B*P*N
  if (idx >= total) return;
  const long per_b = (long)P * N;
  const long b = idx / per_b;
  const long rem = idx - b * per_b;
  const int c = (int)(rem / N);
  const int n = (int)(rem - (long)c * N);
  const int It = n / trh;                      // row tile of n
  const int jmin = (It * trh) / slab;          // first column slab that owns a tile with row tile It
  const int nslot = (NS - jmin + L - 1) / L;   // runs (= row partial slots) of that row tile
  T s = T(0);
  const int Imax = ((n / slab) * slab + slab - 1) / trh;        // row tiles I with I*TRH <= last column of n's slab
  if (flags & 2) {                                              // the partials are read exactly once
    for (int r = 0; r < nslot; ++r) s += __builtin_nontemporal_load(&rowP[(((long)b * NSL + r) * P + c) * (long)N + n]);
    for (int I = 0; I <= Imax && I < NT; ++I)
      s += __builtin_nontemporal_load(&colP[(((long)b * NT + I) * P + c) * (long)N + n]);
  } else {
    for (int r = 0; r < nslot; ++r) s += rowP[(((long)b * NSL + r) * P + c) * (long)N + n];
    for (int I = 0; I <= Imax && I < NT; ++I) s += colP[(((long)b * NT + I) * P + c) * (long)N + n];
  }
  Y[b * sY + (long)c * ldy + n] = s;
}

}  // namespace xk

extern "C" {

// `opts` of the entry points below (no process-wide state; 0 = the shipped behaviour):
//   bit 0: the row / column partials leave with PLAIN stores instead of non-temporal ones, bit 1: the fold reads them
//          with plain loads (written once, read once, milliseconds apart: keeping them out of L2's way was worth 1.2 %
//          of the eigensolver call in round 2);
//   bits 8..15: L, column slabs per workgroup run (0 = 1; row partials per row tile = ceil(slabs / L));
//   bit 2 / bit 3: force 512- / 1024-row tiles (fp64; default: 512 rows for launches of fewer than SYMM_SMALL_LAUNCH
//          workgroups).
// Results do not depend on either (the run length changes the summation order of the row partials — still a fixed
// order); both exist for measurements.

// workspace (elements): row partials (B, NS, P, N) + column partials (B, NT, P, N) — sized for runs of one slab
long xk_dense_symm_workspace_elems(int B, int N, int P, int elem_size) {
  const int vn = 16 / elem_size;
  const long slab = 256L * vn * xk::SYMM_NU;
  const long NS = (N + slab - 1) / slab, NT = (N + 511) / 512;   // (column partials per 512-row tile: the small-launch form)
  const long pc = P > 6 ? 6 : P;
  return (long)B * (NS + NT) * pc * N + xk::SYMM_QUEUE_ELEMS;   // + the run queue of the resident form (last 64 bytes)
}

#define XK_DEFINE_SYMM(SUF, T)                                                                              \
  static int symm_launch_##SUF(const T* A, const T* X, T* Y, T* ws, long ws_elems, int B, int N, int P,     \
                               long lda, long sA, long ldx, long sX, long ldy, long sY, int opts,           \
                               void* stream, int phase) {                                                   \
    if (B < 0 || N < 0 || P < 0 || opts < 0 || opts > 0x0fffffff) return XK_ERR_ARG;                        \
    if (B == 0 || N == 0 || P == 0) return XK_OK;                                                           \
    if (phase != 0 && P > 6) return XK_ERR_UNSUPPORTED;   /* split phases: one column chunk only */         \
    constexpr int VN = xk::Vec16<T>::n;                                                                     \
    /* opts bit 5 (fp64): 8-wave workgroups on 2048 x 2048 tiles, one per compute unit */                   \
    const bool wide8 = sizeof(T) == 8 && (opts & 32) != 0;                                                  \
    const int NWV = wide8 ? 8 : 4;                                                                          \
    const int SLAB = NWV * 64 * VN * xk::SYMM_NU;                                                           \
    if ((N % VN) || (lda % VN) || (sA % VN) || (ldx % VN) || (sX % VN) || ((uintptr_t)A & 15) ||             \
        ((uintptr_t)X & 15) || ((uintptr_t)ws & 15))                                                        \
      return XK_ERR_UNSUPPORTED;                                                                            \
    if ((long)(wide8 ? 2048 : xk::SYMM_TRH) * lda * (long)sizeof(T) > 0x7fffffe0L) return XK_ERR_UNSUPPORTED; \
    hipStream_t st = (hipStream_t)stream;                                                                   \
    const int L = ((opts >> 8) & 0xff) ? ((opts >> 8) & 0xff) : 1, fl = 3 & ~opts;                          \
    const bool persist = (opts & 16) != 0;                                                                  \
    int nslots = (opts >> 16) & 0xfff;                                                                      \
    if (persist && nslots == 0) nslots = 2 * xk::symm_device_cus();                                         \
    if (persist && wide8) nslots = (nslots + 1) / 2;              /* one 8-wave workgroup per compute unit */ \
    if (persist && (nslots <= 0 || ws_elems < xk::SYMM_QUEUE_ELEMS)) return XK_ERR_ARG;                     \
    unsigned* queue = persist ? reinterpret_cast<unsigned*>(ws + (ws_elems - xk::SYMM_QUEUE_ELEMS)) : nullptr; \
    if (persist) ws_elems -= xk::SYMM_QUEUE_ELEMS;                                                          \
    if (L > 64) return XK_ERR_ARG;                                                                          \
    const int NS = (N + SLAB - 1) / SLAB;                                                                   \
    const int NSL = (NS + L - 1) / L;                                                                       \
    /* rows per tile: 1024; 512 (fp64) when the launch would be a few rounds of workgroups only — a 4-operator   \
     * launch of the 8-operator shard is 544 workgroups of 1024 rows on 384 slots: 1.4 rounds, run as 2 */     \
    int trh = xk::SYMM_TRH;                                                                                 \
    {                                                                                                       \
      const int nt1 = (N + xk::SYMM_TRH - 1) / xk::SYMM_TRH;                                                \
      long nr1 = 0;                                                                                         \
      for (int I = 0; I < nt1; ++I) nr1 += (NS - (I * xk::SYMM_TRH) / SLAB + L - 1) / L;                    \
      if (sizeof(T) == 8 && (long)B * nr1 < xk::SYMM_SMALL_LAUNCH) trh = 512;                               \
      if (sizeof(T) == 8 && (opts & 4)) trh = 512;                                                          \
      if (opts & 8) trh = xk::SYMM_TRH;                                                                     \
      if (wide8) trh = 2048;                                                                                \
    }                                                                                                       \
    const int NT = (N + trh - 1) / trh;                                                                     \
    int nruns = 0;                                                                                          \
    for (int I = 0; I < NT; ++I) nruns += (NS - (I * trh) / SLAB + L - 1) / L;                              \
    int c0 = 0;                                                                                             \
    while (c0 < P) {                                                                                        \
      const int pc = (P - c0) >= 6 ? 6 : (P - c0);                                                          \
      const long nrow = (long)B * NSL * pc * N, ncol = (long)B * NT * pc * N;                               \
      if (ws_elems < nrow + ncol) return XK_ERR_ARG;                                                        \
      T* rowP = ws;                                                                                         \
      T* colP = ws + nrow;                                                                                  \
      const size_t lds = (size_t)trh * pc * sizeof(T);                                                      \
      const long nitems_l = (long)B * nruns;                                                                \
      if (nitems_l > 0x7fffffffL) return XK_ERR_UNSUPPORTED;                                                \
      const int nitems = (int)nitems_l;                                                                     \
      const dim3 grid((unsigned)(persist ? (nitems < nslots ? nitems : nslots) : nitems));                  \
      const T* Xc = X + (long)c0 * ldx;                                                                     \
      if (phase != 2) {                                                                                     \
        if (persist) {                                                                                      \
          hipError_t me = hipMemsetAsync(queue, 0, 64, st);                                                 \
          if (me != hipSuccess) return (int)me;                                                             \
        }                                                                                                   \
        switch (pc) {                                                                                       \
          XK_SYMM_CASE(1) XK_SYMM_CASE(2) XK_SYMM_CASE(3) XK_SYMM_CASE(4) XK_SYMM_CASE(5) XK_SYMM_CASE(6)   \
        }                                                                                                   \
        XK_LAUNCH_CHECK();                                                                                  \
      }                                                                                                     \
      if (phase != 1) {                                                                                     \
        const long total = (long)B * pc * N;                                                                \
        hipLaunchKernelGGL((xk::symm_fold<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,     \
                           rowP, colP, Y + (long)c0 * ldy, N, pc, NS, NT, NSL, L, SLAB, ldy, sY, total,     \
                           fl, trh);                                                                        \
        XK_LAUNCH_CHECK();                                                                                  \
      }                                                                                                     \
      c0 += pc;                                                                                             \
    }                                                                                                       \
    return XK_OK;                                                                                           \
  }                                                                                                         \
  int xk_dense_symm_##SUF(const T* A, const T* X, T* Y, T* ws, long ws_elems, int B, int N, int P, long lda, \
                          long sA, long ldx, long sX, long ldy, long sY, int opts, void* stream) {          \
    return symm_launch_##SUF(A, X, Y, ws, ws_elems, B, N, P, lda, sA, ldx, sX, ldy, sY, opts, stream, 0);   \
  }                                                                                                         \
  int xk_dense_symm_tiles_##SUF(const T* A, const T* X, T* ws, long ws_elems, int B, int N, int P,          \
                                long lda, long sA, long ldx, long sX, int opts, void* stream) {             \
    return symm_launch_##SUF(A, X, (T*)nullptr, ws, ws_elems, B, N, P, lda, sA, ldx, sX, 0, 0, opts,        \
                             stream, 1);                                                                    \
  }                                                                                                         \
  int xk_dense_symm_fold_##SUF(T* Y, const T* ws, long ws_elems, int B, int N, int P, long ldy, long sY,    \
                               int opts, void* stream) {                                                    \
    /* the fold never touches A or X: alignment-checked placeholders */                                     \
    return symm_launch_##SUF((const T*)ws, (const T*)ws, Y, (T*)ws, ws_elems, B, N, P, N, 0, N, 0, ldy, sY, \
                             opts, stream, 2);                                                              \
  }

#define XK_SYMM_LAUNCH(PP, RR, PERS)                                                                      \
  hipLaunchKernelGGL((xk::dense_symm_tiles<TT, PP, RR, PERS, 4>), grid, dim3(256), lds, st, A, Xc, rowP, colP, \
                     nruns, N, lda, sA, ldx, sX, NS, NT, NSL, L, fl, queue, nitems)
#define XK_SYMM_WIDE8(PP, PERS)                                                                           \
  do {                                                                                                    \
    const int wrc = xk::symm_launch_wide8<TT, PP, PERS>(grid, lds, st, A, Xc, rowP, colP, nruns, N, lda, sA, ldx, \
                                                        sX, NS, NT, NSL, L, fl, queue, nitems);           \
    if (wrc != XK_OK) return wrc;                                                                         \
  } while (0)
#define XK_SYMM_CASE(PP)                                                                                  \
  case PP:                                                                                                \
    if (wide8) {                                                                                          \
      if (persist) XK_SYMM_WIDE8(PP, true); else XK_SYMM_WIDE8(PP, false);                                \
    } else if (trh == 512) {                                                                              \
      if (persist) XK_SYMM_LAUNCH(PP, 512, true); else XK_SYMM_LAUNCH(PP, 512, false);                    \
    } else {                                                                                              \
      if (persist) XK_SYMM_LAUNCH(PP, 1024, true); else XK_SYMM_LAUNCH(PP, 1024, false);                  \
    }                                                                                                     \
    break;

#define TT double
XK_DEFINE_SYMM(f64, double)
#undef TT
#define TT float
XK_DEFINE_SYMM(f32, float)
#undef TT

}  // extern "C"
