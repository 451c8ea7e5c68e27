// xitorch_amd :: K1sw — the operator-panel product of an EXACTLY SYMMETRIC operator for wide panels (9 .. 16 columns)
// on the matrix cores, streaming only the upper triangle:
//
//        Y[b, c, :] = A_b X[b, c, :],   A_b == A_b^T bit for bit,   9 <= P <= 16   (fp32: BASELINE configs[4], a 16-column
//                                                                    eigen-block on a MatrixLinearOperator N = 32768)
//
// Replaces `torch.matmul(mat, x)` (xitorch/_core/linop.py:695-696) inside the eigensolver (symeig.py:163,221) where K1s
// (xk_symm.hip, VALU, P <= 6 per pass) would re-stream the triangle three times and K1w (xk_wide.hip, MFMA) reads the
// whole matrix: every 64-row x 128-byte sub-tile on or above the diagonal is loaded ONCE and feeds both
//
//        row part   y_I += A_IJ  x_J      M = 16 operator rows,    K = 4 operator columns, N = 16 panel columns
//        col part   y_J += A_IJ^T x_I     M = 16 operator columns, K = 4 operator rows,    N = 16 panel columns
//
// through v_mfma_f32_16x16x4_f32 (exact FMA chains).  The contraction index of the row part is the contiguous one, so
// — as in K1wr (xk_rowswide.hip) — each wave turns its sub-tile through LDS (8 coalesced 16 B/lane non-temporal loads
// -> ds_write_b64 with a 136-byte pitch) and reads it back in the MFMA operand layouts with ds_read_b64 (two k-steps /
// two column blocks per read, conflict-free for both parts); the panel operands: x_J (8 values per lane) arrives with
// each sub-tile's loads from the L1 / L2-resident panel, x_I is parked in LDS once per 64-row band.
//
// Decomposition (no atomics, no block barriers, bit-reproducible): a WAVE owns one tile = TR rows x WS = 8 sub-tiles
// (256 fp32 columns); it walks the tile's 64-row bands top to bottom, each band left to right with the next sub-tile's
// loads in flight, keeps the band's row sums (16 accumulators) and the strip's column sums (64 accumulators, in AGPRs)
// in registers and leaves them as partials:  rowP[b][strip][c][row]  once per band,  colP[b][row tile][c][col]  once
// per tile; `symm_wide_fold` adds, per output element, exactly the slots that exist, in fixed order.  The diagonal
// 64 x 64 block of a band is used WHOLE (both triangles are in storage and equal) for the row part and not at all for
// the column part, so no masks exist anywhere; sub-tiles left of it are "loaded" through an out-of-range buffer offset
// (hardware zeros, no traffic).  Extra traffic of the partials: about 19 % of the triangle bytes at N = 32768.
//
// Three forms live here (opts of the C entry points): the one-wave-per-tile form just described (0), the workgroup-
// cooperative form of round 4 (bit 0: 1024 x 512 super-tiles, three waves per SIMD, partials 7 % of the triangle bytes)
// and the round 6 form (bits 0 + 3, shipped): same super-tiles, the column part fed straight from the load registers.
#include "xk_common.h"

namespace xk {

constexpr int SW_ROWS = 64;                      // rows per band
constexpr int SW_SEG_BYTES = 128;                // bytes of one row inside a sub-tile
constexpr int SW_PITCH = SW_SEG_BYTES + 8;       // 34 dwords: ds_read_b64 of 16 rows / of one row hit 32 distinct banks
constexpr int SW_TILE_LDS = SW_ROWS * SW_PITCH;  // 8704 B per wave
constexpr int SW_XI_LDS = 16 * 64 * 4;           // the band's x_I in the column part's B-operand layout (4096 B)
constexpr int SW_WAVE_LDS = SW_TILE_LDS;                         // 8704 B per wave
constexpr int SW_NSUB = 8;                       // sub-tiles per strip
constexpr long SW_QUEUE_ELEMS = 16;              // workspace elements kept for the queue of the resident launch (64 B, at the end)

typedef float sw_f32x4 __attribute__((ext_vector_type(4)));
typedef float sw_f32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t SwRsrc;

__device__ __forceinline__ SwRsrc sw_rsrc(const void* base, long bytes) {
  const uint64_t v = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  void* b = reinterpret_cast<void*>(((uint64_t)hi << 32) | lo);
  const uint32_t nrec = __builtin_amdgcn_readfirstlane((uint32_t)(bytes > 0xffffffffL ? 0xffffffffL : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(b, (short)0, (int)nrec, 0x00020000);
}

__device__ __forceinline__ sw_f32x4 sw_mma(float a, float b, sw_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// row tiles of strip s: rows [0, min(N, (s + 1) WS)) in pieces of TR
__host__ __device__ inline int sw_tiles_of_strip(int s, int N, int WS, int TR) {
  long top = (long)(s + 1) * WS;
  if (top > N) top = N;
  return (int)((top + TR - 1) / TR);
}

// fp32 only (the 16-wide fp64 MFMA holds the matrix pipe ~120 cycles: a symmetric fp64 kernel of this shape would be
// bound by it at half the triangle rate K1s reaches on the VALU)
__global__ __launch_bounds__(256) void dense_symm_wide_kernel(
    const float* __restrict__ A, const float* __restrict__ X, float* __restrict__ rowP, float* __restrict__ colP,
    int N, int pc, long lda, long sA, long ldx, long sX, int NS, int NT, int TR, int tiles_per_op, long total_tiles) {
  typedef float T;
  constexpr int SEG = SW_SEG_BYTES / (int)sizeof(T);          // 32 columns per sub-tile
  constexpr int WS = SW_NSUB * SEG;                           // 256 columns per strip
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long gt = (long)blockIdx.x * 4 + wave;
  if (gt >= total_tiles) return;
  const int b = __builtin_amdgcn_readfirstlane((int)(gt / tiles_per_op));
  int ti = __builtin_amdgcn_readfirstlane((int)(gt - (long)b * tiles_per_op));
  int s = 0;
  for (;;) {                                                   // (scalar: at most N / 256 steps, once per wave)
    const int cnt = sw_tiles_of_strip(s, N, WS, TR);
    if (ti < cnt) break;
    ti -= cnt;
    ++s;
  }
  s = __builtin_amdgcn_readfirstlane(s);
  const int I = __builtin_amdgcn_readfirstlane(ti);
  const int col0 = s * WS;
  const int r_begin = I * TR;
  int r_end = r_begin + TR;
  r_end = r_end < N ? r_end : N;
  r_end = r_end < col0 + WS ? r_end : col0 + WS;              // bands below the strip's last column: lower triangle only
  char* tile = smem + wave * SW_WAVE_LDS;
  const float* Ab = A + (long)b * sA;
  const float* Xb = X + (long)b * sX;
  const int mm = lane & 15, kq = lane >> 4;                   // MFMA operand coordinates
  const int lrow = lane >> 3, lcol = lane & 7;                // load phase: 8 rows x 8 lanes of 16 B
  const bool cok = mm < pc;
  const float* Xc = Xb + (long)(cok ? mm : 0) * ldx;          // panel column of this lane (columns >= pc: zeros)

  sw_f32x4 acc_col[SW_NSUB][2];
#pragma unroll
  for (int t = 0; t < SW_NSUB; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_col[t][h][r] = 0.f;

  const unsigned ldab = (unsigned)(lda * (long)sizeof(T));
  const unsigned lane_off = (unsigned)lrow * ldab + (unsigned)lcol * 16u;
  const unsigned st_off = (unsigned)lrow * SW_PITCH + (unsigned)lcol * 16u;
  constexpr unsigned POISON = 0x7ffffff0u;

  // sub-tile (row0, t): voff of load tt = lane_off + tt * 8 rows; skipped sub-tiles (left of the diagonal block, or past
  // the matrix) read zeros through the poison offset
  auto issue = [&](sw_f32x4 (&a)[8], const SwRsrc& ra, int row0, int t, bool exists) {
    const int c0 = col0 + t * SEG;
    const bool live = exists && (c0 + SEG > row0) && (c0 < N);
    const unsigned base = live ? lane_off + (unsigned)(c0 - col0) * (unsigned)sizeof(T) : POISON;
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
      const unsigned voff = live ? base + (unsigned)(tt * 8) * ldab : POISON;
      a[tt] = __builtin_bit_cast(sw_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)voff, 0, 2));
    }
  };

  // ---- ONE wave per SIMD (4 per compute unit, ~390 registers): two register buffers — the next sub-tile's loads are
  //      issued before this one's LDS turn and MFMAs, across bands — one LDS tile per wave; x_J of the whole strip (64
  //      values per lane, once per tile) and x_I of the band (16, once per band) stay in registers.  Measured against the
  //      alternatives on 8 x 32768^2 (profiles/r04_k1sw_variants.jsonl): this 3.5 ms; two waves per SIMD with one buffer
  //      3.6-3.7; a ring of four buffers + two LDS tiles 3.9 (register-file moves); any variant that spills inside the
  //      stream 7.6 (a scratch reload turns the counted waits into vmcnt(0)).
  float xj[SW_NSUB][4][2];                                     // xj[t][j][h] = X[c = mm][col0 + 32 t + 8 j + 2 kq + h]
#pragma unroll
  for (int t = 0; t < SW_NSUB; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + t * SEG + 8 * j + 2 * kq;
      sw_f32x2 v = {0.f, 0.f};
      if (cok && col < N) v = *reinterpret_cast<const sw_f32x2*>(Xc + col);
      xj[t][j][0] = v[0];
      xj[t][j][1] = v[1];
    }
  sw_f32x4 abuf[2][8];
  // descriptor of a band: its 64 rows from column col0 on
  auto band_rsrc = [&](int row0) {
    return sw_rsrc(Ab + (long)row0 * lda + col0, ((long)(SW_ROWS - 1) * lda + (N - col0)) * (long)sizeof(T));
  };
  if (r_begin >= r_end) return;
  {
    const SwRsrc r0 = band_rsrc(r_begin);
    issue(abuf[0], r0, r_begin, 0, true);
  }
  for (int row0 = r_begin; row0 < r_end; row0 += SW_ROWS) {
    const SwRsrc ra = band_rsrc(row0);
    const bool more = row0 + SW_ROWS < r_end;
    const int rown = more ? row0 + SW_ROWS : row0;            // (no next band: poisoned loads, results unused)
    const SwRsrc rn = band_rsrc(rown);
    // ---- x_I of the band in the column part's B-operand layout: xi[kk] = X[c = mm][row0 + 4 kk + kq]
    float xi[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float v = Xc[row0 + 4 * kk + kq];                 // (unconditional load, then select: no exec-masked branches)
      xi[kk] = cok ? v : 0.f;
    }
    sw_f32x4 acc_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_row[i][r] = 0.f;
#pragma unroll
    for (int t = 0; t < SW_NSUB; ++t) {
      sw_f32x4(&acur)[8] = abuf[t & 1];
      sw_f32x4(&anxt)[8] = abuf[(t + 1) & 1];
      // next sub-tile's loads go out before this one's LDS turn and MFMAs (the last of a band fetches the next band's
      // first: the prefetch is carried across bands)
      if (t + 1 < SW_NSUB) issue(anxt, ra, row0, t + 1, true);
      else issue(anxt, rn, rown, 0, more);
      __builtin_amdgcn_sched_barrier(0);
      const int c0 = col0 + t * SEG;
      const bool both = (c0 >= row0 + SW_ROWS) && (c0 < N);   // right of the diagonal block: column part too
      // ---- LDS turn: [row][column], 136-byte pitch
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        char* p = tile + st_off + (unsigned)(tt * 8) * SW_PITCH;
        *reinterpret_cast<sw_f32x2*>(p) = sw_f32x2{acur[tt][0], acur[tt][1]};
        *reinterpret_cast<sw_f32x2*>(p + 8) = sw_f32x2{acur[tt][2], acur[tt][3]};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): the wave's own writes have landed
      const char* rrow = tile + (unsigned)mm * SW_PITCH + (unsigned)(2 * kq) * 4u;      // + 16 i rows, + 8 j columns
      const char* rcol = tile + (unsigned)kq * SW_PITCH + (unsigned)(2 * mm) * 4u;      // + 4 kk rows
      // ---- row part: rows 16 i + mm, columns 8 j + 2 kq (+1): one ds_read_b64 = the A operands of two k-steps; the
      //      four row blocks are four independent accumulator chains, visited round-robin
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sw_f32x2 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[i] = *reinterpret_cast<const sw_f32x2*>(rrow + (unsigned)(16 * i) * SW_PITCH + (unsigned)(8 * j) * 4u);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc_row[i] = sw_mma(v[i][h], xj[t][j][h], acc_row[i]);
      }
      // ---- column part: rows 4 kk + kq, columns 2 mm (even block) and 2 mm + 1 (odd block): one ds_read_b64 = the A
      //      operands of both column blocks for one k-step; k-steps of either parity go to accumulators of their own
      //      (four independent chains), added at the end of the sub-tile
      if (both) {
        sw_f32x4 ce = {0.f, 0.f, 0.f, 0.f}, co = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
          const sw_f32x2 w0 = *reinterpret_cast<const sw_f32x2*>(rcol + (unsigned)(4 * kk) * SW_PITCH);
          const sw_f32x2 w1 = *reinterpret_cast<const sw_f32x2*>(rcol + (unsigned)(4 * kk + 4) * SW_PITCH);
          acc_col[t][0] = sw_mma(w0[0], xi[kk], acc_col[t][0]);
          acc_col[t][1] = sw_mma(w0[1], xi[kk], acc_col[t][1]);
          ce = sw_mma(w1[0], xi[kk + 1], ce);
          co = sw_mma(w1[1], xi[kk + 1], co);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc_col[t][0][r] += ce[r];
          acc_col[t][1][r] += co[r];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // the next turn overwrites the tile
    }
    // ---- the band's row sums: rowP[b][s][c][row0 + 16 i + 4 kq + r]   (D: lane holds column c = mm, rows 4 kq + r)
    if (cok) {
      float* rp = rowP + (((long)b * NS + s) * 16 + mm) * (long)N + row0 + 4 * kq;
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(acc_row[i], reinterpret_cast<sw_f32x4*>(rp + 16 * i));
    }
  }
  // ---- the tile's column sums: colP[b][I][c][col0 + 32 t + 8 kq + {e0, o0, e1, o1, e2, o2, e3, o3}]
  if (cok) {
    float* cp = colP + (((long)b * NT + I) * 16 + mm) * (long)N + col0 + 8 * kq;
#pragma unroll
    for (int t = 0; t < SW_NSUB; ++t) {
      if (col0 + t * SEG < N) {
        const sw_f32x4 lo = {acc_col[t][0][0], acc_col[t][1][0], acc_col[t][0][1], acc_col[t][1][1]};
        const sw_f32x4 hi = {acc_col[t][0][2], acc_col[t][1][2], acc_col[t][0][3], acc_col[t][1][3]};
        __builtin_nontemporal_store(lo, reinterpret_cast<sw_f32x4*>(cp + t * SEG));
        __builtin_nontemporal_store(hi, reinterpret_cast<sw_f32x4*>(cp + t * SEG + 4));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Workgroup-cooperative form (opts bit 0): THREE waves per SIMD.  A workgroup owns a super-tile = TR rows x 512 columns,
// its four waves four interleaved combs of 128 columns (4 sub-tiles each, 128 columns apart: 32 column accumulators
// instead of 64, one register buffer, <= 168 registers).  All four walk the same bands; at the end of a band each parks its 16 row accumulators in
// its own (free) LDS tile, the workgroup meets at a barrier and wave w adds up row block w of the four — one row partial
// per 512 columns instead of one per 256 (and per wave), one column partial per 128 columns and TR rows.
// ---------------------------------------------------------------------------------------------------------------
#ifndef XK_SW_PROBE
#define XK_SW_PROBE 0      // scripts/k1sw_probe.py builds 1 (no MFMA: traffic + LDS turn only) and 2 (no matrix loads) to time the halves
#endif
constexpr int SW7_NSUB = 4;
constexpr int SW7_WS = SW7_NSUB * 32;             // 128 columns per wave
constexpr int SW7_SS = 4 * SW7_WS;                // 512 columns per workgroup

__host__ __device__ inline int sw7_tiles_of_sstrip(int S, int N, int TR) {
  long top = (long)(S + 1) * SW7_SS;
  if (top > N) top = N;
  return (int)((top + TR - 1) / TR);
}

// PERSIST (round 5, opts bit 2): `gridDim.x` resident workgroups (three per compute unit of the stream's CU mask) take the
// super-tiles from a queue in global memory until it is empty, like the resident K1s launch (xk_symm.hip): the other batch
// group's launch, on its own stream, moves into the slots this launch's tail frees.  Partial slots are indexed by the
// super-tile, so which workgroup serves it does not enter the result.
template <int PRIO, bool PERSIST>
__global__ __launch_bounds__(256, 3) void dense_symm_wide7_kernel(
    const float* __restrict__ A, const float* __restrict__ X, float* __restrict__ rowP, float* __restrict__ colP,
    int N, int pc, long lda, long sA, long ldx, long sX, int NSS, int NT, int TR, int tiles_per_op,
    unsigned* __restrict__ queue, int nitems) {
  typedef float T;
  constexpr int SEG = 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_next[2];
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    if (threadIdx.x == 0) s_next[par] = (int)atomicAdd(queue, 1u);     // the super-tile after this one
  }
  // (the kernel has 168 registers for three waves per SIMD: everything derived from the lane index is rebuilt per
  //  super-tile instead of living across the queue loop — hoisted, it spilled 9 registers)
  int lane = lane0;
  if (PERSIST) asm volatile("" : "+v"(lane));
  const int b = __builtin_amdgcn_readfirstlane((int)((unsigned)item / (unsigned)tiles_per_op));
  int ti = __builtin_amdgcn_readfirstlane((int)((unsigned)item - (unsigned)b * tiles_per_op));
  // strip-major: consecutive workgroups walk down one 512-column super-strip (row-tile-major order — the workgroups in
  // flight covering whole matrix rows — measured the same: profiles/r04_k1sw_coop_pmc_probe.json)
  int S = 0;
  for (;;) {
    const int cnt = sw7_tiles_of_sstrip(S, N, TR);
    if (ti < cnt) break;
    ti -= cnt;
    ++S;
  }
  int I = ti;
  S = __builtin_amdgcn_readfirstlane(S);
  I = __builtin_amdgcn_readfirstlane(I);
  // this wave's columns: sub-tile t = 32 columns at sbase + 128 t + 32 wave — at any moment the four waves of the
  // workgroup (they walk in step) read 512 contiguous bytes of each matrix row
  const int sbase = S * SW7_SS;
  const int col0 = sbase + wave * 32;
  constexpr int TSTEP = 4 * 32;
  const int colb = sbase;
  const int r_begin = I * TR;
  int r_end = r_begin + TR;
  r_end = r_end < N ? r_end : N;
  r_end = r_end < (S + 1) * SW7_SS ? r_end : (S + 1) * SW7_SS; // the same band range for the four waves
  char* tile = smem + wave * SW_TILE_LDS;
  const float* Ab = A + (long)b * sA;
  const float* Xb = X + (long)b * sX;
  const int mm = lane & 15, kq = lane >> 4;
  const int lrow = lane >> 3, lcol = lane & 7;
  const bool cok = mm < pc;
  const float* Xc = Xb + (long)(cok ? mm : 0) * ldx;
  sw_f32x4 acc_col[SW7_NSUB][2];
#pragma unroll
  for (int t = 0; t < SW7_NSUB; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_col[t][h][r] = 0.f;
  const unsigned ldab = (unsigned)(lda * (long)sizeof(T));
  const unsigned lane_off = (unsigned)lrow * ldab + (unsigned)lcol * 16u;
  const unsigned st_off = (unsigned)lrow * SW_PITCH + (unsigned)lcol * 16u;
  constexpr unsigned POISON = 0x7ffffff0u;
  auto band_rsrc = [&](int row0) {
    return sw_rsrc(Ab + (long)row0 * lda + colb, ((long)(SW_ROWS - 1) * lda + (N - colb)) * (long)sizeof(T));
  };
  // sub-tile (row0, t) of this wave's strip: 8 loads + its x_J (8 values per lane); the same loads on every path
  auto issue = [&](sw_f32x4 (&a)[8], sw_f32x2 (&xj)[4], const SwRsrc& ra, int row0, int t, bool exists) {
    const int c0 = col0 + t * TSTEP;
    const bool live = exists && (c0 + SEG > row0) && (c0 < N);
    const int cj = live ? c0 : colb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const sw_f32x2 v = *reinterpret_cast<const sw_f32x2*>(Xc + cj + 8 * j + 2 * kq);
      xj[j] = sw_f32x2{(live && cok) ? v[0] : 0.f, (live && cok) ? v[1] : 0.f};
    }
    const unsigned base = live ? lane_off + (unsigned)(c0 - colb) * (unsigned)sizeof(T) : POISON;
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
      const unsigned voff = live ? base + (unsigned)(tt * 8) * ldab : POISON;
#if XK_SW_PROBE == 2
      a[tt] = sw_f32x4{(float)voff, (float)tt, (float)row0, (float)t};      // (probe: no matrix traffic)
#else
      a[tt] = __builtin_bit_cast(sw_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)voff, 0, 2));
#endif
    }
  };
  sw_f32x4 abuf[8];
  sw_f32x2 xjr[2][4];
  {
    const SwRsrc r0 = band_rsrc(r_begin);
    issue(abuf, xjr[0], r0, r_begin, 0, true);
  }
  for (int row0 = r_begin; row0 < r_end; row0 += SW_ROWS) {
    const SwRsrc ra = band_rsrc(row0);
    const bool more = row0 + SW_ROWS < r_end;
    const int rown = more ? row0 + SW_ROWS : row0;
    const SwRsrc rn = band_rsrc(rown);
    float xi[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float v = Xc[row0 + 4 * kk + kq];
      xi[kk] = cok ? v : 0.f;
    }
    sw_f32x4 acc_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_row[i][r] = 0.f;
#pragma unroll
    for (int t = 0; t < SW7_NSUB; ++t) {
      sw_f32x2(&xj)[4] = xjr[t & 1];
      const int c0 = col0 + t * TSTEP;
      const bool both = (c0 >= row0 + SW_ROWS) && (c0 < N);
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        char* p = tile + st_off + (unsigned)(tt * 8) * SW_PITCH;
        *reinterpret_cast<sw_f32x2*>(p) = sw_f32x2{abuf[tt][0], abuf[tt][1]};
        *reinterpret_cast<sw_f32x2*>(p + 8) = sw_f32x2{abuf[tt][2], abuf[tt][3]};
      }
      if (t + 1 < SW7_NSUB) issue(abuf, xjr[(t + 1) & 1], ra, row0, t + 1, true);
      else issue(abuf, xjr[(t + 1) & 1], rn, rown, 0, more);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if (PRIO) __builtin_amdgcn_s_setprio(1);
      const char* rrow = tile + (unsigned)mm * SW_PITCH + (unsigned)(2 * kq) * 4u;
      const char* rcol = tile + (unsigned)kq * SW_PITCH + (unsigned)(2 * mm) * 4u;
#if XK_SW_PROBE != 1
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sw_f32x2 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[i] = *reinterpret_cast<const sw_f32x2*>(rrow + (unsigned)(16 * i) * SW_PITCH + (unsigned)(8 * j) * 4u);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc_row[i] = sw_mma(v[i][h], xj[j][h], acc_row[i]);
      }
      if (both) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          const sw_f32x2 w0 = *reinterpret_cast<const sw_f32x2*>(rcol + (unsigned)(4 * kk) * SW_PITCH);
          acc_col[t][0] = sw_mma(w0[0], xi[kk], acc_col[t][0]);
          acc_col[t][1] = sw_mma(w0[1], xi[kk], acc_col[t][1]);
        }
      }
#endif
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- the four waves' row sums of this band through LDS: slot of wave w = its own tile, [row block i][lane] float4
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<sw_f32x4*>(tile + (unsigned)(i * 64 + lane) * 16u) = acc_row[i];
    __syncthreads();
    {
      sw_f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const sw_f32x4 v = *reinterpret_cast<const sw_f32x4*>(smem + w * SW_TILE_LDS + (unsigned)(wave * 64 + lane) * 16u);
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] += v[r];
      }
      if (cok) {
        float* rp = rowP + (((long)b * NSS + S) * 16 + mm) * (long)N + row0 + 16 * wave + 4 * kq;
        __builtin_nontemporal_store(sum, reinterpret_cast<sw_f32x4*>(rp));
      }
    }
    __syncthreads();                                           // the next LDS turn overwrites the slots
  }
  if (cok) {
    float* cp = colP + (((long)b * NT + I) * 16 + mm) * (long)N + col0 + 8 * kq;
#pragma unroll
    for (int t = 0; t < SW7_NSUB; ++t) {
      if (col0 + t * TSTEP < N) {
        const sw_f32x4 lo = {acc_col[t][0][0], acc_col[t][1][0], acc_col[t][0][1], acc_col[t][1][1]};
        const sw_f32x4 hi = {acc_col[t][0][2], acc_col[t][1][2], acc_col[t][0][3], acc_col[t][1][3]};
        __builtin_nontemporal_store(lo, reinterpret_cast<sw_f32x4*>(cp + t * TSTEP));
        __builtin_nontemporal_store(hi, reinterpret_cast<sw_f32x4*>(cp + t * TSTEP + 4));
      }
    }
  }
  if (!PERSIST) break;
  __syncthreads();                                             // s_next[par] is visible; LDS tiles are free again
  item = __builtin_amdgcn_readfirstlane(s_next[par]);
  par ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6 form (opts bit 3): the column part straight from the load registers, a ring of four blocks in flight.
//
// What the counters of the form above say (profiles/r04_k1sw_coop_pmc_probe.json, profiles/r06_k1sw_*.json): the matrix
// pipe is 0.48-0.54 busy with three waves per SIMD, waves wait for ISSUE, not for counters; in the instruction stream every
// operand of both products comes out of LDS (48 DS operations per 64 MFMAs, the column part as ds_read -> lgkmcnt(0) ->
// 4 MFMAs), and a wave has ONE 8 KB register buffer, re-issued only after the previous one went through its LDS turn.
//
// Here a load instruction fetches 4 rows x 256 B in the B-operand layout of the COLUMN part
//        lane (n = lane & 15, q = lane >> 4), load s:  A[rb + 4 q + s][c0 + 4 n .. 4 n + 3]
// so  y_J += A_IJ^T x_I  is  D[panel][column 4 n + e] += sum_q x[rb + 4 q + s][panel] * A[rb + 4 q + s][c0 + 4 n + e]:
// v_mfma_f32_16x16x4_f32 with the loaded register as B and x_I as A, 16 MFMAs per 16 x 64 block on four independent
// accumulators, no LDS, no wait but the load's own.  (The contraction order inside a block is free: rows rb + 4 q + s
// for fixed s share an instruction, which also makes x_I of a 16-row block ONE 16-byte load per lane.)  The row part
// needs the block with rows on the lane index: 4 ds_write_b128 + 4 ds_read_b128 per block (lane (m, q) reads row m,
// columns 4 (q + 4 u) .. + 3 = the A operands of four k-steps), a third of the DS operations, software-pipelined one
// block behind (the reads of block j - 1 are covered by the 16 column MFMAs of block j).  A ring of four blocks =
// 16 loads = 16 KB in flight per wave, two waves per SIMD (256 registers): 128 KB per compute unit, K1s' figure.
// Work split, partial slots, fold, diagonal rule (the 64 x 64 diagonal block: row part on both triangles, no column
// part; blocks left of it: out-of-range loads, no traffic, and — new — no MFMAs either) as in the form above.
// ---------------------------------------------------------------------------------------------------------------
#ifndef XK_SW8_PROBE
#define XK_SW8_PROBE 0     // 1: no MFMA (traffic + LDS turn + partials), 2: no matrix loads (MFMA + LDS), scripts/k1sw_probe.py
#endif
#ifndef XK_SW8_SWZ
#define XK_SW8_SWZ 1       // 1: 256-byte LDS rows, 16-byte slot index XOR row (no bank conflicts either way); 0: 272-byte pitch
#endif
#ifndef XK_SW8_PRIO
#define XK_SW8_PRIO 0      // 1: s_setprio 1 around the MFMA blocks (trial builds, scripts/k1sw_r06.py)
#endif
// blocks (16 rows x 64 columns, four 1 KB loads) in flight per wave.  (A ring of 6 — 24 KB per wave — was built in r06 with
// the band body instantiated three times, a band having 8 blocks and the ring index having to be static: 256 registers
// and 346 spills inside the stream; not kept.)
constexpr int SW8_RING = 4;
constexpr int SW8_PITCH = XK_SW8_SWZ ? 256 : 272;  // bytes per LDS row of a block (16 slots of 16 B)
constexpr int SW8_BUF = 16 * 272;                  // 4352 B: one block in row layout
constexpr int SW8_PARK = 4 * 64 * 16;              // the band's four row-sum blocks of this wave (4096 B)
constexpr int SW8_WAVE_LDS = 2 * SW8_BUF + SW8_PARK;   // 12800 B per wave, 51200 per workgroup

__global__ __launch_bounds__(256, 2) void dense_symm_wide8_kernel(
    const float* __restrict__ A, const float* __restrict__ X, float* __restrict__ rowP, float* __restrict__ colP,
    int N, int pc, long lda, long sA, long ldx, long sX, int NSS, int NT, int TR, int tiles_per_op) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int item = blockIdx.x;
  const int b = __builtin_amdgcn_readfirstlane((int)((unsigned)item / (unsigned)tiles_per_op));
  int ti = __builtin_amdgcn_readfirstlane((int)((unsigned)item - (unsigned)b * tiles_per_op));
  int S = 0;
  for (;;) {
    const int cnt = sw7_tiles_of_sstrip(S, N, TR);
    if (ti < cnt) break;
    ti -= cnt;
    ++S;
  }
  S = __builtin_amdgcn_readfirstlane(S);
  const int I = __builtin_amdgcn_readfirstlane(ti);
  // this wave's two column groups of 64: sbase + 64 wave and sbase + 256 + 64 wave (the four waves together: 2 x 1 KB
  // contiguous per matrix row)
  const int sbase = S * SW7_SS;
  const int cg0 = sbase + 64 * wave;
  const int r_begin = I * TR;
  int r_end = r_begin + TR;
  r_end = r_end < N ? r_end : N;
  r_end = r_end < (S + 1) * SW7_SS ? r_end : (S + 1) * SW7_SS;
  if (r_begin >= r_end) return;
  char* wbase = smem + wave * SW8_WAVE_LDS;
  char* park = wbase + 2 * SW8_BUF;
  const float* Ab = A + (long)b * sA;
  const float* Xb = X + (long)b * sX;
  const int nn = lane & 15, kq = lane >> 4;
  // (lanes of panel columns >= pc read column 0 instead: an output element (panel, .) depends on that panel's operand
  //  lanes only, and the stores below are guarded — no select after a load, which would put a wait next to it)
  const bool cok = nn < pc;
  const float* Xc = Xb + (long)(cok ? nn : 0) * ldx;
  const unsigned ldab = (unsigned)(lda * 4L);
  const unsigned lane_off = (unsigned)(4 * kq) * ldab + (unsigned)nn * 16u + (unsigned)wave * 256u;
  constexpr unsigned POISON = 0x7ffffff0u;
#if XK_SW8_SWZ
  // slot of (row r, 16-byte column chunk c) = c ^ r.  A ds_read_b128 is served in groups of 16 lanes made of 8 lanes of
  // one q and 8 of q ^ 1 whose row sets are {0-3, 12-15} and {4-11}: chunks q + 4 u and (q ^ 1) + 4 u differ in bit 0
  // only, so the 16 slots of a group are distinct; a ds_write_b128 group is 8 consecutive lanes of one row: 8 slots.
  const unsigned wr_off = (unsigned)(4 * kq) * SW8_PITCH + (unsigned)((nn ^ (4 * kq)) * 16);     // row 4 q + s: ^ s below
  const unsigned rd_off = (unsigned)nn * SW8_PITCH + (unsigned)((kq ^ (nn & 3)) * 16);           // chunk q + 4 u: ^ below
  const unsigned rd_hi = (unsigned)(nn & 12);
#else
  const unsigned wr_off = (unsigned)(4 * kq) * SW8_PITCH + (unsigned)nn * 16u;     // + s rows
  const unsigned rd_off = (unsigned)nn * SW8_PITCH + (unsigned)kq * 16u;           // + 64 u bytes
#endif

  // x_J of the wave's two groups in the row part's B-operand layout: xj[g][u] = X[panel nn][cg + 4 (kq + 4 u) .. + 3]
  sw_f32x4 xj[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = cg0 + 256 * g + 4 * (kq + 4 * u);
      const bool ok = cok && (cg0 + 256 * g < N);
      xj[g][u] = *reinterpret_cast<const sw_f32x4*>(Xc + (ok ? c : 0));
    }
  sw_f32x4 acc_col[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc_col[g][e] = sw_f32x4{0.f, 0.f, 0.f, 0.f};

  auto band_rsrc = [&](int row0) {
    return sw_rsrc(Ab + (long)row0 * lda + sbase, ((long)(SW_ROWS - 1) * lda + (N - sbase)) * 4L);
  };
  // the band's x_I in the column part's A-operand layout: xi[i] = X[panel nn][row0 + 16 i + 4 kq .. + 3]
  auto load_xi = [&](sw_f32x4 (&xi)[4], int row0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) xi[i] = *reinterpret_cast<const sw_f32x4*>(Xc + row0 + 16 * i + 4 * kq);
  };
  // block j of a band = row block i = j >> 1, column group g = j & 1: four loads of 4 rows x 256 B
  auto issue = [&](sw_f32x4 (&a)[4], const SwRsrc& ra, unsigned vo, int j) {
    const int i = j >> 1, g = j & 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#if XK_SW8_PROBE == 2
      a[s] = sw_f32x4{(float)vo, (float)s, (float)j, 1.f};
#else
      const unsigned so = (unsigned)(16 * i + s) * ldab + (unsigned)g * 1024u;
      a[s] = __builtin_bit_cast(sw_f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (int)vo, (int)so, 2));
#endif
    }
  };
  // per band: which of the wave's two groups exist (on / right of the diagonal block) and which carry the column part
  auto live_of = [&](int row0, int g) { const int c0 = cg0 + 256 * g; return (c0 >= row0) && (c0 < N); };
  auto both_of = [&](int row0, int g) { const int c0 = cg0 + 256 * g; return (c0 >= row0 + SW_ROWS) && (c0 < N); };

  constexpr int RING = SW8_RING;
  static_assert(8 % RING == 0, "a band has 8 blocks: the ring index must be static");
  sw_f32x4 ring[RING][4];
  sw_f32x4 xi[4];
  {
    const SwRsrc r0 = band_rsrc(r_begin);
    load_xi(xi, r_begin);
#pragma unroll
    for (int j = 0; j < RING; ++j)
      issue(ring[j], r0, live_of(r_begin, j & 1) ? lane_off : POISON, j);
  }
  for (int row0 = r_begin; row0 < r_end; row0 += SW_ROWS) {
    const SwRsrc ra = band_rsrc(row0);
    const bool more = row0 + SW_ROWS < r_end;
    const int rown = more ? row0 + SW_ROWS : row0;
    const SwRsrc rn = band_rsrc(rown);
    bool live[2], both[2];
    unsigned vo[2], von[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      live[g] = live_of(row0, g);
      both[g] = both_of(row0, g);
      vo[g] = live[g] ? lane_off : POISON;
      von[g] = (more && live_of(rown, g)) ? lane_off : POISON;
    }
    sw_f32x4 acc_row[2];
    sw_f32x4 w[4];
#pragma unroll
    for (int j = 0; j <= 8; ++j) {
      // ---- (1) block j - 1 back from LDS in row layout: lane (m = nn, q) <- row m, columns 4 (q + 4 u) .. + 3
      const bool rowpart = j > 0 && live[(j - 1) & 1];
      if (rowpart) {
        const char* rb = wbase + ((j - 1) & 1) * SW8_BUF + rd_off;
#pragma unroll
#if XK_SW8_SWZ
        for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const sw_f32x4*>(rb + ((unsigned)(4 * u) ^ rd_hi) * 16u);
#else
        for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const sw_f32x4*>(rb + 64 * u);
#endif
      }
      if (j < 8) {
        const int i = j >> 1, g = j & 1;
        sw_f32x4(&blk)[4] = ring[j % RING];
        if (live[g]) {
          // ---- (2) block j into LDS: lane (n, q), load s -> row 4 q + s, slot n
          char* wb = wbase + (j & 1) * SW8_BUF;
#pragma unroll
#if XK_SW8_SWZ
          for (int s = 0; s < 4; ++s) *reinterpret_cast<sw_f32x4*>(wb + (wr_off ^ (unsigned)(16 * s)) + s * SW8_PITCH) = blk[s];
#else
          for (int s = 0; s < 4; ++s) *reinterpret_cast<sw_f32x4*>(wb + wr_off + s * SW8_PITCH) = blk[s];
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
#if XK_SW8_PROBE != 1
        if (both[g]) {
          // ---- (3) column part of block j from the load registers: k-step s, four column phases e
          if (XK_SW8_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc_col[g][e] = sw_mma(xi[i][s], blk[s][e], acc_col[g][e]);
          if (XK_SW8_PRIO) __builtin_amdgcn_s_setprio(0);
        }
#else
        if (both[g]) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc_col[g][e][0] += blk[e][e] * xi[i][e];
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        // ---- (4) the ring slot is free: block j + RING (of this band, or of the next one)
        if (j + RING < 8) issue(blk, ra, vo[(j + RING) & 1], j + RING);
        else issue(blk, rn, von[(j + RING) & 1], j + RING - 8);
        // x_I of row block i is dead after the column MFMAs of its second block: the next band's goes straight into the
        // same registers.  In the in-order load queue it sits before the next band's block 2 i (issued RING - 2 steps from
        // now), which is the block that needs it: no wait is added, no second buffer needed.
        if (g == 1) xi[i] = *reinterpret_cast<const sw_f32x4*>(Xc + rown + 16 * i + 4 * kq);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- (5) row part of block j - 1: k-steps (u, e), two alternating accumulator chains
      if (j > 0) {
        const int ip = (j - 1) >> 1, gp = (j - 1) & 1;
        if (gp == 0) {
          acc_row[0] = sw_f32x4{0.f, 0.f, 0.f, 0.f};
          acc_row[1] = sw_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (rowpart) {
#if XK_SW8_PROBE != 1
          if (XK_SW8_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc_row[e & 1] = sw_mma(w[u][e], xj[gp][u][e], acc_row[e & 1]);
          if (XK_SW8_PRIO) __builtin_amdgcn_s_setprio(0);
#else
#pragma unroll
          for (int u = 0; u < 4; ++u) acc_row[u & 1][u] += w[u][u] * xj[gp][u][u];
#endif
        }
        if (gp == 1) {
          sw_f32x4 sum;
#pragma unroll
          for (int r = 0; r < 4; ++r) sum[r] = acc_row[0][r] + acc_row[1][r];
          *reinterpret_cast<sw_f32x4*>(park + (unsigned)(ip * 64 + lane) * 16u) = sum;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- the four waves' row sums of this band: wave w adds up row block w of the four parks
    __syncthreads();
    {
      sw_f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const sw_f32x4 v = *reinterpret_cast<const sw_f32x4*>(smem + q * SW8_WAVE_LDS + 2 * SW8_BUF +
                                                               (unsigned)(wave * 64 + lane) * 16u);
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[r] += v[r];
      }
      if (cok) {
        float* rp = rowP + (((long)b * NSS + S) * 16 + nn) * (long)N + row0 + 16 * wave + 4 * kq;
        __builtin_nontemporal_store(sum, reinterpret_cast<sw_f32x4*>(rp));
      }
    }
    __syncthreads();
  }
  // ---- the tile's column sums: lane (n, q), accumulator e, register r = panel 4 q + r, column cg + 4 n + e
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int c0 = cg0 + 256 * g;
    if (c0 < N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int panel = 4 * kq + r;
        if (panel < pc) {
          float* cp = colP + (((long)b * NT + I) * 16 + panel) * (long)N + c0 + 4 * nn;
          const sw_f32x4 v = {acc_col[g][0][r], acc_col[g][1][r], acc_col[g][2][r], acc_col[g][3][r]};
          __builtin_nontemporal_store(v, reinterpret_cast<sw_f32x4*>(cp));
        }
      }
    }
  }
}

// Y[b][c][n] = sum of the row partials of the strips that hold row n (strip n / WS and every strip right of it) and of
// the column partials of the row tiles that hold column n (tiles 0 .. n / TR), each list in ascending order
__global__ __launch_bounds__(256) void symm_wide_fold(const float* __restrict__ rowP, const float* __restrict__ colP,
                                                       float* __restrict__ Y, int N, int pc, int NS, int NT, int TR,
                                                       int WS, long ldy, long sY, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;       // over B * pc * N
  if (idx >= total) return;
  const long per_b = (long)pc * N;
  const long b = idx / per_b;
  const long rem = idx - b * per_b;
  const int c = (int)(rem / N);
  const int n = (int)(rem - (long)c * N);
  // (eight loads in flight per thread: a serial chain of ~100 dependent strided reads ran at 3.5 TB/s)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const float* rp = rowP + (b * NS * 16 + c) * (long)N + n;
  const long rstep = 16L * N;
  int s = n / WS;
  for (; s + 4 <= NS; s += 4) {
    const float v0 = __builtin_nontemporal_load(rp + s * rstep), v1 = __builtin_nontemporal_load(rp + (s + 1) * rstep);
    const float v2 = __builtin_nontemporal_load(rp + (s + 2) * rstep), v3 = __builtin_nontemporal_load(rp + (s + 3) * rstep);
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; s < NS; ++s) a0 += __builtin_nontemporal_load(rp + s * rstep);
  const float* cp = colP + (b * NT * 16 + c) * (long)N + n;
  const int I1 = n / TR;
  int I = 0;
  for (; I + 4 <= I1 + 1; I += 4) {
    const float v0 = __builtin_nontemporal_load(cp + I * rstep), v1 = __builtin_nontemporal_load(cp + (I + 1) * rstep);
    const float v2 = __builtin_nontemporal_load(cp + (I + 2) * rstep), v3 = __builtin_nontemporal_load(cp + (I + 3) * rstep);
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; I <= I1; ++I) a1 += __builtin_nontemporal_load(cp + I * rstep);
  const float sum = (a0 + a1) + (a2 + a3);
  Y[b * sY + (long)c * ldy + n] = sum;
}

static int symm_wide_tr(int N) { return N >= 4096 ? 512 : 256; }

#ifndef XK_SW_TR_BIG
#define XK_SW_TR_BIG 1024  // rows of a super-tile from order 8192 on (trial builds: 512 / 2048)
#endif
static int symm_wide7_tr(int N) { return N >= 8192 ? XK_SW_TR_BIG : (N >= 2048 ? 512 : 256); }

static int symm_wide(const float* A, const float* X, float* Y, float* ws, long ws_elems, int B, int N, int P, long lda,
                     long sA, long ldx, long sX, long ldy, long sY, int opts, int phase, hipStream_t st) {
  if (opts & 1) {
    // workgroup-cooperative form
    if (P < 1 || P > 16) return XK_ERR_ARG;
    if ((N % SW_ROWS) || (lda % 4) || (sA % 4) || (ldx % 2) || (sX % 2) || ((uintptr_t)A & 15) || ((uintptr_t)X & 7) ||
        ((uintptr_t)ws & 15))
      return XK_ERR_UNSUPPORTED;
    if ((long)SW_ROWS * lda * 4 > 0x7fffffe0L) return XK_ERR_UNSUPPORTED;
    const int TR = symm_wide7_tr(N);
    const int NSS = (N + SW7_SS - 1) / SW7_SS, NT = (N + TR - 1) / TR;
    const long nrow = (long)B * NSS * 16 * N, ncol = (long)B * NT * 16 * N;
    if (ws == nullptr || ws_elems < nrow + ncol) return XK_ERR_ARG;
    float* rowP = ws;
    float* colP = ws + nrow;
    if (phase != 2) {
      int tiles = 0;
      for (int S = 0; S < NSS; ++S) tiles += sw7_tiles_of_sstrip(S, N, TR);
      const size_t lds = 4 * (size_t)SW_TILE_LDS;
      const long nitems_l = (long)B * tiles;
      if (nitems_l > 0x7fffffffL) return XK_ERR_UNSUPPORTED;
      const int nitems = (int)nitems_l;
      if (opts & 8) {
        static bool opted = false;
        if (!opted) {
          (void)hipFuncSetAttribute((const void*)dense_symm_wide8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    4 * SW8_WAVE_LDS);
          opted = true;
        }
        hipLaunchKernelGGL(dense_symm_wide8_kernel, dim3((unsigned)nitems), dim3(256), 4 * (size_t)SW8_WAVE_LDS, st, A, X,
                           rowP, colP, N, P, lda, sA, ldx, sX, NSS, NT, TR, tiles);
      } else if (opts & 4) {
        // resident launch: queue word in the last 64 bytes of the workspace, reset in stream order before the launch
        if (ws_elems < nrow + ncol + SW_QUEUE_ELEMS) return XK_ERR_ARG;
        unsigned* queue = reinterpret_cast<unsigned*>(ws + (ws_elems - SW_QUEUE_ELEMS));
        int nslots = (opts >> 16) & 0xfff;
        if (nslots == 0) {
          int dev = 0, cus = 256;
          if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
          nslots = 3 * (cus > 0 ? cus : 256);
        }
        hipError_t me = hipMemsetAsync(queue, 0, 64, st);
        if (me != hipSuccess) return (int)me;
        auto kern = (opts & 2) ? dense_symm_wide7_kernel<1, true> : dense_symm_wide7_kernel<0, true>;
        hipLaunchKernelGGL(kern, dim3((unsigned)(nitems < nslots ? nitems : nslots)), dim3(256), lds, st, A, X, rowP, colP,
                           N, P, lda, sA, ldx, sX, NSS, NT, TR, tiles, queue, nitems);
      } else {
        auto kern = (opts & 2) ? dense_symm_wide7_kernel<1, false> : dense_symm_wide7_kernel<0, false>;
        hipLaunchKernelGGL(kern, dim3((unsigned)nitems), dim3(256), lds, st, A, X, rowP, colP, N, P, lda, sA,
                           ldx, sX, NSS, NT, TR, tiles, (unsigned*)nullptr, nitems);
      }
      XK_LAUNCH_CHECK();
    }
    if (phase != 1) {
      const long total = (long)B * P * N;
      hipLaunchKernelGGL(symm_wide_fold, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, rowP, colP, Y, N, P,
                         NSS, NT, TR, SW7_SS, ldy, sY, total);
      XK_LAUNCH_CHECK();
    }
    return XK_OK;
  }
  constexpr int WS = SW_NSUB * (SW_SEG_BYTES / 4);
  if (P < 1 || P > 16) return XK_ERR_ARG;
  if ((N % SW_ROWS) || (lda % 4) || (sA % 4) || (ldx % 2) || (sX % 2) || ((uintptr_t)A & 15) || ((uintptr_t)X & 7) ||
      ((uintptr_t)ws & 15))
    return XK_ERR_UNSUPPORTED;
  if ((long)SW_ROWS * lda * 4 > 0x7fffffe0L) return XK_ERR_UNSUPPORTED;
  const int TR = symm_wide_tr(N);
  const int NS = (N + WS - 1) / WS, NT = (N + TR - 1) / TR;
  const long nrow = (long)B * NS * 16 * N, ncol = (long)B * NT * 16 * N;
  if (ws == nullptr || ws_elems < nrow + ncol) return XK_ERR_ARG;
  float* rowP = ws;
  float* colP = ws + nrow;
  if (phase != 2) {
    int tiles = 0;
    for (int s = 0; s < NS; ++s) tiles += sw_tiles_of_strip(s, N, WS, TR);
    const long total = (long)B * tiles;
    const size_t lds = 4 * (size_t)SW_WAVE_LDS;
    hipLaunchKernelGGL(dense_symm_wide_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), lds, st, A, X, rowP, colP,
                       N, P, lda, sA, ldx, sX, NS, NT, TR, tiles, total);
    XK_LAUNCH_CHECK();
  }
  if (phase != 1) {
    const long total = (long)B * P * N;
    hipLaunchKernelGGL(symm_wide_fold, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, rowP, colP, Y, N, P,
                       NS, NT, TR, WS, ldy, sY, total);
    XK_LAUNCH_CHECK();
  }
  return XK_OK;
}

}  // namespace xk

extern "C" {

// workspace (elements): row partials (B, NS, 16, N) + column partials (B, NT, 16, N)
long xk_dense_symm_wide_workspace_elems(int B, int N) {
  constexpr int WS = xk::SW_NSUB * (xk::SW_SEG_BYTES / 4);
  const int TR = xk::symm_wide_tr(N);
  const long NS = (N + WS - 1) / WS, NT = (N + TR - 1) / TR;
  return (long)B * (NS + NT) * 16 * N + xk::SW_QUEUE_ELEMS;     // + the queue of the resident launch (last 64 bytes)
}

int xk_dense_symm_wide_f32(const float* A, const float* X, float* Y, float* ws, long ws_elems, int B, int N, int P,
                           long lda, long sA, long ldx, long sX, long ldy, long sY, int opts, void* stream) {
  if (B < 0 || N < 0 || opts < 0 || (opts & 0xfff0) != 0 || opts > 0x0fffffff || ((opts & 12) && !(opts & 1)) || (opts & 12) == 12) return XK_ERR_ARG;
  if (B == 0 || N == 0) return XK_OK;
  return xk::symm_wide(A, X, Y, ws, ws_elems, B, N, P, lda, sA, ldx, sX, ldy, sY, opts, 0, (hipStream_t)stream);
}

// the two halves as separate launches (the eigensolver's two-group pipeline: tiles on the CU-masked panel stream, the
// fold on the group's own stream); `ws` must stay untouched in between
int xk_dense_symm_wide_tiles_f32(const float* A, const float* X, float* ws, long ws_elems, int B, int N, int P,
                                 long lda, long sA, long ldx, long sX, int opts, void* stream) {
  if (B < 0 || N < 0 || opts < 0 || (opts & 0xfff0) != 0 || opts > 0x0fffffff || ((opts & 12) && !(opts & 1)) || (opts & 12) == 12) return XK_ERR_ARG;
  if (B == 0 || N == 0) return XK_OK;
  return xk::symm_wide(A, X, nullptr, ws, ws_elems, B, N, P, lda, sA, ldx, sX, 0, 0, opts, 1, (hipStream_t)stream);
}

int xk_dense_symm_wide_fold_f32(float* Y, const float* ws, long ws_elems, int B, int N, int P, long ldy, long sY,
                                int opts, void* stream) {
  if (B < 0 || N < 0 || opts < 0 || (opts & 0xfff0) != 0 || opts > 0x0fffffff || ((opts & 12) && !(opts & 1)) || (opts & 12) == 12) return XK_ERR_ARG;
  if (B == 0 || N == 0) return XK_OK;
  return xk::symm_wide((const float*)ws, (const float*)ws, Y, (float*)ws, ws_elems, B, N, P, N, 0, N, 0, ldy, sY, opts,
                       2, (hipStream_t)stream);
}

}  // extern "C"
