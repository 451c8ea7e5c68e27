// Micro-benchmark: how fast can gfx950 stream a (B, N, N) fp64 matrix once with different per-wave access patterns?
// (decides the tile walk of the symmetric panel product; no arithmetic beyond a checksum that keeps the loads alive)
//   hipcc --offload-arch=gfx950 -O3 -o stream_patterns stream_patterns.hip && ./stream_patterns [B N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t Rsrc;

__device__ __forceinline__ Rsrc mk(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ d2 ld(Rsrc r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 2));
}

// MODE 0: "rowline"  — a wave reads NU KB of one row per step (lane: NU 16 B vectors 1 KB apart), walks down TR rows,
//                      ring of R rows in flight; block = W waves on adjacent column ranges        (xk_symm.hip's walk)
// MODE 1: "colsweep" — a wave reads 64 rows x 128 B per step (8 loads of 8 rows x 128 B), walks DOWN TR rows, then takes
//                      its next 128 B strip (strips w, w+W, ...)                                   (xk_symm2.hip's walk)
// MODE 2: "rowsweep" — same 64 x 128 B steps, but the wave walks ALONG the row: its 64 rows, consecutive 128 B
//                      segments of a CW-byte wide tile                                            (xk_rowswide.hip's walk)
// FLAGS: 1 = only the tiles on/above the block diagonal (the symmetric kernels' tile list), 2 = per-block prologue and
// epilogue traffic of the symmetric kernels (stage TR x 6 panel values into LDS behind a barrier, write TR x 6 + TCB/8 x 6
// partial sums at the end), 4 = park every loaded sub-tile in LDS (ds_write_b128 + wait) before it is consumed
template <int MODE, int W, int DEPTH, int FLAGS>
__global__ __launch_bounds__(64 * W) void stream(const double* __restrict__ A, double* __restrict__ out, int N,
                                                 int TR, int TCB, int tiles_r, int tiles_c, const double* __restrict__ X,
                                                 double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int t = blockIdx.x;
  int b, I, J;
  if (FLAGS & 1) {
    // tiles_r * tiles_c is passed as the per-operator count of upper tiles in tiles_c's high half: see run()
    const int ntri = tiles_c >> 16; tiles_c &= 0xffff;
    b = t / ntri; t -= b * ntri;
    const int cpt = TCB / 8;                 // columns per tile
    I = 0;
    for (;; ++I) { const int jmin = (I * TR) / cpt; const int cnt = tiles_c - jmin; if (t < cnt) { J = jmin + t; break; } t -= cnt; }
  } else {
    b = t / (tiles_r * tiles_c);
    t -= b * tiles_r * tiles_c;
    I = t / tiles_c; J = t - I * tiles_c;
  }
  double* lds = reinterpret_cast<double*>(smem);
  if (FLAGS & 2) {
    for (int idx = threadIdx.x; idx < TR * 6; idx += 64 * W) { const int c = idx / TR, r = idx - c * TR; lds[r * 6 + c] = X[((long)b * 6 + c) * N + I * TR + r]; }
    __syncthreads();
  }
  char* wtile = smem + TR * 6 * 8 + wave * 9216;
  const unsigned ldab = (unsigned)N * 8u;
  const char* base = (const char*)(A + ((long)b * N + (long)I * TR) * N) + (long)J * TCB;
  const Rsrc rs = mk(base, (unsigned)(TR - 1) * ldab + (unsigned)TCB);
  d2 acc = {0.0, 0.0};
  if (MODE == 0) {
    constexpr int NU = 2;
    const int wbytes = TCB / W;                     // bytes per wave and row (NU KB)
    unsigned off[NU];
    for (int u = 0; u < NU; ++u) off[u] = wave * wbytes + u * 1024 + lane * 16;
    d2 ring[DEPTH][NU];
#pragma unroll
    for (int r = 0; r < DEPTH; ++r)
#pragma unroll
      for (int u = 0; u < NU; ++u) ring[r][u] = ld(rs, off[u], (unsigned)r * ldab);
    for (int i0 = 0; i0 < TR; i0 += DEPTH) {
#pragma unroll
      for (int r = 0; r < DEPTH; ++r) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          acc += ring[r][u];
          int row = i0 + DEPTH + r;
          row = row < TR ? row : TR - 1;
          ring[r][u] = ld(rs, off[u], (unsigned)row * ldab);
        }
      }
    }
  } else {
    const int lrow = lane >> 3, lcol = lane & 7;
    unsigned rowpart[8];
    for (int k = 0; k < 8; ++k) rowpart[k] = (unsigned)(k * 8 + lrow) * ldab + lcol * 16;
    if (MODE == 3)   // MFMA A-operand layout straight from memory: lane -> row (lane & 15) of a 16-row block, 16 B chunk
      for (int k = 0; k < 8; ++k)   // (lane >> 4) + 4 * (k & 1): an instruction covers 16 rows x 64 B, a pair the full 128 B lines
        rowpart[k] = (unsigned)((k >> 1) * 16 + (lane & 15)) * ldab + (unsigned)((lane >> 4) + 4 * (k & 1)) * 16;
    const int nstrip = TCB / 128, nsub = TR / 64;
    const int per_wave = nstrip / W;
    const int steps = per_wave * nsub;
    d2 buf[DEPTH][8];
    auto stepoff = [&](int n) -> unsigned {
      int k, sub;
      if (MODE == 1 || MODE == 3) { k = n / nsub; sub = n - k * nsub; return (unsigned)(wave + k * W) * 128u + (unsigned)sub * 64u * ldab; }
      // MODE 2: rows fixed per (wave, sub-block), walk along the row
      sub = n / per_wave; k = n - sub * per_wave;
      return (unsigned)(wave * per_wave + k) * 128u + (unsigned)sub * 64u * ldab;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const unsigned so = stepoff(d < steps ? d : steps - 1);
#pragma unroll
      for (int k = 0; k < 8; ++k) buf[d][k] = ld(rs, rowpart[k] + so, 0);
    }
    for (int n0 = 0; n0 < steps; n0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        int nn = n0 + d + DEPTH;
        nn = nn < steps ? nn : steps - 1;
        const unsigned so = stepoff(nn);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (FLAGS & 4) *reinterpret_cast<d2*>(wtile + (k * 8 + lrow) * 144 + lcol * 16) = buf[d][k];
          else acc += buf[d][k];
          buf[d][k] = ld(rs, rowpart[k] + so, 0);
        }
        if (FLAGS & 4) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_s_waitcnt(0xc07f);
          acc += *reinterpret_cast<const d2*>(wtile + (lane & 15) * 144 + (lane >> 4) * 16);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    }
  }
  if (FLAGS & 2) {
    __syncthreads();
    const int cpt = TCB / 8;
    double* pr = part + (long)blockIdx.x * (TR + cpt) * 6;
    for (int idx = threadIdx.x; idx < (TR + cpt) * 6; idx += 64 * W) pr[idx] = lds[idx % (TR * 6)] + acc[0];
  }
  if (acc[0] + acc[1] == 12345.678) out[blockIdx.x] = acc[0];
}

static const double* gX; static double* gPart;
template <int MODE, int W, int DEPTH, int FLAGS = 0>
static void run(const char* name, const double* A, double* out, int B, int N, int TR, int TCB, int lds = 0) {
  // lds: dynamic LDS per block, only there to cap the number of resident blocks per CU (160 KB)
  hipFuncSetAttribute((const void*)stream<MODE, W, DEPTH, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int tiles_r = N / TR;
  int tiles_c = (N * 8) / TCB;
  int grid = B * tiles_r * tiles_c;
  double bytes = (double)B * N * N * 8.0;
  if (FLAGS & 1) {
    int ntri = 0;
    for (int I = 0; I < tiles_r; ++I) ntri += tiles_c - (I * TR) / (TCB / 8);
    grid = B * ntri;
    bytes = (double)grid * TR * TCB;
    tiles_c |= ntri << 16;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  stream<MODE, W, DEPTH, FLAGS><<<grid, 64 * W, lds>>>(A, out, N, TR, TCB, tiles_r, tiles_c, gX, gPart);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 3;
  for (int r = 0; r < reps; ++r) stream<MODE, W, DEPTH, FLAGS><<<grid, 64 * W, lds>>>(A, out, N, TR, TCB, tiles_r, tiles_c, gX, gPart);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("{\"pattern\": \"%s\", \"waves_per_block\": %d, \"depth\": %d, \"tile_rows\": %d, \"tile_bytes_per_row\": %d, "
         "\"blocks\": %d, \"lds_cap\": %d, \"ms\": %.3f, \"TBps\": %.3f}\n", name, W, DEPTH, TR, TCB, grid, lds, ms, bytes / ms / 1e9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 16, N = argc > 2 ? atoi(argv[2]) : 16384;
  double* A; double* out;
  const size_t bytes = (size_t)B * N * N * 8;
  hipMalloc(&A, bytes); hipMalloc(&out, 1 << 24);
  hipMemset(A, 0, bytes);
  { double* x; hipMalloc(&x, (size_t)B * 6 * N * 8); hipMemset(x, 0, (size_t)B * 6 * N * 8); gX = x; hipMalloc(&gPart, (size_t)1 << 30); }
  if (argc > 3 && argv[3][0] == 's') {   // what the symmetric kernels add to the bare stream, one ingredient at a time
    const int L = 120 * 1024;
    run<1, 8, 2, 0>("colsweep 8 waves, 1 block/CU", A, out, B, N, 512, 8192, L);
    run<3, 8, 2, 0>("colsweep, loads in the MFMA operand layout (16 rows x 64 B per instruction), 8 waves 1 block/CU", A, out, B, N, 512, 8192, L);
    run<3, 4, 2, 0>("same, 4 waves x 2 blocks/CU", A, out, B, N, 512, 8192, 80 * 1024);
    run<3, 4, 2, 3>("same, 4 waves x 2 blocks/CU + triangle + prologue/epilogue", A, out, B, N, 512, 8192, 80 * 1024);
    run<3, 4, 2, 3>("same, 4 waves x 3 blocks/CU + triangle + prologue/epilogue", A, out, B, N, 512, 8192, 53 * 1024);
    run<3, 4, 2, 3>("same, 4 waves x 3 blocks/CU, 1024-row tiles", A, out, B, N, 1024, 8192, 53 * 1024);
    run<1, 8, 2, 1>("+ triangle tile list", A, out, B, N, 512, 8192, L);
    run<1, 8, 2, 3>("+ triangle + prologue/epilogue", A, out, B, N, 512, 8192, L);
    run<1, 8, 2, 7>("+ triangle + prologue/epilogue + LDS park", A, out, B, N, 512, 8192, L);
    run<1, 8, 2, 4>("colsweep + LDS park only", A, out, B, N, 512, 8192, L);
    run<1, 4, 2, 7>("4 waves x 2 blocks/CU: triangle + prologue/epilogue + LDS park", A, out, B, N, 512, 8192, 80 * 1024);
    run<1, 4, 2, 7>("4 waves x 3 blocks/CU, 256-row tiles: same", A, out, B, N, 256, 8192, 53 * 1024);
    run<0, 4, 8, 1>("rowline 2 blocks/CU + triangle", A, out, B, N, 1024, 8192, 80 * 1024);
    run<0, 4, 8, 3>("rowline 2 blocks/CU + triangle + prologue/epilogue", A, out, B, N, 1024, 8192, 80 * 1024);
    return 0;
  }
  if (argc > 3) {   // occupancy study: blocks per CU capped through the LDS allocation
    run<1, 8, 2>("colsweep 8 waves depth 2 (16 KB/wave), 1 block/CU", A, out, B, N, 512, 8192, 120 * 1024);
    run<1, 8, 4>("colsweep 8 waves depth 4 (32 KB/wave), 1 block/CU", A, out, B, N, 512, 8192, 120 * 1024);
    run<1, 4, 2>("colsweep 4 waves depth 2, 2 blocks/CU", A, out, B, N, 512, 8192, 80 * 1024);
    run<1, 4, 4>("colsweep 4 waves depth 4, 2 blocks/CU", A, out, B, N, 512, 8192, 80 * 1024);
    run<1, 4, 2>("colsweep 4 waves depth 2, 3 blocks/CU", A, out, B, N, 512, 8192, 53 * 1024);
    run<1, 4, 2>("colsweep 4 waves depth 2, 4 blocks/CU", A, out, B, N, 512, 8192, 40 * 1024);
    run<0, 4, 8>("rowline ring 8 rows (16 KB/wave), 2 blocks/CU", A, out, B, N, 1024, 8192, 80 * 1024);
    run<0, 4, 16>("rowline ring 16 rows (32 KB/wave), 2 blocks/CU", A, out, B, N, 1024, 8192, 80 * 1024);
    run<0, 4, 8>("rowline ring 8 rows, 3 blocks/CU", A, out, B, N, 1024, 8192, 53 * 1024);
    run<0, 4, 8>("rowline ring 8 rows, 4 blocks/CU", A, out, B, N, 1024, 8192, 40 * 1024);
    run<2, 4, 2>("rowsweep full row 4 waves depth 2, 2 blocks/CU", A, out, B, N, 256, N * 8, 80 * 1024);
    run<2, 4, 1>("rowsweep full row 4 waves depth 1, 3 blocks/CU", A, out, B, N, 256, N * 8, 53 * 1024);
    return 0;
  }
  run<0, 4, 8>("rowline 2KB/wave/row, ring 8 rows", A, out, B, N, 1024, 8192);
  run<0, 4, 16>("rowline 2KB/wave/row, ring 16 rows", A, out, B, N, 1024, 8192);
  run<0, 4, 8>("rowline, tile 512 rows", A, out, B, N, 512, 8192);
  run<1, 8, 2>("colsweep 64x128B, 8 waves", A, out, B, N, 512, 8192);
  run<1, 4, 2>("colsweep 64x128B, 4 waves", A, out, B, N, 512, 8192);
  run<1, 4, 4>("colsweep 64x128B, 4 waves depth 4", A, out, B, N, 512, 8192);
  run<1, 8, 2>("colsweep tile 1024 rows", A, out, B, N, 1024, 8192);
  run<2, 8, 2>("rowsweep 64x128B, 8 waves x 64 rows, 8 KB wide", A, out, B, N, 512, 8192);
  run<2, 4, 2>("rowsweep 4 waves x 64 rows, 32 KB wide", A, out, B, N, 256, 32768);
  run<2, 4, 2>("rowsweep 4 waves x 64 rows, full row", A, out, B, N, 256, N * 8);
  run<2, 4, 4>("rowsweep 4 waves, full row, depth 4", A, out, B, N, 256, N * 8);
  run<2, 4, 1>("rowsweep 4 waves, full row, depth 1", A, out, B, N, 256, N * 8);
  return 0;
}
