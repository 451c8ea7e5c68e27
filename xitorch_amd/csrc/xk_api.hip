// xitorch_amd :: ABI bookkeeping
#include "xk_common.h"

extern "C" int xk_abi_version(void) { return 1; }

// ---------------------------------------------------------------------------------------------
// Stream restricted to a subset of the compute units.
//
// The operator-panel product is HBM-bound: it runs as fast on 192-224 of the 256 CUs as on all of them.
// Launching it on a stream whose CU mask leaves `reserve_cus` CUs out keeps those CUs free for the
// latency-bound small kernels of the OTHER half of the batch (LDS-resident Jacobi eigensolver: one
// workgroup with ~100 KB of LDS per matrix, which cannot start on a CU already holding two panel-product
// blocks), so the two really overlap.  The stream lives until xk_stream_destroy / process exit.
// ---------------------------------------------------------------------------------------------
// pattern 0: the LAST `reserve_cus` bits of the linear CU mask are cleared (what ships); pattern 1: every
// (ncu / reserve_cus)-th bit is cleared instead.  Measured with xk_probe_xcc below (profiles/r05_cu_mask_probe.jsonl): the
// driver deals the linear mask out over the XCDs bit by bit, so pattern 0 takes reserve / 8 units from every XCD, pattern 1
// all of them from one XCD — and a mask that leaves an XCD without any unit is not honoured (all units stay usable).
extern "C" int xk_stream_create_cu_masked_pattern(int device, int reserve_cus, int pattern, void** stream_out) {
  if (!stream_out || reserve_cus < 0 || pattern < 0 || pattern > 2) return XK_ERR_ARG;
  if (pattern == 2 && reserve_cus == 0) return XK_ERR_ARG;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return (int)e;
  const int ncu = prop.multiProcessorCount;
  if (reserve_cus >= ncu) return XK_ERR_ARG;
  const int nwords = (ncu + 31) / 32;
  uint32_t mask[64];
  if (nwords > 64) return XK_ERR_UNSUPPORTED;
  for (int w = 0; w < nwords; ++w) mask[w] = 0;
  if (pattern == 2) {
    // the COMPLEMENT of pattern 0 (r06): only the `reserve_cus` units that a pattern-0 stream leaves free — for the chain
    // of a batch group whose panel kernel does not fill the register file of its units (K1sw r06 form: 2 x 202 of 512
    // registers), so that chain workgroups do not move in beside the panel kernel's
    for (int cu = ncu - reserve_cus; cu < ncu; ++cu) mask[cu >> 5] |= (1u << (cu & 31));
  } else if (pattern == 0 || reserve_cus == 0) {
    for (int cu = 0; cu < ncu - reserve_cus; ++cu) mask[cu >> 5] |= (1u << (cu & 31));
  } else {
    const int step = ncu / reserve_cus;                       // clear bit step-1, 2*step-1, ... (reserve_cus of them)
    int cleared = 0;
    for (int cu = 0; cu < ncu; ++cu) {
      const bool clear = step > 0 && (cu % step) == step - 1 && cleared < reserve_cus;
      if (clear) ++cleared;
      else mask[cu >> 5] |= (1u << (cu & 31));
    }
  }
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess) prev = device;
  if (prev != device && hipSetDevice(device) != hipSuccess) return XK_ERR_ARG;
  hipStream_t st = nullptr;
  e = hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask);
  if (prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess) return (int)e;
  *stream_out = (void*)st;
  return XK_OK;
}

extern "C" int xk_stream_create_cu_masked(int device, int reserve_cus, void** stream_out) {
  return xk_stream_create_cu_masked_pattern(device, reserve_cus, 0, stream_out);
}

// Measurement utility: which XCDs (accelerator complex dies) the workgroups of a launch on `stream` land on.  Every
// workgroup reads its XCC id (HW_REG_XCC_ID, bits 3..0) and the id of its compute unit inside the XCD (HW_REG_HW_ID:
// CU_ID bits 11..8, SH bit 12, SE bits 15..13), marks the unit in a per-XCD bit set and counts itself:
// hist[xcc] = workgroups, units[xcc * 4 + w] = bit set of (se, sh, cu) seen.  Used to see what a CU mask leaves.
namespace xk {
__global__ __launch_bounds__(64) void xcc_probe_kernel(unsigned* __restrict__ hist, unsigned* __restrict__ units, int spin) {
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;          // HW_REG_XCC_ID[3:0]
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));                   // HW_REG_HW_ID[15:8]
  if (threadIdx.x == 0) {
    atomicAdd(&hist[xcc], 1u);
    atomicOr(&units[xcc * 8 + (hw >> 5)], 1u << (hw & 31));
  }
  // stay resident long enough for the dispatcher to use every unit the mask allows
  unsigned long long t0 = __builtin_readcyclecounter();
  while ((long long)(__builtin_readcyclecounter() - t0) < (long long)spin) {}
}
}  // namespace xk

extern "C" int xk_probe_xcc(unsigned* hist16, unsigned* units128, int workgroups, int spin_cycles, void* stream) {
  if (!hist16 || !units128 || workgroups < 1 || spin_cycles < 0) return XK_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(hist16, 0, 16 * sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(units128, 0, 128 * sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(xk::xcc_probe_kernel, dim3((unsigned)workgroups), dim3(64), 0, st, hist16, units128, spin_cycles);
  XK_LAUNCH_CHECK();
  return XK_OK;
}

extern "C" int xk_stream_destroy(void* stream) {
  if (!stream) return XK_OK;
  return (int)hipStreamDestroy((hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// Measurement utility: read `bytes` of device memory once with 16 B/lane non-temporal loads and NO arithmetic
// beyond a checksum that keeps the loads alive.  The buffer is walked the way the panel kernels walk an operator
// batch: rows of `pitch_bytes`, tiles of 1024 rows x 8 KB, one 256-thread block per tile, every wave 2 KB of a row
// per step with a ring of 8 rows (16 KB) in flight.  bench.py times it on the operator batch to report what the
// SAME buffer streams at when nothing is computed — the practical ceiling the panel kernels are measured against
// (`roofline.stream_read`; scripts/micro/stream_patterns.hip has the other walks).
// ---------------------------------------------------------------------------------------------
namespace xk {
typedef unsigned int api_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_read_kernel(const char* __restrict__ src, long rows, long pitch,
                                                           int tiles_c, unsigned* __restrict__ scratch) {
  constexpr int DEPTH = 8, TR = 1024, TCB = 8192;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tr = blockIdx.x / tiles_c;
  const int tc = (int)(blockIdx.x - tr * tiles_c);
  const long row0 = tr * TR;
  long nrows = rows - row0;
  nrows = nrows < TR ? nrows : TR;
  const long colb = (long)tc * TCB + wave * 2048 + lane * 16;       // byte column of this lane's first vector
  const bool ok0 = colb + 16 <= pitch, ok1 = colb + 1024 + 16 <= pitch;
  const char* base = src + row0 * pitch + colb;
  api_u4 acc = {0u, 0u, 0u, 0u};
  api_u4 ring[DEPTH][2];
#pragma unroll
  for (int r = 0; r < DEPTH; ++r) {
    const long rr = r < nrows ? r : nrows - 1;
    ring[r][0] = ok0 ? __builtin_nontemporal_load((const api_u4*)(base + rr * pitch)) : acc;
    ring[r][1] = ok1 ? __builtin_nontemporal_load((const api_u4*)(base + rr * pitch + 1024)) : acc;
  }
  for (long i0 = 0; i0 < nrows; i0 += DEPTH) {
#pragma unroll
    for (int r = 0; r < DEPTH; ++r) {
      acc ^= ring[r][0] ^ ring[r][1];
      long rr = i0 + DEPTH + r;
      rr = rr < nrows ? rr : nrows - 1;
      ring[r][0] = ok0 ? __builtin_nontemporal_load((const api_u4*)(base + rr * pitch)) : acc;
      ring[r][1] = ok1 ? __builtin_nontemporal_load((const api_u4*)(base + rr * pitch + 1024)) : acc;
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u && scratch) scratch[blockIdx.x & 1023] = acc[0];
}
}  // namespace xk

extern "C" int xk_stream_read(const void* src, long bytes, long pitch_bytes, void* scratch, void* stream) {
  if (bytes < 0 || pitch_bytes <= 0 || (pitch_bytes & 15) || ((uintptr_t)src & 15)) return XK_ERR_ARG;
  const long rows = bytes / pitch_bytes;
  if (rows == 0) return XK_OK;
  const long tiles_r = (rows + 1023) / 1024, tiles_c = (pitch_bytes + 8191) / 8192;
  if (tiles_r * tiles_c > 0x7fffffffL || tiles_c > 0x7fffffffL) return XK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(xk::stream_read_kernel, dim3((unsigned)(tiles_r * tiles_c)), dim3(256), 0, (hipStream_t)stream,
                     (const char*)src, rows, pitch_bytes, (int)tiles_c, (unsigned*)scratch);
  XK_LAUNCH_CHECK();
  return XK_OK;
}
