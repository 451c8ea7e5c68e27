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
extern "C" int xk_stream_create_cu_masked(int device, int reserve_cus, void** stream_out) {
  if (!stream_out || reserve_cus < 0) return XK_ERR_ARG;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return (int)e;
  const int ncu = prop.multiProcessorCount;
  if (reserve_cus >= ncu) return XK_ERR_ARG;
  const int nwords = (ncu + 31) / 32;
  uint32_t mask[64];
  if (nwords > 64) return XK_ERR_UNSUPPORTED;
  for (int w = 0; w < nwords; ++w) mask[w] = 0;
  for (int cu = 0; cu < ncu - reserve_cus; ++cu) mask[cu >> 5] |= (1u << (cu & 31));
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

extern "C" int xk_stream_destroy(void* stream) {
  if (!stream) return XK_OK;
  return (int)hipStreamDestroy((hipStream_t)stream);
}
