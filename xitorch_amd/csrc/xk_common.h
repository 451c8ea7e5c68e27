// xitorch_amd :: shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
//
// Everything here is written for wave64 / gfx950 only.  No CUDA shims, no
// dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define XK_WAVE 64

// status codes returned over the C ABI (see include/xitorch_amd.h)
#define XK_OK 0
#define XK_ERR_ARG (-1)
#define XK_ERR_UNSUPPORTED (-2)

#define XK_LAUNCH_CHECK()                       \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

namespace xk {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// 16-byte vector type for an element type
template <typename T> struct Vec16;
template <> struct Vec16<double> { typedef d2 type; static constexpr int n = 2; };
template <> struct Vec16<float> { typedef f4 type; static constexpr int n = 4; };

// streaming (read-once) 16 B load: the operator matrix is touched exactly once
// per panel product, so keep it out of the way of the L2-resident panel.
template <typename V>
__device__ __forceinline__ V ld_stream(const V* p) {
  return __builtin_nontemporal_load(p);
}

__device__ __forceinline__ double shfl_xor_t(double v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ float shfl_xor_t(float v, int mask) { return __shfl_xor(v, mask, 64); }

// ---------------------------------------------------------------------------
// Low-latency wave reduction on the DPP path (no LDS crossbar).
//
// `__shfl_xor` lowers to ds_bpermute_b32 (two per double), ~100+ cycles each through the LDS pipeline, so a
// butterfly all-reduce of a double is a ~12-deep chain of them: fine in streaming kernels that have other waves
// to run, but the whole critical path of latency-bound single-workgroup kernels (the small eigensolver makes two
// reductions per Householder step).  Here: row_shr 1/2/4/8 inside each 16-lane row, row_bcast:15 / row_bcast:31
// across the rows (gfx9 DPP controls), total read from lane 63 — six dependent VALU stages.
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_shift_or_zero(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_shift_or_zero(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
// Partner exchange for reduction stages over lane bits 3..0 on the DPP path (no LDS crossbar): the value of a lane
// whose bit MASK differs from this lane's.  For MASK = 8, 2, 1 the partner is exactly lane ^ MASK (row_ror:8,
// quad_perm); for MASK = 4 it is lane ^ 7 (row_half_mirror) — also a bijection between the bit-2 halves, which is
// all a sum reduction needs as long as the stages over bits 1 and 0 follow (they always do here).
template <int MASK> struct PartnerCtrl;
template <> struct PartnerCtrl<8> { static constexpr int ctrl = 0x128; };     // row_ror:8
template <> struct PartnerCtrl<4> { static constexpr int ctrl = 0x141; };     // row_half_mirror
template <> struct PartnerCtrl<2> { static constexpr int ctrl = 0x4e; };      // quad_perm [2,3,0,1]
template <> struct PartnerCtrl<1> { static constexpr int ctrl = 0xb1; };      // quad_perm [1,0,3,2]
template <int MASK>
__device__ __forceinline__ double lane_partner(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), PartnerCtrl<MASK>::ctrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), PartnerCtrl<MASK>::ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int MASK>
__device__ __forceinline__ float lane_partner(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), PartnerCtrl<MASK>::ctrl, 0xf, 0xf, false));
}
// run-time stage mask -> the matching compile-time exchange (the callers' loops are fully unrolled)
template <typename T>
__device__ __forceinline__ T lane_partner_rt(T v, int mask) {
  switch (mask) {
    case 8: return lane_partner<8>(v);
    case 4: return lane_partner<4>(v);
    case 2: return lane_partner<2>(v);
    case 1: return lane_partner<1>(v);
    default: return shfl_xor_t(v, mask);
  }
}
__device__ __forceinline__ double readlane63(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float readlane63(float v) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// sum over the 64 lanes, every lane gets the (bit-identical) total
template <typename T>
__device__ __forceinline__ T wave_sum_dpp(T v) {
  v += dpp_shift_or_zero<0x111, 0xf>(v);      // row_shr:1
  v += dpp_shift_or_zero<0x112, 0xf>(v);      // row_shr:2
  v += dpp_shift_or_zero<0x114, 0xf>(v);      // row_shr:4
  v += dpp_shift_or_zero<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of each row holds the row sum
  v += dpp_shift_or_zero<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
  v += dpp_shift_or_zero<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return readlane63(v);
}

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    T o = shfl_xor_t(v, m);
    v = o > v ? o : v;
  }
  return v;
}

// ---------------------------------------------------------------------------
// Transposing wave reduction ("reduce-scatter over lanes").
//
// Each lane holds NV partial sums v[0..NV).  We want, for every i, the sum of
// v[i] over the 64 lanes.  A plain butterfly costs 6*NV shuffles.  Instead, at
// every stage where the live count is even, lanes pair up values: the lane
// with the stage bit clear keeps the even-indexed value of each pair and ships
// the odd one, its partner does the opposite; the live count halves.  When the
// count becomes odd the remaining stages fall back to the plain butterfly.
//
// After the call, lane l holds in v[0..count) the full sums of original index
//     orig(i, l) = i * 2^h + sum_{s<h} bit_s(l) * 2^s,  bit_s(l) = (l >> (5-s)) & 1
// where h = number of halving stages performed (returned via template consts).
// Lanes that differ only in the low (6-h) bits hold identical copies.
// ---------------------------------------------------------------------------
template <int NV> struct HalvingStages {
  static constexpr int value = (NV % 2 == 0 && NV > 1) ? 1 + HalvingStages<NV / 2>::value : 0;
};
template <> struct HalvingStages<1> { static constexpr int value = 0; };
template <> struct HalvingStages<0> { static constexpr int value = 0; };

// gfx950 half-exchange primitives: v_permlane32_swap swaps lanes 32-63 of `a` with lanes 0-31 of `b`;
// v_permlane16_swap swaps the odd 16-lane rows of `a` with the even rows of `b`.  After the swap
// a' + b' is, in the low half (even rows), the pair-sum of the ORIGINAL a and, in the high half
// (odd rows), the pair-sum of the original b — exactly one stage of the transposing reduction with
// no select and no LDS-crossbar traffic.
__device__ __forceinline__ double swap_add32(double a, double b) {
  auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ float swap_add32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b) {
  auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ float swap_add16(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// butterfly all-reduce (sum) over the 64 lanes of a wave, every lane gets the bit-identical total; all stages on
// the vector ALU (half-exchange swaps for lane bits 5 and 4, DPP partners below)
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
  v = swap_add32(v, v);
  v = swap_add16(v, v);
  v += lane_partner<8>(v);
  v += lane_partner<4>(v);
  v += lane_partner<2>(v);
  v += lane_partner<1>(v);
  return v;
}

template <typename T, int NV>
__device__ __forceinline__ void wave_reduce_scatter(T (&v)[NV], int lane) {
  constexpr int H0 = HalvingStages<NV>::value;
  constexpr int H = H0 > 6 ? 6 : H0;
  int cnt = NV;
#pragma unroll
  for (int s = 0; s < H; ++s) {
    const int mask = 32 >> s;
    if (s < 2) {
      // stages with masks 32 and 16: half-exchange swaps (lane with the stage bit clear ends up with
      // the pair-sum of the even-indexed value, its partner with that of the odd-indexed one)
      const int half0 = cnt / 2;
#pragma unroll
      for (int i = 0; i < NV / 2; ++i) {
        if (i < half0) v[i] = (s == 0) ? swap_add32(v[2 * i], v[2 * i + 1]) : swap_add16(v[2 * i], v[2 * i + 1]);
      }
      cnt = half0;
      continue;
    }
    const bool hi = (lane & mask) != 0;
    const int half = cnt / 2;
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
      if (i < half) {
        T keep = hi ? v[2 * i + 1] : v[2 * i];
        T send = hi ? v[2 * i] : v[2 * i + 1];
        v[i] = keep + lane_partner_rt(send, mask);
      }
    }
    cnt = half;
  }
#pragma unroll
  for (int s = H; s < 6; ++s) {
    const int mask = 32 >> s;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i < (NV >> H)) v[i] += lane_partner_rt(v[i], mask);
    }
  }
}

// index bookkeeping for wave_reduce_scatter
template <int NV>
__device__ __forceinline__ int wave_rs_orig_index(int i, int lane) {
  constexpr int H0 = HalvingStages<NV>::value;
  constexpr int H = H0 > 6 ? 6 : H0;
  int idx = i << H;
#pragma unroll
  for (int s = 0; s < H; ++s) idx += ((lane >> (5 - s)) & 1) << s;
  return idx;
}
template <int NV>
__device__ __forceinline__ bool wave_rs_is_writer(int lane) {
  constexpr int H0 = HalvingStages<NV>::value;
  constexpr int H = H0 > 6 ? 6 : H0;
  return (lane & ((64 >> H) - 1)) == 0;
}
template <int NV> struct WaveRsCount {
  static constexpr int H0 = HalvingStages<NV>::value;
  static constexpr int H = H0 > 6 ? 6 : H0;
  static constexpr int value = NV >> H;
};

// non-negative doubles/floats order like their bit patterns -> atomic max on ints
__device__ __forceinline__ void atomic_max_nonneg(double* addr, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(addr),
            (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
  atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

}  // namespace xk
