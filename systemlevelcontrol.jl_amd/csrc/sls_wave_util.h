// sls_wave_util.h — device helpers shared by the one-wave / twisted kernels (sls_wave_kernel.hip) and the four-wave
// twisted kernel (sls_twisted4_kernel.hip): binary search in LDS, cross-lane primitives measured on gfx950
// (profiles/r01_lds_xlane_microbench.txt), compile-time loops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

#ifndef SLS_GJ_NR
#define SLS_GJ_NR 1          // Newton steps on the v_rcp_f64 seed of every pivot reciprocal (tools/nr_scan.sh: 1 gives the pass counts of 2)
#endif

namespace sls {

__device__ __forceinline__ int wbsearch(const int32_t* a, int n, int32_t key) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int32_t v = a[mid];
    if (v == key) return mid;
    if (v < key) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

// max over the wave, result in every lane: lane ^ 32 by v_permlane32_swap, lane ^ 16..1 by ds_swizzle in bitmask mode
// (immediate pattern; __shfl_xor costs an address computation and a ds_bpermute pair per step)
template <int XOR>
__device__ __forceinline__ double swizzle_xor_f64(double v) {
  constexpr int pattern = 0x1f | (XOR << 10);              // and_mask 0x1f, or_mask 0, xor_mask XOR
  return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), pattern),
                          __builtin_amdgcn_ds_swizzle(__double2loint(v), pattern));
}
__device__ __forceinline__ double wave_max_f64(double v) {
  const auto a = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  v = fmax(__hiloint2double(b[0], a[0]), __hiloint2double(b[1], a[1]));
  v = fmax(v, swizzle_xor_f64<16>(v));
  v = fmax(v, swizzle_xor_f64<8>(v));
  v = fmax(v, swizzle_xor_f64<4>(v));
  v = fmax(v, swizzle_xor_f64<2>(v));
  v = fmax(v, swizzle_xor_f64<1>(v));
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
  return v;
}

// 1/a to ~1 ulp: v_rcp_f64 seed + two Newton steps (the result only feeds a preconditioner)
__device__ __forceinline__ double fast_rcp(double a) {
  double x = __builtin_amdgcn_rcp(a);
  x = __builtin_fma(__builtin_fma(-a, x, 1.0), x, x);
  x = __builtin_fma(__builtin_fma(-a, x, 1.0), x, x);
  return x;
}

__device__ __forceinline__ double readlane_f64(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}
// v[lane] + v[lane ^ 32] with v_permlane32_swap (VALU, no LDS round trip):
// swap(x,x) = { [x.lo32 | x.lo32],  [x.hi32 | x.hi32] } as (lanes 0-31 | lanes 32-63)
__device__ __forceinline__ double xsum32(double v) {
  const auto a = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// v[lane] + v[lane ^ 16] with v_permlane16_swap
__device__ __forceinline__ double xsum16(double v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// compile-time loop: f(std::integral_constant<int, 0>{}), f(<1>), …  — gives every pivot a constexpr index, which the
// immediate-pattern cross-lane instructions (ds_swizzle) need
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// Broadcast inside each lane group of NPL lanes: every lane gets the value held by lane PV of ITS group.
// NPL = 32/16: ds_swizzle_b32 in bitmask mode, lane' = (lane & and_mask) | or_mask within 32-lane groups — measured
// 2.4 CU-cycles per instruction at saturation and ≈10 cycles for a lone wave, against 6 / 14 for ds_bpermute_b32 and
// ≈550 cycles for an LDS write→read round trip (tools/lds_xlane_microbench.hip).  NPL = 64: the group is the whole
// wave, so it is a plain v_readlane (scalar broadcast).
template <int NPL, int PV>
__device__ __forceinline__ double group_bcast(double v) {
  if constexpr (NPL == 64) {
    return readlane_f64(v, PV);
  } else {
    constexpr int and_mask = (NPL == 32) ? 0x00 : 0x10;
    constexpr int pattern = and_mask | (PV << 5);          // [4:0] and, [9:5] or, [14:10] xor, bit 15 = 0
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pattern);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pattern);
    return __hiloint2double(hi, lo);
  }
}

// Lanes of ONE wave exchange data through LDS.  DS instructions of a wave execute in program order, so a ds_read
// issued after a ds_write sees its data; all that is needed is that the compiler keeps that order (wave_barrier is a
// pure scheduling barrier, it emits no instruction).  __syncthreads() here would add s_waitcnt vmcnt(0) — a stall on
// every outstanding GLOBAL store/load (measured: 2.5 k cycles per time step of the residual pass).
#define WSYNC() __builtin_amdgcn_wave_barrier()



}  // namespace sls
