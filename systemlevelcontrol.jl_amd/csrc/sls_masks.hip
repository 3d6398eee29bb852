// sls_masks.hip — the README mask recipe on the device (reference README.md:52-54):
//     𝓢x[t] = (A≠0)^kx(t) ≠ 0,   𝓢u[t] = (B2ᵀ≠0)(A≠0)^ku(t) ≠ 0,   kx = min(d, ⌊αt⌋), ku = min(d+1, ⌊αt⌋),  t = 0..T−1.
// Column c of (A≠0)^k is the set reached from c by walks of exactly k steps along A's pattern (edge q → r iff A[r,q] ≠ 0): one
// wave per column expands the level sets L_0 = {c}, L_{k+1} = ∪_{q∈L_k} rows(A[:,q]) with an LDS bitmap over the states
// (atomicOr dedupes, a scan of the touched word range returns the level sorted), and the actuator sets
// act_k = ∪_{r∈L_k} {j : B2[r,j] ≠ 0} the same way over the inputs.  Two launches, like the host pass (sls_symbolic.cpp:
// localization_masks): count (sizes of every level), then — after the host's prefix sums — fill (every time step's column
// written straight at its CSC position, Julia's Int64 indices in the caller's index base).  HBM-bound integer work: the only
// traffic that matters is the row-index arrays themselves, written once, coalesced per level.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sls_device.h"

namespace sls {

// the set bits of bm[w0..w1] as an ascending list; clears the words.  Returns the count (wave-uniform).
__device__ __forceinline__ int bitmap_to_list(uint32_t* bm, int w0, int w1, int32_t* out, int cap, int lane, int32_t* overflow) {
  int total = 0;
  for (int wb = w0; wb <= w1; wb += 64) {
    const int wi = wb + lane;
    const uint32_t word = (wi <= w1) ? bm[wi] : 0u;
    if (wi <= w1) bm[wi] = 0u;
    const int cnt = __popc(word);
    int incl = cnt;                                   // inclusive prefix over the lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    int pos = total + incl - cnt;
    uint32_t wbits = word;
    while (wbits) {
      const int b = __ffs(wbits) - 1;
      wbits &= wbits - 1;
      if (pos < cap) out[pos] = wi * 32 + b;
      ++pos;
    }
    total += __shfl(incl, 63);
  }
  if (total > cap) { if (lane == 0) *overflow = 1; total = cap; }
  return total;
}

template <bool FILL>
__global__ __launch_bounds__(64) void mask_levels_kernel(const MaskParams p) {
  extern __shared__ uint32_t lds_u32[];
  const int lane = threadIdx.x;
  const int nwx = (p.Nx + 31) >> 5, nwu = (max(p.Nu, 1) + 31) >> 5;
  uint32_t* bmx = lds_u32;                           // bitmap over states
  uint32_t* bmu = bmx + nwx;                         // bitmap over inputs
  int32_t* cur = reinterpret_cast<int32_t*>(bmu + nwu);
  int32_t* nxt = cur + p.cap;
  int32_t* act = nxt + p.cap;
  for (int i = lane; i < nwx + nwu; i += 64) lds_u32[i] = 0u;
  __syncthreads();
  const int K1 = p.kmax + 1;
  for (int c = blockIdx.x; c < p.Nx; c += gridDim.x) {
    int ncur = 1;
    if (lane == 0) cur[0] = c;
    __syncthreads();
    for (int k = 0; k <= p.kmax; ++k) {
      // ---- actuator set of this level ----
      int umin = 0x7fffffff, umax = -1;
      for (int i = lane; i < ncur; i += 64) {
        const int r = cur[i];
        for (int e = p.B_rp[r]; e < p.B_rp[r + 1]; ++e) {
          const int j = p.B_ci[e];
          atomicOr(&bmu[j >> 5], 1u << (j & 31));
          umin = min(umin, j >> 5); umax = max(umax, j >> 5);
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { umin = min(umin, __shfl_xor(umin, off)); umax = max(umax, __shfl_xor(umax, off)); }
      __syncthreads();
      const int nact = (umax >= 0) ? bitmap_to_list(bmu, umin, umax, act, p.cap, lane, p.overflow) : 0;
      __syncthreads();
      if (!FILL) {
        if (lane == 0) { p.cntx[(int64_t)c * K1 + k] = ncur; p.cntu[(int64_t)c * K1 + k] = nact; }
      } else {
        // every time step that uses this level: the column's rows at their CSC position
        for (int t = 0; t < p.T; ++t) {
          if (p.kx[t] == k) {
            int64_t* dst = p.rowx + p.offx[t] + p.prex[(int64_t)k * p.Nx + c];
            for (int i = lane; i < ncur; i += 64) dst[i] = (int64_t)cur[i] + p.base;
          }
          if (p.ku[t] == k) {
            int64_t* dst = p.rowu + p.offu[t] + p.preu[(int64_t)k * p.Nx + c];
            for (int i = lane; i < nact; i += 64) dst[i] = (int64_t)act[i] + p.base;
          }
        }
      }
      if (k == p.kmax) break;
      // ---- next level ----
      int xmin = 0x7fffffff, xmax = -1;
      for (int i = lane; i < ncur; i += 64) {
        const int q = cur[i];
        for (int e = p.A_cp[q]; e < p.A_cp[q + 1]; ++e) {
          const int r = p.A_ri[e];
          atomicOr(&bmx[r >> 5], 1u << (r & 31));
          xmin = min(xmin, r >> 5); xmax = max(xmax, r >> 5);
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { xmin = min(xmin, __shfl_xor(xmin, off)); xmax = max(xmax, __shfl_xor(xmax, off)); }
      __syncthreads();
      const int nn = (xmax >= 0) ? bitmap_to_list(bmx, xmin, xmax, nxt, p.cap, lane, p.overflow) : 0;
      __syncthreads();
      int32_t* sw = cur; cur = nxt; nxt = sw;
      ncur = nn;
    }
    __syncthreads();
  }
}

hipError_t launch_mask_levels(const MaskParams& p, bool fill, int grid, size_t lds_bytes, hipStream_t stream) {
  const void* fn = fill ? reinterpret_cast<const void*>(&mask_levels_kernel<true>) : reinterpret_cast<const void*>(&mask_levels_kernel<false>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  if (fill) hipLaunchKernelGGL(mask_levels_kernel<true>, dim3(grid), dim3(64), lds_bytes, stream, p);
  else hipLaunchKernelGGL(mask_levels_kernel<false>, dim3(grid), dim3(64), lds_bytes, stream, p);
  return hipGetLastError();
}

// ---- index sets of all single-column subproblems (reference src/reduction.jl:11-27) ----------------------
// One wave per column c: the union over k ∈ rows(A[:,c]) of column k of the last mask, deduplicated in the LDS bitmap and
// returned ascending — the order the kernels' destination tables use (the reference's first-appearance order is the same set).
template <bool FILL>
__global__ __launch_bounds__(64) void index_sets_kernel(const IndexSetParams p) {
  extern __shared__ uint32_t lds_u32[];
  const int lane = threadIdx.x;
  const int nwx = (p.Nx + 31) >> 5, nwu = (max(p.Nu, 1) + 31) >> 5;
  uint32_t* bmx = lds_u32;
  uint32_t* bmu = bmx + nwx;
  int32_t* lst = reinterpret_cast<int32_t*>(bmu + nwu);
  for (int i = lane; i < nwx + nwu; i += 64) lds_u32[i] = 0u;
  __syncthreads();
  for (int c = blockIdx.x; c < p.Nx; c += gridDim.x) {
    int xmin = 0x7fffffff, xmax = -1, umin = 0x7fffffff, umax = -1;
    for (int e = p.A_cp[c]; e < p.A_cp[c + 1]; ++e) {
      const int k = p.A_ri[e];
      for (int i = p.Sx_cp[k] + lane; i < p.Sx_cp[k + 1]; i += 64) {
        const int r = p.Sx_ri[i];
        atomicOr(&bmx[r >> 5], 1u << (r & 31));
        xmin = min(xmin, r >> 5); xmax = max(xmax, r >> 5);
      }
      for (int i = p.Su_cp[k] + lane; i < p.Su_cp[k + 1]; i += 64) {
        const int j = p.Su_ri[i];
        atomicOr(&bmu[j >> 5], 1u << (j & 31));
        umin = min(umin, j >> 5); umax = max(umax, j >> 5);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      xmin = min(xmin, __shfl_xor(xmin, off)); xmax = max(xmax, __shfl_xor(xmax, off));
      umin = min(umin, __shfl_xor(umin, off)); umax = max(umax, __shfl_xor(umax, off));
    }
    __syncthreads();
    const int nx = (xmax >= 0) ? bitmap_to_list(bmx, xmin, xmax, lst, p.cap, lane, p.overflow) : 0;
    __syncthreads();
    if (FILL) { int64_t* dst = p.outx + p.ptrx[c]; for (int i = lane; i < nx; i += 64) dst[i] = (int64_t)lst[i] + p.base; }
    else if (lane == 0) p.cntx[c] = nx;
    __syncthreads();
    const int nu = (umax >= 0) ? bitmap_to_list(bmu, umin, umax, lst, p.cap, lane, p.overflow) : 0;
    __syncthreads();
    if (FILL) { int64_t* dst = p.outu + p.ptru[c]; for (int i = lane; i < nu; i += 64) dst[i] = (int64_t)lst[i] + p.base; }
    else if (lane == 0) p.cntu[c] = nu;
    __syncthreads();
  }
}

hipError_t launch_index_sets(const IndexSetParams& p, bool fill, int grid, size_t lds_bytes, hipStream_t stream) {
  const void* fn = fill ? reinterpret_cast<const void*>(&index_sets_kernel<true>) : reinterpret_cast<const void*>(&index_sets_kernel<false>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  if (fill) hipLaunchKernelGGL(index_sets_kernel<true>, dim3(grid), dim3(64), lds_bytes, stream, p);
  else hipLaunchKernelGGL(index_sets_kernel<false>, dim3(grid), dim3(64), lds_bytes, stream, p);
  return hipGetLastError();
}

}  // namespace sls
