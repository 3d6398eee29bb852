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

// ---- the whole symbolic pass of the README recipe per column, on the device (round 3; SURVEY §8 row f1 on the SOLVE path) ----
// One wave per column c.  Levels L_0 = {c} … L_KL by the same bitmap expansion as mask_levels_kernel, every level kept
// (sorted) in an LDS pool; actuator sets U_k likewise.  Then
//     s_x = L_{kx[T−1]+1}(c),  s_u = U_{ku[T−1]+1}(c)          (reference src/reduction.jl:14: rows of (𝓢[T]·(A≠0))[:,c])
//     bit (t, i) of the column's compact mask  ⇔  s_x[i] ∈ L_{kx[t]}(c)   /   s_u[j] ∈ U_{ku[t]}(c)
//     first destinations: value-array offset of time step t + the column's CSC position = exclusive prefix of the level sizes
// — exactly what the host pass (sls_symbolic.cpp: fill_range, compact form) derives from the caller's mask arrays.
// Count pass: sizes (ñx, ñu, pos, nnz, free variables, level sizes); fill pass after the prefix sums: index sets, bit masks, bases.
__device__ __forceinline__ int lbsearch(const int32_t* a, int n, int32_t key) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int32_t v = a[mid];
    if (v == key) return mid;
    if (v < key) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

template <bool FILL>
__global__ __launch_bounds__(64) void column_tables_kernel(const ColumnTableParams p) {
  extern __shared__ uint32_t lds_u32[];
  const int lane = threadIdx.x;
  const int nwx = (p.Nx + 31) >> 5, nwu = (max(p.Nu, 1) + 31) >> 5;
  uint32_t* bmx = lds_u32;
  uint32_t* bmu = bmx + nwx;
  int32_t* px = reinterpret_cast<int32_t*>(bmu + nwu);        // level starts in lx (KL + 2 entries)
  int32_t* pu = px + 68;
  int32_t* chk = pu + 68;                                      // [2][T] regularity counters (fill pass)
  int32_t* kxs = chk + 2 * p.T;                                // level schedule, staged once (a global load per time step and
  int32_t* kus = kxs + p.T;                                    //  column was most of the fill pass)
  int32_t* lx = kus + p.T;
  int32_t* lu = lx + p.cap;
  for (int i = lane; i < nwx + nwu; i += 64) lds_u32[i] = 0u;
  for (int t = lane; t < p.T; t += 64) { kxs[t] = p.kx[t]; kus[t] = p.ku[t]; }
  __syncthreads();
  const int K1 = p.kmax + 1;
  for (int c = blockIdx.x; c < p.Nx; c += gridDim.x) {
    if (lane == 0) { px[0] = 0; px[1] = 1; lx[0] = c; pu[0] = 0; }
    __syncthreads();
    for (int k = 0; k <= p.KL; ++k) {
      const int32_t* cur = lx + px[k];
      const int ncur = px[k + 1] - px[k];
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
      const int u0 = pu[k];
      const int nact = (umax >= 0) ? bitmap_to_list(bmu, umin, umax, lu + u0, p.cap - u0, lane, p.flags) : 0;
      __syncthreads();
      if (lane == 0) pu[k + 1] = u0 + nact;
      if (k < p.KL) {
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
        const int x0 = px[k + 1];
        const int nn = (xmax >= 0) ? bitmap_to_list(bmx, xmin, xmax, lx + x0, p.cap - x0, lane, p.flags) : 0;
        __syncthreads();
        if (lane == 0) px[k + 2] = x0 + nn;
      }
      __syncthreads();
    }
    const int32_t* sx = lx + px[p.qx1];
    const int n = px[p.qx1 + 1] - px[p.qx1];
    const int32_t* su = lu + pu[p.qu1];
    const int m = pu[p.qu1 + 1] - pu[p.qu1];
    const int nm = n + m, wps = (nm + 63) >> 6;
    if (!FILL) {
      int nnzA = 0, nnzB = 0;
      for (int i = lane; i < n; i += 64) {
        const int r = sx[i];
        for (int e = p.A_rowptr[r]; e < p.A_rowptr[r + 1]; ++e)
          if (p.A_val[e] != 0.0 && lbsearch(sx, n, p.A_colidx[e]) >= 0) ++nnzA;
        for (int e = p.B_rowptr[r]; e < p.B_rowptr[r + 1]; ++e)
          if (p.B_val[e] != 0.0 && lbsearch(su, m, p.B_colidx[e]) >= 0) ++nnzB;
      }
      int nfree = 0;
      for (int t = lane; t < p.T; t += 64) {
        const int a = kxs[t], b = kus[t];
        nfree += (px[a + 1] - px[a]) + (pu[b + 1] - pu[b]);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { nnzA += __shfl_xor(nnzA, off); nnzB += __shfl_xor(nnzB, off); nfree += __shfl_xor(nfree, off); }
      if (lane == 0) {
        int32_t* ci = p.col_info + (int64_t)c * 6;
        ci[0] = n; ci[1] = m; ci[2] = lbsearch(sx, n, c); ci[3] = nnzA; ci[4] = nnzB; ci[5] = nfree;
      }
      for (int k = lane; k < K1; k += 64) { p.cntx[(int64_t)c * K1 + k] = px[k + 1] - px[k]; p.cntu[(int64_t)c * K1 + k] = pu[k + 1] - pu[k]; }
    } else {
      int32_t* ip = p.idx_pool + p.idx_off[c];
      for (int i = lane; i < n; i += 64) ip[i] = sx[i];
      for (int i = lane; i < m; i += 64) ip[n + i] = su[i];
      for (int t = lane; t < 2 * p.T; t += 64) chk[t] = 0;
      __syncthreads();
      uint64_t* cm = p.cmask + p.cw_off[c];
      for (int w = 0; w < wps; ++w) {
        const int pos = 64 * w + lane;
        // membership of this position's state / actuator in every level the masks use
        unsigned long long lv = 0;
        if (pos < n) {
          const int id = sx[pos];
          for (int k = 0; k < K1; ++k) if (lbsearch(lx + px[k], px[k + 1] - px[k], id) >= 0) lv |= 1ull << k;
        } else if (pos < nm) {
          const int id = su[pos - n];
          for (int k = 0; k < K1; ++k) if (lbsearch(lu + pu[k], pu[k + 1] - pu[k], id) >= 0) lv |= 1ull << k;
        }
        const int nxw = min(max(n - 64 * w, 0), 64);                  // x positions in this word
        const unsigned long long xbits = nxw >= 64 ? ~0ull : ((1ull << nxw) - 1ull);
        for (int t = 0; t < p.T; ++t) {
          const int lev = (pos < n) ? kxs[t] : kus[t];
          const bool on = pos < nm && ((lv >> lev) & 1ull);
          const unsigned long long word = __ballot(on);
          if (lane == 0) {
            cm[(int64_t)t * wps + w] = word;
            chk[t] += __popcll(word & xbits); chk[p.T + t] += __popcll(word & ~xbits);
          }
        }
      }
      __syncthreads();
      bool bad = false;
      for (int t = lane; t < p.T; t += 64) {
        const int a = kxs[t], b = kus[t];
        // every row of the mask column must lie inside the index set (else rank-in-part ≠ CSC position: irregular)
        bad = bad || chk[t] != px[a + 1] - px[a] || chk[p.T + t] != pu[b + 1] - pu[b];
        int32_t* cb = p.cbase + 2ll * p.T * c;
        cb[2 * t] = (int32_t)(p.offx[t] + p.prex[(int64_t)a * p.Nx + c]);
        cb[2 * t + 1] = (int32_t)(p.offu[t] + p.preu[(int64_t)b * p.Nx + c]);
      }
      if (__any(bad) && lane == 0) p.flags[1] = 1;
    }
    __syncthreads();
  }
}

// exclusive prefix over the columns of cnt[c][k] for every level k: pre[k][c], tot[k].  One 1024-thread workgroup per (level,
// x|u): 1024 columns per round (wave scans + a scan of the 16 wave totals); a one-wave version spent 1 µs per 64 columns.
__global__ __launch_bounds__(1024) void level_prefix_kernel(const int32_t* __restrict__ cntx, const int32_t* __restrict__ cntu, int Nx, int K1,
                                                            int64_t* __restrict__ prex, int64_t* __restrict__ preu,
                                                            int64_t* __restrict__ totx, int64_t* __restrict__ totu) {
  __shared__ long long wsum[16];
  __shared__ long long carry_s;
  const bool isu = blockIdx.x >= (unsigned)K1;
  const int k = isu ? blockIdx.x - K1 : blockIdx.x;
  const int32_t* cnt = isu ? cntu : cntx;
  int64_t* pre = isu ? preu : prex;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int b = 0; b < Nx; b += 1024) {
    const int i = b + tid;
    const long long v = (i < Nx) ? cnt[(int64_t)i * K1 + k] : 0;
    long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const long long t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    long long woff = 0;
    for (int q = 0; q < w; ++q) woff += wsum[q];
    const long long carry = carry_s;
    if (i < Nx) pre[(int64_t)k * Nx + i] = carry + woff + incl - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (tid == 0) (isu ? totu : totx)[k] = carry_s;
}

hipError_t launch_column_tables(const ColumnTableParams& p, bool fill, int grid, size_t lds_bytes, hipStream_t stream) {
  const void* fn = fill ? reinterpret_cast<const void*>(&column_tables_kernel<true>) : reinterpret_cast<const void*>(&column_tables_kernel<false>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  if (fill) hipLaunchKernelGGL(column_tables_kernel<true>, dim3(grid), dim3(64), lds_bytes, stream, p);
  else hipLaunchKernelGGL(column_tables_kernel<false>, dim3(grid), dim3(64), lds_bytes, stream, p);
  return hipGetLastError();
}
hipError_t launch_level_prefix(const int32_t* cntx, const int32_t* cntu, int Nx, int K1, int64_t* prex, int64_t* preu, int64_t* totx,
                               int64_t* totu, hipStream_t stream) {
  hipLaunchKernelGGL(level_prefix_kernel, dim3(2 * K1), dim3(1024), 0, stream, cntx, cntu, Nx, K1, prex, preu, totx, totu);
  return hipGetLastError();
}

}  // namespace sls
