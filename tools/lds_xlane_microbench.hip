// lds_xlane_microbench.hip — per-CU throughput of the cross-lane / LDS primitives the wave kernel can use to broadcast a
// pivot column: ds_bpermute_b32, ds_read2_b64 (broadcast), v_readlane_b32, v_permlane32_swap.  Cycles per wave-instruction
// at 1..16 waves per CU (s_memtime inside one wave + wall clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(int* out, int iters, unsigned long long* cyc) {
  __shared__ double buf[128];
  int lane = threadIdx.x;
  buf[lane] = lane; buf[lane + 64] = lane * 2.0;
  __syncthreads();
  int v0 = lane, v1 = lane * 3, v2 = lane * 5, v3 = lane * 7, v4 = lane + 1, v5 = lane + 2, v6 = lane + 3, v7 = lane + 4;
  double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  int addr = ((lane & 32) + 5) << 2;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {   // 8 independent ds_bpermute_b32
      v0 = __builtin_amdgcn_ds_bpermute(addr, v0); v1 = __builtin_amdgcn_ds_bpermute(addr, v1);
      v2 = __builtin_amdgcn_ds_bpermute(addr, v2); v3 = __builtin_amdgcn_ds_bpermute(addr, v3);
      v4 = __builtin_amdgcn_ds_bpermute(addr, v4); v5 = __builtin_amdgcn_ds_bpermute(addr, v5);
      v6 = __builtin_amdgcn_ds_bpermute(addr, v6); v7 = __builtin_amdgcn_ds_bpermute(addr, v7);
    } else if (MODE == 1) {   // 4 ds_read2_b64 broadcast (8 doubles)
      const double* p = buf + (lane >> 5);
      double a0 = p[0], a1 = p[2], a2 = p[4], a3 = p[6], a4 = p[8], a5 = p[10], a6 = p[12], a7 = p[14];
      d0 += a0 + a1; d1 += a2 + a3; d2 += a4 + a5; d3 += a6 + a7;
      asm volatile("" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3));
    } else if (MODE == 2) {   // 8 v_readlane_b32
      v0 += __builtin_amdgcn_readlane(v1, 5); v1 += __builtin_amdgcn_readlane(v2, 6); v2 += __builtin_amdgcn_readlane(v3, 7);
      v3 += __builtin_amdgcn_readlane(v4, 8); v4 += __builtin_amdgcn_readlane(v5, 9); v5 += __builtin_amdgcn_readlane(v6, 10);
      v6 += __builtin_amdgcn_readlane(v7, 11); v7 += __builtin_amdgcn_readlane(v0, 12);
    } else if (MODE == 3) {   // 8 v_permlane32_swap
      auto a = __builtin_amdgcn_permlane32_swap(v0, v1, false, false); v0 = a[0]; v1 = a[1];
      auto b = __builtin_amdgcn_permlane32_swap(v2, v3, false, false); v2 = b[0]; v3 = b[1];
      auto c = __builtin_amdgcn_permlane32_swap(v4, v5, false, false); v4 = c[0]; v5 = c[1];
      auto d = __builtin_amdgcn_permlane32_swap(v6, v7, false, false); v6 = d[0]; v7 = d[1];
      auto e = __builtin_amdgcn_permlane32_swap(v0, v2, false, false); v0 = e[0]; v2 = e[1];
      auto f = __builtin_amdgcn_permlane32_swap(v1, v3, false, false); v1 = f[0]; v3 = f[1];
      auto g = __builtin_amdgcn_permlane32_swap(v4, v6, false, false); v4 = g[0]; v6 = g[1];
      auto hh = __builtin_amdgcn_permlane32_swap(v5, v7, false, false); v5 = hh[0]; v7 = hh[1];
    } else if (MODE == 4) {   // 8 ds_swizzle broadcast within 32 (bitmask mode: and=0x00 or=5 -> lane 5 of each 32-group)
      v0 = __builtin_amdgcn_ds_swizzle(v0, 0x00A0); v1 = __builtin_amdgcn_ds_swizzle(v1, 0x00A0);
      v2 = __builtin_amdgcn_ds_swizzle(v2, 0x00A0); v3 = __builtin_amdgcn_ds_swizzle(v3, 0x00A0);
      v4 = __builtin_amdgcn_ds_swizzle(v4, 0x00A0); v5 = __builtin_amdgcn_ds_swizzle(v5, 0x00A0);
      v6 = __builtin_amdgcn_ds_swizzle(v6, 0x00A0); v7 = __builtin_amdgcn_ds_swizzle(v7, 0x00A0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + lane] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (int)(d0 + d1 + d2 + d3);
  if (lane == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
  int* out; unsigned long long* cyc; CHK(hipMalloc(&out, 4 * 64 * 256 * 32)); CHK(hipMalloc(&cyc, 8));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const char* names[5] = {"ds_bpermute_b32", "ds_read2_b64 bcast", "v_readlane_b32", "v_permlane32_swap", "ds_swizzle bcast"};
  const int iters = 20000;
  for (int mode = 0; mode < 5; ++mode)
    for (int wpc : {1, 2, 4, 8, 16}) {
      int blocks = prop.multiProcessorCount * wpc; float ms; unsigned long long hc;
      CHK(hipEventRecord(e0));
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); break;
      }
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
      CHK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
      const double n_inst = (mode == 1 ? 4.0 : 8.0) * iters;
      printf("%-20s waves/CU %2d : %7.1f cycles per wave-instr (one wave's view), %6.2f CU-cycles per instr (wall)  [%0.3f ms]\n",
             names[mode], wpc, (double)hc / n_inst, ms * 1e-3 * 2.4e9 / (n_inst * wpc), ms);
    }
  return 0;
}
