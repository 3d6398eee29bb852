// fp64_microbench.hip — calibrates the FP64 ceilings used by DESIGN.md / bench.py's roofline on gfx950:
//   (1) v_mfma_f64_16x16x4_f64  dense rate (TFLOP/s) and cycles per instruction per SIMD
//   (2) v_fma_f64 (VALU)        dense rate
//   (3) v_readlane + v_fma_f64  the scalar-broadcast rank-1 update pattern
//   (4) ds_bpermute/LDS-broadcast patterns
// Build: hipcc -O3 --offload-arch=gfx950 tools/fp64_microbench.hip -o tools/fp64_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, unsigned long long* cyc) {
  d4 acc[NACC];
  for (int a = 0; a < NACC; ++a) acc[a] = d4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, unsigned long long* cyc) {
  double acc[NACC];
  for (int a = 0; a < NACC; ++a) acc[a] = a;
  double x = 1.0 + threadIdx.x * 1e-9, y = 1e-9 * threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = __builtin_fma(acc[q], x, y);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int q = 0; q < NACC; ++q) s += acc[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// rank-1 update pattern: M[i] (16 registers) -= bcast(M[i], p) * r  with p a compile-time lane
template <int NR>
__global__ __launch_bounds__(64) void k_readlane_fma(double* out, int iters, unsigned long long* cyc) {
  double M[NR];
  for (int i = 0; i < NR; ++i) M[i] = 1.0 + 1e-3 * (i + threadIdx.x);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      double r = M[p] * 1e-3;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        double c = readlane_f64(M[i], p);
        M[i] = __builtin_fma(-c, r, M[i]);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NR; ++i) s += M[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// same with LDS broadcast: column values written by one lane, read back by all (b128 broadcast reads)
template <int NR>
__global__ __launch_bounds__(64) void k_lds_bcast_fma(double* out, int iters, unsigned long long* cyc) {
  __shared__ double col[NR];
  double M[NR];
  for (int i = 0; i < NR; ++i) M[i] = 1.0 + 1e-3 * (i + threadIdx.x);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      double r = M[p] * 1e-3;
      if (threadIdx.x == p) {
#pragma unroll
        for (int i = 0; i < NR; ++i) col[i] = M[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NR; ++i) M[i] = __builtin_fma(-col[i], r, M[i]);
      __syncthreads();
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NR; ++i) s += M[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// dependent MFMA chain (latency)
__global__ __launch_bounds__(64) void k_mfma_dep(double* out, int iters, unsigned long long* cyc) {
  d4 acc = d4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  double* out; unsigned long long* cyc;
  CHK(hipMalloc(&out, sizeof(double) * 256 * 4096));
  CHK(hipMalloc(&cyc, 8));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int ncu = prop.multiProcessorCount;
  auto report = [&](const char* name, double flops_per_wave_iter, int waves_per_block, int blocks, int iters, float ms, unsigned long long c, double instr_per_iter) {
    double total = flops_per_wave_iter * waves_per_block * (double)blocks * iters;
    printf("%-34s %8.3f ms  %9.3f TFLOP/s   wave cycles/iter %8.1f  (%.1f cyc per instr-group)\n", name, ms, total / (ms * 1e-3) / 1e12,
           (double)c / iters, (double)c / iters / instr_per_iter);
  };
  unsigned long long hc; float ms;
  for (int rep = 0; rep < 2; ++rep) {
    // (1) MFMA f64, 4 waves per block (1 per SIMD), 1 and 2 blocks per CU
    for (int bpc = 1; bpc <= 2; ++bpc) {
      int iters = 20000, blocks = ncu * bpc;
      CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc); CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1)); CHK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
      char nm[64]; snprintf(nm, 64, "mfma_f64_16x16x4 x4acc %dblk/CU", bpc);
      report(nm, 4 * 2.0 * 16 * 16 * 4, 4, blocks, iters, ms, hc, 4);
    }
    {
      int iters = 20000, blocks = ncu;
      CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc); CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1)); CHK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
      report("mfma_f64_16x16x4 x1acc (dep chain)", 2.0 * 16 * 16 * 4, 4, blocks, iters, ms, hc, 1);
    }
    // (2) VALU fma f64
    for (int bpc = 1; bpc <= 2; ++bpc) {
      int iters = 20000, blocks = ncu * bpc;
      CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc); CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1)); CHK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
      char nm[64]; snprintf(nm, 64, "v_fma_f64 x8acc %dblk/CU", bpc);
      report(nm, 8 * 2.0 * 64, 4, blocks, iters, ms, hc, 8);
    }
    {
      int iters = 20000, blocks = ncu;
      CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fma<1>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc); CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1)); CHK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
      report("v_fma_f64 x1acc (dep chain)", 2.0 * 64, 4, blocks, iters, ms, hc, 1);
    }
    // (3) readlane + fma, single wave per block, 1 block per CU (latency view) and 8 per CU
    for (int bpc : {1, 16}) {
      int iters = 2000, blocks = ncu * bpc;
      CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_readlane_fma<16>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1)); CHK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
      char nm[64]; snprintf(nm, 64, "readlane+fma 16rows x4piv %dw/CU", bpc);
      report(nm, 4 * 16 * 2.0 * 64, 1, blocks, iters, ms, hc, 64);
    }
    for (int bpc : {1, 16}) {
      int iters = 2000, blocks = ncu * bpc;
      CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds_bcast_fma<16>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1)); CHK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
      char nm[64]; snprintf(nm, 64, "lds-bcast+fma 16rows x4piv %dw/CU", bpc);
      report(nm, 4 * 16 * 2.0 * 64, 1, blocks, iters, ms, hc, 64);
    }
    {
      int iters = 20000, blocks = ncu;
      CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mfma_dep, dim3(blocks), dim3(64), 0, 0, out, iters, cyc); CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1)); CHK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
      report("mfma_f64 dep chain 1 wave/CU", 2.0 * 16 * 16 * 4, 1, blocks, iters, ms, hc, 1);
    }
    printf("----\n");
  }
  return 0;
}
