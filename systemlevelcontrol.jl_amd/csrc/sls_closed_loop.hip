// sls_closed_loop.hip — the consumer of an on-device Φ: the closed-loop simulation of reference README.md:62-72
//
//     β[:,t+1] = Σ_{τ=1..min(t,T−1)} Φx[τ+1]·(x[:,t+1−τ] − β[:,t+1−τ])
//     u[:,t]   = Σ_{τ=1..min(t,T)}   Φu[τ]  ·(x[:,t+1−τ] − β[:,t+1−τ])
//     x[:,t+1] = A·x[:,t] + B₁·w(t) + B₂·u[:,t]                                   t = 1..steps−1
//
// for `nscen` disturbance scenarios at once.  The reference runs this in the user script with one sparse mat-vec per
// (t, τ); here the two FIR sums are ONE row-oriented sparse operator over all lags (built once from the masks,
// FirOperator in sls_symbolic.h) applied to the history ŵ = x − β, which is kept for all times with T zero slots in front,
// so that an entry (τ, c) reads ŵ at (k+T)·Nx − (τ·Nx − c) without a lag test.  Φ arrives in the solve's own mask-order
// value array (device) and is gathered once per run into row order.
//
// One time step is one kernel: the wave that owns state row i evaluates β_i, the u rows its B₂ row references (an
// actuator that drives several states is evaluated by each of them, bit-identically) and the plant update, so no
// grid-wide dependency exists inside a step; the steps−1 launches are captured in a hipGraph and replayed.
// HBM-bound: a step reads 12 B per stored Φ entry (value + hoff); ŵ, x stay in L2.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

#include "../../include/sls_mi355x.h"
#include "sls_internal.h"
#include "sls_symbolic.h"

namespace {

struct LoopParams {
  const int32_t *beta_ptr, *u_ptr, *hoff;
  const double* vals;                         // Φ entries in row order (gathered per run)
  const int32_t *A_ptr, *A_idx, *B1_ptr, *B1_idx, *B2_ptr, *B2_idx, *orphan;
  const double *A_val, *B1_val, *B2_val;
  int Nx, Nu, Nw, T, n_orphan;
  long long nscen;
  const double* w;                            // [steps][Nw][nscen] or NULL
  double *x, *u, *what;                       // [steps][Nx][nscen], [steps][Nu][nscen], [T+steps][Nx][nscen]
};

constexpr int kWavesPerBlock = 4;

__global__ void gather_rows_kernel(const double* __restrict__ values, const int32_t* __restrict__ perm, long long n,
                                   double* __restrict__ vals) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) vals[k] = values[perm[k]];
}

// sum over the lanes that share `lane % SCN` (the scenario slot); result in every lane of the group
template <int SCN>
__device__ __forceinline__ double entry_lanes_sum(double v) {
#pragma unroll
  for (int off = 32; off >= SCN; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// lanes = (entry slot le, scenario slot ls): lane = le·SCN + ls.  SCN scenarios share a wave; 64/SCN entries in flight.
template <int SCN>
__global__ void __launch_bounds__(64 * kWavesPerBlock) closed_loop_step_kernel(LoopParams p, int k) {
  constexpr int NE = 64 / SCN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ls = lane % SCN, le = lane / SCN;
  const int row = blockIdx.x * kWavesPerBlock + wave;
  const long long scen = (long long)blockIdx.y * SCN + ls;
  if (row >= p.Nx + p.n_orphan) return;
  const bool live = scen < p.nscen;
  const long long sc = live ? scen : 0;                       // dead lanes read scenario 0 and never write
  const long long ns = p.nscen;
  const double* __restrict__ hist = p.what + (long long)(k + p.T) * p.Nx * ns + sc;
  auto fir = [&](int beg, int end) {                           // this lane's share of Σ_e vals[e]·ŵ[(k+T)·Nx − hoff[e]]
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;               // four entries in flight per lane: the row is a latency chain otherwise
    int e = beg + le;
    for (; e + 3 * NE < end; e += 4 * NE) {
      const double v0 = p.vals[e], v1 = p.vals[e + NE], v2 = p.vals[e + 2 * NE], v3 = p.vals[e + 3 * NE];
      const long long h0 = p.hoff[e], h1 = p.hoff[e + NE], h2 = p.hoff[e + 2 * NE], h3 = p.hoff[e + 3 * NE];
      a0 = fma(v0, hist[-h0 * ns], a0);
      a1 = fma(v1, hist[-h1 * ns], a1);
      a2 = fma(v2, hist[-h2 * ns], a2);
      a3 = fma(v3, hist[-h3 * ns], a3);
    }
    for (; e < end; e += NE) a0 = fma(p.vals[e], hist[-(long long)p.hoff[e] * ns], a0);
    return (a0 + a1) + (a2 + a3);
  };
  if (row >= p.Nx) {                                           // actuator that drives no state: u only
    const int j = p.orphan[row - p.Nx];
    const double uj = entry_lanes_sum<SCN>(fir(p.u_ptr[j], p.u_ptr[j + 1]));
    if (le == 0 && live) p.u[((long long)(k - 1) * p.Nu + j) * ns + scen] = uj;
    return;
  }
  const int i = row;
  const double beta = entry_lanes_sum<SCN>(fir(p.beta_ptr[i], p.beta_ptr[i + 1]));
  double part = 0.0;                                           // A·x[k−1] + B1·w[k−1], this lane's share
  const double* __restrict__ xprev = p.x + (long long)(k - 1) * p.Nx * ns + sc;
  for (int e = p.A_ptr[i] + le; e < p.A_ptr[i + 1]; e += NE) part = fma(p.A_val[e], xprev[(long long)p.A_idx[e] * ns], part);
  if (p.w) {
    const double* __restrict__ wprev = p.w + (long long)(k - 1) * p.Nw * ns + sc;
    for (int e = p.B1_ptr[i] + le; e < p.B1_ptr[i + 1]; e += NE) part = fma(p.B1_val[e], wprev[(long long)p.B1_idx[e] * ns], part);
  }
  double xi = entry_lanes_sum<SCN>(part);
  for (int e = p.B2_ptr[i]; e < p.B2_ptr[i + 1]; ++e) {
    const int j = p.B2_idx[e];
    const double uj = entry_lanes_sum<SCN>(fir(p.u_ptr[j], p.u_ptr[j + 1]));
    xi = fma(p.B2_val[e], uj, xi);
    if (le == 0 && live) p.u[((long long)(k - 1) * p.Nu + j) * ns + scen] = uj;
  }
  if (le == 0 && live) {
    p.x[((long long)k * p.Nx + i) * ns + scen] = xi;
    p.what[((long long)(k + p.T) * p.Nx + i) * ns + scen] = xi - beta;
  }
}

template <int SCN>
hipError_t launch_step(const LoopParams& p, int k, hipStream_t st) {
  const int rows = p.Nx + p.n_orphan;
  dim3 grid((rows + kWavesPerBlock - 1) / kWavesPerBlock, (unsigned)((p.nscen + SCN - 1) / SCN));
  hipLaunchKernelGGL(closed_loop_step_kernel<SCN>, grid, dim3(64 * kWavesPerBlock), 0, st, p, k);
  return hipGetLastError();
}

hipError_t launch_step_any(int scn, const LoopParams& p, int k, hipStream_t st) {
  switch (scn) {
    case 1: return launch_step<1>(p, k, st);
    case 2: return launch_step<2>(p, k, st);
    case 4: return launch_step<4>(p, k, st);
    case 8: return launch_step<8>(p, k, st);
    case 16: return launch_step<16>(p, k, st);
    case 32: return launch_step<32>(p, k, st);
    default: return launch_step<64>(p, k, st);
  }
}

}  // namespace

struct sls_loop {
  sls_ctx* ctx = nullptr;
  int dev = 0;
  sls::FirOperator op;                 // host copy of the dimensions (vectors are released after upload)
  int64_t n_entries = 0;
  void* arena = nullptr;               // operator tables
  const int32_t* d_perm = nullptr;
  LoopParams kp{};
  double* d_vals = nullptr;
  // per-(steps, nscen) state
  double* d_what = nullptr; int64_t what_cap = 0;
  hipStream_t cap_stream = nullptr;
  hipGraphExec_t gexec = nullptr;
  struct Key { const void *w, *x, *u; int64_t steps, nscen; } key{nullptr, nullptr, nullptr, 0, 0};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
};

using namespace sls;

namespace {

int enqueue_steps(sls_loop* L, const LoopParams& p, int64_t steps, hipStream_t st) {
  int scn = 1;
  while (scn < 64 && scn < p.nscen) scn <<= 1;
  const size_t slab = (size_t)p.Nx * p.nscen * sizeof(double);
  HIPCHK(L->ctx, hipMemsetAsync(p.what, 0, slab * (p.T + 1), st));                       // ŵ = 0 up to and including t = 1
  HIPCHK(L->ctx, hipMemsetAsync(p.x, 0, slab, st));                                      // x[:,1] = 0
  if (p.Nu > 0) HIPCHK(L->ctx, hipMemsetAsync(p.u + (size_t)(steps - 1) * p.Nu * p.nscen, 0, (size_t)p.Nu * p.nscen * sizeof(double), st));
  for (int64_t k = 1; k < steps; ++k) HIPCHK(L->ctx, launch_step_any(scn, p, (int)k, st));
  return 0;
}

}  // namespace

extern "C" {

int sls_closed_loop_plan(sls_ctx* ctx, int dev_slot, const sls_dims* dims, const sls_plant* P, const sls_csc_bool* Sx,
                         const sls_csc_bool* Su, sls_loop** loop_out) {
  if (!ctx) return fail(nullptr, SLS_EINVAL, "sls_closed_loop_plan: null context");
  if (!loop_out || !dims || !P) return fail(ctx, SLS_EINVAL, "sls_closed_loop_plan: null argument");
  if (dev_slot < 0 || dev_slot >= (int)ctx->devs.size()) return fail(ctx, SLS_EINVAL, "sls_closed_loop_plan: dev_slot out of range");
  *loop_out = nullptr;
  sls_loop* L = new (std::nothrow) sls_loop();
  if (!L) return fail(ctx, SLS_ENOMEM, "sls_closed_loop_plan: out of memory");
  L->ctx = ctx; L->dev = ctx->devs[dev_slot];
  std::string msg;
  int rc = build_fir_operator(dims, P->A, P->B1, P->B2, Sx, Su, L->op, msg);
  if (rc) { delete L; return fail(ctx, rc, "sls_closed_loop_plan: " + msg); }
  FirOperator& F = L->op;
  L->n_entries = (int64_t)F.hoff.size();
  hipError_t e = hipSetDevice(L->dev);
  if (e != hipSuccess) { delete L; return hipfail(ctx, e, "hipSetDevice"); }
  // one arena for the operator tables
  struct Req { const void* src; size_t bytes; const void** out; };
  std::vector<Req> reqs;
  auto add = [&](const auto& v, const auto** out) {
    reqs.push_back({v.data(), v.size() * sizeof(v[0]), reinterpret_cast<const void**>(out)});
  };
  LoopParams& kp = L->kp;
  add(F.beta_ptr, &kp.beta_ptr); add(F.u_ptr, &kp.u_ptr); add(F.hoff, &kp.hoff); add(F.perm, &L->d_perm);
  add(F.A.ptr, &kp.A_ptr); add(F.A.idx, &kp.A_idx); add(F.A.val, &kp.A_val);
  add(F.B1.ptr, &kp.B1_ptr); add(F.B1.idx, &kp.B1_idx); add(F.B1.val, &kp.B1_val);
  add(F.B2.ptr, &kp.B2_ptr); add(F.B2.idx, &kp.B2_idx); add(F.B2.val, &kp.B2_val);
  add(F.orphan, &kp.orphan);
  auto al = [](size_t b) { return (std::max<size_t>(b, 16) + 255) / 256 * 256; };
  size_t total = 0;
  for (auto& r : reqs) total += al(r.bytes);
  const size_t vals_off = total;
  total += al((size_t)std::max<int64_t>(L->n_entries, 1) * sizeof(double));
  e = hipMalloc(&L->arena, total);
  if (e != hipSuccess) { delete L; return hipfail(ctx, e, "hipMalloc (closed-loop operator)"); }
  size_t off = 0;
  for (auto& r : reqs) {
    if (r.bytes) {
      e = hipMemcpy(static_cast<unsigned char*>(L->arena) + off, r.src, r.bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) { (void)hipFree(L->arena); delete L; return hipfail(ctx, e, "hipMemcpy H2D (closed-loop operator)"); }
    }
    *r.out = static_cast<unsigned char*>(L->arena) + off;
    off += al(r.bytes);
  }
  L->d_vals = reinterpret_cast<double*>(static_cast<unsigned char*>(L->arena) + vals_off);
  kp.vals = L->d_vals;
  kp.Nx = (int)F.Nx; kp.Nu = (int)F.Nu; kp.Nw = (int)F.Nw; kp.T = (int)F.T; kp.n_orphan = (int)F.orphan.size();
  // host vectors are no longer needed
  F.hoff = {}; F.perm = {}; F.beta_ptr = {}; F.u_ptr = {}; F.A = {}; F.B1 = {}; F.B2 = {};
  *loop_out = L;
  return 0;
}

int sls_closed_loop_run(sls_loop* L, void* hip_stream, const double* d_values, const double* d_w, int64_t steps,
                        int64_t nscen, double* d_x, double* d_u) {
  if (!L) return fail(nullptr, SLS_EINVAL, "sls_closed_loop_run: null loop");
  sls_ctx* ctx = L->ctx;
  if (!d_values || !d_x || (!d_u && L->kp.Nu > 0)) return fail(ctx, SLS_EINVAL, "sls_closed_loop_run: null device pointer");
  if (steps < 1 || nscen < 1) return fail(ctx, SLS_EINVAL, "sls_closed_loop_run: steps and nscen must be >= 1");
  if (steps > 0x7ffffff0LL) return fail(ctx, SLS_EUNSUPPORTED, "sls_closed_loop_run: steps exceeds int32");
  HIPCHK(ctx, hipSetDevice(L->dev));
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int64_t need = (steps + L->kp.T) * (int64_t)L->kp.Nx * nscen;
  if (need > L->what_cap) {
    if (L->gexec) { (void)hipGraphExecDestroy(L->gexec); L->gexec = nullptr; }
    if (L->d_what) { HIPCHK(ctx, hipStreamSynchronize(st)); (void)hipFree(L->d_what); L->d_what = nullptr; L->what_cap = 0; }
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(&L->d_what), (size_t)need * sizeof(double)));
    L->what_cap = need;
  }
  if (!L->ev0) { HIPCHK(ctx, hipEventCreate(&L->ev0)); HIPCHK(ctx, hipEventCreate(&L->ev1)); }
  LoopParams p = L->kp;
  p.nscen = nscen; p.w = d_w; p.x = d_x; p.u = d_u; p.what = L->d_what;
  HIPCHK(ctx, hipEventRecord(L->ev0, st));
  if (L->n_entries > 0) {                                      // Φ: mask order → row order
    const int bs = 256;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((L->n_entries + bs - 1) / bs)), dim3(bs), 0, st, d_values, L->d_perm,
                       (long long)L->n_entries, L->d_vals);
    HIPCHK(ctx, hipGetLastError());
  }
  static const bool no_graph = sls_knob("SLS_NO_GRAPH") != nullptr;
  if (no_graph || steps < 3) {
    int rc = enqueue_steps(L, p, steps, st);
    if (rc) return rc;
  } else {
    const sls_loop::Key key{d_w, d_x, d_u, steps, nscen};
    const bool same = L->gexec && key.w == L->key.w && key.x == L->key.x && key.u == L->key.u && key.steps == L->key.steps &&
                      key.nscen == L->key.nscen;
    if (!same) {
      if (L->gexec) { (void)hipGraphExecDestroy(L->gexec); L->gexec = nullptr; }
      if (!L->cap_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&L->cap_stream, hipStreamNonBlocking));
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamBeginCapture(L->cap_stream, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        int rc = enqueue_steps(L, p, steps, L->cap_stream);
        e = hipStreamEndCapture(L->cap_stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
      }
      if (e == hipSuccess) {
        e = hipGraphInstantiate(&L->gexec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
      }
      if (e != hipSuccess) {                                   // capture unavailable (e.g. a runtime that refuses it): plain launches
        L->gexec = nullptr;
        (void)hipGetLastError();
        int rc = enqueue_steps(L, p, steps, st);
        if (rc) return rc;
        HIPCHK(ctx, hipEventRecord(L->ev1, st));
        L->timed = true;
        return 0;
      }
      L->key = key;
    }
    HIPCHK(ctx, hipGraphLaunch(L->gexec, st));
  }
  HIPCHK(ctx, hipEventRecord(L->ev1, st));
  L->timed = true;
  return 0;
}

int sls_closed_loop_run_host(sls_loop* L, const double* d_values, const double* h_w, int64_t steps, int64_t nscen, double* h_x,
                             double* h_u) {
  if (!L) return fail(nullptr, SLS_EINVAL, "sls_closed_loop_run_host: null loop");
  sls_ctx* ctx = L->ctx;
  if (!h_x || (!h_u && L->kp.Nu > 0) || steps < 1 || nscen < 1) return fail(ctx, SLS_EINVAL, "sls_closed_loop_run_host: bad argument");
  HIPCHK(ctx, hipSetDevice(L->dev));
  const size_t nx = (size_t)steps * L->kp.Nx * nscen, nu = (size_t)steps * L->kp.Nu * nscen, nw = (size_t)steps * L->kp.Nw * nscen;
  double* buf = nullptr;
  HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(&buf), (nx + std::max<size_t>(nu, 1) + (h_w ? nw : 0)) * sizeof(double)));
  double *dx = buf, *du = buf + nx, *dw = h_w ? du + std::max<size_t>(nu, 1) : nullptr;
  int rc = 0;
  hipError_t e = hipSuccess;
  if (h_w) e = hipMemcpy(dw, h_w, nw * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = sls_closed_loop_run(L, nullptr, d_values, dw, steps, nscen, dx, du);
    if (!rc) e = hipStreamSynchronize(nullptr);
    if (!rc && e == hipSuccess) e = hipMemcpy(h_x, dx, nx * sizeof(double), hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess && nu) e = hipMemcpy(h_u, du, nu * sizeof(double), hipMemcpyDeviceToHost);
  }
  // the graph is keyed on these buffers: drop it with them
  if (L->gexec) { (void)hipGraphExecDestroy(L->gexec); L->gexec = nullptr; }
  (void)hipFree(buf);
  if (rc) return rc;
  if (e != hipSuccess) return hipfail(ctx, e, "sls_closed_loop_run_host");
  return 0;
}

int sls_closed_loop_last_ms(sls_loop* L, double* ms) {
  if (!L || !ms) return fail(L ? L->ctx : nullptr, SLS_EINVAL, "sls_closed_loop_last_ms: null argument");
  if (!L->timed) return fail(L->ctx, SLS_EINVAL, "sls_closed_loop_last_ms: no run recorded");
  HIPCHK(L->ctx, hipSetDevice(L->dev));
  HIPCHK(L->ctx, hipEventSynchronize(L->ev1));
  float f = 0.f;
  HIPCHK(L->ctx, hipEventElapsedTime(&f, L->ev0, L->ev1));
  *ms = f;
  return 0;
}

int sls_closed_loop_entries(const sls_loop* L, int64_t* n_entries) {
  if (!L || !n_entries) return fail(nullptr, SLS_EINVAL, "sls_closed_loop_entries: null argument");
  *n_entries = L->n_entries;
  return 0;
}

void sls_closed_loop_destroy(sls_loop* L) {
  if (!L) return;
  (void)hipSetDevice(L->dev);
  if (L->gexec) (void)hipGraphExecDestroy(L->gexec);
  if (L->cap_stream) (void)hipStreamDestroy(L->cap_stream);
  if (L->ev0) (void)hipEventDestroy(L->ev0);
  if (L->ev1) (void)hipEventDestroy(L->ev1);
  if (L->d_what) (void)hipFree(L->d_what);
  if (L->arena) (void)hipFree(L->arena);
  delete L;
}

}  // extern "C"
