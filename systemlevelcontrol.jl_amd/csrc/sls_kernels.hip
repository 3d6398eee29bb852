// sls_kernels.hip — gfx950 kernels of the batched column-separable H2 SLS solve.
//
// What one workgroup does = one iteration of the reference's hot loop
// (src/synthesis.jl:37-68): build the localized sub-plant (src/reduction.jl:15,
// GeneralizedPlant.jl:266-285), solve the equality-constrained QP that the reference
// hands to JuMP/Ipopt (src/synthesis.jl:46-62) and scatter the masked optimum
// (src/synthesis.jl:65-67).
//
// Math.  Free variables z = {x_t[m_x[t]], u_t[m_u[t]]}, t = 0..T-1, cost
// ½ zᵀHz + gᵀz with H diagonal, constraints E z = f:
//     block-row 0   :  x_0                         = e_pos
//     block-row k   :  x_k − Ã x_{k−1} − B̃ u_{k−1}  = 0      (1 ≤ k ≤ T−1)
//     block-row T   :      − Ã x_{T−1} − B̃ u_{T−1}  = 0
// z(λ) = H⁻¹(Eᵀλ − g);   S = E H⁻¹ Eᵀ is block tridiagonal with ñx×ñx blocks
//     D_k = [k≥1](Ã Wx_{k−1} Ãᵀ + B̃ Wu_{k−1} B̃ᵀ) + [k≤T−1] Wx_k ,   L_k = −Ã Wx_{k−1}
// (W = mask ⊙ H⁻¹).  S is singular whenever E has dependent or empty rows (it does on
// every README column), so we factor S + δI (block LDLᵀ with explicit inverse pivot
// blocks P_k) and run the method of multipliers / iterated Tikhonov refinement
//     λ ← λ + (S+δI)⁻¹ (f − E z(λ))
// which converges to the unique optimum for every consistent system; z stays in
// H⁻¹(range(Eᵀ) − g) by construction, so stationarity holds exactly and the
// residual ‖f − E z‖∞ is the whole optimality certificate.
//
// "General" kernel (ñx or ñu > 64): 256 threads per subproblem, the working blocks in LDS,
// Ã kept sparse (local CSR + CSC gathered from the shared CSR operator in HBM), the block
// inversion in register tiles on a 16×16 thread grid, P_k streamed to an L2-resident
// workspace.  FP64 throughout.  The ñx ≤ 64 classes live in sls_wave_kernel.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdint.h>
#include "sls_device.h"

namespace sls {

constexpr int BLOCK = 256;

__device__ __forceinline__ int bsearch_i32(const int32_t* a, int n, int32_t key) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    int mid = (lo + hi) >> 1;
    int32_t v = a[mid];
    if (v == key) return mid;
    if (v < key) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

__device__ __forceinline__ double block_max(double v, double* red, int tid) {
  // wave reduce then cross-wave through LDS
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double r = red[0];
  for (int w = 1; w < BLOCK / 64; ++w) r = fmax(r, red[w]);
  __syncthreads();
  return r;
}

// out[i] = Σ_j P[j*n+i]·y[j]   (P symmetric up to rounding; column read = coalesced)
// all threads of the block call this; y in LDS; result valid in `dst` (LDS) after return.
__device__ __forceinline__ void block_sym_matvec(const double* __restrict__ P, const double* y,
                                                 double* dst, double* partial, int n, int tid) {
  const int S = (BLOCK / n) > 0 ? (BLOCK / n) : 1;
  if (n <= BLOCK) {
    const int s = tid / n, i = tid - s * n;
    if (s < S) {
      // eight loads of P in flight per thread (P comes from the workspace: a dependent load→FMA chain costs a full L2 round
      // trip per term); the tail is predicated through a clamped index with a zero weight
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      for (int j0 = s; j0 < n; j0 += 8 * S) {
        double pv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = j0 + u * S;
          const int jc = min(j, n - 1);
          pv[u] = P[(int64_t)jc * n + i];
          yv[u] = (j < n) ? y[jc] : 0.0;
        }
        a0 = fma(pv[0], yv[0], a0); a1 = fma(pv[1], yv[1], a1); a2 = fma(pv[2], yv[2], a2); a3 = fma(pv[3], yv[3], a3);
        a0 = fma(pv[4], yv[4], a0); a1 = fma(pv[5], yv[5], a1); a2 = fma(pv[6], yv[6], a2); a3 = fma(pv[7], yv[7], a3);
      }
      partial[s * n + i] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (tid < n) {
      double acc = 0.0;
      for (int s2 = 0; s2 < S; ++s2) acc += partial[s2 * n + tid];
      dst[tid] = acc;
    }
    __syncthreads();
  }
}

// TIMAX = 6, OG = false: ñx ≤ 96 with both block images in LDS.  TIMAX = 9, OG = true ("wide"): ñx ≤ 144, the Ã·Q image lives in the
// workgroup's global workspace instead (written once and gathered once per block step; the LDS keeps P_k, the dense B̃ and the lists).
template <int TIMAX, bool OG>
__global__ __launch_bounds__(BLOCK) void h2_column_general_kernel(const KernelParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x;
  const int T = p.T;
  const int nmax = p.nmax, mmax = p.mmax;

  // ---- LDS carve (doubles first) ----
  double* dp = reinterpret_cast<double*>(lds_raw);
  double* bufA = dp; dp += general_kernel_block_doubles(nmax, mmax);
  double* bufB = dp; if (!OG) dp += general_kernel_block_doubles(nmax, mmax);
  double* Bd = dp;   if (!OG) dp += (int64_t)nmax * mmax;    // wide variant: the dense B̃ lives in the global workspace as well
  double* csrA_v = dp; dp += p.nnzA_cap;
  double* cscA_v = dp; dp += p.nnzA_cap;
  double* csrB_v = dp; dp += p.nnzB_cap;
  double* hx = dp; dp += nmax;
  double* gx = dp; dp += nmax;
  double* hu = dp; dp += mmax;
  double* gu = dp; dp += mmax;
  double* wprev = dp; dp += nmax;
  double* wcur = dp;  dp += nmax;
  double* wuprev = dp; dp += mmax;
  double* xt = dp;   dp += nmax;
  double* base = dp; dp += nmax;
  double* tmp = dp;  dp += nmax;
  double* tmp2 = dp; dp += nmax;
  double* ut = dp;   dp += mmax;
  double* xbuf = dp; if (OG) dp += 4 * 144;                 // wide variant: exchange buffers (the other variant uses the idle Ã·Q image)
  double* red = dp;  dp += 256;
  double* partial = dp; dp += 256;
  double* vec_lds = dp;
  if (p.vec_in_lds) dp += 3LL * (T + 1) * nmax;
  int32_t* ip = reinterpret_cast<int32_t*>(dp);
  int32_t* sx = ip; ip += nmax;
  int32_t* su = ip; ip += mmax;
  int32_t* csrA_p = ip; ip += nmax + 1;
  int32_t* cscA_p = ip; ip += nmax + 1;
  int32_t* csrA_i = ip; ip += p.nnzA_cap;
  int32_t* cscA_i = ip; ip += p.nnzA_cap;
  int32_t* csrB_p = ip; ip += nmax + 1;
  int32_t* csrB_i = ip; ip += p.nnzB_cap;

  double* facws = p.fac_ws + (int64_t)blockIdx.x * p.fac_stride;
  if constexpr (OG) Bd = facws + (int64_t)(T + 2) * nmax * nmax;       // [T+1 pivot blocks][Ã·Q image][dense B̃]
  double* vecs = p.vec_in_lds ? vec_lds : (p.vec_ws + (int64_t)blockIdx.x * p.vec_stride);

  for (int it_sub = blockIdx.x; it_sub < p.nsub; it_sub += gridDim.x) {
    const SubDesc sd = p.subs[p.order[p.order_off + it_sub]];
    const int n = sd.n, m = sd.m, nm = n + m;
    double* lam = vecs;
    double* qv = vecs + (int64_t)(T + 1) * n;
    double* rv = vecs + 2LL * (T + 1) * n;
    const uint8_t* mask = p.mask_pool + sd.off_mask;
    const int32_t* dest = p.dest_pool + sd.off_dest;

    __syncthreads();   // previous subproblem fully done with LDS
    unsigned long long tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // SLS_PHASE_TIMERS: setup, residual, build, gj, store, sweeps
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    auto lap = [&](int slot) { if (p.dbg) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tc[slot] += now - tlast; tlast = now; } };
    // ---- stage index sets and weights ----
    for (int i = tid; i < n; i += BLOCK) {
      sx[i] = p.idx_pool[sd.off_sx + i];
      hx[i] = sd.has_w ? p.w_pool[sd.off_w + i] : 1.0;
      gx[i] = sd.has_w ? p.w_pool[sd.off_w + nm + i] : 0.0;
    }
    for (int i = tid; i < m; i += BLOCK) {
      su[i] = p.idx_pool[sd.off_su + i];
      hu[i] = sd.has_w ? p.w_pool[sd.off_w + n + i] : 1.0;
      gu[i] = sd.has_w ? p.w_pool[sd.off_w + nm + n + i] : 0.0;
    }
    for (int i = tid; i < n * m; i += BLOCK) Bd[i] = 0.0;
    __syncthreads();

    // ---- gather Ã (CSR + CSC) and B̃ (CSR + dense) from the shared operator ----
    // pass 1: counts
    for (int i = tid; i < n; i += BLOCK) {
      const int g = sx[i];
      int c1 = 0, c2 = 0, c3 = 0;
      for (int e = p.A_rowptr[g]; e < p.A_rowptr[g + 1]; ++e)
        if (p.A_val[e] != 0.0 && bsearch_i32(sx, n, p.A_colidx[e]) >= 0) ++c1;
      for (int e = p.At_rowptr[g]; e < p.At_rowptr[g + 1]; ++e)
        if (p.At_val[e] != 0.0 && bsearch_i32(sx, n, p.At_colidx[e]) >= 0) ++c2;
      for (int e = p.B_rowptr[g]; e < p.B_rowptr[g + 1]; ++e)
        if (p.B_val[e] != 0.0 && bsearch_i32(su, m, p.B_colidx[e]) >= 0) ++c3;
      csrA_p[i + 1] = c1; cscA_p[i + 1] = c2; csrB_p[i + 1] = c3;
    }
    __syncthreads();
    if (tid == 0) {
      csrA_p[0] = 0; cscA_p[0] = 0; csrB_p[0] = 0;
      for (int i = 0; i < n; ++i) {
        csrA_p[i + 1] += csrA_p[i]; cscA_p[i + 1] += cscA_p[i]; csrB_p[i + 1] += csrB_p[i];
      }
    }
    __syncthreads();
    // pass 2: fill
    for (int i = tid; i < n; i += BLOCK) {
      const int g = sx[i];
      int w1 = csrA_p[i], w2 = cscA_p[i], w3 = csrB_p[i];
      for (int e = p.A_rowptr[g]; e < p.A_rowptr[g + 1]; ++e) {
        const double v = p.A_val[e];
        const int loc = (v != 0.0) ? bsearch_i32(sx, n, p.A_colidx[e]) : -1;
        if (loc >= 0) { csrA_i[w1] = loc; csrA_v[w1] = v; ++w1; }
      }
      for (int e = p.At_rowptr[g]; e < p.At_rowptr[g + 1]; ++e) {
        const double v = p.At_val[e];
        const int loc = (v != 0.0) ? bsearch_i32(sx, n, p.At_colidx[e]) : -1;
        if (loc >= 0) { cscA_i[w2] = loc; cscA_v[w2] = v; ++w2; }
      }
      for (int e = p.B_rowptr[g]; e < p.B_rowptr[g + 1]; ++e) {
        const double v = p.B_val[e];
        const int loc = (v != 0.0) ? bsearch_i32(su, m, p.B_colidx[e]) : -1;
        if (loc >= 0) { csrB_i[w3] = loc; csrB_v[w3] = v; Bd[i * m + loc] = v; ++w3; }
      }
    }
    __syncthreads();

    // ---- regularisation scale: largest possible Schur diagonal ----
    double sc = 0.0;
    for (int i = tid; i < n; i += BLOCK) {
      double s = hx[i];
      for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) s = fma(csrA_v[e] * csrA_v[e], hx[csrA_i[e]], s);
      for (int e = csrB_p[i]; e < csrB_p[i + 1]; ++e) s = fma(csrB_v[e] * csrB_v[e], hu[csrB_i[e]], s);
      sc = fmax(sc, s);
    }
    sc = block_max(sc, red, tid);
    const double delta = p.delta_rel * sc;

    for (int i = tid; i < (T + 1) * n; i += BLOCK) lam[i] = 0.0;
    __syncthreads();

    // residual pass: r = f − E z(λ); writes z to the output; returns ‖r‖∞.
    // λ_t, λ_{t+1} are staged in LDS (λ may live in the global workspace) one time step ahead, x_t/u_t and the row carried
    // to the next step (Ãx_t + B̃u_t) are double-buffered: one barrier per time step.  The staging area is the pair of block
    // images, which is idle outside the factorisation.
    auto residual_pass = [&]() -> double {
      double rmax = 0.0;
      double* const stage = bufA;                                   // [3][nmax] λ slices, then [2][nmax] carried rows, then [2][nmax+mmax] x,u
      double* const carry0 = stage + 3 * nmax;
      double* const xu0 = stage + 5 * nmax;
      const int xus = nmax + mmax;
      for (int i = tid; i < n; i += BLOCK) {
        carry0[i] = (i == sd.pos) ? 1.0 : 0.0;                      // f_0 = e_pos
        stage[i] = lam[i];
        stage[nmax + i] = lam[(int64_t)n + i];
      }
      __syncthreads();
      for (int t = 0; t < T; ++t) {
        const double* l0 = stage + (t % 3) * nmax;
        const double* l1 = stage + ((t + 1) % 3) * nmax;
        double* l2 = stage + ((t + 2) % 3) * nmax;
        double* xt_ = xu0 + (t & 1) * xus;
        double* ut_ = xt_ + nmax;
        if (t + 2 <= T) for (int i = tid; i < n; i += BLOCK) l2[i] = lam[(int64_t)(t + 2) * n + i];
        const uint8_t* mk = mask + (int64_t)t * nm;
        const int32_t* ds = dest + (int64_t)t * nm;
        for (int q = tid; q < nm; q += BLOCK) {
          if (q < n) {
            double acc = 0.0;
            for (int e = cscA_p[q]; e < cscA_p[q + 1]; ++e) acc = fma(cscA_v[e], l1[cscA_i[e]], acc);
            const double v = mk[q] ? hx[q] * (l0[q] - acc - gx[q]) : 0.0;
            xt_[q] = v;
            if (mk[q]) { const int d = ds[q]; if (d >= 0) p.out[d] = v; }
          } else {
            const int j = q - n;
            double acc = 0.0;
            for (int i = 0; i < n; ++i) acc = fma(Bd[i * m + j], l1[i], acc);
            const double v = mk[q] ? hu[j] * (-acc - gu[j]) : 0.0;
            ut_[j] = v;
            if (mk[q]) { const int d = ds[q]; if (d >= 0) p.out[d] = v; }
          }
        }
        __syncthreads();
        const double* cin = carry0 + (t & 1) * nmax;
        double* cout = carry0 + ((t + 1) & 1) * nmax;
        for (int i = tid; i < n; i += BLOCK) {
          const double r = cin[i] - xt_[i];
          rv[(int64_t)t * n + i] = r;
          rmax = resid_max(rmax, r);
          double acc = 0.0;
          for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) acc = fma(csrA_v[e], xt_[csrA_i[e]], acc);
          for (int e = csrB_p[i]; e < csrB_p[i + 1]; ++e) acc = fma(csrB_v[e], ut_[csrB_i[e]], acc);
          cout[i] = acc;
        }
      }
      __syncthreads();
      const double* cfin = carry0 + (T & 1) * nmax;
      for (int i = tid; i < n; i += BLOCK) {
        rv[(int64_t)T * n + i] = cfin[i];
        rmax = resid_max(rmax, cfin[i]);
      }
      return block_max(rmax, red, tid);
    };

    lap(0);
    double resid;
    if (!sd.has_w && sd.pos >= 0) {              // g = 0 and λ = 0: z = 0 and r = f = e_pos exactly — no pass needed to know it
      for (int i = tid; i < (T + 1) * n; i += BLOCK) rv[i] = (i == sd.pos) ? 1.0 : 0.0;
      __syncthreads();
      resid = 1.0;
    } else {
      resid = residual_pass();
    }
    lap(1);
    int iters = 0;
    int status = 0;

    if (resid > p.tol) {
      // =================== factor: P_k = (D_k − L_k P_{k−1} L_kᵀ + δI)⁻¹ ===================
      // Thread (ty, tx) = (tid / 16, tid % 16) owns the tile {rows ty + 16a} × {columns tx + 16b}, a, b < TI = ⌈n/16⌉ ≤ 6, of
      // the block being inverted and keeps it in REGISTERS through the whole Gauss–Jordan; per pivot only the pivot row and
      // column travel through LDS (double-buffered: one barrier per pivot).  v1 kept the block in LDS (ping-pong copy per
      // pivot, an integer division and an IEEE FP64 division per element): 10.2 M cycles per ñx = 85 column against 2.5 M.
      const int ty = tid >> 4, tx = tid & 15;
      const int TI = (n + 15) >> 4;
      double* Pcur = bufA;    // holds P_{k−1}
      double* Oth;
      if constexpr (OG) Oth = facws + (int64_t)(T + 1) * nmax * nmax; else Oth = bufB;
      // pivot column / row exchange buffers, double-buffered.  Plain
      // offsets from ONE LDS base keep the accesses ds_read/ds_write — a pointer chosen at run time from an array degrades
      // to FLAT loads and stores (that, and a ping-pong of the whole block through such pointers, is what made v1 slow)
      // Padded to TIMAX·16 entries so that the owners write their whole tile row/column without per-entry guards.
      constexpr int PADN = TIMAX * 16;
      double* colbuf0; double* rowbuf0;
      if constexpr (OG) { colbuf0 = xbuf; rowbuf0 = xbuf + 2 * PADN; } else { colbuf0 = bufB; rowbuf0 = bufB + 2 * PADN; }
      for (int k = 0; k <= T; ++k) {
        // weights of this block row
        if (k >= 1) {
          const uint8_t* mk = mask + (int64_t)(k - 1) * nm;
          for (int i = tid; i < n; i += BLOCK) wprev[i] = mk[i] ? hx[i] : 0.0;
          for (int i = tid; i < m; i += BLOCK) wuprev[i] = mk[n + i] ? hu[i] : 0.0;
        }
        if (k <= T - 1) {
          const uint8_t* mk = mask + (int64_t)k * nm;
          for (int i = tid; i < n; i += BLOCK) wcur[i] = mk[i] ? hx[i] : 0.0;
        } else {
          for (int i = tid; i < n; i += BLOCK) wcur[i] = 0.0;
        }
        __syncthreads();
        double R[TIMAX][TIMAX];
        if (k >= 1) {
          // Oth = Ã·Q,  Q = W − W P W  (W = diag(wprev)).  Row i of Ã is walked once per tile row: each entry (q, v) is
          // read once and applied to the ≤ 6 tile columns (v1 re-read the entry and its weight for every element)
          double wc[TIMAX];
#pragma unroll
          for (int b = 0; b < TIMAX; ++b) wc[b] = wprev[min(tx + 16 * b, n - 1)];
#pragma unroll
          for (int a = 0; a < TIMAX; ++a) {
            const int i = ty + 16 * a;
            if (a < TI && i < n) {
              double acc[TIMAX];
#pragma unroll
              for (int b = 0; b < TIMAX; ++b) acc[b] = 0.0;
              for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) {
                const int q = csrA_i[e];
                const double vw = csrA_v[e] * wprev[q];
                const double* Pq = Pcur + q * n;
#pragma unroll
                for (int b = 0; b < TIMAX; ++b) {
                  const int c = min(tx + 16 * b, n - 1);
                  acc[b] = fma(vw, ((q == c) ? 1.0 : 0.0) - Pq[c] * wc[b], acc[b]);
                }
              }
#pragma unroll
              for (int b = 0; b < TIMAX; ++b) { const int c = tx + 16 * b; if (b < TI && c < n) Oth[i * n + c] = acc[b]; }
            }
          }
          __syncthreads();
        }
        // D'_k = δI + Wx_k + B̃ Wu B̃ᵀ + Oth·Ãᵀ straight into the register tile; rows j of B̃ and Ã are walked once per tile column
#pragma unroll
        for (int b = 0; b < TIMAX; ++b) {
          const int j = tx + 16 * b;
          const bool jok = b < TI && j < n;
#pragma unroll
          for (int a = 0; a < TIMAX; ++a) R[a][b] = (jok && ty + 16 * a == j) ? (delta + wcur[j]) : 0.0;
          if (k >= 1 && jok) {
            for (int e = csrB_p[j]; e < csrB_p[j + 1]; ++e) {
              const int q = csrB_i[e];
              const double bw = wuprev[q] * csrB_v[e];
#pragma unroll
              for (int a = 0; a < TIMAX; ++a) R[a][b] = fma(Bd[min(ty + 16 * a, n - 1) * m + q], bw, R[a][b]);
            }
            for (int e = csrA_p[j]; e < csrA_p[j + 1]; ++e) {
              const int ce = csrA_i[e];
              const double ve = csrA_v[e];
#pragma unroll
              for (int a = 0; a < TIMAX; ++a) R[a][b] = fma(Oth[min(ty + 16 * a, n - 1) * n + ce], ve, R[a][b]);
            }
          }
        }
        if constexpr (!OG) __syncthreads();               // the exchange buffers live in the Ã·Q image: every thread is done gathering from it
        lap(2);
        // Gauss–Jordan in registers (SPD ⇒ no pivoting).  The slot pa = pv / 16 of the pivot inside the tiles is a compile-time
        // constant of the unrolled outer loop, so every register access is static; the pivot row and column are produced
        // by the rank-1 update itself (t_p := 1 + d turns column p into −c·d; the pivot row is patched by its 16 owners).
#pragma unroll
        for (int pa = 0; pa < TIMAX; ++pa) {
          if (pa < TI) {
            for (int pt = 0; pt < 16; ++pt) {
              const int pv = pa * 16 + pt;
              if (pv >= n) break;
              double* cb = colbuf0 + (pv & 1) * PADN; double* rb = rowbuf0 + (pv & 1) * PADN;
              if (tx == pt) {
#pragma unroll
                for (int a = 0; a < TIMAX; ++a) cb[ty + 16 * a] = R[a][pa];
              }
              if (ty == pt) {
#pragma unroll
                for (int b = 0; b < TIMAX; ++b) rb[tx + 16 * b] = R[pa][b];
              }
              __syncthreads();
              const double piv = rb[pv];
              double d = __builtin_amdgcn_rcp(piv);
              d = fma(fma(-piv, d, 1.0), d, d);
              d = fma(fma(-piv, d, 1.0), d, d);
              double ci[TIMAX], tj[TIMAX], tf[TIMAX];
#pragma unroll
              for (int a = 0; a < TIMAX; ++a) ci[a] = cb[min(ty + 16 * a, n - 1)];      // padded rows/columns compute on a
#pragma unroll                                                                              // clamped neighbour, never stored
              for (int b = 0; b < TIMAX; ++b) {
                tj[b] = rb[min(tx + 16 * b, n - 1)] * d;
                tf[b] = (b == pa && tx == pt) ? (1.0 + d) : tj[b];
              }
#pragma unroll
              for (int a = 0; a < TIMAX; ++a) {
#pragma unroll
                for (int b = 0; b < TIMAX; ++b) R[a][b] = fma(-ci[a], tf[b], R[a][b]);
              }
              if (ty == pt) {
#pragma unroll
                for (int b = 0; b < TIMAX; ++b) R[pa][b] = (b == pa && tx == pt) ? d : tj[b];
              }
            }
          }
        }
        __syncthreads();                                  // last readers of Oth / the row-column buffers are done
        lap(3);
        // P_k: to LDS (next block's Ã·Q reads it) and to the workspace
        double* Pk = facws + (int64_t)k * n * n;
#pragma unroll
        for (int a = 0; a < TIMAX; ++a) {
          const int i = ty + 16 * a;
#pragma unroll
          for (int b = 0; b < TIMAX; ++b) {
            const int j = tx + 16 * b;
            if (a < TI && b < TI && i < n && j < n) { Pcur[i * n + j] = R[a][b]; Pk[i * n + j] = R[a][b]; }
          }
        }
        lap(4);
        // (next step begins with a barrier)
      }
      __syncthreads();
      __threadfence_block();

      // =================== refinement loop ===================
      double prev = resid, prev2 = resid;
      int itmax = p.max_iters;
      for (int it = 1; it <= itmax; ++it) {
        iters = it;
        // forward: y_k = r_k + Ã(Wx_{k−1} q_{k−1});  q_k = P_k y_k
        for (int k = 0; k <= T; ++k) {
          for (int i = tid; i < n; i += BLOCK) {
            double acc = rv[(int64_t)k * n + i];
            if (k >= 1) {
              const uint8_t* mk = mask + (int64_t)(k - 1) * nm;
              const double* qp = qv + (int64_t)(k - 1) * n;
              for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) {
                const int q = csrA_i[e];
                if (mk[q]) acc = fma(csrA_v[e] * hx[q], qp[q], acc);
              }
            }
            tmp[i] = acc;
          }
          __syncthreads();
          block_sym_matvec(facws + (int64_t)k * n * n, tmp, tmp2, partial, n, tid);
          for (int i = tid; i < n; i += BLOCK) qv[(int64_t)k * n + i] = tmp2[i];
          __syncthreads();
        }
        // backward: Δλ_k = q_k + P_k (Wx_k (Ãᵀ Δλ_{k+1}));  λ += Δλ.   Δλ_k overwrites q_k.
        for (int k = T; k >= 0; --k) {
          if (k < T) {
            const uint8_t* mk = mask + (int64_t)k * nm;
            const double* dl1 = qv + (int64_t)(k + 1) * n;
            for (int q = tid; q < n; q += BLOCK) {
              double acc = 0.0;
              if (mk[q]) {
                for (int e = cscA_p[q]; e < cscA_p[q + 1]; ++e) acc = fma(cscA_v[e], dl1[cscA_i[e]], acc);
                acc *= hx[q];
              }
              tmp[q] = acc;
            }
            __syncthreads();
            block_sym_matvec(facws + (int64_t)k * n * n, tmp, tmp2, partial, n, tid);
            for (int i = tid; i < n; i += BLOCK) qv[(int64_t)k * n + i] += tmp2[i];
            __syncthreads();
          }
          for (int i = tid; i < n; i += BLOCK) lam[(int64_t)k * n + i] += qv[(int64_t)k * n + i];
        }
        __syncthreads();
        lap(5);
        resid = residual_pass();
        lap(1);
        if (resid <= p.tol) break;
        if (it >= 2 && resid > p.stag * prev) {                       // stagnation ⇒ inconsistent system, unless merely slow
          if (!(p.max_iters_slow > 0 && (resid > p.tol_ok || itmax > p.max_iters) && still_contracting(it >= 3 ? prev2 : prev, prev, resid))) { status = 1; break; }
          itmax = max(itmax, p.max_iters_slow);
        }
        prev2 = prev; prev = resid;
      }
      if (resid <= p.tol_ok) status = 0;
      else if (status == 0) status = 2;
    }
    if (sd.pos < 0 && status == 0) status = 3;
    if (tid == 0) {
      p.status[sd.out_index] = status;
      p.resid[sd.out_index] = resid;
      p.iters[sd.out_index] = iters;
      if (p.dbg) for (int q = 0; q < 8; ++q) p.dbg[sd.out_index * 8 + q] = tc[q];
    }
  }
}

__global__ void scatter_f64_kernel(const double* __restrict__ src, const int64_t* __restrict__ idx,
                                   int64_t n, double* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[idx[i]] = src[i];
}

// Destination tables on the device: expands the compact per-column tables of the host symbolic pass
// (Symbolic::compact — a bit mask over (s_x, s_u) per time step and the value-array index of the column's first x / u entry)
// into the byte mask and the int32 destinations the solve kernels read:  mask[t][i] = bit,  dest[t][i] = base + rank of the
// bit inside its part (x: positions [0, ñx), u: [ñx, ñx+ñu)).  The scatter target of src/synthesis.jl:65-67, computed where
// it is used; 5 B per masked position never cross PCIe.  One workgroup per subproblem, thread per position.
__global__ void expand_tables_kernel(const SubDesc* __restrict__ subs, int nsub, int T, const uint64_t* __restrict__ cmask,
                                     const int32_t* __restrict__ cbase, const int64_t* __restrict__ coff,
                                     uint8_t* __restrict__ mask_pool, int32_t* __restrict__ dest_pool) {
  for (int q = blockIdx.x; q < nsub; q += gridDim.x) {
    const SubDesc sd = subs[q];
    const int n = sd.n, nm = sd.n + sd.m, wps = (nm + 63) >> 6;
    const uint64_t* bits = cmask + coff[sd.out_index];
    const int32_t* cb = cbase + 2ll * T * sd.out_index;
    uint8_t* mk = mask_pool + sd.off_mask;
    int32_t* ds = dest_pool + sd.off_dest;
    const int len = T * nm;
    for (int e = threadIdx.x; e < len; e += blockDim.x) {
      const int t = e / nm, i = e - t * nm;
      const uint64_t* bt = bits + (int64_t)t * wps;
      auto below = [&](int j) {                       // set bits at positions < j
        int cnt = 0;
        for (int w = 0; w < (j >> 6); ++w) cnt += __popcll(bt[w]);
        if (j & 63) cnt += __popcll(bt[j >> 6] & ((1ull << (j & 63)) - 1ull));
        return cnt;
      };
      const bool on = (bt[i >> 6] >> (i & 63)) & 1ull;
      mk[e] = on ? 1 : 0;
      ds[e] = !on ? -1 : ((i < n) ? cb[2 * t] + below(i) : cb[2 * t + 1] + below(i) - below(n));
    }
  }
}

}  // namespace sls

// ---- launchers (called from sls_api.cpp through plain C++ declarations) ----
namespace sls {

template <int TIMAX, bool OG>
static hipError_t launch_general_v(const KernelParams& p, int grid, size_t lds_bytes, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&h2_column_general_kernel<TIMAX, OG>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((h2_column_general_kernel<TIMAX, OG>), dim3(grid), dim3(BLOCK), lds_bytes, stream, p);
  return hipGetLastError();
}
hipError_t launch_general(const KernelParams& p, int grid, size_t lds_bytes, hipStream_t stream, bool wide) {
  return wide ? launch_general_v<9, true>(p, grid, lds_bytes, stream) : launch_general_v<6, false>(p, grid, lds_bytes, stream);
}

hipError_t launch_scatter(const double* src, const int64_t* idx, int64_t n, double* dst, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scatter_f64_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, idx, n, dst);
  return hipGetLastError();
}

hipError_t launch_expand_tables(const SubDesc* subs, int nsub, int T, const uint64_t* cmask, const int32_t* cbase, const int64_t* coff,
                                uint8_t* mask_pool, int32_t* dest_pool, hipStream_t stream) {
  if (nsub <= 0) return hipSuccess;
  hipLaunchKernelGGL(expand_tables_kernel, dim3((unsigned)std::min(nsub, 8192)), dim3(256), 0, stream, subs, nsub, T, cmask, cbase,
                     coff, mask_pool, dest_pool);
  return hipGetLastError();
}

}  // namespace sls
