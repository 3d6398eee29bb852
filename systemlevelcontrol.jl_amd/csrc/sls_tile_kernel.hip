// sls_tile_kernel.hip — the ñx > 64 regime of the batched column-separable H2 SLS solve on FP64 matrix cores.
//
// Same mathematics as sls_kernels.hip (reference src/synthesis.jl:46-62 is the QP being solved): block
// tridiagonal Schur complement S = E H⁻¹ Eᵀ, block LDLᵀ with explicit inverse pivot blocks
//     P_k = (δI + Wx_k + B̃ Wu_{k−1} B̃ᵀ + Ã (W − W P_{k−1} W) Ãᵀ)⁻¹ ,   W = Wx_{k−1},
// method of multipliers on λ until ‖f − E z(λ)‖∞ ≤ tol.  What differs is how a block is held and inverted:
//
//   * a block is a symmetric grid of 16×16 tiles, UPPER tiles only (tile (I,J), I ≤ J, row-major);
//   * the inversion is the blocked symmetric SWEEP operator: for pivot tile q
//         M_qq = L Lᵀ ,  Y_i = M_iq L⁻ᵀ ,  G_i = Y_i L⁻¹ ,   M_ij ← M_ij − Y_i Y_jᵀ (i,j ≠ q),  M_iq ← G_i,  M_qq ← −L⁻ᵀL⁻¹
//     which after all pivots leaves M = −(D'_k)⁻¹ = −P_k; every trailing update is four v_mfma_f64_16x16x4_f64 on one
//     tile, the operand panel Yᵀ staged k-major in LDS so that every operand read is a conflict-free ds_read_b64
//     (lane l reads element 64·s + l of a panel tile for k-step s); Yᵀ_i = L⁻¹·C_iᵀ and Gᵀ_i = L⁻ᵀ·Yᵀ_i are MFMA groups
//     whose B operand is the previous result still in registers (C/D map of the f64 MFMA: row = (l>>4) + 4·reg,
//     col = l&15 — lane (g,c), register s holds element [4s+g][c], which is exactly B[k = 4s+g][c] of k-step s);
//   * only the stored half is ever updated, so the block stays exactly symmetric (no row-for-column stand-in);
//   * ñx ≤ ≈144: the block lives in LDS (MLDS = true); beyond, it is swept in place in its slot of the workgroup's
//     global workspace (L2 / Infinity-Cache resident), which removes the size limit of the other kernels
//     (the reference has none: src/synthesis.jl:46-62);
//   * Ã and B̃2 stay sparse (CSR + CSC gathered once per column from the shared operator); ñu is not limited by LDS
//     (no dense B̃2): random-sparse plants have ñu > ñx.
//   * the substitution sweeps read each stored tile once for both of its mirror images (half the P_k traffic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sls_device.h"

namespace sls {

namespace {

constexpr int TB = kTileThreads;
constexpr int NW = kTileWaves;
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int tbsearch(const int32_t* a, int n, int32_t key) {
  int lo = 0, hi = n - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int32_t v = a[mid];
    if (v == key) return mid;
    if (v < key) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

__device__ __forceinline__ double tblock_max(double v, double* red, int tid) {
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) r = fmax(r, red[w]);
  __syncthreads();
  return r;
}

// a[0] = 0, a[1..n] hold counts on entry and inclusive prefix sums on return; one wave
__device__ __forceinline__ void wave_prefix(int32_t* a, int n, int lane) {
  int carry = 0;
  for (int b = 0; b < n; b += 64) {
    const int i = b + lane;
    int v = (i < n) ? a[i + 1] : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(v, off); if (lane >= off) v += t; }
    if (i < n) a[i + 1] = v + carry;
    carry += __shfl(v, 63);
  }
  if (lane == 0) a[0] = 0;
}

__device__ __forceinline__ int tile_index(int i, int j, int NT) { return i * NT - ((i * (i - 1)) >> 1) + (j - i); }   // i ≤ j

// ---- tile access in the C/D register layout of v_mfma_f64_16x16x4_f64: lane (g,c) = (l>>4, l&15), reg r ↔ (row 4r+g, col c)
template <int RS>
__device__ __forceinline__ d4 tile_load(const double* tp, int g, int c) {
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = tp[(4 * r + g) * RS + c];
  return v;
}
template <int RS>
__device__ __forceinline__ d4 tile_load_t(const double* tp, int g, int c) {   // the transposed tile in the same layout
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = tp[c * RS + 4 * r + g];
  return v;
}
template <int RS>
__device__ __forceinline__ void tile_store(double* tp, int g, int c, d4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) tp[(4 * r + g) * RS + c] = v[r];
}
template <int RS>
__device__ __forceinline__ void tile_store_t(double* tp, int g, int c, d4 v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) tp[c * RS + 4 * r + g] = v[r];
}

// Cholesky of one 16×16 SPD tile by ONE wave, tile in the C/D layout: returns L⁻¹ (lower triangular) — the row operations
// of the factorisation applied to an identity tile carried alongside.  Pivots are clamped at `pivmin` (the block is
// δI + PSD, so a pivot below δ can only be rounding).
__device__ __forceinline__ d4 tile_chol_inverse(d4 t, double pivmin, int lane) {
  const int g = lane >> 4, c = lane & 15;
  d4 x;
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] = (4 * r + g == c) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int rk = k >> 2, gk = k & 3;
    const double rowk = __shfl(t[rk], gk * 16 + c);          // M[k][c]
    const double piv = fmax(__shfl(rowk, k), pivmin);        // M[k][k]
    double rs = __builtin_amdgcn_rsq(piv);
    rs = rs * fma(-0.5 * piv * rs, rs, 1.5);
    rs = rs * fma(-0.5 * piv * rs, rs, 1.5);
    const double lrow = rowk * rs;                           // L[c][k], c ≥ k
    const double xrow = __shfl(x[rk], gk * 16 + c) * rs;     // row k of L⁻¹
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int l = 4 * r + g;
      const double m = __shfl(t[r], g * 16 + k) * rs;        // L[l][k]
      if (l > k) { t[r] = fma(-m, lrow, t[r]); x[r] = fma(-m, xrow, x[r]); }
      if (l == k) { t[r] = lrow; x[r] = xrow; }
    }
  }
  return x;
}

// Blocked symmetric sweep of the NT×NT tile grid at Mb (tile (i,j), i ≤ j, at Mb + tile_index·TSZ, row stride RS):
// on return Mb holds −(M)⁻¹.  For pivot tile q with M_qq = L Lᵀ:
//     Y_i = M_iq L⁻ᵀ (bounded),  G_i = Y_i L⁻¹ = M_iq M_qq⁻¹,   M_ij ← M_ij − Y_i Y_jᵀ (i,j ≠ q),  M_iq ← G_i,  M_qq ← −L⁻ᵀL⁻¹.
// The trailing update goes through Y, never through M_qq⁻¹ itself: with a nearly singular pivot tile (dependent
// constraint rows: eigenvalue δ) the entries of M_qq⁻¹ are 1/δ while G_i C_jᵀ is O(1) — formed as (C_i·M_qq⁻¹)·C_jᵀ it loses
// eleven digits to cancellation, formed as Y_i·Y_jᵀ none.
// Yp: LDS panel of NT tiles (256 doubles each, Yᵀ_i k-major: every MFMA operand read is element 64·s + lane),
// Lb: 512 doubles (L⁻¹ row-major, then its transpose), tl: tile list.  All TB threads call it.
template <int RS, int TSZ>
__device__ __forceinline__ void tile_sweep(double* Mb, int NT, double* Yp, double* Lb, const int32_t* tl, int HT, double pivmin,
                                            int tid) {
  const int lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  double* const Li = Lb;            // L⁻¹[a][b]  at a·16 + b
  double* const LiT = Lb + 256;     // L⁻¹[b][a]  at a·16 + b
  for (int q = 0; q < NT; ++q) {
    // P0: Cholesky of the pivot tile by the last wave
    if (w == NW - 1) {
      d4 v = tile_load<RS>(Mb + (int64_t)tile_index(q, q, NT) * TSZ, g, c);
      v = tile_chol_inverse(v, pivmin, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) { Li[64 * r + lane] = v[r]; LiT[c * 16 + 4 * r + g] = v[r]; }
    }
    __syncthreads();
    // P2: Yᵀ_i = L⁻¹·C_iᵀ, Gᵀ_i = L⁻ᵀ·Yᵀ_i (operand B straight from the result registers of the load / the first product);
    //     panel Yᵀ; M_iq ← G_i; M_qq ← −L⁻ᵀL⁻¹
    {
      double li[4], lit[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { li[s] = Li[64 * s + lane]; lit[s] = LiT[64 * s + lane]; }
      for (int i = w; i < NT; i += NW) {
        if (i == q) {
          d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(li[s], li[s], acc, 0, 0, 0);
          tile_store<RS>(Mb + (int64_t)tile_index(q, q, NT) * TSZ, g, c, -acc);
          continue;
        }
        d4 x;
        if (i > q) x = tile_load<RS>(Mb + (int64_t)tile_index(q, i, NT) * TSZ, g, c);       // M_qi = C_iᵀ
        else       x = tile_load_t<RS>(Mb + (int64_t)tile_index(i, q, NT) * TSZ, g, c);     // (M_iq)ᵀ
        d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) y = __builtin_amdgcn_mfma_f64_16x16x4f64(lit[s], x[s], y, 0, 0, 0);
        d4 gt = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) gt = __builtin_amdgcn_mfma_f64_16x16x4f64(li[s], y[s], gt, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Yp[i * 256 + 64 * r + lane] = y[r];
        if (i > q) tile_store<RS>(Mb + (int64_t)tile_index(q, i, NT) * TSZ, g, c, gt);
        else       tile_store_t<RS>(Mb + (int64_t)tile_index(i, q, NT) * TSZ, g, c, gt);
      }
    }
    __syncthreads();
    // P3: trailing update of every stored tile off the pivot row/column
    for (int t = w; t < HT; t += NW) {
      const int ij = tl[t];
      const int i = ij & 0xffff, j = ij >> 16;
      if (i == q || j == q) continue;
      double* tp = Mb + (int64_t)t * TSZ;
      d4 acc = tile_load<RS>(tp, g, c);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Yp[i * 256 + 64 * s + lane], Yp[j * 256 + 64 * s + lane], acc, 0, 0, 0);
      tile_store<RS>(tp, g, c, acc);
    }
    __syncthreads();
  }
}

}  // namespace

template <bool MLDS>
__global__ __launch_bounds__(TB) void h2_column_tile_kernel(const KernelParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int RS = MLDS ? 17 : 16;
  constexpr int TSZ = MLDS ? kTileLdsTile : 256;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  const int T = p.T;
  const int nmax = p.nmax, mmax = p.mmax;
  const int NTmax = tile_nt(nmax), npadmax = 16 * NTmax, mpadmax = (mmax + 7) / 8 * 8 + 8;

  // ---- LDS carve (must match tile_kernel_lds_bytes) ----
  double* dp = reinterpret_cast<double*>(lds_raw);
  double* R0 = dp; dp += tile_kernel_r0_doubles(nmax, mmax);
  double* Mlds = dp; if (MLDS) dp += (int64_t)tile_ht(NTmax) * kTileLdsTile;
  double* csrA_v = dp; dp += p.nnzA_cap;
  double* cscA_v = dp; dp += p.nnzA_cap;
  double* csrB_v = dp; dp += p.nnzB_cap;
  double* cscB_v = dp; dp += p.nnzB_cap;
  double* wprev = dp; dp += npadmax;
  double* wcur = dp; dp += npadmax;
  double* red = dp; dp += 16;
  int32_t* ip = reinterpret_cast<int32_t*>(dp);
  int32_t* sx = ip; ip += npadmax;
  int32_t* csrA_p = ip; ip += npadmax + 1;
  int32_t* cscA_p = ip; ip += npadmax + 1;
  int32_t* csrA_i = ip; ip += p.nnzA_cap;
  int32_t* cscA_i = ip; ip += p.nnzA_cap;
  int32_t* csrB_p = ip; ip += npadmax + 1;
  int32_t* cscB_p = ip; ip += mpadmax + 1;
  int32_t* csrB_i = ip; ip += p.nnzB_cap;
  int32_t* cscB_i = ip; ip += p.nnzB_cap;
  int32_t* tl = ip; ip += tile_ht(NTmax);

  double* facws = p.fac_ws + (int64_t)blockIdx.x * p.fac_stride;
  double* vecs = p.vec_ws + (int64_t)blockIdx.x * p.vec_stride;

  for (int it_sub = blockIdx.x; it_sub < p.nsub; it_sub += gridDim.x) {
    const SubDesc sd = p.subs[p.order[p.order_off + it_sub]];
    const int n = sd.n, m = sd.m, nm = n + m;
    const int NT = tile_nt(n), HT = tile_ht(NT), npad = 16 * NT;
    double* lam = vecs;
    double* qv = vecs + (int64_t)(T + 1) * n;
    double* rv = vecs + 2LL * (T + 1) * n;
    const uint8_t* mask = p.mask_pool + sd.off_mask;
    const int32_t* dest = p.dest_pool + sd.off_dest;
    const int32_t* su = p.idx_pool + sd.off_su;
    double* Pfull = facws + (int64_t)(T + 1) * HT * 256;          // row-major npad×npad copy of the latest −P_k
    const bool has_w = sd.has_w != 0;
    auto hx = [&](int i) -> double { return has_w ? p.w_pool[sd.off_w + i] : 1.0; };
    auto hu = [&](int j) -> double { return has_w ? p.w_pool[sd.off_w + n + j] : 1.0; };
    auto gx = [&](int i) -> double { return has_w ? p.w_pool[sd.off_w + nm + i] : 0.0; };
    auto gu = [&](int j) -> double { return has_w ? p.w_pool[sd.off_w + nm + n + j] : 0.0; };

    __syncthreads();   // previous subproblem fully done with LDS
    unsigned long long tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // SLS_PHASE_TIMERS: setup, residual, build, sweep, store, substitution
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    auto lap = [&](int slot) { if (p.dbg) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tc[slot] += now - tlast; tlast = now; } };

    // ---- index set, tile list ----
    for (int i = tid; i < n; i += TB) sx[i] = p.idx_pool[sd.off_sx + i];
    for (int t = tid; t < HT; t += TB) {
      int i = 0, r = t;
      while (r >= NT - i) { r -= NT - i; ++i; }
      tl[t] = i | ((i + r) << 16);
    }
    __syncthreads();

    // ---- gather Ã (CSR + CSC) and B̃2 (CSR + CSC) from the shared operator: counts, prefix sums, fill ----
    for (int i = tid; i < n; i += TB) {
      const int gi = sx[i];
      int c1 = 0, c2 = 0, c3 = 0;
      for (int e = p.A_rowptr[gi]; e < p.A_rowptr[gi + 1]; ++e)
        if (p.A_val[e] != 0.0 && tbsearch(sx, n, p.A_colidx[e]) >= 0) ++c1;
      for (int e = p.At_rowptr[gi]; e < p.At_rowptr[gi + 1]; ++e)
        if (p.At_val[e] != 0.0 && tbsearch(sx, n, p.At_colidx[e]) >= 0) ++c2;
      for (int e = p.B_rowptr[gi]; e < p.B_rowptr[gi + 1]; ++e)
        if (p.B_val[e] != 0.0 && tbsearch(su, m, p.B_colidx[e]) >= 0) ++c3;
      csrA_p[i + 1] = c1; cscA_p[i + 1] = c2; csrB_p[i + 1] = c3;
    }
    for (int j = tid; j < m; j += TB) {
      const int gj = su[j];
      int c4 = 0;
      for (int e = p.Bt_rowptr[gj]; e < p.Bt_rowptr[gj + 1]; ++e)
        if (p.Bt_val[e] != 0.0 && tbsearch(sx, n, p.Bt_colidx[e]) >= 0) ++c4;
      cscB_p[j + 1] = c4;
    }
    __syncthreads();
    if (w == 0) wave_prefix(csrA_p, n, lane);
    else if (w == 1) wave_prefix(cscA_p, n, lane);
    else if (w == 2) wave_prefix(csrB_p, n, lane);
    else if (w == 3) wave_prefix(cscB_p, m, lane);
    __syncthreads();
    for (int i = tid; i < n; i += TB) {
      const int gi = sx[i];
      int w1 = csrA_p[i], w2 = cscA_p[i], w3 = csrB_p[i];
      for (int e = p.A_rowptr[gi]; e < p.A_rowptr[gi + 1]; ++e) {
        const double v = p.A_val[e];
        const int loc = (v != 0.0) ? tbsearch(sx, n, p.A_colidx[e]) : -1;
        if (loc >= 0) { csrA_i[w1] = loc; csrA_v[w1] = v; ++w1; }
      }
      for (int e = p.At_rowptr[gi]; e < p.At_rowptr[gi + 1]; ++e) {
        const double v = p.At_val[e];
        const int loc = (v != 0.0) ? tbsearch(sx, n, p.At_colidx[e]) : -1;
        if (loc >= 0) { cscA_i[w2] = loc; cscA_v[w2] = v; ++w2; }
      }
      for (int e = p.B_rowptr[gi]; e < p.B_rowptr[gi + 1]; ++e) {
        const double v = p.B_val[e];
        const int loc = (v != 0.0) ? tbsearch(su, m, p.B_colidx[e]) : -1;
        if (loc >= 0) { csrB_i[w3] = loc; csrB_v[w3] = v; ++w3; }
      }
    }
    for (int j = tid; j < m; j += TB) {
      const int gj = su[j];
      int w4 = cscB_p[j];
      for (int e = p.Bt_rowptr[gj]; e < p.Bt_rowptr[gj + 1]; ++e) {
        const double v = p.Bt_val[e];
        const int loc = (v != 0.0) ? tbsearch(sx, n, p.Bt_colidx[e]) : -1;
        if (loc >= 0) { cscB_i[w4] = loc; cscB_v[w4] = v; ++w4; }
      }
    }
    __syncthreads();

    // ---- regularisation scale: largest possible Schur diagonal ----
    double sc = 0.0;
    for (int i = tid; i < n; i += TB) {
      double s = hx(i);
      for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) s = fma(csrA_v[e] * csrA_v[e], hx(csrA_i[e]), s);
      for (int e = csrB_p[i]; e < csrB_p[i + 1]; ++e) s = fma(csrB_v[e] * csrB_v[e], hu(csrB_i[e]), s);
      sc = fmax(sc, s);
    }
    sc = tblock_max(sc, red, tid);
    const double delta = p.delta_rel * sc;

    for (int i = tid; i < (T + 1) * n; i += TB) lam[i] = 0.0;
    __syncthreads();

    // residual pass: r = f − E z(λ); writes z to the output; returns ‖r‖∞  (one barrier per time step; λ slices, the row
    // carried to the next step and x_t/u_t are staged in the phase-shared LDS region)
    auto residual_pass = [&]() -> double {
      double rmax = 0.0;
      double* const stage = R0;                                       // [3][npad]
      double* const carry0 = stage + 3 * npad;                        // [2][npad]
      double* const xu0 = stage + 5 * npad;                           // [2][npad + mpad]
      const int xus = npad + mpadmax;
      for (int i = tid; i < n; i += TB) {
        carry0[i] = (i == sd.pos) ? 1.0 : 0.0;                        // f_0 = e_pos
        stage[i] = lam[i];
        stage[npad + i] = lam[(int64_t)n + i];
      }
      __syncthreads();
      for (int t = 0; t < T; ++t) {
        const double* l0 = stage + (t % 3) * npad;
        const double* l1 = stage + ((t + 1) % 3) * npad;
        double* l2 = stage + ((t + 2) % 3) * npad;
        double* xt_ = xu0 + (t & 1) * xus;
        double* ut_ = xt_ + npad;
        if (t + 2 <= T) for (int i = tid; i < n; i += TB) l2[i] = lam[(int64_t)(t + 2) * n + i];
        const uint8_t* mk = mask + (int64_t)t * nm;
        const int32_t* ds = dest + (int64_t)t * nm;
        for (int q = tid; q < nm; q += TB) {
          if (q < n) {
            double v = 0.0;
            if (mk[q]) {
              double acc = 0.0;
              for (int e = cscA_p[q]; e < cscA_p[q + 1]; ++e) acc = fma(cscA_v[e], l1[cscA_i[e]], acc);
              v = hx(q) * (l0[q] - acc - gx(q));
              const int d = ds[q]; if (d >= 0) p.out[d] = v;
            }
            xt_[q] = v;
          } else {
            const int j = q - n;
            double v = 0.0;
            if (mk[q]) {
              double acc = 0.0;
              for (int e = cscB_p[j]; e < cscB_p[j + 1]; ++e) acc = fma(cscB_v[e], l1[cscB_i[e]], acc);
              v = hu(j) * (-acc - gu(j));
              const int d = ds[q]; if (d >= 0) p.out[d] = v;
            }
            ut_[j] = v;
          }
        }
        __syncthreads();
        const double* cin = carry0 + (t & 1) * npad;
        double* cout = carry0 + ((t + 1) & 1) * npad;
        for (int i = tid; i < n; i += TB) {
          const double r = cin[i] - xt_[i];
          rv[(int64_t)t * n + i] = r;
          rmax = fmax(rmax, fabs(r));
          double acc = 0.0;
          for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) acc = fma(csrA_v[e], xt_[csrA_i[e]], acc);
          for (int e = csrB_p[i]; e < csrB_p[i + 1]; ++e) acc = fma(csrB_v[e], ut_[csrB_i[e]], acc);
          cout[i] = acc;
        }
      }
      __syncthreads();
      const double* cfin = carry0 + (T & 1) * npad;
      for (int i = tid; i < n; i += TB) {
        rv[(int64_t)T * n + i] = cfin[i];
        rmax = fmax(rmax, fabs(cfin[i]));
      }
      return tblock_max(rmax, red, tid);
    };

    lap(0);
    double resid;
    if (!has_w && sd.pos >= 0) {                 // g = 0 and λ = 0: z = 0 and r = f = e_pos exactly
      for (int i = tid; i < (T + 1) * n; i += TB) rv[i] = (i == sd.pos) ? 1.0 : 0.0;
      __syncthreads();
      resid = 1.0;
    } else {
      resid = residual_pass();
    }
    lap(1);
    int iters = 0;
    int status = 0;

    if (resid > p.tol) {
      // =================== factor: −P_k = sweep(D'_k) ===================
      double* const Yp = R0;
      double* const Lb = R0 + (int64_t)NT * 256;
      double* const strip = R0;                         // [16][npad + 1], build phase only
      const int sst = npad + 1;
      for (int k = 0; k <= T; ++k) {
        double* const slot = facws + (int64_t)k * HT * 256;
        double* const Mb = MLDS ? Mlds : slot;
        // weights of this block row (zero padded)
        {
          const uint8_t* mkp = mask + (int64_t)(k - 1) * nm;
          const uint8_t* mkc = mask + (int64_t)k * nm;
          for (int i = tid; i < npad; i += TB) {
            wprev[i] = (k >= 1 && i < n && mkp[i]) ? hx(i) : 0.0;
            wcur[i] = (k <= T - 1 && i < n && mkc[i]) ? hx(i) : 0.0;
          }
        }
        __syncthreads();
        if (k >= 1) {
          for (int I = 0; I < NT; ++I) {
            // step 1: strip = rows 16I..16I+15 of Ã·Q,  Q = W + W N W  (N = −P_{k−1}, full row-major copy)
            for (int a = w; a < 16; a += NW) {
              const int i = 16 * I + a;
              for (int cb = 0; cb < npad; cb += 64) {
                const int cc = cb + lane;
                double acc = 0.0;
                if (i < n && cc < npad) {
                  const double wc = wprev[cc];
                  for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) {
                    const int q = csrA_i[e];
                    const double vw = csrA_v[e] * wprev[q];
                    if (vw != 0.0) {
                      const double nv = Pfull[(int64_t)q * npad + cc];
                      acc = fma(vw, ((q == cc) ? 1.0 : 0.0) + nv * wc, acc);
                    }
                  }
                }
                if (cc < npad) strip[a * sst + cc] = acc;
              }
            }
            __syncthreads();
            // step 2: tiles (I, J ≥ I) of D' = δI + Wx_k + (Ã·Q)·Ãᵀ; unit = (J, register row group r)
            const int nunits = (NT - I) * 4;
            for (int u = w; u < nunits; u += NW) {
              const int J = I + (u >> 2), r = u & 3;
              const int irow = 4 * r + g, i = 16 * I + irow, j = 16 * J + c;
              double val = (i == j) ? ((j < n) ? delta + wcur[j] : 1.0) : 0.0;
              if (i < n && j < n)
                for (int e = csrA_p[j]; e < csrA_p[j + 1]; ++e) val = fma(strip[irow * sst + csrA_i[e]], csrA_v[e], val);
              Mb[(int64_t)tile_index(I, J, NT) * TSZ + irow * RS + c] = val;
            }
            __syncthreads();
          }
          // B̃ Wu B̃ᵀ: row-owned read-modify-write of the stored half (both orders inside a diagonal tile)
          const uint8_t* mku = mask + (int64_t)(k - 1) * nm + n;
          for (int i = tid; i < n; i += TB) {
            for (int e = csrB_p[i]; e < csrB_p[i + 1]; ++e) {
              const int q = csrB_i[e];
              if (!mku[q]) continue;
              const double vi = csrB_v[e] * hu(q);
              for (int e2 = cscB_p[q]; e2 < cscB_p[q + 1]; ++e2) {
                const int j = cscB_i[e2];
                if ((i >> 4) <= (j >> 4)) {
                  double* el = Mb + (int64_t)tile_index(i >> 4, j >> 4, NT) * TSZ + (i & 15) * RS + (j & 15);
                  *el = fma(vi, cscB_v[e2], *el);
                }
              }
            }
          }
        } else {
          for (int t = w; t < HT; t += NW) {
            const int ij = tl[t];
            const int I = ij & 0xffff, J = ij >> 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * I + 4 * r + g, j = 16 * J + c;
              Mb[(int64_t)t * TSZ + (4 * r + g) * RS + c] = (i == j) ? ((j < n) ? delta + wcur[j] : 1.0) : 0.0;
            }
          }
        }
        __syncthreads();
        lap(2);
        tile_sweep<RS, TSZ>(Mb, NT, Yp, Lb, tl, HT, 0.25 * delta, tid);
        lap(3);
        // write-out: the slot (LDS-resident block only) and the full row-major copy the next build gathers rows from
        for (int t = w; t < HT; t += NW) {
          const int ij = tl[t];
          const int I = ij & 0xffff, J = ij >> 16;
          const d4 x = tile_load<RS>(Mb + (int64_t)t * TSZ, g, c);
          if (MLDS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slot[(int64_t)t * 256 + 64 * r + lane] = x[r];
          }
          if (k < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              Pfull[(int64_t)(16 * I + 4 * r + g) * npad + 16 * J + c] = x[r];
              if (I != J) Pfull[(int64_t)(16 * J + c) * npad + 16 * I + 4 * r + g] = x[r];
            }
          }
        }
        __syncthreads();
        lap(4);
      }

      // out = P_k y = −(N_k y), N_k symmetric tiles of slot k; y, out: LDS vectors of npad entries (y zero padded).
      // Wave w owns tile rows s and NT−1−s (balanced); every stored tile is read once and serves both mirror images.
      double* const pv = R0;                         // [NW][npad]
      double* const yv = R0 + (int64_t)NW * npad;
      double* const ov = yv + npad;
      auto sym_matvec = [&](const double* Ns) {
        for (int i = tid; i < NW * npad; i += TB) pv[i] = 0.0;
        __syncthreads();
        double* mypv = pv + w * npad;
        for (int s = w; s < (NT + 1) / 2; s += NW) {
#pragma unroll 1
          for (int half = 0; half < 2; ++half) {
            const int i = half ? NT - 1 - s : s;
            if (half && i == s) break;
            double yI[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) yI[r] = yv[16 * i + 4 * r + g];
            double pr[4] = {0.0, 0.0, 0.0, 0.0};
            const double* tp = Ns + (int64_t)tile_index(i, i, NT) * 256;
            for (int j = i; j < NT; ++j, tp += 256) {
              double x[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) x[r] = tp[64 * r + lane];
              const double yJ = yv[16 * j + c];
#pragma unroll
              for (int r = 0; r < 4; ++r) pr[r] = fma(x[r], yJ, pr[r]);
              if (j != i) {
                double pc = x[0] * yI[0];
#pragma unroll
                for (int r = 1; r < 4; ++r) pc = fma(x[r], yI[r], pc);
                pc += __shfl_xor(pc, 16);
                pc += __shfl_xor(pc, 32);
                if (g == 0) mypv[16 * j + c] += pc;
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              double v = pr[r];
              v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
              if (c == 0) mypv[16 * i + 4 * r + g] += v;
            }
          }
        }
        __syncthreads();
        for (int i = tid; i < npad; i += TB) {
          double s = 0.0;
#pragma unroll
          for (int ww = 0; ww < NW; ++ww) s += pv[ww * npad + i];
          ov[i] = -s;
        }
        __syncthreads();
      };

      // =================== refinement loop ===================
      double prev = resid;
      for (int it = 1; it <= p.max_iters; ++it) {
        iters = it;
        // forward: y_k = r_k + Ã(Wx_{k−1} q_{k−1});  q_k = P_k y_k
        for (int k = 0; k <= T; ++k) {
          const uint8_t* mk = mask + (int64_t)(k - 1) * nm;
          const double* qp = qv + (int64_t)(k - 1) * n;
          for (int i = tid; i < npad; i += TB) {
            double acc = 0.0;
            if (i < n) {
              acc = rv[(int64_t)k * n + i];
              if (k >= 1)
                for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) {
                  const int q = csrA_i[e];
                  if (mk[q]) acc = fma(csrA_v[e] * hx(q), qp[q], acc);
                }
            }
            yv[i] = acc;
          }
          __syncthreads();
          sym_matvec(facws + (int64_t)k * HT * 256);
          for (int i = tid; i < n; i += TB) qv[(int64_t)k * n + i] = ov[i];
          __syncthreads();
        }
        // backward: Δλ_k = q_k + P_k (Wx_k (Ãᵀ Δλ_{k+1}));  λ += Δλ.   Δλ_k overwrites q_k.
        for (int k = T; k >= 0; --k) {
          if (k < T) {
            const uint8_t* mk = mask + (int64_t)k * nm;
            const double* dl1 = qv + (int64_t)(k + 1) * n;
            for (int q = tid; q < npad; q += TB) {
              double acc = 0.0;
              if (q < n && mk[q]) {
                for (int e = cscA_p[q]; e < cscA_p[q + 1]; ++e) acc = fma(cscA_v[e], dl1[cscA_i[e]], acc);
                acc *= hx(q);
              }
              yv[q] = acc;
            }
            __syncthreads();
            sym_matvec(facws + (int64_t)k * HT * 256);
            for (int i = tid; i < n; i += TB) qv[(int64_t)k * n + i] += ov[i];
            __syncthreads();
          }
          for (int i = tid; i < n; i += TB) lam[(int64_t)k * n + i] += qv[(int64_t)k * n + i];
        }
        __syncthreads();
        lap(5);
        resid = residual_pass();
        lap(1);
        if (resid <= p.tol) break;
        if (it >= 2 && resid > 0.5 * prev) { status = 1; break; }   // stagnation ⇒ inconsistent system
        prev = resid;
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

// ---- test kernel: invert one dense SPD matrix with the tile sweep (tests/test_gpu_tile.py, through sls_debug_tile_invert) ----
template <bool MLDS>
__global__ __launch_bounds__(TB) void tile_invert_kernel(const double* __restrict__ A, int n, double* ws, double* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int RS = MLDS ? 17 : 16;
  constexpr int TSZ = MLDS ? kTileLdsTile : 256;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  const int NT = tile_nt(n), HT = tile_ht(NT);
  double* Yp = reinterpret_cast<double*>(lds_raw);
  double* Lb = Yp + (int64_t)NT * 256;
  double* Mlds = Lb + 512;
  int32_t* tl = reinterpret_cast<int32_t*>(Mlds + (MLDS ? (int64_t)HT * kTileLdsTile : 0));
  double* Mb = MLDS ? Mlds : ws;
  for (int t = tid; t < HT; t += TB) {
    int i = 0, r = t;
    while (r >= NT - i) { r -= NT - i; ++i; }
    tl[t] = i | ((i + r) << 16);
  }
  __syncthreads();
  for (int t = w; t < HT; t += NW) {
    const int I = tl[t] & 0xffff, J = tl[t] >> 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * I + 4 * r + g, j = 16 * J + c;
      Mb[(int64_t)t * TSZ + (4 * r + g) * RS + c] = (i < n && j < n) ? A[(int64_t)i * n + j] : (i == j ? 1.0 : 0.0);
    }
  }
  __syncthreads();
  tile_sweep<RS, TSZ>(Mb, NT, Yp, Lb, tl, HT, 0.0, tid);
  for (int t = w; t < HT; t += NW) {
    const int I = tl[t] & 0xffff, J = tl[t] >> 16;
    const d4 x = tile_load<RS>(Mb + (int64_t)t * TSZ, g, c);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 16 * I + 4 * r + g, j = 16 * J + c;
      if (i < n && j < n) { out[(int64_t)i * n + j] = -x[r]; out[(int64_t)j * n + i] = -x[r]; }
    }
  }
}

}  // namespace sls

// ---- launchers (called from sls_api.cpp through plain C++ declarations) ----
namespace sls {

hipError_t launch_tile(const KernelParams& p, int grid, size_t lds_bytes, hipStream_t stream, bool mlds) {
  hipError_t e;
  if (mlds) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&h2_column_tile_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((h2_column_tile_kernel<true>), dim3(grid), dim3(TB), lds_bytes, stream, p);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&h2_column_tile_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((h2_column_tile_kernel<false>), dim3(grid), dim3(TB), lds_bytes, stream, p);
  }
  return hipGetLastError();
}

// d_A, d_out: device n×n row-major; d_ws: device scratch of tile_ht(nt)·256 doubles (global variant)
hipError_t launch_tile_invert(const double* d_A, int n, double* d_ws, double* d_out, bool mlds, hipStream_t stream) {
  const int nt = tile_nt(n), ht = tile_ht(nt);
  const size_t lds = (size_t)(nt * 256 + 512 + (mlds ? ht * kTileLdsTile : 0)) * 8 + (size_t)ht * 4 + 64;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e;
  if (mlds) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_invert_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((tile_invert_kernel<true>), dim3(1), dim3(TB), lds, stream, d_A, n, d_ws, d_out);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tile_invert_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((tile_invert_kernel<false>), dim3(1), dim3(TB), lds, stream, d_A, n, d_ws, d_out);
  }
  return hipGetLastError();
}

}  // namespace sls
