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

// Workgroup barrier for data that travels through LDS only: waits for this wave's LDS operations, not for its global ones.
// __syncthreads() is a workgroup-scope release/acquire fence + s_barrier, i.e. s_waitcnt vmcnt(0) lgkmcnt(0): every barrier
// of a substitution step then also waited for the P_k tiles just prefetched, the r_k load and the q_k store of the step
// (none of which another thread of the workgroup reads before the next full barrier).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// deterministic block sum (fixed tree: lanes by xor-shuffle, waves in order)
__device__ __forceinline__ double tblock_sum(double v, double* red, int tid) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) r += red[w];
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

// broadcast of lane K of every 16-lane row to the whole row (DPP row_newbcast: two VALU moves, no LDS crossbar)
template <int K>
__device__ __forceinline__ double row_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int SRC>
__device__ __forceinline__ double lane_bcast(double v) {       // value of lane SRC in every lane (through SGPRs)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), SRC), hi = __builtin_amdgcn_readlane(__double2hiint(v), SRC);
  return __hiloint2double(hi, lo);
}

// One elimination step of the Cholesky factorisation of a 16×16 SPD tile held by ONE wave in the C/D layout, with the
// same row operations applied to x (starts as I, ends as L⁻¹).  Column K travels by DPP, row K by one ds_bpermute pair,
// the pivot through SGPRs.
template <int K>
__device__ __forceinline__ void chol_step(d4& t, d4& x, double pivmin, int g, int c) {
  constexpr int rk = K >> 2, gk = K & 3;
  const double rowk = __shfl(t[rk], gk * 16 + c);            // M[K][c]
  const double piv = fmax(lane_bcast<gk * 16 + K>(t[rk]), pivmin);
  double rs = __builtin_amdgcn_rsq(piv);
  rs = rs * fma(-0.5 * piv * rs, rs, 1.5);
  rs = rs * fma(-0.5 * piv * rs, rs, 1.5);
  const double lrow = rowk * rs;                             // L[c][K], c ≥ K
  const double xrow = __shfl(x[rk], gk * 16 + c) * rs;       // row K of L⁻¹
  // rows l = 4r + g: registers r < rk are finished, r > rk are all below the pivot row, r = rk holds the pivot row itself
  // (g = gk), rows below it (g > gk) and finished rows (g < gk) — everything but the last distinction is static
#pragma unroll
  for (int r = rk; r < 4; ++r) {
    const double m = row_bcast<K>(t[r]) * rs;                // L[l][K]
    const double tn = fma(-m, lrow, t[r]), xn = fma(-m, xrow, x[r]);
    if (r > rk) { t[r] = tn; x[r] = xn; }
    else {
      t[r] = (g > gk) ? tn : ((g == gk) ? lrow : t[r]);
      x[r] = (g > gk) ? xn : ((g == gk) ? xrow : x[r]);
    }
  }
}

// Cholesky of one 16×16 SPD tile by ONE wave, tile in the C/D layout: returns L⁻¹ (lower triangular) — the row operations
// of the factorisation applied to an identity tile carried alongside.  Pivots are clamped at `pivmin` (the block is
// δI + PSD, so a pivot below δ can only be rounding).
__device__ __forceinline__ d4 tile_chol_inverse(d4 t, double pivmin, int lane) {
  const int g = lane >> 4, c = lane & 15;
  d4 x;
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] = (4 * r + g == c) ? 1.0 : 0.0;
  chol_step<0>(t, x, pivmin, g, c);  chol_step<1>(t, x, pivmin, g, c);  chol_step<2>(t, x, pivmin, g, c);  chol_step<3>(t, x, pivmin, g, c);
  chol_step<4>(t, x, pivmin, g, c);  chol_step<5>(t, x, pivmin, g, c);  chol_step<6>(t, x, pivmin, g, c);  chol_step<7>(t, x, pivmin, g, c);
  chol_step<8>(t, x, pivmin, g, c);  chol_step<9>(t, x, pivmin, g, c);  chol_step<10>(t, x, pivmin, g, c); chol_step<11>(t, x, pivmin, g, c);
  chol_step<12>(t, x, pivmin, g, c); chol_step<13>(t, x, pivmin, g, c); chol_step<14>(t, x, pivmin, g, c); chol_step<15>(t, x, pivmin, g, c);
  return x;
}

// Blocked symmetric sweep of the NT×NT tile grid at Mb (tile (i,j), i ≤ j, at Mb + tile_index·TSZ, row stride RS):
// on return Mb holds −(M)⁻¹.  For pivot tile q with M_qq = L Lᵀ:
//     Y_i = M_iq L⁻ᵀ (bounded),  G_i = Y_i L⁻¹ = M_iq M_qq⁻¹,   M_ij ← M_ij − Y_i Y_jᵀ (i,j ≠ q),  M_iq ← G_i,  M_qq ← −L⁻ᵀL⁻¹.
// The trailing update goes through Y, never through M_qq⁻¹ itself: with a nearly singular pivot tile (dependent
// constraint rows: eigenvalue δ) the entries of M_qq⁻¹ are 1/δ while G_i C_jᵀ is O(1) — formed as (C_i·M_qq⁻¹)·C_jᵀ it loses
// eleven digits to cancellation, formed as Y_i·Y_jᵀ none.
// Yp: LDS panel(s) of NT tiles (256 doubles each, Yᵀ_i k-major: every MFMA operand read is element 64·s + lane),
// Lb: 512 doubles (L⁻¹ row-major, then its transpose), tl: tile list.  All TB threads call it.
// TWO (block in the workspace): pivot tiles are taken in pairs.  After the panel of the first, only the tile row/column of
// the second is brought up to date (NT tiles); its panel follows, and ONE pass over the stored half applies both rank-16
// updates (eight MFMAs per tile load + store instead of four): the sweep is bound by that traffic — 258 GB per
// random10000_d2 launch at 3.7 TB/s with single panels (profiles/r02_rocprof_random10000.txt).
template <int RS, int TSZ, bool TWO>
__device__ __forceinline__ void tile_sweep(double* Mb, int NT, double* Yp, double* Lb, const int32_t* tl, int HT, double pivmin,
                                            int tid, unsigned long long* sub = nullptr) {
  const int lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  double* const Li = Lb;            // L⁻¹[a][b]  at a·16 + b
  double* const LiT = Lb + 256;     // L⁻¹[b][a]  at a·16 + b
  const d4 z4 = {0.0, 0.0, 0.0, 0.0};
  // panel of pivot tile q into Y (and the swept column/row q into M): P0 (Cholesky by the last wave) + P2
  // have_l: L⁻¹ of this pivot tile is already in Lb — factored by the last wave during the previous trailing update (look-ahead)
  auto panel = [&](int q, double* Y, bool have_l) {
    unsigned long long ts0 = sub ? __builtin_amdgcn_s_memtime() : 0;
    if (!have_l) {
      if (w == NW - 1) {
        d4 v = tile_load<RS>(Mb + (int64_t)tile_index(q, q, NT) * TSZ, g, c);
        v = tile_chol_inverse(v, pivmin, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) { Li[64 * r + lane] = v[r]; LiT[c * 16 + 4 * r + g] = v[r]; }
      }
      __syncthreads();
    }
    if (sub) { const unsigned long long now = __builtin_amdgcn_s_memtime(); sub[0] += now - ts0; ts0 = now; }
    // Yᵀ_i = L⁻¹·C_iᵀ, Gᵀ_i = L⁻ᵀ·Yᵀ_i (operand B straight from the result registers of the load / the first product)
    double li[4], lit[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { li[s] = Li[64 * s + lane]; lit[s] = LiT[64 * s + lane]; }
    for (int i = w; i < NT; i += NW) {
      if (i == q) {
        d4 acc = z4;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(li[s], li[s], acc, 0, 0, 0);
        tile_store<RS>(Mb + (int64_t)tile_index(q, q, NT) * TSZ, g, c, -acc);
        continue;
      }
      d4 x;
      if (i > q) x = tile_load<RS>(Mb + (int64_t)tile_index(q, i, NT) * TSZ, g, c);       // M_qi = C_iᵀ
      else       x = tile_load_t<RS>(Mb + (int64_t)tile_index(i, q, NT) * TSZ, g, c);     // (M_iq)ᵀ
      // two independent accumulators per product: a dependent f64 MFMA waits ≈ 200 cycles for its predecessor
      d4 ya = __builtin_amdgcn_mfma_f64_16x16x4f64(lit[0], x[0], z4, 0, 0, 0);
      d4 yb = __builtin_amdgcn_mfma_f64_16x16x4f64(lit[1], x[1], z4, 0, 0, 0);
      ya = __builtin_amdgcn_mfma_f64_16x16x4f64(lit[2], x[2], ya, 0, 0, 0);
      yb = __builtin_amdgcn_mfma_f64_16x16x4f64(lit[3], x[3], yb, 0, 0, 0);
      const d4 y = ya + yb;
      d4 ga = __builtin_amdgcn_mfma_f64_16x16x4f64(li[0], y[0], z4, 0, 0, 0);
      d4 gb = __builtin_amdgcn_mfma_f64_16x16x4f64(li[1], y[1], z4, 0, 0, 0);
      ga = __builtin_amdgcn_mfma_f64_16x16x4f64(li[2], y[2], ga, 0, 0, 0);
      gb = __builtin_amdgcn_mfma_f64_16x16x4f64(li[3], y[3], gb, 0, 0, 0);
      const d4 gt = ga + gb;
#pragma unroll
      for (int r = 0; r < 4; ++r) Y[i * 256 + 64 * r + lane] = y[r];
      if (i > q) tile_store<RS>(Mb + (int64_t)tile_index(q, i, NT) * TSZ, g, c, gt);
      else       tile_store_t<RS>(Mb + (int64_t)tile_index(i, q, NT) * TSZ, g, c, gt);
    }
    __syncthreads();
    if (sub) { const unsigned long long now = __builtin_amdgcn_s_memtime(); sub[1] += now - ts0; }
  };
  // acc −= Y_i Y_jᵀ, two accumulators
  auto rank16 = [&](const double* Y, int i, int j, d4& a0, d4& a1) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Y[i * 256 + lane], Y[j * 256 + lane], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Y[i * 256 + 64 + lane], Y[j * 256 + 64 + lane], a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Y[i * 256 + 128 + lane], Y[j * 256 + 128 + lane], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Y[i * 256 + 192 + lane], Y[j * 256 + 192 + lane], a1, 0, 0, 0);
  };
  double* const Y0 = Yp;
  double* const Y1 = Yp + (int64_t)NT * 256;
  // LOOK-AHEAD (round 3): the Cholesky factorisation of a pivot tile is a 16-step dependent chain on ONE wave (≈ 4 k cycles)
  // with the other waves at a barrier — 44 % of the inversion on a grid-32 column.  The next pivot tile is final as soon as it
  // has taken this step's rank-16 update(s): the last wave updates it first, factors it from registers and leaves L⁻¹ in Lb
  // (nobody reads Lb during the trailing update: the panel keeps its copy in registers) while the other waves share the rest of
  // the trailing update; the next panel then starts without its serial prologue.
#ifndef SLS_TILE_LOOKAHEAD
#define SLS_TILE_LOOKAHEAD 1
#endif
  bool have_l = false;
  for (int q0 = 0; q0 < NT; q0 += (TWO ? 2 : 1)) {
    const int q1 = q0 + 1;
    const bool two = TWO && q1 < NT;
    panel(q0, Y0, have_l);
    have_l = false;
    if (two) {
      // bring the tile row/column of q1 up to date with the first panel (tiles with the other index ≠ q0), then its panel
      for (int o = w; o < NT; o += NW) {
        if (o == q0) continue;
        const int i = min(o, q1), j = max(o, q1);
        double* tp = Mb + (int64_t)tile_index(i, j, NT) * TSZ;
        d4 a0 = tile_load<RS>(tp, g, c), a1 = z4;
        rank16(Y0, i, j, a0, a1);
        tile_store<RS>(tp, g, c, a0 + a1);
      }
      __syncthreads();
      panel(q1, Y1, false);
    }
    // trailing update of every stored tile: the q0 term off row/column q0, the q1 term off row/column q1 (row/column q1 got
    // its q0 term above and was then replaced by the second panel; row/column q0 only takes the q1 term)
    const int qn = q0 + (two ? 2 : 1);                    // the next pivot tile
    const bool la = SLS_TILE_LOOKAHEAD != 0 && qn < NT && NW > 1;
    const int tn = la ? tile_index(qn, qn, NT) : -1;
    if (la && w == NW - 1) {
      // (qn, qn) takes both terms (qn ≠ q0, q1); it is not stored: the next panel replaces it by −L⁻ᵀL⁻¹ without reading it
      d4 a0 = tile_load<RS>(Mb + (int64_t)tn * TSZ, g, c), a1 = z4;
      rank16(Y0, qn, qn, a0, a1);
      if (two) rank16(Y1, qn, qn, a0, a1);
      const d4 v = tile_chol_inverse(a0 + a1, pivmin, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) { Li[64 * r + lane] = v[r]; LiT[c * 16 + 4 * r + g] = v[r]; }
    } else {
      const int nwk = la ? NW - 1 : NW;                     // waves sharing the rest
      for (int t = w; t < HT; t += nwk) {
        if (t == tn) continue;
        const int ij = tl[t];
        const int i = ij & 0xffff, j = ij >> 16;
        const bool u0 = i != q0 && j != q0 && !(two && (i == q1 || j == q1));
        const bool u1 = two && i != q1 && j != q1;
        if (!u0 && !u1) continue;
        double* tp = Mb + (int64_t)t * TSZ;
        d4 a0 = tile_load<RS>(tp, g, c), a1 = z4;
        if (u0) rank16(Y0, i, j, a0, a1);
        if (u1) rank16(Y1, i, j, a0, a1);
        tile_store<RS>(tp, g, c, a0 + a1);
      }
    }
    __syncthreads();
    have_l = la;
  }
}

}  // namespace

// GW: the build that carries the projected-CG loop for dense cost Hessians (has_w = 2 records); the plain build leaves it out
// (with it the multiplier iteration is inlined at four call sites and the 128-VGPR variant spills 2.5 KB per lane).
// BIG (round 3): the index set is too large for LDS (the two pivot panels alone are 32·ñx doubles; ñx ≳ 250 with typical
// sparse lists).  The SAME carve — panels, Ã·Q strip, staging vectors, sparse lists — then lives in a per-workgroup buffer in
// global memory (L2 / Infinity-Cache resident; only the block-reduction words stay in LDS): every array pointer of the kernel
// derives from that buffer, so the instantiation addresses it with global loads and stores, and the barriers that ordered the
// LDS phases order these too (the waves of a workgroup share the CU's vector L1, which is write-through).  Slower per flop
// than the LDS variants, but there is no index-set size the build refuses any more (the reference has no limit either:
// src/synthesis.jl:46-62).
template <bool MLDS, int WPE, bool GW, bool BIG = false>
__global__ __launch_bounds__(TB, WPE) void h2_column_tile_kernel(const KernelParams p) {
  static_assert(!(BIG && MLDS), "the big variant keeps its block in the workspace");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int RS = MLDS ? 17 : 16;
  constexpr int TSZ = MLDS ? kTileLdsTile : 256;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, c = lane & 15;
  const int T = p.T;
  const int nmax = p.nmax, mmax = p.mmax;
  const int NTmax = tile_nt(nmax), npadmax = 16 * NTmax, mpadmax = (mmax + 7) / 8 * 8 + 8;

  // ---- LDS carve (must match tile_kernel_lds_bytes) ----
  double* dp = BIG ? reinterpret_cast<double*>(p.big_ws + (int64_t)blockIdx.x * p.big_stride) : reinterpret_cast<double*>(lds_raw);
  double* R0 = dp; dp += tile_kernel_r0_doubles(nmax, mmax, p.tile_oth_rows, MLDS);
  double* Mlds = dp; if (MLDS) dp += (int64_t)tile_ht(NTmax) * kTileLdsTile;
  double* csrA_v = dp; dp += p.nnzA_cap;
  double* cscA_v = dp; dp += p.nnzA_cap;
  double* csrB_v = dp; dp += p.nnzB_cap;
  double* cscB_v = dp; dp += p.nnzB_cap;
  double* wprev = dp; dp += npadmax;
  double* wcur = dp; dp += npadmax;
  double* red = BIG ? reinterpret_cast<double*>(lds_raw) : dp; dp += 16;      // (block reductions always through LDS)
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

  // Work queue: the launch's subproblems are ordered by descending ñx (≈ cost) and handed out through one atomic counter
  // (cleared by the host before every launch): columns differ by 4× in passes and by 10× in ñx³, a fixed stride leaves
  // workgroups idle behind the one that drew two long columns.  Every workgroup leaves when the counter passes nsub.
  __shared__ int s_next;
  for (;;) {
    if (tid == 0) s_next = p.work_counter ? atomicAdd(p.work_counter, 1) : -1;
    __syncthreads();
    const int it_sub = s_next;
    __syncthreads();
    if (it_sub < 0 || it_sub >= p.nsub) break;
    // A work item is one subproblem — or, in the CG build, one COUPLED GROUP of columns (has_w = 3: non-diagonal B̃1 block,
    // src/synthesis.jl:42,50): its columns are consecutive subproblems that share s_x, s_u, Ã, B̃2 and differ in mask, right-hand
    // side and weight scale.  `sd`, the mask / destination tables, the factor slots and the vectors below always describe the
    // column in hand (`set_column`); a plain subproblem is a group of one.
    const int first_sub = p.order[p.order_off + it_sub];
    SubDesc sd = p.subs[first_sub];
    const int ncol = (GW && sd.has_w == 3) ? sd.pad_ : 1;
    const int n = sd.n, m = sd.m, nm = n + m;
    const int NT = tile_nt(n), HT = tile_ht(NT), npad = 16 * NT;
    const int64_t vlen = (int64_t)(T + 1) * n, zlen = (int64_t)T * nm;
    const int64_t vstride_col = 3 * vlen + 7 * zlen, fstride_col = (int64_t)(T + 1) * HT * 256 + (int64_t)npad * npad;
    double* qv;                         // q, then Δλ = (S+δI)⁻¹ r
    double* rv;                         // residual r = f − E z of the iterate
    double* rt;                         // residual of the trial point
    double* zc;                         // the primal iterate z = (x_t, u_t)_t, [T][ñx+ñu]
    double* zt;                         // the trial point z + H⁻¹EᵀΔλ
    const uint8_t* mask;
    const int32_t* dest;
    double* fcol;                       // the column's factor slots −P_k
    double* Pfull;                      // row-major npad×npad copy of the latest −P_k
    int gcol = 0;
    auto set_column = [&](int c2) {
      gcol = c2;
      if (c2 > 0 || ncol > 1) sd = p.subs[first_sub + c2];
      double* vb = vecs + (int64_t)c2 * vstride_col;
      qv = vb; rv = vb + vlen; rt = vb + 2 * vlen; zc = vb + 3 * vlen; zt = zc + zlen;
      mask = p.mask_pool + sd.off_mask;
      dest = p.dest_pool + sd.off_dest;
      fcol = facws + (int64_t)c2 * fstride_col;
      Pfull = fcol + (int64_t)(T + 1) * HT * 256;
    };
    set_column(0);
    const int32_t* su = p.idx_pool + sd.off_su;
    auto hx = [&](int i) -> double { return sd.has_w ? p.w_pool[sd.off_w + i] : 1.0; };
    auto hu = [&](int j) -> double { return sd.has_w ? p.w_pool[sd.off_w + n + j] : 1.0; };
    auto gx = [&](int i) -> double { return sd.has_w ? p.w_pool[sd.off_w + nm + i] : 0.0; };
    auto gu = [&](int j) -> double { return sd.has_w ? p.w_pool[sd.off_w + nm + n + j] : 0.0; };
    const bool has_w = sd.has_w != 0;

    __syncthreads();   // previous subproblem fully done with LDS
    unsigned long long tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // SLS_PHASE_TIMERS: setup, residual, build, sweep, store, substitution
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
    auto lap = [&](int slot) { if (p.dbg && p.dbg_level < 3) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tc[slot] += now - tlast; tlast = now; } };

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
    auto schur_scale = [&]() -> double {
      double sc = 0.0;
      for (int i = tid; i < n; i += TB) {
        double s = hx(i);
        for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) s = fma(csrA_v[e] * csrA_v[e], hx(csrA_i[e]), s);
        for (int e = csrB_p[i]; e < csrB_p[i + 1]; ++e) s = fma(csrB_v[e] * csrB_v[e], hu(csrB_i[e]), s);
        sc = fmax(sc, s);
      }
      return tblock_max(sc, red, tid);
    };
    double delta = p.delta_rel * schur_scale();

    // The primal iterate z is kept explicitly and every residual is evaluated from it, r = f − E z: round 1 evaluated
    // z = H⁻¹(Eᵀλ − g) from the accumulated multiplier in every pass, which limits the residual to ≈ 1e-15·‖λ‖ — 2e-12 on the
    // grid plant's near-singular columns, i.e. an error of 2e-12/σ_min ≈ 5e-5 in Φ.  With z stored, a pass only adds
    // Δz = H⁻¹EᵀΔλ for the small correction Δλ = (S+δI)⁻¹r, and the residual floor is that of E z itself (≈ 1e-16).
    // zpass: zdst = zsrc + H⁻¹(Eᵀdl) on the free variables (init: zdst = −H⁻¹g), rdst = f − E zdst; returns ‖rdst‖∞.
    // One barrier per time step; Δλ slices, the row carried to the next step and x_t/u_t are staged in the phase-shared LDS.
    const double* cur_g = nullptr;     // linear term of the solve in progress: per (t, variable), NULL = the column's own g
    bool cur_f = true;                 // right-hand side f = e_pos (the column's) or 0 (projection solves of the CG loop)
    auto zpass = [&](const double* dl, const double* zsrc, double* zdst, double* rdst) -> double {
      double rmax = 0.0;
      double* const stage = R0;                                       // [3][npad]
      double* const carry0 = stage + 3 * npad;                        // [2][npad]
      double* const xu0 = stage + 5 * npad;                           // [2][npad + mpad]
      const int xus = npad + mpadmax;
      for (int i = tid; i < n; i += TB) {
        carry0[i] = (cur_f && i == sd.pos) ? 1.0 : 0.0;               // f_0 = e_pos
        stage[i] = dl ? dl[i] : 0.0;
        stage[npad + i] = dl ? dl[(int64_t)n + i] : 0.0;
      }
      __syncthreads();
      for (int t = 0; t < T; ++t) {
        const double* l0 = stage + (t % 3) * npad;
        const double* l1 = stage + ((t + 1) % 3) * npad;
        double* l2 = stage + ((t + 2) % 3) * npad;
        double* xt_ = xu0 + (t & 1) * xus;
        double* ut_ = xt_ + npad;
        if (dl && t + 2 <= T) for (int i = tid; i < n; i += TB) l2[i] = dl[(int64_t)(t + 2) * n + i];
        const uint8_t* mk = mask + (int64_t)t * nm;
        for (int q = tid; q < nm; q += TB) {
          double v = 0.0;
          if (mk[q]) {
            if (q < n) {
              if (dl) {
                double acc = 0.0;
                for (int e = cscA_p[q]; e < cscA_p[q + 1]; ++e) acc = fma(cscA_v[e], l1[cscA_i[e]], acc);
                v = fma(hx(q), l0[q] - acc, zsrc[(int64_t)t * nm + q]);
              } else {
                v = -hx(q) * (cur_g ? cur_g[(int64_t)t * nm + q] : gx(q));
              }
            } else {
              const int j = q - n;
              if (dl) {
                double acc = 0.0;
                for (int e = cscB_p[j]; e < cscB_p[j + 1]; ++e) acc = fma(cscB_v[e], l1[cscB_i[e]], acc);
                v = fma(hu(j), -acc, zsrc[(int64_t)t * nm + q]);
              } else {
                v = -hu(j) * (cur_g ? cur_g[(int64_t)t * nm + q] : gu(j));
              }
            }
          }
          zdst[(int64_t)t * nm + q] = v;
          if (q < n) xt_[q] = v; else ut_[q - n] = v;
        }
        __syncthreads();
        const double* cin = carry0 + (t & 1) * npad;
        double* cout = carry0 + ((t + 1) & 1) * npad;
        for (int i = tid; i < n; i += TB) {
          const double r = cin[i] - xt_[i];
          rdst[(int64_t)t * n + i] = r;
          rmax = resid_max(rmax, r);
          double acc = 0.0;
          for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) acc = fma(csrA_v[e], xt_[csrA_i[e]], acc);
          for (int e = csrB_p[i]; e < csrB_p[i + 1]; ++e) acc = fma(csrB_v[e], ut_[csrB_i[e]], acc);
          cout[i] = acc;
        }
      }
      __syncthreads();
      const double* cfin = carry0 + (T & 1) * npad;
      for (int i = tid; i < n; i += TB) {
        rdst[(int64_t)T * n + i] = cfin[i];
        rmax = resid_max(rmax, cfin[i]);
      }
      return tblock_max(rmax, red, tid);
    };

    lap(0);
    double resid;
    if (!has_w && sd.pos >= 0) {                 // g = 0: z = 0 and r = f = e_pos exactly
      for (int i = tid; i < (T + 1) * n; i += TB) rv[i] = (i == sd.pos) ? 1.0 : 0.0;
      for (int64_t i = tid; i < zlen; i += TB) zc[i] = 0.0;
      __syncthreads();
      resid = 1.0;
    } else {
      resid = zpass(nullptr, nullptr, zc, rv);
    }
    lap(1);
    int iters = 0;
    int status = 0;
    bool trial_is_answer = false;            // the answer is the trial point zt (residual `resid`), else the iterate zc
    bool grp_done = false;                   // CG build: every column of the group has its certified answer
    unsigned long long ans_trial = 0;        // bit c: the answer of column c of the group is its trial point

    if (resid > p.tol || (GW && (sd.has_w >= 2 || p.objective == 1))) {
      // P_k y = −(N_k y), N_k symmetric tiles of slot k, y an LDS vector of npad entries (zero padded).  Every wave takes a
      // contiguous chunk of the (row-major) tile list, four tiles in flight; every stored tile is read once and serves both
      // mirror images: the row image is accumulated per lane and reduced over the 16 column lanes when the chunk leaves a
      // tile row, the column image is reduced over the 4 row groups per tile.  Per block step: [prefetch the first tiles,
      // clear the partial vectors, build y] barrier [tiles → per-wave partial vectors] barrier [consumer sums the partials].
      double* const pv = R0;                         // [NW][npad]
      double* const yv = R0 + (int64_t)NW * npad;
      double* const sv = yv + npad;                  // staged (masked, weighted) neighbour vector of the y build
      const int mt0 = (HT * w) / NW, mt1 = (HT * (w + 1)) / NW;
      double xf[4][4];
      auto mv_prefetch = [&](const double* Ns) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int t = min(mt0 + u, HT - 1);
#pragma unroll
          for (int r = 0; r < 4; ++r) xf[u][r] = Ns[(int64_t)t * 256 + 64 * r + lane];
        }
        for (int i = tid; i < NW * npad; i += TB) pv[i] = 0.0;
      };
      auto mv_tiles = [&](const double* Ns) {
        double* mypv = pv + w * npad;
        int cur = -1;
        double pr[4] = {0.0, 0.0, 0.0, 0.0}, yI[4] = {0.0, 0.0, 0.0, 0.0};
        auto flush = [&]() {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double v = pr[r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if (c == 0) mypv[16 * cur + 4 * r + g] += v;
            pr[r] = 0.0;
          }
        };
        for (int tb = mt0; tb < mt1; tb += 4) {
          if (tb > mt0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int t = min(tb + u, mt1 - 1);
#pragma unroll
              for (int r = 0; r < 4; ++r) xf[u][r] = Ns[(int64_t)t * 256 + 64 * r + lane];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (tb + u < mt1) {
              const int ij = tl[tb + u];
              const int i = ij & 0xffff, j = ij >> 16;
              if (i != cur) {
                if (cur >= 0) flush();
                cur = i;
#pragma unroll
                for (int r = 0; r < 4; ++r) yI[r] = yv[16 * i + 4 * r + g];
              }
              const double yJ = yv[16 * j + c];
#pragma unroll
              for (int r = 0; r < 4; ++r) pr[r] = fma(xf[u][r], yJ, pr[r]);
              if (j != i) {
                double pc = xf[u][0] * yI[0];
#pragma unroll
                for (int r = 1; r < 4; ++r) pc = fma(xf[u][r], yI[r], pc);
                pc += __shfl_xor(pc, 16);
                pc += __shfl_xor(pc, 32);
                if (g == 0) mypv[16 * j + c] += pc;
              }
            }
          }
        }
        if (cur >= 0) flush();
      };
      auto mv_result = [&](int i) -> double {        // (P_k y)[i] after the barrier that follows mv_tiles
        double sres = 0.0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) sres += pv[ww * npad + i];
        return -sres;
      };

      // =================== multiplier corrections with minimal-residual step lengths ===================
      // r = f − E z.  The plain method of multipliers (Δλ = (S+δI)⁻¹ r, z += H⁻¹EᵀΔλ) contracts the component of r along a
      // singular direction σ by δ/(σ²+δ) per pass: two passes on every well-posed column, but ≈ 0.85 per pass on the grid
      // plant's near-boundary columns (σ² ≈ δ/5), which round 1 stopped at ‖r‖ ≈ 3e-11.  The same two building blocks (one
      // application of the factor, one pass over E) allow an exact line search:
      //   Δλ = (S+δI)⁻¹ r,  trial point z′ = z + H⁻¹EᵀΔλ with its residual r′ = f − E z′,  S Δλ = r − r′,
      //   α = (r·SΔλ)/(SΔλ·SΔλ) minimises ‖r − α SΔλ‖₂,   z ← (1−α) z + α z′,  r ← (1−α) r + α r′.
      // On a well-posed column α = 1 + O(δ) and the trial point of the second step is accepted as it is (the old two passes);
      // once the fast directions are gone, α ≈ (σ²+δ)/σ² removes a slow one in a single step.  An inconsistent system shows
      // as r ⟂ SΔλ (no step length reduces the residual): stop, keep the trial point (a plain step) and flag the column.
      bool consistent = false;       // the system being solved is known to be consistent (projection solves): never give up early
      // ---- state of the projected-CG loop for dense cost Hessians (GW build only; see below) ----
      int cgphase = 0, cg = 0, rises = 0, st_keep = 0, it_keep = 0;
      double rho = 0.0, rho0 = 0.0, rho_acc = 0.0;
      bool proj_ok = true;
      int aa_k = 0, aa_col = 0; bool aa_prev = false; double aa_gmin = 1e300;      // sum-of-norms loop: Anderson acceleration state
      bool need_factor = true;       // the column in hand has no factor yet
      for (;;) {                      // one trip = one diagonal-weight solve (the plain build makes exactly one)
      if (need_factor) {
        // =================== factor: −P_k = sweep(D'_k) ===================
        double* const Yp = R0;                                        // one panel (LDS-resident block) or two
        double* const Lb = R0 + (int64_t)(MLDS ? 1 : 2) * NT * 256;
        // Ã·Q image, `orows` rows at a time (a multiple of 16; the whole image when LDS allows), row stride npad + 1; aliases the
        // sweep's panel
        double* const Oth = R0;
        const int ost = npad + 1;
        const int orows = min(p.tile_oth_rows, npad);
        for (int k = 0; k <= T; ++k) {
          double* const slot = fcol + (int64_t)k * HT * 256;
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
          if (k == 0) {
            // D'_0 = δI + Wx_0 is diagonal: −P_0 written directly, no sweep
            for (int t = w; t < HT; t += NW) {
              const int ij = tl[t];
              const int I = ij & 0xffff, J = ij >> 16;
  #pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int i = 16 * I + 4 * r + g, j = 16 * J + c;
                const double v = (i == j) ? ((j < n) ? -1.0 / (delta + wcur[j]) : -1.0) : 0.0;
                Mb[(int64_t)t * TSZ + (4 * r + g) * RS + c] = v;
                if (MLDS) slot[(int64_t)t * 256 + 64 * r + lane] = v;
                else {
                  Pfull[(int64_t)i * npad + j] = v;
                  if (I != J) Pfull[(int64_t)j * npad + i] = v;
                }
              }
            }
            __syncthreads();
            lap(2);
            continue;
          }
          // Oth = Ã·Q,  Q = W + W N W,  N = −P_{k−1}: row i is a sparse combination of rows of N.  LDS-resident block: N is read
          // where the sweep left it (stored half, the mirror image through the row-stride-17 tiles, conflict-free both ways);
          // block in the workspace: from its full row-major copy (coalesced rows).  Then D' = δI + Wx_k + Oth·Ãᵀ, stored half.
          // Strips of `orows` rows: step 1 fills the strip, step 2 builds the tiles of its tile rows.
          // LDS-resident block built in strips: N is read from the very tiles D' would overwrite, so D' is staged in the slot
          // (workspace, same tile order) and brought in after the last strip
          const bool stage_d = MLDS && orows < npad;
          for (int i0 = 0; i0 < npad; i0 += orows) {
            const int i1 = min(i0 + orows, npad);
            if (MLDS) {
              // tile_index(I,J) = rb(I) + J with rb(I) = I·NT − I(I−1)/2 − I: the address splits into a part that depends on the
              // entry (q) only and a part that depends on the lane (column) only
              constexpr int NCH = 3;                       // npad ≤ 144 for an LDS-resident block
              int L1[NCH], L2[NCH], Jl[NCH]; double wc[NCH];
  #pragma unroll
              for (int ch = 0; ch < NCH; ++ch) {
                const int cc = min(64 * ch + lane, npad - 1);
                const int J = cc >> 4, b = cc & 15;
                Jl[ch] = J; L1[ch] = J * TSZ + b; L2[ch] = (J * NT - ((J * (J - 1)) >> 1) - J) * TSZ + b * RS;
                wc[ch] = wprev[cc];
              }
              for (int i = i0 + w; i < min(i1, n); i += NW) {
                double acc[NCH];
  #pragma unroll
                for (int ch = 0; ch < NCH; ++ch) acc[ch] = 0.0;
                for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) {
                  const int q = csrA_i[e];
                  const double vw = csrA_v[e] * wprev[q];
                  if (vw != 0.0) {
                    const int Iq = q >> 4, a = q & 15;
                    const int U1 = (Iq * NT - ((Iq * (Iq - 1)) >> 1) - Iq) * TSZ + a * RS, U2 = Iq * TSZ + a;
  #pragma unroll
                    for (int ch = 0; ch < NCH; ++ch) {
                      if (64 * ch < npad) {
                        const double nv = Mb[(Iq <= Jl[ch]) ? U1 + L1[ch] : U2 + L2[ch]];
                        acc[ch] = fma(vw, ((q == 64 * ch + lane) ? 1.0 : 0.0) + nv * wc[ch], acc[ch]);
                      }
                    }
                  }
                }
  #pragma unroll
                for (int ch = 0; ch < NCH; ++ch)
                  if (64 * ch + lane < npad) Oth[(int64_t)(i - i0) * ost + 64 * ch + lane] = acc[ch];
              }
            } else {
              for (int i = i0 + w; i < min(i1, n); i += NW) {
                for (int cb = 0; cb < npad; cb += 64) {
                  const int cc = cb + lane;
                  if (cc >= npad) break;
                  const double wcc = wprev[cc];
                  double acc = 0.0;
                  for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) {
                    const int q = csrA_i[e];
                    const double vw = csrA_v[e] * wprev[q];
                    if (vw != 0.0) acc = fma(vw, ((q == cc) ? 1.0 : 0.0) + Pfull[(int64_t)q * npad + cc] * wcc, acc);
                  }
                  Oth[(int64_t)(i - i0) * ost + cc] = acc;
                }
              }
            }
            __syncthreads();
            // step 2: unit = (tile of tile rows [i0/16, i1/16), register row group r), contiguous chunks per wave
            {
              const int I0 = i0 >> 4, I1 = i1 >> 4;
              const int tb0 = tile_index(I0, I0, NT), tb1 = (I1 < NT) ? tile_index(I1, I1, NT) : HT;
              const int nunits = 4 * (tb1 - tb0);
              const int u0 = (nunits * w) / NW, u1 = (nunits * (w + 1)) / NW;
              for (int u = u0; u < u1; ++u) {
                const int t = tb0 + (u >> 2), r = u & 3;
                const int ij = tl[t];
                const int I = ij & 0xffff, J = ij >> 16;
                const int irow = 4 * r + g, i = 16 * I + irow, j = 16 * J + c;
                double val = (i == j) ? ((j < n) ? delta + wcur[j] : 1.0) : 0.0;
                if (i < n && j < n) {
                  const double* orow = Oth + (int64_t)(i - i0) * ost;
                  for (int e = csrA_p[j]; e < csrA_p[j + 1]; ++e) val = fma(orow[csrA_i[e]], csrA_v[e], val);
                }
                if (stage_d) slot[(int64_t)t * 256 + irow * 16 + c] = val;
                else Mb[(int64_t)t * TSZ + irow * RS + c] = val;
              }
            }
            __syncthreads();
          }
          if (stage_d) {
            for (int t = w; t < HT; t += NW) {
  #pragma unroll
              for (int r = 0; r < 4; ++r) Mb[(int64_t)t * TSZ + (4 * r + g) * RS + c] = slot[(int64_t)t * 256 + 64 * r + lane];
            }
            __syncthreads();
          }
          // B̃ Wu B̃ᵀ: row-owned read-modify-write of the stored half (both orders inside a diagonal tile)
          {
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
          }
          __syncthreads();
          lap(2);
          tile_sweep<RS, TSZ, !MLDS>(Mb, NT, Yp, Lb, tl, HT, 0.25 * delta, tid, (p.dbg && p.dbg_level == 2) ? tc + 6 : nullptr);
          lap(3);
          // write-out: the slot (LDS-resident block) or the full row-major copy the next build gathers rows from (workspace block)
          for (int t = w; t < HT; t += NW) {
            const int ij = tl[t];
            const int I = ij & 0xffff, J = ij >> 16;
            const d4 x = tile_load<RS>(Mb + (int64_t)t * TSZ, g, c);
            if (MLDS) {
  #pragma unroll
              for (int r = 0; r < 4; ++r) slot[(int64_t)t * 256 + 64 * r + lane] = x[r];
            }
            if (!MLDS && k < T) {
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
      need_factor = false;
      }
      if (resid > p.tol) {
      double prev = resid, prev2 = resid;
      int itmax = p.max_iters;
      trial_is_answer = false;
      for (int it = 1; it <= itmax; ++it) {
        iters = it;
        // (inside the two sweeps every value that crosses threads does so through LDS — sv, yv, the partial vectors; q_k in the
        //  global workspace is written and read back by the same thread — so their barriers are LDS-only, except when the carve
        //  itself lives in global memory)
        auto sweep_barrier = [&]() { if constexpr (BIG) __syncthreads(); else lds_barrier(); };
        // forward: y_k = r_k + Ã(Wx_{k−1} q_{k−1});  q_k = P_k y_k
        for (int k = 0; k <= T; ++k) {
          const double* Ns = fcol + (int64_t)k * HT * 256;
          mv_prefetch(Ns);
          if (k >= 1) {          // Wx_{k−1} q_{k−1} once, coalesced, into LDS: the sparse products below then gather from LDS
            const uint8_t* mk = mask + (int64_t)(k - 1) * nm;
            const double* qp = qv + (int64_t)(k - 1) * n;
            for (int i = tid; i < n; i += TB) sv[i] = mk[i] ? hx(i) * qp[i] : 0.0;
            sweep_barrier();
          }
          for (int i = tid; i < npad; i += TB) {
            double acc = 0.0;
            if (i < n) {
              acc = rv[(int64_t)k * n + i];
              if (k >= 1)
                for (int e = csrA_p[i]; e < csrA_p[i + 1]; ++e) acc = fma(csrA_v[e], sv[csrA_i[e]], acc);
            }
            yv[i] = acc;
          }
          sweep_barrier();
          mv_tiles(Ns);
          sweep_barrier();
          for (int i = tid; i < n; i += TB) qv[(int64_t)k * n + i] = mv_result(i);
          sweep_barrier();
        }
        // backward: z_k = q_k + P_k (Wx_k (Ãᵀ z_{k+1})), z_k overwrites q_k
        for (int k = T; k >= 0; --k) {
          if (k < T) {
            const double* Ns = fcol + (int64_t)k * HT * 256;
            mv_prefetch(Ns);
            const uint8_t* mk = mask + (int64_t)k * nm;
            const double* dl1 = qv + (int64_t)(k + 1) * n;
            for (int i = tid; i < n; i += TB) sv[i] = dl1[i];
            sweep_barrier();
            for (int q = tid; q < npad; q += TB) {
              double acc = 0.0;
              if (q < n && mk[q]) {
                for (int e = cscA_p[q]; e < cscA_p[q + 1]; ++e) acc = fma(cscA_v[e], sv[cscA_i[e]], acc);
                acc *= hx(q);
              }
              yv[q] = acc;
            }
            sweep_barrier();
            mv_tiles(Ns);
            sweep_barrier();
            for (int i = tid; i < n; i += TB) qv[(int64_t)k * n + i] += mv_result(i);
            sweep_barrier();
          }
        }
        __syncthreads();
        lap(5);
        const double rt_max = zpass(qv, zc, zt, rt);         // the trial point and its residual
        lap(1);
        if (rt_max <= p.tol) { resid = rt_max; trial_is_answer = true; break; }
        double pa = 0.0, pb = 0.0, pc = 0.0;
        for (int64_t i = tid; i < vlen; i += TB) {
          const double r0 = rv[i], sz = r0 - rt[i];
          pa = fma(r0, sz, pa); pb = fma(sz, sz, pb); pc = fma(r0, r0, pc);
        }
        const double da = tblock_sum(pa, red, tid), db = tblock_sum(pb, red, tid), dc = tblock_sum(pc, red, tid);
        // cos² of the angle between r and Sz = the fraction of ‖r‖₂² the best step length removes
        const bool no_progress = !(da > 0.0) || !(da * da > 0.02 * db * dc);
        if (no_progress && !consistent) {
          // nothing to gain along z: an inconsistent system (residual far above the acceptance level) or the FP64 floor
          resid = rt_max; trial_is_answer = true;
          if (rt_max > p.tol_ok) status = 1;
          break;
        }
        const double alpha = (da > 0.0 && db > 0.0) ? da / db : 1.0;
        double rn = 0.0;
        for (int64_t i = tid; i < vlen; i += TB) {
          const double r1 = fma(alpha, rt[i] - rv[i], rv[i]);
          rv[i] = r1;
          rn = fmax(rn, fabs(r1));
        }
        for (int64_t i = tid; i < zlen; i += TB) zc[i] = fma(alpha, zt[i] - zc[i], zc[i]);
        rn = tblock_max(rn, red, tid);
        const bool stalled = it >= 2 && rn > p.stag * prev;
        resid = rn;
        if (stalled && !consistent) {
          // still contracting (still_contracting, sls_device.h): a near-singular consistent column, not an inconsistent one — go on
          const bool patient = p.max_iters_slow > 0 && (rn > p.tol_ok || itmax > p.max_iters) && still_contracting(it >= 3 ? prev2 : prev, prev, rn);
          if (!patient) {                                   // above the acceptance level: inconsistent; below it: the FP64 floor of this column
            if (rn > p.tol_ok) status = 1;
            if (rn > prev) { resid = rt_max; trial_is_answer = true; }      // the step did not even help: keep the trial point
            break;
          }
          itmax = max(itmax, p.max_iters_slow);
        }
        prev2 = prev; prev = rn;
        if (rn <= p.tol) break;
      }
      }
      if constexpr (!GW) {
        break;
      } else {
        // =================== non-diagonal cost Hessian: projected conjugate gradients ===================
        // The weight record carries b·W (W = [C̃1 D̃12] on (s_x,s_u), src/synthesis.jl:50,76-83), so the cost is
        // Σ_t ‖W z_t + d‖² with the dense Hessian G = WᵀW.  The solve above used diag(G) in its place; it is now the
        // constraint preconditioner: K(r) = argmin ½vᵀdiag(G)v − rᵀv s.t. E v = 0 is the same machinery with linear term r
        // and right-hand side 0.  Conjugate gradients in the null space of E (Gould–Hribar–Nocedal):
        //   z feasible,  r = G z + g,  s = −K(r),  p = s;   α = ρ/(p·Gp),  z += αp,  r += αGp,  s = −K(r),  p = s + βp,
        // with ρ = sᵀdiag(G)s and r replaced by its projected part −diag(G)·s after every projection (residual update: r
        // itself stays O(1) at the optimum (= Eᵀμ), and −r·s evaluated from it floors at ≈ 1e-16 — 1e-6 in Φ).
        // Written as a state machine around the one solve loop so that the multiplier iteration is compiled once.
        if (p.objective == 1) {
          // =================== sum-of-norms objective  min Σ_t ‖W z_t‖₂  s.t. E z = f   (SLS_SOLVE_SUM_OF_NORMS) ===================
          // The column-separable bound of the 𝓗∞ norm (BASELINE configs[3]; no reference exists: SURVEY §0 F3; oracle and
          // certificate: oracle/sls_son_oracle.py).  ADMM on  min Σ‖y_t‖  s.t. y = W z, E z = f:
          //   z ← argmin ½‖W z − (y − u)‖² s.t. E z = f     — the diagonal-weight solve above with linear term −W(y − u)
          //   y_t ← (1 − 1/(ρ‖v_t‖))₊ v_t,  v = W z + u      — block soft threshold, one wave per time step
          //   u ← u + W z − y;   ρ starts at 8 / max_t‖W z_t‖ of the 𝓗₂ solution and is doubled / halved every 10 steps when the
          //   primal / dual residual leads by 10×; W z over-relaxed by 1.8 in the y and u updates (7× fewer steps on chain-4096
          //   interior columns than ρ = 1 without relaxation).
          // The answer is the last projected z: feasible to the solve's tolerance whatever the ADMM accuracy.
          constexpr double kRelax = 1.8;
          double* yv_ = zt + zlen;                  // y
          double* uv_ = yv_ + zlen;                 // scaled multiplier u
          double* gl = uv_ + zlen;                  // linear term of the projection in progress
          double* ztmp = gl + zlen;                 // warm start of the next projection
          // Anderson acceleration of the fixed-point map s = (y, u) ↦ F(s) (same scheme as the one-wave kernel's loop): g of this
          // step, F(s) and g of the previous one, the last AAM differences of each; every vector is (y part, u part)
          constexpr int AAM = 5;
          const int64_t L2 = 2 * zlen;
          double* sgc = ztmp + zlen; double* sFp = sgc + L2; double* sgp = sFp + L2; double* sdF = sgp + L2; double* sdG = sdF + AAM * L2;
          auto wof = [&](int q) -> double { return rsqrt((q < n) ? hx(q) : hu(q - n)); };      // W = diag(H)^{1/2}
          // next projection.  Warm start: with the multipliers kept, a change Δg of the linear term moves the iterate by −H⁻¹Δg;
          // from there the residual is E H⁻¹Δg — small once the ADMM steps are small — instead of that of z = −H⁻¹g.
          auto start_projection = [&]() {
            const double* zprev = trial_is_answer ? zt : zc;
            for (int64_t e = tid; e < zlen; e += TB) {
              const int q = (int)(e % nm);
              const double gnew = mask[e] ? -wof(q) * (yv_[e] - uv_[e]) : 0.0;
              ztmp[e] = mask[e] ? zprev[e] - ((q < n) ? hx(q) : hu(q - n)) * (gnew - gl[e]) : 0.0;
              gl[e] = gnew;
            }
            for (int64_t i = tid; i < vlen; i += TB) qv[i] = 0.0;
            __syncthreads();
            cur_g = gl; cur_f = true; consistent = true;
            resid = zpass(qv, ztmp, zc, rv);
            trial_is_answer = false;
          };
          if (cgphase == 0) {                        // the 𝓗₂ solve has just finished: its z starts the iteration
            if (resid > p.tol_ok) break;             // infeasible / not converged: reported as it is
            it_keep = iters;
            const double* z0 = trial_is_answer ? zt : zc;
            double big = 0.0;                        // max_t ‖W z_t‖ of the 𝓗₂ solution sets the scale of ρ
            for (int t = w; t < T; t += NW) {
              double part = 0.0;
              for (int q = lane; q < nm; q += 64)
                if (mask[(int64_t)t * nm + q]) { const double v = wof(q) * z0[(int64_t)t * nm + q]; part = fma(v, v, part); }
#pragma unroll
              for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
              big = fmax(big, part);
            }
            big = sqrt(tblock_max(big, red, tid));
            for (int64_t e = tid; e < zlen; e += TB) { yv_[e] = mask[e] ? wof((int)(e % nm)) * z0[e] : 0.0; uv_[e] = 0.0; gl[e] = 0.0; }
            __syncthreads();
            rho = (big > 0.0) ? 8.0 / big : 1.0; cg = 0; cgphase = 1; aa_k = 0; aa_col = 0; aa_prev = false; aa_gmin = 1e300;
            start_projection();
            continue;
          }
          const double* zp = trial_is_answer ? zt : zc;
          proj_ok = proj_ok && resid <= p.tol_ok;
          double rp2 = 0.0, rd2 = 0.0, nx2 = 0.0, gn2 = 0.0;
          for (int t = w; t < T; t += NW) {          // one wave per time step
            const int64_t o = (int64_t)t * nm;
            double part = 0.0;
            for (int q = lane; q < nm; q += 64)
              if (mask[o + q]) {
                const double v = fma(kRelax, wof(q) * zp[o + q], fma(1.0 - kRelax, yv_[o + q], uv_[o + q]));
                part = fma(v, v, part);
              }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            const double nv = sqrt(part);
            const double sh = (nv * rho > 1.0) ? 1.0 - 1.0 / (rho * nv) : 0.0;
            for (int q = lane; q < nm; q += 64)
              if (mask[o + q]) {
                const double wz = wof(q) * zp[o + q];
                const double xh = fma(kRelax, wz, (1.0 - kRelax) * yv_[o + q]);       // over-relaxed W z
                const double yn = sh * (xh + uv_[o + q]);
                const double dy = yn - yv_[o + q], dp = wz - yn, du = xh - yn;
                rd2 = fma(dy, dy, rd2); rp2 = fma(dp, dp, rp2); nx2 = fma(wz, wz, nx2);
                gn2 = fma(dy, dy, fma(du, du, gn2));
                yv_[o + q] = yn;
                uv_[o + q] += du;
                sgc[o + q] = dy; sgc[zlen + o + q] = du;
              }
          }
          const double rp = sqrt(tblock_sum(rp2, red, tid)), rd = rho * sqrt(tblock_sum(rd2, red, tid));
          const double nx = sqrt(tblock_sum(nx2, red, tid));
          const double gn = sqrt(tblock_sum(gn2, red, tid));
          ++cg;
          const bool converged = fmax(rp, rd) <= p.son_tol * fmax(1.0, nx);
          if (!converged && cg < p.son_maxit && proj_ok) {
            bool aa_reset = false;
            if (cg % 10 == 0) {
              const double sc_ = (rp > 10.0 * rd) ? 2.0 : ((rd > 10.0 * rp) ? 0.5 : 1.0);
              if (sc_ != 1.0) {
                rho *= sc_;
                for (int64_t e = tid; e < zlen; e += TB) uv_[e] /= sc_;
                aa_reset = true;                     // the map changed with ρ
              }
            }
            __syncthreads();
            if (p.son_anderson) {
              if (aa_reset || gn > 10.0 * aa_gmin || cg <= p.son_aa_start) {
                aa_k = 0; aa_col = 0; aa_prev = false; aa_gmin = (aa_reset || cg <= p.son_aa_start) ? 1e300 : gn;
              } else {
                aa_gmin = fmin(aa_gmin, gn);
                if (aa_prev) {
                  double* dF = sdF + (int64_t)aa_col * L2; double* dG = sdG + (int64_t)aa_col * L2;
                  for (int64_t e = tid; e < zlen; e += TB) {
                    if (!mask[e]) continue;
                    dF[e] = yv_[e] - sFp[e]; dF[zlen + e] = uv_[e] - sFp[zlen + e];
                    dG[e] = sgc[e] - sgp[e]; dG[zlen + e] = sgc[zlen + e] - sgp[zlen + e];
                  }
                  aa_col = (aa_col + 1 == AAM) ? 0 : aa_col + 1;
                  aa_k = min(aa_k + 1, AAM);
                }
                for (int64_t e = tid; e < zlen; e += TB) {
                  if (!mask[e]) continue;
                  sFp[e] = yv_[e]; sFp[zlen + e] = uv_[e]; sgp[e] = sgc[e]; sgp[zlen + e] = sgc[zlen + e];
                }
                aa_prev = true;
                __syncthreads();
                if (aa_k > 0) {
                  double Am[AAM][AAM], bv[AAM];
#pragma unroll
                  for (int a = 0; a < AAM; ++a) { bv[a] = 0.0;
#pragma unroll
                    for (int b2 = 0; b2 < AAM; ++b2) Am[a][b2] = 0.0; }
                  for (int64_t e = tid; e < zlen; e += TB) {
                    if (!mask[e]) continue;
                    double gy[AAM], gu2[AAM];
#pragma unroll
                    for (int a = 0; a < AAM; ++a) { gy[a] = (a < aa_k) ? sdG[(int64_t)a * L2 + e] : 0.0; gu2[a] = (a < aa_k) ? sdG[(int64_t)a * L2 + zlen + e] : 0.0; }
                    const double cy = sgc[e], cu = sgc[zlen + e];
#pragma unroll
                    for (int a = 0; a < AAM; ++a) {
                      bv[a] = fma(gy[a], cy, fma(gu2[a], cu, bv[a]));
#pragma unroll
                      for (int b2 = a; b2 < AAM; ++b2) Am[a][b2] = fma(gy[a], gy[b2], fma(gu2[a], gu2[b2], Am[a][b2]));
                    }
                  }
#pragma unroll
                  for (int a = 0; a < AAM; ++a) {
                    bv[a] = tblock_sum(bv[a], red, tid);
#pragma unroll
                    for (int b2 = a; b2 < AAM; ++b2) Am[a][b2] = tblock_sum(Am[a][b2], red, tid);
                  }
                  double tr = 0.0;
#pragma unroll
                  for (int a = 0; a < AAM; ++a) tr += Am[a][a];
                  const double regv = 1e-10 * tr / (double)aa_k;
#pragma unroll
                  for (int a = 0; a < AAM; ++a) {
                    Am[a][a] = (a < aa_k) ? Am[a][a] + regv : 1.0;
#pragma unroll
                    for (int b2 = 0; b2 < a; ++b2) Am[a][b2] = Am[b2][a];
                  }
                  double gam[AAM];
#pragma unroll
                  for (int a = 0; a < AAM; ++a) {
                    const double piv = 1.0 / Am[a][a];
#pragma unroll
                    for (int r = a + 1; r < AAM; ++r) {
                      const double f2 = Am[r][a] * piv;
#pragma unroll
                      for (int c2 = a; c2 < AAM; ++c2) Am[r][c2] = fma(-f2, Am[a][c2], Am[r][c2]);
                      bv[r] = fma(-f2, bv[a], bv[r]);
                    }
                  }
#pragma unroll
                  for (int a = AAM - 1; a >= 0; --a) {
                    double acc = bv[a];
#pragma unroll
                    for (int c2 = a + 1; c2 < AAM; ++c2) acc = fma(-Am[a][c2], gam[c2], acc);
                    gam[a] = acc / Am[a][a];
                  }
                  bool finite = true;
#pragma unroll
                  for (int a = 0; a < AAM; ++a) finite = finite && (fabs(gam[a]) < 1e6);
                  if (finite) {
                    for (int64_t e = tid; e < zlen; e += TB) {
                      if (!mask[e]) continue;
                      double ay = yv_[e], au = uv_[e];
#pragma unroll
                      for (int a = 0; a < AAM; ++a)
                        if (a < aa_k) { ay = fma(-gam[a], sdF[(int64_t)a * L2 + e], ay); au = fma(-gam[a], sdF[(int64_t)a * L2 + zlen + e], au); }
                      yv_[e] = ay; uv_[e] = au;
                    }
                  } else { aa_k = 0; aa_col = 0; aa_prev = false; aa_gmin = 1e300; }
                  __syncthreads();
                }
              }
            }
            start_projection();
            continue;
          }
          cur_g = nullptr; cur_f = true; consistent = false;
          iters = cg;
          status = !proj_ok ? 2 : (converged ? 0 : 2);
          break;
        }
        if (sd.has_w < 2) break;
        // ---- dense Hessian (has_w = 2) and coupled groups (has_w = 3 on the first column, 4 on the others): one machinery.
        // Variables: z_c for the group's columns c; cost Σ_{c,c'} M_cc' z_cᵀ G z_c' + 2 Σ_c g_cᵀ z_c with M = R Rᵀ (R = B1[c_j,c_j];
        // M = 1 for a single column, whose record carries b·W), G = WᵀW; constraints E_c z_c = f_c per column (same Ã, B̃2, the
        // column's own mask and right-hand side).  The constraint preconditioner is block diagonal: column c's solve with
        // M_cc·diag(G), so one projection = one multiplier solve per column, in turn, each with its own factor.
        const bool grp = p.subs[first_sub].has_w == 3;
        const double* wrec0 = p.w_pool + p.subs[first_sub].off_w + 2LL * nm;
        const double* Mmat = grp ? wrec0 + 1 : nullptr;
        const double* wrec = grp ? wrec0 + 1 + (int64_t)ncol * ncol : wrec0;
        const int nzw = (int)wrec[0], nnzw = (int)wrec[1];
        const double* w_rp = wrec + 2;            // CSR of W (b·W for a single column) by z-row: ptr[nzw+1], idx[nnzw] (variable), val[nnzw]
        const double* w_ri = w_rp + nzw + 1;
        const double* w_rv = w_ri + nnzw;
        const double* w_cp = w_rv + nnzw;         // CSC by variable: ptr[nm+1], idx[nnzw] (z-row), val[nnzw]
        const double* w_ci = w_cp + nm + 1;
        const double* w_cv = w_ci + nnzw;
        const double* w_reg = w_cv + nnzw;        // ridge term per variable [ñx+ñu] (sls_set_ridge; zeros when none)
        // per column, after the solve's own vectors: 0 the feasible iterate, 1 gradient (then its projected part), 2 search
        // direction, 3 Hessian·direction (holds s = −K(gradient) between a projection and the direction update), 4 temporary
        auto cvec = [&](int c2, int which) -> double* { return vecs + (int64_t)c2 * vstride_col + 3 * vlen + (int64_t)(2 + which) * zlen; };
        double* vin = R0;                         // LDS: one time slice of the input, then W·slice
        double* ub = R0 + (npad + mpadmax);
        auto gmat_raw = [&](const double* in, double* out) {      // out_t = Wᵀ(W in_t), not masked
          for (int t = 0; t < T; ++t) {
            for (int q = tid; q < nm; q += TB) vin[q] = in[(int64_t)t * nm + q];
            __syncthreads();
            for (int zr = tid; zr < nzw; zr += TB) {
              double acc = 0.0;
              for (int e = (int)w_rp[zr]; e < (int)w_rp[zr + 1]; ++e) acc = fma(w_rv[e], vin[(int)w_ri[e]], acc);
              ub[zr] = acc;
            }
            __syncthreads();
            for (int q = tid; q < nm; q += TB) {
              double acc = 0.0;
              for (int e = (int)w_cp[q]; e < (int)w_cp[q + 1]; ++e) acc = fma(w_cv[e], ub[(int)w_ci[e]], acc);
              out[(int64_t)t * nm + q] = acc;
            }
            __syncthreads();
          }
        };
        auto hess = [&](int which_in, int which_out) {             // out_c = mask_c ⊙ (Σ_c' M_cc' G in_c' + ridge ⊙ in_c)
          for (int c2 = 0; c2 < ncol; ++c2) gmat_raw(cvec(c2, which_in), cvec(c2, 4));
          for (int c2 = 0; c2 < ncol; ++c2) {
            const uint8_t* mk = p.mask_pool + p.subs[first_sub + c2].off_mask;
            const double* in_c = cvec(c2, which_in);
            double* o = cvec(c2, which_out);
            for (int64_t e = tid; e < zlen; e += TB) {
              double acc = 0.0;
              if (mk[e]) {
                acc = w_reg[(int)(e % nm)] * in_c[e];
                for (int c3 = 0; c3 < ncol; ++c3) acc = fma(Mmat ? Mmat[c2 * ncol + c3] : 1.0, cvec(c3, 4)[e], acc);
              }
              o[e] = acc;
            }
          }
          __syncthreads();
        };
        auto start_projection = [&]() {            // s = −K(gradient of the column in hand): linear term, right-hand side 0
          cur_g = cvec(gcol, 1); cur_f = false; consistent = true;
          resid = zpass(nullptr, nullptr, zc, rv);
          trial_is_answer = false;
        };
        auto start_certification = [&]() {         // the column's iterate, taken back onto E z = f by one more multiplier solve
          cur_g = nullptr; cur_f = true; consistent = false;
          for (int64_t i = tid; i < vlen; i += TB) qv[i] = 0.0;
          __syncthreads();
          resid = zpass(qv, cvec(gcol, 0), zc, rv);
          trial_is_answer = false;
        };
        if (cgphase == 0) {                        // the own (diagonal-weight) solve of the column in hand has just finished
          if (resid > p.tol_ok) break;             // infeasible / not converged: the whole group is reported as it is
          it_keep = max(it_keep, iters);
          {
            const double* z0 = trial_is_answer ? zt : zc;
            double* zg = cvec(gcol, 0);
            for (int64_t i = tid; i < zlen; i += TB) zg[i] = z0[i];
          }
          __syncthreads();
          if (gcol + 1 < ncol) {                   // next column of the group: its own factor and solve
            set_column(gcol + 1);
            delta = p.delta_rel * schur_scale();
            need_factor = true;
            iters = 0; status = 0;
            cur_g = nullptr; cur_f = true; consistent = false;
            resid = zpass(nullptr, nullptr, zc, rv);
            trial_is_answer = false;
            continue;
          }
          hess(0, 1);                              // gradient Σ_c' M_cc' G z_c' + g_c
          for (int c2 = 0; c2 < ncol; ++c2) {
            const SubDesc sc2 = p.subs[first_sub + c2];
            const uint8_t* mk = p.mask_pool + sc2.off_mask;
            double* gr = cvec(c2, 1);
            for (int64_t e = tid; e < zlen; e += TB)
              if (mk[e]) gr[e] += p.w_pool[sc2.off_w + nm + (int)(e % nm)];
          }
          __syncthreads();
          cgphase = 1; rho_acc = 0.0;
          set_column(0);
          start_projection();
          continue;
        }
        if (cgphase == 3) {                        // the certification solve of the column in hand has finished
          const int stc = (resid <= p.tol_ok && proj_ok) ? ((sd.pos < 0) ? 3 : 0) : 2;
          if (trial_is_answer) ans_trial |= 1ull << gcol;
          if (tid == 0) {
            p.status[sd.out_index] = stc;
            p.resid[sd.out_index] = resid;
            p.iters[sd.out_index] = it_keep + cg;
          }
          if (gcol + 1 < ncol) { set_column(gcol + 1); start_certification(); continue; }
          grp_done = true;
          break;
        }
        // cgphase 1 (first projection round) or 2: the projection solve of the column in hand has finished
        {
          const double* sans = trial_is_answer ? zt : zc;
          proj_ok = proj_ok && resid <= p.tol_ok;
          double* gr = cvec(gcol, 1);
          double* sk = cvec(gcol, 3);
          double rho_part = 0.0;
          for (int64_t e = tid; e < zlen; e += TB) {
            const int q = (int)(e % nm);
            const double sv_ = sans[e];
            const double dg = 1.0 / ((q < n) ? hx(q) : hu(q - n));
            rho_part = fma(sv_ * dg, sv_, rho_part);
            gr[e] = -dg * sv_;                     // residual update: the projected part of the gradient
            sk[e] = sv_;
          }
          rho_acc += tblock_sum(rho_part, red, tid);
        }
        if (gcol + 1 < ncol) { set_column(gcol + 1); start_projection(); continue; }
        const double rho_new = rho_acc;
        rho_acc = 0.0;
        if (cgphase == 1) {
          rho = rho0 = rho_new;
          for (int c2 = 0; c2 < ncol; ++c2) {
            const double* sk = cvec(c2, 3); double* gp = cvec(c2, 2);
            for (int64_t i = tid; i < zlen; i += TB) gp[i] = sk[i];
          }
          cgphase = 2;
        } else {
          rises = (rho_new >= rho) ? rises + 1 : 0;
          const double beta = (rho > 0.0) ? rho_new / rho : 0.0;
          for (int c2 = 0; c2 < ncol; ++c2) {
            const double* sk = cvec(c2, 3); double* gp = cvec(c2, 2);
            for (int64_t i = tid; i < zlen; i += TB) gp[i] = fma(beta, gp[i], sk[i]);
          }
          rho = rho_new;
        }
        __syncthreads();
        // stop when the projected gradient has dropped ten orders (ρ = ‖projected gradient‖² in the preconditioner's norm), or
        // when ρ no longer decreases (rounding level) — going on from there divides noise by noise
        const bool go_on = cg < 200 * ncol && proj_ok && rho > 1e-20 * rho0 && rho > 1e-30 && rises < 2;
        if (go_on) {
          hess(2, 3);
          double part = 0.0;
          for (int c2 = 0; c2 < ncol; ++c2) {
            const double* gp = cvec(c2, 2); const double* gq = cvec(c2, 3);
            for (int64_t i = tid; i < zlen; i += TB) part = fma(gp[i], gq[i], part);
          }
          const double pgp = tblock_sum(part, red, tid);
          if (pgp > 0.0) {
            const double alpha = rho / pgp;
            for (int c2 = 0; c2 < ncol; ++c2) {
              double* zg = cvec(c2, 0); double* gr = cvec(c2, 1); const double* gp = cvec(c2, 2); const double* gq = cvec(c2, 3);
              for (int64_t i = tid; i < zlen; i += TB) { zg[i] = fma(alpha, gp[i], zg[i]); gr[i] = fma(alpha, gq[i], gr[i]); }
            }
            __syncthreads();
            ++cg;
            set_column(0);
            start_projection();
            continue;
          }
        }
        // certify the iterates: their constraint residuals, evaluated from z itself; the steps may have drifted off E z = f by
        // a few 1e-13 — one more multiplier solve per column takes it back (a diagonal-metric correction of that size)
        cgphase = 3;
        set_column(0);
        start_certification();
      }
      }
      if (resid <= p.tol_ok) { if (status != 2 || !GW) status = 0; }
      else if (status == 0) status = 2;
    }
    // the answer goes to the output array (destination table: mask order or packed)
    if (GW && p.subs[first_sub].has_w >= 2 && p.objective != 1) {
      // dense Hessian / coupled group.  Completed: every column's certified iterate, statuses already written.  Not completed
      // (a column's own solve was infeasible or did not converge): the group is flagged as a whole with that status; the
      // columns solved so far return their diagonal-weight points, the column in hand its last point, the rest stay zero.
      __syncthreads();
      const int last = grp_done ? ncol - 1 : gcol;
      const double* zcur = trial_is_answer ? zt : zc;
      for (int c2 = 0; c2 <= last; ++c2) {
        const SubDesc sc2 = p.subs[first_sub + c2];
        const uint8_t* mk = p.mask_pool + sc2.off_mask;
        const int32_t* ds = p.dest_pool + sc2.off_dest;
        const double* vb = vecs + (int64_t)c2 * vstride_col + 3 * vlen;
        const double* zans = grp_done ? (((ans_trial >> c2) & 1ull) ? vb + zlen : vb) : (c2 < gcol ? vb + 2 * zlen : zcur);
        for (int64_t e = tid; e < zlen; e += TB)
          if (mk[e]) { const int d = ds[e]; if (d >= 0) p.out[d] = zans[e]; }
      }
      if (!grp_done && tid == 0) {
        for (int c2 = 0; c2 < ncol; ++c2) {
          const int64_t oi = p.subs[first_sub + c2].out_index;
          p.status[oi] = status ? status : 2;
          p.resid[oi] = resid;
          p.iters[oi] = iters;
        }
      }
      if (tid == 0 && p.dbg) for (int q = 0; q < 8; ++q) p.dbg[p.subs[first_sub].out_index * 8 + q] = tc[q];
      continue;
    }
    {
      const double* zans = trial_is_answer ? zt : zc;
      __syncthreads();
      for (int64_t e = tid; e < zlen; e += TB)
        if (mask[e]) { const int d = dest[e]; if (d >= 0) p.out[d] = zans[e]; }
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
  double* Lb = Yp + (int64_t)(MLDS ? 1 : 2) * NT * 256;
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
  tile_sweep<RS, TSZ, !MLDS>(Mb, NT, Yp, Lb, tl, HT, 0.0, tid);
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

template <bool MLDS, int WPE, bool GW, bool BIG = false>
static hipError_t launch_tile_v(const KernelParams& p, int grid, size_t lds_bytes, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&h2_column_tile_kernel<MLDS, WPE, GW, BIG>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((h2_column_tile_kernel<MLDS, WPE, GW, BIG>), dim3(grid), dim3(TB), lds_bytes, stream, p);
  return hipGetLastError();
}
// two_per_cu: the variant compiled for 4 waves per SIMD (≤ 128 VGPRs), two workgroups share a CU when LDS allows;
// general_weights: the build with the projected-CG loop (one workgroup per CU)
hipError_t launch_tile(const KernelParams& p, int grid, size_t lds_bytes, hipStream_t stream, bool mlds, bool two_per_cu,
                       bool general_weights, bool big) {
  if (big) {          // carve in the global workspace (p.big_ws): index sets beyond what LDS holds
    if (mlds || !p.big_ws) return hipErrorInvalidValue;
    return general_weights ? launch_tile_v<false, 2, true, true>(p, grid, lds_bytes, stream) : launch_tile_v<false, 2, false, true>(p, grid, lds_bytes, stream);
  }
  if (general_weights) {
    if (two_per_cu && mlds) return launch_tile_v<true, 4, true>(p, grid, lds_bytes, stream);      // small columns: two workgroups per CU
    return mlds ? launch_tile_v<true, 2, true>(p, grid, lds_bytes, stream) : launch_tile_v<false, 2, true>(p, grid, lds_bytes, stream);
  }
  if (mlds) return two_per_cu ? launch_tile_v<true, 4, false>(p, grid, lds_bytes, stream) : launch_tile_v<true, 2, false>(p, grid, lds_bytes, stream);
  return two_per_cu ? launch_tile_v<false, 4, false>(p, grid, lds_bytes, stream) : launch_tile_v<false, 2, false>(p, grid, lds_bytes, stream);
}

// d_A, d_out: device n×n row-major; d_ws: device scratch of tile_ht(nt)·256 doubles (global variant)
hipError_t launch_tile_invert(const double* d_A, int n, double* d_ws, double* d_out, bool mlds, hipStream_t stream) {
  const int nt = tile_nt(n), ht = tile_ht(nt);
  const size_t lds = (size_t)((mlds ? 1 : 2) * nt * 256 + 512 + (mlds ? ht * kTileLdsTile : 0)) * 8 + (size_t)ht * 4 + 64;
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
