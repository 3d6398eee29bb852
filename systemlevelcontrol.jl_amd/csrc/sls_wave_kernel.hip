// sls_wave_kernel.hip — v2 "one wavefront per subproblem" kernel of the H2 column solve (gfx950).
//
// Same mathematics as the general kernel (sls_kernels.hip header / DESIGN.md §3), different machine mapping,
// chosen from measurements on MI355X (profiles/r01_fp64_microbench.txt):
//   * v_mfma_f64_16x16x4 issues every 143 cycles per wave with a 196-cycle dependent latency and tops out at the
//     VALU FP64 rate, so for ñx ≤ 64 blocks whose work is a chain of dependent rank-1 updates it buys nothing;
//   * workgroup barriers cost ~0.5 µs per pivot in the general kernel.
// So: ONE 64-lane wave owns a subproblem, no workgroup barrier anywhere.  The ñx×ñx pivot block lives in
// REGISTERS: lane (h, j) = (lane / NPL, lane % NPL) holds column j of rows i = HS·r + h, r = 0..RPL-1
// (HS = 64/NPL row groups, so all 64 lanes work even when ñx ≤ 32 or ≤ 16).  The block inverse is an in-register
// Gauss–Jordan (SPD, no pivoting): per pivot the pivot row and the pivot column go once through LDS and every lane
// reads back the column entries of its own rows (broadcast ds_read2_b64), i.e. ≈1.5 LDS + 1 v_fma_f64 per register
// row instead of 2×v_readlane + v_fma_f64.  Ã stays sparse: per-lane row/column lists in LDS; the two sparse
// products Ã·Q·Ãᵀ are LDS gathers that use the symmetry of Q.
// P_k is streamed to an L2-resident workspace in the register layout ([k][r][lane], 512-B coalesced rows) and
// re-read by the two substitution sweeps of each refinement pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "sls_device.h"
#include "sls_wave_util.h"

namespace sls {

#ifndef SLS_TILED_GJ
#define SLS_TILED_GJ 1
#endif
#ifndef SLS_WAVE_PACKED_P
#define SLS_WAVE_PACKED_P 1     // P_k of the NPL = 32 classes stored as its upper half where its symmetry allows (0: always the full register image)
#endif
#ifndef SLS_TILED_GJ_WAVE
#define SLS_TILED_GJ_WAVE 1     // the same in the one-wave (throughput) kernel
#endif
#ifndef SLS_TILED_GJ_WAVE64
#define SLS_TILED_GJ_WAVE64 1
#endif
#ifndef SLS_GJ_NR
#define SLS_GJ_NR 1          // Newton steps on the v_rcp_f64 seed of every pivot reciprocal.  The seed is good to ≈1e-8: with 0 steps the
                           // multiplier iteration needs 3–5 passes instead of 2; 1 step gives the same pass counts and residuals as 2 on
                           // every test workload (tools/nr_scan.sh) and is ≈1 % faster
#endif

// Gauss–Jordan on an 8×8 LANE GRID (NPL = 32 classes).  Lane (a, b) = (lane >> 3, lane & 7) holds the TR×TR tile
// {rows a + 8·ri} × {columns b + 8·cj}.  A pivot then needs TR pivot-column values (ds_swizzle broadcast inside the 8-lane
// group: 2·TR instructions) and TR pivot-row values (ds_bpermute from lane (pa, b): 2·TR) for TR² FMAs — 12 cross-lane
// operations + 9 FMAs at ñx ≤ 24 against 26 + 12 in the column layout — and, unlike there, only the 2·TR−1 tile entries in
// the NEXT pivot's row slot and column slot have to be updated before its cross-lane reads can issue: they are fetched
// one step ahead and the rest of the rank-1 update runs in their shadow.  The block changes layout through the (private)
// LDS image on the way in and out.  What is left per pivot is the dependent chain of the next reciprocal (§5 of DESIGN.md).
// Pivot row through LDS instead of 2·TR ds_bpermute: the eight owner lanes (a = q mod 8, known at compile time) store their
// TR values with EXEC set by two scalar moves (no divergent region, see sls_twisted4_kernel.hip: store_row8_*), every lane reads
// the record of its lane-grid column back with ds_read_b128.  ds_bpermute_b32 occupies the CU's LDS pipe for ≈6 cycles per
// instruction whatever the wave count (profiles/r01_lds_xlane_microbench.txt); with eight waves per CU the 8 + 8 cross-lane
// operations of a pivot kept that pipe ≈85 % busy in the throughput regime (chain-4096, round 3).  LDS operations of a wave
// complete in order: the reads below see the stores without a wait, and the next pivot's stores cannot overtake them.
#ifndef SLS_GJ_ROW_LDS
#define SLS_GJ_ROW_LDS 0        // measured: no gain at eight waves per CU (chain-4096 1.52 ms either way), +4 % on a lone wave (two-wave kernel,
#endif                        // README 0.1209 → 0.1265 ms: the store → load round trip is longer than a ds_bpermute) — kept as an A/B switch
#ifndef SLS_GJ_ASM_MASKS
#define SLS_GJ_ASM_MASKS 0      // lane-specific moves of a pivot step as EXEC-masked inline assembly (gj_fix_column / gj_own_row*)
#endif
typedef double gj_d2_t __attribute__((ext_vector_type(2)));
template <int A>
__device__ __forceinline__ void gj_row8_write4(unsigned addr, double v0, double v1, double v2, double v3) {
  const gj_d2_t q0 = {v0, v1}, q1 = {v2, v3};
  asm volatile("s_mov_b64 exec, 0xff\n\ts_lshl_b64 exec, exec, %[sh]\n\t"
               "ds_write_b128 %[a], %[q0]\n\tds_write_b128 %[a], %[q1] offset:16\n\t"
               "s_mov_b64 exec, -1"
               :: [sh] "n"(8 * A), [a] "v"(addr), [q0] "v"(q0), [q1] "v"(q1) : "memory", "scc");
}
template <int A>
__device__ __forceinline__ void gj_row8_write3(unsigned addr, double v0, double v1, double v2) {
  const gj_d2_t q0 = {v0, v1};
  asm volatile("s_mov_b64 exec, 0xff\n\ts_lshl_b64 exec, exec, %[sh]\n\t"
               "ds_write_b128 %[a], %[q0]\n\tds_write_b64 %[a], %[q1] offset:16\n\t"
               "s_mov_b64 exec, -1"
               :: [sh] "n"(8 * A), [a] "v"(addr), [q0] "v"(q0), [q1] "v"(v2) : "memory", "scc");
}

// (s_lshl_b64 writes SCC: every block that shifts EXEC lists "scc" as clobbered — without it the compiler kept a compare result
//  alive across the block and branched on the shift's SCC.)
// The two places where a pivot step treats lanes differently, as EXEC-masked moves between two scalar moves (the lane sets are
// compile-time constants): the compiler turned `own ? a : b` selects back into a divergent region around the FMAs of the pivot
// row, with the lane masks parked in VGPR lanes (v_writelane / v_readlane per use).
//   gj_fix_column<pa>:  t ← v in the lanes of lane-grid column pa (bits pa, pa+8, …)
//   gj_own_row{3,4}<pa, ps>: the tile row of the pivot in the eight lanes of lane-grid row pa ← tj[·], its pivot entry ← d
template <int PA>
__device__ __forceinline__ void gj_fix_column(double& t, double v) {
  asm("s_mov_b32 exec_lo, %[m]\n\ts_mov_b32 exec_hi, %[m]\n\tv_mov_b64 %[t], %[v]\n\ts_mov_b64 exec, -1"
      : [t] "+v"(t) : [m] "n"(0x01010101u << PA), [v] "v"(v));
}
#define SLS_GJ_OWN_ROW4(RP)                                                                                              \
  asm("s_mov_b64 exec, 0xff\n\ts_lshl_b64 exec, exec, %[sh]\n\t"                                                        \
      "v_mov_b64 %[r0], %[t0]\n\tv_mov_b64 %[r1], %[t1]\n\tv_mov_b64 %[r2], %[t2]\n\tv_mov_b64 %[r3], %[t3]\n\t"        \
      "s_mov_b64 exec, 1\n\ts_lshl_b64 exec, exec, %[sp]\n\tv_mov_b64 %[" RP "], %[d]\n\ts_mov_b64 exec, -1"             \
      : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3)                                                       \
      : [sh] "n"(8 * PA), [sp] "n"(9 * PA), [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2), [t3] "v"(t3), [d] "v"(d) : "scc")
#define SLS_GJ_OWN_ROW3(RP)                                                                                              \
  asm("s_mov_b64 exec, 0xff\n\ts_lshl_b64 exec, exec, %[sh]\n\t"                                                        \
      "v_mov_b64 %[r0], %[t0]\n\tv_mov_b64 %[r1], %[t1]\n\tv_mov_b64 %[r2], %[t2]\n\t"                                  \
      "s_mov_b64 exec, 1\n\ts_lshl_b64 exec, exec, %[sp]\n\tv_mov_b64 %[" RP "], %[d]\n\ts_mov_b64 exec, -1"             \
      : [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2)                                                                      \
      : [sh] "n"(8 * PA), [sp] "n"(9 * PA), [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2), [d] "v"(d) : "scc")
template <int PA, int PS>
__device__ __forceinline__ void gj_own_row4(double& r0, double& r1, double& r2, double& r3, double t0, double t1, double t2, double t3, double d) {
  if constexpr (PS == 0) SLS_GJ_OWN_ROW4("r0"); else if constexpr (PS == 1) SLS_GJ_OWN_ROW4("r1");
  else if constexpr (PS == 2) SLS_GJ_OWN_ROW4("r2"); else SLS_GJ_OWN_ROW4("r3");
}
template <int PA, int PS>
__device__ __forceinline__ void gj_own_row3(double& r0, double& r1, double& r2, double t0, double t1, double t2, double d) {
  if constexpr (PS == 0) SLS_GJ_OWN_ROW3("r0"); else if constexpr (PS == 1) SLS_GJ_OWN_ROW3("r1"); else SLS_GJ_OWN_ROW3("r2");
}
#undef SLS_GJ_OWN_ROW4
#undef SLS_GJ_OWN_ROW3

template <int NPL, int RPL, int LDT>
__device__ __forceinline__ void gauss_jordan_tiled(double (&M)[RPL], double* mat, const int lane, const int n) {
  static_assert(NPL == 32 || NPL == 64, "8×8 lane grid: written for the NPL = 32 and 64 classes");
  constexpr int HS = 64 / NPL, NP = HS * RPL;
  constexpr int TR = (NP + 7) / 8;
  constexpr bool ROWLDS = SLS_GJ_ROW_LDS != 0 && (TR == 3 || TR == 4);
  constexpr bool ASMM = SLS_GJ_ASM_MASKS != 0 && (TR == 3 || TR == 4);
  const int h = lane / NPL, j = lane % NPL;
  // opaque copies: otherwise every per-pivot predicate (pv < n, ta == pa, tb == pa) is hoisted out of the block loop as a
  // 64-bit lane mask, spilled into VGPR lanes and fetched back with two v_readlane per use — recomputing costs one compare
  int ta = lane >> 3, tb = lane & 7, nn = __builtin_amdgcn_readfirstlane(n);
  asm volatile("" : "+v"(ta), "+v"(tb), "+s"(nn));
#pragma unroll
  for (int r = 0; r < RPL; ++r) mat[(HS * r + h) * LDT + j] = M[r];
  WSYNC();
  double Tt[TR * TR];
#pragma unroll
  for (int ri = 0; ri < TR; ++ri) {
#pragma unroll
    for (int cj = 0; cj < TR; ++cj) {
      const int i = ta + 8 * ri;
      Tt[ri * TR + cj] = (i < NP) ? mat[i * LDT + tb + 8 * cj] : 0.0;
    }
  }
  WSYNC();
  // row records: 32 B per lane-grid column at the (16-byte aligned) start of the image, which is idle until the tiles go back
  // (every caller's image starts on a 16-byte boundary of the LDS carve)
  const gj_d2_t* rrec = reinterpret_cast<const gj_d2_t*>(__builtin_assume_aligned(mat, 16)) + 2 * tb;
  const unsigned rrec_addr = (unsigned)(uintptr_t)rrec;
  double dnext = fast_rcp(readlane_f64(Tt[0], 0));
  double col[TR], row[TR];
  auto fetch = [&](auto q_c) {
    constexpr int q = decltype(q_c)::value;
    constexpr int qa = q % 8, qs = q / 8;
    constexpr int pattern = 0x18 | (qa << 5);               // lane' = (lane & 0x18) | qa inside each 32-lane half
#pragma unroll
    for (int ri = 0; ri < TR; ++ri) {
      const double v = Tt[ri * TR + qs];
      col[ri] = __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), pattern),
                                 __builtin_amdgcn_ds_swizzle(__double2loint(v), pattern));
    }
    if constexpr (ROWLDS) {
      if constexpr (TR == 4) gj_row8_write4<qa>(rrec_addr, Tt[qs * TR + 0], Tt[qs * TR + 1], Tt[qs * TR + 2], Tt[qs * TR + 3]);
      else gj_row8_write3<qa>(rrec_addr, Tt[qs * TR + 0], Tt[qs * TR + 1], Tt[qs * TR + 2]);
      const gj_d2_t r01 = rrec[0];
      row[0] = r01[0]; row[1] = r01[1];
      if constexpr (TR == 4) { const gj_d2_t r23 = rrec[1]; row[2] = r23[0]; row[3] = r23[1]; }
      else row[2] = reinterpret_cast<const double*>(rrec)[2];
    } else {
      const int src = (qa * 8 + tb) << 2;
#pragma unroll
      for (int cj = 0; cj < TR; ++cj) {
        const double v = Tt[qs * TR + cj];
        row[cj] = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(v)),
                                   __builtin_amdgcn_ds_bpermute(src, __double2loint(v)));
      }
    }
  };
  fetch(std::integral_constant<int, 0>{});
  static_for<NP>([&](auto pv_c) {
    constexpr int pv = decltype(pv_c)::value;
    constexpr int pa = pv % 8, ps = pv / 8;                 // lane-grid coordinate and register slot of row/column pv
    if (pv < nn) {
      const double d = dnext;
      constexpr bool have_next = pv + 1 < NP;
      constexpr int na = (pv + 1) % 8, ns = have_next ? (pv + 1) / 8 : ps;
      double xr = 0.0;
      if constexpr (have_next) {                            // next pivot predicted from three entries of the not yet updated block
        const double a_nn = readlane_f64(Tt[ns * TR + ns], na * 8 + na);
        const double a_pn = readlane_f64(Tt[ps * TR + ns], pa * 8 + na);   // = M[p+1][p] too: the unreduced part stays symmetric
        const double pn = __builtin_fma(-(a_pn * d), a_pn, a_nn);          // (to rounding, which a predicted pivot does not care about)
        xr = __builtin_amdgcn_rcp(pn);
        if (SLS_GJ_NR >= 1) xr = __builtin_fma(__builtin_fma(-pn, xr, 1.0), xr, xr);
        if (SLS_GJ_NR >= 2) xr = __builtin_fma(__builtin_fma(-pn, xr, 1.0), xr, xr);
      }
      double c0[TR], tj[TR], tfix[TR];
#pragma unroll
      for (int ri = 0; ri < TR; ++ri) c0[ri] = col[ri];
#pragma unroll
      for (int cj = 0; cj < TR; ++cj) {
        tj[cj] = row[cj] * d;
        if constexpr (ASMM) tfix[cj] = tj[cj];
        else tfix[cj] = (cj == ps && tb == pa) ? (1.0 + d) : tj[cj];
      }
      if constexpr (ASMM) gj_fix_column<pa>(tfix[ps], 1.0 + d);
      __builtin_amdgcn_sched_barrier(0);
      // phase A: what the next pivot's row/column reads depend on
#pragma unroll
      for (int ri = 0; ri < TR; ++ri) {
#pragma unroll
        for (int cj = 0; cj < TR; ++cj)
          if (ri == ns || cj == ns || ri == ps) Tt[ri * TR + cj] = __builtin_fma(-c0[ri], tfix[cj], Tt[ri * TR + cj]);
      }
      if constexpr (ASMM) {
        if constexpr (TR == 4) gj_own_row4<pa, ps>(Tt[ps * TR + 0], Tt[ps * TR + 1], Tt[ps * TR + 2], Tt[ps * TR + 3], tj[0], tj[1], tj[2], tj[3], d);
        else gj_own_row3<pa, ps>(Tt[ps * TR + 0], Tt[ps * TR + 1], Tt[ps * TR + 2], tj[0], tj[1], tj[2], d);
      } else {
        if (ta == pa) {
#pragma unroll
          for (int cj = 0; cj < TR; ++cj) Tt[ps * TR + cj] = (cj == ps && tb == pa) ? d : tj[cj];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (have_next) fetch(std::integral_constant<int, have_next ? pv + 1 : 0>{});
      __builtin_amdgcn_sched_barrier(0);
      // phase B: the rest of the rank-1 update, in the shadow of those reads
#pragma unroll
      for (int ri = 0; ri < TR; ++ri) {
#pragma unroll
        for (int cj = 0; cj < TR; ++cj)
          if (!(ri == ns || cj == ns || ri == ps)) Tt[ri * TR + cj] = __builtin_fma(-c0[ri], tfix[cj], Tt[ri * TR + cj]);
      }
      if constexpr (have_next) dnext = xr;
    }
  });
  if constexpr (ROWLDS) WSYNC();
#pragma unroll
  for (int ri = 0; ri < TR; ++ri) {
#pragma unroll
    for (int cj = 0; cj < TR; ++cj) {
      const int i = ta + 8 * ri;
      if (i < NP) mat[i * LDT + tb + 8 * cj] = Tt[ri * TR + cj];
    }
  }
  WSYNC();
#pragma unroll
  for (int r = 0; r < RPL; ++r) M[r] = mat[(HS * r + h) * LDT + j];
  WSYNC();
}

// VG = true: λ and r/q/Δλ (the two T-sized vectors) live in a per-workgroup global workspace (L2-resident) instead of
// LDS — the throughput-regime variant: LDS drops from ≈39 KB to ≈18 KB per wave (8 resident waves per CU instead of 4);
// P_k is fetched from global memory in every sweep step anyway, so the extra row per step adds no latency chain.
template <int NPL, int RPL, bool VG, bool SON = false>
__device__ __forceinline__ void wave_solve_column(const KernelParams& p, const SubDesc& sd, double* __restrict__ fac,
                                                  double* __restrict__ gvec, unsigned char* lds_raw) {
  static_assert(!SON || VG, "the sum-of-norms build keeps its vectors in the global workspace");
  constexpr int HS = 64 / NPL;          // row groups
  constexpr int NP = HS * RPL;          // rows held (≥ n)
  constexpr int LDM = NPL + 1;          // padded leading dimension of the LDS matrix image
  const int lane = threadIdx.x;
  const int h = lane / NPL, j = lane % NPL;
  const int n = sd.n, m = sd.m, nm = n + m, T = p.T;
  const int MC = p.w_mcap;
  const int capA = p.w_nzA, capAc = p.w_nzAc, capB = p.w_nzB, capBc = p.w_nzBc;

  // ---- LDS carve (must match wave_kernel_lds_bytes): fixed-size hot arrays first, so that their offsets are
  //      compile-time immediates of the ds_read/ds_write instructions; run-time sized arrays after ----
  double* dp = reinterpret_cast<double*>(lds_raw);
  double* mat = dp;    dp += NP * LDM;       // matrix image of the sparse products; doubles as `us` in residual passes
  double* tmp = dp;    dp += NPL;
  double* tmp2 = dp;   dp += NPL;
  double* hx = dp;     dp += NPL;
  double* gx = dp;     dp += NPL;
  double* wl = dp;     dp += NPL;
  double* hu = dp;     dp += 64;
  double* gu = dp;     dp += 64;
  double* wul = dp;    dp += 64;
  int32_t* sx = reinterpret_cast<int32_t*>(dp);
  int32_t* su = sx + NPL;
  dp += (NPL + 64) / 2;
  double* lam; double* rq;
  if constexpr (VG) { lam = gvec; rq = gvec + (T + 1) * NPL; }
  else { lam = dp; dp += (T + 1) * NPL; rq = dp; dp += (T + 1) * NPL; }
  double* Bd = dp;     dp += NPL * MC;
  // u_t of the current residual pass, [t][q]: lives in the matrix image when it fits there (the image is only used by
  // the block build, never during a residual pass), else in its own array
  double* us = mat;
  if (T * MC > NP * LDM) { us = dp; dp += T * MC; }
  double* arow_v = dp; dp += capA * NPL;
  double* acol_v = dp; dp += capAc * NPL;
  double* brow_v = dp; dp += capB * NPL;
  double* bcol_v = dp; dp += capBc * 64;
  int32_t* ip = reinterpret_cast<int32_t*>(dp);
  int32_t* arow_c = ip; ip += capA * NPL;
  int32_t* acol_c = ip; ip += capAc * NPL;
  int32_t* brow_c = ip; ip += capB * NPL;
  int32_t* bcol_c = ip; ip += capBc * 64;
  uint8_t* mask = reinterpret_cast<uint8_t*>(ip);
  // sum-of-norms build (SLS_SOLVE_SUM_OF_NORMS): ADMM vectors per (t, variable) in the wave's global workspace — the linear
  // term of the projection in progress, y, the scaled multiplier u, and v = over-relaxed W z + u of the step being taken
  double* sgl = nullptr; double* syv = nullptr; double* suv = nullptr; double* svv = nullptr;
  double* sgc = nullptr; double* sFp = nullptr; double* sgp = nullptr; double* sdF = nullptr; double* sdG = nullptr;   // Anderson acceleration
  constexpr int AAM = 5;                 // memory of the acceleration
  const int64_t L1 = (int64_t)T * nm, L2 = 2 * L1;
  if constexpr (SON) {
    sgl = gvec + 2 * (T + 1) * NPL; syv = sgl + L1; suv = syv + L1; svv = suv + L1;
    sgc = svv + L1; sFp = sgc + L2; sgp = sFp + L2; sdF = sgp + L2; sdG = sdF + AAM * L2;
    for (int i = lane; i < 3 * T * nm; i += 64) sgl[i] = 0.0;          // linear term, y, u
  }

  const int32_t* dest = p.dest_pool + sd.off_dest;
  // phase cycle counters (diagnostics only; s_memtime is a scalar op, a handful per subproblem)
  unsigned long long tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
  auto lap = [&](int slot) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tc[slot] += now - tlast; tlast = now; };

  WSYNC();
  // ---- stage index sets, weights, masks; clear the lists ----
  if (lane < NPL) {
    sx[lane] = (lane < n) ? p.idx_pool[sd.off_sx + lane] : 0x7fffffff;
    hx[lane] = (lane < n) ? (sd.has_w ? p.w_pool[sd.off_w + lane] : 1.0) : 0.0;
    gx[lane] = (lane < n && sd.has_w) ? p.w_pool[sd.off_w + nm + lane] : 0.0;
    tmp[lane] = 0.0; tmp2[lane] = 0.0; wl[lane] = 0.0;
  }
  su[lane] = (lane < m) ? p.idx_pool[sd.off_su + lane] : 0x7fffffff;
  hu[lane] = (lane < m) ? (sd.has_w ? p.w_pool[sd.off_w + n + lane] : 1.0) : 0.0;
  gu[lane] = (lane < m && sd.has_w) ? p.w_pool[sd.off_w + nm + n + lane] : 0.0;
  wul[lane] = 0.0;
  for (int i = lane; i < T * nm; i += 64) mask[i] = p.mask_pool[sd.off_mask + i];
  for (int i = lane; i < NPL * MC; i += 64) Bd[i] = 0.0;
  for (int i = lane; i < capA * NPL; i += 64) { arow_v[i] = 0.0; arow_c[i] = 0; }
  for (int i = lane; i < capAc * NPL; i += 64) { acol_v[i] = 0.0; acol_c[i] = 0; }
  for (int i = lane; i < capB * NPL; i += 64) { brow_v[i] = 0.0; brow_c[i] = 0; }
  for (int i = lane; i < capBc * 64; i += 64) { bcol_v[i] = 0.0; bcol_c[i] = 0; }
  for (int i = lane; i < (T + 1) * NPL; i += 64) { lam[i] = 0.0; rq[i] = 0.0; }
  WSYNC();

  // ---- gather Ã (row and column lists per lane) and B̃ (row lists, column lists, dense image) from the
  //      shared operator: coalescing is per CSR row; the operator itself is read once per subproblem ----
  int cntA = 0, cntAc = 0, cntB = 0, cntBc = 0;
  if (lane < n) {
    const int g = sx[lane];
    for (int e = p.A_rowptr[g]; e < p.A_rowptr[g + 1]; ++e) {
      const double v = p.A_val[e];
      const int loc = (v != 0.0) ? wbsearch(sx, n, p.A_colidx[e]) : -1;
      if (loc >= 0 && cntA < capA) { arow_c[cntA * NPL + lane] = loc; arow_v[cntA * NPL + lane] = v; ++cntA; }
    }
    for (int e = p.At_rowptr[g]; e < p.At_rowptr[g + 1]; ++e) {
      const double v = p.At_val[e];
      const int loc = (v != 0.0) ? wbsearch(sx, n, p.At_colidx[e]) : -1;
      if (loc >= 0 && cntAc < capAc) { acol_c[cntAc * NPL + lane] = loc; acol_v[cntAc * NPL + lane] = v; ++cntAc; }
    }
    for (int e = p.B_rowptr[g]; e < p.B_rowptr[g + 1]; ++e) {
      const double v = p.B_val[e];
      const int loc = (v != 0.0) ? wbsearch(su, m, p.B_colidx[e]) : -1;
      if (loc >= 0 && cntB < capB) { brow_c[cntB * NPL + lane] = loc; brow_v[cntB * NPL + lane] = v; Bd[lane * MC + loc] = v; ++cntB; }
    }
  }
  if (lane < m) {
    const int g = su[lane];
    for (int e = p.Bt_rowptr[g]; e < p.Bt_rowptr[g + 1]; ++e) {
      const double v = p.Bt_val[e];
      const int loc = (v != 0.0) ? wbsearch(sx, n, p.Bt_colidx[e]) : -1;
      if (loc >= 0 && cntBc < capBc) { bcol_c[cntBc * 64 + lane] = loc; bcol_v[cntBc * 64 + lane] = v; ++cntBc; }
    }
  }
  const int nzA = wave_max_i32(cntA), nzAc = wave_max_i32(cntAc), nzB = wave_max_i32(cntB), nzBc = wave_max_i32(cntBc);
  WSYNC();
  // The first KR entries of lane j's row list and column list of Ã live in registers for the rest of the solve
  // (covers tri-diagonal and 5-point plants entirely): every gather site otherwise pays an extra dependent LDS
  // round trip just to learn its index.  Longer lists continue from LDS.
  constexpr int KR = 4;
  int arc[KR], acc_[KR];
  double arv[KR], acv[KR];
#pragma unroll
  for (int e = 0; e < KR; ++e) {
    const bool okr = e < capA, okc = e < capAc;
    arc[e] = okr ? arow_c[e * NPL + j] : 0;  arv[e] = okr ? arow_v[e * NPL + j] : 0.0;
    acc_[e] = okc ? acol_c[e * NPL + j] : 0; acv[e] = okc ? acol_v[e * NPL + j] : 0.0;
  }
  auto dotA_row = [&](const double* vec) -> double {      // Σ_e Ã[j][c_e]·vec[c_e]
    // entries ≥ nzA of the register-cached list are (0, 0.0): no branches, the four gathers issue together
    static_assert(KR == 4, "written for four cached entries");
    const double v0 = vec[arc[0]], v1 = vec[arc[1]], v2 = vec[arc[2]], v3 = vec[arc[3]];
    double a = __builtin_fma(arv[1], v1, arv[0] * v0) + __builtin_fma(arv[3], v3, arv[2] * v2);
    for (int e = KR; e < nzA; ++e) a = __builtin_fma(arow_v[e * NPL + j], vec[arow_c[e * NPL + j]], a);
    return a;
  };
  auto dotA_col = [&](const double* vec) -> double {      // Σ_e Ã[c_e][j]·vec[c_e]
    static_assert(KR == 4, "written for four cached entries");
    const double v0 = vec[acc_[0]], v1 = vec[acc_[1]], v2 = vec[acc_[2]], v3 = vec[acc_[3]];
    double a = __builtin_fma(acv[1], v1, acv[0] * v0) + __builtin_fma(acv[3], v3, acv[2] * v2);
    for (int e = KR; e < nzAc; ++e) a = __builtin_fma(acol_v[e * NPL + j], vec[acol_c[e * NPL + j]], a);
    return a;
  };

  // ---- Tikhonov shift: relative to the largest possible Schur diagonal ----
  double sc = 0.0;
  if (lane < n) {
    sc = hx[lane];
    for (int e = 0; e < nzA; ++e) { const double v = arow_v[e * NPL + lane]; sc = __builtin_fma(v * v, hx[arow_c[e * NPL + lane]], sc); }
    for (int e = 0; e < nzB; ++e) { const double v = brow_v[e * NPL + lane]; sc = __builtin_fma(v * v, hu[brow_c[e * NPL + lane]], sc); }
  }
  const double scmax = wave_max_f64(sc);
  double delta = p.delta_rel * scmax;      // set per attempt below

  // ---- residual pass: r = f − E z(λ) into rq, z to the output array, returns ‖r‖∞ ----
  // Given λ every time step is independent, so there is no serial chain here: phase 1 evaluates
  // x_t = Wx_t(λ_t − Ãᵀλ_{t+1} − g_x), u_t = Wu_t(−B̃ᵀλ_{t+1} − g_u) for HS time steps per instruction (lane group h
  // takes t = HS·it + h) and parks x_t in rq[t]; phase 2 walks t downwards and overwrites rq[t] with
  // r_t = f_t − x_t + Ãx_{t−1} + B̃u_{t−1} (r_{t+1} has already consumed x_t by then).
  // z(λ): the SAME expressions feed the residual and, at the end, the output array (bitwise identical values)
  auto x_of = [&](int t) -> double {            // lane (h, j), j < n
    const double* l1 = lam + (t + 1) * NPL;
    const double l0 = lam[t * NPL + j];
    const uint8_t mk = mask[t * nm + j];
    const double acc = dotA_col(l1);
    const double gterm = SON ? sgl[t * nm + j] : gx[j];
    const double v = hx[j] * (l0 - acc - gterm);
    return mk ? v : 0.0;
  };
  // the ñu inputs use few lanes (chain-4096: 10 of 64): lanes are (time slot, input) pairs, 64/MP time steps per instruction
  int lgMP = 0;
  while ((1 << lgMP) < m) ++lgMP;
  const int uq = lane & ((1 << lgMP) - 1), uts = lane >> lgMP, NTS = 64 >> lgMP;
  bool xu_valid = false;                         // `us` holds u(λ) of the current λ (set by the residual pass)
  auto u_of = [&](int t) -> double {            // input uq at time t
    const double* l1 = lam + (t + 1) * NPL;
    double acc = 0.0;
    for (int e = 0; e < nzBc; ++e) acc = __builtin_fma(bcol_v[e * 64 + uq], l1[bcol_c[e * 64 + uq]], acc);
    const double gterm = SON ? ((uq < m) ? sgl[t * nm + n + uq] : 0.0) : gu[uq];
    return mask[t * nm + n + uq] ? hu[uq] * (-acc - gterm) : 0.0;
  };
  // one pass over the destination table at the very end (the residual passes touch no global memory); destinations are
  // fetched eight at a time so that their global loads overlap, u comes from the image the last residual pass left
  auto output_pass = [&]() {
    constexpr int CH = 8;
    if (j < n) {
      for (int t0 = h; t0 < T; t0 += HS * CH) {
        int dd[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int t = t0 + q * HS; dd[q] = (t < T && mask[t * nm + j]) ? dest[t * nm + j] : -1; }
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int t = t0 + q * HS; if (dd[q] >= 0) p.out[dd[q]] = x_of(t); }
      }
    }
    if (uq < m) {
      for (int t0 = uts; t0 < T; t0 += NTS * CH) {
        int dd[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int t = t0 + q * NTS; dd[q] = (t < T && mask[t * nm + n + uq]) ? dest[t * nm + n + uq] : -1; }
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int t = t0 + q * NTS; if (dd[q] >= 0) p.out[dd[q]] = xu_valid ? us[t * MC + uq] : u_of(t); }
      }
    }
  };
  auto residual_pass = [&]() -> double {
    double rmax = 0.0;
    const bool live = j < n;
#pragma unroll 2
    for (int t0 = 0; t0 <= T; t0 += HS) {
      const int t = t0 + h;
      if (t <= T) rq[t * NPL + j] = (t < T && live) ? x_of(t) : 0.0;            // x_T ≡ 0
    }
    if (uq < m) {
#pragma unroll 2
      for (int t = uts; t < T; t += NTS) us[t * MC + uq] = u_of(t);
    }
    xu_valid = true;
    WSYNC();
    const int nround = (T + HS) / HS;                 // covers t = 0..T
#pragma unroll 2
    for (int it = nround - 1; it >= 0; --it) {
      const int t = it * HS + h;
      if (t <= T && live) {
        double acc = (t == 0 && j == sd.pos) ? 1.0 : 0.0;       // f_0 = e_pos
        acc -= rq[t * NPL + j];
        if (t >= 1) {
          const double* xp = rq + (t - 1) * NPL;
          const double* up = us + (t - 1) * MC;
          acc += dotA_row(xp);
          for (int e = 0; e < nzB; ++e) acc = __builtin_fma(brow_v[e * NPL + j], up[brow_c[e * NPL + j]], acc);
        }
        rmax = resid_max(rmax, acc);
        // the other lane groups of this round still read x_{t−1} = rq[t−1] … rq[t] is only read as "own element"
        // by this lane and as x_t by the group handling t+1, which belongs to this same instruction (loads of
        // every group issue before the store below) or to an earlier round.
        rq[t * NPL + j] = acc;
      }
    }
    WSYNC();
    return wave_max_f64(rmax);
  };

  // ACC[r] += Σ_{e < min(nzA, KR)} arv[e]·image[row r][arc[e]]: one straight-line body per entry count (nzA is wave-uniform), so
  // that the reads of all cached entries are in flight together instead of one branch + wait per entry
  auto cached_product = [&](double (&ACC)[RPL]) {
    auto body = [&](auto ne_c) {
      constexpr int NE = decltype(ne_c)::value;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const double* row = mat + (HS * r + h) * LDM;
        double a = ACC[r];
#pragma unroll
        for (int e = 0; e < NE; ++e) a = __builtin_fma(arv[e], row[arc[e]], a);
        ACC[r] = a;
      }
    };
    if constexpr (RPL > 16) {                     // NPL = 64 classes: 40–64 rows per lane, four unrolled copies only cost (measured +4 %)
#pragma unroll
      for (int e = 0; e < KR; ++e) {
        if (e < nzA) {
#pragma unroll
          for (int r = 0; r < RPL; ++r) ACC[r] = __builtin_fma(arv[e], mat[(HS * r + h) * LDM + arc[e]], ACC[r]);
        }
      }
    } else {
      const int ne = nzA < KR ? nzA : KR;
      if (ne == 3) body(std::integral_constant<int, 3>{});
      else if (ne == 4) body(std::integral_constant<int, 4>{});
      else if (ne == 2) body(std::integral_constant<int, 2>{});
      else if (ne == 1) body(std::integral_constant<int, 1>{});
    }
  };

  lap(0);                       // setup: staging + operator gather
  // Two attempts at most.  The first uses the small shift δ_first (1e-15 of the Schur scale): on chain-like columns the
  // singular directions of S are structurally isolated, FP64 elimination survives it and ONE pass reaches 1e-14 (the
  // first-pass residual is δ·‖λ‖) — a third less work and half the P_k traffic (chain-4096: 2.55 → 1.9 ms).  Where
  // round-off gets through a shift that small (grid-like columns: eps/δ), the attempt does not end converged below
  // `tol`; the column is then redone from λ = 0 with the robust shift δ_rel (1e-12), whose result is final.
  double resid = 0.0;
  int iters = 0, status = 0, iters_first = 0;
  // Light size classes only (ñx ≤ 32): there a wasted first try is cheap and the chains live there; the 64-lane classes (ñx up to
  // 64: the grid plant's boundary columns, two thirds of them infeasible as specified) would pay every infeasible column twice.
  const bool two_tries = !SON && NPL <= 32 && p.delta_first > 0.0 && p.delta_first < p.delta_rel;
  for (int attempt = two_tries ? 0 : 1; attempt < 2; ++attempt) {
  delta = (attempt == 0 ? p.delta_first : p.delta_rel) * scmax;
  if (attempt == 1 && two_tries) {
    iters_first = iters;
    for (int i = lane; i < (T + 1) * NPL; i += 64) { lam[i] = 0.0; rq[i] = 0.0; }
    if (lane < NPL) { tmp[lane] = 0.0; tmp2[lane] = 0.0; }
    WSYNC();
  }
  iters = 0; status = 0;
  if (sd.has_w) {
    resid = residual_pass();                    // g ≠ 0: z(0) = −H⁻¹g ≠ 0
  } else {                                      // g = 0: z(0) = 0, r = f = e_pos exactly (rq was cleared above)
    if (lane == 0 && sd.pos >= 0) rq[sd.pos] = 1.0;
    WSYNC();
    resid = (sd.pos >= 0) ? 1.0 : 0.0;
  }
  lap(1);                       // residual passes

  if (resid > p.tol) {
    // =========================== factor: M = P_k = (D_k − L_k P_{k−1} L_kᵀ + δI)⁻¹ ===========================
    auto matvec = [&](const double (&Pk)[RPL]) -> double {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const double y = tmp2[HS * r + h];
        if ((r & 3) == 0) a0 = __builtin_fma(Pk[r], y, a0);
        else if ((r & 3) == 1) a1 = __builtin_fma(Pk[r], y, a1);
        else if ((r & 3) == 2) a2 = __builtin_fma(Pk[r], y, a2);
        else a3 = __builtin_fma(Pk[r], y, a3);
      }
      double part = (a0 + a1) + (a2 + a3);
      if (HS >= 2) part = xsum32(part);
      if (HS >= 4) part = xsum16(part);
      return part;
    };
    // ---- P_k in the workspace: stored half only, packed by columns (element (i ≤ j) at j(j+1)/2 + i), for the NPL = 32 classes
    // (the chains' ñx = 27: 378 of the 896 doubles of the register image — the P_k trips are what is left of this kernel's HBM
    // traffic).  Both directions go through the LDS image, which is idle between the block build and the next one and during
    // the sweeps: stores and loads of the workspace stay whole 512-B rows.
    constexpr bool PK = (NPL == 32) && SLS_WAVE_PACKED_P != 0;
    constexpr int NLMAX = PK ? (NP * (NP + 1) / 2 + 63) / 64 : 1;
    const int NL = PK ? (n * (n + 1) / 2 + 63) / 64 : RPL;
    // Packed storage is decided block by block.  The mirror image of the upper half stands in for the lower half in the later
    // sweeps, which is only as good as the computed inverse is symmetric: in-register Gauss–Jordan leaves a one-sided inverse
    // (D'·P̂ ≈ I to rounding) whose asymmetry grows with the block's condition number, and on a near-singular block mixing its
    // rows and columns destroys exactly the property the multiplier iteration relies on (measured: a feasible ñx = 32 column of
    // a random plant stalled at 1e-5 and was flagged infeasible; symmetrising the registers as well made it worse, and a
    // second, full-image copy of this function — inlined or called — cost the packed build a third to a half of its speed).
    // A block that fails the test below keeps its full register image; chains never do.  Bit k of pkmask: block k is packed.
    unsigned long long pkmask = 0;
    const bool pk_dyn = PK && T + 1 <= 64;
    // slot of block k: packed blocks take NL rows of 64 doubles, full ones RPL, laid out back to back
    auto slot_of = [&](int k) -> int64_t {
      if constexpr (PK) {
        const int nfull = k - __popcll(k < 64 ? pkmask & ((1ull << k) - 1ull) : pkmask);      // (beyond 64 blocks nothing is packed: pkmask = 0)
        return ((int64_t)k * NL + (int64_t)nfull * (RPL - NL)) * 64;
      } else {
        return (int64_t)k * RPL * 64;
      }
    };
    int po1[NLMAX], po2[NLMAX];                                               // image offsets of this lane's packed elements
    if constexpr (PK) {
#pragma unroll
      for (int u = 0; u < NLMAX; ++u) {
        const int e = lane + 64 * u;
        int jj = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
        while ((jj + 1) * (jj + 2) / 2 <= e) ++jj;
        while (jj * (jj + 1) / 2 > e) --jj;
        const int ii = e - jj * (jj + 1) / 2;
        const bool ok = jj < n;
        po1[u] = ok ? ii * LDM + jj : -1;
        po2[u] = ok ? jj * LDM + ii : -1;
      }
    }
    auto store_P = [&](int k, const double (&Mr)[RPL]) {
      bool packed = false;
      if constexpr (PK) {
        if (pk_dyn) {
          // (after the tiled Gauss–Jordan the image already holds P_k row-major with this leading dimension — it is how the block
          //  came back from the lane grid — and nothing has written to it since: the forward substitution uses tmp/tmp2 only)
          if constexpr (!(SLS_TILED_GJ_WAVE != 0)) {
#pragma unroll
            for (int r = 0; r < RPL; ++r) mat[(HS * r + h) * LDM + j] = Mr[r];
            WSYNC();
          }
          // every lane judges its own elements (a sample spread over the block: element e = lane + 64u) against the largest of
          // them — which can only be stricter than the block's largest entry — and one ballot decides, no reduction
          double dmax = 0.0, amax = 0.0;
#pragma unroll
          for (int u = 0; u < NLMAX; ++u)
            if (u < NL && po1[u] >= 0) {
              const double a = mat[po1[u]], b = mat[po2[u]];
              dmax = fmax(dmax, fabs(a - b)); amax = fmax(amax, fmax(fabs(a), fabs(b)));
            }
          packed = __builtin_amdgcn_ballot_w64(dmax > 1e-13 * amax) == 0;
          if (packed) {
            const int64_t so = slot_of(k);
#pragma unroll
            for (int u = 0; u < NLMAX; ++u)
              if (u < NL) fac[so + 64 * u + lane] = (po1[u] >= 0) ? mat[po1[u]] : 0.0;
            pkmask |= 1ull << k;
          }
          WSYNC();
        }
      }
      if (!packed) {
        const int64_t so = slot_of(k);
#pragma unroll
        for (int r = 0; r < RPL; ++r) fac[so + r * 64 + lane] = Mr[r];
      }
    };
    constexpr int NPF = PK ? NLMAX : RPL;                  // registers of a block in flight (packed rows / register image)
    auto fetch_P = [&](int k, double (&Pf)[NPF]) {
      if constexpr (PK) {
        // (a block kept as its full image is not prefetched: it is read where it is used — rare, and the in-flight buffer stays
        //  at the packed size: a buffer of RPL rows cost the <32,14> build 224 B more scratch and 13 % of its speed)
        if (k < 64 && ((pkmask >> k) & 1ull)) {
          const int64_t so = slot_of(k);
#pragma unroll
          for (int u = 0; u < NLMAX; ++u) Pf[u] = (u < NL) ? fac[so + 64 * u + lane] : 0.0;
        }
      } else {
#pragma unroll
        for (int r = 0; r < RPL; ++r) Pf[r] = fac[((int64_t)k * RPL + r) * 64 + lane];
      }
    };
    auto expand_P = [&](int k, const double (&Pf)[NPF], double (&Pk)[RPL]) {
      if constexpr (PK) {
        if (k < 64 && ((pkmask >> k) & 1ull)) {
          WSYNC();
#pragma unroll
          for (int u = 0; u < NLMAX; ++u)
            if (u < NL && po1[u] >= 0) { mat[po1[u]] = Pf[u]; mat[po2[u]] = Pf[u]; }
          WSYNC();
#pragma unroll
          for (int r = 0; r < RPL; ++r) Pk[r] = (HS * r + h < n && j < n) ? mat[(HS * r + h) * LDM + j] : 0.0;
          WSYNC();
        } else {
          const int64_t so = slot_of(k);
#pragma unroll
          for (int r = 0; r < RPL; ++r) Pk[r] = fac[so + r * 64 + lane];
        }
      } else {
#pragma unroll
        for (int r = 0; r < RPL; ++r) Pk[r] = Pf[r];
      }
    };
    double M[RPL];
    for (int k = 0; k <= T; ++k) {
      const double wcur = (k <= T - 1 && j < n && mask[k * nm + j]) ? hx[j] : 0.0;
      // r_k for the fused forward substitution below: with the vectors in the global workspace this is an L2 round trip —
      // issued here, it returns during the block build instead of in front of the matrix–vector product
      const double rk_early = (lane < NPL) ? rq[k * NPL + lane] : 0.0;
      if (k == 0) {
#pragma unroll
        for (int r = 0; r < RPL; ++r) M[r] = (HS * r + h == j) ? (delta + wcur) : 0.0;
      } else {
        const double wj = (j < n && mask[(k - 1) * nm + j]) ? hx[j] : 0.0;
        if (h == 0) wl[j] = wj;
        wul[lane] = (lane < m && mask[(k - 1) * nm + n + lane]) ? hu[lane] : 0.0;
        WSYNC();
        // Q = W − W P W  (P = M), written row-major into the LDS image
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
          const int i = HS * r + h;
          const double wi = wl[i];
          mat[i * LDM + j] = wi * (((i == j) ? 1.0 : 0.0) - M[r] * wj);
        }
        WSYNC();
        // Y = Q Ãᵀ :  Y[i][j] = Σ_e Ã[j][c_e]·Q[i][c_e]   (row list of lane j)
        double Y[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) Y[r] = 0.0;
        if (p.knock_out != 2) cached_product(Y);
        for (int e = KR; e < nzA; ++e) {
          const int c = arow_c[e * NPL + j];
          const double v = arow_v[e * NPL + j];
#pragma unroll
          for (int r = 0; r < RPL; ++r) Y[r] = __builtin_fma(v, mat[(HS * r + h) * LDM + c], Y[r]);
        }
        WSYNC();
        // image ← Yᵀ
        if (j < NP) {
#pragma unroll
          for (int r = 0; r < RPL; ++r) mat[j * LDM + (HS * r + h)] = Y[r];
        }
        WSYNC();
        // Z = Ã Q Ãᵀ (symmetric):  Z[i][j] = Σ_e Ã[j][c_e]·Y[c_e][i] = Σ_e Ã[j][c_e]·image[i][c_e]
#pragma unroll
        for (int r = 0; r < RPL; ++r) M[r] = (HS * r + h == j) ? (delta + wcur) : 0.0;
        if (p.knock_out != 2) cached_product(M);
        for (int e = KR; e < nzA; ++e) {
          const int c = arow_c[e * NPL + j];
          const double v = arow_v[e * NPL + j];
#pragma unroll
          for (int r = 0; r < RPL; ++r) M[r] = __builtin_fma(v, mat[(HS * r + h) * LDM + c], M[r]);
        }
        // + B̃ Wu B̃ᵀ
        for (int e = 0; e < nzB; ++e) {
          const int c = brow_c[e * NPL + j];
          const double v = brow_v[e * NPL + j] * wul[c];
          double bq[RPL];
#pragma unroll
          for (int r = 0; r < RPL; ++r) bq[r] = Bd[(HS * r + h) * MC + c];          // all RPL reads first: one latency, not RPL
#pragma unroll
          for (int r = 0; r < RPL; ++r) M[r] = __builtin_fma(v, bq[r], M[r]);
        }
      }
      lap(2);                   // D' build (sparse products)
      // ---- in-register Gauss–Jordan over the n live pivots (SPD ⇒ no pivoting): M ← M⁻¹ ----
      // The pivot ROW (owner lane group) and the true pivot COLUMN (lanes j == pv of every group) both go
      // through LDS.  (Using the row as a stand-in for the column — legal for an exactly symmetric matrix —
      // amplifies round-off asymmetry by (1−d)/d per pivot and diverges for large pivots: measured, see
      // DESIGN.md §5.)
      if constexpr ((NPL == 32 && SLS_TILED_GJ_WAVE != 0) || (NPL == 64 && SLS_TILED_GJ_WAVE64 != 0)) {
        if (p.knock_out != 1) gauss_jordan_tiled<NPL, RPL, LDM>(M, mat, lane, n);     // 8×8 lane grid, see above (the image is free between the build and the sweeps)
      } else {
      double dnext = fast_rcp(readlane_f64(M[0], 0));      // 1/pivot of pivot 0 (row 0 lives in group 0, register 0)
      static_for<NP>([&](auto pv_c) {
        {
          constexpr int pv = decltype(pv_c)::value;
          constexpr int rp = pv / HS, hp = pv % HS;
          if (pv < n) {
            constexpr int rn = (pv + 1) / HS, hn = (pv + 1) % HS;       // owner of the NEXT pivot row (compile time)
            const double d = dnext;
            unsigned long long ps0 = 0, ps1 = 0, ps2 = 0;
            if (p.dbg_level >= 2) { __builtin_amdgcn_sched_barrier(0); ps0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); }
            // (1) the true pivot COLUMN: every lane takes M[r] of lane pv of its own lane group (group_bcast above).
            // No LDS memory, no write→read dependency, no exec masking: the ds_write2/ds_read2 version of this step
            // measured 555 cycles per pivot for the round trip alone, ds_bpermute saturates the CU at 4 waves.
            // (NPL = 64: one lane group per wave, the broadcast is a scalar v_readlane that feeds its FMA directly —
            //  materialising 40–64 of them at once only spills)
            double c[HS == 1 ? 1 : RPL];
            if constexpr (HS > 1) {
#pragma unroll
              for (int r = 0; r < RPL; ++r) c[r] = group_bcast<NPL, pv>(M[r]);
            }
            if (p.dbg_level >= 2) {   // diagnostic: wait for the column here so the two segments can be told apart
              __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); ps1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0);
            }
            // (2) … while the pivot ROW crosses lane groups in registers (v_permlane32_swap) and the NEXT pivot is
            // predicted from three broadcasts, M[p+1][p+1] − M[p+1][p]·M[p][p+1]·d, so that its reciprocal
            // (v_rcp_f64 + 2 Newton steps, ≈170 dependent cycles) never waits for the LDS round trip.
            double rowj;
            if (HS == 1) rowj = M[rp];
            else if (HS == 2) {
              const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(M[rp]), __double2loint(M[rp]), false, false);
              const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(M[rp]), __double2hiint(M[rp]), false, false);
              // swap(x,x)[0] = x.lo32 in both halves, [1] = x.hi32 in both halves; the owner group is hp
              rowj = __hiloint2double(hi[hp], lo[hp]);
            } else {                               // HS == 4: from lane (hp·NPL + j)
              const int arow = (hp * NPL + j) << 2;
              rowj = __hiloint2double(__builtin_amdgcn_ds_bpermute(arow, __double2hiint(M[rp])),
                                      __builtin_amdgcn_ds_bpermute(arow, __double2loint(M[rp])));
            }
            double xr = 0.0, pn = 1.0;
            constexpr bool have_next = rn < RPL;     // an unused 1/x (pv+1 ≥ n) is harmless
            if constexpr (have_next) {
              const double a_nn = readlane_f64(M[rn], hn * NPL + pv + 1);
              const double a_np = readlane_f64(M[rn], hn * NPL + pv);
              const double a_pn = readlane_f64(M[rp], hp * NPL + pv + 1);
              pn = __builtin_fma(-(a_np * d), a_pn, a_nn);
              xr = __builtin_amdgcn_rcp(pn);
              xr = __builtin_fma(__builtin_fma(-pn, xr, 1.0), xr, xr);
              xr = __builtin_fma(__builtin_fma(-pn, xr, 1.0), xr, xr);
            }
            const double tj = rowj * d;
            const double tfix = (j == pv) ? (1.0 + d) : tj;     // lane pv: c − c(1+d) = −c·d
            if constexpr (HS > 1) {
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int r = 0; r < RPL; ++r) M[r] = __builtin_fma(-c[r], tfix, M[r]);
            } else {
#pragma unroll
              for (int r = 0; r < RPL; ++r) M[r] = __builtin_fma(-readlane_f64(M[r], pv), tfix, M[r]);
            }
            if (h == hp) M[rp] = (j == pv) ? d : tj;
            if constexpr (have_next) dnext = xr;
            if (p.dbg_level >= 2) {
              __builtin_amdgcn_sched_barrier(0);
              asm volatile("" :: "v"(M[0]), "v"(M[RPL - 1]), "v"(dnext));
              ps2 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0);
              tc[6] += ps1 - ps0; tc[7] += ps2 - ps1;
            }
          }
        }
      });
      }
      lap(3);                   // Gauss–Jordan
      // ---- first forward substitution fused here (P_k is in registers): y_k = r_k + Ã(W_{k−1}q_{k−1}), q_k = P_k y_k ----
      {
        if (lane < NPL) {
          double acc = rk_early;
          if (k >= 1) acc += dotA_row(tmp);
          tmp2[lane] = (lane < n) ? acc : 0.0;
        }
        WSYNC();
        const double q = matvec(M);
        if (lane < NPL) {
          rq[k * NPL + lane] = q;
          tmp[lane] = wcur * q;                  // wcur = Wx_k of lane j (0 for k = T); lanes < NPL have j = lane
        }
        WSYNC();
      }
      lap(5);
      // ---- stream the pivot block P_k to the workspace ----
      if (p.knock_out != 3) store_P(k, M);
    }
    WSYNC();
    lap(4);                     // P_k stores (+ loop tail)

    // =========================== multiplier iteration ===========================
    // (sum-of-norms build: one trip of the outer loop per projection — the column's own solve first, then the ADMM steps)
    int admm = 0;
    double rho_s = 1.0;
    int aa_k = 0, aa_col = 0;            // Anderson acceleration: columns in use, next column of the ring
    bool aa_prev = false;                // F(s) and g of the previous step are stored
    double aa_gmin = 1e300;
    for (;;) {
    double prev = resid, prev2 = resid;
    int itmax = p.max_iters;
    for (int it = 1; it <= itmax; ++it) {
      if constexpr (SON) { if (admm > 0) tc[7] += 1; }          // phase timers, sum-of-norms build: slot 7 = multiplier passes of the projections
      iters = it;
      // forward: y_k = r_k + Ã(W_{k−1} q_{k−1});  q_k = P_k y_k   (q_k overwrites r_k in rq)
      // P_{k+1} is prefetched from the workspace while block k is multiplied; the RPL-long dot product is
      // split into four independent accumulators (a dependent v_fma_f64 costs ≈32 cycles on gfx950).
      if (it > 1 || (SON && admm > 0)) {         // pass 1's forward sweep ran inside the factor loop (own solve only)
        double Pn[NPF];
        fetch_P(0, Pn);
        for (int k = 0; k <= T; ++k) {
          const double rk_early = (lane < NPL) ? rq[k * NPL + lane] : 0.0;      // (in flight during the expansion of P_k)
          double Pk[RPL];
          expand_P(k, Pn, Pk);
          if (k < T) fetch_P(k + 1, Pn);
          if (lane < NPL) {
            double acc = rk_early;
            if (k >= 1) acc += dotA_row(tmp);
            tmp2[lane] = (lane < n) ? acc : 0.0;
          }
          WSYNC();
          const double q = matvec(Pk);
          if (lane < NPL) {
            rq[k * NPL + lane] = q;
            const double w = (k <= T - 1 && lane < n && mask[k * nm + lane]) ? hx[lane] : 0.0;
            tmp[lane] = w * q;
          }
          WSYNC();
        }
      }
      // backward: Δλ_k = q_k + P_k (W_k Ãᵀ Δλ_{k+1});  λ += Δλ   (Δλ_k overwrites q_k)
      if (p.knock_out != 4) {
        // Δλ_{k+1} travels from step to step in LDS (`tmp`): read back from the global workspace it was a store → load round
        // trip through L2 on the dependent chain of every block (round 3; the forward sweep always handed W q over in LDS)
        if (lane < NPL) { const double qT = rq[T * NPL + lane]; lam[T * NPL + lane] += qT; tmp[lane] = qT; }
        double Pn[NPF];
        if (T >= 1) fetch_P(T - 1, Pn);
        WSYNC();
        for (int k = T - 1; k >= 0; --k) {
          const double qk_early = (lane < NPL) ? rq[k * NPL + lane] : 0.0;
          const double lk_early = (lane < NPL) ? lam[k * NPL + lane] : 0.0;
          double Pk[RPL];
          expand_P(k, Pn, Pk);
          if (k >= 1) fetch_P(k - 1, Pn);
          if (lane < NPL) {
            double acc = 0.0;
            if (lane < n && mask[k * nm + lane]) {
              acc = dotA_col(tmp);
              acc *= hx[lane];
            }
            tmp2[lane] = acc;
          }
          WSYNC();
          const double dl = matvec(Pk) + qk_early;
          if (lane < NPL) {
            rq[k * NPL + lane] = dl;
            lam[k * NPL + lane] = lk_early + dl;
            tmp[lane] = dl;
          }
          WSYNC();
        }
      }
      lap(5);                   // substitution sweeps
      resid = (p.knock_out == 5) ? 0.0 : residual_pass();
      lap(1);
      if (resid <= p.tol || p.knock_out != 0) break;
      // (a projection of the sum-of-norms loop is a consistent system by construction: slow progress there is a near-singular
      //  direction, not infeasibility — it gets all its passes)
      if (it >= 2 && resid > p.stag * prev && !(SON && admm > 0)) {
        // (the first attempt's tiny shift is not the place to be patient: it hands over to the robust shift)
        const bool patient = p.max_iters_slow > 0 && (attempt == 1 || !two_tries) && (resid > p.tol_ok || itmax > p.max_iters) && still_contracting(it >= 3 ? prev2 : prev, prev, resid);
        if (!patient) { status = 1; break; }
        itmax = max(itmax, p.max_iters_slow);
      }
      prev2 = prev; prev = resid;
    }
    if (resid <= p.tol_ok) status = 0;
    else if (status == 0) status = 2;
    if constexpr (!SON) {
      break;
    } else {
      // =================== sum-of-norms objective  min Σ_t ‖W z_t‖₂  s.t. E z = f  (SLS_SOLVE_SUM_OF_NORMS) ===================
      // Same ADMM as the tile kernel's (sls_tile_kernel.hip; oracle: oracle/sls_son_oracle.py) with this kernel's solve as the
      // projection: the factor stays in the workspace, a step is λ ← 0, two multiplier passes with the linear term −W(y − u),
      // then the block soft threshold.  z(λ) of the last projection is the answer (x_of / u_of read the same linear term).
      if (status != 0) { if (admm > 0) { status = 2; iters = admm; } break; }      // the own solve failed: reported as it is; a projection did not converge: SLS_COL_NOTCONV
      constexpr double kRelax = 1.8;
      const double arel = (admm == 0) ? 1.0 : kRelax;          // first trip: v = W z of the 𝓗₂ solution starts y
      // phase A: v = α·W z + (1−α)·y + u per variable, its squared norm per time step (x part → rq[t][0], u part → rq[t][1])
      {
        const int nround = (T + HS - 1) / HS;
        for (int itx = 0; itx < nround; ++itx) {
          const int t = itx * HS + h;
          double part = 0.0;
          if (t < T && j < n && mask[t * nm + j]) {
            const double wz = sqrt(1.0 / hx[j]) * x_of(t);
            const double v = __builtin_fma(arel, wz, __builtin_fma(1.0 - arel, syv[t * nm + j], suv[t * nm + j]));
            svv[t * nm + j] = v;
            part = v * v;
          }
#pragma unroll
          for (int off = NPL / 2; off > 0; off >>= 1) part += __shfl_xor(part, off);
          if (j == 0 && t < T) rq[t * NPL] = part;
        }
        for (int t0 = 0; t0 < T; t0 += NTS) {
          const int t = t0 + uts;
          double part = 0.0;
          if (t < T && uq < m && mask[t * nm + n + uq]) {
            const double wz = sqrt(1.0 / hu[uq]) * u_of(t);
            const double v = __builtin_fma(arel, wz, __builtin_fma(1.0 - arel, syv[t * nm + n + uq], suv[t * nm + n + uq]));
            svv[t * nm + n + uq] = v;
            part = v * v;
          }
          for (int off = (1 << lgMP) >> 1; off > 0; off >>= 1) part += __shfl_xor(part, off);
          if (uq == 0 && t < T) rq[t * NPL + 1] = part;
        }
        WSYNC();
      }
      bool go_on = true;
      if (admm == 0) {
        double big = 0.0;
        for (int t = lane; t < T; t += 64) big = fmax(big, rq[t * NPL] + rq[t * NPL + 1]);
        big = sqrt(wave_max_f64(big));
        rho_s = (big > 0.0) ? 8.0 / big : 1.0;
        for (int e = lane; e < T * nm; e += 64) { syv[e] = mask[e] ? svv[e] : 0.0; suv[e] = 0.0; }
      } else {
        // phase B: y ← block soft threshold of v, u ← v − y; primal / dual residuals; g = F(s) − s of the fixed-point map
        // s = (y, u) ↦ F(s) for the acceleration below
        double rp2 = 0.0, rd2 = 0.0, nx2 = 0.0, gn2 = 0.0;
        for (int e = lane; e < T * nm; e += 64) {
          if (!mask[e]) continue;
          const int t = e / nm;
          const double nv = sqrt(rq[t * NPL] + rq[t * NPL + 1]);
          const double sh = (nv * rho_s > 1.0) ? 1.0 - 1.0 / (rho_s * nv) : 0.0;
          const double v = svv[e], yo = syv[e], uo = suv[e];
          const double yn = sh * v, un = v - yn;
          const double wz = ((v - uo) - (1.0 - kRelax) * yo) * (1.0 / kRelax);
          const double dp = wz - yn, dy = yn - yo, du = un - uo;
          rp2 = __builtin_fma(dp, dp, rp2); rd2 = __builtin_fma(dy, dy, rd2); nx2 = __builtin_fma(wz, wz, nx2);
          gn2 = __builtin_fma(dy, dy, __builtin_fma(du, du, gn2));
          syv[e] = yn; suv[e] = un;
          sgc[e] = dy; sgc[L1 + e] = du;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          rp2 += __shfl_xor(rp2, off); rd2 += __shfl_xor(rd2, off); nx2 += __shfl_xor(nx2, off); gn2 += __shfl_xor(gn2, off);
        }
        const double rp = sqrt(rp2), rd = rho_s * sqrt(rd2), nx = sqrt(nx2);
        const bool converged = fmax(rp, rd) <= p.son_tol * fmax(1.0, nx);
        bool aa_reset = false;
        if (converged || admm >= p.son_maxit) {
          iters = admm;
          status = converged ? 0 : 2;
          go_on = false;
        } else if (admm % 10 == 0) {
          const double sc_ = (rp > 10.0 * rd) ? 2.0 : ((rd > 10.0 * rp) ? 0.5 : 1.0);
          if (sc_ != 1.0) {
            rho_s *= sc_;
            for (int e = lane; e < T * nm; e += 64) suv[e] /= sc_;
            aa_reset = true;                       // the map changed with ρ
          }
        }
        // Anderson acceleration (type II, memory AAM) of the fixed-point iteration s ← F(s): the next iterate is
        // F(s) − ΔF γ with γ = argmin ‖g − ΔG γ‖₂ over the last AAM differences of F(s) and of g = F(s) − s.  Plain ADMM needs
        // thousands of steps on columns whose optimal support in time is degenerate (chain-4096: median 2461); the accelerated
        // iteration 90–1000 (CPU study: tools/son_anderson_study.py).  Safeguards: restart when ρ changes or when ‖g‖ rises
        // tenfold above its smallest value since the last restart; the normal equations are regularised by 1e-10·trace/k.
        if (go_on && p.son_anderson) {
          const double gn = sqrt(gn2);
          if (aa_reset || gn > 10.0 * aa_gmin || admm <= p.son_aa_start) {
            aa_k = 0; aa_col = 0; aa_prev = false; aa_gmin = (admm <= p.son_aa_start) ? 1e300 : gn;
            if (aa_reset) aa_gmin = 1e300;
          } else {
            aa_gmin = fmin(aa_gmin, gn);
            // ONE pass over the step's vectors: the new difference (F, g of this step minus the previous step's) goes to its ring
            // slot, F and g become "previous", and the Gram matrix ΔGᵀΔG (upper half) and ΔGᵀg are accumulated — every load of
            // an iteration issued together (unconditional loads, selects afterwards): the history of all resident waves
            // (30 vectors each) does not fit the caches, and four separate predicated loops spent 575 k cycles per step waiting
            // for one load after the other (tools/son_phase_breakdown.py)
            const bool had_prev = aa_prev;
            const int slot_new = aa_col;
            if (had_prev) { aa_col = (aa_col + 1 == AAM) ? 0 : aa_col + 1; aa_k = min(aa_k + 1, AAM); }
            aa_prev = true;
            double Am[AAM][AAM], bv[AAM];
#pragma unroll
            for (int a = 0; a < AAM; ++a) { bv[a] = 0.0;
#pragma unroll
              for (int b2 = 0; b2 < AAM; ++b2) Am[a][b2] = 0.0; }
            for (int e = lane; e < T * nm; e += 64) {
              const bool mk = mask[e] != 0;
              const double sy = syv[e], su = suv[e], cy0 = sgc[e], cu0 = sgc[L1 + e];
              const double fy = sFp[e], fu = sFp[L1 + e], py = sgp[e], pu = sgp[L1 + e];
              double gy[AAM], gu2[AAM];
#pragma unroll
              for (int a = 0; a < AAM; ++a) { gy[a] = sdG[(int64_t)a * L2 + e]; gu2[a] = sdG[(int64_t)a * L2 + L1 + e]; }
              const double ndy = cy0 - py, ndu = cu0 - pu;
#pragma unroll
              for (int a = 0; a < AAM; ++a) {
                const bool used = mk && a < aa_k, isnew = had_prev && a == slot_new;
                gy[a] = used ? (isnew ? ndy : gy[a]) : 0.0;
                gu2[a] = used ? (isnew ? ndu : gu2[a]) : 0.0;
              }
              if (mk) {
                if (had_prev) {
                  double* dF = sdF + (int64_t)slot_new * L2; double* dG = sdG + (int64_t)slot_new * L2;
                  dF[e] = sy - fy; dF[L1 + e] = su - fu; dG[e] = ndy; dG[L1 + e] = ndu;
                }
                sFp[e] = sy; sFp[L1 + e] = su; sgp[e] = cy0; sgp[L1 + e] = cu0;
              }
              const double cy = mk ? cy0 : 0.0, cu = mk ? cu0 : 0.0;
#pragma unroll
              for (int a = 0; a < AAM; ++a) {
                bv[a] = __builtin_fma(gy[a], cy, __builtin_fma(gu2[a], cu, bv[a]));
#pragma unroll
                for (int b2 = a; b2 < AAM; ++b2) Am[a][b2] = __builtin_fma(gy[a], gy[b2], __builtin_fma(gu2[a], gu2[b2], Am[a][b2]));
              }
            }
            if (aa_k > 0) {
#pragma unroll
              for (int a = 0; a < AAM; ++a) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) bv[a] += __shfl_xor(bv[a], off);
#pragma unroll
                for (int b2 = a; b2 < AAM; ++b2) {
#pragma unroll
                  for (int off = 32; off > 0; off >>= 1) Am[a][b2] += __shfl_xor(Am[a][b2], off);
                }
              }
              double tr = 0.0;
#pragma unroll
              for (int a = 0; a < AAM; ++a) tr += Am[a][a];
              const double regv = 1e-10 * tr / (double)aa_k;
#pragma unroll
              for (int a = 0; a < AAM; ++a) {
                Am[a][a] = (a < aa_k) ? Am[a][a] + regv : 1.0;          // unused columns: identity rows, γ = 0
#pragma unroll
                for (int b2 = 0; b2 < a; ++b2) Am[a][b2] = Am[b2][a];
              }
              // Cholesky-free Gaussian elimination (SPD after the shift), every lane redundantly
              double gam[AAM];
#pragma unroll
              for (int a = 0; a < AAM; ++a) {
                const double piv = 1.0 / Am[a][a];
#pragma unroll
                for (int r = a + 1; r < AAM; ++r) {
                  const double f2 = Am[r][a] * piv;
#pragma unroll
                  for (int c2 = a; c2 < AAM; ++c2) Am[r][c2] = __builtin_fma(-f2, Am[a][c2], Am[r][c2]);
                  bv[r] = __builtin_fma(-f2, bv[a], bv[r]);
                }
              }
#pragma unroll
              for (int a = AAM - 1; a >= 0; --a) {
                double acc = bv[a];
#pragma unroll
                for (int c2 = a + 1; c2 < AAM; ++c2) acc = __builtin_fma(-Am[a][c2], gam[c2], acc);
                gam[a] = acc / Am[a][a];
              }
              bool finite = true;
#pragma unroll
              for (int a = 0; a < AAM; ++a) finite = finite && (fabs(gam[a]) < 1e6);
              if (finite) {
                for (int e = lane; e < T * nm; e += 64) {
                  const bool mk = mask[e] != 0;
                  double ay = syv[e], au = suv[e];
                  double fy2[AAM], fu2[AAM];
#pragma unroll
                  for (int a = 0; a < AAM; ++a) { fy2[a] = sdF[(int64_t)a * L2 + e]; fu2[a] = sdF[(int64_t)a * L2 + L1 + e]; }
#pragma unroll
                  for (int a = 0; a < AAM; ++a) {
                    const bool used = mk && a < aa_k;
                    ay = __builtin_fma(-gam[a], used ? fy2[a] : 0.0, ay); au = __builtin_fma(-gam[a], used ? fu2[a] : 0.0, au);
                  }
                  if (mk) { syv[e] = ay; suv[e] = au; }
                }
              } else { aa_k = 0; aa_col = 0; aa_prev = false; aa_gmin = 1e300; }
            }
          }
        }
      }
      lap(6);                   // (sum-of-norms build: slot 6 = threshold + acceleration of the ADMM steps)
      if (!go_on) break;
      // next projection: linear term −W(y − u), r = f − E z(λ)
      for (int e = lane; e < T * nm; e += 64) {
        const int q = e % nm;
        const double wq = sqrt(1.0 / ((q < n) ? hx[q] : hu[q - n]));
        sgl[e] = mask[e] ? -wq * (syv[e] - suv[e]) : 0.0;
      }
      // λ is kept (warm start): the new residual is E H⁻¹(g_new − g_old), one multiplier pass takes it to δ·‖Δλ‖ — below the
      // tolerance once the ADMM steps are small, instead of the two passes a start from λ = 0 always needs
      WSYNC();
      status = 0;
      resid = residual_pass();
      lap(1);
      ++admm;
    }
    }
  }
  if (attempt == 0 && p.knock_out == 0 && !(status == 0 && resid <= p.tol)) continue;     // the small shift did not do it: robust shift, from scratch
  break;
  }
  iters += iters_first;
  output_pass();
  if (sd.pos < 0 && status == 0) status = 3;
  if (p.dbg && lane == 0) {
    for (int q = 0; q < 8; ++q) p.dbg[sd.out_index * 8 + q] = tc[q];
  }
  if (lane == 0) {
    p.status[sd.out_index] = status;
    p.resid[sd.out_index] = resid;
    p.iters[sd.out_index] = iters;
  }
}

// register budget: 256 VGPRs (2 waves/SIMD) for NPL ≤ 32, 512 (1 wave/SIMD) for the NPL = 64 classes whose pivot block
// alone takes 2·RPL = 80..128 registers
template <int NPL, int RPL, bool VG, bool SON = false>
#ifndef SLS_WAVE_OCC
#define SLS_WAVE_OCC 2          // experiments: 1 = a whole SIMD's registers per wave (no scratch), four waves per CU
#endif
__global__ __launch_bounds__(64, (NPL == 64 ? 1 : SLS_WAVE_OCC)) void h2_column_wave_kernel(const KernelParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double* fac = p.fac_ws + (int64_t)blockIdx.x * p.fac_stride;
  double* gvec = VG ? p.vec_ws + (int64_t)blockIdx.x * p.vec_stride : nullptr;
  if constexpr (SON) {
    // the ADMM step counts differ by 16× between columns (55 … 874 on chain-4096): columns are drawn from a queue (one atomic
    // counter per launch, cleared by the host before every execute) instead of the static round-robin of the 𝓗₂ builds
    for (;;) {
      int s = 0;
      if (threadIdx.x == 0) s = p.work_counter ? atomicAdd(p.work_counter, 1) : p.nsub;
      s = __builtin_amdgcn_readfirstlane(s);
      if (s >= p.nsub) break;
      const SubDesc sd = p.subs[p.order[p.order_off + s]];
      wave_solve_column<NPL, RPL, VG, SON>(p, sd, fac, gvec, lds_raw);
    }
  } else {
    for (int s = blockIdx.x; s < p.nsub; s += gridDim.x) {
      const SubDesc sd = p.subs[p.order[p.order_off + s]];
      wave_solve_column<NPL, RPL, VG, SON>(p, sd, fac, gvec, lds_raw);
    }
  }
}


// =====================================================================================================================
// Twisted (two-ended) variant for the LATENCY regime — fewer subproblems than SIMDs, e.g. the README chain's 59 columns.
// A column's time is a serial chain over the T+1 block rows.  Here a workgroup of TWO waves owns the column: wave 0
// eliminates upwards from block 0, wave 1 downwards from block T (block LDLᵀ of the same S+δI taken from both ends:
// "twisted factorisation"), they meet in a middle block c, and both substitution sweeps split the same way — the serial
// chain is halved.  Backward Schur step:  G_k = δI + W_k + ÃW_{k−1}Ãᵀ + B̃Wu_{k−1}B̃ᵀ − W_k(Ãᵀ P_{k+1} Ã)W_k ,
// middle block:  D'_c(forward formula) − W_c(Ãᵀ P_{c+1} Ã)W_c.  Same P_k workspace, same refinement, same outputs.
// =====================================================================================================================
// meeting block: the downward wave's blocks cost ≈7 % more (one more sparse product when the masks change), so it gets one fewer
static inline __host__ __device__ int twisted_middle(int T) { return T >= 5 ? (T + 1) / 2 : (T - 1) / 2; }

// PL = true: the pivot blocks P_k stay in LDS (compact ñx×ñx per block) instead of the global workspace — possible in this
// regime because one column owns a whole CU's 160 KB (README: 30·21²·8 B = 106 KB next to 56 KB of working set); the
// solve then moves only its algorithmic bytes through HBM (measured before: 23.8 MB per launch against 0.34 MB).
template <int NPL, int RPL, bool PL>
__device__ __forceinline__ void twisted_solve_column(const KernelParams& p, const SubDesc& sd, double* __restrict__ fac,
                                                     unsigned char* lds_raw) {
  constexpr int HS = 64 / NPL, NP = HS * RPL, LDM = NPL + 1;
  constexpr bool TILED = (NPL == 32) && (SLS_TILED_GJ != 0);   // Gauss–Jordan on an 8×8 lane grid (see gauss_jordan_tiled)
  constexpr int LDT = 40;                                 // leading dimension of the image during layout changes (8·a + b: no bank conflict)
  constexpr int MATSZ = NP * (NPL == 32 ? LDT : LDM);     // must match twisted_kernel_lds_bytes
  constexpr int PRIV = MATSZ + 3 * NPL + 64;             // doubles per wave: mat, tmp, tmp2, wl, wul
  const int wv = threadIdx.x >> 6;                        // 0: upward wave (blocks 0..c), 1: downward wave (blocks T..c+1)
  const int lane = threadIdx.x & 63;
  const int h = lane / NPL, j = lane % NPL;
  const int n = sd.n, m = sd.m, nm = n + m, T = p.T;
  const int MC = p.w_mcap;
  const int capA = p.w_nzA, capAc = p.w_nzAc, capB = p.w_nzB, capBc = p.w_nzBc;
  const int c = twisted_middle(T);

  // ---- LDS carve (must match twisted_kernel_lds_bytes): per-wave private block first, then the shared column data ----
  double* dp = reinterpret_cast<double*>(lds_raw);
  double* priv = dp + wv * PRIV; dp += 2 * PRIV;
  double* mat = priv; double* tmp = mat + MATSZ; double* tmp2 = tmp + NPL; double* wl = tmp2 + NPL;
  double* wul = wl + NPL;
  double* xch = dp;    dp += NP * LDM;                   // W_c(ÃᵀP_{c+1}Ã)W_c handed from wave 1 to wave 0
  double* hx = dp;     dp += NPL;
  double* gx = dp;     dp += NPL;
  double* hu = dp;     dp += 64;
  double* gu = dp;     dp += 64;
  double* red = dp;    dp += 8;                           // [0..1] per-wave maxima, [2] δ
  int32_t* sx = reinterpret_cast<int32_t*>(dp);
  int32_t* su = sx + NPL;
  int32_t* nzs = su + 64;                                 // nzA, nzAc, nzB, nzBc
  dp += (NPL + 64 + 8) / 2;
  double* lam = dp;    dp += (T + 1) * NPL;
  double* rq = dp;     dp += (T + 1) * NPL;
  double* xs = dp;     dp += (T + 1) * NPL;               // x_t of the current residual pass (x_T ≡ 0)
  double* Bd = dp;     dp += NPL * MC;
  double* us = dp;     dp += T * MC;
  double* arow_v = dp; dp += capA * NPL;
  double* acol_v = dp; dp += capAc * NPL;
  double* brow_v = dp; dp += capB * NPL;
  double* bcol_v = dp; dp += capBc * 64;
  int32_t* ip = reinterpret_cast<int32_t*>(dp);
  int32_t* arow_c = ip; ip += capA * NPL;
  int32_t* acol_c = ip; ip += capAc * NPL;
  int32_t* brow_c = ip; ip += capB * NPL;
  int32_t* bcol_c = ip; ip += capBc * 64;
  uint8_t* mask = reinterpret_cast<uint8_t*>(ip);
  double* pl = reinterpret_cast<double*>(lds_raw + p.w_pl_off);          // P_k blocks in LDS (PL only)
  const int32_t* dest = p.dest_pool + sd.off_dest;
  unsigned long long tc[4] = {0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
  auto lap = [&](int slot) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tc[slot] += now - tlast; tlast = now; };
  // element (i,j) of block k at pl[k·n² + i·n + j]; padded rows/columns read a clamped (finite) neighbour — they only
  // ever multiply zeros — and are never written
  const int jc = min(j, n - 1);
  auto p_off = [&](int r) -> int { return min(HS * r + h, n - 1) * n + jc; };      // recomputed per access: 24 VGPRs less than a table
  auto p_ok = [&](int r) -> bool { return HS * r + h < n && j < n; };

  __syncthreads();
  // ---- setup, split between the waves: wave 0 stages indices/weights, then wave 0 gathers the ROW lists of Ã, B̃ (and
  //      the dense B̃ image) while wave 1 copies the masks and gathers the COLUMN lists ----
  if (wv == 0) {
    if (lane < NPL) {
      sx[lane] = (lane < n) ? p.idx_pool[sd.off_sx + lane] : 0x7fffffff;
      hx[lane] = (lane < n) ? (sd.has_w ? p.w_pool[sd.off_w + lane] : 1.0) : 0.0;
      gx[lane] = (lane < n && sd.has_w) ? p.w_pool[sd.off_w + nm + lane] : 0.0;
    }
    su[lane] = (lane < m) ? p.idx_pool[sd.off_su + lane] : 0x7fffffff;
    hu[lane] = (lane < m) ? (sd.has_w ? p.w_pool[sd.off_w + n + lane] : 1.0) : 0.0;
    gu[lane] = (lane < m && sd.has_w) ? p.w_pool[sd.off_w + nm + n + lane] : 0.0;
    for (int i = lane; i < NPL * MC; i += 64) Bd[i] = 0.0;
    for (int i = lane; i < capA * NPL; i += 64) { arow_v[i] = 0.0; arow_c[i] = 0; }
    for (int i = lane; i < capB * NPL; i += 64) { brow_v[i] = 0.0; brow_c[i] = 0; }
  } else {
    for (int i = lane; i < T * nm; i += 64) mask[i] = p.mask_pool[sd.off_mask + i];
    for (int i = lane; i < capAc * NPL; i += 64) { acol_v[i] = 0.0; acol_c[i] = 0; }
    for (int i = lane; i < capBc * 64; i += 64) { bcol_v[i] = 0.0; bcol_c[i] = 0; }
    for (int i = lane; i < (T + 1) * NPL; i += 64) { lam[i] = 0.0; rq[i] = 0.0; xs[i] = 0.0; }
  }
  __syncthreads();
  if (wv == 0) {
    int cntA = 0, cntB = 0;
    if (lane < n) {
      const int g = sx[lane];
      for (int e = p.A_rowptr[g]; e < p.A_rowptr[g + 1]; ++e) {
        const double v = p.A_val[e];
        const int loc = (v != 0.0) ? wbsearch(sx, n, p.A_colidx[e]) : -1;
        if (loc >= 0 && cntA < capA) { arow_c[cntA * NPL + lane] = loc; arow_v[cntA * NPL + lane] = v; ++cntA; }
      }
      for (int e = p.B_rowptr[g]; e < p.B_rowptr[g + 1]; ++e) {
        const double v = p.B_val[e];
        const int loc = (v != 0.0) ? wbsearch(su, m, p.B_colidx[e]) : -1;
        if (loc >= 0 && cntB < capB) { brow_c[cntB * NPL + lane] = loc; brow_v[cntB * NPL + lane] = v; Bd[lane * MC + loc] = v; ++cntB; }
      }
    }
    const int a0 = wave_max_i32(cntA), a2 = wave_max_i32(cntB);
    WSYNC();
    double sc = 0.0;
    if (lane < n) {
      sc = hx[lane];
      for (int e = 0; e < a0; ++e) { const double v = arow_v[e * NPL + lane]; sc = __builtin_fma(v * v, hx[arow_c[e * NPL + lane]], sc); }
      for (int e = 0; e < a2; ++e) { const double v = brow_v[e * NPL + lane]; sc = __builtin_fma(v * v, hu[brow_c[e * NPL + lane]], sc); }
    }
    const double dl_ = p.delta_rel * wave_max_f64(sc);
    if (lane == 0) { nzs[0] = a0; nzs[2] = a2; red[2] = dl_; }
  } else {
    int cntAc = 0, cntBc = 0;
    if (lane < n) {
      const int g = sx[lane];
      for (int e = p.At_rowptr[g]; e < p.At_rowptr[g + 1]; ++e) {
        const double v = p.At_val[e];
        const int loc = (v != 0.0) ? wbsearch(sx, n, p.At_colidx[e]) : -1;
        if (loc >= 0 && cntAc < capAc) { acol_c[cntAc * NPL + lane] = loc; acol_v[cntAc * NPL + lane] = v; ++cntAc; }
      }
    }
    if (lane < m) {
      const int g = su[lane];
      for (int e = p.Bt_rowptr[g]; e < p.Bt_rowptr[g + 1]; ++e) {
        const double v = p.Bt_val[e];
        const int loc = (v != 0.0) ? wbsearch(sx, n, p.Bt_colidx[e]) : -1;
        if (loc >= 0 && cntBc < capBc) { bcol_c[cntBc * 64 + lane] = loc; bcol_v[cntBc * 64 + lane] = v; ++cntBc; }
      }
    }
    const int a1 = wave_max_i32(cntAc), a3 = wave_max_i32(cntBc);
    if (lane == 0) { nzs[1] = a1; nzs[3] = a3; }
  }
  if (lane < NPL) { tmp[lane] = 0.0; tmp2[lane] = 0.0; wl[lane] = 0.0; }
  wul[lane] = 0.0;
  __syncthreads();
  const int nzA = nzs[0], nzAc = nzs[1], nzB = nzs[2], nzBc = nzs[3];
  const double delta = red[2];
  lap(0);                          // setup (wave 1: waiting for it)

  constexpr int KR = 4;
  int arc[KR], acc_[KR];
  double arv[KR], acv[KR];
#pragma unroll
  for (int e = 0; e < KR; ++e) {
    const bool okr = e < capA, okc = e < capAc;
    arc[e] = okr ? arow_c[e * NPL + j] : 0;  arv[e] = okr ? arow_v[e * NPL + j] : 0.0;
    acc_[e] = okc ? acol_c[e * NPL + j] : 0; acv[e] = okc ? acol_v[e * NPL + j] : 0.0;
  }
  auto dotA_row = [&](const double* vec) -> double {
    // entries ≥ nzA of the register-cached list are (0, 0.0): no branches, the four gathers issue together
    static_assert(KR == 4, "written for four cached entries");
    const double v0 = vec[arc[0]], v1 = vec[arc[1]], v2 = vec[arc[2]], v3 = vec[arc[3]];
    double a = __builtin_fma(arv[1], v1, arv[0] * v0) + __builtin_fma(arv[3], v3, arv[2] * v2);
    for (int e = KR; e < nzA; ++e) a = __builtin_fma(arow_v[e * NPL + j], vec[arow_c[e * NPL + j]], a);
    return a;
  };
  auto dotA_col = [&](const double* vec) -> double {
    static_assert(KR == 4, "written for four cached entries");
    const double v0 = vec[acc_[0]], v1 = vec[acc_[1]], v2 = vec[acc_[2]], v3 = vec[acc_[3]];
    double a = __builtin_fma(acv[1], v1, acv[0] * v0) + __builtin_fma(acv[3], v3, acv[2] * v2);
    for (int e = KR; e < nzAc; ++e) a = __builtin_fma(acol_v[e * NPL + j], vec[acol_c[e * NPL + j]], a);
    return a;
  };
  auto wx_of = [&](int k) -> double { return (k >= 0 && k <= T - 1 && j < n && mask[k * nm + j]) ? hx[j] : 0.0; };

  // Z += X Q Xᵀ for a sparse X given by per-lane lists (rows of Ã: arow; rows of Ãᵀ: acol); Q comes from qfun(r, i)
  // ACC[r] += Σ_{e < min(nz, KR)} lv[e]·image[row r][lc[e]] for the register-cached list entries.  One straight-line body per entry
  // count (nz is wave-uniform): with a branch around every entry only the first one had its reads in flight together, the
  // others were issued one ds_read2 at a time behind an s_waitcnt (register pressure), six exposed LDS latencies per entry
  auto cached_product = [&](const int (&lc)[KR], const double (&lv)[KR], int nz, double (&ACC)[RPL]) {
    auto body = [&](auto ne_c) {
      constexpr int NE = decltype(ne_c)::value;
#pragma unroll
      for (int r = 0; r < RPL; ++r) {
        const double* row = mat + (HS * r + h) * LDM;
        double a = ACC[r];
#pragma unroll
        for (int e = 0; e < NE; ++e) a = __builtin_fma(lv[e], row[lc[e]], a);
        ACC[r] = a;
      }
    };
    const int ne = nz < KR ? nz : KR;
    if (ne == 3) body(std::integral_constant<int, 3>{});
    else if (ne == 4) body(std::integral_constant<int, 4>{});
    else if (ne == 2) body(std::integral_constant<int, 2>{});
    else if (ne == 1) body(std::integral_constant<int, 1>{});
  };
  auto sandwich = [&](auto qfun, const int (&lc)[KR], const double (&lv)[KR], const int32_t* lcl, const double* lvl,
                      int nz, double (&Z)[RPL]) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) mat[(HS * r + h) * LDM + j] = qfun(r, HS * r + h);
    WSYNC();
    double Y[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) Y[r] = 0.0;
    cached_product(lc, lv, nz, Y);
    for (int e = KR; e < nz; ++e) {
      const int cc = lcl[e * NPL + j]; const double v = lvl[e * NPL + j];
#pragma unroll
      for (int r = 0; r < RPL; ++r) Y[r] = __builtin_fma(v, mat[(HS * r + h) * LDM + cc], Y[r]);
    }
    WSYNC();
    if (j < NP) {
#pragma unroll
      for (int r = 0; r < RPL; ++r) mat[j * LDM + (HS * r + h)] = Y[r];
    }
    WSYNC();
    cached_product(lc, lv, nz, Z);
    for (int e = KR; e < nz; ++e) {
      const int cc = lcl[e * NPL + j]; const double v = lvl[e * NPL + j];
#pragma unroll
      for (int r = 0; r < RPL; ++r) Z[r] = __builtin_fma(v, mat[(HS * r + h) * LDM + cc], Z[r]);
    }
    WSYNC();
  };
  // Z += B̃ Wu_{ku} B̃ᵀ
  auto add_BWB = [&](int ku, double (&Z)[RPL]) {
    wul[lane] = (lane < m && mask[ku * nm + n + lane]) ? hu[lane] : 0.0;
    WSYNC();
    for (int e = 0; e < nzB; ++e) {
      const int cc = brow_c[e * NPL + j];
      const double v = brow_v[e * NPL + j] * wul[cc];
      double bq[RPL];
#pragma unroll
      for (int r = 0; r < RPL; ++r) bq[r] = Bd[(HS * r + h) * MC + cc];             // all RPL reads first: one latency, not RPL
#pragma unroll
      for (int r = 0; r < RPL; ++r) Z[r] = __builtin_fma(v, bq[r], Z[r]);
    }
    WSYNC();
  };
  auto publish_w = [&](int k) {   // wl[i] = Wx_k[i]
    if (h == 0) wl[j] = wx_of(k);
    WSYNC();
  };

  auto matvec = [&](const double (&Pk)[RPL]) -> double {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
      const double y = tmp2[HS * r + h];
      if ((r & 3) == 0) a0 = __builtin_fma(Pk[r], y, a0);
      else if ((r & 3) == 1) a1 = __builtin_fma(Pk[r], y, a1);
      else if ((r & 3) == 2) a2 = __builtin_fma(Pk[r], y, a2);
      else a3 = __builtin_fma(Pk[r], y, a3);
    }
    double part = (a0 + a1) + (a2 + a3);
    if (HS >= 2) part = xsum32(part);
    if (HS >= 4) part = xsum16(part);
    return part;
  };
  auto load_P = [&](int k, double (&Pk)[RPL]) {
    if constexpr (PL) {
      const double* b = pl + k * n * n;
#pragma unroll
      for (int r = 0; r < RPL; ++r) Pk[r] = b[p_off(r)];
    } else {
#pragma unroll
      for (int r = 0; r < RPL; ++r) Pk[r] = fac[((int64_t)k * RPL + r) * 64 + lane];
    }
  };
  auto store_P = [&](int k, const double (&Pk)[RPL]) {
    if constexpr (PL) {
      double* b = pl + k * n * n;
#pragma unroll
      for (int r = 0; r < RPL; ++r) if (p_ok(r)) b[p_off(r)] = Pk[r];
    } else {
#pragma unroll
      for (int r = 0; r < RPL; ++r) fac[((int64_t)k * RPL + r) * 64 + lane] = Pk[r];
    }
  };

  // in-register Gauss–Jordan (same schedule as the one-wave kernel)
  auto gauss_jordan = [&](double (&M)[RPL]) {
    double dnext = fast_rcp(readlane_f64(M[0], 0));
    static_for<NP>([&](auto pv_c) {
      constexpr int pv = decltype(pv_c)::value;
      constexpr int rp = pv / HS, hp = pv % HS;
      if (pv < n) {
        constexpr int rn = (pv + 1) / HS, hn = (pv + 1) % HS;
        const double d = dnext;
        double cc[RPL];
#pragma unroll
        for (int r = 0; r < RPL; ++r) cc[r] = group_bcast<NPL, pv>(M[r]);
        double rowj;
        if (HS == 1) rowj = M[rp];
        else if (HS == 2) {
          const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(M[rp]), __double2loint(M[rp]), false, false);
          const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(M[rp]), __double2hiint(M[rp]), false, false);
          rowj = __hiloint2double(hi[hp], lo[hp]);
        } else {
          const int arow = (hp * NPL + j) << 2;
          rowj = __hiloint2double(__builtin_amdgcn_ds_bpermute(arow, __double2hiint(M[rp])),
                                  __builtin_amdgcn_ds_bpermute(arow, __double2loint(M[rp])));
        }
        double xr = 0.0;
        constexpr bool have_next = rn < RPL;
        if constexpr (have_next) {
          const double a_nn = readlane_f64(M[rn], hn * NPL + pv + 1);
          const double a_np = readlane_f64(M[rn], hn * NPL + pv);
          const double a_pn = readlane_f64(M[rp], hp * NPL + pv + 1);
          const double pn = __builtin_fma(-(a_np * d), a_pn, a_nn);
          xr = __builtin_amdgcn_rcp(pn);
          if (SLS_GJ_NR >= 1) xr = __builtin_fma(__builtin_fma(-pn, xr, 1.0), xr, xr);
          if (SLS_GJ_NR >= 2) xr = __builtin_fma(__builtin_fma(-pn, xr, 1.0), xr, xr);
        }
        const double tj = rowj * d;
        const double tfix = (j == pv) ? (1.0 + d) : tj;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < RPL; ++r) M[r] = __builtin_fma(-cc[r], tfix, M[r]);
        if (h == hp) M[rp] = (j == pv) ? d : tj;
        if constexpr (have_next) dnext = xr;
      }
    });
  };

  unsigned long long gj_cycles = 0, res_cycles = 0;
  auto invert_block = [&](double (&M)[RPL]) {
    unsigned long long t0 = 0;
    if (p.dbg_level >= 3) { __builtin_amdgcn_sched_barrier(0); t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    if constexpr (TILED) gauss_jordan_tiled<NPL, RPL, LDT>(M, mat, lane, n); else gauss_jordan(M);
    if (p.dbg_level >= 3) { __builtin_amdgcn_sched_barrier(0); gj_cycles += __builtin_amdgcn_s_memtime() - t0; __builtin_amdgcn_sched_barrier(0); }
  };

  // forward Schur block (blocks 0..c): M holds P_{k−1} on entry (k ≥ 1), D'_k on exit
  auto build_up = [&](int k, double (&M)[RPL]) {
    const double wcur = wx_of(k);
    if (k == 0) {
#pragma unroll
      for (int r = 0; r < RPL; ++r) M[r] = (HS * r + h == j) ? (delta + wcur) : 0.0;
      return;
    }
    const double wj = wx_of(k - 1);
    publish_w(k - 1);
    double Z[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) Z[r] = (HS * r + h == j) ? (delta + wcur) : 0.0;
    sandwich([&](int r, int i) { return wl[i] * (((i == j) ? 1.0 : 0.0) - M[r] * wj); }, arc, arv, arow_c, arow_v, nzA, Z);
    add_BWB(k - 1, Z);
#pragma unroll
    for (int r = 0; r < RPL; ++r) M[r] = Z[r];
  };

  // r = f − E z(λ): x_t, u_t first (any order), then r_t = f_t − x_t + Ãx_{t−1} + B̃u_{t−1}; 2·HS time steps per round
  auto x_of = [&](int t) -> double {
    const double l0 = lam[t * NPL + j];
    const uint8_t mk = mask[t * nm + j];
    const double a = dotA_col(lam + (t + 1) * NPL);
    const double v = hx[j] * (l0 - a - gx[j]);
    return mk ? v : 0.0;
  };
  // the ñu inputs use few lanes (README: 8 of 64): lanes are (time slot, input) pairs, 64/MP time steps per instruction
  int lgMP = 0;
  while ((1 << lgMP) < m) ++lgMP;
  const int uq = lane & ((1 << lgMP) - 1), uts = lane >> lgMP, NTS = 64 >> lgMP;
  auto u_of = [&](int t) -> double {            // input uq at time t
    const double* l1 = lam + (t + 1) * NPL;
    double a = 0.0;
    for (int e = 0; e < nzBc; ++e) a = __builtin_fma(bcol_v[e * 64 + uq], l1[bcol_c[e * 64 + uq]], a);
    return mask[t * nm + n + uq] ? hu[uq] * (-a - gu[uq]) : 0.0;
  };
  constexpr int G = 2 * HS;
  const int gid = wv * HS + h;
  unsigned long long seg[4] = {0, 0, 0, 0}, segt = 0;
  bool xu_valid = false;                         // xs/us hold z(λ) of the current λ (set by the residual pass)
  auto residual_pass = [&]() -> double {       // all 128 threads; contains workgroup barriers
    const bool live = j < n;
    xu_valid = true;
    if (p.dbg_level >= 5) segt = __builtin_amdgcn_s_memtime();
    const int nit = (T + G) / G;                  // wave-uniform trip count (t = gid + it·G covers 0..T): the loop unrolls and
#pragma unroll 2                                  // the LDS gathers of consecutive time steps overlap
    for (int it = 0; it < nit; ++it) {
      const int t = gid + it * G;
      const double v = (t < T && live) ? x_of(min(t, T - 1)) : 0.0;
      if (t <= T) xs[t * NPL + j] = v;
    }
    if (uq < m) {
#pragma unroll 2
      for (int t = wv * NTS + uts; t < T; t += 2 * NTS) us[t * MC + uq] = u_of(t);
    }
    if (p.dbg_level >= 5) { const unsigned long long nw = __builtin_amdgcn_s_memtime(); seg[0] += nw - segt; segt = nw; }
    __syncthreads();
    if (p.dbg_level >= 5) { const unsigned long long nw = __builtin_amdgcn_s_memtime(); seg[1] += nw - segt; segt = nw; }
    double rmax = 0.0;
#pragma unroll 2
    for (int it = 0; it < nit; ++it) {
      const int t = gid + it * G;
      const int tc_ = min(t, T), tp = max(tc_ - 1, 0);            // clamped: every lane computes, only valid (t, j) store
      double a = (t == 0 && j == sd.pos) ? 1.0 : 0.0;
      a -= xs[tc_ * NPL + j];
      double b = dotA_row(xs + tp * NPL);
      const double* up = us + tp * MC;
      for (int e = 0; e < nzB; ++e) b = __builtin_fma(brow_v[e * NPL + j], up[brow_c[e * NPL + j]], b);
      if (t >= 1) a += b;
      if (live && t <= T) {
        rmax = resid_max(rmax, a);
        rq[t * NPL + j] = a;
      }
    }
    if (p.dbg_level >= 5) { const unsigned long long nw = __builtin_amdgcn_s_memtime(); seg[2] += nw - segt; segt = nw; }
    rmax = wave_max_f64(rmax);
    if (lane == 0) red[wv] = rmax;
    __syncthreads();
    const double r2 = fmax(red[0], red[1]);
    __syncthreads();
    if (p.dbg_level >= 5) { const unsigned long long nw = __builtin_amdgcn_s_memtime(); seg[3] += nw - segt; segt = nw; }
    return r2;
  };
  // z(λ) is what the last residual pass left in xs/us (λ has not moved since); destinations are fetched eight at a time so
  // that their global loads overlap instead of each store waiting for its own (17 k → ≈4 k cycles on the README chain)
  auto output_pass = [&]() {
    constexpr int CH = 8;
    if (j < n) {
      for (int t0 = gid; t0 < T; t0 += G * CH) {
        int dd[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int t = t0 + q * G; dd[q] = (t < T && mask[t * nm + j]) ? dest[t * nm + j] : -1; }
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int t = t0 + q * G; if (dd[q] >= 0) p.out[dd[q]] = xu_valid ? xs[t * NPL + j] : 0.0; }
      }
    }
    if (uq < m) {
      for (int t0 = wv * NTS + uts; t0 < T; t0 += 2 * NTS * CH) {
        int dd[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int t = t0 + q * 2 * NTS; dd[q] = (t < T && mask[t * nm + n + uq]) ? dest[t * nm + n + uq] : -1; }
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int t = t0 + q * 2 * NTS; if (dd[q] >= 0) p.out[dd[q]] = xu_valid ? us[t * MC + uq] : 0.0; }
      }
    }
  };

  double resid;
  if (sd.has_w) {
    resid = residual_pass();
  } else {
    if (threadIdx.x == 0 && sd.pos >= 0) rq[sd.pos] = 1.0;
    __syncthreads();
    resid = (sd.pos >= 0) ? 1.0 : 0.0;
  }
  int iters = 0, status = 0;

  // one elimination step of the upward / downward wave on block k with P_k in registers; q_k → rq[k]
  auto elim_up = [&](int k, const double (&Pk)[RPL], double wk) {
    if (lane < NPL) {
      double a = rq[k * NPL + lane];
      if (k >= 1) a += dotA_row(tmp);                        // tmp = Wx_{k−1} q_{k−1}
      tmp2[lane] = (lane < n) ? a : 0.0;
    }
    WSYNC();
    const double q = matvec(Pk);
    if (lane < NPL) { rq[k * NPL + lane] = q; tmp[lane] = wk * q; }
    WSYNC();
  };
  auto elim_down = [&](int k, const double (&Pk)[RPL], double wk) {
    if (lane < NPL) {
      double a = rq[k * NPL + lane];
      if (k < T) a += wk * dotA_col(tmp);                    // tmp = q_{k+1}
      tmp2[lane] = (lane < n) ? a : 0.0;
    }
    WSYNC();
    const double q = matvec(Pk);
    if (lane < NPL) { rq[k * NPL + lane] = q; tmp[lane] = q; }
    WSYNC();
  };
  // middle block (wave 0): y_c = r_c + ÃWx_{c−1}q_{c−1} + Wx_c Ãᵀ q_{c+1};  Δλ_c = P_c y_c
  auto middle = [&](const double (&Pc)[RPL]) {
    if (lane < NPL) {
      double a = rq[c * NPL + lane];
      if (c >= 1) a += dotA_row(tmp);
      a += wx_of(c) * dotA_col(rq + (c + 1) * NPL);
      tmp2[lane] = (lane < n) ? a : 0.0;
    }
    WSYNC();
    const double dl = matvec(Pc);
    if (lane < NPL) { rq[c * NPL + lane] = dl; lam[c * NPL + lane] += dl; }
    WSYNC();
  };
  // outward substitution from the middle: wave 0 towards block 0, wave 1 towards block T
  auto outward = [&]() {
    double Pk[RPL];
    if (wv == 0) {
      for (int k = c - 1; k >= 0; --k) {
        load_P(k, Pk);
        if (lane < NPL) {
          double a = 0.0;
          if (lane < n && mask[k * nm + lane]) a = hx[lane] * dotA_col(rq + (k + 1) * NPL);
          tmp2[lane] = a;
        }
        WSYNC();
        const double dl = matvec(Pk) + ((lane < NPL) ? rq[k * NPL + lane] : 0.0);
        if (lane < NPL) { rq[k * NPL + lane] = dl; lam[k * NPL + lane] += dl; }
        WSYNC();
      }
    } else {
      for (int k = c + 1; k <= T; ++k) {
        load_P(k, Pk);
        if (lane < NPL) tmp[lane] = wx_of(k - 1) * rq[(k - 1) * NPL + lane];     // Wx_{k−1} Δλ_{k−1}
        WSYNC();
        if (lane < NPL) tmp2[lane] = (lane < n) ? dotA_row(tmp) : 0.0;
        WSYNC();
        const double dl = matvec(Pk) + ((lane < NPL) ? rq[k * NPL + lane] : 0.0);
        if (lane < NPL) { rq[k * NPL + lane] = dl; lam[k * NPL + lane] += dl; }
        WSYNC();
      }
    }
  };

  if (resid > p.tol) {
    double M[RPL];
    // ---------------- factor from both ends, first elimination sweep fused ----------------
    if (wv == 0) {
      for (int k = 0; k < c; ++k) {
        build_up(k, M);
        invert_block(M);
        store_P(k, M);
        elim_up(k, M, wx_of(k));
      }
    } else {
      double S[RPL];                       // static part δI + Wx_k + ÃWx_{k−1}Ãᵀ + B̃Wu_{k−1}B̃ᵀ, cached while the masks repeat
      bool have_S = false;
      for (int k = T; k > c; --k) {
        // masks of this block equal those of block k+1 (processed just before)?  wave-uniform test
        bool same = have_S && (k + 1 <= T - 1);
        if (same) {
          bool eq = true;
          if (lane < nm) {
            eq = (mask[k * nm + lane] == mask[(k + 1) * nm + lane]) && (mask[(k - 1) * nm + lane] == mask[k * nm + lane]);
          }
          same = __all(eq);
        }
        if (!same) {
          const double wcur = wx_of(k);
          publish_w(k - 1);
#pragma unroll
          for (int r = 0; r < RPL; ++r) S[r] = (HS * r + h == j) ? (delta + wcur) : 0.0;
          sandwich([&](int r, int i) { (void)r; return (i == j) ? wl[i] : 0.0; }, arc, arv, arow_c, arow_v, nzA, S);
          add_BWB(k - 1, S);
          have_S = true;
        }
        if (k == T) {
#pragma unroll
          for (int r = 0; r < RPL; ++r) M[r] = S[r];
        } else {
          double Z[RPL];
#pragma unroll
          for (int r = 0; r < RPL; ++r) Z[r] = 0.0;
          sandwich([&](int r, int i) { (void)i; return M[r]; }, acc_, acv, acol_c, acol_v, nzAc, Z);     // ÃᵀP_{k+1}Ã
          publish_w(k);
          const double wj = wx_of(k);
#pragma unroll
          for (int r = 0; r < RPL; ++r) M[r] = S[r] - wl[HS * r + h] * Z[r] * wj;
        }
        invert_block(M);
        store_P(k, M);
        elim_down(k, M, wx_of(k));
      }
      // hand W_c(ÃᵀP_{c+1}Ã)W_c to wave 0 for the middle block
      double Z[RPL];
#pragma unroll
      for (int r = 0; r < RPL; ++r) Z[r] = 0.0;
      sandwich([&](int r, int i) { (void)i; return M[r]; }, acc_, acv, acol_c, acol_v, nzAc, Z);
      publish_w(c);
      const double wj = wx_of(c);
#pragma unroll
      for (int r = 0; r < RPL; ++r) xch[(HS * r + h) * LDM + j] = wl[HS * r + h] * Z[r] * wj;
    }
    lap(1);                        // own half of the factorisation (+ fused elimination)
    __syncthreads();
    lap(2);                        // waiting for the other wave
    if (wv == 0) {
      build_up(c, M);                                          // M held P_{c−1}
#pragma unroll
      for (int r = 0; r < RPL; ++r) M[r] -= xch[(HS * r + h) * LDM + j];
      invert_block(M);
      store_P(c, M);
      middle(M);
    }
    __syncthreads();
    outward();
    __syncthreads();
    lap(3);                        // middle block + outward substitution

    // ---------------- multiplier iteration ----------------
    double prev = resid, prev2 = resid;
    int itmax = p.max_iters;
    for (int it = 1; it <= itmax; ++it) {
      iters = it;
      if (it > 1) {
        double Pk[RPL];
        if (wv == 0) {
          for (int k = 0; k < c; ++k) { load_P(k, Pk); elim_up(k, Pk, wx_of(k)); }
        } else {
          for (int k = T; k > c; --k) { load_P(k, Pk); elim_down(k, Pk, wx_of(k)); }
        }
        __syncthreads();
        if (wv == 0) { load_P(c, Pk); middle(Pk); }
        __syncthreads();
        outward();
        __syncthreads();
      }
      unsigned long long tr0 = 0;
      if (p.dbg_level >= 4) tr0 = __builtin_amdgcn_s_memtime();
      resid = residual_pass();
      if (p.dbg_level >= 4) res_cycles += __builtin_amdgcn_s_memtime() - tr0;
      if (resid <= p.tol) break;
      if (it >= 2 && resid > p.stag * prev) {
        if (!(p.max_iters_slow > 0 && (resid > p.tol_ok || itmax > p.max_iters) && still_contracting(it >= 3 ? prev2 : prev, prev, resid))) { status = 1; break; }
        itmax = max(itmax, p.max_iters_slow);
      }
      prev2 = prev; prev = resid;
    }
    if (resid <= p.tol_ok) status = 0;
    else if (status == 0) status = 2;
  }
  unsigned long long to0 = 0;
  if (p.dbg_level >= 4) to0 = __builtin_amdgcn_s_memtime();
  output_pass();
  if (p.dbg && lane == 0) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    if (p.dbg_level >= 5) { tc[0] = seg[0]; tc[1] = seg[1]; tc[2] = seg[2]; tc[3] = seg[3]; }
    else if (p.dbg_level >= 4) { tc[0] = res_cycles; tc[2] = now - to0; }    // level 4: slot 0 = residual passes, slot 2 = output pass
    else if (p.dbg_level >= 3) tc[0] = gj_cycles;                     // level 3: slot 0 reports the Gauss–Jordan share of the factor half instead of the setup
    for (int q = 0; q < 4; ++q) p.dbg[sd.out_index * 8 + wv * 4 + q] = (q == 3) ? tc[3] : tc[q];
    p.dbg[sd.out_index * 8 + wv * 4 + 3] = (tc[3] << 32) | ((now - tlast) & 0xffffffffull);   // [3]: hi = middle+outward, lo = passes 2.. + output
  }
  if (sd.pos < 0 && status == 0) status = 3;
  if (threadIdx.x == 0) {
    p.status[sd.out_index] = status;
    p.resid[sd.out_index] = resid;
    p.iters[sd.out_index] = iters;
  }
}

template <int NPL, int RPL, bool PL>
__global__ __launch_bounds__(128, 1) void h2_column_twisted_kernel(const KernelParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double* fac = p.fac_ws + (int64_t)blockIdx.x * p.fac_stride;
  for (int s = blockIdx.x; s < p.nsub; s += gridDim.x) {
    const SubDesc sd = p.subs[p.order[p.order_off + s]];
    twisted_solve_column<NPL, RPL, PL>(p, sd, fac, lds_raw);
  }
}

template <int NPL, int RPL, bool PL>
static hipError_t launch_one_twisted_v(const KernelParams& p, int grid, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&h2_column_twisted_kernel<NPL, RPL, PL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((h2_column_twisted_kernel<NPL, RPL, PL>), dim3(grid), dim3(128), lds, st, p);
  return hipGetLastError();
}
template <int NPL, int RPL>
static hipError_t launch_one_twisted(const KernelParams& p, int grid, size_t lds, hipStream_t st) {
  return p.w_pl_off > 0 ? launch_one_twisted_v<NPL, RPL, true>(p, grid, lds, st)
                        : launch_one_twisted_v<NPL, RPL, false>(p, grid, lds, st);
}

hipError_t launch_twisted(int cls, const KernelParams& p, int grid, size_t lds, hipStream_t st) {
  switch (cls) {
    case 0: return launch_one_twisted<16, 3>(p, grid, lds, st);
    case 1: return launch_one_twisted<16, 4>(p, grid, lds, st);
    case 2: return launch_one_twisted<32, 10>(p, grid, lds, st);
    case 3: return launch_one_twisted<32, 12>(p, grid, lds, st);
    case 4: return launch_one_twisted<32, 14>(p, grid, lds, st);
    case 5: return launch_one_twisted<32, 16>(p, grid, lds, st);
    default: return hipErrorInvalidValue;
  }
}

template <int NPL, int RPL, bool VG>
static hipError_t launch_one_v(const KernelParams& p, int grid, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&h2_column_wave_kernel<NPL, RPL, VG>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((h2_column_wave_kernel<NPL, RPL, VG>), dim3(grid), dim3(64), lds, st, p);
  return hipGetLastError();
}
template <int NPL, int RPL>
static hipError_t launch_one(const KernelParams& p, int grid, size_t lds, hipStream_t st) {
  if constexpr (NPL <= 32) {
    if (p.objective == 1) {                  // sum-of-norms build: ñx ≤ 32, vectors in the global workspace
      if (p.vec_in_lds != 0) return hipErrorInvalidValue;
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&h2_column_wave_kernel<NPL, RPL, true, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((h2_column_wave_kernel<NPL, RPL, true, true>), dim3(grid), dim3(64), lds, st, p);
      return hipGetLastError();
    }
    if (p.vec_in_lds == 0) return launch_one_v<NPL, RPL, true>(p, grid, lds, st);
  }
  if (p.objective == 1) return hipErrorInvalidValue;
  return launch_one_v<NPL, RPL, false>(p, grid, lds, st);
}

// one kernel per size class (a merged multi-class kernel makes the register allocator spill: 604 B scratch per
// lane measured); ragged batches run their classes concurrently on separate streams (sls_api.cpp)
hipError_t launch_wave(int cls, const KernelParams& p, int grid, size_t lds, hipStream_t st) {
  switch (cls) {
    case 0: return launch_one<16, 3>(p, grid, lds, st);    // n ≤ 12
    case 1: return launch_one<16, 4>(p, grid, lds, st);    // n ≤ 16
    case 2: return launch_one<32, 10>(p, grid, lds, st);   // n ≤ 20
    case 3: return launch_one<32, 12>(p, grid, lds, st);   // n ≤ 24
    case 4: return launch_one<32, 14>(p, grid, lds, st);   // n ≤ 28
    case 5: return launch_one<32, 16>(p, grid, lds, st);   // n ≤ 32
    case 6: return launch_one<64, 40>(p, grid, lds, st);   // n ≤ 40
    case 7: return launch_one<64, 48>(p, grid, lds, st);   // n ≤ 48
    case 8: return launch_one<64, 64>(p, grid, lds, st);   // n ≤ 64
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sls
