// sls_twisted4_kernel.hip — LATENCY regime of the H2 column solve, round 3: FOUR waves per subproblem (gfx950).
//
// The two-wave twisted kernel (sls_wave_kernel.hip) halves the serial chain over the T+1 block rows: wave 0 eliminates
// upwards from block 0, wave 1 downwards from block T.  What is left on each wave's chain per block is
//     build D'_k from P_{k∓1} (two sparse products through an LDS image: ≈4.8 k cycles)  →  Gauss–Jordan (≈7.0 k)  →  fused sweep,
// strictly one after the other, on ONE of the CU's four SIMDs while two stand idle (README chain: 59 columns on 256 CUs).
//
// Here each direction gets a HELPER wave on its own SIMD that takes the build off the chain.  With
//     S_k = δI + W_k + B̃ Wu_{k−1} B̃ᵀ + Ã W_{k−1} Ãᵀ            (the diagonal block of E H⁻¹ Eᵀ + δI: static, no P in it)
//     upward:    D'_k = S_k − X P_{k−1} Xᵀ,   X = Ã W_{k−1}        downward:  G_k = S_k − X P_{k+1} Xᵀ,   X = W_k Ãᵀ
// the next block is the Schur complement of the bordered matrix [[G, Xᵀ], [X, S]] on G = the block being inverted RIGHT NOW:
// eliminating G's pivots from the border produces S − X G⁻¹ Xᵀ in place of S.  The chain wave inverts G by the same tiled
// Gauss–Jordan as before and, per pivot, drops the pivot row (its pre-update values) and 1/pivot into LDS; the helper —
// one or two pivots behind — applies that pivot to its own tiles of X and S (6 ds_swizzle + 6 ds_bpermute + 18 FMA, no
// reciprocal chain, ≈250 cycles against the chain wave's ≈330).  When the chain wave has finished block k (store + fused
// sweep), block k+1 is waiting for it in LDS in exactly the lane layout its Gauss–Jordan starts from.  Chain per block:
// Gauss–Jordan + store + sweep; the sparse products are gone (the helper never forms P·anything: same flops as the explicit
// products, but off the chain).  The middle block takes both helpers' contributions.
// The elimination form is as stable as the explicit one: for a PSD bordered matrix the border lies in range(G), so along a
// (near-)null direction of G + δI the border column is O(eps) and enters the update as (eps)²/δ — the same quadratic
// term the explicit form has through Q = W − W P W.
// Hand-offs go through LDS flags (monotone sequence numbers; DS operations of one wave execute in program order, so a
// flag written after the data is seen after the data): chain → helper per pivot, helper → chain per block.  All four waves
// of the workgroup are resident (one workgroup per CU), every wait is on a wave that never waits for the waiter.
// Everything else — setup, residual passes, output — is split four ways instead of two; sweeps, middle block, multiplier
// iteration, status words and the P_k workspace layout are those of the two-wave kernel (bitwise the same mathematics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>
#include "sls_device.h"
#include "sls_wave_util.h"

namespace sls {

#ifndef SLS_T4_EXP
#define SLS_T4_EXP 0          // timing experiments (WRONG results on purpose): 1 = helper does not eliminate, 2 = helper without cross-lane traffic, 3 = without ds_bpermute
#endif

// meeting block.  The upward side carries the mask ramp (its helper recomputes the static part for every block while the
// masks still grow, the downward side's masks repeat) and the middle block's own inversion: it gets the shorter half.
#ifndef SLS_T4_MIDDLE_SHIFT
#define SLS_T4_MIDDLE_SHIFT 0
#endif
static inline __host__ __device__ int twisted4_middle(int T) { return T >= 7 ? (T + 1) / 2 - SLS_T4_MIDDLE_SHIFT : (T - 1) / 2; }

// LDS flags between the waves of one workgroup: RELAXED workgroup-scope atomics are plain ds_read/ds_write (a `volatile` access
// makes the backend wait for every outstanding LDS and global operation after it — ≈250 cycles per pivot on the chain wave,
// measured); ordering comes from the hardware (DS operations of one wave execute in program order) plus compiler barriers.
__device__ __forceinline__ int flag_load(const int* f) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void flag_store(int* f, int v) {
  asm volatile("" ::: "memory");
  __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wait_flag(const int* f, int target) {
  for (;;) {
    if (flag_load(f) >= target) break;
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}

// Broadcast inside each 8-lane group of the lane grid (lane = 8·a + b → the value of lane (a, QA)) WITHOUT the LDS crossbar:
// two DPP row_newbcast moves per 32-bit half, one per half of the 16-lane DPP row (bank masks 0x3 / 0xc).  ds_swizzle does the
// same in one DS instruction per half.  Measured SLOWER (README launch 0.1174 → 0.1239 ms): the waves are bound by instruction
// issue, and four VALU moves per value cost more issue slots than two ds_swizzle; kept as an A/B switch.
#ifndef SLS_T4_SAME
#define SLS_T4_SAME 1          // reuse of the static part of a block while the masks repeat (0: always rebuilt; diagnostics)
#endif
#ifndef SLS_T4_DPP
#define SLS_T4_DPP 0
#endif
template <int QA>
__device__ __forceinline__ double bcast8(double v) {
  if constexpr (SLS_T4_DPP != 0) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    int l2 = __builtin_amdgcn_update_dpp(0, lo, 0x150 + QA, 0xf, 0x3, false);
    l2 = __builtin_amdgcn_update_dpp(l2, lo, 0x150 + 8 + QA, 0xf, 0xc, false);
    int h2 = __builtin_amdgcn_update_dpp(0, hi, 0x150 + QA, 0xf, 0x3, false);
    h2 = __builtin_amdgcn_update_dpp(h2, hi, 0x150 + 8 + QA, 0xf, 0xc, false);
    return __hiloint2double(h2, l2);
  } else {
    constexpr int pattern = 0x18 | (QA << 5);
    return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), pattern), __builtin_amdgcn_ds_swizzle(__double2loint(v), pattern));
  }
}

// LDS byte address of a pointer into the workgroup's LDS (the low half of the flat address)
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)p; }

// Stores by EIGHT lanes of the wave without a divergent region: EXEC is set by two scalar moves around the DS instructions
// (the eight lanes are known at compile time) and restored to all lanes — the code around these calls runs with every lane
// active.  A compiler-generated `if (lane ∈ set) store` cost the chain wave ≈150 cycles per pivot (v_cmp → s_and_saveexec →
// branch), 64-lane stores to junk addresses loaded the LDS pipe that all four waves' cross-lane traffic shares.
// s_lshl_b64 writes SCC, hence the "scc" clobber.  The compiler does not count these DS operations in its s_waitcnt bookkeeping; LDS operations complete in order, so every
// wait it emits for its own operations only becomes stricter.
// Lanes 8·A … 8·A+7 (one row of the lane grid): TR doubles to addr + off0 + 64·q, one more to addr_d + offd, a flag word.
typedef double d2_t __attribute__((ext_vector_type(2)));
// Packed layout of a published pivot: per lane-grid column b a record of SB doubles at rowbuf[pv·RS + SB·b]:
//   TR = 3: {row·d [b], row·d [b+8], row·d [b+16], d}        (SB = 4: two ds_write_b128)
//   TR = 4: {row·d [b], [b+8], [b+16], [b+24], d, –}          (SB = 6: two ds_write_b128 + one ds_write_b64)
// then the flag word: three (four) DS instructions per pivot instead of five (six) — each costs a lone wave ≈ 12 issue cycles.
template <int A, int O0>
__device__ __forceinline__ void store_row8_3(unsigned addr, double v0, double v1, double v2, double d, unsigned addr_flag, int flagval) {
  const d2_t q0 = {v0, v1}, q1 = {v2, d};
  asm volatile("s_mov_b64 exec, 0xff\n\ts_lshl_b64 exec, exec, %[sh]\n\t"
               "ds_write_b128 %[a], %[q0] offset:%[o0]\n\tds_write_b128 %[a], %[q1] offset:%[o1]\n\t"
               "ds_write_b32 %[af], %[vf]\n\ts_mov_b64 exec, -1"
               :: [sh] "n"(8 * A), [a] "v"(addr), [q0] "v"(q0), [q1] "v"(q1), [af] "v"(addr_flag), [vf] "v"(flagval),
                  [o0] "n"(O0), [o1] "n"(O0 + 16) : "memory", "scc");
}
template <int A, int O0>
__device__ __forceinline__ void store_row8_4(unsigned addr, double v0, double v1, double v2, double v3, double d, unsigned addr_flag, int flagval) {
  const d2_t q0 = {v0, v1}, q1 = {v2, v3};
  asm volatile("s_mov_b64 exec, 0xff\n\ts_lshl_b64 exec, exec, %[sh]\n\t"
               "ds_write_b128 %[a], %[q0] offset:%[o0]\n\tds_write_b128 %[a], %[q1] offset:%[o1]\n\tds_write_b64 %[a], %[vd] offset:%[o2]\n\t"
               "ds_write_b32 %[af], %[vf]\n\ts_mov_b64 exec, -1"
               :: [sh] "n"(8 * A), [a] "v"(addr), [q0] "v"(q0), [q1] "v"(q1), [vd] "v"(d), [af] "v"(addr_flag), [vf] "v"(flagval),
                  [o0] "n"(O0), [o1] "n"(O0 + 16), [o2] "n"(O0 + 32) : "memory", "scc");
}
// Lanes 0, 8, 16, … 56 (column 0 of the lane grid): TR doubles to addr + O0 + 64·q
template <int O0>
__device__ __forceinline__ void store_col8_3(unsigned addr, double v0, double v1, double v2) {
  asm volatile("s_mov_b32 exec_lo, 0x01010101\n\ts_mov_b32 exec_hi, 0x01010101\n\t"
               "ds_write_b64 %[a], %[v0] offset:%[o0]\n\tds_write_b64 %[a], %[v1] offset:%[o1]\n\tds_write_b64 %[a], %[v2] offset:%[o2]\n\t"
               "s_mov_b64 exec, -1"
               :: [a] "v"(addr), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [o0] "n"(O0), [o1] "n"(O0 + 64), [o2] "n"(O0 + 128) : "memory");
}
template <int O0>
__device__ __forceinline__ void store_col8_4(unsigned addr, double v0, double v1, double v2, double v3) {
  asm volatile("s_mov_b32 exec_lo, 0x01010101\n\ts_mov_b32 exec_hi, 0x01010101\n\t"
               "ds_write_b64 %[a], %[v0] offset:%[o0]\n\tds_write_b64 %[a], %[v1] offset:%[o1]\n\tds_write_b64 %[a], %[v2] offset:%[o2]\n\tds_write_b64 %[a], %[v3] offset:%[o3]\n\t"
               "s_mov_b64 exec, -1"
               :: [a] "v"(addr), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [o0] "n"(O0), [o1] "n"(O0 + 64), [o2] "n"(O0 + 128), [o3] "n"(O0 + 192) : "memory");
}

// Tiled Gauss–Jordan of the chain wave on the 8×8 lane grid (lane (a, b) holds {rows a + 8·ri} × {columns b + 8·cj}), in place
// on the tiles, publishing every pivot's row and reciprocal for the helper wave: rowbuf[pv] = eight packed records (RS doubles per pivot, see store_row8_*), then *flag = seqbase + pv + 1.
template <int NP, int TR, int RS>
__device__ __forceinline__ void gj_tiles_publish(double (&Tt)[TR * TR], const int lane, const int n, double* rowbuf, int* flag,
                                                 const int seqbase, double* junk) {
  int ta = lane >> 3, tb = lane & 7, nn = __builtin_amdgcn_readfirstlane(n);
  asm volatile("" : "+v"(ta), "+v"(tb), "+s"(nn));
  double dnext = fast_rcp(readlane_f64(Tt[0], 0));
  constexpr int SB = (TR == 3) ? 4 : 6;                   // doubles per published record (see store_row8_*)
  static_assert(RS == 8 * SB, "row stride = eight records");
  const unsigned pub_addr = lds_addr(rowbuf + SB * tb), flag_addr = lds_addr(flag);
  (void)junk;
  double col[TR], row[TR];
  auto fetch = [&](auto q_c) {
    constexpr int q = decltype(q_c)::value;
    constexpr int qa = q % 8, qs = q / 8;
#pragma unroll
    for (int ri = 0; ri < TR; ++ri) {
      col[ri] = bcast8<qa>(Tt[ri * TR + qs]);
    }
    const int src = (qa * 8 + tb) << 2;
#pragma unroll
    for (int cj = 0; cj < TR; ++cj) {
      const double v = Tt[qs * TR + cj];
      row[cj] = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(v)),
                                 __builtin_amdgcn_ds_bpermute(src, __double2loint(v)));
    }
  };
  fetch(std::integral_constant<int, 0>{});
  static_for<NP>([&](auto pv_c) {
    constexpr int pv = decltype(pv_c)::value;
    constexpr int pa = pv % 8, ps = pv / 8;
    if (pv < nn) {
      const double d = dnext;
      constexpr bool have_next = pv + 1 < NP;
      constexpr int na = (pv + 1) % 8, ns = have_next ? (pv + 1) / 8 : ps;
      double xr = 0.0;
      if constexpr (have_next) {
        const double a_nn = readlane_f64(Tt[ns * TR + ns], na * 8 + na);
        const double a_pn = readlane_f64(Tt[ps * TR + ns], pa * 8 + na);
        const double pn = __builtin_fma(-(a_pn * d), a_pn, a_nn);
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
        tfix[cj] = (cj == ps && tb == pa) ? (1.0 + d) : tj[cj];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ri = 0; ri < TR; ++ri) {
#pragma unroll
        for (int cj = 0; cj < TR; ++cj)
          if (ri == ns || cj == ns || ri == ps) Tt[ri * TR + cj] = __builtin_fma(-c0[ri], tfix[cj], Tt[ri * TR + cj]);
      }
      const bool own = ta == pa;
#pragma unroll
      for (int cj = 0; cj < TR; ++cj) {                     // (selects, not a branch: a register write inside a divergent region made the
        const double nv = (cj == ps && tb == pa) ? d : tj[cj];   //  allocator copy the whole tile array around it, ≈30 v_mov per pivot)
        Tt[ps * TR + cj] = own ? nv : Tt[ps * TR + cj];
      }
      if (SLS_T4_EXP != 4) {
        // The helper's copy of this pivot: the row as it was BEFORE the update, times 1/pivot (what an LU step uses; `tj`, not `row`:
        // a store of the raw row made the chain wave wait for its ds_bpermute results a whole reciprocal chain early), 1/pivot,
        // then the flag — stored by the eight owner lanes (a = p mod 8) only, see store_row8_*.
        static_assert(TR == 3 || TR == 4, "store_row8_* are written for three or four tile columns");
        if constexpr (TR == 3) store_row8_3<pa, pv * RS * 8>(pub_addr, tj[0], tj[1], tj[2], d, flag_addr, seqbase + pv + 1);
        else store_row8_4<pa, pv * RS * 8>(pub_addr, tj[0], tj[1], tj[2], tj[3], d, flag_addr, seqbase + pv + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (have_next) fetch(std::integral_constant<int, have_next ? pv + 1 : 0>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ri = 0; ri < TR; ++ri) {
#pragma unroll
        for (int cj = 0; cj < TR; ++cj)
          if (!(ri == ns || cj == ns || ri == ps)) Tt[ri * TR + cj] = __builtin_fma(-c0[ri], tfix[cj], Tt[ri * TR + cj]);
      }
      if constexpr (have_next) dnext = xr;
    }
  });
}

// The helper wave's side of the same elimination: X ← X − X[:,p]·(d·g[p,:]),  S ← S − (X[:,p]·d)·X[:,p]ᵀ for every pivot p the chain
// wave has published (same lane grid; column p of X by ds_swizzle inside the 8-lane group, the same column as a row by
// ds_bpermute from lane (b, p mod 8)).  Columns ≤ p of X are dead after pivot p and are left to whatever the update makes of them.
template <int NP, int TR, int RS>
__device__ __forceinline__ void helper_eliminate(double (&Xw)[TR * TR], double (&Sw)[TR * TR], const int lane, const int n,
                                                 const double* rowbuf, const int* flag, const int seqbase, unsigned long long& wait_cycles) {
  int ta = lane >> 3, tb = lane & 7, nn = __builtin_amdgcn_readfirstlane(n);
  asm volatile("" : "+v"(ta), "+v"(tb), "+s"(nn));
  (void)ta;
  int ready = 0;
  // Column q of X in both index forms, fetched one pivot ahead (as the chain wave does with its pivot row and column): only the
  // entries of the NEXT pivot's column slot have to be up to date before its cross-lane reads issue, the other FMAs of the
  // step run in their shadow.  Two register sets, alternating by pivot parity (no copies).
  double xc[2][TR], xrw[2][TR];
  auto fetchX = [&](auto q_c) {
    constexpr int q = decltype(q_c)::value;
    constexpr int qa = q % 8, qs = q / 8, par = q & 1;
#pragma unroll
    for (int ri = 0; ri < TR; ++ri) {
      xc[par][ri] = bcast8<qa>(Xw[ri * TR + qs]);
    }
    const int src = (tb * 8 + qa) << 2;
#pragma unroll
    for (int cj = 0; cj < TR; ++cj) {
      const double v = Xw[cj * TR + qs];
      xrw[par][cj] = (SLS_T4_EXP == 2 || SLS_T4_EXP == 3) ? v : __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(v)),
                                                                                    __builtin_amdgcn_ds_bpermute(src, __double2loint(v)));
    }
  };
  fetchX(std::integral_constant<int, 0>{});
  // The published row of the NEXT pivot is read speculatively one step ahead, flag first (DS operations execute in order: a
  // flag value that covers the pivot makes the data read after it valid); only when the chain wave had not got there yet
  // does the step poll and read again.  Takes two LDS round trips per pivot off the helper's chain.
  double dn[2] = {0.0, 0.0}, gn[2][TR];
  int fnext = 0;
  constexpr int SB = (TR == 3) ? 4 : 6;
  static_assert(RS == 8 * SB, "row stride = eight records");
  auto read_row = [&](int q, int par) {                   // the record of this lane's grid column: two ds_read_b128 (+ one b64)
    const double* rec = rowbuf + q * RS + SB * tb;
    const d2_t q0 = *reinterpret_cast<const d2_t*>(rec), q1 = *reinterpret_cast<const d2_t*>(rec + 2);
    gn[par][0] = q0[0]; gn[par][1] = q0[1]; gn[par][2] = q1[0];
    if constexpr (TR == 3) dn[par] = q1[1];
    else { gn[par][TR - 1] = q1[1]; dn[par] = rec[4]; }
  };
  auto prefetch_row = [&](auto q_c) {
    constexpr int q = decltype(q_c)::value, par = q & 1;
    fnext = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    read_row(q, par);
  };
  prefetch_row(std::integral_constant<int, 0>{});
  static_for<NP>([&](auto pv_c) {
    constexpr int pv = decltype(pv_c)::value;
    if (pv < nn) {
      constexpr bool have_next = pv + 1 < NP;
      constexpr int ps = pv / 8, ns = have_next ? (pv + 1) / 8 : pv / 8, par = pv & 1;
      const int target = seqbase + pv + 1;
      ready = max(ready, __builtin_amdgcn_readfirstlane(fnext));
      if (ready < target) {
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
        while (ready < target) {
          ready = flag_load(flag);
          if (ready < target) __builtin_amdgcn_s_sleep(1);
        }
        wait_cycles += __builtin_amdgcn_s_memtime() - w0;
        asm volatile("" ::: "memory");
        read_row(pv, par);
      }
      if constexpr (have_next) prefetch_row(std::integral_constant<int, have_next ? pv + 1 : 0>{});
      const double d = dn[par];
      double m[TR];
      const double (&grow)[TR] = gn[par];
      __builtin_amdgcn_sched_barrier(0);
      // phase A: the next pivot's column slot of X
#pragma unroll
      for (int ri = 0; ri < TR; ++ri) Xw[ri * TR + ns] = __builtin_fma(-xc[par][ri], grow[ns], Xw[ri * TR + ns]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (have_next) fetchX(std::integral_constant<int, have_next ? pv + 1 : 0>{});
      __builtin_amdgcn_sched_barrier(0);
      // phase B: the live rest of X (column slots below the pivot's are dead), the stored half of S (tiles ri ≤ cj)
#pragma unroll
      for (int ri = 0; ri < TR; ++ri) m[ri] = xc[par][ri] * d;
#pragma unroll
      for (int ri = 0; ri < TR; ++ri) {
#pragma unroll
        for (int cj = 0; cj < TR; ++cj) {
          if (cj != ns && cj >= ps) Xw[ri * TR + cj] = __builtin_fma(-xc[par][ri], grow[cj], Xw[ri * TR + cj]);      // grow = g[p,:]·d as published
          if (ri <= cj) Sw[ri * TR + cj] = __builtin_fma(-m[ri], xrw[par][cj], Sw[ri * TR + cj]);
        }
      }
    }
  });
}

template <int NPL, int RPL>
__device__ __forceinline__ void twisted4_solve_column(const KernelParams& p, const SubDesc& sd, double* __restrict__ fac,
                                                      unsigned char* lds_raw) {
  static_assert(NPL == 32, "written for the NPL = 32 classes (8×8 lane grid)");
  constexpr int HS = 64 / NPL, NP = HS * RPL;
  constexpr int TR = (NP + 7) / 8, NR = 8 * TR, TT = TR * TR;
  constexpr int LDT = 40;                                 // leading dimension of the tile images (8·a + b: no bank conflict)
  constexpr int RS = (TR == 3) ? 32 : 48;                 // doubles per published pivot: eight packed records (store_row8_*)
  constexpr int PRIVC = NR * LDT + 2 * NPL;               // chain wave: mat, tmp, tmp2
  constexpr int PRIVH = 0;                                // (helper waves keep nothing private in LDS)
  constexpr int DIRSZ = NR * RS + NR * LDT + 2;           // per direction: published rows, hand-off tiles, two flags
  const int wv = threadIdx.x >> 6;                        // 0/1: chain waves (up / down), 2/3: their helpers
  const int dir = wv & 1, helper = wv >> 1;
  const int lane = threadIdx.x & 63;
  const int h = lane / NPL, j = lane % NPL;
  const int ta = lane >> 3, tb = lane & 7;
  const int n = sd.n, m = sd.m, nm = n + m, T = p.T;
  const int MC = p.w_mcap;
  const int capA = p.w_nzA, capAc = p.w_nzAc, capB = p.w_nzB, capBc = p.w_nzBc;
  const int c = twisted4_middle(T);

  // ---- LDS carve (must match twisted4_kernel_lds_bytes) ----
  double* dp = reinterpret_cast<double*>(lds_raw);
  double* privc = dp + dir * PRIVC; dp += 2 * PRIVC;
  double* mat = privc; double* tmp = mat + NR * LDT; double* tmp2 = tmp + NPL;
  dp += 2 * PRIVH;
  double* dirb = dp + dir * DIRSZ;
  double* dirb_other = dp + (1 - dir) * DIRSZ; dp += 2 * DIRSZ;
  double* rowbuf = dirb; double* hand = rowbuf + NR * RS;
  int* flagA = reinterpret_cast<int*>(hand + NR * LDT);   // chain → helper: rows published (sequence number)
  int* flagB = flagA + 2;                                  // helper → chain: tiles of step s are in `hand`
  double* hand_other = dirb_other + NR * RS;
  int* flagB_other = reinterpret_cast<int*>(hand_other + NR * LDT) + 2;
  double* Ad = dp;     dp += NR * LDT;                    // dense image of Ã (zero padded; leading dimension 40: 8·a + b conflict-free)
  double* hx = dp;     dp += NPL;
  double* gx = dp;     dp += NPL;
  double* hu = dp;     dp += 64;
  double* gu = dp;     dp += 64;
  double* red = dp;    dp += 8;                           // [0..3] per-wave maxima, [4] δ
  int32_t* sx = reinterpret_cast<int32_t*>(dp);
  int32_t* su = sx + NPL;
  int32_t* nzs = su + 64;                                 // nzA, nzAc, nzB, nzBc
  dp += (NPL + 64 + 8) / 2;
  double* lam = dp;    dp += (T + 1) * NPL;
  double* rq = dp;     dp += (T + 1) * NPL;
  double* xs = dp;     dp += (T + 1) * NPL;
  double* Bd = dp;     dp += NPL * MC;
  double* us = dp;     dp += T * MC;
  double* wxt = dp;    dp += (T + 1) * NPL;                // Wx_k[j] for every block k (row T = 0): mask ⊙ H⁻¹ expanded once
  double* wut = dp;    dp += T * MC;                       // Wu_k[q]
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
  const int32_t* dest = p.dest_pool + sd.off_dest;
  unsigned long long tc[4] = {0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
  auto lap = [&](int slot) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tc[slot] += now - tlast; tlast = now; };

  __syncthreads();
  // ---- setup, split over the four waves ----
  if (wv == 0) {
    if (lane < NPL) {
      sx[lane] = (lane < n) ? p.idx_pool[sd.off_sx + lane] : 0x7fffffff;
      hx[lane] = (lane < n) ? (sd.has_w ? p.w_pool[sd.off_w + lane] : 1.0) : 0.0;
      gx[lane] = (lane < n && sd.has_w) ? p.w_pool[sd.off_w + nm + lane] : 0.0;
    }
    su[lane] = (lane < m) ? p.idx_pool[sd.off_su + lane] : 0x7fffffff;
    hu[lane] = (lane < m) ? (sd.has_w ? p.w_pool[sd.off_w + n + lane] : 1.0) : 0.0;
    gu[lane] = (lane < m && sd.has_w) ? p.w_pool[sd.off_w + nm + n + lane] : 0.0;
    for (int i = lane; i < capA * NPL; i += 64) { arow_v[i] = 0.0; arow_c[i] = 0; }
    for (int i = lane; i < capB * NPL; i += 64) { brow_v[i] = 0.0; brow_c[i] = 0; }
  } else if (wv == 1) {
    for (int i = lane; i < T * nm; i += 64) mask[i] = p.mask_pool[sd.off_mask + i];
    for (int i = lane; i < capAc * NPL; i += 64) { acol_v[i] = 0.0; acol_c[i] = 0; }
    for (int i = lane; i < capBc * 64; i += 64) { bcol_v[i] = 0.0; bcol_c[i] = 0; }
  } else if (wv == 2) {
    for (int i = lane; i < NR * LDT; i += 64) Ad[i] = 0.0;
    for (int i = lane; i < NPL * MC; i += 64) Bd[i] = 0.0;
  } else {
    for (int i = lane; i < (T + 1) * NPL; i += 64) { lam[i] = 0.0; rq[i] = 0.0; xs[i] = 0.0; }
  }
  if (helper == 0) {
    if (lane < NPL) { tmp[lane] = 0.0; tmp2[lane] = 0.0; }
  } else if (lane < 4) {
    flagA[lane] = 0;                                      // flagA[0..1], flagB[0..1] of this direction
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
    if (lane == 0) { nzs[0] = a0; nzs[2] = a2; red[4] = dl_; }
  } else if (wv == 1) {
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
  } else if (wv == 2) {
    // dense image of Ã for the helpers (row i, column loc): the second factor of Ã W Ãᵀ and the border blocks X
    if (lane < n) {
      const int g = sx[lane];
      for (int e = p.A_rowptr[g]; e < p.A_rowptr[g + 1]; ++e) {
        const double v = p.A_val[e];
        const int loc = (v != 0.0) ? wbsearch(sx, n, p.A_colidx[e]) : -1;
        if (loc >= 0) Ad[lane * LDT + loc] += v;
      }
    }
  } else {
    // weight tables for the helpers: one LDS read per weight later instead of mask → weight → select chains per block
    for (int i = lane; i < (T + 1) * NPL; i += 64) {
      const int k = i / NPL, jj = i % NPL;
      wxt[i] = (k < T && jj < n && mask[k * nm + jj]) ? hx[jj] : 0.0;
    }
    for (int i = lane; i < T * MC; i += 64) {
      const int k = i / MC, q = i % MC;
      wut[i] = (q < m && mask[k * nm + n + q]) ? hu[q] : 0.0;
    }
  }
  __syncthreads();
  const int nzA = nzs[0], nzAc = nzs[1], nzB = nzs[2], nzBc = nzs[3];
  const double delta = red[4];
  lap(0);

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
    return part;
  };
  auto load_P = [&](int k, double (&Pk)[RPL]) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) Pk[r] = fac[((int64_t)k * RPL + r) * 64 + lane];
  };
  auto store_P = [&](int k, const double (&Pk)[RPL]) {
#pragma unroll
    for (int r = 0; r < RPL; ++r) fac[((int64_t)k * RPL + r) * 64 + lane] = Pk[r];
  };

  // r = f − E z(λ): x_t, u_t first (any order), then r_t = f_t − x_t + Ãx_{t−1} + B̃u_{t−1}; 4·HS time steps per round
  auto x_of = [&](int t) -> double {
    const double l0 = lam[t * NPL + j];
    const uint8_t mk = mask[t * nm + j];
    const double a = dotA_col(lam + (t + 1) * NPL);
    const double v = hx[j] * (l0 - a - gx[j]);
    return mk ? v : 0.0;
  };
  int lgMP = 0;
  while ((1 << lgMP) < m) ++lgMP;
  const int uq = lane & ((1 << lgMP) - 1), uts = lane >> lgMP, NTS = 64 >> lgMP;
  auto u_of = [&](int t) -> double {
    const double* l1 = lam + (t + 1) * NPL;
    double a = 0.0;
    for (int e = 0; e < nzBc; ++e) a = __builtin_fma(bcol_v[e * 64 + uq], l1[bcol_c[e * 64 + uq]], a);
    return mask[t * nm + n + uq] ? hu[uq] * (-a - gu[uq]) : 0.0;
  };
  constexpr int G = 4 * HS;
  const int gid = wv * HS + h;
  bool xu_valid = false;
  auto residual_pass = [&]() -> double {       // all 256 threads; contains workgroup barriers
    const bool live = j < n;
    xu_valid = true;
    const int nit = (T + G) / G;
#pragma unroll 2
    for (int it = 0; it < nit; ++it) {
      const int t = gid + it * G;
      const double v = (t < T && live) ? x_of(min(t, T - 1)) : 0.0;
      if (t <= T) xs[t * NPL + j] = v;
    }
    if (uq < m) {
#pragma unroll 2
      for (int t = wv * NTS + uts; t < T; t += 4 * NTS) us[t * MC + uq] = u_of(t);
    }
    __syncthreads();
    double rmax = 0.0;
#pragma unroll 2
    for (int it = 0; it < nit; ++it) {
      const int t = gid + it * G;
      const int tc_ = min(t, T), tp = max(tc_ - 1, 0);
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
    rmax = wave_max_f64(rmax);
    if (lane == 0) red[wv] = rmax;
    __syncthreads();
    const double r2 = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    return r2;
  };
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
      for (int t0 = wv * NTS + uts; t0 < T; t0 += 4 * NTS * CH) {
        int dd[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int t = t0 + q * 4 * NTS; dd[q] = (t < T && mask[t * nm + n + uq]) ? dest[t * nm + n + uq] : -1; }
#pragma unroll
        for (int q = 0; q < CH; ++q) { const int t = t0 + q * 4 * NTS; if (dd[q] >= 0) p.out[dd[q]] = xu_valid ? us[t * MC + uq] : 0.0; }
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

  // ---- chain-wave sweep steps (the two-wave kernel's, unchanged) ----
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
    } else if (wv == 1) {
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

  unsigned long long ph[4] = {0, 0, 0, 0};      // diagnostics (SLS_PHASE_TIMERS ≥ 2): chain wave: hand-off waits / Gauss–Jordan / store + sweep; helper: static + border / elimination / of which waiting
  if (resid > p.tol) {
    double M[RPL];
    if (helper == 0) {
      // =================== chain wave: Gauss–Jordan + store + fused sweep per block; blocks arrive from the helper ===================
      if (dir == 0) {
        // block 0 is diagonal (δ + Wx_0): its inverse needs no elimination, and the helper folds it into block 1's weights
        const double w0 = wx_of(0);
#pragma unroll
        for (int r = 0; r < RPL; ++r) M[r] = (HS * r + h == j && j < n) ? 1.0 / (delta + w0) : 0.0;
        store_P(0, M);
        elim_up(0, M, w0);
      }
      const int s_end = (dir == 0) ? c : T - c;
      for (int s = 1; s <= s_end; ++s) {
        const int k = (dir == 0) ? s : T - s + 1;
        const bool mid = (dir == 0) && (s == c);
        unsigned long long q0 = __builtin_amdgcn_s_memtime();
        wait_flag(flagB, s);
        // the helper hands over the stored half (tiles ri ≤ cj of every lane); the other half is read at the mirror position
        double Tt[TT];
#pragma unroll
        for (int ri = 0; ri < TR; ++ri) {
#pragma unroll
          for (int cj = 0; cj < TR; ++cj)
            Tt[ri * TR + cj] = (ri <= cj) ? hand[(ta + 8 * ri) * LDT + tb + 8 * cj] : hand[(tb + 8 * cj) * LDT + ta + 8 * ri];
        }
        if (mid) {
          wait_flag(flagB_other, T - c + 1);
#pragma unroll
          for (int ri = 0; ri < TR; ++ri) {
#pragma unroll
            for (int cj = 0; cj < TR; ++cj)
              Tt[ri * TR + cj] += (ri <= cj) ? hand_other[(ta + 8 * ri) * LDT + tb + 8 * cj] : hand_other[(tb + 8 * cj) * LDT + ta + 8 * ri];
          }
        }
        if (p.dbg_level >= 2) { const unsigned long long q1 = __builtin_amdgcn_s_memtime(); ph[0] += q1 - q0; q0 = q1; }
        gj_tiles_publish<NP, TR, RS>(Tt, lane, n, rowbuf, flagA, s * 64, mat);
        if (p.dbg_level >= 2) { __builtin_amdgcn_sched_barrier(0); asm volatile("" :: "v"(Tt[0])); const unsigned long long q1 = __builtin_amdgcn_s_memtime(); ph[1] += q1 - q0; q0 = q1; }
        // tiles → column layout (lane (h, j): rows HS·r + h of column j) through the private image
#pragma unroll
        for (int ri = 0; ri < TR; ++ri) {
#pragma unroll
          for (int cj = 0; cj < TR; ++cj) mat[(ta + 8 * ri) * LDT + tb + 8 * cj] = Tt[ri * TR + cj];
        }
        WSYNC();
#pragma unroll
        for (int r = 0; r < RPL; ++r) M[r] = mat[(HS * r + h) * LDT + j];
        WSYNC();
        store_P(k, M);
        if (!mid) { if (dir == 0) elim_up(k, M, wx_of(k)); else elim_down(k, M, wx_of(k)); }
        if (p.dbg_level >= 2) { const unsigned long long q1 = __builtin_amdgcn_s_memtime(); ph[2] += q1 - q0; q0 = q1; }
      }
    } else {
      // =================== helper wave: static part S_k, border X, elimination behind the chain wave's pivots ===================
      // row lists of Ã and B̃ for this lane's tile rows a + 8·ri: first KH / KB entries in registers for the whole column
      constexpr int KH = 4, KB = 2;
      int hac[TR][KH], hbc[TR][KB];
      double hav[TR][KH], hbv[TR][KB];
#pragma unroll
      for (int ri = 0; ri < TR; ++ri) {
        const int i = ta + 8 * ri;
#pragma unroll
        for (int e = 0; e < KH; ++e) { const bool ok = e < capA; hac[ri][e] = ok ? arow_c[e * NPL + i] : 0; hav[ri][e] = ok ? arow_v[e * NPL + i] : 0.0; }
#pragma unroll
        for (int e = 0; e < KB; ++e) { const bool ok = e < capB; hbc[ri][e] = ok ? brow_c[e * NPL + i] : 0; hbv[ri][e] = ok ? brow_v[e * NPL + i] : 0.0; }
      }
      // which blocks can reuse the previous step's static part: masks of (k, k−1) equal those of the block produced before
      // (upward: k−1; downward: k+1 — never block T, whose Wx is 0).  One ballot per block, all loads independent.
      unsigned long long samebits = 0;
      for (int k = 2; k <= T - 1; ++k) {
        const int ko = (dir == 0) ? k - 1 : k + 1;
        bool eq = true;
        if (ko <= T - 1 && lane < nm) eq = (mask[k * nm + lane] == mask[ko * nm + lane]) && (mask[(k - 1) * nm + lane] == mask[(ko - 1) * nm + lane]);
        if (SLS_T4_SAME != 0 && ko <= T - 1 && k < 64 && __all(eq)) samebits |= 1ull << k;
      }
      double Sc[TT];                         // cached static part (stored half) while the masks repeat
      bool have_S = false;
      const int s_end = (dir == 0) ? c : T - c + 1;
      auto load_border = [&](int k, double (&Xn)[TT]) {     // X of block k: upward Ã·Wx_{k−1}, downward Wx_k·Ãᵀ
        const double* wrow = wxt + ((dir == 0) ? k - 1 : k) * NPL;
#pragma unroll
        for (int ri = 0; ri < TR; ++ri) {
          const int i = ta + 8 * ri;
#pragma unroll
          for (int cj = 0; cj < TR; ++cj) {
            const int l = tb + 8 * cj;
            Xn[ri * TR + cj] = (dir == 0) ? Ad[i * LDT + l] * wrow[l] : wrow[i] * Ad[l * LDT + i];
          }
        }
      };
      double Xn[TT];
      if (s_end >= 2) load_border((dir == 0) ? 2 : T - 1, Xn);
      for (int s = 1; s <= s_end; ++s) {
        unsigned long long q0 = __builtin_amdgcn_s_memtime();
        const int k = (dir == 0) ? s : T - s + 1;           // block being produced (downward helper's last step: k = c, the middle)
        const bool mid_down = (dir == 1) && (k == c);
        const bool same = have_S && !mid_down && k < 64 && ((samebits >> k) & 1ull) && !(dir == 0 && s == 2);
        double Sw[TT];
        if (mid_down) {
#pragma unroll
          for (int q = 0; q < TT; ++q) Sw[q] = 0.0;
        } else if (same) {
#pragma unroll
          for (int q = 0; q < TT; ++q) Sw[q] = Sc[q];
        } else {
          const double* wc_ = wxt + k * NPL;
          const double* wp_ = wxt + (k - 1) * NPL;
          const double* wu_ = wut + (k - 1) * MC;
          const bool fold0 = (dir == 0) && (s == 1);         // upward, block 1: P_0 = (δ + Wx_0)⁻¹ is diagonal, W − W P_0 W = δ·w/(δ + w) takes the place of Wx_0
#pragma unroll
          for (int ri = 0; ri < TR; ++ri) {
            const int i = ta + 8 * ri;
            const double di = delta + wc_[i];
#pragma unroll
            for (int cj = 0; cj < TR; ++cj) Sw[ri * TR + cj] = (i == tb + 8 * cj) ? di : 0.0;
          }
#pragma unroll
          for (int e = 0; e < KH; ++e) {
#pragma unroll
            for (int ri = 0; ri < TR; ++ri) {
              const int cc = hac[ri][e];
              const double w0 = wp_[cc];
              const double v = hav[ri][e] * (fold0 ? delta * w0 / (delta + w0) : w0);
#pragma unroll
              for (int cj = ri; cj < TR; ++cj) Sw[ri * TR + cj] = __builtin_fma(v, Ad[(tb + 8 * cj) * LDT + cc], Sw[ri * TR + cj]);
            }
          }
          for (int e = KH; e < nzA; ++e) {
#pragma unroll
            for (int ri = 0; ri < TR; ++ri) {
              const int i = ta + 8 * ri;
              const int cc = arow_c[e * NPL + i];
              const double w0 = wp_[cc];
              const double v = arow_v[e * NPL + i] * (fold0 ? delta * w0 / (delta + w0) : w0);
#pragma unroll
              for (int cj = ri; cj < TR; ++cj) Sw[ri * TR + cj] = __builtin_fma(v, Ad[(tb + 8 * cj) * LDT + cc], Sw[ri * TR + cj]);
            }
          }
#pragma unroll
          for (int e = 0; e < KB; ++e) {
#pragma unroll
            for (int ri = 0; ri < TR; ++ri) {
              const int cc = hbc[ri][e];
              const double v = hbv[ri][e] * wu_[cc];
#pragma unroll
              for (int cj = ri; cj < TR; ++cj) Sw[ri * TR + cj] = __builtin_fma(v, Bd[(tb + 8 * cj) * MC + cc], Sw[ri * TR + cj]);
            }
          }
          for (int e = KB; e < nzB; ++e) {
#pragma unroll
            for (int ri = 0; ri < TR; ++ri) {
              const int i = ta + 8 * ri;
              const int cc = brow_c[e * NPL + i];
              const double v = brow_v[e * NPL + i] * wu_[cc];
#pragma unroll
              for (int cj = ri; cj < TR; ++cj) Sw[ri * TR + cj] = __builtin_fma(v, Bd[(tb + 8 * cj) * MC + cc], Sw[ri * TR + cj]);
            }
          }
          if (!fold0) {
#pragma unroll
            for (int q = 0; q < TT; ++q) Sc[q] = Sw[q];
            have_S = true;
          }
        }
        if (s >= 2) {
          double Xw[TT];
#pragma unroll
          for (int q = 0; q < TT; ++q) Xw[q] = Xn[q];
          if (s + 1 <= s_end) load_border((dir == 0) ? s + 1 : T - s, Xn);      // next step's border: its loads fly during this elimination
          if (p.dbg_level >= 2) { const unsigned long long q1 = __builtin_amdgcn_s_memtime(); ph[0] += q1 - q0; q0 = q1; }
          if (SLS_T4_EXP != 1 && SLS_T4_EXP != 4) helper_eliminate<NP, TR, RS>(Xw, Sw, lane, n, rowbuf, flagA, (s - 1) * 64, ph[2]);
          if (p.dbg_level >= 2) { __builtin_amdgcn_sched_barrier(0); asm volatile("" :: "v"(Sw[0])); const unsigned long long q1 = __builtin_amdgcn_s_memtime(); ph[1] += q1 - q0; q0 = q1; }
        }
#pragma unroll
        for (int ri = 0; ri < TR; ++ri) {
#pragma unroll
          for (int cj = ri; cj < TR; ++cj) hand[(ta + 8 * ri) * LDT + tb + 8 * cj] = Sw[ri * TR + cj];
        }
        if (lane == 0) flag_store(flagB, s);
        WSYNC();
      }
    }
    lap(1);                        // own half of the factorisation
    __syncthreads();
    lap(2);                        // waiting for the other waves
    if (wv == 0) middle(M);        // M = P_c
    __syncthreads();
    outward();
    __syncthreads();
    lap(3);

    // ---------------- multiplier iteration ----------------
    double prev = resid, prev2 = resid;
    int itmax = p.max_iters;
    for (int it = 1; it <= itmax; ++it) {
      iters = it;
      if (it > 1) {
        double Pk[RPL];
        if (wv == 0) {
          for (int k = 0; k < c; ++k) { load_P(k, Pk); elim_up(k, Pk, wx_of(k)); }
        } else if (wv == 1) {
          for (int k = T; k > c; --k) { load_P(k, Pk); elim_down(k, Pk, wx_of(k)); }
        }
        __syncthreads();
        if (wv == 0) { load_P(c, Pk); middle(Pk); }
        __syncthreads();
        outward();
        __syncthreads();
      }
      resid = residual_pass();
      if (resid <= p.tol) break;
      if (it >= 2 && resid > p.stag * prev) {          // (still_contracting: sls_device.h)
        if (!(p.max_iters_slow > 0 && (resid > p.tol_ok || itmax > p.max_iters) && still_contracting(it >= 3 ? prev2 : prev, prev, resid))) { status = 1; break; }
        itmax = max(itmax, p.max_iters_slow);
      }
      prev2 = prev; prev = resid;
    }
    if (resid <= p.tol_ok) status = 0;
    else if (status == 0) status = 2;
  }
  output_pass();
  if (p.dbg && lane == 0) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    if (p.dbg_level <= 1 && wv < 2) {
      for (int q = 0; q < 3; ++q) p.dbg[sd.out_index * 8 + wv * 4 + q] = tc[q];
      p.dbg[sd.out_index * 8 + wv * 4 + 3] = (tc[3] << 32) | ((now - tlast) & 0xffffffffull);
    } else if ((p.dbg_level == 2 && wv < 2) || (p.dbg_level == 3 && wv >= 2)) {       // 2: chain waves' shares, 3: helper waves'
      for (int q = 0; q < 3; ++q) p.dbg[sd.out_index * 8 + dir * 4 + q] = ph[q];
      p.dbg[sd.out_index * 8 + dir * 4 + 3] = tc[1];
    }
  }
  if (sd.pos < 0 && status == 0) status = 3;
  if (threadIdx.x == 0) {
    p.status[sd.out_index] = status;
    p.resid[sd.out_index] = resid;
    p.iters[sd.out_index] = iters;
  }
}

template <int NPL, int RPL>
__global__ __launch_bounds__(256, 1) void h2_column_twisted4_kernel(const KernelParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  double* fac = p.fac_ws + (int64_t)blockIdx.x * p.fac_stride;
  for (int s = blockIdx.x; s < p.nsub; s += gridDim.x) {
    const SubDesc sd = p.subs[p.order[p.order_off + s]];
    twisted4_solve_column<NPL, RPL>(p, sd, fac, lds_raw);
  }
}

template <int NPL, int RPL>
static hipError_t launch_one_twisted4(const KernelParams& p, int grid, size_t lds, hipStream_t st) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&h2_column_twisted4_kernel<NPL, RPL>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((h2_column_twisted4_kernel<NPL, RPL>), dim3(grid), dim3(256), lds, st, p);
  return hipGetLastError();
}

hipError_t launch_twisted4(int cls, const KernelParams& p, int grid, size_t lds, hipStream_t st) {
  switch (cls) {
    case 2: return launch_one_twisted4<32, 10>(p, grid, lds, st);
    case 3: return launch_one_twisted4<32, 12>(p, grid, lds, st);
    case 4: return launch_one_twisted4<32, 14>(p, grid, lds, st);
    case 5: return launch_one_twisted4<32, 16>(p, grid, lds, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sls
