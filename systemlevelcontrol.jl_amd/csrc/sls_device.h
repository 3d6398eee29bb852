// sls_device.h — host/device shared structures of the batched H2 column solver.
// Plain structs only; included by both the C-ABI host code (sls_api.cpp) and the
// HIP kernels (sls_kernels.hip).
#pragma once
#include <stdint.h>

namespace sls {

// One localized subproblem = one disturbance column c of one group
// (reference src/synthesis.jl:37-68, one iteration of the `for cⱼ in Cⱼ` loop body,
// specialised to one column of the group).
struct SubDesc {
  int32_t n;          // ñx = |s_x|            (src/reduction.jl:14)
  int32_t m;          // ñu = |s_u|
  int32_t pos;        // local index of column c inside s_x, or -1 (Ĩ column is zero: reduction.jl:22-23)
  int32_t nnzA;       // nnz(Ã),  Ã = A[s_x,s_x]   (GeneralizedPlant.jl:266)
  int32_t nnzB;       // nnz(B̃2), B̃2 = B2[s_x,s_u] (GeneralizedPlant.jl:268)
  int32_t has_w;      // 1 if a weight record (hinv_x,hinv_u,g_x,g_u) exists, 0 = identity cost
  int32_t cls;        // wave-kernel size class (wave_class_of), -1 = general kernel
  int32_t pad_;
  int64_t off_sx;     // into idx_pool (int32 global state indices, ascending)
  int64_t off_su;     // into idx_pool (int32 global input indices, ascending)
  int64_t off_mask;   // into mask_pool: uint8 [T][n+m], 1 = free variable (synthesis.jl:57-60)
  int64_t off_dest;   // into dest_pool: int32 [T][n+m], destination in the value array, -1 = none
  int64_t off_w;      // into w_pool (doubles): hinv_x[n], hinv_u[m], g_x[n], g_u[m]
  int64_t out_index;  // index of this subproblem in status/resid/iters arrays
};

struct KernelParams {
  // shared operator, resident once per device  (P.A, P.B₂ of the reference plant)
  const int32_t* A_rowptr;  const int32_t* A_colidx;  const double* A_val;    // CSR of A
  const int32_t* At_rowptr; const int32_t* At_colidx; const double* At_val;   // CSR of Aᵀ (= Julia's CSC of A)
  const int32_t* B_rowptr;  const int32_t* B_colidx;  const double* B_val;    // CSR of B2
  const int32_t* Bt_rowptr; const int32_t* Bt_colidx; const double* Bt_val;   // CSR of B2ᵀ (= Julia's CSC of B2)
  // per-subproblem tables
  const SubDesc* subs;
  const int32_t* order;      // processing order (descending predicted cost)
  const int32_t* idx_pool;
  const uint8_t* mask_pool;
  const int32_t* dest_pool;
  const double*  w_pool;
  int32_t nsub;       // subproblems of THIS launch: order[order_off .. order_off+nsub)
  int32_t order_off;
  int32_t T;
  // wave-kernel LDS capacities (per launch)
  int32_t w_mcap, w_nzA, w_nzAc, w_nzB, w_nzBc, w_nm_max;
  int32_t w_pl_off;   // twisted kernel: byte offset of the LDS-resident P_k blocks, 0 = P_k in the global workspace
  // workspaces (per resident workgroup)
  double* fac_ws;  int64_t fac_stride;   // (T+1)·nmax² doubles: the inverse Schur blocks P_k
  double* vec_ws;  int64_t vec_stride;   // 3·(T+1)·nmax doubles when the vectors do not fit in LDS
  int32_t vec_in_lds;
  // LDS carve sizes
  int32_t nmax, mmax, nnzA_cap, nnzB_cap;
  int32_t tile_oth_rows;  // tile kernel: rows of the Ã·Q image of the block build held in LDS at a time (multiple of 16)
  int32_t* work_counter;  // tile kernel: work queue of the launch (next subproblem to hand out), cleared by the host
  unsigned char* big_ws;  // tile kernel, big variant: per-workgroup carve buffer in global memory (what LDS holds otherwise)
  int64_t big_stride;     // … bytes per workgroup
  // outputs
  double*  out;
  int32_t* status;
  double*  resid;
  int32_t* iters;
  // numerics
  double delta_rel;   // Tikhonov shift relative to the largest Schur diagonal
  double delta_first; // one-wave kernel: shift of the first attempt (0 = none): one pass when it survives round-off, else redo
  double tol;         // stop when ‖f − E z‖∞ ≤ tol
  double tol_ok;      // status OK when the final residual ≤ tol_ok
  int32_t max_iters;
  int32_t objective;  // 0 = 𝓗₂ (sum of squares), 1 = sum of norms (tile kernel, CG/ADMM build)
  int32_t son_maxit;  // sum of norms: ADMM step cap
  double son_tol;     // sum of norms: stop when primal and dual residual ≤ son_tol·max(1, ‖W z‖)
  int32_t son_anderson;  // sum of norms, one-wave kernel: Anderson acceleration of the ADMM fixed-point map (1 = on)
  int32_t son_aa_start;  // … first ADMM step whose iterate enters the Anderson history (plain steps before it)
  double stag;        // a pass that leaves more than stag × the previous residual counts as stagnation (inconsistent system)
  // optional phase-cycle counters (diagnostics): 8 uint64 per subproblem, NULL = off
  unsigned long long* dbg;
  int32_t dbg_level;     // 1 = phase laps (cheap), 2 = + stamps inside every pivot (intrusive)
  int32_t max_iters_slow; // pass cap of a column that contracts geometrically but slowly (still_contracting); 0: the plain stagnation rule
  int32_t knock_out;     // timing experiments only (SLS_KNOCK_OUT): a phase of the one-wave kernel is skipped, results are meaningless
};

// LDS bytes the general kernel needs for given caps (must match the carve in the kernel).
// one of the two ñx×ñx block images; outside the factorisation the pair doubles as the staging area of the residual pass
// (3 λ slices + 2 x/u pairs + 2 carried rows), so it is never smaller than that
__host__ __device__ static inline int64_t general_kernel_block_doubles(int nmax, int mmax) {
  const int64_t a = 1LL * nmax * nmax, b = (7LL * nmax + 2LL * mmax + 1) / 2;
  const int64_t c = a > b ? a : b;
  return c > 384 ? c : 384;          // ≥ 4·96: the second image also holds the padded pivot row/column exchange buffers
}
static inline int64_t general_kernel_lds_bytes(int nmax, int mmax, int nnzA, int nnzB, int T, bool vec_in_lds, bool oth_global = false) {
  int64_t d = 0;                 // doubles
  d += (oth_global ? 1LL : 2LL) * general_kernel_block_doubles(nmax, mmax);   // P_k / Ã·Q images (the first also stages the residual pass)
  if (!oth_global) d += 1LL * nmax * mmax;   // dense B̃2 (wide variant: in the global workspace)
  d += 2LL * nnzA + nnzB;        // csr/csc values of Ã, csr values of B̃2
  d += 2LL * nmax + 2LL * mmax;  // hinv_x, g_x, hinv_u, g_u
  d += 2LL * nmax + mmax;        // w_prev, w_cur, wu_prev
  d += 4LL * nmax + mmax;        // xt, base, tmp, tmp2, ut
  if (oth_global) d += 4LL * 144;   // wide variant: padded pivot row/column exchange buffers of their own
  d += 256;                      // block reduction + matvec partials base
  d += 1LL * 256;                // more partials (2·256 total)
  if (vec_in_lds) d += 3LL * (T + 1) * nmax;
  int64_t i = 0;                 // int32
  i += nmax + mmax;              // s_x, s_u
  i += 2LL * (nmax + 1) + 2LL * nnzA;   // csr/csc ptr + idx of Ã
  i += (nmax + 1) + nnzB;        // csr of B̃2
  i += 8;
  return d * 8 + ((i * 4 + 15) / 16) * 16 + 64;
}

// ---- tile kernel (sls_tile_kernel.hip): ñx > 64, blocks held as 16×16 FP64 MFMA tiles, upper triangle only ----
// NT = ⌈ñx/16⌉ tile rows, HT = NT(NT+1)/2 stored tiles.  mlds: the block being inverted lives in LDS (padded tiles of
// 16×17 doubles), otherwise in the workgroup's global workspace (256-double tiles, in place in its P_k slot).
#ifndef SLS_TILE_THREADS
#define SLS_TILE_THREADS 512
#endif
constexpr int kTileThreads = SLS_TILE_THREADS;      // A/B builds: 256 (four waves per column, up to four columns per CU)
constexpr int kTileWaves = kTileThreads / 64;
constexpr int kTileLdsTile = 16 * 17;          // doubles per LDS-resident tile (row stride 17: transposed reads conflict-free)
__host__ __device__ static inline int tile_nt(int n) { return (n + 15) >> 4; }
__host__ __device__ static inline int tile_ht(int nt) { return nt * (nt + 1) / 2; }
// doubles of the phase-shared LDS region: max over {pivot panel Yᵀ + L⁻¹; Ã·Q image; residual staging; sweep vectors}
__host__ __device__ static inline int64_t tile_kernel_r0_doubles(int nmax, int mmax, int oth_rows, bool mlds) {
  const int64_t nt = tile_nt(nmax), npad = 16 * nt, mpad = (mmax + 7) / 8 * 8 + 8;
  int64_t a = (mlds ? 1 : 2) * nt * 256 + 512;         // pivot panel(s) Yᵀ (two for a block in the workspace) + L⁻¹ and its transpose
  const int64_t b = (int64_t)(oth_rows < npad ? oth_rows : npad) * (npad + 1);   // Ã·Q image, oth_rows rows at a time
  const int64_t c = 7 * npad + 2 * mpad;               // residual pass: 3 λ slices, 2 carried rows, 2 (x,u) pairs
  const int64_t d = (kTileWaves + 2) * npad;           // sweeps: per-wave partial vectors + y + out
  a = a > b ? a : b; a = a > c ? a : c; a = a > d ? a : d;
  return a;
}
// fac workspace doubles per workgroup: (T+1) half-tile slots + the full row-major copy of the latest −P_k
static inline int64_t tile_kernel_fac_doubles(int nmax, int T) {
  const int64_t nt = tile_nt(nmax), npad = 16 * nt;
  return (int64_t)(T + 1) * tile_ht((int)nt) * 256 + npad * npad;
}
static inline int64_t tile_kernel_lds_bytes(int nmax, int mmax, int nnzA, int nnzB, bool mlds, int oth_rows = 16) {
  const int64_t nt = tile_nt(nmax), npad = 16 * nt, mpad = (mmax + 7) / 8 * 8 + 8;
  int64_t d = tile_kernel_r0_doubles(nmax, mmax, oth_rows, mlds);
  if (mlds) d += (int64_t)tile_ht((int)nt) * kTileLdsTile;
  d += 2LL * nnzA + 2LL * nnzB;          // csr/csc values of Ã and B̃2
  d += 2 * npad;                         // w_prev, w_cur
  d += 16;                               // block reduction
  int64_t i = 0;
  i += npad;                             // s_x
  i += 2 * (npad + 1) + 2LL * nnzA;      // csr/csc of Ã
  i += (npad + 1) + (mpad + 1) + 2LL * nnzB;   // csr/csc of B̃2
  i += tile_ht((int)nt);                 // tile list
  i += 8;
  return d * 8 + ((i * 4 + 15) / 16) * 16 + 64;
}

// ---- wave kernel size classes: (NPL lanes per row group, RPL rows per lane), capacity n ≤ (64/NPL)·RPL ----
constexpr int kNumWaveClasses = 9;
constexpr int kNumSmallWaveClasses = 6;   // classes 0..5 have NPL ≤ 32 (light on registers: higher occupancy cap)
struct WaveClass { int npl, rpl; };
static inline WaveClass wave_class(int cls) {
  constexpr WaveClass tab[kNumWaveClasses] = {{16, 3}, {16, 4}, {32, 10}, {32, 12}, {32, 14}, {32, 16}, {64, 40}, {64, 48}, {64, 64}};
  return tab[cls];
}
static inline int wave_class_of(int n, int m) {
  if (m > 64 || n < 1) return -1;
  for (int c = 0; c < kNumWaveClasses; ++c) {
    const WaveClass w = wave_class(c);
    if (n <= (64 / w.npl) * w.rpl) return c;
  }
  return -1;
}
// LDS bytes of the wave kernel (must match the carve in wave_solve_column)
static inline int64_t wave_kernel_lds_bytes(int cls, int T, int mcap, int nzA, int nzAc, int nzB, int nzBc, int nm_max,
                                            bool vec_global = false) {
  const WaveClass w = wave_class(cls);
  const int64_t NPL = w.npl, NP = (64 / w.npl) * w.rpl, LDM = NPL + 1;
  const int64_t us_extra = ((int64_t)T * mcap > NP * LDM) ? (int64_t)T * mcap : 0;      // `us` aliases the matrix image when it fits
  int64_t d = NP * LDM + (vec_global ? 0 : 2LL * (T + 1) * NPL) + 5 * NPL + 3 * 64 + NPL * mcap + us_extra + (int64_t)(nzA + nzAc + nzB) * NPL + (int64_t)nzBc * 64;
  int64_t i = (int64_t)(nzA + nzAc + nzB) * NPL + (int64_t)nzBc * 64 + NPL + 64;
  return d * 8 + i * 4 + ((int64_t)T * nm_max + 15) / 16 * 16 + 16;
}

// LDS bytes of the twisted two-wave kernel (must match the carve in twisted_solve_column)
static inline int64_t twisted_kernel_lds_bytes(int cls, int T, int mcap, int nzA, int nzAc, int nzB, int nzBc, int nm_max) {
  const WaveClass w = wave_class(cls);
  const int64_t NPL = w.npl, NP = (64 / w.npl) * w.rpl, LDM = NPL + 1;
  const int64_t priv = NP * (NPL == 32 ? 40 : LDM) + 3 * NPL + 64;     // NPL = 32: image widened for the tiled Gauss–Jordan
  int64_t d = 2 * priv + NP * LDM + 2 * NPL + 2 * 64 + 8 + (NPL + 64 + 8) / 2 + 3LL * (T + 1) * NPL + NPL * mcap + (int64_t)T * mcap +
              (int64_t)(nzA + nzAc + nzB) * NPL + (int64_t)nzBc * 64;
  int64_t i = (int64_t)(nzA + nzAc + nzB) * NPL + (int64_t)nzBc * 64;
  return d * 8 + i * 4 + ((int64_t)T * nm_max + 15) / 16 * 16 + 16;
}

// LDS bytes of the four-wave twisted kernel (must match the carve in twisted4_solve_column; NPL = 32 classes only)
static inline int64_t twisted4_kernel_lds_bytes(int cls, int T, int mcap, int nzA, int nzAc, int nzB, int nzBc, int nm_max) {
  const WaveClass w = wave_class(cls);
  const int64_t NPL = w.npl, NP = (64 / w.npl) * w.rpl;
  const int64_t TR = (NP + 7) / 8, NR = 8 * TR, LDT = 40, RS = (TR == 3) ? 32 : 48;
  const int64_t privc = NR * LDT + 2 * NPL, privh = 0, dirsz = NR * RS + NR * LDT + 2;
  int64_t d = 2 * privc + 2 * privh + 2 * dirsz + NR * LDT + 2 * NPL + 2 * 64 + 8 + (NPL + 64 + 8) / 2 + 4LL * (T + 1) * NPL + NPL * mcap +
              2LL * T * mcap + (int64_t)(nzA + nzAc + nzB) * NPL + (int64_t)nzBc * 64;
  int64_t i = (int64_t)(nzA + nzAc + nzB) * NPL + (int64_t)nzBc * 64;
  return d * 8 + i * 4 + ((int64_t)T * nm_max + 15) / 16 * 16 + 16;
}

// ---- device mask recipe (sls_masks.hip) ----
struct MaskParams {
  int32_t Nx, Nu, T, kmax, base;
  const int32_t* A_cp;      // CSC of (A≠0) by value: colptr[Nx+1], rowval — edges q → r
  const int32_t* A_ri;
  const int32_t* B_rp;      // CSR of (B2≠0) by value: rowptr[Nx+1], colidx — actuators touching state r
  const int32_t* B_ci;
  const int32_t* kx;        // [T]
  const int32_t* ku;        // [T]
  int32_t cap;              // capacity of a level list (entries)
  int32_t* cntx;            // [Nx][kmax+1] out (count pass) / in (fill pass)
  int32_t* cntu;
  const int64_t* prex;      // fill pass: [kmax+1][Nx] exclusive prefix over columns of cntx[·][k]
  const int64_t* preu;
  const int64_t* offx;      // fill pass: [T] offset of 𝓢x[t]'s row indices in rowx
  const int64_t* offu;
  int64_t* rowx;            // fill pass out: all row indices of 𝓢x[0..T), then (rowu) of 𝓢u
  int64_t* rowu;
  int32_t* overflow;        // set when a level does not fit `cap`
};


// Index sets of every single-column subproblem (reference src/reduction.jl:11-27 with cⱼ = {c}):
//   s_x(c) = rows of (𝓢x[T]·(A≠0))[:,c] = ∪_{k ∈ rows(A[:,c])} rows(𝓢x[T][:,k]),   s_u(c) likewise with 𝓢u[T].
struct IndexSetParams {
  int32_t Nx, Nu, base, cap;
  const int32_t* A_cp;  const int32_t* A_ri;     // CSC of (A≠0) by value
  const int32_t* Sx_cp; const int32_t* Sx_ri;    // CSC of the last state mask, every stored entry (findnz is structural)
  const int32_t* Su_cp; const int32_t* Su_ri;    // CSC of the last input mask
  int32_t* cntx;            // [Nx] out (count pass)
  int32_t* cntu;
  const int64_t* ptrx;      // fill pass: [Nx+1] 0-based offsets into outx
  const int64_t* ptru;
  int64_t* outx;            // fill pass out: ascending indices in the caller's base
  int64_t* outu;
  int32_t* overflow;
};

// Per-column tables of the README mask recipe built on the device (sls_masks.hip: column_tables_kernel): index sets, compact
// bit masks and first destinations of every single-column subproblem, straight from the plant pattern — no mask crosses PCIe.
// max(|a|, running maximum) of a residual pass that does not lose a NaN: fmax / v_max_f64 return the other operand, so a
// column whose factorisation broke down (NaN in z) used to report a residual of 0 and pass as converged (tools/fuzz_h2.py seed
// 401, column 51 through the tile kernel: status OK with NaN values).  A NaN becomes +inf and stays: the column ends flagged.
static inline __host__ __device__ double resid_max(double running, double a) {
  const double av = a < 0.0 ? -a : a;
  return (av <= running) ? running : ((av == av) ? av : 1.0 / 0.0);
}

// A pass left more than `stag` (half) of the residual: inconsistent system, or a consistent one that is merely slow?
// An inconsistent column approaches its least-squares residual: grid-32's 646 and random10000_d2's 9 896 infeasible columns show
// r2/r1 ≥ 0.97 in 76 % / 100 % of the cases at the second pass and in every case by the fourth (tools/stag_ratio_hist.py).  A
// consistent column with σ_min(E)² ≈ δ contracts by a constant factor per pass — 0.78 in the one-wave kernel, 0.5 / 0.7
// alternating under the tile kernel's minimal-residual steps on tools/fuzz_h2.py seed 235, column 52 (σ(E) = …, 1.7e-2, 1.7e-5,
// 9.0e-7; 30 passes to 1e-12), which rounds 1–2 flagged infeasible at pass 3.  Rule: above the acceptance level, a column that
// still loses 10 % per pass (or 19 % over two) goes on, under the larger pass cap `max_iters_slow` — and, once it has been granted
// that (itmax raised), also below the acceptance level, down to `tol`: its error in Φ is residual/σ_min, the acceptance level
// alone would leave 5e-4 on that column.
// r0, r1, r2: residuals of the last three iterates (r0 = r1 when only two are known).
static inline __host__ __device__ bool still_contracting(double r0, double r1, double r2) {
  return r2 < 0.9 * r1 || r2 < 0.81 * r0;
}

struct ColumnTableParams {
  int32_t Nx, Nu, T;
  int32_t kmax;             // highest level a mask uses (max over t of kx, ku)
  int32_t KL;               // highest level expanded: max(kmax, kx[T−1] + 1, ku[T−1] + 1)
  int32_t qx1, qu1;         // levels that ARE the index sets: s_x = L_{qx1}(c), s_u = actuators of L_{qu1}(c)
  int32_t cap;              // capacity of each level pool (entries, all levels of one column together)
  const int32_t* A_cp;  const int32_t* A_ri;     // CSC of (A≠0) by value: successors of a state
  const int32_t* B_rp;  const int32_t* B_ci;     // CSR of (B2≠0) by value: actuators touching a state
  const int32_t* A_rowptr; const int32_t* A_colidx; const double* A_val;   // CSR of A   (nnz of Ã)
  const int32_t* B_rowptr; const int32_t* B_colidx; const double* B_val;   // CSR of B2  (nnz of B̃2)
  const int32_t* kx;    const int32_t* ku;       // [T]
  // count pass (out)
  int32_t* cntx;  int32_t* cntu;                 // [Nx][kmax+1] level sizes
  int32_t* col_info;                             // [Nx][6]: ñx, ñu, pos, nnz(Ã), nnz(B̃2), free variables
  int32_t* flags;                                // [0] a level pool overflowed, [1] a mask row lies outside its index set
  // fill pass (in)
  const int64_t* prex;  const int64_t* preu;     // [kmax+1][Nx] exclusive prefix over the columns of the level sizes
  const int64_t* offx;  const int64_t* offu;     // [T] offset of each time step's slice in the mask-order value array
  const int64_t* idx_off;                        // [Nx] first entry of the column's (s_x, s_u) in idx_pool
  const int64_t* cw_off;                         // [Nx] first word of the column in cmask
  // fill pass (out)
  int32_t* idx_pool;  uint64_t* cmask;  int32_t* cbase;
};

}  // namespace sls
