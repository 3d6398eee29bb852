/* sls_mi355x.h — C ABI of the MI355X-native column-separable H2 SLS engine.
 *
 * Drop-in boundary for ONE path of aaltoKEPO/SystemLevelControl.jl:
 *     Φx,Φu = SLS_𝓗₂(P, [𝓢x,𝓢u]; 𝓘)            (reference src/synthesis.jl:11-32)
 * i.e. the @distributed per-column loop _SLS_𝓗₂ (src/synthesis.jl:34-72) with
 * sparsity_dim_reduction (src/reduction.jl:11-27) and the JuMP/Ipopt solve
 * (src/synthesis.jl:46-62) replaced by batched HIP kernels for gfx950.
 *
 * Conventions
 *  - plain C, plain pointers and sizes; no C++/torch types cross this boundary.
 *  - every matrix is handed over exactly as Julia stores a SparseMatrixCSC{Tv,Int}
 *    (reference src/types/GeneralizedPlant.jl:47-55): colptr[ncols+1], rowval[nnz]
 *    (sorted within a column), nzval[nnz]; indices are int64 and may be 1-based
 *    (Julia) or 0-based (C/Python) — sls_dims.index_base says which.
 *  - the caller owns every buffer; the library keeps no host pointer after a
 *    call returns.  Calls block.  No callbacks, no signal handlers, no abort():
 *    every failure is a negative return code plus sls_last_error().
 *  - return codes: 0 = ok; <0 = SLS_E*; >0 (solve calls only) = number of
 *    subproblems whose status is not SLS_COL_OK (their values are still written;
 *    see col_status).  The reference itself never checks the solver status
 *    (src/synthesis.jl:62-65 goes straight from optimize! to value.).
 */
#ifndef SLS_MI355X_H
#define SLS_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLS_ABI_VERSION 2   /* 2: sls_stats.n_refined appended (round 2) */

/* ---- error codes (negative returns) ---- */
#define SLS_EINVAL      (-1)  /* bad argument / inconsistent dimensions           */
#define SLS_ENOTSF      (-2)  /* weights not LQR-shaped: Nz != Nx+Nu (the reference's
                                 view() hard-codes z-rows [s_x; Nx+s_u],
                                 src/reduction.jl:15)                              */
#define SLS_EUNSUPPORTED (-3) /* valid input this build cannot solve (a cost
                                 that leaves a free variable without weight; a
                                 coupled group with a column outside its s_x)      */
#define SLS_EHIP        (-4)  /* HIP runtime error (message in sls_last_error)     */
#define SLS_ENOMEM      (-5)
#define SLS_ENODEVICE   (-6)  /* no gfx950 device / kernels not loadable          */

/* ---- per-subproblem status words (col_status[], one per column of every group) ---- */
#define SLS_COL_OK          0
#define SLS_COL_INFEASIBLE  1 /* E z = f has no solution: the residual stops contracting above the acceptance level (a pass
                                 that leaves more than half of it AND no longer removes 10 % per pass / 19 % over two: a slowly
                                 but steadily contracting column — nearly dependent constraints — is carried on, up to 48
                                 passes); values are the minimum-norm least-squares point — except when the factorisation itself
                                 broke down on an inconsistent, nearly rank-deficient column (reported residual = inf): the column
                                 is flagged all the same, its values are then unspecified.  The reference hands every column to
                                 Ipopt and never reads its status (src/synthesis.jl:62-65).                          */
#define SLS_COL_NOTCONV     2 /* refinement hit the iteration cap above tolerance   */
#define SLS_COL_TRIVIAL     3 /* column not in its own s_x (Ĩ column is zero): Φ = 0 */
#define SLS_COL_SKIPPED     4 /* not owned by this plan's shard                     */
#define SLS_COL_UNSUPPORTED 5 /* not solved, values stay 0.0 (the other columns of the call are solved as usual).  Since round 3
                                 the default build never reports it: an index set whose working set exceeds LDS runs the tile
                                 kernel with that working set in a global buffer (no size limit but device memory).  Only the
                                 experiment switches SLS_TILE=0 / SLS_TILE_BIG=0 (round-1 / round-2 launch lists) bring it back. */

/* ---- sls_create flags ---- */
#define SLS_CREATE_DEFAULT  0u

/* ---- sls_dims.flags ---- */
#define SLS_SOLVE_DEFAULT   0u
#define SLS_SOLVE_SUM_OF_NORMS 1u /* per column: minimise Σ_t ‖[C̃1 D̃12]Φ̃[t]B̃1‖₂ (sum of the per-time-step norms) instead of the 𝓗₂
                                    sum of squares, over the same index sets, masks and achievability constraints — the
                                    column-separable bound of the 𝓗∞ norm, a second-order cone program per column.  NOT in the
                                    reference (it has no 𝓗∞ synthesis; BASELINE configs[3] names one).  Needs a diagonal
                                    [C1 D12]ᵀ[C1 D12] and D11 = 0, else SLS_EUNSUPPORTED.  ADMM whose projection step is the 𝓗₂ machinery (one-wave
                                    kernel for ñx ≤ 32, tile kernel beyond); status SLS_COL_NOTCONV when the step cap is hit. */

/* Julia SparseMatrixCSC{Float64,Int}  (reference src/types/GeneralizedPlant.jl:47-52) */
typedef struct sls_csc_f64 {
  int64_t nrows, ncols;
  const int64_t* colptr;   /* ncols+1 */
  const int64_t* rowval;   /* nnz     */
  const double*  nzval;    /* nnz     */
} sls_csc_f64;

/* Julia SparseMatrixCSC{Bool,Int}: one byte per stored entry.  nzval == NULL means
 * "every stored entry is true".  A stored `false` is NOT in the mask (the reference
 * tests `.≠ 1`, src/synthesis.jl:58-59) but IS part of the structural pattern that
 * sparsity_dim_reduction sees through findnz (src/reduction.jl:14).                */
typedef struct sls_csc_bool {
  int64_t nrows, ncols;
  const int64_t* colptr;
  const int64_t* rowval;
  const uint8_t* nzval;
} sls_csc_bool;

/* P.Nx, P.Nu, P.Nz, P.Nw (reference src/types/GeneralizedPlant.jl:56) and
 * T = length(𝓢x) (src/synthesis.jl:20). */
typedef struct sls_dims {
  int64_t Nx, Nu, Nz, Nw, T;
  int32_t index_base;      /* 1 = Julia, 0 = C   (applies to every colptr/rowval/group_cols) */
  uint32_t flags;          /* SLS_SOLVE_*        */
} sls_dims;

/* The nine-block plant as SLS_𝓗₂ reads it (state feedback: C2 = I, D21/D22 empty are
 * implied and not passed — reference src/types/GeneralizedPlant.jl:91-95).
 * C1, D11, D12 may each be NULL: then the 3-argument Plant(A,B1,B2) defaults apply
 * ([C1 D12] = I, D11 = 0 — src/types/GeneralizedPlant.jl:105-110).                  */
typedef struct sls_plant {
  const sls_csc_f64 *A, *B1, *B2, *C1, *D11, *D12;
} sls_plant;

typedef struct sls_stats {
  int64_t n_subproblems;       /* Σ_groups |c_j|                                   */
  int64_t n_not_ok;
  int64_t n_values_x, n_values_u;   /* Σ_t nnz(𝓢x[t]), Σ_t nnz(𝓢u[t])                */
  int64_t n_free;              /* Σ free variables actually solved for              */
  int32_t max_nx, max_nu;      /* max |s_x|, |s_u| over subproblems                 */
  int32_t max_iters;           /* max refinement passes used by any subproblem      */
  int32_t n_devices;
  double  max_residual;        /* max over OK subproblems of ‖E z − f‖∞             */
  double  flops_alg;           /* Σ F_alg, algorithmic FP64 flops      */
  double  bytes_alg;           /* Σ B_alg, algorithmic HBM bytes       */
  double  t_symbolic_s;        /* host symbolic pass (replaces src/reduction.jl)    */
  double  t_upload_s;          /* H2D of the shared operator + per-column tables    */
  double  t_solve_s;           /* device solve, wall clock around the launches      */
  double  t_download_s;        /* D2H of Φ values                                    */
  int64_t n_refined;           /* columns solved a second time on the tile kernel (slow convergence: near-singular) */
} sls_stats;

typedef struct sls_ctx  sls_ctx;
typedef struct sls_plan sls_plan;

/* ---- context ------------------------------------------------------------------
 * One context owns `ndev` HIP devices (device ordinals in devs[]); the reference
 * analogue is the worker pool of `julia -p N` (src/synthesis.jl:16 nworkers()).
 * devs == NULL selects devices 0..ndev-1.  Returns NULL on failure
 * (sls_last_error(NULL) then describes it).                                       */
sls_ctx* sls_create(const int* devs, int ndev, uint32_t flags);
void     sls_destroy(sls_ctx* ctx);
const char* sls_last_error(const sls_ctx* ctx);   /* ctx may be NULL                 */
int      sls_abi_version(void);
int      sls_device_count(void);                  /* gfx950 devices visible, <0 on error */

/* ---- the drop-in call  (replaces reference src/synthesis.jl:11-32 + :34-72) -----
 *  P        : plant blocks (see sls_plant)
 *  Sx, Su   : arrays of T masks (Nx×Nx and Nu×Nx)               — 𝓢 = [𝓢x, 𝓢u]
 *  groups   : 𝓘 as CSR-like lists: group g holds columns
 *             group_cols[group_ptr[g] .. group_ptr[g+1]) (index_base applies to
 *             group_cols, group_ptr is always 0-based offsets).  ngroups = 0 and
 *             NULL pointers select the default [[i] for i in 1:Nx]
 *             (src/synthesis.jl:15).  Columns ascend strictly inside a group.  A column
 *             listed in several groups is solved once per group, with that group's index
 *             sets, and the contributions are ADDED — what the reference's Φ̃ += … and
 *             (+) fold do (src/synthesis.jl:24,67).  (Only this call: a plan gives every
 *             subproblem its own destinations, so sls_h2_sf_plan / sls_shard_groups /
 *             sls_h2_sf_packed_layout answer such a list with SLS_EINVAL.)
 *  phix_vals[t] / phiu_vals[t] : caller-allocated arrays of nnz(𝓢x[t]) / nnz(𝓢u[t])
 *             doubles, filled IN THE MASK'S CSC nzval ORDER, so that
 *             SparseMatrixCSC(Nx,Nx,𝓢x[t].colptr,𝓢x[t].rowval,phix_vals[t]) is Φx[t]
 *             with a pattern that is the mask's bit for bit (entries the reference
 *             would drop as numerical zeros are stored 0.0; src/synthesis.jl:65-67).
 *             Columns that belong to no group keep 0.0.
 *  col_status : NULL or int32[Σ|c_j|], in group order (SLS_COL_*)
 *  stats      : NULL or filled on return
 * A column's subproblem is solved with its group's index sets s_x, s_u
 * (src/reduction.jl:14).  Cost weights: any [C1 D12] with Nz = Nx+Nu rows
 * (src/synthesis.jl:50,76-83); when [C1 D12]ᵀ[C1 D12] is diagonal on (s_x,s_u) (every
 * Plant(A,B1,B2), every diagonally weighted LQR) the column runs on the kernel of its
 * size class, otherwise on the tile kernel with projected conjugate gradients over the
 * diagonal-weight solve.  The cost couples the columns of a multi-column group only
 * through the block B1[c_j,c_j] (src/synthesis.jl:42): diagonal block ⇒ the group's QP
 * separates by column; otherwise the group is solved jointly (one work item of the tile
 * kernel: conjugate gradients over all its columns, Hessian (B̃1B̃1ᵀ)⊗[C̃1 D̃12]ᵀ[C̃1 D̃12]);
 * a group containing an infeasible column is then flagged as a whole.
 * Columns of the small size classes that converged slowly (four or more passes and a residual above 1e-11, or
 * SLS_COL_NOTCONV: a near-singular constraint matrix) are solved once more on the tile kernel's minimal-residual iteration
 * before the download; stats->n_refined counts them.                                     */
int sls_h2_sf_solve(sls_ctx* ctx, const sls_dims* dims, const sls_plant* P,
                    const sls_csc_bool* Sx, const sls_csc_bool* Su,
                    int64_t ngroups, const int64_t* group_ptr, const int64_t* group_cols,
                    double* const* phix_vals, double* const* phiu_vals,
                    int32_t* col_status, sls_stats* stats);

/* The reference's objective is norm(H, 𝓗₂) + L⁺([Φ̃x,Φ̃u], c_j) with the hook L⁺ hard-wired to 0 (src/synthesis.jl:21,52).
 * Its diagonal quadratic instance — the ridge term  Σ_t Σ_i rx[i]·Φx[t][i,c]² + Σ_j ru[j]·Φu[t][j,c]²  added to every column's
 * cost — is available on the context: the weights (≥ 0, lengths Nx and Nu of the plants solved afterwards; 0 = none for that
 * part) apply to every later sls_h2_sf_solve / sls_h2_sf_plan / batch call on `ctx` until replaced; (0, NULL, 0, NULL) clears.
 * With default weights and B1 = I this equals solving with [C1 D12] = diag(√(1 + r)).  Not built for the sum-of-norms mode. */
int sls_set_ridge(sls_ctx* ctx, int64_t nx, const double* rx, int64_t nu, const double* ru);

/* A batch of independent plants in ONE call and ONE set of kernel launches (latency regime: the README plant's 59 columns fill
 * a quarter of the CUs; four of them take the time of one).  The reference's counterpart is a loop of SLS_𝓗₂ calls
 * (src/synthesis.jl:11), one per plant.  The plants are solved as the block-diagonal composite plant — column c of plant i has
 * exactly the index sets, masks and constraints it has alone, so every per-column result is the one the single call returns —
 * with default groups (one per column).  All plants share T, index_base and flags (dims[i].T etc. must agree).
 *   dims[nplants], P[nplants];  Sx[i], Su[i]: the T masks of plant i;  phix_vals[i][t], phiu_vals[i][t] as in
 *   sls_h2_sf_solve;  col_status: NULL or col_status[i] = NULL / array of dims[i].Nx.  Returns like sls_h2_sf_solve. */
int sls_h2_sf_solve_batch(sls_ctx* ctx, int nplants, const sls_dims* dims, const sls_plant* P,
                          const sls_csc_bool* const* Sx, const sls_csc_bool* const* Su,
                          double* const* const* phix_vals, double* const* const* phiu_vals,
                          int32_t* const* col_status, sls_stats* stats);

/* ---- the same path split into plan / execute ------------------------------------
 * A plan = symbolic pass + everything resident in HBM on ONE device of the context
 * (shared operator A,B2 in CSR; per-subproblem index sets, masks, destination
 * table, factor workspace).  It owns the subproblems of groups
 * [group_begin, group_end) — the shard of this rank (src/synthesis.jl:16 static
 * contiguous chunking; here the caller chooses the cut, see sls_shard_groups).
 * sls_plan_execute is the hot path proper: device-resident in, device-resident out. */
typedef struct sls_plan_info {
  int64_t n_subproblems;       /* owned by this plan                                */
  int64_t n_values;            /* length of the full value array: n_values_x+n_values_u */
  int64_t n_values_x, n_values_u;
  int64_t n_packed;            /* # values this plan writes (its free variables)     */
  int64_t workspace_bytes;     /* device bytes held by the plan                      */
  int32_t max_nx, max_nu, T;
  int32_t device;              /* HIP ordinal                                        */
  double  flops_alg, bytes_alg;/* Σ over owned subproblems              */
  double  t_symbolic_s, t_upload_s;
} sls_plan_info;

int  sls_h2_sf_plan(sls_ctx* ctx, int dev_slot, const sls_dims* dims, const sls_plant* P,
                    const sls_csc_bool* Sx, const sls_csc_bool* Su,
                    int64_t ngroups, const int64_t* group_ptr, const int64_t* group_cols,
                    int64_t group_begin, int64_t group_end, sls_plan** plan_out);
int  sls_plan_get_info(const sls_plan* plan, sls_plan_info* info);

/* Value-array layout ("mask order"): [t=0..T-1: nnz(𝓢x[t]) doubles] then
 * [t=0..T-1: nnz(𝓢u[t]) doubles].  sls_plan_value_offsets fills off_x[T+1], off_u[T+1]
 * (offsets of each t's slice in that array; off_x[T] = n_values_x = off_u[0]).       */
int  sls_plan_value_offsets(const sls_plan* plan, int64_t* off_x, int64_t* off_u);

/* Run the solve on `hip_stream` (a hipStream_t cast to void*; NULL = HIP's null stream,
 * which is also torch's default stream).  d_values: DEVICE pointer.
 *   packed == 0: d_values has n_values doubles; the plan writes only its own entries
 *                (caller zero-fills once; entries of other shards are untouched).
 *   packed == 1: d_values has n_packed doubles; entry k goes to mask-order position
 *                dest[k] (sls_plan_packed_dest) — the layout the RCCL all-gather moves.
 * Asynchronous w.r.t. the host: returns after enqueueing.  Status/residuals stay on
 * the device until sls_plan_fetch_status.                                           */
int  sls_plan_execute(sls_plan* plan, void* hip_stream, double* d_values, int packed);
/* Several resident plans of one device in one call: plan 0 runs on `hip_stream`, the others on their plan-owned streams, which
 * first wait for everything enqueued on `hip_stream` before the call (fork edge) and are joined back into `hip_stream` by an
 * event each: work enqueued on `hip_stream` BEFORE the call has finished with d_values[i] when a plan overwrites it, work
 * enqueued AFTERWARDS sees all results; the host is never blocked.  (SLS_BATCH_FORK=0 in the environment drops the fork
 * edge: then the caller must guarantee that nothing still pending on `hip_stream` reads or writes any d_values[i], i >= 1.)
 * Measured (4 README plans): 0.27 ms per call against 0.48 ms for four calls in a row — a cross-queue event wait costs
 * ≈0.1 ms on this stack; a caller that can merge its plants up front should use sls_h2_sf_solve_batch (one launch, 0.126 ms
 * for the same four).  d_values[i] / packed as above; no plan may appear twice. */
int  sls_plan_execute_batch(sls_plan* const* plans, int nplans, void* hip_stream, double* const* d_values, int packed);
int  sls_plan_synchronize(sls_plan* plan, void* hip_stream);
int  sls_plan_packed_dest(const sls_plan* plan, int64_t* dest /* n_packed, host */);
/* col_status / residual / iters: NULL or arrays of n_subproblems (host). Synchronises. */
int  sls_plan_fetch_status(sls_plan* plan, int32_t* col_status, double* residual, int32_t* iters);
/* Refinement for the resident path (sls_h2_sf_solve does this by itself).  After an execute: reads the statuses, and for the
 * groups whose one-wave / twisted columns stopped between 1e-11 and the acceptance level after four or more passes, ended
 * NOTCONV, or were called infeasible at a small residual (by a twisted kernel: at any residual, after three or more passes) — the
 * signature of a near-singular constraint matrix, where Φ is
 * only determined to residual/σ_min — builds a second plan on the tile kernel (minimal-residual multiplier iteration, 1e-13),
 * runs it on `hip_stream` into `d_values` (layout `packed` as in sls_plan_execute: the refinement numbers its free variables
 * where `plan` put them, so it writes either layout in place) and ATTACHES it to `plan`: every later sls_plan_execute runs it
 * after the main launches, sls_plan_fetch_status reports the refined outcome, and sls_plan_destroy frees it.  The plan does
 * not keep its inputs: pass the same dims / P / masks / group list it was built from.  Waits for the device;
 * *n_refined = subproblems re-solved (0: nothing qualified, nothing attached).  𝓗₂ objective only.                       */
int  sls_plan_refine(sls_plan* plan, const sls_dims* dims, const sls_plant* P,
                     const sls_csc_bool* Sx, const sls_csc_bool* Su,
                     int64_t ngroups, const int64_t* group_ptr, const int64_t* group_cols,
                     void* hip_stream, double* d_values, int packed, int64_t* n_refined);
/* average device time of the solve kernel over the launches since the last call, from
 * HIP events recorded on the launch stream around every sls_plan_execute.             */
int  sls_plan_kernel_time_ms(sls_plan* plan, double* avg_ms, int64_t* n_launches);
/* Human-readable list of the kernels one sls_plan_execute launches ("name nsub= grid= block= lds=;" per launch). */
int  sls_plan_describe(const sls_plan* plan, char* buf, int64_t buflen);
/* convenience for non-torch callers: device buffer management on the plan's device.  */
int  sls_plan_alloc_values(sls_plan* plan, int packed, double** d_values_out);
int  sls_plan_free_values(sls_plan* plan, double* d_values);
int  sls_plan_download(sls_plan* plan, const double* d_values /* mask order */,
                       double* const* phix_vals, double* const* phiu_vals);
void sls_plan_destroy(sls_plan* plan);

/* d_dst[idx[k]] = d_src[k], k < n  — unpack of the all-gathered packed shards.       */
int  sls_scatter_f64(sls_ctx* ctx, int dev_slot, void* hip_stream,
                     const double* d_src, const int64_t* d_idx, int64_t n, double* d_dst);

/* Cost-balanced contiguous cut of the groups into `nshards` ranges (cost model
 * Σ (T+1)·ñx³).  Fills cuts[nshards+1].  Pure host; needs no device.     */
int  sls_shard_groups(const sls_dims* dims, const sls_plant* P,
                      const sls_csc_bool* Sx, const sls_csc_bool* Su,
                      int64_t ngroups, const int64_t* group_ptr, const int64_t* group_cols,
                      int nshards, int64_t* cuts);

/* Host-only symbolic pass for groups [group_begin, group_end): how many values the shard
 * produces (n_packed), the length of the full value array (n_values) and, when dest !=
 * NULL, the mask-order destination of every packed value (dest[n_packed]) — identical to
 * what a plan over the same range reports.  info (nullable) receives the symbolic part of
 * sls_plan_info (device = -1).  Needs no device: lets every rank know every other rank's
 * layout without communication, and lets the N>1 gather path be tested on CPU ranks.  */
int  sls_h2_sf_packed_layout(const sls_dims* dims, const sls_plant* P,
                             const sls_csc_bool* Sx, const sls_csc_bool* Su,
                             int64_t ngroups, const int64_t* group_ptr, const int64_t* group_cols,
                             int64_t group_begin, int64_t group_end,
                             int64_t* n_packed, int64_t* n_values, int64_t* dest, sls_plan_info* info);

/* ---- (d,T)-localization masks: the README recipe (reference README.md:52-54), not part of the package itself -------
 *     𝓢x[t] = (A.≠0)^min(d,  ⌊α(t−1)⌋) .≠ 0                    t = 1..T   (Nx×Nx)
 *     𝓢u[t] = (B₂'.≠0)·(A.≠0)^min(d+1,⌊α(t−1)⌋) .≠ 0           t = 1..T   (Nu×Nx)
 * computed per column as level sets of exact k-step walks (= Boolean matrix powers), on host threads.  Two-call
 * protocol: with rowval_x == NULL only the counts are returned — nnz_x[t], nnz_u[t] (T entries each); the caller then
 * allocates, for every t, colptr_x[t] (Nx+1), rowval_x[t] (nnz_x[t]), colptr_u[t] (Nx+1), rowval_u[t] (nnz_u[t]) and
 * calls again.  Indices follow dims->index_base; rows ascend inside a column (a valid SparseMatrixCSC{Bool,Int} pattern,
 * every stored value true).  Only dims->Nx, Nu, T and index_base are read.  Pure host; needs no device.               */
int  sls_localization_masks(const sls_dims* dims, const sls_csc_f64* A, const sls_csc_f64* B2, int64_t d, double alpha,
                            int64_t* nnz_x, int64_t* nnz_u,
                            int64_t* const* colptr_x, int64_t* const* rowval_x,
                            int64_t* const* colptr_u, int64_t* const* rowval_u);

/* The same recipe computed on the device: one wave per column expands the level sets of A's pattern with an
 * LDS bitmap, in a count launch and a fill launch (csrc/sls_masks.hip); only A's and B2's patterns go up and only the Int64 row
 * indices come down.  Same arguments and two-call protocol as sls_localization_masks, bit-identical output; SLS_EUNSUPPORTED
 * when the state bitmap or a level set does not fit LDS (then use the host version). */
int  sls_localization_masks_device(sls_ctx* ctx, int dev_slot, const sls_dims* dims, const sls_csc_f64* A, const sls_csc_f64* B2,
                                   int64_t d, double alpha, int64_t* nnz_x, int64_t* nnz_u,
                                   int64_t* const* colptr_x, int64_t* const* rowval_x,
                                   int64_t* const* colptr_u, int64_t* const* rowval_u);

/* ---- the same solve with the masks never leaving the device (SURVEY §8 row f1 on the solve path) --------------------------
 * For callers whose masks ARE the README recipe (reference README.md:52-54: 𝓢x[t] = (A.≠0)^min(d,⌊α(t−1)⌋), 𝓢u[t] =
 * (B₂'.≠0)(A.≠0)^min(d+1,⌊α(t−1)⌋)): the plan is built from (A, B₂, d, α, T) alone.  Level sets, index sets
 * (src/reduction.jl:14), the per-column mask slices (src/synthesis.jl:57-60) and the destinations (src/synthesis.jl:65-67) are
 * computed by device kernels (csrc/sls_masks.hip: column_tables_kernel) and stay in HBM; no mask array crosses PCIe in either
 * direction.  The value arrays come back in the CSC order of exactly the masks sls_localization_masks[_device] returns for the
 * same (d, α, T) — a caller that wants Φ as sparse matrices fetches the patterns with that call, a caller that feeds
 * sls_closed_loop_* or its own device code needs no pattern at all.
 * Restrictions (SLS_EUNSUPPORTED otherwise; pass the masks to sls_h2_sf_plan / sls_h2_sf_solve then): default groups [[i] for
 * i in 1:Nx]; the 3-argument Plant's cost ([C1 D12] = I, D11 = 0; C1/D11/D12 NULL or equal to that by value); no ridge term;
 * every mask row inside its column's index set (holds whenever A has a full diagonal, as every README-style plant has);
 * d + 2 ≤ 62; the whole plant on ONE device (the plan covers all columns; mask-order output only, packed = 0).
 * dims->T is the horizon; dims->flags selects the objective as usual.  Plans built this way are executed, queried and
 * destroyed like any other (sls_plan_execute with packed = 0, sls_plan_fetch_status, sls_plan_download, sls_plan_destroy). */
int  sls_h2_sf_plan_localized(sls_ctx* ctx, int dev_slot, const sls_dims* dims, const sls_plant* P, int64_t d, double alpha,
                              sls_plan** plan_out);
/* One-shot form on device slot 0: plan (as above) + solve + download.  phix_vals[t] / phiu_vals[t], col_status, stats and the
 * return value as in sls_h2_sf_solve. */
int  sls_h2_sf_solve_localized(sls_ctx* ctx, const sls_dims* dims, const sls_plant* P, int64_t d, double alpha,
                               double* const* phix_vals, double* const* phiu_vals, int32_t* col_status, sls_stats* stats);

/* ---- closed-loop simulation with an on-device Φ (reference README.md:62-72; a user script there, not package code) ----
 *     β[:,t+1] = Σ_{τ=1..min(t,T−1)} Φx[τ+1]·(x[:,t+1−τ] − β[:,t+1−τ])
 *     u[:,t]   = Σ_{τ=1..min(t,T)}   Φu[τ]  ·(x[:,t+1−τ] − β[:,t+1−τ])
 *     x[:,t+1] = A·x[:,t] + B₁·w(t) + B₂·u[:,t]              t = 1..steps−1,   x[:,1] = β[:,1] = 0
 * for `nscen` disturbance scenarios at once.  sls_closed_loop_plan builds the row-oriented FIR operator from the masks
 * (P->A, B1, B2 are read; Φx must be square: Sx[t] is Nx×Nx) on device `dev_slot` of the context.
 * sls_closed_loop_run: d_values = Φ in the mask-order value array a plan's sls_plan_execute(packed = 0) fills (DEVICE);
 *   d_w [steps][Nw][nscen] (DEVICE; NULL = no disturbance; w(steps) is never read),
 *   d_x [steps][Nx][nscen], d_u [steps][Nu][nscen] (DEVICE, written; time-major, scenario innermost; x[0] = 0, u[steps−1] = 0
 *   as in the README's arrays).  Enqueues on `hip_stream` (NULL = null stream) and returns; steps−1 kernels replayed
 *   from a hipGraph that is cached while (d_w, d_x, d_u, steps, nscen) stay the same.
 * sls_closed_loop_run_host: same with HOST w/x/u (staged through device buffers; d_values stays a device pointer). */
typedef struct sls_loop sls_loop;
int  sls_closed_loop_plan(sls_ctx* ctx, int dev_slot, const sls_dims* dims, const sls_plant* P,
                          const sls_csc_bool* Sx, const sls_csc_bool* Su, sls_loop** loop_out);
int  sls_closed_loop_run(sls_loop* loop, void* hip_stream, const double* d_values, const double* d_w,
                         int64_t steps, int64_t nscen, double* d_x, double* d_u);
int  sls_closed_loop_run_host(sls_loop* loop, const double* d_values, const double* h_w,
                              int64_t steps, int64_t nscen, double* h_x, double* h_u);
/* device time of the last sls_closed_loop_run (HIP events on its stream); synchronises on it */
int  sls_closed_loop_last_ms(sls_loop* loop, double* ms);
/* stored-true Φ entries one time step reads (12 B each: the HBM traffic model of the step kernel) */
int  sls_closed_loop_entries(const sls_loop* loop, int64_t* n_entries);
void sls_closed_loop_destroy(sls_loop* loop);

/* Symbolic pass only (replaces reference src/reduction.jl:11-27 for one group):
 * fills s_x / s_u (index_base of dims), returns their lengths.  Pure host.           */
int  sls_sparsity_dim_reduction(const sls_dims* dims, const sls_csc_f64* A,
                                const sls_csc_bool* Sx_last, const sls_csc_bool* Su_last,
                                const int64_t* cj, int64_t ncj,
                                int64_t* sx_out, int64_t* nsx, int64_t* su_out, int64_t* nsu);

/* The same sets for EVERY single-column subproblem c = 0..Nx-1 at once, on the device (one wave per
 * column, LDS bitmap union of the last masks' columns — csrc/sls_masks.hip).  CSR-style output: s_x(c) = sx_idx[sx_ptr[c]-b ..
 * sx_ptr[c+1]-b), ascending (the order the destination tables use; the reference's unique(findnz) order is the same set),
 * b = dims->index_base; likewise s_u.  Two-call protocol: with sx_idx == NULL only sx_ptr / su_ptr (Nx+1 each) are filled;
 * the caller allocates sx_idx (sx_ptr[Nx]-b entries) and su_idx and calls again.  SLS_EUNSUPPORTED when the state bitmap
 * does not fit LDS.                                                                                                    */
int  sls_index_sets_device(sls_ctx* ctx, int dev_slot, const sls_dims* dims, const sls_csc_f64* A,
                           const sls_csc_bool* Sx_last, const sls_csc_bool* Su_last,
                           int64_t* sx_ptr, int64_t* sx_idx, int64_t* su_ptr, int64_t* su_idx);

#ifdef __cplusplus
}
#endif
#endif /* SLS_MI355X_H */
