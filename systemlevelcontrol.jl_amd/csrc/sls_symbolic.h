// sls_symbolic.h — host symbolic pass of the H2 column solve (pure C++, no HIP).
//
// Replaces, once for all columns, what the reference recomputes per column:
//   src/reduction.jl:14     s_x, s_u  = rows of nz((𝓢[end]·(A≠0))[:,c_j])
//   src/reduction.jl:15     sub-plant view  (index bookkeeping only)
//   src/reduction.jl:22-23  ii_x / Ĩ  (position of the impulse inside s_x)
//   src/synthesis.jl:57-60  mask slices 𝓢x[t][s_x,c_j], 𝓢u[t][s_u,c_j]
//   src/synthesis.jl:65-66  destination of every solved entry in Φx[t], Φu[t]
//   src/synthesis.jl:42,50  B̃1 diagonal block and the (diagonal) cost weights
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <functional>
#include <vector>

#include "../../include/sls_mi355x.h"
#include "sls_device.h"

// Diagnostic knobs (the SLS_* environment variables) are honoured in LAB MODE only — SLS_LAB=1 in the environment when
// the library first looks (tests/conftest.py and the scripts under tools/ set it).  A process that does not ask for it gets the
// shipped routing, tolerances and launch plans whatever SLS_* variables it happens to inherit.
#include <cstdlib>
static inline const char* sls_knob(const char* name) {
  static const bool lab = [] { const char* e = std::getenv("SLS_LAB"); return e != nullptr && e[0] == '1'; }();
  return lab ? std::getenv(name) : nullptr;
}


namespace sls {

// std::vector whose resize(n) leaves trivially-constructible elements uninitialised: the spliced pools are sized once and
// filled by several threads; value-initialising tens of MB serially (and faulting the pages in on one thread) cost more
// than the per-group work itself
template <class T>
struct default_init_alloc : std::allocator<T> {
  template <class U> struct rebind { using other = default_init_alloc<U>; };
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U;
    else ::new (static_cast<void*>(p)) U(static_cast<A&&>(a)...);
  }
};
template <class T> using pool_vec = std::vector<T, default_init_alloc<T>>;

// 0-based int32 CSR with explicit values (host copy of a Julia CSC, possibly transposed)
struct HostCsr {
  int64_t nrows = 0, ncols = 0;
  std::vector<int32_t> ptr;   // nrows+1
  std::vector<int32_t> idx;   // nnz (ascending within a row)
  std::vector<double> val;    // nnz
};

struct GroupSets {
  std::vector<int32_t> sx, su;          // ascending global indices
  std::vector<int32_t> sx_first, su_first;  // first-appearance order (findnz/unique semantics)
};

struct Symbolic {
  int64_t Nx = 0, Nu = 0, T = 0;
  // value-array offsets ("mask order"): x slices then u slices
  std::vector<int64_t> off_x, off_u;    // T+1 each; off_u[0] == off_x[T]
  int64_t n_values = 0;
  // shared operator
  HostCsr A_csr, At_csr, B_csr, Bt_csr;
  int32_t max_row_A = 1, max_row_At = 1, max_row_B = 1, max_row_Bt = 1;   // longest rows (LDS list capacities)
  int32_t max_nm = 1;                                                      // max ñx+ñu over owned subproblems
  int32_t max_wz = 0;                                                      // max # z-rows of a general (non-diagonal) weight record
  // owned subproblems
  pool_vec<SubDesc> subs;
  std::vector<int32_t> order;           // descending cost
  pool_vec<int32_t> idx_pool;
  pool_vec<uint8_t> mask_pool;
  pool_vec<int32_t> dest_pool;          // destinations in the mask-order value array
  pool_vec<int32_t> pdest_pool;         // destinations in the packed array
  pool_vec<int64_t> packed_to_final;    // n_packed
  pool_vec<double> w_pool;
  pool_vec<int32_t> sub_col;            // global column of each subproblem
  int64_t n_packed = 0;
  std::vector<int64_t> pk_base;         // n_subs+1: first packed index of each subproblem (its free variables are contiguous)
  // input, optional: packed bases to use instead of pk_base — the refinement plan of sls_plan_refine numbers its subproblems'
  // free variables where the plan it refines put them, so that it can write into that plan's packed array; packed_to_final is
  // then not built
  std::vector<int64_t> pk_override;
  bool want_packed = true;              // false: pdest_pool / packed_to_final are not built (n_packed still is)
  // Compact tables (requested by the caller, one-device drop-in call): instead of mask_pool / dest_pool (5 B per masked
  // position, the bulk of the pass's writes and of the H2D copy) the pass leaves, per subproblem and time step, a bit mask
  // over (s_x, s_u) and the destinations of the first kept x / u entry; the device expands them (expand_tables_kernel in
  // sls_kernels.hip).  Valid when every column is REGULAR: all stored mask entries true, rows inside the index sets and
  // ascending, so that the k-th set bit of a part goes to base + k.  build_symbolic clears `compact` and fills the explicit
  // pools when some column is not.
  bool compact = false;
  pool_vec<uint64_t> cmask;             // per subproblem T × ceil((ñx+ñu)/64) words, bit i = position i of (s_x, s_u) is free
  pool_vec<int32_t> cbase;              // per subproblem T × 2: value-array index of the first x / first u entry of the column
  pool_vec<int64_t> coff;               // per subproblem: its first word in cmask
  int64_t md_total = 0;                 // length of mask_pool / dest_pool (whether or not they are materialised on the host)
  int32_t max_n = 0, max_m = 0, max_nnzA = 0, max_nnzB = 0;
  double flops_alg = 0.0, bytes_alg = 0.0;
  int64_t n_total_subproblems = 0;      // over ALL groups (for col_status indexing)
  int64_t first_sub_index = 0;          // index of this shard's first subproblem in the global order
};

struct Inputs {
  const sls_dims* dims;
  const sls_plant* P;
  const sls_csc_bool* Sx;
  const sls_csc_bool* Su;
  int64_t ngroups;
  const int64_t* group_ptr;
  const int64_t* group_cols;
  // optional ridge term Σ_t Σ_i rx[i]·Φx[t][i,c]² + Σ_j ru[j]·Φu[t][j,c]² added to every column's cost (sls_set_ridge: the
  // diagonal instance of the reference's L⁺ hook, src/synthesis.jl:21,52); NULL = none
  const double* reg_x = nullptr;   // Nx
  const double* reg_u = nullptr;   // Nu
};

// Host worker threads shared by the symbolic pass and the pinned download: f(0..n-1) run concurrently (f(0) on the calling
// thread); returns when all are done.  The workers are created on first use and parked between jobs.
void host_parallel(int n, const std::function<void(int)>& f);

// returns 0 or SLS_E*; msg filled on error
int validate_inputs(const Inputs& in, std::string& msg);

// index sets of one group (0-based inputs already normalised through `in`)
int group_index_sets(const Inputs& in, const int64_t* cols0, int64_t ncols, GroupSets& out, std::string& msg);

// default groups [[i] for i in 1:Nx] or the caller's, normalised to 0-based
void normalise_groups(const Inputs& in, std::vector<int64_t>& gptr, std::vector<int64_t>& gcols);

// full symbolic pass for groups [gbeg, gend)
int build_symbolic(const Inputs& in, int64_t gbeg, int64_t gend, Symbolic& out, std::string& msg);

// README.md:52-54 mask recipe (see sls_localization_masks in the public header)
int localization_masks(const sls_dims* dims, const sls_csc_f64* A, const sls_csc_f64* B2, int64_t d, double alpha, int64_t* nnz_x,
                       int64_t* nnz_u, int64_t* const* colptr_x, int64_t* const* rowval_x, int64_t* const* colptr_u,
                       int64_t* const* rowval_u, std::string& msg);

// inputs of the device mask recipe (sls_masks.hip): the same checks as localization_masks, the level schedule kx(t), ku(t),
// and the patterns by value as 0-based int32 — CSC of (A≠0), CSR of (B2≠0)
int mask_recipe_inputs(const sls_dims* dims, const sls_csc_f64* A, const sls_csc_f64* B2, int64_t d, double alpha,
                       std::vector<int32_t>& kx, std::vector<int32_t>& ku, int& kmax, std::vector<int32_t>& a_cp,
                       std::vector<int32_t>& a_ri, std::vector<int32_t>& b_rp, std::vector<int32_t>& b_ci, std::string& msg);

// inputs of the device index-set pass (sls_masks.hip: index_sets_kernel): the patterns by value as 0-based int32 CSC
int index_set_inputs(const sls_dims* dims, const sls_csc_f64* A, const sls_csc_bool* Sx_last, const sls_csc_bool* Su_last,
                     std::vector<int32_t>& a_cp, std::vector<int32_t>& a_ri, std::vector<int32_t>& sx_cp, std::vector<int32_t>& sx_ri,
                     std::vector<int32_t>& su_cp, std::vector<int32_t>& su_ri, std::string& msg);

// Host part of the device-resident symbolic route (sls_h2_sf_plan_localized): plant checks, the level schedule and the plant
// patterns of the README recipe, the shared operator in CSR (S.A_csr … S.Bt_csr, longest rows) — everything that does not
// need a mask.  SLS_EUNSUPPORTED when the route does not apply (cost weights other than the 3-argument Plant's, d + 2 > 62).
struct LocalizedHost {
  std::vector<int32_t> kx, ku, a_cp, a_ri, b_rp, b_ci;
  int kmax = 0;
};
int localized_prepare(const sls_dims* dims, const sls_plant* P, int64_t d, double alpha, Symbolic& S, LocalizedHost& L, std::string& msg);

// predicted cost per group (Σ over its columns of (T+1)·ñx³)
int group_costs(const Inputs& in, std::vector<double>& cost, std::string& msg);

// ---- closed-loop simulator (README.md:62-72): the FIR operators β[t+1] = Σ_τ Φx[τ+1]·ŵ[t+1−τ], u[t] = Σ_τ Φu[τ]·ŵ[t+1−τ]
// as ROW lists over all lags, each entry pointing back into the mask-order value array.  Row i of the β operator gathers
// the stored-true entries (i, c) of 𝓢x[τ+1], τ = 1..T−1; row j of the u operator those of 𝓢u[τ], τ = 1..T; inside a row
// entries ascend in (τ, c).  `hoff = τ·Nx − c` addresses the disturbance-estimate history as (k+T)·Nx − hoff at step k.
struct FirOperator {
  int64_t Nx = 0, Nu = 0, Nw = 0, T = 0, n_values = 0;
  std::vector<int32_t> beta_ptr;   // Nx+1, offsets into hoff/perm
  std::vector<int32_t> u_ptr;      // Nu+1, offsets into hoff/perm (the u entries follow the β entries)
  std::vector<int32_t> hoff;       // n_entries
  std::vector<int32_t> perm;       // n_entries: index in the mask-order value array
  HostCsr A, B1, B2;               // plant, row-oriented
  std::vector<int32_t> orphan;     // actuators that drive no state (empty B2 column): still reported in u
};
int build_fir_operator(const sls_dims* dims, const sls_csc_f64* A, const sls_csc_f64* B1, const sls_csc_f64* B2,
                       const sls_csc_bool* Sx, const sls_csc_bool* Su, FirOperator& out, std::string& msg);

}  // namespace sls
