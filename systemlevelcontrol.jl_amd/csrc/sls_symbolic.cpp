// sls_symbolic.cpp — see sls_symbolic.h.  Pure host code (compiled by hipcc or g++).
#include "sls_symbolic.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <chrono>
#include <cstdio>

namespace sls {

namespace {


template <class M>
int check_csc(const M* m, int64_t nr, int64_t nc, int base, const char* name, std::string& msg) {
  if (!m) { msg = std::string(name) + ": null matrix"; return SLS_EINVAL; }
  if (m->nrows != nr || m->ncols != nc) {
    msg = std::string(name) + ": expected " + std::to_string(nr) + "x" + std::to_string(nc) + ", got " +
          std::to_string(m->nrows) + "x" + std::to_string(m->ncols);
    return SLS_EINVAL;
  }
  if (!m->colptr) { msg = std::string(name) + ": null colptr"; return SLS_EINVAL; }
  if (m->colptr[0] != base) { msg = std::string(name) + ": colptr[0] != index_base"; return SLS_EINVAL; }
  for (int64_t c = 0; c < nc; ++c) {
    const int64_t b = m->colptr[c] - base, e = m->colptr[c + 1] - base;
    if (e < b) { msg = std::string(name) + ": colptr not monotone"; return SLS_EINVAL; }
    if (e > b && !m->rowval) { msg = std::string(name) + ": null rowval"; return SLS_EINVAL; }
    int64_t prev = -1;
    for (int64_t k = b; k < e; ++k) {
      const int64_t r = m->rowval[k] - base;
      if (r < 0 || r >= nr) { msg = std::string(name) + ": row index out of range"; return SLS_EINVAL; }
      if (r <= prev) { msg = std::string(name) + ": row indices not strictly ascending within a column"; return SLS_EINVAL; }
      prev = r;
    }
  }
  if (m->colptr[nc] - base > 0x7fffffffLL) { msg = std::string(name) + ": nnz exceeds int32"; return SLS_EUNSUPPORTED; }
  return 0;
}

// CSR of the matrix itself (transpose of the Julia CSC storage)
void csc_to_csr(const sls_csc_f64* m, int base, HostCsr& out) {
  out.nrows = m->nrows; out.ncols = m->ncols;
  const int64_t nnz = m->colptr[m->ncols] - base;
  out.ptr.assign(m->nrows + 1, 0);
  out.idx.resize(nnz); out.val.resize(nnz);
  for (int64_t k = 0; k < nnz; ++k) out.ptr[m->rowval[k] - base + 1]++;
  for (int64_t r = 0; r < m->nrows; ++r) out.ptr[r + 1] += out.ptr[r];
  std::vector<int32_t> w(out.ptr.begin(), out.ptr.end() - 1);
  for (int64_t c = 0; c < m->ncols; ++c)
    for (int64_t k = m->colptr[c] - base; k < m->colptr[c + 1] - base; ++k) {
      const int64_t r = m->rowval[k] - base;
      out.idx[w[r]] = (int32_t)c;
      out.val[w[r]] = m->nzval ? m->nzval[k] : 1.0;
      ++w[r];
    }
}

// the CSC storage reinterpreted as CSR of the transpose
void csc_as_csr_of_transpose(const sls_csc_f64* m, int base, HostCsr& out) {
  out.nrows = m->ncols; out.ncols = m->nrows;
  const int64_t nnz = m->colptr[m->ncols] - base;
  out.ptr.resize(m->ncols + 1);
  out.idx.resize(nnz); out.val.resize(nnz);
  for (int64_t c = 0; c <= m->ncols; ++c) out.ptr[c] = (int32_t)(m->colptr[c] - base);
  for (int64_t k = 0; k < nnz; ++k) {
    out.idx[k] = (int32_t)(m->rowval[k] - base);
    out.val[k] = m->nzval ? m->nzval[k] : 1.0;
  }
}

bool weights_are_default(const Inputs& in) {
  const sls_plant* P = in.P;
  const int base = in.dims->index_base;
  const int64_t Nx = in.dims->Nx, Nu = in.dims->Nu;
  if (!P->C1 && !P->D12) {
    // D11 alone may still be given
  } else {
    if (!P->C1 || !P->D12) return false;
    // [C1 D12] == I(Nx+Nu) by value
    const sls_csc_f64* mats[2] = {P->C1, P->D12};
    const int64_t shift[2] = {0, Nx};
    const int64_t ncs[2] = {Nx, Nu};
    for (int q = 0; q < 2; ++q) {
      const sls_csc_f64* m = mats[q];
      for (int64_t c = 0; c < ncs[q]; ++c) {
        bool diag = false;
        for (int64_t k = m->colptr[c] - base; k < m->colptr[c + 1] - base; ++k) {
          const double v = m->nzval ? m->nzval[k] : 1.0;
          if (v == 0.0) continue;
          if (m->rowval[k] - base == c + shift[q] && v == 1.0) diag = true; else return false;
        }
        if (!diag) return false;
      }
    }
  }
  if (P->D11 && P->D11->nzval) {
    const int64_t nnz = P->D11->colptr[P->D11->ncols] - base;
    for (int64_t k = 0; k < nnz; ++k) if (P->D11->nzval[k] != 0.0) return false;
  } else if (P->D11 && !P->D11->nzval && (P->D11->colptr[P->D11->ncols] - base) > 0) {
    return false;
  }
  return true;
}

}  // namespace

int validate_inputs(const Inputs& in, std::string& msg) {
  if (!in.dims || !in.P) { msg = "null dims/plant"; return SLS_EINVAL; }
  const sls_dims& d = *in.dims;
  if (d.index_base != 0 && d.index_base != 1) { msg = "index_base must be 0 or 1"; return SLS_EINVAL; }
  if (d.Nx <= 0 || d.Nu < 0 || d.T <= 0 || d.Nw <= 0) { msg = "Nx, Nw, T must be positive"; return SLS_EINVAL; }
  if (d.Nx > 0x7fffffffLL || d.Nu > 0x7fffffffLL) { msg = "Nx/Nu exceed int32"; return SLS_EUNSUPPORTED; }
  const int b = d.index_base;
  int rc;
  if ((rc = check_csc(in.P->A, d.Nx, d.Nx, b, "A", msg))) return rc;
  if ((rc = check_csc(in.P->B1, d.Nx, d.Nw, b, "B1", msg))) return rc;
  if ((rc = check_csc(in.P->B2, d.Nx, d.Nu, b, "B2", msg))) return rc;
  if (in.P->C1 || in.P->D12 || in.P->D11) {
    if (d.Nz != d.Nx + d.Nu) {
      msg = "Nz != Nx+Nu: sparsity_dim_reduction's view() assumes z-rows [s_x; Nx+s_u] (reference src/reduction.jl:15)";
      return SLS_ENOTSF;
    }
    if (in.P->C1 && (rc = check_csc(in.P->C1, d.Nz, d.Nx, b, "C1", msg))) return rc;
    if (in.P->D12 && (rc = check_csc(in.P->D12, d.Nz, d.Nu, b, "D12", msg))) return rc;
    if (in.P->D11 && (rc = check_csc(in.P->D11, d.Nz, d.Nw, b, "D11", msg))) return rc;
    if ((in.P->C1 == nullptr) != (in.P->D12 == nullptr)) { msg = "C1 and D12 must be given together"; return SLS_EINVAL; }
  }
  if (!in.Sx || !in.Su) { msg = "null mask arrays"; return SLS_EINVAL; }
  for (int64_t t = 0; t < d.T; ++t) {
    if ((rc = check_csc(&in.Sx[t], d.Nx, d.Nx, b, "Sx[t]", msg))) return rc;
    if ((rc = check_csc(&in.Su[t], d.Nu, d.Nx, b, "Su[t]", msg))) return rc;
  }
  if (in.ngroups < 0) { msg = "ngroups < 0"; return SLS_EINVAL; }
  if (in.ngroups > 0) {
    if (!in.group_ptr || !in.group_cols) { msg = "null group arrays"; return SLS_EINVAL; }
    if (in.group_ptr[0] != 0) { msg = "group_ptr[0] != 0"; return SLS_EINVAL; }
    // A column may belong to ONE group only: every subproblem writes its column of Φ, so two groups holding the same column
    // would be two launches racing for the same destinations (the reference would SUM their contributions, Φ̃ += …,
    // src/synthesis.jl:24,67 — which is never what a partition 𝓘 of the columns means).
    std::vector<uint8_t> seen((size_t)d.Nx, 0);
    for (int64_t g = 0; g < in.ngroups; ++g) {
      if (in.group_ptr[g + 1] < in.group_ptr[g]) { msg = "group_ptr not monotone"; return SLS_EINVAL; }
      int64_t prev = -1;
      for (int64_t k = in.group_ptr[g]; k < in.group_ptr[g + 1]; ++k) {
        const int64_t c = in.group_cols[k] - b;
        if (c < 0 || c >= d.Nx || c >= d.Nw) { msg = "group column out of range"; return SLS_EINVAL; }
        if (c <= prev) { msg = "columns of a group must be strictly ascending"; return SLS_EINVAL; }
        if (seen[(size_t)c]) { msg = "column " + std::to_string(c + b) + " appears in more than one group"; return SLS_EINVAL; }
        seen[(size_t)c] = 1;
        prev = c;
      }
    }
  } else if (d.Nw < d.Nx) {
    msg = "default groups 1:Nx need Nw >= Nx"; return SLS_EINVAL;
  }
  return 0;
}

void normalise_groups(const Inputs& in, std::vector<int64_t>& gptr, std::vector<int64_t>& gcols) {
  const int b = in.dims->index_base;
  if (in.ngroups == 0) {   // src/synthesis.jl:15
    gptr.resize(in.dims->Nx + 1); gcols.resize(in.dims->Nx);
    for (int64_t i = 0; i < in.dims->Nx; ++i) { gptr[i] = i; gcols[i] = i; }
    gptr[in.dims->Nx] = in.dims->Nx;
  } else {
    gptr.assign(in.group_ptr, in.group_ptr + in.ngroups + 1);
    gcols.resize(gptr.back());
    for (size_t k = 0; k < gcols.size(); ++k) gcols[k] = in.group_cols[k] - b;
  }
}

// rows of nz((S_last·(A≠0))[:,c]) merged over the group's columns — src/reduction.jl:14
static void product_rows(const sls_csc_f64* A, const sls_csc_bool* S, int base, const int64_t* cols, int64_t ncols,
                         std::vector<int32_t>& stamp_group, std::vector<int32_t>& stamp_col, int32_t& stamp_ctr,
                         std::vector<int32_t>& first, std::vector<int32_t>& tmp) {
  first.clear();
  const int32_t gstamp = ++stamp_ctr;
  for (int64_t q = 0; q < ncols; ++q) {
    const int64_t c = cols[q];
    const int32_t cstamp = ++stamp_ctr;
    tmp.clear();
    for (int64_t ka = A->colptr[c] - base; ka < A->colptr[c + 1] - base; ++ka) {
      if ((A->nzval ? A->nzval[ka] : 1.0) == 0.0) continue;          // (P.A .≠ 0) is by value
      const int64_t k = A->rowval[ka] - base;
      for (int64_t ks = S->colptr[k] - base; ks < S->colptr[k + 1] - base; ++ks) {   // structural (findnz)
        const int32_t r = (int32_t)(S->rowval[ks] - base);
        if (stamp_col[r] != cstamp) { stamp_col[r] = cstamp; tmp.push_back(r); }
      }
    }
    std::sort(tmp.begin(), tmp.end());
    for (int32_t r : tmp)
      if (stamp_group[r] != gstamp) { stamp_group[r] = gstamp; first.push_back(r); }
  }
}

int group_index_sets(const Inputs& in, const int64_t* cols0, int64_t ncols, GroupSets& out, std::string& msg) {
  (void)msg;
  const int base = in.dims->index_base;
  const int64_t T = in.dims->T;
  static thread_local std::vector<int32_t> sgx, scx, sgu, scu, tmp;
  static thread_local int32_t ctr_x = 0, ctr_u = 0;
  if ((int64_t)sgx.size() != in.dims->Nx || ctr_x > 0x7ffffff0) { sgx.assign(in.dims->Nx, 0); scx.assign(in.dims->Nx, 0); ctr_x = 0; }
  if ((int64_t)sgu.size() != in.dims->Nu || ctr_u > 0x7ffffff0) { sgu.assign(in.dims->Nu, 0); scu.assign(in.dims->Nu, 0); ctr_u = 0; }
  product_rows(in.P->A, &in.Sx[T - 1], base, cols0, ncols, sgx, scx, ctr_x, out.sx_first, tmp);
  product_rows(in.P->A, &in.Su[T - 1], base, cols0, ncols, sgu, scu, ctr_u, out.su_first, tmp);
  out.sx = out.sx_first; std::sort(out.sx.begin(), out.sx.end());
  out.su = out.su_first; std::sort(out.su.begin(), out.su.end());
  return 0;
}

// ---- README.md:52-54: 𝓢x[t] = (A≠0)^kx(t) ≠ 0, 𝓢u[t] = (B2ᵀ≠0)(A≠0)^ku(t) ≠ 0, kx = min(d,⌊αt⌋), ku = min(d+1,⌊αt⌋), t = 0..T−1 ----
// Column c of (A≠0)^k = rows reachable from c by walks of EXACTLY k steps along A's pattern (edge k→r iff A[r,k] ≠ 0).
int localization_masks(const sls_dims* dims, const sls_csc_f64* A, const sls_csc_f64* B2, int64_t d, double alpha, int64_t* nnz_x,
                       int64_t* nnz_u, int64_t* const* colptr_x, int64_t* const* rowval_x, int64_t* const* colptr_u,
                       int64_t* const* rowval_u, std::string& msg) {
  const int base = dims->index_base;
  const int64_t Nx = dims->Nx, Nu = dims->Nu, T = dims->T;
  if (base != 0 && base != 1) { msg = "index_base must be 0 or 1"; return SLS_EINVAL; }
  if (Nx <= 0 || Nu < 0 || T <= 0 || d < 0 || !(alpha >= 0.0)) { msg = "bad Nx/Nu/T/d/alpha"; return SLS_EINVAL; }
  int rc;
  if ((rc = check_csc(A, Nx, Nx, base, "A", msg))) return rc;
  if ((rc = check_csc(B2, Nx, Nu, base, "B2", msg))) return rc;
  const bool fill = rowval_x != nullptr;
  if (fill && (!colptr_x || !colptr_u || !rowval_u)) { msg = "null output arrays"; return SLS_EINVAL; }
  std::vector<int> kx(T), ku(T);
  int kmax = 0;
  for (int64_t t = 0; t < T; ++t) {
    const int64_t f = (int64_t)std::floor(alpha * (double)t);
    kx[t] = (int)std::min<int64_t>(d, f); ku[t] = (int)std::min<int64_t>(d + 1, f);
    kmax = std::max(kmax, std::max(kx[t], ku[t]));
  }
  // row pattern of B2 by value: actuators touching state r
  HostCsr Bcsr; csc_to_csr(B2, base, Bcsr);
  // per column: sizes of every level (x) and of the actuator sets (u)
  std::vector<int32_t> cntx((size_t)Nx * (kmax + 1)), cntu((size_t)Nx * (kmax + 1));
  unsigned hw = std::thread::hardware_concurrency();
  int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(hw ? hw : 1, 16), Nx / 256));
  if (const char* e = sls_knob("SLS_SYMBOLIC_THREADS")) nthreads = std::max(1, std::atoi(e));
  auto levels_of = [&](int64_t c, std::vector<std::vector<int32_t>>& lev, std::vector<std::vector<int32_t>>& act,
                       std::vector<int32_t>& stamp, int32_t& ctr, std::vector<int32_t>& astamp) {
    lev[0].assign(1, (int32_t)c);
    for (int k = 1; k <= kmax; ++k) {
      auto& nx = lev[k]; nx.clear();
      const int32_t s = ++ctr;
      for (int32_t q : lev[k - 1])
        for (int64_t e = A->colptr[q] - base; e < A->colptr[q + 1] - base; ++e) {
          if ((A->nzval ? A->nzval[e] : 1.0) == 0.0) continue;
          const int32_t r = (int32_t)(A->rowval[e] - base);
          if (stamp[r] != s) { stamp[r] = s; nx.push_back(r); }
        }
      std::sort(nx.begin(), nx.end());
    }
    for (int k = 0; k <= kmax; ++k) {
      auto& a = act[k]; a.clear();
      const int32_t s = ++ctr;
      for (int32_t r : lev[k])
        for (int32_t e = Bcsr.ptr[r]; e < Bcsr.ptr[r + 1]; ++e) {
          if (Bcsr.val[e] == 0.0) continue;
          const int32_t j = Bcsr.idx[e];
          if (astamp[j] != s) { astamp[j] = s; a.push_back(j); }
        }
      std::sort(a.begin(), a.end());
    }
  };
  auto worker = [&](int tix, bool do_fill) {
    std::vector<std::vector<int32_t>> lev(kmax + 1), act(kmax + 1);
    std::vector<int32_t> stamp(Nx, 0), astamp(std::max<int64_t>(Nu, 1), 0);
    int32_t ctr = 0;
    const int64_t c0 = Nx * tix / nthreads, c1 = Nx * (tix + 1) / nthreads;
    for (int64_t c = c0; c < c1; ++c) {
      if (ctr > 0x7ffff000) { std::fill(stamp.begin(), stamp.end(), 0); std::fill(astamp.begin(), astamp.end(), 0); ctr = 0; }
      levels_of(c, lev, act, stamp, ctr, astamp);
      if (!do_fill) {
        for (int k = 0; k <= kmax; ++k) { cntx[(size_t)c * (kmax + 1) + k] = (int32_t)lev[k].size(); cntu[(size_t)c * (kmax + 1) + k] = (int32_t)act[k].size(); }
      } else {
        for (int64_t t = 0; t < T; ++t) {
          int64_t o = colptr_x[t][c] - base;
          for (int32_t r : lev[kx[t]]) rowval_x[t][o++] = r + base;
          o = colptr_u[t][c] - base;
          for (int32_t j : act[ku[t]]) rowval_u[t][o++] = j + base;
        }
      }
    }
  };
  auto run_all = [&](bool do_fill) {
    if (nthreads == 1) { worker(0, do_fill); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(worker, t, do_fill);
    for (auto& x : th) x.join();
  };
  run_all(false);
  for (int64_t t = 0; t < T; ++t) {
    int64_t sx = 0, su = 0;
    for (int64_t c = 0; c < Nx; ++c) { sx += cntx[(size_t)c * (kmax + 1) + kx[t]]; su += cntu[(size_t)c * (kmax + 1) + ku[t]]; }
    nnz_x[t] = sx; nnz_u[t] = su;
  }
  if (!fill) return 0;
  for (int64_t t = 0; t < T; ++t) {
    if (!colptr_x[t] || !colptr_u[t] || (nnz_x[t] && !rowval_x[t]) || (nnz_u[t] && !rowval_u[t])) { msg = "null output array for some t"; return SLS_EINVAL; }
    colptr_x[t][0] = base; colptr_u[t][0] = base;
    for (int64_t c = 0; c < Nx; ++c) {
      colptr_x[t][c + 1] = colptr_x[t][c] + cntx[(size_t)c * (kmax + 1) + kx[t]];
      colptr_u[t][c + 1] = colptr_u[t][c] + cntu[(size_t)c * (kmax + 1) + ku[t]];
    }
  }
  run_all(true);
  return 0;
}

int mask_recipe_inputs(const sls_dims* dims, const sls_csc_f64* A, const sls_csc_f64* B2, int64_t d, double alpha,
                       std::vector<int32_t>& kx, std::vector<int32_t>& ku, int& kmax, std::vector<int32_t>& a_cp,
                       std::vector<int32_t>& a_ri, std::vector<int32_t>& b_rp, std::vector<int32_t>& b_ci, std::string& msg) {
  const int base = dims->index_base;
  const int64_t Nx = dims->Nx, Nu = dims->Nu, T = dims->T;
  if (base != 0 && base != 1) { msg = "index_base must be 0 or 1"; return SLS_EINVAL; }
  if (Nx <= 0 || Nu < 0 || T <= 0 || d < 0 || !(alpha >= 0.0)) { msg = "bad Nx/Nu/T/d/alpha"; return SLS_EINVAL; }
  int rc;
  if ((rc = check_csc(A, Nx, Nx, base, "A", msg))) return rc;
  if ((rc = check_csc(B2, Nx, Nu, base, "B2", msg))) return rc;
  kx.resize(T); ku.resize(T); kmax = 0;
  for (int64_t t = 0; t < T; ++t) {
    const int64_t f = (int64_t)std::floor(alpha * (double)t);
    kx[t] = (int32_t)std::min<int64_t>(d, f); ku[t] = (int32_t)std::min<int64_t>(d + 1, f);
    kmax = std::max(kmax, std::max(kx[t], ku[t]));
  }
  a_cp.assign(Nx + 1, 0); a_ri.clear();
  for (int64_t c = 0; c < Nx; ++c) {
    for (int64_t e = A->colptr[c] - base; e < A->colptr[c + 1] - base; ++e)
      if ((A->nzval ? A->nzval[e] : 1.0) != 0.0) a_ri.push_back((int32_t)(A->rowval[e] - base));
    a_cp[c + 1] = (int32_t)a_ri.size();
  }
  HostCsr Bcsr; csc_to_csr(B2, base, Bcsr);
  b_rp.assign(Nx + 1, 0); b_ci.clear();
  for (int64_t r = 0; r < Nx; ++r) {
    for (int32_t e = Bcsr.ptr[r]; e < Bcsr.ptr[r + 1]; ++e)
      if (Bcsr.val[e] != 0.0) b_ci.push_back(Bcsr.idx[e]);
    b_rp[r + 1] = (int32_t)b_ci.size();
  }
  return 0;
}

int localized_prepare(const sls_dims* dims, const sls_plant* P, int64_t d, double alpha, Symbolic& S, LocalizedHost& L, std::string& msg) {
  if (!dims || !P) { msg = "null dims/plant"; return SLS_EINVAL; }
  const sls_dims& dm = *dims;
  if (dm.index_base != 0 && dm.index_base != 1) { msg = "index_base must be 0 or 1"; return SLS_EINVAL; }
  if (dm.Nx <= 0 || dm.Nu < 0 || dm.T <= 0 || dm.Nw <= 0) { msg = "Nx, Nw, T must be positive"; return SLS_EINVAL; }
  if (dm.Nx > 0x7fffffffLL || dm.Nu > 0x7fffffffLL) { msg = "Nx/Nu exceed int32"; return SLS_EUNSUPPORTED; }
  if (dm.Nw < dm.Nx) { msg = "default groups 1:Nx need Nw >= Nx"; return SLS_EINVAL; }
  const int b = dm.index_base;
  int rc;
  if ((rc = check_csc(P->A, dm.Nx, dm.Nx, b, "A", msg))) return rc;
  if ((rc = check_csc(P->B1, dm.Nx, dm.Nw, b, "B1", msg))) return rc;
  if ((rc = check_csc(P->B2, dm.Nx, dm.Nu, b, "B2", msg))) return rc;
  if (P->C1 || P->D12 || P->D11) {
    if (dm.Nz != dm.Nx + dm.Nu) { msg = "Nz != Nx+Nu: sparsity_dim_reduction's view() assumes z-rows [s_x; Nx+s_u] (reference src/reduction.jl:15)"; return SLS_ENOTSF; }
    if (P->C1 && (rc = check_csc(P->C1, dm.Nz, dm.Nx, b, "C1", msg))) return rc;
    if (P->D12 && (rc = check_csc(P->D12, dm.Nz, dm.Nu, b, "D12", msg))) return rc;
    if (P->D11 && (rc = check_csc(P->D11, dm.Nz, dm.Nw, b, "D11", msg))) return rc;
    if ((P->C1 == nullptr) != (P->D12 == nullptr)) { msg = "C1 and D12 must be given together"; return SLS_EINVAL; }
  }
  Inputs in{dims, P, nullptr, nullptr, 0, nullptr, nullptr};
  if (!weights_are_default(in)) {
    msg = "the device-resident symbolic route is built for the 3-argument Plant's cost ([C1 D12] = I, D11 = 0); pass the masks to sls_h2_sf_plan for other weights";
    return SLS_EUNSUPPORTED;
  }
  if ((rc = mask_recipe_inputs(dims, P->A, P->B2, d, alpha, L.kx, L.ku, L.kmax, L.a_cp, L.a_ri, L.b_rp, L.b_ci, msg))) return rc;
  if (L.kmax + 2 > 62) { msg = "localization radius d + 2 > 62 levels: use the mask-based entry point"; return SLS_EUNSUPPORTED; }
  S.Nx = dm.Nx; S.Nu = dm.Nu; S.T = dm.T;
  csc_to_csr(P->A, b, S.A_csr);
  csc_as_csr_of_transpose(P->A, b, S.At_csr);
  csc_to_csr(P->B2, b, S.B_csr);
  csc_as_csr_of_transpose(P->B2, b, S.Bt_csr);
  auto longest = [](const HostCsr& M) {
    int32_t mx = 1;
    for (int64_t r = 0; r < M.nrows; ++r) mx = std::max(mx, M.ptr[r + 1] - M.ptr[r]);
    return mx;
  };
  S.max_row_A = longest(S.A_csr); S.max_row_At = longest(S.At_csr);
  S.max_row_B = longest(S.B_csr); S.max_row_Bt = longest(S.Bt_csr);
  return 0;
}

int index_set_inputs(const sls_dims* dims, const sls_csc_f64* A, const sls_csc_bool* Sx_last, const sls_csc_bool* Su_last,
                     std::vector<int32_t>& a_cp, std::vector<int32_t>& a_ri, std::vector<int32_t>& sx_cp, std::vector<int32_t>& sx_ri,
                     std::vector<int32_t>& su_cp, std::vector<int32_t>& su_ri, std::string& msg) {
  const int base = dims->index_base;
  const int64_t Nx = dims->Nx, Nu = dims->Nu;
  if (base != 0 && base != 1) { msg = "index_base must be 0 or 1"; return SLS_EINVAL; }
  if (Nx <= 0 || Nu < 0) { msg = "bad Nx/Nu"; return SLS_EINVAL; }
  int rc;
  if ((rc = check_csc(A, Nx, Nx, base, "A", msg))) return rc;
  if ((rc = check_csc(Sx_last, Nx, Nx, base, "Sx[T]", msg))) return rc;
  if ((rc = check_csc(Su_last, Nu, Nx, base, "Su[T]", msg))) return rc;
  // as product_rows above: (P.A .≠ 0) is by value, the masks count structurally (findnz of a sparse product keeps every stored
  // entry of 𝓢[end], true or false)
  auto pattern = [&](auto* m, bool by_value, std::vector<int32_t>& cp, std::vector<int32_t>& ri) {
    cp.assign(Nx + 1, 0); ri.clear();
    for (int64_t c = 0; c < Nx; ++c) {
      for (int64_t e = m->colptr[c] - base; e < m->colptr[c + 1] - base; ++e)
        if (!by_value || !m->nzval || m->nzval[e] != 0) ri.push_back((int32_t)(m->rowval[e] - base));
      cp[c + 1] = (int32_t)ri.size();
    }
  };
  pattern(A, true, a_cp, a_ri); pattern(Sx_last, false, sx_cp, sx_ri); pattern(Su_last, false, su_cp, su_ri);
  return 0;
}

int group_costs(const Inputs& in, std::vector<double>& cost, std::string& msg) {
  std::vector<int64_t> gptr, gcols;
  normalise_groups(in, gptr, gcols);
  const int64_t ng = (int64_t)gptr.size() - 1;
  cost.assign(ng, 0.0);
  GroupSets gs;
  for (int64_t g = 0; g < ng; ++g) {
    int rc = group_index_sets(in, gcols.data() + gptr[g], gptr[g + 1] - gptr[g], gs, msg);
    if (rc) return rc;
    const double n = (double)gs.sx.size();
    cost[g] = (double)(gptr[g + 1] - gptr[g]) * ((double)(in.dims->T + 1) * n * n * n + 1.0);
  }
  return 0;
}

// Host worker threads of the symbolic pass, created once per process and parked on a condition variable between passes:
// spawning 16 std::threads three times per call cost ≈1 ms of a 4 ms one-shot chain-4096 call.  Never destroyed (the
// workers are parked when the process exits; joining them from a static destructor would race library unloading).
class HostPool {
 public:
  void run(int n, const std::function<void(int)>& f) {
    std::lock_guard<std::mutex> serial(run_mu_);              // one job at a time, whoever calls
    grow(n - 1);
    {
      std::lock_guard<std::mutex> l(mu_);
      job_ = &f; want_ = n - 1; next_ = 1; pending_ = n - 1; ++gen_;
    }
    cv_.notify_all();
    f(0);
    std::unique_lock<std::mutex> l(mu_);
    done_cv_.wait(l, [&] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  void grow(int workers) {
    while ((int)threads_.size() < workers) {
      threads_.emplace_back([this] {
        uint64_t seen = 0;
        for (;;) {
          const std::function<void(int)>* job = nullptr;
          int t = -1;
          {
            std::unique_lock<std::mutex> l(mu_);
            cv_.wait(l, [&] { return gen_ != seen && next_ <= want_; });
            t = next_++;
            if (next_ > want_) seen = gen_;                   // last index of this generation handed out
            job = job_;
          }
          (*job)(t);
          bool last;
          { std::lock_guard<std::mutex> l(mu_); last = --pending_ == 0; }
          if (last) done_cv_.notify_one();
        }
      });
      threads_.back().detach();
    }
  }
  std::mutex run_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> threads_;
  const std::function<void(int)>* job_ = nullptr;
  uint64_t gen_ = 0;
  int want_ = 0, next_ = 1, pending_ = 0;
};
// One pool per process: a fork()ed child (Python multiprocessing's "fork" start method) inherits the object but not its
// threads, and its condition variables still remember the parent's waiters — the child gets a fresh pool, the old one is leaked.
HostPool& host_pool() {
  static HostPool* p = nullptr;
  static pid_t owner = 0;
  if (!p || owner != getpid()) { p = new HostPool; owner = getpid(); }
  return *p;
}
void host_parallel(int n, const std::function<void(int)>& f) {
  if (n <= 1) { if (n == 1) f(0); return; }
  host_pool().run(n, f);
}

// Symbolic pass of groups [gbeg, gend) into a PARTIAL result (pools start at 0); `sh` holds the shared read-only parts
// (operator in CSR, value-array offsets).  Thread-safe: all scratch is local (group_index_sets uses thread_local scratch).
// Layout of the final pools, known before any of them is written (pass A + prefix sums): per group its index sets and the
// bases of its slices; the per-thread partial results that cannot be placed in advance (weight records, reductions).
struct GroupPlace {
  int32_t n = 0, m = 0;
  int64_t idx_src = 0;      // where pass A left s_x, s_u (thread-local buffer of the owning thread)
  int64_t idx_base = 0;     // idx_pool offset of s_x (s_u follows)
  int64_t md_base = 0;      // mask_pool / dest_pool offset of the group's first column
  int64_t sub_base = 0;     // index of the group's first subproblem
  int64_t cw_base = 0;      // compact tables: cmask offset (words) of the group's first column
};
struct RangePart {
  std::vector<int32_t> idx;              // pass A: s_x, s_u of every group of the range, back to back
  pool_vec<double> w_pool;               // pass B: weight records (spliced afterwards; small)
  int32_t max_n = 0, max_m = 0, max_nnzA = 0, max_nnzB = 0, max_nm = 1, max_nz = 0;
  double flops_alg = 0.0, bytes_alg = 0.0;
  bool irregular = false;                // compact tables: some column of the range is not regular
};

// pass A: index sets of the groups of one range (src/reduction.jl:14)
static int index_sets_range(const Inputs& in, const std::vector<int64_t>& gptr, const std::vector<int64_t>& gcols, int64_t g0,
                            int64_t g1, int64_t gbeg, std::vector<GroupPlace>& place, RangePart& part, std::string& msg) {
  GroupSets gs;
  for (int64_t g = g0; g < g1; ++g) {
    int rc = group_index_sets(in, gcols.data() + gptr[g], gptr[g + 1] - gptr[g], gs, msg);
    if (rc) return rc;
    GroupPlace& gp = place[g - gbeg];
    gp.n = (int32_t)gs.sx.size(); gp.m = (int32_t)gs.su.size();
    gp.idx_src = (int64_t)part.idx.size();
    part.idx.insert(part.idx.end(), gs.sx.begin(), gs.sx.end());
    part.idx.insert(part.idx.end(), gs.su.begin(), gs.su.end());
  }
  return 0;
}

// pass B: everything of the groups [g0, g1) except the packed numbering, written straight into the final pools of S
static int fill_range(const Inputs& in, const std::vector<int64_t>& gptr, const std::vector<int64_t>& gcols, int64_t gbeg,
                      int64_t g0, int64_t g1, const std::vector<GroupPlace>& place, bool def_w, Symbolic& S, RangePart& part,
                      std::vector<int32_t>& nfree_of, std::string& msg) {
  const Symbolic& sh = S;
  const sls_dims& d = *in.dims;
  const int base = d.index_base;
  const int64_t Nx = d.Nx, Nu = d.Nu, T = d.T;
  std::vector<int32_t> map_x(Nx, -1), map_u(Nu, -1);
  std::vector<int32_t> zcount;            // per z-row nonzero counter (diagonality check)
  std::vector<double> d11col;
  if (!def_w) { zcount.assign(Nx + Nu, 0); d11col.assign(Nx + Nu, 0.0); }

  GroupSets gs;
  for (int64_t g = g0; g < g1; ++g) {
    const int64_t* cols = gcols.data() + gptr[g];
    const int64_t nc = gptr[g + 1] - gptr[g];
    const GroupPlace& gp = place[g - gbeg];
    const int32_t n = gp.n, m = gp.m;
    gs.sx.assign(part.idx.begin() + gp.idx_src, part.idx.begin() + gp.idx_src + n);
    gs.su.assign(part.idx.begin() + gp.idx_src + n, part.idx.begin() + gp.idx_src + n + m);
    for (int32_t i = 0; i < n; ++i) map_x[gs.sx[i]] = i;
    for (int32_t i = 0; i < m; ++i) map_u[gs.su[i]] = i;

    // local nnz of Ã, B̃2 (what the kernel's gather will find)
    int32_t nnzA = 0, nnzB = 0;
    for (int32_t i = 0; i < n; ++i) {
      const int32_t r = gs.sx[i];
      for (int32_t e = sh.A_csr.ptr[r]; e < sh.A_csr.ptr[r + 1]; ++e)
        if (sh.A_csr.val[e] != 0.0 && map_x[sh.A_csr.idx[e]] >= 0) ++nnzA;
      for (int32_t e = sh.B_csr.ptr[r]; e < sh.B_csr.ptr[r + 1]; ++e)
        if (sh.B_csr.val[e] != 0.0 && map_u[sh.B_csr.idx[e]] >= 0) ++nnzB;
    }

    // B̃1 = B1[c_j ∩ s_x, c_j] (src/synthesis.jl:42,50).  Diagonal: the group's QP separates by column.  Otherwise the columns are
    // COUPLED through the cost: Σ_t ‖[C̃1 D̃12] Z_t R + D̃11‖²_F with Z_t = [z_t of column 1 … column nc] and R = B1[c_j, c_j], i.e.
    // Hessian (R Rᵀ) ⊗ G over the group's columns, G = [C̃1 D̃12]ᵀ[C̃1 D̃12].  The group then becomes ONE work item of the tile
    // kernel's CG build (has_w = 3 on its first column, 4 on the others): per column the diagonal-weight record with
    // M_cc·diag(G), plus one group record with M = R Rᵀ and W.
    std::vector<double> bdiag(nc, 0.0);
    std::vector<double> Rm;                          // nc × nc, row-major, only for a coupled group
    bool coupled = false;
    for (int64_t q = 0; q < nc; ++q) {
      const int64_t c = cols[q];
      for (int64_t k = in.P->B1->colptr[c] - base; k < in.P->B1->colptr[c + 1] - base; ++k) {
        const int64_t r = in.P->B1->rowval[k] - base;
        const double v = in.P->B1->nzval ? in.P->B1->nzval[k] : 1.0;
        if (r == c) bdiag[q] = v;
        else if (v != 0.0 && nc > 1 && map_x[r] >= 0 && std::binary_search(cols, cols + nc, r)) coupled = true;
      }
    }
    if (coupled && nc > 64) {
      msg = "coupled column group of more than 64 columns: not supported by this build";
      for (int32_t i = 0; i < n; ++i) map_x[gs.sx[i]] = -1;
      for (int32_t i = 0; i < m; ++i) map_u[gs.su[i]] = -1;
      return SLS_EUNSUPPORTED;
    }
    if (coupled) {
      Rm.assign((size_t)nc * nc, 0.0);
      for (int64_t q = 0; q < nc; ++q) {
        const int64_t c = cols[q];
        if (map_x[c] < 0) {
          // the reference multiplies Φ̃ (… × nc) by B̃1 (|c_j ∩ s_x| × nc): a DimensionMismatch there
          msg = "coupled column group with a column outside its own index set s_x: the reference's Φ̃·B̃1 is not defined";
          for (int32_t i = 0; i < n; ++i) map_x[gs.sx[i]] = -1;
          for (int32_t i = 0; i < m; ++i) map_u[gs.su[i]] = -1;
          return SLS_EUNSUPPORTED;
        }
        for (int64_t k = in.P->B1->colptr[c] - base; k < in.P->B1->colptr[c + 1] - base; ++k) {
          const int64_t r = in.P->B1->rowval[k] - base;
          const int64_t* pos = std::lower_bound(cols, cols + nc, r);
          if (pos != cols + nc && *pos == r) Rm[(size_t)(pos - cols) * nc + q] = in.P->B1->nzval ? in.P->B1->nzval[k] : 1.0;   // R[row r, column q]
        }
      }
    }
    auto Mof = [&](int64_t a, int64_t b2) { double sacc = 0.0; for (int64_t w2 = 0; w2 < nc; ++w2) sacc += Rm[(size_t)a * nc + w2] * Rm[(size_t)b2 * nc + w2]; return sacc; };

    // cost weights  W = [C1 D12][[s_x; Nx+s_u], (s_x, s_u)]  (src/synthesis.jl:50, GeneralizedPlant.jl:266-285).
    // Diagonal WᵀW (every Plant(A,B1,B2), every diagonally weighted LQR): H = b²·diag(WᵀW) goes to the kernels as is.
    // Otherwise the Hessian b²·WᵀW is dense on (s_x,s_u): the column record also carries b·W itself (CSR by z-row and CSC by
    // variable); the kernel solves with diag(WᵀW) as constraint preconditioner and conjugate gradients on top (has_w = 2).
    std::vector<double> hdx, hdu;
    bool nondiag = false;
    struct WEnt { int64_t z; int32_t col; double v; };
    std::vector<WEnt> went;                         // selected entries of W, local variable numbering (x: i, u: n + i)
    std::vector<int64_t> zrows;                     // z-rows with an entry, ascending
    const bool reg = in.reg_x || in.reg_u;
    auto rxv = [&](int32_t i) { return in.reg_x ? in.reg_x[gs.sx[i]] : 0.0; };
    auto ruv = [&](int32_t i) { return in.reg_u ? in.reg_u[gs.su[i]] : 0.0; };
    const bool use_w = !def_w || coupled || reg;
    if (use_w) {
      if (zcount.empty()) { zcount.assign(Nx + Nu, 0); d11col.assign(Nx + Nu, 0.0); }
      hdx.assign(n, 0.0); hdu.assign(m, 0.0);
      auto zsel = [&](int64_t z) -> bool { return z < Nx ? map_x[z] >= 0 : map_u[z - Nx] >= 0; };
      std::vector<int64_t> touched;
      auto scan = [&](const sls_csc_f64* M, const std::vector<int32_t>& sel, std::vector<double>& hd, int32_t col0) {
        for (size_t i = 0; i < sel.size(); ++i) {
          const int64_t c = sel[i];
          if (!M) {                                   // NULL C1 / D12: [C1 D12] = I (GeneralizedPlant.jl:105-110)
            const int64_t z = c + (col0 ? Nx : 0);
            hd[i] += 1.0;
            went.push_back({z, col0 + (int32_t)i, 1.0});
            if (zcount[z]++ == 0) touched.push_back(z); else nondiag = true;
            continue;
          }
          for (int64_t k = M->colptr[c] - base; k < M->colptr[c + 1] - base; ++k) {
            const int64_t z = M->rowval[k] - base;
            const double v = M->nzval ? M->nzval[k] : 1.0;
            if (v == 0.0 || !zsel(z)) continue;
            hd[i] += v * v;
            went.push_back({z, col0 + (int32_t)i, v});
            if (zcount[z]++ == 0) touched.push_back(z); else nondiag = true;
          }
        }
      };
      scan(in.P->C1, gs.sx, hdx, 0);
      scan(in.P->D12, gs.su, hdu, n);
      for (int64_t z : touched) zcount[z] = 0;
      if (nondiag || coupled) { zrows = touched; std::sort(zrows.begin(), zrows.end()); }
    }

    const int64_t off_sx = gp.idx_base, off_su = gp.idx_base + n;
    std::copy(gs.sx.begin(), gs.sx.end(), S.idx_pool.begin() + off_sx);
    std::copy(gs.su.begin(), gs.su.end(), S.idx_pool.begin() + off_su);
    const int32_t nm = n + m;

    for (int64_t q = 0; q < nc; ++q) {
      const int64_t c = cols[q];
      SubDesc sd{};
      sd.n = n; sd.m = m; sd.pos = map_x[c]; sd.nnzA = nnzA; sd.nnzB = nnzB;
      sd.cls = wave_class_of(n, m);
      part.max_nm = std::max(part.max_nm, nm);
      sd.off_sx = off_sx; sd.off_su = off_su;
      sd.off_mask = sd.off_dest = gp.md_base + q * T * nm;
      sd.out_index = gp.sub_base + q;
      int64_t nfree = 0;
      if (S.compact) {
        // bit masks + first destinations; regularity (see Symbolic::compact) checked on the way
        const int32_t wps = (nm + 63) / 64;
        const int64_t cw = gp.cw_base + q * T * wps;
        uint64_t* bits = S.cmask.data() + cw;
        int32_t* cb = S.cbase.data() + 2 * T * sd.out_index;
        S.coff[sd.out_index] = cw;
        std::fill(bits, bits + (size_t)T * wps, (uint64_t)0);
        bool regular = true;
        for (int64_t t = 0; t < T; ++t) {
          uint64_t* bt = bits + t * wps;
          auto part_of = [&](const sls_csc_bool* sm, const std::vector<int32_t>& map, int32_t shift, int64_t off) -> int32_t {
            const int64_t k0 = sm->colptr[c] - base, k1 = sm->colptr[c + 1] - base;
            int32_t last = -1;
            for (int64_t k = k0; k < k1; ++k) {
              const int32_t loc = map[sm->rowval[k] - base];
              if ((sm->nzval && sm->nzval[k] != 1) || loc <= last) { regular = false; continue; }   // loc < 0 included
              last = loc;
              bt[(shift + loc) >> 6] |= (uint64_t)1 << ((shift + loc) & 63);
            }
            nfree += k1 - k0;
            return (int32_t)(off + k0);
          };
          cb[2 * t] = part_of(&in.Sx[t], map_x, 0, sh.off_x[t]);
          cb[2 * t + 1] = part_of(&in.Su[t], map_u, n, sh.off_u[t]);
        }
        if (!regular) part.irregular = true;
      } else {
      uint8_t* mk = S.mask_pool.data() + sd.off_mask;
      int32_t* ds = S.dest_pool.data() + sd.off_dest;
      std::fill(mk, mk + (size_t)T * nm, (uint8_t)0);
      std::fill(ds, ds + (size_t)T * nm, -1);
      for (int64_t t = 0; t < T; ++t) {
        const sls_csc_bool* sx = &in.Sx[t];
        for (int64_t k = sx->colptr[c] - base; k < sx->colptr[c + 1] - base; ++k) {
          if (sx->nzval && sx->nzval[k] != 1) continue;                       // `.≠ 1` ⇒ fixed to 0
          const int32_t loc = map_x[sx->rowval[k] - base];
          if (loc < 0) continue;                                              // row outside s_x: no variable
          mk[t * nm + loc] = 1; ds[t * nm + loc] = (int32_t)(sh.off_x[t] + k);
        }
        const sls_csc_bool* su = &in.Su[t];
        for (int64_t k = su->colptr[c] - base; k < su->colptr[c + 1] - base; ++k) {
          if (su->nzval && su->nzval[k] != 1) continue;
          const int32_t loc = map_u[su->rowval[k] - base];
          if (loc < 0) continue;
          mk[t * nm + n + loc] = 1; ds[t * nm + n + loc] = (int32_t)(sh.off_u[t] + k);
        }
        for (int32_t i = 0; i < nm; ++i) nfree += mk[t * nm + i];
      }
      }
      nfree_of[sd.out_index] = (int32_t)nfree;
      // weights record
      sd.has_w = 0; sd.off_w = 0;
      bool bad_w = false;
      if (use_w) {
        const double b = coupled ? 1.0 : bdiag[q];           // coupled: W is stored unscaled, the scale is M = R Rᵀ
        const double b2 = coupled ? Mof(q, q) : b * b;
        if (b2 != 0.0 || (reg && !coupled)) {
          // g = b·Wᵀ d11[:,c]   (coupled: Wᵀ Σ_w R[q,w]·d11[:,c_w])
          std::vector<double> gxv(n, 0.0), guv(m, 0.0);
          bool any_d11 = false;
          if (in.P->D11) {
            std::vector<int64_t> touched;
            for (int64_t w2 = 0; w2 < (coupled ? nc : 1); ++w2) {
              const int64_t cw = coupled ? cols[w2] : c;
              const double rw = coupled ? Rm[(size_t)q * nc + w2] : 1.0;
              if (rw == 0.0) continue;
              for (int64_t k = in.P->D11->colptr[cw] - base; k < in.P->D11->colptr[cw + 1] - base; ++k) {
                const int64_t z = in.P->D11->rowval[k] - base;
                const double v = in.P->D11->nzval ? in.P->D11->nzval[k] : 1.0;
                const bool sel = z < Nx ? map_x[z] >= 0 : map_u[z - Nx] >= 0;
                if (v != 0.0 && sel) { if (d11col[z] == 0.0) touched.push_back(z); d11col[z] += rw * v; any_d11 = true; }
              }
            }
            if (any_d11) {
              auto acc = [&](const sls_csc_f64* M, const std::vector<int32_t>& sel, std::vector<double>& gv, int64_t zoff) {
                for (size_t i = 0; i < sel.size(); ++i) {
                  const int64_t cc = sel[i];
                  if (!M) { gv[i] += b * d11col[cc + zoff]; continue; }
                  for (int64_t k = M->colptr[cc] - base; k < M->colptr[cc + 1] - base; ++k) {
                    const int64_t z = M->rowval[k] - base;
                    const double v = M->nzval ? M->nzval[k] : 1.0;
                    gv[i] += b * v * d11col[z];
                  }
                }
              };
              acc(in.P->C1, gs.sx, gxv, 0);
              acc(in.P->D12, gs.su, guv, Nx);
              for (int64_t z : touched) d11col[z] = 0.0;
            }
          }
          bool ident = !any_d11 && !coupled && !reg;
          for (int32_t i = 0; i < n && ident; ++i) if (hdx[i] != hdx[0]) ident = false;
          for (int32_t i = 0; i < m && ident; ++i) if (hdu[i] != (n ? hdx[0] : hdu[0])) ident = false;
          for (int32_t i = 0; i < n; ++i) if (!(b2 * hdx[i] + rxv(i) > 0.0)) { msg = "zero cost weight on a state variable (singular Hessian): not supported"; bad_w = true; }
          for (int32_t i = 0; i < m; ++i) if (!(b2 * hdu[i] + ruv(i) > 0.0)) { msg = "zero cost weight on an input variable (singular Hessian): not supported"; bad_w = true; }
          if ((!ident || nondiag) && !bad_w) {
            sd.has_w = coupled ? (q == 0 ? 3 : 4) : (nondiag ? 2 : 1);
            sd.off_w = (int64_t)part.w_pool.size();        // local; rebased when the records are spliced
            if (coupled && q == 0) sd.pad_ = (int32_t)nc;
            for (int32_t i = 0; i < n; ++i) part.w_pool.push_back(1.0 / (b2 * hdx[i] + rxv(i)));
            for (int32_t i = 0; i < m; ++i) part.w_pool.push_back(1.0 / (b2 * hdu[i] + ruv(i)));
            for (int32_t i = 0; i < n; ++i) part.w_pool.push_back(gxv[i]);
            for (int32_t i = 0; i < m; ++i) part.w_pool.push_back(guv[i]);
            if (coupled && q == 0) {
              part.w_pool.push_back((double)nc);
              for (int64_t a = 0; a < nc; ++a) for (int64_t b3 = 0; b3 < nc; ++b3) part.w_pool.push_back(Mof(a, b3));
            }
            if ((nondiag && !coupled) || (coupled && q == 0)) {
              // b·W on the selected rows/columns: [nz, nnz] · CSR by z-row (ptr, idx, val) · CSC by variable (ptr, idx, val);
              // integers are stored as doubles (exact below 2^53) so that the record stays in the one weight pool
              const int32_t nz = (int32_t)zrows.size(), nnzw = (int32_t)went.size();
              std::vector<int32_t> rp(nz + 1, 0), cp(nm + 1, 0), ri(nnzw), ci(nnzw);
              std::vector<double> rv(nnzw), cv(nnzw);
              auto zloc = [&](int64_t z) { return (int32_t)(std::lower_bound(zrows.begin(), zrows.end(), z) - zrows.begin()); };
              for (const WEnt& e : went) { rp[zloc(e.z) + 1]++; cp[e.col + 1]++; }
              for (int32_t i = 0; i < nz; ++i) rp[i + 1] += rp[i];
              for (int32_t i = 0; i < nm; ++i) cp[i + 1] += cp[i];
              std::vector<int32_t> rw(rp.begin(), rp.end() - 1), cw(cp.begin(), cp.end() - 1);
              for (const WEnt& e : went) {
                const int32_t zl = zloc(e.z);
                ri[rw[zl]] = e.col; rv[rw[zl]] = b * e.v; ++rw[zl];
                ci[cw[e.col]] = zl; cv[cw[e.col]] = b * e.v; ++cw[e.col];
              }
              part.w_pool.push_back((double)nz); part.w_pool.push_back((double)nnzw);
              for (int32_t v : rp) part.w_pool.push_back((double)v);
              for (int32_t v : ri) part.w_pool.push_back((double)v);
              for (double v : rv) part.w_pool.push_back(v);
              for (int32_t v : cp) part.w_pool.push_back((double)v);
              for (int32_t v : ci) part.w_pool.push_back((double)v);
              for (double v : cv) part.w_pool.push_back(v);
              for (int32_t i = 0; i < n; ++i) part.w_pool.push_back(rxv(i));        // ridge term per variable (zeros when none)
              for (int32_t i = 0; i < m; ++i) part.w_pool.push_back(ruv(i));
              part.max_nz = std::max(part.max_nz, nz);
            }
          }
        } else if (coupled) {
          msg = "coupled column group with a zero row of B1[c_j,c_j]: not supported";
          bad_w = true;
        }
        // b == 0: the cost is constant in Φ; return the minimum-norm feasible point (identity weights)
      }
      if (bad_w) {
        for (int32_t i = 0; i < n; ++i) map_x[gs.sx[i]] = -1;
        for (int32_t i = 0; i < m; ++i) map_u[gs.su[i]] = -1;
        return SLS_EUNSUPPORTED;
      }
      S.subs[sd.out_index] = sd;
      S.sub_col[sd.out_index] = (int32_t)c;
      // algorithmic work, SURVEY §8d
      const double dn = n, dT = (double)T, dnf = (double)nfree;
      part.flops_alg += dn * dn * dnf + (7.0 / 3.0) * (dT + 1) * dn * dn * dn + 6.0 * (dT + 1) * dn * dn + 2.0 * dn * dnf;
      part.bytes_alg += 12.0 * (nnzA + nnzB) + 4.0 * (n + m) + dT * (n + m) / 8.0 + 8.0 * dnf;
    }
    part.max_n = std::max(part.max_n, n); part.max_m = std::max(part.max_m, m);
    part.max_nnzA = std::max(part.max_nnzA, nnzA); part.max_nnzB = std::max(part.max_nnzB, nnzB);
    for (int32_t i = 0; i < n; ++i) map_x[gs.sx[i]] = -1;
    for (int32_t i = 0; i < m; ++i) map_u[gs.su[i]] = -1;
  }

  return 0;
}

int build_symbolic(const Inputs& in, int64_t gbeg, int64_t gend, Symbolic& S, std::string& msg) {
  const sls_dims& d = *in.dims;
  const int base = d.index_base;
  const int64_t Nx = d.Nx, Nu = d.Nu, T = d.T;
  S.Nx = Nx; S.Nu = Nu; S.T = T;

  std::vector<int64_t> gptr, gcols;
  normalise_groups(in, gptr, gcols);
  const int64_t ng = (int64_t)gptr.size() - 1;
  if (gbeg < 0 || gend > ng || gbeg > gend) { msg = "group range out of bounds"; return SLS_EINVAL; }
  S.n_total_subproblems = gptr[ng];
  S.first_sub_index = gptr[gbeg];

  // value-array offsets
  S.off_x.assign(T + 1, 0); S.off_u.assign(T + 1, 0);
  for (int64_t t = 0; t < T; ++t) S.off_x[t + 1] = S.off_x[t] + (in.Sx[t].colptr[Nx] - base);
  S.off_u[0] = S.off_x[T];
  for (int64_t t = 0; t < T; ++t) S.off_u[t + 1] = S.off_u[t] + (in.Su[t].colptr[Nx] - base);
  S.n_values = S.off_u[T];
  if (S.n_values > 0x7fffffffLL) { msg = "more than 2^31 values in Φ: not supported by this build"; return SLS_EUNSUPPORTED; }

  csc_to_csr(in.P->A, base, S.A_csr);
  csc_as_csr_of_transpose(in.P->A, base, S.At_csr);
  csc_to_csr(in.P->B2, base, S.B_csr);
  csc_as_csr_of_transpose(in.P->B2, base, S.Bt_csr);
  auto longest = [](const HostCsr& M) {
    int32_t mx = 1;
    for (int64_t r = 0; r < M.nrows; ++r) mx = std::max(mx, M.ptr[r + 1] - M.ptr[r]);
    return mx;
  };
  S.max_row_A = longest(S.A_csr); S.max_row_At = longest(S.At_csr);
  S.max_row_B = longest(S.B_csr); S.max_row_Bt = longest(S.Bt_csr);

  const bool dbg_t = sls_knob("SLS_DEBUG_TIMING") != nullptr;
  auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tdbg = clk();
  auto tick = [&](const char* what) { if (dbg_t) { const double n = clk(); std::fprintf(stderr, "[sls symbolic] %-24s %8.3f ms\n", what, 1e3 * (n - tdbg)); tdbg = n; } };
  const bool def_w = weights_are_default(in);
  tick("operator CSR + defaults");

  // ---- the per-group work is independent and every pool slice has a size known from (ñx, ñu, T): three passes on host
  // threads, each writing straight into the final pools (no per-thread copies to splice, every page first touched by the
  // thread that fills it):  A index sets → prefix sums → B masks / destinations / descriptors → prefix of free counts →
  // C packed numbering.
  const int64_t ngr = gend - gbeg;
  unsigned hw = std::thread::hardware_concurrency();
  int nthreads = (int)std::min<int64_t>(hw ? hw : 1, 16);
  if (const char* e = sls_knob("SLS_SYMBOLIC_THREADS")) nthreads = std::max(1, std::atoi(e));
  nthreads = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, ngr / 64));       // ≥ 64 groups per thread
  std::vector<GroupPlace> place((size_t)ngr);
  std::vector<RangePart> parts(nthreads);
  std::vector<std::string> msgs(nthreads);
  std::vector<int> rcs(nthreads, 0);
  auto range_of = [&](int t, int64_t& g0, int64_t& g1) { g0 = gbeg + ngr * t / nthreads; g1 = gbeg + ngr * (t + 1) / nthreads; };
  auto run = [&](auto&& f) {
    if (nthreads == 1) { f(0); return; }
    host_pool().run(nthreads, [&f](int t) { f(t); });
  };
  auto first_error = [&]() -> int { for (int t = 0; t < nthreads; ++t) if (rcs[t]) { msg = msgs[t]; return rcs[t]; } return 0; };
  run([&](int t) { int64_t g0, g1; range_of(t, g0, g1); rcs[t] = index_sets_range(in, gptr, gcols, g0, g1, gbeg, place, parts[t], msgs[t]); });
  if (int rc = first_error()) return rc;
  tick("A: index sets");
  int64_t idx_tot = 0, md_tot = 0, sub_tot = 0, cw_tot = 0;
  for (int64_t g = 0; g < ngr; ++g) {
    GroupPlace& gp = place[g];
    const int64_t nc = gptr[gbeg + g + 1] - gptr[gbeg + g];
    gp.idx_base = idx_tot; gp.md_base = md_tot; gp.sub_base = sub_tot; gp.cw_base = cw_tot;
    idx_tot += gp.n + gp.m; md_tot += nc * T * (gp.n + gp.m); sub_tot += nc; cw_tot += nc * T * ((gp.n + gp.m + 63) / 64);
  }
  S.md_total = md_tot;
  if (S.want_packed) S.compact = false;
  S.idx_pool.resize(idx_tot);
  S.subs.resize(sub_tot); S.sub_col.resize(sub_tot);
  std::vector<int32_t> nfree_of((size_t)sub_tot, 0);
  auto pass_b = [&]() -> int {
    if (S.compact) { S.cmask.resize(cw_tot); S.cbase.resize(2 * T * sub_tot); S.coff.resize(sub_tot); }
    else {
      S.mask_pool.resize(md_tot); S.dest_pool.resize(md_tot);
      if (S.want_packed) S.pdest_pool.resize(md_tot);
    }
    run([&](int t) { int64_t g0, g1; range_of(t, g0, g1); rcs[t] = fill_range(in, gptr, gcols, gbeg, g0, g1, place, def_w, S, parts[t], nfree_of, msgs[t]); });
    return first_error();
  };
  if (int rc = pass_b()) return rc;
  if (S.compact) {
    bool irregular = false;
    for (int t = 0; t < nthreads; ++t) irregular = irregular || parts[t].irregular;
    if (irregular) {
      // some column stores a false, an out-of-set or an unsorted mask entry: explicit tables for the whole plan
      S.compact = false;
      pool_vec<uint64_t>().swap(S.cmask); pool_vec<int32_t>().swap(S.cbase); pool_vec<int64_t>().swap(S.coff);
      for (int t = 0; t < nthreads; ++t) {
        RangePart fresh; fresh.idx.swap(parts[t].idx);
        parts[t] = std::move(fresh);
      }
      if (int rc = pass_b()) return rc;
    }
  }
  tick(S.compact ? "B: compact masks + bases" : "B: masks + destinations");
  // reductions, weight records (rebased), packed bases
  int64_t w_tot = 0;
  std::vector<int64_t> w_base(nthreads + 1, 0);
  for (int t = 0; t < nthreads; ++t) {
    const RangePart& Pt = parts[t];
    w_base[t + 1] = w_base[t] + (int64_t)Pt.w_pool.size();
    S.max_n = std::max(S.max_n, Pt.max_n); S.max_m = std::max(S.max_m, Pt.max_m);
    S.max_nnzA = std::max(S.max_nnzA, Pt.max_nnzA); S.max_nnzB = std::max(S.max_nnzB, Pt.max_nnzB);
    S.max_nm = std::max(S.max_nm, Pt.max_nm);
    S.max_wz = std::max(S.max_wz, Pt.max_nz);
    S.flops_alg += Pt.flops_alg; S.bytes_alg += Pt.bytes_alg;
  }
  w_tot = w_base[nthreads];
  S.w_pool.resize(w_tot);
  std::vector<int64_t> pk_base((size_t)sub_tot + 1, 0);
  for (int64_t q = 0; q < sub_tot; ++q) pk_base[q + 1] = pk_base[q] + nfree_of[q];
  S.n_packed = pk_base[sub_tot];
  S.pk_base = pk_base;
  const bool rebased = !S.pk_override.empty();
  if (rebased && (int64_t)S.pk_override.size() != sub_tot) { msg = "packed-base override: one entry per subproblem expected"; return SLS_EINVAL; }
  if (S.want_packed && !rebased) S.packed_to_final.resize(S.n_packed);
  run([&](int t) {
    int64_t g0, g1; range_of(t, g0, g1);
    std::copy(parts[t].w_pool.begin(), parts[t].w_pool.end(), S.w_pool.begin() + w_base[t]);
    const int64_t q0 = g0 < gend ? place[g0 - gbeg].sub_base : sub_tot, q1 = g1 < gend ? place[g1 - gbeg].sub_base : sub_tot;
    for (int64_t q = q0; q < q1; ++q) {
      SubDesc& sd = S.subs[q];
      if (sd.has_w) sd.off_w += w_base[t];
      if (!S.want_packed) continue;
      const int64_t len = T * (sd.n + sd.m);
      const uint8_t* mk = S.mask_pool.data() + sd.off_mask;
      const int32_t* ds = S.dest_pool.data() + sd.off_dest;
      int32_t* pds = S.pdest_pool.data() + sd.off_dest;
      int64_t pk = rebased ? S.pk_override[q] : pk_base[q];
      for (int64_t e = 0; e < len; ++e) {
        if (mk[e]) { pds[e] = (int32_t)pk; if (!rebased) S.packed_to_final[pk] = ds[e]; ++pk; } else pds[e] = -1;
      }
    }
    parts[t] = RangePart();
  });
  tick("C: packed numbering");
  // processing order: descending predicted cost (T+1)·ñx³ (longest first ⇒ short tail)
  S.order.resize(S.subs.size());
  std::iota(S.order.begin(), S.order.end(), 0);
  std::stable_sort(S.order.begin(), S.order.end(), [&](int32_t a, int32_t b) { return S.subs[a].n > S.subs[b].n; });
  return 0;
}

// README.md:62-72 — row-oriented FIR operators of the closed-loop simulation (see FirOperator)
int build_fir_operator(const sls_dims* dims, const sls_csc_f64* A, const sls_csc_f64* B1, const sls_csc_f64* B2,
                       const sls_csc_bool* Sx, const sls_csc_bool* Su, FirOperator& F, std::string& msg) {
  if (!dims || !A || !B1 || !B2 || !Sx || !Su) { msg = "null argument"; return SLS_EINVAL; }
  const sls_dims& d = *dims;
  const int base = d.index_base;
  if (base != 0 && base != 1) { msg = "index_base must be 0 or 1"; return SLS_EINVAL; }
  if (d.Nx <= 0 || d.Nu < 0 || d.Nw <= 0 || d.T <= 0) { msg = "Nx, Nw, T must be positive"; return SLS_EINVAL; }
  if ((d.T + 1) * d.Nx > 0x7fffffffLL) { msg = "(T+1)·Nx exceeds int32"; return SLS_EUNSUPPORTED; }
  int rc;
  if ((rc = check_csc(A, d.Nx, d.Nx, base, "A", msg))) return rc;
  if ((rc = check_csc(B1, d.Nx, d.Nw, base, "B1", msg))) return rc;
  if ((rc = check_csc(B2, d.Nx, d.Nu, base, "B2", msg))) return rc;
  int64_t nval = 0;
  std::vector<int64_t> off_x(d.T + 1), off_u(d.T + 1);
  for (int64_t t = 0; t < d.T; ++t) {
    if ((rc = check_csc(&Sx[t], d.Nx, d.Nx, base, "Sx[t]", msg))) return rc;
    off_x[t] = nval; nval += Sx[t].colptr[d.Nx] - base;
  }
  off_x[d.T] = nval;
  for (int64_t t = 0; t < d.T; ++t) {
    if ((rc = check_csc(&Su[t], d.Nu, d.Nx, base, "Su[t]", msg))) return rc;
    off_u[t] = nval; nval += Su[t].colptr[d.Nx] - base;
  }
  off_u[d.T] = nval;
  if (nval > 0x7fffffffLL) { msg = "value array exceeds int32 indexing"; return SLS_EUNSUPPORTED; }
  F.Nx = d.Nx; F.Nu = d.Nu; F.Nw = d.Nw; F.T = d.T; F.n_values = nval;
  csc_to_csr(A, base, F.A); csc_to_csr(B1, base, F.B1); csc_to_csr(B2, base, F.B2);
  F.orphan.clear();
  for (int64_t j = 0; j < d.Nu; ++j) if (B2->colptr[j + 1] == B2->colptr[j]) F.orphan.push_back((int32_t)j);
  // counting sort by row: slices in ascending lag, columns ascending inside a slice ⇒ rows come out ordered by (τ, c)
  auto each = [&](const sls_csc_bool* S, int64_t t0, int64_t t1, auto&& f) {
    for (int64_t t = t0; t < t1; ++t)
      for (int64_t c = 0; c < d.Nx; ++c)
        for (int64_t k = S[t].colptr[c] - base; k < S[t].colptr[c + 1] - base; ++k)
          if (!S[t].nzval || S[t].nzval[k]) f(t, c, S[t].rowval[k] - base, k);
  };
  F.beta_ptr.assign(d.Nx + 1, 0); F.u_ptr.assign(d.Nu + 1, 0);
  each(Sx, 1, d.T, [&](int64_t, int64_t, int64_t r, int64_t) { F.beta_ptr[r + 1]++; });
  each(Su, 0, d.T, [&](int64_t, int64_t, int64_t r, int64_t) { F.u_ptr[r + 1]++; });
  for (int64_t r = 0; r < d.Nx; ++r) F.beta_ptr[r + 1] += F.beta_ptr[r];
  F.u_ptr[0] = F.beta_ptr[d.Nx];
  for (int64_t r = 0; r < d.Nu; ++r) F.u_ptr[r + 1] += F.u_ptr[r];
  const int64_t nent = F.u_ptr[d.Nu];
  F.hoff.resize(nent); F.perm.resize(nent);
  std::vector<int32_t> wx(F.beta_ptr.begin(), F.beta_ptr.end() - 1), wu(F.u_ptr.begin(), F.u_ptr.end() - 1);
  each(Sx, 1, d.T, [&](int64_t t, int64_t c, int64_t r, int64_t k) {        // Φx[τ+1], τ = t (0-based slice t)
    const int32_t e = wx[r]++;
    F.hoff[e] = (int32_t)(t * d.Nx - c); F.perm[e] = (int32_t)(off_x[t] + k);
  });
  each(Su, 0, d.T, [&](int64_t t, int64_t c, int64_t r, int64_t k) {        // Φu[τ], τ = t+1
    const int32_t e = wu[r]++;
    F.hoff[e] = (int32_t)((t + 1) * d.Nx - c); F.perm[e] = (int32_t)(off_u[t] + k);
  });
  return 0;
}

}  // namespace sls
