/* sls_oracle_c.c — plain-C CPU restatement of one column of the H2 SLS solve.
 *
 * TEST INFRASTRUCTURE ONLY (checker + the timed `cpu_baseline` of bench.py).  Nothing in the
 * product package links or loads this file.  PARITY UNPINNED for Φ (the reference has no golden
 * vector for this path); this file is itself validated against the NumPy oracle
 * (oracle/sls_oracle.py, dense SVD) in tests/test_oracle.py.
 *
 * What it restates (reference src/synthesis.jl:46-62, the QP handed to JuMP/Ipopt), for one column
 * with dense reduced blocks Ã (n×n), B̃2 (n×m) and masks m_x[t], m_u[t]  (src/synthesis.jl:57-60):
 *     min ½ zᵀHz + gᵀz   s.t.  x_0 = e_pos,  x_{t+1} = Ãx_t + B̃u_t (t<T−1),  0 = Ãx_{T−1} + B̃u_{T−1}
 * (src/synthesis.jl:53-55), H diagonal (H = I, g = 0 for Plant(A,B1,B2): GeneralizedPlant.jl:105-110).
 *
 * Algorithm: the canonical dense block-tridiagonal direct method on which SURVEY §8d defines
 * F_alg — Schur complement S = E H⁻¹ Eᵀ (blocks D_k, L_k), block Cholesky (POTRF / TRSM / SYRK per
 * block row), two block substitutions, z = H⁻¹(Eᵀλ − g); S is singular on every README column
 * (dependent / empty constraint rows), so S + δI is factored and the multiplier iteration
 * λ ← λ + (S+δI)⁻¹(f − E z(λ)) is run to ‖f − Ez‖∞ ≤ tol.  Deliberately NOT the GPU kernels'
 * formulation (those use explicit inverse pivot blocks, sparse Ã and no triangular solves).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int n, m, T, pos;
  const double* A;      /* n×n row-major */
  const double* B;      /* n×m row-major */
  const uint8_t* mask;  /* T×(n+m) */
  const double* hinv;   /* n+m or NULL */
  const double* g;      /* n+m or NULL */
} col_t;

static int chol_lower(double* M, int n) { /* in place, row-major, lower */
  for (int j = 0; j < n; ++j) {
    double d = M[j * n + j];
    for (int k = 0; k < j; ++k) d -= M[j * n + k] * M[j * n + k];
    if (d <= 0.0) return -1;
    d = sqrt(d);
    M[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = M[i * n + j];
      for (int k = 0; k < j; ++k) s -= M[i * n + k] * M[j * n + k];
      M[i * n + j] = s / d;
    }
  }
  return 0;
}
static void fsub(const double* C, int n, double* v) { /* C y = v */
  for (int i = 0; i < n; ++i) {
    double s = v[i];
    for (int k = 0; k < i; ++k) s -= C[i * n + k] * v[k];
    v[i] = s / C[i * n + i];
  }
}
static void bsub(const double* C, int n, double* v) { /* Cᵀ y = v */
  for (int i = n - 1; i >= 0; --i) {
    double s = v[i];
    for (int k = i + 1; k < n; ++k) s -= C[k * n + i] * v[k];
    v[i] = s / C[i * n + i];
  }
}

/* x_out: T×n, u_out: T×m.  returns status 0 ok / 1 infeasible / 2 not converged / -1 numerical failure */
int sls_oracle_solve_column(int n, int m, int T, int pos, const double* A, const double* B, const uint8_t* mask,
                            const double* hinv, const double* g, double delta_rel, double tol, int max_iters,
                            double* x_out, double* u_out, double* resid_out, int* iters_out) {
  const int nm = n + m;
  const size_t nn = (size_t)n * n;
  double* Cf = (double*)malloc(sizeof(double) * nn * (T + 1));  /* Cholesky factors C_k            */
  double* Wf = (double*)malloc(sizeof(double) * nn * (T + 1));  /* W_k = L_k C_{k−1}^{−T}          */
  double* L = (double*)malloc(sizeof(double) * nn);
  double* lam = (double*)calloc((size_t)(T + 1) * n, sizeof(double));
  double* y = (double*)malloc(sizeof(double) * (size_t)(T + 1) * n);
  double* r = (double*)malloc(sizeof(double) * (size_t)(T + 1) * n);
  double* wx = (double*)malloc(sizeof(double) * (size_t)T * n);
  double* wu = (double*)malloc(sizeof(double) * (size_t)T * (m > 0 ? m : 1));
  double* tmp = (double*)malloc(sizeof(double) * n);
  int status = 0, iters = 0;
  double resid = 0.0;
  if (!Cf || !Wf || !L || !lam || !y || !r || !wx || !wu || !tmp) { status = -1; goto done; }
  for (int t = 0; t < T; ++t) {
    for (int i = 0; i < n; ++i) wx[t * n + i] = mask[t * nm + i] ? (hinv ? hinv[i] : 1.0) : 0.0;
    for (int j = 0; j < m; ++j) wu[t * m + j] = mask[t * nm + n + j] ? (hinv ? hinv[n + j] : 1.0) : 0.0;
  }
  double sc = 0.0;
  for (int i = 0; i < n; ++i) {
    double s = hinv ? hinv[i] : 1.0;
    for (int q = 0; q < n; ++q) s += A[i * n + q] * A[i * n + q] * (hinv ? hinv[q] : 1.0);
    for (int q = 0; q < m; ++q) s += B[i * m + q] * B[i * m + q] * (hinv ? hinv[n + q] : 1.0);
    if (s > sc) sc = s;
  }
  const double delta = delta_rel * sc;
  /* ---- factor ---- */
  for (int k = 0; k <= T; ++k) {
    double* D = Cf + nn * k;
    memset(D, 0, sizeof(double) * nn);
    for (int i = 0; i < n; ++i) D[i * n + i] = delta + (k <= T - 1 ? wx[k * n + i] : 0.0);
    if (k >= 1) {
      const double* w = wx + (size_t)(k - 1) * n;
      const double* v = wu + (size_t)(k - 1) * m;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
          double s = 0.0;
          for (int q = 0; q < n; ++q) s += A[i * n + q] * w[q] * A[j * n + q];
          for (int q = 0; q < m; ++q) s += B[i * m + q] * v[q] * B[j * m + q];
          D[i * n + j] += s;
        }
      /* L_k = −Ã Wx_{k−1};  W_k = L_k C_{k−1}^{−T}  (row i of W solves C_{k−1} wᵀ = L[i,:]ᵀ) */
      double* W = Wf + nn * k;
      const double* Cp = Cf + nn * (k - 1);
      for (int i = 0; i < n; ++i) {
        for (int q = 0; q < n; ++q) W[i * n + q] = -A[i * n + q] * w[q];
        fsub(Cp, n, W + (size_t)i * n);
      }
      for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
          double s = 0.0;
          for (int q = 0; q < n; ++q) s += W[i * n + q] * W[j * n + q];
          D[i * n + j] -= s;
        }
    }
    if (chol_lower(D, n)) { status = -1; goto done; }
  }
  /* ---- multiplier iteration ---- */
  double prev = 1e300;
  for (int it = 0; it <= max_iters; ++it) {
    /* z(λ) and residual r = f − E z */
    memset(r, 0, sizeof(double) * (size_t)(T + 1) * n);
    if (pos >= 0) r[pos] = 1.0;
    for (int t = 0; t < T; ++t) {
      const double* l0 = lam + (size_t)t * n;
      const double* l1 = lam + (size_t)(t + 1) * n;
      double* xt = x_out + (size_t)t * n;
      double* ut = u_out + (size_t)t * m;
      for (int q = 0; q < n; ++q) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += A[i * n + q] * l1[i];
        xt[q] = wx[t * n + q] * (l0[q] - s - (g ? g[q] : 0.0));
      }
      for (int q = 0; q < m; ++q) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += B[i * m + q] * l1[i];
        ut[q] = wu[t * m + q] * (-s - (g ? g[n + q] : 0.0));
      }
      for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int q = 0; q < n; ++q) s += A[i * n + q] * xt[q];
        for (int q = 0; q < m; ++q) s += B[i * m + q] * ut[q];
        r[(size_t)t * n + i] -= xt[i];
        r[(size_t)(t + 1) * n + i] += s;
      }
    }
    resid = 0.0;
    for (size_t i = 0; i < (size_t)(T + 1) * n; ++i) if (fabs(r[i]) > resid) resid = fabs(r[i]);
    if (resid <= tol) break;
    if (it >= 2 && resid > 0.5 * prev) { status = 1; break; }
    if (it == max_iters) { status = 2; break; }
    prev = resid;
    iters = it + 1;
    /* forward: C_k y_k = r_k − W_k y_{k−1} */
    for (int k = 0; k <= T; ++k) {
      double* yk = y + (size_t)k * n;
      for (int i = 0; i < n; ++i) {
        double s = r[(size_t)k * n + i];
        if (k >= 1) {
          const double* W = Wf + nn * k;
          const double* yp = y + (size_t)(k - 1) * n;
          for (int q = 0; q < n; ++q) s -= W[i * n + q] * yp[q];
        }
        yk[i] = s;
      }
      fsub(Cf + nn * k, n, yk);
    }
    /* backward: C_kᵀ dλ_k = y_k − W_{k+1}ᵀ dλ_{k+1} */
    for (int k = T; k >= 0; --k) {
      double* yk = y + (size_t)k * n;
      if (k < T) {
        const double* W = Wf + nn * (k + 1);
        const double* dn = y + (size_t)(k + 1) * n;
        for (int q = 0; q < n; ++q) {
          double s = 0.0;
          for (int i = 0; i < n; ++i) s += W[i * n + q] * dn[i];
          yk[q] -= s;
        }
      }
      bsub(Cf + nn * k, n, yk);
      for (int i = 0; i < n; ++i) lam[(size_t)k * n + i] += yk[i];
    }
  }
  if (status == 1 && resid <= 1e-9) status = 0;
done:
  if (resid_out) *resid_out = resid;
  if (iters_out) *iters_out = iters;
  free(Cf); free(Wf); free(L); free(lam); free(y); free(r); free(wx); free(wu); free(tmp);
  return status;
}

/* batch over columns with OpenMP (the reference's `julia -p N` analogue on the host cores).
 * Per-column inputs are given as offset tables into flat pools.  Returns the number of threads used. */
int sls_oracle_solve_batch(int ncols, int T, const int32_t* n, const int32_t* m, const int32_t* pos,
                           const int64_t* offA, const int64_t* offB, const int64_t* offM, const int64_t* offX,
                           const int64_t* offU, const double* poolA, const double* poolB, const uint8_t* poolM,
                           double* poolX, double* poolU, double* resid, int32_t* status, int32_t* iters,
                           int nthreads) {
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
  used = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int c = 0; c < ncols; ++c) {
    int it = 0;
    status[c] = sls_oracle_solve_column(n[c], m[c], T, pos[c], poolA + offA[c], poolB + offB[c], poolM + offM[c], NULL,
                                        NULL, 1e-12, 1e-12, 8, poolX + offX[c], poolU + offU[c], &resid[c], &it);
    iters[c] = it;
  }
  return used;
}
