/* examples/solve_readme.c — the drop-in call from plain C (what a Julia/C/Fortran host does through the ABI).
 *
 * Builds the README plant and (d,T)-localization masks of reference README.md:43-54 with 1-BASED indices (Julia's own
 * arrays), calls sls_h2_sf_solve and prints Σ‖Φ‖² (oracle: 893.3262819770).
 *   gcc -std=c99 -Wall -pedantic -Iinclude examples/solve_readme.c -o /tmp/solve_readme \
 *       -Lsystemlevelcontrol.jl_amd -lsls_mi355x -Wl,-rpath,$PWD/systemlevelcontrol.jl_amd -lm
 * Needs an MI355X at run time; compiling and linking it is part of the CPU test-suite (tests/test_host.py).            */
#include <stdio.h>
#include <stdlib.h>
#include "sls_mi355x.h"

enum { NX = 59, NU = 20, T = 29, D = 9 };

int main(void) {
  /* A = I + 0.2 superdiag − 0.2 subdiag (CSC, 1-based) */
  int64_t a_cp[NX + 1], a_rv[3 * NX]; double a_nz[3 * NX];
  int64_t k = 0, c, t;
  for (c = 0; c < NX; ++c) {
    a_cp[c] = k + 1;
    if (c > 0)      { a_rv[k] = c;     a_nz[k++] = 0.2; }    /* row c−1 (0-based) → A[c−1,c] = +0.2 */
    a_rv[k] = c + 1; a_nz[k++] = 1.0;
    if (c < NX - 1) { a_rv[k] = c + 2; a_nz[k++] = -0.2; }   /* A[c+1,c] = −0.2 */
  }
  a_cp[NX] = k + 1;
  /* B1 = I */
  int64_t b1_cp[NX + 1], b1_rv[NX]; double b1_nz[NX];
  for (c = 0; c < NX; ++c) { b1_cp[c] = c + 1; b1_rv[c] = c + 1; b1_nz[c] = 1.0; }
  b1_cp[NX] = NX + 1;
  /* B2 = I[:, {6n+1, 6n+2}] */
  int64_t b2_cp[NU + 1], b2_rv[NU]; double b2_nz[NU];
  for (c = 0; c < NU; ++c) { b2_cp[c] = c + 1; b2_rv[c] = 6 * (c / 2) + (c % 2) + 1; b2_nz[c] = 1.0; }
  b2_cp[NU] = NU + 1;
  sls_csc_f64 A = {NX, NX, a_cp, a_rv, a_nz}, B1 = {NX, NX, b1_cp, b1_rv, b1_nz}, B2 = {NX, NU, b2_cp, b2_rv, b2_nz};
  sls_dims dims = {NX, NU, NX + NU, NX, T, 1, SLS_SOLVE_DEFAULT};
  sls_plant P = {&A, &B1, &B2, NULL, NULL, NULL};

  /* masks through the library's own recipe (two-call protocol) */
  int64_t nnz_x[T], nnz_u[T];
  int rc = sls_localization_masks(&dims, &A, &B2, D, 1.5, nnz_x, nnz_u, NULL, NULL, NULL, NULL);
  if (rc) { fprintf(stderr, "masks: %s\n", sls_last_error(NULL)); return 1; }
  int64_t *cpx[T], *rvx[T], *cpu_[T], *rvu[T];
  sls_csc_bool Sx[T], Su[T];
  double *vx[T], *vu[T];
  for (t = 0; t < T; ++t) {
    cpx[t] = malloc((NX + 1) * sizeof(int64_t)); rvx[t] = malloc((size_t)(nnz_x[t] + 1) * sizeof(int64_t));
    cpu_[t] = malloc((NX + 1) * sizeof(int64_t)); rvu[t] = malloc((size_t)(nnz_u[t] + 1) * sizeof(int64_t));
    vx[t] = calloc((size_t)nnz_x[t] + 1, sizeof(double)); vu[t] = calloc((size_t)nnz_u[t] + 1, sizeof(double));
  }
  rc = sls_localization_masks(&dims, &A, &B2, D, 1.5, nnz_x, nnz_u, cpx, rvx, cpu_, rvu);
  if (rc) { fprintf(stderr, "masks: %s\n", sls_last_error(NULL)); return 1; }
  for (t = 0; t < T; ++t) {
    Sx[t].nrows = NX; Sx[t].ncols = NX; Sx[t].colptr = cpx[t]; Sx[t].rowval = rvx[t]; Sx[t].nzval = NULL;
    Su[t].nrows = NU; Su[t].ncols = NX; Su[t].colptr = cpu_[t]; Su[t].rowval = rvu[t]; Su[t].nzval = NULL;
  }

  int dev = 0;
  sls_ctx* ctx = sls_create(&dev, 1, SLS_CREATE_DEFAULT);
  if (!ctx) { fprintf(stderr, "sls_create: %s\n", sls_last_error(NULL)); return 2; }
  int32_t status[NX];
  sls_stats st;
  rc = sls_h2_sf_solve(ctx, &dims, &P, Sx, Su, 0, NULL, NULL, vx, vu, status, &st);
  if (rc < 0) { fprintf(stderr, "solve: %s\n", sls_last_error(ctx)); return 3; }
  double cost = 0.0;
  for (t = 0; t < T; ++t) {
    for (k = 0; k < nnz_x[t]; ++k) cost += vx[t][k] * vx[t][k];
    for (k = 0; k < nnz_u[t]; ++k) cost += vu[t][k] * vu[t][k];
  }
  printf("columns not solved: %d   sum |Phi|^2 = %.10f   max residual %.1e   solve %.3f ms\n", rc, cost, st.max_residual,
         1e3 * st.t_solve_s);
  sls_destroy(ctx);
  return (rc == 0 && cost > 893.3262819 && cost < 893.3262821) ? 0 : 4;
}
