/* sls_mi355x_debug.h — DIAGNOSTIC entry points of libsls_mi355x.so.
 *
 * Not part of the drop-in boundary (include/sls_mi355x.h): nothing here replaces a reference interface.  These are the
 * hooks the test-suite and the measurement scripts under tools/ use to look inside a plan — kernel phase counters, the
 * factor workspace, the per-column tables, the MFMA tile inversion on its own.  They are declared so that the shared
 * library exports nothing a header does not name (tests/test_host.py checks both directions); a host program has no
 * reason to call them and their signatures may change between rounds without an ABI version bump.
 */
#ifndef SLS_MI355X_DEBUG_H
#define SLS_MI355X_DEBUG_H

#include "sls_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Per-subproblem phase cycle counters (s_memtime) of the last execute: out[n_subproblems * 8].  Only when the plan was
 * built with SLS_PHASE_TIMERS=1..5 in the environment (tools/phase_breakdown*.py explain the slots); SLS_EINVAL otherwise. */
int sls_plan_debug_phase_cycles(sls_plan* plan, unsigned long long* out);

/* Copy `count` doubles of the plan's factor workspace (the pivot blocks P_k as the kernels left them), starting at
 * `offset`, to the host (tools/tile_check_factor.py). */
int sls_plan_debug_read_workspace(sls_plan* plan, int64_t offset, int64_t count, double* out);

/* Invert one dense SPD matrix (host, n×n row-major) with the tile kernel's blocked symmetric FP64-MFMA sweep — the unit
 * test of the MFMA operand / result lane maps (tests/test_gpu_tile.py).  mlds != 0: block resident in LDS. */
int sls_debug_tile_invert(sls_ctx* ctx, int dev_slot, int n, const double* h_A, double* h_out, int mlds);

/* The mask / destination tables of a one-device plan as the solve kernels see them, built on the host (host_tables = 1)
 * or expanded on the device from the compact form (0).  Null outputs: only *md_total (the tables' length) is returned.
 * *was_compact reports whether the device expansion actually ran (0 when some column is not regular). */
int sls_debug_plan_tables(sls_ctx* ctx, int dev_slot, const sls_dims* dims, const sls_plant* P, const sls_csc_bool* Sx,
                          const sls_csc_bool* Su, int64_t ngroups, const int64_t* group_ptr, const int64_t* group_cols,
                          int host_tables, int64_t* md_total, uint8_t* mask_out, int32_t* dest_out, int32_t* was_compact);

/* The same tables (and the concatenated index sets s_x, s_u of all columns, 0-based: *n_idx entries) of a plan built by the
 * device-resident symbolic route (sls_h2_sf_plan_localized), for the bit-for-bit comparison with the host route. */
int sls_debug_plan_tables_localized(sls_ctx* ctx, int dev_slot, const sls_dims* dims, const sls_plant* P, int64_t d, double alpha,
                                    int64_t* md_total, uint8_t* mask_out, int32_t* dest_out, int64_t* n_idx, int32_t* idx_out);

#ifdef __cplusplus
}
#endif
#endif /* SLS_MI355X_DEBUG_H */
