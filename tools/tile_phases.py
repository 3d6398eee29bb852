"""Diagnostics: per-phase cycle breakdown of the tile kernel's columns (SLS_PHASE_TIMERS=1).  usage: tile_phases.py [workload]"""
import ctypes as C, os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
os.environ.setdefault("SLS_PHASE_TIMERS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slc_amd
name = sys.argv[1] if len(sys.argv) > 1 else "grid32"
P, S, meta = slc_amd.workloads.make_workload(name)
ctx = slc_amd.Context([0])
plan = slc_amd.Plan(ctx, P, S)
print(plan.describe())
d = plan.alloc_values()
for _ in range(3):
    plan.execute(d)
plan.synchronize()
ms, n = plan.kernel_time_ms()
lib = ctx._lib
lib.sls_plan_debug_phase_cycles.restype = C.c_int
lib.sls_plan_debug_phase_cycles.argtypes = [C.c_void_p, C.c_void_p]
ns = plan.info["n_subproblems"]
buf = np.zeros(ns * 8, dtype=np.uint64)
assert lib.sls_plan_debug_phase_cycles(plan.handle, buf.ctypes.data) == 0
b = buf.reshape(ns, 8).astype(np.float64)
st, rs, it = plan.fetch_status()
tot = b.sum(1)
print(f"{name}: {ns} subproblems, kernel avg {ms:.4f} ms over {n} launches; s_memtime ticks ≈ shader-clock cycles")
for lo, hi in ((0, 0), (1, 2), (3, 99)):
    sel = (it >= lo) & (it <= hi) if hi < 99 else it >= lo
    if not sel.any():
        continue
    print(f"passes {lo}..{hi}: {int(sel.sum())} columns, status counts {np.bincount(st[sel])}, mean total {np.median(tot[sel]) * 1e-3:.0f} k ticks (median), max {tot[sel].max() * 1e-3:.0f} k")
    print("   slots (setup, residual, build, inversion, store, sweeps, -, -), median k ticks:", " ".join(f"{np.median(b[sel, q]) * 1e-3:8.0f}" for q in range(8)))
