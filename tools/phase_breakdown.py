"""Diagnostics: per-phase cycle breakdown of the wave kernel (SLS_PHASE_TIMERS=1).  usage: phase_breakdown.py [workload]"""
import ctypes as C, os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
os.environ["SLS_PHASE_TIMERS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slc_amd
name = sys.argv[1] if len(sys.argv) > 1 else "readme_chain"
P, S, meta = slc_amd.workloads.make_workload(name)
ctx = slc_amd.Context([0])
plan = slc_amd.Plan(ctx, P, S)
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
names = ["setup", "residual", "build", "gj", "store", "sweeps", "gj:col_rt", "gj:compute"]
tot = b.sum(1)
k = int(np.argmax(tot))
print(f"{name}: {ns} subproblems, kernel avg {ms:.4f} ms over {n} launches; s_memtime ticks ≈ shader-clock cycles (a wave's two columns of chain-4096: 4.0 M ticks in 2.2 ms)")
print("slowest subproblem", k, "iters", it[k], "total ticks", tot[k])
for q in range(8):
    print(f"  {names[q]:9s} max {b[:, q].max():10.0f}  mean {b[:, q].mean():10.0f}  ticks   ({100 * b[k, q] / tot[k]:5.1f}% of slowest)")
