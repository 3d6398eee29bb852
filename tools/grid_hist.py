"""Step history (trial residual, step length) of a few grid-32 columns (SLS_PHASE_TIMERS=3)."""
import ctypes as C, os, sys
os.environ["SLS_PHASE_TIMERS"] = "3"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, slc_amd
P, S, _ = slc_amd.workloads.make_workload("grid32")
cols = [90, 483, 495, 200]
ctx = slc_amd.Context([0]); plan = slc_amd.Plan(ctx, P, S, [[c] for c in cols])
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
st, rs, it = plan.fetch_status()
buf = np.zeros(len(cols) * 8, dtype=np.uint64)
ctx._lib.sls_plan_debug_phase_cycles.argtypes = [C.c_void_p, C.c_void_p]
ctx._lib.sls_plan_debug_phase_cycles(plan.handle, buf.ctypes.data)
h = buf.view(np.float64).reshape(-1, 8)
for q, c in enumerate(cols):
    print(c, "status", st[q], "resid", rs[q], "passes", it[q], "| trial resid", h[q, :4], "| alpha", h[q, 4:])
