"""Phase breakdown of the general (workgroup) kernel on a few grid-32 columns (SLS_PHASE_TIMERS=1)."""
import ctypes as C, os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
os.environ["SLS_PHASE_TIMERS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, slc_amd
P, S, meta = slc_amd.workloads.make_workload("grid32")
cols = [495, 500, 528, 529, 200]
ctx = slc_amd.Context([0]); plan = slc_amd.Plan(ctx, P, S, [[c] for c in cols])
print(plan.describe())
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
st, rs, it = plan.fetch_status()
lib = ctx._lib
lib.sls_plan_debug_phase_cycles.restype = C.c_int; lib.sls_plan_debug_phase_cycles.argtypes = [C.c_void_p, C.c_void_p]
ns = len(cols); buf = np.zeros(ns * 8, dtype=np.uint64)
assert lib.sls_plan_debug_phase_cycles(plan.handle, buf.ctypes.data) == 0
b = buf.reshape(ns, 8).astype(float)
names = ["setup", "residual", "build", "gj", "store", "sweeps"]
for q, c in enumerate(cols):
    tot = b[q, :6].sum()
    print(f"col {c}: status {st[q]} iters {it[q]} total {tot:.0f} cycles | " + ", ".join(f"{names[s]} {b[q, s]:.0f} ({100 * b[q, s] / tot:.0f}%)" for s in range(6)))
