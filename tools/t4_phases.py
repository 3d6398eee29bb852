"""Phase shares inside the four-wave twisted kernel's factor half (SLS_PHASE_TIMERS=1: laps; 2: chain waves; 3: helper waves)."""
import ctypes as C, os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
lvl = str(int(os.environ.setdefault("SLS_PHASE_TIMERS", "2")) % 10)
import slc_amd
name = sys.argv[1] if len(sys.argv) > 1 else "readme_chain"
P, S, meta = slc_amd.workloads.make_workload(name)
ctx = slc_amd.Context([0]); plan = slc_amd.Plan(ctx, P, S)
print(plan.describe())
d = plan.alloc_values()
for _ in range(3): plan.execute(d)
plan.synchronize()
lib = ctx._lib
lib.sls_plan_debug_phase_cycles.restype = C.c_int; lib.sls_plan_debug_phase_cycles.argtypes = [C.c_void_p, C.c_void_p]
ns = plan.info["n_subproblems"]; buf = np.zeros(ns * 8, dtype=np.uint64)
assert lib.sls_plan_debug_phase_cycles(plan.handle, buf.ctypes.data) == 0
b = buf.reshape(ns, 2, 4).astype(np.int64)
if lvl == "1":
    k = int(np.argmax(b[:, 0, 1]))
    for w in (0, 1):
        hi = int(b[k, w, 3]) >> 32; lo = int(b[k, w, 3]) & 0xffffffff
        print(f"chain wave {w} (column {k}): setup {b[k,w,0]}, own factor half {b[k,w,1]}, wait {b[k,w,2]}, middle+outward {hi}, later passes+residual+output {lo}; total {b[k,w,0]+b[k,w,1]+b[k,w,2]+hi+lo}")
else:
    k = int(np.argmax(b[:, 0, 3]))
    names = ("hand-off waits", "Gauss-Jordan", "convert+store+sweep") if lvl == "2" else ("static part + border", "elimination (incl. waits)", "of which waiting for pivots")
    for w in (0, 1):
        print(f"{'chain' if lvl == '2' else 'helper'} wave dir {w} (column {k}): " + ", ".join(f"{nm} {b[k,w,q]}" for q, nm in enumerate(names)) + f"; factor half {b[k,w,3]}")
