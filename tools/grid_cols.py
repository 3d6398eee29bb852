import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, slc_amd
P, S, meta = slc_amd.workloads.make_workload("grid32")
cols = [0, 31, 200, 495, 500, 528, 529, 1023]
ctx = slc_amd.Context([0]); plan = slc_amd.Plan(ctx, P, S, [[c] for c in cols])
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
st, rs, it = plan.fetch_status()
print("delta", os.environ.get("SLS_DELTA_REL"), "tol", os.environ.get("SLS_TOL"), "max_iters", os.environ.get("SLS_MAX_ITERS"))
print(" status", st.tolist(), "iters", it.tolist(), "resid", " ".join("%.1e" % r for r in rs))
