import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, slc_amd
name = sys.argv[1] if len(sys.argv) > 1 else "readme_chain"
P, S, meta = slc_amd.workloads.make_workload(name)
ctx = slc_amd.Context([0]); plan = slc_amd.Plan(ctx, P, S)
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
st, rs, it = plan.fetch_status()
print(name, "delta", os.environ.get("SLS_DELTA_REL"), "max_iters", os.environ.get("SLS_MAX_ITERS"))
print(" resid per column:", " ".join("%.0e" % r for r in rs))
