import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, slc_amd
name = sys.argv[1] if len(sys.argv) > 1 else "readme_chain"
P, S, meta = slc_amd.workloads.make_workload(name)
ctx = slc_amd.Context([0]); plan = slc_amd.Plan(ctx, P, S)
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
st, rs, it = plan.fetch_status()
print(name, "iters histogram", np.bincount(it), "status", np.bincount(st), "resid max %.2e" % rs.max())
for k in np.unique(it): print("  iters", k, "resid range %.1e .. %.1e" % (rs[it == k].min(), rs[it == k].max()))
print(plan.describe())
for _ in range(3):
    plan.execute(d)
plan.synchronize()
ms, nl = plan.kernel_time_ms()
print("kernel avg %.3f ms over %d launches; n_sub %d; max_nx %d max_nu %d" % (ms, nl, plan.info["n_subproblems"], plan.info["max_nx"], plan.info["max_nu"]))
