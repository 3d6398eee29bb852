"""Diagnostics: residual after 1, 2, 3, 4 passes per column (stagnation rule off) — the ratios the stagnation rules judge.
usage: stag_ratio_hist.py [workload]   (sets SLS_STAG / SLS_MAX_ITERS / SLS_MAX_ITERS_SLOW itself)"""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
name = sys.argv[1] if len(sys.argv) > 1 else "grid32"
os.environ["SLS_STAG"] = "2"; os.environ["SLS_MAX_ITERS_SLOW"] = "0"
import slc_amd
P, S, meta = slc_amd.workloads.make_workload(name)
ctx = slc_amd.Context([0])
res = {}
for k in (1, 2, 3, 4, 40):
    os.environ["SLS_MAX_ITERS"] = str(k)
    plan = slc_amd.Plan(ctx, P, S)
    d = plan.alloc_values(); plan.execute(d); plan.synchronize()
    st, rs, it = plan.fetch_status()
    res[k] = (st.copy(), rs.copy(), it.copy())
    del plan
r1, r2, r3, r4 = (res[k][1] for k in (1, 2, 3, 4))
final = res[40]
feas = final[1] <= 1e-9
print(name, "columns", len(r1), "converged with 40 passes:", int(feas.sum()), "max passes among them", int(final[2][feas].max()) if feas.any() else 0)
for label, sel in (("end feasible", feas), ("end infeasible", ~feas)):
    sel = sel & (r1 > 1e-12)
    if not sel.any():
        continue
    q21 = r2[sel] / r1[sel]; q32 = r3[sel] / np.maximum(r2[sel], 1e-300); q43 = r4[sel] / np.maximum(r3[sel], 1e-300)
    print(label, int(sel.sum()), "columns with r1 > tol")
    for nm, q in (("r2/r1", q21), ("r3/r2", q32), ("r4/r3", q43)):
        print("  ", nm, "quantiles 5/25/50/75/95:", " ".join("%.3f" % v for v in np.quantile(q, [0.05, 0.25, 0.5, 0.75, 0.95])),
              " share < 0.5: %.3f, in [0.5, 0.9): %.3f, in [0.9, 0.97): %.3f, ≥ 0.97: %.3f" % ((q < 0.5).mean(), ((q >= 0.5) & (q < 0.9)).mean(), ((q >= 0.9) & (q < 0.97)).mean(), (q >= 0.97).mean()))
