"""grid-32: passes / residuals of the slowly converging columns under different stagnation rules (diagnostics)."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import slc_amd
P, S, _ = slc_amd.workloads.make_workload("grid32")
ctx = slc_amd.Context([0])
plan = slc_amd.Plan(ctx, P, S); d = plan.alloc_values(); plan.execute(d); plan.synchronize()
plan.execute(d); plan.synchronize(); ms, _ = plan.kernel_time_ms()
st, rs, it = plan.fetch_status()
print(os.environ.get("SLS_STAG"), os.environ.get("SLS_MAX_ITERS"), "ms", round(ms, 3), "status", np.bincount(st, minlength=3).tolist(), "passes", np.bincount(it).tolist())
ok = st == 0
print("   ok columns: resid > 1e-12:", int((rs[ok] > 1e-12).sum()), " max resid", rs[ok].max(), " col 90:", st[90], rs[90], it[90], " col 200:", st[200], rs[200], it[200])
