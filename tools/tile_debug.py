"""Per-column status / residual / passes of sample columns under the kernel selection modes (diagnostics)."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slc_amd
wl = slc_amd.workloads

cases = {
    "grid32": lambda: (wl.make_workload("grid32")[:2], [200, 495, 500, 528, 529, 530, 531]),
    "chain_d40": lambda: ((wl.chain_plant(200),) , [60, 100, 101, 140]),
    "grid16_d6": lambda: ((wl.grid_plant(16, 3),), [119, 136, 120]),
    "grid16_d8": lambda: ((wl.grid_plant(16, 3),), [119, 136, 120]),
    "grid16_d5": lambda: ((wl.grid_plant(16, 3),), [119, 136, 120]),
}
spec = {"chain_d40": (40, 90), "grid16_d6": (6, 14), "grid16_d8": (8, 14), "grid16_d5": (5, 20)}
for name in sys.argv[1:]:
    r = cases[name]()
    if name == "grid32":
        (P, S), cols = r
    else:
        (P,), cols = r
        d, T = spec[name]
        S = list(wl.localization_masks(P.A, P.B2, d, T, 1.5))
    for mode in (("0", "0"), ("all", "0"), ("all", "1")):
        os.environ["SLS_TILE"] = mode[0]; os.environ["SLS_TILE_GLOBAL"] = mode[1]
        ctx = slc_amd.Context([0])
        plan = slc_amd.Plan(ctx, P, S, [[c] for c in cols])
        d_ = plan.alloc_values(); plan.execute(d_); plan.synchronize()
        st, rs, it = plan.fetch_status()
        print(name, mode, plan.describe(), "max_nx", plan.info["max_nx"], "max_nu", plan.info["max_nu"])
        print("  status", st.tolist()); print("  resid ", ["%.2e" % r for r in rs]); print("  iters ", it.tolist())
        plan.close(); ctx.close()
