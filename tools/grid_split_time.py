"""grid-32: time of the tile-kernel columns alone, of the one-wave columns alone, and of both (diagnostics for the launch overlap)."""
import os, sys, time
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, slc_amd
P, S, meta = slc_amd.workloads.make_workload("grid32")
ctx = slc_amd.Context([0])
nx = np.diff(((S[0][-1].astype(np.int32)) @ (P.A != 0).astype(np.int32)).tocsc().indptr)
sets = {"all": list(range(P.Nx)), "tile columns (nx > 64)": [c for c in range(P.Nx) if nx[c] > 64], "wave columns (nx <= 64)": [c for c in range(P.Nx) if nx[c] <= 64]}
for name, cols in sets.items():
    plan = slc_amd.Plan(ctx, P, S, [[c] for c in cols])
    d = plan.alloc_values(); plan.execute(d); plan.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): plan.execute(d)
    plan.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {len(cols)} columns, {1e3*dt:.3f} ms per pass; {plan.describe()}")
    plan.close()
