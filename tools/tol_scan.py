import sys, os, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
import numpy as np, slc_amd as slc
from conftest import flat_phi
g = np.load("/root/repo/tests/golden/readme_chain_phi.npz")
P, S, _ = slc.workloads.make_workload("readme_chain")
ctx = slc.Context([0])
plan = slc.Plan(ctx, P, S)
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
t0 = time.perf_counter()
for _ in range(300): plan.execute(d)
plan.synchronize(); dt = (time.perf_counter() - t0) / 300
st, rs, it = plan.fetch_status()
vx, vu = plan.download(d)
got = np.concatenate([*vx, *vu])
print("tol", os.environ.get("SLS_TOL"), "delta", os.environ.get("SLS_DELTA_REL"), "%.4f ms" % (1e3 * dt), "passes hist", np.bincount(it).tolist(), "max resid %.1e" % rs.max(), end=" ")
if g is not None:
    want = np.concatenate([g["vals_x"], g["vals_u"]]); print("max|dPhi| vs golden %.1e" % np.abs(got - want).max())
else: print()
