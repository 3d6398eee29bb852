"""One-wave kernel per size class against the NumPy oracle on random plants, one column per plan (SLS_NO_TWISTED=1): localises
class-specific failures (diagnostics)."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
os.environ["SLS_NO_TWISTED"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp, slc_amd as slc, sls_oracle as o
ctx = slc.Context([0])
for seed in (1, 2, 3):
    rng = np.random.default_rng(40 + seed)
    Nx = 36
    A = sp.random(Nx, Nx, density=0.08, random_state=seed, format="csc") * 0.5 + sp.eye(Nx, format="csc")
    B2 = sp.eye(Nx, format="csc")[:, ::2]
    P = slc.Plant(A, sp.eye(Nx, format="csc"), B2)
    S = list(slc.workloads.localization_masks(P.A, P.B2, 3, 8, 1.5))
    Po = o.OraclePlant(P.A, P.B1, P.B2)
    for c in range(Nx):
        z, info, d = o.solve_group(Po, [c], S[0], S[1])
        plan = slc.Plan(ctx, P, S, [[c]])
        dv = plan.alloc_values(); plan.execute(dv); plan.synchronize()
        st, rs, it = plan.fetch_status()
        desc = plan.describe().split(" ")[0]
        plan.close()
        flag = "" if (st[0] == 0) == (d["resid"] < 1e-9) else "   <<<<< MISMATCH"
        if info["n"] >= 29 or flag:
            print(seed, c, "n", info["n"], desc, "gpu status", st[0], "resid %.0e iters %d" % (rs[0], it[0]), "| oracle resid %.0e" % d["resid"], flag)
