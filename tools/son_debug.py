"""Sum-of-norms mode against the CPU oracle on a small chain (diagnostics)."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, slc_amd as slc, sls_oracle as o, sls_son_oracle as son
from conftest import flat_phi
P = slc.workloads.chain_plant(23)
S = list(slc.workloads.localization_masks(P.A, P.B2, 6, 18, 1.5))
cols = [0, 5, 11, 22]
ctx = slc.Context([0])
plan = slc.Plan(ctx, P, S, [[c] for c in cols], objective="sum_of_norms")
print(plan.describe())
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
vx, vu = plan.download(d); st, rs, it = plan.fetch_status()
Phix, Phiu = slc.assemble_phi(S[0], S[1], vx, vu, dropzeros=False)
ox, ou, dg = son.SLS_SON(o.OraclePlant(P.A, P.B1, P.B2), S, cols=cols)
for q, c in enumerate(cols):
    obj = sum(np.sqrt((Fx[:, c].toarray() ** 2).sum() + (Fu[:, c].toarray() ** 2).sum()) for Fx, Fu in zip(Phix, Phiu))
    err = max(max(abs(Fx[:, c] - Ox[:, c]).max() for Fx, Ox in zip(Phix, ox)), max(abs(Fu[:, c] - Ou[:, c]).max() for Fu, Ou in zip(Phiu, ou)))
    print(c, "status", st[q], "resid %.1e" % rs[q], "admm steps", it[q], "obj gpu %.10f oracle %.10f (gap %.1e)" % (obj, dg[q]["obj"], dg[q]["gap"]), "max |ΔΦ| %.1e" % err)
