"""grid-32 full size: per-column |Φ_gpu − Φ_cport|, with residuals / passes of both (diagnostics)."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import slc_amd, sls_oracle as o, sls_oracle_cport as cp
from conftest import flat_phi
P, S, _ = slc_amd.workloads.make_workload("grid32")
ctx = slc_amd.Context([0])
Phix, Phiu, info = slc_amd.SLS_H2(P, S, ctx=ctx, return_info=True, dropzeros=False)
plan = slc_amd.Plan(ctx, P, S); d = plan.alloc_values(); plan.execute(d); plan.synchronize(); st, rs, it = plan.fetch_status()
ox, ou, oi = cp.SLS_H2(o.OraclePlant(P.A, P.B1, P.B2), S, nthreads=8)
got = np.concatenate([flat_phi(Phix, S[0]), flat_phi(Phiu, S[1])]); want = np.concatenate([flat_phi(ox, S[0]), flat_phi(ou, S[1])])
col = np.concatenate([np.repeat(np.arange(P.Nx), np.diff(M.indptr)) for M in S[0] + S[1]])
err = np.zeros(P.Nx); np.maximum.at(err, col, np.abs(got - want))
feas = oi["status"] == 0
idx = np.argsort(-np.where(feas, err, 0))[:12]
for c in idx:
    print(f"col {c:4d} err {err[c]:.2e} gpu: st {st[c]} resid {rs[c]:.2e} passes {it[c]} | cport: st {oi['status'][c]} resid {oi['resid'][c]:.2e} iters {oi['iters'][c]}")
print("feasible columns with err > 1e-8:", int((feas & (err > 1e-8)).sum()), " gpu resid > 1e-12 among feasible:", int((feas & (rs > 1e-12)).sum()))
dif = np.flatnonzero((st == 0) != (oi["status"] == 0))
for c in dif: print(f"status differs col {c}: gpu st {st[c]} resid {rs[c]:.2e} passes {it[c]} | cport st {oi['status'][c]} resid {oi['resid'][c]:.2e} iters {oi['iters'][c]}")
wp = feas & (st == 0) & (rs <= 1e-12) & (oi["resid"] <= 1e-12)
print("well-posed feasible columns:", int(wp.sum()), "max err", err[wp].max(), "; other feasible:", int((feas & ~wp).sum()), "max err", err[feas & ~wp].max())
