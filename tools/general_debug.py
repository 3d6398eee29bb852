import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, scipy.sparse as sp, slc_amd as slc
from conftest import flat_phi
g = np.load(os.path.join(ROOT, "tests", "golden", "general_weights_phi.npz"))
Nx = int(g["Nx"]); Pc = slc.workloads.chain_plant(Nx); Nu = Pc.Nu
W = sp.csc_matrix((g["W_data"], g["W_indices"], g["W_indptr"]), shape=(Nx + Nu, Nx + Nu))
D11 = sp.csc_matrix((g["D11_data"], g["D11_indices"], g["D11_indptr"]), shape=(Nx + Nu, Nx))
P = slc.Plant(Pc.A, sp.diags(g["b"]).tocsc(), Pc.B2, W[:, :Nx], D11, W[:, Nx:])
S = list(slc.workloads.localization_masks(P.A, P.B2, int(g["d"]), int(g["T"]), float(g["alpha"])))
ctx = slc.Context([0]); plan = slc.Plan(ctx, P, S); d = plan.alloc_values(); plan.execute(d); plan.synchronize()
vx, vu = plan.download(d); st, rs, it = plan.fetch_status()
got = np.concatenate(vx + vu); want = np.concatenate([g["vals_x"], g["vals_u"]])
col = np.concatenate([np.repeat(np.arange(P.Nx), np.diff(M.indptr)) for M in S[0] + S[1]])
for c in range(Nx):
    m = col == c
    print(c, "st", st[c], "resid %.2e" % rs[c], "iters", it[c], "err %.2e" % np.abs(got[m] - want[m]).max(), "oracle resid %.1e" % g["col_resid"][c])

