"""grid-32: the slowly converging columns against the SVD oracle (diagnostics)."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import slc_amd, sls_oracle as o
from conftest import flat_phi
P, S, _ = slc_amd.workloads.make_workload("grid32")
cols = [90, 483, 69, 102, 387, 495]
ctx = slc_amd.Context([0])
Phix, Phiu, info = slc_amd.SLS_H2(P, S, [[c] for c in cols], ctx=ctx, return_info=True, dropzeros=False)
ox, ou, dg = o.SLS_H2(o.OraclePlant(P.A, P.B1, P.B2), S, I=[[c] for c in cols], return_diag=True)
got = np.concatenate([flat_phi(Phix, S[0]), flat_phi(Phiu, S[1])]); want = np.concatenate([flat_phi(ox, S[0]), flat_phi(ou, S[1])])
col = np.concatenate([np.repeat(np.arange(P.Nx), np.diff(M.indptr)) for M in S[0] + S[1]])
for q, c in enumerate(cols):
    m = col == c
    print(c, "n", dg[q]["n"], "status", info["col_status"][q], "err vs SVD", np.abs(got[m] - want[m]).max(), "svd resid", dg[q]["resid"])
