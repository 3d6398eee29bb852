import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, scipy.sparse as sp
import slc_amd
from test_gpu_parity import _weighted_problem
from conftest import flat_phi
P, S, g = _weighted_problem(slc_amd)
variant = sys.argv[1] if len(sys.argv) > 1 else "full"
if variant == "nod11":
    P = slc_amd.Plant(P.A, P.B1, P.B2, P.C1, 0, P.D12)
if variant == "b1":
    P = slc_amd.Plant(P.A, sp.identity(P.Nx, format="csc"), P.B2, P.C1, P.D11, P.D12)
ctx = slc_amd.Context([0])
plan = slc_amd.Plan(ctx, P, S)
d = plan.alloc_values()
plan.execute(d); plan.synchronize()
st, rs, it = plan.fetch_status()
vx, vu = plan.download(d)
got = np.concatenate(vx + vu)
print("variant", variant, "force_general", os.environ.get("SLS_FORCE_GENERAL"))
print("status", st.tolist()); print("iters", it.tolist()); print("resid", ["%.1e" % r for r in rs])
if variant == "full":
    want = np.concatenate([g["vals_x"], g["vals_u"]])
    cols = np.concatenate([np.repeat(np.arange(P.Nx), np.diff(M.indptr)) for M in S[0] + S[1]])
    err = np.array([np.abs(got[cols == c] - want[cols == c]).max() for c in range(P.Nx)])
    print("err/col", ["%.1e" % e for e in err])
