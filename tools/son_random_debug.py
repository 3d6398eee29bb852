import sys, os; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
import numpy as np, scipy.sparse as sp, slc_amd as slc
seed=1
rng = np.random.default_rng(40 + seed)
Nx = 36
A = sp.random(Nx, Nx, density=0.08, random_state=seed, format="csc") * 0.5 + sp.eye(Nx, format="csc")
B2 = sp.eye(Nx, format="csc")[:, ::2]
Nu = B2.shape[1]
q = rng.uniform(0.5, 2.0, Nx); r = rng.uniform(0.5, 2.0, Nu)
C1 = sp.vstack([sp.diags(q), sp.csc_matrix((Nu, Nx))]).tocsc()
D12 = sp.vstack([sp.csc_matrix((Nx, Nu)), sp.diags(r)]).tocsc()
B1 = sp.diags(rng.uniform(0.6, 1.4, Nx)).tocsc()
P = slc.Plant(A, B1, B2, C1, 0, D12)
S = list(slc.workloads.localization_masks(P.A, P.B2, 3, 8, 1.5))
cols = sorted(int(c) for c in rng.permutation(Nx)[:12])
if os.environ.get('SON_COLS'): cols = [int(c) for c in os.environ['SON_COLS'].split(',')]
ctx = slc.Context([0])
for obj in ("h2", "sum_of_norms"):
    plan = slc.Plan(ctx, P, S, [[c] for c in cols], objective=obj)
    print(obj, os.environ.get("SLS_NO_TWISTED"), os.environ.get("SLS_DELTA_FIRST"), plan.describe())
    d = plan.alloc_values(); plan.execute(d); plan.synchronize()
    st, rs, it = plan.fetch_status()
    print("  cols", cols); print("  status", st.tolist()); print("  resid", ["%.0e" % x for x in rs]); print("  iters", it.tolist())
    plan.close()
