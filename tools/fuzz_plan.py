import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("SLS_LAB", "1")
src = open(os.path.join(ROOT, "tools", "fuzz_h2.py")).read().split("modes = {")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "fuzz_h2.py")}
exec(compile(src, "fuzz_h2.py", "exec"), ns)
import numpy as np, slc_amd as slc
seed = int(sys.argv[1]); cols = [int(c) for c in sys.argv[2:]]
P, S, meta = ns["problem"](seed)
ctx = slc.Context([0])
plan = slc.Plan(ctx, P, S)
print(plan.describe())
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
st, rs, it = plan.fetch_status()
print("resident plan: status!=0 at", np.flatnonzero(st != 0).tolist())
for c in cols: print(" col", c, "status", st[c], "resid %.2e" % rs[c], "passes", it[c])
Px, Pu, info = slc.SLS_H2(P, S, ctx=ctx, return_info=True, dropzeros=False, index_base=seed % 2)
print("one-shot: n_refined", info.get("n_refined"), "status!=0 at", np.flatnonzero(info["col_status"] != 0).tolist())
for k, v in (("SLS_TWISTED4", "0"), ("SLS_NO_TWISTED", "1")):
    os.environ[k] = v
    plan2 = slc.Plan(ctx, P, S); d2 = plan2.alloc_values(); plan2.execute(d2); plan2.synchronize()
    st2, rs2, it2 = plan2.fetch_status()
    print(k, plan2.describe()[:90], "status!=0 at", np.flatnonzero(st2 != 0).tolist(), [("%.1e" % rs2[c], int(it2[c])) for c in cols])
    del os.environ[k]
