import os, sys
os.environ["SLS_LAB"]="1"; os.environ["SLS_T4_NMIN"]="0"
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/oracle")
src = open(ROOT+"/tools/fuzz_h2.py").read().split("modes = {")[0]
ns = {"__file__": ROOT+"/tools/fuzz_h2.py"}; exec(compile(src, "f", "exec"), ns)
import numpy as np, slc_amd as slc
ctx = slc.Context([0])
P, S0, meta = ns["problem"](297)
for T in (4,5,6,7,8,10,12,16):
    S = list(slc.workloads.localization_masks(P.A, P.B2, meta["d"], T, 1.5))
    res = {}
    for mode in ("1","0"):
        os.environ["SLS_TWISTED4"]=mode
        plan = slc.Plan(ctx, P, S); desc = plan.describe()
        d = plan.alloc_values(); plan.execute(d); plan.synchronize()
        st, rs, it = plan.fetch_status(); res[mode]=(st.copy(), rs.copy(), desc); plan.close()
    bad = np.flatnonzero(res["1"][0] != res["0"][0])
    print("T", T, res["1"][2].split(" ")[0], "differs at", bad.tolist(), ["%.1e" % res["1"][1][c] for c in bad])
# single small column with a big partner
S = S0
info = slc.Plan(ctx, P, S).info
for small in (15, 33, 24, 17):
    for partner in (0, 5, 20):
        os.environ["SLS_TWISTED4"]="1"
        plan = slc.Plan(ctx, P, S, [[small],[partner]]); desc = plan.describe()
        d = plan.alloc_values(); plan.execute(d); plan.synchronize()
        st, rs, it = plan.fetch_status(); plan.close()
        print("pair", small, partner, desc.split(" ")[0], "status", st.tolist(), ["%.1e" % r for r in rs], it.tolist())
