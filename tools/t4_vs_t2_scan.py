"""Diagnostics: four-wave twisted kernel against the two-wave kernel on tools/fuzz_h2.py plants — statuses and residuals per column.
usage: t4_vs_t2_scan.py seed [seed ...]"""
import os, sys
os.environ.setdefault("SLS_LAB", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
src = open(os.path.join(ROOT, "tools", "fuzz_h2.py")).read().split("modes = {")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "fuzz_h2.py")}
exec(compile(src, "fuzz_h2.py", "exec"), ns)
import numpy as np, slc_amd as slc
ctx = slc.Context([0])
for seed in [int(s) for s in sys.argv[1:]]:
    P, S, meta = ns["problem"](seed)
    out = {}
    for mode in ("1", "0"):
        os.environ["SLS_TWISTED4"] = mode
        plan = slc.Plan(ctx, P, S)
        desc = plan.describe()
        d = plan.alloc_values(); plan.execute(d); plan.synchronize()
        st, rs, it = plan.fetch_status()
        vals = np.concatenate(sum(plan.download(d), []))
        out[mode] = (st.copy(), rs.copy(), it.copy(), vals, desc)
        plan.close()
    del os.environ["SLS_TWISTED4"]
    if "twisted4" not in out["1"][4]:
        continue
    st4, rs4, it4, v4, d4 = out["1"]; st2, rs2, it2, v2, d2 = out["0"]
    bad = np.flatnonzero(st4 != st2)
    both = (st4 == 0) & (st2 == 0)
    colidx = np.concatenate([np.repeat(np.arange(P.Nx), np.diff(M.indptr)) for M in S[0] + S[1]])
    dv = np.zeros(P.Nx); np.maximum.at(dv, colidx, np.abs(v4 - v2))
    worst = float(dv[both].max()) if both.any() else 0.0
    if len(bad) == 0 and worst > 1e-9:
        print(seed, meta, d4.split(" ")[0], "statuses equal, but max|dΦ| over columns both call solved = %.1e (column %d)" % (worst, int(np.argmax(np.where(both, dv, 0)))), flush=True)
    sx = [np.flatnonzero(np.asarray(sum(M[:, c] for M in S[0]).todense()).ravel()).size for c in range(P.Nx)]
    print(seed, meta, d4.split(" ")[0], "columns", P.Nx, "status differs at", [(int(c), "n~%d" % sx[c], int(st4[c]), "%.1e" % rs4[c], int(it4[c]), "two-wave:", int(st2[c]), "%.1e" % rs2[c], int(it2[c])) for c in bad][:6],
          "max|dΦ| over columns both call solved %.1e" % worst, flush=True)
