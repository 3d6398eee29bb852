import os, sys
os.environ["SLS_LAB"]="1"
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/oracle")
src = open(ROOT+"/tools/fuzz_h2.py").read().split("modes = {")[0]
ns = {"__file__": ROOT+"/tools/fuzz_h2.py"}; exec(compile(src, "f", "exec"), ns)
import numpy as np, slc_amd as slc, sls_oracle as o
P, S, meta = ns["problem"](401)
col=51
Po = o.OraclePlant(P.A, P.B1, P.B2, P.C1, P.D11, P.D12)
z, oi, d = o.solve_group(Po, [col], S[0], S[1])
print("oracle resid", d["resid"], "rank", d["rank"], "nfree", len(z), "n", oi["n"], "m", oi["m"], "smin", d["smin"])
E=d["E"]; f=d["f"]; s=np.linalg.svd(E, compute_uv=False); print("sigma tail", s[-6:])
for env in ({}, {"SLS_TILE":"all","SLS_FORCE_GENERAL":"1"}, {"SLS_NO_TWISTED":"1"}):
    for k,v in env.items(): os.environ[k]=v
    ctx = slc.Context([0])
    Px, Pu, info = slc.SLS_H2(P, S, [[col]], ctx=ctx, return_info=True, dropzeros=False)
    got = np.array([(Px if kind == 0 else Pu)[t][(oi["sx"] if kind == 0 else oi["su"])[r], col] for (t, kind, r, _) in oi["var_index"]])
    print(env, "status", info["col_status"], "max_resid", info["max_residual"], "|got|max %.3e" % np.abs(got).max(), "||E got - f||inf %.3e" % np.abs(E@got-f).max())
    ctx.close()
    for k in env: del os.environ[k]
