"""One column of a tools/fuzz_h2.py problem against the oracle: status, residual, passes, max |ΔΦ| (diagnostics; env knobs apply)."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
src = open(os.path.join(ROOT, "tools", "fuzz_h2.py")).read().split("modes = {")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "fuzz_h2.py")}
exec(compile(src, "fuzz_h2.py", "exec"), ns)
import numpy as np, slc_amd as slc, sls_oracle as o
seed, col = int(sys.argv[1]), int(sys.argv[2])
P, S, meta = ns["problem"](seed)
Po = o.OraclePlant(P.A, P.B1, P.B2, P.C1, P.D11, P.D12)
z, oi, d = o.solve_group(Po, [col], S[0], S[1])
E = d["E"]; sv = np.linalg.svd(E, compute_uv=False)
ctx = slc.Context([0])
plan = slc.Plan(ctx, P, S, [[col]]); desc = plan.describe(); plan.close()
Px, Pu, info = slc.SLS_H2(P, S, [[col]], ctx=ctx, return_info=True, dropzeros=False)
got = np.array([(Px if kind == 0 else Pu)[t][(oi["sx"] if kind == 0 else oi["su"])[r], col] for (t, kind, r, _) in oi["var_index"]])
print(desc, "| n", oi["n"], "m", oi["m"], "rank", d["rank"], "of", E.shape, "sigma_min+ %.1e" % sv[d["rank"] - 1], "| status", info["col_status"][0],
      "max resid %.1e" % info["max_residual"], "passes", info["max_iters"], "| max|dPhi| %.2e" % np.abs(got - z).max(), "gpu ‖Ez-f‖ %.1e" % np.abs(E @ got - d["f"]).max(),
      "cost gpu-oracle %.2e" % (np.sum((d["M"] @ got + d["m0"]) ** 2) - np.sum((d["M"] @ z + d["m0"]) ** 2)))
