"""One column of a tools/fuzz_h2.py problem on the default routing: status, residual, passes (diagnostics; env knobs apply)."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
src = open(os.path.join(ROOT, "tools", "fuzz_h2.py")).read().split("modes = {")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "fuzz_h2.py")}
exec(compile(src, "fuzz_h2.py", "exec"), ns)
import slc_amd as slc
seed, col = int(sys.argv[1]), int(sys.argv[2])
P, S, meta = ns["problem"](seed)
ctx = slc.Context([0])
plan = slc.Plan(ctx, P, S, [[col]])
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
st, rs, it = plan.fetch_status()
print(meta, plan.describe(), "status", st[0], "resid %.2e" % rs[0], "passes", it[0])
