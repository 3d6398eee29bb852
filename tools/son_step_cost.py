"""Per-step cost of the sum-of-norms ADMM in the one-wave kernel on chain-4096: Σ steps over the columns, the resident pass time, and the
implied time of one ADMM step of one wave (pass time × resident waves / Σ steps), with and without the Anderson acceleration."""
import os, sys, time
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, slc_amd as slc
P, S, meta = slc.workloads.make_workload("chain4096")
for env in ({}, {"SLS_SON_ANDERSON": "0"}):
    os.environ.update(env)
    ctx = slc.Context([0])
    plan = slc.Plan(ctx, P, S, objective="sum_of_norms")
    dv = plan.alloc_values()
    plan.execute(dv); plan.synchronize()
    t0 = time.perf_counter(); plan.execute(dv); plan.synchronize(); dt = time.perf_counter() - t0
    st, rs, it = plan.fetch_status()
    print(env, "pass %.1f ms" % (1e3 * dt), "sum steps", int(it.sum()), "median", int(np.median(it)), "max", int(it.max()), "not converged", int((st != 0).sum()),
          "| implied step time at 2048 resident waves: %.3f ms" % (1e3 * dt * 2048 / it.sum()), plan.describe()[:120])
    plan.close(); ctx.close()
    for k in env: del os.environ[k]
