"""Phase timers of the sum-of-norms build of the one-wave kernel on chain-4096 (SLS_PHASE_TIMERS=1): per ADMM step, the cycles of the
substitution sweeps, the residual passes, the threshold + Anderson block, and the multiplier passes per projection."""
import ctypes as C, os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
os.environ["SLS_PHASE_TIMERS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, slc_amd as slc
P, S, meta = slc.workloads.make_workload(sys.argv[1] if len(sys.argv) > 1 else "chain4096")
for env in ({}, {"SLS_SON_ANDERSON": "0"}):
    os.environ.update(env)
    ctx = slc.Context([0])
    plan = slc.Plan(ctx, P, S, objective="sum_of_norms")
    dv = plan.alloc_values()
    plan.execute(dv); plan.synchronize()
    lib = ctx._lib
    lib.sls_plan_debug_phase_cycles.restype = C.c_int; lib.sls_plan_debug_phase_cycles.argtypes = [C.c_void_p, C.c_void_p]
    ns = plan.info["n_subproblems"]
    buf = np.zeros(ns * 8, dtype=np.uint64)
    assert lib.sls_plan_debug_phase_cycles(plan.handle, buf.ctypes.data) == 0
    b = buf.reshape(ns, 8).astype(np.float64)
    st, rs, it = plan.fetch_status()
    steps = it.sum()
    print(env, "steps", int(steps), "| per step (10 ns ticks): sweeps %.0f, residual %.0f, threshold+AA %.0f | multiplier passes per projection %.2f"
          % (b[:, 5].sum() / steps, b[:, 1].sum() / steps, b[:, 6].sum() / steps, b[:, 7].sum() / steps))
    plan.close(); ctx.close()
    for k in env: del os.environ[k]
