"""Four-wave twisted kernel (sls_twisted4_kernel.hip) against the golden README vector and the two-wave kernel: values,
statuses, time per resident launch.  Usage: python tools/t4_check.py [reps]"""
import os, sys, time
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slc_amd

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = np.load(os.path.join(ROOT, "tests", "golden", "readme_chain_phi.npz"))
want = np.concatenate([g["vals_x"], g["vals_u"]])
P, S, _ = slc_amd.workloads.make_workload("readme_chain")
for mode in ("1", "0"):
    os.environ["SLS_TWISTED4"] = mode
    ctx = slc_amd.Context([0])
    plan = slc_amd.Plan(ctx, P, S)
    print("SLS_TWISTED4=" + mode, plan.describe(), flush=True)
    d = plan.alloc_values()
    plan.execute(d); plan.synchronize()
    got = np.concatenate(sum(plan.download(d), []))
    st, rs, it = plan.fetch_status()
    print("  max|ΔΦ| vs golden %.3e  status %s  resid max %.2e  iters %s" % (np.abs(got - want).max(), np.unique(st), rs.max(), np.unique(it)), flush=True)
    for _ in range(20):
        plan.execute(d)
    plan.synchronize(); plan.kernel_time_ms()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.execute(d)
    plan.synchronize()
    wall = (time.perf_counter() - t0) / reps
    km, nl = plan.kernel_time_ms()
    print("  wall per launch %.4f ms, kernel (HIP events) %.4f ms over %d" % (1e3 * wall, km, nl), flush=True)
    plan.close(); ctx.close()
