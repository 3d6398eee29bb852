"""Time one workload's resident solve (plan + a few executes): launch list, ms per pass, status / pass histograms and,
with SLS_PHASE_TIMERS=1, the per-phase cycle shares of the workgroup-level kernels (diagnostics)."""
import ctypes as C, os, sys, time
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slc_amd

name = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
P, S, meta = slc_amd.workloads.make_workload(name)
ctx = slc_amd.Context([0])
obj = os.environ.get('SLS_OBJECTIVE', 'h2')
t0 = time.perf_counter(); plan = slc_amd.Plan(ctx, P, S, objective=obj); t1 = time.perf_counter()
print(f"{name}: plan {1e3*(t1-t0):.1f} ms  workspace {plan.info['workspace_bytes']/2**30:.2f} GiB  max_nx {plan.info['max_nx']} max_nu {plan.info['max_nu']}")
for ln in plan.describe().split(";"):
    if ln: print("   ", ln)
d = plan.alloc_values()
plan.execute(d); plan.synchronize(); plan.kernel_time_ms()
t0 = time.perf_counter()
for _ in range(reps): plan.execute(d)
plan.synchronize(); wall = (time.perf_counter() - t0) / reps
ms, nl = plan.kernel_time_ms()
st, rs, it = plan.fetch_status()
print(f"  {1e3*wall:.3f} ms per pass (events {ms:.3f} ms), {P.Nx/wall:.0f} subproblems/s, F_alg {plan.info['flops_alg']/wall/1e12:.2f} TFLOP/s")
print("  status histogram", np.bincount(st, minlength=6).tolist(), " passes histogram", (np.bincount(it).tolist() if it.max() < 40 else "min %d median %d max %d" % (it.min(), np.median(it), it.max())),
      " max resid (ok)", rs[st == 0].max() if (st == 0).any() else None)
if os.environ.get("SLS_PHASE_TIMERS"):
    lib = ctx._lib
    buf = np.zeros(len(st) * 8, dtype=np.uint64)
    lib.sls_plan_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.sls_plan_debug_phase_cycles(plan.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)))
    ph = buf.reshape(-1, 8).astype(float)
    tot = ph.sum(1)
    import scipy.sparse as sp
    nx = np.diff(((S[0][-1].astype(np.int32)) @ (P.A != 0).astype(np.int32)).tocsc().indptr)
    lo = int(os.environ.get("PH_NMIN", "65")); hi = int(os.environ.get("PH_NMAX", "100000"))
    big = (tot > 0) & (nx >= lo) & (nx <= hi)
    print(f"  ({big.sum()} subproblems with {lo} <= nx <= {hi})")
    names = ["setup", "residual", "build1", "invert", "store", "sweeps", "build2", "buildB"]
    print("  phase shares over subproblems with counters (cycles of s_memtime, 100 MHz):")
    for k in range(8): print(f"    {names[k]:9s} {ph[big, k].sum() / tot[big].sum():6.1%}   mean {ph[big, k].mean():12.0f}")
    print(f"    total mean {tot[big].mean():.0f}  max {tot.max():.0f}")
plan.close(); ctx.close()
