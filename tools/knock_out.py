"""Diagnostics: marginal cost of the phases of the one-wave kernel at full occupancy (SLS_KNOCK_OUT=n skips a phase; results of
such a run are meaningless, only its time is read).  usage: knock_out.py [workload]   — run once per value, the plan reads the
variable when it is created."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slc_amd
name = sys.argv[1] if len(sys.argv) > 1 else "chain4096"
P, S, meta = slc_amd.workloads.make_workload(name)
ctx = slc_amd.Context([0])
names = {0: "full", 1: "no Gauss-Jordan", 2: "no sparse products in the build", 3: "no P_k store", 4: "no backward sweep", 5: "no residual pass"}
for ko in (0, 1, 2, 3, 4, 5, 0):
    os.environ["SLS_KNOCK_OUT"] = str(ko)
    plan = slc_amd.Plan(ctx, P, S)
    d = plan.alloc_values()
    for _ in range(30):
        plan.execute(d)
    plan.synchronize()
    ms, n = plan.kernel_time_ms()
    print(f"{name} knock_out={ko} ({names[ko]}): kernel avg {ms:.4f} ms over {n} launches", flush=True)
    del plan
