"""Latency regime, batch of independent plants: k README-chain plans of one context executed on k HIP streams
(sls_plan_execute takes the caller's stream), so that the launches of different plants overlap on the CUs one plant leaves idle.
Prints subproblems/s for k = 1, 2, 4, 8 and the single-plan launch time beside it."""
import os, sys, time
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import slc_amd

name = sys.argv[1] if len(sys.argv) > 1 else "readme_chain"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
P, S, meta = slc_amd.workloads.make_workload(name)
ctx = slc_amd.Context([0])
for k in (1, 2, 4, 8):
    plans = [slc_amd.Plan(ctx, P, S) for _ in range(k)]
    vals = [p.alloc_values() for p in plans]
    streams = [torch.cuda.Stream() for _ in range(k)]
    for p, v, s in zip(plans, vals, streams): p.execute(v, stream=s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for p, v, s in zip(plans, vals, streams): p.execute(v, stream=s.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ok = all((p.fetch_status()[0] == 0).all() for p in plans)
    print(f"{name}: {k} plant(s) per round: {1e3*dt:.4f} ms per round, {k*P.Nx/dt:,.0f} subproblems/s, all solved {ok}")
    # the same through the C ABI's batch entry (plan-owned streams, fork/join on the null stream)
    for label, st in (("null stream", None), ("caller stream", streams[0].cuda_stream)):
        slc_amd.execute_batch(plans, vals, stream=st); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): slc_amd.execute_batch(plans, vals, stream=st)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name}: {k} plant(s) per sls_plan_execute_batch on the {label}: {1e3*dt:.4f} ms per call (host enqueue {1e3*(t1-t0)/reps:.4f} ms), {k*P.Nx/dt:,.0f} subproblems/s")
    for p in plans: p.close()
ctx.close()
