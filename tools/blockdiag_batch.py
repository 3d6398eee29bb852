import os, sys, time
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, "/root/repo")
import numpy as np, scipy.sparse as sp, torch
import slc_amd
P, S, meta = slc_amd.workloads.make_workload("readme_chain")
ctx = slc_amd.Context([0])
for k in (1, 2, 4, 8, 16):
    A = sp.block_diag([P.A] * k, format="csc"); B1 = sp.block_diag([P.B1] * k, format="csc"); B2 = sp.block_diag([P.B2] * k, format="csc")
    Pk = slc_amd.Plant(A, B1, B2)
    Sk = [[sp.block_diag([m] * k, format="csc") for m in S[0]], [sp.block_diag([m] * k, format="csc") for m in S[1]]]
    plan = slc_amd.Plan(ctx, Pk, Sk)
    v = plan.alloc_values()
    plan.execute(v); plan.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): plan.execute(v)
    plan.synchronize()
    dt = (time.perf_counter() - t0) / 300
    print(k, "plants block-diagonal:", plan.describe(), "%.4f ms" % (1e3 * dt), "%.0f subproblems/s" % (k * P.Nx / dt), (plan.fetch_status()[0] == 0).all())
    plan.close()
