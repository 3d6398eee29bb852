"""Closed-loop simulator timing: device time of `steps` time steps (HIP events), per step, and the HBM rate of the model
12 B per stored Φ entry per step.  usage: closed_loop_bench.py [workload] [steps] [nscen]"""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, slc_amd
name = sys.argv[1] if len(sys.argv) > 1 else "chain4096"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 250
nscen = int(sys.argv[3]) if len(sys.argv) > 3 else 1
P, S, meta = slc_amd.workloads.make_workload(name)
ctx = slc_amd.Context([0]); plan = slc_amd.Plan(ctx, P, S)
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
loop = slc_amd.ClosedLoop(ctx, P, S)
dev = torch.device("cuda:0")
w = torch.zeros(steps, P.Nw, nscen, dtype=torch.float64, device=dev); w[9, P.Nx // 2, :] = 1.0
x = torch.empty(steps, P.Nx, nscen, dtype=torch.float64, device=dev); u = torch.empty(steps, P.Nu, nscen, dtype=torch.float64, device=dev)
ms = []
for rep in range(5):
    loop.run(d, w.data_ptr(), steps, nscen, x.data_ptr(), u.data_ptr())
    torch.cuda.synchronize(); ms.append(loop.last_ms())
best = min(ms[1:])
per_step = best / (steps - 1)
gbs = 12.0 * loop.n_entries / (per_step * 1e-3) / 1e9
print(f"{name}: Nx={P.Nx} entries={loop.n_entries} steps={steps} nscen={nscen} graph={'off' if os.environ.get('SLS_NO_GRAPH') else 'on'}: "
      f"first {ms[0]:.3f} ms, best {best:.3f} ms, {per_step*1e3:.2f} us/step, {gbs:.1f} GB/s (12 B/entry model), "
      f"{nscen*(steps-1)/best*1e3:.0f} scenario-steps/s; max|x| {float(x.abs().max()):.3f}")
