"""Scaling sanity: a chain of Nx states (default 65536), d = 12, T = 40 — plan, resident passes, one-shot call, full-system achievability
of the result (Φx[t+1] = AΦx[t] + B2Φu[t]) as the size-independent check."""
import os, sys, time
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, slc_amd as slc
Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
P = slc.workloads.chain_plant(Nx)
t0 = time.perf_counter(); S = list(slc.workloads.localization_masks_native(P.A, P.B2, 12, 40, 1.5)); t1 = time.perf_counter()
print(f"Nx {Nx}: masks {sum(M.nnz for M in S[0] + S[1])} entries in {1e3*(t1-t0):.0f} ms")
ctx = slc.Context([0])
plan = slc.Plan(ctx, P, S)
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
t0 = time.perf_counter()
for _ in range(3): plan.execute(d)
plan.synchronize(); dt = (time.perf_counter() - t0) / 3
st, rs, it = plan.fetch_status()
print(f"  resident pass {1e3*dt:.2f} ms, {Nx/dt:,.0f} subproblems/s, workspace {plan.info['workspace_bytes']/2**30:.2f} GiB, status histogram {np.bincount(st).tolist()}, max resid {rs.max():.1e}")
plan.close()
t0 = time.perf_counter(); Px, Pu, info = slc.SLS_H2(P, S, ctx=ctx, return_info=True); t1 = time.perf_counter()
print(f"  one-shot: library {1e3*(info['t_symbolic_s']+info['t_upload_s']+info['t_solve_s']+info['t_download_s']):.1f} ms (symbolic {1e3*info['t_symbolic_s']:.1f}, upload {1e3*info['t_upload_s']:.1f}, solve {1e3*info['t_solve_s']:.1f}, download {1e3*info['t_download_s']:.1f}), wrapper wall {1e3*(t1-t0):.0f} ms")
A, B2 = P.A.tocsc(), P.B2.tocsc()
T = len(Px)
err = abs(Px[0] - slc.workloads.sp.identity(Nx, format="csc")).max()
for t in range(T - 1): err = max(err, abs(Px[t + 1] - A @ Px[t] - B2 @ Pu[t]).max())
err = max(err, abs(A @ Px[T - 1] + B2 @ Pu[T - 1]).max())
print(f"  achievability in the full system: max violation {err:.1e}")
