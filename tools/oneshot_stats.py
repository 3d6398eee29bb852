"""Timing split of the one-shot drop-in call sls_h2_sf_solve (symbolic pass + H2D + solve + D2H per call)."""
import os, sys, time
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slc_amd
for name in sys.argv[1:] or ["readme_chain", "chain1024", "chain4096"]:
    P, S, meta = slc_amd.workloads.make_workload(name)
    ctx = slc_amd.Context([0])
    slc_amd.SLS_H2(P, S, ctx=ctx)                     # warm-up (module load, allocator)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        Px, Pu, info = slc_amd.SLS_H2(P, S, ctx=ctx, return_info=True)
        wall = time.perf_counter() - t0
        if best is None or wall < best[0]:
            best = (wall, info)
    wall, info = best
    print(f"{name}: wall {1e3 * wall:.2f} ms (python marshalling + assemble included) | symbolic {1e3 * info['t_symbolic_s']:.2f} ms, "
          f"upload {1e3 * info['t_upload_s']:.2f} ms, solve {1e3 * info['t_solve_s']:.2f} ms, download {1e3 * info['t_download_s']:.2f} ms "
          f"| {info['n_subproblems']} subproblems, unsolved {info['n_unsolved']}")
    ctx.close()
