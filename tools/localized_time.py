"""One-shot drop-in call: mask route (sls_h2_sf_solve, masks handed over as 2T host arrays) against the device-resident route
(sls_h2_sf_solve_localized, plan built from (A, B2, d, α, T) on the device).  Library-internal times (sls_stats).
Usage: python tools/localized_time.py [workload|chainN] ..."""
import ctypes as C, os, sys, time
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slc_amd
from slc_amd import _capi

def lib_ms(st):
    return 1e3 * (st["t_symbolic_s"] + st["t_upload_s"] + st["t_solve_s"] + st["t_download_s"])

def localized_raw(ctx, P, d, T, alpha, nnz_x, nnz_u):
    """the C call alone (no pattern download, no sparse assembly): what a device-side consumer pays"""
    m = _capi.Marshalled(P, [], [], None); m.dims.T = T
    vx = [np.zeros(max(n, 1)) for n in nnz_x]; vu = [np.zeros(max(n, 1)) for n in nnz_u]
    dp = C.POINTER(C.c_double)
    px = (dp * T)(*[a.ctypes.data_as(dp) for a in vx]); pu = (dp * T)(*[a.ctypes.data_as(dp) for a in vu])
    st = _capi.sls_stats()
    t0 = time.perf_counter()
    rc = ctx._lib.sls_h2_sf_solve_localized(ctx.handle, C.byref(m.dims), C.byref(m.plant), d, alpha, px, pu, None, C.byref(st))
    wall = time.perf_counter() - t0
    _capi.check(rc, ctx.handle)
    return st.asdict(), wall, rc

for name in (sys.argv[1:] or ["readme_chain", "chain4096"]):
    if name.startswith("chain") and name[5:].isdigit() and name not in slc_amd.workloads.WORKLOADS:
        P = slc_amd.workloads.chain_plant(int(name[5:])); d, T, alpha = 12, 40, 1.5
    else:
        fac, d, T, alpha = slc_amd.workloads.WORKLOADS[name]; P = fac()
    ctx = slc_amd.Context([0])
    t0 = time.perf_counter()
    S = list(slc_amd.workloads.localization_masks_native(P.A, P.B2, d, T, alpha))
    t_masks = time.perf_counter() - t0
    nnz_x = [int(M.nnz) for M in S[0]]; nnz_u = [int(M.nnz) for M in S[1]]
    best_m = best_l = None
    for rep in range(4):
        _, _, im = slc_amd.SLS_H2(P, S, ctx=ctx, return_info=True)
        il, wall, rc = localized_raw(ctx, P, d, T, alpha, nnz_x, nnz_u)
        if rep and (best_m is None or lib_ms(im) < lib_ms(best_m)): best_m = im
        if rep and (best_l is None or lib_ms(il) < lib_ms(best_l[0])): best_l = (il, wall)
    f = lambda st: "total %.3f ms = symbolic %.3f + upload %.3f + solve %.3f + download %.3f" % (lib_ms(st), 1e3 * st["t_symbolic_s"], 1e3 * st["t_upload_s"], 1e3 * st["t_solve_s"], 1e3 * st["t_download_s"])
    print(f"{name}: Nx={P.Nx} values={sum(nnz_x) + sum(nnz_u)}  (host mask recipe on its own: {1e3 * t_masks:.1f} ms)")
    print("  mask route      ", f(best_m), " unsolved", best_m["n_unsolved"])
    print("  device-resident ", f(best_l[0]), " wall of the C call %.3f ms" % (1e3 * best_l[1]))
    ctx.close()
