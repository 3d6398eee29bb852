"""Mask recipe (README.md:52-54): SciPy Boolean powers vs the library's host level-set pass vs the device kernels (sls_masks.hip)."""
import os, sys, time
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import slc_amd
wl = slc_amd.workloads
ctx = slc_amd.Context([0])
for name in sys.argv[1:] or ["readme_chain", "chain4096", "grid32", "random10000_d2"]:
    mk, d, T, alpha = wl.WORKLOADS[name]
    P = mk()
    best = {}
    for label, fn in (("host", lambda: wl.localization_masks_native(P.A, P.B2, d, T, alpha)),
                      ("device", lambda: wl.localization_masks_native(P.A, P.B2, d, T, alpha, ctx=ctx))):
        fn()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); S = fn(); ts.append(time.perf_counter() - t0)
        best[label] = min(ts)
    nnz = sum(M.nnz for M in S[0] + S[1])
    print(f"{name}: {nnz} mask entries; host recipe {1e3*best['host']:.1f} ms, device recipe {1e3*best['device']:.1f} ms (both through the Python wrapper: two calls + SciPy assembly)")
ctx.close()
