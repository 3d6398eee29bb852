import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, slc_amd as slc
wl = slc.workloads
P = wl.chain_plant(65536)
ctx = slc.Context([0])
t0=time.perf_counter(); hx, hu = wl.localization_masks_native(P.A, P.B2, 12, 40, 1.5); t1=time.perf_counter()
dx, du = wl.localization_masks_native(P.A, P.B2, 12, 40, 1.5, ctx=ctx); t2=time.perf_counter()
ok = all(np.array_equal(H.indptr, D.indptr) and np.array_equal(H.indices, D.indices) for H, D in zip(hx + hu, dx + du))
print("Nx 65536: host %.0f ms, device %.0f ms, identical %s, entries %d" % (1e3*(t1-t0), 1e3*(t2-t1), ok, sum(M.nnz for M in dx+du)))
P = wl.random_plant(10000, 4, 2, 1)
hx, hu = wl.localization_masks_native(P.A, P.B2, 2, 25, 1.5)
dx, du = wl.localization_masks_native(P.A, P.B2, 2, 25, 1.5, ctx=ctx)
ok = all(np.array_equal(H.indptr, D.indptr) and np.array_equal(H.indices, D.indices) for H, D in zip(hx + hu, dx + du))
print("random10000_d2: identical", ok, sum(M.nnz for M in dx+du))
