"""Compare the tile kernel's stored pivot blocks −P_k (workspace dump, single-column plan) with a NumPy recursion (diagnostics)."""
import ctypes as C, os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import slc_amd, sls_oracle as o, sls_oracle_cport as cp
wl = slc_amd.workloads

name = sys.argv[1]; col = int(sys.argv[2])
if name == "grid16_d5":
    P = wl.grid_plant(16, 3); S = list(wl.localization_masks(P.A, P.B2, 5, 20, 1.5))
elif name == "chain_d40":
    P = wl.chain_plant(200); S = list(wl.localization_masks(P.A, P.B2, 40, 90, 1.5))
T = len(S[0])
os.environ["SLS_TILE"] = "all"; os.environ["SLS_TILE_GLOBAL"] = sys.argv[3] if len(sys.argv) > 3 else "1"
os.environ["SLS_MAX_ITERS"] = "1"
ctx = slc_amd.Context([0]); plan = slc_amd.Plan(ctx, P, S, [[col]])
print(plan.describe())
d = plan.alloc_values(); plan.execute(d); plan.synchronize()
st, rs, it = plan.fetch_status(); print("status", st, rs, it)
rec = cp.prepare(o.OraclePlant(P.A, P.B1, P.B2), S, [col])[0]
n, m, A, B, mask = rec["n"], rec["m"], rec["A"], rec["B"], rec["mask"].astype(float)
NT = (n + 15) // 16; HT = NT * (NT + 1) // 2; npad = 16 * NT
buf = np.zeros((T + 1) * HT * 256)
lib = ctx._lib
lib.sls_plan_debug_read_workspace.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double)]
assert lib.sls_plan_debug_read_workspace(plan.handle, 0, buf.size, buf.ctypes.data_as(C.POINTER(C.c_double))) == 0
sc = (1 + (A ** 2).sum(1) + (B ** 2).sum(1)).max(); delta = 1e-12 * sc
Pprev = None
for k in range(T + 1):
    Wc = mask[k, :n] if k < T else np.zeros(n)
    D = delta * np.eye(n) + np.diag(Wc)
    if k >= 1:
        W = np.diag(mask[k - 1, :n]); Wu = np.diag(mask[k - 1, n:])
        D = D + B @ Wu @ B.T + A @ (W - W @ Pprev @ W) @ A.T
    Pk = np.linalg.inv(D)
    got = np.zeros((npad, npad)); t = 0
    for I in range(NT):
        for J in range(I, NT):
            tile = -buf[(k * HT + t) * 256:(k * HT + t + 1) * 256].reshape(16, 16)
            got[16 * I:16 * I + 16, 16 * J:16 * J + 16] = tile
            if I != J: got[16 * J:16 * J + 16, 16 * I:16 * I + 16] = tile.T
            t += 1
    err = np.abs(got[:n, :n] - Pk).max() / np.abs(Pk).max()
    act = np.diag(D) > 1e-6
    sub = np.ix_(act, act)
    erra = np.abs(got[:n, :n][sub] - Pk[sub]).max() / np.abs(Pk[sub]).max()
    # one-step check: the same recursion fed with the GPU's own P_{k-1}
    print(f"k={k:2d} relerr {err:.2e} active({act.sum()}) relerr {erra:.2e} |Pact|max {np.abs(Pk[sub]).max():.2e} |P|max {np.abs(Pk).max():.2e} mineig {np.linalg.eigvalsh((D+D.T)/2).min():.2e}")
    Pprev = got[:n, :n].copy() if os.environ.get("FEED_GPU") else Pk
