"""Diagnostics: pivot blocks P_k of one column from the four-wave and the two-wave twisted kernels, block by block (T = 4 plants:
both kernels put the middle at block 1, so every block is comparable).  usage: t4_dump_P.py seed col"""
import os, sys, ctypes as C
os.environ["SLS_LAB"] = "1"; os.environ["SLS_T4_NMIN"] = "0"; os.environ["SLS_T4_TMIN"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
src = open(os.path.join(ROOT, "tools", "fuzz_h2.py")).read().split("modes = {")[0]
ns = {"__file__": os.path.join(ROOT, "tools", "fuzz_h2.py")}
exec(compile(src, "fuzz_h2.py", "exec"), ns)
import numpy as np, slc_amd as slc
seed, col = int(sys.argv[1]), int(sys.argv[2])
P, S, meta = ns["problem"](seed)
T = meta["T"]
ctx = slc.Context([0])
full = slc.Plan(ctx, P, S)
nbig = full.info["max_nx"]; full.close()
# a partner with the largest index set forces the launch's 32-lane class
sizes = [int(sum(M[:, c].nnz for M in S[0])) for c in range(P.Nx)]
partner = int(np.argmax(sizes))
res = {}
for mode in ("1", "0"):
    os.environ["SLS_TWISTED4"] = mode
    plan = slc.Plan(ctx, P, S, [[col], [partner]])
    desc = plan.describe()
    d = plan.alloc_values(); plan.execute(d); plan.synchronize()
    st, rs, it = plan.fetch_status()
    cls = desc.split("<")[1].split(">")[0].split(",")
    RPL = int(cls[1]); stride = (T + 1) * RPL * 64
    lib = ctx._lib
    lib.sls_plan_debug_read_workspace.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_double)]
    bufs = []
    for b in range(2):
        buf = np.zeros(stride)
        assert lib.sls_plan_debug_read_workspace(plan.handle, b * stride, stride, buf.ctypes.data_as(C.POINTER(C.c_double))) == 0
        bufs.append(buf.reshape(T + 1, RPL, 2, 32))
    res[mode] = (desc, st.copy(), rs.copy(), it.copy(), bufs, RPL)
    plan.close()
print(meta, "col", col, "partner", partner)
for mode in ("1", "0"):
    print(" ", res[mode][0].split(" ")[0], "status", res[mode][1].tolist(), ["%.1e" % r for r in res[mode][2]], res[mode][3].tolist())
RPL = res["1"][5]
for b in range(2):
    for k in range(T + 1):
        A4 = res["1"][4][b][k]; A2 = res["0"][4][b][k]
        # image (r, h, j) -> row 2r+h, col j
        M4 = A4.transpose(0, 1, 2).reshape(2 * RPL, 32); M2 = A2.reshape(2 * RPL, 32)
        dif = np.abs(M4 - M2); sc = max(np.abs(M2).max(), 1e-300)
        i, j = np.unravel_index(np.argmax(dif), dif.shape)
        print(f"  workgroup {b} block {k}: max|P4-P2| {dif.max():.2e} (rel {dif.max()/sc:.1e}) at ({i},{j})  |P2|max {np.abs(M2).max():.2e}  diag4 {M4[i,i] if i<32 else 0:.3e} diag2 {M2[i,i] if i<32 else 0:.3e}")

# ---- the downward recursion in NumPy for the small column (workgroup 1 unless it is the costlier one) ----
import sls_oracle as o
Po = o.OraclePlant(P.A, P.B1, P.B2, P.C1, P.D11, P.D12)
for b, cc in ((1, col), (0, partner)):
    sub, It, iix, sx, su = o.sparsity_dim_reduction(Po, [cc], S)
    srt = np.argsort(sx); sxs = sx[srt]; sus = np.sort(su)          # the kernels keep the index sets ascending
    n, m = len(sxs), len(sus)
    A = P.A[sxs][:, sxs].toarray(); B = P.B2[sxs][:, sus].toarray()
    mx = np.array([np.asarray(S[0][t][sxs, cc].todense()).ravel() != 0 for t in range(T)], dtype=float)
    mu = np.array([np.asarray(S[1][t][sus, cc].todense()).ravel() != 0 for t in range(T)], dtype=float)
    sc = (1.0 + (A ** 2).sum(1) + (B ** 2).sum(1)).max(); delta = 1e-12 * sc
    W = lambda k: np.diag(mx[k]) if 0 <= k <= T - 1 else np.zeros((n, n))
    Wu = lambda k: np.diag(mu[k])
    D = lambda k: delta * np.eye(n) + W(k) + (A @ W(k - 1) @ A.T + B @ Wu(k - 1) @ B.T if k >= 1 else 0)
    c = 1 if T < 7 else (T + 1) // 2
    Pd = {}
    Sk = D(T); Pd[T] = np.linalg.inv(Sk)
    for k in range(T - 1, c, -1):
        X = W(k) @ A.T
        Sk = D(k) - X @ Pd[k + 1] @ X.T
        Pd[k] = np.linalg.inv(Sk)
    print(f"workgroup {b} (column {cc}, n = {n}, m = {m}): downward blocks against the NumPy recursion")
    for k in range(T, c, -1):
        for mode in ("1", "0"):
            img = res[mode][4][b][k].reshape(2 * RPL, 32)[:n, :n]
            ref = Pd[k]
            print(f"   block {k} {'four-wave' if mode == '1' else 'two-wave '}: max|P − P_numpy| / max|P_numpy| = {np.abs(img - ref).max() / np.abs(ref).max():.2e}   (|P_numpy|max {np.abs(ref).max():.2e}, smallest eigenvalue of S_k {np.linalg.eigvalsh((np.linalg.inv(ref) + np.linalg.inv(ref).T) / 2).min():.2e})")
    # what is wrong with the four-wave block c+1?  S4 − S_numpy, its dominant entries and rank
    k = c + 1
    img = res["1"][4][b][k].reshape(2 * RPL, 32)[:n, :n]
    S4 = np.linalg.inv(img); S_np = np.linalg.inv(Pd[k])
    E = S4 - S_np
    u, sv, vt = np.linalg.svd(E)
    print(f"   block {k}: S_fourwave − S_numpy: max {np.abs(E).max():.2e}, singular values {sv[:4]}")
    idx = np.argsort(-np.abs(E).ravel())[:6]
    print("      largest entries:", [(int(i // n), int(i % n), float('%.2e' % E.ravel()[i])) for i in idx])
    print("      inactive rows of block", k + 1, ":", np.flatnonzero(np.diag(np.linalg.inv(Pd[k + 1])) < 10 * delta).tolist(), " of block", k, ":", np.flatnonzero(np.diag(S_np) < 10 * delta).tolist(), " W_k diag", mx[k].astype(int).tolist() if k < T else None)
    Xk = W(k) @ A.T
    print("      columns of X_k with nonzeros:", np.flatnonzero(np.abs(Xk).sum(0) > 0).tolist(), " rows:", np.flatnonzero(np.abs(Xk).sum(1) > 0).tolist())
