"""Randomized plants with NON-DIAGONAL cost weights [C1 D12] (banded + random couplings), D11, diagonal B1, random multi-column groups
(decoupled: B1 diagonal) against the NumPy oracle's joint solve of each group; one-shot call.  Prints mismatches per seed."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp, slc_amd as slc, sls_oracle as o
ctx = slc.Context([0])
for seed in [int(x) for x in sys.argv[1:]] or range(1, 9):
    rng = np.random.default_rng(500 + seed)
    Nx = int(rng.integers(20, 50))
    A = sp.random(Nx, Nx, density=rng.uniform(0.03, 0.08), random_state=seed, format="csc") * 0.5 + sp.eye(Nx, format="csc")
    B2 = sp.eye(Nx, format="csc")[:, ::int(rng.integers(1, 3))]
    Nu = B2.shape[1]; Nz = Nx + Nu
    W = sp.lil_matrix(sp.diags(rng.uniform(0.8, 1.6, Nz)) + sp.diags(rng.uniform(-0.3, 0.3, Nz - 1), 1) + sp.diags(rng.uniform(-0.3, 0.3, Nz - 2), -2))
    for _ in range(5): W[int(rng.integers(Nx)), Nx + int(rng.integers(Nu))] = rng.uniform(-0.4, 0.4)
    W = sp.csc_matrix(W)
    B1 = sp.diags(rng.uniform(0.5, 1.5, Nx)).tocsc()
    D11 = sp.random(Nz, Nx, density=0.05, random_state=seed + 3, format="csc") * 0.3
    P = slc.Plant(A, B1, B2, W[:, :Nx], D11, W[:, Nx:])
    d = int(rng.integers(1, 4)); T = int(rng.integers(4, 10))
    S = list(slc.workloads.localization_masks(P.A, P.B2, d, T, 1.5))
    cols = rng.permutation(Nx)[: min(Nx, 18)]
    groups, k = [], 0
    while k < len(cols):
        sz = int(rng.integers(1, 4)); groups.append(sorted(int(c) for c in cols[k:k + sz])); k += sz
    Po = o.OraclePlant(P.A, P.B1, P.B2, P.C1, P.D11, P.D12)
    # keep groups whose reference pairing is the natural one
    groups = [g for g in groups if [int(v) for v in o.sparsity_dim_reduction(Po, g, S)[3] if int(v) in g] == g]
    Px, Pu, info = slc.SLS_H2(P, S, groups, ctx=ctx, return_info=True, dropzeros=False, index_base=seed % 2)
    ox, ou, dg = o.SLS_H2(Po, S, groups, return_diag=True)
    st = info["col_status"]; k = 0; bad = []; nok = 0
    for g, dd in zip(groups, dg):
        stg = st[k:k + len(g)]; k += len(g)
        if 1e-14 < dd["resid"] < 1e-6: continue
        feas = dd["resid"] <= 1e-14
        if feas != bool(np.all(stg == 0)): bad.append((g, stg.tolist(), float("%.0e" % dd["resid"]))); continue
        if feas:
            nok += 1
            err = max(max(abs(X[:, c] - O[:, c]).max() for X, O in zip(Px, ox)) for c in g)
            err = max(err, max(max(abs(U[:, c] - O[:, c]).max() for U, O in zip(Pu, ou)) for c in g))
            if err > 1e-7: bad.append((g, "err %.0e" % err))
    print(seed, dict(Nx=Nx, Nu=Nu, d=d, T=T, groups=len(groups)), "feasible groups", nok, "MISMATCHES" if bad else "ok", bad[:5])
ctx.close()
