"""Randomized COUPLED groups (B1 with off-diagonal entries inside the groups: the joint Hessian (RRᵀ)⊗G of DESIGN §3.3) against the
NumPy oracle's joint solve: random plant size, actuation step, diagonal or banded cost weights, D11, horizon, 1-based indices on odd
seeds; one-shot call.  Groups whose reference pairing is not the natural one are left out.  Prints mismatches per seed."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp, slc_amd as slc, sls_oracle as o
ctx = slc.Context([0])
for seed in [int(x) for x in sys.argv[1:]] or range(1, 9):
    rng = np.random.default_rng(900 + seed)
    Nx = int(rng.integers(24, 60))
    A = sp.random(Nx, Nx, density=rng.uniform(0.04, 0.09), random_state=seed, format="csc") * 0.5 + sp.eye(Nx, format="csc")
    B2 = sp.eye(Nx, format="csc")[:, ::int(rng.integers(1, 3))]
    Nu = B2.shape[1]; Nz = Nx + Nu
    cols = rng.permutation(Nx)[: min(Nx, 24)]
    groups, k = [], 0
    while k < len(cols):
        sz = int(rng.integers(1, 6)); groups.append(sorted(int(c) for c in cols[k:k + sz])); k += sz
    B1 = sp.lil_matrix(sp.diags(rng.uniform(0.6, 1.4, Nx)))
    for g in groups:
        for a in g:
            for b in g:
                if a != b and rng.uniform() < 0.5: B1[a, b] = rng.uniform(-0.5, 0.5)
    B1 = B1.tocsc()
    dense_w = seed % 3 == 0
    if dense_w:
        W = sp.csc_matrix(sp.diags(rng.uniform(0.8, 1.6, Nz)) + sp.diags(rng.uniform(-0.3, 0.3, Nz - 1), 1) + sp.diags(rng.uniform(-0.3, 0.3, Nz - 2), -2))
        C1, D12 = W[:, :Nx], W[:, Nx:]
    else:
        C1 = sp.vstack([sp.diags(rng.uniform(0.5, 2.0, Nx)), sp.csc_matrix((Nu, Nx))]).tocsc()
        D12 = sp.vstack([sp.csc_matrix((Nx, Nu)), sp.diags(rng.uniform(0.5, 2.0, Nu))]).tocsc()
    D11 = sp.random(Nz, Nx, density=0.05, random_state=seed + 7, format="csc") * 0.2
    P = slc.Plant(A, B1, B2, C1, D11, D12)
    d = int(rng.integers(1, 4)); T = int(rng.integers(4, 10))
    S = list(slc.workloads.localization_masks(P.A, P.B2, d, T, 1.5))
    Po = o.OraclePlant(P.A, P.B1, P.B2, P.C1, P.D11, P.D12)
    groups = [g for g in groups if [int(v) for v in o.sparsity_dim_reduction(Po, g, S)[3] if int(v) in g] == g]
    try:
        Px, Pu, info = slc.SLS_H2(P, S, groups, ctx=ctx, return_info=True, dropzeros=False, index_base=seed % 2)
    except Exception as e:
        print(seed, dict(Nx=Nx, Nu=Nu, d=d, T=T, groups=len(groups), dense_w=dense_w), "CALL FAILED", str(e)[:200]); continue
    ox, ou, dg = o.SLS_H2(Po, S, groups, return_diag=True)
    st = info["col_status"]; k = 0; bad = []; nok = 0
    for g, dd in zip(groups, dg):
        stg = st[k:k + len(g)]; k += len(g)
        if 1e-12 < dd["resid"] < 1e-6: continue
        feas = dd["resid"] <= 1e-12
        if feas != bool(np.all(stg == 0)): bad.append((g, stg.tolist(), float("%.0e" % dd["resid"]))); continue
        if feas:
            nok += 1
            err = max(max(abs(X[:, c] - O[:, c]).max() for X, O in zip(Px, ox)) for c in g)
            err = max(err, max(max(abs(U[:, c] - O[:, c]).max() for U, O in zip(Pu, ou)) for c in g))
            if err > 1e-7: bad.append((g, "err %.0e" % err))
    print(seed, dict(Nx=Nx, Nu=Nu, d=d, T=T, groups=len(groups), dense_w=dense_w), "feasible groups", nok, "MISMATCHES" if bad else "ok", bad[:5], flush=True)
ctx.close()
