"""Randomized comparison of the HIP path with the NumPy oracle over plant shapes the fixed tests do not cover: irregular sparse A
(not banded), partial actuation, diagonal weights, D11, random single-column selections; every kernel routing (default, one-wave only,
round-1 general kernel, tile kernel for everything).  Prints per-configuration mismatches: status vs oracle feasibility (columns whose
oracle residual lies between 1e-14 and 1e-6 are 'marginal' — feasible only just, or infeasible only just: DESIGN §2 — and skipped) and the value error of feasible columns."""
import os, sys
os.environ.setdefault("SLS_LAB", "1")      # diagnostic knobs are honoured in lab mode only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, scipy.sparse as sp, slc_amd as slc, sls_oracle as o

def problem(seed):
    rng = np.random.default_rng(seed)
    Nx = int(rng.integers(24, 90))
    dens = rng.uniform(0.02, 0.08)
    A = sp.random(Nx, Nx, density=dens, random_state=seed, format="csc") * 0.5 + sp.eye(Nx, format="csc")
    step = int(rng.integers(1, 4))
    B2 = sp.eye(Nx, format="csc")[:, ::step] * rng.uniform(0.5, 1.5)
    Nu = B2.shape[1]
    weighted = seed % 2 == 0
    if weighted:
        q = rng.uniform(0.5, 2.0, Nx); r = rng.uniform(0.5, 2.0, Nu)
        C1 = sp.vstack([sp.diags(q), sp.csc_matrix((Nu, Nx))]).tocsc()
        D12 = sp.vstack([sp.csc_matrix((Nx, Nu)), sp.diags(r)]).tocsc()
        D11 = sp.random(Nx + Nu, Nx, density=0.03, random_state=seed + 1, format="csc") * 0.2
        B1 = sp.diags(rng.uniform(0.6, 1.4, Nx)).tocsc()
        P = slc.Plant(A, B1, B2, C1, D11, D12)
    else:
        P = slc.Plant(A, sp.eye(Nx, format="csc"), B2)
    d = int(rng.integers(1, 4)); T = int(rng.integers(4, 12))
    S = list(slc.workloads.localization_masks(P.A, P.B2, d, T, 1.5))
    return P, S, dict(Nx=Nx, dens=round(dens, 3), step=step, d=d, T=T, weighted=weighted)

modes = {"default": {}, "one-wave": {"SLS_NO_TWISTED": "1"}, "tile-all": {"SLS_TILE": "all", "SLS_FORCE_GENERAL": "1"}}
seeds = [int(x) for x in sys.argv[1:]] or list(range(1, 13))
for seed in seeds:
    P, S, meta = problem(seed)
    Po = o.OraclePlant(P.A, P.B1, P.B2, P.C1, P.D11, P.D12)
    ox, ou, dg = o.SLS_H2(Po, S, return_diag=True)
    res = np.array([d_["resid"] for d_ in dg]); ns = np.array([d_["n"] for d_ in dg])
    for mode, env in modes.items():
        for k, v in env.items(): os.environ[k] = v
        ctx = slc.Context([0])
        Px, Pu, info = slc.SLS_H2(P, S, ctx=ctx, return_info=True, dropzeros=False, index_base=seed % 2)
        ctx.close()
        for k in env: del os.environ[k]
        st = info["col_status"]
        bad = []
        for c in range(P.Nx):
            if 1e-14 < res[c] < 1e-6: continue
            feas = res[c] <= 1e-14
            if (st[c] == 0) != feas and not (st[c] == 3 and feas):
                bad.append((c, int(ns[c]), int(st[c]), float("%.0e" % res[c])))
            elif feas:
                err = max(max(abs(X[:, c] - O[:, c]).max() for X, O in zip(Px, ox)), max(abs(U[:, c] - O[:, c]).max() for U, O in zip(Pu, ou)))
                # an iterative solve is good to residual/σ_min: 1e-7, or 3× the 1e-12 residual target over the smallest singular value kept
                if err > max(1e-7, 3e-12 / dg[c]["smin"]): bad.append((c, int(ns[c]), "err %.0e" % err, float("%.0e" % res[c])))
        print(seed, meta, "max n", int(ns.max()), mode, "feasible", int((res <= 1e-14).sum()), "MISMATCHES" if bad else "ok", bad[:6])
