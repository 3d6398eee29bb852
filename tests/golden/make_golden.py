"""Generates the committed golden fixtures from the NumPy oracle (oracle/sls_oracle.py).

The reference itself cannot run in the build container (Julia/JuMP/Ipopt absent),
so these vectors are ORACLE outputs, certified by the solver-independent optimality
certificate (feasibility + projected gradient), not reference outputs; see DESIGN.md §2.

  readme_chain_phi.npz      Φ of the README chain (README.md:43-57) in mask-CSC order, per-column
                            costs, certificate maxima.
  reduction_known_answer.json   inputs/expected outputs of reference test/reduction_test.jl:11-24
                            (data only: index sets), 0-based.
  weighted_chain_phi.npz    a small diagonally weighted LQR chain with D11 ≠ 0 and B1 ≠ I
  grouped_chain_phi.npz     README chain solved with multi-column groups 𝓘 = [0:20, 20:40, 40:59]
  infeasible_chain.npz      Nx=23 chain with d=4,T=12: several columns have NO feasible localized
                            response; per-column oracle residuals + least-squares values
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import sls_oracle as o  # noqa: E402


def flat(vals):
    return np.concatenate([np.asarray(v, dtype=np.float64) for v in vals]) if len(vals) else np.zeros(0)


def main():
    # ---- README chain ----
    P = o.readme_chain()
    Sx, Su = o.readme_masks(P.A, P.B2, 9, 29, 1.5)
    Phix, Phiu, dg = o.SLS_H2(P, [Sx, Su], return_diag=True)
    feas = pg = 0.0
    for j in (0, 5, 10, 29, 58):
        z, info, d = o.solve_group(P, [j], Sx, Su)
        f_, p_ = o.certificate(d["E"], d["f"], d["M"], d["m0"], z)
        feas, pg = max(feas, f_), max(pg, p_)
    np.savez_compressed(os.path.join(HERE, "readme_chain_phi.npz"),
                        vals_x=flat(o.values_in_mask_order(Phix, Sx)), vals_u=flat(o.values_in_mask_order(Phiu, Su)),
                        col_cost=np.array([d["cost"] for d in dg]), col_n=np.array([d["n"] for d in dg]),
                        col_m=np.array([d["m"] for d in dg]), col_nfree=np.array([d["nfree"] for d in dg]),
                        cert_feas=feas, cert_projgrad=pg)
    print("readme: total cost", sum(d["cost"] for d in dg), "cert", feas, pg)

    # ---- reduction known-answer (reference test/reduction_test.jl:11-24) ----
    Ab = (P.A != 0).astype(np.int64).tocsc()
    A9 = o._bool_power(Ab, 9)
    S1x = [(A9 != 0).tocsc()]
    S1u = [(((P.B2.T != 0).astype(np.int64)) @ A9 != 0).tocsc()]
    sub, It, iix, sx, su = o.sparsity_dim_reduction(P, np.arange(20), [S1x, S1u])
    json.dump(dict(source="reference test/reduction_test.jl:11-24 (expected values as written there, 0-based here)",
                   Nx=59, cj=list(range(20)), expected_sx=list(range(30)), expected_su=list(range(10)),
                   expected_iix=[1] * 20 + [0] * 10,
                   oracle_sx=[int(v) for v in sx], oracle_su=[int(v) for v in su], oracle_iix=[int(v) for v in iix]),
              open(os.path.join(HERE, "reduction_known_answer.json"), "w"), indent=1)

    # ---- weighted chain: diagonal LQR weights, D11 ≠ 0, B1 = diag(b) ----
    rng = np.random.default_rng(7)
    Nx = 23
    Pc = o.readme_chain(Nx)
    Nu = Pc.Nu
    q = rng.uniform(0.5, 2.0, Nx); r = rng.uniform(0.5, 2.0, Nu)
    C1 = sp.vstack([sp.diags(q), sp.csc_matrix((Nu, Nx))]).tocsc()
    D12 = sp.vstack([sp.csc_matrix((Nx, Nu)), sp.diags(r)]).tocsc()
    b = rng.uniform(0.5, 1.5, Nx)
    B1 = sp.diags(b).tocsc()
    D11 = sp.random(Nx + Nu, Nx, density=0.15, random_state=3, format="csc") * 0.3
    Pw = o.OraclePlant(Pc.A, B1, Pc.B2, C1, D11, D12)
    Sxw, Suw = o.readme_masks(Pw.A, Pw.B2, 6, 18, 1.5)
    Phix, Phiu, dg = o.SLS_H2(Pw, [Sxw, Suw], return_diag=True)
    np.savez_compressed(os.path.join(HERE, "weighted_chain_phi.npz"), Nx=Nx, q=q, r=r, b=b,
                        D11_data=D11.data, D11_indices=D11.indices, D11_indptr=D11.indptr,
                        d=6, T=18, alpha=1.5,
                        vals_x=flat(o.values_in_mask_order(Phix, Sxw)), vals_u=flat(o.values_in_mask_order(Phiu, Suw)),
                        col_cost=np.array([d["cost"] for d in dg]))
    print("weighted: total cost", sum(d["cost"] for d in dg), "max resid", max(d["resid"] for d in dg))

    # ---- infeasible small chain (d too small for the actuator spacing): status fixture ----
    Pi = o.readme_chain(23)
    Sxi, Sui = o.readme_masks(Pi.A, Pi.B2, 4, 12, 1.5)
    Phix, Phiu, dg = o.SLS_H2(Pi, [Sxi, Sui], return_diag=True)
    np.savez_compressed(os.path.join(HERE, "infeasible_chain.npz"), Nx=23, d=4, T=12, alpha=1.5,
                        col_resid=np.array([d["resid"] for d in dg]),
                        vals_x=flat(o.values_in_mask_order(Phix, Sxi)), vals_u=flat(o.values_in_mask_order(Phiu, Sui)))
    print("infeasible chain: columns with residual > 1e-9:", [i for i, d in enumerate(dg) if d["resid"] > 1e-9])

    # ---- grouped README chain ----
    groups = [list(range(0, 20)), list(range(20, 40)), list(range(40, 59))]
    Phix, Phiu, dg = o.SLS_H2(P, [Sx, Su], I=groups, return_diag=True)
    np.savez_compressed(os.path.join(HERE, "grouped_chain_phi.npz"),
                        vals_x=flat(o.values_in_mask_order(Phix, Sx)), vals_u=flat(o.values_in_mask_order(Phiu, Su)),
                        group_cost=np.array([d["cost"] for d in dg]))
    print("grouped: total cost", sum(d["cost"] for d in dg), "max resid", max(d["resid"] for d in dg))


if __name__ == "__main__":
    main()
