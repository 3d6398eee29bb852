"""Golden fixture for COUPLED column groups (reference src/synthesis.jl:42,50: the objective of a group c_j is
Σ_t ‖[C̃1 D̃12][Φ̃x[t]; Φ̃u[t]] B̃1 + D̃11‖²_F with B̃1 = B1[c_j ∩ s_x, c_j]; a non-diagonal block couples the group's columns):
   coupled_group_phi.npz   Nx = 23 chain, B1 tridiagonal (every multi-column group of neighbours is coupled), banded
                           non-diagonal [C1 D12], D11 ≠ 0, groups of 1–4 columns; Φ from the NumPy oracle (dense SVD null-space
                           method on the explicit joint (E, f, M, m0) of each group — oracle/sls_oracle.py:assemble_group).
An ORACLE output (the reference cannot run here: SURVEY §0 F5), certified by the optimality certificate.
Run:  python tests/golden/make_golden_coupled.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import sls_oracle as o  # noqa: E402

GROUPS = [[0, 1, 2], [3, 4], [5], [6, 7, 8, 9], [10, 11], [12, 13, 14], [15], [16, 17], [18, 19, 20], [21, 22]]


def coupled_problem(diagonal_weights=False):
    rng = np.random.default_rng(23)
    Nx = 23
    Pc = o.readme_chain(Nx)
    Nu = Pc.Nu
    Nz = Nx + Nu
    B1 = (sp.diags(rng.uniform(0.7, 1.3, Nx)) + sp.diags(rng.uniform(-0.4, 0.4, Nx - 1), 1) + sp.diags(rng.uniform(-0.4, 0.4, Nx - 1), -1)).tocsc()
    if diagonal_weights:
        W = sp.diags(rng.uniform(0.8, 1.6, Nz)).tocsc()
    else:
        W = sp.diags(rng.uniform(0.8, 1.6, Nz)) + sp.diags(rng.uniform(-0.3, 0.3, Nz - 1), 1) + sp.diags(rng.uniform(-0.3, 0.3, Nz - 2), -2)
        W = sp.csc_matrix(W)
    D11 = sp.random(Nz, Nx, density=0.1, random_state=7, format="csc") * 0.3
    P = o.OraclePlant(Pc.A, B1, Pc.B2, W[:, :Nx], D11, W[:, Nx:])
    return P, W, B1, D11


def main():
    out = {}
    for tag, diag in (("dense", False), ("diag", True)):
        P, W, B1, D11 = coupled_problem(diag)
        Sx, Su = o.readme_masks(P.A, P.B2, 6, 18, 1.5)
        Phix, Phiu, dg = o.SLS_H2(P, [Sx, Su], GROUPS, return_diag=True)
        flat = lambda vals: np.concatenate([np.asarray(v, dtype=np.float64) for v in vals])
        feas = pg = 0.0
        for cj in GROUPS[:4]:
            z, info, d = o.solve_group(P, cj, Sx, Su)
            f_, p_ = o.certificate(d["E"], d["f"], d["M"], d["m0"], z)
            feas, pg = max(feas, f_), max(pg, p_)
        W = sp.csc_matrix(W); B1 = sp.csc_matrix(B1)
        out.update({f"{tag}_W_data": W.data, f"{tag}_W_indices": W.indices, f"{tag}_W_indptr": W.indptr,
                    f"{tag}_vals_x": flat(o.values_in_mask_order(Phix, Sx)), f"{tag}_vals_u": flat(o.values_in_mask_order(Phiu, Su)),
                    f"{tag}_group_cost": np.array([d["cost"] for d in dg]), f"{tag}_group_resid": np.array([d["resid"] for d in dg]),
                    f"{tag}_cert_feas": feas, f"{tag}_cert_projgrad": pg})
        print(tag, "total cost", sum(d["cost"] for d in dg), "group resid", [float("%.1e" % d["resid"]) for d in dg], "cert", feas, pg)
    np.savez_compressed(os.path.join(HERE, "coupled_group_phi.npz"), Nx=P.Nx, d=6, T=18, alpha=1.5,
                        B1_data=B1.data, B1_indices=B1.indices, B1_indptr=B1.indptr,
                        D11_data=D11.data, D11_indices=D11.indices, D11_indptr=D11.indptr,
                        group_ptr=np.cumsum([0] + [len(g) for g in GROUPS]), group_cols=np.concatenate(GROUPS), **out)


if __name__ == "__main__":
    main()
