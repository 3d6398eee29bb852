"""Golden fixture for NON-DIAGONAL cost weights (reference src/synthesis.jl:50,76-83 takes any [C̃1 D̃12]):
   general_weights_phi.npz   Nx = 23 chain, [C1 D12] = banded (Nx+Nu)×(Nx+Nu) matrix (couples neighbouring states and
                             states with inputs), D11 ≠ 0, B1 = diag(b); Φ from the NumPy oracle (dense SVD null-space
                             method on the explicit (E, f, M, m0) — oracle/sls_oracle.py), per-column costs and residuals.
An ORACLE output (the reference cannot run here: SURVEY §0 F5), certified by the optimality certificate.
Run:  python tests/golden/make_golden_general.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import sls_oracle as o  # noqa: E402


def general_problem():
    rng = np.random.default_rng(11)
    Nx = 23
    Pc = o.readme_chain(Nx)
    Nu = Pc.Nu
    Nz = Nx + Nu
    # banded weight: diagonal in [0.8, 1.6] plus two off-diagonals and a few state–input couplings
    W = sp.diags(rng.uniform(0.8, 1.6, Nz)) + sp.diags(rng.uniform(-0.3, 0.3, Nz - 1), 1) + sp.diags(rng.uniform(-0.3, 0.3, Nz - 2), -2)
    W = sp.lil_matrix(W)
    for k in range(6):
        W[int(rng.integers(Nx)), Nx + int(rng.integers(Nu))] = rng.uniform(-0.4, 0.4)
    W = sp.csc_matrix(W)
    b = rng.uniform(0.5, 1.5, Nx)
    D11 = sp.random(Nz, Nx, density=0.1, random_state=5, format="csc") * 0.3
    P = o.OraclePlant(Pc.A, sp.diags(b).tocsc(), Pc.B2, W[:, :Nx], D11, W[:, Nx:])
    return P, W, b, D11


def main():
    P, W, b, D11 = general_problem()
    Sx, Su = o.readme_masks(P.A, P.B2, 5, 14, 1.5)
    Phix, Phiu, dg = o.SLS_H2(P, [Sx, Su], return_diag=True)
    flat = lambda vals: np.concatenate([np.asarray(v, dtype=np.float64) for v in vals])
    feas = pg = 0.0
    for j in (0, 7, 11, 22):
        z, info, d = o.solve_group(P, [j], Sx, Su)
        f_, p_ = o.certificate(d["E"], d["f"], d["M"], d["m0"], z)
        feas, pg = max(feas, f_), max(pg, p_)
    W = sp.csc_matrix(W)
    np.savez_compressed(os.path.join(HERE, "general_weights_phi.npz"), Nx=P.Nx, d=5, T=14, alpha=1.5, b=b,
                        W_data=W.data, W_indices=W.indices, W_indptr=W.indptr,
                        D11_data=D11.data, D11_indices=D11.indices, D11_indptr=D11.indptr,
                        vals_x=flat(o.values_in_mask_order(Phix, Sx)), vals_u=flat(o.values_in_mask_order(Phiu, Su)),
                        col_cost=np.array([d["cost"] for d in dg]), col_resid=np.array([d["resid"] for d in dg]),
                        cert_feas=feas, cert_projgrad=pg)
    print("general weights: total cost", sum(d["cost"] for d in dg), "max resid", max(d["resid"] for d in dg), "cert", feas, pg)


if __name__ == "__main__":
    main()
