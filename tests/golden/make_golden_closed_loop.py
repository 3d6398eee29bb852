"""Golden fixture of the README closed-loop simulation (reference README.md:62-72): x, u of the impulse response
w(t) = δ(t−50)·e₃₀ over 250 steps, computed by the oracle's restatement of the script from the committed golden Φ
(readme_chain_phi.npz).  Only the support window is stored (x and u vanish outside times 51..80).
Run:  python tests/golden/make_golden_closed_loop.py"""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import sls_oracle as o  # noqa: E402


def main():
    P = o.readme_chain()
    Sx, Su = o.readme_masks(P.A, P.B2, 9, 29, 1.5)
    g = np.load(os.path.join(HERE, "readme_chain_phi.npz"))

    def build(masks, flat):
        out, k = [], 0
        for M in masks:
            M = sp.csc_matrix(M); M.sort_indices()
            out.append(sp.csc_matrix((flat[k:k + M.nnz], M.indices, M.indptr), shape=M.shape)); k += M.nnz
        assert k == len(flat)
        return out
    Phix, Phiu = build(Sx, g["vals_x"]), build(Su, g["vals_u"])
    x, u = o.closed_loop(P.A, P.B1, P.B2, Phix, Phiu)            # (Nx, 250), (Nu, 250): column k ↔ time k+1
    nzc = np.flatnonzero(np.abs(x).max(axis=0) + np.abs(u).max(axis=0) > 0)
    lo, hi = int(nzc.min()), int(nzc.max()) + 1
    np.savez_compressed(os.path.join(HERE, "readme_closed_loop.npz"), x=x[:, lo:hi], u=u[:, lo:hi], t0=lo, steps=250,
                        t_imp=50, i_imp=29)
    print("support columns", lo, hi, "max|x|", np.abs(x).max(), "max|u|", np.abs(u).max())


if __name__ == "__main__":
    main()
