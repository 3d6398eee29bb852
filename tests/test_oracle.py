"""CPU tests of the oracle itself: pinned against the reference's only fixture for this path
(test/reduction_test.jl:11-24), against the committed golden vectors, and against the
solver-independent optimality certificate."""
import json
import os

import numpy as np
import scipy.sparse as sp

from conftest import GOLDEN, split_vals


def test_reduction_known_answer_reference_fixture(oracle):
    """reference test/reduction_test.jl:11-24: c_j = 1:20 ⇒ s_x = 1:30, s_u = 1:10,
    ii_x = [ones(20); zeros(10)], Ĩ = I(30)[:,1:20], P̃ == Plant(A[sx,sx], B1[sx,cj], B2[sx,su])."""
    ka = json.load(open(os.path.join(GOLDEN, "reduction_known_answer.json")))
    P = oracle.readme_chain()
    Ab = (P.A != 0).astype(np.int64).tocsc()
    A9 = oracle._bool_power(Ab, 9)
    Sx = [(A9 != 0).tocsc()]
    Su = [((((P.B2.T != 0).astype(np.int64)) @ A9) != 0).tocsc()]
    cj = np.asarray(ka["cj"])
    sub, It, iix, sx, su = oracle.sparsity_dim_reduction(P, cj, [Sx, Su])
    assert sx.tolist() == ka["expected_sx"]
    assert su.tolist() == ka["expected_su"]
    assert iix.astype(int).tolist() == ka["expected_iix"]
    assert np.array_equal(It, np.eye(30)[:, :20])
    assert np.array_equal(sub["A"], P.A[sx][:, sx].toarray())
    assert np.array_equal(sub["B1"], P.B1[sx][:, cj].toarray())
    assert np.array_equal(sub["B2"], P.B2[sx][:, su].toarray())
    # default LQR weights of the 3-argument Plant: [C1 D12] = I on the selected rows/cols
    assert np.array_equal(np.hstack([sub["C1"], sub["D12"]]), np.eye(40))


def test_readme_masks_counts(oracle):
    P = oracle.readme_chain()
    Sx, Su = oracle.readme_masks(P.A, P.B2, 9, 29, 1.5)
    assert sum(s.nnz for s in Sx) == 26413 and sum(s.nnz for s in Su) == 9616


def test_oracle_columns_match_golden_and_certificate(oracle, golden_readme):
    P = oracle.readme_chain()
    Sx, Su = oracle.readme_masks(P.A, P.B2, 9, 29, 1.5)
    for j in (0, 10, 29, 58):
        z, info, d = oracle.solve_group(P, [j], Sx, Su)
        assert abs(d["cost"] - golden_readme["col_cost"][j]) < 1e-10
        assert info["n"] == golden_readme["col_n"][j] and info["m"] == golden_readme["col_m"][j]
        feas, pg = oracle.certificate(d["E"], d["f"], d["M"], d["m0"], z)
        assert feas < 1e-12 and pg < 1e-11
    assert abs(golden_readme["col_cost"].sum() - 893.3262819770) < 1e-8   # SURVEY §8c anchor


def _golden_phi(oracle, golden):
    P = oracle.readme_chain()
    Sx, Su = oracle.readme_masks(P.A, P.B2, 9, 29, 1.5)
    vx = split_vals(golden["vals_x"], [s.nnz for s in Sx])
    vu = split_vals(golden["vals_u"], [s.nnz for s in Su])
    Phix = [sp.csc_matrix((v, s.indices, s.indptr), shape=s.shape) for v, s in zip(vx, Sx)]
    Phiu = [sp.csc_matrix((v, s.indices, s.indptr), shape=s.shape) for v, s in zip(vu, Su)]
    return P, Phix, Phiu


def test_golden_full_system_achievability(oracle, golden_readme):
    """Φx[1] = I, Φx[t+1] = AΦx[t] + B2Φu[t], AΦx[T] + B2Φu[T] = 0 on the FULL plant (z-domain constraint of
    README.md:31) — holds although every column was solved on its own reduced index set."""
    P, Phix, Phiu = _golden_phi(oracle, golden_readme)
    T = len(Phix)
    assert abs(Phix[0] - sp.identity(P.Nx)).max() < 1e-13
    for t in range(T - 1):
        assert abs(Phix[t + 1] - (P.A @ Phix[t] + P.B2 @ Phiu[t])).max() < 1e-12
    assert abs(P.A @ Phix[T - 1] + P.B2 @ Phiu[T - 1]).max() < 1e-12


def test_golden_closed_loop_localized(oracle, golden_readme):
    """README.md:60-76 / res/SLS_H2_Chain.png: impulse at state 30, t = 50 stays inside |i−30| ≤ 9 and
    dies after T = 29 steps, while the open loop is unstable (ρ(A) > 1)."""
    P, Phix, Phiu = _golden_phi(oracle, golden_readme)
    x, u = oracle.closed_loop(P.A, P.B1, P.B2, Phix, Phiu)
    r, c = np.nonzero(np.abs(x) > 1e-9)
    assert r.min() + 1 >= 21 and r.max() + 1 <= 39
    assert c.min() + 1 == 51 and c.max() + 1 <= 79
    assert max(abs(np.linalg.eigvals(P.A.toarray()))) > 1.05


def test_decoupled_group_equals_monolithic(oracle):
    """The column-by-column solve of a multi-column group (B̃1 diagonal) is the monolithic QP's optimum."""
    P = oracle.readme_chain(17)
    Sx, Su = oracle.readme_masks(P.A, P.B2, 6, 10, 1.5)
    z1, _, d1 = oracle.solve_group(P, [2, 3, 4], Sx, Su, decouple=True)
    z2, _, d2 = oracle.solve_group(P, [2, 3, 4], Sx, Su, decouple=False)
    assert np.abs(z1 - z2).max() < 1e-10 and abs(d1["cost"] - d2["cost"]) < 1e-10


def test_c_restatement_matches_numpy_oracle(oracle, golden_readme):
    """oracle/sls_oracle_c.c (block-tridiagonal Cholesky, the timed cpu_baseline) against the NumPy/SVD oracle's golden Φ
    of the README chain, plus its statuses on the infeasible fixture."""
    import sls_oracle_cport as cp
    from conftest import flat_phi
    P = oracle.readme_chain()
    Sx, Su = oracle.readme_masks(P.A, P.B2, 9, 29, 1.5)
    Phix, Phiu, info = cp.SLS_H2(P, [Sx, Su], nthreads=4)
    got = np.concatenate([flat_phi(Phix, Sx), flat_phi(Phiu, Su)])
    want = np.concatenate([golden_readme["vals_x"], golden_readme["vals_u"]])
    assert info["status"].max() == 0 and info["resid"].max() < 1e-12
    assert np.abs(got - want).max() < 1e-8          # same bar as the GPU tests (E⁺ amplifies the 1e-12 residual stop)
    g = np.load(os.path.join(GOLDEN, "infeasible_chain.npz"))
    Pi = oracle.readme_chain(int(g["Nx"]))
    Sxi, Sui = oracle.readme_masks(Pi.A, Pi.B2, int(g["d"]), int(g["T"]), float(g["alpha"]))
    _, _, info_i = cp.SLS_H2(Pi, [Sxi, Sui], nthreads=2)
    assert np.array_equal(info_i["status"] != 0, g["col_resid"] > 1e-9)


def test_c_restatement_under_sanitizers(tmp_path):
    """oracle/sls_oracle_c.c built with AddressSanitizer + UBSan and run on the README chain and the infeasible fixture in a child
    interpreter (the sanitizer runtime has to be preloaded): same statuses and values as the regular build, no report."""
    import shutil
    import subprocess
    import sys
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    here = os.path.dirname(os.path.abspath(__file__))
    odir = os.path.join(here, "..", "oracle")
    lib = str(tmp_path / "libsls_oracle_asan.so")
    b = subprocess.run(["gcc", "-O1", "-g", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-o", lib, os.path.join(odir, "sls_oracle_c.c"), "-lm"],
                       capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-3000:]
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan.so not found")
    code = (
        "import sys, os, numpy as np\n"
        f"sys.path.insert(0, {odir!r}); sys.path.insert(0, {here!r})\n"
        "import sls_oracle as o, sls_oracle_cport as cp\n"
        "from conftest import flat_phi\n"
        "P = o.readme_chain(); Sx, Su = o.readme_masks(P.A, P.B2, 9, 29, 1.5)\n"
        "Px, Pu, info = cp.SLS_H2(P, [Sx, Su], nthreads=2)\n"
        f"g = np.load(os.path.join({GOLDEN!r}, 'readme_chain_phi.npz'))\n"
        "got = np.concatenate([flat_phi(Px, Sx), flat_phi(Pu, Su)]); want = np.concatenate([g['vals_x'], g['vals_u']])\n"
        "assert info['status'].max() == 0 and np.abs(got - want).max() < 1e-8\n"
        f"gi = np.load(os.path.join({GOLDEN!r}, 'infeasible_chain.npz'))\n"
        "Pi = o.readme_chain(int(gi['Nx'])); Sxi, Sui = o.readme_masks(Pi.A, Pi.B2, int(gi['d']), int(gi['T']), float(gi['alpha']))\n"
        "_, _, ii = cp.SLS_H2(Pi, [Sxi, Sui], nthreads=2)\n"
        "assert np.array_equal(ii['status'] != 0, gi['col_resid'] > 1e-9)\n"
        "print('oracle_c sanitized: clean')\n")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", SLS_ORACLE_LIB=lib, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    assert "oracle_c sanitized: clean" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
