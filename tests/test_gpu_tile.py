"""GPU tests of the FP64-MFMA tile kernel (csrc/sls_tile_kernel.hip): the ñx > 64 regime without a size limit
(reference src/synthesis.jl:46-62 has none), against the C restatement of the oracle.  TOL as in test_gpu_parity.py."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import flat_phi

pytestmark = pytest.mark.gpu
TOL = 1e-8


def _c_oracle_flat(P, S, cols):
    import sls_oracle as o
    import sls_oracle_cport as cp
    Po = o.OraclePlant(P.A, P.B1, P.B2)
    ox, ou, info = cp.SLS_H2(Po, S, cols=cols, nthreads=8)
    return np.concatenate([flat_phi(ox, S[0]), flat_phi(ou, S[1])]), info


def _colidx(P, S):
    return np.concatenate([np.repeat(np.arange(P.Nx), np.diff(M.indptr)) for M in S[0] + S[1]])


@pytest.mark.parametrize("n,mlds", [(16, 1), (23, 1), (40, 1), (96, 1), (131, 1), (144, 0), (200, 0), (333, 0)])
def test_tile_sweep_inverts_spd_matrix(slc, gpu_ctx, n, mlds):
    """The blocked symmetric sweep on v_mfma_f64_16x16x4_f64 tiles against numpy's inverse: pins the FP64 MFMA operand and
    result lane maps (A[l&15][l>>4], B[l>>4][l&15], C/D row = (l>>4)+4·reg) with an asymmetric-data matrix."""
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n))
    A = G @ G.T / n + np.diag(rng.uniform(0.5, 2.0, n))
    out = np.zeros((n, n))
    dp = C.POINTER(C.c_double)
    rc = gpu_ctx._lib.sls_debug_tile_invert(gpu_ctx.handle, 0, n, np.ascontiguousarray(A).ctypes.data_as(dp), out.ctypes.data_as(dp), mlds)
    assert rc == 0, slc._capi.last_error(gpu_ctx.handle)
    want = np.linalg.inv(A)
    assert np.abs(out - want).max() < 1e-11 * np.abs(want).max() * np.linalg.cond(A)
    # off-diagonal tiles exist once (exactly symmetric); inside a diagonal tile both triangles are computed, equal to rounding
    assert np.abs(out - out.T).max() < 1e-13 * np.abs(want).max() * np.linalg.cond(A)
    if n > 16:
        assert np.array_equal(out[:16, 16:], out[16:, :16].T)


def test_large_index_sets_are_solved_not_flagged(slc, gpu_ctx):
    """A 16×16 grid with d = 8: interior columns have ñx = 181 — beyond every round-1 kernel (they came back
    SLS_COL_UNSUPPORTED).  They now run on the tile kernel with the block in the global workspace."""
    P = slc.workloads.grid_plant(16, 3)
    S = list(slc.workloads.localization_masks(P.A, P.B2, 8, 14, 1.5))
    cols = [0, 15, 119, 120, 136, 255]                     # 120 is a feasible interior column (ñx = 171)
    plan = slc.Plan(gpu_ctx, P, S, [[c] for c in cols])
    desc = plan.describe()
    plan.close()
    assert "h2_column_tile_kernel<block_in_workspace>" in desc, desc
    Phix, Phiu, info = slc.SLS_H2(P, S, [[c] for c in cols], ctx=gpu_ctx, return_info=True, dropzeros=False)
    st = info["col_status"]
    assert info["max_nx"] > 144
    assert not np.any(st == slc._capi.SLS_COL_UNSUPPORTED)
    got = np.concatenate([flat_phi(Phix, S[0]), flat_phi(Phiu, S[1])])
    want, oinfo = _c_oracle_flat(P, S, cols)
    feasible = oinfo["status"] == 0
    assert np.array_equal(st == 0, feasible) and feasible[3]
    ok = np.isin(_colidx(P, S), np.asarray(cols)[feasible])
    assert ok.any() and np.abs(got[ok] - want[ok]).max() < TOL
    ok120 = _colidx(P, S) == 120
    assert np.abs(got[ok120] - want[ok120]).max() < TOL and np.abs(want[ok120]).max() > 0.1


def test_round1_size_limit_still_reported_when_tile_kernel_is_off(slc):
    """SLS_TILE=0 restores the round-1 launch list: ñx > 144 is never launched and comes back SLS_COL_UNSUPPORTED with zero
    values while the rest of the call is solved (the status word must stay truthful whatever the kernel set)."""
    P = slc.workloads.grid_plant(16, 3)
    S = list(slc.workloads.localization_masks(P.A, P.B2, 8, 14, 1.5))
    cols = [0, 119, 255]
    os.environ["SLS_TILE"] = "0"
    try:
        ctx = slc.Context([0])
        Phix, Phiu, info = slc.SLS_H2(P, S, [[c] for c in cols], ctx=ctx, return_info=True, dropzeros=False)
        ctx.close()
    finally:
        del os.environ["SLS_TILE"]
    st = info["col_status"]
    assert st[1] == slc._capi.SLS_COL_UNSUPPORTED and st[0] != slc._capi.SLS_COL_UNSUPPORTED
    got = np.concatenate([flat_phi(Phix, S[0]), flat_phi(Phiu, S[1])])
    assert np.all(got[_colidx(P, S) == 119] == 0.0)


@pytest.mark.parametrize("one_per_cu", ["0", "1"])
def test_tile_kernel_block_in_lds_on_grid_columns(slc, one_per_cu):
    """Default kernel selection: every ñx > 64 column runs on the tile kernel: grid-32 (ñx = 85, block in LDS) on the columns
    of test_grid_plant_general_kernel_and_infeasible_columns — statuses and feasible values against the C restatement.  Both
    LDS plans: two workgroups per CU (D' built in 16-row strips, staged through the slot) and one (whole Ã·Q image in LDS)."""
    P, S, _ = slc.workloads.make_workload("grid32")
    cols = [0, 31, 200, 495, 500, 528, 529, 1023]
    os.environ["SLS_TILE_ONE_PER_CU"] = one_per_cu
    try:
        ctx = slc.Context([0])
        plan = slc.Plan(ctx, P, S, [[c] for c in cols])
        desc = plan.describe()
        plan.close()
        Phix, Phiu, info = slc.SLS_H2(P, S, [[c] for c in cols], ctx=ctx, return_info=True, dropzeros=False)
        ctx.close()
    finally:
        del os.environ["SLS_TILE_ONE_PER_CU"]
    assert ("per_cu=1" in desc) == (one_per_cu == "1"), desc
    assert "h2_column_tile_kernel<block_in_LDS>" in desc and "general" not in desc, desc
    got = np.concatenate([flat_phi(Phix, S[0]), flat_phi(Phiu, S[1])])
    want, oinfo = _c_oracle_flat(P, S, cols)
    feasible = oinfo["status"] == 0
    assert 0 < feasible.sum() < len(cols)
    assert np.array_equal(info["col_status"] == 0, feasible)
    ok = np.isin(_colidx(P, S), np.asarray(cols)[feasible])
    assert np.abs(got[ok] - want[ok]).max() < TOL


def test_tile_kernel_weighted_and_wide_inputs(slc):
    """Diagonal LQR weights + D11 feed-through + B1 = diag(b) on a 12×12 grid (ñx = 85 interior, the weight record path of the
    tile kernel), checked through the certificate the weighted problem offers: full-system achievability and agreement with
    the workgroup kernel's result (same mathematics, different inversion)."""
    Pg = slc.workloads.grid_plant(12, 2)
    rng = np.random.default_rng(3)
    Nx, Nu = Pg.Nx, Pg.Nu
    q = rng.uniform(0.5, 2.0, Nx); r = rng.uniform(0.5, 2.0, Nu); b = rng.uniform(0.5, 1.5, Nx)
    C1 = sp.vstack([sp.diags(q), sp.csc_matrix((Nu, Nx))]).tocsc()
    D12 = sp.vstack([sp.csc_matrix((Nx, Nu)), sp.diags(r)]).tocsc()
    D11 = sp.random(Nx + Nu, Nx, density=0.02, random_state=4, format="csc") * 0.1
    P = slc.Plant(Pg.A, sp.diags(b).tocsc(), Pg.B2, C1, D11, D12)
    S = list(slc.workloads.localization_masks(P.A, P.B2, 5, 12, 1.5))
    cols = [0, 11, 66, 77, 78, 143]
    res = {}
    for mode in ("0", "all"):
        os.environ["SLS_TILE"] = mode                     # "0": the round-1 workgroup kernel; anything else: the tile kernel
        try:
            ctx = slc.Context([0])
            Phix, Phiu, info = slc.SLS_H2(P, S, [[c] for c in cols], ctx=ctx, return_info=True, dropzeros=False)
            ctx.close()
        finally:
            del os.environ["SLS_TILE"]
        res[mode] = (np.concatenate([flat_phi(Phix, S[0]), flat_phi(Phiu, S[1])]), info["col_status"].copy(), info)
    assert res["all"][2]["max_nx"] > 64
    assert np.array_equal(res["0"][1], res["all"][1])
    okc = np.asarray(cols)[res["0"][1] == 0]
    assert len(okc) >= 2
    ok = np.isin(_colidx(P, S), okc)
    assert np.abs(res["0"][0][ok] - res["all"][0][ok]).max() < TOL


def test_non_diagonal_cost_weights_match_golden(slc, gpu_ctx):
    """[C1 D12] banded (couples neighbouring states, and states with inputs), D11 ≠ 0, B1 = diag(b): the dense Hessian path
    (reference src/synthesis.jl:50,76-83 accepts any [C̃1 D̃12]; round 1 returned SLS_EUNSUPPORTED).  Every column runs on
    the tile kernel (ñx ≤ 13 here: a single pivot tile) with projected conjugate gradients on top of the diagonal-weight
    solve; Φ against the SVD oracle's golden vector, statuses against its residuals."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "general_weights_phi.npz"))
    Nx = int(g["Nx"])
    Pc = slc.workloads.chain_plant(Nx)
    Nu = Pc.Nu
    W = sp.csc_matrix((g["W_data"], g["W_indices"], g["W_indptr"]), shape=(Nx + Nu, Nx + Nu))
    D11 = sp.csc_matrix((g["D11_data"], g["D11_indices"], g["D11_indptr"]), shape=(Nx + Nu, Nx))
    P = slc.Plant(Pc.A, sp.diags(g["b"]).tocsc(), Pc.B2, W[:, :Nx], D11, W[:, Nx:])
    S = list(slc.workloads.localization_masks(P.A, P.B2, int(g["d"]), int(g["T"]), float(g["alpha"])))
    plan = slc.Plan(gpu_ctx, P, S)
    desc = plan.describe()
    plan.close()
    assert "h2_column_tile_kernel" in desc and "wave" not in desc and "twisted" not in desc, desc
    Phix, Phiu, info = slc.SLS_H2(P, S, ctx=gpu_ctx, return_info=True, dropzeros=False)
    feasible = g["col_resid"] < 1e-9
    assert feasible.sum() >= 8
    assert np.array_equal(info["col_status"] == 0, feasible), (info["col_status"], g["col_resid"])
    got = np.concatenate([flat_phi(Phix, S[0]), flat_phi(Phiu, S[1])])
    want = np.concatenate([g["vals_x"], g["vals_u"]])
    ok = np.isin(_colidx(P, S), np.flatnonzero(feasible))
    assert np.abs(got[ok] - want[ok]).max() < TOL


@pytest.mark.parametrize("tag", ["dense", "diag"])
def test_coupled_column_groups_match_golden(slc, gpu_ctx, tag):
    """Multi-column groups whose columns are coupled through a non-diagonal B̃1 = B1[c_j, c_j] (reference src/synthesis.jl:42,50;
    rounds 1–2a returned SLS_EUNSUPPORTED): the group is one work item of the tile kernel's CG build — joint projected conjugate
    gradients over its columns, Hessian (B̃1B̃1ᵀ) ⊗ [C̃1 D̃12]ᵀ[C̃1 D̃12], each column's own diagonal-weight solve as its block of the
    constraint preconditioner.  Φ against the SVD oracle's joint solve of each group (golden vector), with a banded and with a
    diagonal [C1 D12]; D11 ≠ 0; groups of 1–4 columns, one of them a single column with the same B1."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "coupled_group_phi.npz"))
    Nx = int(g["Nx"])
    Pc = slc.workloads.chain_plant(Nx)
    Nu = Pc.Nu
    W = sp.csc_matrix((g[f"{tag}_W_data"], g[f"{tag}_W_indices"], g[f"{tag}_W_indptr"]), shape=(Nx + Nu, Nx + Nu))
    B1 = sp.csc_matrix((g["B1_data"], g["B1_indices"], g["B1_indptr"]), shape=(Nx, Nx))
    D11 = sp.csc_matrix((g["D11_data"], g["D11_indices"], g["D11_indptr"]), shape=(Nx + Nu, Nx))
    P = slc.Plant(Pc.A, B1, Pc.B2, W[:, :Nx], D11, W[:, Nx:])
    S = list(slc.workloads.localization_masks(P.A, P.B2, int(g["d"]), int(g["T"]), float(g["alpha"])))
    gp = g["group_ptr"]; gc = g["group_cols"]
    groups = [[int(c) for c in gc[gp[i]:gp[i + 1]]] for i in range(len(gp) - 1)]
    Phix, Phiu, info = slc.SLS_H2(P, S, groups, ctx=gpu_ctx, return_info=True, dropzeros=False)
    assert np.all(g[f"{tag}_group_resid"] < 1e-9)
    assert np.all(info["col_status"] == 0), info["col_status"]
    got = np.concatenate([flat_phi(Phix, S[0]), flat_phi(Phiu, S[1])])
    want = np.concatenate([g[f"{tag}_vals_x"], g[f"{tag}_vals_u"]])
    assert np.abs(got - want).max() < TOL * max(1.0, np.abs(want).max())
    # and it is a different answer from the decoupled one: solving the same columns one by one (groups of one) ignores R's
    # off-diagonals
    Px1, Pu1 = slc.SLS_H2(P, S, ctx=gpu_ctx, dropzeros=False)
    one = np.concatenate([flat_phi(Px1, S[0]), flat_phi(Pu1, S[1])])
    assert np.abs(one - want).max() > 1e-3


@pytest.mark.parametrize("spacing,d,T,min_ok,tol", [(1, 3, 8, 6, TOL), (2, 4, 10, 2, 1e-6)])
def test_coupled_groups_multi_tile_and_infeasible_groups(slc, gpu_ctx, oracle, spacing, d, T, min_ok, tol):
    """Coupled groups on a grid plant (ñx up to 56: several 16×16 tiles per pivot block; the groups' index sets are unions over
    their columns) against the live NumPy oracle's joint solve; with an actuator on every second node most groups contain an
    infeasible column and are flagged as a whole, and the feasible ones are only just so (the oracle's own residual is 4e-14
    there instead of 1e-15: Φ is determined to residual/σ_min, DESIGN §2 — hence 1e-6 for that case, measured 4e-8)."""
    base = slc.workloads.grid_plant(8, spacing)
    rng = np.random.default_rng(9)
    Nx, Nu = base.Nx, base.Nu
    B1 = sp.lil_matrix(sp.diags(rng.uniform(0.7, 1.3, Nx)))
    groups = [[0, 1], [9, 10, 11], [27, 28], [36, 37, 44, 45], [62, 63], [20]]
    for gq in groups:
        for a in gq:
            for b in gq:
                if a != b and rng.uniform() < 0.7:
                    B1[a, b] = rng.uniform(-0.4, 0.4)
    B1 = B1.tocsc()
    q = rng.uniform(0.5, 2.0, Nx); r = rng.uniform(0.5, 2.0, Nu)
    C1 = sp.vstack([sp.diags(q), sp.csc_matrix((Nu, Nx))]).tocsc()
    D12 = sp.vstack([sp.csc_matrix((Nx, Nu)), sp.diags(r)]).tocsc()
    P = slc.Plant(base.A, B1, base.B2, C1, 0, D12)
    S = list(slc.workloads.localization_masks(P.A, P.B2, d, T, 1.5))
    Phix, Phiu, info = slc.SLS_H2(P, S, groups, ctx=gpu_ctx, return_info=True, dropzeros=False)
    assert info["max_nx"] > 32
    Po = oracle.OraclePlant(P.A, P.B1, P.B2, P.C1, P.D11, P.D12)
    ox, ou, dg = oracle.SLS_H2(Po, S, groups, return_diag=True)
    st = info["col_status"]
    k = 0
    n_ok = 0
    for gq, d in zip(groups, dg):
        stg = st[k:k + len(gq)]; k += len(gq)
        if d["resid"] < 1e-9:
            assert np.all(stg == 0), (gq, stg, d["resid"])
            n_ok += 1
            for c in gq:
                err = max(max(abs(X[:, c] - O[:, c]).max() for X, O in zip(Phix, ox)), max(abs(U[:, c] - O[:, c]).max() for U, O in zip(Phiu, ou)))
                assert err < tol, (gq, c, err)
        else:
            assert np.all(stg != 0), (gq, stg, d["resid"])
    assert n_ok >= min_ok


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_coupled_groups_against_live_oracle(slc, gpu_ctx, oracle, seed):
    """Random small plants, random ascending groups of 1–5 columns (not neighbours: the groups' index sets are unions of distant
    neighbourhoods), random sparse B1 with entries inside the groups, random diagonal weights, D11 ≠ 0, 1-based index arrays on odd
    seeds: every group the oracle finds feasible must match its joint solve; the others must be flagged as a whole."""
    rng = np.random.default_rng(100 + seed)
    Nx, Nu = 40, 40
    A = sp.random(Nx, Nx, density=0.08, random_state=seed, format="csc") * 0.5 + sp.eye(Nx, format="csc")
    B2 = sp.eye(Nx, format="csc")                                   # fully actuated: most groups feasible
    cols = rng.permutation(Nx)[:24]
    groups, k = [], 0
    while k < len(cols):
        sz = int(rng.integers(1, 6)); groups.append(sorted(int(c) for c in cols[k:k + sz])); k += sz
    B1 = sp.lil_matrix(sp.diags(rng.uniform(0.6, 1.4, Nx)))
    for gq in groups:
        for a in gq:
            for b in gq:
                if a != b and rng.uniform() < 0.5:
                    B1[a, b] = rng.uniform(-0.5, 0.5)
    B1 = B1.tocsc()
    q = rng.uniform(0.5, 2.0, Nx); r = rng.uniform(0.5, 2.0, Nu)
    C1 = sp.vstack([sp.diags(q), sp.csc_matrix((Nu, Nx))]).tocsc()
    D12 = sp.vstack([sp.csc_matrix((Nx, Nu)), sp.diags(r)]).tocsc()
    D11 = sp.random(Nx + Nu, Nx, density=0.05, random_state=seed + 7, format="csc") * 0.2
    P = slc.Plant(A, B1, B2, C1, D11, D12)
    S = list(slc.workloads.localization_masks(P.A, P.B2, 2, 6, 1.5))
    Phix, Phiu, info = slc.SLS_H2(P, S, groups, ctx=gpu_ctx, return_info=True, dropzeros=False, index_base=seed % 2)
    Po = oracle.OraclePlant(P.A, P.B1, P.B2, P.C1, P.D11, P.D12)
    ox, ou, dg = oracle.SLS_H2(Po, S, groups, return_diag=True)
    st = info["col_status"]
    k = 0; n_ok = 0; n_mispaired = 0
    for gq, d in zip(groups, dg):
        stg = st[k:k + len(gq)]; k += len(gq)
        # The reference pairs the k-th column of the group with the k-th member of c_j in the FIRST-APPEARANCE order of s_x
        # (src/reduction.jl:14,22-23: Ĩ = I[:, s_x ∈ c_j]); when the neighbourhoods of an ascending group interleave, that order is
        # not ascending and Φ̃x[1][:, c] = e_{pos(c')} for another member c' — a constraint its own mask then contradicts (the
        # oracle, which restates this, reports residual 1).  This library pairs column c with e_pos(c); such
        # groups are not comparable.
        _, _, _, sx, _ = oracle.sparsity_dim_reduction(Po, gq, S)
        if [int(v) for v in sx if int(v) in gq] != gq:
            n_mispaired += 1
            continue
        if d["resid"] < 1e-10:
            assert np.all(stg == 0), (gq, stg, d["resid"])
            n_ok += 1
            for c in gq:
                err = max(max(abs(X[:, c] - O[:, c]).max() for X, O in zip(Phix, ox)), max(abs(U[:, c] - O[:, c]).max() for U, O in zip(Phiu, ou)))
                assert err < 1e-7, (gq, c, err)
        elif d["resid"] > 1e-6:
            assert np.all(stg != 0), (gq, stg, d["resid"])
    assert n_ok >= (len(groups) - n_mispaired) // 2 and n_ok >= 3


def test_coupled_groups_and_sum_of_norms_on_several_device_slots(slc):
    """The in-process multi-device split (packed layout per shard + host scatter; here three slots on the one GPU) with work items
    that are whole groups and with the sum-of-norms objective: identical to the one-device call."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "coupled_group_phi.npz"))
    Nx = int(g["Nx"]); Pc = slc.workloads.chain_plant(Nx); Nu = Pc.Nu
    W = sp.csc_matrix((g["dense_W_data"], g["dense_W_indices"], g["dense_W_indptr"]), shape=(Nx + Nu, Nx + Nu))
    B1 = sp.csc_matrix((g["B1_data"], g["B1_indices"], g["B1_indptr"]), shape=(Nx, Nx))
    D11 = sp.csc_matrix((g["D11_data"], g["D11_indices"], g["D11_indptr"]), shape=(Nx + Nu, Nx))
    P = slc.Plant(Pc.A, B1, Pc.B2, W[:, :Nx], D11, W[:, Nx:])
    S = list(slc.workloads.localization_masks(P.A, P.B2, int(g["d"]), int(g["T"]), float(g["alpha"])))
    gp = g["group_ptr"]; gc = g["group_cols"]
    groups = [[int(c) for c in gc[gp[i]:gp[i + 1]]] for i in range(len(gp) - 1)]
    want = np.concatenate([g["dense_vals_x"], g["dense_vals_u"]])
    ref = None
    for devs in ([0], [0, 0, 0]):
        ctx = slc.Context(devs)
        try:
            Px, Pu, info = slc.SLS_H2(P, S, groups, ctx=ctx, return_info=True, dropzeros=False)
            got = np.concatenate([flat_phi(Px, S[0]), flat_phi(Pu, S[1])])
            assert np.all(info["col_status"] == 0) and np.abs(got - want).max() < TOL * max(1.0, np.abs(want).max())
            Px, Pu, info = slc.SLS_H2(Pc, S, ctx=ctx, return_info=True, dropzeros=False, objective="sum_of_norms")
            son = np.concatenate([flat_phi(Px, S[0]), flat_phi(Pu, S[1])])
            assert np.all(info["col_status"] == 0)
            ref = son if ref is None else ref
            assert np.array_equal(son, ref)
        finally:
            ctx.close()


def test_carve_in_workspace_variant_equals_the_lds_variants(slc, monkeypatch):
    """Round 3: the tile kernel's third home for its working set — panels, Ã·Q strip, staging vectors and sparse lists carved
    from a per-workgroup buffer in global memory instead of LDS (index sets beyond what 160 KiB hold).  SLS_TILE_BIG=all routes
    EVERY tile column through it: the large-index-set grid columns of the test above (diagonal weights) and the non-diagonal
    weight fixture (the CG build) must come out as from the LDS variants."""
    P = slc.workloads.grid_plant(16, 3)
    S = list(slc.workloads.localization_masks(P.A, P.B2, 8, 14, 1.5))
    cols = [0, 15, 119, 120, 136, 255]
    ctx = slc.Context([0])
    ref = slc.SLS_H2(P, S, [[c] for c in cols], ctx=ctx, return_info=True, dropzeros=False)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "general_weights_phi.npz"))
    monkeypatch.setenv("SLS_TILE_BIG", "all")
    plan = slc.Plan(ctx, P, S, [[c] for c in cols])
    desc = plan.describe()
    plan.close()
    assert "h2_column_tile_kernel<block_in_workspace,carve_in_workspace>" in desc and "block_in_LDS" not in desc, desc
    big = slc.SLS_H2(P, S, [[c] for c in cols], ctx=ctx, return_info=True, dropzeros=False)
    assert np.array_equal(big[2]["col_status"], ref[2]["col_status"])
    a = np.concatenate([flat_phi(ref[0], S[0]), flat_phi(ref[1], S[1])])
    b = np.concatenate([flat_phi(big[0], S[0]), flat_phi(big[1], S[1])])
    ok = np.isin(_colidx(P, S), np.asarray(cols)[ref[2]["col_status"] == 0])
    assert np.abs(a[ok] - b[ok]).max() < 1e-10 * max(1.0, np.abs(a[ok]).max())
    # the CG build (dense cost Hessian) in the same variant, against the SVD oracle's golden vector
    Nx = int(g["Nx"])
    Pc = slc.workloads.chain_plant(Nx)
    Nu = Pc.Nu
    W = sp.csc_matrix((g["W_data"], g["W_indices"], g["W_indptr"]), shape=(Nx + Nu, Nx + Nu))
    D11 = sp.csc_matrix((g["D11_data"], g["D11_indices"], g["D11_indptr"]), shape=(Nx + Nu, Nx))
    Pg = slc.Plant(Pc.A, sp.diags(g["b"]).tocsc(), Pc.B2, W[:, :Nx], D11, W[:, Nx:])
    Sg = list(slc.workloads.localization_masks(Pg.A, Pg.B2, int(g["d"]), int(g["T"]), float(g["alpha"])))
    plan = slc.Plan(ctx, Pg, Sg)
    desc = plan.describe()
    plan.close()
    assert "carve_in_workspace,dense_hessian_cg" in desc, desc
    Phix, Phiu, info = slc.SLS_H2(Pg, Sg, ctx=ctx, return_info=True, dropzeros=False)
    feasible = g["col_resid"] < 1e-9
    assert np.array_equal(info["col_status"] == 0, feasible)
    got = np.concatenate([flat_phi(Phix, Sg[0]), flat_phi(Phiu, Sg[1])])
    want = np.concatenate([g["vals_x"], g["vals_u"]])
    okg = np.isin(_colidx(Pg, Sg), np.flatnonzero(feasible))
    assert np.abs(got[okg] - want[okg]).max() < TOL
    ctx.close()


@pytest.mark.timeout(900)
def test_index_sets_beyond_lds_are_solved(slc, gpu_ctx):
    """No size limit left (the reference has none: src/synthesis.jl:46-62).  A 24×24 grid with an actuator on every state and
    d = 20: ñx = 576, ñu = 576 — the two pivot panels alone (32·ñx doubles = 147 KiB) leave no room in LDS, round 2 answered
    SLS_COL_UNSUPPORTED.  Three columns (centre, edge, corner) against the C restatement at 1e-8, and the plan says which
    variant ran."""
    P = slc.workloads.grid_plant(24, 1)
    S = list(slc.workloads.localization_masks(P.A, P.B2, 20, 6, 4.0))
    cols = [12 * 24 + 12, 12 * 24, 0]
    plan = slc.Plan(gpu_ctx, P, S, [[c] for c in cols])
    desc = plan.describe()
    info_p = dict(plan.info)
    plan.close()
    assert "carve_in_workspace" in desc, desc
    assert info_p["max_nx"] >= 500, info_p
    Phix, Phiu, info = slc.SLS_H2(P, S, [[c] for c in cols], ctx=gpu_ctx, return_info=True, dropzeros=False)
    st = info["col_status"]
    assert np.all(st == 0), st
    got = np.concatenate([flat_phi(Phix, S[0]), flat_phi(Phiu, S[1])])
    want, oinfo = _c_oracle_flat(P, S, cols)
    assert np.all(oinfo["status"] == 0)
    ok = np.isin(_colidx(P, S), cols)
    assert np.abs(want[ok]).max() > 0.1
    assert np.abs(got[ok] - want[ok]).max() < TOL * max(1.0, np.abs(want[ok]).max())
