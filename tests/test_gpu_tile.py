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
