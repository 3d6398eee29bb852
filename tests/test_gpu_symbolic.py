"""Destination tables on the device: the one-device drop-in call ships a compact form of the per-column
tables (a bit mask per time step + two base indices) and `expand_tables_kernel` rebuilds, on the MI355X, the byte masks and
int32 scatter destinations of reference src/synthesis.jl:57-60,65-67.  Checked bit for bit against the tables the host pass
builds, and the index sets behind them against the oracle's `sparsity_dim_reduction` (reference src/reduction.jl:14)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _tables(slc, ctx, P, S, groups, host):
    m = slc._capi.Marshalled(P, S[0], S[1], groups)
    n = C.c_int64(0); wc = C.c_int32(-1)
    lib = ctx._lib
    rc = lib.sls_debug_plan_tables(ctx.handle, 0, *m.common_args(), host, C.byref(n), None, None, C.byref(wc))
    assert rc == 0, slc._capi.last_error(ctx.handle)
    mask = np.zeros(n.value, dtype=np.uint8); dest = np.zeros(n.value, dtype=np.int32)
    rc = lib.sls_debug_plan_tables(ctx.handle, 0, *m.common_args(), host, C.byref(n), mask.ctypes.data_as(C.POINTER(C.c_uint8)),
                                   dest.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(wc))
    assert rc == 0, slc._capi.last_error(ctx.handle)
    return mask, dest, wc.value


@pytest.mark.parametrize("name", ["readme_chain", "grid12", "random_wide"])
def test_device_tables_equal_host_tables(slc, gpu_ctx, oracle, name):
    if name == "readme_chain":
        P, S, _ = slc.workloads.make_workload("readme_chain"); groups = None
    elif name == "grid12":
        P = slc.workloads.grid_plant(12, 2)
        S = list(slc.workloads.localization_masks(P.A, P.B2, 5, 12, 1.5)); groups = None      # ñx + ñu > 64: several mask words
    else:
        rng = np.random.default_rng(11)
        A = sp.random(300, 300, density=0.01, random_state=12, format="csc") + sp.eye(300, format="csc")
        B2 = sp.random(300, 120, density=0.02, random_state=13, format="csc")
        P = slc.Plant(A, sp.eye(300, format="csc"), B2)
        S = list(slc.workloads.localization_masks(P.A, P.B2, 3, 7, 1.5))
        groups = [[int(c)] for c in sorted(rng.choice(300, 90, replace=False))]
    mh, dh, _ = _tables(slc, gpu_ctx, P, S, groups, 1)
    md, dd, was_compact = _tables(slc, gpu_ctx, P, S, groups, 0)
    assert was_compact == 1
    assert mh.size > 0 and np.array_equal(mh, md)
    assert np.array_equal(dh, dd)
    # the destinations are a bijection onto the masked entries of the owned columns
    tgt = dd[md == 1]
    assert tgt.min() >= 0 and np.unique(tgt).size == tgt.size
    cols = range(P.Nx) if groups is None else [g[0] for g in groups]
    want = sum(int(M[:, c].nnz) for M in S[0] + S[1] for c in cols)
    assert tgt.size == want
    # the table layout is [column][t][s_x, s_u] with the oracle's index sets (reference src/reduction.jl:14): lengths agree
    Po = oracle.OraclePlant(P.A, P.B1, P.B2)
    T = len(S[0])
    total = 0
    for c in cols:
        _, _, _, sx, su = oracle.sparsity_dim_reduction(Po, [c], S)
        total += T * (len(sx) + len(su))
    assert total == md.size


def test_irregular_masks_fall_back_to_host_tables(slc, gpu_ctx):
    """A stored `false` in a mask (allowed: src/synthesis.jl:57 fixes Φ where 𝓢 .≠ 1) is not expressible in the compact form:
    the plan falls back to explicit host tables, and the answer has the exact zero there."""
    P = slc.workloads.chain_plant(23)
    S = list(slc.workloads.localization_masks(P.A, P.B2, 6, 18, 1.5))
    Sx = [m.copy().astype(np.uint8) for m in S[0]]
    c = 11
    col = Sx[5][:, c].tocoo()
    r = int(col.row[len(col.row) // 2])
    Sx[5] = Sx[5].tolil(); Sx[5][r, c] = 2; Sx[5] = Sx[5].tocsc()                 # stored, ≠ 1 after the `!= 0 → uint8`? keep explicit
    S2 = [Sx, S[1]]
    # marshal by hand so that the stored entry really is a false byte
    m = slc._capi.Marshalled(P, S2[0], S2[1], None)
    k = Sx[5].indptr[c] + int(np.searchsorted(Sx[5].indices[Sx[5].indptr[c]:Sx[5].indptr[c + 1]], r))
    nz = np.ctypeslib.as_array(m.Sx[5].nzval, shape=(Sx[5].nnz,))
    nz[k] = 0
    n = C.c_int64(0); wc = C.c_int32(-1)
    rc = gpu_ctx._lib.sls_debug_plan_tables(gpu_ctx.handle, 0, *m.common_args(), 0, C.byref(n), None, None, C.byref(wc))
    assert rc == 0 and wc.value == 0
    mask = np.zeros(n.value, dtype=np.uint8); dest = np.zeros(n.value, dtype=np.int32)
    rc = gpu_ctx._lib.sls_debug_plan_tables(gpu_ctx.handle, 0, *m.common_args(), 0, C.byref(n), mask.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            dest.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(wc))
    assert rc == 0
    want_free = sum(int(M.nnz) for M in S[0] + S[1]) - 1
    assert int(mask.sum()) == want_free


@pytest.mark.parametrize("name,base", [("readme_chain", 0), ("readme_chain", 1), ("grid32", 1), ("chain1024", 0), ("random", 0)])
def test_device_mask_recipe_equals_host_recipe(slc, gpu_ctx, name, base):
    """sls_localization_masks_device (level-set expansion on the MI355X, csrc/sls_masks.hip) against the host recipe
    sls_localization_masks — itself pinned to SciPy Boolean matrix powers (README.md:52-54) in tests/test_host.py — bit for bit:
    column pointers and Int64 row indices of every 𝓢x[t], 𝓢u[t], in both index bases."""
    wl = slc.workloads
    if name == "random":
        A = sp.random(600, 600, density=0.006, random_state=3, format="csc") + sp.eye(600, format="csc")
        A.data[::7] = 0.0                                            # stored zeros: (A .≠ 0) is by value
        B2 = sp.random(600, 240, density=0.01, random_state=4, format="csc")
        d, T, alpha = 3, 9, 1.5
    else:
        mk, d, T, alpha = wl.WORKLOADS[name]
        P = mk(); A, B2 = P.A, P.B2
    hx, hu = wl.localization_masks_native(A, B2, d, T, alpha, index_base=base)
    dx, du = wl.localization_masks_native(A, B2, d, T, alpha, ctx=gpu_ctx, index_base=base)
    for H, D in zip(hx + hu, dx + du):
        assert H.shape == D.shape and H.nnz == D.nnz
        assert np.array_equal(H.indptr, D.indptr) and np.array_equal(H.indices, D.indices)
    assert sum(M.nnz for M in dx) > 0
    if name == "readme_chain" and base == 0:                          # and against the SciPy recipe itself
        sx, su = wl.localization_masks(A, B2, d, T, alpha)
        for H, D in zip(sx + su, dx + du):
            assert np.array_equal(H.indptr, D.indptr) and np.array_equal(H.indices, D.indices)


@pytest.mark.parametrize("name,base", [("readme_chain", 0), ("readme_chain", 1), ("grid32", 0), ("random", 1)])
def test_device_index_sets_equal_host_reduction(slc, gpu_ctx, name, base):
    """sls_index_sets_device (one wave per column, csrc/sls_masks.hip) against the host sls_sparsity_dim_reduction for every
    column, and against the reference's known answer (test/reduction_test.jl:11-24 via tests/golden/reduction_known_answer.json)
    as sets: the device returns ascending order, the reference unique(findnz(...)) order."""
    import ctypes as C
    import json
    import os
    wl = slc.workloads
    if name == "random":
        A = sp.random(500, 500, density=0.008, random_state=5, format="csc") + sp.eye(500, format="csc")
        A.data[::5] = 0.0                                            # stored zeros are not part of (A .≠ 0)
        B2 = sp.random(500, 200, density=0.01, random_state=6, format="csc")
        d, T, alpha = 3, 7, 1.5
    else:
        mk, d, T, alpha = wl.WORKLOADS[name]
        P = mk(); A, B2 = P.A, P.B2
    Sx, Su = wl.localization_masks_native(A, B2, d, T, alpha)
    Sx_last, Su_last = Sx[-1].copy(), Su[-1].copy()
    if name == "random":
        Sx_last.data[::11] = False                                   # stored-false mask entries still count (findnz is structural)
    dx, du = wl.index_sets_device(gpu_ctx, A, Sx_last, Su_last, index_base=base)
    Nx = A.shape[0]
    assert len(dx) == Nx and sum(len(v) for v in dx) > 0
    # host function, column by column
    lib = slc.load_library(); cap = slc._capi
    i64p = C.POINTER(C.c_int64)

    def csc(M, cls, dt):
        M = sp.csc_matrix(M); M.sort_indices()
        cp = M.indptr.astype(np.int64) + base; rv = M.indices.astype(np.int64) + base
        nz = np.ascontiguousarray(M.data, dtype=dt)
        return cls(M.shape[0], M.shape[1], cp.ctypes.data_as(i64p), rv.ctypes.data_as(i64p),
                   nz.ctypes.data_as(C.POINTER(C.c_double if dt == np.float64 else C.c_uint8))), (cp, rv, nz)
    a, k1 = csc(A, cap.sls_csc_f64, np.float64)
    sx_, k2 = csc(Sx_last, cap.sls_csc_bool, np.uint8)
    su_, k3 = csc(Su_last, cap.sls_csc_bool, np.uint8)
    dims = cap.sls_dims(Nx, B2.shape[1], Nx + B2.shape[1], Nx, 1, base, 0)
    sx = np.zeros(Nx, dtype=np.int64); su = np.zeros(max(B2.shape[1], 1), dtype=np.int64)
    nsx, nsu = C.c_int64(), C.c_int64()
    step = 1 if Nx <= 600 else 7
    for c in range(0, Nx, step):
        cj = np.array([c + base], dtype=np.int64)
        rc = lib.sls_sparsity_dim_reduction(C.byref(dims), C.byref(a), C.byref(sx_), C.byref(su_), cj.ctypes.data_as(i64p), 1,
                                            sx.ctypes.data_as(i64p), C.byref(nsx), su.ctypes.data_as(i64p), C.byref(nsu))
        assert rc == 0
        assert np.array_equal(np.sort(sx[:nsx.value] - base), dx[c]), c
        assert np.array_equal(np.sort(su[:nsu.value] - base), du[c]), c
    if name == "readme_chain":                                        # the reference's own case: masks (A≠0)^9, B2ᵀ(A≠0)^9
        ka = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reduction_known_answer.json")))
        Ab = (A != 0).astype(np.int32).tocsc()
        R = sp.identity(Nx, dtype=np.int32, format="csc")
        for _ in range(9):
            R = ((R @ Ab) != 0).astype(np.int32).tocsc()
        S9 = (R != 0).tocsc(); U9 = ((((B2.T != 0).astype(np.int32)) @ R) != 0).tocsc()
        gx, gu = wl.index_sets_device(gpu_ctx, A, S9, U9, index_base=base)
        # a group's sets are the unions of its columns' sets (reduction.jl:17-20 over cⱼ)
        ux = sorted(set(int(i) for c in ka["cj"] for i in gx[c])); uu = sorted(set(int(i) for c in ka["cj"] for i in gu[c]))
        assert ux == sorted(ka["expected_sx"]) and uu == sorted(ka["expected_su"])


def _tables_localized(slc, ctx, P, d, T, alpha, index_base=0):
    m = slc._capi.Marshalled(P, [], [], None, index_base=index_base)
    m.dims.T = T
    n = C.c_int64(0); ni = C.c_int64(0)
    lib = ctx._lib
    rc = lib.sls_debug_plan_tables_localized(ctx.handle, 0, C.byref(m.dims), C.byref(m.plant), d, alpha, C.byref(n), None, None, C.byref(ni), None)
    assert rc == 0, slc._capi.last_error(ctx.handle)
    mask = np.zeros(n.value, dtype=np.uint8); dest = np.zeros(n.value, dtype=np.int32); idx = np.zeros(ni.value, dtype=np.int32)
    rc = lib.sls_debug_plan_tables_localized(ctx.handle, 0, C.byref(m.dims), C.byref(m.plant), d, alpha, C.byref(n),
                                             mask.ctypes.data_as(C.POINTER(C.c_uint8)), dest.ctypes.data_as(C.POINTER(C.c_int32)),
                                             C.byref(ni), idx.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0, slc._capi.last_error(ctx.handle)
    return mask, dest, idx


@pytest.mark.parametrize("name", ["readme_chain", "grid32", "random2000", "grid12_wide"])
def test_device_resident_route_builds_the_host_routes_tables(slc, gpu_ctx, oracle, name):
    """sls_h2_sf_plan_localized (round 3): index sets, mask slices and destinations of every column computed on the device from
    (A, B2, d, α, T) alone — bit-identical to the tables the host pass derives from the 2T mask arrays of the same recipe
    (sls_debug_plan_tables, host_tables = 1), in both index bases; the index sets against the reference's own recipe through
    the oracle (src/reduction.jl:14) on sampled columns."""
    if name == "readme_chain":
        P = slc.workloads.chain_plant(59); d, T, alpha = 9, 29, 1.5
    elif name == "grid32":
        P = slc.workloads.grid_plant(32, 3); d, T, alpha = 5, 20, 1.5
    elif name == "random2000":
        P = slc.workloads.random_plant(2000, 4, 2, 3); d, T, alpha = 2, 9, 1.5
    else:
        P = slc.workloads.grid_plant(12, 2); d, T, alpha = 5, 12, 1.5          # ñx + ñu > 64: several mask words per time step
    S = list(slc.workloads.localization_masks_native(P.A, P.B2, d, T, alpha))
    mh, dh, _ = _tables(slc, gpu_ctx, P, S, None, 1)
    for base in (0, 1):
        md, dd, idx = _tables_localized(slc, gpu_ctx, P, d, T, alpha, index_base=base)
        assert mh.size > 0 and np.array_equal(mh, md)
        assert np.array_equal(dh, dd)
    Po = oracle.OraclePlant(P.A, P.B1, P.B2)
    rng = np.random.default_rng(1)
    cols = sorted(set([0, P.Nx - 1] + [int(c) for c in rng.choice(P.Nx, 6, replace=False)]))
    sizes = np.diff(((S[0][-1].astype(np.int32)) @ (P.A != 0).astype(np.int32)).tocsc().indptr) + \
        np.diff(((S[1][-1].astype(np.int32)) @ (P.A != 0).astype(np.int32)).tocsc().indptr)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    assert offs[-1] == idx.size
    for c in cols:
        _, _, _, sx, su = oracle.sparsity_dim_reduction(Po, [c], S)
        assert np.array_equal(idx[offs[c]:offs[c + 1]], np.concatenate([np.sort(sx), np.sort(su)]))


def test_device_resident_route_solves_like_the_mask_route(slc, gpu_ctx, golden_readme):
    """The drop-in solve through the device-resident route (sls_h2_sf_solve_localized) and through the mask arrays
    (sls_h2_sf_solve): same tables → same launch list → the same Φ bit for bit; against the golden README vector to 1e-8; the
    resident plan (Plan.localized) likewise; and what the route does not cover is refused, not approximated."""
    P = slc.workloads.chain_plant(59)
    d, T, alpha = 9, 29, 1.5
    S = list(slc.workloads.localization_masks_native(P.A, P.B2, d, T, alpha))
    Px, Pu, info = slc.SLS_H2_localized(P, d, T, alpha, ctx=gpu_ctx, return_info=True, dropzeros=False)
    Qx, Qu, info2 = slc.SLS_H2(P, S, ctx=gpu_ctx, return_info=True, dropzeros=False)
    assert info["n_unsolved"] == 0 and info["n_subproblems"] == 59 and info["n_free"] == 36029
    want = np.concatenate([golden_readme["vals_x"], golden_readme["vals_u"]])
    got = np.concatenate([M.data for M in Px + Pu])
    ref = np.concatenate([M.data for M in Qx + Qu])
    assert np.array_equal(got, ref)
    assert np.abs(got - want).max() < 1e-8
    assert abs(info["flops_alg"] - info2["flops_alg"]) <= 1e-9 * info2["flops_alg"]
    plan = slc.Plan.localized(gpu_ctx, P, d, T, alpha)
    assert plan.info["n_values"] == 36029 and "twisted" in plan.describe()
    dv = plan.alloc_values(); plan.execute(dv); plan.synchronize()
    vx, vu = plan.download(dv)
    assert np.array_equal(np.concatenate(vx + vu), ref)
    with pytest.raises(slc.SLSError) as ei:                       # packed output is the mask route's
        plan.execute(dv, packed=True)
    plan.close()
    # grid-32 through both routes: statuses and values identical (tile kernel, infeasible columns included)
    Pg = slc.workloads.grid_plant(32, 3)
    Sg = list(slc.workloads.localization_masks_native(Pg.A, Pg.B2, 5, 20, 1.5))
    a = slc.SLS_H2_localized(Pg, 5, 20, 1.5, ctx=gpu_ctx, return_info=True, dropzeros=False)
    b = slc.SLS_H2(Pg, Sg, ctx=gpu_ctx, return_info=True, dropzeros=False)
    assert np.array_equal(a[2]["col_status"], b[2]["col_status"])
    ok = b[2]["col_status"] == 0
    colidx = np.concatenate([np.repeat(np.arange(Pg.Nx), np.diff(M.indptr)) for M in Sg[0] + Sg[1]])
    va = np.concatenate([M.data for M in a[0] + a[1]]); vb = np.concatenate([M.data for M in b[0] + b[1]])
    assert np.array_equal(va[ok[colidx]], vb[ok[colidx]])
    # non-default cost weights are the mask route's
    import scipy.sparse as sp2
    C1 = sp2.vstack([sp2.diags(np.full(59, 2.0)), sp2.csc_matrix((P.Nu, 59))]).tocsc()
    D12 = sp2.vstack([sp2.csc_matrix((59, P.Nu)), sp2.eye(P.Nu)]).tocsc()
    Pw = slc.Plant(P.A, P.B1, P.B2, C1, 0, D12)
    with pytest.raises(slc.SLSError) as ei:
        slc.SLS_H2_localized(Pw, d, T, alpha, ctx=gpu_ctx)
    assert ei.value.code == slc._capi.SLS_EUNSUPPORTED
    # a plant without a full diagonal: exact-walk level sets are not nested, a mask row falls outside the index set
    A2 = sp2.diags([np.ones(58), np.ones(58)], [1, -1]).tocsc()
    Pn = slc.Plant(A2, sp2.eye(59, format="csc"), P.B2)
    with pytest.raises(slc.SLSError) as ei:
        slc.SLS_H2_localized(Pn, 4, 10, 1.5, ctx=gpu_ctx)
    assert ei.value.code == slc._capi.SLS_EUNSUPPORTED
