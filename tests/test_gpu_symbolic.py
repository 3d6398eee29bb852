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
