"""Synthetic plants and (d,T)-localization masks used by the tests and by bench.py.

None of this is in the reference package: the README builds its plant and masks in the
user script (README.md:43-54).  The recipes here restate that script and extend it to the
BASELINE.json configurations:
  configs[0]/[1]  README chain           Nx=59,  Nu=20, d=9,  T=29, α=1.5
  configs[2]      2-D grid               Nx=n², actuators every `act_every`-th state, d=5, T=20
  configs[3]      long chain             Nx=4096, d=12, T=40      (𝓗₂ on the 𝓗∞ config's plant)
  configs[4]      random sparse A        Nx=10000, avg degree 4
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .plant import Plant


def chain_plant(Nx=59, Nu=None):
    """README.md:43-47: A = I + 0.2·superdiag − 0.2·subdiag; B1 = I; B2 = I[:, {6n+1, 6n+2}]."""
    A = (sp.identity(Nx) + sp.diags(0.2 * np.ones(Nx - 1), 1) - sp.diags(0.2 * np.ones(Nx - 1), -1)).tocsc()
    cols = [6 * n + k for n in range((Nx + 5) // 6) for k in (0, 1) if 6 * n + k < Nx]
    if Nu is None:
        Nu = 20 if Nx == 59 else len(cols)
    B2 = sp.identity(Nx, format="csc")[:, cols[:Nu]]
    return Plant(A, sp.identity(Nx, format="csc"), B2)


def grid_plant(n=32, act_every=3):
    """n×n 4-neighbour grid, A = I ± 0.2 on links (antisymmetric like the chain),
    actuators on every `act_every`-th state."""
    N = n * n
    rows, cols, vals = [], [], []
    for i in range(n):
        for j in range(n):
            k = i * n + j
            rows.append(k); cols.append(k); vals.append(1.0)
            for di, dj in ((0, 1), (1, 0)):
                ii, jj = i + di, j + dj
                if ii < n and jj < n:
                    k2 = ii * n + jj
                    rows += [k, k2]; cols += [k2, k]; vals += [0.2, -0.2]
    A = sp.csc_matrix((vals, (rows, cols)), shape=(N, N))
    B2 = sp.identity(N, format="csc")[:, list(range(0, N, act_every))]
    return Plant(A, sp.identity(N, format="csc"), B2)


def random_plant(Nx=10000, avg_degree=4, act_every=2, seed=1):
    """Random sparse A (≈avg_degree off-diagonal nonzeros per column, unit diagonal),
    actuators on every `act_every`-th state; explicit seed."""
    rng = np.random.default_rng(seed)
    nnz = int(Nx * avg_degree)
    r = rng.integers(0, Nx, nnz); c = rng.integers(0, Nx, nnz)
    v = rng.uniform(-0.3, 0.3, nnz)
    keep = r != c
    A = sp.csc_matrix((v[keep], (r[keep], c[keep])), shape=(Nx, Nx)) + sp.identity(Nx)
    A = sp.csc_matrix(A); A.sum_duplicates()
    B2 = sp.identity(Nx, format="csc")[:, list(range(0, Nx, act_every))]
    return Plant(A, sp.identity(Nx, format="csc"), B2)


def _bool_power(Mb, k):
    R = sp.identity(Mb.shape[0], dtype=np.int32, format="csc")
    for _ in range(int(k)):
        R = ((R @ Mb) != 0).astype(np.int32).tocsc()
    return R


def localization_masks(A, B2, d, T, alpha):
    """README.md:53-54 with 1-based t = 1..T:
         𝓢x[t] = (A≠0)^min(d,   ⌊α(t−1)⌋) ≠ 0
         𝓢u[t] = (B2'≠0)·(A≠0)^min(d+1, ⌊α(t−1)⌋) ≠ 0
    Returns two lists of boolean CSC matrices with sorted indices."""
    Ab = (sp.csc_matrix(A) != 0).astype(np.int32).tocsc()
    Bb = (sp.csc_matrix(B2).T != 0).astype(np.int32).tocsc()
    cache = {}
    Sx, Su = [], []
    for t in range(T):
        kx = min(d, int(np.floor(alpha * t)))
        ku = min(d + 1, int(np.floor(alpha * t)))
        for k in (kx, ku):
            if k not in cache:
                prev = max((q for q in cache if q < k), default=None)
                base = cache[prev] if prev is not None else sp.identity(Ab.shape[0], dtype=np.int32, format="csc")
                R = base
                for _ in range(k - (prev or 0)):
                    R = ((R @ Ab) != 0).astype(np.int32).tocsc()
                cache[k] = R
        sx = (cache[kx] != 0).tocsc(); sx.sort_indices()
        su = ((Bb @ cache[ku]) != 0).tocsc(); su.sort_indices()
        Sx.append(sx); Su.append(su)
    return Sx, Su


def localization_masks_native(A, B2, d, T, alpha, ctx=None, index_base=0):
    """Same masks as `localization_masks` (README.md:53-54), computed by the library's host-threaded level-set expansion
    (sls_localization_masks) instead of SciPy Boolean matrix powers — ≈40× faster at Nx = 4096 — or, with a Context, by the
    device kernels of sls_localization_masks_device (csrc/sls_masks.hip)."""
    import ctypes as C
    from . import _capi
    lib = _capi.load_library()
    if ctx is not None:
        def call(*a):
            return lib.sls_localization_masks_device(ctx.handle, 0, *a)
    else:
        call = lib.sls_localization_masks
    A = sp.csc_matrix(A, dtype=np.float64); B2 = sp.csc_matrix(B2, dtype=np.float64)
    A.sort_indices(); B2.sort_indices()
    Nx, Nu = A.shape[0], B2.shape[1]
    keep = []

    def f64(M):
        cp = np.ascontiguousarray(M.indptr, dtype=np.int64) + index_base
        rv = np.ascontiguousarray(M.indices, dtype=np.int64) + index_base
        nz = np.ascontiguousarray(M.data, dtype=np.float64)
        keep.extend([cp, rv, nz])
        i64p = C.POINTER(C.c_int64)
        return _capi.sls_csc_f64(M.shape[0], M.shape[1], cp.ctypes.data_as(i64p), rv.ctypes.data_as(i64p),
                                 nz.ctypes.data_as(C.POINTER(C.c_double)))
    a, b = f64(A), f64(B2)
    dims = _capi.sls_dims(Nx, Nu, Nx + Nu, Nx, T, index_base, 0)
    nx = np.zeros(T, dtype=np.int64); nu = np.zeros(T, dtype=np.int64)
    i64p = C.POINTER(C.c_int64)
    _capi.check(call(C.byref(dims), C.byref(a), C.byref(b), int(d), float(alpha),
                     nx.ctypes.data_as(i64p), nu.ctypes.data_as(i64p), None, None, None, None), ctx.handle if ctx else None)
    cpx = [np.zeros(Nx + 1, dtype=np.int64) for _ in range(T)]; rvx = [np.zeros(max(int(k), 1), dtype=np.int64) for k in nx]
    cpu = [np.zeros(Nx + 1, dtype=np.int64) for _ in range(T)]; rvu = [np.zeros(max(int(k), 1), dtype=np.int64) for k in nu]
    arr = lambda lst: (i64p * T)(*[x.ctypes.data_as(i64p) for x in lst])
    _capi.check(call(C.byref(dims), C.byref(a), C.byref(b), int(d), float(alpha),
                     nx.ctypes.data_as(i64p), nu.ctypes.data_as(i64p), arr(cpx), arr(rvx), arr(cpu), arr(rvu)), ctx.handle if ctx else None)
    b0 = index_base
    Sx = [sp.csc_matrix((np.ones(int(nx[t]), dtype=bool), rvx[t][: int(nx[t])] - b0, cpx[t] - b0), shape=(Nx, Nx)) for t in range(T)]
    Su = [sp.csc_matrix((np.ones(int(nu[t]), dtype=bool), rvu[t][: int(nu[t])] - b0, cpu[t] - b0), shape=(Nu, Nx)) for t in range(T)]
    return Sx, Su


def index_sets_device(ctx, A, Sx_last, Su_last, index_base=0):
    """s_x(c), s_u(c) of every single-column subproblem (reference src/reduction.jl:11-27 with cⱼ = {c}) from the device pass
    sls_index_sets_device (csrc/sls_masks.hip).  Returns two lists of ascending 0-based index arrays."""
    import ctypes as C
    from . import _capi
    lib = _capi.load_library()
    i64p = C.POINTER(C.c_int64)
    keep = []

    def csc(M, cls, dt):
        M = sp.csc_matrix(M); M.sort_indices()
        cp = np.ascontiguousarray(M.indptr, dtype=np.int64) + index_base
        rv = np.ascontiguousarray(M.indices, dtype=np.int64) + index_base
        nz = np.ascontiguousarray(M.data, dtype=dt)
        keep.extend([cp, rv, nz])
        return cls(M.shape[0], M.shape[1], cp.ctypes.data_as(i64p), rv.ctypes.data_as(i64p),
                   nz.ctypes.data_as(C.POINTER(C.c_double if dt == np.float64 else C.c_uint8)))
    a = csc(A, _capi.sls_csc_f64, np.float64)
    sx = csc(Sx_last, _capi.sls_csc_bool, np.uint8); su = csc(Su_last, _capi.sls_csc_bool, np.uint8)
    Nx, Nu = a.nrows, su.nrows
    dims = _capi.sls_dims(Nx, Nu, Nx + Nu, Nx, 1, index_base, 0)
    px = np.zeros(Nx + 1, dtype=np.int64); pu = np.zeros(Nx + 1, dtype=np.int64)
    _capi.check(lib.sls_index_sets_device(ctx.handle, 0, C.byref(dims), C.byref(a), C.byref(sx), C.byref(su),
                                          px.ctypes.data_as(i64p), None, pu.ctypes.data_as(i64p), None), ctx.handle)
    ix = np.zeros(max(int(px[-1]) - index_base, 1), dtype=np.int64); iu = np.zeros(max(int(pu[-1]) - index_base, 1), dtype=np.int64)
    _capi.check(lib.sls_index_sets_device(ctx.handle, 0, C.byref(dims), C.byref(a), C.byref(sx), C.byref(su),
                                          px.ctypes.data_as(i64p), ix.ctypes.data_as(i64p), pu.ctypes.data_as(i64p),
                                          iu.ctypes.data_as(i64p)), ctx.handle)
    px -= index_base; pu -= index_base
    return ([ix[px[c]:px[c + 1]] - index_base for c in range(Nx)], [iu[pu[c]:pu[c + 1]] - index_base for c in range(Nx)])


WORKLOADS = {
    # name: (plant factory, d, T, alpha)
    "readme_chain": (lambda: chain_plant(59), 9, 29, 1.5),
    "grid32": (lambda: grid_plant(32, 3), 5, 20, 1.5),
    "grid32_dense_act": (lambda: grid_plant(32, 2), 5, 20, 1.5),
    "chain4096": (lambda: chain_plant(4096), 12, 40, 1.5),
    "chain1024": (lambda: chain_plant(1024), 12, 40, 1.5),
    "chain4096_T12": (lambda: chain_plant(4096), 12, 12, 1.5),      # short horizon (infeasible; occupancy experiments only)
    "random10000_d2": (lambda: random_plant(10000, 4, 2, 1), 2, 25, 1.5),
    # BASELINE configs[4] family with an actuator on every state: same random graph, same index sets (ñx ≤ 322), but every
    # column is FEASIBLE (with every 2nd state actuated 9 896 of 10 000 are not) — the workload that carries value parity on
    # the large-block paths.  Deviation from configs[4] as written: d = 2 (d = 6 is not localized) and B2 = I.
    "random10000_d2_act1": (lambda: random_plant(10000, 4, 1, 1), 2, 25, 1.5),
    "chain512_d20": (lambda: chain_plant(512), 20, 46, 1.5),        # ñx = 43: the <64,48> class of the one-wave kernel
    "chain512_d28": (lambda: chain_plant(512), 28, 62, 1.5),        # ñx = 59: the <64,64> class
}


def make_workload(name):
    fac, d, T, alpha = WORKLOADS[name]
    P = fac()
    Sx, Su = localization_masks_native(P.A, P.B2, d, T, alpha)
    return P, [Sx, Su], dict(name=name, d=d, T=T, alpha=alpha, Nx=P.Nx, Nu=P.Nu)
