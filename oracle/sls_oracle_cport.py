"""ctypes front-end of the C restatement (oracle/sls_oracle_c.c) — TEST INFRASTRUCTURE ONLY.

Builds, with the NumPy oracle's own `sparsity_dim_reduction` (reference src/reduction.jl:11-27), the dense
reduced blocks of every column and hands them to the C batch solver (OpenMP over columns = the reference's
`julia -p N` analogue).  Used by tests (second, independent-of-NumPy-SVD checker at sizes the SVD oracle is
too slow for) and by bench.py's `cpu_baseline` leg ("port").  Default plant weights only (H = I, g = 0)."""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import scipy.sparse as sp

import sls_oracle as o

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("SLS_ORACLE_LIB") or os.path.join(_HERE, "_build", "libsls_oracle.so")   # override: the sanitizer build of tests/test_oracle.py


def load():
    if not os.path.exists(LIB):
        raise ImportError(f"{LIB} missing: run `make -C oracle`")
    lib = C.CDLL(LIB)
    i32p, i64p, dp, u8p = (C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_uint8))
    lib.sls_oracle_solve_batch.restype = C.c_int
    lib.sls_oracle_solve_batch.argtypes = [C.c_int, C.c_int, i32p, i32p, i32p, i64p, i64p, i64p, i64p, i64p,
                                           dp, dp, u8p, dp, dp, dp, i32p, i32p, C.c_int]
    return lib


def prepare(P, S, cols):
    """Dense per-column inputs (this is the part the reference does with view()/JuMP model building)."""
    Sx, Su = S
    T = len(Sx)
    recs = []
    for c in cols:
        sub, It, iix, sx, su = o.sparsity_dim_reduction(P, [c], [Sx, Su])
        n, m = len(sx), len(su)
        mask = np.zeros((T, n + m), dtype=np.uint8)
        for t in range(T):
            mask[t, :n] = o._mask_block(Sx[t], sx, [c]).ravel()
            mask[t, n:] = o._mask_block(Su[t], su, [c]).ravel()
        pos = int(np.flatnonzero(sx == c)[0]) if c in sx else -1
        recs.append(dict(c=c, n=n, m=m, pos=pos, A=np.ascontiguousarray(sub["A"]), B=np.ascontiguousarray(sub["B2"]),
                         mask=mask, sx=sx, su=su))
    return recs


def solve_batch(recs, T, nthreads=0):
    """Returns (list of (x[T,n], u[T,m]), resid, status, iters, seconds, threads)."""
    lib = load()
    nc = len(recs)
    n = np.array([r["n"] for r in recs], dtype=np.int32); m = np.array([r["m"] for r in recs], dtype=np.int32)
    pos = np.array([r["pos"] for r in recs], dtype=np.int32)
    offA = np.zeros(nc, dtype=np.int64); offB = np.zeros(nc, dtype=np.int64); offM = np.zeros(nc, dtype=np.int64)
    offX = np.zeros(nc, dtype=np.int64); offU = np.zeros(nc, dtype=np.int64)
    a = b = k = x = u = 0
    for i, r in enumerate(recs):
        offA[i], offB[i], offM[i], offX[i], offU[i] = a, b, k, x, u
        a += r["n"] ** 2; b += r["n"] * r["m"]; k += T * (r["n"] + r["m"]); x += T * r["n"]; u += T * r["m"]
    poolA = np.concatenate([r["A"].ravel() for r in recs]); poolB = np.concatenate([r["B"].ravel() for r in recs] + [np.zeros(1)])
    poolM = np.concatenate([r["mask"].ravel() for r in recs])
    poolX = np.zeros(max(x, 1)); poolU = np.zeros(max(u, 1))
    resid = np.zeros(nc); status = np.zeros(nc, dtype=np.int32); iters = np.zeros(nc, dtype=np.int32)
    p = lambda arr, t: arr.ctypes.data_as(C.POINTER(t))
    t0 = time.perf_counter()
    used = lib.sls_oracle_solve_batch(nc, T, p(n, C.c_int32), p(m, C.c_int32), p(pos, C.c_int32), p(offA, C.c_int64),
                                      p(offB, C.c_int64), p(offM, C.c_int64), p(offX, C.c_int64), p(offU, C.c_int64),
                                      p(poolA, C.c_double), p(poolB, C.c_double), p(poolM, C.c_uint8),
                                      p(poolX, C.c_double), p(poolU, C.c_double), p(resid, C.c_double),
                                      p(status, C.c_int32), p(iters, C.c_int32), int(nthreads))
    dt = time.perf_counter() - t0
    out = [(poolX[offX[i]:offX[i] + T * r["n"]].reshape(T, r["n"]), poolU[offU[i]:offU[i] + T * r["m"]].reshape(T, r["m"]))
           for i, r in enumerate(recs)]
    return out, resid, status, iters, dt, used


def SLS_H2(P, S, cols=None, nthreads=0):
    """Φx, Φu (lists of CSC) from the C restatement; default weights, one group per column."""
    Sx, Su = S
    T = len(Sx)
    cols = list(range(P.Nx)) if cols is None else list(cols)
    recs = prepare(P, S, cols)
    sols, resid, status, iters, dt, used = solve_batch(recs, T, nthreads)
    Px = [sp.lil_matrix((P.Nx, P.Nx)) for _ in range(T)]
    Pu = [sp.lil_matrix((P.Nu, P.Nx)) for _ in range(T)]
    for r, (x, u) in zip(recs, sols):
        for t in range(T):
            for i in np.flatnonzero(r["mask"][t, : r["n"]]):
                Px[t][r["sx"][i], r["c"]] = x[t, i]
            for i in np.flatnonzero(r["mask"][t, r["n"]:]):
                Pu[t][r["su"][i], r["c"]] = u[t, i]
    return [M.tocsc() for M in Px], [M.tocsc() for M in Pu], dict(resid=resid, status=status, iters=iters, seconds=dt, threads=used)
