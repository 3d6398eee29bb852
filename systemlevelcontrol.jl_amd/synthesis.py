"""SLS_𝓗₂ — host-side mirror of the reference entry point, backed by the HIP engine.

Reference: src/synthesis.jl
  :11      SLS_𝓗₂(P::AbstractGeneralizedPlant, 𝓢::AbstractVector; 𝓘=nothing)
  :13,30   non-StateFeedback plant ⇒ returns `nothing`
  :15      default 𝓘 = [[i] for i in 1:P.Nx]
  :16      static contiguous partition of the groups over nworkers()
  :24-27   @distributed (+) over the chunks; returns eachcol(Φ) → (Φx, Φu)
The per-column work (src/synthesis.jl:34-72, src/reduction.jl:11-27) runs inside
libsls_mi355x.so; this module only marshals.  ('₂' is not a legal Python identifier
character, so the function is spelled `SLS_H2`.)
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np
import scipy.sparse as sp

from . import _capi
from .plant import GeneralizedPlant, StateFeedback


class Context:
    """Owns the HIP devices used by a solve — the analogue of the `julia -p N` worker pool."""

    def __init__(self, devices=None):
        lib = _capi.load_library()
        if devices is None:
            devices = [0]
        arr = (C.c_int * len(devices))(*devices)
        self._lib = lib
        self.devices = list(devices)
        self.handle = lib.sls_create(arr, len(devices), 0)
        if not self.handle:
            raise _capi.SLSError(_capi.SLS_ENODEVICE, _capi.last_error(None))

    def set_ridge(self, rx=None, ru=None):
        """Ridge term Σ_t Σ_i rx[i]·Φx[t][i,c]² + Σ_j ru[j]·Φu[t][j,c]² added to every column's cost in later solves on this
        context (sls_set_ridge: the diagonal instance of the reference's L⁺ hook, src/synthesis.jl:21,52).  None/None clears."""
        rx = np.zeros(0) if rx is None else np.ascontiguousarray(rx, dtype=np.float64)
        ru = np.zeros(0) if ru is None else np.ascontiguousarray(ru, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _capi.check(self._lib.sls_set_ridge(self.handle, rx.size, rx.ctypes.data_as(dp) if rx.size else None,
                                            ru.size, ru.ctypes.data_as(dp) if ru.size else None), self.handle)

    def close(self):
        if getattr(self, "handle", None):
            self._lib.sls_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None or not _default_ctx.handle:
        _default_ctx = Context([0])
    return _default_ctx


class Plan:
    """Symbolic pass + device-resident tables for a shard of the groups (sls_h2_sf_plan)."""

    def __init__(self, ctx: Context, P, S, groups=None, group_range=None, dev_slot=0, objective="h2"):
        Sx, Su = S
        self.ctx = ctx
        self._lib = ctx._lib
        self.m = _capi.Marshalled(P, Sx, Su, groups, flags=_objective_flags(objective))
        ng = self.m.ngroups if groups is not None else P.Nx
        gb, ge = (0, ng) if group_range is None else group_range
        h = C.c_void_p()
        _capi.check(self._lib.sls_h2_sf_plan(ctx.handle, dev_slot, *self.m.common_args(), gb, ge, C.byref(h)),
                    ctx.handle)
        self.handle = h
        info = _capi.sls_plan_info()
        _capi.check(self._lib.sls_plan_get_info(self.handle, C.byref(info)))
        self.info = info.asdict()
        T = self.info["T"]
        ox = np.zeros(T + 1, dtype=np.int64); ou = np.zeros(T + 1, dtype=np.int64)
        _capi.check(self._lib.sls_plan_value_offsets(self.handle, ox.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     ou.ctypes.data_as(C.POINTER(C.c_int64))))
        self.off_x, self.off_u = ox, ou
        self._owned_values = []

    @classmethod
    def localized(cls, ctx, P, d, T, alpha, dev_slot=0, objective="h2", index_base=0):
        """sls_h2_sf_plan_localized: the plan of the README's (d, T, α) localization built from the plant alone — level sets,
        index sets, mask slices and destinations computed on the device, no mask array crosses PCIe.  Values come back in the
        CSC order of `workloads.localization_masks_native(P.A, P.B2, d, T, alpha)`."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self._lib = ctx._lib
        empty = [sp.csc_matrix((P.Nx, P.Nx), dtype=bool)] * 0
        self.m = _capi.Marshalled(P, empty, empty, None, index_base=index_base, flags=_objective_flags(objective))
        self.m.dims.T = int(T)
        h = C.c_void_p()
        _capi.check(self._lib.sls_h2_sf_plan_localized(ctx.handle, dev_slot, C.byref(self.m.dims), C.byref(self.m.plant), int(d),
                                                       float(alpha), C.byref(h)), ctx.handle)
        self.handle = h
        info = _capi.sls_plan_info()
        _capi.check(self._lib.sls_plan_get_info(self.handle, C.byref(info)))
        self.info = info.asdict()
        ox = np.zeros(T + 1, dtype=np.int64); ou = np.zeros(T + 1, dtype=np.int64)
        _capi.check(self._lib.sls_plan_value_offsets(self.handle, ox.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     ou.ctypes.data_as(C.POINTER(C.c_int64))))
        self.off_x, self.off_u = ox, ou
        self.m.nnz_x = [int(v) for v in np.diff(ox)]
        self.m.nnz_u = [int(v) for v in np.diff(ou)]
        self._owned_values = []
        return self

    # -- device buffers managed by the library (for callers without torch) --
    def alloc_values(self, packed=False):
        p = C.c_void_p()
        _capi.check(self._lib.sls_plan_alloc_values(self.handle, int(packed), C.byref(p)), self.ctx.handle)
        self._owned_values.append(p.value)
        return p.value

    def execute(self, d_values, packed=False, stream=None):
        _capi.check(self._lib.sls_plan_execute(self.handle, stream, d_values, int(packed)), self.ctx.handle)

    def synchronize(self, stream=None):
        _capi.check(self._lib.sls_plan_synchronize(self.handle, stream), self.ctx.handle)

    def refine(self, d_values, packed=False, stream=None):
        """sls_plan_refine: after an execute, re-solve the near-singular columns of the one-wave / twisted kernels on the tile
        kernel and attach that pass to the plan (later executes run it too).  Returns the number of subproblems refined."""
        n = C.c_int64()
        _capi.check(self._lib.sls_plan_refine(self.handle, *self.m.common_args(), stream, d_values, int(packed), C.byref(n)), self.ctx.handle)
        return n.value

    def packed_dest(self):
        d = np.zeros(max(self.info["n_packed"], 1), dtype=np.int64)
        _capi.check(self._lib.sls_plan_packed_dest(self.handle, d.ctypes.data_as(C.POINTER(C.c_int64))))
        return d[: self.info["n_packed"]]

    def fetch_status(self):
        n = self.info["n_subproblems"]
        st = np.zeros(max(n, 1), dtype=np.int32); it = np.zeros(max(n, 1), dtype=np.int32)
        rs = np.zeros(max(n, 1), dtype=np.float64)
        _capi.check(self._lib.sls_plan_fetch_status(self.handle, st.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    rs.ctypes.data_as(C.POINTER(C.c_double)),
                                                    it.ctypes.data_as(C.POINTER(C.c_int32))), self.ctx.handle)
        return st[:n], rs[:n], it[:n]

    def describe(self):
        buf = C.create_string_buffer(4096)
        _capi.check(self._lib.sls_plan_describe(self.handle, buf, len(buf)))
        return buf.value.decode()

    def kernel_time_ms(self):
        avg = C.c_double(); n = C.c_int64()
        _capi.check(self._lib.sls_plan_kernel_time_ms(self.handle, C.byref(avg), C.byref(n)), self.ctx.handle)
        return avg.value, n.value

    def download(self, d_values):
        """D2H of a mask-order value array → (list of T arrays for Φx, list of T arrays for Φu)."""
        T = self.info["T"]
        vx = [np.zeros(max(n, 1), dtype=np.float64) for n in self.m.nnz_x]
        vu = [np.zeros(max(n, 1), dtype=np.float64) for n in self.m.nnz_u]
        px = (C.POINTER(C.c_double) * T)(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in vx])
        pu = (C.POINTER(C.c_double) * T)(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in vu])
        _capi.check(self._lib.sls_plan_download(self.handle, d_values, px, pu), self.ctx.handle)
        return ([a[:n] for a, n in zip(vx, self.m.nnz_x)], [a[:n] for a, n in zip(vu, self.m.nnz_u)])

    def close(self):
        if getattr(self, "handle", None):
            for p in self._owned_values:
                self._lib.sls_plan_free_values(self.handle, p)
            self._owned_values = []
            self._lib.sls_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def assemble_phi(Sx, Su, vals_x, vals_u, dropzeros=True):
    """Φx[t] = SparseMatrixCSC(Nx,Nx, 𝓢x[t].colptr, 𝓢x[t].rowval, vals) (+ dropzeros!, which is what the
    reference's sparse `+` accumulation does to numerical zeros: src/synthesis.jl:65-67)."""
    def build(Sm, v):
        Sm = sp.csc_matrix(Sm)
        if not Sm.has_sorted_indices:
            Sm = Sm.copy(); Sm.sort_indices()
        M = sp.csc_matrix((np.asarray(v, dtype=np.float64).copy(), Sm.indices.copy(), Sm.indptr.copy()), shape=Sm.shape)
        if dropzeros:
            M.eliminate_zeros()
        return M
    return [build(s, v) for s, v in zip(Sx, vals_x)], [build(s, v) for s, v in zip(Su, vals_u)]


def execute_batch(plans, d_values, packed=False, stream=None):
    """Several independent plants in one call (sls_plan_execute_batch): plan i writes d_values[i]; the launches overlap on the
    device, `stream` (None = the null stream) sees all results.  The reference's counterpart is one SLS_𝓗₂ call per plant."""
    n = len(plans)
    if n != len(d_values):
        raise ValueError("one value array per plan")
    if n == 0:
        return
    hp = (C.c_void_p * n)(*[p.handle for p in plans])
    dv = (C.c_void_p * n)(*[int(v) for v in d_values])
    _capi.check(plans[0]._lib.sls_plan_execute_batch(hp, n, stream, dv, int(packed)), plans[0].ctx.handle)


def SLS_H2_batch(plants, masks, *, ctx: Context | None = None, return_info=False, dropzeros=True, objective="h2", index_base=0):
    """[(Φx, Φu) for each plant] = one sls_h2_sf_solve_batch call over independent plants that share T: a loop of reference
    `SLS_𝓗₂(P, 𝓢)` calls (src/synthesis.jl:11) run as ONE set of kernel launches (the block-diagonal composite plant)."""
    n = len(plants)
    if n == 0 or n != len(masks):
        raise ValueError("one [𝓢x, 𝓢u] per plant")
    ctx = ctx or default_context()
    lib = ctx._lib
    ms = [_capi.Marshalled(P, S[0], S[1], None, index_base=index_base, flags=_objective_flags(objective)) for P, S in zip(plants, masks)]
    T = len(masks[0][0])
    dims = (_capi.sls_dims * n)(*[m.dims for m in ms])
    pl = (_capi.sls_plant * n)(*[m.plant for m in ms])
    bp = C.POINTER(_capi.sls_csc_bool)
    sx = (bp * n)(*[C.cast(m.Sx, bp) for m in ms]); su = (bp * n)(*[C.cast(m.Su, bp) for m in ms])
    dp, i32p = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    dpp = C.POINTER(dp)
    vals, keep = [], []
    pxs, pus, sts = (dpp * n)(), (dpp * n)(), (i32p * n)()
    status = []
    for i, m in enumerate(ms):
        vx = [np.zeros(max(k, 1), dtype=np.float64) for k in m.nnz_x]
        vu = [np.zeros(max(k, 1), dtype=np.float64) for k in m.nnz_u]
        px = (dp * T)(*[a.ctypes.data_as(dp) for a in vx]); pu = (dp * T)(*[a.ctypes.data_as(dp) for a in vu])
        st = np.zeros(max(m.n_sub, 1), dtype=np.int32)
        keep += [px, pu]
        pxs[i] = C.cast(px, dpp); pus[i] = C.cast(pu, dpp); sts[i] = st.ctypes.data_as(i32p)
        vals.append((vx, vu)); status.append(st)
    stats = _capi.sls_stats()
    rc = lib.sls_h2_sf_solve_batch(ctx.handle, n, dims, pl, sx, su, pxs, pus, sts, C.byref(stats))
    _capi.check(rc, ctx.handle)
    out = []
    for m, S, (vx, vu) in zip(ms, masks, vals):
        out.append(assemble_phi(S[0], S[1], [a[:k] for a, k in zip(vx, m.nnz_x)], [a[:k] for a, k in zip(vu, m.nnz_u)],
                                dropzeros=dropzeros))
    if return_info:
        info = stats.asdict()
        info["col_status"] = [st[: m.n_sub].copy() for st, m in zip(status, ms)]
        info["n_unsolved"] = rc
        return out, info
    if rc > 0:
        warnings.warn(f"SLS_H2_batch: {rc} subproblems not solved — pass return_info=True for the per-column status",
                      RuntimeWarning, stacklevel=2)
    return out


def SLS_H2_localized(P, d, T, alpha, *, ctx: Context | None = None, return_info=False, dropzeros=True, objective="h2", index_base=0):
    """Φx, Φu = SLS_𝓗₂(P, [𝓢x, 𝓢u]) for the README's own masks (README.md:52-54), given as (d, T, α) instead of 2T sparse
    matrices: sls_h2_sf_solve_localized builds index sets, mask slices and destinations on the device.  The patterns needed to
    return Φ as sparse matrices are fetched with the device mask recipe (row indices only come down)."""
    if not isinstance(P, GeneralizedPlant) or P.Ts is not StateFeedback:
        return None
    from .workloads import localization_masks_native
    ctx = ctx or default_context()
    lib = ctx._lib
    Sx, Su = localization_masks_native(P.A, P.B2, d, T, alpha, ctx=ctx)
    empty = []
    m = _capi.Marshalled(P, empty, empty, None, index_base=index_base, flags=_objective_flags(objective))
    m.dims.T = int(T)
    nnz_x = [int(M.nnz) for M in Sx]; nnz_u = [int(M.nnz) for M in Su]
    vx = [np.zeros(max(n, 1), dtype=np.float64) for n in nnz_x]
    vu = [np.zeros(max(n, 1), dtype=np.float64) for n in nnz_u]
    px = (C.POINTER(C.c_double) * T)(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in vx])
    pu = (C.POINTER(C.c_double) * T)(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in vu])
    status = np.zeros(max(P.Nx, 1), dtype=np.int32)
    stats = _capi.sls_stats()
    rc = lib.sls_h2_sf_solve_localized(ctx.handle, C.byref(m.dims), C.byref(m.plant), int(d), float(alpha), px, pu,
                                       status.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(stats))
    _capi.check(rc, ctx.handle)
    st = stats.asdict()
    if st["n_values_x"] != sum(nnz_x) or st["n_values_u"] != sum(nnz_u):
        raise RuntimeError("device symbolic route and device mask recipe disagree on the pattern sizes")
    Phix, Phiu = assemble_phi(Sx, Su, [a[:n] for a, n in zip(vx, nnz_x)], [a[:n] for a, n in zip(vu, nnz_u)], dropzeros=dropzeros)
    if return_info:
        st["col_status"] = status[: P.Nx].copy()
        st["n_unsolved"] = rc
        return Phix, Phiu, st
    if rc > 0:
        warnings.warn(f"SLS_H2_localized: {rc} of {P.Nx} subproblems not solved — pass return_info=True for the per-column status",
                      RuntimeWarning, stacklevel=2)
    return Phix, Phiu


def _objective_flags(objective):
    if objective in ("h2", None):
        return _capi.SLS_SOLVE_DEFAULT
    if objective == "sum_of_norms":
        return _capi.SLS_SOLVE_SUM_OF_NORMS
    raise ValueError(f"unknown objective {objective!r}")


def SLS_Hinf_bound(P, S, I=None, **kw):
    """Φx, Φu minimising, per column, Σ_t ‖[C̃1 D̃12]Φ̃[t]B̃1‖₂ — the column-separable bound of the 𝓗∞ norm — over the same
    localized constraints as SLS_𝓗₂ (SLS_SOLVE_SUM_OF_NORMS).  NOT in the reference: it has no 𝓗∞ synthesis;
    BASELINE.json configs[3] names one.  Diagonal weights, D11 = 0."""
    return SLS_H2(P, S, I, objective="sum_of_norms", **kw)


def SLS_H2(P, S, I=None, *, ctx: Context | None = None, return_info=False, dropzeros=True, index_base=0, objective="h2"):
    """Φx, Φu = SLS_𝓗₂(P, [𝓢x, 𝓢u]; 𝓘)   — drop-in for reference src/synthesis.jl:11.

    P : GeneralizedPlant (state feedback).  Any other feedback structure returns None,
        exactly like the reference (src/synthesis.jl:13,30-32).
    S : [𝓢x, 𝓢u], two length-T lists of boolean sparse matrices (Nx×Nx, Nu×Nx).
    I : optional list of column groups (0-based column indices, ascending inside a group).
    index_base : 0, or 1 to marshal every index array the way Julia stores it (what the `ccall` binding hands over).
    Returns two length-T lists of scipy CSC matrices (Φx[t] Nx×Nx, Φu[t] Nu×Nx).
    """
    if not isinstance(P, GeneralizedPlant) or P.Ts is not StateFeedback:
        return None
    Sx, Su = S
    ctx = ctx or default_context()
    lib = ctx._lib
    m = _capi.Marshalled(P, Sx, Su, None if I is None else [list(g) for g in I], index_base=index_base,
                         flags=_objective_flags(objective))
    T = len(Sx)
    vx = [np.zeros(max(n, 1), dtype=np.float64) for n in m.nnz_x]
    vu = [np.zeros(max(n, 1), dtype=np.float64) for n in m.nnz_u]
    px = (C.POINTER(C.c_double) * T)(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in vx])
    pu = (C.POINTER(C.c_double) * T)(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in vu])
    status = np.zeros(max(m.n_sub, 1), dtype=np.int32)
    stats = _capi.sls_stats()
    rc = lib.sls_h2_sf_solve(ctx.handle, *m.common_args(), px, pu,
                             status.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(stats))
    _capi.check(rc, ctx.handle)
    Phix, Phiu = assemble_phi(Sx, Su, [a[:n] for a, n in zip(vx, m.nnz_x)],
                              [a[:n] for a, n in zip(vu, m.nnz_u)], dropzeros=dropzeros)
    if return_info:
        info = stats.asdict()
        info["col_status"] = status[: m.n_sub].copy()
        info["n_unsolved"] = rc
        return Phix, Phiu, info
    if rc > 0:
        # the reference never checks Ipopt's status (src/synthesis.jl:62-65); a caller that does not ask for the per-column
        # status words still gets a signal, like julia/SLSMI355X.jl's @warn
        st = status[: m.n_sub]
        n_uns = int((st == _capi.SLS_COL_UNSUPPORTED).sum())
        warnings.warn(f"SLS_H2: {rc} of {m.n_sub} subproblems not solved "
                      f"({int((st == _capi.SLS_COL_INFEASIBLE).sum())} infeasible, "
                      f"{int((st == _capi.SLS_COL_NOTCONV).sum())} not converged, {n_uns} unsupported); "
                      "their columns hold the least-squares point (zeros if unsupported) — "
                      "pass return_info=True for the per-column status", RuntimeWarning, stacklevel=2)
    return Phix, Phiu
