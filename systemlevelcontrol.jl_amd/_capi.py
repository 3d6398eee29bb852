"""ctypes binding of include/sls_mi355x.h (libsls_mi355x.so).

This is the same marshalling the Julia `ccall` wrapper performs:
SciPy's CSC triplet (indptr, indices, data) is Julia's (colptr, rowval, nzval) with
index_base = 0 instead of 1.

The library is the ONLY compute backend of this package.  If it cannot be loaded the
import fails loudly; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsls_mi355x.so")

# ---- constants (mirror the header) ----
SLS_ABI_VERSION = 2
SLS_EINVAL, SLS_ENOTSF, SLS_EUNSUPPORTED, SLS_EHIP, SLS_ENOMEM, SLS_ENODEVICE = -1, -2, -3, -4, -5, -6
SLS_SOLVE_DEFAULT, SLS_SOLVE_SUM_OF_NORMS = 0, 1
SLS_COL_OK, SLS_COL_INFEASIBLE, SLS_COL_NOTCONV, SLS_COL_TRIVIAL, SLS_COL_SKIPPED, SLS_COL_UNSUPPORTED = 0, 1, 2, 3, 4, 5

ERROR_NAMES = {SLS_EINVAL: "SLS_EINVAL", SLS_ENOTSF: "SLS_ENOTSF", SLS_EUNSUPPORTED: "SLS_EUNSUPPORTED",
               SLS_EHIP: "SLS_EHIP", SLS_ENOMEM: "SLS_ENOMEM", SLS_ENODEVICE: "SLS_ENODEVICE"}


class SLSError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {msg}")
        self.code = code


class sls_csc_f64(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("colptr", C.POINTER(C.c_int64)),
                ("rowval", C.POINTER(C.c_int64)), ("nzval", C.POINTER(C.c_double))]


class sls_csc_bool(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("colptr", C.POINTER(C.c_int64)),
                ("rowval", C.POINTER(C.c_int64)), ("nzval", C.POINTER(C.c_uint8))]


class sls_dims(C.Structure):
    _fields_ = [("Nx", C.c_int64), ("Nu", C.c_int64), ("Nz", C.c_int64), ("Nw", C.c_int64), ("T", C.c_int64),
                ("index_base", C.c_int32), ("flags", C.c_uint32)]


class sls_plant(C.Structure):
    _fields_ = [(k, C.POINTER(sls_csc_f64)) for k in ("A", "B1", "B2", "C1", "D11", "D12")]


class sls_stats(C.Structure):
    _fields_ = [("n_subproblems", C.c_int64), ("n_not_ok", C.c_int64), ("n_values_x", C.c_int64),
                ("n_values_u", C.c_int64), ("n_free", C.c_int64), ("max_nx", C.c_int32), ("max_nu", C.c_int32),
                ("max_iters", C.c_int32), ("n_devices", C.c_int32), ("max_residual", C.c_double),
                ("flops_alg", C.c_double), ("bytes_alg", C.c_double), ("t_symbolic_s", C.c_double),
                ("t_upload_s", C.c_double), ("t_solve_s", C.c_double), ("t_download_s", C.c_double), ("n_refined", C.c_int64)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class sls_plan_info(C.Structure):
    _fields_ = [("n_subproblems", C.c_int64), ("n_values", C.c_int64), ("n_values_x", C.c_int64),
                ("n_values_u", C.c_int64), ("n_packed", C.c_int64), ("workspace_bytes", C.c_int64),
                ("max_nx", C.c_int32), ("max_nu", C.c_int32), ("T", C.c_int32), ("device", C.c_int32),
                ("flops_alg", C.c_double), ("bytes_alg", C.c_double), ("t_symbolic_s", C.c_double),
                ("t_upload_s", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


EXPORTS = [
    "sls_create", "sls_destroy", "sls_last_error", "sls_abi_version", "sls_device_count",
    "sls_h2_sf_solve", "sls_h2_sf_solve_batch", "sls_set_ridge", "sls_h2_sf_plan", "sls_plan_get_info", "sls_plan_value_offsets",
    "sls_plan_execute", "sls_plan_execute_batch", "sls_plan_refine", "sls_plan_synchronize", "sls_plan_packed_dest", "sls_plan_fetch_status",
    "sls_plan_kernel_time_ms", "sls_plan_alloc_values", "sls_plan_free_values", "sls_plan_download",
    "sls_plan_destroy", "sls_scatter_f64", "sls_shard_groups", "sls_sparsity_dim_reduction",
    "sls_h2_sf_packed_layout", "sls_plan_describe", "sls_localization_masks", "sls_localization_masks_device",
    "sls_index_sets_device", "sls_h2_sf_plan_localized", "sls_h2_sf_solve_localized",
    "sls_closed_loop_plan", "sls_closed_loop_run", "sls_closed_loop_run_host", "sls_closed_loop_last_ms",
    "sls_closed_loop_entries", "sls_closed_loop_destroy",
]

_lib = None


def load_library(path: str | None = None):
    """dlopen libsls_mi355x.so and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C systemlevelcontrol.jl_amd/csrc).  There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own libamdhip64 under the system runtime's SONAME; whichever copy the process loads
    # first serves both, and torch only finds its GPUs through its own.  A process that also uses torch tensors/streams
    # (tests, bench.py, dist.py) must therefore load torch's copy before this library pulls in /opt/rocm's.
    if "torch" not in sys.modules and not os.environ.get("SLS_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(p)
    vp, i64p, i32p, dp = C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
    dpp = C.POINTER(C.POINTER(C.c_double))
    lib.sls_create.restype = vp; lib.sls_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_uint32]
    lib.sls_destroy.restype = None; lib.sls_destroy.argtypes = [vp]
    lib.sls_last_error.restype = C.c_char_p; lib.sls_last_error.argtypes = [vp]
    lib.sls_abi_version.restype = C.c_int; lib.sls_abi_version.argtypes = []
    lib.sls_device_count.restype = C.c_int; lib.sls_device_count.argtypes = []
    common = [C.POINTER(sls_dims), C.POINTER(sls_plant), C.POINTER(sls_csc_bool), C.POINTER(sls_csc_bool),
              C.c_int64, i64p, i64p]
    lib.sls_h2_sf_solve.restype = C.c_int
    lib.sls_h2_sf_solve.argtypes = [vp] + common + [dpp, dpp, i32p, C.POINTER(sls_stats)]
    lib.sls_set_ridge.restype = C.c_int; lib.sls_set_ridge.argtypes = [vp, C.c_int64, dp, C.c_int64, dp]
    lib.sls_h2_sf_solve_batch.restype = C.c_int
    lib.sls_h2_sf_solve_batch.argtypes = [vp, C.c_int, C.POINTER(sls_dims), C.POINTER(sls_plant), C.POINTER(C.POINTER(sls_csc_bool)),
                                          C.POINTER(C.POINTER(sls_csc_bool)), C.POINTER(dpp), C.POINTER(dpp), C.POINTER(i32p),
                                          C.POINTER(sls_stats)]
    lib.sls_h2_sf_plan.restype = C.c_int
    lib.sls_h2_sf_plan.argtypes = [vp, C.c_int] + common + [C.c_int64, C.c_int64, C.POINTER(vp)]
    lib.sls_plan_get_info.restype = C.c_int; lib.sls_plan_get_info.argtypes = [vp, C.POINTER(sls_plan_info)]
    lib.sls_plan_value_offsets.restype = C.c_int; lib.sls_plan_value_offsets.argtypes = [vp, i64p, i64p]
    lib.sls_plan_execute.restype = C.c_int; lib.sls_plan_execute.argtypes = [vp, vp, vp, C.c_int]
    lib.sls_plan_synchronize.restype = C.c_int; lib.sls_plan_synchronize.argtypes = [vp, vp]
    lib.sls_plan_refine.restype = C.c_int; lib.sls_plan_refine.argtypes = [vp] + common + [vp, vp, C.c_int, i64p]
    lib.sls_plan_packed_dest.restype = C.c_int; lib.sls_plan_packed_dest.argtypes = [vp, i64p]
    lib.sls_plan_fetch_status.restype = C.c_int; lib.sls_plan_fetch_status.argtypes = [vp, i32p, dp, i32p]
    lib.sls_plan_kernel_time_ms.restype = C.c_int; lib.sls_plan_kernel_time_ms.argtypes = [vp, dp, i64p]
    lib.sls_plan_describe.restype = C.c_int; lib.sls_plan_describe.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.sls_plan_alloc_values.restype = C.c_int; lib.sls_plan_alloc_values.argtypes = [vp, C.c_int, C.POINTER(vp)]
    lib.sls_plan_free_values.restype = C.c_int; lib.sls_plan_free_values.argtypes = [vp, vp]
    lib.sls_plan_download.restype = C.c_int; lib.sls_plan_download.argtypes = [vp, vp, dpp, dpp]
    lib.sls_plan_destroy.restype = None; lib.sls_plan_destroy.argtypes = [vp]
    lib.sls_scatter_f64.restype = C.c_int; lib.sls_scatter_f64.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int64, vp]
    lib.sls_shard_groups.restype = C.c_int; lib.sls_shard_groups.argtypes = common + [C.c_int, i64p]
    lib.sls_h2_sf_packed_layout.restype = C.c_int
    lib.sls_h2_sf_packed_layout.argtypes = common + [C.c_int64, C.c_int64, i64p, i64p, i64p, C.POINTER(sls_plan_info)]
    i64pp = C.POINTER(C.POINTER(C.c_int64))
    lib.sls_localization_masks.restype = C.c_int
    lib.sls_localization_masks.argtypes = [C.POINTER(sls_dims), C.POINTER(sls_csc_f64), C.POINTER(sls_csc_f64), C.c_int64,
                                           C.c_double, i64p, i64p, i64pp, i64pp, i64pp, i64pp]
    lib.sls_localization_masks_device.restype = C.c_int
    lib.sls_localization_masks_device.argtypes = [vp, C.c_int] + lib.sls_localization_masks.argtypes
    lib.sls_sparsity_dim_reduction.restype = C.c_int
    lib.sls_sparsity_dim_reduction.argtypes = [C.POINTER(sls_dims), C.POINTER(sls_csc_f64), C.POINTER(sls_csc_bool),
                                               C.POINTER(sls_csc_bool), i64p, C.c_int64, i64p, i64p, i64p, i64p]
    lib.sls_index_sets_device.restype = C.c_int
    lib.sls_index_sets_device.argtypes = [vp, C.c_int, C.POINTER(sls_dims), C.POINTER(sls_csc_f64), C.POINTER(sls_csc_bool),
                                          C.POINTER(sls_csc_bool), i64p, i64p, i64p, i64p]
    lib.sls_h2_sf_plan_localized.restype = C.c_int
    lib.sls_h2_sf_plan_localized.argtypes = [vp, C.c_int, C.POINTER(sls_dims), C.POINTER(sls_plant), C.c_int64, C.c_double, C.POINTER(vp)]
    lib.sls_h2_sf_solve_localized.restype = C.c_int
    lib.sls_h2_sf_solve_localized.argtypes = [vp, C.POINTER(sls_dims), C.POINTER(sls_plant), C.c_int64, C.c_double, dpp, dpp, i32p,
                                              C.POINTER(sls_stats)]
    lib.sls_closed_loop_plan.restype = C.c_int
    lib.sls_closed_loop_plan.argtypes = [vp, C.c_int, C.POINTER(sls_dims), C.POINTER(sls_plant), C.POINTER(sls_csc_bool),
                                         C.POINTER(sls_csc_bool), C.POINTER(vp)]
    lib.sls_closed_loop_run.restype = C.c_int
    lib.sls_closed_loop_run.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int64, vp, vp]
    lib.sls_closed_loop_run_host.restype = C.c_int
    lib.sls_closed_loop_run_host.argtypes = [vp, vp, dp, C.c_int64, C.c_int64, dp, dp]
    lib.sls_closed_loop_last_ms.restype = C.c_int; lib.sls_closed_loop_last_ms.argtypes = [vp, dp]
    lib.sls_closed_loop_entries.restype = C.c_int; lib.sls_closed_loop_entries.argtypes = [vp, i64p]
    lib.sls_closed_loop_destroy.restype = None; lib.sls_closed_loop_destroy.argtypes = [vp]
    lib.sls_plan_execute_batch.restype = C.c_int
    lib.sls_plan_execute_batch.argtypes = [C.POINTER(vp), C.c_int, vp, C.POINTER(vp), C.c_int]
    # diagnostics outside the public header
    lib.sls_debug_tile_invert.restype = C.c_int; lib.sls_debug_tile_invert.argtypes = [vp, C.c_int, C.c_int, dp, dp, C.c_int]
    lib.sls_plan_debug_read_workspace.restype = C.c_int; lib.sls_plan_debug_read_workspace.argtypes = [vp, C.c_int64, C.c_int64, dp]
    lib.sls_debug_plan_tables.restype = C.c_int
    lib.sls_debug_plan_tables.argtypes = [vp, C.c_int, C.POINTER(sls_dims), C.POINTER(sls_plant), C.POINTER(sls_csc_bool),
                                          C.POINTER(sls_csc_bool), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int,
                                          C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.sls_debug_plan_tables_localized.restype = C.c_int
    lib.sls_debug_plan_tables_localized.argtypes = [vp, C.c_int, C.POINTER(sls_dims), C.POINTER(sls_plant), C.c_int64, C.c_double,
                                                    C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                                                    C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    if lib.sls_abi_version() != SLS_ABI_VERSION:
        raise ImportError(f"ABI mismatch: library {lib.sls_abi_version()} vs binding {SLS_ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def last_error(ctx=None) -> str:
    s = load_library().sls_last_error(ctx)
    return s.decode("utf-8", "replace") if s else ""


def check(rc: int, ctx=None):
    if rc < 0:
        raise SLSError(rc, last_error(ctx) or last_error(None))
    return rc


# ---------------------------------------------------------------------------
# marshalling helpers: keep the numpy arrays alive next to the ctypes structs
# ---------------------------------------------------------------------------
class Marshalled:
    """Holds ctypes views of a problem (plant, masks, groups) plus the arrays backing them."""

    def __init__(self, P, Sx, Su, groups=None, index_base=0, flags=0):
        """index_base = 1 hands every colptr/rowval/group column over exactly as Julia stores them (used by the tests to
        exercise the path the `ccall` binding takes); groups are always given 0-based on the Python side."""
        self.keep = []
        self.base = int(index_base)
        T = len(Sx)
        if len(Su) != T:
            raise ValueError("𝓢x and 𝓢u must have the same length T")
        self.dims = sls_dims(P.Nx, P.Nu, P.Nz, P.Nw, T, self.base, int(flags))
        self.plant = sls_plant()
        for name, attr in (("A", "A"), ("B1", "B1"), ("B2", "B2"), ("C1", "C1"), ("D11", "D11"), ("D12", "D12")):
            M = getattr(P, attr)
            setattr(self.plant, name, C.pointer(self._f64(M)) if M is not None else None)
        self.Sx = (sls_csc_bool * T)(*[self._bool(m) for m in Sx])
        self.Su = (sls_csc_bool * T)(*[self._bool(m) for m in Su])
        if groups is None:
            self.ngroups = 0
            self.group_ptr = None
            self.group_cols = None
            self.n_sub = P.Nx
        else:
            ptr = np.zeros(len(groups) + 1, dtype=np.int64)
            ptr[1:] = np.cumsum([len(g) for g in groups])
            cols = np.asarray([c for g in groups for c in g], dtype=np.int64) + self.base
            self.keep += [ptr, cols]
            self.ngroups = len(groups)
            self.group_ptr = ptr.ctypes.data_as(C.POINTER(C.c_int64))
            self.group_cols = cols.ctypes.data_as(C.POINTER(C.c_int64)) if len(cols) else None
            self.n_sub = int(ptr[-1])
        self.nnz_x = [int(sp.csc_matrix(m).nnz) for m in Sx]
        self.nnz_u = [int(sp.csc_matrix(m).nnz) for m in Su]

    def _csc_arrays(self, M):
        M = sp.csc_matrix(M)
        if not M.has_sorted_indices:
            M = M.copy(); M.sort_indices()
        colptr = np.ascontiguousarray(M.indptr, dtype=np.int64) + self.base
        rowval = np.ascontiguousarray(M.indices, dtype=np.int64) + self.base
        return M, colptr, rowval

    def _f64(self, M):
        M, colptr, rowval = self._csc_arrays(M)
        nz = np.ascontiguousarray(M.data, dtype=np.float64)
        self.keep += [colptr, rowval, nz]
        s = sls_csc_f64(M.shape[0], M.shape[1], colptr.ctypes.data_as(C.POINTER(C.c_int64)),
                        rowval.ctypes.data_as(C.POINTER(C.c_int64)), nz.ctypes.data_as(C.POINTER(C.c_double)))
        self.keep.append(s)
        return s

    def _bool(self, M):
        M, colptr, rowval = self._csc_arrays(M)
        nz = np.ascontiguousarray(M.data != 0, dtype=np.uint8)
        self.keep += [colptr, rowval, nz]
        return sls_csc_bool(M.shape[0], M.shape[1], colptr.ctypes.data_as(C.POINTER(C.c_int64)),
                            rowval.ctypes.data_as(C.POINTER(C.c_int64)), nz.ctypes.data_as(C.POINTER(C.c_uint8)))

    def common_args(self):
        return (C.byref(self.dims), C.byref(self.plant), self.Sx, self.Su, self.ngroups, self.group_ptr,
                self.group_cols)
