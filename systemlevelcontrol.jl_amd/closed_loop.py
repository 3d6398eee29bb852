"""Closed-loop simulation with the Φ that lives on the device — mirror of the reference README's script.

Reference README.md:62-72 (user code there, not part of the package):
    β[:,t+1] = Σ_{τ=1..min(t,T−1)} Φx[τ+1]·(x[:,t+1−τ] − β[:,t+1−τ])
    u[:,t]   = Σ_{τ=1..min(t,T)}   Φu[τ]  ·(x[:,t+1−τ] − β[:,t+1−τ])
    x[:,t+1] = A·x[:,t] + B₁·w(t) + B₂·u[:,t]
The recursion runs in libsls_mi355x.so (sls_closed_loop_*: one HIP kernel per time step, replayed from a
hipGraph); this module only marshals.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi


class ClosedLoop:
    """FIR operators of the controller built from the masks (sls_closed_loop_plan) on one device of `ctx`."""

    def __init__(self, ctx, P, S, dev_slot=0):
        Sx, Su = S
        self.ctx, self._lib = ctx, ctx._lib
        self.m = _capi.Marshalled(P, Sx, Su, None)
        self.Nx, self.Nu, self.Nw, self.T = P.Nx, P.Nu, P.Nw, len(Sx)
        h = C.c_void_p()
        _capi.check(self._lib.sls_closed_loop_plan(ctx.handle, dev_slot, C.byref(self.m.dims), C.byref(self.m.plant),
                                                   self.m.Sx, self.m.Su, C.byref(h)), ctx.handle)
        self.handle = h
        n = C.c_int64()
        _capi.check(self._lib.sls_closed_loop_entries(self.handle, C.byref(n)))
        self.n_entries = n.value

    def run(self, d_values, d_w, steps, nscen, d_x, d_u, stream=None):
        """Device pointers in, device pointers out (asynchronous on `stream`)."""
        _capi.check(self._lib.sls_closed_loop_run(self.handle, stream, d_values, d_w, int(steps), int(nscen), d_x, d_u),
                    self.ctx.handle)

    def simulate(self, d_values, w=None, steps=250, nscen=None):
        """d_values: device pointer of the mask-order Φ (Plan.execute(packed=False)).
        w: None or array [steps, Nw] / [steps, Nw, nscen] (host).  Returns x [steps, Nx(, nscen)], u [steps, Nu(, nscen)]:
        row k is the README's x[:,k+1], u[:,k+1] (1-based)."""
        squeeze = False
        if w is not None:
            w = np.asarray(w, dtype=np.float64)
            if w.ndim == 2:
                w = w[:, :, None]; squeeze = nscen is None
            if w.shape[0] != steps or w.shape[1] != self.Nw:
                raise ValueError(f"w must be [steps={steps}, Nw={self.Nw}(, nscen)], got {w.shape}")
            nscen = w.shape[2]
            w = np.ascontiguousarray(w)
        elif nscen is None:
            nscen, squeeze = 1, True
        x = np.zeros((steps, self.Nx, nscen)); u = np.zeros((steps, max(self.Nu, 1), nscen))
        dp = C.POINTER(C.c_double)
        _capi.check(self._lib.sls_closed_loop_run_host(self.handle, d_values, None if w is None else w.ctypes.data_as(dp),
                                                       int(steps), int(nscen), x.ctypes.data_as(dp), u.ctypes.data_as(dp)),
                    self.ctx.handle)
        u = u[:, : self.Nu]
        return (x[:, :, 0], u[:, :, 0]) if squeeze else (x, u)

    def last_ms(self):
        ms = C.c_double()
        _capi.check(self._lib.sls_closed_loop_last_ms(self.handle, C.byref(ms)), self.ctx.handle)
        return ms.value

    def close(self):
        if getattr(self, "handle", None):
            self._lib.sls_closed_loop_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
