"""sls_son_oracle.py — CPU oracle of the per-column SUM-OF-NORMS SLS problem (test infrastructure only).

NO REFERENCE EXISTS for this problem: aaltoKEPO/SystemLevelControl.jl has no 𝓗∞ / SOCP synthesis at all (SURVEY §0 F3;
BASELINE.json configs[3] names one).  PARITY UNPINNED by construction; the oracle is pinned by a solver-independent
certificate instead (primal–dual gap + feasibility), see `certificate`.

Problem (one disturbance column c, the same localized index sets, masks and achievability constraints as SLS_𝓗₂ —
reference src/reduction.jl:11-27, src/synthesis.jl:53-60):

    minimise   Σ_t ‖ W z_t ‖₂            z_t = (Φx[t][s_x,c], Φu[t][s_u,c]) on the mask,  W = b·diag([C̃1 D̃12]) ≥ 0
    subject to E z = f                   (Φx[1] = Ĩ, Φx[t+1] = ÃΦx[t] + B̃2Φu[t], 0 = ÃΦx[T] + B̃2Φu[T])

Σ_t‖Φ[t](:,c)‖₂ bounds the column's contribution to the 𝓗∞ norm of the closed loop (‖Σ_t Φ[t]e^{−jωt}(:,c)‖₂ ≤ Σ_t‖Φ[t](:,c)‖₂
for every ω), it is column-separable like the 𝓗₂ cost, and it is the second-order cone program "epigraph of the per-time-step
norms over the same E z = f": min Σ_t τ_t s.t. ‖W z_t‖ ≤ τ_t.  Restricted to diagonal weights and D11 = 0 (with D11 ≠ 0 the
constant rows of fixed variables enter every norm; not built).

Algorithm here: ADMM on  min Σ‖y_t‖ s.t. y = W z, E z = f  with EXACT projections (dense pseudo-inverse of E W⁻¹), run far
past the tolerance the GPU path stops at.  A different implementation of the projection than the kernels' (those use the
block-tridiagonal factor of E H⁻¹ Eᵀ + δI and a multiplier iteration).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

import sls_oracle as o


def _column_problem(P, c, Sx, Su):
    """(E, f, w, tslice, info): free-variable system of column c, diagonal weight per free variable, block ranges per t."""
    E, f, M, m0, info = o.assemble_group(P, [c], Sx, Su)
    if np.abs(m0).max() > 0:
        raise ValueError("sum-of-norms oracle: D11 must vanish on the column's rows")
    nz = info["W"].shape[0]
    T = info["T"]
    nfree = E.shape[1]
    w = np.zeros(nfree)
    tslice = []
    tvar = np.array([t for (t, _, _, _) in info["var_index"]], dtype=np.int64)
    for t in range(T):
        idx = np.flatnonzero(tvar == t)
        tslice.append(idx)
        Mt = M[t * nz:(t + 1) * nz][:, idx]
        G = Mt.T @ Mt
        if np.abs(G - np.diag(np.diag(G))).max() > 1e-14 * max(1.0, np.abs(G).max()):
            raise ValueError("sum-of-norms oracle: [C1 D12]ᵀ[C1 D12] must be diagonal on (s_x, s_u)")
        w[idx] = np.sqrt(np.diag(G))
    if (w <= 0).any():
        raise ValueError("zero cost weight on a free variable")
    return E, f, w, tslice, info


def solve_column(P, c, Sx, Su, iters=20000, tol=1e-11, rho=1.0):
    """Returns (z, diag) with diag = dict(obj, resid, gap, iters, feasible)."""
    E, f, w, tslice, info = _column_problem(P, c, Sx, Su)
    nfree = E.shape[1]
    if nfree == 0:
        return np.zeros(0), dict(obj=0.0, resid=float(np.abs(f).max()), gap=0.0, iters=0, feasible=np.abs(f).max() < 1e-9, info=info)
    # projection onto {E z = f} in the metric W²:  z = v − W⁻²Eᵀ(E W⁻²Eᵀ)⁺(E v − f)
    Ew = E / w                                    # E W⁻¹
    pinv = np.linalg.pinv(Ew, rcond=1e-11)        # (E W⁻¹)⁺
    x_part = pinv @ f                             # particular solution in x = W z coordinates
    resid = float(np.abs(Ew @ x_part - f).max())
    feasible = resid < 1e-9
    # null-space projector in x coordinates:  x ↦ x − pinv·(Ew x)
    def project(xv):
        return xv - pinv @ (Ew @ xv - f)
    x = x_part.copy(); y = x.copy(); u = np.zeros(nfree)
    it = 0
    for it in range(1, iters + 1):
        x = project(y - u)
        v = x + u
        y_old = y
        y = np.zeros(nfree)
        for idx in tslice:
            nv = np.linalg.norm(v[idx])
            if nv > 1.0 / rho:
                y[idx] = (1.0 - 1.0 / (rho * nv)) * v[idx]
        u = u + x - y
        rp = np.linalg.norm(x - y); rd = rho * np.linalg.norm(y - y_old)
        if it % 25 == 0:                          # residual balancing
            if rp > 10 * rd:
                rho *= 2.0; u /= 2.0
            elif rd > 10 * rp:
                rho /= 2.0; u *= 2.0
        if max(rp, rd) < tol * max(1.0, np.linalg.norm(x)):
            break
    z = x / w
    obj = sum(np.linalg.norm(x[idx]) for idx in tslice)
    gap = certificate(E, f, w, tslice, z, rho * u)
    return z, dict(obj=float(obj), resid=float(np.abs(E @ z - f).max()), gap=gap, iters=it, feasible=feasible, info=info)


def certificate(E, f, w, tslice, z, nu):
    """Primal–dual gap of a candidate.  Dual of  min Σ‖W z_t‖ s.t. Ez = f  is  max fᵀμ s.t. ‖W⁻¹(Eᵀμ)_t‖₂ ≤ 1 ∀t.
    `nu` (the scaled ADMM multiplier ρu, ‖nu_t‖ ≤ 1 at convergence) proposes Eᵀμ ≈ W nu; μ is fitted by least squares and
    scaled into the dual feasible set, so the returned gap = primal − dual ≥ 0 is a rigorous optimality bound whatever
    solver produced z."""
    mu = np.linalg.lstsq(E.T, w * nu, rcond=None)[0]
    s = (E.T @ mu) / w
    scale = max(1.0, max((np.linalg.norm(s[idx]) for idx in tslice), default=1.0))
    dual = float(f @ mu) / scale
    primal = sum(np.linalg.norm((w * z)[idx]) for idx in tslice)
    return float(primal - dual)


def SLS_SON(P, S, cols=None, **kw):
    """Φx, Φu (lists of CSC) minimising Σ_t‖W Φ[t](:,c)‖₂ per column; diag list per column."""
    Sx, Su = S
    T = len(Sx)
    cols = list(range(P.Nx)) if cols is None else list(cols)
    Px = [sp.lil_matrix((P.Nx, P.Nx)) for _ in range(T)]
    Pu = [sp.lil_matrix((P.Nu, P.Nx)) for _ in range(T)]
    diags = []
    for c in cols:
        z, dg = solve_column(P, c, Sx, Su, **kw)
        info = dg.pop("info")
        for q, (t, kind, r, _) in enumerate(info["var_index"]):
            if kind == 0:
                Px[t][info["sx"][r], c] = z[q]
            else:
                Pu[t][info["su"][r], c] = z[q]
        dg["n"] = info["n"]; dg["m"] = info["m"]
        diags.append(dg)
    return [M.tocsc() for M in Px], [M.tocsc() for M in Pu], diags
