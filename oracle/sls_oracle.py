"""CPU oracle for the column-separable H2 SLS solve  --  TEST INFRASTRUCTURE ONLY.

This file is a NumPy/SciPy FP64 *restatement* of the reference algorithm
(aaltoKEPO/SystemLevelControl.jl, `SLS_𝓗₂`), written from the reference's
source text.  It is the checker for the HIP path; nothing in the product
package (`systemlevelcontrol.jl_amd/`) may import it.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` use it.

PARITY PINNING.  The reference holds exactly one fixture that touches this
path: test/reduction_test.jl:11-24 (index sets of `sparsity_dim_reduction`).
`tests/test_oracle.py` pins `sparsity_dim_reduction` below against it.  The
reference has NO test or golden vector for the values of Φ (test/runtests.jl:10-13
never calls SLS_𝓗₂) and Julia/JuMP/Ipopt are not installed in the build
container, so for Φ itself **parity is unpinned**: what stands in is the
solver-independent optimality certificate `certificate()` (feasibility +
projected-gradient = 0 of a strictly convex equality-constrained QP, whose
optimum is unique, so any solver that converges — Ipopt included — returns it).

The solve here is deliberately a *different algorithm* from the GPU one
(dense SVD least squares on the explicitly assembled constraint matrix vs. the
block-tridiagonal Schur-complement recursion of the kernels), so agreement
between the two is evidence, not tautology.

Third-party arithmetic the reference delegates to (absent from /root/reference):
JuMP 1.10.0 -> MathOptInterface 1.14.1 -> Ipopt 3.14.10 (Ipopt_jll 300.1400.1000)
-> MUMPS_seq 5.5.1 -> OpenBLAS32 0.3.17   (Manifest.toml:177,183,206,253,269,309).
Call sites: src/synthesis.jl:46 (Model(Ipopt.Optimizer)), :47-60 (model build),
:62 (optimize!), :65-66 (value.).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

__all__ = [
    "OraclePlant", "readme_chain", "readme_masks", "sparsity_dim_reduction",
    "assemble_group", "solve_group", "SLS_H2", "certificate", "closed_loop",
    "values_in_mask_order",
]


# --------------------------------------------------------------------------
# Plant container (the nine blocks the path reads; GeneralizedPlant.jl:45-67)
# --------------------------------------------------------------------------
class OraclePlant:
    """State-feedback generalized plant.

    3-argument form synthesises the LQR-shaped weights exactly as
    src/types/GeneralizedPlant.jl:105-110:  [C1 D12] = I(Nx+Nu), D11 = 0.
    """

    def __init__(self, A, B1, B2, C1=None, D11=None, D12=None):
        self.A = sp.csc_matrix(A, dtype=np.float64)
        self.B1 = sp.csc_matrix(B1, dtype=np.float64)
        self.B2 = sp.csc_matrix(B2, dtype=np.float64)
        self.Nx = self.A.shape[0]
        self.Nw = self.B1.shape[1]
        self.Nu = self.B2.shape[1]
        if C1 is None:
            CD = sp.identity(self.Nx + self.Nu, dtype=np.float64, format="csc")
            C1 = CD[:, : self.Nx]
            D12 = CD[:, self.Nx:]
        self.C1 = sp.csc_matrix(C1, dtype=np.float64)
        self.D12 = sp.csc_matrix(D12, dtype=np.float64)
        self.Nz = self.C1.shape[0]
        if D11 is None:
            D11 = sp.csc_matrix((self.Nz, self.Nw), dtype=np.float64)
        self.D11 = sp.csc_matrix(D11, dtype=np.float64)


def readme_chain(Nx=59, Nu=None):
    """README.md:43-47.  A = I + 0.2*superdiag - 0.2*subdiag; B1 = I;
    B2 = I[:, {6n+1, 6n+2}] (1-based) truncated to Nu columns (README: 20)."""
    A = (sp.identity(Nx) + sp.diags(0.2 * np.ones(Nx - 1), 1)
         - sp.diags(0.2 * np.ones(Nx - 1), -1)).tocsc()
    cols = [6 * n + k for n in range((Nx + 5) // 6) for k in (0, 1) if 6 * n + k < Nx]
    if Nu is None:
        Nu = 20 if Nx == 59 else len(cols)
    cols = cols[:Nu]
    B2 = sp.identity(Nx, format="csc")[:, cols]
    return OraclePlant(A, sp.identity(Nx, format="csc"), B2)


def _bool_power(Mb, k):
    R = sp.identity(Mb.shape[0], dtype=np.int64, format="csc")
    for _ in range(int(k)):
        R = ((R @ Mb) != 0).astype(np.int64).tocsc()
    return R


def readme_masks(A, B2, d, T, alpha):
    """README.md:53-54 (1-based t = 1..T):
       Sx[t] = (A≠0)^min(d,  floor(α(t-1))) ≠ 0
       Su[t] = (B2'≠0)(A≠0)^min(d+1,floor(α(t-1))) ≠ 0."""
    Ab = (sp.csc_matrix(A) != 0).astype(np.int64).tocsc()
    Bb = (sp.csc_matrix(B2).T != 0).astype(np.int64).tocsc()
    Sx, Su = [], []
    cache = {}
    for t in range(T):
        kx = min(d, int(np.floor(alpha * t)))
        ku = min(d + 1, int(np.floor(alpha * t)))
        for k in (kx, ku):
            if k not in cache:
                cache[k] = _bool_power(Ab, k)
        sx = (cache[kx] != 0).tocsc()
        su = ((Bb @ cache[ku]) != 0).tocsc()
        sx.sort_indices(); su.sort_indices()
        Sx.append(sx); Su.append(su)
    return Sx, Su


# --------------------------------------------------------------------------
# src/reduction.jl:11-27  (state-feedback branch)
# --------------------------------------------------------------------------
def _unique_first_appearance(v):
    _, idx = np.unique(v, return_index=True)
    return v[np.sort(idx)]


def sparsity_dim_reduction(P, cj, S):
    """Returns (sub-plant blocks dict, I_tilde, ii_x, s_x, s_u); indices 0-based.

    reduction.jl:14: s = unique(findnz((S[end]*(A.≠0))[:,cj])[1]) — rows of the
    structural nonzeros of the Bool product restricted to the group's columns,
    in first-appearance (column-major) order.
    reduction.jl:15 + GeneralizedPlant.jl:266-285: view(P,(sx,[sx;Nx.+su]),(sx,cj,su)).
    reduction.jl:22-23: ii_x = sx ∈ cj ; Ĩ = [I(ñx)[:,ii_x]  0].
    """
    cj = np.asarray(cj, dtype=np.int64)
    Sx, Su = S
    Ab = (P.A != 0).astype(np.int64).tocsc()
    out = []
    for Sj in (Sx, Su):
        last = sp.csc_matrix(Sj[-1])
        patt = sp.csc_matrix((np.ones(last.nnz, dtype=np.int64), last.indices, last.indptr),
                             shape=last.shape)  # structural pattern (findnz semantics)
        prod = (patt @ Ab).tocsc()
        prod.sort_indices()
        rows = np.concatenate([prod.indices[prod.indptr[c]:prod.indptr[c + 1]] for c in cj]) \
            if len(cj) else np.zeros(0, dtype=np.int64)
        out.append(_unique_first_appearance(rows.astype(np.int64)))
    sx, su = out
    zrows = np.concatenate([sx, P.Nx + su])
    sub = dict(
        A=P.A[sx][:, sx].toarray(), B1=P.B1[sx][:, cj].toarray(), B2=P.B2[sx][:, su].toarray(),
        C1=P.C1[zrows][:, sx].toarray(), D11=P.D11[zrows][:, cj].toarray(),
        D12=P.D12[zrows][:, su].toarray(),
    )
    iix = np.isin(sx, cj)
    nx, nw = len(sx), len(cj)
    It = np.zeros((nx, nw))
    ksel = np.flatnonzero(iix)
    It[ksel, np.arange(len(ksel))] = 1.0
    return sub, It, iix, sx, su


# --------------------------------------------------------------------------
# src/synthesis.jl:40-60 : the per-group QP, assembled explicitly
# --------------------------------------------------------------------------
def _mask_block(Sj_t, rows, cols):
    """(S[t][rows, cols] == 1) as a dense bool array (synthesis.jl:58-59 tests `.≠ 1`)."""
    M = sp.csc_matrix(Sj_t)
    return np.asarray(M[rows][:, cols].todense()) == 1


def assemble_group(P, cj, Sx, Su):
    """Build (E, f, M, m0, index bookkeeping) for one group of columns.

    Variables: Φ̃x[t] (ñx×ñw), Φ̃u[t] (ñu×ñw), t = 0..T-1, only the entries whose
    mask is 1 (the others are `fix`ed to 0 and leave the problem: synthesis.jl:57-60).
    Constraints (synthesis.jl:53-55), every row of s_x, every column of the group:
        Φ̃x[0] = Ĩ ;  Φ̃x[t+1] = ÃΦ̃x[t] + B̃2Φ̃u[t] ;  0 = ÃΦ̃x[T-1] + B̃2Φ̃u[T-1].
    Cost (synthesis.jl:50,52,76-83):  Σ_t ‖[C̃1 D̃12][Φ̃x;Φ̃u][t]·B̃1 + D̃11‖_F².
    """
    cj = np.asarray(cj, dtype=np.int64)
    T = len(Sx)
    sub, It, iix, sx, su = sparsity_dim_reduction(P, cj, [Sx, Su])
    n, m, w = len(sx), len(su), len(cj)
    B1t = sub["B1"][iix, :]                       # synthesis.jl:42
    if B1t.shape[0] != w:
        raise ValueError("group columns not all inside s_x: L*Φ*R is not conformable "
                         "(the reference would throw a DimensionMismatch)")
    W = np.hstack([sub["C1"], sub["D12"]])         # (ñx+ñu) x (ñx+ñu) for LQR-shaped weights
    mx = [_mask_block(Sx[t], sx, cj) for t in range(T)]   # n x w
    mu = [_mask_block(Su[t], su, cj) for t in range(T)]   # m x w
    # free-variable enumeration: t-major, then x before u, then column-major (vec order)
    var_index = []        # (t, kind, local_row, local_col)
    pos = {}
    for t in range(T):
        for kind, msk in ((0, mx[t]), (1, mu[t])):
            for c in range(w):
                for r in np.flatnonzero(msk[:, c]):
                    pos[(t, kind, r, c)] = len(var_index)
                    var_index.append((t, kind, r, c))
    nfree = len(var_index)
    nrows = (T + 1) * n * w
    E = np.zeros((nrows, nfree))
    f = np.zeros(nrows)

    def row(k, i, c):
        return (k * w + c) * n + i

    for c in range(w):
        for i in range(n):
            f[row(0, i, c)] = It[i, c]
    At, Bt = sub["A"], sub["B2"]
    for (t, kind, r, c), q in pos.items():
        if kind == 0:
            E[row(t, r, c), q] += 1.0                       # +Φ̃x[t] in block-row t
            E[row(t + 1, 0, c):row(t + 1, 0, c) + n, q] -= At[:, r]   # -ÃΦ̃x[t] in block-row t+1
        else:
            E[row(t + 1, 0, c):row(t + 1, 0, c) + n, q] -= Bt[:, r]
    # cost  Σ_t ‖ W Φ_t B̃1 + D̃11 ‖_F² = ‖ M z + m0 ‖²,   vec(WΦB) = (Bᵀ⊗W) vec(Φ)
    nz = W.shape[0]
    M = np.zeros((T * nz * w, nfree))
    m0 = np.tile(sub["D11"].reshape(-1, order="F"), T)
    for (t, kind, r, c), q in pos.items():
        wcol = W[:, r] if kind == 0 else W[:, n + r]
        # Φ[r_full, c] multiplies B̃1[c, :] on the right
        for c2 in range(w):
            if B1t[c, c2] != 0.0:
                M[(t * w + c2) * nz:(t * w + c2 + 1) * nz, q] += wcol * B1t[c, c2]
    info = dict(sx=sx, su=su, n=n, m=m, w=w, T=T, mx=mx, mu=mu, var_index=var_index,
                iix=iix, It=It, At=At, Bt=Bt, B1t=B1t, W=W)
    return E, f, M, m0, info


def _solve_dense(E, f, M, m0, rcond):
    nfree = E.shape[1]
    if nfree == 0:
        return np.zeros(0), 0, np.inf
    U, s, Vt = np.linalg.svd(E, full_matrices=True)
    tol = rcond * (s[0] if len(s) else 1.0)
    r = int((s > tol).sum())
    smin = float(s[r - 1]) if r else np.inf               # smallest singular value kept: |ΔΦ| of an iterative solve ≈ residual / smin
    zp = Vt[:r].T @ ((U[:, :r].T @ f) / s[:r])           # min-norm (least-squares) particular solution
    N = Vt[r:].T                                          # null(E)
    if N.shape[1]:
        wv = np.linalg.lstsq(M @ N, -(M @ zp + m0), rcond=None)[0]
        return zp + N @ wv, r, smin
    return zp, r, smin


def solve_group(P, cj, Sx, Su, rcond=1e-11, decouple=True):
    """Solve one group's QP by dense SVD (null-space method).  Returns (z, info, diag).

    When B̃1 is diagonal the group's QP is separable by column (block-diagonal E and M in
    the column index), so it is solved column by column — the same optimum at a fraction of
    the SVD cost; `decouple=False` forces the monolithic solve (used by a test to show the
    two agree)."""
    E, f, M, m0, info = assemble_group(P, cj, Sx, Su)
    B1t = info["B1t"]
    if decouple and info["w"] > 1 and np.count_nonzero(B1t - np.diag(np.diag(B1t))) == 0:
        n, w, T = info["n"], info["w"], info["T"]
        nz = info["W"].shape[0]
        z = np.zeros(E.shape[1]); rank = 0; smin = np.inf
        vcol = np.array([c for (_, _, _, c) in info["var_index"]], dtype=np.int64)
        for c in range(w):
            vsel = np.flatnonzero(vcol == c)
            rsel = np.concatenate([np.arange((k * w + c) * n, (k * w + c + 1) * n) for k in range(T + 1)])
            msel = np.concatenate([np.arange((t * w + c) * nz, (t * w + c + 1) * nz) for t in range(T)])
            zc, rc, sm = _solve_dense(E[np.ix_(rsel, vsel)], f[rsel], M[np.ix_(msel, vsel)], m0[msel], rcond)
            z[vsel] = zc; rank += rc; smin = min(smin, sm)
        resid = float(np.abs(E @ z - f).max()) if E.shape[0] else 0.0
        return z, info, dict(resid=resid, rank=rank, smin=smin, cost=float(np.sum((M @ z + m0) ** 2)), E=E, f=f, M=M, m0=m0)
    nfree = E.shape[1]
    if nfree == 0:
        return np.zeros(0), info, dict(resid=float(np.abs(f).max(initial=0.0)), rank=0, smin=np.inf, cost=float(m0 @ m0))
    U, s, Vt = np.linalg.svd(E, full_matrices=True)
    tol = rcond * (s[0] if len(s) else 1.0)
    r = int((s > tol).sum())
    zp = Vt[:r].T @ ((U[:, :r].T @ f) / s[:r])           # min-norm particular solution
    N = Vt[r:].T                                          # null(E)
    if N.shape[1]:
        wv = np.linalg.lstsq(M @ N, -(M @ zp + m0), rcond=None)[0]
        z = zp + N @ wv
    else:
        z = zp
    resid = float(np.abs(E @ z - f).max()) if E.shape[0] else 0.0
    cost = float(np.sum((M @ z + m0) ** 2))
    return z, info, dict(resid=resid, rank=r, smin=(float(s[r - 1]) if r else np.inf), cost=cost, E=E, f=f, M=M, m0=m0)


def certificate(E, f, M, m0, z):
    """Solver-independent optimality certificate of  min ‖Mz+m0‖² s.t. Ez=f :
       feasibility ‖Ez−f‖∞  and  projected gradient ‖P_null(E) Mᵀ(Mz+m0)‖∞."""
    feas = float(np.abs(E @ z - f).max()) if E.shape[0] else 0.0
    g = M.T @ (M @ z + m0)
    # projection onto null(E):  g - Eᵀ (E Eᵀ)⁺ E g
    y = np.linalg.lstsq(E.T, g, rcond=None)[0]
    pg = g - E.T @ y
    return feas, float(np.abs(pg).max(initial=0.0))


# --------------------------------------------------------------------------
# src/synthesis.jl:11-32,65-67 : driver + masked scatter
# --------------------------------------------------------------------------
def SLS_H2(P, S, I=None, return_diag=False):
    """Returns (Φx, Φu): lists (length T) of CSC matrices Nx×Nx / Nu×Nx.

    synthesis.jl:15: default one group per column.  synthesis.jl:65-67: the solved
    values are multiplied by the mask and scattered to (s_x, c_j) / (s_u, c_j);
    sparse `+` drops numerical zeros, so the final pattern ⊆ mask."""
    Sx, Su = S
    T = len(Sx)
    groups = [[i] for i in range(P.Nx)] if I is None else [list(g) for g in I]
    Px = [sp.lil_matrix((P.Nx, P.Nx)) for _ in range(T)]
    Pu = [sp.lil_matrix((P.Nu, P.Nx)) for _ in range(T)]
    diags = []
    for cj in groups:
        z, info, dg = solve_group(P, cj, Sx, Su)
        diags.append(dict(cols=list(cj), n=info["n"], m=info["m"], nfree=len(z),
                          resid=dg["resid"], rank=dg["rank"], smin=dg["smin"], cost=dg["cost"]))
        for q, (t, kind, r, c) in enumerate(info["var_index"]):
            if kind == 0:
                Px[t][info["sx"][r], cj[c]] += z[q]
            else:
                Pu[t][info["su"][r], cj[c]] += z[q]
    Phix = [m.tocsc() for m in Px]
    Phiu = [m.tocsc() for m in Pu]
    if return_diag:
        return Phix, Phiu, diags
    return Phix, Phiu


def values_in_mask_order(Phi, Smask):
    """Flatten Φ[t] to the value array aligned with the mask's CSC nzval order
    (the layout the C ABI hands back: SURVEY §8b 'Ownership')."""
    out = []
    for F, Sm in zip(Phi, Smask):
        Sm = sp.csc_matrix(Sm); Sm.sort_indices()
        F = sp.csc_matrix(F)
        rows = Sm.indices
        cols = np.repeat(np.arange(Sm.shape[1]), np.diff(Sm.indptr))
        out.append(np.asarray(F[rows, cols]).ravel().astype(np.float64))
    return out


def closed_loop(A, B1, B2, Phix, Phiu, steps=250, t_imp=50, i_imp=29, w=None):
    """README.md:62-72 (0-based storage; impulse w(t)=δ(t-50)e_30, 1-based; or an explicit disturbance
    sequence w[steps, Nw] whose row t−1 is the README's w(t)).
       β[:,t+1] = Σ_{τ=1..min(t,T-1)} Φx[τ+1](x[:,t+1-τ] − β[:,t+1-τ])
       u[:,t]   = Σ_{τ=1..min(t,T)}   Φu[τ]  (x[:,t+1-τ] − β[:,t+1-τ])
       x[:,t+1] = A x[:,t] + B1 w(t) + B2 u[:,t]."""
    A = sp.csc_matrix(A); B1 = sp.csc_matrix(B1); B2 = sp.csc_matrix(B2)
    Nx, Nu = A.shape[0], B2.shape[1]
    T = len(Phix)
    x = np.zeros((Nx, steps + 1)); beta = np.zeros_like(x); u = np.zeros((Nu, steps + 1))
    for t in range(1, steps):          # 1-based t as in the README
        b = np.zeros(Nx)
        for tau in range(1, min(t, T - 1) + 1):
            b += Phix[tau] @ (x[:, t + 1 - tau] - beta[:, t + 1 - tau])   # Φx[τ+1] 1-based = index τ
        beta[:, t + 1] = b
        uu = np.zeros(Nu)
        for tau in range(1, min(t, T) + 1):
            uu += Phiu[tau - 1] @ (x[:, t + 1 - tau] - beta[:, t + 1 - tau])
        u[:, t] = uu
        if w is None:
            wt = np.zeros(B1.shape[1])
            if t == t_imp:
                wt[i_imp] = 1.0
        else:
            wt = np.asarray(w[t - 1], dtype=np.float64)
        x[:, t + 1] = A @ x[:, t] + B1 @ wt + B2 @ uu
    return x[:, 1:], u[:, 1:]
