"""Host-side mirror of the reference's plant type, restricted to what SLS_𝓗₂ reads.

Reference: src/types/GeneralizedPlant.jl
  :45-67   struct GeneralizedPlant{T,Ts} (nine SparseMatrixCSC blocks + Nx,Nz,Ny,Nw,Nu)
  :70-99   9-argument constructor (eltype promotion, StateFeedback detection,
           to_sparse_matrix / fix_feedthrough, SF defaults C2 = I, D21/D22 empty)
  :101-103 6-argument constructor  (C2 = I ⇒ state feedback)
  :105-110 3-argument constructor  ([C1 D12] = I(Nx+Nu), D11 = 0)
  :190     Plant(args...) = GeneralizedPlant(args...)
  :291-311 validate_GeneralizedPlant (dimension errors)
Only the fields are mirrored; the plant algebra (adjoint, view, getindex, ==) is
outside the solve path and stays in Julia.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


class StateFeedback:  # src/types/FeedbackStructures.jl:26
    pass


class OutputFeedback:  # src/types/FeedbackStructures.jl:19
    pass


def _to_sparse_matrix(M, dtype=np.float64):
    """src/types/conversions.jl:11-13: numbers → 1×1, vectors → n×1, matrices as they are."""
    if sp.issparse(M):
        out = sp.csc_matrix(M, dtype=dtype)
    else:
        a = np.asarray(M, dtype=dtype)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)
        out = sp.csc_matrix(a)
    out.sort_indices()
    return out


class GeneralizedPlant:
    """State-feedback generalized plant  P = [A B1 B2; C1 D11 D12; I 0 0].

    `Plant(A, B1, B2)` and `Plant(A, B1, B2, C1, D11, D12)` as in the reference.
    Dimension mismatches raise ValueError with the reference's messages
    (ErrorException there, GeneralizedPlant.jl:296-310)."""

    def __init__(self, A, B1, B2, C1=None, D11=None, D12=None, C2=None, D21=None, D22=None):
        self.A = _to_sparse_matrix(A)
        self.B1 = _to_sparse_matrix(B1)
        self.B2 = _to_sparse_matrix(B2)
        nx, nu = self.A.shape[0], self.B2.shape[1]
        self.default_weights = C1 is None
        if C1 is None:
            if D11 is not None or D12 is not None:
                raise ValueError("C1, D11 and D12 must be given together")
            CD = sp.identity(nx + nu, dtype=np.float64, format="csc")   # GeneralizedPlant.jl:106-108
            C1, D12, D11 = CD[:, :nx], CD[:, nx:], 0
        self.C1 = _to_sparse_matrix(C1)
        self.D12 = _to_sparse_matrix(D12)
        D11m = _to_sparse_matrix(D11)
        if D11m.nnz == 0 or not np.any(D11m.data):                       # fix_feedthrough, conversions.jl:15
            D11m = sp.csc_matrix((self.C1.shape[0], self.B1.shape[1]), dtype=np.float64)
        self.D11 = D11m
        # feedback structure: GeneralizedPlant.jl:76 — C2 == I and D21 empty/zero ⇒ StateFeedback
        is_sf = (C2 is None) and (D21 is None or np.size(D21) == 0) and (D22 is None or np.size(D22) == 0)
        self.Ts = StateFeedback if is_sf else OutputFeedback
        self.C2 = sp.identity(nx, format="csc") if is_sf else _to_sparse_matrix(C2)
        self.D21 = sp.csc_matrix((0, self.B1.shape[1])) if is_sf else _to_sparse_matrix(D21)
        self.D22 = sp.csc_matrix((0, nu)) if is_sf else _to_sparse_matrix(D22)
        self.Nx, self.Nz, self.Ny = nx, self.C1.shape[0], self.C2.shape[0]
        self.Nw, self.Nu = self.B1.shape[1], nu
        self._validate()

    def _validate(self):
        A, B1, B2, C1, D11, D12 = self.A, self.B1, self.B2, self.C1, self.D11, self.D12
        nx = A.shape[0]
        if A.shape[1] != nx or nx == 0:
            raise ValueError(f"A must be nonempty and square, but has dimensions ({A.shape[0]}×{A.shape[1]}).")
        if B1.shape[0] != nx or B2.shape[0] != nx:
            raise ValueError(f"The number of rows of A (={nx}) does not match either B₁ (={B1.shape[0]}) or B₂ (={B2.shape[0]}).")
        if C1.shape[1] != nx:
            raise ValueError(f"The number of columns of A (={nx}) does not match either C₁ (={C1.shape[1]}) or C₂ (={nx}).")
        if D11.shape[0] != C1.shape[0] or D12.shape[0] != C1.shape[0]:
            raise ValueError(f"The number of rows of C₁ (={C1.shape[0]}) does not match either D₁₁ (={D11.shape[0]}) or D₁₂ (={D12.shape[0]}).")
        if D11.shape[1] != B1.shape[1]:
            raise ValueError(f"The number of columns of B₁ (={B1.shape[1]}) does not match either D₁₁ (={D11.shape[1]}) or D₂₁ (={B1.shape[1]}).")
        if D12.shape[1] != B2.shape[1]:
            raise ValueError(f"The number of columns of B₂ (={B2.shape[1]}) does not match either D₁₂ (={D12.shape[1]}) or D₂₂ (={B2.shape[1]}).")

    def __iter__(self):  # src/types/operations.jl:24-33
        return iter((self.A, self.B1, self.B2, self.C1, self.D11, self.D12, self.C2, self.D21, self.D22))

    def __repr__(self):  # GeneralizedPlant.jl:289
        r, c = self.Nx + self.Nz + self.Ny, self.Nx + self.Nu + self.Nw
        return (f"{r}×{c} GeneralizedPlant{{Float64, {self.Ts.__name__}}} w/ {self.Nx} states, "
                f"{self.Ny} outputs, {self.Nu} controls.")


def Plant(*args, **kwargs):  # GeneralizedPlant.jl:190
    return GeneralizedPlant(*args, **kwargs)
