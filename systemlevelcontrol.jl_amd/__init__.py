"""MI355X-native engine for the column-separable 𝓗₂ SLS solve of SystemLevelControl.jl.

Public surface mirrors the reference's exports for this path (src/SystemLevelControl.jl:9-23):
    Plant, GeneralizedPlant, StateFeedback, OutputFeedback, SLS_H2 (the reference's SLS_𝓗₂;
    '₂' is not a legal Python identifier character, so the ASCII spelling is the name)
plus the plan/execute split of the C ABI (include/sls_mi355x.h).

The directory is named after the reference (`systemlevelcontrol.jl_amd`), which is not an
importable Python identifier; the repo-root shim `slc_amd.py` loads it as module `slc_amd`.
The compute backend is libsls_mi355x.so (hand-written HIP for gfx950); nothing here falls
back to the CPU.
"""
from .plant import GeneralizedPlant, OutputFeedback, Plant, StateFeedback
from .synthesis import SLS_H2, SLS_H2_batch, SLS_H2_localized, SLS_Hinf_bound, execute_batch, Context, Plan, assemble_phi, default_context
from .closed_loop import ClosedLoop
from ._capi import SLSError, load_library
from . import _capi, closed_loop, dist, workloads

__all__ = ["Plant", "GeneralizedPlant", "StateFeedback", "OutputFeedback", "SLS_H2", "Context", "Plan",
           "assemble_phi", "default_context", "SLS_Hinf_bound", "SLS_H2_batch", "SLS_H2_localized", "execute_batch", "ClosedLoop", "SLSError", "load_library", "dist", "workloads"]
