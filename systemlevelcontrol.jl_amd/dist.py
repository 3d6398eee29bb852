"""Column sharding across ranks + reassembly of {Φx[t], Φu[t]} with ONE all-gather.

Reference analogue (src/synthesis.jl:16,24-27): the groups 𝓘 are cut into contiguous
chunks, one per worker, each worker solves its columns, and `@distributed (+)` folds
the per-worker sparse matrices on the master.  Here: one process per GPU
(torch.distributed; backend "nccl" is RCCL over xGMI), contiguous *cost-balanced* cuts
(sls_shard_groups), every rank solves its shard into a packed value vector, and a
single all_gather_into_tensor moves the packed shards; a scatter kernel then places
them in the mask-order value array.  The columns are independent, so there is no other
data-path collective.

torch is plumbing here (device buffers, streams, the process group); the solve itself
is libsls_mi355x.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi


def shard_groups(P, S, groups, nshards):
    """Cost-balanced contiguous cuts (host only).  Returns int64[nshards+1]."""
    lib = _capi.load_library()
    m = _capi.Marshalled(P, S[0], S[1], groups)
    cuts = np.zeros(nshards + 1, dtype=np.int64)
    _capi.check(lib.sls_shard_groups(*m.common_args(), nshards, cuts.ctypes.data_as(C.POINTER(C.c_int64))))
    return cuts


def shard_cost_report(P, S, groups, nshards):
    """What the cost-balanced cut of `shard_groups` looks like: (cuts, per-shard predicted cost, max/mean imbalance).
    The cost model is the library's own (sls_shard_groups: |c_j|·((T+1)·ñx³ + 1) per group, ñx from src/reduction.jl:14),
    restated here from the masks so that a rank can report the imbalance of its cut without a device."""
    import scipy.sparse as sp
    cuts = shard_groups(P, S, groups, nshards)
    T = len(S[0])
    last = sp.csc_matrix(S[0][-1])
    patt = sp.csc_matrix((np.ones(last.nnz, dtype=np.int32), last.indices, last.indptr), shape=last.shape)   # findnz: structural
    prod = (patt @ (sp.csc_matrix(P.A) != 0).astype(np.int32)).tocsc()
    if groups is None:
        n = np.diff(prod.indptr).astype(np.float64)
        cost = (T + 1) * n ** 3 + 1.0
    else:
        cost = np.zeros(len(groups))
        for g, cols in enumerate(groups):
            rows = np.unique(np.concatenate([prod.indices[prod.indptr[c]:prod.indptr[c + 1]] for c in cols])) if len(cols) else np.zeros(0)
            cost[g] = len(cols) * ((T + 1) * float(len(rows)) ** 3 + 1.0)
    per = np.array([cost[int(cuts[r]):int(cuts[r + 1])].sum() for r in range(nshards)])
    return cuts, per, float(per.max() / max(per.mean(), 1e-300))


def packed_layout(P, S, groups, group_range, index_base=0):
    """Host-only symbolic pass of one shard: (dest int64[n_packed], n_values, info dict)."""
    lib = _capi.load_library()
    m = _capi.Marshalled(P, S[0], S[1], groups, index_base=index_base)
    npk, nval = C.c_int64(), C.c_int64()
    info = _capi.sls_plan_info()
    gb, ge = group_range
    _capi.check(lib.sls_h2_sf_packed_layout(*m.common_args(), gb, ge, C.byref(npk), C.byref(nval), None, C.byref(info)))
    dest = np.zeros(max(npk.value, 1), dtype=np.int64)
    _capi.check(lib.sls_h2_sf_packed_layout(*m.common_args(), gb, ge, None, None,
                                            dest.ctypes.data_as(C.POINTER(C.c_int64)), None))
    return dest[: npk.value], nval.value, info.asdict()


class HipLocalSolver:
    """The product's local solver: a device plan over this rank's shard."""

    def __init__(self, ctx, P, S, groups, group_range, objective="h2"):
        from .synthesis import Plan
        self.plan = Plan(ctx, P, S, groups, group_range, objective=objective)
        self.n_packed = self.plan.info["n_packed"]
        self.info = self.plan.info

    def dest(self):
        return self.plan.packed_dest()

    def solve_into(self, packed_tensor):
        import torch
        stream = torch.cuda.current_stream(packed_tensor.device).cuda_stream
        self.plan.execute(packed_tensor.data_ptr(), packed=True, stream=stream)

    def refine(self, tensor, packed=True):
        """sls_plan_refine on this shard (after a solve into `tensor`): near-singular columns of the one-wave / twisted kernels are
        re-solved on the tile kernel, in place, and that pass stays attached to the plan for every later solve."""
        import torch
        stream = torch.cuda.current_stream(tensor.device).cuda_stream
        return self.plan.refine(tensor.data_ptr(), packed=packed, stream=stream)


class ColumnShardedH2:
    """N-rank solve of SLS_𝓗₂: shard → local solve → one all-gather → unpack.

    `local_solver_factory(group_range)` must return an object with `.n_packed`,
    `.dest()` (int64 destinations in the mask-order value array) and
    `.solve_into(tensor)`; the default builds a HipLocalSolver on `device`.
    Tests on CPU ranks (gloo) pass a factory that fills the packed vector from
    precomputed values, which exercises everything here except the HIP kernels."""

    def __init__(self, P, S, groups=None, *, device=None, process_group=None, local_solver_factory=None, ctx=None,
                 always_gather=False, objective="h2"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.device = torch.device(device if device is not None else "cpu")
        # always_gather: run the collective + unpack even at world size 1 (rehearses the N>1 data path on one GPU)
        self.gather = self.world > 1 or (always_gather and dist.is_initialized())
        self._pipe = None
        ng = len(groups) if groups is not None else P.Nx
        self.cuts = shard_groups(P, S, groups, self.world)
        rng = (int(self.cuts[self.rank]), int(self.cuts[self.rank + 1]))
        self.group_range = rng
        if local_solver_factory is None:
            if self.device.type != "cuda":
                raise RuntimeError("the HIP local solver needs a cuda (ROCm) device; there is no CPU fallback")
            from .synthesis import Context
            self.ctx = ctx or Context([self.device.index or 0])
            self.local = HipLocalSolver(self.ctx, P, S, groups, rng, objective=objective)
        else:
            self.ctx = ctx
            self.local = local_solver_factory(rng)
        self.n_groups = ng
        # --- static layout exchange (setup, not part of a step) ---
        n_local = int(self.local.n_packed)
        dest_local = np.asarray(self.local.dest(), dtype=np.int64)
        _, self.n_values, _ = packed_layout(P, S, groups, (0, 0))
        counts = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        mine = torch.tensor([n_local], dtype=torch.int64, device=self.device)
        if self.gather:
            self._all_gather(counts, mine)
        else:
            counts.copy_(mine)
        self.counts = counts.cpu().numpy()
        self.max_packed = max(int(self.counts.max()), 1)
        # padding entries are routed to a dump slot one past the end of the value array
        dpad = torch.full((self.max_packed,), self.n_values, dtype=torch.int64, device=self.device)
        if n_local:
            dpad[:n_local] = torch.from_numpy(dest_local).to(self.device)
        self.unpack_idx = torch.empty(self.world * self.max_packed, dtype=torch.int64, device=self.device)
        if self.gather:
            self._all_gather(self.unpack_idx, dpad)
        else:
            self.unpack_idx.copy_(dpad)
        # --- step buffers ---
        self.packed = torch.zeros(self.max_packed, dtype=torch.float64, device=self.device)
        self.gathered = torch.zeros(self.world * self.max_packed, dtype=torch.float64, device=self.device)
        self.values = torch.zeros(self.n_values + 1, dtype=torch.float64, device=self.device)

    def _all_gather(self, out, inp):
        """all_gather_into_tensor on the process group.  With the nccl backend (= RCCL) device tensors go straight
        over xGMI.  A gloo group cannot move device memory, so there (tests only: several ranks sharing one GPU)
        the vectors are staged through the host."""
        dist = self.dist
        if inp.is_cuda and dist.get_backend(self.pg) == "gloo":
            self.torch.cuda.current_stream(inp.device).synchronize()
            ho = self.torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(ho, inp.cpu(), group=self.pg)
            out.copy_(ho)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.pg)

    def _unpack(self, src, stream_ptr):
        """values[unpack_idx[k]] = src[k] on the given stream (device) / index_copy_ (CPU ranks)."""
        if self.device.type == "cuda" and self.ctx is not None:
            _capi.check(self.ctx._lib.sls_scatter_f64(self.ctx.handle, 0, stream_ptr, src.data_ptr(),
                                                      self.unpack_idx.data_ptr(), src.numel(), self.values.data_ptr()),
                        self.ctx.handle)
        else:
            self.values.index_copy_(0, self.unpack_idx[: src.numel()], src)

    def _direct(self):
        """One rank, HIP solver, no collective requested: the solve writes the mask-order array itself (no unpack launch).
        Measured equal to packed output + unpack kernel on chain-4096 (3.12 ms either way on the same box) and 3 µs
        faster per pass on the README chain."""
        return (not self.gather) and self.device.type == "cuda" and self.ctx is not None and hasattr(self.local, "plan")

    def step(self):
        """One pass of the hot path over this rank's shard + reassembly on every rank, ordered on the current stream."""
        torch = self.torch
        if self._direct():
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self.local.plan.execute(self.values.data_ptr(), packed=False, stream=stream)
            return self.values[: self.n_values]
        self.local.solve_into(self.packed)
        if self.gather:
            self._all_gather(self.gathered, self.packed)                             # RCCL over xGMI
            src = self.gathered
        else:
            src = self.packed
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None
        self._unpack(src, stream)
        return self.values[: self.n_values]

    def step_async(self):
        """Throughput form of `step` for a stream of independent solves: the all-gather + unpack of pass k run on a side
        stream while the current stream already solves pass k+1 (two packed/gathered buffer pairs; events keep a buffer
        from being refilled before its gather has consumed it).  Every pass does the full work of `step`; call `flush()`
        before reading `values` or stopping a timer.  On CPU ranks and on the gloo staging path this is `step`."""
        torch = self.torch
        if self._direct() or self.device.type != "cuda" or self.ctx is None or \
                (self.gather and self.dist.get_backend(self.pg) == "gloo"):
            return self.step()
        if self._pipe is None:
            self._pipe = {"side": torch.cuda.Stream(device=self.device), "k": 0,
                          "solved": [torch.cuda.Event(), torch.cuda.Event()], "free": [torch.cuda.Event(), torch.cuda.Event()],
                          "packed": [self.packed, torch.zeros_like(self.packed)],
                          "gathered": [self.gathered, torch.zeros_like(self.gathered)]}
        pp = self._pipe
        b = pp["k"] & 1
        main, side = torch.cuda.current_stream(self.device), pp["side"]
        main.wait_event(pp["free"][b])                 # no-op until the buffer pair has been through a gather once
        self.local.solve_into(pp["packed"][b])
        pp["solved"][b].record(main)
        with torch.cuda.stream(side):
            side.wait_event(pp["solved"][b])
            if self.gather:
                self._all_gather(pp["gathered"][b], pp["packed"][b])
                src = pp["gathered"][b]
            else:
                src = pp["packed"][b]
            self._unpack(src, side.cuda_stream)
            pp["free"][b].record(side)
        pp["k"] += 1
        return self.values[: self.n_values]

    def refine(self):
        """Once, after the first `step`: every rank lets its plan re-solve the columns that converged slowly (near-singular
        constraint matrix) on the tile kernel and keeps that pass attached, then the step is repeated so that
        every rank holds the refined Φ.  Returns the number of subproblems this rank refines (0 on ranks with none; CPU test
        ranks without a plan return 0)."""
        if not hasattr(self.local, "refine"):
            return 0
        if self._direct():
            n = self.local.refine(self.values, packed=False)
        else:
            n = self.local.refine(self.packed, packed=True)
        self.step()
        return int(n)

    def flush(self):
        """Make the current stream wait for everything `step_async` has in flight."""
        if self._pipe is not None:
            main = self.torch.cuda.current_stream(self.device)
            for ev in self._pipe["free"]:
                main.wait_event(ev)

    def subproblems_owned(self):
        return int(getattr(self.local, "info", {}).get("n_subproblems", 0))
