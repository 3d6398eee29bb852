"""N>1 path on CPU ranks (gloo, world_size 2): cost-balanced column sharding, the single all-gather of
packed shards and the unpack — everything of dist.ColumnShardedH2 except the HIP kernels, whose place is
taken by a stand-in local solver that fills the packed vector from the golden Φ."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import slc_amd
        g = np.load(os.path.join(GOLDEN, "readme_chain_phi.npz"))
        full = np.concatenate([g["vals_x"], g["vals_u"]])
        P, S, _ = slc_amd.workloads.make_workload("readme_chain")

        class GoldenLocal:
            def __init__(self, rng):
                self._dest, _, self.info = slc_amd.dist.packed_layout(P, S, None, rng)
                self.n_packed = len(self._dest)

            def dest(self):
                return self._dest

            def solve_into(self, t):
                t[: self.n_packed] = torch.from_numpy(full[self._dest])

        sh = slc_amd.dist.ColumnShardedH2(P, S, None, device="cpu", local_solver_factory=GoldenLocal)
        vals = sh.step().numpy().copy()
        ok = bool(np.array_equal(vals, full))
        q.put((rank, ok, sh.group_range, int(sh.local.n_packed), sh.subproblems_owned()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shard_allgather_unpack():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    # contiguous cover of the 59 groups, both ranks own work, packed counts add up to Σfree
    assert res[0][2][0] == 0 and res[0][2][1] == res[1][2][0] and res[1][2][1] == 59
    assert res[0][3] + res[1][3] == 36029
    assert res[0][4] + res[1][4] == 59


def _worker_big(rank, world, port, q, workload):
    """Full-size cut of a BASELINE plant over `world` CPU ranks: every rank runs the host symbolic pass of its own shard, the
    packed counts are exchanged by the same all-gather the GPU path uses, and a stand-in solver writes each value's own
    mask-order index — after one step every entry of the value array must hold its index (each written exactly once)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import slc_amd
        P, S, _ = slc_amd.workloads.make_workload(workload)

        class IndexLocal:
            def __init__(self, rng):
                self._dest, _, self.info = slc_amd.dist.packed_layout(P, S, None, rng)
                self.n_packed = len(self._dest)

            def dest(self):
                return self._dest

            def solve_into(self, t):
                t[: self.n_packed] = torch.from_numpy(self._dest.astype(np.float64))

        sh = slc_amd.dist.ColumnShardedH2(P, S, None, device="cpu", local_solver_factory=IndexLocal)
        vals = sh.step().numpy()
        ok = bool(np.array_equal(vals, np.arange(sh.n_values, dtype=np.float64)))
        cuts, per, imb = slc_amd.dist.shard_cost_report(P, S, None, world)
        q.put((rank, ok, tuple(int(c) for c in sh.cuts), int(sh.local.n_packed), int(sh.n_values), [int(c) for c in sh.counts],
               tuple(int(c) for c in cuts), float(imb)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,workload", [(2, "chain4096"), (8, "chain4096"), (2, "random10000_d2")])
def test_full_size_cuts_are_balanced_and_cover_every_value(world, workload):
    """The strong-scaling case of the N > 1 path (bench.py's "strong" record) on CPU ranks: chain-4096 over 2 and 8 ranks and
    the random-sparse plant over 2 — cuts identical on every rank and contiguous, predicted-cost imbalance max/mean ≤ 1.05,
    packed counts equal to what each rank's symbolic pass reports and adding up to the whole value array, and the all-gather +
    unpack placing every value exactly once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_big, args=(r, world, port, q, workload)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), [r[:2] for r in res]
    cuts = res[0][2]
    Nx = 4096 if workload == "chain4096" else 10000
    assert cuts[0] == 0 and cuts[-1] == Nx and all(a < b for a, b in zip(cuts, cuts[1:]))
    for r in res:
        assert r[2] == cuts and r[6] == cuts and r[5] == [x[3] for x in res]
        assert r[7] <= 1.05, r[7]
    assert sum(r[3] for r in res) == res[0][4]


def test_random10000_cut_over_eight_ranks_is_balanced():
    """The eight-way cut of the random-sparse plants (ñx from 1 to 322: costs differ by 10⁷ between columns) without spawning
    ranks: imbalance ≤ 1.05, and the shards' packed counts (host symbolic pass) add up to the value array."""
    sys.path.insert(0, ROOT)
    import slc_amd
    for name in ("random10000_d2", "random10000_d2_act1"):
        P, S, _ = slc_amd.workloads.make_workload(name)
        cuts, per, imb = slc_amd.dist.shard_cost_report(P, S, None, 8)
        assert imb <= 1.05, (name, imb, per)
        tot = 0
        for r in range(8):
            _, nval, info = slc_amd.dist.packed_layout(P, S, None, (int(cuts[r]), int(cuts[r + 1])))
            tot += info["n_packed"]
        assert tot == nval == sum(m.nnz for m in S[0] + S[1])
