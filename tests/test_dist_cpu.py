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
