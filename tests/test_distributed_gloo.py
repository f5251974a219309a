"""N>1 path on the CPU: world_size-2 gloo.  Covers the sharded-table lookup exchange (bucketing,
de-duplication, the two all_to_all rounds, remapping) and the dense-gradient all-reduce.
The owner-side row gather is injected (test infrastructure); on the GPU it is the HIP kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _cpu_gather(table, idx):
    return table[idx.long()]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from p_companion_amd import distributed as pdist
    r, w, _ = pdist.init_from_env("cpu")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    full = torch.randn(1000, 16, generator=g)
    tab = pdist.ShardedFeatureTable(pdist.ShardedFeatureTable.shard(full, rank, world), 1000, rank, world,
                                    gather_fn=_cpu_gather)
    gi = torch.Generator().manual_seed(100 + rank)
    ids = torch.randint(-1, 1000, (37, 11), generator=gi, dtype=torch.int32)
    ids[0, :5] = ids[1, :5]                                   # duplicates across the batch
    rows, remap = tab.lookup(ids)
    ok = remap.shape == ids.shape and remap.dtype == torch.int32
    ext = torch.cat([rows, torch.zeros(1, 16)])
    want = torch.cat([full, torch.zeros(1, 16)])[ids.long()]
    ok = ok and torch.equal(ext[remap.long()], want)
    ok = ok and rows.shape[0] == len(torch.unique(ids[ids >= 0]))            # de-duplicated on the wire
    ok = ok and bool(((remap < 0) == (ids < 0)).all())
    grad = torch.full((10,), float(rank + 1))
    pdist.all_reduce_mean_(grad, world)
    ok = ok and torch.allclose(grad, torch.full((10,), (1 + world) / 2))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sharded_lookup_and_grad_allreduce_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
    assert res == {0: True, 1: True}


def test_sharded_lookup_single_rank():
    from p_companion_amd import distributed as pdist
    full = torch.arange(50 * 4, dtype=torch.float32).view(50, 4)
    tab = pdist.ShardedFeatureTable(full, 50, 0, 1, gather_fn=_cpu_gather)
    ids = torch.tensor([[3, -1, 3], [49, 0, -1]], dtype=torch.int32)
    rows, remap = tab.lookup(ids)
    ext = torch.cat([rows, torch.zeros(1, 4)])
    assert torch.equal(ext[remap.long()], torch.cat([full, torch.zeros(1, 4)])[ids.long()])
