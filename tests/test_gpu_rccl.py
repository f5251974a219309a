"""The N > 1 code path over RCCL on the one GPU this suite has: a ONE-RANK 'nccl' process group with every exchange
forced through torch.distributed (PC_DIST_FORCE=1).  The world-2 semantics of the exchanges are covered on gloo
(tests/test_distributed_gloo.py, tests/test_gpu_sharded.py); what this adds is that every collective call of the path --
its dtypes, shapes, devices, split lists -- is one RCCL accepts, and that bench.py's multi-rank branches (gradient hooks,
barrier + max-over-ranks timing, the sharded-lookup loader) run end to end on that backend.  Needs an MI355X."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ, PC_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PC_DIST_BACKEND", None)
    return env


def _bench(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline", "--no-sustained", *flags],
                         env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exchange", ["native", "hook"])
def test_bench_multi_rank_branches_run_over_rccl(exchange):
    """Product2Vec: the flat-gradient exchange per step, barrier + MAX all-reduce of the timing; joint step at T = 100 and
    T = 34800.  native: the library's exchange slot over its own RCCL communicator (pc_exchange_adam per Product2Vec step,
    pc_joint_train_epoch_dp for the joint epochs); hook: torch.distributed from Python per step (T = 34800: dense segment
    all-reduce + row-list all-gathers)."""
    line = _bench("--steps", "10", "--warmup", "3", "--no-large", "--no-dropout-legs", "--exchange", exchange)
    assert line["rccl"]["backend"].startswith("nccl") and line["rccl"]["ranks_seen"] == [0] and line["rccl"]["world"] == 1
    assert line["n_gpus"] == 1 and line["value"] > 1e6 and line["config"]["parallelism"] == "dp1"
    j, jr = line["joint"], line["joint_num_types_34800"]
    if exchange == "native":
        assert line["rccl"]["exchange"].startswith("rccl"), line["rccl"]
        assert "pc_joint_train_epoch_dp" in j["config"]["launch"] and j["config"]["exchange"].startswith("rccl")
        assert "pc_joint_train_epoch_dp" in jr["config"]["launch"]
        # ABI 8: at T > 512 the optimizer is sharded over the replicas (ncclReduceScatter, Adam on 1/world, ncclAllGather)
        assert jr["config"]["optimizer"].startswith("sharded") and j["config"]["optimizer"].startswith("all-reduce")
    else:
        assert "all-reduce" in j["config"]["launch"] and "hook" in line["rccl"]["exchange"]
    assert j["value"] > 1e6
    assert jr["value"] > 1e6 and jr["config"]["final_loss"] == jr["config"]["final_loss"]        # (not NaN)


@pytest.mark.timeout(900)
def test_bench_sharded_lookup_and_cross_replica_batchnorm_run_over_rccl():
    """--table sharded: the loader's two constant-shape all_to_all rounds per batch (int32 requests, fp32 rows) and the
    capacity agreement (MAX all-reduce of an int64); --sync-bn: the three-phase step with the statistics all-reduce."""
    line = _bench("--phase", "p2v", "--table", "sharded", "--sync-bn", "--steps", "10", "--warmup", "3")
    assert line["config"]["table"] == "sharded" and line["config"]["batchnorm"] == "cross-replica"
    assert line["config"]["sharded_lookup"] and line["value"] > 5e5


def _worker_exchanges():
    """(subprocess body) the general-purpose lookup with split lists and the row-list exchange, against their no-collective forms"""
    import torch.distributed as dist
    from p_companion_amd import distributed as pdist, ops
    rank, world, local = pdist.init_from_env("cuda")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
    dev = torch.device("cuda", local)
    g = torch.Generator().manual_seed(0)
    table = torch.randn(5000, 128, generator=g).to(dev)
    sh = pdist.ShardedFeatureTable(table, 5000, 0, 1)
    ids = torch.randint(-1, 5000, (64, 7), generator=g).to(dev)
    rows, remap = sh.lookup(ids)
    ref = torch.where((ids >= 0)[..., None], table[ids.clamp(min=0)], torch.zeros((), device=dev))
    got = torch.where((remap >= 0)[..., None], rows[remap.clamp(min=0)], torch.zeros((), device=dev))
    assert torch.equal(got.view_as(ref), ref)
    T = 2000
    ex = pdist.TableRowExchange(1)
    tabs = [torch.zeros(T, 64, device=dev), torch.zeros(T, 64, device=dev)]
    touched = [torch.unique(torch.randint(0, T, (300,), generator=g)).to(torch.int32).to(dev) for _ in range(2)]
    want = []
    for t, ids_ in zip(tabs, touched):
        t[ids_.long()] = torch.randn(ids_.numel(), 64, generator=g).to(dev)
        want.append(t.clone())
    ex(tabs, touched)
    assert all(torch.equal(a, b) for a, b in zip(tabs, want)) and ex.last_bytes["row_lists_per_rank"] > 0
    dist.destroy_process_group()
    print("exchanges ok")


def test_lookup_and_row_list_exchange_over_rccl():
    out = subprocess.run([sys.executable, "-c", "import tests.test_gpu_rccl as t; t._worker_exchanges()"], env=_env(),
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0 and "exchanges ok" in out.stdout, out.stderr[-3000:]


def _worker_native_collectives():
    """(subprocess body) the library's own communicator on one rank: pc_rccl_alltoall / pc_rccl_allreduce_sum_f64 / the
    cross-stream chain, and the sharded lookup with BOTH rounds on that communicator."""
    import torch.distributed as dist
    from p_companion_amd import distributed as pdist, ops
    rank, world, local = pdist.init_from_env("cuda")
    dev = torch.device("cuda", local)
    ex = pdist.make_exchange(world, rank=rank, kind="rccl", device=dev)
    assert ex.native and ex.kind.startswith("rccl")
    base = ex.stats()
    assert base["issued"] == 3 and base["chained"] == 2          # the probe: all-reduce | all-to-all on a side stream | all-reduce
    # all-to-all of every dtype the lookup sends (int32 request lists, fp32 rows); one rank: the slice comes back
    send_i = torch.arange(1000, dtype=torch.int32, device=dev)
    recv_i = torch.full_like(send_i, -7)
    ex.all_to_all(send_i, recv_i)
    send_f = torch.randn(300, 128, device=dev)
    recv_f = torch.zeros_like(send_f)
    ex.all_to_all(send_f, recv_f)
    d = torch.arange(ops.BN_SYNC_DOUBLES, dtype=torch.float64, device=dev) * 0.5
    ex.all_reduce_sum_f64_(d)
    torch.cuda.synchronize()
    assert torch.equal(recv_i, send_i) and torch.equal(recv_f, send_f)
    assert torch.equal(d, torch.arange(ops.BN_SYNC_DOUBLES, dtype=torch.float64, device=dev) * 0.5)
    s1 = ex.stats()
    # (the probe ran on two streams of its own: the first collective here changes stream once; the next two follow it on the
    # same stream, where stream order is the chain)
    assert s1["issued"] == base["issued"] + 3 and s1["chained"] == base["chained"] + 1
    import pytest as _pt
    with _pt.raises(ValueError):
        ex.all_to_all(send_i, send_i)                            # aliased
    # two streams, alternating: every change of stream is chained (a wait on an event at the other stream's tail)
    side = torch.cuda.Stream(dev)
    g = torch.ones(4096, device=dev)
    for i in range(4):
        with torch.cuda.stream(side):
            ex.all_to_all(send_f, recv_f)
        ex.all_reduce_mean_(g)
    torch.cuda.synchronize()
    s2 = ex.stats()
    assert s2["issued"] == s1["issued"] + 8 and s2["chained"] == s1["chained"] + 8
    assert torch.equal(g, torch.ones_like(g))
    # the sharded lookup with its two rounds on the library's communicator == the same lookup on torch's
    gen = torch.Generator().manual_seed(0)
    table = torch.randn(5000, 128, generator=gen).to(dev)
    B, K = 64, 5
    nb = torch.randint(-1, 5000, (B, 9), generator=gen, dtype=torch.int32).to(dev)
    batch = {"anchor_idx": torch.randint(0, 5000, (B,), generator=gen, dtype=torch.int32).to(dev),
             "positive_idx": torch.randint(0, 5000, (B,), generator=gen, dtype=torch.int32).to(dev),
             "negative_idx": torch.randint(0, 5000, (B, K), generator=gen, dtype=torch.int32).to(dev),
             "neighbor_compact": ops.unique_neighbors(nb)}
    # (the owner-side slot of an id is handed out by integer atomics: the request lists of two calls may differ in order, the
    # rows every index of the remapped batch points at may not)
    zrow = torch.zeros(1, 128, device=dev)
    n_ids = 2 * B + B * K + batch["neighbor_compact"]["nb_rows"].numel()
    with _pt.raises(ValueError):                                 # a native exchange: no lazy capacity agreement inside a lookup
        pdist.ShardedFeatureTable(table, 5000, 0, 1, exchange=ex).lookup_batch(batch)
    with _pt.raises(ValueError):                                 # ... and no variable-split lookup on torch's communicator
        pdist.ShardedFeatureTable(table, 5000, 0, 1, exchange=ex, capacity=n_ids).lookup(batch["anchor_idx"])
    for exch in (ex, None):
        sh = pdist.ShardedFeatureTable(table, 5000, 0, 1, exchange=exch, capacity=pdist.ShardedFeatureTable.capacity_for(n_ids, 1))
        tab, rb = sh.lookup_batch(batch)
        assert sh.overflowed() == 0
        for k in ("anchor_idx", "positive_idx", "negative_idx"):
            assert torch.equal(tab[rb[k].long().reshape(-1)], table[batch[k].long().reshape(-1)]), (exch is None, k)
        nu = int(batch["neighbor_compact"]["n_unique"])
        want_rows = batch["neighbor_compact"]["nb_rows"][:nu + 1].long()
        got_rows = rb["neighbor_compact"]["nb_rows"][:nu + 1].long()
        ext, ext_t = torch.cat([tab, zrow]), torch.cat([table, zrow])
        assert torch.equal(ext[got_rows], ext_t[want_rows]), exch is None             # (-1 = the padding row: a zero row)
    assert ex.stats()["issued"] == s2["issued"] + 2              # both rounds went through the library's communicator
    # ABI 8: the sharded optimizer's two halves (one rank: both are the identity) and pc_exchange_adam_plan against the plain form
    t = torch.randn(4096, device=dev)
    ref = t.clone()
    ex.reduce_scatter_mean_(t)
    ex.all_gather_(t)
    torch.cuda.synchronize()
    assert torch.equal(t, ref) and ex.stats()["issued"] == s2["issued"] + 4
    n = 10000
    gen2 = torch.Generator().manual_seed(3)
    p0, g0 = torch.randn(n, generator=gen2).to(dev), torch.randn(n, generator=gen2).to(dev)
    outs = []
    for shard in (False, True):
        p_, g_ = p0.clone(), g0.clone()
        m_, v_ = torch.zeros_like(p_), torch.zeros_like(p_)
        cnt, sc = torch.zeros(1, dtype=torch.int64, device=dev), torch.zeros(2, device=dev)
        for step in (1, 2, 3):
            ops.exchange_adam(ex, p_, g_, m_, v_, cnt, step, sc, 1e-2, shard=shard)
        ops.exchange_adam(ex, p_, g_, m_, v_, cnt, 0, sc, 1e-2, shard=shard)       # the device-counter form
        torch.cuda.synchronize()
        outs.append((p_, m_, v_, int(cnt)))
    assert all(torch.equal(a, b) for a, b in zip(outs[0][:3], outs[1][:3])) and outs[0][3] == outs[1][3] == 4
    ex.close()
    dist.destroy_process_group()
    print("native collectives ok")


def test_library_alltoall_and_the_cross_stream_chain_over_rccl():
    out = subprocess.run([sys.executable, "-c", "import tests.test_gpu_rccl as t; t._worker_native_collectives()"], env=_env(),
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0 and "native collectives ok" in out.stdout, out.stderr[-3000:]
