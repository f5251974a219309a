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


def _cpu_bucket(arrays, G, C, counts, send_ids, overflow, hot_rows=0, hot_ids=None, hot_served=None):
    """Test-side restatement of pc_shard_bucket[_hot]'s contract (include/pcompanion_hip.h) for CPU tensors."""
    counts.zero_(); send_ids.fill_(-1)
    hot = {int(v): k for k, v in enumerate(hot_ids.tolist())} if hot_ids is not None else None
    outs = []
    for ids, n_dev, add in arrays:
        live = ids.numel() if n_dev is None else min(ids.numel(), int(n_dev) + add)
        out = torch.full_like(ids, -1)
        for pos in range(live):
            i = int(ids[pos])
            if i < 0:
                continue
            hs = (hot.get(i, -1) if hot is not None else (i if i < hot_rows else -1)) if hot_rows else -1
            if hs >= 0:                                          # the replicated hot set: no request slot
                out[pos] = G * C + hs
                if hot_served is not None:
                    hot_served += 1
                continue
            o = i % G
            slot = int(counts[o]); counts[o] += 1
            if slot < C:
                send_ids[o * C + slot] = i // G
                out[pos] = o * C + slot
            else:
                overflow += 1
        outs.append(out)
    return outs


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
    # the per-step, fixed-capacity form (what the loader runs): request lists of C ids per peer, rows back, the batch's
    # indices over the [G*C, D] buffer; the unique-neighbour list's live length is a device scalar
    gb = torch.Generator().manual_seed(200 + rank)
    B, K = 16, 5
    nb = torch.sort(torch.randperm(1000, generator=gb)[:40]).values.to(torch.int32)
    batch = {"anchor_idx": torch.randint(0, 1000, (B,), generator=gb, dtype=torch.int32),
             "positive_idx": torch.randint(0, 1000, (B,), generator=gb, dtype=torch.int32),
             "negative_idx": torch.randint(0, 1000, (B, K), generator=gb, dtype=torch.int32),
             "neighbor_compact": {"nb_rows": torch.cat([nb, torch.tensor([-1, 777, 778], dtype=torch.int32)]),   # 2 scratch entries
                                  "weight": torch.ones(43), "slot_row": torch.zeros(B, 4, dtype=torch.int32),
                                  "n_unique": 40, "n_unique_dev": torch.tensor([40], dtype=torch.int32)}}
    tab2 = pdist.ShardedFeatureTable(pdist.ShardedFeatureTable.shard(full, rank, world), 1000, rank, world,
                                     gather_fn=lambda t, i: torch.cat([t, torch.zeros(1, 16)])[i.long()], bucket_fn=_cpu_bucket)
    rows2, rb = tab2.lookup_batch(batch)
    ok = ok and rows2.shape == (world * tab2.capacity, 16) and tab2.overflowed() == 0
    ext2 = torch.cat([rows2, torch.zeros(1, 16)])
    wantf = torch.cat([full, torch.zeros(1, 16)])
    for k in ("anchor_idx", "positive_idx", "negative_idx"):
        ok = ok and torch.equal(ext2[rb[k].long()], wantf[batch[k].long()])
    nbr = rb["neighbor_compact"]["nb_rows"]
    ok = ok and torch.equal(ext2[nbr[:41].long()], wantf[batch["neighbor_compact"]["nb_rows"][:41].long()])
    ok = ok and bool((nbr[41:] == -1).all()) and int(nbr[40]) == -1
    ok = ok and tab2.bytes_per_peer == {"request_ids": 4 * tab2.capacity, "rows": 64 * tab2.capacity}
    # the replicated hot set (configs[4]): ids of the set are served from the local replica behind the exchange buffer -- the
    # rows every index resolves to are the same bits, the request lists are shorter by exactly the entries served
    zg = lambda t, i: torch.cat([t, torch.zeros(1, 16)])[i.long()]
    for hot_ids in (None, torch.tensor(sorted([5, 999, 17, 400, 401, 402, 3, 250]), dtype=torch.int32)):
        H = 64 if hot_ids is None else hot_ids.numel()
        hb = {k: (v.clone() if torch.is_tensor(v) else dict(v)) for k, v in batch.items()}
        hot_list = list(range(H)) if hot_ids is None else hot_ids.tolist()
        hb["negative_idx"][:, 0] = torch.tensor([hot_list[i % H] for i in range(B)], dtype=torch.int32)     # a Zipf-like head
        hb["anchor_idx"][:3] = torch.tensor(hot_list[:3], dtype=torch.int32)
        plain = pdist.ShardedFeatureTable(pdist.ShardedFeatureTable.shard(full, rank, world), 1000, rank, world, gather_fn=zg,
                                          bucket_fn=_cpu_bucket, capacity=tab2.capacity)
        hot = pdist.ShardedFeatureTable(pdist.ShardedFeatureTable.shard(full, rank, world), 1000, rank, world, gather_fn=zg,
                                        bucket_fn=_cpu_bucket, capacity=tab2.capacity, hot_rows=H, hot_ids=hot_ids)
        rep = hot.build_hot_replica()                            # a collective: both ranks
        ok = ok and torch.equal(rep, full[torch.tensor(hot_list).long()])
        rows_p, rb_p = plain.lookup_batch(hb)
        used_p = int(plain._bufs["counts"].sum())
        rows_h, rb_h = hot.lookup_batch(hb)
        used_h = int(hot._bufs["counts"].sum())
        ok = ok and rows_h.shape == (world * hot.capacity + H, 16) and rows_p.shape == (world * plain.capacity, 16)
        ext_p, ext_h = torch.cat([rows_p, torch.zeros(1, 16)]), torch.cat([rows_h, torch.zeros(1, 16)])
        for k in ("anchor_idx", "positive_idx", "negative_idx"):
            ok = ok and torch.equal(ext_h[rb_h[k].long()], ext_p[rb_p[k].long()])
            ok = ok and torch.equal(ext_h[rb_h[k].long()], wantf[hb[k].long()])
        ok = ok and torch.equal(ext_h[rb_h["neighbor_compact"]["nb_rows"][:41].long()], wantf[hb["neighbor_compact"]["nb_rows"][:41].long()])
        served = hot.hot_rows_served()
        in_set = sum(int(x) in set(hot_list) for k in ("anchor_idx", "positive_idx", "negative_idx") for x in hb[k].reshape(-1).tolist())
        in_set += sum(int(x) in set(hot_list) for x in hb["neighbor_compact"]["nb_rows"][:41].tolist())
        ok = ok and served == in_set >= B + 3 and used_p - used_h == served and hot.hot_rows_served() == 0
        ok = ok and hot.overflowed() == 0 and bool((rb_h["negative_idx"][:, 0] >= world * hot.capacity).all())
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


# ------------------------------------------------------------------ row-list exchange of the type-table gradients
def _joint_half_batch(rank, B, P, T_live, seed=0):
    g = torch.Generator().manual_seed(seed)
    full = {"query_idx": torch.randint(0, P, (2 * B,), generator=g, dtype=torch.int32),
            "query_types": torch.randint(0, T_live, (2 * B,), generator=g),
            "positive_types": torch.randint(0, T_live, (2 * B, 1), generator=g),
            "negative_types": torch.randint(0, T_live, (2 * B, 1), generator=g),
            "positive_items": torch.randn(2 * B, 128, generator=g), "negative_items": torch.randn(2 * B, 128, generator=g)}
    half = {k: v[rank * B:(rank + 1) * B] for k, v in full.items()}
    return full, half


def _exchange_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import joint_oracle
    from p_companion_amd import distributed as pdist
    pdist.init_from_env("cpu")
    T, P, B = 3000, 200, 48
    gt = torch.Generator().manual_seed(1)
    st = joint_oracle.init_state(3, torch.randn(P, 128, generator=gt), T)
    full, half = _joint_half_batch(rank, B, P, 25, seed=5)
    loc = joint_oracle.train_step({k: v.clone() for k, v in st.items()}, half, joint_oracle.new_moments(st), 1)
    ref = joint_oracle.train_step({k: v.clone() for k, v in st.items()}, full, joint_oracle.new_moments(st), 1)
    names = ("complementary_type_embeddings.weight", "query_type_embeddings.weight")
    tables = [loc["grads"][n].clone() for n in names]
    touched = [torch.nonzero(t.abs().amax(1) > 0).reshape(-1).to(torch.int32) for t in tables]       # ascending, as the kernel lists them
    ex = pdist.TableRowExchange(world, gather_fn=lambda t, i: t[i.long()],
                                assign_fn=lambda t, i, r: t.index_copy_(0, i.long(), r),
                                add_fn=lambda t, i, r: t.index_add_(0, i.long(), r))
    ex(tables, touched)
    ok = True
    for n, t in zip(names, tables):
        # the mean over the two ranks' B-sample means = the gradient of the 2B-sample mean
        ok = ok and float((t - ref["grads"][n]).abs().max()) <= 1e-7 + 1e-5 * float(ref["grads"][n].abs().max())
        ok = ok and bool((t[ref["grads"][n].abs().amax(1) == 0] == 0).all())          # untouched rows stay exactly zero
    # all ranks hold bit-identical tables (fixed summation order: rank 0's list, then rank 1's)
    both = [torch.empty_like(tables[0]) for _ in range(world)]
    dist.all_gather(both, tables[0])
    ok = ok and torch.equal(both[0], both[1])
    ok = ok and ex.last_bytes["row_lists_per_rank"] < ex.last_bytes["dense_tables"] // 20
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_type_table_row_list_exchange_world2_equals_the_concatenated_batch():
    """Two ranks, each with the oracle's gradients of its half batch (T = 3000 > 512, 25 live types): after the row-list
    exchange both hold the gradients of the single-process step on the concatenated batch; far fewer bytes than the
    dense tables."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
    assert res == {0: True, 1: True}


# ------------------------------------------------------------------ ONE communicator, one launch order per step
def _order_worker(rank, world, port, q):
    """A step's three collectives (two lookup rounds on the loader's side, the gradient exchange on the step's) all go through
    the ONE object make_exchange returned: every rank issues the same sequence of calls."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from p_companion_amd import distributed as pdist
    pdist.init_from_env("cpu")
    ex = pdist.make_exchange(world, rank=rank, kind="auto")          # gloo: torch.distributed behind the slot AND behind the lookups
    ok = not ex.native and callable(ex.all_to_all) and callable(ex.all_reduce_sum_f64_)
    calls = []
    a2a, ar = ex.all_to_all, ex.all_reduce_sum_f64_
    ex.all_to_all = lambda s, r: (calls.append(("a2a", s.numel() * s.element_size())), a2a(s, r))[1]
    ex.all_reduce_sum_f64_ = lambda t: (calls.append(("sum64", t.numel())), ar(t))[1]
    g = torch.Generator().manual_seed(0)
    full = torch.randn(1000, 16, generator=g)
    tab = pdist.ShardedFeatureTable(pdist.ShardedFeatureTable.shard(full, rank, world), 1000, rank, world,
                                    gather_fn=lambda t, i: torch.cat([t, torch.zeros(1, 16)])[i.long()], bucket_fn=_cpu_bucket, exchange=ex)
    wantf = torch.cat([full, torch.zeros(1, 16)])
    for step in range(3):
        gb = torch.Generator().manual_seed(300 + 10 * step + rank)
        B, K = 8, 5
        nb = torch.sort(torch.randperm(1000, generator=gb)[:20]).values.to(torch.int32)
        batch = {"anchor_idx": torch.randint(0, 1000, (B,), generator=gb, dtype=torch.int32),
                 "positive_idx": torch.randint(0, 1000, (B,), generator=gb, dtype=torch.int32),
                 "negative_idx": torch.randint(0, 1000, (B, K), generator=gb, dtype=torch.int32),
                 "neighbor_compact": {"nb_rows": torch.cat([nb, torch.tensor([-1], dtype=torch.int32)]), "weight": torch.ones(21),
                                      "slot_row": torch.zeros(B, 4, dtype=torch.int32), "n_unique": 20,
                                      "n_unique_dev": torch.tensor([20], dtype=torch.int32)}}
        rows, rb = tab.lookup_batch(batch)
        ext = torch.cat([rows, torch.zeros(1, 16)])
        ok = ok and torch.equal(ext[rb["anchor_idx"].long()], wantf[batch["anchor_idx"].long()])
        stats = torch.full((6,), float(rank + 1), dtype=torch.float64)
        ex.all_reduce_sum_f64_(stats)                                 # (cross-replica BatchNorm sums of the step)
        ok = ok and torch.equal(stats, torch.full((6,), 3.0, dtype=torch.float64))
        grad = torch.full((10,), float(rank + 1))
        ex.register(grad)
        ex.all_reduce_mean_ = None                                   # (the slot itself is driven through pc_exchange_adam on the GPU)
        pdist.all_reduce_mean_(grad, world)
        ok = ok and torch.allclose(grad, torch.full((10,), 1.5))
    seqs = [None] * world
    dist.all_gather_object(seqs, calls)
    ok = ok and seqs[0] == seqs[1] and [c[0] for c in calls] == ["a2a", "a2a", "sum64"] * 3
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_one_exchange_object_carries_every_collective_of_a_step_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_order_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
    assert res == {0: True, 1: True}


def test_rendezvous_deadline_ends_the_process():
    """make_exchange's construction and probe run under a host deadline: a rank whose peer never joins leaves with exit code 75
    (the launcher then ends the job) instead of sitting in the rendezvous; exceptions inside are re-raised to the caller."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import time; from p_companion_amd.distributed import _run_with_deadline as r; "
            "r(lambda: time.sleep(30), 0.5, 'test rendezvous', 0); print('returned')")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=60)
    assert out.returncode == 75 and "returned" not in out.stdout and "did not return" in out.stderr
    from p_companion_amd.distributed import _run_with_deadline
    assert _run_with_deadline(lambda: 41 + 1, 5.0, "quick", 0) == 42
    with pytest.raises(KeyError):
        _run_with_deadline(lambda: {}["x"], 5.0, "raises", 0)
    # on_expire="abandon" (the native exchange's probe): the caller gets control back and can agree on the fallback
    import time
    from p_companion_amd.distributed import _Expired, probe_timeout_s, group_timeout_s
    t0 = time.perf_counter()
    with pytest.raises(_Expired):
        _run_with_deadline(lambda: time.sleep(5), 0.3, "parked rendezvous", 0, on_expire="abandon")
    assert time.perf_counter() - t0 < 2.0
    assert probe_timeout_s() <= 90 and probe_timeout_s() <= group_timeout_s() <= 300       # first contact resolves inside a 10-minute harness limit
