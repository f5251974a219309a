"""Row-sharded feature table on the GPU (BASELINE configs[3]/[4]): device-resident bucketing, the
HIP owner-side gather, the fused step over the exchanged rows.  world = 1 in this process; world = 2 as two child
processes sharing the card over gloo (started by conftest.py before this process touches the GPU)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cfg():
    from types import SimpleNamespace
    return SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                           MARGIN=1.0, DEVICE=torch.device("cuda"))


def test_shard_bucket_properties_config4_shape():
    """One rank of BASELINE configs[3]: 10 M products over 8 ranks, B = 4096 (its ~160 k lookups per step): every id gets
    a slot in its owner's list, lists hold the owner-local row, padding maps to -1, nothing overflows at the default
    capacity, the non-live tail of the neighbour list is ignored."""
    from p_companion_amd import ops
    from p_companion_amd.distributed import ShardedFeatureTable
    G, P, B, K, N = 8, 10_000_000, 4096, 5, 32
    g = torch.Generator().manual_seed(0)
    a = torch.randint(0, P, (B,), generator=g, dtype=torch.int32).cuda()
    p = torch.randint(0, P, (B,), generator=g, dtype=torch.int32).cuda()
    ng = torch.randint(0, P, (B * K,), generator=g, dtype=torch.int32).cuda()
    n_unique = 88000
    nb = torch.sort(torch.randperm(P, generator=g)[:n_unique]).values.to(torch.int32)
    nb_rows = torch.cat([nb, torch.tensor([-1], dtype=torch.int32), torch.full((5000,), 123456789, dtype=torch.int32)]).cuda()
    n_dev = torch.tensor([n_unique], dtype=torch.int32).cuda()
    n_ids = 2 * B + B * K + nb_rows.numel()
    C = ShardedFeatureTable.capacity_for(n_ids, G)
    counts = torch.zeros(G, dtype=torch.int32).cuda()
    send = torch.zeros(G * C, dtype=torch.int32).cuda()
    over = torch.zeros(1, dtype=torch.int32).cuda()
    outs = ops.shard_bucket([(a, None, 0), (nb_rows, n_dev, 1), (p, None, 0), (ng, None, 0)], G, C, counts, send, over)
    assert int(over) == 0
    send_h, counts_h = send.cpu().numpy().reshape(G, C), counts.cpu().numpy()
    live = [a.cpu().numpy(), nb_rows.cpu().numpy()[:n_unique + 1], p.cpu().numpy(), ng.cpu().numpy()]
    assert counts_h.sum() == sum(int((x >= 0).sum()) for x in live)
    assert counts_h.max() < C and abs(counts_h.max() - counts_h.mean()) < 6 * np.sqrt(counts_h.mean())
    used = np.zeros(G * C, bool)
    for ids, rm in zip(live, outs):
        rm = rm.cpu().numpy()
        assert np.all(rm[len(ids):] == -1)                       # the scratch tail of the neighbour list
        rm = rm[:len(ids)]
        assert np.array_equal(rm < 0, ids < 0)
        ok = ids >= 0
        owner, slot = rm[ok] // C, rm[ok] % C
        assert np.array_equal(owner, ids[ok] % G)
        assert np.array_equal(send_h[owner, slot], ids[ok] // G)
        assert not used[rm[ok]].any() and len(np.unique(rm[ok])) == ok.sum()      # every occurrence its own slot
        used[rm[ok]] = True
    for o in range(G):
        assert np.all(send_h[o, counts_h[o]:] == -1)
    # a capacity that is too small is reported, not overrun
    small = torch.zeros(G * 100, dtype=torch.int32).cuda()
    ops.shard_bucket([(a, None, 0)], G, 100, counts, small, over)
    assert int(over) == B - 8 * 100 and int(counts.sum()) == B


@pytest.mark.parametrize("products,batch", [(30000, 1024), (1_250_000, 4096)])
def test_sharded_step_world1_equals_replicated(products, batch):
    """The chain bench.py --table sharded runs -- loader (unique neighbour layout) -> pc_shard_bucket -> HIP gather ->
    fused step over the gathered buffer -- against the same batches over the replicated table: loss, gradients and
    BatchNorm statistics bit for bit.  1.25 M rows = one rank's shard of the 10 M-product configuration."""
    from p_companion_amd import distributed as pdist
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import Product2Vec
    bpg = generate_scaled_bpg(products, 100, seed=2)
    table = bpg.cuda()["features"]
    sharded = pdist.ShardedFeatureTable(table, bpg.num_products, 0, 1)
    ld_s = SimilarityIndexLoader(bpg, batch, seed=3, drop_last=True, sharded=sharded)
    ld_r = SimilarityIndexLoader(bpg, batch, seed=3, drop_last=True)
    torch.manual_seed(0)
    m_s, m_r = Product2Vec(cfg()).cuda().train(), Product2Vec(cfg()).cuda().train()
    m_r.load_state_dict(m_s.state_dict())
    n = 0
    for bs, br in zip(ld_s, ld_r):
        assert "table" in bs and bs["table"].shape[0] == sharded.capacity
        assert torch.equal(bs["neighbor_compact"]["slot_row"], br["neighbor_compact"]["slot_row"])
        ls, lr = m_s.train_step_indexed(bs["table"], bs), m_r.train_step_indexed(table, br)
        assert torch.equal(ls, lr)
        assert torch.equal(m_s.flatten_parameters()[1], m_r.flatten_parameters()[1])
        n += 1
        if n == 3:
            break
    assert torch.equal(m_s.ffn[1].running_var, m_r.ffn[1].running_var)
    assert sharded.overflowed() == 0
    sharded.raise_if_overflowed()


def test_sharded_lookup_makes_no_host_sync():
    """The per-step exchange must not wait for the device: with a long kernel queued in front, lookup_batch returns
    while that kernel is still running (the stream is not idle when the call comes back)."""
    from p_companion_amd import distributed as pdist
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    bpg = generate_scaled_bpg(30000, 100, seed=2)
    table = bpg.cuda()["features"]
    sharded = pdist.ShardedFeatureTable(table, bpg.num_products, 0, 1)
    ld = SimilarityIndexLoader(bpg, 1024, seed=3, drop_last=True, prefetch=False)
    batch = next(iter(ld))
    sharded.lookup_batch(batch)                                    # allocate the buffers
    torch.cuda.synchronize()
    x = torch.randn(8192, 8192, device="cuda")
    ev = torch.cuda.Event()
    for _ in range(20):
        x = x @ x * 1e-4                                           # ~20 x 1.1 TFLOP of queued work
    ev.record()
    sharded.lookup_batch(batch)
    assert not ev.query(), "lookup_batch waited for the device"
    torch.cuda.synchronize()


def test_sharded_step_world2_on_one_card(world2_job):
    """Two ranks (child processes, gloo) on the one card: the sharded step of each rank equals its replicated step bit
    for bit; the bucket capacity and per-peer bytes are the documented ones."""
    results = world2_job()
    assert len(results) == 2
    assert all(r.get("ok") for r in results), "\n".join(str(r.get("error", r)) for r in results)
    for r in results:
        assert r["steps"] == 3 and r["worst"] == 0.0
        assert r["bytes_per_peer"]["rows"] == 512 * r["capacity"]
        # ABI 8: reduce-scatter -> Adam on the rank's half -> all-gather == all-reduce + whole Adam, bit for bit at world 2
        so = r["sharded_optimizer"]
        assert so["ok"] and so["same"] and so["moments_outside_own_half"] == 0.0 and so["ranks_equal"], so
        assert so["steps"] == 7 and so["flat"] == so["real"]
        # the replicated hot set under the sharded table at world 2: same rows, same step, shorter request lists
        hs = r["hot_set"]
        assert hs["ok"] and hs["same"] and hs["served"] == hs["request_slots_saved"] > 0 and hs["table_rows"] == 2 * hs["capacity"] + 512, hs


@pytest.mark.parametrize("popularity", ["identity", "permuted"])
def test_hot_set_serves_the_zipf_head_from_the_replica(popularity):
    """BASELINE configs[4]'s hot rows: the H most popular products replicated behind the exchange buffer
    (pc_shard_bucket_hot).  Same loader seed with and without the hot set: identical id batches, the rows every index of the
    batch resolves to are bit-identical, the fused step's loss and gradients are bit-identical, and the request lists shrink
    by exactly the entries served -- the Zipf head's share of the negatives (about a third at H = 1024 over 200 k products)."""
    from p_companion_amd import distributed as pdist
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import Product2Vec
    P, B, H = 200_000, 2048, 1024
    bpg = generate_scaled_bpg(P, 100, seed=2)
    table = bpg.cuda()["features"]
    pop = None
    if popularity == "permuted":
        pop = torch.randperm(P, generator=torch.Generator().manual_seed(11)).to(torch.int32)
    hot_ids = pdist.ShardedFeatureTable.hot_ids_from_popularity(pop, H)
    plain = pdist.ShardedFeatureTable(table, P, 0, 1)
    hot = pdist.ShardedFeatureTable(table, P, 0, 1, hot_rows=H, hot_ids=hot_ids)
    mk = lambda sh: SimilarityIndexLoader(bpg, B, seed=3, drop_last=True, sharded=sh, negatives="zipf",
                                          popularity=None if pop is None else pop.numpy(), prefetch=False)
    ld_p, ld_h = mk(plain), mk(hot)
    want_rep = table[(hot_ids if hot_ids is not None else torch.arange(H)).long().cuda()]
    assert torch.equal(hot.hot_replica, want_rep)                                # built at the loader's construction
    torch.manual_seed(0)
    m_p, m_h = Product2Vec(cfg()).cuda().train(), Product2Vec(cfg()).cuda().train()
    m_h.load_state_dict(m_p.state_dict())
    zrow = torch.zeros(1, 128, device="cuda")
    served_total = neg_total = 0
    for n, (bp, bh) in enumerate(zip(ld_p, ld_h)):
        used_p, used_h = int(plain._bufs["counts"].sum()), int(hot._bufs["counts"].sum())
        assert bh["table"].shape[0] == hot.capacity + H and bp["table"].shape[0] == plain.capacity
        assert torch.equal(bh["table"][hot.capacity:], want_rep)
        ep, eh = torch.cat([bp["table"], zrow]), torch.cat([bh["table"], zrow])
        for k in ("anchor_idx", "positive_idx", "negative_idx"):
            assert torch.equal(eh[bh[k].long()], ep[bp[k].long()]), k            # the same rows, bit for bit
        nu = int(bp["neighbor_compact"]["n_unique"])
        assert torch.equal(eh[bh["neighbor_compact"]["nb_rows"][:nu + 1].long()], ep[bp["neighbor_compact"]["nb_rows"][:nu + 1].long()])
        served = hot.hot_rows_served()
        assert used_p - used_h == served > 0                                     # request-list occupancy down by what the replica served
        in_replica = int((bh["negative_idx"] >= hot.capacity).sum())
        assert in_replica <= served
        served_total += in_replica
        neg_total += bh["negative_idx"].numel()
        lp, lh = m_p.train_step_indexed(bp["table"], bp), m_h.train_step_indexed(bh["table"], bh)
        assert torch.equal(lp, lh) and torch.equal(m_p.flatten_parameters()[1], m_h.flatten_parameters()[1])
        if n == 2:
            break
    share = served_total / neg_total
    # Zipf(1) over P ranks: P(rank <= H) = H_H / H_P = 7.51 / 12.78 = 0.59 of the PROPOSALS at P = 200 k (0.37 at 100 M); rejections
    # (the anchor, its positives, duplicates within a sample's five) move it a little
    assert 0.45 < share < 0.70, share
    assert hot.overflowed() == 0 and plain.overflowed() == 0
    with pytest.raises(ValueError):
        pdist.ShardedFeatureTable(table, P, 0, 1, hot_rows=4, hot_ids=torch.tensor([3, 2, 1, 0]))
    with pytest.raises(RuntimeError):
        pdist.ShardedFeatureTable(table, P, 0, 1, hot_rows=4, capacity=plain.capacity).lookup_batch(
            next(iter(SimilarityIndexLoader(bpg, B, seed=3, drop_last=True, prefetch=False))))
