"""Child process of tests/test_gpu_sharded.py::test_sharded_step_world2_on_one_card: one rank of a 2-rank job
(gloo rendezvous on 127.0.0.1, both ranks on the one MI355X of the box).  Each rank owns rows r % 2 == rank of the
feature table, builds ITS batches with the device sampler (seed 1 + rank), fetches their rows through the
device-resident sharded lookup (pc_shard_bucket + two all_to_all rounds + HIP row gather) and runs the fused step;
the same batch over the replicated table must give the same loss and gradients bit for bit (the gathered rows are the
same numbers; the step is deterministic).  Writes {"ok": bool, ...} as JSON to argv[3]."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, port, out = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0",
                      PC_DIST_BACKEND="gloo", PC_FORCE_DEVICE="0")
    res = {"ok": False, "rank": rank}
    try:
        from types import SimpleNamespace
        import torch
        import torch.distributed as dist
        from p_companion_amd import distributed as pdist
        from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
        from p_companion_amd.product2vec import Product2Vec
        r, w, _ = pdist.init_from_env("cuda")
        assert (r, w) == (rank, 2)
        dev = torch.device("cuda", 0)
        cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                              MARGIN=1.0, DEVICE=dev)
        bpg = generate_scaled_bpg(20000, 100, seed=0)
        table = bpg.cuda(dev)["features"]
        sharded = pdist.ShardedFeatureTable(pdist.ShardedFeatureTable.shard(table, rank, 2), bpg.num_products, rank, 2)
        ld_s = SimilarityIndexLoader(bpg, 1024, seed=1 + rank, drop_last=True, device=dev, sharded=sharded)
        ld_r = SimilarityIndexLoader(bpg, 1024, seed=1 + rank, drop_last=True, device=dev)
        torch.manual_seed(0)
        m_s, m_r = Product2Vec(cfg).to(dev).train(), Product2Vec(cfg).to(dev).train()
        m_r.load_state_dict(m_s.state_dict())
        steps = 0
        worst = 0.0
        for bs, br in zip(ld_s, ld_r):
            assert bs["table"].shape == (2 * sharded.capacity, 128)
            ls = m_s.train_step_indexed(bs["table"], bs)
            lr = m_r.train_step_indexed(table, br)
            gs, gr = m_s.flatten_parameters()[1], m_r.flatten_parameters()[1]
            worst = max(worst, float((gs - gr).abs().max()), abs(float(ls) - float(lr)))
            pdist.all_reduce_mean_(gs, 2)                       # the data-parallel gradient exchange, both ranks
            steps += 1
            if steps == 3:
                break
        res.update(ok=bool(worst == 0.0 and sharded.overflowed() == 0), worst=worst, steps=steps,
                   capacity=sharded.capacity, bytes_per_peer=sharded.bytes_per_peer)
        res["sharded_optimizer"] = sharded_optimizer_check(rank, dev)
        res["hot_set"] = hot_set_check(rank, dev, bpg, table)
        res["ok"] = bool(res["ok"] and res["sharded_optimizer"]["ok"] and res["hot_set"]["ok"])
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                       # noqa: BLE001 -- reported to the parent test
        import traceback
        res["error"] = traceback.format_exc()
        res["ok"] = False
    with open(out, "w") as f:
        json.dump(res, f)


def hot_set_check(rank, dev, bpg, table):
    """The replicated hot set at world 2 (two ranks on the one card, gloo): Zipf negatives, the 512 most popular products
    replicated on both ranks.  Against the same loader without the hot set: the rows the batch resolves to and the fused step's
    loss / gradients bit for bit, the request lists shorter by exactly the entries the replica served."""
    from types import SimpleNamespace
    import torch
    from p_companion_amd import distributed as pdist
    from p_companion_amd.data import SimilarityIndexLoader
    from p_companion_amd.product2vec import Product2Vec
    H = 512
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0, DEVICE=dev)
    local = pdist.ShardedFeatureTable.shard(table, rank, 2)
    plain = pdist.ShardedFeatureTable(local, bpg.num_products, rank, 2)
    hot = pdist.ShardedFeatureTable(local, bpg.num_products, rank, 2, hot_rows=H)
    mk = lambda sh: SimilarityIndexLoader(bpg, 1024, seed=7 + rank, drop_last=True, device=dev, sharded=sh, negatives="zipf", prefetch=False)
    ld_p, ld_h = mk(plain), mk(hot)                               # (capacity agreement + build_hot_replica: collectives, both ranks)
    same = torch.equal(hot.hot_replica, table[:H])
    torch.manual_seed(0)
    m_p, m_h = Product2Vec(cfg).to(dev).train(), Product2Vec(cfg).to(dev).train()
    m_h.load_state_dict(m_p.state_dict())
    zrow = torch.zeros(1, 128, device=dev)
    served_sum = saved_sum = 0
    for n, (bp, bh) in enumerate(zip(ld_p, ld_h)):
        used_p, used_h = int(plain._bufs["counts"].sum()), int(hot._bufs["counts"].sum())
        ep, eh = torch.cat([bp["table"], zrow]), torch.cat([bh["table"], zrow])
        for k in ("anchor_idx", "positive_idx", "negative_idx"):
            same = same and torch.equal(eh[bh[k].long()], ep[bp[k].long()])
        served = hot.hot_rows_served()
        served_sum += served
        saved_sum += used_p - used_h
        lp, lh = m_p.train_step_indexed(bp["table"], bp), m_h.train_step_indexed(bh["table"], bh)
        same = same and torch.equal(lp, lh) and torch.equal(m_p.flatten_parameters()[1], m_h.flatten_parameters()[1])
        if n == 2:
            break
    return {"ok": bool(same and served_sum == saved_sum > 0 and hot.overflowed() == 0), "same": bool(same), "served": served_sum,
            "request_slots_saved": saved_sum, "table_rows": int(bh["table"].shape[0]), "capacity": int(hot.capacity)}


def sharded_optimizer_check(rank, dev):
    """ABI 8 at world 2: the joint step's optimizer sharded over the replicas (reduce-scatter of the flat gradient, Adam on this
    rank's half of the flat buffers, all-gather of the parameters) against the all-reduce + whole-Adam form -- the same
    parameters bit for bit (two addends commute), moments untouched outside this rank's half, both ranks identical."""
    from types import SimpleNamespace
    import torch
    import torch.distributed as dist
    from p_companion_amd import distributed as pdist
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader, generate_scaled_bpg
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
    T, B = 601, 256
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=0.0, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3,
                          NUM_TYPES=T, DEVICE=dev)
    bpg = generate_scaled_bpg(3000, 40, seed=3)
    out = {}

    def make(shard):
        torch.manual_seed(5)
        m = PCompanion(cfg, bpg.cuda(dev)["features"]).to(dev).train()
        o = FusedAdam(m, lr=1e-2)
        ex = pdist.make_exchange(2, rank=rank, kind="callback", device=dev)
        g = GraphedJointStep(m, o, B, warmup=1, mode="direct", exchange=ex, shard_optimizer=shard)
        ld = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), B, shuffle=True, seed=20 + rank, device=dev,
                                      out=g.static)
        return m, o, g, ld

    m_a, o_a, g_a, ld_a = make(False)
    m_b, o_b, g_b, ld_b = make(True)
    assert g_b.shard_optimizer and not g_a.shard_optimizer
    n_real = sum(p.numel() for _, p in m_b._named_flat())
    flat_b = m_b.flatten_parameters()[0]
    assert flat_b.numel() % 2 == 0 and flat_b.numel() == n_real            # (29 024 + 2 * 64 T floats: even at any T)
    steps = 0
    for ba, bb in zip(ld_a, ld_b):
        if ba["query_idx"].numel() != B:
            continue
        la, _ = g_a(ba)
        lb, _ = g_b(bb)
        assert torch.equal(la, lb), (steps, la, lb)
        steps += 1
        if steps == 4:
            break
    got = g_b.run_epoch(ld_b, drop_last=True, max_steps=3)            # pc_joint_train_epoch_plan, shard_optimizer = 1
    ref = g_a.run_epoch(ld_a, drop_last=True, max_steps=3)
    same = torch.equal(got, ref)
    for (k, pa), (_, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        same = same and torch.equal(pa, pb)
    half = flat_b.numel() // 2
    lo, hi = rank * half, (rank + 1) * half
    own = o_b.exp_avg[lo:hi]
    other = torch.cat([o_b.exp_avg[:lo], o_b.exp_avg[hi:]])
    # both ranks hold the same parameters: compare a checksum over the process group
    chk = torch.stack([flat_b.double().sum(), flat_b.double().abs().sum()]).cpu()
    both = [torch.zeros_like(chk) for _ in range(2)]
    dist.all_gather(both, chk)
    out.update(ok=bool(same and float(other.abs().max()) == 0.0 and float(own.abs().max()) > 0.0 and torch.equal(both[0], both[1])
                       and int(o_a.step_count) == int(o_b.step_count) == 7),
               same=bool(same), moments_outside_own_half=float(other.abs().max()), ranks_equal=bool(torch.equal(both[0], both[1])),
               steps=int(o_b.step_count), flat=int(flat_b.numel()), real=int(n_real))
    return out


if __name__ == "__main__":
    main()
