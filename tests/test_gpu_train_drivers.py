"""Phase drivers and metrics on the GPU: Metrics.evaluate_model vs the reference's golden value,
pretrain_product2vec / train end to end on the reference's own 1k-product graph, checkpoint
layouts.  Needs an MI355X."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import joint_oracle


def cfg(tmp, **over):
    c = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                        MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=100, DEVICE=torch.device("cuda"),
                        LEARNING_RATE=1e-3, BATCH_SIZE=256, PRODUCT2VEC_EPOCHS=1, NUM_EPOCHS=2, MODEL_DIR=str(tmp))
    c.__dict__.update(over)
    return c


def test_metrics_golden(golden, tmp_path):
    from p_companion_amd.metrics import Metrics
    from p_companion_amd.p_companion import PCompanion
    g = golden("g8_metrics.npz")
    st = {k[5:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("init.")}
    c = cfg(tmp_path)
    model = PCompanion(c, st["product_embeddings.weight"])
    model.load_state_dict(st)
    model = model.to(c.DEVICE)
    batch = {"query_idx": torch.from_numpy(g["query_idx"]), "query_types": torch.from_numpy(g["query_types"]),
             "positive_items": torch.from_numpy(g["positive_items"]), "target_features": torch.from_numpy(g["target_features"])}
    m = Metrics.evaluate_model(model, [batch], c.DEVICE)
    for name, val in zip(g["metric_names"], g["metric_values"]):
        assert abs(m[str(name)] - float(val)) < 1e-5, (name, m[str(name)], float(val))
    # rows >= B can never hit (metrics.py:95-100): hit@k is bounded by 1/K
    assert m["hit@10"] <= 1.0 / 3 + 1e-6


def test_hit_rank_and_cosine_vs_torch():
    from p_companion_amd import ops
    g = torch.Generator().manual_seed(0)
    sims = torch.randn(30, 10, generator=g)
    rank = ops.hit_rank(sims.cuda()).cpu()
    for k in (1, 3, 10):
        top = torch.topk(sims, k, dim=1).indices
        want = (top == torch.arange(30).unsqueeze(1)).any(1)
        assert torch.equal(rank < k, want)
    x, y = torch.randn(7, 3, 128, generator=g), torch.randn(7, 128, generator=g)
    cos = ops.cosine_rows(x.cuda(), y.cuda()).cpu().view(7, 3)
    np.testing.assert_allclose(cos, torch.cosine_similarity(x, y.unsqueeze(1), dim=-1), atol=1e-6)


def test_pipeline_end_to_end(golden, tmp_path):
    """train.py:main() on the reference's own 1k-product graph: both phases, checkpoints in the
    reference's dict layouts, the loss falling over the epochs of both phases, metrics in range, the checkpoint the
    best epoch's.  (Values against the reference's own run of the same loops: tests/test_gpu_epoch_goldens.py -- there the
    negatives / shuffles are the reference's; here they are the throughput samplers' own.)"""
    from p_companion_amd import train as drv
    from p_companion_amd.data import (ComplementaryIndexDataset, ComplementaryIndexLoader, IntBPG,
                                      SimilarityIndexLoader)
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    c = cfg(tmp_path, NUM_TYPES=bpg.n_types, PRODUCT2VEC_EPOCHS=3)
    torch.manual_seed(0)
    c.RECORD_STEP_LOSSES = True
    emb = drv.pretrain_product2vec(c, bpg)
    assert len(emb) == 1000
    p2v = drv.pretrain_product2vec.last_model
    per_epoch = p2v.step_losses.view(3, -1).mean(1)               # 12 steps per epoch (2949 pairs, B = 256)
    assert p2v.step_losses.numel() == 36 and torch.isfinite(p2v.step_losses).all()
    assert per_epoch[2] < per_epoch[1] < per_epoch[0] and 0.3 < float(per_epoch[2]) < float(per_epoch[0]) < 1.2
    # (the reference's own two epochs on this graph go 0.99 -> 0.55, tests/golden/g10_p2v_epochs.npz)
    assert int(p2v.state_dict()["ffn.1.num_batches_tracked"]) == 4 * 36
    E = torch.stack([emb[f"P{i:06d}"] for i in range(1000)])
    assert torch.isfinite(E).all() and float(E.std()) > 1e-3
    ck = torch.load(os.path.join(c.MODEL_DIR, "product2vec.pth"), weights_only=True)
    assert set(ck) == {"model_state_dict", "embeddings", "type_to_idx"}                # pretrain_product2vec.py:44-49
    assert "ffn.1.running_var" in ck["model_state_dict"] and ck["embeddings"]["P000999"].shape == (128,)
    tr = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), 256, shuffle=True)
    va = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "val"), 256, shuffle=False)
    b = next(iter(tr))
    assert b["positive_types"].shape == (256, 1) and b["query_types"].dtype == torch.int32
    lab = b["label"].cpu().numpy()
    tt = bpg.type_idx[bpg.complementary_pairs[0, 1]]            # spot-check the label rules (data_loader.py:148-153)
    assert set(np.unique(lab).tolist()) <= {-1, 1}
    assert bool((b["positive_types"].cpu().numpy()[lab == -1] == 0).all())
    model = drv.train(c, tr, va, emb)
    best = torch.load(os.path.join(c.MODEL_DIR, "best_model.pth"), weights_only=True)
    assert set(best) == {"epoch", "model_state_dict", "optimizer_state_dict", "metrics"}      # train.py:63-70
    assert set(best["metrics"]) == {"hit@1", "hit@3", "hit@10", "type_diversity", "mean_relevance"}
    n_steps = 2 * len(tr)
    assert model.step_losses.numel() == n_steps and torch.isfinite(model.step_losses).all()
    first, second = model.step_losses[:len(tr)].mean(), model.step_losses[len(tr):].mean()
    assert second < first and 0.5 < float(second) < float(first) < 2.0           # (the reference's run: 1.35 -> 0.85, g11)
    assert len(model.epoch_metrics) == 2
    for m in model.epoch_metrics:
        assert 0.0 <= m["hit@1"] <= m["hit@3"] <= m["hit@10"] <= 1.0 / 3 + 1e-6     # rows >= B can never hit (metrics.py:95-100)
        assert 0.0 < m["type_diversity"] <= 1.0 and -1.0 <= m["mean_relevance"] <= 1.0
    hits = [m["hit@10"] for m in model.epoch_metrics]
    want_epoch = 0 if hits[0] >= hits[1] else 1                                    # strict improvement only (train.py:62)
    assert best["epoch"] == want_epoch and best["metrics"] == model.epoch_metrics[want_epoch]
    steps_at_best = (want_epoch + 1) * len(tr)
    assert all(float(s["step"]) == steps_at_best for s in best["optimizer_state_dict"]["state"].values())
    # reference-style (unfused, torch Adam) loop on the same data also runs
    c2 = cfg(tmp_path, NUM_TYPES=bpg.n_types, NUM_EPOCHS=1)
    drv.train(c2, tr, va, emb, fused=False)


def test_complementary_batch_builder_rules(golden):
    """pc_build_complementary_batch vs the label rules of data_loader.py:148-153 (oracle) and the
    N(0,1) filler's moments."""
    from oracle import data_oracle
    from p_companion_amd import ops
    from p_companion_amd.data import ComplementaryIndexDataset, IntBPG
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    ds = ComplementaryIndexDataset(bpg, "train", seed=3)
    rows = np.ascontiguousarray(ds.pairs[:512], np.int32)
    g = bpg.cuda()
    b = ops.build_complementary_batch(torch.from_numpy(rows).cuda(), g["features"], g["type_idx"], bpg.n_types, 11, 5)
    feats = bpg.features
    filler = []
    for i, (q, t, lab) in enumerate(rows):
        want = data_oracle.complementary_sample_ints(q, t, lab, bpg.type_idx, bpg.n_types)
        assert int(b["query_idx"][i]) == q and int(b["query_types"][i]) == want["query_types"]
        assert int(b["positive_types"][i, 0]) == want["positive_types"]
        assert int(b["negative_types"][i, 0]) == want["negative_types"]
        real, fill = ("positive_items", "negative_items") if lab == 1 else ("negative_items", "positive_items")
        assert np.array_equal(b[real][i].cpu().numpy(), feats[t])
        assert np.array_equal(b["target_features"][i].cpu().numpy(), feats[t])
        filler.append(b[fill][i].cpu().numpy())
    f = np.concatenate(filler)
    assert abs(f.mean()) < 0.02 and abs(f.std() - 1.0) < 0.02 and abs((f ** 3).mean()) < 0.05
    b2 = ops.build_complementary_batch(torch.from_numpy(rows).cuda(), g["features"], g["type_idx"], bpg.n_types, 11, 6)
    assert not torch.equal(b2["negative_items"], b["negative_items"])            # a new step draws new fillers


def test_type_filtered_retrieval_vs_oracle(golden, tmp_path):
    """PCompanionInference.recommend (inference.py:64-124): model forward, then per predicted type
    top-n products of that type by <projected embedding, features>.  Candidate search on the GPU
    (pc_retrieve_topk) against the numpy restatement, on the reference's own 1k-product graph;
    round trip through a best_model.pth-layout checkpoint."""
    from p_companion_amd.data import IntBPG
    from p_companion_amd.inference import PCompanionInference
    from p_companion_amd.p_companion import PCompanion
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    c = cfg(tmp_path, NUM_TYPES=bpg.n_types)
    torch.manual_seed(3)
    table = torch.randn(bpg.num_products, 128)
    model = PCompanion(c, table)
    path = os.path.join(str(tmp_path), "best_model.pth")
    torch.save({"epoch": 0, "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "optimizer_state_dict": {}, "metrics": {}}, path)
    ids = ["P%06d" % i for i in range(bpg.num_products)]
    inf = PCompanionInference(path, c, bpg, product_ids=ids)

    q = torch.arange(0, 200, dtype=torch.int32)
    types, idx, sc = inf.recommend_batch(q, 10)
    out = inf.model({"query_idx": q.cuda(), "query_types": inf.type_idx[q.long().cuda()]})
    proj = out["projected_embeddings"].reshape(-1, 128).cpu().numpy()
    tflat = out["complementary_types"].reshape(-1).cpu().numpy()
    ref = joint_oracle.recommend(proj, tflat, bpg.type_idx, bpg.features, 10)
    idx, sc = idx.reshape(-1, 10).cpu().numpy(), sc.reshape(-1, 10).cpu().numpy()
    for r, (rid, rsc) in enumerate(ref):
        k = len(rid)
        assert (idx[r, k:] == -1).all()
        np.testing.assert_allclose(sc[r, :k], rsc, rtol=1e-5, atol=1e-5)
        assert (idx[r, :k] == rid).all() or np.allclose(np.sort(sc[r, :k]), np.sort(rsc), atol=1e-5)
        assert (bpg.type_idx[idx[r, :k]] == tflat[r]).all()              # only products of the predicted type

    rec = inf.recommend("P000007", num_recommendations=5)
    assert rec["complementary_types"] == types[7].tolist()
    assert len(rec["recommendations"]) == len(rec["scores"]) <= 3
    assert all(len(x) <= 5 and all(isinstance(p, str) for p in x) for x in rec["recommendations"])
    with pytest.raises(ValueError):
        inf.recommend("P999999")
    with pytest.raises(FileNotFoundError):
        PCompanionInference(os.path.join(str(tmp_path), "missing.pth"), c, bpg)


@pytest.mark.parametrize("mode", ["direct", "graph"])
def test_graphed_joint_step_equals_eager_steps(mode):
    """GraphedJointStep (HIP-graph replay of pc_joint_train_step + pc_adam_step, loader building into the graph's
    fixed buffers) against the same steps launched eagerly: the same losses, top-k and parameters after 8 steps, bit
    for bit (pc_joint_fused_step has no float atomics for T <= 512: fixed-order slab sums, one-hot table gradients)."""
    from types import SimpleNamespace
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader, generate_scaled_bpg
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=0.0, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3,
                          NUM_TYPES=40, DEVICE="cuda")
    bpg = generate_scaled_bpg(3000, 40, seed=3)
    B = 512

    def make():
        torch.manual_seed(5)
        m = PCompanion(cfg, bpg.cuda("cuda")["features"]).to("cuda").train()
        return m, FusedAdam(m, lr=1e-2)

    m_e, o_e = make()
    m_g, o_g = make()
    graphed = GraphedJointStep(m_g, o_g, B, warmup=2, mode=mode)
    ld_e = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), B, shuffle=True, seed=9, device="cuda")
    ld_g = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), B, shuffle=True, seed=9, device="cuda",
                                    out=graphed.static)
    steps = 0
    for be, bg in zip(ld_e, ld_g):
        if be["query_idx"].numel() != B:
            continue
        assert bg["positive_items"].data_ptr() == graphed.static["positive_items"].data_ptr()     # built in place
        le, te = m_e.train_step(be, optimizer=o_e)
        lg, tg = graphed(bg)
        assert torch.equal(te, tg), f"top-k differs at step {steps}"
        assert torch.equal(le, lg), f"losses differ at step {steps}: {le} vs {lg}"      # no float atomics: bit equality
        steps += 1
        if steps == 8:
            break
    assert steps == 8 and (graphed.graph if mode == "graph" else graphed.prepared) is not None     # 2 warm-up steps, 6 replays / direct calls
    assert mode == "graph" or graphed.prepared.calls == 6
    for (k, pe), (_, pg) in zip(m_e.named_parameters(), m_g.named_parameters()):
        assert torch.equal(pe, pg), k              # the fused step is bitwise reproducible (T <= 512), replay or eager
    with pytest.raises(ValueError):
        graphed({k: v[:7] for k, v in graphed.static.items()})


@pytest.mark.parametrize("types,dropout,k", [(40, 0.0, 3), (40, 0.1, 3), (300, 0.0, 3), (100, 0.0, 2), (128, 0.1, 4), (600, 0.1, 3), (900, 0.25, 2)])
def test_deferred_batches_built_inside_the_step(types, dropout, k):
    """ComplementaryIndexLoader(deferred=True) + GraphedJointStep: the batch is built by the step's first kernel
    (pc_joint_fused_step_pairs; T <= 128) or by the builder's launch inside the same call (larger tables).  Against the
    loader that builds first: the same losses, top-k and parameters bit for bit, and after each step the fixed buffers
    hold exactly the batch the builder writes (ids, type ids, feature rows and the N(0,1) filler bits)."""
    from types import SimpleNamespace
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader, generate_scaled_bpg
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=dropout, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=k,
                          NUM_TYPES=types, DEVICE="cuda")
    bpg = generate_scaled_bpg(3000, 40, seed=3)
    B = 500                                                       # (not a multiple of the 16-sample tile)

    def make():
        torch.manual_seed(5)
        m = PCompanion(cfg, bpg.cuda("cuda")["features"]).to("cuda").train()
        m.type_transition._dropout_seed = 1234
        return m, FusedAdam(m, lr=1e-2)

    m_a, o_a = make()
    m_b, o_b = make()
    g_a = GraphedJointStep(m_a, o_a, B, warmup=1, mode="direct")
    g_b = GraphedJointStep(m_b, o_b, B, warmup=1, mode="direct")
    ld_a = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), B, shuffle=True, seed=9, device="cuda", out=g_a.static)
    ld_b = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), B, shuffle=True, seed=9, device="cuda", out=g_b.static,
                                    deferred=True)
    steps = 0
    for ba, bb in zip(ld_a, ld_b):
        if ba["query_idx"].numel() != B:
            continue
        assert "_deferred" in bb and "_deferred" not in ba
        la, ta = g_a(ba)
        lb, tb = g_b(bb)
        assert "_deferred" not in bb
        assert torch.equal(ta, tb) and torch.equal(la, lb), (steps, la, lb)
        for k in ("query_idx", "query_types", "positive_types", "negative_types", "positive_items", "negative_items"):
            assert torch.equal(g_a.static[k], g_b.static[k]), (steps, k)
        steps += 1
        if steps == 5:
            break
    assert steps == 5 and g_b.prepared.calls == 4
    for (k, pa), (_, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert torch.equal(pa, pb), k
    # a deferred batch handed to anything else is built on request
    it = iter(ld_b)
    nb = next(it)
    assert "_deferred" in nb
    ld_b.materialize(nb)
    assert "_deferred" not in nb and bool((nb["query_types"] == bpg.cuda("cuda")["type_idx"][nb["query_idx"].long()]).all())


@pytest.mark.parametrize("types,dropout", [(40, 0.1), (600, 0.0), (600, 0.1)])
def test_run_epoch_equals_the_loop_over_the_loader(types, dropout):
    """GraphedJointStep.run_epoch (pc_joint_train_epoch: train.py:36-57 as one foreign call, ragged last batch included)
    against iterating the same loader and stepping batch by batch: per-step losses and the parameters after the epoch,
    bit for bit -- also at T = 600, the large-table path: its table gradients are the sorted form (reproducible at any number
    of touched rows) and, without dropout, the epoch call forms every step's distinct-type list one step AHEAD (a riding
    workgroup of the previous step's finish kernel) where the loop launches present_types_kernel per step: same list, same bits."""
    from types import SimpleNamespace
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader, generate_scaled_bpg
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=dropout, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3,
                          NUM_TYPES=types, DEVICE="cuda")
    bpg = generate_scaled_bpg(1500, 40, seed=3)
    B = 448

    def make(deferred):
        torch.manual_seed(5)
        m = PCompanion(cfg, bpg.cuda("cuda")["features"]).to("cuda").train()
        m.type_transition._dropout_seed = 77
        o = FusedAdam(m, lr=1e-2)
        g = GraphedJointStep(m, o, B, warmup=0, mode="direct")
        ld = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), B, shuffle=True, seed=2, device="cuda",
                                      out=g.static, deferred=deferred)
        return m, o, g, ld

    m_a, o_a, g_a, ld_a = make(False)
    m_b, o_b, g_b, ld_b = make(True)
    n = len(ld_a.dataset)
    ref = []
    for batch in ld_a:                                            # the plain loop: full batches through the fixed buffers,
        if batch["query_idx"].numel() == B:                       # the ragged last one through train_step
            ref.append(g_a(batch)[0].clone())
        else:
            ref.append(m_a.train_step(batch, optimizer=o_a)[0].clone())
    got = g_b.run_epoch(ld_b)
    assert got.shape == (len(ref), 3) and len(ref) == (n + B - 1) // B and n % B != 0
    exact = True
    for i, r in enumerate(ref):
        assert torch.equal(got[i], r) if exact else torch.allclose(got[i], r, rtol=1e-5, atol=1e-6), (i, got[i], r)
    for (k, pa), (_, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert torch.equal(pa, pb) if exact else torch.allclose(pa, pb, rtol=1e-4, atol=1e-6), k
    assert int(o_a.step_count) == int(o_b.step_count) == len(ref) and ld_a.step == ld_b.step == len(ref)
    # a second epoch continues the counters (loader step, dropout offset, Adam step)
    got2 = g_b.run_epoch(ld_b, drop_last=True)
    assert got2.shape[0] == n // B and int(o_b.step_count) == len(ref) + n // B and torch.isfinite(got2).all()


def test_pairs_path_clamps_and_counts_ids_outside_the_tables():
    """pc_joint_fused_step_pairs validates what it derives from the pairs: a query / target id outside the product table is
    clamped and counted (never dereferenced), the step stays finite, and the model raises when the counter is read."""
    from types import SimpleNamespace
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader, generate_scaled_bpg
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=0.0, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3,
                          NUM_TYPES=40, DEVICE="cuda")
    bpg = generate_scaled_bpg(2000, 40, seed=3)
    B = 128
    torch.manual_seed(5)
    m = PCompanion(cfg, bpg.cuda("cuda")["features"]).to("cuda").train()
    o = FusedAdam(m, lr=1e-2)
    g = GraphedJointStep(m, o, B, warmup=1, mode="direct")
    ld = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), B, shuffle=True, seed=4, device="cuda",
                                  out=g.static, deferred=True)
    it = iter(ld)
    g(next(it))                                                   # eager warm-up step (builds its batch itself)
    g(next(it))                                                   # first prepared step
    assert m.index_errors() == 0
    batch = next(it)
    loader, rows, step = batch["_deferred"]
    rows = rows.clone()
    rows[5, 0] = 2000                                             # query id == num_products
    rows[9, 1] = -3                                               # negative target id
    batch["_deferred"] = (loader, rows, step)
    losses, _ = g(batch)
    assert torch.isfinite(losses).all()
    with pytest.raises(IndexError):
        m.raise_index_errors()                                    # (reads and clears the device counter)
    rows[5, 0], rows[9, 1] = 0, 1
    rows[0, 0] = 5000
    batch["_deferred"] = (loader, rows, step + 1)
    g(batch)
    assert m.index_errors() == 1


def test_graphed_joint_step_with_grad_hook_equals_fused_adam():
    """The data-parallel form (fused step without its Adam -> grad_hook(flat gradients) -> optimizer.step()) with an identity
    hook against the single-process form (Adam inside the finish kernel): same bits (pc_adam_update is one definition)."""
    from types import SimpleNamespace
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader, generate_scaled_bpg
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=0.0, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3,
                          NUM_TYPES=40, DEVICE="cuda")
    bpg = generate_scaled_bpg(2000, 40, seed=3)
    B, seen = 256, []

    def make(hook):
        torch.manual_seed(5)
        m = PCompanion(cfg, bpg.cuda("cuda")["features"]).to("cuda").train()
        o = FusedAdam(m, lr=1e-2)
        g = GraphedJointStep(m, o, B, warmup=1, mode="direct", grad_hook=hook)
        ld = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), B, shuffle=True, seed=4, device="cuda",
                                      out=g.static, deferred=True)
        return m, o, g, ld

    m_a, o_a, g_a, ld_a = make(None)
    m_b, o_b, g_b, ld_b = make(lambda gflat: seen.append(gflat.numel()))
    n = 0
    for ba, bb in zip(ld_a, ld_b):
        if ba["query_idx"].numel() != B:
            continue
        la, _ = g_a(ba)
        lb, _ = g_b(bb)
        assert torch.equal(la, lb), (n, la, lb)
        n += 1
        if n == 5:
            break
    assert len(seen) == 5 and int(o_a.step_count) == int(o_b.step_count) == 5
    for (k, pa), (_, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert torch.equal(pa, pb), k
    with pytest.raises(ValueError):
        g_b.run_epoch(ld_b)


def test_train_model_dense_host_batches_prefetched(golden):
    """train_model (product2vec.py:113-170) over DENSE HOST batches in the reference's collate format: the loop
    prefetches batch i+1 to the device on a side stream while batch i trains (data.prefetch_to_device) and runs the
    fused dense step; the parameters must equal those of the plain loop (copy, dense_loss, backward, step)."""
    from types import SimpleNamespace as NS
    from p_companion_amd.data import IntBPG, prefetch_to_device
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    feats = torch.from_numpy(bpg.features)
    gen = torch.Generator().manual_seed(4)
    batches = []
    for _ in range(5):
        ids = torch.randint(0, 1000, (48, 7 + 6), generator=gen)
        nb = feats[ids[:, 7:]] * (torch.rand(48, 6, 1, generator=gen) > 0.25)            # zero rows = collate padding
        batches.append({"anchor": feats[ids[:, 0]], "positive": feats[ids[:, 1]], "negative": feats[ids[:, 2:7]],
                        "anchor_neighbors": nb, "anchor_ids": ["x"] * 48})
    class Loader:
        dataset = NS(bpg=bpg)

        def __iter__(self):
            return iter(batches)

    c = cfg(None)
    torch.manual_seed(2)
    m1 = Product2Vec(c).to(c.DEVICE)
    torch.manual_seed(2)
    m2 = Product2Vec(c).to(c.DEVICE)
    o1, o2 = FusedAdam(m1, lr=1e-2), FusedAdam(m2, lr=1e-2)
    emb = m1.train_model(Loader(), o1, num_epochs=1)
    assert len(emb) == 1000
    m2.train()
    for b in batches:
        loss = m2.dense_loss({k: v.cuda() for k, v in b.items() if isinstance(v, torch.Tensor)})
        o2.zero_grad()
        loss.backward()
        o2.step()
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p1, p2, rtol=0, atol=1e-6), k
    got = [b["anchor"].data_ptr() for b in prefetch_to_device(batches, "cuda")]             # order kept, all delivered
    assert len(got) == 5
