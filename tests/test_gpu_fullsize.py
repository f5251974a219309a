"""BASELINE.json's full sizes (configs[1]: 100 k products, 100 types, B = 4096, neighbour lists padded to the batch max <= 32;
configs[2]: the same catalogue through the joint step at T = 100 and at config.py:27's NUM_TYPES = 34800), through the entry
points bench.py times.

Against the ORACLE (it takes ~3 s per Product2Vec step and 0.2-1.1 s per joint step at these sizes on the GPU box's host cores):
  * configs[1]: one batch of the loader bench.py uses (device sampler, unique-row layout) through pc_p2v_train_step_unique +
    pc_adam_step vs p2v_oracle.train_step on the dense batch the reference's collate_fn would have built: loss <= 1e-4, all twelve
    gradients, BatchNorm running statistics, parameters after the Adam step;
  * configs[2]: the 100 k catalogue through ComplementaryIndexLoader(deferred=True) + GraphedJointStep -- one step through
    pc_joint_fused_step_pairs (top-k, the three losses, ten gradients) and three steps through run_epoch (pc_joint_train_epoch:
    per-step losses, parameters after three Adam steps) vs joint_oracle.train_step on the materialised batches, at T = 100 and at
    T = 34800 without and with the explicit hidden-layer dropout mask.
And through identities the reference semantics imply (no oracle needed):
  * the three row layouts of the step (every slot a row / padding once / every distinct product once) give the
    same loss, embeddings and gradients (identical rows are identical at every layer);
  * permuting the samples of a batch permutes nothing observable (BatchNorm, the mean loss and every gradient are
    symmetric in the samples);
  * the step is bitwise reproducible run to run (no float atomics anywhere on the path);
  * top-k of the similarities == torch.topk on the same matrix (index-exact), type loss / item loss additivity;
  * the FUSED step's selection at B = 4096, T = 34800 (sub-chunk maxima + exact refinement: the [B,T] matrix never exists) against
    the module path's dense matrix, with and without dropout.
Needs an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fresh(ops, seed=5):
    sizes = [int(np.prod(s)) for s in ops.P2V_SHAPES]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    g = torch.Generator(device="cpu").manual_seed(seed)
    flat = (torch.randn(int(offs[-1]), generator=g) * 0.05).cuda()
    gflat = torch.zeros_like(flat)
    params = {k: flat[offs[i]:offs[i + 1]].view(s) for i, (k, s) in enumerate(zip(ops.P2V_KEYS, ops.P2V_SHAPES))}
    grads = {k: gflat[offs[i]:offs[i + 1]].view(s) for i, (k, s) in enumerate(zip(ops.P2V_KEYS, ops.P2V_SHAPES))}
    params["ffn.1.weight"].fill_(1.0)
    params["ffn.1.running_mean"] = torch.zeros(256, device="cuda")
    params["ffn.1.running_var"] = torch.ones(256, device="cuda")
    params["ffn.1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device="cuda")
    return params, grads, gflat


@pytest.fixture(scope="module")
def full_batch():
    from p_companion_amd.data import generate_scaled_bpg, SimilarityIndexLoader
    bpg = generate_scaled_bpg(100_000, 100, seed=0)
    loader = SimilarityIndexLoader(bpg, 4096, seed=1, drop_last=True, compact=False, prefetch=False)
    batch = next(iter(loader))
    return bpg, bpg.cuda()["features"], batch


def test_row_layouts_agree_at_full_size(full_batch):
    from p_companion_amd import ops
    bpg, table, batch = full_batch
    nb = batch["neighbor_idx"]
    assert nb.shape == (4096, 32)
    res = {}
    for name, layout in (("dense", nb), ("compact", ops.compact_neighbors(nb)), ("unique", ops.unique_neighbors(nb))):
        p, g, gf = _fresh(ops)
        out = ops.p2v_train_step(p, g, table, batch["anchor_idx"], batch["positive_idx"], batch["negative_idx"], layout,
                                 1.0, want_emb=True)
        res[name] = (float(out["loss"]), out["anchor_emb"].clone(), gf.clone(), p["ffn.1.running_var"].clone())
    uq = ops.unique_neighbors(nb)
    assert uq["n_unique"] < 0.75 * int((nb >= 0).sum())          # the layouts really differ in row count
    l0, e0, g0, v0 = res["dense"]
    tol = 2e-6 + 2e-4 * float(g0.abs().max())
    for name in ("compact", "unique"):
        l, e, g, v = res[name]
        assert abs(l - l0) < 2e-6, (name, l, l0)
        assert float((e - e0).abs().max()) < 2e-5
        assert float((g - g0).abs().max()) < tol, (name, float((g - g0).abs().max()), tol)
        assert torch.allclose(v, v0, atol=1e-6)


def test_step_is_bitwise_reproducible_and_sample_symmetric(full_batch):
    from p_companion_amd import ops
    bpg, table, batch = full_batch
    nb = batch["neighbor_idx"]

    def run(order):
        b = {k: batch[k][order].contiguous() for k in ("anchor_idx", "positive_idx", "negative_idx")}
        p, g, gf = _fresh(ops)
        out = ops.p2v_train_step(p, g, table, b["anchor_idx"], b["positive_idx"], b["negative_idx"],
                                 ops.unique_neighbors(nb[order].contiguous()), 1.0)
        return float(out["loss"]), gf.clone(), out["d_pos"].clone()

    ident = torch.arange(4096, device="cuda")
    l1, g1, d1 = run(ident)
    l2, g2, d2 = run(ident)
    assert l1 == l2 and torch.equal(g1, g2) and torch.equal(d1, d2)          # bit for bit
    perm = torch.randperm(4096, generator=torch.Generator().manual_seed(9)).cuda()
    l3, g3, d3 = run(perm)
    assert abs(l3 - l1) < 2e-6
    assert float((g3 - g1).abs().max()) < 2e-6 + 2e-4 * float(g1.abs().max())
    assert float((d3 - d1[perm]).abs().max()) < 2e-5                          # per-sample distances follow the samples


def test_joint_similarities_topk_and_loss_additivity_at_T34800():
    """configs[0]'s table size (NUM_TYPES = 34800, config.py:27) at B = 4096: top-3 indices are exactly
    torch.topk's on the same similarity matrix; the joint loss is ALPHA * item + (1 - ALPHA) * type of the two
    separately computed hinges (p_companion.py:79-93)."""
    from types import SimpleNamespace
    from p_companion_amd.p_companion import PCompanion
    T, B = 34800, 4096
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                          MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=T, DEVICE=torch.device("cuda"),
                          LEARNING_RATE=1e-3)
    g = torch.Generator().manual_seed(2)
    torch.manual_seed(2)
    model = PCompanion(cfg, torch.randn(5000, 128, generator=g)).cuda().eval()
    batch = {"query_idx": torch.randint(0, 5000, (B,), generator=g, dtype=torch.int32).cuda(),
             "query_types": torch.randint(0, T, (B,), generator=g).cuda(),
             "positive_types": torch.randint(0, T, (B, 1), generator=g).cuda(),
             "negative_types": torch.randint(0, T, (B, 1), generator=g).cuda(),
             "positive_items": torch.randn(B, 128, generator=g).cuda(), "negative_items": torch.randn(B, 128, generator=g).cuda()}
    with torch.no_grad():
        out = model(batch)
        sims = out["type_similarities"]
        assert sims.shape == (B, T)
        ref_v, ref_i = torch.topk(sims, 3, dim=1)
        got = out["complementary_types"]
        same = (got == ref_i)
        # positions that differ must be exact ties in value
        assert bool(same.all()) or torch.equal(torch.gather(sims, 1, got)[~same], ref_v[~same])
        total = model.compute_loss(batch, out)
        tl = model._compute_type_loss(sims, batch["positive_types"].squeeze(-1), batch["negative_types"].squeeze(-1))
        il = model._compute_item_loss(out["projected_embeddings"], batch["positive_items"], batch["negative_items"])
        assert abs(float(total) - (0.8 * float(il) + 0.2 * float(tl))) < 1e-5


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_fused_selection_at_full_size_against_the_dense_similarity_matrix(p):
    """B = 4096 rows over NUM_TYPES = 34800 (config.py:27), with and without config.py:12's DROPOUT: the FUSED step never forms the
    [B,T] similarity matrix -- it keeps the maximum of every 64-type sub-chunk and re-forms the sub-chunks that can hold a row's K
    best -- while the module path (p_companion.py:57-65 op by op) writes all 142 M similarities and selects with pc_topk_rows.
    Same parameters, same dropout mask (seed and offset set alike): the fused selection must be the dense one up to the rounding of
    two fp32 summation orders -- its types' dense similarities within 2e-6 (relative to the row's scale) of the dense K best, in
    descending order, K different types per row, and index-exact on all but a handful of rows."""
    from types import SimpleNamespace
    from p_companion_amd.p_companion import PCompanion
    T, B, K = 34800, 4096, 3
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=p,
                          MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=K, NUM_TYPES=T, DEVICE=torch.device("cuda"),
                          LEARNING_RATE=1e-3)
    g = torch.Generator().manual_seed(7)
    torch.manual_seed(7)
    model = PCompanion(cfg, torch.randn(5000, 128, generator=g)).cuda().train()
    batch = {"query_idx": torch.randint(0, 5000, (B,), generator=g, dtype=torch.int32).cuda(),
             "query_types": torch.randint(0, T, (B,), generator=g).cuda(),            # ~3 900 distinct query types
             "positive_types": torch.randint(0, T, (B, 1), generator=g).cuda(),
             "negative_types": torch.randint(0, T, (B, 1), generator=g).cuda(),
             "positive_items": torch.randn(B, 128, generator=g).cuda(), "negative_items": torch.randn(B, 128, generator=g).cuda()}
    tt = model.type_transition
    tt._dropout_seed, tt._dropout_step = 4242, 0
    with torch.no_grad():
        sims = model(batch)["type_similarities"].double()                             # the dense matrix, module path
    tt._dropout_step = 0
    _, got = model.train_step(batch)                                                  # the fused step, same mask
    got = got.long()
    assert got.shape == (B, K) and int(got.min()) >= 0 and int(got.max()) < T
    assert bool((got[:, 0] != got[:, 1]).all() and (got[:, 1] != got[:, 2]).all() and (got[:, 0] != got[:, 2]).all())
    best_v, best_i = sims.topk(K, dim=1)
    mine = sims.gather(1, got)
    tol = 2e-6 * sims.abs().amax(1, keepdim=True)
    assert bool((mine >= best_v - tol).all()), float((best_v - mine).max())
    assert bool((mine[:, :-1] >= mine[:, 1:] - tol).all())
    assert int((got != best_i).any(1).sum()) <= 4                                     # (rounding-level ties only)


# ------------------------------------------------------------------------------------------------------------------------
# ORACLE parity at full size, through the entry points bench.py times
# ------------------------------------------------------------------------------------------------------------------------
def _adam_close(actual, desired, steps, tight, lr=1e-3):
    """Parameters after `steps` Adam steps: nothing further from the oracle than the steps can move it, and all but a
    handful of elements (gradients at rounding level, where the first Adam steps are sign-like) within `tight`."""
    d = (actual - torch.as_tensor(desired)).abs()
    assert float(d.max()) <= 1.05 * lr * steps, float(d.max())
    assert float((d <= tight).float().mean()) >= 0.999, float((d <= tight).float().mean())


@pytest.fixture(scope="module")
def full_bpg():
    from p_companion_amd.data import generate_scaled_bpg
    return generate_scaled_bpg(100_000, 100, seed=0)               # bench.py's catalogue (configs[1] and [2])


def test_config1_unique_layout_step_and_adam_against_the_oracle(full_bpg):
    """configs[1] as bench.py runs it (product2vec.py:126-159 per step): a batch of the throughput loader -- device Philox
    negatives, every distinct neighbour product once -- through Product2Vec.train_step_indexed (pc_p2v_train_step_unique)
    and FusedAdam.step (pc_adam_step), against the oracle on the dense [B,N,128] batch collate_fn would have built."""
    from types import SimpleNamespace
    from oracle import p2v_oracle
    from p_companion_amd.data import SimilarityIndexLoader
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    bpg = full_bpg
    B = 4096
    loader = SimilarityIndexLoader(bpg, B, shuffle=True, sampler="philox", seed=1, drop_last=True, device="cuda",
                                   reuse_buffers=True)
    assert loader.unique
    batch = next(iter(loader))
    nbc = batch["neighbor_compact"]
    assert "weight" in nbc                                         # the unique-row layout (multiplicities)
    # the layout says which product sits in every (sample, slot): exactly the anchor's co-view out-neighbours (bpg.py:24-38),
    # the rest of the row padding (-1 = collate_fn's zero row, data_loader.py:186-198)
    nb_dense = nbc["nb_rows"].cpu().numpy()[nbc["slot_row"].cpu().numpy()]
    anchors = batch["anchor_idx"].cpu().numpy()
    assert nb_dense.shape[0] == B and nb_dense.shape[1] <= 32 and nb_dense.shape[1] == batch["n_pad"]
    for i in range(0, B, 97):
        lo, hi = bpg.cv_rowptr[anchors[i]], bpg.cv_rowptr[anchors[i] + 1]
        row = nb_dense[i]
        assert sorted(row[row >= 0].tolist()) == sorted(bpg.cv_col[lo:hi].tolist()) and hi - lo >= 1
    neg = batch["negative_idx"].cpu().numpy()
    assert neg.shape == (B, 5) and bool((neg != anchors[:, None]).all())
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                          MARGIN=1.0, BATCH_SIZE=B, LEARNING_RATE=1e-3, DEVICE=torch.device("cuda"))
    torch.manual_seed(0)
    model = Product2Vec(cfg).cuda().train()
    opt = FusedAdam(model, lr=1e-3)
    st = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    table = bpg.cuda("cuda")["features"]
    loss = float(model.train_step_indexed(table, batch))
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}
    opt.step()
    after = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    dense = p2v_oracle.gather_batch(torch.from_numpy(bpg.features), anchors, batch["positive_idx"].cpu().numpy(), neg, nb_dense)
    assert dense["anchor_neighbors"].shape == (B, nb_dense.shape[1], 128)
    ref = p2v_oracle.train_step(st, dense, 1.0, p2v_oracle.new_moments(st), 1)
    assert abs(loss - float(ref["loss"])) < 1e-4, (loss, float(ref["loss"]))          # the north_star's bound
    for k in p2v_oracle.TRAINABLE:
        r = ref["grads"][k]
        if k == "ffn.0.bias":
            assert float(grads[k].abs().max()) < 1e-6                                  # analytically zero (BatchNorm follows)
            continue
        tol = 2e-6 + 2e-4 * float(r.abs().max())
        assert float((grads[k] - r).abs().max()) < tol, (k, float((grads[k] - r).abs().max()), tol)
    for k in ("ffn.1.running_mean", "ffn.1.running_var"):
        assert torch.allclose(after[k], st[k], rtol=1e-5, atol=1e-5), k                # (the oracle updated `st` in place)
    assert int(after["ffn.1.num_batches_tracked"]) == int(st["ffn.1.num_batches_tracked"]) == 4
    for k in p2v_oracle.TRAINABLE:
        if k != "ffn.0.bias":
            _adam_close(after[k], st[k], 1, 5e-5)


def _joint_setup(bpg, T, dropout, B=4096):
    from types import SimpleNamespace
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=float(dropout), MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3,
                          NUM_TYPES=T, DEVICE=torch.device("cuda"))
    torch.manual_seed(0)
    model = PCompanion(cfg, bpg.cuda("cuda")["features"]).cuda().train()
    model.type_transition._dropout_seed, model.type_transition._dropout_step = 4242, 0
    opt = FusedAdam(model, lr=1e-3)
    step = GraphedJointStep(model, opt, B, warmup=0, mode="direct")
    ds = ComplementaryIndexDataset(bpg, "train")
    loader = ComplementaryIndexLoader(ds, B, shuffle=True, seed=0, device="cuda", out=step.static, deferred=True)
    plain = ComplementaryIndexLoader(ds, B, shuffle=True, seed=0, device="cuda")          # the same batches, built up front
    return cfg, model, opt, step, loader, plain


def _host_batches(plain, n, bpg, B=4096):
    """The first n batches of the epoch as host tensors, each checked against the label rules of data_loader.py:133-157."""
    out = []
    for b in plain:
        lab = b["label"].cpu().numpy()
        hb = {k: v.detach().cpu().clone() for k, v in b.items() if torch.is_tensor(v)}
        q, qt = hb["query_idx"].numpy(), hb["query_types"].numpy()
        assert q.shape == (B,) and np.array_equal(qt, bpg.type_idx[q])
        pt, nt = hb["positive_types"].numpy()[:, 0], hb["negative_types"].numpy()[:, 0]
        tgt_feat = hb["target_features"].numpy()
        pos, neg = hb["positive_items"].numpy(), hb["negative_items"].numpy()
        comp = lab == 1
        assert comp.any() and (~comp).any()
        assert np.array_equal(pos[comp], tgt_feat[comp]) and np.array_equal(neg[~comp], tgt_feat[~comp])
        assert bool((pt[~comp] == 0).all()) and np.array_equal(nt[comp], (pt[comp] + 1) % bpg.n_types)
        out.append(hb)
        if len(out) == n:
            break
    return out


def _hidden_mask(dropout, offset, B):
    from oracle import philox_oracle
    if dropout <= 0.0:
        return None
    return torch.from_numpy(philox_oracle.dropout_mask(4242, offset, philox_oracle.STREAM_HIDDEN, B * 32, dropout)).view(B, 32)


@pytest.mark.parametrize("T,dropout", [(100, 0.0), (34800, 0.0), (34800, 0.1)])
def test_config2_fused_pairs_step_against_the_oracle(full_bpg, T, dropout):
    """configs[2] through pc_joint_fused_step_pairs (one step of train.py:42-48, the batch built inside the step's first
    kernel) at B = 4096 over the 100 k catalogue: top-k index-exact, the three losses <= 1e-4, ten gradients."""
    from oracle import joint_oracle
    B = 4096
    cfg, model, opt, step, loader, plain = _joint_setup(full_bpg, T, dropout)
    st0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    hb = _host_batches(plain, 1, full_bpg)[0]
    model.flatten_parameters()
    opt.fused_state()
    step._prepare()                                    # (as run_epoch does: the first call then already takes the pairs form)
    batch = next(iter(loader))
    assert "_deferred" in batch
    losses, topk = step(batch)
    assert step.prepared is not None and step.prepared.calls == 1 and "_deferred" not in batch
    for k in ("query_idx", "query_types", "positive_types", "negative_types", "positive_items", "negative_items"):
        assert torch.equal(step.static[k].cpu(), hb[k].reshape(step.static[k].shape)), k     # the step built THIS batch
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.requires_grad}
    ref = joint_oracle.train_step({k: v.clone() for k, v in st0.items()}, hb, joint_oracle.new_moments(st0), 1,
                                  cfg.MARGIN, cfg.ALPHA, cfg.NUM_COMP_TYPES, hidden_mask=_hidden_mask(dropout, 0, B))
    want = ref["out"]["complementary_types"]
    got = topk.cpu().long()
    differ = (got != want).any(1)
    if bool(differ.any()):
        # only where two similarities of a row agree to fp32 rounding may the selections differ
        sims = ref["out"]["type_similarities"]
        mine, best = sims.gather(1, got), sims.gather(1, want)
        assert int(differ.sum()) <= 2 and bool(((mine - best).abs() <= 2e-6 * sims.abs().amax(1, keepdim=True))[differ].all())
    lo = losses.cpu()
    assert abs(float(lo[0]) - float(ref["loss"])) < 1e-4 and abs(float(lo[1]) - float(ref["type_loss"])) < 1e-4 \
        and abs(float(lo[2]) - float(ref["item_loss"])) < 1e-4, (lo, ref["loss"], ref["type_loss"], ref["item_loss"])
    slack = float(differ.sum()) * 4.0 / (B * cfg.NUM_COMP_TYPES)                          # (a swapped near-tie moves one row's share)
    for k in joint_oracle.TRAINABLE:
        r = ref["grads"][k]
        tol = 2e-6 + 2e-4 * float(r.abs().max()) + slack
        assert float((grads[k] - r).abs().max()) < tol, (k, float((grads[k] - r).abs().max()), tol)
    untouched = ref["grads"]["query_type_embeddings.weight"].abs().sum(1) == 0
    assert bool((grads["query_type_embeddings.weight"][untouched] == 0).all())


@pytest.mark.parametrize("T,dropout", [(100, 0.0), (34800, 0.0), (34800, 0.1)])
def test_config2_run_epoch_against_the_oracle(full_bpg, T, dropout):
    """configs[2] through GraphedJointStep.run_epoch (pc_joint_train_epoch: what bench.py's joint legs time) at B = 4096
    over the 100 k catalogue: the first three steps of the epoch vs three oracle steps on the materialised batches --
    per-step losses <= 1e-4 and every parameter after three dense Adam steps (untouched type rows move too once their
    moments are non-zero, p_companion.py:36-43 + train.py:24)."""
    from oracle import joint_oracle
    B, n = 4096, 3
    cfg, model, opt, step, loader, plain = _joint_setup(full_bpg, T, dropout)
    st = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    init = {k: v.clone() for k, v in st.items()}
    hbs = _host_batches(plain, n, full_bpg)
    got = step.run_epoch(loader, drop_last=True, max_steps=n)
    assert got.shape == (n, 3) and int(opt.step_count) == n and loader.step == n
    mom = joint_oracle.new_moments(st)
    for i, hb in enumerate(hbs):
        ref = joint_oracle.train_step(st, hb, mom, i + 1, cfg.MARGIN, cfg.ALPHA, cfg.NUM_COMP_TYPES,
                                      hidden_mask=_hidden_mask(dropout, i, B))
        g = got[i].cpu()
        assert abs(float(g[0]) - float(ref["loss"])) < 1e-4 and abs(float(g[1]) - float(ref["type_loss"])) < 1e-4 \
            and abs(float(g[2]) - float(ref["item_loss"])) < 1e-4, (i, g, ref["loss"], ref["type_loss"], ref["item_loss"])
    after = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    for k in joint_oracle.TRAINABLE:
        _adam_close(after[k], st[k], n, 5e-5)
        assert not torch.equal(after[k], init[k]), k
    assert torch.equal(after["product_embeddings.weight"], init["product_embeddings.weight"])    # frozen (p_companion.py:26-29)
