"""Size-independent properties at BASELINE.json's full sizes (configs[1]: 100 k products, 100 types, B = 4096,
neighbour lists padded to 32; configs[0]'s T = 34800 for the joint similarities).  The oracle finishes these
sizes in minutes, not seconds, so here the HIP path is checked against itself through identities the reference
semantics imply:
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
