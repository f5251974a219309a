"""The device-side catalogue generator (csrc/generator.hip; SyntheticDataGenerator, src/data/synthetic_data.py:11-153,
restated per source node) against the host restatement data.generate_scaled_bpg, and BASELINE configs[3] at its REAL
catalogue size on one GPU: 10 M products, the row-sharded lookup chain bit-equal to the replicated table.  Needs an MI355X."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cfg(dim=128):
    return SimpleNamespace(PRODUCT_EMB_DIM=dim, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                           MARGIN=1.0, DEVICE=torch.device("cuda"))


@pytest.mark.parametrize("n", [1, 5, 4095, 4096, 4097, 8192, 1_000_003])
def test_exclusive_scan(n):
    from p_companion_amd import ops
    rng = np.random.default_rng(n)
    c = rng.integers(0, 40, n).astype(np.int32)
    out, total = ops.exclusive_scan_i32(torch.from_numpy(c).cuda())
    ref = np.concatenate([[0], np.cumsum(c.astype(np.int64))])
    assert total == int(ref[-1])
    assert np.array_equal(out.cpu().numpy().astype(np.int64), ref)


def test_device_generator_matches_the_host_generators_statistics():
    """P = 100 k, the configs[1] catalogue: the structural invariants of the generated graph, and its degree / pair-count
    / category / feature statistics against data.generate_scaled_bpg (two independent draws of the same distributions)."""
    from p_companion_amd.data import generate_device_bpg, generate_scaled_bpg
    P, T = 100_000, 100
    d = generate_device_bpg(P, T, seed=0).to_host()
    h = generate_scaled_bpg(P, T, seed=0)
    per_cat = T // 5
    # ---- invariants
    deg = np.diff(d.cv_rowptr)
    assert d.cv_rowptr[0] == 0 and d.cv_rowptr[-1] == len(d.cv_col) and deg.min() >= 0 and deg.max() <= 32
    src = np.repeat(np.arange(P), deg)
    assert d.cv_col.min() >= 0 and d.cv_col.max() < P and not np.any(d.cv_col == src)           # flag bits cleared, no self loops
    key = src.astype(np.int64) * P + d.cv_col
    assert len(np.unique(key)) == len(key)                                                      # targets distinct per row
    sp = d.similarity_pairs
    assert np.isin(sp[:, 0].astype(np.int64) * P + sp[:, 1], key).all()                         # similarity is a subset of co-view
    assert np.array_equal(np.repeat(np.arange(P), np.diff(d.sim_rowptr)), sp[:, 0]) and np.array_equal(d.sim_col, sp[:, 1])
    cp = d.complementary_pairs
    ckey = cp[:, 0].astype(np.int64) * P + cp[:, 1]
    assert not np.isin(ckey, key).any() and len(np.unique(ckey)) == len(ckey) and not np.any(cp[:, 0] == cp[:, 1])
    assert np.array_equal(d.category, np.minimum(d.type_idx // per_cat, 4))
    # ---- statistics, device draw vs host draw
    hdeg = np.diff(h.cv_rowptr)
    assert abs(deg.mean() - hdeg.mean()) < 0.02 * hdeg.mean()
    hist_d = np.bincount(deg, minlength=33) / P
    hist_h = np.bincount(hdeg, minlength=33) / P
    assert np.abs(np.cumsum(hist_d) - np.cumsum(hist_h)).max() < 0.01                           # degree distribution
    same = lambda g, s, t: float((g.category[s] == g.category[t]).mean())
    hsrc = np.repeat(np.arange(P), hdeg)
    assert abs(same(d, src, d.cv_col) - same(h, hsrc, h.cv_col)) < 0.005                        # 1.5x same-category preference
    assert abs(same(d, src, d.cv_col) - 0.2 / (0.2 + 0.8 * 2 / 3)) < 0.01
    assert abs(len(sp) / P - len(h.similarity_pairs) / P) < 0.03 * len(h.similarity_pairs) / P
    assert abs(len(cp) / P - len(h.complementary_pairs) / P) < 0.03 * len(h.complementary_pairs) / P
    assert abs(same(d, cp[:, 0], cp[:, 1]) - same(h, h.complementary_pairs[:, 0], h.complementary_pairs[:, 1])) < 0.005
    assert np.abs(np.bincount(d.type_idx, minlength=T) / P - 1.0 / T).max() < 0.002             # types uniform
    # features: N(0,1) everywhere, + 1.0 on the category's 20-dim block
    f = d.features
    blk = np.zeros_like(f, dtype=bool)
    blk[np.arange(P)[:, None], d.category[:, None] * 20 + np.arange(20)[None, :]] = True
    assert abs(f[blk].mean() - 1.0) < 0.005 and abs(f[~blk].mean()) < 0.002
    assert abs(f[blk].std() - 1.0) < 0.005 and abs(f[~blk].std() - 1.0) < 0.002
    # tails: Box-Muller over the open interval reaches |x| > 4 at the normal rate
    assert 2e-5 < float((np.abs(f[~blk]) > 4).mean()) < 1.2e-4


def test_device_generator_is_a_pure_function_of_seed_and_product():
    """Same seed -> same bits; another seed -> another catalogue; a rank's shard of the feature table (rows rank::world) is
    exactly those rows of the whole table; D = 256 rows carry the same category block."""
    from p_companion_amd.data import generate_device_bpg
    a = generate_device_bpg(50_000, 100, seed=7)
    b = generate_device_bpg(50_000, 100, seed=7)
    for k, v in a.arrays.items():
        if torch.is_tensor(v):
            assert torch.equal(v, b.arrays[k]), k
    c = generate_device_bpg(50_000, 100, seed=8)
    assert not torch.equal(a.arrays["cv_col"][:1000], c.arrays["cv_col"][:1000])
    for rank in (0, 3):
        s = generate_device_bpg(50_000, 100, seed=7, rank=rank, world=4, with_complementary=False)
        assert torch.equal(s.arrays["features"], a.arrays["features"][rank::4])
        assert torch.equal(s.arrays["cv_col"], a.arrays["cv_col"])                              # the graph is replicated
    w = generate_device_bpg(20_000, 100, seed=7, dim=256, with_complementary=False)
    f, t = w.arrays["features"].cpu().numpy(), w.arrays["type_idx"].cpu().numpy()
    cat = np.minimum(t // 20, 4)
    blk = np.zeros_like(f, dtype=bool)
    blk[np.arange(len(f))[:, None], cat[:, None] * 20 + np.arange(20)[None, :]] = True
    assert f.shape == (20_000, 256) and abs(f[blk].mean() - 1.0) < 0.01 and abs(f[~blk].mean()) < 0.003


def test_loader_and_fused_step_run_over_a_device_graph():
    """The throughput loader + fused step over a DeviceBPG equal the same loader over its host copy (IntBPG) bit for bit:
    nothing in the device path depends on host arrays."""
    from p_companion_amd.data import SimilarityIndexLoader, generate_device_bpg
    from p_companion_amd.product2vec import Product2Vec
    d = generate_device_bpg(30_000, 100, seed=1)
    h = d.to_host()
    ld_d = SimilarityIndexLoader(d, 1024, seed=3, drop_last=True)
    ld_h = SimilarityIndexLoader(h, 1024, seed=3, drop_last=True)
    assert len(ld_d) == len(ld_h)
    torch.manual_seed(0)
    m_d, m_h = Product2Vec(cfg()).cuda().train(), Product2Vec(cfg()).cuda().train()
    m_h.load_state_dict(m_d.state_dict())
    table = d.cuda()["features"]
    for i, (bd, bh) in enumerate(zip(ld_d, ld_h)):
        for k in ("anchor_idx", "positive_idx", "negative_idx"):
            assert torch.equal(bd[k], bh[k]), k
        ld_, lh_ = m_d.train_step_indexed(table, bd), m_h.train_step_indexed(h.cuda()["features"], bh)
        assert torch.equal(ld_, lh_)
        assert torch.equal(m_d.flatten_parameters()[1], m_h.flatten_parameters()[1])
        if i == 2:
            break


def test_config3_catalogue_10M_sharded_chain_equals_replicated():
    """BASELINE configs[3] at its real size on one GPU: 10 M products x 128 (5.1 GB of features, ~160 M co-view edges)
    generated in HBM; the row-sharded lookup chain (loader -> pc_shard_bucket -> HIP gather -> fused step over the gathered
    rows, world 1) against the same batches over the table itself: loss, gradients and BatchNorm statistics bit for bit
    over three steps."""
    from p_companion_amd import distributed as pdist
    from p_companion_amd.data import SimilarityIndexLoader, generate_device_bpg
    from p_companion_amd.product2vec import Product2Vec
    P, B = 10_000_000, 4096
    bpg = generate_device_bpg(P, 100, seed=0, with_complementary=False)
    assert bpg.arrays["features"].shape == (P, 128)
    e = int(bpg.arrays["cv_col"].numel())
    assert 15.0 < e / P < 16.5 and bpg.n_similarity_pairs > 2.5 * P
    table = bpg.cuda()["features"]
    sharded = pdist.ShardedFeatureTable(table, bpg.num_products, 0, 1)
    ld_s = SimilarityIndexLoader(bpg, B, seed=3, drop_last=True, sharded=sharded)
    ld_r = SimilarityIndexLoader(bpg, B, seed=3, drop_last=True)
    assert not ld_s.unique                                        # the compact layout: no O(P) scan per batch at this size
    torch.manual_seed(0)
    m_s, m_r = Product2Vec(cfg()).cuda().train(), Product2Vec(cfg()).cuda().train()
    m_r.load_state_dict(m_s.state_dict())
    it_s, it_r = iter(ld_s), iter(ld_r)
    for _ in range(3):
        bs, br = next(it_s), next(it_r)
        assert "table" in bs and bs["table"].shape[0] == sharded.capacity
        ls, lr = m_s.train_step_indexed(bs["table"], bs), m_r.train_step_indexed(table, br)
        assert torch.isfinite(ls).all() and torch.equal(ls, lr)
        assert torch.equal(m_s.flatten_parameters()[1], m_r.flatten_parameters()[1])
    assert torch.equal(m_s.ffn[1].running_var, m_r.ffn[1].running_var)
    sharded.raise_if_overflowed()
    ld_s.check_errors()
