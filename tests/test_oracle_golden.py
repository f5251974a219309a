"""Pins the oracle (CPU restatement) to vectors produced by running the reference itself
(tests/golden/make_golden.py).  CPU only."""
import random

import numpy as np
import pytest
import torch

from oracle import data_oracle, joint_oracle, p2v_oracle
from oracle.mt import Random


def t(x):
    return torch.from_numpy(np.asarray(x))


def assert_adam_close(actual, desired, steps, tight, lr=1e-3):
    """Adam's first steps move a parameter by ~lr*sign(g): an element whose gradient is
    rounding noise (|g| ~ 1e-8) may legitimately differ by up to a full lr-step, so bound
    the worst case by steps*lr and require all but 0.1% of elements to agree tightly."""
    d = np.abs(np.asarray(actual) - np.asarray(desired))
    assert d.max() <= 1.05 * lr * steps, d.max()
    assert (d <= tight).mean() >= 0.999, (d > tight).mean()


def load_state(g, prefix, keys=None):
    st = {}
    for k in g.files:
        if k.startswith(prefix):
            st[k[len(prefix):]] = t(g[k]).clone()
    return st


# ------------------------------------------------------------------ G1: CPython random
@pytest.mark.parametrize("seed", [0, 1, 12345, 2**40 + 7])
def test_mt19937_stream(golden, seed):
    g = golden("g1_mt19937.npz")
    r = Random(seed)
    tag = f"s{seed}_"
    assert [r.getrandbits(10) for _ in range(64)] == g[tag + "getrandbits10"].tolist()
    assert [r.getrandbits(32) for _ in range(16)] == g[tag + "getrandbits32"].tolist()
    assert [r.choice_index(1000) for _ in range(64)] == g[tag + "choice1000"].tolist()
    assert [r.random() for _ in range(16)] == g[tag + "random"].tolist()
    assert r.shuffle_perm(32).tolist() == g[tag + "shuffle32"].tolist()


def test_mt19937_vs_live_cpython():
    """CPython's own `random` is on every box: cross-check beyond the fixture."""
    for seed in (0, 3, 2**33 + 5):
        random.seed(seed)
        r = Random(seed)
        for n in (1, 2, 7, 1000, 100000, 2**31 + 11, 2**40 + 3):
            assert [r.randbelow(n) for _ in range(50)] == [random.randrange(n) for _ in range(50)]
        assert [r.getrandbits(53) for _ in range(20)] == [random.getrandbits(53) for _ in range(20)]


# ------------------------------------------------------------------ G3: negative sampler
@pytest.mark.parametrize("seed", [0, 7])
def test_negative_sampler(golden, seed):
    ints = golden("g2_bpg1000.npz")
    g = golden("g3_negatives.npz")
    pairs = ints["similarity_pairs"]
    got = Random(seed).negative_samples(1000, pairs, pairs[:256, 0], 5)
    assert np.array_equal(got, g[f"s{seed}_negatives"])
    # rejection rules (data_loader.py:33-38)
    for a, row in zip(pairs[:256, 0], got):
        pos = set(pairs[pairs[:, 0] == a, 1].tolist())
        assert len(set(row.tolist())) == 5 and a not in row and not (set(row.tolist()) & pos)


# ------------------------------------------------------------------ G7: collate padding
def test_collate_padding(golden):
    g = golden("g7_collate.npz")
    degs = g["degrees"]
    starts = np.concatenate([[0], np.cumsum(degs)])
    lists = [np.arange(starts[i], starts[i + 1]) for i in range(len(degs))]
    idx = data_oracle.collate_neighbors(lists)
    rows = np.concatenate([g["neighbor_rows"], np.zeros((1, 128), np.float32)])
    assert np.array_equal(rows[idx], g["anchor_neighbors"])


# ------------------------------------------------------------------ G4: Product2Vec step
def _p2v_check(g, batch, after3, after1_keys_src):
    st = load_state(g, "init.")
    mom = p2v_oracle.new_moments(st)
    losses = []
    for step in range(1, 4):
        r = p2v_oracle.train_step(st, batch, 1.0, mom, step)
        losses.append(float(r["loss"]))
        if step == 1:
            first = r
            bn1 = {k: st[k].clone() for k in st if "running" in k or "num_batches" in k}
            after1 = {k: st[k].clone() for k in st}
    np.testing.assert_allclose(losses, g["losses"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(first["anchor_emb"], g["anchor_emb"], atol=2e-6)
    np.testing.assert_allclose(first["positive_emb"], g["positive_emb"], atol=2e-6)
    np.testing.assert_allclose(first["pos_distance"], g["pos_distance"], atol=5e-6)
    np.testing.assert_allclose(first["neg_distance"], g["neg_distance"], atol=5e-6)
    for k in p2v_oracle.TRAINABLE:
        ref = g["grad." + k]
        if k == "ffn.0.bias":
            # d(loss)/d(b0) is analytically 0 (BatchNorm removes the shift): both sides hold
            # rounding noise only
            assert np.abs(ref).max() < 1e-7 and first["grads"][k].abs().max() < 1e-7
            continue
        np.testing.assert_allclose(first["grads"][k], ref, atol=1e-6 + 1e-4 * np.abs(ref).max())
    for k, v in bn1.items():
        np.testing.assert_allclose(v, g["bn_after1." + k], rtol=1e-6, atol=1e-6)
    assert int(bn1["ffn.1.num_batches_tracked"]) == 4       # anchor, neighbours, positive, negative
    for k in after1_keys_src.files:
        if k.startswith("after1.") and k[7:] in p2v_oracle.TRAINABLE and k[7:] != "ffn.0.bias":
            assert_adam_close(after1[k[7:]], after1_keys_src[k], 1, 2e-6)
    for k in p2v_oracle.TRAINABLE:
        if k == "ffn.0.bias":
            continue           # Adam amplifies the rounding-noise gradient: not comparable
        assert_adam_close(st[k], after3["after3." + k], 3, 2e-5)
    return first


def test_p2v_step_tiny(golden):
    g = golden("g4_p2v_tiny.npz")
    batch = {k[6:]: t(g[k]) for k in g.files if k.startswith("batch.")}
    first = _p2v_check(g, batch, g, g)
    np.testing.assert_allclose(first["negative_emb"], g["negative_emb"], atol=2e-6)


def test_p2v_step_b256(golden):
    g = golden("g4_p2v_b256.npz")
    ints = golden("g2_bpg1000.npz")
    # the oracle's loader semantics rebuild the reference's collated batch in index form
    nb = data_oracle.collate_neighbors(
        [data_oracle.neighbors(ints["cv_rowptr"], ints["cv_col"], a) for a in g["anchor_idx"]])
    assert np.array_equal(nb, g["neighbor_idx"])
    assert np.array_equal(ints["similarity_pairs"][:256, 0], g["anchor_idx"])
    assert np.array_equal(ints["similarity_pairs"][:256, 1], g["positive_idx"])
    assert np.array_equal(Random(3).negative_samples(1000, ints["similarity_pairs"], g["anchor_idx"]),
                          g["negative_idx"])
    batch = p2v_oracle.gather_batch(t(ints["features"]), g["anchor_idx"], g["positive_idx"],
                                    g["negative_idx"], g["neighbor_idx"])
    first = _p2v_check(g, batch, golden("g4_p2v_b256_params3.npz"), golden("g4_p2v_b256_params3.npz"))
    np.testing.assert_allclose(first["negative_emb"][:32], g["negative_emb_first32"], atol=2e-6)


# ------------------------------------------------------------------ G5: eval export
def test_generate_all_embeddings(golden):
    g = golden("g5_p2v_eval.npz")
    st = load_state(g, "init.")
    out = p2v_oracle.generate_all_embeddings(t(g["features"]), g["cv_rowptr"], g["cv_col"], st)
    np.testing.assert_allclose(out, g["embeddings"], atol=3e-6)
    deg = np.diff(g["cv_rowptr"])
    assert (deg == 0).any() and (deg > 0).any()


# ------------------------------------------------------------------ G6: joint step
@pytest.mark.parametrize("T", [100, 300])
def test_joint_step(golden, T):
    g = golden(f"g6_joint_t{T}.npz")
    st = load_state(g, "init.")
    batch = {k[6:]: t(g[k]) for k in g.files if k.startswith("batch.")}
    mom = joint_oracle.new_moments(st)
    losses = []
    for step in range(1, 4):
        r = joint_oracle.train_step(st, batch, mom, step)
        losses.append(float(r["loss"]))
        if step == 1:
            first = r
        if step in (1, 3):
            tag = f"after{step}."
            for k in joint_oracle.TRAINABLE:
                np.testing.assert_allclose(st[k], g[tag + k], atol=2e-6)
                np.testing.assert_allclose(mom[k][0], g[f"{tag}exp_avg.{k}"], atol=1e-7)
                np.testing.assert_allclose(mom[k][1], g[f"{tag}exp_avg_sq.{k}"], atol=1e-9)
    np.testing.assert_allclose(losses, g["losses"], atol=2e-6)
    assert np.array_equal(first["out"]["complementary_types"].numpy(), g["complementary_types"])
    np.testing.assert_allclose(first["out"]["projected_embeddings"], g["projected_embeddings"], atol=2e-6)
    sims = first["out"]["type_similarities"].numpy()
    np.testing.assert_allclose(sims if T <= 100 else sims[:, :128], g["type_similarities"], atol=2e-6)
    assert abs(float(first["type_loss"]) - float(g["type_loss"])) < 2e-6
    assert abs(float(first["item_loss"]) - float(g["item_loss"])) < 2e-6
    for k in joint_oracle.TRAINABLE:
        np.testing.assert_allclose(first["grads"][k], g["grad." + k], atol=1e-7)
    # "untouched rows still move": a type row with zero gradient in step 3 has changed by step 3
    gq = g["grad.query_type_embeddings.weight"]
    touched = np.abs(gq).sum(1) > 0
    assert touched.sum() < T


# ------------------------------------------------------------------ G8: metrics
def test_metrics(golden):
    g = golden("g8_metrics.npz")
    st = load_state(g, "init.")
    res = joint_oracle.evaluate_batch(st, t(g["query_idx"]), t(g["query_types"]), t(g["positive_items"]),
                                      t(g["target_features"]))
    for name, val in zip(g["metric_names"], g["metric_values"]):
        assert abs(res[str(name)] - float(val)) < 1e-6, name
