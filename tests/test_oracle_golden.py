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


# ------------------------------------------------------------------ G6 at T = 1000 (SURVEY 8c's size)
def test_joint_step_t1000(golden):
    g = golden("g6_joint_t1000.npz")
    st = load_state(g, "init.")
    batch = {k[6:]: t(g[k]) for k in g.files if k.startswith("batch.")}
    mom = joint_oracle.new_moments(st)
    losses = []
    for step in range(1, 4):
        r = joint_oracle.train_step(st, batch, mom, step)
        losses.append(float(r["loss"]))
        if step == 1:
            assert np.array_equal(r["out"]["complementary_types"].numpy(), g["complementary_types"])
            np.testing.assert_allclose(r["out"]["type_similarities"][:, :128], g["type_similarities"], atol=2e-6)
            for k in joint_oracle.TRAINABLE:
                np.testing.assert_allclose(r["grads"][k], g["grad." + k], atol=1e-7)
    np.testing.assert_allclose(losses, g["losses"], atol=2e-6)
    for k in joint_oracle.TRAINABLE:
        np.testing.assert_allclose(st[k], g["after3." + k], atol=2e-6)
        np.testing.assert_allclose(mom[k][0], g[f"after3.exp_avg.{k}"], atol=1e-7)


# ------------------------------------------------------------------ G10: Product2Vec.train_model, two epochs
def p2v_epoch_batches(ints, g):
    """The batches DataLoader(SimilarityDataset, B, shuffle=False, collate_fn) yields (data_loader.py:45-71,171-206), in
    index form, with the negatives of the CPython stream after random.seed(seed) (:27-40)."""
    pairs = ints["similarity_pairs"]
    B, S = int(g["batch_size"]), len(pairs)
    rng = Random(int(g["seed"]))
    at = 0
    for epoch in range(int(g["epochs"])):
        for lo in range(0, S, B):
            ids = np.arange(lo, min(lo + B, S))
            neg = rng.negative_samples(1000, pairs, pairs[ids, 0], 5)
            assert np.array_equal(neg, g["negative_idx"][at:at + len(ids)])          # bit-exact negative-sample indices
            at += len(ids)
            yield data_oracle.similarity_batch(ints, ids, neg)
    assert at == len(g["negative_idx"])


def test_p2v_train_model_epochs(golden):
    """The oracle's loop body iterated over the reference's own two-epoch run of Product2Vec.train_model
    (product2vec.py:113-170): 24 steps with a ragged last batch (133 of 256), BatchNorm running statistics carried
    through 96 calls, then the eval-mode export over all 1 000 products (32 of them without out-neighbours)."""
    g = golden("g10_p2v_epochs.npz")
    ints = golden("g2_bpg1000.npz")
    st = load_state(g, "init.")
    mom = p2v_oracle.new_moments(st)
    feats = t(ints["features"])
    losses = []
    for step, b in enumerate(p2v_epoch_batches(ints, g), 1):
        batch = p2v_oracle.gather_batch(feats, b["anchor_idx"], b["positive_idx"], b["negative_idx"], b["neighbor_idx"])
        losses.append(float(p2v_oracle.train_step(st, batch, 1.0, mom, step)["loss"]))
    assert len(losses) == len(g["losses"]) == 24 and len(ints["similarity_pairs"]) % 256 == 133
    np.testing.assert_allclose(losses, g["losses"], rtol=0, atol=1e-5)
    assert int(st["ffn.1.num_batches_tracked"]) == int(g["final.ffn.1.num_batches_tracked"]) == 96
    # ffn.0.bias has an analytically zero gradient (BatchNorm removes the shift); Adam turns its rounding noise into
    # steps of up to lr, on either side differently, and the running MEAN carries that bias: bounded by steps * lr, and
    # without the bias's own drift it agrees closely
    d_mean = (st["ffn.1.running_mean"] - t(g["final.ffn.1.running_mean"])).abs()
    assert float(d_mean.max()) <= 24e-3
    drift = st["ffn.0.bias"] - t(g["final.ffn.0.bias"])
    print("running_mean: max diff %.2e, b0 drift max %.2e" % (float(d_mean.max()), float(drift.abs().max())))
    np.testing.assert_allclose(st["ffn.1.running_var"], g["final.ffn.1.running_var"], rtol=1e-5, atol=1e-5)
    for k in p2v_oracle.TRAINABLE:
        if k != "ffn.0.bias":
            assert_adam_close(st[k], g["final." + k], 24, 5e-5)
    # the export from the REFERENCE's final weights (the oracle's export alone), then from the oracle's own
    emb = p2v_oracle.generate_all_embeddings(feats, ints["cv_rowptr"], ints["cv_col"], load_state(g, "final."))
    np.testing.assert_allclose(emb, g["embeddings"], atol=5e-6)
    emb = p2v_oracle.generate_all_embeddings(feats, ints["cv_rowptr"], ints["cv_col"], st)
    np.testing.assert_allclose(emb, g["embeddings"], atol=1e-3)
    assert (np.diff(ints["cv_rowptr"]) == 0).sum() == 32


# ------------------------------------------------------------------ G11: train.train, two epochs
def joint_epoch_batches(ints, g, which, epoch):
    """The batches of DataLoader(ComplementaryDataset, B, collate_fn) (data_loader.py:133-157) in the order the reference's
    loader visited the samples, with the randn_like filler rows it drew as input data."""
    pairs, order, filler = g[which + "_pairs"], g[which + "_order"][epoch], g[which + "_filler"][epoch]
    B, n_types = int(g["batch_size"]), len(ints["type_names"])
    feats = ints["features"]
    for lo in range(0, len(order), B):
        rows = pairs[order[lo:lo + B]]
        f = filler[lo:lo + B]
        ii = [data_oracle.complementary_sample_ints(q, tg, lab, ints["type_idx"], n_types) for q, tg, lab in rows]
        real = feats[rows[:, 1]]
        pos = rows[:, 2:3] == 1
        yield {"query_idx": t(rows[:, 0].astype(np.int64)), "label": rows[:, 2],
               "query_types": t(np.array([i["query_types"] for i in ii], np.int64)),
               "positive_types": t(np.array([[i["positive_types"]] for i in ii], np.int64)),
               "negative_types": t(np.array([[i["negative_types"]] for i in ii], np.int64)),
               "positive_items": t(np.where(pos, real, f)), "negative_items": t(np.where(pos, f, real)),
               "target_features": t(real)}


def test_joint_train_epochs(golden):
    """The oracle's loop body + evaluate_batch iterated over the reference's own two-epoch run of train.train
    (train.py:16-72) on the g2 graph with g10's embeddings: 48 steps (ragged last batches), metrics over three
    validation batches after each epoch, the best-checkpoint rule."""
    g = golden("g11_joint_epochs.npz")
    ints = golden("g2_bpg1000.npz")
    st = load_state(g, "init.")
    st["product_embeddings.weight"] = t(golden("g10_p2v_epochs.npz")["embeddings"]).clone()
    # the datasets' pair order and split are the CPython stream's (data_loader.py:108-126), train then val from one stream
    rng = Random(int(g["seed"]))
    cp, sp = ints["complementary_pairs"], ints["similarity_pairs"]
    allp = np.concatenate([np.c_[cp, np.ones(len(cp), np.int32)], np.c_[sp, -np.ones(len(sp), np.int32)]])
    n = len(allp)
    assert np.array_equal(allp[rng.shuffle_perm(n)][:int(0.8 * n)], g["train_pairs"])
    assert np.array_equal(allp[rng.shuffle_perm(n)][int(0.8 * n):int(0.9 * n)], g["val_pairs"])
    mom = joint_oracle.new_moments(st)
    names = [str(x) for x in g["metric_names"]]
    losses, step, best, best_epoch = [], 0, 0.0, None
    for epoch in range(int(g["epochs"])):
        for batch in joint_epoch_batches(ints, g, "train", epoch):
            step += 1
            losses.append(float(joint_oracle.train_step(st, batch, mom, step)["loss"]))
        acc, nb = dict.fromkeys(names, 0.0), 0
        for batch in joint_epoch_batches(ints, g, "val", epoch):
            r = joint_oracle.evaluate_batch(st, batch["query_idx"], batch["query_types"], batch["positive_items"],
                                            batch["target_features"])
            for k in names:
                acc[k] += r[k]
            nb += 1
        assert nb == 3
        for k, want in zip(names, g["metric_values"][epoch]):
            assert abs(acc[k] / nb - want) < 1e-6, (epoch, k, acc[k] / nb, want)
        if acc["hit@10"] / nb > best:                                   # train.py:62
            best, best_epoch = acc["hit@10"] / nb, epoch
            best_state = {k: st[k].clone() for k in joint_oracle.TRAINABLE}
    assert step == 48 and best_epoch == int(g["best_epoch"])
    np.testing.assert_allclose(losses, g["losses"], rtol=0, atol=5e-6)
    for k in joint_oracle.TRAINABLE:
        np.testing.assert_allclose(st[k], g["final." + k], atol=1e-5, err_msg=k)
        np.testing.assert_allclose(best_state[k], g["best." + k], atol=1e-5, err_msg=k)
        np.testing.assert_allclose(mom[k][0], g["final.exp_avg." + k], atol=1e-6, err_msg=k)
        np.testing.assert_allclose(mom[k][1], g["final.exp_avg_sq." + k], atol=1e-7, err_msg=k)
        assert float(g["final.step." + k]) == 48.0
