"""Integer BPG, scalable generator, loaders' host logic.  CPU only."""
import numpy as np
import pytest

from oracle import data_oracle
from p_companion_amd.data import ComplementaryIndexDataset, IntBPG, generate_scaled_bpg


def test_intbpg_from_reference_graph(golden):
    z = golden("g2_bpg1000.npz")
    bpg = IntBPG.from_arrays(z)
    assert bpg.num_products == 1000 and bpg.n_types == 20
    for pid in (0, 17, 500, 999):
        assert np.array_equal(bpg.get_neighbors(pid), data_oracle.neighbors(z["cv_rowptr"], z["cv_col"], pid))
    # positives CSR == {pair[1] for pair in similar_pairs if pair[0] == anchor} (data_loader.py:31)
    sp = z["similarity_pairs"]
    for a in np.unique(sp[:50, 0]):
        assert set(bpg.sim_col[bpg.sim_rowptr[a]:bpg.sim_rowptr[a + 1]].tolist()) == set(sp[sp[:, 0] == a, 1].tolist())
    # every similarity anchor has a co-view out-neighbour (synthetic_data.py:110-119)
    assert bpg.degree(sp[:, 0]).min() >= 1


def test_scaled_generator_distributions():
    g = generate_scaled_bpg(20000, 100, seed=0)
    deg = np.diff(g.cv_rowptr)
    assert 14.5 < deg.mean() < 16.5 and deg.max() <= 32                      # mean 16, cap 32
    assert 2.3 < len(g.similarity_pairs) / 20000 < 3.3                       # reference cfg1: 2.95 per product
    assert 3.5 < len(g.complementary_pairs) / 20000 < 5.0                    # reference cfg1: 4.52
    assert g.degree(g.similarity_pairs[:, 0]).min() >= 1
    # feature recipe: +1.0 on the category block (synthetic_data.py:50-52)
    for c in range(5):
        rows = g.features[g.category == c]
        assert abs(rows[:, 20 * c:20 * c + 20].mean() - 1.0) < 0.05
        other = np.delete(rows, np.s_[20 * c:20 * c + 20], axis=1)
        assert abs(other.mean()) < 0.05
    assert g.type_idx.max() == 99 and np.array_equal(g.category, g.type_idx // 20)
    # same-category co-views are 1.5x as likely as cross-category ones
    src = np.repeat(np.arange(20000), deg)
    same = (g.category[src] == g.category[g.cv_col]).mean()
    assert abs(same - (0.2 * 1.5) / (0.2 * 1.5 + 0.8)) < 0.02
    # no self loops / duplicate edges; complementary pairs are not co-viewed
    key = src.astype(np.int64) * 20000 + g.cv_col
    assert len(np.unique(key)) == len(key) and (src != g.cv_col).all()
    ck = g.complementary_pairs[:, 0].astype(np.int64) * 20000 + g.complementary_pairs[:, 1]
    assert not np.isin(ck, key).any()
    g2 = generate_scaled_bpg(20000, 100, seed=0)
    assert np.array_equal(g.cv_col, g2.cv_col) and np.array_equal(g.features, g2.features)   # seeded


def test_complementary_dataset_split_and_labels(golden):
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    tr, va, te = (ComplementaryIndexDataset(bpg, m, seed=0) for m in ("train", "val", "test"))
    n = len(bpg.complementary_pairs) + len(bpg.similarity_pairs)
    assert (len(tr), len(va)) == (int(0.8 * n), int(0.9 * n) - int(0.8 * n)) and len(tr) + len(va) + len(te) >= n - 1
    assert set(np.unique(tr.pairs[:, 2]).tolist()) == {-1, 1}
    for q, t, lab in tr.pairs[:20]:
        want = data_oracle.complementary_sample_ints(q, t, lab, bpg.type_idx, bpg.n_types)
        tt = int(bpg.type_idx[t])
        assert want["positive_types"] == (tt if lab == 1 else 0)
        assert want["negative_types"] == (tt if lab == -1 else (tt + 1) % bpg.n_types)


@pytest.mark.parametrize("seed", [0, 11])
def test_complementary_dataset_cpython_mode_matches_reference(golden, seed):
    """J1 parity mode: pair order after the reference's random.shuffle + 80/10/10 split, and the integer fields of
    the first 256 samples of each mode, against vectors captured from the reference's ComplementaryDataset (G9)."""
    z = golden("g9_complementary.npz")
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    assert bpg.n_types == int(z["n_types"])
    for mode in ("train", "val", "test"):
        ds = ComplementaryIndexDataset(bpg, mode, seed=seed, sampler="cpython")
        want = z[f"s{seed}_{mode}_pairs"]
        assert np.array_equal(ds.pairs, want), mode
        n = len(z[f"s{seed}_{mode}_query_idx"])
        assert np.array_equal(ds.pairs[:n, 0], z[f"s{seed}_{mode}_query_idx"])
        assert np.array_equal(ds.pairs[:n, 2], z[f"s{seed}_{mode}_label"])
        # the oracle's label rules against what the reference's __getitem__ returned
        for i in range(n):
            q, t, lab = ds.pairs[i]
            o = data_oracle.complementary_sample_ints(q, t, lab, bpg.type_idx, bpg.n_types)
            for k in ("query_types", "positive_types", "negative_types"):
                assert o[k] == int(z[f"s{seed}_{mode}_{k}"][i]), (mode, i, k)
            assert bool(z[f"s{seed}_{mode}_positive_is_target"][i]) == (lab == 1)
    # one shared stream, datasets built back to back like train.py:111-112: train is the seed's first shuffle,
    # val comes from the continued stream (and is therefore NOT the fresh-seed val split)
    from p_companion_amd import ops
    rng = ops.CPythonRandom(seed)
    tr = ComplementaryIndexDataset(bpg, "train", sampler="cpython", rng=rng)
    va = ComplementaryIndexDataset(bpg, "val", sampler="cpython", rng=rng)
    assert np.array_equal(tr.pairs, z[f"s{seed}_train_pairs"])
    assert not np.array_equal(va.pairs, z[f"s{seed}_val_pairs"])
    import random
    random.seed(seed)
    n = len(bpg.complementary_pairs) + len(bpg.similarity_pairs)
    a = list(range(n)); random.shuffle(a)
    b = list(range(n)); random.shuffle(b)
    allp = np.concatenate([np.concatenate([bpg.complementary_pairs, np.ones((len(bpg.complementary_pairs), 1), np.int32)], 1),
                           np.concatenate([bpg.similarity_pairs, -np.ones((len(bpg.similarity_pairs), 1), np.int32)], 1)])
    assert np.array_equal(va.pairs, allp[np.array(b)][int(0.8 * n):int(0.9 * n)])


def test_unique_neighbor_layout_host_construction():
    """ops.unique_neighbors (the host-side mirror of pc_build_similarity_batch_unique): rows = distinct products
    ascending, weights = multiplicities (+ padding count), slot map and per-row slot lists consistent."""
    import torch
    from p_companion_amd import ops
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 40, size=(16, 6)).astype(np.int32)
    idx[rng.random(idx.shape) < 0.3] = -1
    u = ops.unique_neighbors(torch.from_numpy(idx))
    U = u["n_unique"]
    rows, w, slot = u["nb_rows"].numpy(), u["weight"].numpy(), u["slot_row"].numpy()
    real = idx >= 0
    assert np.array_equal(rows[:U], np.unique(idx[real])) and rows[U] == -1
    assert w[:U].sum() == real.sum() and w[U] == (~real).sum()
    assert np.array_equal(rows[slot][real], idx[real]) and (slot[~real] == U).all()
    ro, rs = u["ref_off"].numpy(), u["ref_slot"].numpy()
    assert ro[0] == 0 and ro[U] == ro[U + 1] == real.sum() and len(rs) == real.sum()
    for r in range(U):
        s = rs[ro[r]:ro[r + 1]]
        assert len(s) == w[r] and (np.diff(s) > 0).all() and (slot.reshape(-1)[s] == r).all()


def test_recommend_oracle_matches_bruteforce():
    """oracle.joint_oracle.recommend (restatement of inference.py:90-118) on a toy catalogue."""
    from oracle import joint_oracle
    rng = np.random.default_rng(1)
    feats = rng.standard_normal((50, 128)).astype(np.float32)
    types = rng.integers(0, 5, 50)
    proj = rng.standard_normal((4, 128)).astype(np.float32)
    want = np.array([0, 3, 3, 4])
    out = joint_oracle.recommend(proj, want, types, feats, 3)
    for r, (ids, sc) in enumerate(out):
        cand = np.nonzero(types == want[r])[0]
        s = feats[cand] @ proj[r]
        best = cand[np.argsort(-s)[:3]]
        assert np.array_equal(ids, best) and np.allclose(sc, np.sort(s)[::-1][:3], atol=1e-5)


def test_epoch_plans_are_permutations_deterministic_in_seed_and_epoch():
    """Host logic of the two index loaders' epoch turnover (CPU device: same code, CPU generator).  SimilarityIndexLoader:
    the order is a permutation of the pairs, the plan carries per batch the largest neighbour count and the number of real
    slots, both are functions of (seed, epoch) only.  ComplementaryIndexLoader.epoch_pairs: a row permutation of the
    dataset's labelled pairs, advancing the epoch counter."""
    import torch
    from p_companion_amd.data import ComplementaryIndexLoader, SimilarityIndexLoader
    bpg = generate_scaled_bpg(600, 12, seed=4)
    S = bpg.similarity_pairs.shape[0]
    B = 64

    def plan(seed, epoch, drop_last):
        ld = SimilarityIndexLoader.__new__(SimilarityIndexLoader)           # (the constructor uploads the graph for the device sampler)
        ld.bpg, ld.batch_size, ld.shuffle, ld.seed, ld.drop_last, ld.device, ld.epoch = bpg, B, True, seed, drop_last, "cpu", epoch
        ld._deg = bpg.degree(bpg.similarity_pairs[:, 0])
        ld._n_pairs = S
        return ld._epoch_plan(S)

    for drop_last in (True, False):
        perm, st = plan(3, 0, drop_last)
        perm2, st2 = plan(3, 0, drop_last)
        assert torch.equal(perm, perm2) and torch.equal(st, st2)
        assert sorted(perm.tolist()) == list(range(S))
        assert not torch.equal(perm, plan(3, 1, drop_last)[0]) and not torch.equal(perm, plan(4, 0, drop_last)[0])
        deg = bpg.degree(bpg.similarity_pairs[:, 0])[perm.numpy()]
        n = S // B if drop_last else (S + B - 1) // B
        assert st.shape == (n, 2)
        for i in range(n):
            d = deg[i * B:(i + 1) * B]
            assert int(st[i, 0]) == int(d.max()) and int(st[i, 1]) == int(d.sum())

    ds = ComplementaryIndexDataset(bpg, "train")
    ld = ComplementaryIndexLoader.__new__(ComplementaryIndexLoader)
    ld.dataset, ld.batch_size, ld.shuffle, ld.seed, ld.device, ld.epoch = ds, B, True, 5, "cpu", 0
    e0 = ld.epoch_pairs()
    e1 = ld.epoch_pairs()
    assert ld.epoch == 2 and e0.shape == (len(ds), 3) and e0.dtype == torch.int32 and not torch.equal(e0, e1)
    as_rows = lambda t: sorted(map(tuple, t.tolist()))
    assert as_rows(e0) == as_rows(e1) == sorted(map(tuple, np.asarray(ds.pairs, np.int32).tolist()))
    ld.epoch = 0
    assert torch.equal(ld.epoch_pairs(), e0)
    ld.shuffle = False
    assert torch.equal(ld.epoch_pairs(), torch.from_numpy(np.ascontiguousarray(ds.pairs, np.int32)))
