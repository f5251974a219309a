"""The loaders on the GPU: ComplementaryDataset batches against the reference's golden vectors (data_loader.py:108-157), the Zipf and
uniform negative samplers against their oracles, the epoch order (a keyed Feistel bijection instead of torch.randperm), the buffer
ring of the training loops.  Needs an MI355X."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cfg(**over):
    c = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                        MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=40, DEVICE=torch.device("cuda"),
                        LEARNING_RATE=1e-3, BATCH_SIZE=64, PRODUCT2VEC_EPOCHS=1, NUM_EPOCHS=1, MODEL_DIR="/tmp/pc_r3_models")
    c.__dict__.update(over)
    return c


# ------------------------------------------------------------------ epoch order on the device
@pytest.mark.parametrize("n", [1, 2, 3, 17, 256, 4097, 274013, 1 << 20])
def test_epoch_permutation_is_the_oracles_bijection(n):
    from oracle import philox_oracle
    from p_companion_amd import ops
    for seed, epoch in ((0, 0), (1000020, 3), (2 ** 40 + 5, 77)):
        p = ops.epoch_permutation(n, seed, epoch, "cuda").cpu().numpy()
        assert np.array_equal(p, philox_oracle.epoch_permutation(n, seed, epoch))
        assert np.array_equal(np.sort(p), np.arange(n))
    if n > 16:                                                    # epochs and seeds give different orders
        a = ops.epoch_permutation(n, 5, 0, "cuda")
        assert not torch.equal(a, ops.epoch_permutation(n, 5, 1, "cuda"))
        assert not torch.equal(a, ops.epoch_permutation(n, 6, 0, "cuda"))


def test_shuffle_rows_and_epoch_plan():
    from oracle import philox_oracle
    from p_companion_amd import ops
    rng = np.random.default_rng(0)
    n = 50021
    rows = torch.from_numpy(rng.integers(-5, 1 << 20, (n, 3)).astype(np.int32)).cuda()
    out = ops.shuffle_rows_i32(rows, 12345, 4)
    perm = philox_oracle.epoch_permutation(n, 12345, 4)
    assert np.array_equal(out.cpu().numpy(), rows.cpu().numpy()[perm])
    deg = torch.from_numpy(rng.integers(0, 33, n).astype(np.int32)).cuda()
    B = 4096
    for nb in (n // B, (n + B - 1) // B):                          # drop_last and the ragged last batch
        order = torch.from_numpy(perm).cuda()
        plan = ops.epoch_plan(order, deg, n, B, nb).cpu().numpy()
        d = deg.cpu().numpy()[perm]
        for b in range(nb):
            seg = d[b * B:(b + 1) * B]
            assert plan[b, 0] == seg.max() and plan[b, 1] == seg.sum()
        ident = ops.epoch_plan(None, deg, n, B, nb).cpu().numpy()
        assert ident[0, 0] == deg.cpu().numpy()[:B].max() and ident[0, 1] == deg.cpu().numpy()[:B].sum()


def test_loaders_run_without_torch_randperm(monkeypatch):
    """Both throughput loaders draw their epoch order with the library's own kernels: torch.randperm is never called on
    the device, every epoch is a permutation of the dataset, and two loaders with the same seed agree."""
    from p_companion_amd.data import (ComplementaryIndexDataset, ComplementaryIndexLoader, SimilarityIndexLoader,
                                      generate_scaled_bpg)
    real = torch.randperm

    def guarded(*a, **k):
        dev = k.get("device")
        assert dev is None or torch.device(dev).type != "cuda", "torch.randperm on the device (ATen + rocprim sort kernels)"
        return real(*a, **k)

    monkeypatch.setattr(torch, "randperm", guarded)
    bpg = generate_scaled_bpg(3000, 20, seed=0)
    S = bpg.similarity_pairs.shape[0]
    seen = []
    for rep in range(2):
        ld = SimilarityIndexLoader(bpg, 256, shuffle=True, sampler="philox", seed=3, drop_last=False, device="cuda")
        epochs = []
        for _ in range(2):
            anchors = torch.cat([b["anchor_idx"] for b in ld]).cpu().numpy()
            assert anchors.shape[0] == S
            assert np.array_equal(np.sort(anchors), np.sort(bpg.similarity_pairs[:, 0]))
            epochs.append(anchors)
        assert not np.array_equal(epochs[0], epochs[1])
        seen.append(epochs)
    assert np.array_equal(seen[0][0], seen[1][0]) and np.array_equal(seen[0][1], seen[1][1])
    ds = ComplementaryIndexDataset(bpg, "train")
    cl = ComplementaryIndexLoader(ds, 512, shuffle=True, seed=1, device="cuda")
    e0, e1 = cl.epoch_pairs().cpu().numpy(), cl.epoch_pairs().cpu().numpy()
    key = lambda a: np.sort(a[:, 0].astype(np.int64) * (1 << 40) + a[:, 1].astype(np.int64) * 4 + a[:, 2] + 1)
    assert np.array_equal(key(e0), key(ds.pairs)) and np.array_equal(key(e1), key(ds.pairs))
    assert not np.array_equal(e0, e1)


# ------------------------------------------------------------------ Zipf sampler: every wave terminates
def test_zipf_sampler_terminates_and_reports_when_an_anchor_has_too_few_candidates():
    """5 products, the anchor's positives are products 1..3: only product 4 is eligible, K = 3 negatives are asked for.
    The kernel returns (the unbounded rejection loop of round 2 would spin for ever), fills the one eligible product,
    pads with -1 and counts the sample; the loader refuses such a graph up front."""
    from p_companion_amd import ops
    sim_pairs = torch.tensor([[0, 1], [0, 2], [0, 3]], dtype=torch.int32).cuda()
    rowptr = torch.tensor([0, 3, 3, 3, 3, 3], dtype=torch.int32).cuda()
    col = torch.tensor([1, 2, 3], dtype=torch.int32).cuda()
    graph = {"sim_pairs": sim_pairs, "sim_rowptr": rowptr, "sim_col": col, "n_products": 5}
    thr = torch.from_numpy(ops.zipf_octave_thresholds(5).view(np.int32).copy()).cuda()
    failed = torch.zeros(1, dtype=torch.int32, device="cuda")
    # k_neg + 1 < n_products holds (the C-side guard), eligibility does not
    ng = ops.sample_negatives_zipf(torch.zeros(4, dtype=torch.int32, device="cuda"), graph, 3, 1, 0, thr, failed=failed)
    torch.cuda.synchronize()
    assert int(failed) == 4
    for row in ng.cpu().numpy():
        assert row[0] == 4 and row[1] == -1 and row[2] == -1
    # a healthy anchor next to it: bit-identical to a run without the counter (the bound never triggers)
    g2 = dict(graph, sim_pairs=torch.tensor([[4, 0]], dtype=torch.int32).cuda(),
              sim_rowptr=torch.tensor([0, 0, 0, 0, 0, 1], dtype=torch.int32).cuda(), sim_col=torch.tensor([0], dtype=torch.int32).cuda())
    failed.zero_()
    a = ops.sample_negatives_zipf(torch.zeros(64, dtype=torch.int32, device="cuda"), g2, 3, 1, 0, thr, failed=failed)
    b = ops.sample_negatives_zipf(torch.zeros(64, dtype=torch.int32, device="cuda"), g2, 3, 1, 0, thr)
    assert int(failed) == 0 and torch.equal(a, b)
    assert set(np.unique(a.cpu().numpy()).tolist()) <= {1, 2, 3}


# ------------------------------------------------------------------ the loader's buffer ring
@pytest.mark.parametrize("unique", [True, False])
def test_loader_buffer_ring_hands_out_the_same_batches(unique):
    """reuse_buffers=True (what bench.py and train_model's driver use): batches built into a ring of preallocated buffers,
    consumed one at a time, are bit-identical to the freshly allocated ones over several epochs (epoch boundaries, a ragged
    last batch, the abandoned-iterator path), and training on them gives bit-identical parameters."""
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    bpg = generate_scaled_bpg(8000, 40, seed=3)
    table = bpg.cuda()["features"]
    mk = lambda ring: SimilarityIndexLoader(bpg, 512, seed=5, drop_last=False, device="cuda", unique=unique, reuse_buffers=ring)
    a, b = mk(True), mk(False)
    torch.manual_seed(0)
    ma, mb = Product2Vec(cfg()).to("cuda").train(), Product2Vec(cfg()).to("cuda").train()
    mb.load_state_dict(ma.state_dict())
    oa, ob = FusedAdam(ma, lr=1e-3), FusedAdam(mb, lr=1e-3)
    n = 0
    for epoch in range(3):
        for i, (x, y) in enumerate(zip(a, b)):
            for k in ("anchor_idx", "positive_idx", "negative_idx"):
                assert torch.equal(x[k], y[k]), (epoch, i, k)
            nx, ny = x["neighbor_compact"], y["neighbor_compact"]
            assert torch.equal(nx["slot_row"], ny["slot_row"])
            nu = int(nx["n_unique"]) if unique else nx["nb_rows"].numel() - 1
            assert torch.equal(nx["nb_rows"][:nu + 1], ny["nb_rows"][:nu + 1])
            if x["anchor_idx"].numel() >= 2:
                la, lb = ma.train_step_indexed(table, x), mb.train_step_indexed(table, y)
                oa.step(); ob.step()
                assert torch.equal(la, lb)
            n += 1
            if epoch == 1 and i == 3:
                break                                             # abandon this epoch's iterators mid-way
    assert n > 20
    for (k, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
        assert torch.equal(p, q), k


def test_look_ahead_builder_queued_by_the_step_hands_out_the_same_batches():
    """batch["_after_step"] (what train_step_indexed calls behind its launches): the loader's next look-ahead builder is queued
    then instead of at the next hand-out.  Same batches, bit for bit, as a loader that never offers it (kick_after_step False) and
    as one whose consumer ignores it; calling it twice queues nothing twice; an epoch's last `depth` batches carry none."""
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    bpg = generate_scaled_bpg(6000, 30, seed=2)
    mk = lambda: SimilarityIndexLoader(bpg, 512, seed=9, drop_last=False, device="cuda", reuse_buffers=True)
    kicked, ignored, plain = mk(), mk(), mk()
    plain.kick_after_step = False
    n = len(kicked)
    for epoch in range(2):
        seen = 0
        for i, (x, y, z) in enumerate(zip(kicked, ignored, plain)):
            assert "_after_step" not in z
            assert ("_after_step" in x) == (i + 4 < n) == ("_after_step" in y), i     # prefetch depth 4 with the ring
            nu = int(x["neighbor_compact"]["n_unique"])
            for other in (y, z):
                for k in ("anchor_idx", "positive_idx", "negative_idx"):
                    assert torch.equal(x[k], other[k]), (epoch, i, k)
                assert int(other["neighbor_compact"]["n_unique"]) == nu
                for k in ("nb_rows", "weight"):
                    assert torch.equal(x["neighbor_compact"][k][:nu + 1], other["neighbor_compact"][k][:nu + 1]), (epoch, i, k)
                assert torch.equal(x["neighbor_compact"]["slot_row"], other["neighbor_compact"]["slot_row"])
                rows = x["anchor_idx"].numel() * 7 + nu + 1
                assert torch.equal(x["neighbor_compact"]["step_rows"][:rows], other["neighbor_compact"]["step_rows"][:rows])
            if "_after_step" in x:
                x["_after_step"]()
                x["_after_step"]()                                   # nothing owed any more: a no-op
            seen += 1
        assert seen == n


@pytest.mark.parametrize("mode", ["train", "val"])
def test_complementary_batch_against_reference_golden(golden, mode):
    """J1 end to end in parity mode: sampler='cpython' pair order + pc_build_complementary_batch on the device against
    what the reference's ComplementaryDataset.__getitem__ returned for the same random.seed (G9): every integer
    field bit-exact, the real item row is the target's feature row, the other one is filler."""
    from p_companion_amd import ops
    from p_companion_amd.data import ComplementaryIndexDataset, IntBPG
    z = golden("g9_complementary.npz")
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    ds = ComplementaryIndexDataset(bpg, mode, seed=11, sampler="cpython")
    n = len(z[f"s11_{mode}_query_idx"])
    rows = torch.from_numpy(np.ascontiguousarray(ds.pairs[:n], np.int32)).cuda()
    g = bpg.cuda()
    b = ops.build_complementary_batch(rows, g["features"], g["type_idx"], bpg.n_types, 5, 0)
    for k, dk in (("query_idx", "query_idx"), ("query_types", "query_types"), ("positive_types", "positive_types"),
                  ("negative_types", "negative_types")):
        assert np.array_equal(b[dk].reshape(-1).cpu().numpy(), z[f"s11_{mode}_{k}"]), k
    pos_is = z[f"s11_{mode}_positive_is_target"]
    tgt = torch.from_numpy(bpg.features[ds.pairs[:n, 1]]).cuda()
    is_pos = (b["positive_items"] == tgt).all(1).cpu().numpy()
    is_neg = (b["negative_items"] == tgt).all(1).cpu().numpy()
    assert np.array_equal(is_pos, pos_is) and np.array_equal(is_neg, ~pos_is)
    assert torch.equal(b["target_features"], tgt)


# ------------------------------------------------------------------ Zipf negatives (BASELINE configs[4])
def test_zipf_negatives_bit_exact_vs_oracle_and_rules(golden):
    from oracle import philox_oracle
    from p_companion_amd import ops
    from p_companion_amd.data import IntBPG
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    g = bpg.cuda()
    P = bpg.num_products
    thr = ops.zipf_octave_thresholds(P)
    thr_d = torch.from_numpy(thr.view(np.int32).copy()).cuda()
    rs = np.random.default_rng(0)
    perm = rs.permutation(P).astype(np.int32)
    pair_ids = torch.arange(0, 300, dtype=torch.int32).cuda()
    for pm in (None, perm):
        got = ops.sample_negatives_zipf(pair_ids, g, 5, 77, 3, thr_d, None if pm is None else torch.from_numpy(pm).cuda())
        want = philox_oracle.zipf_negatives(np.arange(300), bpg.similarity_pairs, bpg.sim_rowptr, bpg.sim_col, P, 5, 77, 3, thr, pm)
        assert np.array_equal(got.cpu().numpy(), want)
        for b in range(300):                                     # the reference's rejection rules
            a = bpg.similarity_pairs[b, 0]
            pos = set(bpg.sim_col[bpg.sim_rowptr[a]:bpg.sim_rowptr[a + 1]].tolist())
            row = want[b].tolist()
            assert a not in row and not (pos & set(row)) and len(set(row)) == 5


def test_zipf_negatives_follow_one_over_rank():
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    bpg = generate_scaled_bpg(50000, 100, seed=4)
    ld = SimilarityIndexLoader(bpg, 4096, seed=9, drop_last=True, negatives="zipf", prefetch=False)
    cnt = np.zeros(bpg.num_products, np.int64)
    n = 0
    for b in ld:
        # the FIRST negative of a sample is one draw of the distribution (the later ones exclude the earlier: the "no
        # repeats" rule of data_loader.py:36 thins the head)
        np.add.at(cnt, b["negative_idx"][:, 0].cpu().numpy().reshape(-1), 1)
        n += 1
        if n == 40:
            break
    total = cnt.sum()
    h = np.sum(1.0 / np.arange(1, bpg.num_products + 1))
    # octave masses: ranks [2^j, 2^(j+1)) each carry ~ln 2 / H_P of the draws (product id = rank - 1 here)
    for j in (0, 3, 6, 9, 12, 15):
        lo, hi = (1 << j) - 1, min((1 << (j + 1)) - 1, bpg.num_products)
        want = np.sum(1.0 / np.arange(lo + 1, hi + 1)) / h
        got = cnt[lo:hi].sum() / total
        assert abs(got - want) < 0.006, (j, got, want)
    assert cnt[0] > 10 * max(cnt[1000:1010].mean(), 1)           # the head is heavy
