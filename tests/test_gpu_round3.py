"""Round-3 additions on the GPU: the loaders' own epoch-order kernels (no torch.randperm), the bounded Zipf sampler,
the epoch runner's error / hyper-parameter conventions, the fused joint step against the ORACLE at the reference's own
NUM_TYPES (config.py:27), and the absorbed attention's saved tensors.  Needs an MI355X."""
from types import SimpleNamespace

import ctypes

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


def joint_batch(B, P, T, seed=0, dev="cuda"):
    g = torch.Generator().manual_seed(seed)
    return {"query_idx": torch.randint(0, P, (B,), generator=g, dtype=torch.int32).to(dev),
            "query_types": torch.randint(0, T, (B,), generator=g).to(dev),
            "positive_types": torch.randint(0, T, (B, 1), generator=g).to(dev),
            "negative_types": torch.randint(0, T, (B, 1), generator=g).to(dev),
            "positive_items": torch.randn(B, 128, generator=g).to(dev),
            "negative_items": torch.randn(B, 128, generator=g).to(dev)}


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


# ------------------------------------------------------------------ train(): the epoch-runner path
def _small_bpg():
    from p_companion_amd.data import generate_scaled_bpg
    return generate_scaled_bpg(600, 20, seed=0)


def test_train_epoch_runner_raises_index_error_for_an_out_of_range_type_id(tmp_path):
    """on train()'s default path (warmup = 0 -> straight to pc_joint_train_epoch) the bad-id counter did
    not exist yet, so the kernel clamped silently and the per-epoch raise_index_errors() could never fire."""
    from p_companion_amd import train as ptrain
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader
    bpg = _small_bpg()
    c = cfg(NUM_TYPES=int(bpg.n_types), BATCH_SIZE=64, MODEL_DIR=str(tmp_path))
    table = torch.from_numpy(bpg.features).cuda()
    tr = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), 64, shuffle=True, device="cuda")
    va = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "val"), 64, shuffle=False, device="cuda")
    assert len(tr.dataset) >= 64
    # a product table smaller than the graph's product ids is the one range _check_ranges cannot see through a device
    # table; poison one type id on the device instead (the loader's type table is what the kernels read)
    tr.type_idx = tr.type_idx.clone()
    tr.type_idx[int(tr.dataset.pairs[0, 0])] = int(bpg.n_types) + 5
    with pytest.raises(IndexError):
        ptrain.train(c, tr, va, table)


def test_train_restores_the_callers_loader_and_follows_lr_changes(tmp_path):
    from p_companion_amd import train as ptrain
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
    bpg = _small_bpg()
    c = cfg(NUM_TYPES=int(bpg.n_types), BATCH_SIZE=64, MODEL_DIR=str(tmp_path))
    table = torch.from_numpy(bpg.features).cuda()
    tr = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), 64, shuffle=True, device="cuda")
    va = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "val"), 64, shuffle=False, device="cuda")
    ptrain.train(c, tr, va, table)
    assert tr.out is None                                         # train() borrowed the loader, it did not keep it
    b = next(iter(tr))
    assert "_deferred" not in b and b["query_idx"].numel() > 0

    # the fused update reads lr / betas / eps from optimizer.param_groups on every call, like the eager path
    def run(direct):
        torch.manual_seed(0)
        m = PCompanion(c, table).to("cuda").train()
        opt = FusedAdam(m, lr=1e-3)
        step = GraphedJointStep(m, opt, 64, warmup=0, mode="direct") if direct else None
        for s in range(4):
            if s == 2:
                opt.param_groups[0]["lr"] = 5e-2                 # an LR scheduler's step
            bt = joint_batch(64, 600, int(bpg.n_types), seed=s)
            if direct:
                step(bt)
            else:
                m.train_step(bt, optimizer=opt)
        return {k: v.detach().clone() for k, v in m.state_dict().items()}
    a, e = run(True), run(False)
    for k in a:
        assert torch.equal(a[k], e[k]), k


# ------------------------------------------------------------------ the reference's own NUM_TYPES against the oracle
def test_fused_joint_step_at_reference_num_types_against_the_oracle():
    """config.py:27 NUM_TYPES = 34800, B = 256 (config.py:19): loss, top-k (bit-exact) and all ten gradients of the fused
    step against oracle.joint_oracle.train_step, incl. the rows of both [34800,64] tables that must stay exactly zero; and
    the parameters after the finish kernel's Adam update."""
    from oracle import joint_oracle
    from p_companion_amd.p_companion import PCompanion
    from p_companion_amd.product2vec import FusedAdam
    T, B, P = 34800, 256, 1000
    g = torch.Generator().manual_seed(5)
    table = torch.randn(P, 128, generator=g)
    torch.manual_seed(6)
    m = PCompanion(cfg(NUM_TYPES=T), table).to("cuda").train()
    opt = FusedAdam(m, lr=1e-3)
    st0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    b = joint_batch(B, P, 20, seed=11)                            # 20 live types (synthetic_data.py:16-17) of 34800 rows
    lf, tf = m.train_step(b, optimizer=opt)
    hb = {k: v.cpu() for k, v in b.items()}
    st = {k: v.clone() for k, v in st0.items()}
    ref = joint_oracle.train_step(st, hb, joint_oracle.new_moments(st0), 1)
    assert abs(float(lf[0]) - float(ref["loss"])) < 1e-5
    assert abs(float(lf[1]) - float(ref["type_loss"])) < 1e-5 and abs(float(lf[2]) - float(ref["item_loss"])) < 1e-5
    assert np.array_equal(tf.cpu().numpy(), ref["out"]["complementary_types"].numpy())
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        gr = ref["grads"][k]
        assert float((p.grad.cpu() - gr).abs().max()) <= 1e-6 + 1e-4 * float(gr.abs().max()), k
        if k.endswith("type_embeddings.weight"):
            zero_rows = gr.abs().amax(1) == 0
            assert int(zero_rows.sum()) > 34000
            assert float(p.grad.cpu()[zero_rows].abs().max()) == 0.0, k
        d = (p.detach().cpu() - st[k]).abs()
        assert float((d <= 2e-5).float().mean()) >= 0.99 and float(d.max()) <= 2.1e-3, k


# ------------------------------------------------------------------ absorbed attention
def test_attention_saves_no_kv_buffer_and_matches_the_oracle_with_bias():
    """The K|V projections are absorbed into the per-sample side: the saved tensors are [B,4,D]-sized, not [B*N,2D], and
    forward / every gradient equal the oracle's nn.MultiheadAttention restatement with NON-ZERO in_proj / out_proj biases
    (torch initialises them to 0, which would hide an error in the bias algebra: the key bias leaves the softmax, the
    value bias enters through sum_n p_n)."""
    from oracle import p2v_oracle
    from p_companion_amd import ops
    g = torch.Generator().manual_seed(2)
    B, N, D = 37, 11, 128
    st = p2v_oracle.init_state(3)
    st["attention.in_proj_bias"] = torch.randn(3 * D, generator=g) * 0.3
    st["attention.out_proj.bias"] = torch.randn(D, generator=g) * 0.3
    q = torch.randn(B, D, generator=g)
    kv = torch.randn(B, N, D, generator=g)
    params = {k: v.cuda() for k, v in st.items()}
    out, sv = ops.attention_forward(params, q.cuda(), kv.cuda())
    assert "kv" not in sv and sv["qt"].shape == (B, 4, D) and sv["c"].shape == (B, 4, D)
    assert sum(v.numel() for v in sv.values() if torch.is_tensor(v)) < B * N * 2 * D
    qr, kr = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in st.items() if k.startswith("attention.")}
    ref = p2v_oracle.attention(qr, kr, dict(st, **pr))
    assert float((out.cpu() - ref.detach()).abs().max()) < 2e-5
    w = torch.randn(B, D, generator=g)
    (ref * w).sum().backward()
    grads, dq, dk = ops.attention_backward(params, q.cuda(), kv.cuda(), w.cuda(), sv)
    tol = lambda r: 2e-6 + 2e-4 * float(r.abs().max())
    assert float((dq.cpu() - qr.grad).abs().max()) <= tol(qr.grad)
    assert float((dk.cpu() - kr.grad).abs().max()) <= tol(kr.grad)
    for k, p in pr.items():
        assert float((grads[k].cpu() - p.grad).abs().max()) <= tol(p.grad), k
    # the key-bias gradient is analytically zero: the reference holds rounding noise there, this path an exact 0
    assert float(grads["attention.in_proj_bias"][D:2 * D].abs().max()) == 0.0
    assert float(pr["attention.in_proj_bias"].grad[D:2 * D].abs().max()) < 1e-5


# ------------------------------------------------------------------ big tables: deterministic gradients + row lists
def _pc_big(T, P=2000, seed=3):
    from p_companion_amd.p_companion import PCompanion
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(P, 128, generator=g)
    torch.manual_seed(seed + 1)
    return PCompanion(cfg(NUM_TYPES=T), table).to("cuda").train()


@pytest.mark.parametrize("T,live", [(34800, 100), (34800, 20), (2000, 60)])
def test_fused_joint_step_is_bitwise_reproducible_at_the_reference_num_types(T, live):
    """config.py:27 NUM_TYPES = 34800 (and any T > 512 with <= 512 touched rows per table): the table gradients are
    fixed-order sums -- two runs of the same step from the same state give bit-identical gradients of all ten tensors,
    and the in-kernel Adam update is bit-identical too.  (Round 2: float atomics above T = 512.)"""
    from p_companion_amd.product2vec import FusedAdam
    B = 4096
    m1, m2 = _pc_big(T), _pc_big(T)
    o1, o2 = FusedAdam(m1, lr=1e-2), FusedAdam(m2, lr=1e-2)
    for s in range(3):
        b = joint_batch(B, 2000, live, seed=70 + s)
        l1, t1 = m1.train_step(b, optimizer=o1)
        l2, t2 = m2.train_step(b, optimizer=o2)
        assert torch.equal(l1, l2) and torch.equal(t1, t2)
        assert torch.equal(m1._gflat, m2._gflat), s
        for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert torch.equal(p1, p2), (s, k)
    b = joint_batch(B, 2000, live, seed=99)
    m1.train_step(b)
    g1 = m1._gflat.clone()
    for _ in range(3):
        m1.train_step(b)
        assert torch.equal(g1, m1._gflat)


def test_touched_row_lists_and_the_row_list_exchange_on_the_gpu():
    """pc_joint_fused_touched lists exactly the rows of the two [T,64] tables that received a gradient (ascending); two
    replicas' lists merged by TableRowExchange.merge with the HIP row movers reproduce the step on the concatenated batch."""
    from p_companion_amd import distributed as pdist
    from p_companion_amd import ops
    from p_companion_amd.p_companion import GraphedJointStep
    from p_companion_amd.product2vec import FusedAdam
    T, B, K = 34800, 512, 3
    names = ("complementary_type_embeddings.weight", "query_type_embeddings.weight")
    full = joint_batch(2 * B, 2000, 60, seed=4)
    reps, lists = [], []
    for r in range(2):
        m = _pc_big(T)
        step = GraphedJointStep(m, FusedAdam(m), B, warmup=0, mode="direct", grad_hook=lambda g: g)   # gradients only
        half = {k: v[r * B:(r + 1) * B].contiguous() for k, v in full.items()}
        step(half)
        rc, rq, nt = ops.joint_fused_touched(step.prepared.ws, B, T, K)
        n_c, n_q = (int(v) for v in nt.tolist())
        params = dict(m.named_parameters())
        per_table = []
        for nm, ids in ((names[0], rc[:n_c]), (names[1], rq[:n_q])):
            g = params[nm].grad
            nz = torch.nonzero(g.abs().amax(1) > 0).reshape(-1).to(torch.int32)
            # (a touched row can sum to exactly zero only by accident; the list must cover every non-zero row, ascending)
            assert torch.equal(torch.sort(ids).values, ids) and len(torch.unique(ids)) == ids.numel()
            assert bool(torch.isin(nz, ids).all()) and ids.numel() <= nz.numel() + 2
            per_table.append((ids.clone(), ops.gather_rows(g, ids)))
        reps.append((m, params))
        lists.append(per_table)
    ref = _pc_big(T)
    ref.train_step(full)
    rparams = dict(ref.named_parameters())
    for ti, nm in enumerate(names):
        for r in range(2):
            g = reps[r][1][nm].grad
            pdist.TableRowExchange.merge(g, [lists[0][ti], lists[1][ti]], 2, ops.scatter_rows, ops.scatter_add_rows, lists[r][ti][0])
        assert torch.equal(reps[0][1][nm].grad, reps[1][1][nm].grad)                       # both replicas: the same bits
        want = rparams[nm].grad
        assert float((reps[0][1][nm].grad - want).abs().max()) <= 1e-7 + 1e-5 * float(want.abs().max()), nm


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


def _fork_digest():
    """60 steps of the fused Product2Vec step through the throughput loader -> a digest of the parameters, the BatchNorm
    statistics and the losses."""
    import hashlib
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    bpg = generate_scaled_bpg(20_000, 100, seed=3)
    table = bpg.cuda()["features"]
    torch.manual_seed(0)
    m = Product2Vec(cfg()).to("cuda").train()
    opt = FusedAdam(m, lr=1e-3)
    h = hashlib.sha256()
    n = 0
    for b in SimilarityIndexLoader(bpg, 1024, seed=2, drop_last=True, device="cuda", reuse_buffers=True):
        loss = m.train_step_indexed(table, b)
        opt.step()
        h.update(loss.detach().cpu().numpy().tobytes())
        n += 1
        if n == 60:
            break
    torch.cuda.synchronize()
    h.update(m.flatten_parameters()[0].detach().cpu().numpy().tobytes())
    h.update(m.ffn[1].running_var.cpu().numpy().tobytes())
    return h.hexdigest()


def test_side_queue_fork_changes_no_bit():
    """The fused step runs the attention block's few-row weight gradients and the BatchNorm-backward finalize on the
    library's side queue (csrc/common.h PcFork; include/pcompanion_hip.h "Library-owned device state").  Same digest of 60
    steps' losses, parameters and running statistics with the side queue, with everything on the caller's stream
    (pc_set_option(PC_OPT_SIDE_QUEUE, 0)), and again with it after pc_release_device_state() destroyed and the next step
    re-created it."""
    from p_companion_amd import _lib
    L = _lib.lib()
    v = ctypes.c_int(-1)
    assert L.pc_get_option(_lib.PC_OPT_SIDE_QUEUE, ctypes.byref(v)) == 0 and v.value == 1      # the default
    assert L.pc_set_option(99, 1) == -1 and L.pc_set_option(_lib.PC_OPT_SIDE_QUEUE, 2) == -1
    outs = []
    try:
        for on in (1, 0, 1):
            assert L.pc_set_option(_lib.PC_OPT_SIDE_QUEUE, on) == 0
            outs.append(_fork_digest())
            torch.cuda.synchronize()
            assert L.pc_release_device_state() == 0
    finally:
        L.pc_set_option(_lib.PC_OPT_SIDE_QUEUE, 1)
    assert outs[0] == outs[1] == outs[2]


def test_fused_p2v_step_under_stream_capture_stays_on_one_queue():
    """A stream that is being captured keeps the fused step on itself (no side queue inside a capture): the captured step,
    replayed, gives the eager step's loss and gradients bit for bit."""
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import Product2Vec
    bpg = generate_scaled_bpg(20_000, 100, seed=4)
    table = bpg.cuda()["features"]
    torch.manual_seed(0)
    m = Product2Vec(cfg()).to("cuda").train()
    b = next(iter(SimilarityIndexLoader(bpg, 512, seed=1, drop_last=True, device="cuda")))
    loss_e = m.train_step_indexed(table, b).clone()
    grad_e = m.flatten_parameters()[1].clone()
    m.ffn[1].running_mean.zero_(); m.ffn[1].running_var.fill_(1.0); m.ffn[1].num_batches_tracked.zero_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        m.train_step_indexed(table, b)                            # (workspaces for this stream exist before the capture)
        m.ffn[1].running_mean.zero_(); m.ffn[1].running_var.fill_(1.0); m.ffn[1].num_batches_tracked.zero_()
        with torch.cuda.graph(g, stream=side):
            loss_c = m.train_step_indexed(table, b)
    torch.cuda.current_stream().wait_stream(side)
    m.flatten_parameters()[1].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss_c, loss_e) and torch.equal(m.flatten_parameters()[1], grad_e)
