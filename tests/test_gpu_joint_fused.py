"""The fused P-Companion joint step (train.py:42-48 as one C-ABI call; csrc/joint_fused.hip) against the launch-per-op step and the
oracle: small tables, the reference's NUM_TYPES = 34800 (config.py:27) with and without config.py:12's DROPOUT, other K, ragged
batches, ids outside the tables, exact ties of the per-sample top-K, bitwise reproducibility, the touched-row lists.  Needs an
MI355X."""
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


def joint_batch(B, P, T, seed=0, dev="cuda"):
    g = torch.Generator().manual_seed(seed)
    return {"query_idx": torch.randint(0, P, (B,), generator=g, dtype=torch.int32).to(dev),
            "query_types": torch.randint(0, T, (B,), generator=g).to(dev),
            "positive_types": torch.randint(0, T, (B, 1), generator=g).to(dev),
            "negative_types": torch.randint(0, T, (B, 1), generator=g).to(dev),
            "positive_items": torch.randn(B, 128, generator=g).to(dev),
            "negative_items": torch.randn(B, 128, generator=g).to(dev)}


# ------------------------------------------------------------------ big tables: deterministic gradients + row lists
def _pc_big(T, P=2000, seed=3, dropout=0.0):
    from p_companion_amd.p_companion import PCompanion
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(P, 128, generator=g)
    torch.manual_seed(seed + 1)
    m = PCompanion(cfg(NUM_TYPES=T, DROPOUT=dropout), table).to("cuda").train()
    m.type_transition._dropout_seed, m.type_transition._dropout_step = 977, 0       # (two models built alike draw the same masks)
    return m


# ------------------------------------------------------------------ fused joint step (pc_joint_fused_step)
def _pc(T, P=300, seed=3, **over):
    from p_companion_amd.p_companion import PCompanion
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(P, 128, generator=g)
    torch.manual_seed(seed + 1)
    return PCompanion(cfg(NUM_TYPES=T, **over), table).to("cuda").train()


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


@pytest.mark.parametrize("T,live,dropout", [(34800, 100, 0.0), (34800, 20, 0.0), (2000, 60, 0.0), (34800, 100, 0.1), (34800, 34800, 0.1)])
def test_fused_joint_step_is_bitwise_reproducible_at_the_reference_num_types(T, live, dropout):
    """config.py:27 NUM_TYPES = 34800, without and with config.py:12's DROPOUT = 0.1 (every sample then selects its own K types:
    thousands of touched rows, summed through the sorted form): the table gradients are fixed-order sums -- two runs of the same
    step from the same state give bit-identical gradients of all ten tensors, and the in-kernel Adam update is bit-identical
    too.  (Round 2: float atomics above T = 512; rounds 3-4: above 512 touched rows per table.)"""
    from p_companion_amd.product2vec import FusedAdam
    B = 4096
    m1, m2 = _pc_big(T, dropout=dropout), _pc_big(T, dropout=dropout)
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
    m1.type_transition._dropout_step = 50
    m1.train_step(b)
    g1 = m1._gflat.clone()
    for _ in range(3):
        m1.type_transition._dropout_step = 50                 # (the same mask again)
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


@pytest.mark.parametrize("T,B", [(40, 64), (100, 250), (300, 1000), (512, 333), (513, 100), (2000, 600)])
def test_fused_joint_step_equals_launch_per_op_step_and_oracle(T, B):
    """The three-launch step against the launch-per-op sequence (pc_joint_train_step) and the oracle: losses, top-k
    (bit-exact), every gradient.  B not a multiple of the 16-sample tile, T on both sides of the 512 boundary (LDS
    similarity row + one-hot table gradients | per-distinct-query-type similarity rows + atomics)."""
    from oracle import joint_oracle
    m = _pc(T)
    st0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    b = joint_batch(B, 300, T, seed=T)
    lf, tf = m.train_step(b)
    gf = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m.use_fused_joint = False
    ll, tl = m.train_step(b)
    assert torch.equal(tf, tl)
    assert torch.allclose(lf, ll, rtol=1e-5, atol=1e-6)
    for k, p in m.named_parameters():
        if p.grad is not None:
            tol = 1e-6 + 1e-4 * float(p.grad.abs().max())
            assert float((gf[k] - p.grad).abs().max()) <= tol, k
    hb = {k: v.cpu() for k, v in b.items()}
    ref = joint_oracle.train_step({k: v.clone() for k, v in st0.items()}, hb, joint_oracle.new_moments(st0), 1)
    assert abs(float(lf[0]) - float(ref["loss"])) < 1e-5
    assert np.array_equal(tf.cpu().numpy(), ref["out"]["complementary_types"].numpy())
    for k, g in ref["grads"].items():
        assert float((gf[k].cpu() - g).abs().max()) <= 1e-6 + 1e-4 * float(g.abs().max()), k


@pytest.mark.parametrize("K", [1, 2, 4])
@pytest.mark.parametrize("T,B", [(40, 1), (40, 23), (100, 16), (300, 50), (700, 33)])
def test_fused_joint_step_other_k_and_tiny_batches(K, T, B):
    """NUM_COMP_TYPES other than the reference's 3 (the run-time-K instantiations of the tile / gradient kernels) and
    batches of a single partial tile, in all three table regimes (T <= 128: gradient products in the tile kernel;
    <= 512: the gradient-product kernel; above: per-query-type similarity rows) against the oracle: loss, top-K
    (bit-exact), every gradient; and the Adam update applied by the finish kernel against the oracle's."""
    from oracle import joint_oracle
    from p_companion_amd.product2vec import FusedAdam
    m = _pc(T, NUM_COMP_TYPES=K)
    opt = FusedAdam(m, lr=1e-3)
    st0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    b = joint_batch(B, 300, T, seed=7 * T + K)
    lf, tf = m.train_step(b, optimizer=opt)
    assert tf.shape == (B, K)
    hb = {k: v.cpu() for k, v in b.items()}
    st = {k: v.clone() for k, v in st0.items()}
    ref = joint_oracle.train_step(st, hb, joint_oracle.new_moments(st0), 1, k=K)
    assert abs(float(lf[0]) - float(ref["loss"])) < 1e-5
    assert np.array_equal(tf.cpu().numpy(), ref["out"]["complementary_types"].numpy())
    for k, p in m.named_parameters():
        if p.grad is not None:
            g = ref["grads"][k]
            assert float((p.grad.cpu() - g).abs().max()) <= 1e-6 + 1e-4 * float(g.abs().max()), k
            # one Adam step (|update| <= lr = 1e-3); an element whose gradient is ~1e-8 = eps may move differently for a
            # 1e-10 difference in that gradient, so: nearly all elements agree closely, none by more than the step itself
            d = (p.detach().cpu() - st[k]).abs()
            assert float((d <= 2e-5).float().mean()) >= 0.99 and float(d.max()) <= 2.1e-3, k


def test_fused_joint_step_is_bitwise_reproducible_and_adam_in_kernel():
    """T <= 512: no float atomic anywhere in the step -> run-to-run bit equality of every gradient; the Adam update
    applied by the finish kernel == pc_adam_step on those gradients (same arithmetic, same order)."""
    from p_companion_amd.product2vec import FusedAdam
    T, B = 100, 4096
    m1, m2 = _pc(T, P=5000), _pc(T, P=5000)
    o1, o2 = FusedAdam(m1, lr=1e-2), FusedAdam(m2, lr=1e-2)
    for s in range(4):
        b = joint_batch(B, 5000, T, seed=40 + s)
        l1, t1 = m1.train_step(b, optimizer=o1)                 # three launches, Adam inside
        l2, t2 = m2.train_step(b)                               # gradients, then the separate Adam launch
        o2.step()
        assert torch.equal(l1, l2) and torch.equal(t1, t2)
        for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            if p1.grad is not None:
                assert torch.equal(p1.grad, p2.grad), (s, k)
            assert torch.allclose(p1, p2, rtol=0, atol=1e-7), (s, k)
    assert int(o1.step_count) == int(o2.step_count) == 4
    assert torch.allclose(o1.exp_avg, o2.exp_avg, rtol=0, atol=1e-9)
    # and twice the same step from the same state: bit-identical gradients
    b = joint_batch(B, 5000, T, seed=99)
    m1.train_step(b)
    g1 = m1._gflat.clone()
    m1.train_step(b)
    assert torch.equal(g1, m1._gflat)


@pytest.mark.parametrize("T", [40, 300, 700])
def test_fused_joint_step_clamps_and_counts_bad_ids(T):
    m = _pc(T)
    b = joint_batch(48, 300, 40, seed=5)
    b["query_types"][7] = T + 1
    b["query_idx"][3] = 300
    b["negative_types"][11, 0] = -2
    losses, topk = m.train_step(b)                               # no out-of-bounds access: ids clamped in the kernel
    assert torch.isfinite(losses).all()
    assert m.index_errors() == 3
    m.train_step(joint_batch(48, 300, 40, seed=6))
    m.raise_index_errors()


def test_fused_joint_step_at_reference_num_types():
    """config.py:27 NUM_TYPES = 34800, B = 4096, 100 live query types: per-distinct-type similarity rows; top-k equals
    torch.topk of the full [B,T] product formed by the launch-per-op path, gradients agree, untouched table rows get
    exactly zero gradient."""
    T, B = 34800, 4096
    m = _pc(T, P=20000)
    g = torch.Generator().manual_seed(1)
    b = joint_batch(B, 20000, 100, seed=2)                       # types drawn from the 100 live ones
    lf, tf = m.train_step(b)
    gf = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    out = m.eval()(b)
    m.train()
    ref_top = torch.topk(out["type_similarities"], 3, dim=1).indices
    assert torch.equal(tf.long(), ref_top)
    m.use_fused_joint = False
    ll, tl = m.train_step(b)
    assert torch.equal(tf, tl) and torch.allclose(lf, ll, rtol=1e-5, atol=1e-6)
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert float((gf[k] - p.grad).abs().max()) <= 1e-6 + 1e-4 * float(p.grad.abs().max()), k
    touched = torch.zeros(T, dtype=torch.bool, device="cuda")
    touched[tf.long().reshape(-1)] = True
    touched[b["positive_types"].reshape(-1)] = True
    touched[b["negative_types"].reshape(-1)] = True
    assert float(gf["complementary_type_embeddings.weight"][~touched].abs().max()) == 0.0


# ------------------------------------------------------------------ the reference as shipped: NUM_TYPES = 34800 with DROPOUT = 0.1
@pytest.mark.parametrize("B,k,p", [(256, 3, 0.1), (250, 3, 0.1), (77, 2, 0.5)])
def test_fused_joint_step_at_reference_num_types_with_dropout_against_the_oracle(B, k, p):
    """config.py:12 DROPOUT = 0.1 + config.py:27 NUM_TYPES = 34800 (round 3 sent this to the launch-per-op path): with hidden-layer
    dropout the similarity row is formed per SAMPLE (sample_hidden_kernel, sample_sims_max_kernel, sample_topk_refine_kernel).  Loss, top-k (index-exact),
    all ten gradients, untouched table rows exactly zero and the in-kernel Adam against oracle.joint_oracle.train_step with the
    same mask as an explicit input (oracle.philox_oracle.dropout_mask restates the generator)."""
    from oracle import joint_oracle, philox_oracle
    from p_companion_amd import ops
    from p_companion_amd.p_companion import PCompanion
    from p_companion_amd.product2vec import FusedAdam
    T, P = 34800, 1000
    assert ops.joint_fused_supported(T, k, p)
    g = torch.Generator().manual_seed(5)
    table = torch.randn(P, 128, generator=g)
    torch.manual_seed(6)
    m = PCompanion(cfg(NUM_TYPES=T, DROPOUT=p, NUM_COMP_TYPES=k), table).to("cuda").train()
    opt = FusedAdam(m, lr=1e-3)
    st0 = {kk: v.detach().cpu().clone() for kk, v in m.state_dict().items()}
    b = joint_batch(B, P, 20, seed=11)                            # 20 live types (synthetic_data.py:16-17) of 34800 rows
    tt = m.type_transition
    tt._dropout_seed, tt._dropout_step = 777, 3
    hmask = torch.from_numpy(philox_oracle.dropout_mask(777, 3, philox_oracle.STREAM_HIDDEN, B * 32, p)).view(B, 32)
    lf, tf = m.train_step(b, optimizer=opt)
    assert tt._dropout_step == 4
    hb = {kk: v.cpu() for kk, v in b.items()}
    st = {kk: v.clone() for kk, v in st0.items()}
    ref = joint_oracle.train_step(st, hb, joint_oracle.new_moments(st0), 1, k=k, hidden_mask=hmask)
    plain = joint_oracle.train_step({kk: v.clone() for kk, v in st0.items()}, hb, joint_oracle.new_moments(st0), 1, k=k)
    assert abs(float(ref["loss"]) - float(plain["loss"])) > 1e-5                 # the mask matters
    assert abs(float(lf[0]) - float(ref["loss"])) < 1e-5
    assert abs(float(lf[1]) - float(ref["type_loss"])) < 1e-5 and abs(float(lf[2]) - float(ref["item_loss"])) < 1e-5
    assert np.array_equal(tf.cpu().numpy(), ref["out"]["complementary_types"].numpy())
    # samples of one query type no longer share their top-k (the per-type shortcut would have been wrong)
    qt = hb["query_types"].numpy()
    tk = tf.cpu().numpy()
    assert any(len({tuple(r) for r in tk[qt == t]}) > 1 for t in np.unique(qt))
    for kk, prm in m.named_parameters():
        if prm.grad is None:
            continue
        gr = ref["grads"][kk]
        assert float((prm.grad.cpu() - gr).abs().max()) <= 1e-6 + 1e-4 * float(gr.abs().max()), kk
        if kk.endswith("type_embeddings.weight"):
            zero_rows = gr.abs().amax(1) == 0
            assert int(zero_rows.sum()) > 33000
            assert float(prm.grad.cpu()[zero_rows].abs().max()) == 0.0, kk
        d = (prm.detach().cpu() - st[kk]).abs()
        assert float((d <= 2e-5).float().mean()) >= 0.99 and float(d.max()) <= 2.1e-3, kk
    # bitwise reproducible (no float atomic up to 512 touched rows per table)
    m2 = PCompanion(cfg(NUM_TYPES=T, DROPOUT=p, NUM_COMP_TYPES=k), table).to("cuda").train()
    m2.load_state_dict(st0)
    m2.type_transition._dropout_seed, m2.type_transition._dropout_step = 777, 3
    l2, t2 = m2.train_step(b, optimizer=FusedAdam(m2, lr=1e-3))
    assert torch.equal(l2, lf) and torch.equal(t2, tf)
    for (kk, a_), (_, b_) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(a_, b_), kk


@pytest.mark.parametrize("k", [1, 3, 4])
def test_per_sample_topk_resolves_ties_like_the_dense_path(k):
    """The per-row selection keeps the maximum of every 64-type sub-chunk and re-forms the sub-chunks whose upper bound reaches the
    K-th largest lower bound; the exact two-word keys decide among those.  A complementary table made of 12 distinct rows repeated
    over T = 2000 types makes EVERY similarity row a field of exact ties (each value ~167 times, every sub-chunk maximum equal to
    the row's best): all 32 sub-chunks are candidates and the selected types must be the lowest indices of the best groups, in
    order -- what pc_topk_rows (tie rule of torch.topk) returns on the oracle's similarity matrix."""
    from oracle import joint_oracle, philox_oracle
    from p_companion_amd import ops
    from p_companion_amd.p_companion import PCompanion
    T, P, B, p = 2000, 300, 200, 0.1
    g = torch.Generator().manual_seed(3)
    table = torch.randn(P, 128, generator=g)
    torch.manual_seed(4)
    m = PCompanion(cfg(NUM_TYPES=T, DROPOUT=p, NUM_COMP_TYPES=k), table).to("cuda").train()
    with torch.no_grad():
        base = torch.randn(12, 64, generator=g)
        m.complementary_type_embeddings.weight.copy_(base[torch.arange(T) % 12].cuda())
    st0 = {kk: v.detach().cpu().clone() for kk, v in m.state_dict().items()}
    b = joint_batch(B, P, 50, seed=2)
    tt = m.type_transition
    tt._dropout_seed, tt._dropout_step = 99, 0
    hmask = torch.from_numpy(philox_oracle.dropout_mask(99, 0, philox_oracle.STREAM_HIDDEN, B * 32, p)).view(B, 32)
    _, tf = m.train_step(b)
    ref = joint_oracle.forward(st0, b["query_idx"].cpu(), b["query_types"].cpu(), k, hidden_mask=hmask)
    want = ops.topk_rows(ref["type_similarities"].contiguous().cuda(), k).cpu().numpy()
    got = tf.cpu().numpy()
    assert np.array_equal(got, want), f"{(got != want).any(1).sum()} of {B} rows differ"
    assert (got < 12 * k + 12).all()                               # the winners are the first members of their groups


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_selection_among_sub_chunks_within_the_error_bound_is_right_to_rounding(p):
    """Pass 1 of the T > 512 selection knows a sub-chunk's maximum only to within its error bound (two bf16 pieces per operand); pass 2
    must then look at EVERY sub-chunk that could hold one of the K best.  A complementary table of 12 base rows repeated over
    T = 2000 types, each copy scaled by 1 + d with |d| <= 3e-5, puts all 32 sub-chunk maxima of a row within that bound of each
    other while every similarity is a distinct number: the selected types must be the K best of the oracle's similarity row up to
    the rounding of the two fp32 summation orders (their oracle values within 2e-6 relative of the oracle's own K best, in
    descending order), for rows = samples (dropout) and rows = distinct query types (no dropout)."""
    from oracle import joint_oracle, philox_oracle
    from p_companion_amd.p_companion import PCompanion
    T, P, B, k = 2000, 300, 200, 3
    g = torch.Generator().manual_seed(13)
    table = torch.randn(P, 128, generator=g)
    torch.manual_seed(14)
    m = PCompanion(cfg(NUM_TYPES=T, DROPOUT=p, NUM_COMP_TYPES=k), table).to("cuda").train()
    with torch.no_grad():
        base = torch.randn(12, 64, generator=g)
        scale = 1.0 + (torch.rand(T, 1, generator=g) * 2 - 1) * 3e-5
        m.complementary_type_embeddings.weight.copy_((base[torch.arange(T) % 12] * scale).cuda())
    st0 = {kk: v.detach().cpu().clone() for kk, v in m.state_dict().items()}
    b = joint_batch(B, P, 50, seed=2)
    tt = m.type_transition
    tt._dropout_seed, tt._dropout_step = 99, 0
    hmask = torch.from_numpy(philox_oracle.dropout_mask(99, 0, philox_oracle.STREAM_HIDDEN, B * 32, p)).view(B, 32) if p > 0 else None
    _, tf = m.train_step(b)
    ref = joint_oracle.forward(st0, b["query_idx"].cpu(), b["query_types"].cpu(), k, hidden_mask=hmask)
    sims = ref["type_similarities"].double()
    got = tf.cpu().long()
    assert got.shape == (B, k) and int(got.min()) >= 0 and int(got.max()) < T
    assert all(len(set(r.tolist())) == k for r in got)                               # K different types per row
    best = sims.topk(k, dim=1).values                                                # the oracle's own K best, descending
    mine = sims.gather(1, got)
    tol = 2e-6 * sims.abs().amax(1, keepdim=True)
    assert bool((mine >= best - tol).all()), float((best - mine).max())
    assert bool((mine[:, :-1] >= mine[:, 1:] - tol).all())                           # in descending order (to rounding)
    # and the field really is inside pass 1's error bound: the 32 sub-chunk maxima of a row spread over less than 1e-4 of its scale
    sub_max = sims[:, :1984].view(B, 31, 64).amax(2)
    assert float(((sub_max.amax(1) - sub_max.amin(1)) / sims.abs().amax(1)).max()) < 1e-4


@pytest.mark.parametrize("p", [0.0, 0.1])
def test_selection_leaves_indices_in_range_when_every_similarity_is_nan(p):
    """A diverged model (NaN weights) must not turn into an out-of-bounds gather: with no candidate at all the per-row selection
    falls back to the first K types, the step completes and the tile kernel reads rows inside the table."""
    from p_companion_amd.p_companion import PCompanion
    T, P, B, k = 600, 300, 200, 3
    g = torch.Generator().manual_seed(1)
    torch.manual_seed(2)
    m = PCompanion(cfg(NUM_TYPES=T, DROPOUT=p, NUM_COMP_TYPES=k), torch.randn(P, 128, generator=g)).to("cuda").train()
    with torch.no_grad():
        next(q for n, q in m.named_parameters() if "type_transition" in n and q.dim() == 2).fill_(float("nan"))
    b = joint_batch(B, P, 50, seed=2)
    _, tf = m.train_step(b)
    torch.cuda.synchronize()
    tk = tf.cpu().numpy()
    assert tk.shape == (B, k) and tk.min() >= 0 and tk.max() < T
