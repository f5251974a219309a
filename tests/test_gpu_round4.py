"""Round-4 additions on the GPU: the library's re-entrancy across host threads and streams (ABI 5 states the side queue of
the fused Product2Vec step), FusedAdam under graph capture, bench.py's self-launching N-rank form, configs[4] at its real
shape (100 M products x 256, Zipf negatives).  Needs an MI355X."""
import hashlib
import json
import os
import subprocess
import sys
import threading
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cfg(**over):
    c = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                        MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=40, DEVICE=torch.device("cuda"),
                        LEARNING_RATE=1e-3, BATCH_SIZE=64)
    c.__dict__.update(over)
    return c


def _clone_batch(b):
    cl = lambda v: v.clone() if torch.is_tensor(v) else v
    return {k: ({kk: (int(vv) if kk == "n_unique" else cl(vv)) for kk, vv in v.items()} if isinstance(v, dict) else cl(v))
            for k, v in b.items()}


# ------------------------------------------------------------------ ABI 5: re-entrant across host threads and streams
def test_two_host_threads_step_two_models_on_two_streams_bit_equal_to_serial():
    """include/pcompanion_hip.h: "two threads may step two models on two streams of one device concurrently".  Each thread owns a
    model, an optimizer, a stream (hence its own workspaces and its own side queue of the fused step) and a list of prebuilt
    batches; losses, parameters and BatchNorm statistics after 40 steps equal the serial run's bit for bit."""
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    bpg = generate_scaled_bpg(20_000, 100, seed=3)
    table = bpg.cuda()["features"]
    steps = 40
    batches = []
    for seed in (11, 12):
        bs = []
        for b in SimilarityIndexLoader(bpg, 1024, seed=seed, drop_last=True, device="cuda"):
            bs.append(_clone_batch(b))
            if len(bs) == steps:
                break
        batches.append(bs)
    torch.cuda.synchronize()

    def models():
        # (built in the calling thread: torch.manual_seed / the initialisers draw from ONE process-wide CPU generator)
        out = []
        for which in (0, 1):
            torch.manual_seed(which)
            out.append(Product2Vec(cfg()).to("cuda").train())
        torch.cuda.synchronize()
        return out

    def run(which, stream, out, m):
        try:
            with torch.cuda.stream(stream):
                opt = FusedAdam(m, lr=1e-3)
                h = hashlib.sha256()
                losses = []
                for b in batches[which]:
                    losses.append(m.train_step_indexed(table, b))
                    opt.step()
                stream.synchronize()
                for l in losses:
                    h.update(l.cpu().numpy().tobytes())
                h.update(m.flatten_parameters()[0].detach().cpu().numpy().tobytes())
                h.update(m.ffn[1].running_mean.cpu().numpy().tobytes())
                h.update(m.ffn[1].running_var.cpu().numpy().tobytes())
                out[which] = h.hexdigest()
        except BaseException as e:                               # (surface a worker's failure in the main thread)
            out[which] = e

    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    serial = {}
    ms = models()
    run(0, s[0], serial, ms[0])
    run(1, s[1], serial, ms[1])
    assert all(isinstance(v, str) for v in serial.values()), serial
    for _ in range(2):                                            # twice: first use and re-use of the two side queues
        par = {}
        ms = models()
        th = [threading.Thread(target=run, args=(i, s[i], par, ms[i])) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(300)
        assert par == serial, (par, serial)
    assert serial[0] != serial[1]


# ------------------------------------------------------------------ Adam inside a captured graph
@pytest.mark.parametrize("how", ["unfused", "k5"])
def test_graph_mode_of_the_launch_per_op_step_advances_adam_like_eager(how):
    """GraphedJointStep mode 'graph' captures PCompanion.train_step(optimizer=...) -> FusedAdam.step() for the configurations
    pc_joint_fused_step does not serve.  The captured Adam must read the DEVICE step counter (a host step number baked into
    the graph would freeze the bias corrections at their capture-time value): parameters and the step counter after 2 eager
    warm-up steps + 7 replays equal 9 eager steps."""
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
    from tests.test_gpu_round3 import joint_batch
    k = 5 if how == "k5" else 3
    c = cfg(NUM_COMP_TYPES=k, NUM_TYPES=60)
    g = torch.Generator().manual_seed(2)
    table = torch.randn(500, 128, generator=g).cuda()
    B = 256

    def make():
        torch.manual_seed(7)
        m = PCompanion(c, table).to("cuda").train()
        if how == "unfused":
            m.use_fused_joint = False
        return m, FusedAdam(m, lr=1e-2)

    m_e, o_e = make()
    m_g, o_g = make()
    graphed = GraphedJointStep(m_g, o_g, B, warmup=2, mode="graph")
    n = 9
    for i in range(n):
        b = joint_batch(B, 500, 60, seed=100 + i)
        b["query_types"] = b["query_types"].to(torch.int32)
        b["positive_types"] = b["positive_types"].to(torch.int32)
        b["negative_types"] = b["negative_types"].to(torch.int32)
        le, _ = m_e.train_step(b, optimizer=o_e)
        lg, _ = graphed(b)
        assert torch.allclose(le, lg, rtol=1e-5, atol=1e-6), (i, le, lg)
    assert graphed.graph is not None
    torch.cuda.synchronize()
    assert int(o_e.step_count) == int(o_g.step_count) == n
    for (name, pe), (_, pg) in zip(m_e.named_parameters(), m_g.named_parameters()):
        # (the launch-per-op path's table scatter-adds use float atomics: equal up to summation order)
        assert torch.allclose(pe, pg, rtol=1e-4, atol=2e-5), (name, float((pe - pg).abs().max()))
    # the optimizer keeps reading the device counter afterwards: a checkpoint load must not resurrect the host's copy
    sd = o_g.state_dict()
    o_g.load_state_dict(sd)
    assert o_g._host_step is None and int(o_g.step_count) == n


# ------------------------------------------------------------------ bench.py --gpus N starts its N ranks itself
@pytest.mark.timeout(900)
def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's N > 1 form if it calls the script like the N = 1 one):
    the script starts two ranks itself, rank 0's line reports both.  Rehearsed on the one card of this box (PC_FORCE_DEVICE=0)
    over gloo -- two ranks cannot share a GPU under RCCL; RCCL itself is rehearsed with one rank in tests/test_gpu_rccl.py."""
    env = dict(os.environ, PC_DIST_BACKEND="gloo", PC_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PC_DIST_FORCE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--products", "20000", "--batch", "512", "--no-cpu-baseline", "--no-sustained", "--no-large",
                          "--no-dropout-legs", "--no-ref-types"], env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 1024 and line["config"]["parallelism"] == "dp2"
    rccl = dict(line["rccl"])
    exchange = rccl.pop("exchange")                     # (ABI 6: the gradient exchange through the library's slot; gloo behind it here)
    assert rccl == {"backend": "gloo", "world": 2, "ranks_seen": [0, 1], "launcher": "self"} and "gloo" in exchange
    assert "pc_joint_train_epoch_dp" in line["joint"]["config"]["launch"]
    assert line["value"] > 0 and line["joint"]["value"] > 0 and line["joint"]["config"]["parallelism"] == "dp2"
    # more ranks than GPUs, not a rehearsal: refused with a message, before any rank starts
    env.pop("PC_FORCE_DEVICE")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1)], env=env,
                         capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert bad.returncode != 0 and "visible GPUs" in bad.stderr and not bad.stdout.strip()


# ------------------------------------------------------------------ BASELINE configs[4] at its real shape
class _SparseRows:
    """sim_rowptr of a 100 M-product graph restricted to the rows one batch touches (the oracle indexes it with a and a + 1)."""

    def __init__(self, d):
        self.d = d

    def __getitem__(self, i):
        return self.d[int(i)]


@pytest.mark.timeout(1200)
def test_config4_100M_products_dim256_zipf_negatives():
    """BASELINE configs[4] on one GPU: 100 M products x 256 (102 GB of features, 1.6e9 co-view edges, ~2.8e8 similarity pairs)
    generated in HBM, Zipf(1) negatives.  (a) the batches' negatives equal oracle/philox_oracle.zipf_negatives for those
    anchors bit for bit (the oracle reads only the graph rows the batch touches, copied out of HBM); (b) the compact layout the
    loader picks at this size and the dense layout (every slot its own row) agree on loss, gradients and BatchNorm
    statistics; (c) three fused steps + Adam, run twice from the same state, end in bit-identical parameters; (d) every buffer
    the package allocated sits between intact guard bands afterwards."""
    from oracle import philox_oracle
    from p_companion_amd import ops
    from p_companion_amd.data import SimilarityIndexLoader, generate_device_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    from tests.test_gpu_soak import GuardedAllocator
    if torch.cuda.get_device_properties(0).total_memory < 200e9:
        pytest.skip("needs the 288 GB of an MI355X")
    torch.cuda.empty_cache()
    P, B, D = 100_000_000, 4096, 256
    bpg = generate_device_bpg(P, 100, seed=0, dim=D, with_complementary=False)
    g = bpg.cuda()
    table = g["features"]
    assert table.shape == (P, D) and g["cv_col"].numel() > 15 * P and bpg.n_similarity_pairs > 2.5 * P
    ga = GuardedAllocator()
    old_alloc, ops._allocator = ops._allocator, ga
    try:
        c = cfg(PRODUCT_EMB_DIM=D)

        def run():
            torch.manual_seed(0)
            m = Product2Vec(c).to("cuda").train()
            opt = FusedAdam(m, lr=1e-3)
            ld = SimilarityIndexLoader(bpg, B, seed=1, drop_last=True, negatives="zipf")
            assert not ld.unique and ld.compact
            it = iter(ld)
            kept, losses = [], []
            for _ in range(3):
                b = next(it)
                kept.append(_clone_batch(b))
                losses.append(m.train_step_indexed(table, b).clone())
                opt.step()
            ld.check_errors()
            torch.cuda.synchronize()
            return m, kept, torch.cat(losses)

        m1, kept, l1 = run()
        m2, kept2, l2 = run()
        # (c) bitwise reproducible, batches included
        assert torch.isfinite(l1).all() and torch.equal(l1, l2)
        assert torch.equal(m1.flatten_parameters()[0], m2.flatten_parameters()[0])
        assert torch.equal(m1.ffn[1].running_var, m2.ffn[1].running_var)
        for x, y in zip(kept, kept2):
            assert torch.equal(x["negative_idx"], y["negative_idx"]) and torch.equal(x["neighbor_compact"]["nb_rows"], y["neighbor_compact"]["nb_rows"])

        # (a) negatives vs the oracle, on the rows of the graph the batches touch
        ld = SimilarityIndexLoader(bpg, B, seed=1, drop_last=True, negatives="zipf")
        perm, _ = ld._epoch_plan(bpg.n_similarity_pairs)
        thr = ops.zipf_octave_thresholds(P)
        head = 0
        for step, b in enumerate(kept[:2]):
            pids = perm[step * B:(step + 1) * B].long()
            pairs = g["sim_pairs"][pids].cpu().numpy()                       # [B,2]
            assert np.array_equal(pairs[:, 0], b["anchor_idx"].cpu().numpy()) and np.array_equal(pairs[:, 1], b["positive_idx"].cpu().numpy())
            ua = np.unique(pairs[:, 0])
            ua_dev = torch.from_numpy(ua).cuda().long()
            lo = g["sim_rowptr"][ua_dev].cpu().numpy().astype(np.int64)
            hi = g["sim_rowptr"][ua_dev + 1].cpu().numpy().astype(np.int64)
            rows, off, cols = {}, 0, []
            for a, l, h in zip(ua.tolist(), lo.tolist(), hi.tolist()):      # ascending anchors, positives laid end to end
                rows[a] = off
                cols.append(g["sim_col"][l:h].cpu().numpy())
                off += h - l
                rows[a + 1] = off
            ref = philox_oracle.zipf_negatives(np.arange(B), pairs, _SparseRows(rows), np.concatenate(cols), P, 5, ld.seed, step, thr)
            got = b["negative_idx"].cpu().numpy()
            assert np.array_equal(got, ref), f"step {step}: {(got != ref).sum()} of {got.size} negatives differ"
            head += int((got < 1000).sum())
        assert 0.25 < head / (2 * B * 5) < 0.50                              # ~37 % of Zipf(1) draws fall on the 1000 most popular of 1e8

        # (b) compact vs dense layout on the first batch, from the same fresh state
        b0 = kept[0]
        nbc = b0["neighbor_compact"]
        dense_nb = nbc["nb_rows"][nbc["slot_row"].long()].contiguous()      # [B,N]: every slot its product (-1 = padding)
        res = []
        for layout in (nbc, dense_nb):
            torch.manual_seed(0)
            m = Product2Vec(c).to("cuda").train()
            loss = m.train_step_indexed(table, dict(b0, neighbor_compact=layout) if isinstance(layout, dict) else
                                        {k: v for k, v in dict(b0, neighbor_idx=layout).items() if k != "neighbor_compact"})
            res.append((float(loss), m.flatten_parameters()[1].clone(), m.ffn[1].running_var.clone(), m.ffn[1].running_mean.clone()))
        (lc, gc_, vc, mc), (ld_, gd, vd, md) = res
        assert abs(lc - float(l1[0])) == 0.0
        assert abs(lc - ld_) < 2e-6, (lc, ld_)
        assert float((gc_ - gd).abs().max()) < 2e-6 + 2e-4 * float(gd.abs().max())
        assert torch.allclose(vc, vd, atol=1e-6) and torch.allclose(mc, md, atol=1e-6)
        # (d)
        nblocks, nbytes = ga.verify()
        assert nblocks >= 6 and nbytes > 1e9
    finally:
        ops._allocator = old_alloc
        ops._ws_cache.clear()
        del table, g, bpg
        torch.cuda.empty_cache()


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
    from tests.test_gpu_round3 import joint_batch
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
    from tests.test_gpu_round3 import joint_batch
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
    from tests.test_gpu_round3 import joint_batch
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
    from tests.test_gpu_round3 import joint_batch
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
