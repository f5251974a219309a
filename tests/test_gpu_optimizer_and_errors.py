"""FusedAdam in torch.optim.Adam's checkpoint layout (train.py:63-70) and the reference's error conventions on the GPU: a BatchNorm call
group of one row, ids outside the tables, backward through an eval-mode forward, the epoch runner's IndexError and learning-rate
handling, FusedAdam under graph capture.  Needs an MI355X."""
from types import SimpleNamespace

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


def test_fused_adam_state_dict_is_torch_adams_and_resumes():
    """FusedAdam.state_dict() == what torch.optim.Adam holds after the same steps on the same gradients;
    load_state_dict() continues a run bit-for-bit; torch.optim.Adam.load_state_dict() reads the file."""
    from p_companion_amd.p_companion import PCompanion
    from p_companion_amd.product2vec import FusedAdam
    c = cfg()
    g = torch.Generator().manual_seed(1)
    table = torch.randn(300, 128, generator=g)

    def make():
        torch.manual_seed(3)
        return PCompanion(c, table).to("cuda").train()

    m1 = make()
    o1 = FusedAdam(m1, lr=1e-2)
    assert o1.state_dict()["state"] == {}                       # like torch before the first step
    m_t = make()
    o_t = torch.optim.Adam(m_t.parameters(), lr=1e-2)
    for s in range(3):
        b = joint_batch(64, 300, 40, seed=s)
        m1.train_step(b)
        o1.step()
        m_t.train_step(b)                                        # same gradients into .grad of the torch-optimised copy
        o_t.step()
    sd, sd_t = o1.state_dict(), o_t.state_dict()
    assert sorted(sd["state"]) == sorted(sd_t["state"])         # the frozen product table (index 0) holds no state
    assert 0 not in sd["state"]
    for k in sd["state"]:
        assert float(sd["state"][k]["step"]) == float(sd_t["state"][k]["step"]) == 3.0
        for n in ("exp_avg", "exp_avg_sq"):
            assert sd["state"][k][n].shape == sd_t["state"][k][n].shape
            assert torch.allclose(sd["state"][k][n], sd_t["state"][k][n], rtol=1e-4, atol=1e-7), (k, n)
    assert sd["param_groups"][0]["params"] == sd_t["param_groups"][0]["params"]
    # torch's optimizer reads the fused optimizer's file ...
    m_l = make()
    o_l = torch.optim.Adam(m_l.parameters(), lr=1e-2)
    o_l.load_state_dict(sd)
    assert torch.equal(o_l.state_dict()["state"][1]["exp_avg"].cpu(), sd["state"][1]["exp_avg"].cpu())
    # ... and a resumed fused run continues exactly like the uninterrupted one
    m2 = make()
    m2.load_state_dict(m1.state_dict())
    o2 = FusedAdam(m2, lr=1e-2)
    o2.load_state_dict(sd_t)                                     # torch's own layout is accepted
    o2.load_state_dict(sd)
    b = joint_batch(64, 300, 40, seed=9)
    m1.train_step(b); o1.step()
    m2.train_step(b); o2.step()
    assert int(o2.step_count) == 4
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p1, p2, rtol=0, atol=2e-7), k      # (type-table scatter-adds are float atomics: not bitwise)


def test_p2v_fused_adam_state_roundtrip():
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    torch.manual_seed(0)
    m = Product2Vec(cfg()).to("cuda").train()
    o = FusedAdam(m)
    g = torch.Generator().manual_seed(0)
    table = torch.randn(100, 128, generator=g).cuda()
    b = {"anchor_idx": torch.randint(0, 100, (8,), generator=g, dtype=torch.int32).cuda(),
         "positive_idx": torch.randint(0, 100, (8,), generator=g, dtype=torch.int32).cuda(),
         "negative_idx": torch.randint(0, 100, (8, 5), generator=g, dtype=torch.int32).cuda(),
         "neighbor_idx": torch.randint(-1, 100, (8, 4), generator=g, dtype=torch.int32).cuda()}
    m.train_step_indexed(table, b)
    o.step()
    sd = o.state_dict()
    assert len(sd["state"]) == 12 and sd["state"][0]["exp_avg"].shape == (256, 128)
    t = torch.optim.Adam(m.parameters())
    t.load_state_dict(sd)
    o2 = FusedAdam(m)
    o2.load_state_dict(t.state_dict())
    assert torch.equal(o2.exp_avg, o.exp_avg) and torch.equal(o2.exp_avg_sq, o.exp_avg_sq) and int(o2.step_count) == 1


def test_single_row_batch_raises_like_batchnorm():
    """A batch of ONE triplet: nn.BatchNorm1d raises in training mode (the reference would, product2vec.py:132);
    so does the fused index step (reachable through a loader with drop_last=False)."""
    from p_companion_amd.product2vec import Product2Vec
    m = Product2Vec(cfg()).to("cuda").train()
    table = torch.randn(50, 128).cuda()
    b = {"anchor_idx": torch.tensor([3], dtype=torch.int32).cuda(), "positive_idx": torch.tensor([4], dtype=torch.int32).cuda(),
         "negative_idx": torch.tensor([[5, 6, 7, 8, 9]], dtype=torch.int32).cuda(),
         "neighbor_idx": torch.tensor([[1, 2, -1]], dtype=torch.int32).cuda()}
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        m.train_step_indexed(table, b)


def test_eval_mode_forward_works_but_backward_raises():
    from p_companion_amd.product2vec import Product2Vec
    m = Product2Vec(cfg()).to("cuda").eval()
    x = torch.randn(6, 128).cuda()
    nb = torch.randn(6, 3, 128).cuda()
    with torch.no_grad():
        ref = m(x, nb)
    y = m(x, nb)                                                  # grad enabled, eval mode: inference still works
    assert torch.equal(y, ref) and y.requires_grad
    with pytest.raises(NotImplementedError, match="eval-mode"):
        y.sum().backward()
    xg = x.clone().requires_grad_(True)
    with pytest.raises(NotImplementedError):
        m.get_initial_embedding(xg).sum().backward()


def test_out_of_range_ids_are_reported():
    from p_companion_amd.p_companion import PCompanion
    c = cfg()
    m = PCompanion(c, torch.randn(100, 128)).to("cuda").train()
    b = joint_batch(32, 100, 40)
    m.train_step(b)
    assert m.index_errors() == 0
    out = m(b)
    m.compute_loss(b, out)
    m.raise_index_errors()                                        # nothing to report
    # ids inside int32 but outside the tables: counted, reported as IndexError at the next collection point.
    # (Only the validation launch is exercised: the step itself is not run on the bad batch.)
    bad = dict(b)
    bad["query_types"] = b["query_types"].clone()
    bad["query_types"][3] = 40
    bad["query_idx"] = b["query_idx"].clone()
    bad["query_idx"][0] = 100
    bad["query_idx"][1] = -1
    m._validate((bad["query_idx"], 100), (bad["query_types"].to(torch.int32), 40))
    assert m.index_errors() == 3
    m._validate((bad["query_idx"], 100))
    with pytest.raises(IndexError, match="outside the embedding tables"):
        m.raise_index_errors()
    m.raise_index_errors()                                        # the counter was cleared


def test_train_refuses_a_graph_with_more_types_than_tables(tmp_path):
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader, generate_scaled_bpg
    from p_companion_amd import train as drivers
    bpg = generate_scaled_bpg(500, 20, seed=1)
    c = cfg(NUM_TYPES=10, NUM_EPOCHS=1, MODEL_DIR=str(tmp_path))
    ld = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), 64, device="cuda")
    with pytest.raises(IndexError, match="NUM_TYPES"):
        drivers.train(c, ld, ld, torch.from_numpy(bpg.features))


# ------------------------------------------------------------------ Adam inside a captured graph
@pytest.mark.parametrize("how", ["unfused", "k5"])
def test_graph_mode_of_the_launch_per_op_step_advances_adam_like_eager(how):
    """GraphedJointStep mode 'graph' captures PCompanion.train_step(optimizer=...) -> FusedAdam.step() for the configurations
    pc_joint_fused_step does not serve.  The captured Adam must read the DEVICE step counter (a host step number baked into
    the graph would freeze the bias corrections at their capture-time value): parameters and the step counter after 2 eager
    warm-up steps + 7 replays equal 9 eager steps."""
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
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


def test_padded_flat_buffers_change_no_bit():
    """flatten_parameters(pad_multiple=7) -- what the sharded optimizer asks for at an odd world size: the flat buffers' length is
    a multiple of 7 with zeros behind the last parameter, every parameter / gradient stays a view of them, a re-flatten of a model
    that was flattened unpadded keeps its values, and three fused steps + FusedAdam give the unpadded twin's bits; the checkpoint
    layout (torch.optim.Adam's) is the same."""
    from p_companion_amd.p_companion import PCompanion
    from p_companion_amd.product2vec import FusedAdam
    T = 45
    c = cfg(NUM_TYPES=T)
    g = torch.Generator().manual_seed(1)
    table = torch.randn(300, 128, generator=g)
    b = joint_batch(96, 300, T, seed=3)
    models = []
    for pad in (None, 7):
        torch.manual_seed(9)
        m = PCompanion(c, table).to("cuda").train()
        o = FusedAdam(m, lr=1e-2)
        if pad is not None:
            m.flatten_parameters()                                 # flattened unpadded first: the padded call rebuilds the buffers
            before = {k: p.detach().clone() for k, p in m.named_parameters()}
            flat, gflat = m.flatten_parameters(pad_multiple=pad)
            n_real = sum(p.numel() for _, p in m._named_flat())
            assert n_real % pad != 0 and flat.numel() % pad == 0 and 0 < flat.numel() - n_real < pad
            assert float(flat[n_real:].abs().max()) == 0.0 and float(gflat[n_real:].abs().max()) == 0.0
            for k, p in m.named_parameters():
                assert torch.equal(p, before[k]), k
            off = 0
            for _, p in m._named_flat():
                assert p.data_ptr() == flat.data_ptr() + 4 * off and p.grad.data_ptr() == gflat.data_ptr() + 4 * off
                off += p.numel()
            assert m.flatten_parameters()[0].data_ptr() == flat.data_ptr()            # sticky: a later plain call keeps the padding
        for _ in range(3):
            m.train_step(b, optimizer=o)
        models.append((m, o))
    (m0, o0), (m1, o1) = models
    for (k, p0), (_, p1) in zip(m0.named_parameters(), m1.named_parameters()):
        assert torch.equal(p0, p1), k
    s0, s1 = o0.state_dict(), o1.state_dict()
    assert s0["state"].keys() == s1["state"].keys()
    for i in s0["state"]:
        assert torch.equal(s0["state"][i]["exp_avg"], s1["state"][i]["exp_avg"]) and float(s0["state"][i]["step"]) == float(s1["state"][i]["step"]) == 3.0


@pytest.mark.parametrize("dim", [128, 256])
def test_adam_riding_in_the_steps_last_launch_equals_the_separate_launch(dim):
    """pc_p2v_train_step_unique_adam: torch.optim.Adam's update applied by the slab reduce right behind each gradient (and by rider
    workgroups for the gradients other kernels finished) against the same step followed by FusedAdam.step() (pc_adam_step_at):
    parameters, both moments, the step counter and the losses of six steps, bit for bit; the gradient buffer holds the same
    gradients afterwards.  A sync_reduce step and a non-unique batch fall back to the optimizer's own launch."""
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    bpg = generate_scaled_bpg(20_000, 100, seed=4, dim=dim)
    table = bpg.cuda()["features"]
    c = cfg(PRODUCT_EMB_DIM=dim)
    twins = []
    for _ in range(2):
        torch.manual_seed(1)
        m = Product2Vec(c).to("cuda").train()
        twins.append((m, FusedAdam(m, lr=3e-3)))
    (m_a, o_a), (m_b, o_b) = twins
    ld_a = SimilarityIndexLoader(bpg, 512, seed=1, drop_last=True, device="cuda")
    ld_b = SimilarityIndexLoader(bpg, 512, seed=1, drop_last=True, device="cuda")
    for n, (ba, bb) in enumerate(zip(ld_a, ld_b)):
        assert "weight" in ba["neighbor_compact"]                                  # the unique-neighbour layout
        la = m_a.train_step_indexed(table, ba)
        o_a.step()
        lb = m_b.train_step_indexed(table, bb, optimizer=o_b)                      # no o_b.step(): the step applied it
        assert torch.equal(la, lb), n
        assert torch.equal(m_a.flatten_parameters()[1], m_b.flatten_parameters()[1]), n       # the gradients themselves
        if n == 5:
            break
    assert torch.equal(m_a.flatten_parameters()[0], m_b.flatten_parameters()[0])
    assert torch.equal(o_a.exp_avg, o_b.exp_avg) and torch.equal(o_a.exp_avg_sq, o_b.exp_avg_sq)
    assert int(o_a.step_count) == int(o_b.step_count) == 6 and o_b._host_step == 6
    assert torch.equal(m_a.ffn[1].running_var, m_b.ffn[1].running_var)
    # the fallbacks: a batch in the plain padded layout steps through the optimizer's own launch, same result
    ld_c = SimilarityIndexLoader(bpg, 256, seed=5, drop_last=True, device="cuda", compact=False, prefetch=False, unique=False)
    bc = next(iter(ld_c))
    m_a.train_step_indexed(table, bc); o_a.step()
    m_b.train_step_indexed(table, bc, optimizer=o_b)
    assert torch.equal(m_a.flatten_parameters()[0], m_b.flatten_parameters()[0]) and int(o_b.step_count) == 7
    with pytest.raises(TypeError):
        m_a.train_step_indexed(table, bc, optimizer=o_b)                           # another module's optimizer


@pytest.mark.parametrize("dim,reuse,negatives", [(128, True, "uniform"), (128, False, "zipf"), (256, True, "uniform")])
def test_loader_made_step_rows_change_no_bit(dim, reuse, negatives):
    """pc_p2v_concat_step_rows + pc_p2v_train_step_unique_rows: the loader concatenates the step's row indices behind its builder
    (n_unique read on the device) and the step starts with Linear0, its transposed weights riding in the BatchNorm finalize
    launch -- against the step that concatenates in a launch of its own: row list, losses, gradients, parameters and optimizer
    state over six steps, bit for bit; with and without the buffer ring, with the Zipf negatives overwritten behind the builder,
    with and without the optimizer riding."""
    from p_companion_amd import ops
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    bpg = generate_scaled_bpg(20_000, 100, seed=4, dim=dim)
    table = bpg.cuda()["features"]
    c = cfg(PRODUCT_EMB_DIM=dim)
    twins = []
    for _ in range(2):
        torch.manual_seed(1)
        m = Product2Vec(c).to("cuda").train()
        twins.append((m, FusedAdam(m, lr=3e-3)))
    (m_a, o_a), (m_b, o_b) = twins
    kw = dict(seed=1, drop_last=True, device="cuda", reuse_buffers=reuse, negatives=negatives)
    ld_a = SimilarityIndexLoader(bpg, 512, **kw)
    ld_a.step_rows = False
    ld_b = SimilarityIndexLoader(bpg, 512, **kw)
    for n, (ba, bb) in enumerate(zip(ld_a, ld_b)):
        nbc = bb["neighbor_compact"]
        assert "step_rows" in nbc and "step_rows" not in ba["neighbor_compact"]
        nu = int(nbc["n_unique"])
        want = torch.cat([bb["anchor_idx"], nbc["nb_rows"][:nu + 1], bb["positive_idx"], bb["negative_idx"].reshape(-1)])
        assert torch.equal(nbc["step_rows"][:want.numel()], want), n
        assert torch.equal(ba["negative_idx"], bb["negative_idx"])
        riding = n % 2 == 0                                                        # both entries: adam NULL and not
        la = m_a.train_step_indexed(table, ba, optimizer=o_a if riding else None)
        lb = m_b.train_step_indexed(table, bb, optimizer=o_b if riding else None)
        if not riding:
            o_a.step(); o_b.step()
        assert torch.equal(la, lb), n
        assert torch.equal(m_a.flatten_parameters()[1], m_b.flatten_parameters()[1]), n
        if n == 5:
            break
    assert torch.equal(m_a.flatten_parameters()[0], m_b.flatten_parameters()[0])
    assert torch.equal(o_a.exp_avg, o_b.exp_avg) and torch.equal(o_a.exp_avg_sq, o_b.exp_avg_sq)
    assert torch.equal(m_a.ffn[1].running_mean, m_b.ffn[1].running_mean)
    assert torch.equal(m_a.ffn[1].running_var, m_b.ffn[1].running_var)
    assert int(m_a.ffn[1].num_batches_tracked) == int(m_b.ffn[1].num_batches_tracked) == 24
    # the C-ABI's refusals: a row list too short for the call, a NULL row list
    a, p_, ng = bb["anchor_idx"], bb["positive_idx"], bb["negative_idx"]
    with pytest.raises(ValueError):
        ops.concat_step_rows(a, p_, ng, nbc["nb_rows"], nbc["n_unique_dev"], out=torch.empty(16, dtype=torch.int32, device="cuda"))
    short = dict(nbc, step_rows=nbc["step_rows"][:64].contiguous())
    with pytest.raises(ValueError):
        m_b.train_step_indexed(table, dict(bb, neighbor_compact=short))


def test_module_stays_copyable_and_picklable_after_fused_steps(tmp_path):
    """The step wrapper keeps C structs over the module's buffers between steps -- beside the module, not in it: a trained module
    still deep-copies and torch.save()s whole (the reference saves state_dicts, scripts/pretrain_product2vec.py:44-49, but a
    caller may pickle the module), and the copy trains on from the same bits through its own buffers."""
    import copy
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    bpg = generate_scaled_bpg(5000, 20, seed=4)
    table = bpg.cuda()["features"]
    torch.manual_seed(1)
    m = Product2Vec(cfg()).to("cuda").train()
    o = FusedAdam(m, lr=1e-3)
    it = iter(SimilarityIndexLoader(bpg, 256, seed=1, drop_last=True, device="cuda", prefetch=False))
    b1, b2 = next(it), next(it)
    m.train_step_indexed(table, b1, optimizer=o)
    m2 = copy.deepcopy(m)
    torch.save(m, str(tmp_path / "module.pt"))
    m3 = torch.load(str(tmp_path / "module.pt"), weights_only=False)
    for other in (m2, m3):
        for (k, p), (_, q) in zip(m.named_parameters(), other.named_parameters()):
            assert torch.equal(p, q) and p.data_ptr() != q.data_ptr(), k
    l1 = m.train_step_indexed(table, b2)
    l2 = m2.train_step_indexed(table, b2)
    assert torch.equal(l1, l2) and torch.equal(m.flatten_parameters()[1], m2.flatten_parameters()[1])
    assert m.flatten_parameters()[0].data_ptr() != m2.flatten_parameters()[0].data_ptr()
