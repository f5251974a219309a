"""The reference-shaped modules (Product2Vec, PCompanion, ...) on the GPU against the reference's
golden vectors: state_dict compatibility, dense/autograd mode, fused mode, eval export,
device sampler.  Needs an MI355X."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import joint_oracle, p2v_oracle, philox_oracle

P2V_STATE_KEYS = ["ffn.0.weight", "ffn.0.bias", "ffn.1.weight", "ffn.1.bias", "ffn.1.running_mean",
                  "ffn.1.running_var", "ffn.1.num_batches_tracked", "ffn.3.weight", "ffn.3.bias", "ffn.5.weight",
                  "ffn.5.bias", "attention.in_proj_weight", "attention.in_proj_bias", "attention.out_proj.weight",
                  "attention.out_proj.bias"]
PC_STATE_KEYS = ["product_embeddings.weight", "type_transition.encoder.weight", "type_transition.encoder.bias",
                 "type_transition.decoder.weight", "type_transition.decoder.bias",
                 "item_prediction.type_projection.weight", "item_prediction.type_projection.bias",
                 "item_prediction.item_projection.weight", "item_prediction.item_projection.bias",
                 "query_type_embeddings.weight", "complementary_type_embeddings.weight"]


def cfg(**over):
    c = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                        MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=100, DEVICE=torch.device("cuda"),
                        LEARNING_RATE=1e-3, BATCH_SIZE=256, PRODUCT2VEC_EPOCHS=1)
    c.__dict__.update(over)
    return c


def golden_state(g, prefix="init."):
    return {k[len(prefix):]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith(prefix)}


# ------------------------------------------------------------------ Product2Vec
def test_p2v_state_dict_layout_and_load(golden):
    from p_companion_amd.product2vec import Product2Vec
    m = Product2Vec(cfg())
    assert list(m.state_dict().keys()) == P2V_STATE_KEYS
    g = golden("g4_p2v_tiny.npz")
    m.load_state_dict(golden_state(g))                              # a reference checkpoint loads as is
    torch.manual_seed(200)
    a = Product2Vec(cfg())
    ref = golden("g4_p2v_b256.npz")                                 # reference built with manual_seed(200)
    for k in ("ffn.0.weight", "ffn.5.bias", "attention.in_proj_weight", "attention.out_proj.weight"):
        assert np.array_equal(a.state_dict()[k].numpy(), ref["init." + k])   # same default initialisers / RNG order


def test_p2v_dense_autograd_mode_golden(golden):
    """Reference loop body (product2vec.py:130-159) run unmodified against the drop-in module:
    model(batch[...]) x3, F.pairwise_distance loss, torch.optim.Adam."""
    import torch.nn.functional as F
    from p_companion_amd.product2vec import Product2Vec
    g = golden("g4_p2v_tiny.npz")
    c = cfg()
    model = Product2Vec(c)
    model.load_state_dict(golden_state(g))
    model = model.to(c.DEVICE)
    opt = torch.optim.Adam(model.parameters(), lr=c.LEARNING_RATE)
    batch = {k[6:]: torch.from_numpy(g[k]).cuda() for k in g.files if k.startswith("batch.")}
    model.train()
    losses = []
    for step in range(3):
        anchor_emb = model(batch["anchor"], batch.get("anchor_neighbors"))
        positive_emb = model(batch["positive"])
        negative_emb = model(batch["negative"])
        pos_distance = F.pairwise_distance(anchor_emb, positive_emb)
        anchor_expanded = anchor_emb.unsqueeze(1).expand(-1, negative_emb.size(1), -1)
        neg_distance = torch.mean(F.pairwise_distance(anchor_expanded, negative_emb, p=2), dim=1)
        loss = F.relu(c.MARGIN - pos_distance + neg_distance).mean()
        opt.zero_grad()
        loss.backward()
        if step == 0:
            np.testing.assert_allclose(anchor_emb.detach().cpu(), g["anchor_emb"], atol=2e-5)
            np.testing.assert_allclose(negative_emb.detach().cpu(), g["negative_emb"], atol=2e-5)
            for k, p in model.named_parameters():
                if k != "ffn.0.bias":
                    ref = g["grad." + k]
                    np.testing.assert_allclose(p.grad.cpu(), ref, atol=2e-6 + 2e-4 * np.abs(ref).max(), err_msg=k)
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g["losses"], atol=1e-4)
    assert int(model.state_dict()["ffn.1.num_batches_tracked"]) == 12
    # the module's own fused loss kernel gives the same number
    model.load_state_dict(golden_state(g))
    l2 = model.dense_loss(batch)
    assert abs(float(l2) - g["losses"][0]) < 1e-4


def test_p2v_train_model_index_loader(golden):
    """train_model (product2vec.py:113-170) over the index loader on the reference's own graph;
    parity sampler => the first batch is the golden one (same CPython stream, seed 3)."""
    from p_companion_amd.data import IntBPG, SimilarityIndexLoader
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    ints = golden("g2_bpg1000.npz")
    g = golden("g4_p2v_b256.npz")
    bpg = IntBPG.from_arrays(ints)
    loader = SimilarityIndexLoader(bpg, 256, shuffle=False, sampler="cpython", seed=3)
    first = next(iter(loader))
    assert np.array_equal(first["negative_idx"].cpu().numpy(), g["negative_idx"])        # bit-exact negatives
    assert np.array_equal(first["neighbor_idx"].cpu().numpy(), g["neighbor_idx"])
    c = cfg()
    model = Product2Vec(c)
    model.load_state_dict(golden_state(g))
    model = model.to(c.DEVICE).train()
    opt = FusedAdam(model, lr=1e-3)
    table = bpg.cuda()["features"]
    losses = []
    for _ in range(3):
        losses.append(float(model.train_step_indexed(table, first)))
        opt.step()
    np.testing.assert_allclose(losses, g["losses"], atol=1e-4)
    # full epoch through train_model + export
    loader = SimilarityIndexLoader(bpg, 256, shuffle=True, sampler="philox", seed=5)
    emb = model.train_model(loader, opt, num_epochs=1)
    assert len(emb) == 1000 and emb["P000000"].shape == (128,) and not emb["P000000"].is_cuda
    assert all(torch.isfinite(v).all() for v in list(emb.values())[:50])


def test_p2v_generate_all_embeddings_golden(golden):
    from p_companion_amd.product2vec import Product2Vec
    g = golden("g5_p2v_eval.npz")
    c = cfg()
    model = Product2Vec(c)
    model.load_state_dict(golden_state(g))
    model = model.to(c.DEVICE)
    table = model.generate_embedding_table(torch.from_numpy(g["features"]).cuda(), g["cv_rowptr"], g["cv_col"])
    np.testing.assert_allclose(table.cpu(), g["embeddings"], atol=1e-5)
    # reference-style graph object (nodes dict + get_neighbors): Dict[str, Tensor] out
    class Graph:
        def __init__(s):
            s.nodes = {f"P{i:06d}": {"features": torch.from_numpy(g["features"][i])} for i in range(32)}
        def get_neighbors(s, pid, edge_type=None):
            i = int(pid[1:])
            return [f"P{j:06d}" for j in g["cv_col"][g["cv_rowptr"][i]:g["cv_rowptr"][i + 1]]]
    d = model.generate_all_embeddings(Graph())
    np.testing.assert_allclose(torch.stack([d[f"P{i:06d}"] for i in range(32)]), g["embeddings"], atol=1e-5)


def test_p2v_error_conventions():
    from p_companion_amd.product2vec import Product2Vec
    m = Product2Vec(cfg()).cuda()
    with pytest.raises(ValueError):
        m.get_initial_embedding(torch.zeros(2, 2, 2, 128).cuda())        # product2vec.py:46
    with pytest.raises(ValueError):
        m.train(); m.get_initial_embedding(torch.zeros(1, 128).cuda())   # BatchNorm1d with one row
    with pytest.raises(TypeError):
        m(torch.zeros(4, 128))                                          # CPU tensor: no fallback


# ------------------------------------------------------------------ device sampler
def test_device_sampler_matches_oracle(golden):
    from p_companion_amd import ops
    from p_companion_amd.data import IntBPG
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    gdev = bpg.cuda()
    ids = np.arange(40, 104, dtype=np.int32)
    n_pad = int(bpg.degree(bpg.similarity_pairs[ids, 0]).max())
    a, p, ng, nb = ops.build_similarity_batch(torch.from_numpy(ids).cuda(), gdev, n_pad, 5, seed=(7 << 33) + 9, step=12345)
    ra, rp, rng_, rnb = philox_oracle.build_batch(ids, bpg.similarity_pairs, bpg.cv_rowptr, bpg.cv_col, bpg.sim_rowptr,
                                                  bpg.sim_col, 1000, n_pad, 5, (7 << 33) + 9, 12345)
    assert np.array_equal(a.cpu().numpy(), ra) and np.array_equal(p.cpu().numpy(), rp)
    assert np.array_equal(ng.cpu().numpy(), rng_)                      # bit-exact
    assert np.array_equal(nb.cpu().numpy(), rnb)


# ------------------------------------------------------------------ PCompanion
@pytest.mark.parametrize("T", [100, 300])
def test_pcompanion_module_mode_golden(golden, T):
    """train.py:42-48 run unmodified against the drop-in module (str query ids, torch Adam)."""
    from p_companion_amd.p_companion import PCompanion
    g = golden(f"g6_joint_t{T}.npz")
    c = cfg(NUM_TYPES=T)
    st = golden_state(g)
    table = st["product_embeddings.weight"]
    model = PCompanion(c, {f"P{i:06d}": table[i] for i in range(table.shape[0])})
    assert list(model.state_dict().keys()) == PC_STATE_KEYS
    model.load_state_dict(st)
    model = model.to(c.DEVICE).train()
    opt = torch.optim.Adam(model.parameters(), lr=c.LEARNING_RATE)
    batch = {k[6:]: torch.from_numpy(g[k]).cuda() for k in g.files if k.startswith("batch.")}
    batch["query_ids"] = [f"P{int(i):06d}" for i in g["batch.query_idx"]]
    del batch["query_idx"]
    losses = []
    for step in range(3):
        outputs = model(batch)
        loss = model.compute_loss(batch, outputs)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            assert outputs["complementary_types"].dtype == torch.int64
            assert np.array_equal(outputs["complementary_types"].cpu().numpy(), g["complementary_types"])
            np.testing.assert_allclose(outputs["projected_embeddings"].detach().cpu(), g["projected_embeddings"], atol=2e-5)
            sims = outputs["type_similarities"].detach().cpu().numpy()
            np.testing.assert_allclose(sims if T <= 100 else sims[:, :128], g["type_similarities"], atol=2e-5)
            tl = model._compute_type_loss(outputs["type_similarities"], batch["positive_types"].squeeze(-1),
                                          batch["negative_types"].squeeze(-1))
            il = model._compute_item_loss(outputs["projected_embeddings"], batch["positive_items"], batch["negative_items"])
            assert abs(float(tl) - float(g["type_loss"])) < 1e-5 and abs(float(il) - float(g["item_loss"])) < 1e-5
            for k, p in model.named_parameters():
                if p.requires_grad:
                    ref = g["grad." + k]
                    np.testing.assert_allclose(p.grad.cpu(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), err_msg=k)
            assert model.product_embeddings.weight.grad is None
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g["losses"], atol=1e-4)
    with pytest.raises(KeyError):
        model({**batch, "query_ids": ["P999999"] * len(batch["query_ids"])})      # p_companion.py:48


@pytest.mark.parametrize("T", [100, 300])
def test_pcompanion_fused_step_golden(golden, T):
    from p_companion_amd.p_companion import PCompanion
    from p_companion_amd.product2vec import FusedAdam
    g = golden(f"g6_joint_t{T}.npz")
    c = cfg(NUM_TYPES=T)
    st = golden_state(g)
    model = PCompanion(c, st["product_embeddings.weight"])
    model.load_state_dict(st)
    model = model.to(c.DEVICE).train()
    opt = FusedAdam(model, lr=1e-3)
    batch = {k[6:]: torch.from_numpy(g[k]).cuda() for k in g.files if k.startswith("batch.")}
    losses = []
    for step in range(3):
        ls, topk = model.train_step(batch)
        if step == 0:
            assert np.array_equal(topk.cpu().numpy(), g["complementary_types"])
            assert abs(float(ls[1]) - float(g["type_loss"])) < 1e-5 and abs(float(ls[2]) - float(g["item_loss"])) < 1e-5
            for k, p in model.named_parameters():
                if p.requires_grad:
                    ref = g["grad." + k]
                    np.testing.assert_allclose(p.grad.cpu(), ref, atol=1e-6 + 1e-4 * np.abs(ref).max(), err_msg=k)
        opt.step()
        losses.append(float(ls[0]))
        if step in (0, 2):
            for k, p in model.named_parameters():
                if p.requires_grad:
                    d = (p.detach().cpu() - torch.from_numpy(g[f"after{step + 1}.{k}"])).abs()
                    assert float(d.max()) <= 1.05e-3 * (step + 1) and float((d <= 2e-5).float().mean()) >= 0.999, k
    np.testing.assert_allclose(losses, g["losses"], atol=1e-4)


def test_joint_eval_forward_and_topk_ties():
    from p_companion_amd import ops
    sims = torch.tensor([[1.0, 3.0, 3.0, 2.0, 3.0], [0.0, -1.0, 5.0, 5.0, 4.0]]).cuda()
    idx, val = ops.topk_rows(sims, 3, want_values=True)
    assert idx.cpu().tolist() == [[1, 2, 4], [2, 3, 4]]            # descending, ties -> lower index first
    assert val.cpu().tolist() == [[3.0, 3.0, 3.0], [5.0, 5.0, 4.0]]
    big = torch.randn(7, 34800, generator=torch.Generator().manual_seed(3))
    got = ops.topk_rows(big.cuda(), 3).cpu().long()
    assert torch.equal(got, torch.topk(big, 3, dim=1).indices)


@pytest.mark.parametrize("T", [250, 2000])
def test_pcompanion_fused_step_table_gradient_paths(T):
    """The fused step forms the two type-table gradients three ways depending on the table size: as one-hot products
    through the grouped weight-gradient launch (T <= 512, T % 4 == 0: the golden tests above), from per-workgroup LDS
    copies summed in fixed order (T = 250), with hardware float atomics (T = 2000).  Each must equal the oracle's
    autograd gradients (dense IndexBackward semantics of the reference, p_companion.py:60-65,95-103)."""
    from oracle import joint_oracle
    from p_companion_amd.p_companion import PCompanion
    c = cfg(NUM_TYPES=T)
    gen = torch.Generator().manual_seed(T)
    P, B = 500, 333
    table = torch.randn(P, 128, generator=gen)
    st = joint_oracle.init_state(T + 1, table, T)
    model = PCompanion(c, table)
    model.load_state_dict(st)
    model = model.to(c.DEVICE).train()
    batch = {"query_idx": torch.randint(0, P, (B,), generator=gen).int(),
             "query_types": torch.randint(0, T, (B,), generator=gen),
             "positive_types": torch.randint(0, T, (B, 1), generator=gen),
             "negative_types": torch.randint(0, T, (B, 1), generator=gen),
             "positive_items": torch.randn(B, 128, generator=gen),
             "negative_items": torch.randn(B, 128, generator=gen)}
    ref = joint_oracle.train_step({k: v.clone() for k, v in st.items()}, batch, joint_oracle.new_moments(st), 1)
    ls, topk = model.train_step({k: v.cuda() for k, v in batch.items()})
    assert torch.equal(topk.cpu().long(), ref["out"]["complementary_types"])
    assert abs(float(ls[0]) - float(ref["loss"])) < 1e-5
    for k, p in model.named_parameters():
        if p.requires_grad:
            want = ref["grads"][k]
            got = p.grad.cpu()
            assert torch.allclose(got, want, rtol=1e-4, atol=1e-6 + 1e-5 * float(want.abs().max())), k
    # module mode (forward / compute_loss / backward through autograd) on the same batch: a NUM_TYPES that is not a
    # multiple of 4 makes the similarity product's contraction / gradient dimensions odd (zero-padded in ops.py)
    for p in model.parameters():
        if p.grad is not None:
            p.grad.zero_()
    dev_batch = {k: v.cuda() for k, v in batch.items()}
    outputs = model(dev_batch)
    loss = model.compute_loss(dev_batch, outputs)
    loss.backward()
    assert abs(float(loss) - float(ref["loss"])) < 1e-5
    for k, p in model.named_parameters():
        if p.requires_grad:
            want = ref["grads"][k]
            assert torch.allclose(p.grad.cpu(), want, rtol=1e-4, atol=1e-6 + 1e-5 * float(want.abs().max())), "module mode: " + k
    # a caller's OWN loss on the outputs (not compute_loss): forward ran as one fused launch sequence, so backward has to
    # rebuild the per-op graph (_LazyJointForward) -- gradients against torch autograd over the oracle's forward
    for p in model.parameters():
        if p.grad is not None:
            p.grad.zero_()
    outputs = model(dev_batch)
    own = outputs["projected_embeddings"].square().mean() + 0.3 * outputs["type_similarities"].tanh().mean()
    own.backward()
    leaves = {n: st[n].clone().requires_grad_(True) for n in joint_oracle.TRAINABLE}
    work = dict(st)
    work.update(leaves)
    o = joint_oracle.forward(work, batch["query_idx"], batch["query_types"], 3)
    ref_own = o["projected_embeddings"].square().mean() + 0.3 * o["type_similarities"].tanh().mean()
    want_all = torch.autograd.grad(ref_own, [leaves[n] for n in joint_oracle.TRAINABLE], allow_unused=True)
    assert abs(float(own) - float(ref_own)) < 1e-5
    for n, want in zip(joint_oracle.TRAINABLE, want_all):
        got = dict(model.named_parameters())[n].grad
        want = torch.zeros_like(st[n]) if want is None else want
        got = torch.zeros_like(want) if got is None else got.cpu()
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-7 + 1e-5 * float(want.abs().max())), "own loss: " + n


def test_p2v_forward_shapes_and_missing_neighbours():
    """product2vec.py:70-81: neighbours that are None or have size(0) == 0 are ignored (the FFN output is returned);
    :31-46 / :48-68: 1-D, 2-D and 3-D inputs keep their rank; numbers against the oracle in eval mode."""
    from p_companion_amd.product2vec import Product2Vec
    st = p2v_oracle.init_state(13)
    st["ffn.1.running_mean"] = 0.1 * torch.randn(256, generator=torch.Generator().manual_seed(1))
    st["ffn.1.running_var"] = 0.5 + torch.rand(256, generator=torch.Generator().manual_seed(2))
    m = Product2Vec(cfg())
    m.load_state_dict(st)
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(3)
    x, nb = torch.randn(6, 128, generator=g), torch.randn(6, 5, 128, generator=g)
    with torch.no_grad():
        base = m(x.cuda())
        assert torch.equal(base, m(x.cuda(), None))
        assert torch.equal(base, m(x.cuda(), torch.zeros(0, 5, 128).cuda()))
        assert torch.equal(base, m.get_initial_embedding(x.cuda()))
        np.testing.assert_allclose(base.cpu(), p2v_oracle.ffn(x, st, False), atol=5e-6)
        one = m(x[0].cuda())                                             # a single feature vector (generate_all_embeddings)
        assert one.shape == (128,)
        np.testing.assert_allclose(one.cpu(), base[0].cpu(), atol=5e-6)
        three = m.get_initial_embedding(nb.cuda())                        # [B,N,D] -> [B,N,D]
        assert three.shape == (6, 5, 128)
        full = m(x.cuda(), nb.cuda())
        np.testing.assert_allclose(full.cpu(), p2v_oracle.forward(x, nb, st, False), atol=1e-5)
        single = m.apply_attention(base[0], three[0])                     # query [D], keys [N,D] -> [D]
        assert single.shape == (128,)
        np.testing.assert_allclose(single.cpu(), full[0].cpu(), atol=1e-5)


def test_p2v_dense_loss_fused_path_equals_module_calls(golden):
    """dense_loss on a device batch takes ONE fused step (_FusedDenseLoss); the reference's loop body spelled out with
    the module's own calls (four FFN passes, attention, the hinge, autograd) must give the same loss, the same
    gradients and the same BatchNorm running statistics, on the golden tiny batch and on a ragged random one."""
    import torch.nn.functional as F
    from p_companion_amd.product2vec import Product2Vec
    g = golden("g4_p2v_tiny.npz")
    c = cfg()
    gb = {k[6:]: torch.from_numpy(g[k]).cuda() for k in g.files if k.startswith("batch.")}
    gen = torch.Generator().manual_seed(8)
    rb = {"anchor": torch.randn(37, 128, generator=gen).cuda(), "positive": torch.randn(37, 128, generator=gen).cuda(),
          "negative": torch.randn(37, 5, 128, generator=gen).cuda(),
          "anchor_neighbors": (torch.randn(37, 9, 128, generator=gen) * (torch.rand(37, 9, 1, generator=gen) > 0.3)).cuda()}
    for batch in (gb, rb):
        ms = []
        for fused in (True, False):
            m = Product2Vec(c)
            m.load_state_dict(golden_state(g))
            m = m.to(c.DEVICE).train()
            if fused:
                loss = m.dense_loss(batch)
            else:
                a = m(batch["anchor"], batch["anchor_neighbors"])
                p = m(batch["positive"])
                n = m(batch["negative"])
                dn = torch.mean(F.pairwise_distance(a.unsqueeze(1).expand(-1, n.size(1), -1), n, p=2), dim=1)
                loss = F.relu(c.MARGIN - F.pairwise_distance(a, p) + dn).mean()
            loss.backward()
            ms.append((m, float(loss)))
        (mf, lf), (mm, lm) = ms
        assert abs(lf - lm) < 1e-5
        for (k, pf), (_, pm) in zip(mf.named_parameters(), mm.named_parameters()):
            tol = 2e-6 + 2e-4 * float(pm.grad.abs().max())
            assert torch.allclose(pf.grad, pm.grad, rtol=0, atol=tol), k
        for k in ("ffn.1.running_mean", "ffn.1.running_var", "ffn.1.num_batches_tracked"):
            assert torch.allclose(mf.state_dict()[k].float(), mm.state_dict()[k].float(), atol=1e-6), k


@pytest.mark.parametrize("k", [1, 3, 4, 6])
def test_topk_rows_against_torch(k):
    """pc_topk_rows (p_companion.py:64) for every compile-time / run-time K form, vector and scalar row reads, ties
    and -inf entries: torch.topk on the CPU is the reference (ties: lower index first, as torch's CPU kernel does for
    these inputs is NOT guaranteed, so ties are checked by value and by the lowest-index rule directly)."""
    from p_companion_amd import ops
    g = torch.Generator().manual_seed(k)
    for T in (7, 64, 1001, 34800):
        x = torch.randn(13, T, generator=g)
        if T > 64:
            x[:, ::97] = float("-inf")
            x[3, 5] = x[3, 900] = 9.0                      # a tie for the top value: the lower column wins
        kk = min(k, T)
        idx, val = ops.topk_rows(x.cuda(), kk, want_values=True)
        ref = torch.topk(x, kk, dim=1)
        assert torch.equal(val.cpu(), ref.values)
        assert torch.equal(torch.gather(x, 1, idx.cpu().long()), ref.values)
        if T > 64:
            assert int(idx[3, 0]) == 5 and (kk < 2 or int(idx[3, 1]) == 900)
    allneg = torch.full((2, 40), float("-inf"))
    idx = ops.topk_rows(allneg.cuda(), min(k, 8))
    assert idx.cpu().tolist() == [list(range(min(k, 8)))] * 2      # lowest indices, as before
