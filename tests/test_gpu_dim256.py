"""PRODUCT_EMB_DIM = 256 (BASELINE configs[4]: "100M products, dim=256"; the reference's modules take their dims from
the config, product2vec.py:14-29): FFN 256 -> 256 -> 256 -> 256, attention embed 256 with 4 heads of 64, the triplet loss
and the fused index step, against the oracle (which is dimension-generic) -- small cases and one that reaches the
large-tile kernels (paired column tiles, full-tile weight gradients incl. the BatchNorm-backward-on-load variant at
256 input columns, the split K|V gradient).  Needs an MI355X."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import p2v_oracle

D = 256


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, atol, what=""):
    a, b = torch.as_tensor(a).detach().cpu().float(), torch.as_tensor(b).detach().cpu().float()
    err = float((a - b).abs().max())
    assert err <= atol, f"{what}: max err {err:.3e} > {atol:.1e}"


def _state(seed=11):
    st = p2v_oracle.init_state(seed, d=D)
    st["ffn.1.weight"] = 1.0 + 0.1 * rnd(256, seed=seed + 1)
    st["ffn.1.bias"] = 0.1 * rnd(256, seed=seed + 2)
    st["attention.in_proj_bias"] = 0.05 * rnd(3 * D, seed=seed + 3)
    st["attention.out_proj.bias"] = 0.05 * rnd(D, seed=seed + 4)
    return st


@pytest.mark.parametrize("B,N", [(1, 1), (9, 7), (70, 33)])
def test_attention_dim256(B, N):
    from p_companion_amd import ops
    st = _state()
    q, kv, dout = rnd(B, D, seed=50), rnd(B, N, D, seed=51), rnd(B, D, seed=52)
    names = [k for k in p2v_oracle.TRAINABLE if k.startswith("attention")]
    leaves = {k: st[k].clone().requires_grad_(True) for k in names}
    work = dict(st); work.update(leaves)
    qi, ki = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    ref = p2v_oracle.attention(qi, ki, work)
    (ref * dout).sum().backward()
    dst = {k: v.clone().cuda() for k, v in st.items()}
    out, sv = ops.attention_forward(dst, q.cuda(), kv.cuda())
    close(out, ref, 5e-6, "attention out")
    grads, dq, dk = ops.attention_backward(dst, q.cuda(), kv.cuda(), dout.cuda(), sv)
    close(dq, qi.grad, 2e-5, "dquery")
    close(dk, ki.grad, 2e-5, "dkeys")
    for k in names:
        close(grads[k], leaves[k].grad, 4e-5 * max(1.0, float(leaves[k].grad.abs().max())), k)


@pytest.mark.parametrize("starts,rows", [([0], 37), ([0, 8, 56, 64], 104), ([0, 9001, 24001], 25704)])
def test_ffn_dim256(starts, rows):
    from p_companion_amd import ops
    st = _state()
    x = rnd(rows, D, seed=60)
    dy = rnd(rows, D, seed=61)
    bounds = list(starts) + [rows]
    names = [k for k in p2v_oracle.TRAINABLE if k.startswith("ffn")]
    leaves = {k: st[k].clone().requires_grad_(True) for k in names}
    ref_st = {k: v.clone() for k, v in st.items()}
    ref_st.update(leaves)
    xi = x.clone().requires_grad_(True)
    ys = [p2v_oracle.ffn(xi[bounds[i]:bounds[i + 1]], ref_st, True) for i in range(len(starts))]
    ref = torch.cat(ys)
    (ref * dy).sum().backward()
    dst = {k: v.clone().cuda() for k, v in st.items()}
    y, sv = ops.ffn_forward_train(dst, x.cuda(), None, rows, starts)
    close(y, ref, 2e-5, "ffn forward")
    close(dst["ffn.1.running_var"], ref_st["ffn.1.running_var"], 1e-5, "running_var")
    grads, dx = ops.ffn_backward(dst, x.cuda(), None, dy.cuda(), sv, need_dx=True)
    close(dx, xi.grad, 2e-6 + 2e-4 * float(xi.grad.abs().max()), "dx")
    for k in names:
        g = leaves[k].grad
        if k == "ffn.0.bias":
            continue
        close(grads[k], g, 2e-6 + 2e-4 * float(g.abs().max()), k)


def _batch(B, N, P, seed):
    g = torch.Generator().manual_seed(seed)
    nb = torch.randint(0, P, (B, N), generator=g, dtype=torch.int32)
    deg = torch.randint(1, N + 1, (B,), generator=g)
    nb[torch.arange(N)[None, :] >= deg[:, None]] = -1
    return {"anchor_idx": torch.randint(0, P, (B,), generator=g, dtype=torch.int32),
            "positive_idx": torch.randint(0, P, (B,), generator=g, dtype=torch.int32),
            "negative_idx": torch.randint(0, P, (B, 5), generator=g, dtype=torch.int32), "neighbor_idx": nb}


@pytest.mark.parametrize("B,N,P", [(24, 6, 60), (1024, 10, 3000)])
def test_product2vec_module_dim256_fused_step_vs_oracle(B, N, P):
    from p_companion_amd import ops
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=D, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                          MARGIN=1.0, DEVICE=torch.device("cuda"))
    torch.manual_seed(5)
    model = Product2Vec(cfg).to("cuda").train()
    assert model.ffn[0].weight.shape == (256, D) and model.attention.in_proj_weight.shape == (3 * D, D)
    st0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    table = rnd(P, D, seed=1)
    batch = _batch(B, N, P, 2)
    dbatch = {k: v.cuda() for k, v in batch.items()}
    dense = p2v_oracle.gather_batch(table, batch["anchor_idx"], batch["positive_idx"], batch["negative_idx"], batch["neighbor_idx"])
    st = {k: v.clone() for k, v in st0.items()}
    ref = p2v_oracle.train_step(st, dense, 1.0, p2v_oracle.new_moments(st), 1)
    for layout in ("dense", "unique"):
        model.load_state_dict(st0)
        b = dict(dbatch)
        if layout == "unique":
            b["neighbor_compact"] = ops.unique_neighbors(dbatch["neighbor_idx"])
        loss = model.train_step_indexed(table.cuda(), b)
        assert abs(float(loss) - float(ref["loss"])) < 2e-5, (layout, float(loss), float(ref["loss"]))
        for k, p in model.named_parameters():
            if k == "ffn.0.bias":
                continue
            g = ref["grads"][k]
            close(p.grad, g, 3e-6 + 3e-4 * float(g.abs().max()), f"{layout} grad {k}")
        close(model.ffn[1].running_mean, st["ffn.1.running_mean"], 1e-5, "running_mean")
    opt = FusedAdam(model)
    opt.step()
    # eval export (generate_all_embeddings' batched pass) at D = 256
    model.eval()
    with torch.no_grad():
        e = model(table[:16].cuda(), table[:16 * 3].cuda().view(16, 3, D))
    st_e = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ref_e = p2v_oracle.forward(table[:16], table[:48].view(16, 3, D), st_e, False)
    close(e, ref_e, 3e-5, "eval forward")


# ------------------------------------------------------------------ the joint step at PRODUCT_EMB_DIM = 256
def test_joint_step_dim256_matches_the_oracle():
    """p_companion.py:26-43 / item_prediction.py:11-20 take PRODUCT_EMB_DIM from config: at 256 (BASELINE configs[4]) the
    joint step runs through the per-op module path (pc_linear_*, pc_topk_rows, pc_hadamard_*_dim, pc_joint_loss_dim) --
    forward outputs, the three losses, top-k (bit-exact), every gradient and the parameters after one Adam step against
    the oracle; metrics and retrieval accept the 256-wide rows."""
    from types import SimpleNamespace
    from oracle import joint_oracle
    from p_companion_amd import ops
    from p_companion_amd.p_companion import PCompanion
    from p_companion_amd.product2vec import FusedAdam
    D, T, P, B, K = 256, 60, 500, 96, 3
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=D, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0,
                          ALPHA=0.8, NUM_COMP_TYPES=K, NUM_TYPES=T, DEVICE=torch.device("cuda"), LEARNING_RATE=1e-3)
    g = torch.Generator().manual_seed(0)
    table = torch.randn(P, D, generator=g)
    torch.manual_seed(1)
    m = PCompanion(cfg, table).to("cuda").train()
    assert not m.use_fused_joint
    import pytest
    from p_companion_amd.p_companion import GraphedJointStep
    with pytest.raises(ValueError, match="PRODUCT_EMB_DIM"):          # the fixed-buffer / fused / epoch forms are 128-wide: said, not guessed
        GraphedJointStep(m, FusedAdam(m), B)
    st0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    b = {"query_idx": torch.randint(0, P, (B,), generator=g, dtype=torch.int32).cuda(),
         "query_types": torch.randint(0, T, (B,), generator=g).cuda(),
         "positive_types": torch.randint(0, T, (B, 1), generator=g).cuda(),
         "negative_types": torch.randint(0, T, (B, 1), generator=g).cuda(),
         "positive_items": torch.randn(B, D, generator=g).cuda(), "negative_items": torch.randn(B, D, generator=g).cuda()}
    hb = {k: v.cpu() for k, v in b.items()}
    st = {k: v.clone() for k, v in st0.items()}
    ref = joint_oracle.train_step(st, hb, joint_oracle.new_moments(st0), 1)
    # the reference's loop body, module surface
    out = m(b)
    assert out["projected_embeddings"].shape == (B, K, D)
    assert np.array_equal(out["complementary_types"].cpu().numpy(), ref["out"]["complementary_types"].numpy())
    assert float((out["type_similarities"].detach().cpu() - ref["out"]["type_similarities"]).abs().max()) < 2e-5
    assert float((out["projected_embeddings"].detach().cpu() - ref["out"]["projected_embeddings"]).abs().max()) < 2e-4
    loss = m.compute_loss(b, out)
    assert abs(float(loss) - float(ref["loss"])) < 1e-4
    # the fused-interface loop body (train_step + optimizer)
    opt = FusedAdam(m, lr=1e-3)
    losses, topk = m.train_step(b, optimizer=opt)
    assert abs(float(losses[0]) - float(ref["loss"])) < 1e-4
    assert abs(float(losses[1]) - float(ref["type_loss"])) < 1e-4 and abs(float(losses[2]) - float(ref["item_loss"])) < 1e-4
    assert np.array_equal(topk.cpu().numpy(), ref["out"]["complementary_types"].numpy())
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        gr = ref["grads"][k]
        assert float((p.grad.cpu() - gr).abs().max()) <= 2e-6 + 2e-4 * float(gr.abs().max()), k
        d = (p.detach().cpu() - st[k]).abs()
        assert float((d <= 2e-5).float().mean()) >= 0.99 and float(d.max()) <= 2.1e-3, k
    # metrics / retrieval kernels at the 256-wide rows
    x = torch.randn(B, K, D, generator=g).cuda(); y = torch.randn(B, D, generator=g).cuda()
    cos = ops.cosine_rows(x, y).view(B, K).cpu()
    want = torch.nn.functional.cosine_similarity(x.cpu(), y.cpu()[:, None, :], dim=-1)
    assert float((cos - want).abs().max()) < 1e-5
