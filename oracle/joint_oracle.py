"""ORACLE (test infrastructure, not product code) -- P-Companion joint step on the CPU.

Restates rows J2-J8 of SURVEY.md section 8(a) in plain torch-CPU tensor algebra:
PCompanion.forward / compute_loss (p_companion.py:45-119), ComplementaryTypeTransition
(type_transition.py:15-20), ComplementaryItemPrediction (item_prediction.py:22-40),
dense-gradient Adam over every trainable tensor (train.py:24,46-48) and
Metrics.evaluate_model (metrics.py:62-117).  Pinned by tests/golden/g6_*.npz, g8_*.npz.

State dict keys are the reference's: product_embeddings.weight (frozen),
type_transition.{encoder,decoder}.{weight,bias},
item_prediction.{type_projection,item_projection}.{weight,bias},
query_type_embeddings.weight, complementary_type_embeddings.weight.
"""
import math

import torch

TRAINABLE = (
    "type_transition.encoder.weight", "type_transition.encoder.bias",
    "type_transition.decoder.weight", "type_transition.decoder.bias",
    "item_prediction.type_projection.weight", "item_prediction.type_projection.bias",
    "item_prediction.item_projection.weight", "item_prediction.item_projection.bias",
    "query_type_embeddings.weight", "complementary_type_embeddings.weight")


def init_state(seed, table, num_types, d=128, l=64):
    g = torch.Generator().manual_seed(seed)

    def lin(o, i):
        b = 1 / math.sqrt(i)
        return ((torch.rand((o, i), generator=g) * 2 - 1) * b, (torch.rand((o,), generator=g) * 2 - 1) * b)

    st = {"product_embeddings.weight": table.clone()}
    for name, (o, i) in (("type_transition.encoder", (l // 2, l)), ("type_transition.decoder", (l, l // 2)),
                         ("item_prediction.type_projection", (d, l)),
                         ("item_prediction.item_projection", (d, d))):
        st[name + ".weight"], st[name + ".bias"] = lin(o, i)
    st["query_type_embeddings.weight"] = torch.randn(num_types, l, generator=g)
    st["complementary_type_embeddings.weight"] = torch.randn(num_types, l, generator=g)
    return st


def forward(st, query_idx, query_types, k=3, hidden_mask=None):
    """p_companion.py:45-77 with integer product ids (the str->idx map of :47-49 is host
    glue).  hidden_mask [B,32] (optional): nn.Dropout on the hidden layer (type_transition.py:17) as an explicit
    multiplier (0 or 1/(1-p)); None = dropout off."""
    q = st["product_embeddings.weight"][query_idx.long()]
    t = st["query_type_embeddings.weight"][query_types.long()]
    h = torch.relu(t @ st["type_transition.encoder.weight"].T + st["type_transition.encoder.bias"])
    if hidden_mask is not None:
        h = h * hidden_mask
    c = h @ st["type_transition.decoder.weight"].T + st["type_transition.decoder.bias"]
    ec = st["complementary_type_embeddings.weight"]
    sims = c @ ec.T
    top = torch.topk(sims, k=k, dim=1)
    ce = ec[top.indices]                                                     # [B,K,L]
    pi = q @ st["item_prediction.item_projection.weight"].T + st["item_prediction.item_projection.bias"]
    tp = ce @ st["item_prediction.type_projection.weight"].T + st["item_prediction.type_projection.bias"]
    return {"projected_embeddings": pi.unsqueeze(1) * tp, "complementary_types": top.indices,
            "type_similarities": sims}


def type_loss(sims, pos_types, neg_types, margin):
    """p_companion.py:95-103"""
    ar = torch.arange(sims.shape[0])
    return torch.clamp(margin - sims[ar, pos_types.long()] + sims[ar, neg_types.long()], min=0).mean()


def item_loss(proj, pos_items, neg_items, margin):
    """p_companion.py:105-119 -- torch.norm, no eps; sign as in the reference."""
    dp = torch.linalg.vector_norm(proj - pos_items.unsqueeze(1), dim=-1)
    dn = torch.linalg.vector_norm(proj - neg_items.unsqueeze(1), dim=-1)
    return torch.clamp(margin - dp + dn, min=0).mean()


def compute_loss(out, batch, margin, alpha):
    """p_companion.py:79-93"""
    tl = type_loss(out["type_similarities"], batch["positive_types"].reshape(-1),
                   batch["negative_types"].reshape(-1), margin)
    il = item_loss(out["projected_embeddings"], batch["positive_items"], batch["negative_items"], margin)
    return alpha * il + (1 - alpha) * tl, tl, il


def new_moments(st):
    return {k: (torch.zeros_like(st[k]), torch.zeros_like(st[k])) for k in TRAINABLE}


def train_step(st, batch, moments, step, margin=1.0, alpha=0.8, k=3, lr=1e-3, hidden_mask=None):
    """One iteration of train.train's loop body (train.py:42-48).  Dense Adam: rows of the
    type tables that received zero gradient still move once their moments are non-zero."""
    from .p2v_oracle import adam_step
    leaves = {n: st[n].detach().clone().requires_grad_(True) for n in TRAINABLE}
    work = dict(st)
    work.update(leaves)
    out = forward(work, batch["query_idx"], batch["query_types"], k, hidden_mask=hidden_mask)
    loss, tl, il = compute_loss(out, batch, margin, alpha)
    grads = dict(zip(TRAINABLE, torch.autograd.grad(loss, [leaves[n] for n in TRAINABLE])))
    with torch.no_grad():
        adam_step({n: st[n] for n in TRAINABLE}, grads, moments, step, lr)
    return dict(loss=loss.detach(), type_loss=tl.detach(), item_loss=il.detach(), grads=grads,
                out={a: b.detach() for a, b in out.items()})


def evaluate_batch(st, query_idx, query_types, positive_items, target_features, k=3):
    """Metrics.evaluate_model for one batch (metrics.py:84-113).  Quirk kept: similarities
    is [B*K, B] and ground truth is arange(B*K), so rows >= B can never hit (:95-100)."""
    with torch.no_grad():
        out = forward(st, query_idx, query_types, k)
        proj = out["projected_embeddings"]
        sims = proj.reshape(-1, proj.shape[-1]) @ target_features.T
        gt = torch.arange(sims.shape[0])
        res = {}
        for kk in (1, 3, min(10, sims.shape[1])):
            kq = min(kk, sims.shape[1])
            top = torch.topk(sims, kq, dim=1).indices
            res[f"hit@{kk}"] = (top == gt.unsqueeze(1)).any(1).float().mean().item()
        types = out["complementary_types"]
        res["type_diversity"] = torch.unique(types, dim=1).shape[1] / types.shape[1]
        res["mean_relevance"] = torch.cosine_similarity(proj, positive_items.unsqueeze(1), dim=-1).mean().item()
    return res


def recommend(proj, types, type_idx, features, n):
    """inference.py:90-118 restated with numpy: per (row) predicted type, candidates = products of that
    type in node order (bpg.get_products_by_type, bpg.py:40-43), similarities = proj @ features[cand].T,
    torch.topk(similarities, min(n, len(cand))).  Returns lists of (candidate ids, scores) per row."""
    import numpy as np
    out = []
    for r in range(proj.shape[0]):
        cand = np.nonzero(type_idx == types[r])[0]
        if cand.size == 0:
            out.append((np.zeros(0, np.int64), np.zeros(0, np.float32)))
            continue
        sims = features[cand].astype(np.float64) @ proj[r].astype(np.float64)
        k = min(n, cand.size)
        top = np.argsort(-sims, kind="stable")[:k]
        out.append((cand[top], sims[top].astype(np.float32)))
    return out
