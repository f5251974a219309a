"""ORACLE (test infrastructure, not product code) -- Product2Vec path on the CPU.

A from-scratch restatement, in plain torch-CPU tensor algebra, of what the reference
computes for rows P5-P11 of SURVEY.md section 8(a).  Nothing here is imported by the
product package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may use it, and only as the checker / the timed CPU baseline.

Parity pin: tests/test_oracle_golden.py checks every function below against vectors
produced by running the reference itself (tests/golden/make_golden.py).

State is a flat dict keyed by the reference's state_dict names
(product2vec.py:14-29): ffn.0.{weight,bias}, ffn.1.{weight,bias,running_mean,
running_var,num_batches_tracked}, ffn.3.*, ffn.5.*, attention.in_proj_{weight,bias},
attention.out_proj.{weight,bias}.
"""
import math

import torch

BN_EPS = 1e-5          # nn.BatchNorm1d default (product2vec.py:16)
BN_MOMENTUM = 0.1
PAIRWISE_EPS = 1e-6    # F.pairwise_distance default (product2vec.py:137)
TRAINABLE = (
    "ffn.0.weight", "ffn.0.bias", "ffn.1.weight", "ffn.1.bias", "ffn.3.weight", "ffn.3.bias",
    "ffn.5.weight", "ffn.5.bias", "attention.in_proj_weight", "attention.in_proj_bias",
    "attention.out_proj.weight", "attention.out_proj.bias")


def init_state(seed, d=128, h=256):
    """Default torch initialisers of product2vec.py:14-29, restated: Linear =
    kaiming_uniform(a=sqrt(5)) => U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias;
    MHA in_proj = xavier_uniform, biases 0; out_proj weight as Linear, bias 0."""
    g = torch.Generator().manual_seed(seed)

    def uni(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    st = {}
    for name, (o, i) in (("ffn.0", (h, d)), ("ffn.3", (h, h)), ("ffn.5", (d, h))):
        st[name + ".weight"] = uni((o, i), 1 / math.sqrt(i))
        st[name + ".bias"] = uni((o,), 1 / math.sqrt(i))
    st["ffn.1.weight"] = torch.ones(h)
    st["ffn.1.bias"] = torch.zeros(h)
    st["ffn.1.running_mean"] = torch.zeros(h)
    st["ffn.1.running_var"] = torch.ones(h)
    st["ffn.1.num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)
    st["attention.in_proj_weight"] = uni((3 * d, d), math.sqrt(6.0 / (3 * d + d)))
    st["attention.in_proj_bias"] = torch.zeros(3 * d)
    st["attention.out_proj.weight"] = uni((d, d), 1 / math.sqrt(d))
    st["attention.out_proj.bias"] = torch.zeros(d)
    return st


def ffn(x, st, training, update_running=True):
    """get_initial_embedding on a 2-D row block (product2vec.py:31-46 -> ffn :14-21).
    x [R,128] -> [R,128].  In training mode BatchNorm uses the statistics of exactly
    these R rows (biased variance) and updates the running buffers in place."""
    h0 = x @ st["ffn.0.weight"].T + st["ffn.0.bias"]
    if training:
        mean = h0.mean(0)
        var = h0.var(0, unbiased=False)
        if update_running:
            r = h0.shape[0]
            with torch.no_grad():
                unb = var * (r / (r - 1)) if r > 1 else var
                st["ffn.1.running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
                st["ffn.1.running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unb)
                st["ffn.1.num_batches_tracked"] += 1
    else:
        mean, var = st["ffn.1.running_mean"], st["ffn.1.running_var"]
    z1 = (h0 - mean) / torch.sqrt(var + BN_EPS) * st["ffn.1.weight"] + st["ffn.1.bias"]
    a1 = torch.tanh(z1)
    a2 = torch.tanh(a1 @ st["ffn.3.weight"].T + st["ffn.3.bias"])
    return a2 @ st["ffn.5.weight"].T + st["ffn.5.bias"]


def attention(query, keys, st, heads=4, mask=None):
    """apply_attention (product2vec.py:48-68): nn.MultiheadAttention, one query token,
    keys == values, NO key-padding mask.  query [B,D], keys [B,N,D] -> [B,D].
    mask [B,heads,N] (optional): the attention-weight dropout of nn.MultiheadAttention(dropout=p) in training mode
    (F.multi_head_attention_forward: softmax, THEN dropout on the probabilities, then the weighted sum) as an explicit
    multiplier tensor (0 or 1/(1-p)) -- ATen's own mask stream cannot be reproduced, so the mask is an input here."""
    b, n, d = keys.shape
    hd = d // heads
    w, bias = st["attention.in_proj_weight"], st["attention.in_proj_bias"]
    q = query @ w[:d].T + bias[:d]
    k = keys @ w[d:2 * d].T + bias[d:2 * d]
    v = keys @ w[2 * d:].T + bias[2 * d:]
    q = q.view(b, heads, 1, hd) * (1.0 / math.sqrt(hd))
    k = k.view(b, n, heads, hd).transpose(1, 2)           # [B,H,N,hd]
    v = v.view(b, n, heads, hd).transpose(1, 2)
    p = torch.softmax((q * k).sum(-1), dim=-1)            # [B,H,N]
    if mask is not None:
        p = p * mask
    o = (p.unsqueeze(-1) * v).sum(2).reshape(b, d)        # heads concatenated
    return o @ st["attention.out_proj.weight"].T + st["attention.out_proj.bias"]


def forward(features, neighbors, st, training, attn_mask=None):
    """Product2Vec.forward (product2vec.py:70-81) for 2-D features / 3-D neighbours."""
    if features.dim() == 3:
        b, n, d = features.shape
        return ffn(features.reshape(-1, d), st, training).reshape(b, n, -1)
    emb = ffn(features, st, training)
    if neighbors is not None and neighbors.shape[0] > 0:
        b, n, d = neighbors.shape
        nb = ffn(neighbors.reshape(-1, d), st, training).reshape(b, n, -1)
        emb = attention(emb, nb, st, mask=attn_mask)
    return emb


def triplet_loss(anchor_emb, positive_emb, negative_emb, margin):
    """product2vec.py:137-154.  NB the sign: relu(margin - d+ + d-)."""
    d_pos = torch.linalg.vector_norm(anchor_emb - positive_emb + PAIRWISE_EPS, dim=-1)
    d_neg = torch.linalg.vector_norm(anchor_emb.unsqueeze(1) - negative_emb + PAIRWISE_EPS,
                                     dim=-1).mean(1)
    return torch.relu(margin - d_pos + d_neg).mean(), d_pos, d_neg


def adam_step(params, grads, moments, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (scripts/pretrain_product2vec.py:34, train.py:24);
    no weight decay, no amsgrad.  In place; `step` is the 1-based step number."""
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k in params:
        g = grads[k]
        if g is None:
            continue
        m, v = moments[k]
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m, denom, value=-lr / bc1)


def train_step(st, batch, margin, moments, step, lr=1e-3, attn_mask=None):
    """One iteration of Product2Vec.train_model's loop body (product2vec.py:126-159).
    batch: anchor [B,D], positive [B,D], negative [B,K,D], anchor_neighbors [B,N,D] or None.
    Mutates st (params, BN buffers) and moments.  Returns a dict of intermediates."""
    leaves = {k: st[k].detach().clone().requires_grad_(True) for k in TRAINABLE}
    work = dict(st)
    work.update(leaves)
    a = forward(batch["anchor"], batch.get("anchor_neighbors"), work, True, attn_mask=attn_mask)
    p = forward(batch["positive"], None, work, True)
    n = forward(batch["negative"], None, work, True)
    loss, d_pos, d_neg = triplet_loss(a, p, n, margin)
    grads = dict(zip(TRAINABLE, torch.autograd.grad(loss, [leaves[k] for k in TRAINABLE])))
    with torch.no_grad():
        adam_step({k: st[k] for k in TRAINABLE}, grads, moments, step, lr)
    return dict(loss=loss.detach(), anchor_emb=a.detach(), positive_emb=p.detach(),
                negative_emb=n.detach(), pos_distance=d_pos.detach(),
                neg_distance=d_neg.detach(), grads=grads)


def new_moments(st):
    return {k: (torch.zeros_like(st[k]), torch.zeros_like(st[k])) for k in TRAINABLE}


def gather_batch(features, anchor_idx, positive_idx, negative_idx, neighbor_idx):
    """Index form -> the dense batch collate_fn builds (data_loader.py:171-206):
    neighbour slot -1 is a zero row (the padding of :186-198)."""
    ftab = torch.cat([features, torch.zeros(1, features.shape[1])])
    li = lambda t: torch.as_tensor(t).long()
    return {"anchor": features[li(anchor_idx)], "positive": features[li(positive_idx)],
            "negative": features[li(negative_idx)],
            "anchor_neighbors": ftab[li(neighbor_idx)] if neighbor_idx is not None else None}


def generate_all_embeddings(features, cv_rowptr, cv_col, st):
    """product2vec.py:83-111, eval mode.  Pass 1: ffn(x) for every product.  Pass 2, for
    products with >=1 co-view out-neighbour: forward(emb, neighbour_features) where emb
    is ALREADY the FFN output, so the FFN runs a second time on the query (:105-108 ->
    :73) before attention over ffn(neighbour features)."""
    with torch.no_grad():
        e1 = ffn(features, st, False)
        out = e1.clone()
        e2 = ffn(e1, st, False)                 # the double application
        for i in range(features.shape[0]):
            lo, hi = int(cv_rowptr[i]), int(cv_rowptr[i + 1])
            if hi > lo:
                nb = e1[torch.as_tensor(cv_col[lo:hi]).long()]   # ffn(neighbour features)
                out[i] = attention(e2[i:i + 1], nb.unsqueeze(0), st)[0]
    return out
