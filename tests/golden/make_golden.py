#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING the reference in this container.

Test infrastructure only.  Imports the reference from /root/reference (read-only,
never shipped, absent on the GPU box) with a stub config, pins every RNG
(PYTHONHASHSEED=0, random.seed, torch.manual_seed), and dumps plain arrays
(.npz, no pickled objects) under tests/golden/.  Only the vectors are committed.

    PYTHONHASHSEED=0 python tests/golden/make_golden.py

Vectors:
  g1_mt19937.npz      CPython `random` stream facts the host sampler must reproduce
  g2_bpg1000.npz      the reference's synthetic BPG (seed 0) in integer form
  g3_negatives.npz    SimilarityDataset._get_negative_samples for the first 256 pairs
  g4_p2v_tiny.npz     Product2Vec train step, B=8, N=6 (2 zero-padded rows), DROPOUT=0
  g4_p2v_b256.npz     Product2Vec train step on the first 256 samples of the dataset
  g4_p2v_b256_params3.npz   parameters after 3 Adam steps on that batch
  g5_p2v_eval.npz     generate_all_embeddings on a 32-node toy graph
  g6_joint_t100.npz / g6_joint_t300.npz   P-Companion joint step (fwd, losses, grads, Adam x3)
  g7_collate.npz      collate_fn zero-padding of ragged neighbour lists
  g8_metrics.npz      Metrics.evaluate_model on one fixed batch
  g9_complementary.npz   ComplementaryDataset: pair order after random.shuffle + the 80/10/10 split for
                      train / val / test, and the integer fields / item rules of the first 256 samples of each

  g6_joint_t1000.npz  the same joint step at NUM_TYPES = 1000, B = 256 (the first reference-made
                      vector through the T > 512 kernels)
  g10_p2v_epochs.npz  the reference's OWN Product2Vec.train_model (product2vec.py:113-170) run for two epochs over
                      DataLoader(SimilarityDataset, 256, shuffle=False, collate_fn) on the g2 graph: per-step losses,
                      the negatives it drew, the final state_dict (BatchNorm buffers included) and the embedding dict
                      generate_all_embeddings returns for all 1 000 products
  g11_joint_epochs.npz   the reference's OWN train.train (train.py:16-72) run for two epochs on the g2 graph with g10's
                      embeddings as the pretrained table: pair order / split of both datasets, the order the shuffling
                      DataLoader visited the samples in, the randn_like filler rows it drew (input data), per-step
                      losses, the five metrics of Metrics.evaluate_model after each epoch (three val batches, ragged
                      last), the final parameters and Adam moments, and what best_model.pth held

    PYTHONHASHSEED=0 python tests/golden/make_golden.py g9      # only the named vectors
"""
import os
import sys

if os.environ.get("PYTHONHASHSEED") != "0":
    # set-iteration order of str keys feeds pair order, neighbour order, type_to_idx
    os.environ["PYTHONHASHSEED"] = "0"
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    os.execv(sys.executable, [sys.executable] + sys.argv)

sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

import copy
import random
import types

import numpy as np
import torch

torch.set_num_threads(4)
OUT = os.path.dirname(os.path.abspath(__file__))


def stub_config(**over):
    c = types.SimpleNamespace(
        PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4,
        DROPOUT=0.0, PRODUCT2VEC_EPOCHS=1, MARGIN=1.0, NEG_SAMPLES=5, BATCH_SIZE=256,
        LEARNING_RATE=0.001, NUM_EPOCHS=1, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=100,
        DEVICE=torch.device("cpu"), MODEL_DIR="/tmp/pcompanion_golden")
    for k, v in over.items():
        setattr(c, k, v)
    return c


def pid2int(pid):
    return int(pid[1:])


def sd_to_np(sd, prefix=""):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, {len(arrs)} arrays")


# --------------------------------------------------------------------------- G1
def g1():
    out = {}
    for s in (0, 1, 12345, 2**40 + 7):
        random.seed(s)
        tag = f"s{s}_"
        out[tag + "getrandbits10"] = np.array([random.getrandbits(10) for _ in range(64)], np.int64)
        out[tag + "getrandbits32"] = np.array([random.getrandbits(32) for _ in range(16)], np.int64)
        out[tag + "choice1000"] = np.array([random.choice(range(1000)) for _ in range(64)], np.int64)
        out[tag + "random"] = np.array([random.random() for _ in range(16)], np.float64)
        lst = list(range(32))
        random.shuffle(lst)
        out[tag + "shuffle32"] = np.array(lst, np.int64)
        out[tag + "sample_100_10"] = np.array(random.sample(range(100), 10), np.int64)
        out[tag + "randint_2_4"] = np.array([random.randint(2, 4) for _ in range(32)], np.int64)
    save("g1_mt19937.npz", **out)


# --------------------------------------------------------------------------- G2
def build_bpg(seed=0):
    from src.data.synthetic_data import SyntheticDataGenerator
    random.seed(seed)
    torch.manual_seed(seed)
    gen = SyntheticDataGenerator(stub_config())
    bpg = gen.generate_unified_bpg()
    return gen, bpg


def bpg_to_int(bpg):
    """Integer form of the reference BPG.  Iteration orders are the reference's own."""
    pids = list(bpg.nodes.keys())
    assert [pid2int(p) for p in pids] == list(range(len(pids)))
    feats = torch.stack([bpg.nodes[p]["features"] for p in pids]).numpy()
    cats = ["electronics", "clothing", "sports", "home", "office"]
    cat = np.array([cats.index(bpg.nodes[p]["category"]) for p in pids], np.int32)
    type_order = list(bpg.get_all_types())          # set iteration == type_to_idx order
    type_to_idx = {t: i for i, t in enumerate(type_order)}
    typ = np.array([type_to_idx[bpg.nodes[p]["type"]] for p in pids], np.int32)

    def pairs(seq):
        return np.array([[pid2int(a), pid2int(b)] for a, b in seq], np.int32).reshape(-1, 2)

    # co-view out-neighbour lists in the order get_neighbors() -> set iteration yields them
    rowptr = [0]
    col = []
    for p in pids:
        nb = bpg.get_neighbors(p, edge_type="co_view")
        col.extend(pid2int(n) for n in nb)
        rowptr.append(len(col))
    return dict(
        features=feats, category=cat, type_idx=typ,
        type_names=np.array(type_order),
        co_view=pairs(bpg.edges["co_view"]),
        purchase_after_view=pairs(bpg.edges["purchase_after_view"]),
        co_purchase=pairs(bpg.edges["co_purchase"]),
        similarity_pairs=pairs(bpg.similarity_pairs),
        complementary_pairs=pairs(bpg.complementary_pairs),
        cv_rowptr=np.array(rowptr, np.int32), cv_col=np.array(col, np.int32))


# --------------------------------------------------------------------------- G3
def g3(bpg):
    from src.data.data_loader import SimilarityDataset
    ds = SimilarityDataset(bpg, stub_config())
    out = {}
    for s in (0, 7):
        random.seed(s)
        negs = []
        for i in range(256):
            anchor_id, _ = ds.similar_pairs[i]
            negs.append([pid2int(n) for n in ds._get_negative_samples(anchor_id)])
        out[f"s{s}_negatives"] = np.array(negs, np.int32)
    save("g3_negatives.npz", **out)


# --------------------------------------------------------------------------- G4
def p2v_step_capture(model, batch, cfg, n_steps):
    """Mirror of Product2Vec.train_model's loop body (product2vec.py:126-164) with
    captures.  Calls the REFERENCE module for every computation."""
    import torch.nn.functional as F
    opt = torch.optim.Adam(model.parameters(), lr=cfg.LEARNING_RATE)
    cap = {}
    losses = []
    model.train()
    for step in range(n_steps):
        anchor_emb = model(batch["anchor"], batch.get("anchor_neighbors"))
        positive_emb = model(batch["positive"])
        negative_emb = model(batch["negative"])
        pos_d = F.pairwise_distance(anchor_emb, positive_emb)
        ae = anchor_emb.unsqueeze(1).expand(-1, negative_emb.size(1), -1)
        neg_d = torch.mean(F.pairwise_distance(ae, negative_emb, p=2), dim=1)
        loss = F.relu(cfg.MARGIN - pos_d + neg_d).mean()
        opt.zero_grad()
        loss.backward()
        if step == 0:
            cap["anchor_emb"] = anchor_emb.detach().numpy().copy()
            cap["positive_emb"] = positive_emb.detach().numpy()
            cap["negative_emb"] = negative_emb.detach().numpy()
            cap["pos_distance"] = pos_d.detach().numpy()
            cap["neg_distance"] = neg_d.detach().numpy()
            for n, p in model.named_parameters():
                cap["grad." + n] = p.grad.detach().numpy().copy()
            cap.update(sd_to_np({k: v for k, v in model.state_dict().items()
                                 if "running" in k or "num_batches" in k}, "bn_after1."))
        opt.step()
        losses.append(loss.item())
        if step == 0:
            cap.update(sd_to_np(model.state_dict(), "after1."))
    cap["losses"] = np.array(losses, np.float64)
    return cap


def g4(bpg, ints):
    from src.models.product2vec import Product2Vec
    from src.data.data_loader import SimilarityDataset, collate_fn
    cfg = stub_config()

    # ---- tiny: B=8, N=6, rows 4.. of samples 2 and 5 zero-padded
    torch.manual_seed(100)
    model = Product2Vec(cfg)
    init = sd_to_np(model.state_dict(), "init.")
    g = torch.Generator().manual_seed(101)
    B, N = 8, 6
    batch = {
        "anchor": torch.randn(B, 128, generator=g),
        "positive": torch.randn(B, 128, generator=g),
        "negative": torch.randn(B, 5, 128, generator=g),
        "anchor_neighbors": torch.randn(B, N, 128, generator=g),
    }
    batch["anchor_neighbors"][2, 4:] = 0.0
    batch["anchor_neighbors"][5, 3:] = 0.0
    cap = p2v_step_capture(model, batch, cfg, 3)
    for k in list(cap):
        if k.startswith("after1.") and not (k.endswith("ffn.5.weight") or k.endswith("ffn.1.weight")
                                             or k.endswith("out_proj.weight") or "running" in k):
            cap.pop(k)
    cap.update(sd_to_np(model.state_dict(), "after3."))
    save("g4_p2v_tiny.npz", **init, **{"batch." + k: v.numpy() for k, v in batch.items()}, **cap)

    # ---- B=256: first 256 samples of the reference dataset, reference collate
    ds = SimilarityDataset(bpg, cfg)
    random.seed(3)
    samples = [ds[i] for i in range(256)]
    batch = collate_fn(samples)
    feats = torch.from_numpy(ints["features"])
    anchor_idx = np.array([pid2int(p) for p in batch["anchor_ids"]], np.int32)
    positive_idx = np.array([pid2int(p) for p in batch["positive_id"]], np.int32)
    negative_idx = np.array([[pid2int(p) for p in row] for row in batch["negative_ids"]], np.int32)
    nmax = batch["anchor_neighbors"].shape[1]
    neighbor_idx = np.full((256, nmax), -1, np.int32)
    for i, a in enumerate(anchor_idx):
        lo, hi = ints["cv_rowptr"][a], ints["cv_rowptr"][a + 1]
        neighbor_idx[i, :hi - lo] = ints["cv_col"][lo:hi]
    # the integer form must reconstruct the reference's dense batch exactly
    zrow = torch.zeros(1, 128)
    ftab = torch.cat([feats, zrow])
    assert torch.equal(ftab[torch.from_numpy(neighbor_idx).long()], batch["anchor_neighbors"])
    assert torch.equal(feats[torch.from_numpy(anchor_idx).long()], batch["anchor"])
    assert torch.equal(feats[torch.from_numpy(negative_idx).long()], batch["negative"])
    torch.manual_seed(200)
    model = Product2Vec(cfg)
    init = sd_to_np(model.state_dict(), "init.")
    tb = {k: batch[k] for k in ("anchor", "positive", "negative", "anchor_neighbors")}
    cap = p2v_step_capture(model, tb, cfg, 3)
    after3 = sd_to_np(model.state_dict(), "after3.")
    after1 = {k: cap.pop(k) for k in list(cap) if k.startswith("after1.")}
    neg_full = cap.pop("negative_emb")
    cap["negative_emb_first32"] = neg_full[:32]
    save("g4_p2v_b256.npz", **init, anchor_idx=anchor_idx, positive_idx=positive_idx,
         negative_idx=negative_idx, neighbor_idx=neighbor_idx, **cap)
    save("g4_p2v_b256_params3.npz", **after3,
         **{k: v for k, v in after1.items() if k.endswith("ffn.3.weight") or "attention" in k})


# --------------------------------------------------------------------------- G5
def g5():
    from src.models.product2vec import Product2Vec
    from src.data.bpg import BehaviorProductGraph
    cfg = stub_config()
    torch.manual_seed(300)
    model = Product2Vec(cfg)
    # non-trivial running stats so eval-mode BN is exercised
    with torch.no_grad():
        model.ffn[1].running_mean.copy_(torch.randn(256) * 0.1)
        model.ffn[1].running_var.copy_(torch.rand(256) + 0.5)
    init = sd_to_np(model.state_dict(), "init.")
    g = torch.Generator().manual_seed(301)
    bpg = BehaviorProductGraph()
    P = 32
    feats = torch.randn(P, 128, generator=g)
    for i in range(P):
        bpg.add_node(f"P{i:06d}", {"features": feats[i], "type": "t", "category": "c"})
    random.seed(302)
    for i in range(P):
        if i % 5 == 4:
            continue                      # degree-0 nodes keep the plain FFN embedding
        for j in random.sample(range(P), random.randint(1, 6)):
            if j != i:
                bpg.add_edge(f"P{i:06d}", f"P{j:06d}", "co_view")
    rowptr, col = [0], []
    for i in range(P):
        nb = bpg.get_neighbors(f"P{i:06d}", edge_type="co_view")
        col.extend(pid2int(n) for n in nb)
        rowptr.append(len(col))
    import logging
    logging.disable(logging.CRITICAL)
    emb = model.generate_all_embeddings(bpg)
    E = torch.stack([emb[f"P{i:06d}"] for i in range(P)]).numpy()
    save("g5_p2v_eval.npz", **init, features=feats.numpy(), cv_rowptr=np.array(rowptr, np.int32),
         cv_col=np.array(col, np.int32), embeddings=E)


# --------------------------------------------------------------------------- G6
def g6(T, B, seed):
    from src.models.p_companion import PCompanion
    cfg = stub_config(NUM_TYPES=T)
    g = torch.Generator().manual_seed(seed)
    P = 500
    table = torch.randn(P, 128, generator=g)
    pretrained = {f"P{i:06d}": table[i] for i in range(P)}
    torch.manual_seed(seed + 1)
    model = PCompanion(cfg, pretrained)
    init = sd_to_np(model.state_dict(), "init.")
    qidx = torch.randint(0, P, (B,), generator=g)
    batch = {
        "query_ids": [f"P{int(i):06d}" for i in qidx],
        "query_types": torch.randint(0, T, (B,), generator=g),
        "positive_types": torch.randint(0, T, (B, 1), generator=g),
        "negative_types": torch.randint(0, T, (B, 1), generator=g),
        "positive_items": torch.randn(B, 128, generator=g),
        "negative_items": torch.randn(B, 128, generator=g),
    }
    opt = torch.optim.Adam(model.parameters(), lr=cfg.LEARNING_RATE)
    cap = {}
    losses = []
    model.train()
    for step in range(3):
        out = model(batch)
        loss = model.compute_loss(batch, out)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            cap["projected_embeddings"] = out["projected_embeddings"].detach().numpy()
            cap["complementary_types"] = out["complementary_types"].numpy()
            sims = out["type_similarities"].detach()
            cap["type_similarities"] = sims.numpy() if T <= 100 else sims[:, :128].numpy()
            cap["topk_values"] = torch.topk(sims, 3, dim=1).values.numpy()
            cap["type_loss"] = model._compute_type_loss(
                sims, batch["positive_types"].squeeze(-1), batch["negative_types"].squeeze(-1)).item()
            cap["item_loss"] = model._compute_item_loss(
                out["projected_embeddings"], batch["positive_items"], batch["negative_items"]).item()
            for n, p in model.named_parameters():
                if p.grad is not None:
                    cap["grad." + n] = p.grad.detach().numpy().copy()
        opt.step()
        losses.append(loss.item())
        if step in (0, 2):
            tag = f"after{step + 1}."
            cap.update(sd_to_np({k: v for k, v in model.state_dict().items()
                                 if k != "product_embeddings.weight"}, tag))
            for i, p in enumerate(opt.param_groups[0]["params"]):
                st = opt.state[p]
                if "exp_avg" not in st:
                    continue                       # frozen product table: no grad, no state
                name = [n for n, q in model.named_parameters() if q is p][0]
                cap[f"{tag}exp_avg.{name}"] = st["exp_avg"].numpy().copy()
                cap[f"{tag}exp_avg_sq.{name}"] = st["exp_avg_sq"].numpy().copy()
    cap["losses"] = np.array(losses, np.float64)
    tb = {"batch.query_idx": qidx.numpy().astype(np.int32)}
    for k in ("query_types", "positive_types", "negative_types", "positive_items", "negative_items"):
        tb["batch." + k] = batch[k].numpy()
    save(f"g6_joint_t{T}.npz", **init, **tb, **cap)


# --------------------------------------------------------------------------- G10 / G11
class _RecordingBar:
    """Stands where tqdm stands in the reference's loops (product2vec.py:125, train.py:34): hands the loader's batches on
    unchanged and keeps the running means the loop reports through set_postfix; the per-step losses are recovered from
    them in float64 (n * mean_n - (n - 1) * mean_{n-1}: the loop's own total_loss)."""
    means = None

    def __init__(self, it, desc=None, **kw):
        self.it = it

    def __enter__(self):
        self.n = 0
        return self

    def __exit__(self, *a):
        return False

    def __iter__(self):
        return iter(self.it)

    def set_postfix(self, d):
        self.n += 1
        type(self).means.append((self.n, float(d["loss"])))


def _losses_from_means(means):
    out, prev = [], 0.0
    for n, m in means:
        if n == 1:
            prev = 0.0
        out.append(n * m - prev)
        prev = n * m
    return np.array(out, np.float64)


def g10(bpg, ints, seed=1000, epochs=2):
    import logging
    import src.models.product2vec as ref_p2v
    from torch.utils.data import DataLoader
    from src.data.data_loader import SimilarityDataset, collate_fn
    logging.disable(logging.CRITICAL)
    cfg = stub_config()
    ds = SimilarityDataset(bpg, cfg)
    drawn = []

    def recording_collate(samples):
        drawn.append(np.array([[pid2int(p) for p in s["negative_ids"]] for s in samples], np.int32))
        return collate_fn(samples)

    loader = DataLoader(ds, batch_size=cfg.BATCH_SIZE, shuffle=False, num_workers=0, collate_fn=recording_collate)
    torch.manual_seed(seed)
    model = ref_p2v.Product2Vec(cfg)
    init = sd_to_np(model.state_dict(), "init.")
    opt = torch.optim.Adam(model.parameters(), lr=cfg.LEARNING_RATE)
    random.seed(seed)                                   # the negatives' stream (data_loader.py:33)
    class Bar(_RecordingBar):
        means = []
    keep, ref_p2v.tqdm = ref_p2v.tqdm, Bar
    try:
        emb = model.train_model(loader, opt, num_epochs=epochs)          # <- the reference's own function
    finally:
        ref_p2v.tqdm = keep
    losses = _losses_from_means(Bar.means)
    E = torch.stack([emb[f"P{i:06d}"] for i in range(len(emb))]).numpy()
    assert list(emb.keys()) == [f"P{i:06d}" for i in range(len(emb))]
    print("g10: %d steps, losses %.4f .. %.4f, degree-0 products %d" % (
        len(losses), losses[0], losses[-1], int((np.diff(ints["cv_rowptr"]) == 0).sum())))
    save("g10_p2v_epochs.npz", **init, **sd_to_np(model.state_dict(), "final."), seed=np.array(seed, np.int64),
         epochs=np.array(epochs, np.int64), batch_size=np.array(cfg.BATCH_SIZE, np.int64), losses=losses,
         negative_idx=np.concatenate(drawn), embeddings=E)
    return emb


def g11(bpg, ints, pretrained, seed=1100, epochs=2, T=100):
    """train.py imports src/utils/visualization.py at module level, which imports seaborn (absent here).  train.train
    never touches either; an EMPTY module object named seaborn lets `import train` through."""
    import logging
    from torch.utils.data import DataLoader, Dataset
    sys.modules.setdefault("seaborn", types.ModuleType("seaborn"))
    import train as ref_train
    from src.data.data_loader import ComplementaryDataset, collate_fn
    from src.utils.metrics import Metrics
    logging.disable(logging.CRITICAL)
    cfg = stub_config(NUM_TYPES=T, NUM_EPOCHS=epochs)
    os.makedirs(cfg.MODEL_DIR, exist_ok=True)
    best_path = os.path.join(cfg.MODEL_DIR, "best_model.pth")
    if os.path.exists(best_path):
        os.remove(best_path)
    feats = torch.from_numpy(ints["features"])
    random.seed(seed)
    torch.manual_seed(seed)
    train_ds = ComplementaryDataset(bpg, cfg, mode="train")          # back to back from one stream, as train.py:111-112
    val_ds = ComplementaryDataset(bpg, cfg, mode="val")

    class Visits(Dataset):
        def __init__(self, inner):
            self.inner, self.order = inner, []

        def __len__(self):
            return len(self.inner)

        def __getitem__(self, i):
            self.order.append(int(i))
            return self.inner[i]

    fill = {"train": [], "val": []}

    def recording(which):
        def collate(samples):
            out = collate_fn(samples)
            lab = out["label"]
            f = torch.where((lab == 1).unsqueeze(1), out["negative_items"], out["positive_items"])
            fill[which].append(f.numpy().copy())
            return out
        return collate

    tr_vis, va_vis = Visits(train_ds), Visits(val_ds)
    train_loader = DataLoader(tr_vis, batch_size=cfg.BATCH_SIZE, shuffle=True, num_workers=0,
                              collate_fn=recording("train"))
    val_loader = DataLoader(va_vis, batch_size=cfg.BATCH_SIZE, shuffle=False, num_workers=0,
                            collate_fn=recording("val"))

    made = {}

    class Bar(_RecordingBar):
        means = []

    def make_model(config, emb):
        made["model"] = ref_train.PCompanion_(config, emb)
        made["init"] = sd_to_np(made["model"].state_dict(), "init.")
        return made["model"]

    def make_adam(params, lr):
        made["opt"] = ref_train.Adam_(params, lr=lr)
        return made["opt"]

    class RecMetrics:
        per_epoch = []

        @staticmethod
        def evaluate_model(model, loader, device):
            m = Metrics.evaluate_model(model, loader, device)
            RecMetrics.per_epoch.append(dict(m))
            return m

    ref_train.PCompanion_, ref_train.Adam_ = ref_train.PCompanion, ref_train.Adam
    keep = (ref_train.tqdm, ref_train.PCompanion, ref_train.Adam, ref_train.Metrics)
    ref_train.tqdm, ref_train.PCompanion, ref_train.Adam, ref_train.Metrics = Bar, make_model, make_adam, RecMetrics
    torch.manual_seed(seed + 1)                       # model init, then the loader's shuffles and the randn_like fillers
    try:
        ref_train.train(cfg, train_loader, val_loader, pretrained)      # <- the reference's own function
    finally:
        ref_train.tqdm, ref_train.PCompanion, ref_train.Adam, ref_train.Metrics = keep
    model, opt = made["model"], made["opt"]
    losses = _losses_from_means(Bar.means)
    n_tr, n_va = len(train_ds), len(val_ds)
    assert len(tr_vis.order) == epochs * n_tr and len(va_vis.order) == epochs * n_va
    names = sorted(RecMetrics.per_epoch[0])
    out = {k: v for k, v in made["init"].items() if k != "init.product_embeddings.weight"}
    out.update(sd_to_np({k: v for k, v in model.state_dict().items() if k != "product_embeddings.weight"}, "final."))
    for p in opt.param_groups[0]["params"]:
        st = opt.state[p]
        if "exp_avg" in st:
            name = [n for n, q in model.named_parameters() if q is p][0]
            out["final.exp_avg." + name] = st["exp_avg"].numpy().copy()
            out["final.exp_avg_sq." + name] = st["exp_avg_sq"].numpy().copy()
            out["final.step." + name] = np.array(float(st["step"]))
    best = torch.load(best_path, weights_only=False)
    out.update(sd_to_np({k: v for k, v in best["model_state_dict"].items() if k != "product_embeddings.weight"}, "best."))
    pairs = lambda ds: np.array([[pid2int(q), pid2int(t), lab] for q, t, lab in ds.pairs], np.int32)
    print("g11: %d steps, losses %.4f .. %.4f; metrics %s; best epoch %d" % (
        len(losses), losses[0], losses[-1], RecMetrics.per_epoch, best["epoch"]))
    save("g11_joint_epochs.npz", **out, seed=np.array(seed, np.int64), epochs=np.array(epochs, np.int64),
         num_types=np.array(T, np.int64), batch_size=np.array(cfg.BATCH_SIZE, np.int64),
         train_pairs=pairs(train_ds), val_pairs=pairs(val_ds),
         train_order=np.array(tr_vis.order, np.int32).reshape(epochs, n_tr),
         val_order=np.array(va_vis.order, np.int32).reshape(epochs, n_va),
         train_filler=np.concatenate(fill["train"]).reshape(epochs, n_tr, -1),
         val_filler=np.concatenate(fill["val"]).reshape(epochs, n_va, -1),
         losses=losses, metric_names=np.array(names),
         metric_values=np.array([[m[k] for k in names] for m in RecMetrics.per_epoch], np.float64),
         best_epoch=np.array(best["epoch"], np.int64),
         best_metric_values=np.array([best["metrics"][k] for k in names], np.float64))


# --------------------------------------------------------------------------- G7
def g7():
    from src.data.data_loader import collate_fn
    g = torch.Generator().manual_seed(400)
    degs = [3, 1, 6, 2, 6, 4, 1, 5]
    samples = []
    for i, d in enumerate(degs):
        samples.append({
            "anchor_ids": f"P{i:06d}", "anchor": torch.randn(128, generator=g),
            "positive": torch.randn(128, generator=g), "negative": torch.randn(5, 128, generator=g),
            "positive_id": f"P{i + 100:06d}", "negative_ids": [f"P{j:06d}" for j in range(5)],
            "anchor_neighbors": torch.randn(d, 128, generator=g)})
    out = collate_fn(samples)
    flat = torch.cat([s["anchor_neighbors"] for s in samples]).numpy()
    save("g7_collate.npz", degrees=np.array(degs, np.int32), neighbor_rows=flat,
         anchor=out["anchor"].numpy(), negative=out["negative"].numpy(),
         anchor_neighbors=out["anchor_neighbors"].numpy())


# --------------------------------------------------------------------------- G8
def g8():
    from src.models.p_companion import PCompanion
    from src.utils.metrics import Metrics
    cfg = stub_config(NUM_TYPES=100)
    g = torch.Generator().manual_seed(500)
    P, B = 200, 48
    table = torch.randn(P, 128, generator=g)
    torch.manual_seed(501)
    model = PCompanion(cfg, {f"P{i:06d}": table[i] for i in range(P)})
    init = sd_to_np(model.state_dict(), "init.")
    qidx = torch.randint(0, P, (B,), generator=g)
    batch = {
        "query_ids": [f"P{int(i):06d}" for i in qidx],
        "query_types": torch.randint(0, 100, (B,), generator=g),
        "positive_items": torch.randn(B, 128, generator=g),
        "target_features": torch.randn(B, 128, generator=g),
    }
    m = Metrics.evaluate_model(model, [batch], torch.device("cpu"))
    save("g8_metrics.npz", **init, query_idx=qidx.numpy().astype(np.int32),
         query_types=batch["query_types"].numpy(), positive_items=batch["positive_items"].numpy(),
         target_features=batch["target_features"].numpy(),
         metric_names=np.array(sorted(m)), metric_values=np.array([m[k] for k in sorted(m)], np.float64))


# --------------------------------------------------------------------------- G9
def g9(bpg, ints):
    """ComplementaryDataset (data_loader.py:90-157) on the reference's own graph (= g2): `pairs` after
    _create_product_pairs' random.shuffle and split, per mode, from a fresh random.seed(s); and for the first 256
    samples of each mode what __getitem__ returns: the integer fields, and which of positive_items / negative_items
    IS the target's feature row (the other one is torch.randn_like filler -- input data, not pinned)."""
    from src.data.data_loader import ComplementaryDataset
    import logging
    logging.disable(logging.CRITICAL)
    cfg = stub_config()
    feats = torch.from_numpy(ints["features"])
    out = {}
    for s in (0, 11):
        for mode in ("train", "val", "test"):
            random.seed(s)
            torch.manual_seed(s)
            ds = ComplementaryDataset(bpg, cfg, mode)
            tag = f"s{s}_{mode}_"
            out[tag + "pairs"] = np.array([[pid2int(q), pid2int(t), lab] for q, t, lab in ds.pairs], np.int32)
            n = min(256, len(ds))
            rows = [ds[i] for i in range(n)]
            out[tag + "query_idx"] = np.array([pid2int(r["query_ids"]) for r in rows], np.int32)
            for k in ("query_types", "positive_types", "negative_types", "label"):
                out[tag + k] = np.array([int(r[k].reshape(-1)[0]) for r in rows], np.int32)
            tgt = out[tag + "pairs"][:n, 1]
            pos_is = np.array([bool(torch.equal(r["positive_items"], feats[t])) for r, t in zip(rows, tgt)])
            neg_is = np.array([bool(torch.equal(r["negative_items"], feats[t])) for r, t in zip(rows, tgt)])
            assert np.all(pos_is ^ neg_is)                       # exactly one of the two is the real row
            assert all(torch.equal(r["target_features"], feats[t]) for r, t in zip(rows, tgt))
            out[tag + "positive_is_target"] = pos_is
        # the type index the dataset builds (set iteration order) is the one g2 stores
        assert [ds.type_to_idx[t] for t in ints["type_names"]] == list(range(len(ints["type_names"])))
    out["n_types"] = np.array(len(ds.type_to_idx), np.int32)
    save("g9_complementary.npz", **out)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    want = lambda name: not only or name in only
    if want("g1"):
        g1()
    gen, bpg = build_bpg(0)
    ints = bpg_to_int(bpg)
    f16 = dict(ints)
    if want("g2"):
        save("g2_bpg1000.npz", **f16)
    if want("g3"):
        g3(bpg)
    if want("g4"):
        g4(bpg, ints)
    if want("g5"):
        g5()
    if want("g6"):
        g6(100, 64, 600)
        g6(300, 64, 700)
    if want("g6") or "g6_t1000" in only:
        g6(1000, 256, 800)
    if want("g7"):
        g7()
    if want("g8"):
        g8()
    if want("g9"):
        g9(bpg, ints)
    if want("g10") or want("g11"):
        emb = g10(bpg, ints)
    if want("g11"):
        g11(bpg, ints, emb)
    print("done")
