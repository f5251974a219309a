"""Whole-epoch parity on the GPU against fixtures made by running the reference's OWN loop functions
(tests/golden/make_golden.py g10 / g11 / g6 at T = 1000):

  g10  Product2Vec.train_model (product2vec.py:113-170), two epochs over DataLoader(SimilarityDataset, 256, shuffle=False)
       on the reference's 1k-product graph -> per-step losses, final state_dict (BatchNorm buffers), the embedding dict;
  g11  train.train (train.py:16-72), two epochs with g10's embeddings -> per-step losses, Metrics.evaluate_model per
       epoch over three validation batches, final parameters + Adam moments, the best checkpoint;
  g6   one joint step at NUM_TYPES = 1000 (the T > 512 kernels).

State that is only visible over many steps is compared here: running statistics across 96 BatchNorm calls with a ragged
last batch, Adam step counts, eval-mode export on real degree-0 products, metrics averaged over batches, checkpoint choice.
Needs an MI355X."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import data_oracle, p2v_oracle


def cfg(tmp, **over):
    c = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                        MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=100, DEVICE=torch.device("cuda"),
                        LEARNING_RATE=1e-3, BATCH_SIZE=256, PRODUCT2VEC_EPOCHS=2, NUM_EPOCHS=2, MODEL_DIR=str(tmp))
    c.__dict__.update(over)
    return c


def adam_close(actual, desired, steps, tight, lr=1e-3, name=""):
    """tests/test_oracle_golden.py's rule: an element whose gradient is rounding noise may be off by a whole lr-step per
    step (Adam normalises the noise), so the worst case is bounded by steps * lr and 99.9 % must agree tightly."""
    d = (torch.as_tensor(actual).float().cpu() - torch.as_tensor(desired).float()).abs()
    assert float(d.max()) <= 1.05 * lr * steps, (name, float(d.max()))
    assert float((d <= tight).float().mean()) >= 0.999, (name, float((d > tight).float().mean()))


class Recorded:
    """A loader handed to train_model: passes the batches through and keeps the negatives they carried."""

    def __init__(self, inner):
        self.inner, self.dataset, self.negatives = inner, inner.dataset, []

    def __iter__(self):
        for b in self.inner:
            self.negatives.append(b["negative_idx"].cpu().numpy().copy())
            yield b


def check_p2v_against_g10(g, model, emb, negatives=None):
    steps = len(g["losses"])
    np.testing.assert_allclose(model.step_losses.numpy(), g["losses"], rtol=0, atol=1e-4)        # north_star: fp32 loss within 1e-4
    if negatives is not None:
        assert np.array_equal(np.concatenate(negatives), g["negative_idx"])                      # bit-exact negative-sample indices
    sd = model.state_dict()
    assert int(sd["ffn.1.num_batches_tracked"]) == int(g["final.ffn.1.num_batches_tracked"]) == 4 * steps
    np.testing.assert_allclose(sd["ffn.1.running_var"].cpu(), g["final.ffn.1.running_var"], rtol=2e-4, atol=1e-5)
    # the running mean carries ffn.0.bias, whose gradient is analytically zero: Adam walks it by rounding noise (the CPU
    # oracle differs from the reference by 3.5e-3 here for the same reason, tests/test_oracle_golden.py)
    assert float((sd["ffn.1.running_mean"].cpu() - torch.from_numpy(g["final.ffn.1.running_mean"])).abs().max()) <= steps * 1e-3
    for k in p2v_oracle.TRAINABLE:
        if k != "ffn.0.bias":
            adam_close(sd[k], g["final." + k], steps, 1e-4, name=k)
    E = torch.stack([emb[f"P{i:06d}"] for i in range(len(emb))]).numpy()
    assert E.shape == g["embeddings"].shape == (1000, 128)
    np.testing.assert_allclose(E, g["embeddings"], rtol=0, atol=1e-3)


def p2v_model(g, c):
    from p_companion_amd.product2vec import Product2Vec
    torch.manual_seed(int(g["seed"]))
    model = Product2Vec(c).to(c.DEVICE)
    for k, v in model.state_dict().items():                         # torch.manual_seed(s) yields the reference's initial weights
        assert torch.equal(v.cpu(), torch.from_numpy(g["init." + k])), k
    model.record_step_losses = True
    return model


def test_train_model_index_loader_reproduces_the_references_epochs(golden, tmp_path):
    """The build's train_model over its index loader (CPython-stream negatives in the reference's consumption order,
    dataset order, ragged last batch of 133) and the fused step + FusedAdam, against the reference's run."""
    from p_companion_amd.data import IntBPG, SimilarityIndexLoader
    from p_companion_amd.product2vec import FusedAdam
    g = golden("g10_p2v_epochs.npz")
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    c = cfg(tmp_path)
    model = p2v_model(g, c)
    loader = Recorded(SimilarityIndexLoader(bpg, int(g["batch_size"]), shuffle=False, sampler="cpython", seed=int(g["seed"]),
                                            device="cuda"))
    emb = model.train_model(loader, FusedAdam(model, lr=c.LEARNING_RATE), num_epochs=int(g["epochs"]))
    assert list(emb) == [f"P{i:06d}" for i in range(1000)]
    check_p2v_against_g10(g, model, emb, loader.negatives)
    # the eval-mode export alone, from the REFERENCE's final weights: 32 products without out-neighbours keep ffn(x)
    model.load_state_dict({k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("final.")})
    table = model.generate_embedding_table(bpg.cuda("cuda")["features"], bpg.cv_rowptr, bpg.cv_col).cpu().numpy()
    np.testing.assert_allclose(table, g["embeddings"], rtol=0, atol=2e-5)
    assert int((np.diff(bpg.cv_rowptr) == 0).sum()) == 32


@pytest.mark.parametrize("optimizer", ["torch", "fused"])
def test_train_model_reference_format_batches_reproduce_the_references_epochs(golden, tmp_path, optimizer):
    """The literal plug-in surface: dense batches in the reference's collate format (host tensors, string ids) through
    train_model's dense branch -- model(...) four times / the fused dense step, loss.backward(), torch.optim.Adam."""
    from p_companion_amd.data import IntBPG
    from p_companion_amd.product2vec import FusedAdam
    from oracle.mt import Random
    g = golden("g10_p2v_epochs.npz")
    ints = golden("g2_bpg1000.npz")
    bpg = IntBPG.from_arrays(ints)
    c = cfg(tmp_path)
    model = p2v_model(g, c)
    feats = torch.from_numpy(ints["features"])
    pairs, B, rng = ints["similarity_pairs"], int(g["batch_size"]), Random(int(g["seed"]))

    class Loader:
        dataset = SimpleNamespace(bpg=bpg)

        def __iter__(self):
            for lo in range(0, len(pairs), B):
                ids = np.arange(lo, min(lo + B, len(pairs)))
                b = data_oracle.similarity_batch(ints, ids, rng.negative_samples(1000, pairs, pairs[ids, 0], 5))
                d = p2v_oracle.gather_batch(feats, b["anchor_idx"], b["positive_idx"], b["negative_idx"], b["neighbor_idx"])
                d["anchor_ids"] = [f"P{i:06d}" for i in b["anchor_idx"]]
                yield d

    opt = torch.optim.Adam(model.parameters(), lr=c.LEARNING_RATE) if optimizer == "torch" else FusedAdam(model, lr=c.LEARNING_RATE)
    emb = model.train_model(Loader(), opt, num_epochs=int(g["epochs"]))
    check_p2v_against_g10(g, model, emb)


# --------------------------------------------------------------------------------------------------------- g11
def joint_batches(ints, g, which, epoch, device):
    pairs, order, filler = g[which + "_pairs"], g[which + "_order"][epoch], g[which + "_filler"][epoch]
    B, n_types = int(g["batch_size"]), len(ints["type_names"])
    feats, tidx = ints["features"], ints["type_idx"]
    for lo in range(0, len(order), B):
        rows = pairs[order[lo:lo + B]]
        f = filler[lo:lo + B]
        tt = tidx[rows[:, 1]]
        pos = rows[:, 2] == 1
        real = feats[rows[:, 1]]
        up = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x)).to(device=device, dtype=dt)
        yield {"query_idx": up(rows[:, 0], torch.int32), "query_types": up(tidx[rows[:, 0]], torch.int64),
               "positive_types": up(np.where(pos, tt, 0)[:, None], torch.int64),                      # data_loader.py:148-150
               "negative_types": up(np.where(pos, (tt + 1) % n_types, tt)[:, None], torch.int64),
               "positive_items": up(np.where(pos[:, None], real, f), torch.float32),
               "negative_items": up(np.where(pos[:, None], f, real), torch.float32),
               "target_features": up(real, torch.float32), "label": up(rows[:, 2], torch.int64)}


class EpochLoader:
    def __init__(self, ints, g, which):
        self.ints, self.g, self.which, self.epoch = ints, g, which, 0

    def __iter__(self):
        e, self.epoch = self.epoch, self.epoch + 1
        return joint_batches(self.ints, self.g, self.which, e, "cuda")


@pytest.mark.parametrize("fused", [True, False])
def test_train_reproduces_the_references_epochs(golden, tmp_path, fused):
    """The build's train() (fused: pc_joint_train_step + Adam in its last kernel; unfused: model(batch) / compute_loss /
    backward / torch.optim.Adam) over the batches the reference's loaders produced, against the reference's train.train."""
    from p_companion_amd import train as drv
    from p_companion_amd.p_companion import PCompanion
    g = golden("g11_joint_epochs.npz")
    ints = golden("g2_bpg1000.npz")
    table = torch.from_numpy(golden("g10_p2v_epochs.npz")["embeddings"])
    emb = {f"P{i:06d}": table[i] for i in range(1000)}
    c = cfg(tmp_path, NUM_TYPES=int(g["num_types"]), NUM_EPOCHS=int(g["epochs"]))
    torch.manual_seed(int(g["seed"]) + 1)
    probe = PCompanion(c, emb)
    for k, v in probe.state_dict().items():
        if k != "product_embeddings.weight":
            assert torch.equal(v, torch.from_numpy(g["init." + k])), k
    torch.manual_seed(int(g["seed"]) + 1)
    model = drv.train(c, EpochLoader(ints, g, "train"), EpochLoader(ints, g, "val"), emb, fused=fused)
    steps = len(g["losses"])
    np.testing.assert_allclose(model.step_losses.numpy(), g["losses"], rtol=0, atol=1e-4)
    names = [str(x) for x in g["metric_names"]]
    n_val = g["val_pairs"].shape[0]
    for e, m in enumerate(model.epoch_metrics):
        for k, want in zip(names, g["metric_values"][e]):
            # hit@k: a mean over three batches of per-batch means over 3 * B rows -- one row more or less = 1 / (3 * 3 * 235)
            tol = 1.0 / (3 * 3 * 235) + 1e-6 if k.startswith("hit@") else (1e-6 if k == "type_diversity" else 1e-4)
            assert abs(m[k] - want) <= tol, (e, k, m[k], want)
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("final.") and not k.startswith(("final.exp_avg", "final.step")):
            adam_close(sd[k[6:]], g[k], steps, 5e-5, name=k)
    best = torch.load(os.path.join(c.MODEL_DIR, "best_model.pth"), weights_only=True)
    assert best["epoch"] == int(g["best_epoch"])                                    # the checkpoint rule (train.py:62-70)
    for k, want in zip(names, g["best_metric_values"]):
        assert abs(best["metrics"][k] - want) <= 1.0 / (3 * 3 * 235) + 1e-4
    for k in g.files:
        if k.startswith("best."):
            adam_close(best["model_state_dict"][k[5:]], g[k], steps, 5e-5, name=k)
    # the optimizer state the checkpoint carries is torch.optim.Adam's layout with the reference's moments
    pnames = [n for n, _ in model.named_parameters()]
    st = best["optimizer_state_dict"]["state"]
    seen = 0
    for idx, s in st.items():
        n = pnames[idx]
        assert float(s["step"]) == float(g["final.step." + n]) == steps
        np.testing.assert_allclose(s["exp_avg"], g["final.exp_avg." + n], rtol=0, atol=2e-5, err_msg=n)
        np.testing.assert_allclose(s["exp_avg_sq"], g["final.exp_avg_sq." + n], rtol=0, atol=1e-6, err_msg=n)
        seen += 1
    assert seen == 10 and "product_embeddings.weight" not in [pnames[i] for i in st]


# --------------------------------------------------------------------------------------------------------- g6, T = 1000
@pytest.mark.parametrize("path", ["fused", "module"])
def test_joint_step_t1000_golden(golden, tmp_path, path):
    """The first reference-made vector through the T > 512 kernels (per-sample similarity product + top-k refinement,
    sorted table gradients, dense-equivalent Adam over [1000,64] tables): three steps on one batch."""
    from p_companion_amd.p_companion import PCompanion
    from p_companion_amd.product2vec import FusedAdam
    g = golden("g6_joint_t1000.npz")
    st = {k[5:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("init.")}
    c = cfg(tmp_path, NUM_TYPES=1000)
    model = PCompanion(c, st["product_embeddings.weight"])
    model.load_state_dict(st)
    model = model.to(c.DEVICE).train()
    dev = lambda k, dt: torch.from_numpy(g["batch." + k]).to(device="cuda", dtype=dt)
    batch = {"query_idx": dev("query_idx", torch.int32), "query_types": dev("query_types", torch.int64),
             "positive_types": dev("positive_types", torch.int64), "negative_types": dev("negative_types", torch.int64),
             "positive_items": dev("positive_items", torch.float32), "negative_items": dev("negative_items", torch.float32)}
    losses = []
    if path == "fused":
        opt = FusedAdam(model, lr=c.LEARNING_RATE)
        for step in range(3):
            l, topk = model.train_step(batch, optimizer=opt)
            losses.append(float(l[0]))
            if step == 0:
                assert np.array_equal(topk.cpu().numpy(), g["complementary_types"])              # index-exact, descending order
                assert abs(float(l[1]) - float(g["type_loss"])) < 1e-4 and abs(float(l[2]) - float(g["item_loss"])) < 1e-4
    else:
        opt = torch.optim.Adam(model.parameters(), lr=c.LEARNING_RATE)
        for step in range(3):
            out = model(batch)
            loss = model.compute_loss(batch, out)
            opt.zero_grad()
            loss.backward()
            if step == 0:
                assert np.array_equal(out["complementary_types"].cpu().numpy(), g["complementary_types"])
                np.testing.assert_allclose(out["type_similarities"][:, :128].detach().cpu(), g["type_similarities"], atol=2e-5)
                np.testing.assert_allclose(out["projected_embeddings"].detach().cpu(), g["projected_embeddings"], atol=2e-5)
                for n, p in model.named_parameters():
                    if p.grad is not None and "grad." + n in g.files:
                        ref = g["grad." + n]
                        np.testing.assert_allclose(p.grad.cpu(), ref, atol=1e-7 + 2e-4 * np.abs(ref).max(), err_msg=n)
            opt.step()
            losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, g["losses"], rtol=0, atol=1e-4)
    sd = model.state_dict()
    for k in g.files:
        if k.startswith("after3.") and "exp_avg" not in k:
            adam_close(sd[k[7:]], g[k], 3, 2e-5, name=k)
