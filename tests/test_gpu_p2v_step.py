"""Product2Vec training step on the GPU (C ABI pc_p2v_train_step + pc_adam_step) against the
reference's golden vectors and the oracle.  Needs an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import p2v_oracle


def _load(golden, name):
    g = golden(name)
    st = {k[5:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("init.")}
    return g, st


def _flat(st, keys):
    return torch.cat([st[k].reshape(-1) for k in keys])


def _run_steps(ops, st, table, batch, n_steps):
    """Flat parameter / gradient / moment buffers with per-tensor views, as the modules use."""
    keys = ops.P2V_KEYS
    sizes = [int(np.prod(s)) for s in ops.P2V_SHAPES]
    flat = torch.cat([st[k].reshape(-1) for k in keys]).cuda()
    gflat = torch.zeros_like(flat); m = torch.zeros_like(flat); v = torch.zeros_like(flat)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    params = {k: flat[offs[i]:offs[i + 1]].view(s) for i, (k, s) in enumerate(zip(keys, ops.P2V_SHAPES))}
    grads = {k: gflat[offs[i]:offs[i + 1]].view(s) for i, (k, s) in enumerate(zip(keys, ops.P2V_SHAPES))}
    for k in ("ffn.1.running_mean", "ffn.1.running_var", "ffn.1.num_batches_tracked"):
        params[k] = st[k].clone().cuda()
    step = torch.zeros(1, dtype=torch.int64, device="cuda"); scal = torch.zeros(2, device="cuda")
    outs, first_grads, after1 = [], None, None
    nbr = batch.get("neighbor_idx")
    for i in range(n_steps):
        out = ops.p2v_train_step(params, grads, table, batch["anchor_idx"], batch["positive_idx"],
                                 batch["negative_idx"], nbr, 1.0, want_emb=True)
        if i == 0:
            first_grads = {k: g.clone().cpu() for k, g in grads.items()}
            bn1 = {k: params[k].clone().cpu() for k in params if "running" in k or "num_batches" in k}
        ops.adam_step(flat, gflat, m, v, step, scal)
        if i == 0:
            after1 = {k: params[k].clone().cpu() for k in keys}
        outs.append({k: t.clone().cpu() for k, t in out.items()})
    return outs, first_grads, bn1, after1, {k: params[k].cpu() for k in keys}


def _adam_close(actual, desired, steps, tight, lr=1e-3):
    d = (actual - torch.as_tensor(desired)).abs()
    assert float(d.max()) <= 1.05 * lr * steps
    assert float((d <= tight).float().mean()) >= 0.999


def _check(g, outs, first_grads, bn1, final, after3):
    losses = [float(o["loss"]) for o in outs]
    # north_star: fp32 loss within 1e-4 of the reference
    np.testing.assert_allclose(losses, g["losses"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(outs[0]["anchor_emb"], g["anchor_emb"], atol=2e-5)
    np.testing.assert_allclose(outs[0]["d_pos"], g["pos_distance"], atol=5e-5)
    np.testing.assert_allclose(outs[0]["d_neg"], g["neg_distance"], atol=5e-5)
    for k in p2v_oracle.TRAINABLE:
        ref = g["grad." + k]
        if k == "ffn.0.bias":
            assert float(first_grads[k].abs().max()) < 1e-6      # analytically zero (BatchNorm)
            continue
        np.testing.assert_allclose(first_grads[k], ref, atol=2e-6 + 2e-4 * np.abs(ref).max(), err_msg=k)
    for k, v in bn1.items():
        np.testing.assert_allclose(v, g["bn_after1." + k], rtol=1e-5, atol=1e-5)
    for k in p2v_oracle.TRAINABLE:
        if k != "ffn.0.bias":
            _adam_close(final[k], after3["after3." + k], 3, 5e-5)


def test_train_step_golden_tiny(golden):
    from p_companion_amd import ops
    g, st = _load(golden, "g4_p2v_tiny.npz")
    b = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("batch.")}
    # dense golden batch -> table + indices (index form is the C ABI's input)
    B, N = b["anchor_neighbors"].shape[:2]
    table = torch.cat([b["anchor"], b["positive"], b["negative"].reshape(-1, 128),
                       b["anchor_neighbors"].reshape(-1, 128)]).cuda()
    ar = lambda lo, n: torch.arange(lo, lo + n, dtype=torch.int32).cuda()
    nb = ar(B + B + 5 * B, B * N).view(B, N).clone()
    nb[2, 4:] = -1; nb[5, 3:] = -1                     # the zero-padded rows of the fixture
    batch = {"anchor_idx": ar(0, B), "positive_idx": ar(B, B), "negative_idx": ar(2 * B, 5 * B).view(B, 5),
             "neighbor_idx": nb}
    outs, fg, bn1, after1, final = _run_steps(ops, st, table, batch, 3)
    _check(g, outs, fg, bn1, final, g)


def test_train_step_golden_b256(golden):
    from p_companion_amd import ops
    g, st = _load(golden, "g4_p2v_b256.npz")
    ints = golden("g2_bpg1000.npz")
    table = torch.from_numpy(ints["features"]).cuda()
    batch = {k: torch.from_numpy(g[k]).cuda() for k in ("anchor_idx", "positive_idx", "negative_idx", "neighbor_idx")}
    outs, fg, bn1, after1, final = _run_steps(ops, st, table, batch, 3)
    _check(g, outs, fg, bn1, final, golden("g4_p2v_b256_params3.npz"))


def test_train_step_features_scaled_1e4_vs_oracle(golden):
    """The golden B = 256 batch with its features scaled by 1e4 (operands far from unit scale through Linear0 and its
    weight gradient; BatchNorm brings the rest back): loss within the north_star's 1e-4 of the oracle, gradients close."""
    from p_companion_amd import ops
    g, st = _load(golden, "g4_p2v_b256.npz")
    ints = golden("g2_bpg1000.npz")
    feats = torch.from_numpy(ints["features"]) * 1e4
    batch = {k: torch.from_numpy(g[k]) for k in ("anchor_idx", "positive_idx", "negative_idx", "neighbor_idx")}
    outs, fg, bn1, after1, final = _run_steps(ops, st, feats.cuda(), {k: v.cuda() for k, v in batch.items()}, 1)
    ost = {k: v.clone() for k, v in st.items()}
    dense = p2v_oracle.gather_batch(feats, batch["anchor_idx"], batch["positive_idx"], batch["negative_idx"], batch["neighbor_idx"])
    ref = p2v_oracle.train_step(ost, dense, 1.0, p2v_oracle.new_moments(ost), 1)
    assert abs(float(outs[0]["loss"]) - float(ref["loss"])) < 1e-4
    for k in p2v_oracle.TRAINABLE:
        if k == "ffn.0.bias":
            continue
        r = ref["grads"][k]
        np.testing.assert_allclose(fg[k], r, atol=3e-6 + 3e-4 * float(r.abs().max()), err_msg=k)


def test_train_step_no_neighbors_vs_oracle():
    """anchor_neighbors absent (data_loader.py:67-69): embedding = plain FFN, 3 BatchNorm calls,
    attention parameters receive no gradient."""
    from p_companion_amd import ops
    st = p2v_oracle.init_state(5)
    g = torch.Generator().manual_seed(6)
    table = torch.randn(64, 128, generator=g)
    B = 16
    ai = torch.randint(0, 64, (B,), generator=g, dtype=torch.int32)
    pi = torch.randint(0, 64, (B,), generator=g, dtype=torch.int32)
    ni = torch.randint(0, 64, (B, 5), generator=g, dtype=torch.int32)
    # oracle forward/grad without attention
    leaves = {k: st[k].clone().requires_grad_(True) for k in p2v_oracle.TRAINABLE}
    work = {k: v.clone() for k, v in st.items()}; work.update(leaves)
    a = p2v_oracle.forward(table[ai.long()], None, work, True)
    p = p2v_oracle.forward(table[pi.long()], None, work, True)
    n = p2v_oracle.forward(table[ni.long()], None, work, True)
    loss, _, _ = p2v_oracle.triplet_loss(a, p, n, 1.0)
    grads_ref = torch.autograd.grad(loss, [leaves[k] for k in p2v_oracle.TRAINABLE], allow_unused=True)
    params = {k: v.clone().cuda() for k, v in st.items()}
    grads = {k: torch.full_like(params[k], 7.0) for k in ops.P2V_KEYS}
    out = ops.p2v_train_step(params, grads, table.cuda(), ai.cuda(), pi.cuda(), ni.cuda(), None, 1.0)
    assert abs(float(out["loss"]) - float(loss.detach())) < 1e-5
    for k, gr in zip(p2v_oracle.TRAINABLE, grads_ref):
        if gr is None:
            assert float(grads[k].abs().max()) == 0.0, k
        elif k != "ffn.0.bias":
            np.testing.assert_allclose(grads[k].cpu(), gr, atol=2e-6 + 2e-4 * float(gr.abs().max()), err_msg=k)
    assert int(params["ffn.1.num_batches_tracked"]) == 3


# ------------------------------------------------------------------ compact neighbour rows
def test_train_step_compact_golden_b256(golden):
    """pc_p2v_train_step_compact (zero-padding rows carried once, weighted in BatchNorm) against
    the REFERENCE's golden step: 37% of this batch's neighbour slots are padding."""
    from p_companion_amd import ops
    g, st = _load(golden, "g4_p2v_b256.npz")
    ints = golden("g2_bpg1000.npz")
    table = torch.from_numpy(ints["features"]).cuda()
    batch = {k: torch.from_numpy(g[k]).cuda() for k in ("anchor_idx", "positive_idx", "negative_idx")}
    nb = torch.from_numpy(g["neighbor_idx"]).cuda()
    comp = ops.compact_neighbors(nb)
    assert comp["nb_rows"].numel() - 1 == int((g["neighbor_idx"] >= 0).sum()) < nb.numel()
    batch["neighbor_idx"] = comp
    outs, fg, bn1, after1, final = _run_steps(ops, st, table, batch, 3)
    _check(g, outs, fg, bn1, final, golden("g4_p2v_b256_params3.npz"))


def test_train_step_compact_golden_tiny(golden):
    from p_companion_amd import ops
    g, st = _load(golden, "g4_p2v_tiny.npz")
    b = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("batch.")}
    B, N = b["anchor_neighbors"].shape[:2]
    table = torch.cat([b["anchor"], b["positive"], b["negative"].reshape(-1, 128),
                       b["anchor_neighbors"].reshape(-1, 128)]).cuda()
    ar = lambda lo, n: torch.arange(lo, lo + n, dtype=torch.int32).cuda()
    nb = ar(B + B + 5 * B, B * N).view(B, N).clone()
    nb[2, 4:] = -1; nb[5, 3:] = -1
    batch = {"anchor_idx": ar(0, B), "positive_idx": ar(B, B), "negative_idx": ar(2 * B, 5 * B).view(B, 5),
             "neighbor_idx": ops.compact_neighbors(nb)}
    outs, fg, bn1, after1, final = _run_steps(ops, st, table, batch, 3)
    _check(g, outs, fg, bn1, final, g)


def test_compact_builder_matches_dense_builder(golden):
    from p_companion_amd import ops
    from p_companion_amd.data import IntBPG
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    gdev = bpg.cuda()
    ids = np.arange(100, 356, dtype=np.int32)
    deg = bpg.degree(bpg.similarity_pairs[ids, 0])
    n_pad = int(deg.max())
    a, p, ng, nb = ops.build_similarity_batch(torch.from_numpy(ids).cuda(), gdev, n_pad, 5, seed=9, step=4)
    a2, p2, ng2, comp = ops.build_similarity_batch_compact(torch.from_numpy(ids).cuda(), gdev, n_pad, 5, 9, 4,
                                                          int(deg.sum()))
    assert torch.equal(a, a2) and torch.equal(p, p2) and torch.equal(ng, ng2)
    ref = ops.compact_neighbors(nb)
    assert torch.equal(comp["nb_rows"], ref["nb_rows"]) and torch.equal(comp["slot_row"], ref["slot_row"])
    # slot map reconstructs the dense matrix
    rows = comp["nb_rows"].cpu().numpy()
    assert np.array_equal(rows[comp["slot_row"].cpu().numpy()], nb.cpu().numpy())


def test_cross_replica_batchnorm_equals_concatenated_batch():
    """pc_p2v_train_step_compact_sync: two data-parallel replicas whose BatchNorm sums are added
    between the phases reproduce the single-device step on the concatenated batch -- loss (mean of the two),
    gradients (mean of the two) and running statistics.  The two replicas are run here one after the other
    in one process; `reduce` plays the all-reduce by adding the other replica's buffer."""
    from p_companion_amd import ops
    from p_companion_amd.data import generate_scaled_bpg, SimilarityIndexLoader
    torch.manual_seed(0)
    bpg = generate_scaled_bpg(4000, 20, seed=3)
    table = bpg.cuda()["features"]
    B = 256
    loader = SimilarityIndexLoader(bpg, 2 * B, seed=5, drop_last=True, compact=False, prefetch=False)
    big = next(iter(loader))                                   # dense [2B, N] neighbour indices
    nb = big["neighbor_idx"]
    halves = []
    for h in range(2):
        sl = slice(h * B, (h + 1) * B)
        halves.append({"anchor_idx": big["anchor_idx"][sl].contiguous(), "positive_idx": big["positive_idx"][sl].contiguous(),
                       "negative_idx": big["negative_idx"][sl].contiguous(),
                       "neighbor_compact": ops.compact_neighbors(nb[sl].contiguous())})   # same n_pad on both replicas
    whole = dict(big, neighbor_compact=ops.compact_neighbors(nb))

    def fresh():
        sizes = [int(np.prod(s)) for s in ops.P2V_SHAPES]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        g = torch.Generator(device="cpu").manual_seed(11)
        flat = (torch.randn(int(offs[-1]), generator=g) * 0.05).cuda()
        gflat = torch.zeros_like(flat)
        params = {k: flat[offs[i]:offs[i + 1]].view(s) for i, (k, s) in enumerate(zip(ops.P2V_KEYS, ops.P2V_SHAPES))}
        grads = {k: gflat[offs[i]:offs[i + 1]].view(s) for i, (k, s) in enumerate(zip(ops.P2V_KEYS, ops.P2V_SHAPES))}
        params["ffn.1.weight"].fill_(1.0)
        params["ffn.1.running_mean"] = torch.zeros(256, device="cuda")
        params["ffn.1.running_var"] = torch.ones(256, device="cuda")
        params["ffn.1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device="cuda")
        return params, grads, gflat

    # reference: one device, the concatenated batch
    p0, g0, gf0 = fresh()
    out0 = ops.p2v_train_step(p0, g0, table, whole["anchor_idx"], whole["positive_idx"], whole["negative_idx"],
                              whole["neighbor_compact"], 1.0)

    # two replicas, phases interleaved by hand so that each "all-reduce" sees both contributions
    import ctypes
    from p_companion_amd import _lib
    reps = [fresh() for _ in range(2)]
    outs, bufs = [], []
    for r in range(2):
        bufs.append({k: torch.zeros(ops.BN_SYNC_DOUBLES, dtype=torch.float64, device="cuda") for k in ("fwd", "bl", "bg")})
    state = []
    for r, (params, grads, _) in enumerate(reps):
        st, dev = ops.p2v_struct(params)
        gst, _ = ops.p2v_struct(grads, with_buffers=False)
        hb = halves[r]
        nbc = hb["neighbor_compact"]
        out = {"loss": torch.empty(1, device="cuda"), "d_pos": torch.empty(B, device="cuda"), "d_neg": torch.empty(B, device="cuda")}
        n = nbc["slot_row"].shape[1]
        nbytes = _lib.lib().pc_p2v_train_step_workspace_bytes(B, n, 5)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")          # one workspace per replica: it carries the state
        state.append((st, gst, hb, nbc, out, n, nbytes, ws))
        outs.append(out)

    def phase(r, ph):
        st, gst, hb, nbc, out, n, nbytes, ws = state[r]
        ops.check(_lib.lib().pc_p2v_train_step_compact_sync(
            ctypes.byref(st), ctypes.byref(gst), ops._p(table), ops._p(hb["anchor_idx"]), ops._p(hb["positive_idx"]),
            ops._p(hb["negative_idx"]), ops._p(nbc["nb_rows"]), nbc["nb_rows"].numel() - 1, ops._p(nbc["slot_row"]), B, n, 5,
            1.0, ops._p(out["loss"]), ops._p(out["d_pos"]), ops._p(out["d_neg"]), None, ph, ops._p(bufs[r]["fwd"]),
            ops._p(bufs[r]["bl"]), ops._p(bufs[r]["bg"]), ops._p(ws), nbytes, ops._stream()), "sync step")

    for r in range(2): phase(r, 0)
    tot = bufs[0]["fwd"] + bufs[1]["fwd"]
    for r in range(2): bufs[r]["fwd"].copy_(tot)
    for r in range(2): phase(r, 1)
    tot = bufs[0]["bl"] + bufs[1]["bl"]
    for r in range(2): bufs[r]["bg"].copy_(tot)
    for r in range(2): phase(r, 2)
    torch.cuda.synchronize()

    loss_dp = 0.5 * (float(outs[0]["loss"]) + float(outs[1]["loss"]))
    assert abs(loss_dp - float(out0["loss"])) < 2e-6, (loss_dp, float(out0["loss"]))
    g_dp = 0.5 * (reps[0][2] + reps[1][2])
    tol = 2e-6 + 2e-4 * float(gf0.abs().max())
    assert float((g_dp - gf0).abs().max()) < tol, float((g_dp - gf0).abs().max())
    for r in range(2):                                         # batch-wide running statistics on every replica
        assert torch.allclose(reps[r][0]["ffn.1.running_mean"], p0["ffn.1.running_mean"], atol=1e-6)
        assert torch.allclose(reps[r][0]["ffn.1.running_var"], p0["ffn.1.running_var"], atol=1e-6)
    # and per-replica statistics are NOT the same thing (the test would be vacuous otherwise)
    p1, g1, gf1 = fresh()
    o1 = ops.p2v_train_step(p1, g1, table, halves[0]["anchor_idx"], halves[0]["positive_idx"], halves[0]["negative_idx"],
                            halves[0]["neighbor_compact"], 1.0)
    assert float((gf1 - reps[0][2]).abs().max()) > 10 * tol


def test_unique_neighbour_layout_equals_dense_and_golden(golden):
    """pc_p2v_train_step_unique: every distinct neighbour product carried once (multiplicity-weighted BatchNorm,
    summed slot gradients) gives the dense step's loss and gradients; also against the reference's golden step."""
    from p_companion_amd import ops
    from p_companion_amd.data import generate_scaled_bpg, SimilarityIndexLoader
    bpg = generate_scaled_bpg(3000, 20, seed=4)            # small catalogue: many repeated neighbours
    table = bpg.cuda()["features"]
    loader = SimilarityIndexLoader(bpg, 512, seed=2, drop_last=True, compact=False, prefetch=False)
    batch = next(iter(loader))
    nb = batch["neighbor_idx"]
    uq = ops.unique_neighbors(nb)
    real = int((nb >= 0).sum())
    assert uq["n_unique"] < 0.8 * real                       # the layout must actually merge rows here

    def fresh():
        sizes = [int(np.prod(s)) for s in ops.P2V_SHAPES]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        g = torch.Generator(device="cpu").manual_seed(5)
        flat = (torch.randn(int(offs[-1]), generator=g) * 0.05).cuda()
        gflat = torch.zeros_like(flat)
        params = {k: flat[offs[i]:offs[i + 1]].view(s) for i, (k, s) in enumerate(zip(ops.P2V_KEYS, ops.P2V_SHAPES))}
        grads = {k: gflat[offs[i]:offs[i + 1]].view(s) for i, (k, s) in enumerate(zip(ops.P2V_KEYS, ops.P2V_SHAPES))}
        params["ffn.1.weight"].fill_(1.0)
        params["ffn.1.running_mean"] = torch.zeros(256, device="cuda")
        params["ffn.1.running_var"] = torch.ones(256, device="cuda")
        params["ffn.1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device="cuda")
        return params, grads, gflat

    pd, gd, gfd = fresh()
    od = ops.p2v_train_step(pd, gd, table, batch["anchor_idx"], batch["positive_idx"], batch["negative_idx"], nb, 1.0,
                            want_emb=True)
    pu, gu, gfu = fresh()
    ou = ops.p2v_train_step(pu, gu, table, batch["anchor_idx"], batch["positive_idx"], batch["negative_idx"], uq, 1.0,
                            want_emb=True)
    assert abs(float(od["loss"]) - float(ou["loss"])) < 2e-6
    assert float((od["anchor_emb"] - ou["anchor_emb"]).abs().max()) < 2e-5
    tol = 2e-6 + 2e-4 * float(gfd.abs().max())
    assert float((gfd - gfu).abs().max()) < tol, float((gfd - gfu).abs().max())
    assert torch.allclose(pd["ffn.1.running_var"], pu["ffn.1.running_var"], atol=1e-6)

    # device builder == host construction (same rows, weights, slot map), and it leaves its counters zeroed
    g = bpg.cuda()
    perm = torch.arange(512, dtype=torch.int32, device="cuda")
    deg = np.minimum(bpg.degree(bpg.similarity_pairs[:512, 0]), 32)
    n_pad, n_real = int(deg.max()), int(deg.sum())
    for rep in range(2):                                      # twice: the scratch counters must come back to zero
        a, p_, ng, u = ops.build_similarity_batch_unique(perm, g, n_pad, 5, 7, 3, n_real)
        a2, p2, ng2, dense = ops.build_similarity_batch(perm, g, n_pad, 5, 7, 3)
        ref = ops.unique_neighbors(dense)
        U = int(u["n_unique"][0])
        assert U == ref["n_unique"]
        assert torch.equal(u["nb_rows"][:U + 1], ref["nb_rows"]) and torch.equal(u["weight"][:U + 1], ref["weight"])
        assert torch.equal(u["slot_row"], ref["slot_row"]) and torch.equal(a, a2) and torch.equal(ng, ng2)
        assert torch.equal(u["ref_off"][:U + 2], ref["ref_off"])
        ro, rs, rr = u["ref_off"].cpu().numpy(), u["ref_slot"].cpu().numpy(), ref["ref_slot"].cpu().numpy()
        for r in range(0, U, 97):                              # same slots per row (arrival order is free)
            assert sorted(rs[ro[r]:ro[r + 1]].tolist()) == rr[ro[r]:ro[r + 1]].tolist()



def test_train_step_unique_golden_b256(golden):
    """The REFERENCE's golden step (256 samples of its own 1k-product graph, 37 % padding, repeated neighbours)
    through pc_p2v_train_step_unique: 3 steps, losses / gradients / BN statistics / parameters after Adam."""
    from p_companion_amd import ops
    g, st = _load(golden, "g4_p2v_b256.npz")
    ints = golden("g2_bpg1000.npz")
    table = torch.from_numpy(ints["features"]).cuda()
    batch = {k: torch.from_numpy(g[k]).cuda() for k in ("anchor_idx", "positive_idx", "negative_idx")}
    nb = torch.from_numpy(g["neighbor_idx"]).cuda()
    uq = ops.unique_neighbors(nb)
    assert uq["n_unique"] < int((g["neighbor_idx"] >= 0).sum())
    batch["neighbor_idx"] = uq
    outs, fg, bn1, after1, final = _run_steps(ops, st, table, batch, 3)
    _check(g, outs, fg, bn1, final, golden("g4_p2v_b256_params3.npz"))
