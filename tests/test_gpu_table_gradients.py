"""Gradients of P-Companion's two [NUM_TYPES, 64] embedding tables at NUM_TYPES > 512 (src/models/p_companion.py:36-43: dense
autograd gradients of which a batch touches few rows; config.py:27 ships NUM_TYPES = 34800): the fused step sorts the source rows
by destination (stable) and adds every destination's run in source order -- table_sort_kernel / table_segsum_kernel,
csrc/joint_fused.hip.  Against the oracle, and bit for bit run to run at ANY number of touched rows (with config.py:12's
DROPOUT = 0.1 every sample selects its own K types: thousands of touched rows of the complementary table).  Needs an MI355X."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import joint_oracle, philox_oracle


def _model(T, K, dropout, P=2000, seed=3):
    from p_companion_amd.p_companion import PCompanion
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=dropout, MARGIN=1.0,
                          ALPHA=0.8, NUM_COMP_TYPES=K, NUM_TYPES=T, DEVICE=torch.device("cuda"), LEARNING_RATE=1e-3)
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(P, 128, generator=g)
    torch.manual_seed(seed + 1)
    m = PCompanion(cfg, table).to("cuda").train()
    m.type_transition._dropout_seed, m.type_transition._dropout_step = 4242, 0
    return m, cfg


def _batch(B, P, live, seed):
    g = torch.Generator().manual_seed(seed)
    return {"query_idx": torch.randint(0, P, (B,), generator=g, dtype=torch.int32),
            "query_types": torch.randint(0, live, (B,), generator=g), "positive_types": torch.randint(0, live, (B, 1), generator=g),
            "negative_types": torch.randint(0, live, (B, 1), generator=g), "positive_items": torch.randn(B, 128, generator=g),
            "negative_items": torch.randn(B, 128, generator=g)}


# (T, live types of the hinge / query ids, dropout, B, K, sorted path?)
CASES = [(34800, 34800, 0.1, 4096, 3, True),     # thousands of touched rows in both tables
         (34800, 4, 0.1, 4096, 3, True),         # four hot rows with ~2 000 source rows each: runs cut into quarters
         (34800, 100, 0.1, 4096, 4, True),       # K = 4: 24 576 source rows, the sort kernel's capacity
         (3000, 3000, 0.0, 777, 2, True),        # ragged batch, dropout off (rows of the similarity product = distinct query types)
         (70000, 5000, 0.1, 2048, 3, False)]     # NUM_TYPES > 65535: the bitmap-list path (float atomics beyond 512 touched rows)


@pytest.mark.parametrize("T,live,dropout,B,K,sorted_path", CASES)
def test_table_gradients_against_the_oracle_and_run_to_run(T, live, dropout, B, K, sorted_path):
    from p_companion_amd import _lib
    L = _lib.lib()
    # (the sorted form is the default wherever the lists fit its sort kernel; pc_set_option(PC_OPT_SORTED_TABLE_GRADIENTS, 0) keeps
    # the LDS-table form -- reproducible up to 512 touched rows per table -- for steps without dropout: both are run)
    for option in ((1, 0) if dropout == 0.0 and sorted_path else (1,)):
        assert L.pc_set_option(_lib.PC_OPT_SORTED_TABLE_GRADIENTS, option) == 0
        try:
            _check_table_gradients(T, live, dropout, B, K, sorted_path and option == 1)
        finally:
            assert L.pc_set_option(_lib.PC_OPT_SORTED_TABLE_GRADIENTS, 1) == 0


def _check_table_gradients(T, live, dropout, B, K, sorted_path, hb=None):
    model, cfg = _model(T, K, dropout)
    st0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    hb = _batch(B, 2000, live, seed=11) if hb is None else hb
    db = {k: v.cuda() for k, v in hb.items()}
    tt = model.type_transition

    def run():
        tt._dropout_step = 0
        losses, topk = model.train_step(db)
        return losses.clone(), topk.clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.requires_grad}

    l1, t1, g1 = run()
    mask = None
    if dropout > 0:
        mask = torch.from_numpy(philox_oracle.dropout_mask(4242, 0, philox_oracle.STREAM_HIDDEN, B * 32, dropout)).view(B, 32)
    ref = joint_oracle.train_step({k: v.clone() for k, v in st0.items()}, hb, joint_oracle.new_moments(st0), 1, cfg.MARGIN, cfg.ALPHA,
                                  K, hidden_mask=mask)
    assert abs(float(l1[0]) - float(ref["loss"])) < 1e-4
    same = (t1.cpu().long() == ref["out"]["complementary_types"]).all(1)
    assert int((~same).sum()) <= 2                                 # (two similarities of a row equal to fp32 rounding at most)
    slack = float((~same).sum()) * 4.0 / (B * K)
    for k in joint_oracle.TRAINABLE:
        r = ref["grads"][k]
        d = float((g1[k].cpu() - r).abs().max())
        assert d < 2e-6 + 2e-4 * float(r.abs().max()) + slack, (k, d)
    for nm in ("query_type_embeddings.weight", "complementary_type_embeddings.weight"):
        untouched = ref["grads"][nm].abs().sum(1) == 0
        assert bool((g1[nm].cpu()[untouched] == 0).all()), nm      # rows no sample points at stay exactly zero
    if sorted_path:
        for _ in range(2):                                         # bit for bit, whatever the number of touched rows
            l2, t2, g2 = run()
            assert torch.equal(l1, l2) and torch.equal(t1, t2)
            for k in g1:
                assert torch.equal(g1[k], g2[k]), k
    else:
        l2, t2, g2 = run()
        assert torch.equal(t1, t2)
        for k in g1:
            assert torch.allclose(g1[k], g2[k], rtol=1e-5, atol=1e-7), k


def test_runs_of_every_length_class_and_their_boundaries():
    """The consumer of the sorted table gradients treats a destination's run by its length (joint_fused.hip, table_segsum_kernel:
    up to 4 rows / up to 64 / up to 256 / longer, in 256-row parts added by the last workgroup to finish): positive, negative and
    query types drawn so that runs of 1 .. 5, 63 .. 66, 255 .. 258, 511 .. 513 and ~1 100 rows all occur in one step (the selected
    types add a few rows here and there: the boundaries are hit from both sides over the three lists)."""
    T, B, K = 34800, 4096, 3
    sizes = [1, 2, 3, 4, 5, 6, 63, 64, 65, 66, 255, 256, 257, 258, 511, 512, 513]
    hb = _batch(B, 2000, T, seed=31)

    def staircase(first_type, stride):
        ids, t = [], first_type
        for n in sizes:
            ids += [t] * n
            t += stride
        ids += [t] * (B - len(ids))                          # the rest (~1 100 rows) on one more type
        g = torch.Generator().manual_seed(first_type)
        return torch.tensor(ids)[torch.randperm(B, generator=g)]

    hb["positive_types"] = staircase(1000, 7).view(B, 1)
    hb["negative_types"] = staircase(20000, 2175).view(B, 1) % T      # (spread over the sort kernel's ranges of the table)
    hb["query_types"] = staircase(5, 1)
    _check_table_gradients(T, T, 0.1, B, K, True, hb=hb)


def test_touched_row_lists_are_the_distinct_destinations_at_thousands_of_rows():
    """pc_joint_fused_touched after a step with > 512 touched rows per table (the lists a data-parallel job exchanges,
    distributed.TableRowExchange): ascending, distinct, exactly the rows with a gradient."""
    from p_companion_amd import ops
    from p_companion_amd.p_companion import GraphedJointStep
    from p_companion_amd.product2vec import FusedAdam
    T, B, K = 34800, 4096, 3
    model, cfg = _model(T, K, 0.1)
    step = GraphedJointStep(model, FusedAdam(model), B, warmup=0, mode="direct", grad_hook=lambda g: g)       # gradients only
    step({k: v.cuda() for k, v in _batch(B, 2000, T, seed=5).items()})
    rc, rq, nt = ops.joint_fused_touched(step.prepared.ws, B, T, K)
    n_c, n_q = (int(v) for v in nt.tolist())
    params = dict(model.named_parameters())
    db = step.static
    # destinations of the two lists (p_companion.py:45-77, 95-103): E_c -- the K selected types of every sample and its hinge's
    # positive / negative type; E_q -- the query types
    dest_c = torch.unique(torch.cat([step.complementary_types.reshape(-1).long(), db["positive_types"].reshape(-1).long(),
                                     db["negative_types"].reshape(-1).long()]))
    dest_q = torch.unique(db["query_types"].reshape(-1).long())
    for nm, ids, dest in (("complementary_type_embeddings.weight", rc[:n_c], dest_c), ("query_type_embeddings.weight", rq[:n_q], dest_q)):
        assert torch.equal(ids.long().cpu(), dest.cpu()), nm           # ascending and distinct (torch.unique sorts)
        g = params[nm].grad
        has = torch.nonzero(g.abs().amax(1) > 0).reshape(-1)
        assert bool(torch.isin(has, dest.to(has.device)).all())        # (a listed row may hold a zero gradient: an inactive hinge)
    assert n_c > 512 and n_q > 512


def test_giant_runs_hand_their_sums_over_reliably():
    """A destination most of the batch points at is summed by several workgroups (256 rows each) on different XCDs; the last one to
    finish adds their 64-row sums, handed over through memory behind an agent-scope release / acquire (joint_fused.hip,
    table_segsum_kernel).  A stale read there would show as a run-to-run difference: thirty steps on the same batch -- four hot types
    with ~2 000 rows each and eight with ~500, the rest spread -- all bit for bit the first one."""
    T, B, K = 34800, 4096, 3
    model, cfg = _model(T, K, 0.1)
    hb = _batch(B, 2000, T, seed=41)
    g = torch.Generator().manual_seed(7)
    hot = torch.randint(0, 4, (B,), generator=g) * 8000 + 17
    warm = torch.randint(0, 8, (B,), generator=g) * 4000 + 333
    hb["positive_types"] = hot.view(B, 1)
    hb["negative_types"] = torch.where(torch.rand(B, generator=g) < 0.5, hot, warm).view(B, 1)
    hb["query_types"] = warm
    db = {k: v.cuda() for k, v in hb.items()}
    tt = model.type_transition
    first = None
    for i in range(30):
        tt._dropout_step = 0
        model.train_step(db)
        cur = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
        if first is None:
            first = cur
        else:
            for k in first:
                assert torch.equal(first[k], cur[k]), (i, k)
