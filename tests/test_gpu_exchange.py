"""The data-parallel exchange slot (ABI 6: pc_exchange_fn, pc_exchange_adam, pc_joint_train_epoch_dp, pc_rccl_*): a replica's
gradient exchange issued from the step's own foreign call.  The reference is single-process (train.py:46-48:
loss.backward(); optimizer.step()); a replica averages the gradients between the two.
  * a one-replica exchange (the identity) must change no bit against the single-process epoch / step;
  * the slot is called once per step, with the flat gradient buffer, BETWEEN the gradient kernels and Adam;
  * on RCCL itself: the library's own communicator in a one-rank 'nccl' job (what one MI355X allows);
  * two ranks on the one card (gloo behind the slot): replicas stay bit-identical and follow the mean gradient.
Needs an MI355X."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _joint(types=100, dropout=0.0, B=448, seed=5, exchange=None, products=1500):
    from types import SimpleNamespace
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader, generate_scaled_bpg
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=dropout, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3,
                          NUM_TYPES=types, DEVICE="cuda")
    bpg = generate_scaled_bpg(products, 40, seed=3)
    torch.manual_seed(seed)
    m = PCompanion(cfg, bpg.cuda("cuda")["features"]).to("cuda").train()
    m.type_transition._dropout_seed = 77
    o = FusedAdam(m, lr=1e-2)
    g = GraphedJointStep(m, o, B, warmup=0, mode="direct", exchange=exchange)
    ld = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), B, shuffle=True, seed=2, device="cuda", out=g.static,
                                  deferred=True)
    return m, o, g, ld


@pytest.mark.parametrize("types,dropout", [(100, 0.0), (40, 0.1), (34800, 0.0), (34800, 0.1)])
def test_one_replica_epoch_through_the_slot_equals_the_single_process_epoch(types, dropout):
    """pc_joint_train_epoch_dp with an exchange that is the identity (one replica: the mean of one) against
    pc_joint_train_epoch: per-step losses, parameters, Adam moments and the step counter, bit for bit -- the fused step
    without its Adam + pc_adam_step_at over the flat buffers is the same update as the finish kernel's own.  The slot is
    called once per step with the flat gradient buffer."""
    from p_companion_amd import ops
    calls = []

    def identity(ptr, n, stream):
        calls.append((ptr, n, stream))

    ex = ops.CallbackExchange(identity)
    m_a, o_a, g_a, ld_a = _joint(types, dropout)
    m_b, o_b, g_b, ld_b = _joint(types, dropout, exchange=ex)
    ref = g_a.run_epoch(ld_a, drop_last=True)
    got = g_b.run_epoch(ld_b, drop_last=True)
    steps = ref.shape[0]
    assert steps >= 2 and got.shape == ref.shape and torch.equal(got, ref)
    gflat = m_b.flatten_parameters()[1]
    assert len(calls) == steps and all(c[0] == gflat.data_ptr() and c[1] == gflat.numel() for c in calls)
    assert all(c[2] == torch.cuda.current_stream().cuda_stream for c in calls)
    for (k, pa), (_, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert torch.equal(pa, pb), k
    assert torch.equal(o_a.exp_avg, o_b.exp_avg) and torch.equal(o_a.exp_avg_sq, o_b.exp_avg_sq)
    assert int(o_a.step_count) == int(o_b.step_count) == steps
    # a second epoch continues the counters; the per-step form (fused step + pc_exchange_adam) is the same update again
    ref2 = g_a.run_epoch(ld_a, drop_last=True)
    got2 = []
    for batch in ld_b:
        if batch["query_idx"].numel() == g_b.batch_size:
            got2.append(g_b(batch)[0].clone())
    assert len(got2) == ref2.shape[0] and all(torch.equal(a, b) for a, b in zip(got2, ref2))
    for (k, pa), (_, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert torch.equal(pa, pb), k
    assert int(o_b.step_count) == 2 * steps and len(calls) == 2 * steps


def test_the_slot_sits_between_the_gradients_and_adam_and_errors_surface():
    """An exchange that halves the gradient buffer must give exactly the parameters of a step taken with half the gradient
    (Adam sees what the exchange left); an exception inside a Python exchange ends the call with that exception."""
    from p_companion_amd import ops
    m_a, o_a, g_a, ld_a = _joint()
    tensors = {}

    def halve(ptr, n, stream):
        tensors[ptr].mul_(0.5)

    ex = ops.CallbackExchange(halve)
    m_b, o_b, g_b, ld_b = _joint(exchange=ex)
    gb = m_b.flatten_parameters()[1]
    tensors[gb.data_ptr()] = gb
    batch_a = next(iter(ld_a))
    batch_b = next(iter(ld_b))
    # reference: gradients only, halve by hand, Adam
    ld_a.materialize(batch_a)                                           # (a deferred batch: built on request)
    la, _ = m_a.train_step(batch_a)
    ga = m_a.flatten_parameters()[1]
    ga.mul_(0.5)
    o_a.step()
    lb, _ = g_b(batch_b)
    assert torch.equal(la, lb)
    for (k, pa), (_, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert torch.equal(pa, pb), k

    def boom(ptr, n, stream):
        raise RuntimeError("exchange failed on purpose")

    m_c, o_c, g_c, ld_c = _joint(exchange=ops.CallbackExchange(boom))
    with pytest.raises(RuntimeError, match="on purpose"):
        g_c.run_epoch(ld_c, drop_last=True)
    torch.cuda.synchronize()


def test_p2v_optimizer_step_with_a_one_replica_exchange_changes_no_bit():
    from types import SimpleNamespace
    from p_companion_amd import ops
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0,
                          DEVICE="cuda")
    bpg = generate_scaled_bpg(5000, 40, seed=1)
    table = bpg.cuda("cuda")["features"]
    torch.manual_seed(0)
    m_a, m_b = Product2Vec(cfg).to("cuda").train(), Product2Vec(cfg).to("cuda").train()
    m_b.load_state_dict(m_a.state_dict())
    o_a, o_b = FusedAdam(m_a, lr=1e-3), FusedAdam(m_b, lr=1e-3)
    n_calls = []
    ex = ops.CallbackExchange(lambda ptr, n, stream: n_calls.append(n))
    for i, b in enumerate(SimilarityIndexLoader(bpg, 512, seed=1, drop_last=True, device="cuda")):
        la = m_a.train_step_indexed(table, b)
        o_a.step()
        lb = m_b.train_step_indexed(table, b)
        o_b.step(exchange=ex)
        assert torch.equal(la, lb)
        if i == 3:
            break
    assert n_calls == [m_b.flatten_parameters()[0].numel()] * 4
    assert torch.equal(m_a.flatten_parameters()[0], m_b.flatten_parameters()[0]) and int(o_b.step_count) == 4


# ----------------------------------------------------------------------------------------------- RCCL itself (one rank)
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker_rccl():
    """(subprocess body) a one-rank 'nccl' job: the library's own RCCL communicator behind the slot"""
    import torch.distributed as dist
    from p_companion_amd import distributed as pdist, ops
    rank, world, local = pdist.init_from_env("cuda")
    assert dist.get_backend() == "nccl" and world == 1 and ops.rccl_available()
    ex = pdist.make_exchange(world, rank=rank, kind="rccl")
    assert isinstance(ex, ops.RcclExchange) and ex.kind.startswith("rccl")
    x = torch.randn(100_003, device="cuda")
    want = x.clone()
    ex.all_reduce_mean_(x)
    assert torch.equal(x, want)                                        # the mean over one rank
    m_a, o_a, g_a, ld_a = _joint(100, 0.0)
    m_b, o_b, g_b, ld_b = _joint(100, 0.0, exchange=ex)
    ref = g_a.run_epoch(ld_a, drop_last=True)
    got = g_b.run_epoch(ld_b, drop_last=True)
    assert torch.equal(ref, got)
    for (k, pa), (_, pb) in zip(m_a.named_parameters(), m_b.named_parameters()):
        assert torch.equal(pa, pb), k
    m_c, o_c, g_c, ld_c = _joint(34800, 0.1, exchange=ex)              # the reference's shipped configuration, dense exchange
    m_d, o_d, g_d, ld_d = _joint(34800, 0.1)
    assert torch.equal(g_c.run_epoch(ld_c, drop_last=True), g_d.run_epoch(ld_d, drop_last=True))
    for (k, pa), (_, pb) in zip(m_c.named_parameters(), m_d.named_parameters()):
        assert torch.equal(pa, pb), k
    ex.close()
    dist.destroy_process_group()
    print("rccl exchange ok")


def test_native_rccl_exchange_one_rank():
    env = dict(os.environ, PC_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PC_DIST_BACKEND", None)
    out = subprocess.run([sys.executable, "-c", "import tests.test_gpu_exchange as t; t._worker_rccl()"], env=env,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "rccl exchange ok" in out.stdout, out.stderr[-3000:]


# ----------------------------------------------------------------------------------------------- two ranks on one card (gloo)
def _worker_world2():
    """(subprocess body) rank of a two-rank job on the one card, gloo behind the exchange slot.  Both ranks start from the same
    parameters and see DIFFERENT batches (loader seed = rank); after an epoch through pc_joint_train_epoch_dp they must hold
    bit-identical parameters, and the first step must equal the step taken by hand with the mean of the two ranks' gradients."""
    rank, port, out = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0",
                      PC_DIST_BACKEND="gloo", PC_FORCE_DEVICE="0")
    res = {"ok": False, "rank": rank}
    try:
        import torch.distributed as dist
        from types import SimpleNamespace
        from p_companion_amd import distributed as pdist
        from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader, generate_scaled_bpg
        from p_companion_amd.p_companion import GraphedJointStep, PCompanion
        from p_companion_amd.product2vec import FusedAdam
        r, w, _ = pdist.init_from_env("cuda")
        assert (r, w) == (rank, 2)
        ex = pdist.make_exchange(2, rank=rank)
        assert "gloo" in ex.kind
        cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=0.0, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=100,
                              DEVICE="cuda")
        bpg = generate_scaled_bpg(1500, 40, seed=3)
        B = 256

        def make(exchange):
            torch.manual_seed(5)
            m = PCompanion(cfg, bpg.cuda("cuda")["features"]).to("cuda").train()
            o = FusedAdam(m, lr=1e-2)
            g = GraphedJointStep(m, o, B, warmup=0, mode="direct", exchange=exchange)
            ld = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), B, shuffle=True, seed=10 + rank, device="cuda",
                                          out=g.static, deferred=True)
            return m, o, g, ld

        # by hand, one step: local gradients, mean over the ranks, Adam
        m_h, o_h, g_h, ld_h = make(None)
        b0 = next(iter(ld_h))
        ld_h.materialize(b0)
        m_h.train_step(b0)
        gh = m_h.flatten_parameters()[1]
        dist.all_reduce(gh)
        gh.mul_(0.5)
        o_h.step()
        # through the slot: one step's worth of pairs, then the rest of the epoch
        m_e, o_e, g_e, ld_e = make(ex)
        ex.register(m_e.flatten_parameters()[1])
        losses = g_e.run_epoch(ld_e, drop_last=True, max_steps=1)
        first_equal = all(torch.equal(pa, pb) for (_, pa), (_, pb) in zip(m_h.named_parameters(), m_e.named_parameters()))
        losses = g_e.run_epoch(ld_e, drop_last=True)
        flat = m_e.flatten_parameters()[0]
        both = [torch.empty_like(flat) for _ in range(2)]
        dist.all_gather(both, flat)
        res.update(ok=bool(first_equal and torch.equal(both[0], both[1]) and torch.isfinite(losses).all()),
                   first_equal=bool(first_equal), replicas_equal=bool(torch.equal(both[0], both[1])), steps=int(losses.shape[0]) + 1,
                   kind=ex.kind)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:                                       # noqa: BLE001 -- reported to the parent test
        import traceback
        res["error"] = traceback.format_exc()
    with open(out, "w") as f:
        json.dump(res, f)


@pytest.mark.timeout(600)
def test_two_replicas_on_one_card_follow_the_mean_gradient(tmp_path):
    port = str(_free_port())
    outs = [str(tmp_path / f"r{r}.json") for r in range(2)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", "import tests.test_gpu_exchange as t; t._worker_world2()",
                               str(r), port, outs[r]], env=env, cwd=ROOT) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=500) == 0
    for o in outs:
        res = json.load(open(o))
        assert res.get("ok"), res
        assert res["first_equal"] and res["replicas_equal"] and res["steps"] >= 3
