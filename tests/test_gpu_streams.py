"""The library across streams, host threads and stream capture: the side queue of the fused Product2Vec step changes no bit, a
captured step stays on one queue, two host threads stepping two models on two streams equal the serial run (ABI 5: "Library-owned
device state").  Needs an MI355X."""
from types import SimpleNamespace
import ctypes
import hashlib
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def cfg(**over):
    c = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                        MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=40, DEVICE=torch.device("cuda"),
                        LEARNING_RATE=1e-3, BATCH_SIZE=64, PRODUCT2VEC_EPOCHS=1, NUM_EPOCHS=1, MODEL_DIR="/tmp/pc_r3_models")
    c.__dict__.update(over)
    return c


def _fork_digest():
    """60 steps of the fused Product2Vec step through the throughput loader -> a digest of the parameters, the BatchNorm
    statistics and the losses."""
    import hashlib
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    bpg = generate_scaled_bpg(20_000, 100, seed=3)
    table = bpg.cuda()["features"]
    torch.manual_seed(0)
    m = Product2Vec(cfg()).to("cuda").train()
    opt = FusedAdam(m, lr=1e-3)
    h = hashlib.sha256()
    n = 0
    for b in SimilarityIndexLoader(bpg, 1024, seed=2, drop_last=True, device="cuda", reuse_buffers=True):
        loss = m.train_step_indexed(table, b)
        opt.step()
        h.update(loss.detach().cpu().numpy().tobytes())
        n += 1
        if n == 60:
            break
    torch.cuda.synchronize()
    h.update(m.flatten_parameters()[0].detach().cpu().numpy().tobytes())
    h.update(m.ffn[1].running_var.cpu().numpy().tobytes())
    return h.hexdigest()


def _clone_batch(b):
    cl = lambda v: v.clone() if torch.is_tensor(v) else v
    return {k: ({kk: (int(vv) if kk == "n_unique" else cl(vv)) for kk, vv in v.items()} if isinstance(v, dict) else cl(v))
            for k, v in b.items()}


def test_side_queue_fork_changes_no_bit():
    """The fused step runs the attention block's few-row weight gradients and the BatchNorm-backward finalize on the
    library's side queue (csrc/common.h PcFork; include/pcompanion_hip.h "Library-owned device state").  Same digest of 60
    steps' losses, parameters and running statistics with the side queue, with everything on the caller's stream
    (pc_set_option(PC_OPT_SIDE_QUEUE, 0)), and again with it after pc_release_device_state() destroyed and the next step
    re-created it."""
    from p_companion_amd import _lib
    L = _lib.lib()
    v = ctypes.c_int(-1)
    assert L.pc_get_option(_lib.PC_OPT_SIDE_QUEUE, ctypes.byref(v)) == 0 and v.value == 1      # the default
    assert L.pc_set_option(99, 1) == -1 and L.pc_set_option(_lib.PC_OPT_SIDE_QUEUE, 2) == -1
    outs = []
    try:
        for on in (1, 0, 1):
            assert L.pc_set_option(_lib.PC_OPT_SIDE_QUEUE, on) == 0
            outs.append(_fork_digest())
            torch.cuda.synchronize()
            assert L.pc_release_device_state() == 0
        # the BatchNorm-backward finalize on the side queue beside dW3 (rounds 3-5) or on the step's own queue (default): same bits
        assert L.pc_get_option(_lib.PC_OPT_BN_FINALIZE_SIDE, ctypes.byref(v)) == 0 and v.value == 0
        assert L.pc_set_option(_lib.PC_OPT_BN_FINALIZE_SIDE, 1) == 0
        outs.append(_fork_digest())
        torch.cuda.synchronize()
        assert L.pc_set_option(_lib.PC_OPT_BN_FINALIZE_SIDE, 0) == 0
        # the triplet hinge inside the attention backward's first launch (default) or as its own launch (rounds 1-5): same bits
        assert L.pc_get_option(_lib.PC_OPT_FUSED_LOSS, ctypes.byref(v)) == 0 and v.value == 1
        assert L.pc_set_option(_lib.PC_OPT_FUSED_LOSS, 0) == 0
        outs.append(_fork_digest())
        torch.cuda.synchronize()
        # ... and with the hinge riding, the out-projection's forward chain in the same launch (default) or on its own
        assert L.pc_set_option(_lib.PC_OPT_FUSED_LOSS, 1) == 0
        assert L.pc_get_option(_lib.PC_OPT_FUSED_OUT_CHAIN, ctypes.byref(v)) == 0 and v.value == 1
        assert L.pc_set_option(_lib.PC_OPT_FUSED_OUT_CHAIN, 0) == 0
        outs.append(_fork_digest())
        torch.cuda.synchronize()
    finally:
        L.pc_set_option(_lib.PC_OPT_FUSED_OUT_CHAIN, 1)
        L.pc_set_option(_lib.PC_OPT_SIDE_QUEUE, 1)
        L.pc_set_option(_lib.PC_OPT_BN_FINALIZE_SIDE, 0)
        L.pc_set_option(_lib.PC_OPT_FUSED_LOSS, 1)
    assert outs[0] == outs[1] == outs[2] == outs[3] == outs[4] == outs[5]


def test_fused_p2v_step_under_stream_capture_stays_on_one_queue():
    """A stream that is being captured keeps the fused step on itself (no side queue inside a capture): the captured step,
    replayed, gives the eager step's loss and gradients bit for bit."""
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import Product2Vec
    bpg = generate_scaled_bpg(20_000, 100, seed=4)
    table = bpg.cuda()["features"]
    torch.manual_seed(0)
    m = Product2Vec(cfg()).to("cuda").train()
    b = next(iter(SimilarityIndexLoader(bpg, 512, seed=1, drop_last=True, device="cuda")))
    loss_e = m.train_step_indexed(table, b).clone()
    grad_e = m.flatten_parameters()[1].clone()
    m.ffn[1].running_mean.zero_(); m.ffn[1].running_var.fill_(1.0); m.ffn[1].num_batches_tracked.zero_()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        m.train_step_indexed(table, b)                            # (workspaces for this stream exist before the capture)
        m.ffn[1].running_mean.zero_(); m.ffn[1].running_var.fill_(1.0); m.ffn[1].num_batches_tracked.zero_()
        with torch.cuda.graph(g, stream=side):
            loss_c = m.train_step_indexed(table, b)
    torch.cuda.current_stream().wait_stream(side)
    m.flatten_parameters()[1].zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss_c, loss_e) and torch.equal(m.flatten_parameters()[1], grad_e)


# ------------------------------------------------------------------ ABI 5: re-entrant across host threads and streams
def test_two_host_threads_step_two_models_on_two_streams_bit_equal_to_serial():
    """include/pcompanion_hip.h: "two threads may step two models on two streams of one device concurrently".  Each thread owns a
    model, an optimizer, a stream (hence its own workspaces and its own side queue of the fused step) and a list of prebuilt
    batches; losses, parameters and BatchNorm statistics after 40 steps equal the serial run's bit for bit."""
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec
    bpg = generate_scaled_bpg(20_000, 100, seed=3)
    table = bpg.cuda()["features"]
    steps = 40
    batches = []
    for seed in (11, 12):
        bs = []
        for b in SimilarityIndexLoader(bpg, 1024, seed=seed, drop_last=True, device="cuda"):
            bs.append(_clone_batch(b))
            if len(bs) == steps:
                break
        batches.append(bs)
    torch.cuda.synchronize()

    def models():
        # (built in the calling thread: torch.manual_seed / the initialisers draw from ONE process-wide CPU generator)
        out = []
        for which in (0, 1):
            torch.manual_seed(which)
            out.append(Product2Vec(cfg()).to("cuda").train())
        torch.cuda.synchronize()
        return out

    def run(which, stream, out, m):
        try:
            with torch.cuda.stream(stream):
                opt = FusedAdam(m, lr=1e-3)
                h = hashlib.sha256()
                losses = []
                for b in batches[which]:
                    losses.append(m.train_step_indexed(table, b))
                    opt.step()
                stream.synchronize()
                for l in losses:
                    h.update(l.cpu().numpy().tobytes())
                h.update(m.flatten_parameters()[0].detach().cpu().numpy().tobytes())
                h.update(m.ffn[1].running_mean.cpu().numpy().tobytes())
                h.update(m.ffn[1].running_var.cpu().numpy().tobytes())
                out[which] = h.hexdigest()
        except BaseException as e:                               # (surface a worker's failure in the main thread)
            out[which] = e

    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    serial = {}
    ms = models()
    run(0, s[0], serial, ms[0])
    run(1, s[1], serial, ms[1])
    assert all(isinstance(v, str) for v in serial.values()), serial
    for _ in range(2):                                            # twice: first use and re-use of the two side queues
        par = {}
        ms = models()
        th = [threading.Thread(target=run, args=(i, s[i], par, ms[i])) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(300)
        assert par == serial, (par, serial)
    assert serial[0] != serial[1]
