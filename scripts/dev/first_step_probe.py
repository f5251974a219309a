"""Developer probe: where the host's time goes between an opening synchronize and the first kernel of a loop's first step -- the
loader's hand-out, the step wrapper up to the foreign call, the foreign call itself (its first launch goes out within a few us of
its start) -- for the first three steps of ten regions, with the loader's look-ahead builder queued by the step (kick) or at the
next hand-out.   python scripts/dev/first_step_probe.py"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
from p_companion_amd import _lib, ops
from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
from p_companion_amd.product2vec import FusedAdam, Product2Vec

dev = torch.device("cuda:0")
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0,
                      BATCH_SIZE=4096, LEARNING_RATE=1e-3, DEVICE=dev)
bpg = generate_scaled_bpg(100_000, 100, seed=0)
table = bpg.cuda(dev)["features"]
stamps = {}
L = _lib.lib()
real = L.pc_p2v_train_step_unique_rows


def timed(*a):
    stamps["c0"] = time.perf_counter()
    rc = real(*a)
    stamps["c1"] = time.perf_counter()
    return rc


for kick in (True, False):
    torch.manual_seed(0)
    model = Product2Vec(cfg).to(dev).train()
    opt = FusedAdam(model, lr=1e-3)
    loader = SimilarityIndexLoader(bpg, 4096, shuffle=True, sampler="philox", seed=1, drop_last=True, device=dev, reuse_buffers=True)
    loader.kick_after_step = kick

    def batches():
        while True:
            for b in loader:
                yield b
    it = batches()
    L.pc_p2v_train_step_unique_rows = timed
    try:
        for _ in range(30):
            model.train_step_indexed(table, next(it), optimizer=opt)
        rows = []
        for r in range(10):
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            per = []
            t_reg = time.perf_counter()
            ev0.record()
            for i in range(20):
                t0 = time.perf_counter()
                b = next(it)
                t1 = time.perf_counter()
                model.train_step_indexed(table, b, optimizer=opt)
                t2 = time.perf_counter()
                per.append((t1 - t0, stamps["c0"] - t1, stamps["c1"] - stamps["c0"], t2 - stamps["c1"]))
            ev1.record()
            torch.cuda.synchronize()
            rows.append((per, time.perf_counter() - t_reg, ev0.elapsed_time(ev1)))
        us = lambda x: f"{1e6 * x:6.1f}"
        print(f"kick={kick}: per step host us [hand-out | wrapper before the call | the call | after the call]")
        for i in (0, 1, 2, 10):
            cols = list(zip(*[r[0][i] for r in rows]))
            print(f"  step {i:2d}: " + " | ".join(us(statistics.median(c)) for c in cols))
        print(f"  region of 20 steps: wall {1e3 * statistics.median(r[1] for r in rows) / 20:.4f} ms/step, device events "
              f"{statistics.median(r[2] for r in rows) / 20:.4f} ms/step", flush=True)
    finally:
        L.pc_p2v_train_step_unique_rows = real
