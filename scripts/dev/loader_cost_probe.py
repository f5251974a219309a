"""Developer probe: what the device loader costs the fused step -- the same step over eight PREBUILT batches (nothing on the loader's
stream, no ring events) against the step fed by the loader, alternating on one box.   python scripts/dev/loader_cost_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
from p_companion_amd.product2vec import FusedAdam, Product2Vec

dev = torch.device("cuda:0")
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0,
                      BATCH_SIZE=4096, LEARNING_RATE=1e-3, DEVICE=dev)
bpg = generate_scaled_bpg(100_000, 100, seed=0)
table = bpg.cuda(dev)["features"]
torch.manual_seed(0)
m = Product2Vec(cfg).to(dev).train()
opt = FusedAdam(m, lr=1e-3)
it = iter(SimilarityIndexLoader(bpg, 4096, seed=1, drop_last=True, device=dev))          # fresh tensors: safe to keep
fixed = [next(it) for _ in range(8)]
for b in fixed:
    int(b["neighbor_compact"]["n_unique"]); b.pop("_after_step", None)
del it
ld = SimilarityIndexLoader(bpg, 4096, seed=1, drop_last=True, device=dev, reuse_buffers=True)


def gen():
    while True:
        for b in ld:
            yield b
g = gen()


def run(n, src):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        m.train_step_indexed(table, src(i), optimizer=opt)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


run(100, lambda i: next(g)); run(50, lambda i: fixed[i % 8])
for r in range(4):
    a = run(300, lambda i: fixed[i % 8])
    b = run(300, lambda i: next(g))
    print(f"round {r}: prebuilt batches {a:.4f} ms/step, through the loader {b:.4f} ms/step ({1e3 * (b - a):+.1f} us)", flush=True)
