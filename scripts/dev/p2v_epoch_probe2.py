"""Developer probe: the bench's loop shape (count read back every step) across an epoch boundary, with the loader's
__iter__ pieces timed by monkey-patching."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch, numpy as np
from p_companion_amd import data as D
from p_companion_amd.product2vec import FusedAdam, Product2Vec
dev = torch.device("cuda")
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0, BATCH_SIZE=4096, LEARNING_RATE=1e-3, DEVICE=dev)
bpg = D.generate_scaled_bpg(100000, 100, seed=0)
torch.manual_seed(0)
model = Product2Vec(cfg).to(dev).train()
opt = FusedAdam(model, lr=1e-3)
table = bpg.cuda(dev)["features"]
ld = D.SimilarityIndexLoader(bpg, 4096, shuffle=True, sampler="philox", seed=1, drop_last=True, device=dev)
def batches():
    while True:
        for b in ld:
            yield b
it = batches()
for _ in range(10):
    b = next(it); int(b["neighbor_compact"]["n_unique"]); model.train_step_indexed(table, b); opt.step()
torch.cuda.synchronize()
ts = []
T0 = t0 = time.perf_counter()
for i in range(150):
    b = next(it)
    t1 = time.perf_counter()
    int(b["neighbor_compact"]["n_unique"])
    t2 = time.perf_counter()
    model.train_step_indexed(table, b); opt.step()
    t3 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1, t3 - t2)); t0 = t3
torch.cuda.synchronize()
tot = time.perf_counter() - T0
print(f"150 steps: {tot*1e3/150:.3f} ms/step")
worst = sorted(range(150), key=lambda i: -sum(ts[i]))[:5]
for i in sorted(worst):
    print(f"  iter {i}: next(it) {ts[i][0]*1e3:.2f} ms, count {ts[i][1]*1e3:.2f} ms, step {ts[i][2]*1e3:.2f} ms")
