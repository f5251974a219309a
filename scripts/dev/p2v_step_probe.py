"""Developer probe: host time per iteration of the P2V training loop across epoch boundaries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
from p_companion_amd.product2vec import FusedAdam, Product2Vec
dev = torch.device("cuda")
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0, BATCH_SIZE=4096, LEARNING_RATE=1e-3, DEVICE=dev)
bpg = generate_scaled_bpg(100000, 100, seed=0)
torch.manual_seed(0)
model = Product2Vec(cfg).to(dev).train()
opt = FusedAdam(model, lr=1e-3)
table = bpg.cuda(dev)["features"]
ld = SimilarityIndexLoader(bpg, 4096, shuffle=True, sampler="philox", seed=1, drop_last=True, device=dev)
def batches():
    while True:
        for b in ld:
            yield b
it = batches()
for _ in range(10):
    model.train_step_indexed(table, next(it)); opt.step()
torch.cuda.synchronize()
ts = []
T0 = t0 = time.perf_counter()
for i in range(160):
    b = next(it)
    t1 = time.perf_counter()
    model.train_step_indexed(table, b); opt.step()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1)); t0 = t2
torch.cuda.synchronize()
tot = time.perf_counter() - T0
print(f"160 steps: {tot*1e3/160:.3f} ms/step")
worst = sorted(range(160), key=lambda i: -(ts[i][0] + ts[i][1]))[:6]
for i in sorted(worst):
    print(f"  iter {i}: next(it) {ts[i][0]*1e3:.2f} ms, step {ts[i][1]*1e3:.2f} ms")
med = sorted(a + b for a, b in ts)[80]
print(f"  median iteration {med*1e3:.3f} ms")
