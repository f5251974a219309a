"""Developer probe: how far does the host run ahead of the GPU in the bench loop (enqueue time vs execution time)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
from p_companion_amd.data import generate_scaled_bpg, SimilarityIndexLoader
from p_companion_amd.product2vec import Product2Vec, FusedAdam

cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0,
                      DEVICE=torch.device("cuda"), LEARNING_RATE=1e-3)
bpg = generate_scaled_bpg(100000, 100, 0)
model = Product2Vec(cfg).to("cuda"); opt = FusedAdam(model, lr=1e-3); model.flatten_parameters()
loader = SimilarityIndexLoader(bpg, 4096, seed=1, drop_last=True)
table = bpg.cuda()["features"]
def batches():
    while True:
        t = time.perf_counter()
        for b in loader:
            yield b, time.perf_counter() - t
            t = time.perf_counter()
it = batches()
for _ in range(10):
    b, _ = next(it); model.train_step_indexed(table, b); opt.step()
torch.cuda.synchronize()
t0 = time.perf_counter(); loads = []
for i in range(130):
    b, tl = next(it); loads.append(tl)
    model.train_step_indexed(table, b); opt.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
loads = np.array(loads) * 1e3
print(f"host enqueue {1e3*(t1-t0)/130:.3f} ms/step, total {1e3*(t2-t0)/130:.3f} ms/step; loader next(): median {np.median(loads):.3f} ms, max {loads.max():.2f} ms at steps {np.argsort(-loads)[:3]}")
