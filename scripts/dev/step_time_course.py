"""Per-step device time of the Product2Vec step from a cold start (one HIP event per step, read afterwards): what the first
few hundred steps of a process look like -- the driver's headline flags time steps 5..24."""
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
torch.manual_seed(0)
model = Product2Vec(cfg).to(dev).train()
opt = FusedAdam(model, lr=float(os.environ.get("PC_LR", "1e-3")))      # PC_LR=0: the parameters stand still -- does the step still get faster?
table = bpg.cuda(dev)["features"]
pre = float(os.environ.get("PREHEAT_MS", "0"))
if pre > 0:                                   # an unrelated busy kernel before the first step: is the ramp the clock's?
    x = torch.randn(8192, 8192, device=dev)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < pre:
        y = x @ x
    torch.cuda.synchronize()
loader = SimilarityIndexLoader(bpg, 4096, shuffle=True, sampler="philox", seed=1, drop_last=True, device=dev, reuse_buffers=True)
n = 400
evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
def batches():
    while True:
        for b in loader:
            yield b
it = batches()
evs[0].record()
host = []
for i in range(n):
    t0 = time.perf_counter()
    b = next(it)
    model.train_step_indexed(table, b)
    opt.step()
    evs[i + 1].record()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
for lo in (0, 5, 10, 15, 25, 45, 65, 100, 150, 200, 300):
    hi = min(n, lo + (5 if lo < 15 else 20 if lo < 100 else 50))
    print(f"steps {lo:3d}..{hi:3d}: device {sum(ms[lo:hi]) / (hi - lo):.4f} ms   host enqueue {1e3 * sum(host[lo:hi]) / (hi - lo):.4f} ms", flush=True)
