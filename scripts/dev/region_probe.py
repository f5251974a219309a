"""Developer probe: the driver-flag region (--warmup 5 --steps 20) of bench.py's Product2Vec leg taken apart -- per-step device time
(one event behind each step), the wall clock of the region, and the idle time at its start; then the same 20 steps again after
more warm-up.   python scripts/dev/region_probe.py"""
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
opt = FusedAdam(model, lr=1e-3)
table = bpg.cuda(dev)["features"]
loader = SimilarityIndexLoader(bpg, 4096, shuffle=True, sampler="philox", seed=1, drop_last=True, device=dev, reuse_buffers=True)
def batches():
    while True:
        for b in loader:
            yield b
it = batches()
def step():
    b = next(it)
    return model.train_step_indexed(table, b, optimizer=opt)

def region(steps, label):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record()
    hosts = []
    for i in range(steps):
        h0 = time.perf_counter()
        step()
        evs[i + 1].record()
        hosts.append(time.perf_counter() - h0)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    print(f"{label}: wall {1e3 * wall / steps:.4f} ms/step (enqueue done after {1e3 * t_enq:.2f} ms of {1e3 * wall:.2f}); device per step: "
          + " ".join(f"{m:.3f}" for m in ms) + f" | host per step: " + " ".join(f"{1e3 * h:.2f}" for h in hosts), flush=True)

for _ in range(5):
    step()
region(20, "steps 5..24 (the driver's region)")
region(20, "steps 25..44")
for _ in range(150):
    step()
region(20, "steps 195..214")
region(60, "steps 215..274")
