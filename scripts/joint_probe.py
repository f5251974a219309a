"""Developer probe: joint-step throughput at config 3 (100k products, T=100, B=4096)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
from p_companion_amd.data import generate_scaled_bpg, ComplementaryIndexDataset, ComplementaryIndexLoader
from p_companion_amd.p_companion import PCompanion
from p_companion_amd.product2vec import FusedAdam
dev = torch.device("cuda"); B = int(os.environ.get("B", 4096)); T = int(os.environ.get("T", 100))
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=0.0, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=T, DEVICE=dev)
bpg = generate_scaled_bpg(100000, min(T, 100), 0)
loader = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), B, shuffle=True)
model = PCompanion(cfg, bpg.cuda()["features"]).to(dev).train(); opt = FusedAdam(model)
batches = []
for i, b in enumerate(loader):
    if b["query_idx"].numel() == B: batches.append(b)
    if len(batches) == 8: break
for b in batches[:3]: model.train_step(b); opt.step()
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
for r in range(10):
    for b in batches: losses, _ = model.train_step(b); opt.step(); n += 1
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"joint fused: {dt*1e3:.3f} ms/step  {B/dt:.0f} triplets/s  loss {float(losses[0]):.4f}")
