"""Developer tool: the PCIe-inclusive rate of the drop-in DENSE path (the reference's own batch format).

bench.py times the index path: the feature table and the graph are resident in HBM and a batch is four small int32
arrays built on the device.  A caller that keeps the reference's DataLoader hands over dense host tensors instead
(data_loader.py:57-69,171-206: anchor [B,128], positive [B,128], negative [B,5,128], anchor_neighbors [B,N,128]) =
512 * (N + 7) bytes per triplet over PCIe, then the module-mode step (autograd over the HIP Functions: every slot
its own row, no identical-row merging).  This measures that loop body with pinned host batches:
    host -> device copies + forward + loss + backward + FusedAdam,  B = 4096, N = 32.
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from types import SimpleNamespace
import torch
from p_companion_amd.product2vec import FusedAdam, Product2Vec

B, N, steps, warm = 4096, 32, 30, 5
dev = torch.device("cuda")
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0,
                      BATCH_SIZE=B, LEARNING_RATE=1e-3, DEVICE=dev)
torch.manual_seed(0)
model = Product2Vec(cfg).to(dev).train()
opt = FusedAdam(model, lr=1e-3)
g = torch.Generator().manual_seed(1)
host = [{"anchor": torch.randn(B, 128, generator=g).pin_memory(), "positive": torch.randn(B, 128, generator=g).pin_memory(),
         "negative": torch.randn(B, 5, 128, generator=g).pin_memory(),
         "anchor_neighbors": torch.randn(B, N, 128, generator=g).pin_memory()} for _ in range(4)]
nbytes = sum(t.numel() * 4 for t in host[0].values())


def step(hb, copy=True):
    b = {k: v.to(dev, non_blocking=True) for k, v in hb.items()} if copy else hb
    loss = model.dense_loss(b)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss


from p_companion_amd.data import prefetch_to_device


def overlapped():
    def gen(count):
        for i in range(count):
            yield host[i % 4]
    for hb in prefetch_to_device(gen(warm), dev):
        step(hb, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for hb in prefetch_to_device(gen(steps), dev):
        step(hb, False)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    print(f"dense path, pinned host batches, next batch's copy on a side stream (train_model's loop): {ms:.3f} ms/step = "
          f"{B / ms * 1e3 / 1e6:.3f} M triplets/s")


overlapped()
for mode in ("pcie", "resident"):
    batches = host if mode == "pcie" else [{k: v.to(dev) for k, v in hb.items()} for hb in host]
    for i in range(warm):
        step(batches[i % 4], mode == "pcie")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(batches[i % 4], mode == "pcie")
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    print(f"dense path, batches {'from pinned host memory' if mode == 'pcie' else 'already on the device'}: {ms:.3f} ms/step = "
          f"{B / ms * 1e3 / 1e6:.3f} M triplets/s ({nbytes / 1e6:.1f} MB per batch{'; %.1f GB/s over PCIe if the step were copy only' % (nbytes / ms / 1e6) if mode == 'pcie' else ''})")
t = torch.empty(nbytes // 4, dtype=torch.float32).pin_memory()
d = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    d.copy_(t, non_blocking=True)
torch.cuda.synchronize()
print(f"host -> device copy alone: {nbytes * 10 / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
