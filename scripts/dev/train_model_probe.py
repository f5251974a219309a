"""Developer probe: Product2Vec.train_model (the reference's API: product2vec.py:113-170) against a bare loop of fused steps over the
same loader -- ms per step of each, on one box, alternating.   python scripts/dev/train_model_probe.py"""
import logging, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
from p_companion_amd.product2vec import FusedAdam, Product2Vec

logging.disable(logging.CRITICAL)
dev = torch.device("cuda:0")
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0,
                      BATCH_SIZE=4096, LEARNING_RATE=1e-3, DEVICE=dev)
bpg = generate_scaled_bpg(100_000, 100, seed=0)
table = bpg.cuda(dev)["features"]
torch.manual_seed(0)
model = Product2Vec(cfg).to(dev).train()
opt = FusedAdam(model, lr=1e-3)
loader = SimilarityIndexLoader(bpg, 4096, shuffle=True, sampler="philox", seed=1, drop_last=True, device=dev, reuse_buffers=True)
model.generate_all_embeddings = lambda bpg_: {}          # (the export at the end of train_model is not what is timed here)
steps_per_epoch = len(loader)
EPOCHS = 8


def api(direct=True):
    loader.yields_device_batches = direct           # False: through prefetch_to_device's staging wrapper (what train_model did before)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.train_model(loader, opt, num_epochs=EPOCHS)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / (EPOCHS * steps_per_epoch)


def bare():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(EPOCHS):
        for b in loader:
            model.train_step_indexed(table, b, optimizer=opt)
    torch.cuda.synchronize()
    model.eval()
    return 1e3 * (time.perf_counter() - t0) / (EPOCHS * steps_per_epoch)


api(); bare()
for r in range(4):
    print(f"round {r}: train_model {api():.4f} ms/step (through the staging wrapper {api(False):.4f}), bare loop of fused steps {bare():.4f} ms/step "
          f"({EPOCHS} epochs x {steps_per_epoch} steps)", flush=True)
