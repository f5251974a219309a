"""Developer probe: why steps 5..24 of a run are slower than later ones (the driver's region is exactly those).  Per-step device
times (one event behind each step) for:
  (a) a fresh model, lr 1e-3                       -- the bench's situation
  (b) a fresh model, lr 0                          -- no training progress: does the step still get faster?
  (c) the TRAINED weights of (a) under a NEW loader, a NEW optimizer and new workspaces' first touches
      -- warm-up of code paths without the early-training operands
python scripts/dev/early_steps_probe.py"""
import os, sys, time, copy
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


def run(label, model, lr, n=60, seed=1):
    opt = FusedAdam(model, lr=lr)
    loader = SimilarityIndexLoader(bpg, 4096, shuffle=True, sampler="philox", seed=seed, drop_last=True, device=dev, reuse_buffers=True)

    def batches():
        while True:
            for b in loader:
                yield b
    it = batches()
    losses = []
    for _ in range(5):
        model.train_step_indexed(table, next(it), optimizer=opt)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    evs[0].record()
    for i in range(n):
        losses.append(model.train_step_indexed(table, next(it), optimizer=opt))
        evs[i + 1].record()
    torch.cuda.synchronize()
    ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
    ls = [float(l) for l in losses]
    grp = lambda a, b: sum(ms[a:b]) / (b - a)
    print(f"{label}: mean ms/step of steps 6-24 {grp(1, 20):.4f} | 25-44 {grp(20, 40):.4f} | 45-64 {grp(40, 60):.4f}; loss {ls[0]:.3f} -> {ls[19]:.3f} -> {ls[-1]:.3f}")
    print("   per step: " + " ".join(f"{m:.3f}" for m in ms[:30]), flush=True)


torch.manual_seed(0)
m_a = Product2Vec(cfg).to(dev).train()
init = copy.deepcopy(m_a.state_dict())
run("(a) fresh model, lr 1e-3", m_a, 1e-3)
for _ in range(2):
    pass
m_b = Product2Vec(cfg).to(dev).train()
m_b.load_state_dict(init)
run("(b) fresh model, lr 0   ", m_b, 0.0)
# (a)'s model has now seen 65 steps; train it on to ~300 with another loader, then time it under a NEW loader / optimizer
opt = FusedAdam(m_a, lr=1e-3)
ld = SimilarityIndexLoader(bpg, 4096, shuffle=True, sampler="philox", seed=7, drop_last=True, device=dev, reuse_buffers=True)
k = 0
while k < 240:
    for b in ld:
        m_a.train_step_indexed(table, b, optimizer=opt)
        k += 1
torch.cuda.synchronize()
time.sleep(1.0)                                         # (an idle device in front of the region, as at process start)
run("(c) trained weights, new loader + optimizer", m_a, 1e-3, seed=11)
m_d = Product2Vec(cfg).to(dev).train()
m_d.load_state_dict(init)
time.sleep(1.0)
run("(d) fresh weights again (after everything is warm)", m_d, 1e-3, seed=13)
