"""Developer probe: the fused Product2Vec step on ONE prebuilt batch (no loader kernels on the side stream) against the
same step fed by the throughput loader: what the side stream's batch construction costs the step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace
from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
from p_companion_amd.product2vec import FusedAdam, Product2Vec
dev = torch.device("cuda")
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0,
                      BATCH_SIZE=4096, LEARNING_RATE=1e-3, DEVICE=dev)
bpg = generate_scaled_bpg(100_000, 100, seed=0)
table = bpg.cuda(dev)["features"]
torch.manual_seed(0)
m = Product2Vec(cfg).to(dev).train()
opt = FusedAdam(m, lr=1e-3)
ld = SimilarityIndexLoader(bpg, 4096, seed=1, drop_last=True, device=dev, reuse_buffers=os.environ.get("PROBE_RING", "1") == "1")
it = iter(SimilarityIndexLoader(bpg, 4096, seed=1, drop_last=True, device=dev))
batches = [next(it) for _ in range(8)]
for b in batches:
    int(b["neighbor_compact"]["n_unique"])
torch.cuda.synchronize()
def run(n, src):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        m.train_step_indexed(table, src(i)); opt.step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
if os.environ.get("PROBE_MODE", "both") in ("both", "fixed"):
  run(20, lambda i: batches[i % 8])
  print("fixed batches  ms/step:", round(run(200, lambda i: batches[i % 8]), 4))
def gen():
    while True:
        for b in ld:
            yield b
g = gen()
if os.environ.get("PROBE_MODE", "both") in ("both", "loader"):
  run(20, lambda i: next(g))
  print("through loader ms/step:", round(run(200, lambda i: next(g)), 4))
# host-side time of the three calls of a loader-fed step (a call that blocks on the device shows up here)
import statistics
tl, ts, to = [], [], []
torch.cuda.synchronize()
t_all = time.perf_counter()
for i in range(300):
    t0 = time.perf_counter(); b = next(g); t1 = time.perf_counter()
    m.train_step_indexed(table, b); t2 = time.perf_counter()
    opt.step(); t3 = time.perf_counter()
    tl.append(t1 - t0); ts.append(t2 - t1); to.append(t3 - t2)
t_enq = time.perf_counter() - t_all
torch.cuda.synchronize()
t_tot = time.perf_counter() - t_all
f = lambda v: "mean %.3f  p50 %.3f  p90 %.3f  max %.3f ms" % (1e3 * statistics.mean(v), 1e3 * statistics.median(v), 1e3 * sorted(v)[int(0.9 * len(v))], 1e3 * max(v))
print("loader next():", f(tl)); print("train_step   :", f(ts)); print("opt.step     :", f(to))
print("enqueue all %.3f ms/step, total %.3f ms/step" % (1e3 * t_enq / 300, 1e3 * t_tot / 300))
