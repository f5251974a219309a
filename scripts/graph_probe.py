"""Developer probe: eager vs hipGraph replay of pc_p2v_train_step + Adam."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from types import SimpleNamespace
from p_companion_amd import ops
from p_companion_amd.data import generate_scaled_bpg, SimilarityIndexLoader
from p_companion_amd.product2vec import Product2Vec, FusedAdam
dev = torch.device("cuda")
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0, DEVICE=dev)
bpg = generate_scaled_bpg(100000, 100, 0)
model = Product2Vec(cfg).to(dev).train(); opt = FusedAdam(model)
loader = SimilarityIndexLoader(bpg, 4096, seed=1, drop_last=True)
table = bpg.cuda()["features"]
it = iter(loader); b0 = next(it)
static = {k: v.clone() for k, v in b0.items()}
for _ in range(3):
    model.train_step_indexed(table, static); opt.step()
torch.cuda.synchronize()
def timeit(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def eager():
    model.train_step_indexed(table, static); opt.step()
print("eager ms", timeit(eager))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): eager()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    loss = model.train_step_indexed(table, static); opt.step()
print("graph ms", timeit(g.replay), "loss", float(loss))
