"""Developer probe: where the host spends its time inside the throughput loader's next() (which call blocks)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace
from p_companion_amd import ops
from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
from p_companion_amd.product2vec import FusedAdam, Product2Vec
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label] += time.perf_counter() - t; cnt[label] += 1; return r
    setattr(obj, name, g)
wrap(torch.cuda.Event, "synchronize", "Event.synchronize")
wrap(torch.cuda.Event, "record", "Event.record")
wrap(torch.cuda.Event, "query", "Event.query")
wrap(torch.cuda.Stream, "wait_event", "Stream.wait_event")
wrap(ops, "build_similarity_batch_unique", "build_unique (ctypes + views)")
dev = torch.device("cuda")
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0,
                      BATCH_SIZE=4096, LEARNING_RATE=1e-3, DEVICE=dev)
bpg = generate_scaled_bpg(100_000, 100, seed=0)
table = bpg.cuda(dev)["features"]
m = Product2Vec(cfg).to(dev).train(); opt = FusedAdam(m, lr=1e-3)
ld = SimilarityIndexLoader(bpg, 4096, seed=1, drop_last=True, device=dev, reuse_buffers=True)
def gen():
    while True:
        for b in ld: yield b
g = gen()
for _ in range(30):
    m.train_step_indexed(table, next(g)); opt.step()
torch.cuda.synchronize(); acc.clear(); cnt.clear()
tn = ts = 0.0
t_all = time.perf_counter()
for _ in range(200):
    t0 = time.perf_counter(); b = next(g); t1 = time.perf_counter()
    m.train_step_indexed(table, b); opt.step(); t2 = time.perf_counter()
    tn += t1 - t0; ts += t2 - t1
torch.cuda.synchronize()
print("ms/step total %.3f | next() %.3f | step+opt %.3f" % (1e3 * (time.perf_counter() - t_all) / 200, 1e3 * tn / 200, 1e3 * ts / 200))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-32s %8.3f ms/step  (%d calls/step)" % (k, 1e3 * v / 200, cnt[k] / 200))
# device-side latency of one builder (first kernel start -> last kernel end on the builder's stream), while training runs
side = ld._side
lat = []
orig_build = ops.build_similarity_batch_unique
def timed(*a, **k):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(torch.cuda.current_stream()); r = orig_build(*a, **k); e.record(torch.cuda.current_stream())
    lat.append((s, e)); return r
ops.build_similarity_batch_unique = timed
ld.ops.build_similarity_batch_unique = timed
for _ in range(100):
    m.train_step_indexed(table, next(g)); opt.step()
torch.cuda.synchronize()
v = sorted(s.elapsed_time(e) for s, e in lat)
print("builder device latency ms: p10 %.3f p50 %.3f p90 %.3f" % (v[len(v) // 10], v[len(v) // 2], v[9 * len(v) // 10]))
