"""Developer probe: does a LOWER-priority builder queue (or a higher-priority training queue) take the loader's builders out of the
training kernels' way?  Per-step device times of the Product2Vec loop with (a) everything at the default priority (the shipped
arrangement), (b) the loader's side stream at the device's least priority, (c) the training loop on a greatest-priority stream."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
from p_companion_amd.product2vec import FusedAdam, Product2Vec

dev = torch.device("cuda:0")
torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
print("priority range: least", lo.value, "greatest", hi.value, flush=True)

def raw_stream(priority):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, priority) == 0          # 1 = hipStreamNonBlocking
    return torch.cuda.ExternalStream(s.value, device=dev)

cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0,
                      BATCH_SIZE=4096, LEARNING_RATE=1e-3, DEVICE=dev)
bpg = generate_scaled_bpg(100_000, 100, seed=0)
table = bpg.cuda(dev)["features"]

def run(label, side_priority=None, main_priority=None, steps=240):
    torch.manual_seed(0)
    model = Product2Vec(cfg).to(dev).train()
    opt = FusedAdam(model, lr=1e-3)
    loader = SimilarityIndexLoader(bpg, 4096, shuffle=True, sampler="philox", seed=1, drop_last=True, device=dev, reuse_buffers=True)
    if side_priority is not None:
        loader._side = raw_stream(side_priority)
    main = raw_stream(main_priority) if main_priority is not None else torch.cuda.current_stream(dev)
    def batches():
        while True:
            for b in loader:
                yield b
    it = batches()
    with torch.cuda.stream(main):
        for _ in range(40):
            model.train_step_indexed(table, next(it), optimizer=opt)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        main.synchronize()
        evs[0].record(main)
        for i in range(steps):
            model.train_step_indexed(table, next(it), optimizer=opt)
            evs[i + 1].record(main)
        main.synchronize()
    torch.cuda.synchronize()
    ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(4, steps))
    tot = evs[4].elapsed_time(evs[steps]) / (steps - 4)
    print(f"{label}: mean {tot:.4f} ms/step, median {ms[len(ms) // 2]:.4f}, p25 {ms[len(ms) // 4]:.4f}, p90 {ms[int(len(ms) * 0.9)]:.4f}", flush=True)

for rep in range(2):
    run("default / default")
    run("side queue at the least priority", side_priority=lo.value)
    run("training loop at the greatest priority", main_priority=hi.value)
