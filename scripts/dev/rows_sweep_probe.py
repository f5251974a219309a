"""Developer probe: how the persistent NT kernels' time follows the ROW COUNT (tile rounds against workgroup slots).

For R in a sweep: the FFN forward (Linear0 + finalize + Linear3 + Linear5) as one timed call, the plain 256 x 256 and 128 x 256
products alone.  A staircase (flat between multiples of slots x 128 rows, a jump behind them) says a launch's time is its LAST
round's, i.e. what a better balance of the last round could recover."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from p_companion_amd import ops

dev = "cuda"
torch.manual_seed(0)
params = {k: (torch.randn(s, device=dev) * 0.05) for k, s in zip(ops.P2V_KEYS, ops.p2v_shapes(128))}
params["ffn.1.weight"].fill_(1.0)
params["ffn.1.running_mean"] = torch.zeros(256, device=dev)
params["ffn.1.running_var"] = torch.ones(256, device=dev)
params["ffn.1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device=dev)
table = torch.randn(100000, 128, device=dev)


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


w256 = torch.randn(256, 256, device=dev) * 0.05
w128 = torch.randn(128, 256, device=dev) * 0.05
b256 = torch.zeros(256, device=dev); b128 = torch.zeros(128, device=dev)
print("R tiles | ffn_fwd us | lin 256x256 us | lin 128x256 us | ffn us per 1k rows", flush=True)
rows = sorted(set(list(range(40960, 106497, 4096)) + [86400, 64000, 65536, 66048, 67584, 98304, 99328]))
if len(sys.argv) > 1:
    rows = [int(a) for a in sys.argv[1:]]
for R in rows:
    idx = torch.randint(0, 100000, (R,), device=dev, dtype=torch.int32)
    seg = [0, 4096, R - 5 * 4096 - 4096, R - 5 * 4096]
    x = torch.randn(R, 256, device=dev) * 0.1
    t_ffn = timed(lambda: ops.ffn_forward_train(params, table, idx, R, seg))
    t_256 = timed(lambda: ops.linear_forward(x, w256, b256))
    t_128 = timed(lambda: ops.linear_forward(x, w128, b128))
    print(f"{R:7d} {(R + 127) // 128:5d} | {t_ffn:8.1f} | {t_256:8.1f} | {t_128:8.1f} | {t_ffn / R * 1e3:6.3f}", flush=True)
