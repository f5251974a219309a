"""Developer probe: gemm_nt steady-state vs per-tile overhead (K sweep at fixed tile count)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p_companion_amd import ops

def run(M, N, K, act=0, reps=20):
    x = torch.randn(M, K, device="cuda") * 0.1
    w = torch.randn(N, K, device="cuda") * 0.1
    b = torch.zeros(N, device="cuda")
    for _ in range(3): ops.linear_forward(x, w, b, act=act)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): y = ops.linear_forward(x, w, b, act=act)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    tf = 2.0 * M * N * K / us / 1e6
    print(f"M={M:7d} N={N:4d} K={K:5d} act={act}: {us:8.1f} us  {tf:6.1f} TFLOP/s", flush=True)

for M in (65536, 131072, 117449):
    for K in (128, 256, 1024, 4096):
        run(M, 256, K)
run(65536, 128, 256); run(65536, 128, 4096)
run(131072, 256, 128, act=1); run(131072, 256, 256, act=1)
