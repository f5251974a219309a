"""Developer probe: gemm_tn steady state vs per-launch overhead (row sweep at fixed output shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p_companion_amd import ops

def run(R, No, Ni, reps=20):
    dy = torch.randn(R, No, device="cuda") * 0.1
    x = torch.randn(R, Ni, device="cuda") * 0.1
    for _ in range(3): ops.linear_backward_weight(dy, x, No, Ni)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.linear_backward_weight(dy, x, No, Ni)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"R={R:8d} No={No:4d} Ni={Ni:4d}: {us:8.1f} us  {2.0 * R * No * Ni / us / 1e6:6.1f} TFLOP/s (incl. slab reduce)", flush=True)

for R in (8192, 16384, 32768, 65536, 86400, 117449, 245760, 983040):
    run(R, 128, 256); run(R, 256, 256)
