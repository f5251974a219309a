"""Developer probe: time pc_linear_forward at FFN shapes (dbg variants via PC_NT_DBG)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from p_companion_amd import ops
R = 159744
for (k, n) in ((128, 256), (256, 256), (256, 128)):
    x = torch.randn(R, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.05; b = torch.randn(n, device="cuda")
    for _ in range(3): ops.linear_forward(x, w, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.linear_forward(x, w, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"dbg={os.environ.get('PC_NT_DBG','0')} K={k} N={n}: {dt*1e6:8.1f} us  {2*R*k*n/dt/1e12:6.1f} TF/s", flush=True)
