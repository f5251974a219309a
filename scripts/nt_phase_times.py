"""Developer tool: where a persistent gemm_nt_kernel workgroup spends its time (K loop vs epilogue).
Needs a library built with  PC_EXTRA_HIPCC_FLAGS=-DPC_NT_TIMING python -m p_companion_amd.build --force
Runs bench.py's step a few times, then prints per fused variant the mean shader clocks per tile in each phase."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from p_companion_amd import _lib
lib = _lib.lib()
fn = lib.pc_debug_nt_timing
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
fn.restype = ctypes.c_int
sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-sustained", "--phase", "p2v"]
import runpy
buf = (ctypes.c_ulonglong * 128)()
torch.cuda.init()
torch.zeros(1, device="cuda")
fn(buf, 1)
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
torch.cuda.synchronize()
fn(buf, 0)
names = {0: "plain", 1: "tanh", 3: "dtanh (dZ2)", 6: "plain + BN sums (Linear0)", 10: "dtanh_bn + BN-bwd sums (dZ1)", 13: "BN+tanh prologue, tanh (Linear3)"}
for i in range(32):
    k, e, n, wgs = buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]
    if n:
        print(f"{names.get(i, str(i)):36s} tiles/wg {n / wgs:5.2f}  K loop {k / n:9.0f} clk/tile  epilogue {e / n:9.0f} clk/tile  (epilogue share {e / (k + e):.2f})  per launch&wg: {(k + e) / wgs:9.0f} clk")
