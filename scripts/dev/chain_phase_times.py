"""Developer tool: shader-clock stamps of the few-row chain kernel's phases (workgroup 0).
Needs  PC_EXTRA_HIPCC_FLAGS=-DPC_CHAIN_TIMING python -m p_companion_amd.build --force  (NOT a production build)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from p_companion_amd import _lib, ops
from oracle import p2v_oracle
lib = _lib.lib()
fn = lib.pc_debug_chain_timing
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
fn.restype = ctypes.c_int
B, N, D = 4096, 32, 128
st = {k: v.cuda() for k, v in p2v_oracle.init_state(0).items()}
q = torch.randn(B, D, device="cuda"); kv = torch.randn(B, N, D, device="cuda")
for _ in range(3):
    out, sv = ops.attention_forward(st, q, kv)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
fn(buf, 1)
out, sv = ops.attention_forward(st, q, kv)
torch.cuda.synchronize()
fn(buf, 0)
v = [int(x) for x in buf if x]
print("stamps:", len(v))
print("deltas (clk @100MHz or shader clk):", [v[i + 1] - v[i] for i in range(len(v) - 1)])
