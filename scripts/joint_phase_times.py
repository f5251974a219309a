"""Developer tool: where a joint_tile_kernel workgroup spends its time.  Needs a library built with
PC_EXTRA_HIPCC_FLAGS=-DPC_JOINT_TIMING python -m p_companion_amd.build --force
Runs a few fused joint steps (B=4096, T=100) and prints the 100 MHz wall-clock deltas between the phase boundaries of
wave 0 of workgroups 0 and 128 in the last launch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from types import SimpleNamespace
import torch
from p_companion_amd import _lib
from p_companion_amd.p_companion import PCompanion
from p_companion_amd.product2vec import FusedAdam
lib = _lib.lib()
fn = lib.pc_debug_joint_timing
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
fn.restype = ctypes.c_int
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B, P = 4096, 100000
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=0.0, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=T, DEVICE="cuda")
g = torch.Generator().manual_seed(0)
m = PCompanion(cfg, torch.randn(P, 128, generator=g)).to("cuda").train()
o = FusedAdam(m)
b = {"query_idx": torch.randint(0, P, (B,), generator=g, dtype=torch.int32).cuda(), "query_types": torch.randint(0, min(T, 100), (B,), generator=g).cuda(),
     "positive_types": torch.randint(0, min(T, 100), (B, 1), generator=g).cuda(), "negative_types": torch.randint(0, min(T, 100), (B, 1), generator=g).cuda(),
     "positive_items": torch.randn(B, 128, generator=g).cuda(), "negative_items": torch.randn(B, 128, generator=g).cuda()}
for _ in range(10):
    m.train_step(b, optimizer=o)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
fn(buf)
names = ["ids+weight requests", "row gathers", "A: h || pi 0-3", "B: c, pi 4-7", "C: sims", "D: top-k", "gather E_c rows", "E: tp", "F: per-sample losses", "G: dce, dh", "H: dt", "W: hand-off + bias sums", "W: d itm_w", "W: d typ_w, d dec_w, d enc_w", "W: tables"]
for wg in (0, 1):
    t = [buf[16 * wg + i] for i in range(16)]
    print(f"workgroup {0 if wg == 0 else 128}: total {(t[15] - t[0]) / 100:.2f} us")
    for i, n in enumerate(names):
        print(f"   {n:28s} {(t[i + 1] - t[i]) / 100:7.2f} us")
