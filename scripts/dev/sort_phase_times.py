"""Developer probe (GPU box; needs a -DPC_SORT_TIMING build: PC_EXTRA_HIPCC_FLAGS=-DPC_SORT_TIMING python -m p_companion_amd.build --force):
shader clocks between the phases of table_sort_kernel for the complementary table's list of one fused step at T = 34800."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from types import SimpleNamespace
import torch
from p_companion_amd import _lib
from p_companion_amd.p_companion import PCompanion

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--bench":
    # the benchmark's own workload (bench.py --phase joint --types 34800 --dropout 0.1: its catalogue, model and batches)
    import contextlib, io
    sys.argv = ["bench.py", "--phase", "joint", "--types", "34800", "--dropout", "0.1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                "--no-ref-types", "--no-dropout-legs"]
    sys.path.insert(0, ROOT)
    import bench
    with contextlib.redirect_stdout(io.StringIO()):
        try:
            bench.main()
        except SystemExit:
            pass
else:
    T, B, K, p = 34800, 4096, 3, float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
    SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=p, MARGIN=1.0, ALPHA=0.8,
                          NUM_COMP_TYPES=K, NUM_TYPES=T, DEVICE=torch.device("cuda"), LEARNING_RATE=1e-3)
    torch.manual_seed(SEED)
    g = torch.Generator().manual_seed(0)
    m = PCompanion(cfg, torch.randn(2000, 128, generator=g)).cuda().train()
    b = {"query_idx": torch.randint(0, 2000, (B,), generator=g, dtype=torch.int32).cuda(), "query_types": torch.randint(0, 100, (B,), generator=g).cuda(),
         "positive_types": torch.randint(0, 100, (B, 1), generator=g).cuda(), "negative_types": torch.randint(0, 100, (B, 1), generator=g).cuda(),
         "positive_items": torch.randn(B, 128, generator=g).cuda(), "negative_items": torch.randn(B, 128, generator=g).cuda()}
    for _ in range(5):
        m.train_step(b)
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 1024)()
assert L.pc_debug_sort_timing(out) == 0
t = list(out)
# stamps 0..10 of table_sort_kernel (PC_ST in csrc/joint_fused.hip): the phase that ENDS at stamp i
names = ["read the list + clear the bins", "count the range's bins (LDS atomics)", "per-thread bin sums", "block scan (rows, runs, lists)",
         "run starts back into the bins + the workgroup's list atomics", "barrier", "row base", "placement (returning LDS atomics)",
         "run table + run lists", "write-out of the order"]
live = [b for b in range(64) if t[16 * b]]
nr = len(live) // 2
for b in live:
    seq = t[16 * b:16 * b + 11]
    print("workgroup %2d (list %d, range %2d): total %d clk: " % (b, b // nr, b % nr, seq[-1] - seq[0])
          + ", ".join("%s %d" % (names[i - 1], seq[i] - seq[i - 1]) for i in range(1, 11)))
# the eight grid regions of table_segsum_kernel: runs of up to 4 rows / 5..64 / 65..256 / parts of longer ones, per table
reg = ["short c", "medium c", "long c", "giant c", "short q", "medium q", "long q", "giant q"]
print("table_segsum_kernel, all steps: " + "; ".join("%s: longest wave %d clk, %d waves with work, longest run %d rows" % (reg[i], t[1000 + i], t[1008 + i], t[1016 + i]) for i in range(8)))
print("a medium run of >= 48 rows: entry + count at %d clk, row numbers in the bitmap at %d, queue at %d, rows added at %d" % tuple(t[900:904]))
