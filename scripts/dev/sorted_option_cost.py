"""Developer probe (GPU box): the joint step at NUM_TYPES = 34800 WITHOUT dropout, the per-workgroup LDS tables
(pc_set_option(PC_OPT_SORTED_TABLE_GRADIENTS, 0): rounds 1-4's form) against the sorted form (1: the default since round 5):
python scripts/dev/sorted_option_cost.py"""
import contextlib, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from p_companion_amd import _lib
import bench
for value in (0, 1, 0, 1):
    assert _lib.lib().pc_set_option(_lib.PC_OPT_SORTED_TABLE_GRADIENTS, value) == 0
    sys.argv = ["bench.py", "--phase", "joint", "--types", "34800", "--no-cpu-baseline", "--no-ref-types", "--no-dropout-legs"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        try:
            bench.main()
        except SystemExit:
            pass
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    print("sorted option", value, "ms_per_step", d["ms_per_step"], flush=True)
