"""bench.py --gpus N without a launcher starts its ranks itself (two ranks on the one card over gloo).  Needs an MI355X."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ bench.py --gpus N starts its N ranks itself
@pytest.mark.timeout(900)
def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's N > 1 form if it calls the script like the N = 1 one):
    the script starts two ranks itself, rank 0's line reports both.  Rehearsed on the one card of this box (PC_FORCE_DEVICE=0)
    over gloo -- two ranks cannot share a GPU under RCCL; RCCL itself is rehearsed with one rank in tests/test_gpu_rccl.py."""
    env = dict(os.environ, PC_DIST_BACKEND="gloo", PC_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PC_DIST_FORCE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--products", "20000", "--batch", "512", "--no-cpu-baseline", "--no-sustained", "--no-large",
                          "--no-dropout-legs", "--no-ref-types"], env=env, capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 1024 and line["config"]["parallelism"] == "dp2"
    rccl = dict(line["rccl"])
    exchange = rccl.pop("exchange")                     # (ABI 6: the gradient exchange through the library's slot; gloo behind it here)
    timeouts = rccl.pop("timeouts_s")                   # first-contact deadlines, all well inside a 10-minute harness limit
    assert timeouts["probe"] <= 90 and timeouts["process_group"] <= 300 and timeouts["secondary_leg"] <= 180
    rccl.pop("native_probe", None)
    assert "[bench] rccl {" in out.stderr               # the block is out before the first timed leg
    assert rccl == {"backend": "gloo", "world": 2, "ranks_seen": [0, 1], "launcher": "self"} and "gloo" in exchange
    assert "pc_joint_train_epoch_dp" in line["joint"]["config"]["launch"]
    assert line["value"] > 0 and line["joint"]["value"] > 0 and line["joint"]["config"]["parallelism"] == "dp2"
    # more ranks than GPUs, not a rehearsal: refused with a message, before any rank starts
    env.pop("PC_FORCE_DEVICE")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1)], env=env,
                         capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert bad.returncode != 0 and "visible GPUs" in bad.stderr and not bad.stdout.strip()


@pytest.mark.timeout(900)
def test_bench_world2_drives_the_large_legs():
    """The N > 1 branches of the `large_catalogue` legs, which only an 8-GPU run would otherwise reach: configs[3] (10 M products,
    sharded lookup), configs[4] (100 M x 256 row-sharded over the two ranks, Zipf negatives, the replicated hot set) and the
    hot_set comparison -- two ranks on the one card of this box over gloo (a rehearsal of the code path, not a timing)."""
    env = dict(os.environ, PC_DIST_BACKEND="gloo", PC_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PC_DIST_FORCE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--phase", "p2v",
                          "--no-cpu-baseline", "--no-sustained", "--no-dropout-legs"], env=env, capture_output=True, text=True,
                         timeout=800, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    lc = line["large_catalogue"]
    assert line["n_gpus"] == 2 and "error" not in lc["config3"] and "error" not in lc["config4"], lc
    c4 = lc["config4"]["sharded_lookup"]
    assert c4["hot_rows"] == 1024 and c4["hot_rows_served_per_batch"] > 1000          # the Zipf head served from the replica
    assert lc["config3"]["sharded_lookup"]["hot_rows"] == 0                            # uniform negatives: no hot set
    hs = lc["hot_set"]
    assert hs["with"]["sharded_lookup"]["hot_rows_served_per_batch"] > 1000 and hs["without"]["sharded_lookup"]["hot_rows_served"] is None
    assert abs(hs["with"]["final_loss"] - hs["without"]["final_loss"]) < 1e-6          # the same rows: the same training


@pytest.mark.timeout(600)
def test_bench_world2_leg_watchdog_keeps_the_headline():
    """N > 1: a secondary leg that does not finish inside its deadline (here: 0.25 s, which the joint leg's set-up alone exceeds)
    must not cost the headline -- rank 0 prints the line with the legs measured so far and the leg reported as an error, every
    rank leaves, the launcher exits 0."""
    env = dict(os.environ, PC_DIST_BACKEND="gloo", PC_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PC_BENCH_LEG_DEADLINE_S="0.25")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PC_DIST_FORCE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--products", "20000", "--batch", "512", "--no-cpu-baseline", "--no-sustained", "--no-large",
                          "--no-dropout-legs", "--no-ref-types"], env=env, capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0                      # the headline leg ran first and is in the line
    assert "did not finish within 0.25 s" in line["joint"]["error"], line.get("joint")
    assert "still running after 0.25 s" in out.stderr


@pytest.mark.timeout(600)
def test_bench_world2_total_budget_skips_secondary_legs():
    """N > 1: once the process is older than PC_BENCH_TOTAL_BUDGET_S no further secondary leg is started (the harness ends the job
    at its own limit, line or no line); the ranks agree on it, the headline is in the line, the skipped legs say so, exit 0."""
    env = dict(os.environ, PC_DIST_BACKEND="gloo", PC_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PC_BENCH_TOTAL_BUDGET_S="0.5")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PC_DIST_FORCE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--products", "20000", "--batch", "512", "--no-cpu-baseline", "--no-sustained", "--no-large",
                          "--no-ref-types"], env=env, capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["joint"]["error"].startswith("skipped:") and line["joint_dropout_0p1"]["error"].startswith("skipped:"), line.get("joint")
    assert "not started" in out.stderr
