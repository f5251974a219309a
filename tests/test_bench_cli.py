"""bench.py's command line on a box WITHOUT a GPU: the self-launching N-rank form must refuse, with a message and a non-zero
exit code, before it starts a single rank."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_without_two_devices_fails_loudly():
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this node has two GPUs: the refusal cannot be provoked")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PC_FORCE_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode != 0
    assert "--gpus 2 needs 2 visible GPUs" in r.stderr and not r.stdout.strip()
