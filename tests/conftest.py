import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


# ---- world-2 job of tests/test_gpu_sharded.py: two ranks sharing the one card.  The children are started HERE, at
# session start, before this process makes its first GPU call (a process that has initialised the GPU must not be the
# one that launches further GPU programs on this pool); the test only collects their results.
_WORLD2 = {}


def pytest_collection_finish(session):
    """Runs after collection and BEFORE any test (hence before this process's first GPU call): the children are started
    only when the world-2 test is among the selected items (a run narrowed with -k / a node id leaves no orphans)."""
    if not any(item.name.startswith("test_sharded_step_world2_on_one_card") for item in session.items):
        return
    import socket
    import subprocess
    import tempfile
    import torch
    if torch.cuda.device_count() < 1:                  # (counting devices does not initialise the GPU)
        return
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    d = tempfile.mkdtemp(prefix="pc_world2_")
    worker = os.path.join(ROOT, "tests", "sharded_world2_worker.py")
    procs = []
    for rank in range(2):
        out = os.path.join(d, f"rank{rank}.json")
        log = open(os.path.join(d, f"rank{rank}.log"), "w")
        procs.append((subprocess.Popen([sys.executable, worker, str(rank), str(port), out], stdout=log, stderr=log), out, log))
    _WORLD2["procs"] = procs


def pytest_sessionfinish(session, exitstatus):
    """Children that are still running (the test that collects them failed early, was interrupted, ...) are ended by
    their exact PIDs and their logs closed."""
    for p, _, log in _WORLD2.pop("procs", []):
        if p.poll() is None:
            p.terminate()
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
                p.wait(timeout=20)
        if not log.closed:
            log.close()


@pytest.fixture(scope="session")
def world2_job():
    def collect():
        import json
        if "procs" not in _WORLD2:
            pytest.skip("no GPU: the world-2 job was not started")
        res = []
        for p, out, log in _WORLD2["procs"]:
            p.wait(timeout=600)
            log.close()
            if os.path.exists(out):
                res.append(json.load(open(out)))
            else:
                res.append({"ok": False, "error": open(log.name).read()[-2000:]})
        return res
    return collect
