"""Developer probe: host time per batch of SimilarityIndexLoader across epoch boundaries (no training step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
bpg = generate_scaled_bpg(100000, 100, seed=0)
ld = SimilarityIndexLoader(bpg, 4096, shuffle=True, seed=1, device="cuda")
for ep in range(3):
    t0 = time.perf_counter()
    it = iter(ld)
    ts = []
    for i, b in enumerate(it):
        t1 = time.perf_counter(); ts.append(t1 - t0); t0 = t1
        int(b["neighbor_compact"]["n_unique"])
        if i % 16 == 0:
            torch.cuda.synchronize()
    print(f"epoch {ep}: batches {len(ts)} first {ts[0]*1e3:.2f} ms second {ts[1]*1e3:.2f} median {sorted(ts)[len(ts)//2]*1e3:.3f} max-after-first {max(ts[1:])*1e3:.2f} last {ts[-1]*1e3:.2f}", flush=True)
