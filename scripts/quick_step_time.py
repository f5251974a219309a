"""Developer probe: time pc_p2v_train_step + Adam at the config-2 shape (not the bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from p_companion_amd import ops
from p_companion_amd.data import generate_scaled_bpg, SimilarityIndexLoader

P = int(os.environ.get("P", 100000)); B = int(os.environ.get("B", 4096)); steps = int(os.environ.get("STEPS", 20))
t0 = time.time(); bpg = generate_scaled_bpg(P, 100, 0); print("gen", time.time() - t0, flush=True)
loader = SimilarityIndexLoader(bpg, B, seed=1, drop_last=True)
sizes = [int(np.prod(s)) for s in ops.P2V_SHAPES]; offs = np.concatenate([[0], np.cumsum(sizes)])
flat = (torch.randn(int(offs[-1]), device="cuda") * 0.05); gflat = torch.zeros_like(flat); m = torch.zeros_like(flat); v = torch.zeros_like(flat)
params = {k: flat[offs[i]:offs[i+1]].view(s) for i, (k, s) in enumerate(zip(ops.P2V_KEYS, ops.P2V_SHAPES))}
grads = {k: gflat[offs[i]:offs[i+1]].view(s) for i, (k, s) in enumerate(zip(ops.P2V_KEYS, ops.P2V_SHAPES))}
params["ffn.1.weight"].fill_(1.0)
params["ffn.1.running_mean"] = torch.zeros(256, device="cuda"); params["ffn.1.running_var"] = torch.ones(256, device="cuda")
params["ffn.1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device="cuda")
stepc = torch.zeros(1, dtype=torch.int64, device="cuda"); scal = torch.zeros(2, device="cuda")
table = bpg.cuda()["features"]
it = iter(loader); n = 0; losses = []
for batch in it:
    if n == 5:
        torch.cuda.synchronize(); t0 = time.time()
    out = ops.p2v_train_step(params, grads, table, batch["anchor_idx"], batch["positive_idx"], batch["negative_idx"], batch["neighbor_idx"], 1.0)
    ops.adam_step(flat, gflat, m, v, stepc, scal)
    losses.append(out["loss"])
    n += 1
    if n == 5 + steps: break
torch.cuda.synchronize(); dt = (time.time() - t0) / steps
print("N pad", batch["neighbor_idx"].shape, "ms/step", dt * 1e3, "triplets/s", B / dt, "loss", [round(float(l), 4) for l in losses[::5]])
