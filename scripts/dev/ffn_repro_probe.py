"""Developer probe (GPU): is the FFN forward bitwise reproducible run to run at many rows?  Prints which outputs differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from p_companion_amd import ops

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 117_000
g = torch.Generator().manual_seed(0)
shapes = {"ffn.0.weight": (256, dim), "ffn.0.bias": (256,), "ffn.1.weight": (256,), "ffn.1.bias": (256,), "ffn.3.weight": (256, 256),
          "ffn.3.bias": (256,), "ffn.5.weight": (dim, 256), "ffn.5.bias": (dim,), "attention.in_proj_weight": (3 * dim, dim),
          "attention.in_proj_bias": (3 * dim,), "attention.out_proj.weight": (dim, dim), "attention.out_proj.bias": (dim,)}
params = {k: (torch.randn(*s, generator=g) * 0.05).cuda() for k, s in shapes.items()}
params["ffn.1.weight"].fill_(1.0)
params["ffn.1.running_mean"] = torch.zeros(256, device="cuda")
params["ffn.1.running_var"] = torch.ones(256, device="cuda")
params["ffn.1.num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device="cuda")
table = torch.randn(rows, dim, generator=g).cuda()
segs = [0, 4096, rows - 6 * 4096, rows - 5 * 4096]
ref = None
for it in range(6):
    y, sv = ops.ffn_forward_train(params, table, None, rows, segs, update_running=False)
    torch.cuda.synchronize()
    cur = {"y": y.clone(), "h0": sv["h0"].clone(), "a1": sv["a1"].clone(), "a2": sv["a2"].clone()}
    if ref is None:
        ref = cur
        continue
    for k in cur:
        d = (cur[k] != ref[k])
        n = int(d.sum())
        if n:
            r = torch.nonzero(d.any(1)).reshape(-1)
            print(f"iter {it}: {k} differs in {n} elements, {r.numel()} rows, first rows {r[:8].tolist()} last {r[-3:].tolist()}, tiles {sorted(set((r // 128).tolist()))[:10]}")
print("done", dim, rows)
