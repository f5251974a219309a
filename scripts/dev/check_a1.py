import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from types import SimpleNamespace
from p_companion_amd import ops
from p_companion_amd.product2vec import Product2Vec
torch.manual_seed(0)
cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0, MARGIN=1.0, DEVICE="cuda")
m = Product2Vec(cfg).cuda().train()
params = m._tensor_dict() if hasattr(m, "_tensor_dict") else dict(m.named_parameters())
for rows in (300, 5000, 70001):
    table = torch.randn(rows, 128, device="cuda")
    y, sv = ops.ffn_forward_train(params, table, None, rows, [0, rows // 3], update_running=False)
    bn = sv["bn"]
    starts = [0, rows // 3, rows]
    ref = torch.empty_like(sv["h0"])
    for s in range(2):
        lo, hi = starts[s], starts[s + 1]
        ref[lo:hi] = torch.tanh(sv["h0"][lo:hi] * bn[2, s] + bn[3, s])
    err = (sv["a1"] - ref).abs()
    print(rows, "a1 max err", float(err.max()), "rows bad", int((err.max(1).values > 1e-5).sum()), "first bad", (err.max(1).values > 1e-5).nonzero()[:8].flatten().tolist())
