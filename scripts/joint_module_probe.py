import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from types import SimpleNamespace
import torch
from p_companion_amd.p_companion import PCompanion
from p_companion_amd.product2vec import FusedAdam
B, P = 4096, 100000
for T in (100, 34800):
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=0.0, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3, NUM_TYPES=T, DEVICE=torch.device("cuda"))
    torch.manual_seed(0)
    table = torch.randn(P, 128)
    model = PCompanion(cfg, table).cuda().train()
    g = torch.Generator().manual_seed(1)
    batch = {"query_idx": torch.randint(0, P, (B,), generator=g).int().cuda(), "query_types": torch.randint(0, T, (B,), generator=g).cuda(),
             "positive_types": torch.randint(0, T, (B, 1), generator=g).cuda(), "negative_types": torch.randint(0, T, (B, 1), generator=g).cuda(),
             "positive_items": torch.randn(B, 128, generator=g).cuda(), "negative_items": torch.randn(B, 128, generator=g).cuda()}
    for name, opt in (("torch.optim.Adam", torch.optim.Adam(model.parameters(), lr=1e-3)), ("FusedAdam", FusedAdam(model, lr=1e-3))):
        def step():
            out = model(batch); loss = model.compute_loss(batch, out); opt.zero_grad(); loss.backward(); opt.step()
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): step()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
        print(f"T={T} module mode + {name}: {ms:.3f} ms/step = {B/ms/1e3:.2f} M triplets/s")
    opt = FusedAdam(model, lr=1e-3)
    def fstep():
        model.train_step(batch); opt.step()
    for _ in range(5): fstep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fstep()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
    print(f"T={T} fused train_step + FusedAdam (eager): {ms:.3f} ms/step = {B/ms/1e3:.2f} M triplets/s")
