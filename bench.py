#!/usr/bin/env python3
"""bench.py -- triplets/sec of P-Companion's two embedding-learning hot paths on MI355X
(BASELINE.json metric: "Product2Vec pretrain + P-Companion joint step").

Default run (no flags), one process per GPU:
  * phase 1 = BASELINE configs[1] -- the headline `value`: 100k products, 100 types, D=128, B=4096 triplets per GPU
    and step, 5 negatives, neighbours padded to the batch max (<=32), synthetic BPG, random-init weights.  One step =
    device batch build (Philox negative sampling + CSR neighbour rows) + gather + 4 FFN/BatchNorm call groups +
    attention + triplet hinge + full backward + Adam, everything inside the timed region.
  * phase 2 = configs[2] -- object `joint` of the same JSON line: the P-Companion joint step (forward, type + item
    hinge, backward, Adam over both type tables and the four Linears), B=4096 per GPU, T = --types (100) and, as
    `joint_num_types_34800`, the reference's own config.py:27 NUM_TYPES.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

ONE JSON line on rank 0.  `roofline` is for the dominant kernel family of the headline phase (gemm_nt_kernel):
algorithmic FLOPs and bytes / HIP-event time on the launch stream; BOTH fractions are printed (matrix cores: fp32-grade
products as six v_mfma_f32_32x32x16_bf16 over a three-way bf16 split, i.e. dense bf16 peak / 6; HBM: 8 TB/s) and
`bound` names the roof that is nearer (the larger lower bound on the launch time).  `cpu_baseline` = the oracle port
timed on this box's host cores (compute-only over >= 20 steps, and end to end with the host loader).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, Peak FP32 (matrix)
BF16_MFMA_PEAK_TFLOPS = 2500.0    # same guide, dense bf16 (no sparsity)
BF16_PRODUCTS = 6                 # common.h split3: x = p0 + p1 + p2 (bf16 each), x*y ~ the six products of weight <= 2
NT_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / BF16_PRODUCTS     # 416.7 fp32-equivalent TFLOP/s
TN_PROFILE_STEPS = 4       # untimed steps behind the region whose weight-gradient (TN) launches carry HIP-event brackets
PROFILE_EVERY = 20         # timed steps between two steps whose gemm_nt launches carry HIP-event brackets (each bracketed step pays ~60 us of event packets)
HBM_PEAK_GBS = 8000.0


def flops_per_triplet(n, dim=128):
    """SURVEY.md section 8(d), every padded slot its own row: at dim = 128 fwd 328,192*N + 1,900,544 and fwd+bwd =
    3*fwd - 65,536*(N+7) (no input gradient of Linear0)."""
    ffn_row = 2 * (dim * 256 + 256 * 256 + 256 * dim)
    fwd = ffn_row * (n + 7) + 4 * dim * dim * n + 4 * dim * dim + 4 * dim * n
    return 3 * fwd - 2 * dim * 256 * (n + 7)


def nt_algorithmic_bytes(b, nbc, dim=128):
    """Compulsory HBM bytes of the six persistent gemm_nt_kernel products of one step:
    rows R = [anchor B | neighbour rows nbc | positive B | negatives 5B]; per row (d = dim * 4 bytes, 1 KB = a 256-wide
    hidden row) Linear0 d + 1 KB (gathered x -> H0), Linear3 1+1+1 (H0 -> A2, and A1 = tanh(BN(H0)) saved for dW3),
    Linear5 1 KB + d (A2 -> Y), dZ2 d+1+1 (dY, A2 -> dZ2), dZ1 1+1+1 (dZ2, H0 -> dZ1); per neighbour row dKeys 2d + d
    ([dK|dV] -> dY).  Weights are L2-resident.  (Round 2 had a seventh launch, the K|V projection of the neighbour rows:
    absorbed into the single-query attention in round 3.)"""
    r = 7 * b + nbc
    d = dim * 4
    return r * ((d + 1024) + 3072 + (1024 + d) + (d + 2048) + 3072) + nbc * 3 * d


def executed_flops_per_step(b, nbc, dim=128):
    """FLOPs of the GEMMs one step actually runs: FFN forward + backward without dX of Linear0 over R = 7b + nbc rows
    (Linear0 / dW0: 2*dim*256 each, Linear3 / dZ1 / dW3: 2*256*256 each, Linear5 / dZ2 / dW5: 2*256*dim each), dKeys over the
    nbc neighbour rows (2 * 2dim * dim) and, per sample, the q / out projections with their backward (6 * 2 dim^2) plus
    the per-head products of the absorbed K and V projections (6 * 2 dim^2 / ... one dim x dim product each)."""
    ffn = 2 * (2 * dim * 256) + 3 * (2 * 256 * 256) + 3 * (2 * 256 * dim)
    return float(ffn) * (7 * b + nbc) + 4.0 * dim * dim * nbc + 24.0 * dim * dim * b


def bytes_per_triplet(n, dim=128):
    return 4 * dim * (n + 7) + 4 * (n + 7)


def committed_pmc(kind, match):
    """HBM bytes per launch of the kernels whose name `match` accepts, from the newest committed rocprofv3 --pmc passes of
    the SAME workload (profiles/<round tag>[_<kind>]_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE in separate passes,
    KiB -> bytes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  kind: "" = configs[1] (100 k products),
    "joint" / "jointd" / "joint34800" / "joint34800d" = the joint step at T = 100 / 100 with DROPOUT = 0.1 / 34800 / 34800 with DROPOUT = 0.1,
    "p2vd" = configs[1] with DROPOUT = 0.1, "cfg3" = 10 M products through the sharded lookup, "big" =
    configs[4] (100 M x 256, Zipf).  None if no profile of that kind is committed (PMC counters cannot be collected from
    inside the bench process) -- a leg never borrows another workload's traffic."""
    import re
    pat = re.compile(r"^(r\d+[a-z0-9]*)_(?:(joint34800d|joint34800|jointd|joint|p2vd|big|cfg3)_)?pmc_traffic\.json$")
    files = []
    for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        m = pat.match(f)
        if m and (m.group(2) or "") == kind:
            files.append(f)
    if not files:
        return None
    with open(os.path.join(ROOT, "profiles", files[-1])) as f:
        d = json.load(f)
    sel = [v for k, v in d.items() if match(k)]
    n = sum(v["launches"] for v in sel)
    if not n:
        return None
    return {"hbm_bytes_per_launch": round(sum((v["fetch_bytes_corrected"] + v["write_bytes"]) * v["launches"] for v in sel) / n),
            "source": os.path.join("profiles", files[-1])}


def two_roof(flops, nbytes, seconds, peak_tflops):
    """Both roofline fractions of one launch: time the matrix cores / HBM would need at their peaks over the measured
    time; the binding roof is the larger lower bound."""
    t_mfma = flops / (peak_tflops * 1e12)
    t_hbm = nbytes / (HBM_PEAK_GBS * 1e9)
    fm, fh = (t_mfma / seconds, t_hbm / seconds) if seconds > 0 else (0.0, 0.0)
    return ("mfma" if t_mfma >= t_hbm else "hbm"), fm, fh


# ----------------------------------------------------------------------------------------------- CPU legs
def p2v_cpu_baseline(bpg, batch, min_steps=20, max_seconds=100.0, dropout=0.0):
    """The oracle (CPU restatement pinned to the reference's golden vectors) timed on this box's host cores: same
    workload shape, fwd+bwd+Adam.  compute-only: one pre-gathered dense batch, >= 20 steps; end to end: the host
    loader in parity mode (exact CPython negative sampler + CSR neighbour rows) + the dense gather of
    data_loader.py:45-88,171-206 + the step, per batch."""
    from oracle import p2v_oracle
    from p_companion_amd.data import SimilarityIndexLoader
    st = p2v_oracle.init_state(0)
    feats = torch.from_numpy(bpg.features)
    nbc = batch["neighbor_compact"]
    nb_dense = nbc["nb_rows"].cpu().numpy()[nbc["slot_row"].cpu().numpy()]      # the reference's padded [B,N] layout
    dense = p2v_oracle.gather_batch(feats, batch["anchor_idx"].cpu().numpy(), batch["positive_idx"].cpu().numpy(),
                                    batch["negative_idx"].cpu().numpy(), nb_dense)
    mom = p2v_oracle.new_moments(st)
    b = dense["anchor"].shape[0]
    # DROPOUT > 0 (nn.MultiheadAttention(dropout=p), product2vec.py:23-28): the probabilities' mask is an explicit input of the
    # oracle; a mask of the leg's keep probability and scale, drawn once (the work does not depend on which elements it drops)
    amask = None
    if dropout > 0.0:
        g = torch.Generator().manual_seed(0)
        amask = torch.bernoulli(torch.full((b, 4, dense["anchor_neighbors"].shape[1]), 1.0 - dropout), generator=g) / (1.0 - dropout)
    p2v_oracle.train_step(st, dense, 1.0, mom, 1, attn_mask=amask)            # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < min_steps:
        p2v_oracle.train_step(st, dense, 1.0, mom, n + 2, attn_mask=amask)
        n += 1
        if time.perf_counter() - t0 > max_seconds:
            break
    el = time.perf_counter() - t0
    if dropout > 0.0:                                        # (the secondary legs report the compute-only figure)
        return {"value": b * n / el, "unit": "triplets/s", "cores": torch.get_num_threads(), "kind": "port", "os_cpu_count": os.cpu_count(),
                "sample": f"{n} steps of the same workload (B={b}, N={dense['anchor_neighbors'].shape[1]}, attention dropout p={dropout} as an "
                          f"explicit mask, fwd+bwd+Adam, pre-gathered batch) on {os.cpu_count()} host cpus, torch {torch.get_num_threads()} threads"}
    # end to end: a few batches through the host loader
    ld = SimilarityIndexLoader(bpg, b, shuffle=True, sampler="cpython", seed=0, drop_last=True, device="cpu",
                               compact=False, prefetch=False, unique=False)
    it = iter(ld)
    t1 = time.perf_counter()
    m = 0
    while m < 3:
        hb = next(it)
        d2 = p2v_oracle.gather_batch(feats, hb["anchor_idx"].numpy(), hb["positive_idx"].numpy(),
                                     hb["negative_idx"].numpy(), hb["neighbor_idx"].numpy())
        p2v_oracle.train_step(st, d2, 1.0, mom, n + m + 2)
        m += 1
    el2 = time.perf_counter() - t1
    return {"value": b * n / el, "unit": "triplets/s", "cores": torch.get_num_threads(), "kind": "port",
            "os_cpu_count": os.cpu_count(),
            "sample": f"{n} steps of the same workload (B={b}, N={dense['anchor_neighbors'].shape[1]}, fwd+bwd+Adam, "
                      f"pre-gathered batch) on {os.cpu_count()} host cpus, torch {torch.get_num_threads()} threads",
            "end_to_end": {"value": b * m / el2, "unit": "triplets/s",
                           "sample": f"{m} batches through the host loader (exact CPython negative sampler, CSR neighbour "
                                     f"rows, dense gather) + the step"}}


def joint_cpu_baseline(model, batch, cfg, min_steps=20, max_seconds=40.0):
    """oracle.joint_oracle.train_step on the same batch and the same initial state, on the host cores.  DROPOUT > 0: the
    hidden-layer mask is an explicit input of the oracle (nn.Dropout's stream cannot be reproduced); the timed steps use a mask of
    the leg's keep probability and scale drawn once with torch.bernoulli -- what the oracle does with it does not depend on which
    elements it drops."""
    from oracle import joint_oracle
    st = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    hb = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}
    mom = joint_oracle.new_moments(st)
    p = float(getattr(cfg, "DROPOUT", 0.0))
    mask = None
    if p > 0.0:
        g = torch.Generator().manual_seed(0)
        mask = torch.bernoulli(torch.full((hb["query_idx"].numel(), 32), 1.0 - p), generator=g) / (1.0 - p)
    joint_oracle.train_step(st, hb, mom, 1, cfg.MARGIN, cfg.ALPHA, cfg.NUM_COMP_TYPES, hidden_mask=mask)
    t0 = time.perf_counter()
    n = 0
    while n < min_steps:
        joint_oracle.train_step(st, hb, mom, n + 2, cfg.MARGIN, cfg.ALPHA, cfg.NUM_COMP_TYPES, hidden_mask=mask)
        n += 1
        if time.perf_counter() - t0 > max_seconds:
            break
    el = time.perf_counter() - t0
    b = hb["query_idx"].numel()
    return {"value": b * n / el, "unit": "triplets/s", "cores": torch.get_num_threads(), "kind": "port",
            "os_cpu_count": os.cpu_count(),
            "sample": f"{n} steps of the same batch (B={b}, T={st['query_type_embeddings.weight'].shape[0]}, "
                      + (f"hidden-layer dropout p={p} as an explicit mask, " if p > 0.0 else "") + "forward + both "
                      f"hinges + autograd backward + dense Adam) on {os.cpu_count()} host cpus, torch "
                      f"{torch.get_num_threads()} threads"}


# ----------------------------------------------------------------------------------------------- joint phase
def joint_algorithmic_bytes(b, t, k=3):
    """per triplet 512 (query row) + 1024 (pos/neg item rows) + 256 (query-type row) + K*256 (comp-type
    rows) + ~40 B of indices; per step T*L*4 to stream E_c for the similarities and 28 B per parameter of dense Adam
    over 2*T*64 + 29,024 parameters."""
    return b * (512 + 1024 + 256 + k * 256 + 40) + t * 64 * 4 + 28 * (2 * t * 64 + 29024)


# The joint legs choose their own number of timed steps (the contract's K is the Product2Vec headline's): a 0.05 ms step timed over
# 100 steps is a 5 ms region -- the chip is still coming up from idle (0.0476 ms over 100 steps, 0.0459 over 1000; T = 34800: 0.127
# over 25, 0.1215 over 250: scripts/dev/first_steps_probe.py) -- so these legs time at least 46 / 30 ms.
JOINT_MIN_STEPS = 1000
JOINT_REF_MIN_STEPS = 250
BENCH_T0 = time.monotonic()  # process start (module import): the legs' total budget counts from here
LOADER_OPTS = {}             # developer A/B (PC_BENCH_SET_OPTIONS "rows=0"): attributes set on the Product2Vec leg's loader
EXCHANGE = {"ex": None}      # the replicas' gradient exchange (distributed.make_exchange), set once in main() when N > 1


def run_joint(args, rank, world, dev, types, steps, warmup, want_cpu, dropout=0.0, exchange=None):
    """BASELINE configs[2]: 100k products, T types, B pairs per GPU: PCompanion forward + both hinge losses + backward
    + Adam (pc_joint_train_step + pc_adam_step), loader batch construction included."""
    from types import SimpleNamespace
    from p_companion_amd import distributed as pdist
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader, generate_scaled_bpg
    from p_companion_amd.p_companion import GraphedJointStep, PCompanion
    from p_companion_amd.product2vec import FusedAdam
    exchange = EXCHANGE["ex"] if exchange is None else exchange
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=float(dropout), MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3,
                          NUM_TYPES=types, DEVICE=dev)
    # configs[2] is quoted on the 100 k catalogue; the complementary-pair dataset is built from host arrays, so a catalogue that
    # only exists in HBM (--products > 2 M: configs[3]/[4] of the Product2Vec phase) is not what this phase trains over
    jproducts = args.products if args.products <= 2_000_000 else 100_000
    bpg = run_joint.bpg if getattr(run_joint, "bpg", None) is not None else generate_scaled_bpg(jproducts, min(args.types, 100), seed=0)
    run_joint.bpg = bpg
    torch.manual_seed(0)
    model = PCompanion(cfg, bpg.cuda(dev)["features"]).to(dev).train()       # frozen table: synthetic stand-in for the P2V export
    opt = FusedAdam(model, lr=1e-3)
    flat, gflat = model.flatten_parameters()
    # one process: the fixed-shape step is captured once as a HIP graph and replayed, the loader builds each batch
    # straight into the graph's input buffers.  N > 1 keeps eager launches (the gradient all-reduce sits between).
    graphed = None
    multi = pdist.collectives_on(world)                  # (world > 1, or the one-rank RCCL rehearsal PC_DIST_FORCE=1)
    if not multi and not args.no_graph:
        graphed = GraphedJointStep(model, opt, args.batch, mode="auto" if args.joint_launch == "epoch" else args.joint_launch)
    elif not args.no_graph:
        # one process per GPU: the fused step writes gradients only, the flat gradient buffer (29 k weights + both type tables)
        # is averaged over the replicas, then the Adam launch.  --exchange native (default): the library's exchange slot --
        # ncclAllReduce(ncclAvg) on its own RCCL communicator, issued from the step's own call (pc_joint_train_epoch_dp: the
        # replica's whole epoch as one foreign call); hook: a Python grad_hook per step (T > 512: the two [T,64] table gradients
        # as row lists with one host-visible read per step, the 29 k dense weights as one all-reduce)
        try:
            if exchange is not None:
                # (T > 512: the optimizer sharded over the replicas -- reduce-scatter, Adam on 1/world of the flat buffers,
                # all-gather -- ABI 8; GraphedJointStep pads and registers the flat buffers)
                graphed = GraphedJointStep(model, opt, args.batch, mode="direct", exchange=exchange)
                flat, gflat = model.flatten_parameters()
            else:
                graphed = GraphedJointStep(model, opt, args.batch, mode="direct", grad_hook=lambda g: pdist.all_reduce_mean_(g, world))
                graphed.grad_hook = pdist.joint_grad_hook(model, graphed, world)
        except ValueError:
            graphed = None
    # 'epoch': train.py:36-57's loop over the epoch's batches as one foreign call (pc_joint_train_epoch / _dp) -- the same steps,
    # enqueued from C back to back; falls back to one call per step where the fused step does not serve the configuration
    by_epoch = (graphed is not None and args.joint_launch == "epoch" and graphed.mode == "direct"
                and (not multi or graphed.exchange is not None))
    # direct mode: the loader hands its batches over unbuilt and the step's first kernel builds them (same values)
    loader = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), args.batch, shuffle=True, seed=rank,
                                      device=dev, out=graphed.static if graphed else None,
                                      deferred=graphed is not None and graphed.mode == "direct")

    def batches():
        while True:
            for b in loader:
                if b["query_idx"].numel() == args.batch:
                    yield b

    it = batches()

    def step(b):
        if graphed is not None:
            return graphed(b)[0]
        losses, _ = model.train_step(b)
        if exchange is not None:
            exchange.register(gflat)
            opt.step(exchange=exchange)
        else:
            pdist.all_reduce_mean_(gflat, world)
            opt.step()
        return losses

    last = None
    for _ in range(max(warmup, 4)):                   # (the graph is captured after GraphedJointStep's eager warm-up steps)
        last = next(it)
        step(last)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    if multi:
        torch.cuda.synchronize()          # (the device drained first: torch's communicator never runs beside the library's)
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    trace = os.environ.get("PC_BENCH_TRACE")
    done = 0
    while by_epoch and done < steps:
        ep = graphed.run_epoch(loader, drop_last=True, max_steps=steps - done)
        done += int(ep.shape[0])
        losses = ep[-1]
    for i in range(0 if not by_epoch else steps, steps):
        last = next(it)
        losses = step(last)
        if trace and i % 5 == 0:                     # debugging aid: localise a device fault (synchronises: not for timing)
            torch.cuda.synchronize()
            print(f"[trace] joint T={types} step {i} loader step {loader.step} ok", file=sys.stderr, flush=True)
    ev[1].record()
    host_ms = 1e3 * (time.perf_counter() - t0) / steps       # time the host needed to ENQUEUE a step (>= device time: host-bound)
    torch.cuda.synchronize()
    if multi:
        torch.distributed.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if multi:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    el = float(t)
    if rank != 0:
        return None
    value = world * args.batch * steps / el
    dev_ms = ev[0].elapsed_time(ev[1]) / steps               # HIP events on the launch stream, around the timed steps
    alg = joint_algorithmic_bytes(args.batch, types)
    # (the committed counter passes: DROPOUT = 0 runs of T = 100 and T = 34800, and T = 34800 with DROPOUT = 0.1 -- the reference as
    # shipped; other legs carry no traffic figure)
    pmc_kind = ("joint" if types == 100 else f"joint{types}") if dropout == 0.0 and types in (100, 34800) else \
               ("joint34800d" if types == 34800 else "jointd" if types == 100 else None) if abs(dropout - 0.1) < 1e-9 else None
    pmc = committed_pmc(pmc_kind, lambda k: k == "_step_total") if pmc_kind else None
    achieved = alg / (dev_ms * 1e-3) / 1e9
    out = {"metric": "triplets/sec (P-Companion joint step: fwd + type/item hinge + bwd + Adam)", "value": round(value, 1),
           "unit": "triplets/s", "steps": steps, "ms_per_step": round(1e3 * el / steps, 4),
           "host_enqueue_ms_per_step": round(host_ms, 4),
           # (the wall clock of a 45 ms region is at the mercy of one host hiccup; the HIP events around the same steps are not)
           "wall_over_device": round(1e3 * el / steps / dev_ms, 3) if dev_ms > 0 else None,
           "config": {"workload": f"P-Companion joint step, {jproducts} products, NUM_TYPES={types}, dim=128, "
                                  f"batch={args.batch}/GPU, K=3 (loader batch construction included)",
                      "global_batch": world * args.batch, "parallelism": f"dp{world}", "final_loss": round(float(losses[0]), 5),
                      "dropout": float(dropout),
                      "topk": ("exact: index-identical to torch.topk of the full [B,T] similarity product" if types <= 512 else
                               "per-sample selection over a re-associated fp32 product (sub-chunk maxima, then an exact top-K over each row's K "
                               "best sub-chunks): identical to torch.topk of the oracle's product except at fp32-rounding TIES -- the full-size "
                               "test allows <= 2 of 4096 rows to differ there (tests/test_gpu_fullsize.py), the one integer output of the "
                               "path that is not asserted bit-exact at this size"),
                      # N > 1: what one replica hands to the all-reduce per step -- the whole flat gradient buffer behind the slot
                      # (T = 34800: both dense [T,64] tables, 17.8 MB: about what the touched rows would cost as constant-shape
                      # padded lists at B = 4096); the Python hooks send the touched rows (sizes read back)
                      "exchange_bytes_per_step": (int(gflat.numel()) * 4 if multi and graphed is not None and graphed.exchange is not None else None),
                      "exchange": (getattr(exchange, "kind", None) if multi and graphed is not None and graphed.exchange is not None else
                                   "python grad_hook per step (torch.distributed)" if multi else None),
                      "optimizer": (("sharded over the replicas: reduce-scatter of the flat gradient, Adam on this rank's 1/%d of the flat "
                                     "buffers, all-gather of the parameters (pc_exchange_adam_plan)" % world)
                                    if multi and graphed is not None and getattr(graphed, "shard_optimizer", False) else
                                    "all-reduce of the flat gradient, the whole Adam on every rank" if multi else "one process"),
                      "launch": (("pc_joint_train_epoch_dp: the replica's epoch (fused step without Adam, exchange slot, Adam) enqueued by one foreign call"
                                  if multi else "pc_joint_train_epoch: the epoch's steps enqueued by one foreign call") if by_epoch else
                                 {"direct": "fused step, arguments resolved once (one foreign call per step)" +
                                            ("; gradients only, then all-reduce of the flat gradient buffer and the Adam launch" if multi else ""),
                                  "graph": "hipGraph replay"}[graphed.mode] if graphed is not None else "eager module calls"),
                      "kernels_per_step": ("the launch-per-op sequence (pc_joint_train_step + pc_adam_step, ~25 launches)" if graphed is None or graphed.mode == "graph" else
                                           "2 (tile kernel: batch construction, forward, losses, backward and the tile's gradient slab; finish: slab "
                                            "sums + Adam)" if types <= 128 else
                                            "4 (batch builder, tile kernel, gradient products, finish + Adam)" if types <= 512 else
                                            ("7 (hidden rows of the B samples + G = E_c dec_w, sub-chunk maxima of the similarity rows with the table-gradient "
                                             "clear riding, exact top-K over each row's K best sub-chunks, tile kernel incl. batch construction and the "
                                             "weight-gradient slabs, counting sort of the table gradients' source rows by destination, per-destination sums "
                                             "in ascending source order, finish + Adam)" if dropout > 0 else
                                             "7 (hidden rows of the distinct query types + G = E_c dec_w, sub-chunk maxima of the similarity rows with the "
                                             "table-gradient clear riding, exact top-K over each row's K best sub-chunks, tile kernel incl. batch construction "
                                             "and the weight-gradient slabs, counting sort of the table gradients' source rows by destination, per-destination "
                                             "sums in ascending source order, finish + Adam -- whose first workgroup forms the NEXT step's distinct-type list; "
                                             "an epoch's first step has an eighth launch for its own list)"))},
           "roofline": {"bound": "hbm", "kernel": "the whole step (batch builder + the fused step's kernels: a dependent chain)",
                        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5),
                        "algorithmic_bytes_per_step": alg, "device_ms_per_step": round(dev_ms, 4),
                        "traffic": pmc["hbm_bytes_per_launch"] if pmc else None,
                        "traffic_source": pmc["source"] if pmc else None,
                        "note": "2.6 KB gathered per triplet + T*256 B of E_c + 28 B/parameter of dense Adam per step "
                                "(SURVEY 8d); launch/latency-bound, not bandwidth-bound"},
           # (T = 34800: ~1 s per oracle step -- dense [B,T] similarities and dense Adam over both tables -- bounded to ~20 s)
           "cpu_baseline": (joint_cpu_baseline(model, last, cfg, min_steps=20 if types <= 1000 else 8,
                                               max_seconds=40.0 if types <= 1000 else 20.0) if want_cpu else None)}
    return out


# ----------------------------------------------------------------------------------------------- P2V phase
def run_p2v(args, rank, world, dev, products, steps, warmup, want_cpu, profile_kernels=True, sustained=False, dim=None,
            negatives=None, table_mode=None, dropout=0.0, pmc_kind="", exchange=None, hot_rows=0):
    """One Product2Vec leg.  dim / negatives / table_mode default to the command line's; dropout = config.py:12's DROPOUT of
    the attention probabilities (0.0: the parity setting, every golden test; 0.1: the reference as shipped)."""
    from types import SimpleNamespace
    from p_companion_amd import distributed as pdist
    from p_companion_amd import ops
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec

    from p_companion_amd.data import generate_device_bpg
    multi = pdist.collectives_on(world)                  # (world > 1, or the one-rank RCCL rehearsal PC_DIST_FORCE=1)
    exchange = EXCHANGE["ex"] if exchange is None else exchange
    dim = args.dim if dim is None else dim
    negatives = args.negatives if negatives is None else negatives
    table_mode = args.table if table_mode is None else table_mode
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=dim, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=float(dropout),
                          MARGIN=1.0, BATCH_SIZE=args.batch, LEARNING_RATE=1e-3, DEVICE=dev)
    # the catalogue: host restatement (numpy; what the CPU baseline also reads) up to 2 M products, otherwise generated IN
    # HBM by csrc/generator.hip (configs[3]/[4]: 10 M / 100 M products cannot exist as host arrays); a sharded table is
    # then generated shard by shard (rows rank::world), never as a whole
    on_device = args.generator == "device" or (args.generator == "auto" and products > 2_000_000)
    t_gen = time.perf_counter()
    if on_device:
        shard = table_mode == "sharded"
        bpg = generate_device_bpg(products, args.types, seed=0, dim=dim, device=dev, rank=rank if shard else 0,
                                  world=world if shard else 1, with_complementary=False)
        torch.cuda.synchronize()
    else:
        bpg = generate_scaled_bpg(products, args.types, seed=0, dim=dim)
    t_gen = time.perf_counter() - t_gen
    if products == args.products and not on_device and dim == 128:
        run_joint.bpg = bpg                                   # the joint phase trains over the same catalogue
    torch.manual_seed(0)
    model = Product2Vec(cfg).to(dev)
    model.train()
    opt = FusedAdam(model, lr=cfg.LEARNING_RATE)
    flat, gflat = model.flatten_parameters()
    if exchange is not None:
        exchange.register(gflat)
    table = bpg.cuda(dev)["features"]
    sharded = None
    if table_mode == "sharded":
        # row r on rank r % world; the loader runs the per-batch exchange on its side stream, one batch ahead
        local = table if on_device else pdist.ShardedFeatureTable.shard(table, rank, world)
        # (the lookup rounds on the communicator that carries the gradient exchange: one cross-rank launch order per step)
        # hot_rows: the replicated hot set of the Zipf head (configs[4]): ids below it are served from a local replica
        sharded = pdist.ShardedFeatureTable(local, bpg.num_products, rank, world, exchange=exchange,
                                            hot_rows=hot_rows if negatives == "zipf" else 0)
    loader = SimilarityIndexLoader(bpg, args.batch, shuffle=True, sampler="philox", seed=1 + rank, drop_last=True,
                                   device=dev, sharded=sharded, negatives=negatives, reuse_buffers=True)
    for k_, v_ in LOADER_OPTS.items():
        setattr(loader, k_, v_)

    def batches():
        while True:
            for b in loader:
                yield b

    it = batches()
    prof = ops.KernelProfile(capacity=32 * max(steps, 1))
    if not args.profile_all:
        prof.set_kinds(["gemm_nt_kernel"])
    n_sum = real_sum = slot_sum = 0

    def step(b, profile=None):
        tab = b.get("table", table)                       # sharded: the rows this batch's exchange delivered
        sync = ((exchange.all_reduce_sum_f64_ if exchange is not None else (lambda t: torch.distributed.all_reduce(t)))
                if (args.sync_bn and multi) else None)
        if not multi:
            # one process: optimizer.step() rides in the step's last gradient launch (pc_p2v_train_step_unique_adam)
            return model.train_step_indexed(tab, b, profile=profile, optimizer=opt)
        loss = model.train_step_indexed(tab, b, profile=profile, sync_reduce=sync)
        if exchange is not None:
            opt.step(exchange=exchange)                   # pc_exchange_adam: the replicas' mean gradient + Adam, one foreign call
        else:
            pdist.all_reduce_mean_(gflat, world)
            opt.step()
        return loss

    last = None
    for _ in range(warmup):
        last = next(it)
        step(last)
    # HIP-event brackets around the dominant kernel family cost the stream ~5 us per bracket side (a marker packet each):
    # every PROFILE_EVERY-th step of the timed region carries them, the others run as a caller's loop does
    profiled_steps = 0
    # (every loader call of the region is inside it: `steps` next() calls and `steps` train steps between the two clocks, as in a
    # caller's loop -- round 4 took the first batch before the clock started)
    if multi:
        torch.cuda.synchronize()          # (the device drained first: torch's communicator never runs beside the library's)
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        last = next(it)
        n_sum += last["n_pad"]
        nbc_ = last["neighbor_compact"]
        real_sum += int(nbc_["n_unique"]) if "weight" in nbc_ else nbc_["nb_rows"].numel() - 1      # rows carried
        slot_sum += nbc_.get("n_real", nbc_["nb_rows"].numel() - 1)                                   # real slots
        # (not the region's first step: its launches follow the opening synchronize with an empty queue behind them)
        bracket = profile_kernels and (args.profile_all or i % PROFILE_EVERY == min(PROFILE_EVERY // 2, steps - 1))
        profiled_steps += 1 if bracket else 0
        loss = step(last, profile=prof if bracket else None)
    host_ms = 1e3 * (time.perf_counter() - t0) / max(steps, 1)      # the host's time to enqueue a step (incl. its waits)
    torch.cuda.synchronize()
    if multi:
        torch.distributed.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if multi:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    el = float(t)
    hot_served = None
    if sharded is not None:
        sharded.raise_if_overflowed()
        if sharded.hot_rows:
            hot_served = sharded.hot_rows_served()          # entries served from the replica since construction (warm-up included)
    # the second-largest kernel family (the weight-gradient products: gemm_tn8_kernel x3 + gemm_tn_group_kernel): HIP-event
    # brackets around ITS launches over a few extra steps OUTSIDE the timed region (every bracket is two event packets on the
    # stream; the timed region carries the dominant family's only).  All ranks step together (the step holds collectives).
    tn_prof = None
    if profile_kernels and not args.profile_all:
        tn_prof = ops.KernelProfile(capacity=64)
        tn_prof.set_kinds(["gemm_tn_kernel"])
        for _ in range(TN_PROFILE_STEPS):
            step(next(it), profile=tn_prof)
        torch.cuda.synchronize()
    if rank != 0:
        return None

    n_avg = n_sum / max(steps, 1)
    rows_avg = real_sum / max(steps, 1)
    slots_avg = slot_sum / max(steps, 1)
    value = world * args.batch * steps / el
    res = {"value": value, "ms_per_step": 1e3 * el / steps, "n_avg": n_avg, "final_loss": float(loss),
           "host_enqueue_ms_per_step": round(host_ms, 4), "profiled_steps": profiled_steps,
           "distinct_neighbour_rows": rows_avg, "real_neighbour_slots": slots_avg,
           "sharded_lookup": ({"capacity_rows_per_peer": sharded.capacity, "bytes_per_peer_and_step": sharded.bytes_per_peer,
                               "host_syncs_per_step": 0, "hot_rows": sharded.hot_rows,
                               # (the loader builds a few batches ahead of the step: lookups counted = batches BUILT since construction)
                               "hot_rows_served": hot_served,
                               "hot_rows_served_per_batch": (round(hot_served / max(loader.step, 1), 1) if hot_served is not None else None),
                               "ids_per_batch": 2 * args.batch + 5 * args.batch + int(round(rows_avg)) + 1}
                              if sharded is not None else None),
           "rows_saved_by_duplicate_neighbours": round(1.0 - (7 * args.batch + rows_avg + 1) / (7 * args.batch + slots_avg + 1), 4)}
    if profile_kernels:
        nt = prof.summary("gemm_nt_kernel")
        tn = (tn_prof if tn_prof is not None else prof).summary("gemm_tn_kernel")
        tn_steps = TN_PROFILE_STEPS if tn_prof is not None else profiled_steps
        tn_traffic = committed_pmc(pmc_kind, lambda k: "gemm_tn" in k)
        tn_red = committed_pmc(pmc_kind, lambda k: "tn_reduce" in k)
        sm = prof.summary("gemm_nt_small_kernel")
        traffic = committed_pmc(pmc_kind, lambda k: "gemm_nt_kernel<" in k and "<1, 2," not in k)
        launches = max(nt["launches"], 1)
        sec = nt["total_ms"] * 1e-3 / launches
        fl = nt["total_flops"] / launches
        # bytes of the family per STEP over its launches per step as counted (six products; the two backward ones may run as two
        # launches each -- the positives' / negatives' rows early on a side queue: eight launches of the same bytes)
        alg = nt_algorithmic_bytes(args.batch, rows_avg + 1, dim) / (launches / max(profiled_steps, 1))
        bound, fm, fh = two_roof(fl, alg, sec, NT_PEAK_TFLOPS)
        tfl = fl / sec / 1e12 if sec > 0 else 0.0
        gbs = alg / sec / 1e9 if sec > 0 else 0.0
        exe = executed_flops_per_step(args.batch, rows_avg + 1, dim)
        # the contraction, not the gather, binds the Product2Vec step (1.7 kFLOP per gathered byte) -- `frac`
        # is the matrix-core fraction of the dominant kernel family.  Beside it: the same launches against HBM with the DESIGN's own
        # per-launch activation traffic (`frac_hbm_design`: every layer's [R,256] in and out of HBM -- which roof is nearer by that
        # count is `nearer_roof_by_design_bytes`), and with section 8(d)'s bytes -- the gathered rows, 512 (N + 7) B per triplet, the
        # only bytes the algorithm has to move -- spread over the family's launches (`frac_hbm_8d`)
        alg8d = bytes_per_triplet(n_avg, dim) * args.batch / (launches / max(profiled_steps, 1))
        step_pmc = committed_pmc(pmc_kind, lambda k: k == "_step_total")
        res["roofline"] = {
            # 1.7 kFLOP per gathered byte -- the contraction, not the gather, binds this step, so `frac` is the matrix-core
            # fraction; which roof is nearer by the DESIGN's own per-launch activation bytes is `nearer_roof_by_design_bytes`
            "bound": "mfma", "bound_basis": "SURVEY 8(d) algorithmic bytes and flops", "kernel": "gemm_nt_kernel",
            "achieved": round(tfl, 2), "peak": round(NT_PEAK_TFLOPS, 1), "unit": "TFLOP/s", "frac": round(fm, 4),
            "frac_mfma": round(fm, 4), "achieved_tflops": round(tfl, 2), "peak_tflops": round(NT_PEAK_TFLOPS, 1),
            "frac_of_fp32_mfma_peak": round(tfl / FP32_MFMA_PEAK_TFLOPS, 4),
            "frac_hbm_design": round(fh, 4), "achieved_gbs_design": round(gbs, 1), "nearer_roof_by_design_bytes": bound,
            "frac_hbm_8d": round(alg8d / sec / 1e9 / HBM_PEAK_GBS, 5) if sec > 0 else 0.0, "algorithmic_bytes_per_launch_8d": round(alg8d),
            "bound_note": "intensity %.1f FLOP/B by the design's activation bytes vs the ridge %.1f FLOP/B of (2500 dense bf16 TFLOP/s / "
                          "6 products) over 8 TB/s; by section 8(d)'s gathered bytes the intensity is %.0f FLOP/B"
                          % (fl / alg, NT_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS, fl / alg8d),
            "peak_note": "fp32 in / fp32 accumulate / fp32-grade result on the bf16 matrix cores: 6 bf16 MFMA products per fp32 "
                         "product, so the matrix peak is 2500 / 6 fp32-equivalent TFLOP/s; `achieved_tflops` counts each fp32 product once",
            "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
            "traffic_source": traffic["source"] if traffic else None,
            "algorithmic_bytes_per_launch": round(alg), "flops_per_launch": fl,
            "launches": nt["launches"], "launches_per_step": round(launches / max(profiled_steps, 1), 2), "avg_launch_us": round(1e6 * sec, 2),
            "family_us_per_step": round(1e3 * nt["total_ms"] / max(profiled_steps, 1), 1),
            "bracketed_steps": profiled_steps,              # (every PROFILE_EVERY-th timed step carries the HIP-event brackets)
            "share_of_step": round(nt["total_ms"] / max(profiled_steps, 1) / (el * 1e3 / steps), 3),
            # the weight-gradient family (dW0, dW3, dW5 + the attention block's grouped dW): fp32-grade six-product bf16 form like
            # the NT family, so the same matrix peak; each launch writes per-workgroup partial-sum slabs that
            # tn_reduce_group_kernel folds (slab bytes: the committed counter passes of the same command)
            "gemm_tn_kernel": ({"achieved_tflops": round(tn["total_flops"] / max(tn["total_ms"], 1e-9) / 1e9, 2),
                                "peak_tflops": round(NT_PEAK_TFLOPS, 1),
                                "frac": round(tn["total_flops"] / max(tn["total_ms"], 1e-9) / 1e9 / NT_PEAK_TFLOPS, 4),
                                "flops_per_launch": round(tn["total_flops"] / max(tn["launches"], 1)),
                                "avg_launch_us": round(1e3 * tn["total_ms"] / max(tn["launches"], 1), 2),
                                "launches": tn["launches"], "launches_per_step": round(tn["launches"] / max(tn_steps, 1), 2),
                                "family_us_per_step": round(1e3 * tn["total_ms"] / max(tn_steps, 1), 1),
                                "share_of_step": round(tn["total_ms"] / max(tn_steps, 1) / (el * 1e3 / steps), 3),
                                "bracketed_steps": tn_steps,
                                "measured": "HIP events around the family's launches over %d steps after the timed region" % tn_steps
                                            if tn_prof is not None else "--profile-all: inside the timed region",
                                "traffic": tn_traffic["hbm_bytes_per_launch"] if tn_traffic else None,
                                "slab_reduce_traffic": tn_red["hbm_bytes_per_launch"] if tn_red else None,
                                "traffic_source": tn_traffic["source"] if tn_traffic else None}
                               if tn["launches"] else None),
            "gemm_nt_small_kernel": ({"launches": sm["launches"],
                                      "avg_launch_us": round(1e3 * sm["total_ms"] / max(sm["launches"], 1), 2),
                                      "share_of_step": round(sm["total_ms"] / max(profiled_steps, 1) / (el * 1e3 / steps), 3)}
                                     if sm["launches"] else None),
            # the whole step is occupancy-, epilogue- and launch-structure-bound, not roofline-bound: reported as what
            # it executes, against the peaks of the units it uses
            "whole_step": {"executed_flops_per_triplet": round(exe / args.batch),
                           "executed_tflops": round(exe / args.batch * value / world / 1e12, 2),
                           "executed_frac_of_matrix_peak": round(exe / args.batch * value / world / 1e12 / NT_PEAK_TFLOPS, 4),
                           "reference_equivalent_flops_per_triplet": flops_per_triplet(round(n_avg), dim),
                           "gather_bytes_per_triplet": bytes_per_triplet(round(n_avg), dim),
                           "frac_hbm_gather": round(bytes_per_triplet(n_avg, dim) * value / world / 1e9 / HBM_PEAK_GBS, 5),
                           # every kernel of the step, from the committed counter passes of the same command (loader included)
                           "hbm_traffic_bytes": step_pmc["hbm_bytes_per_launch"] if step_pmc else None,
                           "hbm_traffic_source": step_pmc["source"] if step_pmc else None,
                           "hbm_traffic_over_gather_bytes": (round(step_pmc["hbm_bytes_per_launch"] / (bytes_per_triplet(n_avg, dim) * args.batch), 1)
                                                             if step_pmc else None)}}
    if want_cpu and last is not None:
        # `last` is a set of views into one slot of the loader's buffer ring (reuse_buffers=True); the sustained leg below
        # rebuilds that slot many times over, with other row counts.  The CPU baseline is timed on THIS batch: keep a copy.
        cl = lambda v: v.clone() if torch.is_tensor(v) else v
        last = {k: ({kk: (int(vv) if kk == "n_unique" else cl(vv)) for kk, vv in v.items()} if isinstance(v, dict) else cl(v))
                for k, v in last.items()}
    if sustained:
        # The driver's flags make the headline region short (20 steps = 22 ms) and it never crosses a loader epoch boundary
        # (every 67 steps).  This leg is the same loop, un-bracketed, over >= 300 steps after >= 20 warm-up steps, three
        # times: min / median / max of the per-repeat ms_per_step.
        reps = []
        steps_s = min(max(300, int(np.ceil(3.2 * len(loader)))), 1000)   # >= 3 epoch boundaries inside every repeat (configs[1]: 67 steps per epoch)
        for _ in range(20):
            step(next(it))
        for _rep in range(3):
            if multi:
                torch.cuda.synchronize()
                torch.distributed.barrier()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(steps_s):
                step(next(it))
            torch.cuda.synchronize()
            if multi:
                torch.distributed.barrier()
            tt = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=dev)
            if multi:
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            reps.append(1e3 * float(tt) / steps_s)
        reps.sort()
        res["sustained"] = {"steps": steps_s, "warmup": 20, "repeats": 3, "epoch_boundaries_per_repeat": steps_s // max(len(loader), 1),
                            "ms_per_step": {"min": round(reps[0], 4), "median": round(reps[1], 4), "max": round(reps[2], 4)},
                            "value_median": round(world * args.batch / (reps[1] * 1e-3), 1), "unit": "triplets/s"}
    res["catalogue"] = {"products": products, "dim": dim, "generator": "device (csrc/generator.hip)" if on_device else "host (numpy)",
                        "seconds": round(t_gen, 3), "negatives": negatives, "table": table_mode, "dropout": float(dropout),
                        "neighbour_layout": "unique rows" if loader.unique else "compact (every real slot its own row)",
                        "hbm_bytes": bpg.nbytes() if on_device else None,
                        "similarity_pairs": len(loader) * args.batch}
    if want_cpu and sharded is None and not on_device and dim == 128:    # (the oracle reads host arrays; a sharded batch indexes its own gathered table)
        res["cpu_baseline"] = p2v_cpu_baseline(bpg, last, dropout=float(dropout))
    prof.close()
    if tn_prof is not None:
        tn_prof.close()
    return res


def run_dropin_dense(args, dev, steps, warmup):
    """The LITERAL plug-in surface (three import lines change): the reference's loop body
    (product2vec.py:126-164), statement for statement, over device-resident batches in the reference's collate format
    (anchor [B,128], positive [B,128], negative [B,5,128], anchor_neighbors [B,N,128] zero-padded), `model(...)` four times
    through the autograd Functions of p_companion_amd.product2vec (every padded slot its own row: no identical-row merging, no
    index form), torch's own pairwise_distance / relu / mean, loss.backward(), torch.optim.Adam.  One process."""
    from types import SimpleNamespace
    import torch.nn.functional as F
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import Product2Vec
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                          MARGIN=1.0, BATCH_SIZE=args.batch, LEARNING_RATE=1e-3, DEVICE=dev)
    bpg = run_joint.bpg if getattr(run_joint, "bpg", None) is not None else generate_scaled_bpg(args.products, args.types, seed=0)
    table = bpg.cuda(dev)["features"]
    ftab = torch.cat([table, torch.zeros(1, table.shape[1], device=dev)])           # slot -1 = collate_fn's zero padding row
    loader = SimilarityIndexLoader(bpg, args.batch, shuffle=True, sampler="philox", seed=1, drop_last=True, device=dev,
                                   compact=False, prefetch=False, unique=False)
    batches = []
    for b in loader:                                          # four distinct batches, resident, in the reference's format
        batches.append({"anchor": table[b["anchor_idx"].long()], "positive": table[b["positive_idx"].long()],
                        "negative": table[b["negative_idx"].long()], "anchor_neighbors": ftab[b["neighbor_idx"].long()],
                        "anchor_ids": None})
        if len(batches) == 4:
            break
    torch.manual_seed(0)
    self = Product2Vec(cfg).to(dev)
    optimizer = torch.optim.Adam(self.parameters(), lr=cfg.LEARNING_RATE)
    self.train()
    device = dev

    def body(batch):
        batch = {k: v.to(device) if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
        anchor_emb = self(batch['anchor'], batch.get('anchor_neighbors'))
        positive_emb = self(batch['positive'])
        negative_emb = self(batch['negative'])
        pos_distance = F.pairwise_distance(anchor_emb, positive_emb)
        anchor_expanded = anchor_emb.unsqueeze(1).expand(-1, negative_emb.size(1), -1)
        neg_distance = torch.mean(F.pairwise_distance(anchor_expanded, negative_emb, p=2), dim=1)
        loss = F.relu(self.config.MARGIN - pos_distance + neg_distance).mean()
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        return loss

    for i in range(warmup):
        body(batches[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = body(batches[i % 4])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n = batches[0]["anchor_neighbors"].shape[1]
    return {"value": round(args.batch * steps / el, 1), "unit": "triplets/s", "ms_per_step": round(1e3 * el / steps, 4), "steps": steps,
            "final_loss": round(float(loss), 5),
            "workload": f"the reference's loop body (product2vec.py:126-164) unmodified over device-resident reference-format batches "
                        f"(B={args.batch}, N={n} padded slots, 5 negatives; {sum(t.numel() * 4 for t in batches[0].values() if torch.is_tensor(t)) / 1e6:.0f} MB "
                        "per batch), model(...) x 4 through the HIP autograd Functions, torch pairwise_distance, loss.backward(), "
                        "torch.optim.Adam; batches handed over from host memory add their PCIe time (DESIGN.md section 7)"}


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set), relay rank 0's JSON line and
    exit with the worst child's code.  Runs BEFORE anything touches the GPU in this process: it only counts devices
    (torch.cuda.device_count() does not initialise HIP) and waits for its children."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    forced = os.environ.get("PC_FORCE_DEVICE")               # (rehearsal: every rank on one card, with PC_DIST_BACKEND=gloo)
    if ndev < n and forced is None:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, this node shows {ndev} (one process per GPU; set "
              "PC_FORCE_DEVICE + PC_DIST_BACKEND=gloo only to rehearse the N-rank code path on one card)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PC_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []
    rd = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    worst, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                worst = worst or rc
                for o in sorted(live):                           # a rank died: its peers would sit in a collective until the
                    procs[o].kill()                              # backend's timeout -- end exactly these children, by handle
        if live:
            time.sleep(0.05)
    rd.join(10.0)
    # rank 0's JSON line goes to stdout alone; anything else a library printed there (gloo announces its connections on stdout)
    # is passed on to stderr
    for ln in b"".join(chunks).decode(errors="replace").splitlines():
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    return 1 if worst else 0


def rccl_info(world, dev):
    """What the driver needs to SEE N ranks: backend, world size and the ranks an all_gather over the process group returned."""
    import torch.distributed as dist
    from p_companion_amd import distributed as pdist
    if not (pdist.collectives_on(world) and dist.is_initialized()):
        return None
    backend = dist.get_backend()
    t = torch.tensor([dist.get_rank()], dtype=torch.int64, device=dev if backend != "gloo" else "cpu")
    seen = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(seen, t)
    return {"backend": backend + (" (RCCL over xGMI)" if backend == "nccl" else ""), "world": world,
            "ranks_seen": sorted(int(v) for v in seen.tolist()), "launcher": "self" if os.environ.get("PC_BENCH_SELF_LAUNCHED") else "torch.distributed.run"}


FINAL = {"emit": None, "rank": 0}     # main() installs the function that prints the line from whatever legs exist so far


def leg_watchdog(name, world):
    """N > 1 only.  A secondary leg that HANGS (a collective some rank never joins: nothing raises) would take the headline
    down with it when the harness kills the job.  Every rank arms the same timer when it enters the leg
    (PC_BENCH_LEG_DEADLINE_S, default 180 s); if the leg is still running then, rank 0 prints the line with what has been
    measured -- the headline leg ran first -- and this leg reported as an error, and every rank leaves with os._exit (a thread
    parked inside a collective cannot be unwound)."""
    if world <= 1:
        return None
    import threading
    # (the headline leg -- catalogue, model, loader set-up, warm-up, the timed region, the sustained leg -- gets its own, longer limit)
    deadline = float(os.environ.get("PC_BENCH_HEADLINE_DEADLINE_S", "300")) if name == "headline" else \
        float(os.environ.get("PC_BENCH_LEG_DEADLINE_S", "180"))

    def expire():
        print(f"[bench] rank {FINAL['rank']}: leg `{name}` still running after {deadline:g} s -- ending the job with the legs measured so far",
              file=sys.stderr, flush=True)
        code = 75
        if FINAL["rank"] == 0 and FINAL["emit"] is not None:
            try:
                FINAL["emit"]({name: {"error": f"did not finish within {deadline:g} s (PC_BENCH_LEG_DEADLINE_S); job ended", "leg": name}})
                code = 0 if name != "headline" else 75
            except Exception:                                   # noqa: BLE001
                pass
        elif FINAL["rank"] != 0:
            code = 0
        sys.stdout.flush()
        os._exit(code)

    t = threading.Timer(deadline, expire)
    t.daemon = True
    t.start()
    return t


def guarded(name, fn, world):
    """A SECONDARY leg must not take the headline down with it (an allocation that does not fit a smaller card, a path no
    multi-GPU box has exercised yet): its failure is reported in its place.  At N > 1 the ranks agree on the outcome (MIN
    all-reduce of an ok flag): a failure every rank hits alike -- same code, same shapes -- is reported by all of them and the
    line still goes out; a failure on SOME ranks only leaves the others inside the leg's collectives, which the process
    group's bounded timeout ends (then the launcher ends the job: nothing can report that case)."""
    err = None
    res = None
    # The line goes out only when every leg is through, and the harness ends the job at ITS limit (600 s, line or no line): a run in
    # which set-up, first contact and the legs before this one were slow -- not hung: the watchdogs see nothing -- must stop adding
    # legs while the line can still be printed.  Past PC_BENCH_TOTAL_BUDGET_S (default 330 s since the process started; a default
    # one-GPU run takes ~180 s) a secondary leg is not started and says so in its place; at N > 1 the ranks agree (MAX of the
    # elapsed times, on the drained device: the previous leg ended behind a synchronize), so that all of them skip the same legs.
    elapsed = time.monotonic() - BENCH_T0
    if world > 1:
        import torch.distributed as dist
        el = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    budget = float(os.environ.get("PC_BENCH_TOTAL_BUDGET_S", "330"))
    if elapsed > budget:
        print(f"[bench] leg `{name}` not started: {elapsed:.0f} s since the process started > PC_BENCH_TOTAL_BUDGET_S = {budget:g} s", file=sys.stderr, flush=True)
        return {"error": f"skipped: {elapsed:.0f} s since the process started > PC_BENCH_TOTAL_BUDGET_S = {budget:g} s (the line has to go out "
                         f"inside the harness's limit)", "leg": name}
    watchdog = leg_watchdog(name, world)
    try:
        res = fn()
    except Exception as e:                                      # noqa: BLE001 -- reported, not swallowed
        import traceback
        traceback.print_exc(file=sys.stderr)
        err = f"{type(e).__name__}: {e}"[:400]
        try:
            torch.cuda.synchronize()
        except Exception:                                       # noqa: BLE001
            pass
    if world > 1:
        import torch.distributed as dist
        flag = torch.tensor([0 if err else 1], dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else torch.device("cuda", torch.cuda.current_device()))
        try:
            torch.cuda.synchronize()      # (a loader may still have lookups of batches nobody will train on in flight on the library's communicator)
        except Exception:                 # noqa: BLE001
            pass
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and err is None:
            err = "failed on another rank"
    if watchdog is not None:
        watchdog.cancel()
    if err:
        return {"error": err, "leg": name}
    return res


def leg(res, keys=("value", "ms_per_step", "host_enqueue_ms_per_step", "final_loss", "catalogue", "roofline", "sharded_lookup", "cpu_baseline")):
    """A secondary Product2Vec leg of the line: its own value / ms_per_step / roofline, nothing borrowed from the headline."""
    if res is None or "error" in res:
        return res
    out = {"unit": "triplets/s"}
    for k in keys:
        if res.get(k) is not None:
            out[k] = round(res[k], 4 if k != "value" else 1) if isinstance(res[k], float) else res[k]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--products", type=int, default=100_000)
    ap.add_argument("--types", type=int, default=100)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--table", choices=["replicated", "sharded"], default="replicated")
    ap.add_argument("--dim", type=int, default=128, choices=[128, 256], help="PRODUCT_EMB_DIM (256: BASELINE configs[4])")
    ap.add_argument("--negatives", choices=["uniform", "zipf"], default="uniform", help="zipf: configs[4] (P(rank) ~ 1/rank, product 0 the most popular)")
    ap.add_argument("--generator", choices=["auto", "host", "device"], default="auto",
                    help="where the synthetic catalogue is drawn: host numpy (<= 2 M products) or straight into HBM")
    ap.add_argument("--phase", choices=["both", "p2v", "joint"], default="both",
                    help="both (default) = BASELINE's whole metric: configs[1] as the headline value + configs[2] as `joint`")
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="DROPOUT of the headline legs (config.py:12 ships 0.1; 0.0 is the parity setting of every golden test). "
                         "The default line carries the 0.1 legs beside the 0.0 ones either way")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N > 1: BatchNorm statistics over all replicas' rows (two 16 KB all-reduces per step) instead of "
                         "each replica's own batch")
    ap.add_argument("--profile-all", action="store_true",
                    help="HIP-event brackets around every GEMM launch (TN and few-row kernels too), not only the dominant "
                         "gemm_nt_kernel family: ~60 us/step of event packets")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="joint phase: plain module calls (PCompanion.train_step + optimizer) per step")
    ap.add_argument("--joint-launch", choices=["epoch", "auto", "direct", "graph"], default="epoch",
                    help="joint phase, one process: 'direct' = the fused step with its arguments resolved once; 'graph' = HIP-graph replay")
    ap.add_argument("--large-catalogue", type=int, default=0,
                    help="also time the P2V step over this many products (e.g. 2000000: few duplicate neighbours to merge)")
    ap.add_argument("--exchange", choices=["native", "hook"], default="native",
                    help="N > 1: the gradient exchange through the library's exchange slot (its own RCCL communicator, issued from the "
                         "step's call) or from Python hooks per step (torch.distributed; the joint step's [T,64] tables as row lists)")
    ap.add_argument("--no-large", action="store_true",
                    help="skip the `large_catalogue` legs (BASELINE configs[3]: 10 M products through the row-sharded lookup; "
                         "configs[4]: 100 M products x 256, Zipf negatives; both generated in HBM)")
    ap.add_argument("--no-dropout-legs", action="store_true", help="skip the legs at the reference's shipped DROPOUT = 0.1 (config.py:12)")
    ap.add_argument("--no-ref-types", action="store_true", help="skip the joint leg at the reference's NUM_TYPES = 34800")
    ap.add_argument("--hot-rows", type=int, default=1024,
                    help="replicated hot set under the row-sharded table with Zipf negatives (configs[4] at N > 1, and the `hot_set` leg): "
                         "this many most popular products live on every rank; 0 = off")
    ap.add_argument("--no-dropin", action="store_true", help="skip the `dropin_dense` leg (the reference's loop body over reference-format batches)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the `sustained` leg (3 x >= 300 P2V steps across epoch boundaries)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))           # (nothing has touched the GPU in this process)

    from p_companion_amd import distributed as pdist
    if os.environ.get("PC_BENCH_SET_OPTIONS"):
        # developer A/B on one box: "3=1,2=0" -> pc_set_option(3, 1), pc_set_option(2, 0) before anything runs (include/pcompanion_hip.h PC_OPT_*)
        from p_companion_amd import _lib
        for kv in os.environ["PC_BENCH_SET_OPTIONS"].split(","):
            if kv.startswith("rows="):
                LOADER_OPTS["step_rows"] = bool(int(kv[5:]))       # (the loader's row concatenation, a Python-side switch)
                continue
            if kv.startswith("kick="):
                LOADER_OPTS["kick_after_step"] = bool(int(kv[5:]))  # (the step queues the loader's look-ahead builder)
                continue
            o, v = (int(x) for x in kv.split("="))
            if _lib.lib().pc_set_option(o, v) != 0:
                raise SystemExit(f"PC_BENCH_SET_OPTIONS: pc_set_option({o}, {v}) refused")
    rank, world, local = pdist.init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = int(os.environ.get("PC_FORCE_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    want_cpu = not args.no_cpu_baseline and world == 1
    plain = args.products == 100_000 and args.dim == 128 and args.table == "replicated" and args.negatives == "uniform"
    rccl = rccl_info(world, dev)
    if pdist.collectives_on(world) and args.exchange != "hook":
        # the replicas' gradient exchange through the library's slot: its own RCCL communicator ('native'; verified by one
        # all-reduce of a known vector and agreed over the ranks, else the host-driven form takes over) -- gloo rehearsals get
        # torch.distributed behind the same slot
        EXCHANGE["ex"] = pdist.make_exchange(world, rank=rank, kind="auto", device=dev)
        if rccl is not None:
            rccl["exchange"] = EXCHANGE["ex"].kind
    elif rccl is not None:
        rccl["exchange"] = "python hooks (torch.distributed), --exchange hook"
    if rccl is not None:
        if pdist.last_probe is not None:
            rccl["native_probe"] = pdist.last_probe              # {"native", "reason" (the fallback's, if any), "seconds"}
        rccl["timeouts_s"] = {"probe": pdist.probe_timeout_s(), "process_group": pdist.group_timeout_s(),
                              "secondary_leg": float(os.environ.get("PC_BENCH_LEG_DEADLINE_S", "180"))}
        if rank == 0:
            # first contact with a multi-GPU node: what the ranks agreed on, BEFORE the first timed leg (stderr: stdout carries
            # the one JSON line)
            print("[bench] rccl " + json.dumps(rccl), file=sys.stderr, flush=True)
    FINAL["rank"] = rank

    p2v = joint = joint_ref = large = None
    extra = {}
    def emit(errors=None):
        """Print THE line from the legs measured so far (closure over main's leg variables).  errors: {leg name: {"error": ...}} of a
        leg the watchdog gave up on."""
        if p2v is None and joint is None:
            print(json.dumps({"metric": "triplets/sec (Product2Vec pretrain step: gather+fwd+bwd+Adam)", "value": None, "unit": "triplets/s",
                              "n_gpus": world, "error": "the headline leg did not complete", **(errors or {})}), flush=True)
            return
        common = {"unit": "triplets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                  "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
        if rccl:
            common["rccl"] = rccl
        # developer knobs compiled into the library (pc_build_flags: PC_EXP_* builds compute WRONG numbers by design): 0 for the product
        from p_companion_amd import _lib as _l
        common["build_flags"] = int(_l.lib().pc_build_flags())
        if common["build_flags"]:
            print(f"[bench] WARNING: developer-knob library (pc_build_flags = {common['build_flags']:#x}): not the product's numbers",
                  file=sys.stderr, flush=True)
        if p2v is None:                                           # --phase joint: the joint step is the line
            out = dict(common)
            out.update(joint)
            out["steps"] = joint["steps"]
            if joint_ref:
                out["joint_num_types_34800"] = joint_ref
            out.update(extra)
            print(json.dumps(out), flush=True)
            return
        out = {"metric": "triplets/sec (Product2Vec pretrain step: gather+fwd+bwd+Adam)" +
                         (" + P-Companion joint step under `joint`" if joint else ""),
               "value": round(p2v["value"], 1), **common, "ms_per_step": round(p2v["ms_per_step"], 4),
               "host_enqueue_ms_per_step": p2v.get("host_enqueue_ms_per_step"),
               "host_enqueue_note": "wall time of the host between the region's two clocks / steps: its own work (loader hand-out, step wrapper, "
                                    "the foreign call's launches, the look-ahead builder: scripts/dev/first_step_probe.py takes it apart) plus the "
                                    "time it waits behind a full launch queue; below ms_per_step = the device is the bound, not the host",
               "dtype_note": "fp32 storage, accumulation and result accuracy; the large GEMMs evaluate each fp32 product as six "
                             "bf16 matrix-core products of a three-way split (error <= the fp32 MFMA's, tests/test_gpu_ops.py)",
               "config": {"workload": f"Product2Vec GAT pretrain, {args.products} products, {args.types} types, dim={args.dim}, "
                                      f"batch={args.batch}/GPU, 5 negatives, neighbours padded to batch max "
                                      f"(avg N={p2v['n_avg']:.1f})", "global_batch": world * args.batch,
                          "dropout": args.dropout,
                          "dropout_note": "DROPOUT = 0.0 is the parity setting (ATen's mask stream cannot be reproduced: every golden-vector "
                                          "test runs at 0); the reference ships config.py:12 DROPOUT = 0.1 -- legs `p2v_dropout_0p1`, "
                                          "`joint_dropout_0p1`, `joint_num_types_34800_dropout_0p1` of this line run that setting",
                          "table": args.table, "sharded_lookup": p2v.get("sharded_lookup"), "parallelism": f"dp{world}",
                          "batchnorm": "cross-replica" if (args.sync_bn and pdist.collectives_on(world)) else "per-replica",
                          "final_loss": round(p2v["final_loss"], 5),
                          "neighbour_rows": "identical rows of the neighbour call carried once: avg %.0f distinct products (%.0f "
                                            "real slots) + 1 shared padding row, of %d neighbour slots per step; the duplicates "
                                            "are a property of the catalogue (%.0f %% of all FFN rows saved at %d products; none to "
                                            "speak of at 10 M / 100 M products: `large_catalogue` of this line)"
                                            % (p2v["distinct_neighbour_rows"], p2v["real_neighbour_slots"],
                                               args.batch * round(p2v["n_avg"]), 100 * p2v["rows_saved_by_duplicate_neighbours"],
                                               args.products)},
               "sustained": p2v.get("sustained"), "catalogue": p2v.get("catalogue"),
               "roofline": p2v.get("roofline"), "cpu_baseline": p2v.get("cpu_baseline")}
        if large:
            out["large_catalogue_extra"] = {"products": args.large_catalogue, "value": round(large["value"], 1),
                                            "ms_per_step": round(large["ms_per_step"], 4),
                                            "rows_saved_by_duplicate_neighbours": large["rows_saved_by_duplicate_neighbours"]}
        if joint:
            out["joint"] = joint
        if joint_ref:
            out["joint_num_types_34800"] = joint_ref
        out.update(extra)
        out.update(errors or {})
        print(json.dumps(out), flush=True)
    if rank == 0:
        FINAL["emit"] = emit
    headline_dog = leg_watchdog("headline", world)               # (N > 1: a headline leg that hangs ends the job with an error line)
    if args.phase in ("both", "p2v"):
        p2v = run_p2v(args, rank, world, dev, args.products, args.steps, args.warmup, want_cpu, sustained=not args.no_sustained,
                      dropout=args.dropout, pmc_kind="" if plain else ("big" if args.products == 100_000_000 else "cfg3" if args.products == 10_000_000 else "none"))
        if headline_dog is not None:
            headline_dog.cancel()
            headline_dog = None
        if args.large_catalogue:
            large = run_p2v(args, rank, world, dev, args.large_catalogue, max(args.steps // 2, 5), args.warmup, False,
                            profile_kernels=False)
        if plain and not args.no_dropout_legs and args.dropout == 0.0:
            r = guarded("p2v_dropout_0p1", lambda: run_p2v(args, rank, world, dev, args.products, max(args.steps, 30), args.warmup, want_cpu,
                                                           dropout=0.1, pmc_kind="p2vd"), world)
            if rank == 0:
                extra["p2v_dropout_0p1"] = leg(r)
        if plain and world == 1 and not args.no_dropin:
            # the literal plug-in surface: the reference's loop body over reference-format batches (one process: the reference has no N > 1)
            extra["dropin_dense"] = guarded("dropin_dense", lambda: run_dropin_dense(args, dev, max(args.steps, 20), 5), world)
    if headline_dog is not None:
        headline_dog.cancel()
    if args.phase in ("both", "joint"):
        # (with the Product2Vec phase in front, the joint legs are secondary to the headline value: guarded like the others)
        first = (lambda name, fn: fn()) if args.phase == "joint" else (lambda name, fn: guarded(name, fn, world))
        joint = first("joint", lambda: run_joint(args, rank, world, dev, args.types, max(args.steps * 4, JOINT_MIN_STEPS), max(args.warmup, 10), want_cpu,
                                                 dropout=args.dropout))
        if not args.no_dropout_legs and args.dropout == 0.0:
            extra["joint_dropout_0p1"] = guarded("joint_dropout_0p1", lambda: run_joint(args, rank, world, dev, args.types, max(args.steps * 4, JOINT_MIN_STEPS),
                                                                                         max(args.warmup, 10), want_cpu, dropout=0.1), world)
        if not args.no_ref_types and args.types != 34800:
            joint_ref = guarded("joint_num_types_34800", lambda: run_joint(args, rank, world, dev, 34800, max(args.steps, JOINT_REF_MIN_STEPS), max(args.warmup, 10),
                                                                           want_cpu, dropout=args.dropout), world)
            if not args.no_dropout_legs and args.dropout == 0.0:
                extra["joint_num_types_34800_dropout_0p1"] = guarded(
                    "joint_num_types_34800_dropout_0p1",
                    lambda: run_joint(args, rank, world, dev, 34800, max(args.steps, JOINT_REF_MIN_STEPS), max(args.warmup, 10), want_cpu, dropout=0.1), world)
    if args.phase in ("both", "p2v") and plain and not args.no_large:
        # BASELINE configs[3] / [4] at their real sizes, catalogue generated in HBM (this rank's shard of the table when N > 1).
        # One GPU holds configs[4] whole (102 GB of 288); the smaller catalogues are released first.
        run_joint.bpg = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        lc = {}
        r = guarded("config3", lambda: run_p2v(args, rank, world, dev, 10_000_000, 30, 10, False, dim=128, negatives="uniform",
                                               table_mode="sharded", pmc_kind="cfg3"), world)
        if rank == 0:
            lc["config3"] = dict(leg(r), workload=f"BASELINE configs[3]: 10 M products, dim=128, table row-sharded over {world} GPU(s) "
                                                  f"(pc_shard_bucket + two constant-shape all-to-all rounds + pc_gather_rows per batch), batch={args.batch}/GPU")
        del r
        gc.collect()
        torch.cuda.empty_cache()
        r = guarded("config4", lambda: run_p2v(args, rank, world, dev, 100_000_000, 30, 10, False, dim=256, negatives="zipf",
                                               table_mode="sharded" if world > 1 else "replicated", pmc_kind="big",
                                               hot_rows=args.hot_rows), world)
        if rank == 0:
            lc["config4"] = dict(leg(r), workload=f"BASELINE configs[4]: 100 M products, dim=256, Zipf(1) negatives, "
                                                  + (f"table row-sharded over {world} GPUs" if world > 1 else "whole table (102 GB) on one GPU")
                                                  + f", batch={args.batch}/GPU; "
                                                  + (f"the {args.hot_rows} most popular products replicated on every rank (pc_shard_bucket_hot)"
                                                     if world > 1 and args.hot_rows else "one process holds every row: no lookup to cache")
                                                  + "; an LDS cache of hot rows measured unnecessary (DESIGN.md section 7: gather traffic 1.02x algorithmic)")
        del r
        gc.collect()
        torch.cuda.empty_cache()
        if args.hot_rows:
            # the replicated hot set at work where one card can show it: 10 M products, Zipf negatives, the sharded lookup chain
            # (pc_shard_bucket[_hot] + two constant-shape exchange rounds + pc_gather_rows; G = 1: the rounds are copies) with and
            # without the H most popular products served from the replica
            hs = {}
            for name, h in (("without", 0), ("with", args.hot_rows)):
                rr = guarded("hot_set_" + name, lambda h=h: run_p2v(args, rank, world, dev, 10_000_000, 20, 8, False, dim=128, negatives="zipf",
                                                                    table_mode="sharded", pmc_kind="none", hot_rows=h, profile_kernels=False), world)
                if rank == 0:
                    hs[name] = leg(rr, keys=("value", "ms_per_step", "final_loss", "sharded_lookup"))
                del rr
                gc.collect()
                torch.cuda.empty_cache()
            if rank == 0:
                lc["hot_set"] = dict(hs, workload=f"10 M products, dim=128, Zipf(1) negatives, table row-sharded over {world} GPU(s), "
                                                  f"hot_rows={args.hot_rows}: request-list entries per batch = ids_per_batch - hot_rows_served_per_batch")
        gc.collect()
        torch.cuda.empty_cache()
        extra["large_catalogue"] = lc
    if rank != 0:
        return
    emit()


if __name__ == "__main__":
    main()
