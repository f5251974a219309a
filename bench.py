#!/usr/bin/env python3
"""bench.py -- triplets/sec of the Product2Vec GAT triplet pretrain step on MI355X.

Workload (BASELINE.json configs[1]): 100k products, 100 types, D=128,
B=4096 triplets per GPU and step, 5 negatives, neighbours padded to the batch max (<=32),
synthetic BPG (data='synthetic', random-init weights).  One step = device batch build
(Philox negative sampling + CSR neighbour rows) + gather + 4 FFN/BatchNorm call groups +
attention + triplet hinge + full backward + Adam: everything inside the timed region.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant
kernel gemm_nt_kernel: algorithmic FLOPs / HIP-event time on the launch stream, against the
matrix-core peak of the instruction it issues: fp32-grade products as six v_mfma_f32_32x32x16_bf16
over a three-way bf16 split of the fp32 operands, i.e. the dense bf16 peak / 6) and `cpu_baseline` (the oracle port timed on this box's host cores).
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
HBM_PEAK_GBS = 8000.0


def flops_per_triplet(n):
    """fwd 328,192*N + 1,900,544; fwd+bwd = 3*fwd - 65,536*(N+7)."""
    fwd = 328192 * n + 1900544
    return 3 * fwd - 65536 * (n + 7)


def nt_algorithmic_bytes(b, nbc):
    """Compulsory HBM bytes of the 7 persistent gemm_nt_kernel launches of one step:
    rows R = [anchor B | neighbour rows nbc | positive B | negatives 5B]; per row Linear0 0.5+1 KB (gathered
    x -> H0), Linear3 1+1 (H0 -> A2), Linear5 1+0.5 (A2 -> Y), dZ2 0.5+1+1 (dY, A2 -> dZ2), dZ1 1+1+1
    (dZ2, H0 -> dZ1); per neighbour row K|V projection 0.5+1 and dKeys 1+0.5.  Weights are L2-resident."""
    r = 7 * b + nbc
    return 1024 * (r * (1.5 + 2.0 + 1.5 + 2.5 + 3.0) + nbc * (1.5 + 1.5))


def executed_flops_per_step(b, nbc):
    """FLOPs of the GEMMs one step actually runs: FFN forward + backward without dX of Linear0 over
    R = 7b + nbc rows (720,896 per row), K|V projection + dKeys + dW_kv over the nbc neighbour rows and the
    q / out projections with their backward over the b samples (196,608 each)."""
    return 720896.0 * (7 * b + nbc) + 196608.0 * (nbc + b)


def bytes_per_triplet(n):
    return 512 * (n + 7) + 4 * (n + 7)


def pmc_traffic_per_launch():
    """HBM bytes per gemm_nt_kernel launch from the committed rocprofv3 --pmc passes of this same
    command (profiles/*_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE in separate passes, KiB -> bytes,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if no profile is committed:
    PMC counters cannot be collected from inside the bench process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    nt = [v for k, v in d.items() if "gemm_nt_kernel<" in k and "<1, 2," not in k]      # the persistent family (not the few-row kernels)
    n = sum(v["launches"] for v in nt)
    if not n:
        return None
    return {"hbm_bytes_per_launch": round(sum((v["fetch_bytes_corrected"] + v["write_bytes"]) * v["launches"] for v in nt) / n),
            "source": os.path.relpath(files[-1], ROOT)}


def cpu_baseline(bpg, batch, seconds=15.0):
    """The oracle (CPU restatement pinned to the reference's golden vectors) timed on this
    box's host cores: same workload shape, pre-gathered dense batch, fwd+bwd+Adam."""
    from oracle import p2v_oracle
    st = p2v_oracle.init_state(0)
    feats = torch.from_numpy(bpg.features)
    nbc = batch["neighbor_compact"]
    nb_dense = nbc["nb_rows"].cpu().numpy()[nbc["slot_row"].cpu().numpy()]      # the reference's padded [B,N] layout
    dense = p2v_oracle.gather_batch(feats, batch["anchor_idx"].cpu().numpy(), batch["positive_idx"].cpu().numpy(),
                                    batch["negative_idx"].cpu().numpy(), nb_dense)
    mom = p2v_oracle.new_moments(st)
    b = dense["anchor"].shape[0]
    p2v_oracle.train_step(st, dense, 1.0, mom, 1)            # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        p2v_oracle.train_step(st, dense, 1.0, mom, n + 2)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds or n >= 50:
            break
    return {"value": b * n / el, "unit": "triplets/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} steps of the same workload (B={b}, N={dense['anchor_neighbors'].shape[1]}, fwd+bwd+Adam, "
                      f"pre-gathered batch) on {os.cpu_count()} host cpus, torch {torch.get_num_threads()} threads"}


def joint_phase(args, rank, world, dev):
    """BASELINE configs[2]: 100k products, T types, B pairs per GPU: PCompanion forward + both hinge
    losses + backward + Adam as pc_joint_train_step / pc_adam_step.  HBM/launch-latency bound
    (~2.6 KB and ~0.3 MFLOP per triplet): the roofline object reports the gather bandwidth."""
    from types import SimpleNamespace
    from p_companion_amd import distributed as pdist
    from p_companion_amd.data import ComplementaryIndexDataset, ComplementaryIndexLoader, generate_scaled_bpg
    from p_companion_amd.p_companion import PCompanion
    from p_companion_amd.product2vec import FusedAdam
    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, DROPOUT=0.0, MARGIN=1.0, ALPHA=0.8, NUM_COMP_TYPES=3,
                          NUM_TYPES=args.types, DEVICE=dev)
    bpg = generate_scaled_bpg(args.products, min(args.types, 100), seed=0)
    torch.manual_seed(0)
    model = PCompanion(cfg, bpg.cuda(dev)["features"]).to(dev).train()       # frozen table: synthetic stand-in for the P2V export
    opt = FusedAdam(model, lr=1e-3)
    flat, gflat = model.flatten_parameters()
    # one process: the fixed-shape step (pc_joint_train_step + pc_adam_step) is captured once as a HIP graph and
    # replayed, the loader builds each batch straight into the graph's input buffers.  N > 1 keeps eager launches
    # (the gradient all-reduce sits between the two calls).
    graphed = None
    if world == 1 and not args.no_graph:
        from p_companion_amd.p_companion import GraphedJointStep
        graphed = GraphedJointStep(model, opt, args.batch)
    loader = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), args.batch, shuffle=True, seed=rank,
                                      device=dev, out=graphed.static if graphed else None)

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
        pdist.all_reduce_mean_(gflat, world)
        opt.step()
        return losses

    for _ in range(args.warmup):
        step(next(it))
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step(next(it))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    el = float(t)
    if rank != 0:
        return
    value = world * args.batch * args.steps / el
    bytes_per = 2600.0
    out = {"metric": "triplets/sec (P-Companion joint step: fwd + type/item hinge + bwd + Adam)", "value": round(value, 1),
           "unit": "triplets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * el / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"P-Companion joint step, {args.products} products, {args.types} types, dim=128, "
                                  f"batch={args.batch}/GPU, K=3 (loader batch construction included)",
                      "global_batch": world * args.batch, "parallelism": f"dp{world}", "final_loss": round(float(losses[0]), 5),
                      "launch": "hipGraph replay" if graphed is not None else "eager"},
           "roofline": {"bound": "hbm", "achieved": round(bytes_per * value / world / 1e9, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(bytes_per * value / world / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                        "note": "2.6 KB gathered per triplet (SURVEY 8d); the step is ~30 launches of a few us each: launch/latency bound"},
           "cpu_baseline": None}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--products", type=int, default=100_000)
    ap.add_argument("--types", type=int, default=100)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--table", choices=["replicated", "sharded"], default="replicated")
    ap.add_argument("--phase", choices=["p2v", "joint"], default="p2v",
                    help="p2v = BASELINE configs[1] (the headline line); joint = configs[2], the P-Companion joint step")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N > 1: BatchNorm statistics over all replicas' rows (two 16 KB all-reduces per step) instead of "
                         "each replica's own batch")
    ap.add_argument("--profile-all", action="store_true",
                    help="HIP-event brackets around every GEMM launch (TN and few-row kernels too), not only the dominant "
                         "gemm_nt_kernel family: ~60 us/step of event packets")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="joint phase: eager launches instead of the HIP-graph replay")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    from p_companion_amd import distributed as pdist
    rank, world, local = pdist.init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    local = int(os.environ.get("PC_FORCE_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.phase == "joint":
        return joint_phase(args, rank, world, dev)

    from types import SimpleNamespace
    from p_companion_amd import ops
    from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
    from p_companion_amd.product2vec import FusedAdam, Product2Vec

    cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                          MARGIN=1.0, BATCH_SIZE=args.batch, LEARNING_RATE=1e-3, DEVICE=dev)
    bpg = generate_scaled_bpg(args.products, args.types, seed=0)
    torch.manual_seed(0)
    model = Product2Vec(cfg).to(dev)
    model.train()
    opt = FusedAdam(model, lr=cfg.LEARNING_RATE)
    flat, gflat = model.flatten_parameters()
    loader = SimilarityIndexLoader(bpg, args.batch, shuffle=True, sampler="philox", seed=1 + rank, drop_last=True,
                                   device=dev)
    table = bpg.cuda(dev)["features"]
    sharded = None
    if args.table == "sharded":
        sharded = pdist.ShardedFeatureTable(pdist.ShardedFeatureTable.shard(table, rank, world), bpg.num_products,
                                            rank, world)

    def batches():
        while True:
            for b in loader:
                yield b

    it = batches()
    prof = ops.KernelProfile(capacity=32 * max(args.steps, 1))
    if not args.profile_all:
        prof.set_kinds(["gemm_nt_kernel"])
    n_sum = 0
    real_sum = 0
    slot_sum = 0

    def step(b, profile=None):
        tab = table
        if sharded is not None:
            nbc = b["neighbor_compact"]
            uq = "weight" in nbc
            nrows = nbc["nb_rows"][: int(nbc["n_unique"]) + 1] if uq else nbc["nb_rows"]
            ids = torch.cat([b["anchor_idx"], nrows, b["positive_idx"], b["negative_idx"].reshape(-1)])
            tab, remap = sharded.lookup(ids)
            B, M1, K = b["anchor_idx"].numel(), nrows.numel(), b["negative_idx"].shape[1]
            o = np.cumsum([0, B, M1, B, B * K])
            b = {"anchor_idx": remap[o[0]:o[1]].contiguous(), "positive_idx": remap[o[2]:o[3]].contiguous(),
                 "negative_idx": remap[o[3]:o[4]].view(B, K).contiguous(),
                 "neighbor_compact": dict({"nb_rows": remap[o[1]:o[2]].contiguous(), "slot_row": nbc["slot_row"]},
                                          **({k: nbc[k] for k in ("weight", "n_unique", "ref_off", "ref_slot")} if uq else {}))}
        sync = (lambda t: torch.distributed.all_reduce(t)) if (args.sync_bn and world > 1) else None
        loss = model.train_step_indexed(tab, b, profile=profile, sync_reduce=sync)
        pdist.all_reduce_mean_(gflat, world)
        opt.step()
        return loss

    last = None
    for _ in range(args.warmup):
        last = next(it)
        step(last)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = next(it)
        n_sum += last["n_pad"]
        nbc_ = last["neighbor_compact"]
        real_sum += int(nbc_["n_unique"]) if "weight" in nbc_ else nbc_["nb_rows"].numel() - 1      # rows carried
        slot_sum += nbc_.get("n_real", nbc_["nb_rows"].numel() - 1)                                   # real slots
        loss = step(last, profile=prof)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    el = time.perf_counter() - t0
    t = torch.tensor([el], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    el = float(t)
    if rank != 0:
        return

    n_avg = n_sum / max(args.steps, 1)
    value = world * args.batch * args.steps / el
    nt = prof.summary("gemm_nt_kernel")
    tn = prof.summary("gemm_tn_kernel")
    sm = prof.summary("gemm_nt_small_kernel")
    traffic = pmc_traffic_per_launch()
    achieved = nt["total_flops"] / (nt["total_ms"] * 1e-3) / 1e12 if nt["total_ms"] > 0 else 0.0
    roof = {"bound": "mfma", "kernel": "gemm_nt_kernel", "achieved": round(achieved, 2),
            "peak": round(NT_PEAK_TFLOPS, 1), "unit": "TFLOP/s", "frac": round(achieved / NT_PEAK_TFLOPS, 4),
            "peak_note": "fp32 in / fp32 accumulate / fp32-grade result on the bf16 matrix cores: 6 bf16 MFMA products per "
                         "fp32 product, so peak = 2500 dense bf16 TFLOP/s / 6; `achieved` counts each fp32 product once",
            "executed_bf16_tflops": round(achieved * BF16_PRODUCTS, 1),
            "frac_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
            "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
            "traffic_source": traffic["source"] if traffic else None,
            "algorithmic_bytes_per_launch": round(nt_algorithmic_bytes(args.batch, real_sum / max(args.steps, 1) + 1) / 7),
            "launches": nt["launches"], "avg_launch_us": round(1e3 * nt["total_ms"] / max(nt["launches"], 1), 2),
            "flops_per_launch": nt["total_flops"] / max(nt["launches"], 1),
            "share_of_step": round(nt["total_ms"] / (el * 1e3), 3),
            "gemm_tn_kernel": ({"achieved": round(tn["total_flops"] / max(tn["total_ms"], 1e-9) / 1e9, 2),
                                "launches": tn["launches"], "share_of_step": round(tn["total_ms"] / (el * 1e3), 3)}
                               if tn["launches"] else None),                     # bracketed with --profile-all only
            "gemm_nt_small_kernel": ({"launches": sm["launches"],
                                      "avg_launch_us": round(1e3 * sm["total_ms"] / max(sm["launches"], 1), 2),
                                      "share_of_step": round(sm["total_ms"] / (el * 1e3), 3)} if sm["launches"] else None),
            "whole_step": {"flops_per_triplet": flops_per_triplet(round(n_avg)),
                           # `achieved` prices the step at the reference's dense formulation (every slot its own
                           # row); `executed` counts the rows actually multiplied (identical rows carried once)
                           "executed_flops_per_triplet": round(executed_flops_per_step(args.batch, real_sum / max(args.steps, 1) + 1) / args.batch),
                           "executed": round(executed_flops_per_step(args.batch, real_sum / max(args.steps, 1) + 1) / args.batch * value / world / 1e12, 2),
                           "achieved": round(flops_per_triplet(n_avg) * value / world / 1e12, 2),
                           "frac_mfma": round(flops_per_triplet(n_avg) * value / world / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                           "gather_bytes_per_triplet": bytes_per_triplet(round(n_avg)),
                           "frac_hbm_gather": round(bytes_per_triplet(n_avg) * value / world / 1e9 / HBM_PEAK_GBS, 5)}}
    out = {"metric": "triplets/sec (Product2Vec pretrain step: gather+fwd+bwd+Adam)", "value": round(value, 1),
           "unit": "triplets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * el / args.steps, 4), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "dtype_note": "fp32 storage, accumulation and result accuracy; the large GEMMs evaluate each fp32 product as six "
                         "bf16 matrix-core products of a three-way split (error <= the fp32 MFMA's, tests/test_gpu_ops.py)",
           "config": {"workload": f"Product2Vec GAT pretrain, {args.products} products, {args.types} types, dim=128, "
                                  f"batch={args.batch}/GPU, 5 negatives, neighbours padded to batch max "
                                  f"(avg N={n_avg:.1f})", "global_batch": world * args.batch,
                      "table": args.table, "parallelism": f"dp{world}",
                      "batchnorm": "cross-replica" if (args.sync_bn and world > 1) else "per-replica", "final_loss": round(float(loss), 5),
                      "neighbour_rows": "identical rows of the neighbour call carried once: avg %.0f distinct products (%.0f "
                                        "real slots) + 1 shared padding row, of %d neighbour slots per step"
                                        % (real_sum / max(args.steps, 1), slot_sum / max(args.steps, 1),
                                           args.batch * round(n_avg))},
           "roofline": roof}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(bpg, last, args.cpu_seconds)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
