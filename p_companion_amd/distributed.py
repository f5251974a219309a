"""Multi-GPU plumbing: one process per GPU, torch.distributed over
RCCL/xGMI ('nccl' backend on ROCm) on the GPU box, 'gloo' in the CPU tests.

The path is data-parallel over triplets.  Per step and rank:
  1. (row-sharded table only) lookup all-to-all: ids bucketed by owner (row r lives on rank
     r % G, local row r // G), de-duplicated per destination, exchanged with all_to_all, rows
     gathered by the owner (HIP row gather) and returned with a second all_to_all.  The fused
     step then runs unchanged on the compact table of received rows with remapped indices.
     The table is frozen (p_companion.py:26-29, synthetic_data.py:50-58): no backward exchange.
  2. fused forward/backward on the local triplets (BatchNorm statistics are those of the
     local call groups: the weak-scaling run processes G independent batches per step).
  3. one all-reduce of the flat dense gradient (793 KB for Product2Vec), averaged, then Adam.
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(device_type="cuda"):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # PC_DIST_BACKEND / PC_FORCE_DEVICE: rehearsal of the N>1 path on a one-GPU box (gloo, every rank
        # on the same card); the driver's multi-GPU runs use neither
        backend = os.environ.get("PC_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda":
            local = int(os.environ.get("PC_FORCE_DEVICE", local))
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def all_reduce_mean_(flat, world):
    """Dense-gradient exchange: one bucket (the whole flat gradient buffer)."""
    if world > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / world)
    return flat


class ShardedFeatureTable:
    """[P,D] feature table row-sharded cyclically over the ranks of `group`.

    lookup(ids) returns (rows[U,D], remap) with rows[remap[i]] == table[ids[i]] (ids < 0 map
    to -1: the zero-row sentinel of the collate padding).  `gather_fn(local_table, idx_int32)`
    performs the owner-side row gather: the HIP kernel on the GPU; the CPU tests inject their
    own (test infrastructure only -- the product never falls back)."""

    def __init__(self, local_rows, num_products, rank, world, gather_fn=None, group=None):
        self.local = local_rows
        self.P, self.rank, self.world, self.group = int(num_products), rank, world, group
        if gather_fn is None:
            from . import ops
            gather_fn = ops.gather_rows
        self.gather_fn = gather_fn

    @staticmethod
    def shard(full_table, rank, world):
        return full_table[rank::world].contiguous()

    def lookup_batch(self, batch):
        """Index batch over GLOBAL product ids -> (compact table of the rows this rank needs, the same batch over
        that table): what the fused step consumes unchanged (unique / compact neighbour layouts)."""
        nbc = batch["neighbor_compact"]
        uq = "weight" in nbc
        nrows = nbc["nb_rows"][: int(nbc["n_unique"]) + 1] if uq else nbc["nb_rows"]
        ids = torch.cat([batch["anchor_idx"], nrows, batch["positive_idx"], batch["negative_idx"].reshape(-1)])
        tab, remap = self.lookup(ids)
        B, M1, K = batch["anchor_idx"].numel(), nrows.numel(), batch["negative_idx"].shape[1]
        o = np.cumsum([0, B, M1, B, B * K])
        out = {"anchor_idx": remap[o[0]:o[1]].contiguous(), "positive_idx": remap[o[2]:o[3]].contiguous(),
               "negative_idx": remap[o[3]:o[4]].view(B, K).contiguous(),
               "neighbor_compact": dict({"nb_rows": remap[o[1]:o[2]].contiguous(), "slot_row": nbc["slot_row"]},
                                        **({k: nbc[k] for k in ("weight", "n_unique", "ref_off", "ref_slot")} if uq else {}))}
        return tab, out

    def lookup(self, ids):
        dev = ids.device
        flat = ids.reshape(-1).to(torch.int64)
        valid = flat >= 0
        uniq, inv = torch.unique(flat[valid], return_inverse=True)          # de-duplicate before the wire
        owner = uniq % self.world
        order = torch.argsort(owner, stable=True)
        send_ids = (uniq[order] // self.world).to(torch.int32)
        send_counts = torch.bincount(owner, minlength=self.world)
        recv_counts = torch.empty_like(send_counts)
        if self.world > 1:
            dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        else:
            recv_counts.copy_(send_counts)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        req = torch.empty(sum(rc), dtype=torch.int32, device=dev)
        if self.world > 1:
            dist.all_to_all_single(req, send_ids, rc, sc, group=self.group)
        else:
            req.copy_(send_ids)
        rows_out = self.gather_fn(self.local, req) if req.numel() else self.local.new_zeros((0, self.local.shape[1]))
        rows_in = torch.empty(uniq.numel(), self.local.shape[1], dtype=self.local.dtype, device=dev)
        if self.world > 1:
            dist.all_to_all_single(rows_in, rows_out, sc, rc, group=self.group)
        else:
            rows_in.copy_(rows_out)
        # rows_in is in `order`; position of uniq[j] in rows_in = rank of j in order
        pos = torch.empty_like(order)
        pos[order] = torch.arange(order.numel(), device=dev)
        remap = torch.full_like(flat, -1)
        remap[valid] = pos[inv]
        return rows_in, remap.to(torch.int32).reshape(ids.shape)
