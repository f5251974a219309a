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


def collectives_on(world):
    """True when the exchange steps go through torch.distributed: always for world > 1; for world == 1 only in the one-rank
    rehearsal (PC_DIST_FORCE=1), which drives the whole N > 1 code path -- process group, every collective call with its
    dtypes and shapes, the gradient hooks -- over RCCL on a one-GPU box (tests/test_gpu_rccl.py)."""
    return world > 1 or os.environ.get("PC_DIST_FORCE", "0") == "1"


def init_from_env(device_type="cuda"):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if collectives_on(world) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # PC_DIST_BACKEND / PC_FORCE_DEVICE: rehearsal of the N>1 path on a one-GPU box (gloo, every rank
        # on the same card); the driver's multi-GPU runs use neither
        backend = os.environ.get("PC_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        if device_type == "cuda":
            local = int(os.environ.get("PC_FORCE_DEVICE", local))
            torch.cuda.set_device(local)
        # (a bounded collective timeout: a rank that died inside a step must end the job, not park its peers for the backend's
        # default half hour)
        import datetime
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=group_timeout_s()))
    return rank, world, local


def group_timeout_s():
    """Collective timeout of the process group (PC_DIST_TIMEOUT_S, default 300 s): a rank that died inside a step ends the
    job well inside a 10-minute harness limit instead of parking its peers for the backend's half hour."""
    return int(os.environ.get("PC_DIST_TIMEOUT_S", "300"))


def probe_timeout_s():
    """Host deadline of the native communicator's construction + probe (PC_DIST_PROBE_TIMEOUT_S, default 90 s, never above the
    group's timeout): first contact with a new node must resolve -- native exchange, torch.distributed fallback, or exit
    code 75 -- long before the harness gives up on the job."""
    return float(min(int(os.environ.get("PC_DIST_PROBE_TIMEOUT_S", "90")), group_timeout_s()))


def all_reduce_mean_(flat, world):
    """Dense-gradient exchange: one bucket (the whole flat gradient buffer)."""
    if collectives_on(world):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.mul_(1.0 / world)
    return flat


class _TorchCollectives:
    """The step's collectives on torch.distributed's communicator (gloo in the CPU tests; the all-torch `--exchange hook` form
    on the GPU): the same three calls RcclExchange offers, so ShardedFeatureTable and the BatchNorm sums take either."""

    def __init__(self, group=None):
        self.group = group

    def all_to_all(self, send, recv):
        dist.all_to_all_single(recv, send, group=self.group)          # equal splits
        return recv

    def all_reduce_sum_f64_(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


class _Expired(Exception):
    pass


def _run_with_deadline(fn, seconds, what, rank, on_expire="exit"):
    """fn() on a helper thread, waited for at most `seconds`.  ncclCommInitRank and the first collectives are host-blocking
    rendezvous: a peer that died before joining leaves this rank inside them for good, and nothing in-process can unwind a
    thread parked in a collective library.  on_expire="exit": the PROCESS ends with code 75 (no re-exec, no retry): the
    launcher then ends the job, which is the only recovery an asymmetric failure has.  on_expire="abandon": _Expired is
    raised and the helper thread is left behind (a daemon: it does not keep the process alive) -- for a step whose failure
    the ranks can still agree on over another communicator.  Exceptions of fn are re-raised here."""
    import threading
    box = {}

    def run():
        try:
            if torch.cuda.is_available():
                torch.cuda.set_device(box["dev"])
            box["res"] = fn()
        except BaseException as e:                            # noqa: BLE001 -- handed to the waiting thread
            box["err"] = e

    box["dev"] = torch.cuda.current_device() if torch.cuda.is_available() else None
    th = threading.Thread(target=run, daemon=True, name="pc-" + what)
    th.start()
    th.join(seconds)
    if th.is_alive():
        import sys
        if on_expire == "abandon":
            print(f"[p_companion_amd] rank {rank}: {what} did not return within {seconds:.0f} s; leaving it behind",
                  file=sys.stderr, flush=True)
            raise _Expired(what)
        print(f"[p_companion_amd] rank {rank}: {what} did not return within {seconds:.0f} s: a peer has most likely failed "
              "before joining; ending this process (exit code 75) so that the launcher ends the job", file=sys.stderr, flush=True)
        os._exit(75)
    if "err" in box:
        raise box["err"]
    return box.get("res")


last_probe = None          # what the latest make_exchange(kind='auto'/'rccl') found: {"native", "reason", "seconds"}


def make_exchange(world, rank=None, group=None, kind="auto", device=None):
    """The gradient exchange of a data-parallel replica as an ops.Exchange for the library's exchange slot
    (pc_exchange_adam, pc_joint_train_epoch_dp): issued from the step's own foreign call on the step's stream.
      'rccl'     the library's own RCCL communicator (ncclAllReduce with ncclAvg over xGMI): one rank draws the unique id,
                 torch.distributed broadcasts its 128 bytes, every rank joins (ncclCommInitRank) on its current device;
      'callback' torch.distributed.all_reduce behind a Python trampoline (what the gloo tests use; it blocks the host);
      'auto'     'rccl' when the process group's backend is nccl and the RCCL entry points resolve, else 'callback'.
    None when no collective runs (world == 1 outside the one-rank rehearsal).

    ONE communicator per step.  Whatever this returns also carries the step's OTHER collectives (`.all_to_all(send, recv)`:
    the sharded table's lookup rounds; `.all_reduce_sum_f64_(t)`: cross-replica BatchNorm sums): 'rccl' runs all of them on
    the library's communicator, which chains collectives that sit on different streams (pcompanion_hip.h "ORDER");
    'callback' runs all of them on torch's.  Two communicators with collectives in flight at once -- torch's on the loader
    stream, the library's on the step's stream -- may start them in different orders on different ranks and deadlock; hand
    the SAME object to ShardedFeatureTable(exchange=...) and to the optimizer (bench.py does).  torch's communicator is then
    used only where the device is drained: construction, capacity agreement, the barriers around a timed region.

    After construction 'rccl' is verified -- an all-reduce of a known vector and an all-to-all of known slices on two side
    streams, i.e. the step's own pattern -- under a host deadline (probe_timeout_s(): 90 s).  The outcome is agreed by a
    MIN all-reduce over the process group (itself under a deadline):
      * every rank succeeded: the native exchange;
      * a failure or a TIME-OUT on any rank (RCCL missing, an init error, a wrong probe result, a rendezvous that never
        completes on this node): every rank drops its communicator -- one whose construction is still parked in the
        rendezvous is left behind on its daemon thread, its streams never touched again -- and takes 'callback';
      * the agreement itself does not complete (a peer is gone): exit code 75, the launcher ends the job.
    `last_probe` (module attribute) records what happened for the bench line: {"native": bool, "reason": str, "seconds": float}."""
    if not collectives_on(world):
        return None
    from . import ops
    rank = dist.get_rank(group) if rank is None else rank
    backend = dist.get_backend(group)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available()
                                             else torch.device("cpu"))

    def callback():
        def fn(ptr, n, stream):
            # the slot hands over a raw device address; the replicas' flat gradient buffers are registered by address
            t = fn.tensors.get(ptr)
            if t is None or t.numel() != n:
                raise RuntimeError("callback exchange: unknown gradient buffer %#x (register it with exchange.register(t))" % ptr)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            t.mul_(1.0 / world)
        fn.tensors = {}

        def known(ptr, n):
            t = fn.tensors.get(ptr)
            if t is None or t.numel() != n:
                raise RuntimeError("callback exchange: unknown flat buffer %#x (register it with exchange.register(t))" % ptr)
            return t

        def rs(ptr, n_per, stream):
            # the sharded optimizer's first half: slice `rank` must hold the mean (the other slices are unspecified) -- gloo has
            # no reduce-scatter, so the whole buffer is reduced; NCCL's reduce_scatter_tensor where the backend offers it
            t = known(ptr, n_per * world)
            if backend == "nccl":
                own = t[rank * n_per:(rank + 1) * n_per]
                dist.reduce_scatter_tensor(own, t, op=dist.ReduceOp.SUM, group=group)
                own.mul_(1.0 / world)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                t.mul_(1.0 / world)

        def ag(ptr, n_per, stream):
            t = known(ptr, n_per * world)
            dist.all_gather([t[r * n_per:(r + 1) * n_per] for r in range(world)], t[rank * n_per:(rank + 1) * n_per].clone(), group=group)

        ex = ops.CallbackExchange(fn, kind=f"torch.distributed.all_reduce ({backend}) behind a Python trampoline",
                                  reduce_scatter=rs, all_gather=ag, rank=rank, world=world)
        ex.register = lambda t: fn.tensors.__setitem__(t.data_ptr(), t)
        tc = _TorchCollectives(group)
        ex.all_to_all, ex.all_reduce_sum_f64_ = tc.all_to_all, tc.all_reduce_sum_f64_
        ex.native = False
        return ex

    want = kind
    if kind == "auto":
        want = "rccl" if backend == "nccl" and ops.rccl_available() else "callback"
    if want == "callback":
        return callback()
    if want != "rccl":
        raise ValueError("make_exchange: kind 'auto', 'rccl' or 'callback'")
    ok, ex = 1, None
    deadline = probe_timeout_s()
    holder = {}
    import time as _time
    t_probe = _time.perf_counter()
    reason = "ok"

    def build_and_probe():
        id_t = torch.zeros(128, dtype=torch.uint8, device=dev if backend == "nccl" else "cpu")
        if rank == 0:
            id_t.copy_(torch.frombuffer(bytearray(ops.RcclExchange.unique_id()), dtype=torch.uint8))
        dist.broadcast(id_t, src=0, group=group)
        e = holder["ex"] = ops.RcclExchange(bytes(id_t.cpu().numpy().tobytes()), rank, world)
        probe = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=dev)
        send = (torch.arange(world * 256, device=dev, dtype=torch.int32) // 256 + 1000 * rank).contiguous()   # slice p: 1000 rank + p
        recv = torch.empty_like(send)
        # two SIDE streams stand in for the step's and the loader's: if a collective never completes, the streams the job
        # goes on to use (the default stream included) hold nothing of the probe's
        main_s, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        main_s.wait_stream(torch.cuda.current_stream(dev))
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(main_s):
            e.all_reduce_mean_(probe)                            # "the step's stream"
        with torch.cuda.stream(side):
            e.all_to_all(send, recv)                             # "the loader's stream": chained behind the all-reduce by the library
        with torch.cuda.stream(main_s):
            e.all_reduce_mean_(probe)                            # ... and this one behind the all-to-all
        main_s.synchronize()
        side.synchronize()
        want_recv = (1000 * (torch.arange(world * 256, device=dev, dtype=torch.int32) // 256) + rank)
        good = torch.allclose(probe, torch.full_like(probe, (world + 1) / 2.0), rtol=1e-6, atol=0) and torch.equal(recv, want_recv)
        return int(bool(good) and (world == 1 or e.stats()["chained"] >= 2))

    abandoned = False
    try:
        ok = _run_with_deadline(build_and_probe, deadline, "the native RCCL exchange's construction and probe", rank, on_expire="abandon")
        if not ok:
            reason = "probe returned wrong values"
    except _Expired:
        ok, abandoned, reason = 0, True, f"construction / probe did not complete within {deadline:.0f} s on rank {rank}"
    except Exception as e:                                   # noqa: BLE001 -- any failure means: the host-driven exchange
        import sys
        print(f"[p_companion_amd] rank {rank}: native RCCL exchange unavailable ({e}); using torch.distributed", file=sys.stderr, flush=True)
        ok, reason = 0, f"{type(e).__name__}: {e}"[:200]
    ex = holder.get("ex")

    def agree():
        flag = torch.tensor([ok], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item())

    agreed = _run_with_deadline(agree, max(deadline, 30.0), "the ranks' agreement on the exchange kind", rank)
    global last_probe
    if agreed == 1:
        ex.register = lambda t: None
        ex.native = True
        last_probe = {"native": True, "reason": "ok", "seconds": round(_time.perf_counter() - t_probe, 2)}
        return ex
    if ok and reason == "ok":
        reason = "failed on another rank"
    last_probe = {"native": False, "reason": reason, "seconds": round(_time.perf_counter() - t_probe, 2)}
    if ex is not None and not abandoned:                     # built here, failed elsewhere (or its probe failed): not left behind
        try:
            ex.close()
        except Exception:                                    # noqa: BLE001
            pass
    if kind == "rccl":
        raise RuntimeError("make_exchange(kind='rccl'): the native exchange failed its check on some rank")
    return callback()


class ShardedFeatureTable:
    """[P,D] feature table row-sharded cyclically over the ranks of `group`: product r lives on rank r % G as local row
    r // G.

    lookup_batch(batch): the per-step exchange, DEVICE-RESIDENT -- no host synchronisation, constant shapes:
      1. pc_shard_bucket: the batch's id arrays -> per-owner request lists send_ids[G][C] (fixed capacity C, unused
         slots -1) and the batch's indices over the [G][C][D] buffer the exchange will return;
      2. all_to_all of the request lists (C int32 per peer), owner-side HIP row gather, all_to_all of the rows
         (C x D fp32 per peer) -- pc_rccl_alltoall on the library's communicator when `exchange` is the native
         RcclExchange (the same communicator as the gradient all-reduce: one cross-rank launch order per step),
         torch.distributed.all_to_all_single otherwise;
      3. the fused step runs unchanged over that buffer.  The table is frozen (p_companion.py:26-29,
         synthetic_data.py:50-58): no backward exchange.
    The unique-neighbour layout already carries every distinct neighbour product once; the remaining ids (anchors,
    positives, negatives) are sent as they are -- at the catalogue sizes sharding is for (>= 10 M products) repeats among
    a batch's ~37 k of them are < 1 %.  A bucket overflow (a batch whose ids pile up on one owner beyond C) increments
    a device counter; raise_if_overflowed() reports it where the caller synchronises anyway.

    lookup(ids): general-purpose variant for arbitrary id tensors (tools, tests): de-duplicates, exact sizes, but
    reads the bucket sizes back to the host.

    HOT SET (hot_rows = H > 0; BASELINE configs[4]: Zipf-skewed negatives): the H most popular products -- `hot_ids`
    (ascending product ids; None = the ids below H: popularity rank = product id, the Zipf sampler's default) -- are
    REPLICATED on every rank: fetched once at construction through the same constant-shape exchange (build_hot_replica, a
    collective), kept as a [H, D] tensor, and copied behind every batch's exchange buffer as rows [G*C, G*C + H) of the
    table the step reads.  pc_shard_bucket_hot maps an id of the set there instead of giving it a request slot: the
    request lists lose the Zipf head (37 % of all negative draws at H = 1024 over 100 M products), the wire with them; the
    rows the batch's indices resolve to are the same bits.  `hot_served` counts the entries served from the replica.

    `gather_fn(local_table, idx_int32)` / `bucket_fn(...)`: the owner-side row gather and the bucketing -- the HIP
    kernels on the GPU; the CPU tests inject their own (test infrastructure only -- the product never falls back)."""

    def __init__(self, local_rows, num_products, rank, world, gather_fn=None, group=None, capacity=None, bucket_fn=None,
                 exchange=None, hot_rows=0, hot_ids=None):
        self.local = local_rows
        self.P, self.rank, self.world, self.group = int(num_products), rank, world, group
        # the two lookup rounds go where the step's gradient exchange goes (make_exchange: ONE communicator per step); without
        # an exchange object they are torch.distributed's all_to_all_single (the all-torch form: bench.py --exchange hook)
        self.collectives = exchange if exchange is not None else _TorchCollectives(group)
        if gather_fn is None:
            from . import ops
            gather_fn = ops.gather_rows
        if bucket_fn is None:
            from . import ops
            bucket_fn = ops.shard_bucket
        self.gather_fn, self.bucket_fn = gather_fn, bucket_fn
        self.capacity = capacity                 # rows per peer and step; None: sized from the first batch
        self._bufs = None
        self.bytes_per_peer = None
        self.hot_rows = int(hot_rows)
        if self.hot_rows < 0 or self.hot_rows > self.P:
            raise ValueError("hot_rows: between 0 and the number of products")
        self.hot_ids = None
        if hot_ids is not None:
            hot_ids = torch.as_tensor(hot_ids, dtype=torch.int32, device=local_rows.device).contiguous()
            if hot_ids.numel() != self.hot_rows or (hot_ids.numel() > 1 and bool((hot_ids[1:] <= hot_ids[:-1]).any())):
                raise ValueError("hot_ids: hot_rows distinct product ids in ascending order")
            self.hot_ids = hot_ids
        self.hot_replica = None                  # [H, D]: built by build_hot_replica() (a collective) before the first lookup
        self.hot_served = None

    @staticmethod
    def shard(full_table, rank, world):
        return full_table[rank::world].contiguous()

    @staticmethod
    def hot_ids_from_popularity(popularity, hot_rows):
        """The hot set of a popularity permutation (product id per popularity rank, what SimilarityIndexLoader(negatives='zipf',
        popularity=...) samples over): its first hot_rows entries, ascending.  None for the identity ranking (ids below hot_rows)."""
        if popularity is None:
            return None
        return torch.sort(torch.as_tensor(popularity)[:int(hot_rows)].to(torch.int32)).values

    @staticmethod
    def capacity_for(ids_per_step, world, slack=1.25):
        """Ids hash uniformly over the owners (r % G of a scattered id set): mean R / G per bucket, standard deviation
        ~sqrt(R / G); 25 % headroom is > 20 sigma at R ~ 1e5."""
        return int(ids_per_step) if world == 1 else int(slack * ids_per_step / world) + 1024

    def agree_capacity(self, capacity):
        """The per-peer request capacity is a split size of two all_to_all_single calls: it MUST be the same number on
        every rank.  Ranks may hold different graphs or batch sizes, so the value each rank derived locally is
        all-reduced (MAX) over the group once, at construction time (one tiny host-visible collective)."""
        capacity = int(capacity)
        if collectives_on(self.world) and dist.is_initialized():
            if getattr(self.collectives, "native", False) and self.local.is_cuda:
                torch.cuda.synchronize(self.local.device)        # torch's communicator only on a drained device (make_exchange)
            dev = self.local.device if dist.get_backend(self.group) != "gloo" else torch.device("cpu")
            t = torch.tensor([capacity], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            capacity = int(t.item())
        self.capacity = capacity
        return capacity

    def _buffers(self, n_ids, dev):
        if self.capacity is None:
            if getattr(self.collectives, "native", False):
                # agree_capacity is a collective on TORCH's communicator; here it would run from the loader's side stream with
                # the library's collectives possibly in flight (two communicators, two launch orders: make_exchange's docstring)
                raise ValueError("ShardedFeatureTable(exchange=<native>): pass `capacity`, or let SimilarityIndexLoader agree it "
                                 "at construction (before the first step) -- not lazily inside a lookup")
            self.agree_capacity(self.capacity_for(n_ids, self.world))
        if self._bufs is None or self._bufs["send_ids"].device != dev:
            G, C = self.world, self.capacity
            i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=dev)
            self._bufs = {"counts": i32(G), "send_ids": i32(G * C), "overflow": i32(1)}
            self.bytes_per_peer = {"request_ids": 4 * C, "rows": 4 * C * self.local.shape[1]}
        return self._bufs

    def build_hot_replica(self):
        """The replica of the hot set on this rank: the H rows fetched once through the table's own constant-shape exchange
        (bucket without a hot set, two all-to-all rounds, owner-side gather).  A COLLECTIVE: every rank calls it, at the same
        point, before the first lookup_batch (SimilarityIndexLoader does at construction)."""
        H = self.hot_rows
        if H == 0 or self.hot_replica is not None:
            return self.hot_replica
        dev = self.local.device
        ids = self.hot_ids if self.hot_ids is not None else torch.arange(H, dtype=torch.int32, device=dev)
        G = self.world
        C = H                                        # (every rank's H requests fit one owner's bucket whatever the set)
        i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
        counts, send_ids, overflow = i32(G), i32(G * C), i32(1)
        (remap,) = self.bucket_fn([(ids, None, 0)], G, C, counts, send_ids, overflow)
        req = torch.empty_like(send_ids)
        if collectives_on(G):
            self.collectives.all_to_all(send_ids, req)
        else:
            req.copy_(send_ids)
        rows_out = self.gather_fn(self.local, req)
        tab = torch.empty_like(rows_out)
        if collectives_on(G):
            self.collectives.all_to_all(rows_out, tab)
        else:
            tab = rows_out
        self.hot_replica = tab[remap.long()].contiguous()
        self.hot_served = i32(1)
        return self.hot_replica

    def hot_rows_served(self) -> int:
        """Entries served from the replica since the last call (synchronises)."""
        if self.hot_served is None:
            return 0
        n = int(self.hot_served.item())
        self.hot_served.zero_()
        return n

    def lookup_batch(self, batch):
        """Index batch over GLOBAL product ids -> (table of the rows this rank asked for, [G*C (+ H), D]; the same batch over
        that table).  Unique / compact neighbour layouts; asynchronous."""
        nbc = batch["neighbor_compact"]
        uq = "weight" in nbc
        a, p, ng = batch["anchor_idx"], batch["positive_idx"], batch["negative_idx"]
        dev = a.device
        B, K = a.numel(), ng.shape[1]
        nb_rows = nbc["nb_rows"]
        n_live = nbc.get("n_unique_dev") if uq else None                      # [1] int32 on the device: rows 0..n_unique are live
        bufs = self._buffers(2 * B + B * K + nb_rows.numel(), dev)
        G, C, H = self.world, self.capacity, self.hot_rows
        jobs = [(a, None, 0), (nb_rows, n_live, 1), (p, None, 0), (ng.reshape(-1), None, 0)]
        if H:
            if self.hot_replica is None:
                raise RuntimeError("ShardedFeatureTable(hot_rows=...): call build_hot_replica() (a collective) before the first lookup")
            outs = self.bucket_fn(jobs, G, C, bufs["counts"], bufs["send_ids"], bufs["overflow"], hot_rows=H, hot_ids=self.hot_ids,
                                  hot_served=self.hot_served)
        else:
            outs = self.bucket_fn(jobs, G, C, bufs["counts"], bufs["send_ids"], bufs["overflow"])
        req = torch.empty_like(bufs["send_ids"])
        if collectives_on(G):
            self.collectives.all_to_all(bufs["send_ids"], req)                    # equal splits: C int32 per peer
        else:
            req.copy_(bufs["send_ids"])
        rows_out = self.gather_fn(self.local, req)                                # -1 -> zero row
        # the table the step reads: the exchange buffer [G*C, D], then the hot replica [H, D]
        full = torch.empty(G * C + H, rows_out.shape[1], dtype=rows_out.dtype, device=dev) if (H or collectives_on(G)) else None
        if collectives_on(G):
            self.collectives.all_to_all(rows_out, full[:G * C])                   # C x D fp32 per peer
        elif H:
            full[:G * C].copy_(rows_out)
        if H:
            full[G * C:].copy_(self.hot_replica)
        tab = full if full is not None else rows_out
        out = {"anchor_idx": outs[0], "positive_idx": outs[2], "negative_idx": outs[3].view(B, K),
               "neighbor_compact": dict({"nb_rows": outs[1], "slot_row": nbc["slot_row"]},
                                        **({k: nbc[k] for k in ("weight", "n_unique", "n_unique_dev", "ref_off", "ref_slot", "n_real")
                                            if k in nbc} if uq else {}))}
        if "n_pad" in batch:
            out["n_pad"] = batch["n_pad"]
        return tab, out

    def overflowed(self) -> int:
        """Ids that did not fit their owner's bucket since the last call (synchronises)."""
        if self._bufs is None:
            return 0
        n = int(self._bufs["overflow"].item())
        if n:
            self._bufs["overflow"].zero_()
        return n

    def raise_if_overflowed(self):
        n = self.overflowed()
        if n:
            raise RuntimeError(f"{n} product ids did not fit the per-peer request capacity {self.capacity} of the sharded "
                               "lookup: construct ShardedFeatureTable with a larger `capacity`")

    def lookup(self, ids):
        if getattr(self.collectives, "native", False):
            raise ValueError("ShardedFeatureTable.lookup (variable splits, host read-backs) runs on torch.distributed's "
                             "communicator; with the native exchange use lookup_batch (constant shapes, the library's communicator)")
        dev = ids.device
        flat = ids.reshape(-1).to(torch.int64)
        valid = flat >= 0
        uniq, inv = torch.unique(flat[valid], return_inverse=True)          # de-duplicate before the wire
        owner = uniq % self.world
        order = torch.argsort(owner, stable=True)
        send_ids = (uniq[order] // self.world).to(torch.int32)
        send_counts = torch.bincount(owner, minlength=self.world)
        recv_counts = torch.empty_like(send_counts)
        if collectives_on(self.world):
            dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        else:
            recv_counts.copy_(send_counts)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        req = torch.empty(sum(rc), dtype=torch.int32, device=dev)
        if collectives_on(self.world):
            dist.all_to_all_single(req, send_ids, rc, sc, group=self.group)
        else:
            req.copy_(send_ids)
        rows_out = self.gather_fn(self.local, req) if req.numel() else self.local.new_zeros((0, self.local.shape[1]))
        rows_in = torch.empty(uniq.numel(), self.local.shape[1], dtype=self.local.dtype, device=dev)
        if collectives_on(self.world):
            dist.all_to_all_single(rows_in, rows_out, sc, rc, group=self.group)
        else:
            rows_in.copy_(rows_out)
        # rows_in is in `order`; position of uniq[j] in rows_in = rank of j in order
        pos = torch.empty_like(order)
        pos[order] = torch.arange(order.numel(), device=dev)
        remap = torch.full_like(flat, -1)
        remap[valid] = pos[inv]
        return rows_in, remap.to(torch.int32).reshape(ids.shape)


class TableRowExchange:
    """Data-parallel exchange of the gradients of P-Companion's two [NUM_TYPES, 64] embedding tables as ROW LISTS
    (north_star: "reduce-scatter for the sparse grads"; src/models/p_companion.py:36-43 + train.py:46-48).

    The reference's autograd materialises dense [T,64] gradients of which a batch touches few rows (20 live types at its
    own catalogue, 100 at the benchmark's; at most B (K + 3) rows).  A dense all-reduce moves 2 x T x 256 B per step --
    17.8 MB at config.py:27's T = 34800 -- where the touched rows are n x 260 B (26 KB at 100 live types).  Per step:
      1. every rank's fused step has left ascending lists of its touched rows (pc_joint_fused_touched) and the locally
         summed rows in its dense gradient tables;
      2. all_gather of the two counts (one int64[2] message), then of the row ids and of the rows, padded to the largest
         count (constant shapes per call: no variable-size collective);
      3. every rank clears its own touched rows and adds the G lists IN RANK ORDER, each scaled by 1/G: a fixed summation
         order, so all ranks hold bit-identical mean gradients (ids are distinct inside a list: no atomics contend).
    The remaining 29 k dense weights travel as one all-reduce of their flat segment.

    gather_fn(table, ids) / assign_fn(table, ids, rows) / add_fn(table, ids, rows): the HIP row movers on the GPU
    (ops.gather_rows / scatter_rows / scatter_add_rows); the CPU tests inject torch equivalents (test infrastructure only)."""

    def __init__(self, world, group=None, gather_fn=None, assign_fn=None, add_fn=None):
        self.world, self.group = int(world), group
        if gather_fn is None:
            from . import ops
            gather_fn, assign_fn, add_fn = ops.gather_rows, ops.scatter_rows, ops.scatter_add_rows
        self.gather_fn, self.assign_fn, self.add_fn = gather_fn, assign_fn, add_fn
        self.last_bytes = None

    @staticmethod
    def merge(table_grad, lists, world, assign_fn, add_fn, own_ids):
        """table_grad [T,L]: this rank's dense table gradient; lists = [(ids_r, rows_r)] in RANK ORDER; own_ids: the rows this
        rank touched (cleared first).  Leaves the mean over the ranks in table_grad."""
        if own_ids.numel():
            assign_fn(table_grad, own_ids, torch.zeros(own_ids.numel(), table_grad.shape[1], dtype=table_grad.dtype,
                                                       device=table_grad.device))
        inv = 1.0 / world
        for ids, rows in lists:
            if ids.numel():
                add_fn(table_grad, ids, rows * inv)
        return table_grad

    def __call__(self, tables, touched):
        """tables: [grad_comp [T,L], grad_query [T,L]]; touched: [ids_comp, ids_query] (int32, exact length, ascending)."""
        G = self.world
        dev = tables[0].device
        n_loc = torch.tensor([int(t.numel()) for t in touched], dtype=torch.int64, device=dev)
        counts = torch.empty(G, 2, dtype=torch.int64, device=dev)
        if collectives_on(G):
            dist.all_gather_into_tensor(counts.view(-1), n_loc, group=self.group)
        else:
            counts[0] = n_loc
        counts = counts.cpu()                                   # (the one host-visible read of the exchange)
        sent = 0
        for ti, (tab, ids) in enumerate(zip(tables, touched)):
            cap = int(counts[:, ti].max())
            if cap == 0:
                continue
            L = tab.shape[1]
            ids_pad = torch.full((cap,), -1, dtype=torch.int32, device=dev)
            rows_pad = torch.zeros(cap, L, dtype=tab.dtype, device=dev)
            n = int(ids.numel())
            if n:
                ids_pad[:n] = ids
                rows_pad[:n] = self.gather_fn(tab, ids)
            all_ids = torch.empty(G, cap, dtype=torch.int32, device=dev)
            all_rows = torch.empty(G, cap, L, dtype=tab.dtype, device=dev)
            if collectives_on(G):
                dist.all_gather_into_tensor(all_ids.view(-1), ids_pad, group=self.group)
                dist.all_gather_into_tensor(all_rows.view(-1), rows_pad.view(-1), group=self.group)
            else:
                all_ids[0], all_rows[0] = ids_pad, rows_pad
            lists = [(all_ids[r, :int(counts[r, ti])].contiguous(), all_rows[r, :int(counts[r, ti])].contiguous()) for r in range(G)]
            self.merge(tab, lists, G, self.assign_fn, self.add_fn, ids)
            sent += cap * (4 + 4 * L)
        self.last_bytes = {"row_lists_per_rank": sent, "dense_tables": sum(t.numel() * 4 for t in tables)}
        return tables


def joint_grad_hook(model, step, world, group=None):
    """The data-parallel gradient exchange of the joint step as GraphedJointStep's grad_hook: T <= 512 -- one all-reduce of
    the flat gradient buffer (both tables are 51 KB at T = 100); T > 512 -- the 29 k dense weights as one all-reduce of
    their flat segment and the two [T,64] tables as row lists (TableRowExchange)."""
    T = model.query_type_embeddings.weight.shape[0]
    if T <= 512 or not collectives_on(world):
        return lambda gflat: all_reduce_mean_(gflat, world)
    from . import ops
    ex = TableRowExchange(world, group)
    names = [k for k, _ in model._named_flat()]
    tab_names = ("query_type_embeddings.weight", "complementary_type_embeddings.weight")

    def hook(gflat):
        if step.prepared is None:
            # GraphedJointStep's eager warm-up steps (the launch-per-op path keeps no touched-row lists): the dense exchange,
            # same mean, 2 x T x 256 B more on the wire for those few steps
            return all_reduce_mean_(gflat, world)
        params = dict(model.named_parameters())
        gq, gc = params[tab_names[0]].grad, params[tab_names[1]].grad
        off, lo = 0, None                                       # the dense weights: the maximal runs of the flat buffer between tables
        for k in names + [None]:
            if k is None or k in tab_names:
                if lo is not None:
                    all_reduce_mean_(gflat[lo:off], world)       # (ops.JOINT_KEYS puts all eight Linear tensors first: one call)
                    lo = None
            elif lo is None:
                lo = off
            if k is not None:
                off += params[k].numel()
        rc, rq, nt = ops.joint_fused_touched(step.prepared.ws, step.batch_size, T, int(model.config.NUM_COMP_TYPES))
        n_c, n_q = (int(v) for v in nt.tolist())
        ex([gc, gq], [rc[:n_c], rq[:n_q]])
        step.last_exchange_bytes = ex.last_bytes
        return gflat
    return hook
