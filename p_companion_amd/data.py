"""Integer-form Behavior-Product-Graph, the scalable synthetic generator and the index loaders.

Replaces, for the hot path, the reference's string-keyed Python objects:
  src/data/bpg.py                BehaviorProductGraph (Dict[str,dict] + Set[(src,tgt)], O(E) scans)
  src/data/synthetic_data.py     SyntheticDataGenerator (O(N^2) itertools.combinations)
  src/data/data_loader.py        SimilarityDataset / ComplementaryDataset / collate_fn
with CSR arrays that live on the device: neighbours are an O(deg) row of (cv_rowptr, cv_col),
a batch is a handful of int32 index arrays, and the feature gather happens inside the HIP
kernels.  Product i is the reference's "P%06d" % i (synthetic_data.py:43).
"""
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch

CATEGORIES = ("electronics", "clothing", "sports", "home", "office")     # synthetic_data.py:36


def _csr_from_pairs(src, dst, n):
    order = np.argsort(src, kind="stable")
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, src + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), dst[order].astype(np.int32)


@dataclass
class IntBPG:
    """The BPG in integer form.  All arrays host numpy; .cuda() uploads what the kernels read."""
    features: np.ndarray               # [P,128] float32   bpg.nodes[pid]['features']
    type_idx: np.ndarray               # [P] int32         index into the dataset's type_to_idx
    category: np.ndarray               # [P] int32
    cv_rowptr: np.ndarray              # [P+1] int32       co-view out-neighbours (bpg.get_neighbors)
    cv_col: np.ndarray                 # [E] int32
    similarity_pairs: np.ndarray       # [S,2] int32       (Bcv & Bpv) - Bcp, list order = dataset order
    complementary_pairs: np.ndarray    # [C,2] int32       Bcp - (Bpv | Bcv)
    n_types: int
    sim_rowptr: np.ndarray = field(default=None)   # positives of each anchor (data_loader.py:31)
    sim_col: np.ndarray = field(default=None)
    _dev: Dict[str, torch.Tensor] = field(default=None, repr=False)

    def __post_init__(self):
        if self.sim_rowptr is None:
            sp = self.similarity_pairs
            self.sim_rowptr, self.sim_col = _csr_from_pairs(sp[:, 0].astype(np.int64), sp[:, 1], self.num_products)

    @property
    def num_products(self):
        return self.features.shape[0]

    def degree(self, pids):
        return self.cv_rowptr[np.asarray(pids) + 1] - self.cv_rowptr[np.asarray(pids)]

    def get_neighbors(self, pid):
        """bpg.py:24-38 for edge_type='co_view' (directed out-neighbours), O(deg)."""
        return self.cv_col[self.cv_rowptr[pid]:self.cv_rowptr[pid + 1]]

    @classmethod
    def from_arrays(cls, z):
        """From the integer-form arrays of tests/golden/g2_bpg1000.npz (the reference's own graph)."""
        return cls(features=np.ascontiguousarray(z["features"], np.float32), type_idx=z["type_idx"].astype(np.int32),
                   category=z["category"].astype(np.int32), cv_rowptr=z["cv_rowptr"].astype(np.int32),
                   cv_col=z["cv_col"].astype(np.int32), similarity_pairs=z["similarity_pairs"].astype(np.int32),
                   complementary_pairs=z["complementary_pairs"].astype(np.int32),
                   n_types=int(z["type_idx"].max()) + 1)

    def cuda(self, device="cuda"):
        if self._dev is None:
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(device)
            self._dev = {
                "features": t(self.features, torch.float32), "type_idx": t(self.type_idx, torch.int32),
                "cv_rowptr": t(self.cv_rowptr, torch.int32), "cv_col": t(self.cv_col, torch.int32),
                "sim_pairs": t(self.similarity_pairs, torch.int32), "sim_rowptr": t(self.sim_rowptr, torch.int32),
                "sim_col": t(self.sim_col, torch.int32), "n_products": self.num_products}
        return self._dev


def generate_scaled_bpg(num_products=100_000, num_types=100, seed=0, mean_degree=16.0, degree_cap=32,
                        dim=128) -> IntBPG:
    """Scalable restatement of SyntheticDataGenerator's distributions (synthetic_data.py:11-153).

    The reference enumerates all P(P-1)/2 pairs and keeps 10% of them, so it cannot go past
    ~10k products.  Here edges are drawn per source node with the reference's conditional
    probabilities:
      * type uniform over num_types = 5 categories x num_types/5 (:35-46);
      * features N(0,1)^dim, +1.0 on dims [20c, 20c+20) of category c (:50-52);
      * co-view out-degree: the reference's node i can only point at later nodes, so its degree
        is Binomial(P-1-i, p): a uniform mixture of Poisson means from 0 to 2*mean.  Restated as
        Poisson(2*mean*(1-u)), u ~ U(0,1), capped at degree_cap; same-category targets are
        1.5x as likely (:107-108);
      * purchase-after-view | co-view 0.2; co-purchase | PAV 0.15 (x0.5 same category);
        similarity pair = co-view & PAV & not co-purchase (:110-121);
      * complementary pair = co-purchase without co-view, 0.15 (x0.5 same category) (:122-127),
        scaled to ~4.5 per product (measured 4.52 at the reference's 1k products).
    Deviation: targets are uniform over all other products rather than 'later' ones.
    """
    rng = np.random.Generator(np.random.Philox(seed))
    P = int(num_products)
    per_cat = max(num_types // 5, 1)
    type_idx = rng.integers(0, num_types, P, dtype=np.int32)
    category = np.minimum(type_idx // per_cat, 4).astype(np.int32)
    feats = rng.standard_normal((P, dim), dtype=np.float32)
    cols = category[:, None] * 20 + np.arange(20)[None, :]
    np.add.at(feats, (np.arange(P)[:, None], cols), 1.0)

    def draw_targets(src, p_same, p_diff):
        """uniform target != src, kept w.p. p_same / p_diff by category agreement (rejection)"""
        tgt = rng.integers(0, P, src.shape[0], dtype=np.int64)
        todo = np.arange(src.shape[0])
        while todo.size:
            same = category[tgt[todo]] == category[src[todo]]
            bad = (tgt[todo] == src[todo]) | (rng.random(todo.size) >= np.where(same, p_same, p_diff))
            todo = todo[bad]
            tgt[todo] = rng.integers(0, P, todo.size, dtype=np.int64)
        return tgt

    lam = 2.0 * mean_degree * (1.0 - rng.random(P))
    deg = np.minimum(rng.poisson(lam), degree_cap).astype(np.int64)
    src = np.repeat(np.arange(P, dtype=np.int64), deg)
    tgt = draw_targets(src, 1.0, 2.0 / 3.0)       # cv_prob x1.5 when same category
    key = np.unique(src * P + tgt)                      # drop repeated (src,tgt); sorted by src
    src, tgt = key // P, key % P
    perm = rng.permutation(src.shape[0])                # neighbour order within a row: arbitrary, as a set's
    order = perm[np.argsort(src[perm], kind="stable")]
    src, tgt = src[order], tgt[order]
    cv_rowptr, cv_col = _csr_from_pairs(src, tgt, P)

    same = category[src] == category[tgt]
    pav = rng.random(src.shape[0]) < 0.2
    cp = rng.random(src.shape[0]) < np.where(same, 0.075, 0.15)
    sim_mask = pav & ~cp
    sim = np.stack([src[sim_mask], tgt[sim_mask]], 1).astype(np.int32)
    sim = sim[rng.permutation(sim.shape[0])]            # list(set) order is arbitrary (:150)

    csrc = np.repeat(np.arange(P, dtype=np.int64), rng.poisson(4.5, P))
    ctgt = draw_targets(csrc, 0.5, 1.0)            # cp_prob x0.5 when same category
    ckey = np.setdiff1d(np.unique(csrc * P + ctgt), key)   # not co-viewed
    comp = np.stack([ckey // P, ckey % P], 1).astype(np.int32)
    comp = comp[rng.permutation(comp.shape[0])]
    return IntBPG(features=feats, type_idx=type_idx, category=category, cv_rowptr=cv_rowptr, cv_col=cv_col,
                  similarity_pairs=sim, complementary_pairs=comp, n_types=int(num_types))


class DeviceBPG:
    """The integer BPG with every array RESIDENT IN HBM, generated there (ops.generate_catalogue -> csrc/generator.hip):
    BASELINE configs[3]/[4] -- 10 M and 100 M products -- cannot be built as host numpy arrays (generate_scaled_bpg's
    np.unique over src * P + tgt alone needs tens of GB of host memory and minutes).  Same attribute surface as IntBPG for
    the device paths (the throughput loaders, the fused steps, bench.py); host-side consumers (the CPython parity
    sampler, the CPU baseline) take an IntBPG.

    rank / world: this process holds rows rank, rank + world, ... of the feature table (the cyclic shard of SURVEY 8e);
    the graph arrays are replicated -- they are a pure function of (seed, product id), identical on every rank."""

    def __init__(self, arrays, n_types, dim, rank=0, world=1, seed=0):
        self.arrays = arrays
        self.n_types = int(n_types)
        self.dim = int(dim)
        self.rank, self.world, self.seed = int(rank), int(world), int(seed)
        self.num_products = int(arrays["n_products"])
        self.max_degree = int(arrays["max_degree"])

    # --- the few IntBPG attributes the device paths read
    @property
    def n_similarity_pairs(self):
        return int(self.arrays["sim_pairs"].shape[0])

    def cuda(self, device="cuda"):
        a = self.arrays
        g = {k: a[k] for k in ("type_idx", "cv_rowptr", "cv_col", "sim_pairs", "sim_rowptr", "sim_col")}
        g["n_products"] = self.num_products
        if "features" in a:
            g["features"] = a["features"]
        return g

    def nbytes(self):
        return sum(v.numel() * v.element_size() for v in self.arrays.values() if torch.is_tensor(v))

    def to_host(self):
        """IntBPG copy (small catalogues only: tests, the CPU baseline)."""
        a = self.arrays
        if self.world != 1:
            raise ValueError("to_host(): generate with world = 1 (the whole feature table)")
        c = lambda k: a[k].cpu().numpy()
        per_cat = max(self.n_types // 5, 1)
        ti = c("type_idx")
        return IntBPG(features=c("features"), type_idx=ti, category=np.minimum(ti // per_cat, 4).astype(np.int32),
                      cv_rowptr=c("cv_rowptr"), cv_col=c("cv_col"), similarity_pairs=c("sim_pairs"),
                      complementary_pairs=(c("comp_pairs") if "comp_pairs" in a else np.zeros((0, 2), np.int32)),
                      n_types=self.n_types, sim_rowptr=c("sim_rowptr"), sim_col=c("sim_col"))


def generate_device_bpg(num_products=100_000, num_types=100, seed=0, mean_degree=16.0, degree_cap=32, dim=128,
                        device="cuda", rank=0, world=1, with_complementary=True) -> DeviceBPG:
    """generate_scaled_bpg's distributions, drawn on the device straight into HBM (see DeviceBPG)."""
    from . import ops
    arrays = ops.generate_catalogue(num_products, num_types, seed, mean_degree, degree_cap, dim, device, rank=rank,
                                    world=world, with_complementary=with_complementary)
    return DeviceBPG(arrays, num_types, dim, rank=rank, world=world, seed=seed)


class _LazyCount:
    """Device-computed integer read back through pinned memory; int() waits for the copy (normally long done)."""

    def __init__(self, host_tensor, event):
        self._t, self._ev, self._v = host_tensor, event, None

    def __int__(self):
        if self._v is None:
            self._ev.synchronize()
            self._v = int(self._t[0])
        return self._v

    __index__ = __int__


class SimilarityIndexLoader:
    """Index-form counterpart of DataLoader(SimilarityDataset, shuffle=True, collate_fn)
    (scripts/pretrain_product2vec.py:24-30).  Yields device index batches; the dense feature
    copies of data_loader.py:50-55 never exist.

    sampler='philox' (throughput): the epoch permutation lives on the device, negatives come
      from the HIP Philox sampler (same rejection rules as data_loader.py:33-38);
    sampler='cpython' (parity): dataset order / random.shuffle and negatives are drawn from the
      exact CPython `random` stream in the reference's consumption order, so negative indices
      are bit-identical to the reference for the same random.seed().
    """

    def __init__(self, bpg: IntBPG, batch_size: int, shuffle=True, sampler="philox", seed=0, k_neg=5,
                 drop_last=False, device="cuda", compact=True, prefetch=True, unique=True, sharded=None,
                 negatives="uniform", popularity=None, reuse_buffers=False):
        from . import ops
        self.ops = ops
        self.bpg = bpg
        self.dataset = self                      # train_model reads train_loader.dataset.bpg (product2vec.py:170)
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.sampler = sampler
        self.seed = seed
        self.k_neg = k_neg
        self.drop_last = drop_last
        self.device = device
        self.compact = compact          # carry the zero-padding rows once (pc_p2v_train_step_compact)
        # ... and every distinct neighbour product once (pc_p2v_train_step_unique): identical rows of one
        # BatchNorm call are identical all the way through the FFN
        self.unique = unique and compact and sampler == "philox" and torch.device(device).type == "cuda"
        # build batch i+1 on a side stream while the consumer trains on batch i (what the reference's
        # DataLoader workers do on the host, scripts/pretrain_product2vec.py:24-30); same batches either way
        self.prefetch = prefetch and sampler == "philox" and torch.device(device).type == "cuda"
        # batches are born on the device (the philox builders): train_model iterates this loader directly, without its staging wrapper
        self.yields_device_batches = sampler == "philox" and torch.device(device).type == "cuda"
        self.epoch = 0
        self.step = 0
        # reuse_buffers (training loops: train_model, bench.py): batches are built into a RING of preallocated buffers instead
        # of fresh tensors (ops.BatchBuffers: no allocation and, above all, no cross-stream free per step).  A batch handed out
        # stays valid until RING - depth - RING_EVERY + 1 (= 9 with the defaults: the builder of hand-out j + RING is queued at
        # hand-out j + RING - depth and waits only for the latest ring event, which may be RING_EVERY - 1 hand-outs old)
        # further batches have been requested, PROVIDED it is consumed on the stream that is current when it is handed out
        # (the ring's events are recorded there): a loop that consumes each batch before asking for the next may use it;
        # code that keeps batches (tests collecting an epoch) must not.
        self.reuse_buffers = bool(reuse_buffers)
        self._ring, self._ring_done = None, None
        # the step's row list [anchor | neighbour rows | positive | negatives] is concatenated HERE, behind the builder on its
        # stream (ops.concat_step_rows), so that the fused step needs no launch of its own in front of Linear0
        self.step_rows = True
        # the fused step queues the loader's next look-ahead builder right behind its own launches (batch["_after_step"])
        self.kick_after_step = True
        # negatives='zipf' (BASELINE configs[4]; an extension, the reference draws uniformly): P(rank) ~ 1 / rank over the
        # popularity permutation `popularity` ([P] int32 product id per rank; None: product 0 is the most popular),
        # same rejection rules, device sampler only
        if negatives not in ("uniform", "zipf"):
            raise ValueError("negatives: 'uniform' or 'zipf'")
        if negatives == "zipf" and sampler != "philox":
            raise ValueError("Zipf negatives come from the device sampler (sampler='philox')")
        self.negatives = negatives
        self._zipf = None
        if negatives == "zipf":
            thr = ops.zipf_octave_thresholds(bpg.num_products)
            self._zipf = (torch.from_numpy(thr.view(np.int32).copy()).to(device),
                          None if popularity is None else torch.as_tensor(np.ascontiguousarray(popularity, np.int32)).to(device))
        self.sharded = sharded          # distributed.ShardedFeatureTable: batches then carry their own gathered `table`
        if sharded is not None and not (compact and sampler == "philox"):
            raise ValueError("the sharded lookup consumes the compact / unique neighbour layouts of the device sampler")
        self.g = bpg.cuda(device)
        if sampler == "cpython":
            self.rng = ops.CPythonRandom(seed)
        elif sampler != "philox":
            raise ValueError("sampler must be 'philox' or 'cpython'")
        self._on_device = isinstance(bpg, DeviceBPG)          # graph generated in HBM: no host copy of anything
        if self._on_device:
            if sampler != "philox" or torch.device(device).type != "cuda":
                raise ValueError("a DeviceBPG feeds the device sampler (sampler='philox', device='cuda')")
            self._n_pairs = bpg.n_similarity_pairs
            self._deg = None
            self._deg_dev = bpg.arrays["pair_deg"]
            self._max_deg = bpg.max_degree
            if unique and bpg.num_products > 4_000_000:
                # the unique-row layout scans a per-product counter array every batch (O(P): 1.2 GB of traffic at 100 M
                # products) to merge duplicate neighbours that a catalogue this size does not have (2 % at 2 M products)
                self.unique = False
        else:
            self._n_pairs = int(bpg.similarity_pairs.shape[0])
            self._deg = bpg.degree(bpg.similarity_pairs[:, 0])
            self._max_deg = int(self._deg.max() if len(self._deg) else 0)
        if sharded is not None and sharded.capacity is None:
            # the request capacity per peer must be the same constant on every rank: sized for the largest possible batch
            # (anchor + positive + k negatives + a full neighbour list per sample, + the padding row), not for this rank's first one
            max_ids = self.batch_size * (2 + k_neg + self._max_deg) + 1
            sharded.agree_capacity(sharded.capacity_for(max_ids, sharded.world))     # MAX over the ranks: a split size of the exchange
        if sharded is not None and getattr(sharded, "hot_rows", 0):
            sharded.build_hot_replica()          # (a collective, once: the replicated hot set of the Zipf head, before the first lookup)
        if negatives == "zipf":
            # the rejection sampler needs k_neg eligible products for every anchor (not itself, not one of its positives)
            max_pos = self._max_deg if self._on_device else (int(np.diff(bpg.sim_rowptr).max()) if len(bpg.sim_rowptr) > 1 else 0)
            if max_pos + k_neg + 1 > bpg.num_products:
                raise ValueError(f"Zipf negatives: an anchor has {max_pos} positives, so fewer than k_neg = {k_neg} of the "
                                 f"{bpg.num_products} products are eligible")
            self._zipf_failed = torch.zeros(1, dtype=torch.int32, device=device)

    def __len__(self):
        n = self._n_pairs
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _epoch_plan(self, S):
        """Throughput mode: this epoch's order of the similarity pairs, drawn ON THE DEVICE (deterministic in (seed, epoch)),
        and per batch the padded neighbour count and the number of real neighbour slots -- the two integers the host needs
        to size a batch.  The plan of epoch e + 1 is launched on a side stream when epoch e starts, so that neither the
        draw (4 ms on the host at 275 k pairs: four steps' worth, with the host only one batch ahead of the device) nor its
        read-back is ever waited for.  Returns (perm [S] int32 on the device, plan [n_batches, 2] int64 on the host)."""
        dev = torch.device(self.device)
        n, B = len(self), self.batch_size

        def launch(epoch):
            if dev.type == "cuda":
                # two launches of the library's own kernels (a keyed Feistel bijection + a per-batch max / sum): no
                # torch.randperm, i.e. no ATen / rocprim sort kernels on the loader path
                order = self.ops.epoch_permutation(S, int(self.seed) * 1000003 + 17, epoch, dev) if self.shuffle else None
                st = self.ops.epoch_plan(order, self._deg_dev, S, B, n)
                if order is None:
                    order = torch.arange(S, device=dev, dtype=torch.int32)
                return order, st
            g = torch.Generator(device=dev)                      # host loader (CPU tests): plain torch on the host
            g.manual_seed(int(self.seed) * 1000003 + epoch)
            order = torch.randperm(S, device=dev, generator=g) if self.shuffle else torch.arange(S, device=dev)
            d = self._deg_dev.to(torch.int64)[order]
            d = d[:n * B] if n * B <= S else torch.nn.functional.pad(d, (0, n * B - S))
            d = d.view(n, B)
            st = torch.stack([d.max(1).values, d.sum(1)], 1)
            return order.to(torch.int32), st

        if getattr(self, "_deg_dev", None) is None:
            self._deg_dev = torch.from_numpy(np.ascontiguousarray(self._deg, np.int32)).to(dev)
        if dev.type != "cuda":
            perm, st = launch(self.epoch)
            return perm, st
        if getattr(self, "_plan_stream", None) is None:
            self._plan_stream = torch.cuda.Stream(dev)
        cur = torch.cuda.current_stream(dev)

        def prefetch(epoch):
            self._plan_stream.wait_stream(cur)                   # (_deg_dev and the allocator's blocks come from there)
            with torch.cuda.stream(self._plan_stream):
                perm, st = launch(epoch)
                # two persistent pinned buffers, alternating (a fresh pinned allocation per epoch is a slow driver call)
                bufs = getattr(self, "_plan_host", None)
                if bufs is None or bufs[0].shape[0] != n:
                    bufs = self._plan_host = [torch.empty(n, 2, dtype=torch.int64).pin_memory() for _ in range(2)]
                host = bufs[epoch & 1]
                host.copy_(st, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._plan_stream)
            return epoch, S, perm, host, ev

        nxt = getattr(self, "_next_plan", None)
        if nxt is None or nxt[0] != self.epoch or nxt[1] != S:
            nxt = prefetch(self.epoch)
        _, _, perm, host, ev = nxt
        ev.synchronize()
        cur.wait_event(ev)
        self._last_plan_ev = ev
        perm.record_stream(cur)
        if getattr(self, "_side", None) is not None:
            perm.record_stream(self._side)
        self._next_plan = prefetch(self.epoch + 1)
        return perm, host

    def __iter__(self):
        S = self._n_pairs
        if self.sampler == "cpython":
            perm = self.rng.shuffle(S) if self.shuffle else np.arange(S, dtype=np.int64)
        else:
            perm = None                                   # (philox: the order lives on the device only)
        plan = None
        if perm is None:
            perm_dev, plan = self._epoch_plan(S)
        elif torch.device(self.device).type == "cuda":
            perm_host = torch.from_numpy(perm.astype(np.int32))
            # persistent pinned buffer + non_blocking: a pageable H2D copy (or a fresh pinned allocation) makes the
            # host wait for all queued GPU work -- one pipeline bubble per epoch
            if getattr(self, "_perm_pinned", None) is None or self._perm_pinned.numel() != S:
                self._perm_pinned = torch.empty(S, dtype=torch.int32).pin_memory()
            self._perm_pinned.copy_(perm_host)
            perm_dev = torch.empty(S, dtype=torch.int32, device=self.device)
            perm_dev.copy_(self._perm_pinned, non_blocking=True)
        else:
            perm_dev = torch.from_numpy(perm.astype(np.int32)).to(self.device)
        self.epoch += 1
        ring_ok = self.reuse_buffers and self.prefetch and self.sampler == "philox" and self.compact and self.sharded is None
        base = getattr(self, "_ring_base", 0)
        # the sampler's step counter of batch i of this epoch: a function of (epoch, i) alone -- not of how many builders
        # were queued ahead when an epoch was abandoned (prefetch depth 2 or 4)
        step0 = (self.epoch - 1) * len(self)

        def make(i):
            self.step = step0 + i
            lo, hi = i * self.batch_size, min((i + 1) * self.batch_size, S)
            if plan is not None:
                n_pad, n_real_plan = int(plan[i, 0]), int(plan[i, 1])
            else:
                ids = perm[lo:hi]
                n_pad = int(self._deg[ids].max())           # collate_fn pads to the batch maximum
            nbc = out = None
            if self.sampler == "philox" and self.compact and n_pad > 0:
                n_real = n_real_plan if plan is not None else int(np.minimum(self._deg[ids], n_pad).sum())
                slot = self._ring_slot(base + i) if ring_ok else None
                out = slot.views(hi - lo, n_pad, n_real, self.unique) if slot is not None else None
                if self.unique:
                    a, p, ng, nbc = self.ops.build_similarity_batch_unique(perm_dev[lo:hi], self.g, n_pad, self.k_neg,
                                                                           self.seed, self.step, n_real, out=out)
                    # the row count is needed on the host (kernel grids): fetched through pinned memory behind the
                    # builder, read when the batch is handed out (a whole step later when prefetching)
                    host_n = slot.host_n if slot is not None else torch.empty(1, dtype=torch.int32).pin_memory()
                    host_n.copy_(nbc["n_unique"], non_blocking=True)
                    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(self.device))
                    nbc["n_unique_dev"] = nbc["n_unique"]              # [1] int32 on the device (the sharded lookup reads it there)
                    nbc["n_unique"] = _LazyCount(host_n, ev)
                    nbc["n_real"] = n_real
                else:
                    a, p, ng, nbc = self.ops.build_similarity_batch_compact(perm_dev[lo:hi], self.g, n_pad, self.k_neg,
                                                                            self.seed, self.step, n_real, out=out)
                nb = None
            elif self.sampler == "philox":
                a, p, ng, nb = self.ops.build_similarity_batch(perm_dev[lo:hi], self.g, n_pad, self.k_neg,
                                                               self.seed, self.step)
            else:
                pairs = self.bpg.similarity_pairs[ids]
                negs = self.rng.negative_samples(self.bpg.num_products, self.bpg.sim_rowptr, self.bpg.sim_col,
                                                 pairs[:, 0], self.k_neg)
                nbr = np.full((len(ids), n_pad), -1, np.int32)
                for r, a_ in enumerate(pairs[:, 0]):
                    nb_ = self.bpg.get_neighbors(a_)
                    nbr[r, :len(nb_)] = nb_
                up = lambda x: torch.from_numpy(np.ascontiguousarray(x, np.int32)).to(self.device)
                a, p, ng, nb = up(pairs[:, 0]), up(pairs[:, 1]), up(negs), (up(nbr) if n_pad else None)
            if self._zipf is not None:
                self.ops.sample_negatives_zipf(perm_dev[lo:hi], self.g, self.k_neg, self.seed, self.step, self._zipf[0],
                                               self._zipf[1], out=ng, failed=self._zipf_failed)
            if nbc is not None and "n_unique_dev" in nbc and self.step_rows and self.sharded is None:
                # (a row-sharded table renumbers the rows: its lookup hands the step a remapped batch)
                nbc["step_rows"] = self.ops.concat_step_rows(a, p, ng, nbc["nb_rows"], nbc["n_unique_dev"],
                                                             out=out["step_rows"] if out is not None else None)
            self.step += 1
            batch = {"anchor_idx": a, "positive_idx": p, "negative_idx": ng, "n_pad": n_pad}
            if nb is not None:
                batch["neighbor_idx"] = nb
                if self.compact:
                    batch["neighbor_compact"] = self.ops.compact_neighbors(nb)
            if nbc is not None:
                batch["neighbor_compact"] = nbc
            if self.sharded is not None:
                # row-sharded table: the two all-to-all rounds of this batch's rows run here, on the builder's stream,
                # one batch ahead of the step that consumes them (they overlap the previous step's kernels)
                tab, remapped = self.sharded.lookup_batch(batch)
                batch = dict(remapped, table=tab, n_pad=n_pad)
            return batch

        n = len(self)
        if not self.prefetch:
            for i in range(n):
                yield make(i)
            self._end_of_epoch_checks()
            return
        if getattr(self, "_side", None) is None:
            # (default priority: a HIGH-priority builder stream doubled the step time -- 1.80 vs 0.91 ms -- its kernels then
            # pre-empt the persistent GEMM workgroups)
            self._side = torch.cuda.Stream(self.device)
        side = self._side
        base = getattr(self, "_ring_base", 0)                 # hand-outs of earlier epochs: the ring's sequence runs across epochs
        clean = getattr(self, "_ring_clean", True)              # False: the previous epoch's iterator was abandoned mid-way
        self._ring_clean = False
        if not clean:
            self._ring_done = None
        if ring_ok and clean and getattr(self, "_last_plan_ev", None) is not None and plan is not None:
            # the builders need this epoch's permutation (plan stream), not the training stream's backlog: waiting for
            # that would drain the pipeline at every epoch boundary; the ring's own events order the buffer reuse
            side.wait_event(self._last_plan_ev)
        else:
            side.wait_stream(torch.cuda.current_stream(self.device))     # the epoch permutation was uploaded there
        def launch(i):
            with torch.cuda.stream(side):
                b = make(i)
                ev = torch.cuda.Event()
                ev.record(side)
            return b, ev
        # TWO batches ahead: the builder of batch i + 2 is queued when batch i is handed out, so that its event has completed
        # by the time the batch is asked for (the host runs about one step ahead of the device: one batch ahead, the event
        # was pending about every other time and the wait below cost the training stream a barrier packet per step)
        depth = max(1, int(getattr(self, "prefetch_depth", 4 if ring_ok else 2)))
        if ring_ok and self.RING < depth + self.RING_EVERY:
            raise ValueError(f"prefetch_depth = {depth}: the buffer ring ({self.RING} slots, one event per {self.RING_EVERY} hand-outs) "
                             f"covers at most {self.RING - self.RING_EVERY} batches in flight")
        from collections import deque
        ahead = deque(launch(j) for j in range(min(depth, n)))
        owed = []                                              # the look-ahead builder not yet queued (at most one)

        def kick():
            # queue the owed look-ahead builder.  The fused step calls this (batch["_after_step"]) right behind its own launches:
            # the builder's ~0.1 ms of host work then never stands between a drained device and the next step's first kernel --
            # a loop that steps through other code gets it at the next hand-out, as before.
            if owed:
                ahead.append(launch(owed.pop()))
        for i in range(n):
            kick()
            batch, ev = ahead.popleft()
            cur = torch.cuda.current_stream(self.device)
            if ring_ok and (base + i) % self.RING_EVERY == 0:
                # everything the training stream has queued so far -- the steps over batches < i -- precedes this event.  A
                # builder that reuses a slot waits for the LATEST such event: it covers the slot's previous batch as long as
                # RING >= depth + RING_EVERY, and still leaves the builder `depth` steps of lead.  One event record per
                # RING_EVERY steps on the training stream (each is a ~14 us hole in front of the step's first kernel)
                # instead of one per freed tensor.
                done = torch.cuda.Event()
                done.record(cur)
                self._ring_done = done
            # the builder ran a step ago: normally its event has completed, and then nothing needs to be queued (a
            # cross-stream wait costs the consuming stream a barrier packet, ~10-40 us in front of every step's first kernel)
            if not ev.query():
                # The HOST waits for the builder, not the training stream: a cross-stream wait is a barrier packet in front
                # of the step's first kernel (measured: a 67 us hole at every step boundary, scripts/dev/fixed_batch_probe.py),
                # while the host runs far ahead of the device (0.3 ms of enqueue work per ~1 ms step) and loses nothing by
                # blocking for a builder that was queued two steps ago.
                ev.synchronize()
            if not ring_ok:
                for v in batch.values():
                    for t in (v.values() if isinstance(v, dict) else [v]):
                        if torch.is_tensor(t):
                            t.record_stream(cur)        # allocated on the side stream, consumed on this one
            # the look-ahead builder is queued BEHIND the hand-out (round 6): by the step itself once its launches are out
            # (kick), else when the consumer comes back for batch i + 1 -- so the ~0.1 ms of host work of a builder launch does
            # not stand between a drained device and the first kernel of a loop's first step (profiles/r06_region_probe.txt: a
            # region's first step ran 1.03-1.14 ms against 0.83).  The builder still has depth - 1 steps of lead; its ring wait
            # is the latest event, as before.
            if i + depth < n:
                owed.append(i + depth)
                if self.kick_after_step:
                    batch["_after_step"] = kick
            yield batch
        self._ring_base = base + n
        self._ring_clean = True
        self._end_of_epoch_checks()

    RING = 16
    RING_EVERY = 4

    def _ring_slot(self, i):
        """Slot i % RING of the buffer ring, safe to overwrite: builder i runs `depth` (4) hand-outs ahead; the slot last held
        batch i - RING, whose step was queued before hand-out i - RING + 1; the latest recorded event is that of a hand-out
        >= i - depth - RING_EVERY + 1 >= i - RING + 1 -- the builder's stream waits for it."""
        if self._ring is None:
            self._ring = [self.ops.BatchBuffers(self.batch_size, self._max_deg, self.k_neg, self.device) for _ in range(self.RING)]
            self._ring_done = None
        if self._ring_done is not None and i >= self.RING:
            torch.cuda.current_stream(self.device).wait_event(self._ring_done)      # (the builder's stream: make() runs under it)
        return self._ring[i % self.RING]

    def _end_of_epoch_checks(self, wait=False):
        """Device-side error counters, read WITHOUT stalling the pipeline: at the end of an epoch the counters are copied to
        pinned memory behind the epoch's last builder; the copy of the epoch before is examined (it completed long ago).
        A sharded lookup whose request bucket overflowed trained on zero rows for the ids that did not fit, and a Zipf
        sampler that ran out of proposals completed its negatives in rank order -- both raise instead of continuing
        silently, at most one epoch late.  check_errors() (wait=True) synchronises and reports at once."""
        counters = []
        if self.sharded is not None and self.sharded._bufs is not None:
            counters.append(("overflow", self.sharded._bufs["overflow"]))
        zf = getattr(self, "_zipf_failed", None)
        if zf is not None:
            counters.append(("zipf", zf))
        pend = getattr(self, "_err_pending", None)
        self._err_pending = None
        if counters and torch.device(self.device).type == "cuda":
            host = torch.empty(len(counters), dtype=torch.int32).pin_memory()
            for i, (_, t) in enumerate(counters):
                host[i:i + 1].copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._err_pending = ([n for n, _ in counters], host, ev)
        elif counters:
            pend = ([n for n, _ in counters], torch.cat([t.reshape(1).cpu() for _, t in counters]), None)
        if wait and self._err_pending is not None:
            pend, self._err_pending = self._err_pending, None
        if pend is None:
            return
        names, host, ev = pend
        if ev is not None:
            ev.synchronize()
        for nm, v in zip(names, host.tolist()):
            if v and nm == "overflow":
                raise RuntimeError(f"{v} product ids did not fit the per-peer request capacity {self.sharded.capacity} of the "
                                   "sharded lookup (they were trained as zero rows): construct ShardedFeatureTable with a "
                                   "larger `capacity`")
            if v and nm == "zipf":
                raise RuntimeError(f"Zipf negative sampler: {v} samples ran out of proposals (their negatives were completed "
                                   "in rank order): the anchors' positives cover the head of the popularity order")

    def check_errors(self):
        """Synchronising form of the per-epoch check (end of training, tests)."""
        self._end_of_epoch_checks(wait=True)


class ComplementaryIndexDataset:
    """Index-form ComplementaryDataset (data_loader.py:90-157): labelled pairs = complementary
    (+1) and similarity (-1) pairs, shuffled, split 80/10/10 by mode; per sample
      label +1: positive_types = t(tgt), negative_types = (t(tgt)+1) % n_types,
                positive_items = feat(tgt), negative_items = randn
      label -1: positive_types = 0, negative_types = t(tgt),
                positive_items = randn, negative_items = feat(tgt)              (:148-153)
    The randn filler items are input data (drawn here from torch's generator on the device)."""

    def __init__(self, bpg: IntBPG, mode="train", seed=0, sampler="philox", rng=None):
        """sampler='philox' (throughput): each mode draws its own counter-based permutation.
        sampler='cpython' (parity): the shuffle of data_loader.py:119 is CPython's random.shuffle over the list
        [complementary pairs (+1) ... similarity pairs (-1)] (:113-116), replayed bit-exactly by the host MT19937
        restatement (pc_mt_shuffle): after random.seed(seed) the pair order, the split and therefore every integer
        field of every sample equal the reference's (tests/golden/g9_complementary.npz).  `rng`: an
        ops.CPythonRandom to draw from instead of a fresh one -- the reference builds its train and val datasets
        back to back from ONE global stream (train.py:111-112)."""
        self.bpg = bpg
        cp, sp = bpg.complementary_pairs, bpg.similarity_pairs
        pairs = np.concatenate([np.concatenate([cp, np.ones((len(cp), 1), np.int32)], 1),
                                np.concatenate([sp, -np.ones((len(sp), 1), np.int32)], 1)])
        if sampler == "cpython":
            if rng is None:
                from . import ops
                rng = ops.CPythonRandom(seed)
            pairs = pairs[rng.shuffle(len(pairs))]
        elif sampler == "philox":
            rs = np.random.Generator(np.random.Philox([seed, {"train": 0, "val": 1, "test": 2}[mode]]))
            pairs = pairs[rs.permutation(len(pairs))]       # each mode shuffles independently (:119-126)
        else:
            raise ValueError("sampler must be 'philox' or 'cpython'")
        self.sampler = sampler
        n = len(pairs)
        lo, hi = {"train": (0, int(0.8 * n)), "val": (int(0.8 * n), int(0.9 * n)), "test": (int(0.9 * n), n)}[mode]
        self.pairs = pairs[lo:hi]

    def __len__(self):
        return len(self.pairs)


class ComplementaryIndexLoader:
    """DataLoader(ComplementaryDataset, collate_fn) in index form (train.py:115-129)."""

    def __init__(self, dataset: ComplementaryIndexDataset, batch_size, shuffle=True, seed=0, device="cuda", out=None,
                 deferred=False):
        """`out`: fixed device buffers every FULL batch is built into (GraphedJointStep.static); the dict handed out
        is then the same tensors each time, valid until the next batch is requested.
        `deferred` (needs `out`): a full batch is handed out UNBUILT -- the dict carries the labelled pairs under
        "_deferred" and GraphedJointStep builds the batch inside its first kernel (pc_joint_fused_step_pairs: same values,
        one launch and a 4 MB round trip less); the tensors of the dict hold the batch AFTER the step.  Anything else that
        wants the batch first calls `materialize(batch)`."""
        self.out = out
        self.deferred = bool(deferred) and out is not None
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.seed = seed
        self.device = device
        self.epoch = 0
        g = dataset.bpg.cuda(device)
        self.features, self.type_idx = g["features"], g["type_idx"]
        self.step = 0
        bpg = dataset.bpg
        if len(bpg.type_idx) and (int(bpg.type_idx.max()) >= int(bpg.n_types) or int(bpg.type_idx.min()) < 0):
            raise IndexError(f"type ids of the graph reach {int(bpg.type_idx.max())} but n_types = {bpg.n_types}: the "
                             "batches' negative types are taken modulo n_types (data_loader.py:150)")

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def make_batch(self, rows_dev, rows_host=None):
        """One HIP launch (pc_build_complementary_batch) builds the whole batch from [B,3] device pairs."""
        from . import ops
        out = self.out if (self.out is not None and rows_dev.shape[0] == self.batch_size) else None
        if out is not None and rows_dev.is_cuda:
            # fixed buffers: one foreign call with the arguments resolved once (the host must stay ahead of a ~70 us step)
            if getattr(self, "_prepared", None) is None:
                self._prepared = ops.PreparedComplementaryBuilder(self.features, self.type_idx, self.dataset.bpg.n_types,
                                                                  self.seed, out)
                self._static_batch = dict(out)
                self._source = (self.features, self.type_idx, int(self.dataset.bpg.n_types), int(self.seed))
            if self.deferred:
                self._static_batch["_deferred"] = (self, rows_dev, self.step)
            else:
                self._prepared(rows_dev, self.step)
            self.step += 1
            self._static_batch["label"] = rows_dev[:, 2]
            return self._static_batch
        batch = ops.build_complementary_batch(rows_dev, self.features, self.type_idx, self.dataset.bpg.n_types,
                                              self.seed, self.step, out=out)
        self.step += 1
        batch["label"] = rows_dev[:, 2]
        return batch

    def materialize(self, batch):
        """Build a deferred batch now (the builder's own launch); a batch that is built already passes through."""
        d = batch.pop("_deferred", None) if isinstance(batch, dict) else None
        if d is not None:
            self._prepared(d[1], d[2])
        return batch

    def epoch_pairs(self):
        """The next epoch's labelled pairs [n,3] in batch order on the device (what __iter__ slices its batches from;
        GraphedJointStep.run_epoch hands them to pc_joint_train_epoch whole)."""
        # shuffled ON THE DEVICE (pc_shuffle_rows_i32, one launch): a host permutation of 580 k pairs takes longer than the
        # 141 steps of that epoch run, and the loop would wait for it at every epoch boundary.  Deterministic in (seed, epoch)
        # per device type.
        if getattr(self, "_pairs_all", None) is None:
            self._pairs_all = torch.from_numpy(np.ascontiguousarray(self.dataset.pairs, np.int32)).to(self.device)
        e = self.epoch
        self.epoch += 1
        if not self.shuffle:
            return self._pairs_all
        if self._pairs_all.is_cuda:
            from . import ops
            return ops.shuffle_rows_i32(self._pairs_all, (int(self.seed) + 7) * 1000003, e)     # one launch of the library's own kernel
        g = torch.Generator(device=self._pairs_all.device)           # host loader (CPU tests)
        g.manual_seed((int(self.seed) + 7) * 1000003 + e)
        order = torch.randperm(self._pairs_all.shape[0], device=self._pairs_all.device, generator=g)
        return self._pairs_all.index_select(0, order)

    def __iter__(self):
        pairs_dev = self.epoch_pairs()
        for i in range(len(self)):
            yield self.make_batch(pairs_dev[i * self.batch_size:(i + 1) * self.batch_size])


def prefetch_to_device(loader, device):
    """Iterate `loader` with batch i+1's host -> device copies running on a side stream while batch i trains
    (dense reference batches are 512 * (N + 7) bytes per triplet: 82 MB at B = 4096 -- as long on PCIe as the step
    itself).  Tensors already on the device pass through; pageable host tensors are pinned first (one host copy)
    so that the transfer is asynchronous.  String lists and other values are handed on untouched."""
    device = torch.device(device)
    if device.type != "cuda":
        for batch in loader:
            yield batch
        return
    side = torch.cuda.Stream(device=device)

    def stage(batch):
        out, ev = {}, None
        with torch.cuda.stream(side):
            for k, v in batch.items():
                if isinstance(v, torch.Tensor) and v.device != device:
                    if v.device.type == "cpu" and not v.is_pinned():
                        v = v.pin_memory()
                    v = v.to(device, non_blocking=True)
                out[k] = v
            ev = torch.cuda.Event()
            ev.record(side)
        return out, ev

    it = iter(loader)
    try:
        nxt = stage(next(it))
    except StopIteration:
        return
    while nxt is not None:
        cur, ev = nxt
        try:
            nxt = stage(next(it))
        except StopIteration:
            nxt = None
        torch.cuda.current_stream(device).wait_event(ev)
        for v in cur.values():
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(torch.cuda.current_stream(device))
        yield cur
