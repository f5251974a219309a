"""PCompanion on MI355X -- drop-in for the reference's src/models/p_companion.py.

Same constructor (config, pretrained_embeddings: Dict[str, Tensor]), forward(batch) ->
{'projected_embeddings', 'complementary_types', 'type_similarities'}, compute_loss(batch,
outputs), _compute_type_loss / _compute_item_loss, attributes and state_dict keys
(p_companion.py:10-119).  Numbers come from the HIP kernels; nn.Embedding / nn.Linear are
parameter containers only.  Two ways in:
  * module mode: forward/compute_loss build an autograd graph of HIP Functions (works with
    loss.backward() + any torch optimizer; gradients are dense like the reference's);
  * fused mode: train_step(batch) = pc_joint_train_step -- forward, both hinge losses and the
    whole backward with the type-hinge gradient kept sparse and the type tables updated by
    row scatter-add -- gradients written into .grad, Adam as one launch (FusedAdam).
"""
from typing import Dict

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .functional import embedding, hadamard, linear
from .item_prediction import ComplementaryItemPrediction
from .product2vec import _FlatParamsMixin
from .type_transition import ComplementaryTypeTransition


class _JointLoss(torch.autograd.Function):
    """compute_loss (p_companion.py:79-119) as one kernel; the backward hands autograd the
    dense d(loss)/d(type_similarities) its graph needs (pc_expand_type_grad)."""

    @staticmethod
    def forward(ctx, sims, proj, pos_types, neg_types, pos_items, neg_items, margin, alpha, which):
        losses, dsv, dproj = ops.joint_loss(sims, proj, pos_types, neg_types, pos_items, neg_items, margin, alpha)
        ctx.save_for_backward(dsv, dproj, pos_types, neg_types)
        ctx.num_types = sims.shape[1]
        return losses[which]

    @staticmethod
    def backward(ctx, g):
        dsv, dproj, pos_types, neg_types = ctx.saved_tensors
        dense = ops.expand_type_grad(dsv, pos_types, neg_types, ctx.num_types)
        return dense * g, dproj * g, None, None, None, None, None, None, None


class _LazyJointForward(torch.autograd.Function):
    """forward(batch) in training mode as ONE fused launch sequence (pc_joint_forward).  The reference's loop hands
    the outputs straight to compute_loss, which then takes the fused step and never differentiates through them; if
    anything else does (a caller's own loss on `projected_embeddings` / `type_similarities`), backward rebuilds the
    per-op autograd graph of the same forward and differentiates that -- same gradients, paid only when used."""

    @staticmethod
    def forward(ctx, module, query_idx, query_types, k, dropout, *weights):
        params = module._tensor_dict(dropout)
        params["product_embeddings.weight"] = module.product_embeddings.weight
        sims, topk, proj, _ = ops.joint_forward(params, query_idx, query_types, k)
        ctx.module, ctx.args = module, (query_idx, query_types, k, dropout)
        ctx.nw = len(weights)
        ctx.mark_non_differentiable(topk)
        return sims, topk, proj

    @staticmethod
    def backward(ctx, dsims, _dtopk, dproj):
        module = ctx.module
        weights = [p for p in module.parameters() if p.requires_grad]
        with torch.enable_grad():
            sims, _, proj = module._forward_graph(*ctx.args)
            outs, gouts = [], []
            for o, g in ((sims, dsims), (proj, dproj)):
                if g is not None:
                    outs.append(o); gouts.append(g.contiguous())
            grads = torch.autograd.grad(outs, weights, gouts, allow_unused=True)
        return (None, None, None, None, None) + tuple(grads)


class _FusedJointLoss(torch.autograd.Function):
    """compute_loss(batch, outputs) when `outputs` is what forward(batch) just returned in training mode: the whole
    loop body train.py:42-46 is then known, and forward + both hinges + backward run as ONE pc_joint_train_step
    (type-hinge gradient kept sparse, type-table gradients without a dense [B,T] detour) instead of autograd walking
    the per-op graph.  The parameter gradients are formed here and handed to autograd in backward."""

    @staticmethod
    def forward(ctx, module, query_idx, query_types, pos_types, neg_types, pos_items, neg_items, dropout, *weights):
        names = [k for k, p in module.named_parameters() if p.requires_grad]
        grads = {k: torch.empty_like(w) for k, w in zip(names, weights)}
        params = module._tensor_dict(dropout)
        params["product_embeddings.weight"] = module.product_embeddings.weight
        t, k = module.query_type_embeddings.weight.shape[0], int(module.config.NUM_COMP_TYPES)
        step = ops.joint_train_step
        if getattr(module, "use_fused_joint", True) and ops.joint_fused_supported(t, k, dropout[0] if dropout else 0.0):
            step = ops.joint_fused_step
        losses, _ = step(params, grads, query_idx, query_types, pos_types, neg_types, pos_items, neg_items, k,
                         float(module.config.MARGIN), float(module.config.ALPHA))
        ctx.grads = [grads[k] for k in names]
        return losses[0].reshape(())

    @staticmethod
    def backward(ctx, g):
        return (None,) * 8 + tuple(torch._foreach_mul(ctx.grads, g))      # one multi-tensor launch


class PCompanion(nn.Module, _FlatParamsMixin):
    _flat_keys = ops.JOINT_KEYS

    def __init__(self, config, pretrained_embeddings):
        super().__init__()
        self.config = config
        if config.PRODUCT_EMB_DIM not in (128, 256) or config.TYPE_EMB_DIM != ops.L:
            raise ValueError("the gfx950 kernels are built for PRODUCT_EMB_DIM = 128 (config.py:8) or 256 (BASELINE "
                             "configs[4]) and TYPE_EMB_DIM = 64 (config.py:9)")
        # PRODUCT_EMB_DIM = 256 (the reference takes every dimension from config: p_companion.py:26-43, item_prediction.py:11-20):
        # the per-op module path (pc_linear_*, pc_topk_rows, pc_hadamard_*_dim, pc_joint_loss_dim, autograd between them);
        # the fused single-pass kernels are built for 128
        self.dim = int(config.PRODUCT_EMB_DIM)
        self.use_fused_joint = self.dim == ops.D

        if isinstance(pretrained_embeddings, torch.Tensor):
            # index-mode extension: row i is product i ("P%06d" % i)
            embedding_matrix = pretrained_embeddings.detach().float()
            self.product_to_idx = _IdentityIds(embedding_matrix.shape[0])
        else:
            product_ids = list(pretrained_embeddings.keys())
            self.product_to_idx = {pid: idx for idx, pid in enumerate(product_ids)}
            embedding_matrix = torch.stack([pretrained_embeddings[pid].detach().float().cpu()
                                            for pid in product_ids])       # one stack, not a Python row loop (:20-23)

        self.product_embeddings = nn.Embedding.from_pretrained(embedding_matrix, freeze=True)
        self.type_transition = ComplementaryTypeTransition(config)
        self.item_prediction = ComplementaryItemPrediction(config)
        self.query_type_embeddings = nn.Embedding(config.NUM_TYPES, config.TYPE_EMB_DIM)
        self.complementary_type_embeddings = nn.Embedding(config.NUM_TYPES, config.TYPE_EMB_DIM)

    # ------------------------------------------------------------------ helpers
    def _query_indices(self, batch, device):
        if "query_idx" in batch:
            return batch["query_idx"].to(device=device, dtype=torch.int32).contiguous()
        return torch.tensor([self.product_to_idx[pid] for pid in batch["query_ids"]],      # KeyError as :48
                            dtype=torch.int32).to(device)

    def _tensor_dict(self, dropout=None):
        d = dict(self.named_parameters())
        if dropout is not None:
            d[ops.DROPOUT_KEY] = dropout
        return d

    def _next_dropout(self):
        """(p, seed, offset) of this training-mode forward's hidden-layer dropout (type_transition.py:13,17), None = off."""
        return self.type_transition._next_dropout() if self.training else None

    @staticmethod
    def _i32(t):
        return t.reshape(-1).to(torch.int32).contiguous()

    # ------------------------------------------------------------------ index validation
    def _validate(self, *jobs):
        """Counts ids outside their tables on the device (pc_check_indices); nothing is read back here.  jobs:
        (int32 tensor, table rows).  The reference raises at the lookup (p_companion.py:48-54); here the count is
        collected by raise_index_errors() at a point where the host synchronises anyway (train() does it per epoch)."""
        ops.check_indices([(t, hi, False) for t, hi in jobs], self._bad_counter())

    def _bad_counter(self):
        """The device counter of out-of-range ids (created on first use; every path that hands ids to a kernel -- _validate,
        train_step, GraphedJointStep's prepared step and epoch runner -- goes through here, so raise_index_errors() can
        always fire)."""
        dev = self.query_type_embeddings.weight.device
        bad = getattr(self, "_bad", None)
        if bad is None or bad.device != dev:
            bad = self._bad = torch.zeros(1, dtype=torch.int32, device=dev)
        return bad

    def index_errors(self) -> int:
        """Number of out-of-range product / type ids the device has seen since the last call (synchronises)."""
        bad = getattr(self, "_bad", None)
        if bad is None:
            return 0
        n = int(bad.item())
        if n:
            bad.zero_()
        return n

    def raise_index_errors(self):
        n = self.index_errors()
        if n:
            raise IndexError(f"{n} product / type ids of the processed batches lie outside the embedding tables "
                             f"(products: {self.product_embeddings.weight.shape[0]} rows, types: "
                             f"{self.query_type_embeddings.weight.shape[0]} rows)")

    # ------------------------------------------------------------------ reference surface
    def forward(self, batch):
        dev = self.query_type_embeddings.weight.device
        k = int(self.config.NUM_COMP_TYPES)
        query_indices = self._query_indices(batch, dev)
        query_types = self._i32(batch["query_types"].to(dev))
        self._validate((query_indices, self.product_embeddings.weight.shape[0]),
                       (query_types, self.query_type_embeddings.weight.shape[0]))
        if self.dim != ops.D:
            self._pending = None
            train = torch.is_grad_enabled() and self.training
            with torch.set_grad_enabled(train):
                sims, topk, proj = self._forward_graph(query_indices, query_types, k, self._next_dropout() if self.training else None)
            return {"projected_embeddings": proj, "complementary_types": topk.long(), "type_similarities": sims}
        if not (torch.is_grad_enabled() and self.training):
            sims, topk, proj, _ = ops.joint_forward(self._tensor_dict(self._next_dropout()), query_indices, query_types, k)
            return {"projected_embeddings": proj, "complementary_types": topk.long(), "type_similarities": sims}

        self._pending = None
        weights = [p for p in self.parameters() if p.requires_grad]
        dropout = self._next_dropout()
        similarities, top_k, projected_embeddings = _LazyJointForward.apply(self, query_indices, query_types, k, dropout, *weights)
        outputs = {"projected_embeddings": projected_embeddings, "complementary_types": top_k.long(),
                   "type_similarities": similarities}
        # remembered so that compute_loss(batch, outputs) on exactly this pair can take the fused step
        self._pending = (outputs, projected_embeddings, similarities, query_indices, query_types,
                         tuple(p._version for p in self.parameters()), dropout)
        return outputs

    def _forward_graph(self, query_indices, query_types, k, dropout=None):
        """p_companion.py:45-77 op by op through the autograd Functions (what _LazyJointForward differentiates);
        `dropout`: the (p, seed, offset) the fused forward used, so both see the same hidden-layer mask."""
        query_embeddings = embedding(self.product_embeddings.weight, query_indices)
        query_type_emb = embedding(self.query_type_embeddings.weight, query_types)
        comp_base = self.type_transition(query_type_emb, _dropout=dropout)
        similarities = linear(comp_base, self.complementary_type_embeddings.weight)       # c . E_c^T
        top_k = ops.topk_rows(similarities.detach(), k)                                     # indices: no gradient
        comp_type_embeddings = embedding(self.complementary_type_embeddings.weight, top_k)
        projected_embeddings = self.item_prediction(query_embeddings, comp_type_embeddings)
        return similarities, top_k, projected_embeddings

    def _loss(self, batch, outputs, which):
        dev = outputs["type_similarities"].device
        pos_t, neg_t = self._i32(batch["positive_types"].to(dev)), self._i32(batch["negative_types"].to(dev))
        t = outputs["type_similarities"].shape[1]
        self._validate((pos_t, t), (neg_t, t))
        return _JointLoss.apply(
            outputs["type_similarities"].contiguous(), outputs["projected_embeddings"].contiguous(), pos_t, neg_t,
            batch["positive_items"].to(dev).float().contiguous(), batch["negative_items"].to(dev).float().contiguous(),
            float(self.config.MARGIN), float(self.config.ALPHA), which)

    def compute_loss(self, batch, outputs):
        """Compute combined loss for type transition and item prediction (p_companion.py:79-93)"""
        pend = getattr(self, "_pending", None)
        self._pending = None
        if (pend is not None and outputs is pend[0] and outputs.get("projected_embeddings") is pend[1]
                and outputs.get("type_similarities") is pend[2] and self.training and torch.is_grad_enabled()
                and pend[5] == tuple(p._version for p in self.parameters())
                and not any(isinstance(batch[k], torch.Tensor) and batch[k].requires_grad
                            for k in ("positive_items", "negative_items"))):
            # the reference's loop body, recognised: forward(batch) then compute_loss(batch, its outputs), parameters
            # untouched in between -> one fused step (the autograd graph forward recorded is simply dropped)
            dev = pend[3].device
            weights = [p for p in self.parameters() if p.requires_grad]
            pt, nt = self._i32(batch["positive_types"].to(dev)), self._i32(batch["negative_types"].to(dev))
            t = self.query_type_embeddings.weight.shape[0]
            self._validate((pt, t), (nt, t))
            return _FusedJointLoss.apply(
                self, pend[3], pend[4], pt, nt,
                batch["positive_items"].to(dev).float().contiguous(), batch["negative_items"].to(dev).float().contiguous(),
                pend[6], *weights)
        return self._loss(batch, outputs, 0)

    def _compute_type_loss(self, type_similarities, positive_types, negative_types):
        b = type_similarities.shape[0]
        z = torch.zeros(b, self.dim, device=type_similarities.device)
        proj = torch.zeros(b, 1, self.dim, device=type_similarities.device)
        return _JointLoss.apply(type_similarities.contiguous(), proj, self._i32(positive_types),
                                self._i32(negative_types), z, z, float(self.config.MARGIN), 0.0, 1)

    def _compute_item_loss(self, projected_embeddings, positive_items, negative_items):
        b = projected_embeddings.shape[0]
        dev = projected_embeddings.device
        sims = torch.zeros(b, 4, device=dev)
        zi = torch.zeros(b, dtype=torch.int32, device=dev)
        return _JointLoss.apply(sims, projected_embeddings.contiguous(), zi, zi, positive_items.float().contiguous(),
                                negative_items.float().contiguous(), float(self.config.MARGIN), 1.0, 2)

    # ------------------------------------------------------------------ fused loop body
    def train_step(self, batch, optimizer=None):
        """train.py:42-46 (forward, compute_loss, zero_grad, backward) as one C-ABI call; with `optimizer` (a FusedAdam
        over this module) also train.py:48 optimizer.step(), applied by the step's last kernel.
        Returns (losses[3] = total/type/item on the device, complementary_types[B,K]).
        The fused form (pc_joint_fused_step: two launches at T <= 128) serves K <= 4 (T > 512 with dropout: the similarity row
        per sample instead of per distinct query type); anything else takes the launch-per-op sequence pc_joint_train_step
        (self.use_fused_joint = False forces it)."""
        if self.dim != ops.D:
            # PRODUCT_EMB_DIM = 256: the reference's loop body through the per-op path (train.py:42-48)
            self.flatten_parameters()
            for p in self.parameters():
                if p.grad is not None:
                    p.grad.zero_()
            outputs = self(batch)
            loss = self.compute_loss(batch, outputs)
            loss.backward()
            with torch.no_grad():
                lt = self._loss(batch, outputs, 1)
                li = self._loss(batch, outputs, 2)
            if optimizer is not None:
                optimizer.step()
            return torch.stack([loss.detach(), lt, li]), outputs["complementary_types"].to(torch.int32)
        self.flatten_parameters()
        dev = self.query_type_embeddings.weight.device
        drop = self._next_dropout()
        params = self._tensor_dict(drop)
        params["product_embeddings.weight"] = self.product_embeddings.weight
        grads = {k: p.grad for k, p in self.named_parameters() if p.grad is not None}
        qi, qt = self._query_indices(batch, dev), self._i32(batch["query_types"].to(dev))
        pt, nt = self._i32(batch["positive_types"].to(dev)), self._i32(batch["negative_types"].to(dev))
        pos = batch["positive_items"].to(dev).float().contiguous()
        neg = batch["negative_items"].to(dev).float().contiguous()
        t = self.query_type_embeddings.weight.shape[0]
        k = int(self.config.NUM_COMP_TYPES)
        if getattr(self, "use_fused_joint", True) and ops.joint_fused_supported(t, k, drop[0] if drop else 0.0):
            bad = self._bad_counter()
            adam = None
            if optimizer is not None:
                if not hasattr(optimizer, "fused_state") or optimizer.module is not self:
                    raise TypeError("train_step(optimizer=...) takes the FusedAdam built over this module")
                adam = optimizer.fused_state()
            return ops.joint_fused_step(params, grads, qi, qt, pt, nt, pos, neg, k, float(self.config.MARGIN),
                                        float(self.config.ALPHA), bad=bad, adam=adam)
        self._validate((qi, self.product_embeddings.weight.shape[0]), (qt, t), (pt, t), (nt, t))
        out = ops.joint_train_step(params, grads, qi, qt, pt, nt, pos, neg, k, float(self.config.MARGIN),
                                   float(self.config.ALPHA))
        if optimizer is not None:
            optimizer.step()
        return out

    def _named_flat(self):
        sd = dict(self.named_parameters())
        return [(k, sd[k]) for k in self._flat_keys]


class GraphedJointStep:
    """train.py:42-48 (forward, compute_loss, zero_grad, backward, optimizer.step) for a FIXED batch size with the host
    out of the loop.  The batch lives in fixed device buffers (`.static`; the loader builds straight into them:
    ComplementaryIndexLoader(..., out=step.static)); Adam's step counter and bias corrections live on the device.
    After `warmup` ordinary steps:
      mode 'direct' (default where pc_joint_fused_step serves the configuration): the fused step with every
        argument resolved once -- one foreign call per iteration (ops.PreparedJointStep).  Measured on MI355X: the
        per-node cost of a HIP-graph replay (~5 us) exceeds what a launch costs the device when the host keeps its
        queue filled, so for 3-12 kernels the direct form is the faster one;
      mode 'graph': the launch sequence recorded once as a HIP graph and replayed (the launch-per-op sequence of
        pc_joint_train_step + pc_adam_step is ~25 launches: there the replay wins).
    Data-parallel replicas pass `exchange` (an ops.Exchange: distributed.make_exchange(world) -- ncclAllReduce(ncclAvg) on the
    library's own RCCL communicator): mode 'direct' then runs the fused step WITHOUT its Adam (gradients only) and
    pc_exchange_adam (the exchange on the step's stream, then Adam) -- two foreign calls per step and no Python between the
    kernels; run_epoch is pc_joint_train_epoch_dp, the whole epoch of the replica in one call.  `grad_hook` (a Python callable
    on the flat gradient buffer, e.g. distributed.joint_grad_hook: the row-list exchange of the [T,64] tables at T > 512) is the
    host-driven form: fused step, grad_hook, optimizer.step() per iteration; no run_epoch.  'graph' is a single-process form (a
    collective would have to sit inside the captured sequence)."""

    def __init__(self, model, optimizer, batch_size, warmup=3, mode="auto", grad_hook=None, exchange=None, shard_optimizer=None):
        """shard_optimizer (with `exchange`): the optimizer sharded over the replicas (include/pcompanion_hip.h ABI 8:
        reduce-scatter of the flat gradient, Adam on this rank's 1/world of the flat buffers, all-gather of the parameters)
        instead of an all-reduce and the whole dense Adam on every rank.  None = automatic: at NUM_TYPES > 512, where the two
        [NUM_TYPES,64] tables make the flat buffer megabytes (17.8 MB at config.py:27's 34800: the full update is 125 MB of
        traffic per rank and step), when the exchange offers the pair."""
        from .product2vec import FusedAdam
        if not isinstance(optimizer, FusedAdam):
            raise TypeError("GraphedJointStep drives pc_adam_step / the fused step's Adam: pass a FusedAdam")
        self.model, self.optimizer = model, optimizer
        dev = model.query_type_embeddings.weight.device
        if dev.type != "cuda":
            raise RuntimeError("GraphedJointStep needs the model on the GPU")
        if getattr(model, "dim", ops.D) != ops.D:
            # (the fixed batch buffers, the fused step and the epoch calls are PRODUCT_EMB_DIM = 128 forms; at 256 the loop body is
            # PCompanion.train_step(batch, optimizer): the per-op kernels, oracle-checked)
            raise ValueError(f"GraphedJointStep serves PRODUCT_EMB_DIM = {ops.D}; step a dim-{model.dim} model with PCompanion.train_step")
        b = int(batch_size)
        i32 = lambda *shape: ops.alloc(int(np.prod(shape)), torch.int32, dev, zero=True).view(*shape)
        f32 = lambda: ops.alloc(b * ops.D, torch.float32, dev, zero=True).view(b, ops.D)
        self.static = {"query_idx": i32(b), "query_types": i32(b), "positive_types": i32(b, 1),
                       "negative_types": i32(b, 1), "positive_items": f32(), "negative_items": f32()}
        self.batch_size, self.warmup = b, int(warmup)
        if mode not in ("auto", "direct", "graph"):
            raise ValueError("mode: 'auto', 'direct' or 'graph'")
        p = float(getattr(model.config, "DROPOUT", 0.0))
        fused_ok = getattr(model, "use_fused_joint", True) and ops.joint_fused_supported(
            model.query_type_embeddings.weight.shape[0], int(model.config.NUM_COMP_TYPES), p)
        if mode == "direct" and not fused_ok:
            raise ValueError("mode 'direct' needs a configuration pc_joint_fused_step serves")
        self.mode = ("direct" if fused_ok else "graph") if mode == "auto" else mode
        self.grad_hook = grad_hook
        self.exchange = exchange
        if grad_hook is not None and exchange is not None:
            raise ValueError("grad_hook OR exchange")
        if (grad_hook is not None or exchange is not None) and self.mode != "direct":
            raise ValueError("grad_hook / exchange need mode 'direct' (a configuration pc_joint_fused_step serves)")
        if shard_optimizer is None:
            shard_optimizer = (exchange is not None and getattr(exchange, "rs_fn", None) is not None
                               and model.query_type_embeddings.weight.shape[0] > 512)
        if shard_optimizer and (exchange is None or getattr(exchange, "rs_fn", None) is None):
            raise ValueError("shard_optimizer needs an exchange with the reduce-scatter / all-gather pair")
        self.shard_optimizer = bool(shard_optimizer)
        if exchange is not None:
            # the slices of the sharded form are world equal parts of the flat buffers; a host-driven exchange finds the
            # buffers by address
            flat, gflat = model.flatten_parameters(pad_multiple=exchange.world if self.shard_optimizer else None)
            if hasattr(exchange, "register"):
                exchange.register(gflat)
                exchange.register(flat)
        self.graph = self.prepared = None
        self._eager_steps = 0
        self.losses = self.complementary_types = None

    def load(self, batch):
        """Copy a batch dict into the fixed buffers (a no-op for tensors that already are those buffers)."""
        for k, dst in self.static.items():
            src = batch[k]
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src.reshape(dst.shape), non_blocking=True)

    def _eager(self):
        if self.grad_hook is not None or self.exchange is not None:
            self.losses, self.complementary_types = self.model.train_step(self.static)
            if self.grad_hook is not None:
                self.grad_hook(self.model.flatten_parameters()[1])
            self.optimizer.step(exchange=self.exchange, shard=self.shard_optimizer)
            return
        self.losses, self.complementary_types = self.model.train_step(self.static, optimizer=self.optimizer)

    def _prepare(self):
        m = self.model
        m.flatten_parameters()
        params = m._tensor_dict()
        params["product_embeddings.weight"] = m.product_embeddings.weight
        grads = {k: q.grad for k, q in m.named_parameters() if q.grad is not None}
        bad = m._bad_counter()                     # (never None: the kernel clamps AND counts ids outside the tables)
        tt = m.type_transition
        p = float(getattr(m.config, "DROPOUT", 0.0))
        drop = None
        if p > 0.0:
            if tt._dropout_seed is None:
                tt._next_dropout()
            drop = (p, tt._dropout_seed)
        self.prepared = ops.PreparedJointStep(params, grads, self.static, int(m.config.NUM_COMP_TYPES), float(m.config.MARGIN),
                                              float(m.config.ALPHA), bad=bad, dropout=drop,
                                              adam=self.optimizer.fused_state() if self.grad_hook is None and self.exchange is None else None)
        self._gflat = m.flatten_parameters()[1]

    def _refresh_hyper(self):
        g = self.optimizer.param_groups[0]
        self.prepared.set_hyper(g["lr"], g["betas"], g["eps"])

    def __call__(self, batch=None):
        deferred = None
        if batch is not None:
            if batch["query_idx"].numel() != self.batch_size:
                raise ValueError("GraphedJointStep: fixed batch size %d" % self.batch_size)
            deferred = batch.get("_deferred") if isinstance(batch, dict) else None
            if deferred is not None and (self.prepared is None or batch["query_idx"].data_ptr() != self.static["query_idx"].data_ptr()):
                deferred[0].materialize(batch)           # (warm-up steps / graph mode / foreign buffers: the builder's own launch)
                deferred = None
            self.load(batch)
        if self.prepared is not None:
            tt = self.model.type_transition
            if not self.model.training:
                raise RuntimeError("GraphedJointStep: the model left training mode")
            off = tt._dropout_step
            tt._dropout_step += 1
            self._refresh_hyper()
            if deferred is not None:
                # ComplementaryIndexLoader(..., out=self.static, deferred=True): the batch is built by the step's first kernel
                batch.pop("_deferred", None)
                loader, rows_dev, step = deferred
                self.losses, self.complementary_types = self.prepared.from_pairs(rows_dev, loader._source, step, off)
            else:
                self.losses, self.complementary_types = self.prepared(off)
            if self.grad_hook is not None:                 # data-parallel: gradients only above; average, then Adam
                self.grad_hook(self._gflat)
                self.optimizer.step()
            elif self.exchange is not None:                # the same from one foreign call (pc_exchange_adam[_plan]), no host hook
                self.optimizer.step(exchange=self.exchange, shard=self.shard_optimizer)
            return self.losses, self.complementary_types
        if self.graph is None:
            if self._eager_steps < self.warmup:
                self._eager_steps += 1
                self._eager()
                return self.losses, self.complementary_types
            if self.mode == "direct":
                self._prepare()
                return self(None)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                self._eager()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = graph
        self.graph.replay()
        return self.losses, self.complementary_types


def _graphed_run_epoch(self, loader, drop_last=False, max_steps=None):
    """train.py:36-57 train_epoch over one epoch of `loader` (a ComplementaryIndexLoader built with out=self.static) as ONE
    foreign call (pc_joint_train_epoch): the host enqueues every step's launches back to back, nothing is read back per
    step.  Returns the per-step losses [steps, 3] = (loss, type, item) on the device; `.mean(0)` is the epoch's average
    (train.py:50-57).  Same values, bit for bit, as iterating the loader and calling self(batch)."""
    if self.mode != "direct" or self.grad_hook is not None:
        raise ValueError("run_epoch needs mode 'direct' without a grad_hook (a configuration pc_joint_fused_step serves; "
                         "replicas pass exchange=)")
    if not self.model.training:
        raise RuntimeError("GraphedJointStep: the model left training mode")
    if loader.batch_size != self.batch_size or loader.out is None or \
            loader.out["query_idx"].data_ptr() != self.static["query_idx"].data_ptr():
        raise ValueError("run_epoch: build the loader with batch_size=%d and out=step.static" % self.batch_size)
    if self.prepared is None:
        self.model.flatten_parameters()
        if self.exchange is None:
            self.optimizer.fused_state()
        self._prepare()
    pairs = loader.epoch_pairs()
    if max_steps is not None:
        pairs = pairs[:int(max_steps) * self.batch_size]
    if getattr(loader, "_source", None) is None:
        loader._source = (loader.features, loader.type_idx, int(loader.dataset.bpg.n_types), int(loader.seed))
    tt = self.model.type_transition
    self._refresh_hyper()
    if self.exchange is not None:
        # a replica: fused step without Adam, the exchange slot, Adam over the flat buffers -- per step, all from one call
        flat, gflat = self.model.flatten_parameters()
        m, v, step_count, scalars, t_first = self.optimizer.epoch_state()
        losses, steps = self.prepared.run_epoch_dp(pairs, loader._source, loader.step, flat, gflat, m, v, step_count, t_first,
                                                   scalars, self.exchange, drop_last=drop_last, dropout_offset=tt._dropout_step,
                                                   shard=self.shard_optimizer)
        self.optimizer.advance(steps)
    else:
        losses, steps = self.prepared.run_epoch(pairs, loader._source, loader.step, drop_last=drop_last, dropout_offset=tt._dropout_step)
    tt._dropout_step += steps
    loader.step += steps
    self.losses, self.complementary_types = (losses[-1] if steps else None), self.prepared.topk
    return losses


GraphedJointStep.run_epoch = _graphed_run_epoch


class _IdentityIds:
    """product_to_idx for an integer-id table: 'P000123' -> 123 without a 100M-entry dict."""

    def __init__(self, n):
        self.n = n

    def __getitem__(self, pid):
        i = int(pid[1:]) if isinstance(pid, str) else int(pid)
        if not 0 <= i < self.n:
            raise KeyError(pid)
        return i

    def __len__(self):
        return self.n
