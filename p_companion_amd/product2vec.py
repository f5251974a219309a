"""Product2Vec on MI355X -- drop-in for the reference's src/models/product2vec.py.

Same constructor, method names/signatures, batch-dict keys and state_dict keys
(ffn.0.weight ... attention.out_proj.bias, incl. the BatchNorm buffers); every number is
produced by the HIP kernels of libpcompanion_hip.so through the C ABI.  torch.nn
submodules are kept ONLY as parameter containers (so checkpoints interchange and the
default initialisers / RNG consumption are the reference's); their forward() is never
called and there is no CPU fallback.

Two ways in:
  * dense tensors (reference loader compatible): forward(features[, neighbors]) builds
    the autograd graph from two custom Functions (FFN, attention) -> works with
    loss.backward() and any torch optimizer, exactly like the reference module;
  * index batches (anchor_idx / positive_idx / negative_idx / neighbor_idx over a device
    resident feature table): train_model runs the fused step pc_p2v_train_step -- gather,
    4 BatchNorm call groups, attention, loss, whole backward -- with gradients written
    straight into .grad (views of one flat buffer) and, with FusedAdam, one Adam launch.
"""
import logging
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops

import weakref

_STEP_CACHE = weakref.WeakKeyDictionary()        # Product2Vec -> (buffer addresses, tensor dicts, C structs) of train_step_indexed
_FFN_KEYS = ops.P2V_KEYS[:8]
_ATT_KEYS = ops.P2V_KEYS[8:]


class _FFNFunction(torch.autograd.Function):
    """get_initial_embedding on a [R,128] block, training mode (product2vec.py:31-46)."""

    @staticmethod
    def forward(ctx, module, x, *weights):
        params = module._tensor_dict()
        y, sv = ops.ffn_forward_train(params, x, None, x.shape[0], [0], update_running=True)
        ctx.module, ctx.sv = module, sv
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        params = ctx.module._tensor_dict()
        grads, dx = ops.ffn_backward(params, x, None, dy.contiguous(), ctx.sv, need_dx=ctx.needs_input_grad[1])
        return (None, dx) + tuple(grads[k] for k in _FFN_KEYS)


class _AttentionFunction(torch.autograd.Function):
    """apply_attention (product2vec.py:48-68): query [B,D], keys [B,N,D] -> [B,D]."""

    @staticmethod
    def forward(ctx, module, query, keys, *weights):
        ctx.dropout = module._next_dropout()                 # one fresh mask per forward; the backward regenerates it
        params = module._tensor_dict(ctx.dropout)
        out, sv = ops.attention_forward(params, query, keys)
        ctx.module, ctx.sv = module, sv
        ctx.save_for_backward(query, keys)
        return out

    @staticmethod
    def backward(ctx, dout):
        query, keys = ctx.saved_tensors
        params = ctx.module._tensor_dict(ctx.dropout)
        grads, dq, dk = ops.attention_backward(params, query, keys, dout.contiguous(), ctx.sv)
        return (None, dq, dk) + tuple(grads[k] for k in _ATT_KEYS)


class _NoEvalBackward(torch.autograd.Function):
    """Eval-mode forward (BatchNorm running statistics, no dropout) with gradients enabled.  The reference would
    differentiate through it (product2vec.py:31-46 has no mode switch in autograd); the HIP path implements the
    backward of the TRAINING graph only, so the output carries a node that raises when differentiated -- inference
    code that merely forgot torch.no_grad() keeps working, a fine-tuning / saliency pass fails loudly instead of
    silently receiving no gradient."""

    @staticmethod
    def forward(ctx, y, what, *deps):
        ctx.what = what
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        raise NotImplementedError(
            f"{ctx.what}: backward through the eval-mode forward is not implemented in the HIP path "
            "(call .train(), or wrap inference in torch.no_grad())")


def _guard_eval(y, what, deps):
    deps = [d for d in deps if isinstance(d, torch.Tensor) and d.requires_grad]
    if torch.is_grad_enabled() and deps:
        return _NoEvalBackward.apply(y, what, *deps)
    return y


class _TripletLossFunction(torch.autograd.Function):
    """The loss expression of train_model (product2vec.py:137-154) as one kernel."""

    @staticmethod
    def forward(ctx, a, p, n, margin):
        out = ops.triplet_loss(a, p, n, margin, need_grad=True)
        ctx.save_for_backward(out["da"], out["dp"], out["dn"])
        return out["loss"].reshape(())

    @staticmethod
    def backward(ctx, g):
        da, dp, dn = ctx.saved_tensors
        return da * g, dp * g, dn * g, None


class _FusedDenseLoss(torch.autograd.Function):
    """The whole loop body of train_model (product2vec.py:132-154: four FFN calls, attention, the hinge) on a DENSE
    reference batch as one fused HIP step: the batch's rows are laid end to end as a temporary table
    [anchor B | neighbours B*N | positive B | negatives B*K] addressed by identity indices, so the four calls run as
    the segments of one launch sequence and forward + backward are a single pc_p2v_train_step (every slot its own
    row, zero padding rows included, exactly as the reference computes them).  The parameter gradients are formed
    in forward and handed to autograd in backward (scaled by the incoming gradient)."""

    @staticmethod
    def forward(ctx, module, anchor, positive, negative, neighbors, *weights):
        b, k, n = anchor.shape[0], negative.shape[1], neighbors.shape[1]
        dev = anchor.device
        d = anchor.shape[1]
        table = torch.cat([anchor, neighbors.reshape(-1, d), positive, negative.reshape(-1, d)])
        key = (b, n, k, dev)
        idx = module._dense_idx.get(key)
        if idx is None:
            ar = lambda lo, cnt: torch.arange(lo, lo + cnt, dtype=torch.int32, device=dev)
            idx = (ar(0, b), ar(b + b * n, b), ar(2 * b + b * n, b * k).reshape(b, k), ar(b, b * n).reshape(b, n))
            module._dense_idx.clear()
            module._dense_idx[key] = idx
        names = [nm for nm, _ in module.named_parameters()]
        grads = {nm: torch.empty_like(w) for nm, w in zip(names, weights)}
        out = ops.p2v_train_step(module._tensor_dict(module._next_dropout()), grads, table, idx[0], idx[1], idx[2], idx[3],
                                 float(module.config.MARGIN))
        ctx.grads = [grads[nm] for nm in names]
        return out["loss"].reshape(())

    @staticmethod
    def backward(ctx, g):
        return (None, None, None, None, None) + tuple(torch._foreach_mul(ctx.grads, g))      # one multi-tensor launch


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (defaults of scripts/pretrain_product2vec.py:34) as ONE HIP
    launch over the module's flat parameter buffer.  Opt-in: any torch optimizer works too."""

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.module = module
        super().__init__(list(module.parameters()), dict(lr=lr, betas=betas, eps=eps))
        self._state_ready = False

    def _ensure(self):
        flat, gflat = self.module.flatten_parameters()
        if not self._state_ready or self.exp_avg.data_ptr() == 0 or self.exp_avg.numel() != flat.numel() \
                or self.exp_avg.device != flat.device:
            self.exp_avg = ops.alloc(flat.numel(), torch.float32, flat.device, zero=True)
            self.exp_avg_sq = ops.alloc(flat.numel(), torch.float32, flat.device, zero=True)
            self.step_count = torch.zeros(1, dtype=torch.int64, device=flat.device)
            self.scalars = torch.zeros(2, dtype=torch.float32, device=flat.device)
            self._state_ready = True
            # the step number as the HOST knows it: valid while every increment of the device counter went through step()
            # (then the update is ONE launch, pc_adam_step_at); None once a fused step advanced the counter itself
            # (fused_state()) -- the two-launch form then reads the device counter
            self._host_step = 0
            self._device_counter_only = False     # True once the device counter has been advanced behind the host's back
        return flat, gflat

    @torch.no_grad()
    def step(self, closure=None, exchange=None, shard=False):
        """exchange (ops.Exchange, optional): a data-parallel replica's gradient exchange, issued from the same foreign call as
        the update (pc_exchange_adam) -- the flat gradient buffer holds the replicas' mean afterwards.
        shard=True: the optimizer sharded over the replicas (pc_exchange_adam_plan: reduce-scatter, Adam on this rank's slice of
        the flat buffers, all-gather of the parameters); the module's flat buffers must be padded to a multiple of the world
        size (module.flatten_parameters(pad_multiple=exchange.world) before the first step)."""
        flat, gflat = self._ensure()
        g = self.param_groups[0]
        if self._host_step is not None and flat.is_cuda and torch.cuda.is_current_stream_capturing():
            # a step that is being CAPTURED (GraphedJointStep mode 'graph' records optimizer.step() inside torch.cuda.graph)
            # must not bake the host's step number into the graph as a kernel argument: every replay would run Adam with
            # the capture-time bias corrections.  The device-counter form (pc_adam_step) is replay-safe; the host stops
            # counting for good (replays advance the device counter without passing through here).
            self._host_step = None
            self._device_counter_only = True
        if self._host_step is not None:
            self._host_step += 1
            if exchange is not None:
                ops.exchange_adam(exchange, flat, gflat, self.exp_avg, self.exp_avg_sq, self.step_count, self._host_step,
                                  self.scalars, g["lr"], g["betas"], g["eps"], shard=shard)
                return
            ops.adam_step_at(flat, gflat, self.exp_avg, self.exp_avg_sq, self.step_count, self._host_step, g["lr"],
                             g["betas"], g["eps"])
            return
        if exchange is not None:
            ops.exchange_adam(exchange, flat, gflat, self.exp_avg, self.exp_avg_sq, self.step_count, 0, self.scalars, g["lr"],
                              g["betas"], g["eps"], shard=shard)
            return
        ops.adam_step(flat, gflat, self.exp_avg, self.exp_avg_sq, self.step_count, self.scalars, g["lr"],
                      g["betas"], g["eps"])

    def riding_state(self):
        """For a step that applies this optimizer's update in its own last gradient launch (Product2Vec.train_step_indexed(
        optimizer=...): pc_p2v_train_step_unique_adam): the flat buffers, the hyper-parameters and the step number t of THIS
        update.  Counts the step (the caller must not call step() for it).  None when the host does not know the step number
        (a captured graph / a fused joint step has advanced the device counter on its own): the caller then steps separately."""
        flat, gflat = self._ensure()
        if self._host_step is None or (flat.is_cuda and torch.cuda.is_current_stream_capturing()):
            return None
        self._host_step += 1
        g = self.param_groups[0]
        return {"param": flat, "grad": gflat, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step_count": self.step_count,
                "t": self._host_step, "lr": g["lr"], "betas": g["betas"], "eps": g["eps"]}

    def epoch_state(self):
        """What an epoch-in-one-call of a replica (ops.PreparedJointStep.run_epoch_dp) needs: the flat moment buffers, the device
        step counter, the scratch scalars and t_first -- the Adam step number of the epoch's first step when the host knows it,
        else 0 (the device counter decides).  Report the steps the call ran with advance()."""
        self._ensure()
        t_first = self._host_step + 1 if self._host_step is not None else 0
        return self.exp_avg, self.exp_avg_sq, self.step_count, self.scalars, t_first

    def advance(self, steps):
        if self._host_step is not None:
            self._host_step += int(steps)

    def zero_grad(self, set_to_none=False):
        _, gflat = self._ensure()
        gflat.zero_()

    def fused_state(self):
        """What a fused training step needs to apply this optimizer's update in its own last kernel
        (ops.joint_fused_step(adam=...)): per-parameter views of the flat moment buffers, the device step counter and
        the hyper-parameters.  The caller must NOT call step() for that iteration."""
        self._ensure()
        self._host_step = None                    # (the fused step advances the device counter: the host no longer knows it)
        self._device_counter_only = True
        m, v, off = {}, {}, 0
        for name, p in self.module._named_flat():
            n = p.numel()
            m[name] = self.exp_avg[off:off + n].view(p.shape)
            v[name] = self.exp_avg_sq[off:off + n].view(p.shape)
            off += n
        g = self.param_groups[0]
        return {"exp_avg": m, "exp_avg_sq": v, "step_count": self.step_count, "lr": g["lr"], "betas": g["betas"],
                "eps": g["eps"]}

    # ------------------------------------------------------------------ checkpoint layout of torch.optim.Adam
    def _slices(self):
        """[(index in param_groups[0]['params'], offset, numel, shape)] of the parameters that live in the flat buffer."""
        self._ensure()
        pos = {id(p): i for i, p in enumerate(self.param_groups[0]["params"])}
        out, off = [], 0
        for _, p in self.module._named_flat():
            out.append((pos[id(p)], off, p.numel(), tuple(p.shape)))
            off += p.numel()
        return out

    def state_dict(self):
        """The dict torch.optim.Adam.state_dict() would return for the same parameters after the same steps
        (train.py:66 saves it as 'optimizer_state_dict'): per-parameter 'step' / 'exp_avg' / 'exp_avg_sq' sliced out
        of the flat moment buffers, parameters without a gradient (the frozen product table, p_companion.py:26-29)
        hold no state, like in torch.  torch.optim.Adam(model.parameters()).load_state_dict() reads it."""
        n = len(self.param_groups[0]["params"])
        group = {"lr": self.param_groups[0]["lr"], "betas": tuple(self.param_groups[0]["betas"]),
                 "eps": self.param_groups[0]["eps"], "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(n))}
        state = {}
        steps = int(self.step_count) if self._state_ready else 0
        if steps > 0:
            for idx, off, m, shape in self._slices():
                state[idx] = {"step": torch.tensor(float(steps)),
                              "exp_avg": self.exp_avg[off:off + m].view(shape).clone(),
                              "exp_avg_sq": self.exp_avg_sq[off:off + m].view(shape).clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, state_dict):
        """Accepts torch.optim.Adam's layout (also what state_dict() above emits): a resumed run continues with the
        saved moments and step count."""
        groups = state_dict["param_groups"]
        if len(groups) != 1:
            raise ValueError("FusedAdam: one parameter group expected")
        n = len(self.param_groups[0]["params"])
        if len(groups[0]["params"]) != n:
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
        for k in ("lr", "betas", "eps"):
            if k in groups[0]:
                self.param_groups[0][k] = tuple(groups[0][k]) if k == "betas" else groups[0][k]
        if groups[0].get("weight_decay", 0) or groups[0].get("amsgrad", False) or groups[0].get("maximize", False):
            raise ValueError("FusedAdam implements torch.optim.Adam's defaults only (no weight decay / amsgrad / maximize)")
        order = {pid: i for i, pid in enumerate(groups[0]["params"])}
        state = {order[k] if k in order else k: v for k, v in state_dict["state"].items()}
        slices = self._slices()
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        steps = set()
        for idx, off, m, shape in slices:
            st = state.get(idx)
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != shape:
                raise ValueError(f"FusedAdam.load_state_dict: parameter {idx} has shape {shape}, state has "
                                 f"{tuple(st['exp_avg'].shape)}")
            self.exp_avg[off:off + m].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + m].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdam keeps ONE step count for all parameters; the state holds %s" % sorted(steps))
        loaded = steps.pop() if steps else 0
        self.step_count.fill_(loaded)
        # a prepared fused step (fused_state()) or a captured graph keeps advancing the DEVICE counter: the host's copy would
        # drift from it after the next such step, so it stays out of the picture once that has happened
        self._host_step = None if self._device_counter_only else loaded


class _FlatParamsMixin:
    """Parameters as views of one flat fp32 buffer (+ a flat gradient buffer whose views are
    the .grad tensors), so the fused step writes gradients in place and Adam is one launch."""

    _flat_keys = ()

    def _named_flat(self):
        sd = dict(self.named_parameters())
        return [(k, sd[k]) for k in self._flat_keys]

    def flatten_parameters(self, pad_multiple=None):
        """pad_multiple (sticky once given): the flat buffers' length is rounded up to a multiple of it (zeros behind the last
        parameter) -- the sharded optimizer splits them into world equal slices (ops.exchange_adam(shard=True))."""
        if pad_multiple is not None:
            self._flat_pad = max(1, int(pad_multiple))
        pad = getattr(self, "_flat_pad", 1)
        items = self._named_flat()
        flat = getattr(self, "_flat", None)
        ok = flat is not None and flat.device == items[0][1].device and flat.numel() % pad == 0
        if ok:
            off = 0
            for _, p in items:
                if p.data_ptr() != flat.data_ptr() + 4 * off or p.grad is None or \
                        p.grad.data_ptr() != self._gflat.data_ptr() + 4 * off:
                    ok = False
                    break
                off += p.numel()
        if not ok:
            dev = items[0][1].device
            n = sum(p.numel() for _, p in items)
            n = (n + pad - 1) // pad * pad
            flat = ops.alloc(n, torch.float32, dev, zero=True)
            gflat = ops.alloc(n, torch.float32, dev, zero=True)
            off = 0
            for _, p in items:
                m = p.numel()
                flat[off:off + m].copy_(p.data.reshape(-1))
                if p.grad is not None:
                    gflat[off:off + m].copy_(p.grad.reshape(-1))
                p.data = flat[off:off + m].view(p.shape)
                p.grad = gflat[off:off + m].view(p.shape)
                off += m
            self._flat, self._gflat = flat, gflat
        return self._flat, self._gflat


class Product2Vec(nn.Module, _FlatParamsMixin):
    _flat_keys = ops.P2V_KEYS

    def __init__(self, config):
        super().__init__()
        self.config = config
        if config.PRODUCT_EMB_DIM not in (128, 256) or (config.HIDDEN_SIZE, config.NUM_ATTENTION_HEADS) != (ops.H, ops.HEADS):
            raise ValueError("the gfx950 kernels are built for PRODUCT_EMB_DIM = 128 (config.py:8) or 256 (BASELINE "
                             "configs[4]), HIDDEN_SIZE = 256, NUM_ATTENTION_HEADS = 4 (config.py:10-11)")
        self.dim = int(config.PRODUCT_EMB_DIM)
        # parameter containers only -- same construction order as product2vec.py:14-29, so
        # torch.manual_seed(s) yields the reference's initial weights
        self.ffn = nn.Sequential(
            nn.Linear(config.PRODUCT_EMB_DIM, config.HIDDEN_SIZE),
            nn.BatchNorm1d(config.HIDDEN_SIZE),
            nn.Tanh(),
            nn.Linear(config.HIDDEN_SIZE, config.HIDDEN_SIZE),
            nn.Tanh(),
            nn.Linear(config.HIDDEN_SIZE, config.PRODUCT_EMB_DIM))
        self.attention = nn.MultiheadAttention(embed_dim=config.PRODUCT_EMB_DIM,
                                               num_heads=config.NUM_ATTENTION_HEADS,
                                               dropout=config.DROPOUT, batch_first=True)
        self.last_embedding_table = None
        self._dropout_seed, self._dropout_step = None, 0
        self._dense_idx = {}                      # identity index arrays of the fused dense-batch step, by (B, N, K)

    # ------------------------------------------------------------------ plumbing
    def _tensor_dict(self, dropout=None):
        d = dict(self.named_parameters())
        d.update(dict(self.named_buffers()))
        if dropout is not None:
            d[ops.DROPOUT_KEY] = dropout
        return d

    def _next_dropout(self):
        """(p, seed, offset) of the attention-weight dropout of this training-mode forward
        (nn.MultiheadAttention(dropout=config.DROPOUT), product2vec.py:23-28), None when it is off.  The masks are the
        build's own counter-based stream (include/pcompanion_hip.h pc_dropout) -- ATen's cannot be reproduced, so a
        run with DROPOUT > 0 matches the reference in distribution, not bit for bit; the seed is drawn once from
        torch's generator (torch.manual_seed makes runs repeatable), the offset counts forwards."""
        p = float(getattr(self.config, "DROPOUT", 0.0))
        if not self.training or p == 0.0:
            return None
        if self._dropout_seed is None:
            self._dropout_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._dropout_step += 1
        return (p, self._dropout_seed, self._dropout_step - 1)

    def _weights(self, keys):
        d = dict(self.named_parameters())
        return tuple(d[k] for k in keys)

    @staticmethod
    def _dev(t):
        if not t.is_cuda:
            raise TypeError("Product2Vec runs on the GPU only: move inputs to config.DEVICE (no CPU fallback)")
        return t.contiguous().float()

    # ------------------------------------------------------------------ reference surface
    def get_initial_embedding(self, features: torch.Tensor) -> torch.Tensor:
        """Get initial embedding through FFN (product2vec.py:31-46)."""
        if features.dim() == 1:
            return self._ffn(features.unsqueeze(0)).squeeze(0)
        elif features.dim() == 2:
            return self._ffn(features)
        elif features.dim() == 3:
            B, N, D = features.shape
            return self._ffn(features.reshape(-1, D)).reshape(B, N, -1)
        raise ValueError(f"Unexpected input dimension: {features.dim()}")

    def _ffn(self, x):
        x = self._dev(x)
        if self.training:
            if x.shape[0] == 1:
                raise ValueError(f"Expected more than 1 value per channel when training, got input size {x.shape}")
            if torch.is_grad_enabled():
                return _FFNFunction.apply(self, x, *self._weights(_FFN_KEYS))
            y, _ = ops.ffn_forward_train(self._tensor_dict(), x, None, x.shape[0], [0], True)
            return y
        with torch.no_grad():
            y = ops.ffn_forward_eval(self._tensor_dict(), x, None, x.shape[0])
        return _guard_eval(y, "Product2Vec.get_initial_embedding", [x] + list(self._weights(_FFN_KEYS)))

    def apply_attention(self, query: torch.Tensor, key_value: torch.Tensor) -> torch.Tensor:
        """Apply attention mechanism with proper reshaping (product2vec.py:48-68)."""
        if query.dim() == 1:
            query = query.unsqueeze(0).unsqueeze(0)
        elif query.dim() == 2:
            query = query.unsqueeze(1)
        if key_value.dim() == 2:
            key_value = key_value.unsqueeze(0)
        if query.size(1) != 1:
            raise ValueError("the HIP attention kernel handles one query token per sample "
                             "(the only use in product2vec.py:70-81)")
        q2 = self._dev(query[:, 0, :])
        kv = self._dev(key_value)
        if torch.is_grad_enabled() and self.training:
            out = _AttentionFunction.apply(self, q2, kv, *self._weights(_ATT_KEYS))
        else:
            with torch.no_grad():
                out, _ = ops.attention_forward(self._tensor_dict(self._next_dropout()), q2, kv)   # (dropout off in eval)
            if not self.training:
                out = _guard_eval(out, "Product2Vec.apply_attention", [q2, kv] + list(self._weights(_ATT_KEYS)))
        out = out.unsqueeze(1)
        if query.size(0) == 1 and query.size(1) == 1:
            out = out.squeeze(0).squeeze(0)
        elif query.size(1) == 1:
            out = out.squeeze(1)
        return out

    def forward(self, features: torch.Tensor, neighbors: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Forward pass through Product2Vec model (product2vec.py:70-81)."""
        embeddings = self.get_initial_embedding(features)
        if neighbors is not None and neighbors.size(0) > 0:
            neighbor_embeddings = self.get_initial_embedding(neighbors)
            embeddings = self.apply_attention(embeddings, neighbor_embeddings)
        return embeddings

    # ------------------------------------------------------------------ P11
    @torch.no_grad()
    def generate_embedding_table(self, features: torch.Tensor, cv_rowptr: np.ndarray, cv_col: np.ndarray):
        """Batched device pass of generate_all_embeddings (product2vec.py:83-111), eval mode:
        pass 1 e1 = ffn(x) for every product; pass 2, for products with co-view out-neighbours,
        attention(query = ffn(e1)  [the reference re-applies the FFN to the stored embedding,
        :105-108 -> :73], keys = ffn(neighbour features) = e1[neighbours]) over the product's
        exact neighbour list (no padding in this pass).  Products are grouped by degree so each
        group is one rectangular attention launch."""
        was_training = self.training
        self.eval()
        params = self._tensor_dict()
        x = self._dev(features)
        P = x.shape[0]
        e1 = ops.ffn_forward_eval(params, x, None, P)
        e2 = ops.ffn_forward_eval(params, e1, None, P)
        out = e1.clone()
        deg = np.diff(cv_rowptr)
        cv_col = np.asarray(cv_col)
        for d in np.unique(deg):
            if d == 0:
                continue
            nodes = np.nonzero(deg == d)[0]
            starts = cv_rowptr[nodes].astype(np.int64)
            nb = cv_col[(starts[:, None] + np.arange(d)[None, :]).reshape(-1)]
            nb_idx = torch.from_numpy(np.ascontiguousarray(nb, np.int32)).to(x.device)
            node_idx = torch.from_numpy(nodes.astype(np.int32)).to(x.device)
            keys = ops.gather_rows(e1, nb_idx).view(len(nodes), int(d), self.dim)
            query = ops.gather_rows(e2, node_idx)
            upd, _ = ops.attention_forward(params, query, keys)
            ops.scatter_rows(out, node_idx, upd)
        self.train(was_training)
        self.last_embedding_table = out
        return out

    def generate_all_embeddings(self, bpg) -> Dict[str, torch.Tensor]:
        """Generate embeddings for all products in the BPG (product2vec.py:83-111).
        Accepts the integer BPG (p_companion_amd.data.IntBPG) or a reference-style
        BehaviorProductGraph (nodes dict + edges['co_view'] set).  Returns Dict[str, Tensor[128]]
        on the CPU like the reference; the device table stays in self.last_embedding_table."""
        from .data import IntBPG
        if isinstance(bpg, IntBPG):
            ids = [f"P{i:06d}" for i in range(bpg.num_products)]
            feats = bpg.cuda(self._device())["features"]
            rowptr, col = bpg.cv_rowptr, bpg.cv_col
        else:
            ids = list(bpg.nodes.keys())
            pos = {pid: i for i, pid in enumerate(ids)}
            feats = torch.stack([bpg.nodes[p]["features"] for p in ids]).to(self._device())
            lists = [[pos[t] for t in bpg.get_neighbors(p, edge_type="co_view")] for p in ids]
            rowptr = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int64)
            col = np.array([t for l in lists for t in l], np.int32)
        table = self.generate_embedding_table(feats, np.asarray(rowptr), np.asarray(col))
        cpu = table.cpu()
        return {pid: cpu[i] for i, pid in enumerate(ids)}

    def _device(self):
        return next(self.parameters()).device

    # ------------------------------------------------------------------ P9 loop
    def train_step_indexed(self, table, batch, profile=None, sync_reduce=None, optimizer=None):
        """One loop-body iteration (product2vec.py:130-158 minus optimizer.step) on an index
        batch.  Gradients land in .grad (flat-buffer views); returns the device loss tensor.
        sync_reduce: see ops.p2v_train_step (cross-replica BatchNorm statistics for data-parallel runs).
        optimizer (a FusedAdam over this module): product2vec.py:158's optimizer.step() as well -- inside the step's last
        gradient launch where the batch's layout carries it (the device loader's unique-neighbour batches, one process), by the
        optimizer's own launch otherwise.  Either way the caller does NOT call optimizer.step() for this iteration."""
        flat, gflat = self.flatten_parameters()
        # the tensor dicts and the C structs over them are rebuilt only when a buffer they describe has moved (the parameters and
        # gradients are views of the flat buffers; the BatchNorm buffers are checked by address): a module-tree walk and ~25
        # tensor checks per step otherwise stand between a drained device and the step's first launch
        bn = self.ffn[1]
        key = (flat.data_ptr(), gflat.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
               bn.num_batches_tracked.data_ptr(), self.dim)
        cache = _STEP_CACHE.get(self)               # (kept beside the module, not in it: ctypes structs do not pickle / deepcopy)
        if cache is None or cache[0] != key:
            params = self._tensor_dict()
            grads = {k: p.grad for k, p in self.named_parameters()}
            st, dev = ops.p2v_struct(params)
            gst, _ = ops.p2v_struct(grads, with_buffers=False)
            cache = _STEP_CACHE[self] = (key, params, grads, (st, gst, dev))
        params, grads, structs = dict(cache[1]), cache[2], cache[3]
        drop = self._next_dropout()
        if drop is not None:
            params[ops.DROPOUT_KEY] = drop
        nbr = batch.get("neighbor_compact", batch.get("neighbor_idx"))      # compact rows when the loader built them
        adam = None
        if optimizer is not None:
            if not hasattr(optimizer, "riding_state") or optimizer.module is not self:
                raise TypeError("train_step_indexed(optimizer=...) takes the FusedAdam built over this module")
            if sync_reduce is None and isinstance(nbr, dict) and "weight" in nbr:
                adam = optimizer.riding_state()
        out = ops.p2v_train_step(params, grads, table, batch["anchor_idx"], batch["positive_idx"],
                                 batch["negative_idx"], nbr, float(self.config.MARGIN), profile=profile,
                                 sync_reduce=sync_reduce, adam=adam, structs=structs)
        if optimizer is not None and adam is None:
            optimizer.step()
        after = batch.get("_after_step")           # the device loader's look-ahead builder, queued behind this step's launches
        if after is not None:
            after()
        return out["loss"]

    def train_model(self, train_loader, optimizer, num_epochs=10) -> Dict[str, torch.Tensor]:
        """Train Product2Vec model and generate embeddings for all products
        (product2vec.py:113-170).  Index batches take the fused HIP step; dense reference
        batches take the autograd path.  The loss stays on the device; it is read back once
        per epoch for the log line (the reference syncs every step, :162).  With self.record_step_losses = True (or
        config.RECORD_STEP_LOSSES) the per-step losses (what the reference's progress bar averages, :161-163) are kept in self.step_losses [steps]."""
        device = self.config.DEVICE
        logger = logging.getLogger(__name__)
        self.to(device)
        bpg = train_loader.dataset.bpg
        table = None
        history = [] if (getattr(self, "record_step_losses", False) or getattr(self.config, "RECORD_STEP_LOSSES", False)) else None
        # a device index loader hands out batches that are already where the step reads them: iterated directly (the staging
        # wrapper would put a cross-stream wait in front of every step's first kernel)
        direct = bool(getattr(train_loader, "yields_device_batches", False))
        from .data import prefetch_to_device
        for epoch in range(num_epochs):
            self.train()
            total, pending = None, []
            num_batches = 0

            def fold(total, pending):
                # the epoch's loss sum, on the device: one small reduction per 256 steps instead of an add kernel per step
                part = torch.cat(pending).sum().reshape(1)
                return part if total is None else total + part
            # dense batches: next batch's PCIe copy overlaps this step
            for batch in (train_loader if direct else prefetch_to_device(train_loader, device)):
                if not direct:
                    batch = {k: v.to(device) if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
                if "anchor_idx" in batch:
                    if table is None:
                        table = bpg.cuda(device)["features"]
                    if hasattr(optimizer, "riding_state") and optimizer.module is self:
                        loss = self.train_step_indexed(table, batch, optimizer=optimizer)   # zero_grad + backward + step, fused
                    else:
                        loss = self.train_step_indexed(table, batch)                        # zero_grad + backward, fused
                        optimizer.step()
                else:
                    loss = self.dense_loss(batch)
                    optimizer.zero_grad()
                    loss.backward()
                    optimizer.step()
                    loss = loss.detach().reshape(1)
                pending.append(loss.reshape(1))
                if len(pending) == 256:
                    total, pending = fold(total, pending), []
                num_batches += 1
                if history is not None:
                    history.append(loss.detach().reshape(1).clone())
            if pending:
                total = fold(total, pending)
            if num_batches:
                logger.info(f"Epoch {epoch + 1}/{num_epochs}, Loss: {float(total) / num_batches:.4f}")
        if history is not None:
            self.step_losses = torch.cat(history).cpu() if history else torch.zeros(0)
        self.eval()
        return self.generate_all_embeddings(bpg)

    def dense_loss(self, batch):
        """product2vec.py:132-154 on a dense reference batch (anchor/positive/negative[/anchor_neighbors])."""
        nb = batch.get("anchor_neighbors")
        a, p, n = batch["anchor"], batch["positive"], batch["negative"]
        if (self.training and torch.is_grad_enabled() and nb is not None and nb.dim() == 3 and nb.shape[0] == a.shape[0]
                and nb.shape[1] > 0 and a.dim() == 2 and p.dim() == 2 and n.dim() == 3 and a.shape[0] > 1
                and not any(t.requires_grad for t in (a, p, n, nb))):
            # training on device batches: the fused step (same numbers as the four module calls below, one launch
            # sequence instead of four forward + four backward ones)
            dev = self.ffn[0].weight.device
            f = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
            return _FusedDenseLoss.apply(self, f(a), f(p), f(n), f(nb), *[w for _, w in self.named_parameters()])
        anchor_emb = self(batch["anchor"], batch.get("anchor_neighbors"))
        positive_emb = self(batch["positive"])
        negative_emb = self(batch["negative"])
        if negative_emb.dim() == 2:
            negative_emb = negative_emb.unsqueeze(1)
        return _TripletLossFunction.apply(anchor_emb.contiguous(), positive_emb.contiguous(),
                                          negative_emb.contiguous(), float(self.config.MARGIN))
