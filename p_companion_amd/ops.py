"""Tensor-level wrappers over the C ABI (include/pcompanion_hip.h).

torch is used here for device memory and the current stream only: every computation is a
HIP kernel of libpcompanion_hip.so.  All tensors must be CUDA(ROCm), contiguous, fp32 /
int32; anything else raises (no silent conversion on the hot path, no CPU fallback).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import D, H, HEADS, L, PC_MAX_SEG, AttnSaved, FfnSaved, JointSaved, JointTensors, P2VTensors, Segments, check

P2V_KEYS = ("ffn.0.weight", "ffn.0.bias", "ffn.1.weight", "ffn.1.bias", "ffn.3.weight", "ffn.3.bias",
            "ffn.5.weight", "ffn.5.bias", "attention.in_proj_weight", "attention.in_proj_bias",
            "attention.out_proj.weight", "attention.out_proj.bias")
P2V_FIELDS = ("w0", "b0", "gamma", "beta", "w3", "b3", "w5", "b5", "in_proj_w", "in_proj_b", "out_proj_w",
              "out_proj_b")
P2V_SHAPES = ((H, D), (H,), (H,), (H,), (H, H), (H,), (D, H), (D,), (3 * D, D), (3 * D,), (D, D), (D,))
P2V_BUFFERS = (("ffn.1.running_mean", "running_mean"), ("ffn.1.running_var", "running_var"),
               ("ffn.1.num_batches_tracked", "num_batches_tracked"))

JOINT_KEYS = ("type_transition.encoder.weight", "type_transition.encoder.bias",
              "type_transition.decoder.weight", "type_transition.decoder.bias",
              "item_prediction.type_projection.weight", "item_prediction.type_projection.bias",
              "item_prediction.item_projection.weight", "item_prediction.item_projection.bias",
              "query_type_embeddings.weight", "complementary_type_embeddings.weight")
JOINT_FIELDS = ("enc_w", "enc_b", "dec_w", "dec_b", "typ_w", "typ_b", "itm_w", "itm_b", "query_types",
                "comp_types")


def _req(t, dtype, name, shape=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name}: expected a CUDA/ROCm tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_ws_cache = {}


def _default_alloc(n, dtype, device, zero=False):
    return (torch.zeros if zero else torch.empty)(int(n), dtype=dtype, device=device)


# Every device buffer the kernels WRITE through this package's own allocations -- workspaces (slabs included), the flat
# parameter / gradient / moment buffers, the fixed batch and output buffers of the prepared steps -- comes from this hook.
# tests/test_gpu_soak.py swaps in an allocator that brackets each buffer with sentinel-filled guard bands and checks them
# after hundreds of thousands of steps; production code never touches it.
_allocator = _default_alloc


def alloc(n, dtype, device, zero=False):
    return _allocator(n, dtype, device, zero)


def workspace(nbytes, device, tag="ws"):
    """Grow-only scratch buffer per (device, stream, tag); contents never outlive a call."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, tag, _allocator)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = alloc(max(int(nbytes), 256), torch.uint8, device)
        _ws_cache[key] = buf
    return buf


def make_segments(starts, rows):
    """starts: row starts of each BatchNorm call group, e.g. [0, B, B+B*N]; rows = total."""
    if not 1 <= len(starts) <= PC_MAX_SEG:
        raise ValueError("1..4 segments")
    s = Segments()
    s.weighted_row = -1
    s.weight = 1.0
    s.nseg = len(starts)
    arr = list(starts) + [rows] * (PC_MAX_SEG + 1 - len(starts))
    for i, v in enumerate(arr):
        s.start[i] = int(v)
    for i in range(s.nseg):
        n = s.start[i + 1] - s.start[i]
        if n == 1:
            # nn.BatchNorm1d raises the same in training mode (torch/nn/functional.py _verify_batch_size)
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             f"torch.Size([1, {H}])")
    return s


def p2v_shapes(d=D):
    """Parameter shapes of product2vec.py:14-29 for PRODUCT_EMB_DIM = d (128, or 256: BASELINE configs[4]); HIDDEN_SIZE
    stays 256."""
    return ((H, d), (H,), (H,), (H,), (H, H), (H,), (d, H), (d,), (3 * d, d), (3 * d,), (d, d), (d,))


def p2v_dim(tensors):
    d = int(tensors["ffn.0.weight"].shape[1])
    if d not in (128, 256):
        raise ValueError(f"the gfx950 kernels are built for PRODUCT_EMB_DIM 128 or 256, got {d}")
    return d


def p2v_struct(tensors, with_buffers=True):
    """tensors: mapping reference-state_dict-key -> tensor (parameters or gradients)."""
    st = P2VTensors()
    dev = None
    d = p2v_dim(tensors)
    st.dim = d
    for key, field, shape in zip(P2V_KEYS, P2V_FIELDS, p2v_shapes(d)):
        t = _req(tensors[key], torch.float32, key, shape)
        dev = t.device
        setattr(st, field, t.data_ptr())
    if with_buffers:
        for key, field in P2V_BUFFERS:
            t = tensors.get(key)
            if t is not None:
                _req(t, torch.int64 if "num_batches" in key else torch.float32, key)
                setattr(st, field, t.data_ptr())
    _set_dropout(st, tensors.get(DROPOUT_KEY))
    return st, dev


DROPOUT_KEY = "__dropout__"      # optional entry of a tensor dict: (p, seed, offset) of the module's training-mode dropout


def _set_dropout(st, d):
    if d is not None and float(d[0]) > 0.0:
        if not 0.0 < float(d[0]) < 1.0:
            raise ValueError(f"dropout probability has to be between 0 and 1, but got {d[0]}")
        st.dropout.p, st.dropout.seed, st.dropout.offset = float(d[0]), int(d[1]), int(d[2])


def _new_p2v_grads(device, d=D):
    return {k: torch.empty(s, dtype=torch.float32, device=device) for k, s in zip(P2V_KEYS, p2v_shapes(d))}


# ----------------------------------------------------------------------------- P6
def ffn_forward_train(params, table, idx, rows, seg_starts, update_running=True):
    """Product2Vec.get_initial_embedding in training mode over `rows` rows made of
    len(seg_starts) BatchNorm call groups.  Returns (y[rows,D], saved)."""
    st, dev = p2v_struct(params)
    _req(table, torch.float32, "table")
    if idx is not None:
        _req(idx, torch.int32, "idx", (rows,))
    seg = make_segments(seg_starts, rows)
    y = torch.empty(rows, st.dim, dtype=torch.float32, device=dev)
    sv = {"h0": torch.empty(rows, H, dtype=torch.float32, device=dev),
          "a2": torch.empty(rows, H, dtype=torch.float32, device=dev),
          "a1": torch.empty(rows, H, dtype=torch.float32, device=dev),      # tanh(BN(h0)): written by Linear3's kernel, read by dW3
          "bn": torch.empty(4, PC_MAX_SEG, H, dtype=torch.float32, device=dev),
          "seg_starts": list(seg_starts), "rows": rows}
    nbytes = _lib.lib().pc_p2v_ffn_workspace_bytes(rows)
    ws = workspace(nbytes, dev)
    check(_lib.lib().pc_p2v_ffn_forward_train(ctypes.byref(st), _p(table), _p(idx), rows, ctypes.byref(seg),
                                              1 if update_running else 0, _p(y), ctypes.byref(_ffn_saved(sv)),
                                              _p(ws), nbytes, _stream()), "pc_p2v_ffn_forward_train")
    return y, sv


def _ffn_saved(sv):
    s = FfnSaved()
    s.h0, s.a2 = sv["h0"].data_ptr(), sv["a2"].data_ptr()
    s.a1 = sv["a1"].data_ptr() if sv.get("a1") is not None else None
    bn = sv["bn"]
    s.bn_mean, s.bn_invstd, s.bn_scale, s.bn_shift = (bn[i].data_ptr() for i in range(4))
    return s


def ffn_forward_eval(params, table, idx, rows):
    st, dev = p2v_struct(params)
    _req(table, torch.float32, "table")
    if idx is not None:
        _req(idx, torch.int32, "idx", (rows,))
    y = torch.empty(rows, st.dim, dtype=torch.float32, device=dev)
    nbytes = _lib.lib().pc_p2v_ffn_workspace_bytes(rows)
    ws = workspace(nbytes, dev)
    check(_lib.lib().pc_p2v_ffn_forward_eval(ctypes.byref(st), _p(table), _p(idx), rows, _p(y), _p(ws), nbytes,
                                             _stream()), "pc_p2v_ffn_forward_eval")
    return y


def ffn_backward(params, table, idx, dy, sv, need_dx=False, grads=None, accumulate=False):
    st, dev = p2v_struct(params)
    rows = sv["rows"]
    d = st.dim
    _req(dy, torch.float32, "dy", (rows, d))
    if grads is None:
        grads = _new_p2v_grads(dev, d)
        accumulate = False
    gst, _ = p2v_struct(grads, with_buffers=False)
    seg = make_segments(sv["seg_starts"], rows)
    dx = torch.empty(rows, d, dtype=torch.float32, device=dev) if need_dx else None
    nbytes = _lib.lib().pc_p2v_ffn_workspace_bytes(rows)
    ws = workspace(nbytes, dev)
    check(_lib.lib().pc_p2v_ffn_backward(ctypes.byref(st), ctypes.byref(gst), _p(table), _p(idx), rows,
                                         ctypes.byref(seg), _p(dy), ctypes.byref(_ffn_saved(sv)), _p(dx),
                                         1 if accumulate else 0, _p(ws), nbytes, _stream()), "pc_p2v_ffn_backward")
    return grads, dx


# ----------------------------------------------------------------------------- P7
def _attn_saved(sv):
    s = AttnSaved()
    s.q, s.qt, s.probs, s.c, s.sp, s.ctx = (sv[k].data_ptr() for k in ("q", "qt", "probs", "c", "sp", "ctx"))
    return s


def attention_forward(params, query, keys):
    """query [B,D], keys [B,N,D] -> out [B,D], saved."""
    st, dev = p2v_struct(params)
    b, n, _ = keys.shape
    d = st.dim
    _req(query, torch.float32, "query", (b, d))
    _req(keys, torch.float32, "keys", (b, n, d))
    out = torch.empty(b, d, dtype=torch.float32, device=dev)
    sv = {"q": torch.empty(b, d, dtype=torch.float32, device=dev),
          "qt": torch.empty(b, HEADS, d, dtype=torch.float32, device=dev),      # Wk_h^T q_h: the key projection, absorbed
          "c": torch.empty(b, HEADS, d, dtype=torch.float32, device=dev),       # sum_n pm_n key_n per head (value projection follows)
          "sp": torch.empty(b, HEADS, dtype=torch.float32, device=dev),
          "probs": torch.empty(b, HEADS, n, dtype=torch.float32, device=dev),
          "ctx": torch.empty(b, d, dtype=torch.float32, device=dev)}
    nbytes = _lib.lib().pc_p2v_attention_workspace_bytes_dim(b, n, d)
    ws = workspace(nbytes, dev)
    check(_lib.lib().pc_p2v_attention_forward(ctypes.byref(st), _p(query), _p(keys), b, n, _p(out),
                                              ctypes.byref(_attn_saved(sv)), _p(ws), nbytes, _stream()),
          "pc_p2v_attention_forward")
    return out, sv


def attention_backward(params, query, keys, dout, sv, grads=None, accumulate=False):
    st, dev = p2v_struct(params)
    b, n, _ = keys.shape
    d = st.dim
    _req(dout, torch.float32, "dout", (b, d))
    if grads is None:
        grads = _new_p2v_grads(dev, d)
        accumulate = False
    gst, _ = p2v_struct(grads, with_buffers=False)
    dq = torch.empty(b, d, dtype=torch.float32, device=dev)
    dk = torch.empty(b, n, d, dtype=torch.float32, device=dev)
    nbytes = _lib.lib().pc_p2v_attention_workspace_bytes_dim(b, n, d)
    ws = workspace(nbytes, dev)
    check(_lib.lib().pc_p2v_attention_backward(ctypes.byref(st), ctypes.byref(gst), _p(query), _p(keys), b, n,
                                               _p(dout), ctypes.byref(_attn_saved(sv)), _p(dq), _p(dk),
                                               1 if accumulate else 0, _p(ws), nbytes, _stream()),
          "pc_p2v_attention_backward")
    return grads, dq, dk


# ----------------------------------------------------------------------------- P9 / P10
def triplet_loss(a, p, n, margin, need_grad=True):
    """a,p [B,D]; n [B,K,D].  Returns dict(loss[1], d_pos[B], d_neg[B], da, dp, dn)."""
    b, k, d = n.shape
    if d not in (128, 256):
        raise ValueError(f"embedding width {d}: the gfx950 kernels serve 128 and 256")
    _req(a, torch.float32, "anchor_emb", (b, d)); _req(p, torch.float32, "positive_emb", (b, d))
    _req(n, torch.float32, "negative_emb", (b, k, d))
    dev = a.device
    out = {"loss": torch.empty(1, dtype=torch.float32, device=dev),
           "d_pos": torch.empty(b, dtype=torch.float32, device=dev),
           "d_neg": torch.empty(b, dtype=torch.float32, device=dev)}
    if need_grad:
        out.update(da=torch.empty_like(a), dp=torch.empty_like(p), dn=torch.empty_like(n))
    check(_lib.lib().pc_p2v_triplet_loss_dim(_p(a), _p(p), _p(n), b, k, d, float(margin), _p(out["loss"]),
                                             _p(out["d_pos"]), _p(out["d_neg"]), _p(out.get("da")), _p(out.get("dp")),
                                             _p(out.get("dn")), _stream()), "pc_p2v_triplet_loss_dim")
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, step_count, scalars, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
    n = param.numel()
    for t, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, torch.float32, nm)
        if t.numel() != n:
            raise ValueError("adam_step: size mismatch")
    _req(step_count, torch.int64, "step_count"); _req(scalars, torch.float32, "scalars", (2,))
    check(_lib.lib().pc_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, _p(step_count), _p(scalars),
                                  float(lr), float(betas[0]), float(betas[1]), float(eps), _stream()), "pc_adam_step")


def adam_step_at(param, grad, exp_avg, exp_avg_sq, step_count, t, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
    """pc_adam_step_at: the update of step number t (host-known) as one launch; step_count (device int64 [1]) is left = t."""
    n = param.numel()
    for x, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(x, torch.float32, nm)
        if x.numel() != n:
            raise ValueError("adam_step_at: size mismatch")
    _req(step_count, torch.int64, "step_count")
    check(_lib.lib().pc_adam_step_at(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, _p(step_count), int(t), float(lr),
                                     float(betas[0]), float(betas[1]), float(eps), _stream()), "pc_adam_step_at")


# ----------------------------------------------------------------------------- data-parallel exchange slot (ABI 6)
# A pc_exchange_fn travels through ctypes as a plain address (fn) with its context (ctx): the native one is the library's own
# pc_rccl_allreduce_mean over its own RCCL communicator -- no Python between a step's gradient kernels and its Adam launch;
# CallbackExchange wraps a Python callable for backends RCCL does not serve (gloo in the tests).
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


class ExchangePlan(ctypes.Structure):
    """pc_exchange_plan (ABI 8): the plain exchange (all_reduce_mean) and the two halves of the sharded optimizer's."""
    _fields_ = [("all_reduce_mean", ctypes.c_void_p), ("reduce_scatter_mean", ctypes.c_void_p), ("all_gather", ctypes.c_void_p),
                ("ctx", ctypes.c_void_p), ("rank", ctypes.c_int), ("world", ctypes.c_int), ("shard_optimizer", ctypes.c_int)]


class Exchange:
    fn = None       # ctypes.c_void_p: address of a pc_exchange_fn
    ctx = None      # ctypes.c_void_p
    rs_fn = ag_fn = None      # ctypes.c_void_p: addresses of the pc_shard_collective_fn pair (None: no sharded form)
    rank, world = 0, 1

    def plan(self, shard=False):
        """The pc_exchange_plan of this exchange (kept alive by the caller for the duration of the foreign call).
        shard=True: reduce-scatter -> Adam on this rank's slice -> all-gather (needs the sharded pair)."""
        if shard and (self.rs_fn is None or self.ag_fn is None):
            raise ValueError("this exchange has no reduce-scatter / all-gather pair")
        val = lambda f: f.value if f is not None else None
        return ExchangePlan(val(self.fn), val(self.rs_fn), val(self.ag_fn), val(self.ctx), int(self.rank), int(self.world),
                            1 if shard else 0)

    def all_reduce_mean_(self, t):
        """The exchange applied to a device tensor, in place, on the current stream (what a step's call does through the slot)."""
        _req(t, torch.float32, "grad")
        rc = EXCHANGE_FN(self.fn.value)(self.ctx, _p(t), t.numel(), _stream())
        check(rc, "pc_exchange_fn")
        return t


def rccl_available():
    return bool(_lib.lib().pc_rccl_available())


class RcclExchange(Exchange):
    """The library's own RCCL communicator (pc_rccl_*): ncclAllReduce(ncclAvg) behind the exchange slot, plus the step's
    other collectives -- all_to_all (the row-sharded table's two lookup rounds) and all_reduce_sum_f64_ (cross-replica
    BatchNorm sums) -- so that everything a step exchanges runs on ONE communicator, chained inside the library when two of
    them sit on different streams (include/pcompanion_hip.h, "ORDER").  One rank draws the unique id
    (RcclExchange.unique_id()), every rank constructs with the same 128 bytes -- a collective (ncclCommInitRank) on the
    current device."""

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        check(_lib.lib().pc_rccl_unique_id(buf), "pc_rccl_unique_id")
        return buf.raw

    def __init__(self, unique_id, rank, world):
        if len(unique_id) != 128:
            raise ValueError("RcclExchange: a 128-byte ncclUniqueId")
        L = _lib.lib()
        h = ctypes.c_void_p()
        check(L.pc_rccl_comm_create(ctypes.c_char_p(bytes(unique_id)), int(rank), int(world), ctypes.byref(h)), "pc_rccl_comm_create")
        self.ctx = h
        self.fn = ctypes.cast(L.pc_rccl_allreduce_mean, ctypes.c_void_p)
        self.rs_fn = ctypes.cast(L.pc_rccl_reduce_scatter_mean, ctypes.c_void_p)
        self.ag_fn = ctypes.cast(L.pc_rccl_all_gather, ctypes.c_void_p)
        self.rank, self.world = int(rank), int(world)
        self.kind = "rccl (library-owned communicator, ncclAvg)"

    def close(self):
        if self.ctx is not None and self.ctx.value:
            torch.cuda.synchronize()
            check(_lib.lib().pc_rccl_comm_destroy(self.ctx), "pc_rccl_comm_destroy")
            self.ctx = None

    def all_to_all(self, send, recv):
        """pc_rccl_alltoall on the current stream: equal splits, peer p's slice of `send` lands in peer p's `recv` at this
        rank's slot.  Any dtype; both contiguous device tensors of the same byte size, a multiple of the world size."""
        nbytes = send.numel() * send.element_size()
        if not (send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous()):
            raise ValueError("all_to_all: contiguous device tensors")
        if recv.numel() * recv.element_size() != nbytes or nbytes % self.world or send.data_ptr() == recv.data_ptr():
            raise ValueError("all_to_all: send / recv of equal size (a multiple of the world size), not aliased")
        check(_lib.lib().pc_rccl_alltoall(self.ctx, _p(send), _p(recv), nbytes // self.world, _stream()), "pc_rccl_alltoall")
        return recv

    def all_reduce_sum_f64_(self, t):
        """pc_rccl_allreduce_sum_f64 on the current stream (cross-replica BatchNorm: ops.p2v_train_step(sync_reduce=...))."""
        _req(t, torch.float64, "buf")
        check(_lib.lib().pc_rccl_allreduce_sum_f64(self.ctx, _p(t), t.numel(), _stream()), "pc_rccl_allreduce_sum_f64")
        return t

    def reduce_scatter_mean_(self, t):
        """pc_rccl_reduce_scatter_mean on the current stream: slice `rank` of t <- the mean over the ranks of that slice."""
        _req(t, torch.float32, "buf")
        if t.numel() % self.world:
            raise ValueError("reduce_scatter_mean_: a multiple of the world size")
        check(_lib.lib().pc_rccl_reduce_scatter_mean(self.ctx, _p(t), t.numel() // self.world, _stream()), "pc_rccl_reduce_scatter_mean")
        return t

    def all_gather_(self, t):
        """pc_rccl_all_gather on the current stream: every rank's slice `rank` of t -> all ranks' t."""
        _req(t, torch.float32, "buf")
        if t.numel() % self.world:
            raise ValueError("all_gather_: a multiple of the world size")
        check(_lib.lib().pc_rccl_all_gather(self.ctx, _p(t), t.numel() // self.world, _stream()), "pc_rccl_all_gather")
        return t

    def stats(self):
        """{collectives issued on the communicator, cross-stream waits the library inserted between them}."""
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        check(_lib.lib().pc_rccl_comm_stats(self.ctx, ctypes.byref(a), ctypes.byref(b)), "pc_rccl_comm_stats")
        return {"issued": int(a.value), "chained": int(b.value)}


class CallbackExchange(Exchange):
    """A Python callable behind the slot: pyfn(ptr, n, stream) must leave the mean over the replicas in the n floats at
    device address ptr, ordered on the stream (it may block).  An exception inside is reported as PC_ECOMM and re-raised by
    the wrapper that made the call."""

    def __init__(self, pyfn, kind="python callback", reduce_scatter=None, all_gather=None, rank=0, world=1):
        """reduce_scatter / all_gather (optional): pyfn-style callables (ptr, n_per_rank, stream) for the sharded optimizer's
        two halves over the buffer at device address ptr (world * n_per_rank floats)."""
        self.error = None

        def wrap(f):
            def tramp(_ctx, buf, n, stream):
                try:
                    f(int(buf or 0), int(n), int(stream or 0))
                    return 0
                except BaseException as e:            # noqa: BLE001 -- nothing may unwind through the C frames
                    self.error = e
                    return -5
            return EXCHANGE_FN(tramp)

        self._tramp = wrap(pyfn)                      # (kept alive with the object: the C side holds a bare address)
        self.fn = ctypes.cast(self._tramp, ctypes.c_void_p)
        if reduce_scatter is not None and all_gather is not None:
            self._tramp_rs, self._tramp_ag = wrap(reduce_scatter), wrap(all_gather)
            self.rs_fn = ctypes.cast(self._tramp_rs, ctypes.c_void_p)
            self.ag_fn = ctypes.cast(self._tramp_ag, ctypes.c_void_p)
        self.rank, self.world = int(rank), int(world)
        self.ctx = ctypes.c_void_p(0)
        self.kind = kind

    def reraise(self):
        e, self.error = self.error, None
        if e is not None:
            raise e


def exchange_adam(exchange, param, grad, exp_avg, exp_avg_sq, step_count, t, scalars, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                  shard=False):
    """pc_exchange_adam: the replicas' mean gradient (exchange may be None: single process), then Adam -- optimizer.step() of a
    replica as one foreign call.  t >= 1: the host-known step number; t == 0: the device counter (scalars required).
    shard=True (pc_exchange_adam_plan, ABI 8): reduce-scatter of the flat gradient, Adam on this rank's 1/world of the flat
    buffers, all-gather of the updated parameters; the buffers' length must be a multiple of exchange.world."""
    n = param.numel()
    if shard:
        if exchange is None:
            raise ValueError("exchange_adam(shard=True) needs an exchange")
        for x, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
            _req(x, torch.float32, nm, (n,))
        if n % exchange.world:
            raise ValueError(f"exchange_adam(shard=True): {n} floats are not a multiple of the world size {exchange.world} "
                             "(flatten_parameters(pad_multiple=world))")
        _req(step_count, torch.int64, "step_count")
        plan = exchange.plan(shard=True)
        rc = _lib.lib().pc_exchange_adam_plan(ctypes.byref(plan), _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, _p(step_count),
                                              int(t), _p(scalars), float(lr), float(betas[0]), float(betas[1]), float(eps), _stream())
        if rc and isinstance(exchange, CallbackExchange):
            exchange.reraise()
        check(rc, "pc_exchange_adam_plan")
        return
    for x, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(x, torch.float32, nm)
        if x.numel() != n:
            raise ValueError("exchange_adam: size mismatch")
    _req(step_count, torch.int64, "step_count")
    rc = _lib.lib().pc_exchange_adam(exchange.fn if exchange is not None else None, exchange.ctx if exchange is not None else None,
                                     _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, _p(step_count), int(t), _p(scalars),
                                     float(lr), float(betas[0]), float(betas[1]), float(eps), _stream())
    if rc and isinstance(exchange, CallbackExchange):
        exchange.reraise()
    check(rc, "pc_exchange_adam")


class KernelProfile:
    """HIP-event brackets around the GEMM launches of the fused step (bench.py roofline leg)."""
    KINDS = {"gemm_nt_kernel": 0, "gemm_tn_kernel": 1, "gemm_nt_small_kernel": 2}

    def __init__(self, capacity):
        self.handle = ctypes.c_void_p()
        check(_lib.lib().pc_profile_create(int(capacity), ctypes.byref(self.handle)), "pc_profile_create")

    def reset(self):
        check(_lib.lib().pc_profile_reset(self.handle), "pc_profile_reset")

    def set_kinds(self, names):
        """Record only the brackets of these kernel families (each bracket costs two event packets)."""
        mask = 0
        for n in names:
            mask |= 1 << self.KINDS[n]
        check(_lib.lib().pc_profile_set_kinds(self.handle, mask), "pc_profile_set_kinds")

    def summary(self, kind):
        n, ms, fl = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        check(_lib.lib().pc_profile_summary(self.handle, self.KINDS[kind], ctypes.byref(n), ctypes.byref(ms),
                                            ctypes.byref(fl)), "pc_profile_summary")
        return {"launches": n.value, "total_ms": ms.value, "total_flops": fl.value}

    def close(self):
        if self.handle:
            _lib.lib().pc_profile_destroy(self.handle)
            self.handle = ctypes.c_void_p()


BN_SYNC_DOUBLES = PC_MAX_SEG * 2 * H + PC_MAX_SEG


def p2v_train_step(params, grads, table, anchor_idx, positive_idx, negative_idx, neighbor_idx, margin,
                   want_emb=False, profile=None, sync_reduce=None, adam=None, structs=None):
    """One loop-body iteration of Product2Vec.train_model in index form (grads overwritten).

    sync_reduce: None = BatchNorm statistics of this batch; else a callable `reduce(buf)` that sums a
    [BN_SYNC_DOUBLES] float64 device tensor over the data-parallel replicas in place (e.g.
    `lambda t: dist.all_reduce(t)`): the step then runs as pc_p2v_train_step_compact_sync's three phases
    with batch-wide (cross-replica) BatchNorm statistics.  Compact neighbour layout only.

    adam (unique-neighbour layout, no sync_reduce): {"param", "grad", "exp_avg", "exp_avg_sq" (the flat buffers `grads` are views
    of), "step_count", "t" >= 1, "lr", "betas", "eps"} -- torch.optim.Adam's update inside the step's last gradient launch
    (pc_p2v_train_step_unique_adam): optimizer.step() costs no launch of its own.

    structs: (st, gst, dev) built earlier by p2v_struct over the SAME tensors (a training loop whose parameters are views of
    fixed flat buffers builds them once: ~25 tensor checks per step less between a drained device and the step's first launch);
    the dropout entry of `params` is applied to it per call."""
    if structs is not None:
        st, gst, dev = structs
        st.dropout.p = 0.0
        _set_dropout(st, params.get(DROPOUT_KEY))
    else:
        st, dev = p2v_struct(params)
        gst, _ = p2v_struct(grads, with_buffers=False)
    b = anchor_idx.numel()
    k = negative_idx.shape[1]
    compact = isinstance(neighbor_idx, dict)          # {"nb_rows": [M+1], "slot_row": [B,N]} (+ "weight", "n_unique")
    unique = compact and "weight" in neighbor_idx
    if compact:
        nb_rows, slot_row = neighbor_idx["nb_rows"], neighbor_idx["slot_row"]
        n = slot_row.shape[1]
        n_real = int(neighbor_idx["n_unique"]) if unique else nb_rows.numel() - 1
        _req(nb_rows, torch.int32, "nb_rows"); _req(slot_row, torch.int32, "slot_row", (b, n))
        if unique:
            _req(neighbor_idx["weight"], torch.float32, "weight")
            if nb_rows.numel() < n_real + 1 or neighbor_idx["weight"].numel() < n_real + 1:
                raise ValueError("nb_rows / weight shorter than n_unique + 1")
    else:
        n = 0 if neighbor_idx is None else neighbor_idx.shape[1]
    _req(table, torch.float32, "table")
    _req(anchor_idx, torch.int32, "anchor_idx", (b,)); _req(positive_idx, torch.int32, "positive_idx", (b,))
    _req(negative_idx, torch.int32, "negative_idx", (b, k))
    if n and not compact:
        _req(neighbor_idx, torch.int32, "neighbor_idx", (b, n))
    out = {"loss": alloc(1, torch.float32, dev), "d_pos": alloc(b, torch.float32, dev), "d_neg": alloc(b, torch.float32, dev)}
    if table.shape[1] != st.dim:
        raise ValueError(f"feature table width {table.shape[1]} != PRODUCT_EMB_DIM {st.dim} of the parameters")
    if want_emb:
        out["anchor_emb"] = torch.empty(b, st.dim, dtype=torch.float32, device=dev)
    nbytes = _lib.lib().pc_p2v_train_step_workspace_bytes_dim(b, n, k, st.dim)
    ws = workspace(nbytes, dev, "step")
    if sync_reduce is not None:
        if not compact:
            raise ValueError("cross-replica BatchNorm needs the compact neighbour layout")
        fwd = torch.zeros(BN_SYNC_DOUBLES, dtype=torch.float64, device=dev)
        bwd_local = torch.zeros(BN_SYNC_DOUBLES, dtype=torch.float64, device=dev)
        bwd_global = None

        def phase(ph):
            if unique:
                check(_lib.lib().pc_p2v_train_step_unique(
                    ctypes.byref(st), ctypes.byref(gst), _p(table), _p(anchor_idx), _p(positive_idx), _p(negative_idx),
                    _p(nb_rows), _p(neighbor_idx["weight"]), n_real, _p(slot_row), _p(neighbor_idx["ref_off"]),
                    _p(neighbor_idx["ref_slot"]), b, n, k, float(margin), _p(out["loss"]),
                    _p(out["d_pos"]), _p(out["d_neg"]), _p(out.get("anchor_emb")), None, ph, _p(fwd), _p(bwd_local),
                    _p(bwd_global), _p(ws), nbytes, _stream()), "pc_p2v_train_step_unique")
                return
            check(_lib.lib().pc_p2v_train_step_compact_sync(
                ctypes.byref(st), ctypes.byref(gst), _p(table), _p(anchor_idx), _p(positive_idx), _p(negative_idx),
                _p(nb_rows), n_real, _p(slot_row), b, n, k, float(margin), _p(out["loss"]), _p(out["d_pos"]),
                _p(out["d_neg"]), _p(out.get("anchor_emb")), ph, _p(fwd), _p(bwd_local), _p(bwd_global), _p(ws), nbytes,
                _stream()), "pc_p2v_train_step_compact_sync")
        phase(0)
        sync_reduce(fwd)
        phase(1)
        bwd_global = bwd_local.clone()
        sync_reduce(bwd_global)
        phase(2)
        return out
    if adam is not None and not unique:
        raise ValueError("p2v_train_step(adam=...): the unique-neighbour layout (the device loader's) carries the fused optimizer step")
    step_rows = neighbor_idx.get("step_rows") if unique else None
    if unique and (adam is not None or step_rows is not None):
        af = None
        if adam is not None:
            af = _lib.AdamFused()
            n_flat = adam["param"].numel()
            for key in ("param", "grad", "exp_avg", "exp_avg_sq"):
                _req(adam[key], torch.float32, key, (n_flat,))
            _req(adam["step_count"], torch.int64, "step_count")
            af.param, af.grad, af.exp_avg, af.exp_avg_sq = (adam[key].data_ptr() for key in ("param", "grad", "exp_avg", "exp_avg_sq"))
            af.n, af.step_count, af.t = n_flat, adam["step_count"].data_ptr(), int(adam["t"])
            af.lr, af.beta1, af.beta2, af.eps = float(adam["lr"]), float(adam["betas"][0]), float(adam["betas"][1]), float(adam["eps"])
        afp = ctypes.byref(af) if af is not None else None
        if step_rows is not None:
            # the loader concatenated the step's row indices behind its builder (concat_step_rows): the step starts with Linear0
            _req(step_rows, torch.int32, "step_rows")
            if step_rows.numel() < b * (2 + k) + n_real + 1:
                raise ValueError("step_rows shorter than B * (2 + K) + n_unique + 1")
            check(_lib.lib().pc_p2v_train_step_unique_rows(
                ctypes.byref(st), ctypes.byref(gst), _p(table), _p(step_rows), _p(nb_rows), _p(neighbor_idx["weight"]), n_real,
                _p(slot_row), _p(neighbor_idx["ref_off"]), _p(neighbor_idx["ref_slot"]), b, n, k, float(margin), _p(out["loss"]),
                _p(out["d_pos"]), _p(out["d_neg"]), _p(out.get("anchor_emb")), profile.handle if profile else None,
                _p(ws), nbytes, afp, _stream()), "pc_p2v_train_step_unique_rows")
            return out
        check(_lib.lib().pc_p2v_train_step_unique_adam(
            ctypes.byref(st), ctypes.byref(gst), _p(table), _p(anchor_idx), _p(positive_idx), _p(negative_idx),
            _p(nb_rows), _p(neighbor_idx["weight"]), n_real, _p(slot_row), _p(neighbor_idx["ref_off"]),
            _p(neighbor_idx["ref_slot"]), b, n, k, float(margin), _p(out["loss"]),
            _p(out["d_pos"]), _p(out["d_neg"]), _p(out.get("anchor_emb")), profile.handle if profile else None,
            _p(ws), nbytes, afp, _stream()), "pc_p2v_train_step_unique_adam")
        return out
    if unique:
        check(_lib.lib().pc_p2v_train_step_unique(
            ctypes.byref(st), ctypes.byref(gst), _p(table), _p(anchor_idx), _p(positive_idx), _p(negative_idx),
            _p(nb_rows), _p(neighbor_idx["weight"]), n_real, _p(slot_row), _p(neighbor_idx["ref_off"]),
            _p(neighbor_idx["ref_slot"]), b, n, k, float(margin), _p(out["loss"]),
            _p(out["d_pos"]), _p(out["d_neg"]), _p(out.get("anchor_emb")), profile.handle if profile else None, -1, None,
            None, None, _p(ws), nbytes, _stream()), "pc_p2v_train_step_unique")
        return out
    if compact:
        check(_lib.lib().pc_p2v_train_step_compact(
            ctypes.byref(st), ctypes.byref(gst), _p(table), _p(anchor_idx), _p(positive_idx), _p(negative_idx),
            _p(nb_rows), n_real, _p(slot_row), b, n, k, float(margin), _p(out["loss"]), _p(out["d_pos"]),
            _p(out["d_neg"]), _p(out.get("anchor_emb")), profile.handle if profile else None, _p(ws), nbytes,
            _stream()), "pc_p2v_train_step_compact")
        return out
    check(_lib.lib().pc_p2v_train_step(ctypes.byref(st), ctypes.byref(gst), _p(table), _p(anchor_idx),
                                       _p(positive_idx), _p(negative_idx), _p(neighbor_idx) if n else None, b, n, k,
                                       float(margin), _p(out["loss"]), _p(out["d_pos"]), _p(out["d_neg"]),
                                       _p(out.get("anchor_emb")), profile.handle if profile else None, _p(ws),
                                       nbytes, _stream()), "pc_p2v_train_step")
    return out


# ----------------------------------------------------------------------------- P1-P4
def build_similarity_batch(pair_ids, graph, n_pad, k_neg, seed, step):
    """graph: dict of int32 CUDA tensors sim_pairs[S,2], cv_rowptr, cv_col, sim_rowptr, sim_col
    and n_products.  Returns anchor_idx, positive_idx, negative_idx[B,K], neighbor_idx[B,n_pad]."""
    b = pair_ids.numel()
    dev = pair_ids.device
    _req(pair_ids, torch.int32, "pair_ids")
    for k in ("sim_pairs", "cv_rowptr", "cv_col", "sim_rowptr", "sim_col"):
        _req(graph[k], torch.int32, k)
    a = torch.empty(b, dtype=torch.int32, device=dev)
    p = torch.empty(b, dtype=torch.int32, device=dev)
    ng = torch.empty(b, k_neg, dtype=torch.int32, device=dev)
    nb = torch.empty(b, n_pad, dtype=torch.int32, device=dev) if n_pad > 0 else None
    check(_lib.lib().pc_build_similarity_batch(_p(pair_ids), b, _p(graph["sim_pairs"]), _p(graph["cv_rowptr"]),
                                               _p(graph["cv_col"]), _p(graph["sim_rowptr"]), _p(graph["sim_col"]),
                                               int(graph["n_products"]), n_pad, k_neg, int(seed), int(step), _p(a),
                                               _p(p), _p(ng), _p(nb), _stream()), "pc_build_similarity_batch")
    return a, p, ng, nb


class BatchBuffers:
    """One slot of a loader's ring of batch buffers, sized for the largest batch (B samples, n_max neighbour slots each):
    the builders write into views of it (`out=`), so a steady-state step allocates nothing and -- the point -- frees
    nothing: every tensor that is allocated on the builder's stream, used on the training stream and then freed costs
    the TRAINING stream an event-record packet at the free (torch's allocator does that for record_stream'ed blocks);
    nine such tensors per batch were a 40-70 us hole at every step boundary (scripts/dev/fixed_batch_probe.py)."""

    def __init__(self, b, n_max, k_neg, device):
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=device)
        slots = b * max(n_max, 1)
        self.b, self.n_max, self.k = b, n_max, k_neg
        self.a, self.p, self.ng = i32(b), i32(b), i32(b * k_neg)
        self.nb_rows, self.ref_off, self.ref_slot, self.slot_row = i32(slots + 2), i32(slots + 3), i32(slots + 1), i32(slots)
        self.weight = torch.empty(slots + 2, dtype=torch.float32, device=device)
        self.n_unique, self.row_off = i32(1), i32(b + 1)
        self.step_rows = i32(b * (2 + k_neg) + slots + 2)      # [anchor | nb_rows | positive | negatives] (concat_step_rows)
        self.host_n = torch.empty(1, dtype=torch.int32).pin_memory()

    def views(self, b, n_pad, n_real, unique):
        if b > self.b or n_pad > self.n_max or n_real > self.b * max(self.n_max, 1):
            raise ValueError("batch larger than the ring's buffers")
        v = {"a": self.a[:b], "p": self.p[:b], "ng": self.ng[:b * self.k].view(b, self.k), "nb_rows": self.nb_rows[:n_real + 1],
             "slot_row": self.slot_row[:b * n_pad].view(b, n_pad), "row_off": self.row_off[:b + 1]}
        if unique:
            v.update(weight=self.weight[:n_real + 1], ref_off=self.ref_off[:n_real + 2], ref_slot=self.ref_slot[:max(n_real, 1)],
                     n_unique=self.n_unique, step_rows=self.step_rows[:b * (2 + self.k) + n_real + 1])
        return v


def build_similarity_batch_compact(pair_ids, graph, n_pad, k_neg, seed, step, n_real, out=None):
    """Same batch with the neighbour rows compacted (pc_build_similarity_batch_compact): returns
    anchor_idx, positive_idx, negative_idx, {"nb_rows": [n_real+1], "slot_row": [B,n_pad]}.  n_real =
    sum of the batch's (capped) co-view degrees, known to the host loader."""
    b = pair_ids.numel()
    dev = pair_ids.device
    _req(pair_ids, torch.int32, "pair_ids")
    if out is not None:
        a, p, ng, nb_rows, slot_row, row_off = (out[k] for k in ("a", "p", "ng", "nb_rows", "slot_row", "row_off"))
    else:
        a = torch.empty(b, dtype=torch.int32, device=dev)
        p = torch.empty(b, dtype=torch.int32, device=dev)
        ng = torch.empty(b, k_neg, dtype=torch.int32, device=dev)
        nb_rows = torch.empty(n_real + 1, dtype=torch.int32, device=dev)
        slot_row = torch.empty(b, n_pad, dtype=torch.int32, device=dev)
        row_off = torch.empty(b + 1, dtype=torch.int32, device=dev)
    check(_lib.lib().pc_build_similarity_batch_compact(
        _p(pair_ids), b, _p(graph["sim_pairs"]), _p(graph["cv_rowptr"]), _p(graph["cv_col"]), _p(graph["sim_rowptr"]),
        _p(graph["sim_col"]), int(graph["n_products"]), n_pad, k_neg, int(seed), int(step), _p(a), _p(p), _p(ng),
        _p(nb_rows), _p(slot_row), _p(row_off), _stream()), "pc_build_similarity_batch_compact")
    return a, p, ng, {"nb_rows": nb_rows, "slot_row": slot_row}


_uq_scratch = {}


def build_similarity_batch_unique(pair_ids, graph, n_pad, k_neg, seed, step, n_real, out=None):
    """Same batch in the unique-neighbour layout (pc_build_similarity_batch_unique): returns anchor_idx,
    positive_idx, negative_idx, {"nb_rows": [n_real+1], "weight": [n_real+1] fp32, "slot_row": [B,n_pad],
    "n_unique": [1] int32 DEVICE tensor}.  Only the first n_unique+1 entries of nb_rows / weight are meaningful;
    the caller reads n_unique back (the loader does so asynchronously, one batch ahead)."""
    b = pair_ids.numel()
    dev = pair_ids.device
    _req(pair_ids, torch.int32, "pair_ids")
    npr = int(graph["n_products"])
    slots = b * n_pad
    key = (dev, npr, torch.cuda.current_stream(dev).cuda_stream)      # (per stream: two loaders over one graph must not share counters)
    need = _lib.lib().pc_build_similarity_batch_unique_scratch_bytes(npr, slots)
    if key not in _uq_scratch or _uq_scratch[key].numel() < need:
        # per-product counters (first 4*P bytes): zero-filled once, every call leaves them zeroed
        _uq_scratch[key] = torch.zeros(max(need, _lib.lib().pc_build_similarity_batch_unique_scratch_bytes(npr, 2 * slots)),
                                       dtype=torch.uint8, device=dev)
    scratch = _uq_scratch[key]
    if out is not None:
        a, p, ng, nb_rows, weight, slot_row, n_unique, ref_off, ref_slot = (
            out[k] for k in ("a", "p", "ng", "nb_rows", "weight", "slot_row", "n_unique", "ref_off", "ref_slot"))
    else:
        a = torch.empty(b, dtype=torch.int32, device=dev)
        p = torch.empty(b, dtype=torch.int32, device=dev)
        ng = torch.empty(b, k_neg, dtype=torch.int32, device=dev)
        nb_rows = torch.empty(n_real + 1, dtype=torch.int32, device=dev)
        weight = torch.empty(n_real + 1, dtype=torch.float32, device=dev)
        slot_row = torch.empty(b, n_pad, dtype=torch.int32, device=dev)
        n_unique = torch.empty(1, dtype=torch.int32, device=dev)
        ref_off = torch.empty(n_real + 2, dtype=torch.int32, device=dev)
        ref_slot = torch.empty(max(n_real, 1), dtype=torch.int32, device=dev)
    check(_lib.lib().pc_build_similarity_batch_unique(
        _p(pair_ids), b, _p(graph["sim_pairs"]), _p(graph["cv_rowptr"]), _p(graph["cv_col"]), _p(graph["sim_rowptr"]),
        _p(graph["sim_col"]), npr, n_pad, k_neg, int(seed), int(step), int(n_real), _p(a), _p(p), _p(ng), _p(nb_rows),
        _p(weight), _p(slot_row), _p(ref_off), _p(ref_slot), _p(n_unique), _p(scratch), scratch.numel(), _stream()),
        "pc_build_similarity_batch_unique")
    return a, p, ng, {"nb_rows": nb_rows, "weight": weight, "slot_row": slot_row, "ref_off": ref_off,
                      "ref_slot": ref_slot, "n_unique": n_unique}


def concat_step_rows(anchor_idx, positive_idx, negative_idx, nb_rows, n_unique_dev, out=None):
    """pc_p2v_concat_step_rows on the current stream: [anchor | nb_rows[0 .. n_unique] | positive | negatives] with n_unique read
    on the device -- what the fused step would otherwise concatenate in a launch of its own.  nb_rows: the builder's [n_real + 1]
    buffer.  Returns the int32 row list (`out` or a new tensor of B * (2 + K) + n_real + 1 entries)."""
    b, k = anchor_idx.numel(), negative_idx.shape[1]
    cap = nb_rows.numel()
    for t, nm in ((anchor_idx, "anchor_idx"), (positive_idx, "positive_idx"), (negative_idx, "negative_idx"), (nb_rows, "nb_rows"),
                  (n_unique_dev, "n_unique")):
        _req(t, torch.int32, nm)
    need = b * (2 + k) + cap
    if out is None:
        out = torch.empty(need, dtype=torch.int32, device=anchor_idx.device)
    _req(out, torch.int32, "step_rows")
    if out.numel() < need:
        raise ValueError(f"step_rows: {out.numel()} entries, B * (2 + K) + len(nb_rows) = {need} needed")
    check(_lib.lib().pc_p2v_concat_step_rows(_p(anchor_idx), _p(positive_idx), _p(negative_idx), _p(nb_rows), _p(n_unique_dev), cap, b,
                                             k, _p(out), out.numel(), _stream()), "pc_p2v_concat_step_rows")
    return out


def unique_neighbors(neighbor_idx):
    """Host-side construction of the unique-neighbour layout from a dense [B,N] index matrix (-1 = padding)."""
    idx = neighbor_idx.cpu().numpy()
    real = idx >= 0
    u, inv, cnt = np.unique(idx[real], return_inverse=True, return_counts=True)
    nb_rows = np.concatenate([u, [-1]]).astype(np.int32)
    weight = np.concatenate([cnt, [idx.size - int(real.sum())]]).astype(np.float32)
    slot = np.full(idx.shape, len(u), np.int32)
    slot[real] = inv.astype(np.int32)
    order = np.argsort(slot.reshape(-1), kind="stable").astype(np.int32)[:int(real.sum())]   # real slots grouped by row
    ref_off = np.concatenate([[0], np.cumsum(cnt), [int(real.sum())]]).astype(np.int32)      # [U + 2]: padding row lists none
    dev = neighbor_idx.device
    return {"nb_rows": torch.from_numpy(nb_rows).to(dev), "weight": torch.from_numpy(weight).to(dev),
            "slot_row": torch.from_numpy(slot).to(dev), "ref_off": torch.from_numpy(ref_off).to(dev),
            "ref_slot": torch.from_numpy(order).to(dev), "n_unique": int(len(u))}


def compact_neighbors(neighbor_idx):
    """Host-side compaction of a dense [B,N] neighbour index matrix (-1 = padding slot)."""
    idx = neighbor_idx.cpu().numpy()
    real = idx >= 0
    nb_rows = np.concatenate([idx[real], [-1]]).astype(np.int32)
    slot = np.full(idx.shape, int(real.sum()), np.int32)
    slot[real] = np.arange(int(real.sum()), dtype=np.int32)
    dev = neighbor_idx.device
    return {"nb_rows": torch.from_numpy(nb_rows).to(dev), "slot_row": torch.from_numpy(slot).to(dev)}


class CPythonRandom:
    """Host-side exact restatement of the CPython `random` stream the reference samples
    from (csrc/host_mt.cpp).  Host numpy arrays in, host numpy arrays out."""

    def __init__(self, seed=0):
        L_ = _lib.lib()
        self._buf = ctypes.create_string_buffer(L_.pc_mt_state_bytes())
        self.seed(seed)

    def seed(self, s):
        check(_lib.lib().pc_mt_seed(self._buf, abs(int(s))), "pc_mt_seed")

    def getrandbits(self, k):
        return int(_lib.lib().pc_mt_getrandbits(self._buf, k))

    def randbelow(self, n):
        return int(_lib.lib().pc_mt_randbelow(self._buf, n))

    def shuffle(self, n):
        perm = np.arange(n, dtype=np.int64)
        check(_lib.lib().pc_mt_shuffle(self._buf, perm.ctypes.data, n), "pc_mt_shuffle")
        return perm

    def negative_samples(self, n_products, sim_rowptr, sim_col, anchors, k=5):
        sim_rowptr = np.ascontiguousarray(sim_rowptr, np.int32)
        sim_col = np.ascontiguousarray(sim_col, np.int32)
        anchors = np.ascontiguousarray(anchors, np.int32)
        out = np.empty((len(anchors), k), np.int32)
        check(_lib.lib().pc_mt_negative_samples(self._buf, n_products, sim_rowptr.ctypes.data, sim_col.ctypes.data,
                                                anchors.ctypes.data, len(anchors), k, out.ctypes.data),
              "pc_mt_negative_samples")
        return out


# ----------------------------------------------------------------------------- joint step
def joint_struct(tensors, table=None):
    st = JointTensors()
    dev = None
    for key, field in zip(JOINT_KEYS, JOINT_FIELDS):
        t = _req(tensors[key], torch.float32, key)
        dev = t.device
        setattr(st, field, t.data_ptr())
    tbl = tensors.get("product_embeddings.weight") if table is None else table
    if tbl is not None:
        _req(tbl, torch.float32, "product_embeddings.weight")
        st.product_table = tbl.data_ptr()
    _set_dropout(st, tensors.get(DROPOUT_KEY))
    return st, dev


def _joint_saved(sv):
    s = JointSaved()
    s.h, s.c, s.pi, s.tp = (sv[k].data_ptr() for k in ("h", "c", "pi", "tp"))
    return s


def joint_forward(params, query_idx, query_types, k):
    st, dev = joint_struct(params)
    b = query_idx.numel()
    t = params["query_type_embeddings.weight"].shape[0]
    _req(query_idx, torch.int32, "query_idx", (b,)); _req(query_types, torch.int32, "query_types", (b,))
    f = dict(dtype=torch.float32, device=dev)
    sims = torch.empty(b, t, **f)
    topk = torch.empty(b, k, dtype=torch.int32, device=dev)
    proj = torch.empty(b, k, D, **f)
    sv = {"h": torch.empty(b, L // 2, **f), "c": torch.empty(b, L, **f), "pi": torch.empty(b, D, **f),
          "tp": torch.empty(b * k, D, **f)}
    check(_lib.lib().pc_joint_forward(ctypes.byref(st), _p(query_idx), _p(query_types), b, t, k, _p(sims), _p(topk),
                                      _p(proj), ctypes.byref(_joint_saved(sv)), None, 0, _stream()),
          "pc_joint_forward")
    return sims, topk, proj, sv


def _width(d):
    """PRODUCT_EMB_DIM of the row-wise joint kernels (item_prediction.py:11-20, p_companion.py:105-119 take it from config)."""
    if int(d) not in (128, 256):
        raise ValueError(f"embedding width {d}: the gfx950 kernels serve PRODUCT_EMB_DIM 128 and 256")
    return int(d)


def joint_loss(sims, proj, pos_types, neg_types, pos_items, neg_items, margin, alpha, need_grad=True):
    b, t = sims.shape
    k, d = proj.shape[1], _width(proj.shape[2])
    dev = sims.device
    _req(sims, torch.float32, "type_similarities"); _req(proj, torch.float32, "projected_embeddings", (b, k, d))
    _req(pos_types, torch.int32, "positive_types", (b,)); _req(neg_types, torch.int32, "negative_types", (b,))
    _req(pos_items, torch.float32, "positive_items", (b, d)); _req(neg_items, torch.float32, "negative_items", (b, d))
    losses = torch.empty(3, dtype=torch.float32, device=dev)
    dsv = torch.empty(b, 2, dtype=torch.float32, device=dev) if need_grad else None
    dproj = torch.empty_like(proj) if need_grad else None
    partials = torch.empty(2 * b, dtype=torch.float32, device=dev)
    check(_lib.lib().pc_joint_loss_dim(_p(sims), _p(proj), _p(pos_types), _p(neg_types), _p(pos_items), _p(neg_items),
                                       b, t, k, d, float(margin), float(alpha), _p(losses), _p(dsv), _p(dproj),
                                       _p(partials), _stream()), "pc_joint_loss_dim")
    return losses, dsv, dproj


def expand_type_grad(dsv, pos_types, neg_types, num_types):
    b = dsv.shape[0]
    dense = torch.empty(b, num_types, dtype=torch.float32, device=dsv.device)
    check(_lib.lib().pc_expand_type_grad(_p(dsv), _p(pos_types), _p(neg_types), b, num_types, _p(dense), _stream()),
          "pc_expand_type_grad")
    return dense


def joint_train_step(params, grads, query_idx, query_types, pos_types, neg_types, pos_items, neg_items, k, margin,
                     alpha):
    st, dev = joint_struct(params)
    gst, _ = joint_struct(grads, table=params["product_embeddings.weight"])
    b = query_idx.numel()
    t = params["query_type_embeddings.weight"].shape[0]
    for x, nm in ((query_idx, "query_idx"), (query_types, "query_types"), (pos_types, "positive_types"),
                  (neg_types, "negative_types")):
        _req(x, torch.int32, nm, (b,))
    _req(pos_items, torch.float32, "positive_items", (b, D)); _req(neg_items, torch.float32, "negative_items", (b, D))
    losses = torch.empty(3, dtype=torch.float32, device=dev)
    topk = torch.empty(b, k, dtype=torch.int32, device=dev)
    nbytes = _lib.lib().pc_joint_workspace_bytes(b, t, k)
    ws = workspace(nbytes, dev, "joint")
    check(_lib.lib().pc_joint_train_step(ctypes.byref(st), ctypes.byref(gst), _p(query_idx), _p(query_types),
                                         _p(pos_types), _p(neg_types), _p(pos_items), _p(neg_items), b, t, k,
                                         float(margin), float(alpha), _p(losses), _p(topk), _p(ws), nbytes, _stream()),
          "pc_joint_train_step")
    return losses, topk


def joint_fused_supported(num_types, k, dropout_p=0.0):
    return bool(_lib.lib().pc_joint_fused_supported(int(num_types), int(k), float(dropout_p)))


def joint_fused_step(params, grads, query_idx, query_types, pos_types, neg_types, pos_items, neg_items, k, margin, alpha,
                     bad=None, adam=None):
    """pc_joint_fused_step: the joint loop body as two launches (T <= 128; the gradient products get their own kernel up to 512).  adam: None (gradients only) or a dict with
    'exp_avg' / 'exp_avg_sq' (tensor dicts keyed like `params`), 'step_count' ([1] int64), 'lr', 'betas', 'eps': the
    optimizer update then happens in the last kernel.  Returns (losses[3], complementary_types[B,K])."""
    st, dev = joint_struct(params)
    gst, _ = joint_struct(grads, table=params["product_embeddings.weight"])
    b = query_idx.numel()
    t = params["query_type_embeddings.weight"].shape[0]
    for x, nm in ((query_idx, "query_idx"), (query_types, "query_types"), (pos_types, "positive_types"),
                  (neg_types, "negative_types")):
        _req(x, torch.int32, nm, (b,))
    _req(pos_items, torch.float32, "positive_items", (b, D)); _req(neg_items, torch.float32, "negative_items", (b, D))
    losses = torch.empty(3, dtype=torch.float32, device=dev)
    topk = torch.empty(b, k, dtype=torch.int32, device=dev)
    nbytes = _lib.lib().pc_joint_fused_workspace_bytes(b, t, k)
    ws = workspace(nbytes, dev, "joint_fused")
    m_ref = v_ref = step = None
    lr, b1, b2, eps = 0.0, 0.0, 0.0, 0.0
    if adam is not None:
        mst, _ = joint_struct(adam["exp_avg"], table=params["product_embeddings.weight"])
        vst, _ = joint_struct(adam["exp_avg_sq"], table=params["product_embeddings.weight"])
        m_ref, v_ref = ctypes.byref(mst), ctypes.byref(vst)
        step = _req(adam["step_count"], torch.int64, "step_count")
        lr, (b1, b2), eps = float(adam["lr"]), adam["betas"], float(adam["eps"])
    if bad is not None:
        _req(bad, torch.int32, "bad", (1,))
    check(_lib.lib().pc_joint_fused_step(
        ctypes.byref(st), ctypes.byref(gst), m_ref, v_ref, _p(step), lr, float(b1), float(b2), eps, _p(query_idx),
        _p(query_types), _p(pos_types), _p(neg_types), _p(pos_items), _p(neg_items), b, t, k,
        int(params["product_embeddings.weight"].shape[0]), float(margin), float(alpha), _p(losses), _p(topk), _p(bad),
        _p(ws), nbytes, _stream()), "pc_joint_fused_step")
    return losses, topk


def joint_fused_touched(ws, b, t, k):
    """pc_joint_fused_touched: (rows_comp, rows_query, n_touched) -- int32 device tensors aliasing the fused step's
    workspace `ws` (a uint8 tensor): ascending touched rows of the two [T,64] tables (capacity-sized; the first
    n_touched[0] / n_touched[1] entries are live) after a step at T > 512."""
    rc, rq, nt = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    check(_lib.lib().pc_joint_fused_touched(_p(ws), ws.numel(), int(b), int(t), int(k), ctypes.byref(rc), ctypes.byref(rq),
                                            ctypes.byref(nt)), "pc_joint_fused_touched")
    base = ws.data_ptr()

    def view(ptr, n):
        off = ptr.value - base
        return ws[off:off + 4 * n].view(torch.int32)
    return view(rc, min(b * (k + 2), t)), view(rq, min(b, t)), view(nt, 2)


class PreparedJointStep:
    """pc_joint_fused_step with every argument resolved once: the loop body then costs one foreign call (the per-step
    argument marshalling of joint_fused_step -- ~50 tensor checks, dict and struct building -- is as long as the three
    kernels themselves).  All buffers are fixed: parameters / gradients / moments (flat-buffer views), the batch
    tensors, the workspace, the outputs.  dropout: (p, seed) or None; the offset advances with every call."""

    def __init__(self, params, grads, batch, k, margin, alpha, bad=None, adam=None, dropout=None):
        self._keep = (params, grads, batch, bad, adam)
        self.st, dev = joint_struct(params)
        self.gst, _ = joint_struct(grads, table=params["product_embeddings.weight"])
        qi, qt, pt, nt = (batch[n].reshape(-1) for n in ("query_idx", "query_types", "positive_types", "negative_types"))
        b = qi.numel()
        t = params["query_type_embeddings.weight"].shape[0]
        for x, nm in ((qi, "query_idx"), (qt, "query_types"), (pt, "positive_types"), (nt, "negative_types")):
            _req(x, torch.int32, nm, (b,))
        pos, neg = batch["positive_items"], batch["negative_items"]
        _req(pos, torch.float32, "positive_items", (b, D)); _req(neg, torch.float32, "negative_items", (b, D))
        self.losses = alloc(3, torch.float32, dev)
        self.topk = alloc(b * k, torch.int32, dev).view(b, k)
        nbytes = _lib.lib().pc_joint_fused_workspace_bytes(b, t, k)
        self.ws = alloc(max(int(nbytes), 256), torch.uint8, dev)
        m_ref = v_ref = step = None
        lr = b1 = b2 = eps = 0.0
        if adam is not None:
            self.mst, _ = joint_struct(adam["exp_avg"], table=params["product_embeddings.weight"])
            self.vst, _ = joint_struct(adam["exp_avg_sq"], table=params["product_embeddings.weight"])
            m_ref, v_ref = ctypes.byref(self.mst), ctypes.byref(self.vst)
            step = _req(adam["step_count"], torch.int64, "step_count")
            lr, (b1, b2), eps = float(adam["lr"]), adam["betas"], float(adam["eps"])
        if bad is not None:
            _req(bad, torch.int32, "bad", (1,))
        self.dropout = dropout
        self.calls = 0
        if dropout is not None:
            self.st.dropout.p, self.st.dropout.seed = float(dropout[0]), int(dropout[1])
        self._fn = _lib.lib().pc_joint_fused_step
        self._args = [ctypes.byref(self.st), ctypes.byref(self.gst), m_ref, v_ref, _p(step), lr, float(b1), float(b2), eps,
                      _p(qi), _p(qt), _p(pt), _p(nt), _p(pos), _p(neg), b, t, k,
                      int(params["product_embeddings.weight"].shape[0]), float(margin), float(alpha), _p(self.losses),
                      _p(self.topk), _p(bad), _p(self.ws), nbytes]
        self._idx = (qi, qt, pt, nt)
        self._pairs_fn = _lib.lib().pc_joint_fused_step_pairs
        self._pairs_src = None
        self._hyper = (1e-3, 0.9, 0.999, 1e-8)

    def set_hyper(self, lr, betas, eps):
        """The optimizer's hyper-parameters are plain doubles among the resolved arguments: refreshed from
        optimizer.param_groups by the caller before a step / an epoch, so that an LR scheduler or a load_state_dict()
        after preparation reaches the fused update exactly as it reaches the eager path."""
        self._hyper = (float(lr), float(betas[0]), float(betas[1]), float(eps))       # (run_epoch_dp: Adam over the flat buffers)
        if self._args[2] is not None:
            self._args[5], self._args[6], self._args[7], self._args[8] = self._hyper

    def __call__(self, dropout_offset=0):
        if self.dropout is not None:
            self.st.dropout.offset = int(dropout_offset)
        rc = self._fn(*self._args, _stream())
        if rc:
            check(rc, "pc_joint_fused_step")
        self.calls += 1
        return self.losses, self.topk

    def from_pairs(self, rows_dev, source, step, dropout_offset=0):
        """The loader's batch construction and the step as one call (pc_joint_fused_step_pairs): `rows_dev` [B,3] int32
        labelled pairs, `source` = (features, type_idx, n_types, seed) of the dataset, `step` the loader's batch counter.
        The batch tensors this object was prepared with are OUTPUTS here: they hold the batch afterwards."""
        if self._pairs_src is None or self._pairs_src[0] is not source:
            features, type_idx, n_types, seed = source
            _req(features, torch.float32, "features", (self._args[18], D)); _req(type_idx, torch.int32, "type_idx", (self._args[18],))
            self._pairs_src = (source, [_p(features), _p(type_idx), int(n_types), int(seed)])
        _req(rows_dev, torch.int32, "pairs", (self._args[15], 3))
        if self.dropout is not None:
            self.st.dropout.offset = int(dropout_offset)
        rc = self._pairs_fn(*self._args[:9], ctypes.c_void_p(rows_dev.data_ptr()), *self._pairs_src[1], int(step),
                            *self._args[9:], _stream())
        if rc:
            check(rc, "pc_joint_fused_step_pairs")
        self.calls += 1
        return self.losses, self.topk


def _prepared_run_epoch(self, pairs_dev, source, first_step, drop_last=False, dropout_offset=0):
    """train.py:36-57 over `pairs_dev` [n,3] (epoch order, on the device) as one foreign call (pc_joint_train_epoch).
    Returns the per-step losses [n_steps,3] (device) and the number of steps."""
    if self._args[2] is None:
        raise ValueError("run_epoch needs the optimizer state (PreparedJointStep(adam=...))")
    features, type_idx, n_types, seed = source
    _req(features, torch.float32, "features", (self._args[18], D)); _req(type_idx, torch.int32, "type_idx", (self._args[18],))
    n = int(pairs_dev.shape[0])
    _req(pairs_dev, torch.int32, "pairs", (n, 3))
    b = self._args[15]
    steps = n // b if drop_last else (n + b - 1) // b
    losses = torch.empty(max(steps, 1), 3, dtype=torch.float32, device=pairs_dev.device)
    if self.dropout is not None:
        self.st.dropout.offset = int(dropout_offset)
    a = self._args
    rc = _lib.lib().pc_joint_train_epoch(*a[:9], _p(pairs_dev), n, _p(features), _p(type_idx), int(n_types), int(seed),
                                         int(first_step), *a[9:16], int(bool(drop_last)), *a[16:21], _p(losses), *a[22:],
                                         _stream())
    if rc:
        check(rc, "pc_joint_train_epoch")
    self.calls += steps
    self._keep_epoch = (pairs_dev, features, type_idx)
    return losses[:steps], steps


PreparedJointStep.run_epoch = _prepared_run_epoch


def _prepared_run_epoch_dp(self, pairs_dev, source, first_step, flat, gflat, exp_avg, exp_avg_sq, step_count, t_first, scalars,
                           exchange, drop_last=True, dropout_offset=0, shard=False):
    """pc_joint_train_epoch_plan: the epoch of a data-parallel REPLICA as one foreign call -- per step the fused step without its
    Adam, the exchange (ops.RcclExchange / CallbackExchange / None) and Adam over the flat buffers; shard=True: reduce-scatter,
    Adam on this rank's slice, all-gather (ABI 8).  The object must have been prepared WITHOUT adam (gradients only) over
    parameters / gradients that are views of flat / gflat."""
    if self._args[2] is not None:
        raise ValueError("run_epoch_dp: prepare the step without adam= (the epoch applies Adam over the flat buffers itself)")
    features, type_idx, n_types, seed = source
    _req(features, torch.float32, "features", (self._args[18], D)); _req(type_idx, torch.int32, "type_idx", (self._args[18],))
    n = int(pairs_dev.shape[0])
    _req(pairs_dev, torch.int32, "pairs", (n, 3))
    nf = flat.numel()
    for x, nm in ((flat, "param_flat"), (gflat, "grad_flat"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(x, torch.float32, nm, (nf,))
    _req(step_count, torch.int64, "step_count")
    b = self._args[15]
    steps = n // b if drop_last else (n + b - 1) // b
    losses = torch.empty(max(steps, 1), 3, dtype=torch.float32, device=pairs_dev.device)
    if self.dropout is not None:
        self.st.dropout.offset = int(dropout_offset)
    a = self._args
    if shard and (exchange is None or nf % exchange.world):
        raise ValueError("run_epoch_dp(shard=True): an exchange, and flat buffers whose length is a multiple of its world size")
    plan = exchange.plan(shard=bool(shard)) if exchange is not None else None
    rc = _lib.lib().pc_joint_train_epoch_plan(a[0], a[1], _p(flat), _p(gflat), _p(exp_avg), _p(exp_avg_sq), nf, _p(step_count),
                                              int(t_first), _p(scalars), float(self._hyper[0]), float(self._hyper[1]),
                                              float(self._hyper[2]), float(self._hyper[3]),
                                              ctypes.byref(plan) if plan is not None else None,
                                              _p(pairs_dev), n, _p(features), _p(type_idx), int(n_types), int(seed), int(first_step),
                                              *a[9:16], int(bool(drop_last)), *a[16:21], _p(losses), *a[22:], _stream())
    if rc and isinstance(exchange, CallbackExchange):
        exchange.reraise()
    if rc:
        check(rc, "pc_joint_train_epoch_plan")
    self.calls += steps
    self._keep_epoch = (pairs_dev, features, type_idx, flat, gflat, exp_avg, exp_avg_sq, step_count, scalars, exchange)
    return losses[:steps], steps


PreparedJointStep.run_epoch_dp = _prepared_run_epoch_dp


class PreparedComplementaryBuilder:
    """pc_build_complementary_batch into FIXED output buffers with the arguments resolved once (see PreparedJointStep)."""

    def __init__(self, features, type_idx, n_types, seed, out):
        _req(features, torch.float32, "features"); _req(type_idx, torch.int32, "type_idx")
        self.b = out["query_idx"].numel()
        for k in ("query_idx", "query_types", "positive_types", "negative_types"):
            _req(out[k], torch.int32, k)
        for k in ("positive_items", "negative_items"):
            _req(out[k], torch.float32, k, (self.b, D))
        self._keep = (features, type_idx, out)
        self._fn = _lib.lib().pc_build_complementary_batch
        self._tail = [_p(out["query_idx"]), _p(out["query_types"]), _p(out["positive_types"]), _p(out["negative_types"]),
                      _p(out["positive_items"]), _p(out["negative_items"]), _p(out.get("target_features"))]
        self._head = [_p(features), _p(type_idx), int(n_types), int(seed)]

    def __call__(self, rows_dev, step):
        rc = self._fn(ctypes.c_void_p(rows_dev.data_ptr()), self.b, *self._head, int(step), *self._tail, _stream())
        if rc:
            check(rc, "pc_build_complementary_batch")


# ----------------------------------------------------------------------------- building blocks
def _pad_cols(t, mult=4):
    """[..., c] -> [..., ceil(c / mult) * mult] with zero columns (the kernels move 16-byte chunks: contraction and
    gradient dimensions that are not multiples of 4 -- an odd NUM_TYPES in module mode -- are zero-padded here)."""
    c = t.shape[-1]
    return t if c % mult == 0 else torch.nn.functional.pad(t, (0, mult - c % mult)).contiguous()


def linear_forward(x, w, b=None, idx=None, act=0, rows=None):
    out_dim, in_dim = w.shape
    if in_dim % 4:                                   # zero columns add nothing to x . w
        x, w = _pad_cols(x.reshape(-1, in_dim)), _pad_cols(w)
        in_dim = w.shape[1]
    _req(x, torch.float32, "x"); _req(w, torch.float32, "weight")
    if b is not None:
        _req(b, torch.float32, "bias", (out_dim,))
    if idx is not None:
        _req(idx, torch.int32, "idx")
        rows = idx.numel()
    elif rows is None:
        rows = x.numel() // in_dim
    y = torch.empty(rows, out_dim, dtype=torch.float32, device=w.device)
    check(_lib.lib().pc_linear_forward(_p(x), _p(idx), rows, in_dim, _p(w), _p(b), out_dim, act, _p(y), _stream()),
          "pc_linear_forward")
    return y


def linear_backward_input(dy, w):
    out_dim, in_dim = w.shape
    rows = dy.numel() // out_dim
    if out_dim % 4:                                  # dx = dy . w contracts over out_dim
        dy = _pad_cols(dy.reshape(rows, out_dim))
        w = torch.cat([w, w.new_zeros(dy.shape[1] - out_dim, in_dim)])
        out_dim = dy.shape[1]
    _req(dy, torch.float32, "dy"); _req(w, torch.float32, "weight")
    dx = torch.empty(rows, in_dim, dtype=torch.float32, device=w.device)
    wt = torch.empty(in_dim, out_dim, dtype=torch.float32, device=w.device)
    check(_lib.lib().pc_linear_backward_input(_p(dy), rows, out_dim, _p(w), in_dim, 0, None, _p(dx), _p(wt),
                                              _stream()), "pc_linear_backward_input")
    return dx


def linear_backward_weight(dy, x, out_dim, in_dim, idx=None, want_bias=True):
    rows = dy.numel() // out_dim
    if out_dim % 4 or in_dim % 4:
        dyp, xp = _pad_cols(dy.reshape(rows, out_dim)), _pad_cols(x.reshape(-1, in_dim))
        dw, db = linear_backward_weight(dyp, xp, dyp.shape[1], xp.shape[1], idx, want_bias)
        return dw[:out_dim, :in_dim].contiguous(), (db[:out_dim].contiguous() if db is not None else None)
    _req(dy, torch.float32, "dy"); _req(x, torch.float32, "x")
    if idx is not None:
        _req(idx, torch.int32, "idx", (rows,))
    dev = dy.device
    dw = torch.empty(out_dim, in_dim, dtype=torch.float32, device=dev)
    db = torch.empty(out_dim, dtype=torch.float32, device=dev) if want_bias else None
    nbytes = _lib.lib().pc_linear_backward_weight_workspace_bytes(rows, out_dim, in_dim)
    ws = workspace(nbytes, dev)
    check(_lib.lib().pc_linear_backward_weight(_p(dy), rows, out_dim, _p(x), _p(idx), in_dim, _p(dw), _p(db), 0,
                                               _p(ws), nbytes, _stream()), "pc_linear_backward_weight")
    return dw, db


def topk_rows(sims, k, want_values=False):
    b, t = sims.shape
    _req(sims, torch.float32, "sims")
    idx = torch.empty(b, k, dtype=torch.int32, device=sims.device)
    val = torch.empty(b, k, dtype=torch.float32, device=sims.device) if want_values else None
    check(_lib.lib().pc_topk_rows(_p(sims), b, t, k, _p(idx), _p(val), _stream()), "pc_topk_rows")
    return (idx, val) if want_values else idx


def hadamard_forward(pi, tp, k):
    b, d = pi.shape[0], _width(pi.shape[1])
    _req(pi, torch.float32, "pi", (b, d)); _req(tp, torch.float32, "tp", (b * k, d))
    proj = torch.empty(b, k, d, dtype=torch.float32, device=pi.device)
    check(_lib.lib().pc_hadamard_forward_dim(_p(pi), _p(tp), b, k, d, _p(proj), _stream()), "pc_hadamard_forward_dim")
    return proj


def hadamard_backward(dproj, pi, tp):
    b, k, d = dproj.shape
    _width(d)
    _req(dproj, torch.float32, "dproj", (b, k, d))
    dpi = torch.empty(b, d, dtype=torch.float32, device=pi.device)
    dtp = torch.empty(b * k, d, dtype=torch.float32, device=pi.device)
    check(_lib.lib().pc_hadamard_backward_dim(_p(dproj), _p(pi), _p(tp), b, k, d, _p(dpi), _p(dtp), _stream()),
          "pc_hadamard_backward_dim")
    return dpi, dtp


def gather_rows(table, idx):
    rows = idx.numel()
    width = table.shape[1]
    _req(table, torch.float32, "table"); _req(idx, torch.int32, "idx")
    out = torch.empty(rows, width, dtype=torch.float32, device=table.device)
    check(_lib.lib().pc_gather_rows(_p(table), _p(idx), rows, width, _p(out), _stream()), "pc_gather_rows")
    return out


def scatter_add_rows(table, idx, src):
    rows = idx.numel()
    width = table.shape[1]
    _req(table, torch.float32, "table"); _req(idx, torch.int32, "idx"); _req(src, torch.float32, "src", (rows, width))
    check(_lib.lib().pc_scatter_add_rows(_p(table), _p(idx), rows, width, _p(src), _stream()), "pc_scatter_add_rows")
    return table


def scatter_rows(out, idx, src):
    rows, width = idx.numel(), out.shape[1]
    _req(out, torch.float32, "out"); _req(idx, torch.int32, "idx"); _req(src, torch.float32, "src", (rows, width))
    check(_lib.lib().pc_scatter_rows(_p(out), _p(idx), rows, width, _p(src), _stream()), "pc_scatter_rows")
    return out


def act_backward(dy, y, act):
    _req(dy, torch.float32, "dy"); _req(y, torch.float32, "y", dy.shape)
    dx = torch.empty_like(dy)
    check(_lib.lib().pc_act_backward(_p(dy), _p(y), dy.numel(), act, _p(dx), _stream()), "pc_act_backward")
    return dx


def zipf_octave_thresholds(n_products):
    """Cumulative 32-bit thresholds of the octave masses of P(rank) ~ 1 / rank, rank = 1..n_products (host, float64 once;
    the device and the oracle then compare integers only).  octave j = ranks [2^j, min(2^(j+1), P + 1))."""
    n_oct = int(n_products).bit_length()
    mass = []
    for j in range(n_oct):
        lo, hi = 1 << j, min((1 << (j + 1)) - 1, int(n_products))
        if hi - lo < 4096:
            mass.append(float(np.sum(1.0 / np.arange(lo, hi + 1, dtype=np.float64))))
        else:                                    # Euler-Maclaurin: sum_{k=lo}^{hi} 1/k, error < 1e-12 for lo >= 4096
            mass.append(float(np.log(hi / lo) + 0.5 / lo + 0.5 / hi + (1.0 / lo ** 2 - 1.0 / hi ** 2) / 12.0))
    cum = np.cumsum(mass) / np.sum(mass)
    thr = np.minimum(np.floor(cum * 4294967296.0), 4294967295.0).astype(np.uint64).astype(np.uint32)
    thr[-1] = 0xFFFFFFFF
    return thr


def exclusive_scan_i32(counts):
    """pc_exclusive_scan_i32: int32 [n] on the device -> (offsets int32 [n + 1], total as a Python int; raises when the
    total does not fit the int32 offsets)."""
    _req(counts, torch.int32, "counts")
    n = counts.numel()
    out = torch.empty(n + 1, dtype=torch.int32, device=counts.device)
    total = torch.zeros(1, dtype=torch.int64, device=counts.device)
    nbytes = _lib.lib().pc_scan_scratch_bytes(n)
    scratch = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=counts.device)
    check(_lib.lib().pc_exclusive_scan_i32(_p(counts), n, _p(out), _p(total), _p(scratch), nbytes, _stream()),
          "pc_exclusive_scan_i32")
    t = int(total.item())
    if t >= 2 ** 31:
        raise OverflowError(f"{t} entries do not fit int32 offsets")
    return out, t


def generate_catalogue(num_products, num_types, seed, mean_degree, degree_cap, dim, device, rank=0, world=1,
                       comp_mean=4.5, with_complementary=True, with_features=True):
    """csrc/generator.hip end to end; returns a dict of device tensors (see data.DeviceBPG)."""
    L = _lib.lib()
    P = int(num_products)
    dev = torch.device(device)
    i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=dev)
    out = {"n_products": P}
    type_idx = i32(P)
    check(L.pc_gen_types(P, int(num_types), int(seed), _p(type_idx), _stream()), "pc_gen_types")
    out["type_idx"] = type_idx
    if with_features:
        n_local = (P - rank + world - 1) // world
        feats = torch.empty(n_local, dim, dtype=torch.float32, device=dev)
        check(L.pc_gen_features(rank, world, n_local, int(dim), int(num_types), int(seed), _p(feats), _stream()), "pc_gen_features")
        out["features"] = feats
    deg, cand = i32(P), (i32(P) if with_complementary else None)
    check(L.pc_gen_degrees(P, float(mean_degree), int(degree_cap), float(comp_mean), int(seed), _p(deg), _p(cand), _stream()),
          "pc_gen_degrees")
    cv_rowptr, n_edges = exclusive_scan_i32(deg)
    # (offsets and ids are int32: exclusive_scan_i32 raises when a total reaches 2^31; the kernels index the two-int pair
    # array with size_t, so pair counts up to that limit are safe -- 100 M products: 1.6e9 edges, 275 M pairs)
    cv_col, sim_count = i32(max(n_edges, 1)), i32(P)
    check(L.pc_gen_coview(P, int(num_types), int(seed), _p(cv_rowptr), _p(cv_col), _p(sim_count), _stream()), "pc_gen_coview")
    sim_rowptr, n_sim = exclusive_scan_i32(sim_count)
    sim_pairs, sim_col, pair_deg = i32(max(n_sim, 1), 2), i32(max(n_sim, 1)), i32(max(n_sim, 1))
    check(L.pc_gen_similarity(P, _p(cv_rowptr), _p(cv_col), _p(sim_rowptr), _p(sim_pairs), _p(sim_col), _p(pair_deg), _stream()),
          "pc_gen_similarity")
    out.update(cv_rowptr=cv_rowptr, cv_col=cv_col[:n_edges], sim_rowptr=sim_rowptr, sim_pairs=sim_pairs[:n_sim],
               sim_col=sim_col[:n_sim], pair_deg=pair_deg[:n_sim], max_degree=int(degree_cap))
    del deg, sim_count
    if with_complementary:
        cnt = i32(P)
        check(L.pc_gen_complementary(P, int(num_types), int(seed), _p(cand), _p(cv_rowptr), _p(cv_col), _p(cnt), None, None,
                                     _stream()), "pc_gen_complementary")
        comp_rowptr, n_comp = exclusive_scan_i32(cnt)
        comp = i32(max(n_comp, 1), 2)
        check(L.pc_gen_complementary(P, int(num_types), int(seed), _p(cand), _p(cv_rowptr), _p(cv_col), None, _p(comp_rowptr),
                                     _p(comp), _stream()), "pc_gen_complementary")
        out["comp_pairs"] = comp[:n_comp]
    return out


def epoch_permutation(n, seed, epoch, device):
    """pc_epoch_permutation: the epoch's order of n dataset positions, int32 [n] on the device (keyed Feistel bijection,
    cycle-walked: no sort kernels, no torch.randperm)."""
    out = torch.empty(int(n), dtype=torch.int32, device=device)
    check(_lib.lib().pc_epoch_permutation(int(n), int(seed) & (2 ** 64 - 1), int(epoch), _p(out), _stream()), "pc_epoch_permutation")
    return out


def shuffle_rows_i32(rows, seed, epoch):
    """pc_shuffle_rows_i32: rows [n,w] int32 -> the rows in the epoch's order (a new tensor)."""
    _req(rows, torch.int32, "rows")
    n, w = rows.shape
    out = torch.empty_like(rows)
    check(_lib.lib().pc_shuffle_rows_i32(_p(rows), n, w, int(seed) & (2 ** 64 - 1), int(epoch), _p(out), _stream()), "pc_shuffle_rows_i32")
    return out


def epoch_plan(order, deg, n, batch, n_batches):
    """pc_epoch_plan: per batch (padded neighbour count, real neighbour slots) -> int64 [n_batches,2] on the device."""
    _req(deg, torch.int32, "deg")
    if order is not None:
        _req(order, torch.int32, "order")
    plan = torch.empty(n_batches, 2, dtype=torch.int64, device=deg.device)
    check(_lib.lib().pc_epoch_plan(_p(order), _p(deg), int(n), int(batch), int(n_batches), _p(plan), _stream()), "pc_epoch_plan")
    return plan


def sample_negatives_zipf(pair_ids, graph, k_neg, seed, step, thresholds, perm=None, out=None, failed=None):
    """pc_sample_negatives_zipf: returns negative_idx [B,k_neg] (int32, device).  thresholds: uint32 device tensor from
    zipf_octave_thresholds (viewed as int32 storage).  failed: optional int32 [1] device counter of samples that ran out
    of proposals (see the header)."""
    b = pair_ids.numel()
    _req(pair_ids, torch.int32, "pair_ids"); _req(thresholds, torch.int32, "thresholds")
    if perm is not None:
        _req(perm, torch.int32, "perm", (int(graph["n_products"]),))
    ng = out if out is not None else torch.empty(b, k_neg, dtype=torch.int32, device=pair_ids.device)
    check(_lib.lib().pc_sample_negatives_zipf(_p(pair_ids), b, _p(graph["sim_pairs"]), _p(graph["sim_rowptr"]), _p(graph["sim_col"]),
                                              int(graph["n_products"]), int(k_neg), int(seed), int(step), _p(thresholds),
                                              thresholds.numel(), _p(perm), _p(ng), _p(failed), _stream()), "pc_sample_negatives_zipf")
    return ng


def shard_bucket(arrays, world, capacity, counts, send_ids, overflow, hot_rows=0, hot_ids=None, hot_served=None):
    """pc_shard_bucket[_hot].  arrays: up to four (ids int32 tensor, live-length device tensor or None, add) triples; returns
    the remapped index tensors (same shapes).  counts [world], send_ids [world*capacity], overflow [1]: int32 device.
    hot_rows > 0: ids of the replicated hot set (hot_ids ascending [hot_rows] int32, None = the ids below hot_rows) map to
    world * capacity + their position in the set and take no request slot; hot_served [1] int32 counts them."""
    n = len(arrays)
    if not 1 <= n <= 4:
        raise ValueError("1..4 id arrays per launch")
    outs = [torch.empty_like(_req(t, torch.int32, "ids")) for t, _, _ in arrays]
    _req(counts, torch.int32, "counts", (world,)); _req(send_ids, torch.int32, "send_ids", (world * capacity,))
    _req(overflow, torch.int32, "overflow", (1,))
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t, _, _ in arrays])
    lens = (ctypes.c_int * n)(*[t.numel() for t, _, _ in arrays])
    ndev = (ctypes.c_void_p * n)(*[(_req(d, torch.int32, "live length").data_ptr() if d is not None else None) for _, d, _ in arrays])
    nadd = (ctypes.c_int * n)(*[int(a) for _, _, a in arrays])
    optrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    if hot_rows:
        if hot_ids is not None:
            _req(hot_ids, torch.int32, "hot_ids", (int(hot_rows),))
        if hot_served is not None:
            _req(hot_served, torch.int32, "hot_served", (1,))
        check(_lib.lib().pc_shard_bucket_hot(ptrs, lens, ndev, nadd, optrs, n, int(world), int(capacity), _p(hot_ids), int(hot_rows),
                                             _p(counts), _p(send_ids), _p(overflow), _p(hot_served), _stream()), "pc_shard_bucket_hot")
        return outs
    check(_lib.lib().pc_shard_bucket(ptrs, lens, ndev, nadd, optrs, n, int(world), int(capacity), _p(counts), _p(send_ids),
                                     _p(overflow), _stream()), "pc_shard_bucket")
    return outs


def dropout_hidden(x, dropout):
    """x * mask for the type-transition hidden layer (pc_dropout_hidden); dropout = (p, seed, offset)."""
    _req(x, torch.float32, "x")
    d = _lib.Dropout()
    d.p, d.seed, d.offset = float(dropout[0]), int(dropout[1]), int(dropout[2])
    if not 0.0 < d.p < 1.0:
        raise ValueError(f"dropout probability has to be between 0 and 1, but got {dropout[0]}")
    y = torch.empty_like(x)
    check(_lib.lib().pc_dropout_hidden(_p(x), x.numel(), ctypes.byref(d), _p(y), _stream()), "pc_dropout_hidden")
    return y


def check_indices(jobs, bad):
    """jobs: up to four (int32 index tensor, table rows, allow -1) triples; adds the number of out-of-range entries
    to the device counter `bad` ([1] int32).  Asynchronous: nothing is read back here."""
    n = len(jobs)
    if not 1 <= n <= 4:
        raise ValueError("1..4 index arrays per launch")
    _req(bad, torch.int32, "bad", (1,))
    ptrs = (ctypes.c_void_p * n)(*[_req(t, torch.int32, "idx").data_ptr() for t, _, _ in jobs])
    cnt = (ctypes.c_int * n)(*[t.numel() for t, _, _ in jobs])
    hi = (ctypes.c_int * n)(*[int(h) for _, h, _ in jobs])
    pad = (ctypes.c_int * n)(*[1 if a else 0 for _, _, a in jobs])
    check(_lib.lib().pc_check_indices(ptrs, cnt, hi, pad, n, _p(bad), _stream()), "pc_check_indices")


def hit_rank(sims):
    rows, cols = sims.shape
    _req(sims, torch.float32, "similarities")
    rank = torch.empty(rows, dtype=torch.int32, device=sims.device)
    check(_lib.lib().pc_hit_rank(_p(sims), rows, cols, _p(rank), _stream()), "pc_hit_rank")
    return rank


def cosine_rows(x, y):
    b, k, d = x.shape
    _width(d)
    _req(x, torch.float32, "predictions", (b, k, d)); _req(y, torch.float32, "ground_truth", (b, d))
    out = torch.empty(b * k, dtype=torch.float32, device=x.device)
    check(_lib.lib().pc_cosine_rows_dim(_p(x), _p(y), b, k, d, _p(out), _stream()), "pc_cosine_rows_dim")
    return out


def build_complementary_batch(pairs, features, type_idx, n_types, seed, step, want_targets=True, out=None):
    """pairs [B,3] int32 (query, target, label) on the device -> the joint-step batch dict.  `out`: a dict of
    preallocated tensors of those shapes to build into (fixed buffers of a graphed step)."""
    b = pairs.shape[0]
    _req(pairs, torch.int32, "pairs", (b, 3)); _req(features, torch.float32, "features"); _req(type_idx, torch.int32, "type_idx")
    dev = pairs.device
    i32 = lambda: torch.empty(b, dtype=torch.int32, device=dev)
    f32 = lambda: torch.empty(b, D, dtype=torch.float32, device=dev)
    if out is not None:
        out = dict(out)
        for k in ("query_idx", "query_types", "positive_types", "negative_types"):
            _req(out[k], torch.int32, k)
            if out[k].numel() != b:
                raise ValueError("build_complementary_batch: out[%r] has %d elements, batch is %d" % (k, out[k].numel(), b))
        for k in ("positive_items", "negative_items"):
            _req(out[k], torch.float32, k, (b, D))
        want_targets = "target_features" in out
    else:
        out = {"query_idx": i32(), "query_types": i32(), "positive_types": i32(), "negative_types": i32(),
               "positive_items": f32(), "negative_items": f32()}
        if want_targets:
            out["target_features"] = f32()
    check(_lib.lib().pc_build_complementary_batch(
        _p(pairs), b, _p(features), _p(type_idx), int(n_types), int(seed), int(step), _p(out["query_idx"]),
        _p(out["query_types"]), _p(out["positive_types"]), _p(out["negative_types"]), _p(out["positive_items"]),
        _p(out["negative_items"]), _p(out.get("target_features")), _stream()), "pc_build_complementary_batch")
    out["positive_types"] = out["positive_types"].view(b, 1)
    out["negative_types"] = out["negative_types"].view(b, 1)
    return out


def retrieve_topk(proj, types, type_rowptr, type_col, table, n):
    """pc_retrieve_topk: proj [R,128] fp32, types [R] int32 -> (idx [R,n] int32 product indices, -1 = none;
    scores [R,n] fp32).  inference.py:90-118 for all rows at once."""
    d = _width(table.shape[1])
    proj = _req(proj.reshape(-1, d), torch.float32, "proj")
    r = proj.shape[0]
    _req(types, torch.int32, "types", (r,))
    _req(type_rowptr, torch.int32, "type_rowptr")
    _req(type_col, torch.int32, "type_col")
    _req(table, torch.float32, "table")
    out_idx = torch.empty(r, n, dtype=torch.int32, device=proj.device)
    out_sc = torch.empty(r, n, dtype=torch.float32, device=proj.device)
    check(_lib.lib().pc_retrieve_topk_dim(_p(proj), _p(types), r, _p(type_rowptr), _p(type_col), _p(table),
                                          type_rowptr.numel() - 1, int(n), d, _p(out_idx), _p(out_sc), _stream()),
          "pc_retrieve_topk_dim")
    return out_idx, out_sc
