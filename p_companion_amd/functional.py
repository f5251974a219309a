"""Autograd wrappers over the HIP building blocks (module / dense mode).

Each Function's forward and backward are HIP kernels reached through the C ABI; autograd
only routes the gradient tensors between them (as it does for the reference's ATen ops).
"""
import torch

from . import ops

_ACT = {None: 0, "none": 0, "tanh": 1, "relu": 2}


def _c(t):
    if not t.is_cuda:
        raise TypeError("HIP path: tensors must live on the GPU (no CPU fallback)")
    return t.contiguous().float()


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b)   (nn.Linear; act fused in the GEMM epilogue)"""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x, w = _c(x), _c(w)
        y = ops.linear_forward(x, w, None if b is None else _c(b), act=act)
        ctx.act = act
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y if act else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.act:
            dy = ops.act_backward(dy, y, ctx.act)        # dy * act'(y), one elementwise HIP kernel
        dx = ops.linear_backward_input(dy, w) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = ops.linear_backward_weight(dy, x, w.shape[0], w.shape[1], want_bias=ctx.has_bias)
        return dx, dw, db, None


def linear(x, w, b=None, act=None):
    return _Linear.apply(x, w, b, _ACT[act])


class _Embedding(torch.autograd.Function):
    """table[idx]  (nn.Embedding forward = row gather; backward = row-sparse scatter-add)"""

    @staticmethod
    def forward(ctx, table, idx):
        table = _c(table)
        idx32 = idx.reshape(-1).to(torch.int32).contiguous()
        ctx.save_for_backward(idx32)
        ctx.shape = table.shape
        return ops.gather_rows(table, idx32).view(*idx.shape, table.shape[1])

    @staticmethod
    def backward(ctx, dy):
        (idx32,) = ctx.saved_tensors
        dt = torch.zeros(ctx.shape, dtype=torch.float32, device=dy.device)
        ops.scatter_add_rows(dt, idx32, dy.reshape(idx32.numel(), -1).contiguous())
        return dt, None


def embedding(table, idx):
    return _Embedding.apply(table, idx)


class _Hadamard(torch.autograd.Function):
    """proj[b,k,:] = pi[b,:] * tp[b*K+k,:]   (item_prediction.py:38)"""

    @staticmethod
    def forward(ctx, pi, tp, k):
        pi, tp = _c(pi), _c(tp)
        ctx.save_for_backward(pi, tp)
        return ops.hadamard_forward(pi, tp, k)

    @staticmethod
    def backward(ctx, dproj):
        pi, tp = ctx.saved_tensors
        dpi, dtp = ops.hadamard_backward(dproj.contiguous(), pi, tp)
        return dpi, dtp, None


def hadamard(pi, tp, k):
    return _Hadamard.apply(pi, tp, k)


class _DropoutHidden(torch.autograd.Function):
    """nn.Dropout on the type-transition hidden layer (type_transition.py:17) with the build's counter-based mask;
    the backward multiplies by the same mask (regenerated from seed / offset)."""

    @staticmethod
    def forward(ctx, x, dropout):
        ctx.dropout = dropout
        return ops.dropout_hidden(_c(x), dropout)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout_hidden(dy.contiguous(), ctx.dropout), None


def dropout_hidden(x, dropout):
    return _DropoutHidden.apply(x, dropout)
