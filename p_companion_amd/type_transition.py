"""ComplementaryTypeTransition on MI355X -- drop-in for src/models/type_transition.py.
decoder(dropout(relu(encoder(x)))), Linear 64->32->64 (type_transition.py:11-19), each
Linear one few-row MFMA launch (pc_linear_forward: v_mfma_f32_32x32x2_f32, exact fp32 products) with the ReLU fused
in the epilogue."""
import torch
import torch.nn as nn

from .functional import dropout_hidden, linear


class ComplementaryTypeTransition(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        # parameter containers (same names/initialisers as type_transition.py:11-13)
        self.encoder = nn.Linear(config.TYPE_EMB_DIM, config.TYPE_EMB_DIM // 2)
        self.decoder = nn.Linear(config.TYPE_EMB_DIM // 2, config.TYPE_EMB_DIM)
        self.dropout = nn.Dropout(config.DROPOUT)
        self._dropout_seed, self._dropout_step = None, 0

    def _next_dropout(self):
        """(p, seed, offset) of this training-mode forward's hidden-layer dropout, None when off.  The mask is the
        build's own counter-based stream (pc_dropout in the header): same distribution as nn.Dropout, not ATen's bits."""
        p = float(getattr(self.config, "DROPOUT", 0.0))
        if not self.training or p == 0.0:
            return None
        if self._dropout_seed is None:
            self._dropout_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self._dropout_step += 1
        return (p, self._dropout_seed, self._dropout_step - 1)

    def forward(self, query_type_embedding, _dropout="draw"):
        """_dropout: internal -- PCompanion passes the (p, seed, offset) its fused forward already used, so that a
        lazily rebuilt autograd graph sees the same mask; 'draw' = a fresh one (stand-alone use)."""
        drop = self._next_dropout() if _dropout == "draw" else _dropout
        shape = query_type_embedding.shape
        x = query_type_embedding.reshape(-1, shape[-1])
        h = linear(x, self.encoder.weight, self.encoder.bias, act="relu")
        if drop is not None:
            h = dropout_hidden(h, drop)
        complementary_base = linear(h, self.decoder.weight, self.decoder.bias)
        return complementary_base.reshape(*shape[:-1], -1)
