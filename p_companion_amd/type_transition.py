"""ComplementaryTypeTransition on MI355X -- drop-in for src/models/type_transition.py.
decoder(dropout(relu(encoder(x)))), Linear 64->32->64 (type_transition.py:11-19), each
Linear one few-row MFMA launch (pc_linear_forward: v_mfma_f32_32x32x2_f32, exact fp32 products) with the ReLU fused
in the epilogue."""
import torch
import torch.nn as nn

from .functional import linear


class ComplementaryTypeTransition(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        # parameter containers (same names/initialisers as type_transition.py:11-13)
        self.encoder = nn.Linear(config.TYPE_EMB_DIM, config.TYPE_EMB_DIM // 2)
        self.decoder = nn.Linear(config.TYPE_EMB_DIM // 2, config.TYPE_EMB_DIM)
        self.dropout = nn.Dropout(config.DROPOUT)

    def forward(self, query_type_embedding):
        if self.training and float(self.config.DROPOUT) != 0.0:
            raise NotImplementedError("DROPOUT != 0 is not implemented in the HIP path; set config.DROPOUT = 0")
        shape = query_type_embedding.shape
        x = query_type_embedding.reshape(-1, shape[-1])
        h = linear(x, self.encoder.weight, self.encoder.bias, act="relu")
        complementary_base = linear(h, self.decoder.weight, self.decoder.bias)
        return complementary_base.reshape(*shape[:-1], -1)
