"""ComplementaryItemPrediction on MI355X -- drop-in for src/models/item_prediction.py.
item_projection(q)[:, None, :] * type_projection(c)  (item_prediction.py:22-40): two
Linear launches (pc_linear_forward, see type_transition.py) and one Hadamard kernel."""
import torch
import torch.nn as nn

from .functional import hadamard, linear


class ComplementaryItemPrediction(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        # parameter containers (same names/order as item_prediction.py:11-20)
        self.type_projection = nn.Linear(config.TYPE_EMB_DIM, config.PRODUCT_EMB_DIM)
        self.item_projection = nn.Linear(config.PRODUCT_EMB_DIM, config.PRODUCT_EMB_DIM)

    def forward(self, query_item_embedding, complementary_type_embeddings):
        """query_item_embedding [B,128], complementary_type_embeddings [B,K,64] -> [B,K,128]"""
        projected_item = linear(query_item_embedding, self.item_projection.weight, self.item_projection.bias)
        batch_size, num_types, _ = complementary_type_embeddings.shape
        type_projections = linear(complementary_type_embeddings.reshape(batch_size * num_types, -1),
                                  self.type_projection.weight, self.type_projection.bias)
        return hadamard(projected_item, type_projections, num_types)
