"""Serving counterpart of the reference's inference.py:10-124 (PCompanionInference).

The reference file is a non-running stub (it builds PCompanion without its embeddings argument and
feeds forward() keys it does not read), so the semantics are taken from its recommend() body
(:64-124): run the model on the query, then for every predicted complementary type take the products
of that type, score them by <projected embedding, product features> and keep torch.topk of the scores.
Here the candidate search of all (query, type) rows is one launch of pc_retrieve_topk over a
type-grouped CSR of the product table."""
import os
from typing import Any, Dict, List

import numpy as np
import torch

from . import ops
from .data import IntBPG
from .p_companion import PCompanion


class PCompanionInference:
    def __init__(self, model, config, bpg: IntBPG, product_ids: List[str] = None):
        """model: a trained PCompanion, or the path of a best_model.pth written by train.train
        (train.py:63-70 layout: 'model_state_dict' holds every tensor, including the frozen table)."""
        self.config = config
        self.device = config.DEVICE
        self.bpg = bpg
        if isinstance(model, (str, os.PathLike)):
            model = self._load_model(model)
        self.model = model.to(self.device)
        self.model.eval()
        self.product_ids = product_ids
        self._index = {pid: i for i, pid in enumerate(product_ids)} if product_ids is not None else None
        g = bpg.cuda(self.device)
        self.features = g["features"]
        self.type_idx = g["type_idx"]
        # bpg.get_products_by_type(t) (bpg.py:40-43): products of type t in node order
        order = np.argsort(bpg.type_idx, kind="stable").astype(np.int32)
        counts = np.bincount(bpg.type_idx, minlength=bpg.n_types)
        rowptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        self.type_rowptr = torch.from_numpy(rowptr).to(self.device)
        self.type_col = torch.from_numpy(order).to(self.device)

    def _load_model(self, model_path):
        """Load trained model weights (inference.py:28-36)"""
        if not os.path.exists(model_path):
            raise FileNotFoundError(f"Model file not found: {model_path}")
        checkpoint = torch.load(model_path, map_location="cpu", weights_only=True)
        sd = checkpoint["model_state_dict"]
        model = PCompanion(self.config, sd["product_embeddings.weight"])
        model.load_state_dict(sd)
        return model

    def _to_index(self, query_id):
        if isinstance(query_id, (int, np.integer)):
            idx = int(query_id)
        elif self._index is not None:
            if query_id not in self._index:
                raise ValueError(f"Product ID {query_id} not found in BPG")      # inference.py:40-41
            idx = self._index[query_id]
        else:
            idx = int(str(query_id).lstrip("P"))
        if not 0 <= idx < self.bpg.num_products:
            raise ValueError(f"Product ID {query_id} not found in BPG")
        return idx

    @torch.no_grad()
    def recommend_batch(self, query_idx: torch.Tensor, num_recommendations: int = 10):
        """All queries at once.  Returns complementary_types [B,K] int64, product indices [B,K,n] int32
        (-1 past the end of a short type) and scores [B,K,n]."""
        query_idx = query_idx.to(self.device).to(torch.int32).contiguous()
        batch = {"query_idx": query_idx, "query_types": self.type_idx[query_idx.long()]}
        out = self.model(batch)
        types = out["complementary_types"]
        b, k = types.shape
        idx, sc = ops.retrieve_topk(out["projected_embeddings"].contiguous().reshape(b * k, -1),
                                    types.to(torch.int32).reshape(-1).contiguous(), self.type_rowptr, self.type_col,
                                    self.features, int(num_recommendations))
        return types, idx.reshape(b, k, -1), sc.reshape(b, k, -1)

    def recommend(self, query_id, num_recommendations: int = 10) -> Dict[str, Any]:
        """Generate complementary product recommendations (inference.py:64-124): same result dict."""
        q = torch.tensor([self._to_index(query_id)], dtype=torch.int32)
        types, idx, sc = self.recommend_batch(q, num_recommendations)
        types, idx, sc = types[0].cpu().numpy(), idx[0].cpu().numpy(), sc[0].cpu().numpy()
        recommendations, scores = [], []
        for j in range(len(types)):
            keep = idx[j] >= 0
            if not keep.any():
                continue                                                          # `if not type_products: continue`
            ids = idx[j][keep]
            recommendations.append([self.product_ids[i] for i in ids] if self.product_ids is not None else ids.tolist())
            scores.append(sc[j][keep])
        return {"complementary_types": types.tolist(), "recommendations": recommendations, "scores": scores}
