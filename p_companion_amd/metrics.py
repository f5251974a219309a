"""Metrics on the device -- drop-in for src/utils/metrics.py (Metrics.evaluate_model runs every
epoch in train.train and selects the best checkpoint, train.py:55-70).

The similarity product [B*K,128] x [128,B] is the shared NT GEMM (pc_linear_forward: fp32 MFMA for few rows, fp32-grade
six-product bf16 MFMA for many), hit@k is a rank kernel
(one wave per row), relevance a cosine kernel; only the final scalar means are read back.
Reproduces the reference's quirk: ground truth is arange(B*K) against B columns, so rows >= B can
never hit (metrics.py:95-100)."""
from typing import Dict

import numpy as np
import torch

from . import ops


class Metrics:
    @staticmethod
    def hit_at_k(predictions: torch.Tensor, ground_truth: torch.Tensor, k: int) -> float:
        """predictions [R, C] scores; ground_truth [R]; metrics.py:7-27.  The device kernel covers
        the only use in the reference (ground_truth == arange(R))."""
        k = min(k, predictions.size(1))
        if not torch.equal(ground_truth.cpu(), torch.arange(predictions.size(0))):
            raise NotImplementedError("hit_at_k kernel: ground_truth must be arange(rows) (metrics.py:100)")
        rank = ops.hit_rank(predictions.contiguous().float())
        return float((rank < k).float().mean())

    @staticmethod
    def type_diversity(predicted_types: torch.Tensor) -> float:
        """metrics.py:29-42: unique COLUMNS of [B,K] / K (tiny integer matrix: host side)."""
        if predicted_types.numel() == 0:
            return 0.0
        cols = np.unique(predicted_types.cpu().numpy(), axis=1)
        return cols.shape[1] / predicted_types.size(1)

    @staticmethod
    def mean_relevance(predictions: torch.Tensor, ground_truth: torch.Tensor) -> float:
        """metrics.py:44-60"""
        return float(ops.cosine_rows(predictions.contiguous().float(), ground_truth.contiguous().float()).mean())

    @staticmethod
    def evaluate_model(model: torch.nn.Module, data_loader, device) -> Dict[str, float]:
        """metrics.py:62-117"""
        model.eval()
        metrics = {"hit@1": 0.0, "hit@3": 0.0, "hit@10": 0.0, "type_diversity": 0.0, "mean_relevance": 0.0}
        num_batches = 0
        with torch.no_grad():
            for batch in data_loader:
                batch = {k: v.to(device) if torch.is_tensor(v) else v for k, v in batch.items()}
                outputs = model(batch)
                proj = outputs["projected_embeddings"]
                targets = batch["target_features"].float().contiguous()
                similarities = ops.linear_forward(proj.reshape(-1, proj.size(-1)).contiguous(), targets)   # [B*K, B]
                rank = ops.hit_rank(similarities)
                for k in [1, 3, min(10, similarities.size(1))]:
                    metrics[f"hit@{k}"] += float((rank < min(k, similarities.size(1))).float().mean())
                metrics["type_diversity"] += Metrics.type_diversity(outputs["complementary_types"])
                metrics["mean_relevance"] += Metrics.mean_relevance(proj, batch["positive_items"].to(device))
                num_batches += 1
        for key in metrics:
            metrics[key] /= max(num_batches, 1)
        return metrics
