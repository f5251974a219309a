"""Phase drivers -- counterparts of scripts/pretrain_product2vec.py:11-53 and train.py:16-72.

Same call signatures and the same checkpoint dict layouts (so files interchange with the
reference): product2vec.pth = {'model_state_dict', 'embeddings', 'type_to_idx'};
best_model.pth = {'epoch', 'model_state_dict', 'optimizer_state_dict', 'metrics'}."""
import logging
import os
from typing import Dict

import torch

from .data import ComplementaryIndexDataset, ComplementaryIndexLoader, IntBPG, SimilarityIndexLoader
from .metrics import Metrics
from .p_companion import PCompanion
from .product2vec import FusedAdam, Product2Vec


def pretrain_product2vec(config, similarity_dataset) -> Dict[str, torch.Tensor]:
    """Pretrain Product2Vec model and save embeddings (scripts/pretrain_product2vec.py:11-53).
    `similarity_dataset`: an IntBPG (index loader built here), an index loader, or any iterable
    of reference-style dense batches with a .dataset.bpg attribute."""
    logger = logging.getLogger(__name__)
    if isinstance(similarity_dataset, IntBPG):
        loader = SimilarityIndexLoader(similarity_dataset, config.BATCH_SIZE, shuffle=True, sampler="philox",
                                       device=config.DEVICE, reuse_buffers=True)     # (train_model consumes each batch before the next)
    else:
        loader = similarity_dataset
    model = Product2Vec(config).to(config.DEVICE)
    optimizer = FusedAdam(model, lr=config.LEARNING_RATE)
    embeddings_dict = model.train_model(train_loader=loader, optimizer=optimizer, num_epochs=config.PRODUCT2VEC_EPOCHS)
    os.makedirs(config.MODEL_DIR, exist_ok=True)
    save_path = os.path.join(config.MODEL_DIR, "product2vec.pth")
    bpg = loader.dataset.bpg
    torch.save({"model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "embeddings": embeddings_dict,
                "type_to_idx": bpg.type_to_idx if hasattr(bpg, "type_to_idx") else None}, save_path)
    logger.info(f"Saved pretrained Product2Vec model and embeddings to {save_path}")
    pretrain_product2vec.last_model = model
    return embeddings_dict


def _cpu_state(sd):
    return {"state": {k: {n: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for n, v in st.items()}
                      for k, st in sd["state"].items()}, "param_groups": sd["param_groups"]}


def _check_ranges(config, loader, model):
    """The kernels index the [NUM_TYPES,64] tables with the batch's type ids and the product table with its product
    ids; the reference raises IndexError / KeyError for an id outside them (p_companion.py:48-54).  Checked once
    here for the index loaders (whose ids come from the graph), per step on the device otherwise (PCompanion.index_errors)."""
    bpg = getattr(getattr(loader, "dataset", None), "bpg", None)
    if bpg is None or not hasattr(bpg, "n_types"):
        return
    if int(bpg.n_types) > int(config.NUM_TYPES):
        raise IndexError(f"the graph has {bpg.n_types} types but config.NUM_TYPES = {config.NUM_TYPES}: type ids "
                         "would index past the type embedding tables")
    if bpg.num_products > model.product_embeddings.weight.shape[0]:
        raise IndexError(f"the graph has {bpg.num_products} products but the pretrained table holds "
                         f"{model.product_embeddings.weight.shape[0]} rows")


def train(config, train_loader, val_loader, pretrained_embeddings, fused=True):
    """Train P-Companion model (train.py:16-72): loop, per-epoch Metrics.evaluate_model, best
    hit@10 checkpoint.  fused=True runs the loop body as pc_joint_train_step + one Adam launch;
    fused=False runs model(batch) / compute_loss / backward / torch Adam like the reference.
    Returns the model (the reference returns None); model.step_losses [steps] holds every step's total loss (what the
    reference's progress bar averages, :50-51) and model.epoch_metrics the metrics dict of every epoch (:54)."""
    logger = logging.getLogger(__name__)
    model = PCompanion(config, pretrained_embeddings).to(config.DEVICE)
    optimizer = FusedAdam(model, lr=config.LEARNING_RATE) if fused else \
        torch.optim.Adam(model.parameters(), lr=config.LEARNING_RATE)
    for ld in (train_loader, val_loader):
        _check_ranges(config, ld, model)
    best_hit10 = 0.0
    history, epoch_metrics = [], []
    # the index loader of this package on the GPU: train.py:36-57's loop over an epoch runs as ONE foreign call
    # (GraphedJointStep.run_epoch -> pc_joint_train_epoch; same steps, same values as the loop below)
    epoch_runner = None
    loader_out = (train_loader.out, getattr(train_loader, "_prepared", None)) if isinstance(train_loader, ComplementaryIndexLoader) else None
    if fused and isinstance(train_loader, ComplementaryIndexLoader) and torch.device(config.DEVICE).type == "cuda" and \
            train_loader.out is None and len(train_loader.dataset) >= train_loader.batch_size:
        from .p_companion import GraphedJointStep
        step = GraphedJointStep(model, optimizer, train_loader.batch_size, warmup=0, mode="auto")
        if step.mode == "direct":
            train_loader.out, train_loader._prepared = step.static, None
            epoch_runner = step
    for epoch in range(config.NUM_EPOCHS):
        model.train()
        total = None
        nb = 0
        if epoch_runner is not None:
            per_step = epoch_runner.run_epoch(train_loader)
            total, nb = per_step[:, 0].sum().reshape(1), int(per_step.shape[0])
            history.append(per_step[:, 0].detach().clone())
        for batch in (train_loader if epoch_runner is None else ()):
            batch = {k: v.to(config.DEVICE) if isinstance(v, torch.Tensor) else v for k, v in batch.items()}
            if fused:
                losses, _ = model.train_step(batch, optimizer=optimizer)      # Adam applied by the step's last kernel
                loss = losses[0:1]
            else:
                outputs = model(batch)
                loss = model.compute_loss(batch, outputs)
                optimizer.zero_grad()
                loss.backward()
                loss = loss.detach().reshape(1)
                optimizer.step()
            total = loss.clone() if total is None else total + loss
            history.append(loss.detach().clone())
            nb += 1
        if nb:
            logger.info(f"Epoch {epoch + 1}/{config.NUM_EPOCHS}, Loss: {float(total) / nb:.4f}")
        model.raise_index_errors()                   # ids outside the tables seen by the device this epoch -> IndexError
        metrics = Metrics.evaluate_model(model, val_loader, config.DEVICE)
        epoch_metrics.append(dict(metrics))
        for name, value in metrics.items():
            logger.info(f"{name}: {value:.4f}")
        if metrics["hit@10"] > best_hit10:
            best_hit10 = metrics["hit@10"]
            os.makedirs(config.MODEL_DIR, exist_ok=True)
            torch.save({"epoch": epoch,
                        "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                        "optimizer_state_dict": _cpu_state(optimizer.state_dict()),      # torch.optim.Adam's layout in both modes
                        "metrics": metrics}, os.path.join(config.MODEL_DIR, "best_model.pth"))
        logger.info(f"Best Hit@10: {best_hit10:.4f}")
    if epoch_runner is not None:                     # the caller's loader leaves as it came (its batches no longer alias the step's buffers)
        train_loader.out, train_loader._prepared = loader_out
        train_loader._static_batch = None
    model.step_losses = torch.cat([h.reshape(-1) for h in history]).cpu() if history else torch.zeros(0)
    model.epoch_metrics = epoch_metrics
    return model


def main(config, bpg: IntBPG):
    """train.py:74-134 on an integer BPG: Product2Vec pretrain -> embeddings -> P-Companion."""
    embeddings = pretrain_product2vec(config, bpg)
    table = pretrain_product2vec.last_model.last_embedding_table
    tr = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "train"), config.BATCH_SIZE, shuffle=True,
                                  device=config.DEVICE)
    va = ComplementaryIndexLoader(ComplementaryIndexDataset(bpg, "val"), config.BATCH_SIZE, shuffle=False,
                                  device=config.DEVICE)
    return train(config, tr, va, table if table is not None else embeddings)
