"""ORACLE (test infrastructure).  Batch construction semantics of the reference loader on
the integer-form BPG, numpy only."""
import numpy as np


def neighbors(cv_rowptr, cv_col, pid):
    """BehaviorProductGraph.get_neighbors(pid, 'co_view') (bpg.py:24-38): directed
    out-neighbours; order = the CSR's stored order (the reference's set-iteration order
    in the golden fixtures)."""
    return cv_col[cv_rowptr[pid]:cv_rowptr[pid + 1]]


def collate_neighbors(lists, pad=-1):
    """collate_fn (data_loader.py:186-198) in index form: ragged neighbour lists are
    right-padded to the batch maximum; pad slots gather an all-zero feature row."""
    nmax = max(len(l) for l in lists)
    out = np.full((len(lists), nmax), pad, np.int32)
    for i, l in enumerate(lists):
        out[i, :len(l)] = l
    return out


def similarity_batch(ints, sample_ids, negatives):
    """SimilarityDataset.__getitem__ + collate_fn (data_loader.py:45-71,171-206) for the
    given dataset positions, negatives supplied by the sampler."""
    pairs = ints["similarity_pairs"][np.asarray(sample_ids)]
    nb = collate_neighbors([neighbors(ints["cv_rowptr"], ints["cv_col"], a) for a in pairs[:, 0]])
    return dict(anchor_idx=pairs[:, 0].astype(np.int32), positive_idx=pairs[:, 1].astype(np.int32),
                negative_idx=np.asarray(negatives, np.int32), neighbor_idx=nb)


def complementary_sample_ints(query, target, label, type_idx, n_types):
    """ComplementaryDataset.__getitem__ integer fields (data_loader.py:133-157):
    label +1: positive_types = t(target), negative_types = (t(target)+1) % n_types;
    label -1: positive_types = 0,         negative_types = t(target)."""
    tt = int(type_idx[target])
    return dict(query_types=int(type_idx[query]),
                positive_types=tt if label == 1 else 0,
                negative_types=tt if label == -1 else (tt + 1) % n_types)
