"""ORACLE (test infrastructure).  CPU restatement of the throughput-mode negative sampler:
the rejection rules of SimilarityDataset._get_negative_samples (data_loader.py:33-38) on a
Philox4x32-10 stream (Salmon et al., SC'11; constants of Random123) keyed by (seed; draw
block, sample, step) -- the counter layout documented in csrc/sampler.hip.  Pure Python
integers: use for small batches only."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    x0, x1, x2, x3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * x0, M1 * x2
        x0, x1, x2, x3 = ((p1 >> 32) ^ x1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ x3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return x0, x1, x2, x3


class Stream:
    def __init__(self, seed, step, sample):
        self.key = (seed & MASK, (seed >> 32) & MASK)
        self.block = 0
        self.sample, self.step = sample, step
        self.buf = []

    def next(self):
        if not self.buf:
            self.buf = list(philox4x32_10((self.block, self.sample, self.step & MASK, (self.step >> 32) & MASK), self.key))
            self.block += 1
        return self.buf.pop(0)

    def below(self, n):
        bits = int(n).bit_length()
        r = self.next() >> (32 - bits)
        while r >= n:
            r = self.next() >> (32 - bits)
        return r


def build_batch(pair_ids, sim_pairs, cv_rowptr, cv_col, sim_rowptr, sim_col, n_products, n_pad, k, seed, step):
    B = len(pair_ids)
    a = sim_pairs[pair_ids, 0].astype(np.int32)
    p = sim_pairs[pair_ids, 1].astype(np.int32)
    neg = np.zeros((B, k), np.int32)
    nb = np.full((B, n_pad), -1, np.int32)
    for b in range(B):
        s = Stream(seed, step, b)
        pos = set(sim_col[sim_rowptr[a[b]]:sim_rowptr[a[b] + 1]].tolist())
        got = []
        while len(got) < k:
            c = s.below(n_products)
            if c != a[b] and c not in pos and c not in got:
                got.append(c)
        neg[b] = got
        row = cv_col[cv_rowptr[a[b]]:cv_rowptr[a[b] + 1]][:n_pad]
        nb[b, :len(row)] = row
    return a, p, neg, nb
