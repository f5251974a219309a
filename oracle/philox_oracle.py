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


# ---------------------------------------------------------------------------------------------------------
# Dropout masks of the HIP path (csrc/common.h pc_dropout_keep4).  ATen's dropout stream cannot be reproduced
# (SURVEY.md section 7), so the build draws its own: element e of a dropped tensor belongs to group e >> 2; the four
# keep decisions of a group are the four words of Philox4x32-10(counter = (group, stream, offset lo, offset hi),
# key = seed), word i decides element 4 * group + i: KEEP iff word >= floor(p * 2^32); kept values are scaled by
# 1 / (1 - p) in fp32.  stream 0 = attention probabilities [B, heads, N] (nn.MultiheadAttention(dropout=p),
# product2vec.py:23-28), stream 1 = the type-transition hidden layer [B, 32] (nn.Dropout, type_transition.py:13,17).
# `offset` is the module's training-step counter (one fresh mask per forward).
STREAM_ATTENTION, STREAM_HIDDEN = 0, 1


def dropout_threshold(p):
    return min(int(float(np.float32(p)) * 4294967296.0), 4294967295)


def dropout_mask(seed, offset, stream, n_elements, p):
    """[n_elements] float32 multipliers: 0 for a dropped element, fp32(1 / (1 - p)) for a kept one."""
    if p <= 0:
        return np.ones(n_elements, np.float32)
    thr = dropout_threshold(p)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    key = (seed & MASK, (seed >> 32) & MASK)
    out = np.empty((n_elements + 3) // 4 * 4, np.float32)
    for g in range(len(out) // 4):
        w = philox4x32_10((g & MASK, stream, offset & MASK, (offset >> 32) & MASK), key)
        for i in range(4):
            out[4 * g + i] = scale if w[i] >= thr else np.float32(0.0)
    return out[:n_elements]


# ---------------------------------------------------------------------------------------------------------
# Zipf(s = 1) negatives of the HIP path (csrc/sampler.hip zipf_negatives_kernel; BASELINE configs[4] -- an extension,
# the reference samples uniformly).  Same integer procedure: octave by cumulative 32-bit thresholds, a rank in it by j
# random bits, accepted with probability 2^j / rank; then the rejection rules of data_loader.py:33-38.
ZIPF_TAG = 0x5a495046


def zipf_negatives(pair_ids, sim_pairs, sim_rowptr, sim_col, n_products, k, seed, step, thresholds, perm=None):
    thr = [int(x) for x in np.asarray(thresholds).astype(np.uint32)]
    out = np.zeros((len(pair_ids), k), np.int32)
    for b, pid in enumerate(pair_ids):
        a = int(sim_pairs[pid, 0])
        pos = set(sim_col[sim_rowptr[a]:sim_rowptr[a + 1]].tolist())
        s = Stream(seed ^ ZIPF_TAG, step, b)
        got = []
        while len(got) < k:
            r0 = s.next()
            j = 0
            while j + 1 < len(thr) and r0 > thr[j]:
                j += 1
            base = 1 << j
            while True:                                   # inside the chosen octave until a rank is accepted
                kk = base + ((s.next() >> (32 - j)) if j else 0)
                if kk > n_products:
                    continue
                if s.next() * kk < (base << 32):
                    break
            c = int(perm[kk - 1]) if perm is not None else kk - 1
            if c != a and c not in pos and c not in got:
                got.append(c)
        out[b] = got
    return out


# ---------------------------------------------------------------------------------------------------------
# The loaders' epoch order on the device (csrc/sampler.hip epoch_permutation_kernel; replaces torch.randperm): a keyed
# six-round balanced Feistel bijection over 2^k >= n (k even), cycle-walked into [0, n).  No reference counterpart
# (DataLoader(shuffle=True) draws from torch's CPU generator, scripts/pretrain_product2vec.py:24-30): pinned only
# against this restatement and by the bijection property.
M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return x, z ^ (z >> 31)


def feistel_keys(n, seed, epoch):
    bits = 2
    while bits < 62 and (1 << bits) < n:
        bits += 2
    x = (seed * 0xD1342543DE82EF95 + epoch * 0x2545F4914F6CDD1D + 0x1234567) & M64
    keys = []
    for _ in range(6):
        x, z = _splitmix64(x)
        keys.append(z >> 32)
    return bits // 2, keys


def epoch_permutation(n, seed, epoch):
    half, keys = feistel_keys(n, seed, epoch)
    mask = np.uint64((1 << half) - 1)
    M32 = np.uint64(0xFFFFFFFF)

    def apply(x):
        L, R = x >> np.uint64(half), x & mask
        for k in keys:
            v = (R * np.uint64(0xCC9E2D51) + np.uint64(k)) & M32
            v ^= v >> np.uint64(15); v = (v * np.uint64(0x85EBCA6B)) & M32
            v ^= v >> np.uint64(13); v = (v * np.uint64(0xC2B2AE35)) & M32
            v ^= v >> np.uint64(16)
            L, R = R, L ^ (v & mask)
        return (L << np.uint64(half)) | R

    x = apply(np.arange(n, dtype=np.uint64))
    while True:
        out = x >= np.uint64(n)
        if not out.any():
            break
        x[out] = apply(x[out])
    return x.astype(np.int32)
