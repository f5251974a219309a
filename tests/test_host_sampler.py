"""Host-side exact sampler of the product (csrc/host_mt.cpp through the C ABI) against the
golden vectors, live CPython and the oracle.  Host code only: runs without a GPU."""
import random

import numpy as np
import pytest

from oracle.mt import Random as OracleRandom


@pytest.fixture(scope="module")
def CPy():
    from p_companion_amd.ops import CPythonRandom
    return CPythonRandom


@pytest.mark.parametrize("seed", [0, 1, 12345, 2**40 + 7])
def test_stream_golden(golden, CPy, seed):
    g = golden("g1_mt19937.npz")
    r = CPy(seed)
    tag = f"s{seed}_"
    assert [r.getrandbits(10) for _ in range(64)] == g[tag + "getrandbits10"].tolist()
    assert [r.getrandbits(32) for _ in range(16)] == g[tag + "getrandbits32"].tolist()
    assert [r.randbelow(1000) for _ in range(64)] == g[tag + "choice1000"].tolist()
    # shuffle after the same prefix as the fixture (64 + 16 + 64 draws, then 16 random() = 32 words)
    r3 = CPy(seed)
    [r3.getrandbits(10) for _ in range(64)]; [r3.getrandbits(32) for _ in range(16)]
    [r3.randbelow(1000) for _ in range(64)]; [r3.getrandbits(32) for _ in range(32)]
    assert r3.shuffle(32).tolist() == g[tag + "shuffle32"].tolist()


def test_stream_vs_live_cpython_and_oracle(CPy):
    for seed in (0, 3, 2**33 + 5):
        random.seed(seed)
        r, o = CPy(seed), OracleRandom(seed)
        for n in (1, 2, 7, 1000, 100000, 2**31 + 11, 2**40 + 3):
            want = [random.randrange(n) for _ in range(40)]
            assert [r.randbelow(n) for _ in range(40)] == want
            assert [o.randbelow(n) for _ in range(40)] == want
        lst = list(range(100)); random.shuffle(lst)
        assert r.shuffle(100).tolist() == lst


@pytest.mark.parametrize("seed", [0, 7])
def test_negative_samples_golden(golden, CPy, seed):
    from p_companion_amd.data import IntBPG
    bpg = IntBPG.from_arrays(golden("g2_bpg1000.npz"))
    g = golden("g3_negatives.npz")
    got = CPy(seed).negative_samples(1000, bpg.sim_rowptr, bpg.sim_col, bpg.similarity_pairs[:256, 0], 5)
    assert np.array_equal(got, g[f"s{seed}_negatives"])                    # bit-exact vs the reference
    ora = OracleRandom(seed).negative_samples(1000, bpg.similarity_pairs, bpg.similarity_pairs[:256, 0], 5)
    assert np.array_equal(got, ora)


def test_negative_samples_impossible_request_is_refused(CPy):
    # 4 products, anchor 0 has positives {1,2}: only product 3 is admissible, k=2 can never finish
    rowptr = np.array([0, 2, 2, 2, 2], np.int32); col = np.array([1, 2], np.int32)
    from p_companion_amd._lib import HipKernelError
    with pytest.raises(HipKernelError):
        CPy(0).negative_samples(4, rowptr, col, np.array([0], np.int32), 2)
