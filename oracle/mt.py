"""ORACLE (test infrastructure).  ctypes loader for oracle/mt_oracle.c (CPython `random`
stream + data_loader.py:27-40 negative sampler) and its build recipe."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmt_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "mt_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", src, "-o", _SO])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.orc_state_size.restype = ctypes.c_size_t
        L.orc_seed_u64.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        L.orc_u32.restype = ctypes.c_uint32
        L.orc_u32.argtypes = [ctypes.c_void_p]
        L.orc_getrandbits.restype = ctypes.c_uint64
        L.orc_getrandbits.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.orc_random.restype = ctypes.c_double
        L.orc_random.argtypes = [ctypes.c_void_p]
        L.orc_randbelow.restype = ctypes.c_uint64
        L.orc_randbelow.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        L.orc_shuffle.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        L.orc_negative_samples.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                           ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
        _lib = L
    return _lib


class Random:
    """The subset of random.Random the reference consumes, on the C restatement."""

    def __init__(self, seed=0):
        self._buf = ctypes.create_string_buffer(lib().orc_state_size())
        self.seed(seed)

    def seed(self, s):
        lib().orc_seed_u64(self._buf, abs(int(s)))

    def getrandbits(self, k):
        return int(lib().orc_getrandbits(self._buf, k))

    def random(self):
        return float(lib().orc_random(self._buf))

    def randbelow(self, n):
        return int(lib().orc_randbelow(self._buf, n))

    def choice_index(self, n):
        return self.randbelow(n)

    def randint(self, a, b):
        return a + self.randbelow(b - a + 1)

    def shuffle_perm(self, n):
        x = np.arange(n, dtype=np.int64)
        lib().orc_shuffle(self._buf, x.ctypes.data, n)
        return x

    def negative_samples(self, n_products, pairs, anchors, k=5):
        pairs = np.ascontiguousarray(pairs, np.int32)
        anchors = np.ascontiguousarray(anchors, np.int32)
        out = np.empty((len(anchors), k), np.int32)
        lib().orc_negative_samples(self._buf, n_products, pairs.ctypes.data, len(pairs),
                                   anchors.ctypes.data, len(anchors), k, out.ctypes.data)
        return out
