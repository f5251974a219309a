// P2, exact: SimilarityDataset._get_negative_samples (src/data/data_loader.py:27-40) on the
// stream CPython's `random` module produces, so negative indices are bit-identical to the
// reference's for the same random.seed().  Host code (the stream is inherently sequential).
//
// CPython facts relied on (Modules/_randommodule.c, Lib/random.py, 3.10):
//   random.seed(int n)  -> init_by_array(32-bit little-endian digits of |n|)
//   getrandbits(k<=32)  -> genrand_uint32() >> (32-k)
//   choice(seq)         -> seq[_randbelow(len(seq))]; _randbelow(n): k = n.bit_length(),
//                          r = getrandbits(k) until r < n
//   shuffle(x)          -> for i = len-1 .. 1: j = _randbelow(i+1); swap
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/pcompanion_hip.h"

namespace {

class CPythonRandom {
public:
    void seed(uint64_t s) {
        std::vector<uint32_t> key;
        do { key.push_back((uint32_t)(s & 0xffffffffu)); s >>= 32; } while (s);
        fill_linear(19650218u);
        size_t i = 1, j = 0;
        for (size_t n = kN > key.size() ? kN : key.size(); n; --n) {
            st_[i] = (st_[i] ^ (mix(st_[i - 1]) * 1664525u)) + key[j] + (uint32_t)j;
            if (++i == kN) { st_[0] = st_[kN - 1]; i = 1; }
            if (++j == key.size()) j = 0;
        }
        for (size_t n = kN - 1; n; --n) {
            st_[i] = (st_[i] ^ (mix(st_[i - 1]) * 1566083941u)) - (uint32_t)i;
            if (++i == kN) { st_[0] = st_[kN - 1]; i = 1; }
        }
        st_[0] = 0x80000000u;
        at_ = kN;
    }

    uint32_t word() {
        if (at_ == kN) regenerate();
        uint32_t y = st_[at_++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        return y ^ (y >> 18);
    }

    uint32_t bits(int k) { return word() >> (32 - k); }

    uint64_t below(uint64_t n) {
        int k = 0;
        for (uint64_t t = n; t; t >>= 1) ++k;
        for (;;) {
            uint64_t r;
            if (k <= 32) r = bits(k);
            else { uint64_t lo = word(); r = lo | ((uint64_t)(word() >> (64 - k)) << 32); }
            if (r < n) return r;
        }
    }

private:
    static constexpr size_t kN = 624, kM = 397;
    uint32_t st_[kN];
    size_t at_;

    static uint32_t mix(uint32_t x) { return x ^ (x >> 30); }
    void fill_linear(uint32_t s) {
        st_[0] = s;
        for (size_t i = 1; i < kN; ++i) st_[i] = 1812433253u * mix(st_[i - 1]) + (uint32_t)i;
    }
    static uint32_t twist(uint32_t u, uint32_t v) {
        uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
        return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
    }
    void regenerate() {
        for (size_t i = 0; i < kN; ++i) st_[i] = st_[(i + kM) % kN] ^ twist(st_[i], st_[(i + 1) % kN]);
        at_ = 0;
    }
};

}  // namespace

extern "C" size_t pc_mt_state_bytes(void) { return sizeof(CPythonRandom); }

extern "C" int pc_mt_seed(void* state, uint64_t seed) {
    if (!state) return PC_EINVAL;
    static_cast<CPythonRandom*>(state)->seed(seed);
    return PC_OK;
}

extern "C" uint32_t pc_mt_getrandbits(void* state, int k) {
    if (!state || k < 1 || k > 32) return 0;
    return static_cast<CPythonRandom*>(state)->bits(k);
}

extern "C" uint64_t pc_mt_randbelow(void* state, uint64_t n) {
    if (!state || n == 0) return 0;
    return static_cast<CPythonRandom*>(state)->below(n);
}

extern "C" int pc_mt_shuffle(void* state, int64_t* perm, int64_t n) {
    if (!state || (!perm && n > 0) || n < 0) return PC_EINVAL;
    CPythonRandom* r = static_cast<CPythonRandom*>(state);
    for (int64_t i = n - 1; i >= 1; --i) {
        int64_t j = (int64_t)r->below((uint64_t)i + 1);
        int64_t t = perm[i]; perm[i] = perm[j]; perm[j] = t;
    }
    return PC_OK;
}

// anchors[n] -> out[n][k].  The anchor's positives come from the similarity CSR (the set
// {pair[1] : pair[0] == anchor} of data_loader.py:31); candidates are product indices
// 0..n_products-1 in bpg.nodes insertion order (synthetic_data.py:42-43,84-85).
extern "C" int pc_mt_negative_samples(void* state, int32_t n_products, const int32_t* sim_rowptr,
                                      const int32_t* sim_col, const int32_t* anchors, int64_t n, int k,
                                      int32_t* out) {
    if (!state || !sim_rowptr || !sim_col || !anchors || !out || n < 0 || k <= 0) return PC_EINVAL;
    if (n_products <= 0) return PC_EINVAL;
    CPythonRandom* r = static_cast<CPythonRandom*>(state);
    for (int64_t s = 0; s < n; ++s) {
        const int32_t anchor = anchors[s];
        if (anchor < 0 || anchor >= n_products) return PC_EINVAL;
        const int32_t lo = sim_rowptr[anchor], hi = sim_rowptr[anchor + 1];
        if ((int64_t)n_products - 1 - (hi - lo) < k) return PC_ESHAPE;   // the reference would loop forever
        int32_t* row = out + s * k;
        int have = 0;
        while (have < k) {
            const int32_t cand = (int32_t)r->below((uint64_t)n_products);
            bool reject = cand == anchor;
            for (int32_t e = lo; !reject && e < hi; ++e) reject = sim_col[e] == cand;
            for (int j = 0; !reject && j < have; ++j) reject = row[j] == cand;
            if (!reject) row[have++] = cand;
        }
    }
    return PC_OK;
}
