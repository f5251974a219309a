// The synthetic catalogue ON THE DEVICE (SyntheticDataGenerator, src/data/synthetic_data.py:11-153, restated scalably).
//
// The reference enumerates all P(P-1)/2 product pairs (synthetic_data.py:94-98) and cannot go past ~10 k products;
// data.generate_scaled_bpg restates its distributions per source node on the HOST (whole numpy arrays: fine at 100 k,
// impossible at BASELINE configs[3]/[4]: 10 M and 100 M products).  These kernels draw the same distributions straight
// into HBM, one Philox4x32-10 stream per (seed; product, purpose), so that
//   * every array is a pure function of (seed, product id): any row range / any rank::world shard of the feature table
//     can be generated alone, and the replicated graph arrays come out identical on every rank;
//   * nothing is ever materialised on the host.
// Distributions (synthetic_data.py lines in brackets):
//   type      uniform over num_types = 5 categories x num_types/5                                   [35-46]
//   features  N(0,1)^dim, + 1.0 on dims [20c, 20c + 20) of category c                               [50-52]
//   co-view   out-degree Poisson(2 mean (1 - u)), u ~ U(0,1), capped; targets uniform over the other products, kept with
//             probability 1 (same category) / 2/3 (different) -- the 1.5x of [107-108]; distinct within a row
//   similarity pair = co-view edge with purchase-after-view (0.2) and without co-purchase (0.075 same / 0.15)  [110-121]
//   complementary pair = Poisson(4.5) targets per product, kept 0.5 (same) / 1.0 (different), not co-viewed    [122-127]
#include "common.h"

enum { GEN_TYPE = 1, GEN_FEAT = 2, GEN_DEG = 3, GEN_EDGE = 4, GEN_COMP = 5 };
#define GEN_CAP_MAX 64

__device__ __forceinline__ int gen_type_of(uint64_t seed, uint32_t product, int num_types) {
    Philox r(seed, GEN_TYPE, product);
    return (int)r.below((uint32_t)num_types);
}
__device__ __forceinline__ int gen_category(int type, int num_types) {
    const int per = num_types / 5 > 0 ? num_types / 5 : 1;
    const int c = type / per;
    return c < 4 ? c : 4;
}
__device__ __forceinline__ double gen_u01(Philox& r) {       // (0,1), 53-bit-free: two words
    const uint32_t a = r.next(), b = r.next();
    return (((double)a * 4294967296.0 + (double)b) + 0.5) * (1.0 / 18446744073709551616.0);
}
// Poisson(lam) by inversion (sequential search; lam <= 64 here), fp64: the tail terms of lam = 32 are ~1e-14
__device__ __forceinline__ int gen_poisson(Philox& r, double lam) {
    const double u = gen_u01(r);
    double p = exp(-lam), cdf = p;
    int k = 0;
    while (u > cdf && k < 1000) { k++; p *= lam / (double)k; cdf += p; }
    return k;
}

__global__ void gen_types_kernel(uint32_t P, int num_types, uint64_t seed, int32_t* type_idx) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) type_idx[i] = gen_type_of(seed, i, num_types);
}

// one thread per 16-B chunk of a feature row; row k of the output is product first + k * stride
__global__ void gen_features_kernel(uint32_t first, uint32_t stride, uint32_t n_local, int dim, int num_types, uint64_t seed,
                                    float* features) {
    const int cpr = dim / 4;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n_local * cpr) return;
    const uint32_t k = (uint32_t)(t / cpr), c = (uint32_t)(t % cpr);
    const uint32_t prod = first + k * stride;
    uint32_t x0 = c, x1 = prod, x2 = GEN_FEAT, x3 = 0, k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
        const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1;
        const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float2 n0 = pc_box_muller(x0, x1), n1 = pc_box_muller(x2, x3);
    float v[4] = {n0.x, n0.y, n1.x, n1.y};
    const int cat = gen_category(gen_type_of(seed, prod, num_types), num_types);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int d = 4 * (int)c + q;
        if (d >= 20 * cat && d < 20 * cat + 20) v[q] += 1.0f;
    }
    *reinterpret_cast<float4*>(features + (size_t)k * dim + 4 * c) = make_float4(v[0], v[1], v[2], v[3]);
}

// co-view out-degree and the number of complementary candidates of every product
__global__ void gen_degrees_kernel(uint32_t P, double mean_degree, int cap, double comp_mean, uint64_t seed, int32_t* deg,
                                   int32_t* comp_cand) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    Philox r(seed, GEN_DEG, i);
    const double u = gen_u01(r);
    int d = gen_poisson(r, 2.0 * mean_degree * (1.0 - u));
    d = d < cap ? d : cap;
    if ((uint32_t)d > P - 1) d = (int)(P - 1);
    deg[i] = d;
    if (comp_cand) {
        int c = gen_poisson(r, comp_mean);
        comp_cand[i] = c < GEN_CAP_MAX ? c : GEN_CAP_MAX;
    }
}

// ---- exclusive scan of int32 counts -> int32 offsets [n + 1] (three launches: chunk sums, scan of the sums, offsets)
#define SCAN_CHUNK 4096
__global__ __launch_bounds__(1024) void scan_chunk_sums_kernel(const int32_t* in, uint32_t n, long long* sums) {
    __shared__ long long red[16];
    const uint32_t base = blockIdx.x * SCAN_CHUNK + threadIdx.x * 4;
    long long s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) s += base + i < n ? in[base + i] : 0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int i = 0; i < 16; i++) t += red[i];
        sums[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(long long* sums, int nblocks, long long* total) {
    __shared__ long long part[1024];
    const int t = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int lo = t * per, hi = min(nblocks, lo + per);
    long long s = 0;
    for (int i = lo; i < hi; i++) s += sums[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const long long v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    long long run = part[t] - s;
    for (int i = lo; i < hi; i++) { const long long v = sums[i]; sums[i] = run; run += v; }
    if (t == 1023) *total = part[1023];
}
__global__ __launch_bounds__(1024) void scan_offsets_kernel(const int32_t* in, uint32_t n, const long long* sums, int32_t* out) {
    __shared__ int wsum[16];
    const uint32_t base = blockIdx.x * SCAN_CHUNK + threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    int inc = s;                                             // inclusive scan of the thread sums inside the wave
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < w; i++) woff += wsum[i];
    long long run = sums[blockIdx.x] + woff + (inc - s);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (base + i < n) out[base + i] = (int32_t)run;
        run += v[i];
    }
    if (base <= n && n < base + 4) {                        // the thread whose range holds position n writes out[n] = total
        long long r2 = sums[blockIdx.x] + woff + (inc - s);
        for (uint32_t i = base; i < n; i++) r2 += v[i - base];
        out[n] = (int32_t)r2;
    }
}

extern "C" size_t pc_scan_scratch_bytes(int64_t n) {
    if (n <= 0) return 0;
    return (size_t)((n + SCAN_CHUNK) / SCAN_CHUNK + 2) * sizeof(long long);
}

// out[i] = sum_{j<i} in[j], i = 0..n (out[n] = total).  total_out (device int64, optional).  The total must fit int32.
extern "C" int pc_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int64_t* total_out, void* scratch,
                                     size_t scratch_bytes, void* stream) {
    if (!in || !out || n <= 0 || n >= (1ll << 32) - SCAN_CHUNK || !scratch) return PC_EINVAL;
    if (scratch_bytes < pc_scan_scratch_bytes(n)) return PC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    // chunks cover positions 0..n (one more than the input) so that some thread owns out[n]
    const int nblk = (int)((n + SCAN_CHUNK) / SCAN_CHUNK);
    long long* sums = (long long*)scratch;
    long long* total = total_out ? (long long*)total_out : sums + nblk + 1;
    PC_LAUNCH(scan_chunk_sums_kernel, dim3(nblk), dim3(1024), 0, st, in, (uint32_t)n, sums);
    PC_LAUNCH(scan_sums_kernel, dim3(1), dim3(1024), 0, st, sums, nblk, total);
    PC_LAUNCH(scan_offsets_kernel, dim3(nblk), dim3(1024), 0, st, in, (uint32_t)n, sums, out);
    return pc_launch_status();
}

// ---- co-view rows: one thread per source product (degree <= cap <= 64: the row is checked linearly for repeats).
// cv_col[rowptr[i] + j] = target | (similarity flag << 31); sim_count[i] = number of flagged edges.
__global__ void gen_coview_kernel(uint32_t P, int num_types, uint64_t seed, const int32_t* rowptr, int32_t* cv_col,
                                  int32_t* sim_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int lo = rowptr[i], deg = rowptr[i + 1] - lo;
    const int cat = gen_category(gen_type_of(seed, i, num_types), num_types);
    Philox r(seed, GEN_EDGE, i);
    int nsim = 0;
    for (int j = 0; j < deg; j++) {
        uint32_t t;
        bool same;
        while (true) {
            t = r.below(P);
            if (t == i) continue;
            same = gen_category(gen_type_of(seed, t, num_types), num_types) == cat;
            if (!same && r.next() >= 2863311531u) continue;           // keep with probability 2/3
            bool dup = false;
            for (int q = 0; q < j && !dup; q++) dup = ((uint32_t)cv_col[lo + q] & 0x7fffffffu) == t;
            if (!dup) break;
        }
        const bool pav = r.next() < 858993459u;                        // 0.2
        const bool cp = r.next() < (same ? 322122547u : 644245094u);   // 0.075 / 0.15
        const bool sim = pav && !cp;
        nsim += sim ? 1 : 0;
        cv_col[lo + j] = (int32_t)(t | (sim ? 0x80000000u : 0u));
    }
    sim_count[i] = nsim;
}

// similarity pairs in source order (= the positives' CSR): clears the flag bits of cv_col on the way
__global__ void gen_similarity_kernel(uint32_t P, const int32_t* rowptr, int32_t* cv_col, const int32_t* sim_rowptr,
                                      int32_t* sim_pairs, int32_t* sim_col, int32_t* pair_deg) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int lo = rowptr[i], hi = rowptr[i + 1];
    int k = sim_rowptr[i];
    for (int e = lo; e < hi; e++) {
        const uint32_t v = (uint32_t)cv_col[e];
        const int t = (int)(v & 0x7fffffffu);
        if (v & 0x80000000u) {
            sim_pairs[2 * (size_t)k] = (int32_t)i; sim_pairs[2 * (size_t)k + 1] = t;
            sim_col[k] = t;
            if (pair_deg) pair_deg[k] = hi - lo;
            k++;
            cv_col[e] = t;
        }
    }
}

// complementary pairs: count pass (out == NULL) and write pass regenerate the same draws
__global__ void gen_complementary_kernel(uint32_t P, int num_types, uint64_t seed, const int32_t* comp_cand,
                                         const int32_t* cv_rowptr, const int32_t* cv_col, int32_t* count,
                                         const int32_t* comp_rowptr, int32_t* comp_pairs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int n = comp_cand[i];
    const int lo = cv_rowptr[i], hi = cv_rowptr[i + 1];
    const int cat = gen_category(gen_type_of(seed, i, num_types), num_types);
    Philox r(seed, GEN_COMP, i);
    uint32_t got[GEN_CAP_MAX];
    int k = 0;
    for (int j = 0; j < n; j++) {
        uint32_t t;
        while (true) {
            t = r.below(P);
            if (t == i) continue;
            const bool same = gen_category(gen_type_of(seed, t, num_types), num_types) == cat;
            if (same && (r.next() & 1u)) continue;                    // keep with probability 0.5
            break;
        }
        bool drop = false;                                             // not co-viewed, no repeats (np.unique / setdiff1d)
        for (int e = lo; e < hi && !drop; e++) drop = ((uint32_t)cv_col[e] & 0x7fffffffu) == t;
        for (int q = 0; q < k && !drop; q++) drop = got[q] == t;
        if (!drop) got[k++] = t;
    }
    if (count) count[i] = k;
    if (comp_pairs) {
        const int base = comp_rowptr[i];
        for (int q = 0; q < k; q++) { comp_pairs[2 * (size_t)(base + q)] = (int32_t)i; comp_pairs[2 * (size_t)(base + q) + 1] = (int32_t)got[q]; }
    }
}

extern "C" int pc_gen_types(int64_t n_products, int num_types, uint64_t seed, int32_t* type_idx, void* stream) {
    if (n_products <= 0 || n_products >= (1ll << 31) || num_types <= 0 || !type_idx) return PC_EINVAL;
    PC_LAUNCH(gen_types_kernel, dim3((unsigned)((n_products + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint32_t)n_products,
              num_types, seed, type_idx);
    return pc_launch_status();
}

extern "C" int pc_gen_features(int64_t first, int64_t stride, int64_t n_local, int dim, int num_types, uint64_t seed,
                               float* features, void* stream) {
    if (first < 0 || stride <= 0 || n_local <= 0 || first + (n_local - 1) * stride >= (1ll << 31) || !features) return PC_EINVAL;
    if (dim < 100 || dim % 4 || num_types <= 0) return PC_ESHAPE;     // the category block spans dims [0, 100)
    const uint64_t total = (uint64_t)n_local * (dim / 4);
    PC_LAUNCH(gen_features_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint32_t)first,
              (uint32_t)stride, (uint32_t)n_local, dim, num_types, seed, features);
    return pc_launch_status();
}

extern "C" int pc_gen_degrees(int64_t n_products, double mean_degree, int degree_cap, double comp_mean, uint64_t seed,
                              int32_t* deg, int32_t* comp_cand, void* stream) {
    if (n_products <= 1 || n_products >= (1ll << 31) || !deg || mean_degree <= 0 || mean_degree > 32 || degree_cap < 1 ||
        degree_cap > GEN_CAP_MAX || comp_mean < 0 || comp_mean > 32)
        return PC_EINVAL;
    PC_LAUNCH(gen_degrees_kernel, dim3((unsigned)((n_products + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint32_t)n_products,
              mean_degree, degree_cap, comp_mean, seed, deg, comp_cand);
    return pc_launch_status();
}

extern "C" int pc_gen_coview(int64_t n_products, int num_types, uint64_t seed, const int32_t* cv_rowptr, int32_t* cv_col,
                             int32_t* sim_count, void* stream) {
    if (n_products <= 1 || n_products >= (1ll << 31) || num_types <= 0 || !cv_rowptr || !cv_col || !sim_count) return PC_EINVAL;
    PC_LAUNCH(gen_coview_kernel, dim3((unsigned)((n_products + 127) / 128)), dim3(128), 0, (hipStream_t)stream, (uint32_t)n_products,
              num_types, seed, cv_rowptr, cv_col, sim_count);
    return pc_launch_status();
}

extern "C" int pc_gen_similarity(int64_t n_products, const int32_t* cv_rowptr, int32_t* cv_col, const int32_t* sim_rowptr,
                                 int32_t* sim_pairs, int32_t* sim_col, int32_t* pair_deg, void* stream) {
    if (n_products <= 1 || n_products >= (1ll << 31) || !cv_rowptr || !cv_col || !sim_rowptr || !sim_pairs || !sim_col) return PC_EINVAL;
    PC_LAUNCH(gen_similarity_kernel, dim3((unsigned)((n_products + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
              (uint32_t)n_products, cv_rowptr, cv_col, sim_rowptr, sim_pairs, sim_col, pair_deg);
    return pc_launch_status();
}

extern "C" int pc_gen_complementary(int64_t n_products, int num_types, uint64_t seed, const int32_t* comp_cand,
                                    const int32_t* cv_rowptr, const int32_t* cv_col, int32_t* count,
                                    const int32_t* comp_rowptr, int32_t* comp_pairs, void* stream) {
    if (n_products <= 1 || n_products >= (1ll << 31) || num_types <= 0 || !comp_cand || !cv_rowptr || !cv_col) return PC_EINVAL;
    if (!count && !comp_pairs) return PC_EINVAL;
    if (comp_pairs && !comp_rowptr) return PC_EINVAL;
    PC_LAUNCH(gen_complementary_kernel, dim3((unsigned)((n_products + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
              (uint32_t)n_products, num_types, seed, comp_cand, cv_rowptr, cv_col, count, comp_rowptr, comp_pairs);
    return pc_launch_status();
}
