// P1-P4 on device: index-form batch construction for the Product2Vec loop.
//   (anchor, positive) = sim_pairs[pair_id]                      data_loader.py:46
//   neighbours = co-view CSR row of the anchor, -1 padded        bpg.py:24-38, data_loader.py:186-198
//   negatives  = k distinct uniform products, rejecting the anchor, the anchor's positives
//                and repeats                                     data_loader.py:27-40
// The reference draws from CPython's sequential MT19937 stream (host path: host_mt.cpp, bit
// exact).  This is the throughput path: the same rejection rules on a counter-based
// Philox4x32-10 stream keyed by (seed; sample, step), so every sample draws independently.
#include "common.h"

__device__ __forceinline__ void pairs_negatives_body(const int32_t* pair_ids, int B, const int32_t* sim_pairs,
                                                     const int32_t* sim_rowptr, const int32_t* sim_col, int n_products,
                                                     int K, uint64_t seed, uint64_t step, int32_t* anchor_idx,
                                                     int32_t* positive_idx, int32_t* negative_idx, int b) {
    if (b >= B) return;
    const int pid = pair_ids[b];
    const int a = sim_pairs[2 * (size_t)pid], pos = sim_pairs[2 * (size_t)pid + 1];      // (pair counts reach 2^30 at 100 M+ products)
    anchor_idx[b] = a;
    positive_idx[b] = pos;
    const int lo = sim_rowptr[a], hi = sim_rowptr[a + 1];
    Philox rng(seed, step, (uint32_t)b);
    // the anchor's positives (<= 8 of them: nearly always) and the negatives drawn so far live in registers: the loop used to
    // re-read both from global memory for every candidate -- a chain of dependent L2 round trips per draw (same draws, same
    // decisions: the stream and the rejection rules are untouched)
    constexpr int NPOS = 8, NNEG = 8;
    int pos_r[NPOS], neg_r[NNEG];
#pragma unroll
    for (int j = 0; j < NPOS; j++) pos_r[j] = lo + j < hi ? sim_col[lo + j] : -1;
#pragma unroll
    for (int j = 0; j < NNEG; j++) neg_r[j] = -1;
    int got = 0;
    while (got < K) {
        const int c = (int)rng.below((uint32_t)n_products);
        bool ok = c != a;
#pragma unroll
        for (int j = 0; j < NPOS; j++) ok = ok && pos_r[j] != c;
        for (int j = lo + NPOS; ok && j < hi; j++) ok = sim_col[j] != c;
        if (K <= NNEG) {
#pragma unroll
            for (int j = 0; j < NNEG; j++) ok = ok && neg_r[j] != c;
        } else {
            for (int j = 0; ok && j < got; j++) ok = negative_idx[(size_t)b * K + j] != c;
        }
        if (ok) {
            negative_idx[(size_t)b * K + got] = c;
#pragma unroll
            for (int j = 0; j < NNEG; j++) neg_r[j] = j == got ? c : neg_r[j];
            got++;
        }
    }
}

__global__ void build_pairs_negatives_kernel(const int32_t* pair_ids, int B, const int32_t* sim_pairs,
                                             const int32_t* sim_rowptr, const int32_t* sim_col, int n_products,
                                             int K, uint64_t seed, uint64_t step, int32_t* anchor_idx,
                                             int32_t* positive_idx, int32_t* negative_idx) {
    pairs_negatives_body(pair_ids, B, sim_pairs, sim_rowptr, sim_col, n_products, K, seed, step, anchor_idx, positive_idx,
                         negative_idx, blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ void build_neighbors_kernel(const int32_t* anchor_idx, int B, const int32_t* cv_rowptr,
                                       const int32_t* cv_col, int n_pad, int32_t* neighbor_idx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n_pad) return;
    const int b = t / n_pad, j = t % n_pad;
    const int a = anchor_idx[b];
    const int lo = cv_rowptr[a], deg = cv_rowptr[a + 1] - lo;
    neighbor_idx[t] = j < deg ? cv_col[lo + j] : -1;
}

// Compact neighbour layout: every real neighbour of the batch once, in slot order, then ONE row
// with index -1 standing for all the zero-padding slots.  row_off = exclusive scan of the (capped)
// degrees; single workgroup (B is a few thousand).
// (The degree reads are random accesses into a row-pointer array of up to 400 MB -- 100 M products -- where a dependent
// chain of them is a chain of TLB misses: one thread's anchors are read eight at a time, all sixteen row pointers in flight;
// the single-anchor loop this replaces took 236 us per batch at 100 M products.)
__global__ __launch_bounds__(1024) void degree_scan_kernel(const int32_t* anchor_idx, int B, const int32_t* cv_rowptr,
                                                           int n_pad, int32_t* row_off) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (B + 1023) / 1024;
    const int lo = t * per, hi = min(B, lo + per);
    auto degrees8 = [&](int b0, int (&d)[8]) {
        int a[8], r0[8], r1[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = b0 + u < hi ? anchor_idx[b0 + u] : anchor_idx[lo < B ? lo : 0];
#pragma unroll
        for (int u = 0; u < 8; u++) { r0[u] = cv_rowptr[a[u]]; r1[u] = cv_rowptr[a[u] + 1]; }
#pragma unroll
        for (int u = 0; u < 8; u++) d[u] = b0 + u < hi ? min(r1[u] - r0[u], n_pad) : 0;
    };
    int s = 0;
    for (int b0 = lo; b0 < hi; b0 += 8) {
        int d[8];
        degrees8(b0, d);
#pragma unroll
        for (int u = 0; u < 8; u++) s += d[u];
    }
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                 // Hillis-Steele inclusive scan of the partials
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;                               // exclusive prefix of this thread's chunk
    for (int b0 = lo; b0 < hi; b0 += 8) {
        int d[8];
        degrees8(b0, d);
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (b0 + u < hi) { row_off[b0 + u] = run; run += d[u]; }
    }
    if (t == 1023) row_off[B] = part[1023];
}

__global__ void build_neighbors_compact_kernel(const int32_t* anchor_idx, int B, const int32_t* cv_rowptr,
                                               const int32_t* cv_col, int n_pad, const int32_t* row_off,
                                               int32_t* nb_rows, int32_t* slot_row) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n_pad) return;
    const int b = t / n_pad, j = t % n_pad;
    const int a = anchor_idx[b];
    const int lo = cv_rowptr[a], deg = min(cv_rowptr[a + 1] - lo, n_pad);
    const int M = row_off[B];
    if (j < deg) {
        nb_rows[row_off[b] + j] = cv_col[lo + j];
        slot_row[t] = row_off[b] + j;
    } else {
        slot_row[t] = M;
    }
    if (t == 0) nb_rows[M] = -1;
}

extern "C" int pc_build_similarity_batch_compact(const int32_t* pair_ids, int batch, const int32_t* sim_pairs,
                                                 const int32_t* cv_rowptr, const int32_t* cv_col,
                                                 const int32_t* sim_rowptr, const int32_t* sim_col, int n_products,
                                                 int n_pad, int k_neg, uint64_t seed, uint64_t step,
                                                 int32_t* anchor_idx, int32_t* positive_idx, int32_t* negative_idx,
                                                 int32_t* nb_rows, int32_t* slot_row, int32_t* row_off, void* stream) {
    if (!pair_ids || !sim_pairs || !cv_rowptr || !cv_col || !sim_rowptr || !sim_col || !anchor_idx ||
        !positive_idx || !negative_idx || !nb_rows || !slot_row || !row_off)
        return PC_EINVAL;
    if (batch <= 0 || n_pad <= 0 || k_neg <= 0 || n_products <= k_neg + 1) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    PC_LAUNCH(build_pairs_negatives_kernel, dim3((batch + 127) / 128), dim3(128), 0, st, pair_ids, batch, sim_pairs,
              sim_rowptr, sim_col, n_products, k_neg, seed, step, anchor_idx, positive_idx, negative_idx);
    PC_TRY(pc_launch_status());
    PC_LAUNCH(degree_scan_kernel, dim3(1), dim3(1024), 0, st, anchor_idx, batch, cv_rowptr, n_pad, row_off);
    PC_TRY(pc_launch_status());
    const int total = batch * n_pad;
    PC_LAUNCH(build_neighbors_compact_kernel, dim3((total + 255) / 256), dim3(256), 0, st, anchor_idx, batch, cv_rowptr,
              cv_col, n_pad, row_off, nb_rows, slot_row);
    return pc_launch_status();
}

// ---------------------------------------------------------------------------------------
// Unique neighbour rows.  The co-view neighbours of a batch repeat (35 % of the 88 k real slots of a
// 4096-anchor batch over 100 k products): identical table rows give identical FFN rows inside one
// BatchNorm call, so -- exactly like the zero-padding rows -- each distinct product is carried ONCE with
// its multiplicity (BatchNorm weight, and the sum of its slots' gradients on the way back).  All integer
// work, deterministic: rows come out in ascending product order.
//   cnt[P]   occurrences (zero between calls: this sequence clears what it touched)
//   rank[P]  row of a present product
#define UQ_CHUNK 4096
// The negatives sampler and the occurrence count are independent (the count reads the anchors straight from the pair list):
// ONE launch, workgroups [0, pair_blocks) of 128 threads sample, the rest count (two slots per thread) -- a launch of its own
// costs ~4.5 us of latency, and every builder launch shares the chip with the training stream's persistent kernels.
__global__ __launch_bounds__(128) void uq_pairs_count_kernel(const int32_t* pair_ids, int B, const int32_t* sim_pairs,
                                                             const int32_t* sim_rowptr, const int32_t* sim_col, int n_products,
                                                             int K, uint64_t seed, uint64_t step, int32_t* anchor_idx,
                                                             int32_t* positive_idx, int32_t* negative_idx, int pair_blocks,
                                                             const int32_t* cv_rowptr, const int32_t* cv_col, int n_pad,
                                                             int32_t* cnt) {
    if ((int)blockIdx.x < pair_blocks) {
        pairs_negatives_body(pair_ids, B, sim_pairs, sim_rowptr, sim_col, n_products, K, seed, step, anchor_idx, positive_idx,
                             negative_idx, blockIdx.x * 128 + threadIdx.x);
        return;
    }
    const int t0 = (((int)blockIdx.x - pair_blocks) * 128 + threadIdx.x) * 2;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int t = t0 + u;
        if (t >= B * n_pad) return;
        const int b = t / n_pad, j = t % n_pad;
        const int a = sim_pairs[2 * (size_t)pair_ids[b]];
        const int lo = cv_rowptr[a], deg = min(cv_rowptr[a + 1] - lo, n_pad);
        if (j < deg) atomicAdd(&cnt[cv_col[lo + j]], 1);
    }
}

// per chunk of products: (number present, number of slots) -> blocksum[2 * blk], blocksum[2 * blk + 1]
// THREADS = 1024 (chunks of 4096 products) or 256 (chunks of 1024): the builders run BESIDE the training stream's persistent
// kernels, and a 1024-thread workgroup takes sixteen wave slots of ONE CU for its whole life, a 256-thread one four
template <int THREADS>
__global__ __launch_bounds__(THREADS) void uq_block_sums_kernel(const int32_t* cnt, int P, int32_t* blocksum) {
    constexpr int NWV = THREADS / 64;
    __shared__ int red[2][NWV];
    const int base = blockIdx.x * (4 * THREADS) + threadIdx.x * 4;
    int s = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { const int v = base + i < P ? cnt[base + i] : 0; s += v > 0 ? 1 : 0; c += v; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int ts = 0, tc = 0;
        for (int i = 0; i < NWV; i++) { ts += red[0][i]; tc += red[1][i]; }
        blocksum[2 * blockIdx.x] = ts;
        blocksum[2 * blockIdx.x + 1] = tc;
    }
}

// row of every present product (ascending product order), its multiplicity, and the start of its slot list
// blocksum: the per-chunk sums of uq_block_sums_kernel (NOT scanned): every workgroup adds up its predecessors' sums itself
// (25 chunks at 100 k products; a scan launch of its own cost more than these few loads); the last one writes n_unique
template <int THREADS>
__global__ __launch_bounds__(THREADS) void uq_assign_kernel(const int32_t* cnt, int P, const int32_t* blocksum, int32_t* rank,
                                                         int32_t* nb_rows, float* nb_weight, int32_t* ref_off,
                                                         int32_t* cursor, int32_t* n_unique) {
    constexpr int NWV = THREADS / 64;
    __shared__ int wsum[2][NWV];
    __shared__ int boff[2];
    {
        int ps = 0, pc = 0;
        for (int i = threadIdx.x; i < (int)blockIdx.x; i += THREADS) { ps += blocksum[2 * i]; pc += blocksum[2 * i + 1]; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { ps += __shfl_xor(ps, o, 64); pc += __shfl_xor(pc, o, 64); }
        if ((threadIdx.x & 63) == 0) { wsum[0][threadIdx.x >> 6] = ps; wsum[1][threadIdx.x >> 6] = pc; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int a = 0, b = 0;
            for (int i = 0; i < NWV; i++) { a += wsum[0][i]; b += wsum[1][i]; }
            boff[0] = a; boff[1] = b;
            if (blockIdx.x == gridDim.x - 1) n_unique[0] = a + blocksum[2 * blockIdx.x];
        }
        __syncthreads();
    }
    const int base = blockIdx.x * (4 * THREADS) + threadIdx.x * 4;
    int f[4], c[4], s = 0, cs = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { c[i] = base + i < P ? cnt[base + i] : 0; f[i] = c[i] > 0; s += f[i]; cs += c[i]; }
    // exclusive scans of the per-thread counts: within the wave by shuffles, across the 16 waves through LDS
    int incl = s, inclc = cs;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64), u = __shfl_up(inclc, o, 64);
        if ((threadIdx.x & 63) >= o) { incl += v; inclc += u; }
    }
    if ((threadIdx.x & 63) == 63) { wsum[0][threadIdx.x >> 6] = incl; wsum[1][threadIdx.x >> 6] = inclc; }     // (reused: the barrier above)
    __syncthreads();
    int woff = 0, woffc = 0;
    for (int i = 0; i < (int)(threadIdx.x >> 6); i++) { woff += wsum[0][i]; woffc += wsum[1][i]; }
    int r = boff[0] + woff + incl - s;
    int ro = boff[1] + woffc + inclc - cs;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (f[i]) {
            rank[base + i] = r; nb_rows[r] = base + i; nb_weight[r] = (float)c[i]; ref_off[r] = ro; cursor[r] = 0;
            r++; ro += c[i];
        }
}

__global__ void uq_slots_kernel(const int32_t* anchor_idx, int B, const int32_t* cv_rowptr, const int32_t* cv_col,
                                int n_pad, const int32_t* rank, const int32_t* n_unique, int n_real, int32_t* cnt,
                                int32_t* nb_rows, float* nb_weight, int32_t* slot_row, int32_t* ref_off,
                                int32_t* cursor, int32_t* ref_slot) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * n_pad) return;
    const int b = t / n_pad, j = t % n_pad;
    const int a = anchor_idx[b];
    const int lo = cv_rowptr[a], deg = min(cv_rowptr[a + 1] - lo, n_pad);
    const int U = n_unique[0];
    int row = U;
    if (j < deg) {
        const int pid = cv_col[lo + j];
        row = rank[pid];
        cnt[pid] = 0;                                       // leave the counters zeroed for the next batch
        // the row's slot list fills in arrival order (integer atomics); its consumer sorts the handful of
        // entries of a row before summing, so the result does not depend on that order
        ref_slot[ref_off[row] + atomicAdd(&cursor[row], 1)] = t;
    }
    slot_row[t] = row;
    if (t == 0) { nb_rows[U] = -1; nb_weight[U] = (float)(B * n_pad - n_real); ref_off[U] = n_real; ref_off[U + 1] = n_real; }
}

// layout: cnt[P] | rank[P] | blocksum[2 * nblk] | cursor[S]      (S = slots: an upper bound of the row count)
// chunk of the per-product scans: 1024 products (256-thread workgroups) up to UQ_SMALL_MAX products, 4096 beyond (every
// workgroup adds up its predecessors' chunk sums itself: quadratic in the chunk count)
#ifndef UQ_SMALL_MAX
#define UQ_SMALL_MAX (1 << 20)
#endif
static inline int uq_chunk(int n_products) { return n_products <= UQ_SMALL_MAX ? 1024 : UQ_CHUNK; }
extern "C" size_t pc_build_similarity_batch_unique_scratch_bytes(int n_products, int max_slots) {
    if (n_products <= 0 || max_slots <= 0) return 0;
    const size_t nblk = ((size_t)n_products + 1023) / 1024;         // (the smaller chunk: an upper bound for either)
    return ((size_t)n_products * 2 + 2 * nblk + (size_t)max_slots + 1) * sizeof(int32_t);
}

extern "C" int pc_build_similarity_batch_unique(const int32_t* pair_ids, int batch, const int32_t* sim_pairs,
                                                const int32_t* cv_rowptr, const int32_t* cv_col,
                                                const int32_t* sim_rowptr, const int32_t* sim_col, int n_products,
                                                int n_pad, int k_neg, uint64_t seed, uint64_t step, int n_real,
                                                int32_t* anchor_idx, int32_t* positive_idx, int32_t* negative_idx,
                                                int32_t* nb_rows, float* nb_weight, int32_t* slot_row,
                                                int32_t* ref_off, int32_t* ref_slot, int32_t* n_unique, void* scratch,
                                                size_t scratch_bytes, void* stream) {
    if (!pair_ids || !sim_pairs || !cv_rowptr || !cv_col || !sim_rowptr || !sim_col || !anchor_idx ||
        !positive_idx || !negative_idx || !nb_rows || !nb_weight || !slot_row || !ref_off || !ref_slot || !n_unique ||
        !scratch)
        return PC_EINVAL;
    if (batch <= 0 || n_pad <= 0 || k_neg <= 0 || n_products <= k_neg + 1 || n_real < 0 || n_real > batch * n_pad)
        return PC_EINVAL;
    const int chunk = uq_chunk(n_products);
    const int nblk = (n_products + chunk - 1) / chunk;
    if (nblk > 4096) return PC_ESHAPE;                        // 16.7 M products with one scan workgroup
    const int total = batch * n_pad;
    if (scratch_bytes < pc_build_similarity_batch_unique_scratch_bytes(n_products, total)) return PC_EWORKSPACE;
    int32_t* cnt = (int32_t*)scratch;
    int32_t* rank = cnt + n_products;
    int32_t* blocksum = rank + n_products;
    int32_t* cursor = blocksum + 2 * (((size_t)n_products + 1023) / 1024);      // (the layout of ..._scratch_bytes)
    hipStream_t st = (hipStream_t)stream;
    // four launches (round 2: six): sampler ∥ occurrence count, per-chunk sums, row assignment (own prefix of the sums), slots
    const int pair_blocks = (batch + 127) / 128;
    PC_LAUNCH(uq_pairs_count_kernel, dim3(pair_blocks + (total + 255) / 256), dim3(128), 0, st, pair_ids, batch, sim_pairs,
              sim_rowptr, sim_col, n_products, k_neg, seed, step, anchor_idx, positive_idx, negative_idx, pair_blocks, cv_rowptr,
              cv_col, n_pad, cnt);
    PC_TRY(pc_launch_status());
    if (chunk == 1024) {
        PC_LAUNCH(uq_block_sums_kernel<256>, dim3(nblk), dim3(256), 0, st, cnt, n_products, blocksum);
        PC_LAUNCH(uq_assign_kernel<256>, dim3(nblk), dim3(256), 0, st, cnt, n_products, blocksum, rank, nb_rows, nb_weight, ref_off,
                  cursor, n_unique);
    } else {
        PC_LAUNCH(uq_block_sums_kernel<1024>, dim3(nblk), dim3(1024), 0, st, cnt, n_products, blocksum);
        PC_LAUNCH(uq_assign_kernel<1024>, dim3(nblk), dim3(1024), 0, st, cnt, n_products, blocksum, rank, nb_rows, nb_weight, ref_off,
                  cursor, n_unique);
    }
    PC_LAUNCH(uq_slots_kernel, dim3((total + 255) / 256), dim3(256), 0, st, anchor_idx, batch, cv_rowptr, cv_col, n_pad,
              rank, n_unique, n_real, cnt, nb_rows, nb_weight, slot_row, ref_off, cursor, ref_slot);
    return pc_launch_status();
}

extern "C" int pc_build_similarity_batch(const int32_t* pair_ids, int batch, const int32_t* sim_pairs,
                                         const int32_t* cv_rowptr, const int32_t* cv_col,
                                         const int32_t* sim_rowptr, const int32_t* sim_col, int n_products,
                                         int n_pad, int k_neg, uint64_t seed, uint64_t step, int32_t* anchor_idx,
                                         int32_t* positive_idx, int32_t* negative_idx, int32_t* neighbor_idx,
                                         void* stream) {
    if (!pair_ids || !sim_pairs || !cv_rowptr || !cv_col || !sim_rowptr || !sim_col || !anchor_idx ||
        !positive_idx || !negative_idx)
        return PC_EINVAL;
    if (batch <= 0 || n_pad < 0 || k_neg <= 0 || n_products <= k_neg + 1) return PC_EINVAL;
    if (n_pad > 0 && !neighbor_idx) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    PC_LAUNCH(build_pairs_negatives_kernel, dim3((batch + 127) / 128), dim3(128), 0, st, pair_ids, batch,
                       sim_pairs, sim_rowptr, sim_col, n_products, k_neg, seed, step, anchor_idx, positive_idx,
                       negative_idx);
    PC_TRY(pc_launch_status());
    if (n_pad > 0) {
        const int total = batch * n_pad;
        PC_LAUNCH(build_neighbors_kernel, dim3((total + 255) / 256), dim3(256), 0, st, anchor_idx, batch,
                           cv_rowptr, cv_col, n_pad, neighbor_idx);
        PC_TRY(pc_launch_status());
    }
    return PC_OK;
}

// ---------------------------------------------------------------------------------------
// Zipf(s = 1) negatives (BASELINE configs[4]: "Zipf-skewed negative sampling"; no reference counterpart -- the
// reference draws uniformly, data_loader.py:34): candidate = perm[k - 1] with P(k) proportional to 1 / k over the
// popularity ranks k = 1..P, then the SAME rejection rules as data_loader.py:33-38 (not the anchor, not one of its
// positives, not drawn before).  Integer arithmetic only, so the host restatement (philox_oracle.zipf_negatives) is bit
// exact: the octaves [2^j, 2^(j+1)) carry nearly equal mass under 1 / k -- an octave is picked by comparing one 32-bit
// draw with the cumulative thresholds octave_cum[j] (computed once on the host), a rank inside it is proposed
// uniformly (j random bits) and accepted with probability 2^j / k (r * k < 2^j * 2^32 for a third draw r).
#define PC_ZIPF_MAX_TRIES 4096
__global__ void zipf_negatives_kernel(const int32_t* pair_ids, int B, const int32_t* sim_pairs, const int32_t* sim_rowptr,
                                      const int32_t* sim_col, int n_products, int K, uint64_t seed, uint64_t step,
                                      const uint32_t* octave_cum, int n_octaves, const int32_t* perm,
                                      int32_t* negative_idx, int32_t* failed) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int a = sim_pairs[2 * (size_t)pair_ids[b]];
    const int lo = sim_rowptr[a], hi = sim_rowptr[a + 1];
    Philox rng(seed ^ 0x5a495046ull, step, (uint32_t)b);          // its own stream ("ZIPF"), apart from the uniform sampler's
    int got = 0;
    // Every wave must leave: an anchor whose positives cover the head of the popularity order accepts rarely, and one with
    // fewer than K eligible products never.  After PC_ZIPF_MAX_TRIES proposals the remaining negatives are the first
    // eligible products in rank order (deterministic; counted in `failed` so that the caller can tell), -1 if none is left.
    int tries = 0;
    while (got < K) {
        if (++tries > PC_ZIPF_MAX_TRIES) {
            if (failed && got < K) atomicAdd(failed, 1);
            for (int k1 = 0; got < K && k1 < n_products; k1++) {
                const int c = perm ? perm[k1] : k1;
                bool ok = c != a;
                for (int q = lo; ok && q < hi; q++) ok = sim_col[q] != c;
                for (int q = 0; ok && q < got; q++) ok = negative_idx[(size_t)b * K + q] != c;
                if (ok) negative_idx[(size_t)b * K + got++] = c;
            }
            while (got < K) negative_idx[(size_t)b * K + got++] = -1;
            break;
        }
        const uint32_t r0 = rng.next();
        int j = 0;
        while (j + 1 < n_octaves && r0 > octave_cum[j]) j++;
        const uint32_t base = 1u << j;
        uint32_t k;
        while (true) {                                            // inside the chosen octave until a rank is accepted
            k = base + (j ? (rng.next() >> (32 - j)) : 0u);
            if (k > (uint32_t)n_products) continue;               // the last octave may be partial (it starts inside [1, P]: the host checks)
            const uint32_t r2 = rng.next();
            if ((uint64_t)r2 * k < ((uint64_t)base << 32)) break; // accept with probability 2^j / k
        }
        const int c = perm ? perm[k - 1] : (int)(k - 1);
        bool ok = c != a;
        for (int q = lo; ok && q < hi; q++) ok = sim_col[q] != c;
        for (int q = 0; ok && q < got; q++) ok = negative_idx[(size_t)b * K + q] != c;
        if (ok) negative_idx[(size_t)b * K + got++] = c;
    }
}

extern "C" int pc_sample_negatives_zipf(const int32_t* pair_ids, int batch, const int32_t* sim_pairs,
                                        const int32_t* sim_rowptr, const int32_t* sim_col, int n_products, int k_neg,
                                        uint64_t seed, uint64_t step, const uint32_t* octave_cum, int n_octaves,
                                        const int32_t* perm, int32_t* negative_idx, int32_t* failed, void* stream) {
    if (!pair_ids || !sim_pairs || !sim_rowptr || !sim_col || !octave_cum || !negative_idx) return PC_EINVAL;
    if (batch <= 0 || k_neg <= 0 || n_products <= k_neg + 1 || n_octaves < 1 || n_octaves > 31) return PC_EINVAL;
    if ((1ll << (n_octaves - 1)) > (long long)n_products) return PC_EINVAL;      // the last octave must start inside [1, P]
    PC_LAUNCH(zipf_negatives_kernel, dim3((batch + 127) / 128), dim3(128), 0, (hipStream_t)stream, pair_ids, batch, sim_pairs,
              sim_rowptr, sim_col, n_products, k_neg, seed, step, octave_cum, n_octaves, perm, negative_idx, failed);
    return pc_launch_status();
}

// ---------------------------------------------------------------------------------------
// J1 on device: ComplementaryDataset.__getitem__ + collate_fn (data_loader.py:133-157) for a batch
// of labelled pairs (query, target, label):
//   label +1: positive_types = t(target), negative_types = (t(target)+1) % n_types,
//             positive_items = feat(target), negative_items = N(0,1) filler
//   label -1: positive_types = 0, negative_types = t(target),
//             positive_items = N(0,1) filler, negative_items = feat(target)
// The filler is input DATA (torch.randn_like in the reference's worker): here Philox4x32-10 +
// Box-Muller keyed by (seed; row, 16-B chunk, step) (common.h pc_filler_chunk).  One thread per 16-B chunk of a row.
__global__ void build_complementary_batch_kernel(const int32_t* pairs, int B, const float* features,
                                                 const int32_t* type_idx, int n_types, uint64_t seed, uint64_t step,
                                                 int32_t* query_idx, int32_t* query_types, int32_t* pos_types,
                                                 int32_t* neg_types, float* pos_items, float* neg_items,
                                                 float* target_features) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * (PC_D / 4)) return;
    const int b = t / (PC_D / 4), c = t % (PC_D / 4);
    const int q = pairs[3 * b], tg = pairs[3 * b + 1], lab = pairs[3 * b + 2];
    const float4 f = *reinterpret_cast<const float4*>(features + (size_t)tg * PC_D + 4 * c);
    const float4 fill = pc_filler_chunk(seed, step, (uint32_t)t);
    const bool pos = lab == 1;
    *reinterpret_cast<float4*>(pos_items + (size_t)b * PC_D + 4 * c) = pos ? f : fill;
    *reinterpret_cast<float4*>(neg_items + (size_t)b * PC_D + 4 * c) = pos ? fill : f;
    if (target_features) *reinterpret_cast<float4*>(target_features + (size_t)b * PC_D + 4 * c) = f;
    if (c == 0) {
        const int tt = type_idx[tg];
        query_idx[b] = q;
        query_types[b] = type_idx[q];
        pos_types[b] = pos ? tt : 0;
        neg_types[b] = pos ? (tt + 1) % n_types : tt;
    }
}

extern "C" int pc_build_complementary_batch(const int32_t* pairs, int batch, const float* features,
                                            const int32_t* type_idx, int n_types, uint64_t seed, uint64_t step,
                                            int32_t* query_idx, int32_t* query_types, int32_t* pos_types,
                                            int32_t* neg_types, float* pos_items, float* neg_items,
                                            float* target_features, void* stream) {
    if (!pairs || !features || !type_idx || !query_idx || !query_types || !pos_types || !neg_types || !pos_items ||
        !neg_items || batch <= 0 || n_types <= 0)
        return PC_EINVAL;
    const int total = batch * (PC_D / 4);
    PC_LAUNCH(build_complementary_batch_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, pairs, batch,
              features, type_idx, n_types, seed, step, query_idx, query_types, pos_types, neg_types, pos_items,
              neg_items, target_features);
    return pc_launch_status();
}


// ---------------------------------------------------------------------------------------
// The epoch order of the loaders (DataLoader(shuffle=True): scripts/pretrain_product2vec.py:24-30, train.py:115-121)
// without a sort and without storage: a keyed bijection of [0, 2^k) (balanced Feistel network, six rounds, k = the even
// number of bits covering n) cycle-walked into [0, n) -- position i maps through the network until the value is < n.
// Deterministic in (seed, epoch), integer arithmetic only (restated for the tests by philox_oracle.epoch_permutation), and
// it replaces torch.randperm, whose radix / merge sort kernels were the last third-party kernels on the loader path.
struct FeistelKeys { uint32_t k[6]; int half; };     // half = k / 2 bits per side
static inline uint64_t pc_splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static FeistelKeys feistel_keys(uint64_t n, uint64_t seed, uint64_t epoch) {
    FeistelKeys f;
    int bits = 2;
    while (bits < 62 && (1ull << bits) < n) bits += 2;
    f.half = bits / 2;
    uint64_t x = seed * 0xD1342543DE82EF95ull + epoch * 0x2545F4914F6CDD1Dull + 0x1234567ull;
    for (int r = 0; r < 6; r++) f.k[r] = (uint32_t)(pc_splitmix64(x) >> 32);
    return f;
}
__device__ __forceinline__ uint32_t feistel_apply(uint32_t x, const FeistelKeys& f) {
    const uint32_t mask = (1u << f.half) - 1u;
    uint32_t L = x >> f.half, R = x & mask;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        uint32_t v = R * 0xCC9E2D51u + f.k[r];
        v ^= v >> 15; v *= 0x85EBCA6Bu; v ^= v >> 13; v *= 0xC2B2AE35u; v ^= v >> 16;
        const uint32_t t = L ^ (v & mask);
        L = R; R = t;
    }
    return (L << f.half) | R;
}
__device__ __forceinline__ uint32_t epoch_perm_at(uint32_t i, uint32_t n, const FeistelKeys& f) {
    uint32_t x = feistel_apply(i, f);
    while (x >= n) x = feistel_apply(x, f);          // the cycle of i returns to [0, n): at most 4 n / n expected steps
    return x;
}

__global__ void epoch_permutation_kernel(uint32_t n, FeistelKeys f, int32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)epoch_perm_at(i, n, f);
}

extern "C" int pc_epoch_permutation(int n, uint64_t seed, uint64_t epoch, int32_t* out, void* stream) {
    if (n <= 0 || !out) return PC_EINVAL;
    const FeistelKeys f = feistel_keys((uint64_t)n, seed, epoch);
    PC_LAUNCH(epoch_permutation_kernel, dim3(((unsigned)n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (uint32_t)n, f, out);
    return pc_launch_status();
}

// out[i][:] = rows[perm(i)][:], width int32 per row (the labelled pairs [n,3] of the complementary loader): the shuffled
// epoch in one launch
__global__ void shuffle_rows_kernel(const int32_t* rows, uint32_t n, int width, FeistelKeys f, int32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = epoch_perm_at(i, n, f);
    for (int c = 0; c < width; c++) out[(size_t)i * width + c] = rows[(size_t)s * width + c];
}

extern "C" int pc_shuffle_rows_i32(const int32_t* rows, int n, int width, uint64_t seed, uint64_t epoch, int32_t* out,
                                   void* stream) {
    if (n <= 0 || width <= 0 || !rows || !out || rows == out) return PC_EINVAL;
    const FeistelKeys f = feistel_keys((uint64_t)n, seed, epoch);
    PC_LAUNCH(shuffle_rows_kernel, dim3(((unsigned)n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rows, (uint32_t)n, width, f, out);
    return pc_launch_status();
}

// The two integers per batch the host needs to size it (collate_fn pads the neighbour lists to the batch maximum,
// data_loader.py:186-198): plan[b] = (max, sum) of deg[order[i]] over the batch's positions i in [b B, (b+1) B), i < n.
// order == NULL: the identity (shuffle = False).  One workgroup per batch.
__global__ __launch_bounds__(256) void epoch_plan_kernel(const int32_t* order, const int32_t* deg, int n, int B, long long* plan) {
    __shared__ int smx[256];
    __shared__ long long ssm[256];
    const int b = blockIdx.x;
    int mx = 0;
    long long sm = 0;
    const long long i_end = min((long long)(b + 1) * B, (long long)n);      // ((b + 1) * B overflows an int for n near 2^31)
    for (long long i = (long long)b * B + threadIdx.x; i < i_end; i += 256) {
        const int d = deg[order ? order[i] : i];
        mx = max(mx, d);
        sm += d;
    }
    smx[threadIdx.x] = mx; ssm[threadIdx.x] = sm;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) {
            smx[threadIdx.x] = max(smx[threadIdx.x], smx[threadIdx.x + o]);
            ssm[threadIdx.x] += ssm[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { plan[2 * b] = smx[0]; plan[2 * b + 1] = ssm[0]; }
}

extern "C" int pc_epoch_plan(const int32_t* order, const int32_t* deg, int n, int batch, int n_batches, int64_t* plan,
                             void* stream) {
    if (!deg || !plan || n <= 0 || batch <= 0 || n_batches <= 0) return PC_EINVAL;
    PC_LAUNCH(epoch_plan_kernel, dim3(n_batches), dim3(256), 0, (hipStream_t)stream, order, deg, n, batch, (long long*)plan);
    return pc_launch_status();
}
