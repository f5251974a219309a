// P6: Product2Vec.get_initial_embedding = Linear(128->256) -> BatchNorm1d -> tanh ->
// Linear(256->256) -> tanh -> Linear(256->128)  (product2vec.py:14-21,31-46), forward
// (train / eval) and backward, as three fused fp32-MFMA passes each way:
//
//   fwd 1  gather rows + Linear0 + per-tile BatchNorm partial sums        (gemm_nt, STAT_SUMSQ)
//          bn_finalize: fp64 fold of the partials -> mean/invstd/scale/shift, running stats
//   fwd 2  BN-apply + tanh on the fly -> Linear3 -> tanh                  (gemm_nt, PRO_BNTANH)
//   fwd 3  Linear5
//   bwd 1  dZ2 = (dY W5) * (1-A2^2)            ; dW5, db5                 (gemm_nt + gemm_tn)
//   bwd 2  dZ1 = (dZ2 W3) * (1-A1^2) + BN-backward partial sums ; dW3, db3 (A1 recomputed from H0)
//          bn_bwd_finalize -> dgamma, dbeta, per-segment coefficients
//   bwd 3  dH0 = BN backward (elementwise, in place) ; dW0, db0 with the row gather fused
//
// BatchNorm statistics are per SEGMENT (= per reference FFN call), see pc_segments.
#include "common.h"

int pc_opt_bn_finalize_side();     // (gemm_tn.hip: pc_set_option)

#define BN_EPS 1e-5f
#define BN_MOMENTUM 0.1f

// ---------------------------------------------------------------------------------------
// partial[t][j] over 128-row tiles -> per-segment statistics.  grid = H/32 blocks of
// (32 columns x 32 tile lanes): every column is independent, so the fold is spread over 8 CUs
// and each thread walks tiles/32 entries in two chains; fp64 accumulation.
#define FIN_COLS 32
#define FIN_LANES 32
__device__ __forceinline__ void fold_partials(const float* p1, const float* p2, int t0, int t1, int j, int q,
                                              double (*red)[FIN_LANES][FIN_COLS], double* o1, double* o2) {
    double a = 0.0, b = 0.0, a1 = 0.0, b1 = 0.0;
    int t = t0 + q;
    for (; t + FIN_LANES < t1; t += 2 * FIN_LANES) {
        a += (double)p1[(size_t)t * PC_H + j];                b += (double)p2[(size_t)t * PC_H + j];
        a1 += (double)p1[(size_t)(t + FIN_LANES) * PC_H + j]; b1 += (double)p2[(size_t)(t + FIN_LANES) * PC_H + j];
    }
    if (t < t1) { a += (double)p1[(size_t)t * PC_H + j]; b += (double)p2[(size_t)t * PC_H + j]; }
    red[0][q][threadIdx.x] = a + a1;
    red[1][q][threadIdx.x] = b + b1;
    __syncthreads();
    if (q == 0) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
        for (int i = 0; i < FIN_LANES; i++) { s1 += red[0][i][threadIdx.x]; s2 += red[1][i][threadIdx.x]; }
        *o1 = s1; *o2 = s2;
    }
    __syncthreads();
}

// All segments in ONE pass, 16-byte loads (round 6).  The folds above walk a segment's tiles with one dword load per lane and
// tile: a wave-load moves 256 B and the kernel is bound by the address pipe of its eight CUs, not by latency (5.6 us for 4
// tiles, 8.8 for 352, 13.1 for 704 when run alone -- scripts/dev/bn_finalize_probe.sh -- and 12-14 us inside the step, four
// segments one after the other).  Here a thread owns FOUR adjacent columns (one float4 per tile and array), a workgroup is
// 8 column groups x 64 tile lanes (512 threads: 128 registers per thread would spill the 64 accumulator registers' neighbours),
// so a wave-load moves 1 KB (eight tile rows x 128 B) and a 352-tile fold is six loads per lane and array, all in flight
// before the first use.  A value is added to its segment's accumulator by predicate (tiles never
// straddle segments).  Lanes are reduced in a fixed order: xor-shuffles over the wave's eight tile lanes, then one LDS exchange
// over the eight waves.  Result: thread f < 128 holds the two sums of (segment f / 32, column f % 32 of the workgroup's 32).
#define FIN4_CG 8
#define FIN4_LANES 64
#define FIN4_WAVES (FIN4_CG * FIN4_LANES / 64)
#define FIN4_BATCH 6
__device__ __forceinline__ void fold_partials_all4(const float* __restrict__ p1, const float* __restrict__ p2, const SegInfo& si,
                                                   int col0, double (*red)[2][PC_MAX_SEG][FIN_COLS], double* o1, double* o2) {
    const int cg = threadIdx.x, q = threadIdx.y;
    const int tid = q * FIN4_CG + cg, wave = tid >> 6, lane = tid & 63;
    double a[PC_MAX_SEG][4], b[PC_MAX_SEG][4];
#pragma unroll
    for (int s = 0; s < PC_MAX_SEG; s++)
#pragma unroll
        for (int c = 0; c < 4; c++) { a[s][c] = 0.0; b[s][c] = 0.0; }
    const int t_end = si.tile0[si.nseg];
    const int b1 = si.nseg > 1 ? si.tile0[1] : t_end, b2 = si.nseg > 2 ? si.tile0[2] : t_end, b3 = si.nseg > 3 ? si.tile0[3] : t_end;
    const size_t c0 = (size_t)col0 + cg * 4;
    for (int t = si.tile0[0] + q; t < t_end; t += FIN4_BATCH * FIN4_LANES) {
        float4 x[FIN4_BATCH], y[FIN4_BATCH];
#pragma unroll
        for (int u = 0; u < FIN4_BATCH; u++) {
            const int tt = t + u * FIN4_LANES;
            const int tc = tt < t_end ? tt : t;                    // (a lane past the end re-reads its first tile: no branch, no use)
            x[u] = *reinterpret_cast<const float4*>(p1 + (size_t)tc * PC_H + c0);
            y[u] = *reinterpret_cast<const float4*>(p2 + (size_t)tc * PC_H + c0);
        }
#pragma unroll
        for (int u = 0; u < FIN4_BATCH; u++) {
            const int tt = t + u * FIN4_LANES;
            const int sg = tt < t_end ? (tt >= b1) + (tt >= b2) + (tt >= b3) : -1;
#pragma unroll
            for (int s = 0; s < PC_MAX_SEG; s++) {
                const bool on = sg == s;
                a[s][0] += on ? (double)x[u].x : 0.0; a[s][1] += on ? (double)x[u].y : 0.0;
                a[s][2] += on ? (double)x[u].z : 0.0; a[s][3] += on ? (double)x[u].w : 0.0;
                b[s][0] += on ? (double)y[u].x : 0.0; b[s][1] += on ? (double)y[u].y : 0.0;
                b[s][2] += on ? (double)y[u].z : 0.0; b[s][3] += on ? (double)y[u].w : 0.0;
            }
        }
    }
    // the wave's eight tile lanes (lane bits 3..5), fixed order
#pragma unroll
    for (int s = 0; s < PC_MAX_SEG; s++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { a[s][c] += __shfl_xor(a[s][c], o, 64); b[s][c] += __shfl_xor(b[s][c], o, 64); }
        }
    if (lane < FIN4_CG) {
#pragma unroll
        for (int s = 0; s < PC_MAX_SEG; s++)
#pragma unroll
            for (int c = 0; c < 4; c++) { red[wave][0][s][cg * 4 + c] = a[s][c]; red[wave][1][s][cg * 4 + c] = b[s][c]; }
    }
    __syncthreads();
    if (tid < PC_MAX_SEG * FIN_COLS) {
        const int s = tid / FIN_COLS, col = tid % FIN_COLS;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < FIN4_WAVES; w++) { s1 += red[w][0][s][col]; s2 += red[w][1][s][col]; }
        *o1 = s1; *o2 = s2;
    }
}

// Cross-replica BatchNorm: the per-segment sums of THIS replica, in fp64, in the
// exchange layout sums[PC_BN_SYNC_DOUBLES] = [seg][2][H] sums then [seg] row counts.  The host adds the
// buffers of all replicas (one all-reduce) and hands the result to the finalize kernels below.
__global__ __launch_bounds__(FIN_COLS * FIN_LANES) void bn_fold_kernel(const float* p1, const float* p2, SegInfo si,
                                                                       double* sums) {
    __shared__ double red[2][FIN_LANES][FIN_COLS];
    const int j = blockIdx.x * FIN_COLS + threadIdx.x, q = threadIdx.y;
    for (int s = 0; s < PC_MAX_SEG; s++) {
        double a = 0.0, b = 0.0;
        if (s < si.nseg) fold_partials(p1, p2, si.tile0[s], si.tile0[s + 1], j, q, red, &a, &b);
        if (q == 0) {
            sums[(2 * s) * PC_H + j] = a;
            sums[(2 * s + 1) * PC_H + j] = b;
            if (j == 0) sums[2 * PC_MAX_SEG * PC_H + s] = s < si.nseg ? (double)si.count[s] : 0.0;
        }
    }
}

// gsum: NULL = statistics of this replica (fold the per-tile partials here); else the all-reduced exchange buffer.
// (Round 4 tried the four segments side by side in 32-64 narrower workgroups -- 4-8 columns each: 12.6 -> 9.0 us at configs[1], but
// a wave then touches 16 tile rows per load instead of 2, and beside configs[4]'s loader kernels (random reads over 114 GB: the
// TLB is theirs) every one of those is a miss: 115-144 us there under the profiler.  Back to 32 adjacent columns per tile row.
// Round 6: the same 32 columns x 32 tile lanes, all segments in one pass with sixteen loads in flight per thread
// (fold_partials_all); thread (column, q = s) finalizes segment s, the running statistics are chained in segment order.)
__global__ __launch_bounds__(FIN4_CG * FIN4_LANES) void bn_finalize_fwd_kernel(
    const float* psum, const float* psq, SegInfo si, const double* gsum, const float* gamma, const float* beta,
    float* running_mean, float* running_var, int64_t* nbt, int update_running, float* mean_o, float* invstd_o,
    float* scale_o, float* shift_o, TransposeBatch ride, int tiles_x, int tiles_y) {
    __shared__ double red[FIN4_WAVES][2][PC_MAX_SEG][FIN_COLS];
    __shared__ float seg_mean[PC_MAX_SEG][FIN_COLS], seg_unb[PC_MAX_SEG][FIN_COLS];
    __shared__ int seg_live[PC_MAX_SEG][FIN_COLS];
    const int tid = threadIdx.y * FIN4_CG + threadIdx.x;
    if (blockIdx.x >= PC_H / FIN_COLS) {
        // riders (the unsplit step with loader-made row indices): the transposed weights of the attention and of the backward --
        // nothing before this launch's successor needs them, and the eight finalize workgroups leave 248 CUs idle
        static_assert(sizeof(red) >= 32 * 33 * sizeof(float), "transpose tile");
        transpose_tile_body<FIN4_CG * FIN4_LANES>(ride, (int)blockIdx.x - PC_H / FIN_COLS, tiles_x, tiles_y,
                                                  reinterpret_cast<float(*)[33]>(&red[0][0][0][0]), tid);
        return;
    }
    double a = 0.0, b = 0.0;
    if (!gsum) fold_partials_all4(psum, psq, si, blockIdx.x * FIN_COLS, red, &a, &b);
    if (tid < PC_MAX_SEG * FIN_COLS) {
        const int s = tid / FIN_COLS, col = tid % FIN_COLS, j = blockIdx.x * FIN_COLS + col;
        double n = s < si.nseg ? (double)si.count[s] : 0.0;   // logical rows (a weighted row counts wmult times)
        if (gsum && s < si.nseg) { a = gsum[(2 * s) * PC_H + j]; b = gsum[(2 * s + 1) * PC_H + j]; n = gsum[2 * PC_MAX_SEG * PC_H + s]; }
        int live = 0;
        if (s < si.nseg && n > 0) {
            const double m = a / n;
            double var = b / n - m * m;
            if (var < 0.0) var = 0.0;
            const float mf = (float)m, vf = (float)var;
            const float is = 1.0f / sqrtf(vf + BN_EPS);
            const float sc = gamma[j] * is;
            mean_o[s * PC_H + j] = mf;
            invstd_o[s * PC_H + j] = is;
            scale_o[s * PC_H + j] = sc;
            shift_o[s * PC_H + j] = beta[j] - mf * sc;
            seg_mean[s][col] = mf;
            seg_unb[s][col] = n > 1 ? (float)(var * (n / (n - 1))) : vf;
            live = 1;
        }
        seg_live[s][col] = live;
    }
    __syncthreads();
    if (tid < FIN_COLS && update_running) {
        // nn.BatchNorm1d updates its buffers once per CALL: the segments in call order (product2vec.py:132-134)
        const int j = blockIdx.x * FIN_COLS + tid;
        float rm = running_mean[j], rv = running_var[j];
        int nseen = 0;
        for (int s = 0; s < si.nseg; s++) {
            if (!seg_live[s][tid]) continue;
            rm = BN_MOMENTUM * seg_mean[s][tid] + (1.0f - BN_MOMENTUM) * rm;
            rv = BN_MOMENTUM * seg_unb[s][tid] + (1.0f - BN_MOMENTUM) * rv;
            nseen++;
        }
        running_mean[j] = rm;
        running_var[j] = rv;
        if (j == 0 && nbt) *nbt += nseen;
    }
}

__global__ void bn_eval_coeff_kernel(const float* gamma, const float* beta, const float* running_mean,
                                     const float* running_var, float* scale_o, float* shift_o) {
    const int j = threadIdx.x;
    const float is = 1.0f / sqrtf(running_var[j] + BN_EPS);
    const float sc = gamma[j] * is;
    scale_o[j] = sc;
    shift_o[j] = beta[j] - running_mean[j] * sc;
}

// per-tile (sum dz1, sum dz1*h0) -> dgamma, dbeta (+ per-segment means c1 = dbeta_s/n, c2 = dgamma_s/n)
// (round 6: one pass over all segments like the forward finalize, on the step's own queue between dZ1 and dW3 instead of the
// side queue: the two cross-queue hops it needed there cost the main queue ~6.5 us each)
__global__ __launch_bounds__(FIN4_CG * FIN4_LANES) void bn_finalize_bwd_kernel(
    const float* psum, const float* pdot, SegInfo si, const double* lsum, const double* gsum, const float* mean,
    const float* invstd, float* dgamma, float* dbeta, int accumulate, float* c1, float* c2) {
    __shared__ double red[FIN4_WAVES][2][PC_MAX_SEG][FIN_COLS];
    __shared__ double seg_g[PC_MAX_SEG][FIN_COLS], seg_b[PC_MAX_SEG][FIN_COLS];
    const int tid = threadIdx.y * FIN4_CG + threadIdx.x;
    double a = 0.0, b = 0.0;
    if (!lsum) fold_partials_all4(psum, pdot, si, blockIdx.x * FIN_COLS, red, &a, &b);
    if (tid < PC_MAX_SEG * FIN_COLS) {
        const int s = tid / FIN_COLS, col = tid % FIN_COLS, j = blockIdx.x * FIN_COLS + col;
        double tg = 0.0, tb = 0.0;
        if (s < si.nseg) {
            double n = si.count[s];                           // logical rows (a weighted row counts wmult times)
            if (lsum) { a = lsum[(2 * s) * PC_H + j]; b = lsum[(2 * s + 1) * PC_H + j]; }
            // the tiles carry the raw moment sum dz1*h0: sum dz1*xhat = invstd * (sum dz1*h0 - mean * sum dz1)
            const double is = (double)invstd[s * PC_H + j], mu = (double)mean[s * PC_H + j];
            b = is * (b - mu * a);
            tb = a;                                           // dbeta / dgamma: this replica's rows
            tg = b;
            double ga = a, gb = b;                            // the BN-backward means run over ALL replicas' rows
            if (gsum) {
                ga = gsum[(2 * s) * PC_H + j];
                gb = is * (gsum[(2 * s + 1) * PC_H + j] - mu * ga);
                n = gsum[2 * PC_MAX_SEG * PC_H + s];
            }
            c1[s * PC_H + j] = n > 0 ? (float)(ga / n) : 0.f;
            c2[s * PC_H + j] = n > 0 ? (float)(gb / n) : 0.f;
        }
        seg_g[s][col] = tg;
        seg_b[s][col] = tb;
    }
    __syncthreads();
    if (tid < FIN_COLS) {
        const int j = blockIdx.x * FIN_COLS + tid;
        double tg = 0.0, tb = 0.0;
        for (int s = 0; s < si.nseg; s++) { tg += seg_g[s][tid]; tb += seg_b[s][tid]; }     // segment order, as before
        dgamma[j] = accumulate ? dgamma[j] + (float)tg : (float)tg;
        dbeta[j] = accumulate ? dbeta[j] + (float)tb : (float)tb;
    }
}

// dH0 = scale_s * (dZ1 - c1_s - xhat*c2_s), xhat = (H0 - mean_s)*invstd_s; in place over dz [R,256]
__global__ void bn_bwd_apply_kernel(float* dz, const float* h0, int rows, SegInfo si, const float* mean,
                                    const float* invstd, const float* scale, const float* c1, const float* c2) {
    const int c4 = (threadIdx.x & 63) * 4;                     // 64 threads x float4 = one 256-wide row
    const int rl = threadIdx.x >> 6;                           // 4 rows per block pass
    for (int r = blockIdx.x * 4 + rl; r < rows; r += gridDim.x * 4) {
        const int s = seg_of_row(si, r);
        const float m = row_multiplicity(si, r);                  // sum over the rows this one stands for
        float4 g = *reinterpret_cast<float4*>(dz + (size_t)r * PC_H + c4);
        const float4 h = *reinterpret_cast<const float4*>(h0 + (size_t)r * PC_H + c4);
        const float4 mu = *reinterpret_cast<const float4*>(mean + s * PC_H + c4);
        const float4 is = *reinterpret_cast<const float4*>(invstd + s * PC_H + c4);
        const float4 sc = *reinterpret_cast<const float4*>(scale + s * PC_H + c4);
        const float4 a = *reinterpret_cast<const float4*>(c1 + s * PC_H + c4);
        const float4 b = *reinterpret_cast<const float4*>(c2 + s * PC_H + c4);
        g.x = sc.x * (g.x - m * (a.x + (h.x - mu.x) * is.x * b.x));
        g.y = sc.y * (g.y - m * (a.y + (h.y - mu.y) * is.y * b.y));
        g.z = sc.z * (g.z - m * (a.z + (h.z - mu.z) * is.z * b.z));
        g.w = sc.w * (g.w - m * (a.w + (h.w - mu.w) * is.w * b.w));
        *reinterpret_cast<float4*>(dz + (size_t)r * PC_H + c4) = g;
    }
}

// out[c][r] = in[r][c]   (weights are <= 256x256: one pass, LDS tile 32x33)
__global__ void transpose_kernel(const float* in, int rows, int cols, float* out) {
    __shared__ float t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = by + i, c = bx + threadIdx.x;
        t[i][threadIdx.x] = (r < rows && c < cols) ? in[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = bx + i, r = by + threadIdx.x;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = t[threadIdx.x][i];
    }
}

// several small transposes in ONE launch (blockIdx.z picks the matrix): the backward passes need
// W^T of 2-3 weights each step and a launch boundary costs more than the transpose itself
__global__ void transpose_batch_kernel(TransposeBatch tb) {
    __shared__ float t[32][33];
    const TransposeJob j = tb.job[blockIdx.z];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    if (tb.zero && blockIdx.x + blockIdx.y + blockIdx.z == 0)
        for (int i = threadIdx.y * 32 + threadIdx.x; i < tb.nzero; i += 256) tb.zero[i] = 0.f;
    if (bx >= j.cols || by >= j.rows) return;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = by + i, c = bx + threadIdx.x;
        t[i][threadIdx.x] = (r < j.rows && c < j.cols) ? j.in[(size_t)r * j.cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int c = bx + i, r = by + threadIdx.x;
        if (r < j.rows && c < j.cols) j.out[(size_t)c * j.rows + r] = t[threadIdx.x][i];
    }
}

int launch_transpose_batch(const TransposeBatch& tb, hipStream_t st) {
    int mx = 0, my = 0;
    for (int i = 0; i < tb.n; i++) {
        mx = tb.job[i].cols > mx ? tb.job[i].cols : mx;
        my = tb.job[i].rows > my ? tb.job[i].rows : my;
    }
    PC_LAUNCH(transpose_batch_kernel, dim3((mx + 31) / 32, (my + 31) / 32, tb.n), dim3(32, 8), 0, st, tb);
    return pc_launch_status();
}

int launch_transpose(const float* in, int rows, int cols, float* out, hipStream_t st) {
    PC_LAUNCH(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, st, in, rows, cols, out);
    return pc_launch_status();
}

// ---------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct FfnWs {
    float *stat_a, *stat_b;     // [ntiles][H] each
    float *dz2, *dz1;           // [R,H] each (backward; also h0/a2 scratch for eval)
    float *w5t, *w3t, *w0t;     // transposed weights
    float *c1, *c2;             // [MAX_SEG][H]
    float *coef;                // eval: scale, shift [2][H]
    // one region per weight gradient (dW5, the two halves of dW3, dW0): their slab sums are folded by ONE launch at the
    // end of the step (TnDefer), so no region may be reused in between
    float *slabs[4]; size_t slab_floats;
    size_t total;
};

static FfnWs ffn_ws_layout(void* base, int rows) {
    FfnWs w;
    const int max_tiles = (rows + 127) / 128 + PC_MAX_SEG;
    size_t off = 0;
    auto take = [&](size_t floats) {
        float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
        off += align256(floats * sizeof(float));
        return p;
    };
    w.stat_a = take((size_t)max_tiles * PC_H);
    w.stat_b = take((size_t)max_tiles * PC_H);
    w.dz2 = take((size_t)rows * PC_H);
    w.dz1 = take((size_t)rows * PC_H);
    w.w5t = take(PC_H * 256);                  // [H, D], D <= 256
    w.w3t = take(PC_H * PC_H);
    w.w0t = take(256 * PC_H);
    w.c1 = take(PC_MAX_SEG * PC_H);
    w.c2 = take(PC_MAX_SEG * PC_H);
    w.coef = take(2 * PC_H);
    w.slab_floats = gemm_tn_workspace_floats(rows, PC_H, PC_H);
    for (size_t f : {gemm_tn_workspace_floats(rows, PC_D, PC_H), gemm_tn_workspace_floats(rows, PC_H, PC_D)})
        if (f > w.slab_floats) w.slab_floats = f;
    for (int i = 0; i < 4; i++) w.slabs[i] = take(w.slab_floats);
    w.total = off;
    return w;
}

extern "C" size_t pc_p2v_ffn_workspace_bytes(int rows) {
    if (rows <= 0) return 0;
    return ffn_ws_layout(nullptr, rows).total;
}

static inline int p2v_dim(const pc_p2v_tensors* p) { return p && p->dim == 256 ? 256 : PC_D; }

static int ffn_check(const pc_p2v_tensors* p, const float* table, int rows, const pc_segments* seg, void* ws,
                     size_t ws_bytes) {
    if (!p || !table || rows <= 0 || !ws) return PC_EINVAL;
    if (p->dim != 0 && p->dim != 128 && p->dim != 256) return PC_ESHAPE;
    if (!p->w0 || !p->b0 || !p->gamma || !p->beta || !p->w3 || !p->b3 || !p->w5 || !p->b5) return PC_EINVAL;
    if (!seg_valid(seg, rows)) return PC_EINVAL;
    if (ws_bytes < pc_p2v_ffn_workspace_bytes(rows)) return PC_EWORKSPACE;
    return PC_OK;
}

static NtArgs nt_plain(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
                       int N, int K, const SegInfo& si) {
    NtArgs a = {};
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.seg = si;
    return a;
}

// part 1: gather + Linear0 + per-tile sums (+ this replica's folded sums into `local_sums` when exchanging);
// part 2: statistics (from `global_sums` when given) -> BN-tanh -> Linear3 -> tanh -> Linear5
int ffn_forward_part1(const pc_p2v_tensors* p, const float* table, const int32_t* idx, int rows, const pc_segments* seg,
                      const pc_ffn_saved* sv, double* local_sums, void* ws, size_t ws_bytes, void* stream) {
    PC_TRY(ffn_check(p, table, rows, seg, ws, ws_bytes));
    if (!sv || !sv->h0 || !sv->a2 || !sv->bn_mean || !sv->bn_invstd || !sv->bn_scale || !sv->bn_shift) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int D = p2v_dim(p);
    const SegInfo si = make_seginfo(seg, rows, 128);
    FfnWs w = ffn_ws_layout(ws, rows);
    NtArgs g1 = nt_plain(table, D, p->w0, D, p->b0, sv->h0, PC_H, rows, PC_H, D, si);
    g1.gather = idx;
    g1.stats = NT_STAT_SUMSQ; g1.stat_sum = w.stat_a; g1.stat_aux = w.stat_b;
    PC_TRY(launch_gemm_nt(g1, st));
    if (local_sums) {
        PC_LAUNCH(bn_fold_kernel, dim3(PC_H / FIN_COLS), dim3(FIN_COLS, FIN_LANES), 0, st, w.stat_a, w.stat_b, si, local_sums);
        PC_TRY(pc_launch_status());
    }
    return PC_OK;
}

int ffn_forward_part2(const pc_p2v_tensors* p, int rows, const pc_segments* seg, int update_running, float* y,
                      const pc_ffn_saved* sv, const double* global_sums, void* ws, size_t ws_bytes, void* stream,
                      const TransposeBatch* ride) {
    if (!y || (update_running && (!p->running_mean || !p->running_var))) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const SegInfo si = make_seginfo(seg, rows, 128);
    FfnWs w = ffn_ws_layout(ws, rows);
    TransposeBatch tb = {};
    int tiles_x = 0, tiles_y = 0;
    if (ride && ride->n > 0) { tb = *ride; transpose_batch_tiles(tb, &tiles_x, &tiles_y); }
    PC_LAUNCH(bn_finalize_fwd_kernel, dim3(PC_H / FIN_COLS + tiles_x * tiles_y * tb.n), dim3(FIN4_CG, FIN4_LANES), 0, st, w.stat_a,
              w.stat_b, si, global_sums, p->gamma, p->beta, p->running_mean, p->running_var, p->num_batches_tracked, update_running,
              sv->bn_mean, sv->bn_invstd, sv->bn_scale, sv->bn_shift, tb, tiles_x, tiles_y);
    PC_TRY(pc_launch_status());

    NtArgs g2 = nt_plain(sv->h0, PC_H, p->w3, PC_H, p->b3, sv->a2, PC_H, rows, PC_H, PC_H, si);
    g2.prologue = NT_PRO_BNTANH; g2.pscale = sv->bn_scale; g2.pshift = sv->bn_shift;
    g2.pro_out = sv->a1; g2.ldpo = PC_H;               // A1 = tanh(BN(H0)) leaves the kernel that forms it anyway (dW3 reads it)
    g2.epilogue = NT_EPI_TANH;
    PC_TRY(launch_gemm_nt(g2, st));

    const int D = p2v_dim(p);
    NtArgs g3 = nt_plain(sv->a2, PC_H, p->w5, PC_H, p->b5, y, D, rows, D, PC_H, si);
    return launch_gemm_nt(g3, st);
}

extern "C" int pc_p2v_ffn_forward_train(const pc_p2v_tensors* p, const float* table, const int32_t* idx, int rows,
                                        const pc_segments* seg, int update_running, float* y,
                                        const pc_ffn_saved* sv, void* ws, size_t ws_bytes, void* stream) {
    if (!y) return PC_EINVAL;
    PC_TRY(ffn_forward_part1(p, table, idx, rows, seg, sv, nullptr, ws, ws_bytes, stream));
    return ffn_forward_part2(p, rows, seg, update_running, y, sv, nullptr, ws, ws_bytes, stream, nullptr);
}

extern "C" int pc_p2v_ffn_forward_eval(const pc_p2v_tensors* p, const float* table, const int32_t* idx, int rows,
                                       float* y, void* ws, size_t ws_bytes, void* stream) {
    PC_TRY(ffn_check(p, table, rows, nullptr, ws, ws_bytes));
    if (!y || !p->running_mean || !p->running_var) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const SegInfo si = make_seginfo(nullptr, rows, 128);
    FfnWs w = ffn_ws_layout(ws, rows);
    float* h0 = w.dz2;
    float* a2 = w.dz1;
    PC_LAUNCH(bn_eval_coeff_kernel, dim3(1), dim3(PC_H), 0, st, p->gamma, p->beta, p->running_mean,
                       p->running_var, w.coef, w.coef + PC_H);
    PC_TRY(pc_launch_status());
    const int D = p2v_dim(p);
    NtArgs g1 = nt_plain(table, D, p->w0, D, p->b0, h0, PC_H, rows, PC_H, D, si);
    g1.gather = idx;
    PC_TRY(launch_gemm_nt(g1, st));
    NtArgs g2 = nt_plain(h0, PC_H, p->w3, PC_H, p->b3, a2, PC_H, rows, PC_H, PC_H, si);
    g2.prologue = NT_PRO_BNTANH; g2.pscale = w.coef; g2.pshift = w.coef + PC_H;
    g2.epilogue = NT_EPI_TANH;
    PC_TRY(launch_gemm_nt(g2, st));
    NtArgs g3 = nt_plain(a2, PC_H, p->w5, PC_H, p->b5, y, D, rows, D, PC_H, si);
    return launch_gemm_nt(g3, st);
}

// part 1: everything up to the BN-backward partial sums (+ this replica's folded sums into `local_sums`);
// part 2: dgamma/dbeta (local rows), BN-backward coefficients (from `global_sums` when given), dW0/db0 (/dx)
// the transposed weights the backward multiplies by (W5^T, W3^T, and W0^T for a dx consumer) as jobs of a transpose launch
int ffn_transposes(const pc_p2v_tensors* p, void* ws, int rows, int with_dx, TransposeBatch* tb) {
    FfnWs w = ffn_ws_layout(ws, rows);
    const int D = p2v_dim(p);
    if (tb->n + 2 + (with_dx ? 1 : 0) > PC_TRANSPOSE_JOBS) return PC_EINVAL;
    tb->job[tb->n++] = {p->w5, w.w5t, D, PC_H};                      // [D,H] -> [H,D]
    tb->job[tb->n++] = {p->w3, w.w3t, PC_H, PC_H};
    if (with_dx) tb->job[tb->n++] = {p->w0, w.w0t, PC_H, D};         // [H,D] -> [D,H]
    return PC_OK;
}

// transposed != 0: ffn_transposes' products are in the workspace already (the fused step forms every transposed weight
// of the step in one launch).  defer (optional): the slab sums join the caller's list instead of being launched here.
int ffn_backward_part1(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* table, const int32_t* idx,
                       int rows, const pc_segments* seg, const float* dy, const pc_ffn_saved* sv, int with_dx,
                       int accumulate, double* local_sums, void* ws, size_t ws_bytes, void* stream, int transposed,
                       TnDefer* defer) {
    PC_TRY(ffn_check(p, table, rows, seg, ws, ws_bytes));
    if (!g || !dy || !sv || !sv->h0 || !sv->a2) return PC_EINVAL;
    if (!g->w0 || !g->b0 || !g->gamma || !g->beta || !g->w3 || !g->b3 || !g->w5 || !g->b5) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const SegInfo si = make_seginfo(seg, rows, 128);
    FfnWs w = ffn_ws_layout(ws, rows);
    TnDefer own;
    tn_defer_init(&own);
    TnDefer* df = defer ? defer : &own;

    const int D = p2v_dim(p);
    if (!transposed) {
        TransposeBatch tb = {};
        PC_TRY(ffn_transposes(p, ws, rows, with_dx, &tb));
        PC_TRY(launch_transpose_batch(tb, st));
    }

    // dZ2 = (dY W5) * (1 - A2^2)
    NtArgs b1 = nt_plain(dy, D, w.w5t, D, nullptr, w.dz2, PC_H, rows, PC_H, D, si);
    b1.epilogue = NT_EPI_DTANH; b1.aux = sv->a2; b1.ldaux = PC_H;
    PC_TRY(launch_gemm_nt(b1, st));

    // dW5 = dY^T A2, db5
    TnArgs t5 = {};
    t5.Z = dy; t5.ldz = D; t5.A = sv->a2; t5.lda = PC_H; t5.R = rows; t5.No = D; t5.Ni = PC_H; t5.seg = si;
    t5.dW = g->w5; t5.lddw = PC_H; t5.db = g->b5; t5.accumulate = accumulate; t5.slabs = w.slabs[0];
    t5.slab_floats = w.slab_floats;
    if (D == 256 && rows >= 8192 && (size_t)2 * 128 * ((size_t)128 * PC_H + 128) <= w.slab_floats) {
        // PRODUCT_EMB_DIM = 256: dW5 is a 256 x 256 gradient like dW3 -- the paired half-slice launch instead of the full tile
        // (eight accumulator blocks per wave at 256 registers: 32 B of scratch, 133 us at configs[4])
        const size_t half = (size_t)128 * ((size_t)128 * PC_H + 128);
        PC_TRY(launch_gemm_tn_halves(t5, w.slabs[0], w.slabs[0] + half, half, st, df));
    } else {
        PC_TRY(launch_gemm_tn(t5, st, df));
    }

    // dZ1 = (dZ2 W3) * (1 - A1^2), A1 = tanh(BN(H0)); plus BN-backward partial sums
    NtArgs b2 = nt_plain(w.dz2, PC_H, w.w3t, PC_H, nullptr, w.dz1, PC_H, rows, PC_H, PC_H, si);
    b2.epilogue = NT_EPI_DTANH_BN; b2.aux = sv->h0; b2.ldaux = PC_H; b2.escale = sv->bn_scale; b2.eshift = sv->bn_shift;
    b2.stats = NT_STAT_BNBWD; b2.stat_sum = w.stat_a; b2.stat_aux = w.stat_b;
    PC_TRY(launch_gemm_nt(b2, st));
    if (df->fork && !local_sums) {
        // the BatchNorm-backward finalize (only dW0 reads c1 / c2).  Rounds 3-5: on the side queue beside dW3 -- the fork and the
        // join each cost the main queue ~6.5 us (profiles/r06a_step_timeline.md) and the 8-workgroup kernel took 12-32 us there.
        // Round 6: on the step's own queue, between dZ1 and dW3: 12-14 us exposed, no hops, -8 us per step in A/B
        // (pc_set_option(PC_OPT_BN_FINALIZE_SIDE, 1): the old placement)
        if (pc_opt_bn_finalize_side()) {
            PC_TRY(pc_fork_begin(df->fork, 1, st));
            PC_LAUNCH(bn_finalize_bwd_kernel, dim3(PC_H / FIN_COLS), dim3(FIN4_CG, FIN4_LANES), 0, df->fork->side, w.stat_a, w.stat_b, si,
                      nullptr, nullptr, sv->bn_mean, sv->bn_invstd, g->gamma, g->beta, accumulate, w.c1, w.c2);
            PC_TRY(pc_launch_status());
            df->bn_finalized = 1;
        } else {
            PC_LAUNCH(bn_finalize_bwd_kernel, dim3(PC_H / FIN_COLS), dim3(FIN4_CG, FIN4_LANES), 0, st, w.stat_a, w.stat_b, si,
                      nullptr, nullptr, sv->bn_mean, sv->bn_invstd, g->gamma, g->beta, accumulate, w.c1, w.c2);
            PC_TRY(pc_launch_status());
            df->bn_finalized = 2;                            // done, on the step's queue: nothing to join
        }
    }

    // dW3 = dZ2^T A1, db3 (A1 as the forward saved it; without the optional buffer it is recomputed from H0 in the loader:
    // the in-place tanh(BN(.)) on every landed stage made this the longest kernel of the step, 123 us)
    TnArgs t3 = {};
    t3.Z = w.dz2; t3.ldz = PC_H; t3.A = sv->a1 ? sv->a1 : sv->h0; t3.lda = PC_H; t3.R = rows; t3.No = PC_H; t3.Ni = PC_H; t3.seg = si;
    if (!sv->a1) { t3.prologue = NT_PRO_BNTANH; t3.pscale = sv->bn_scale; t3.pshift = sv->bn_shift; }
    t3.dW = g->w3; t3.lddw = PC_H; t3.db = g->b3; t3.accumulate = accumulate; t3.slabs = w.slabs[1];
    t3.slab_floats = w.slab_floats;
    if (sv->a1) {
        // as two 128 x 256 halves of the output rows: the 256 x 256 tile needs 8 blocks per wave at 256 registers (six splits
        // and a spill per 48 MFMAs: 102 us); the half-size kernel runs 44 us per half (scripts/dev/nt_decompose.sh; same box,
        // whole step: 1.068 -> 1.062 ms)
        // (round 4: many rows -- the two halves as ONE launch of 2 x 128 row slices: A1 comes from HBM once, half the slabs)
        if (rows >= 8192 && (size_t)128 * ((size_t)128 * PC_H + 128) <= w.slab_floats) {
            PC_TRY(launch_gemm_tn_halves(t3, w.slabs[1], w.slabs[2], w.slab_floats, st, df));
        } else {
            for (int half = 0; half < 2; half++) {
                TnArgs th = t3;
                th.Z = w.dz2 + half * (PC_H / 2); th.No = PC_H / 2;
                th.dW = g->w3 + (size_t)half * (PC_H / 2) * PC_H; th.db = g->b3 + half * (PC_H / 2);
                th.slabs = w.slabs[1 + half];
                PC_TRY(launch_gemm_tn(th, st, df));
            }
        }
    } else {
        PC_TRY(launch_gemm_tn(t3, st, df));
    }
    if (local_sums) {
        PC_LAUNCH(bn_fold_kernel, dim3(PC_H / FIN_COLS), dim3(FIN_COLS, FIN_LANES), 0, st, w.stat_a, w.stat_b, si, local_sums);
        PC_TRY(pc_launch_status());
    }
    if (!defer) PC_TRY(launch_tn_reduce_deferred(&own, st));
    return PC_OK;
}

int ffn_backward_part2(const pc_p2v_tensors* g, const float* table, const int32_t* idx, int rows,
                       const pc_segments* seg, const pc_ffn_saved* sv, float* dx, int accumulate,
                       const double* local_sums, const double* global_sums, void* ws, size_t ws_bytes, void* stream,
                       TnDefer* defer) {
    hipStream_t st = (hipStream_t)stream;
    const SegInfo si = make_seginfo(seg, rows, 128);
    FfnWs w = ffn_ws_layout(ws, rows);
    if (defer && defer->bn_finalized) {
        if (local_sums || global_sums) return PC_EINVAL;
        if (defer->bn_finalized == 1) PC_TRY(pc_fork_join(defer->fork, 0, st));      // part 1 ran the finalize on the side queue
    } else {
        PC_LAUNCH(bn_finalize_bwd_kernel, dim3(PC_H / FIN_COLS), dim3(FIN4_CG, FIN4_LANES), 0, st, w.stat_a, w.stat_b, si, local_sums,
                  global_sums, sv->bn_mean, sv->bn_invstd, g->gamma, g->beta, accumulate, w.c1, w.c2);
        PC_TRY(pc_launch_status());
    }

    // dW0 = dH0^T X (rows gathered again from the table), db0.  Without a dx consumer the BatchNorm
    // backward is applied to dZ1 on the fly inside the loader and dH0 never touches HBM.
    const int D = p2v_dim(g);                                     // (the gradient struct carries the same dim)
    TnArgs t0 = {};
    t0.Z = w.dz1; t0.ldz = PC_H; t0.A = table; t0.lda = D; t0.gather = idx; t0.R = rows; t0.No = PC_H;
    t0.Ni = D; t0.seg = si;
    t0.dW = g->w0; t0.lddw = D; t0.db = g->b0; t0.accumulate = accumulate; t0.slabs = w.slabs[3];
    t0.slab_floats = w.slab_floats;
    if (dx) {
        int blocks = (rows + 3) / 4;
        if (blocks > 4096) blocks = 4096;
        PC_LAUNCH(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, w.dz1, sv->h0, rows, si, sv->bn_mean,
                  sv->bn_invstd, sv->bn_scale, w.c1, w.c2);
        PC_TRY(pc_launch_status());
    } else {
        t0.zaux = sv->h0; t0.ldzaux = PC_H;
        t0.z_mean = sv->bn_mean; t0.z_invstd = sv->bn_invstd; t0.z_scale = sv->bn_scale; t0.z_c1 = w.c1; t0.z_c2 = w.c2;
    }
    PC_TRY(launch_gemm_tn(t0, st, defer));

    if (dx) {
        NtArgs b3 = nt_plain(w.dz1, PC_H, w.w0t, PC_H, nullptr, dx, D, rows, D, PC_H, si);
        PC_TRY(launch_gemm_nt(b3, st));
    }
    return PC_OK;
}

extern "C" int pc_p2v_ffn_backward(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* table,
                                   const int32_t* idx, int rows, const pc_segments* seg, const float* dy,
                                   const pc_ffn_saved* sv, float* dx, int accumulate, void* ws, size_t ws_bytes,
                                   void* stream) {
    TnDefer df;
    tn_defer_init(&df);
    PC_TRY(ffn_backward_part1(p, g, table, idx, rows, seg, dy, sv, dx != nullptr, accumulate, nullptr, ws, ws_bytes, stream, 0, &df));
    PC_TRY(ffn_backward_part2(g, table, idx, rows, seg, sv, dx, accumulate, nullptr, nullptr, ws, ws_bytes, stream, &df));
    return launch_tn_reduce_deferred(&df, (hipStream_t)stream);
}
