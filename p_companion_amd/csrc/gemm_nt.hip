// C[M,N] = prologue(A)[M,K] . W[N,K]^T (+bias) -> epilogue, exact fp32 on the matrix cores.
//
// One kernel serves every "activation x weight^T" product of the two hot paths
// (nn.Linear forward, and dX = dY . W through a pre-transposed W): product2vec.py:14-21
// (ffn), nn.MultiheadAttention's in/out projections (:23-28), type_transition.py:11-12,
// item_prediction.py:11-20, p_companion.py:60-63 (similarities).
//
// gfx950 mapping: 128x128 output tile per 256-thread workgroup (4 waves, each 64x64 =
// 2x2 v_mfma_f32_32x32x2_f32 accumulators, 64 VGPRs), K stepped by 32 through a
// double-buffered LDS image (row stride 36 floats => ds_read_b128 of 16 distinct rows hits
// 16 distinct 16-B slots: conflict-free).  Rows of A may be gathered by index straight from
// the feature table (the BPG neighbour gather), tiles never straddle a BatchNorm segment,
// and the epilogue can emit the per-tile column statistics BatchNorm needs, so the FFN's
// first Linear, the row gather and the BN reduction are one pass over HBM.
#include "common.h"

#define BM 128
#define BN 128
#define BK 32
#define LDS_LD 36

int gemm_nt_tiles(const SegInfo& si) { return si.tile0[PC_MAX_SEG]; }

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NtArgs a, int ntn) {
    __shared__ __attribute__((aligned(16))) float smem[2][2][BM * LDS_LD];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int tile_n = blockIdx.x % ntn, tile_m = blockIdx.x / ntn;
    const int seg = seg_of_tile(a.seg, tile_m);
    const int row0 = a.seg.start[seg] + (tile_m - a.seg.tile0[seg]) * BM;
    const int row_end = a.seg.start[seg + 1];
    const int n0 = tile_n * BN;

    // ---- loader mapping: 8 threads cover one 128-B row chunk, 32 rows per pass, 4 passes
    const int lr = tid >> 3, lc = (tid & 7) * 4;
    const float* arow[4];
    const float* wrow[4];
    bool av[4], wv[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        int r = row0 + lr + 32 * p;
        bool v = r < row_end;
        int src = r;
        if (v && a.gather) { src = a.gather[r]; v = src >= 0; }
        av[p] = v;
        arow[p] = a.A + (size_t)(v ? src : 0) * a.lda + lc;
        int n = n0 + lr + 32 * p;
        wv[p] = n < a.N;
        wrow[p] = a.W + (size_t)(wv[p] ? n : 0) * a.ldw + lc;
    }
    const float* ps = a.prologue == NT_PRO_BNTANH ? a.pscale + (size_t)seg * a.K + lc : nullptr;
    const float* psh = a.prologue == NT_PRO_BNTANH ? a.pshift + (size_t)seg * a.K + lc : nullptr;

    float4 ra[4], rw[4];
    auto gload = [&](int k0) {
        const bool kv = k0 + lc < a.K;       // K need only be a multiple of 4: the tail chunk is zero-filled
#pragma unroll
        for (int p = 0; p < 4; p++) {
            ra[p] = (av[p] && kv) ? *reinterpret_cast<const float4*>(arow[p] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
            rw[p] = (wv[p] && kv) ? *reinterpret_cast<const float4*>(wrow[p] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (ps && kv) {
            float4 s = *reinterpret_cast<const float4*>(ps + k0);
            float4 h = *reinterpret_cast<const float4*>(psh + k0);
#pragma unroll
            for (int p = 0; p < 4; p++) {
                ra[p].x = fast_tanh(ra[p].x * s.x + h.x);
                ra[p].y = fast_tanh(ra[p].y * s.y + h.y);
                ra[p].z = fast_tanh(ra[p].z * s.z + h.z);
                ra[p].w = fast_tanh(ra[p].w * s.w + h.w);
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            *reinterpret_cast<float4*>(&smem[buf][0][(lr + 32 * p) * LDS_LD + lc]) = ra[p];
            *reinterpret_cast<float4*>(&smem[buf][1][(lr + 32 * p) * LDS_LD + lc]) = rw[p];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nk = (a.K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int frag = (lane & 31) * LDS_LD + 4 * (lane >> 5);
    for (int kt = 0; kt < nk; kt++) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        const float* As = &smem[cur][0][wm * 64 * LDS_LD + frag];
        const float* Ws = &smem[cur][1][wn * 64 * LDS_LD + frag];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            // lanes 0-31 hold k = 8kk+0..3, lanes 32-63 hold k = 8kk+4..7 (same for A and W)
            float4 a0 = *reinterpret_cast<const float4*>(As + kk * 8);
            float4 a1 = *reinterpret_cast<const float4*>(As + 32 * LDS_LD + kk * 8);
            float4 b0 = *reinterpret_cast<const float4*>(Ws + kk * 8);
            float4 b1 = *reinterpret_cast<const float4*>(Ws + 32 * LDS_LD + kk * 8);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[0][0] = mfma32(av0[r], bv0[r], acc[0][0]);
                acc[0][1] = mfma32(av0[r], bv1[r], acc[0][1]);
                acc[1][0] = mfma32(av1[r], bv0[r], acc[1][0]);
                acc[1][1] = mfma32(av1[r], bv1[r], acc[1][1]);
            }
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  acc[mt][nt][reg]: row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
        const int col = n0 + wn * 64 + nt * 32 + (lane & 31);
        const bool cv = col < a.N;
        const float bias = (cv && a.bias) ? a.bias[col] : 0.f;
        float es = 0.f, eh = 0.f, mu = 0.f, is = 0.f;
        if (cv && a.epilogue == NT_EPI_DTANH_BN) {
            es = a.escale[(size_t)seg * a.N + col];
            eh = a.eshift[(size_t)seg * a.N + col];
        }
        if (cv && a.stats == NT_STAT_BNBWD) {
            mu = a.mean[(size_t)seg * a.N + col];
            is = a.invstd[(size_t)seg * a.N + col];
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int row = row0 + wm * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const bool ok = cv && row < row_end;
                float v = acc[mt][nt][reg] + bias;
                float ax = 0.f;
                if (ok && a.aux) ax = a.aux[(size_t)row * a.ldaux + col];
                switch (a.epilogue) {
                    case NT_EPI_TANH: v = fast_tanh(v); break;
                    case NT_EPI_RELU: v = v > 0.f ? v : 0.f; break;
                    case NT_EPI_DTANH: v = v * (1.f - ax * ax); break;
                    case NT_EPI_DTANH_BN: { float s = fast_tanh(ax * es + eh); v = v * (1.f - s * s); break; }
                    case NT_EPI_DRELU: v = ax > 0.f ? v : 0.f; break;
                    default: break;
                }
                if (ok) {
                    a.C[(size_t)row * a.ldc + col] = v;
                    if (a.stats == NT_STAT_SUMSQ) { s1[nt] += v; s2[nt] += v * v; }
                    else if (a.stats == NT_STAT_BNBWD) { s1[nt] += v; s2[nt] += v * ((ax - mu) * is); }
                }
            }
        }
    }
    if (a.stats != NT_STAT_NONE) {
        // lanes l and l^32 hold the same column; then fold the two M-waves through LDS
        float* red = &smem[0][0][0];   // [2 stats][2 wm][128 cols]; all MFMA reads are done (barrier above)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            s1[nt] += __shfl_xor(s1[nt], 32, 64);
            s2[nt] += __shfl_xor(s2[nt], 32, 64);
            if (lane < 32) {
                red[(0 * 2 + wm) * BN + wn * 64 + nt * 32 + lane] = s1[nt];
                red[(1 * 2 + wm) * BN + wn * 64 + nt * 32 + lane] = s2[nt];
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.N) {
            a.stat_sum[(size_t)tile_m * a.N + n0 + tid] = red[0 * BN + tid] + red[1 * BN + tid];
            a.stat_aux[(size_t)tile_m * a.N + n0 + tid] = red[2 * BN + tid] + red[3 * BN + tid];
        }
    }
}

int launch_gemm_nt(const NtArgs& a, hipStream_t st) {
    if (!a.A || !a.W || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return PC_EINVAL;
    if (a.K % 4 != 0 || a.lda % 4 != 0 || a.ldw % 4 != 0) return PC_ESHAPE;
    if (((uintptr_t)a.A | (uintptr_t)a.W) & 15) return PC_ESHAPE;
    const int ntm = gemm_nt_tiles(a.seg);
    const int ntn = (a.N + BN - 1) / BN;
    if (ntm <= 0) return PC_EINVAL;
    const int pb = pc_prof_begin(PC_KIND_GEMM_NT, 2.0 * a.M * (double)a.N * a.K, st);
    PC_LAUNCH(gemm_nt_kernel, dim3(ntm * ntn), dim3(256), 0, st, a, ntn);
    pc_prof_end(pb, st);
    return pc_launch_status();
}
