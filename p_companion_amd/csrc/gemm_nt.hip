// C[M,N] = prologue(A)[M,K] . W[N,K]^T (+bias) -> epilogue, exact fp32 on the matrix cores.
//
// One kernel family serves every "activation x weight^T" product of the two hot paths
// (nn.Linear forward, and dX = dY . W through a pre-transposed W): product2vec.py:14-21
// (ffn), nn.MultiheadAttention's in/out projections (:23-28), type_transition.py:11-12,
// item_prediction.py:11-20, p_companion.py:60-63 (similarities).
//
// These are SKINNY products: M = hundreds of thousands of rows, N,K <= 256.  At the fp32
// MFMA rate they sit within 2x of the HBM roofline, so the kernel is built around the row
// stream, not around K:
//   * a workgroup owns 128 rows x the FULL N (up to 256 columns: 8 waves as 2 x 4, each 64x64
//     = 2x2 v_mfma_f32_32x32x2_f32 accumulators) so A is read from HBM exactly once;
//   * workgroups are PERSISTENT over row tiles and the K-chunk prefetch runs across tile
//     boundaries (the first chunk of the next tile -- including its gather indices -- is in
//     flight during the last chunk of this one), so short K (4-8 chunks) behaves like one
//     long pipelined loop;
//   * rows of A may be gathered by index straight from the feature table (BPG neighbour
//     gather, -1 = zero row), tiles never straddle a BatchNorm segment, A can be transformed on
//     load (BN-apply + tanh) and the epilogue can emit per-tile BatchNorm partial sums;
//   * the epilogue goes through LDS so that C (and the aux operand of the d-activation
//     epilogues) move as 16-B per lane, 8 full 128-B row segments per wave-instruction.
// LDS image: [row][36 floats] (stride 144 B => ds_read_b128 of 16 distinct rows hits 16
// distinct 16-B slots), double-buffered; operands are fetched with a permuted k order (lanes
// 0-31 take k = 8j..8j+3, lanes 32-63 k = 8j+4..8j+7, identically for A and W) so one b128
// read feeds four MFMAs.
#include "common.h"

#define PLD 36          // row stride (floats) of the per-wave epilogue transposition patch

int gemm_nt_tiles(const SegInfo& si) { return si.tile0[PC_MAX_SEG]; }

// NWM x NWN waves of 64x64 each: BM = 64*NWM rows, BN = 64*NWN columns.
//   <2,4> 128x256, 8 waves, one workgroup per CU   (N > 128)
//   <2,2> 128x128, 4 waves, two workgroups per CU  (N <= 128, many rows)
//   <1,2>  64x128, 2 waves                         (few rows: more, smaller tiles fill more CUs)
// PRO / EPI / STATS are compile-time: a runtime switch per output element costs ~1200 scalar
// branches per tile and wave (measured: 12 us of a 33 us tile) and the unused fusions' registers.
template <int NWM, int NWN, int BK, int OCC, bool PRO, int EPI, int STATS>
__global__ __launch_bounds__(64 * NWM * NWN, OCC) void gemm_nt_kernel(NtArgs a, int ntn, int total_tiles) {
    constexpr int THREADS = 64 * NWM * NWN;
    constexpr int BM = 64 * NWM, BN = 64 * NWN;
    constexpr int LDS_LD = BK + 4;               // row stride: (BK+4)*4 B = odd multiple of 16 B => conflict-free b128 reads
    constexpr int CPR = BK / 4;                  // float4 chunks per row per K-step
    constexpr int APT = BM * CPR / THREADS;      // float4 of A per thread per chunk, all from one row
    constexpr int WPT = BN * CPR / THREADS;      // float4 of W per thread per chunk
    constexpr int TPR_A = CPR / APT, TPR_W = CPR / WPT;
    constexpr int BUF0 = (BM + BN) * LDS_LD;     // floats per stage: A image then W image
    constexpr int BUF = BUF0 > (THREADS / 64) * 32 * PLD ? BUF0 : (THREADS / 64) * 32 * PLD;   // also hosts the epilogue patches
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w % NWM, wn = w / NWM;
    const int la_row = tid / TPR_A, la_c = (tid % TPR_A) * APT * 4;       // loader: A row / first float
    const int lw_row = tid / TPR_W, lw_c = (tid % TPR_W) * WPT * 4;       // loader: W row / first float
    const int nk = (a.K + BK - 1) / BK;
    constexpr bool pro = PRO;

    // ---- per-tile state ------------------------------------------------------------------
    int tile = blockIdx.x;
    int row0 = 0, row_end = 0, n0 = 0, seg = 0;
    const float* aptr = a.A;  bool aval = false;
    const float* wptr = a.W;  bool wval = false;
    auto tile_geom = [&](int t, int& r0, int& rend, int& nn0, int& sg) {
        const int tm = t / ntn, tn = t % ntn;
        sg = seg_of_tile(a.seg, tm);
        r0 = a.seg.start[sg] + (tm - a.seg.tile0[sg]) * BM;
        rend = a.seg.start[sg + 1];
        nn0 = tn * BN;
    };
    // source row of this thread's A row: -1 = zero row (padding slot or past the segment end)
    auto a_source = [&](int r0, int rend) -> int {
        const int r = r0 + la_row;
        if (r >= rend) return -1;
        return a.gather ? a.gather[r] : r;
    };
    auto make_ptrs = [&](int src, int nn0, const float*& ap, bool& av, const float*& wp, bool& wv) {
        av = src >= 0;
        ap = a.A + (size_t)(av ? src : 0) * a.lda + la_c;
        const int n = nn0 + lw_row;
        wv = n < a.N;
        wp = a.W + (size_t)(wv ? n : 0) * a.ldw + lw_c;
    };

    float4 ra[APT], rw[WPT];
    auto gload = [&](int k0, const float* ap, bool av, const float* wp, bool wv, int sg) {
#pragma unroll
        for (int j = 0; j < APT; j++) {
            const bool kv = k0 + la_c + 4 * j < a.K;     // K is a multiple of 4: a tail chunk is zero-filled
            ra[j] = (av && kv) ? *reinterpret_cast<const float4*>(ap + k0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (pro && kv) {
                const float4 s = *reinterpret_cast<const float4*>(a.pscale + (size_t)sg * a.K + k0 + la_c + 4 * j);
                const float4 h = *reinterpret_cast<const float4*>(a.pshift + (size_t)sg * a.K + k0 + la_c + 4 * j);
                ra[j].x = fast_tanh(ra[j].x * s.x + h.x);
                ra[j].y = fast_tanh(ra[j].y * s.y + h.y);
                ra[j].z = fast_tanh(ra[j].z * s.z + h.z);
                ra[j].w = fast_tanh(ra[j].w * s.w + h.w);
            }
        }
#pragma unroll
        for (int j = 0; j < WPT; j++) {
            const bool kv = k0 + lw_c + 4 * j < a.K;
            rw[j] = (wv && kv) ? *reinterpret_cast<const float4*>(wp + k0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
        float* As = smem + buf * BUF;
        float* Ws = As + BM * LDS_LD;
#pragma unroll
        for (int j = 0; j < APT; j++) *reinterpret_cast<float4*>(&As[la_row * LDS_LD + la_c + 4 * j]) = ra[j];
#pragma unroll
        for (int j = 0; j < WPT; j++) *reinterpret_cast<float4*>(&Ws[lw_row * LDS_LD + lw_c + 4 * j]) = rw[j];
    };

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    };

    // aux operand of the d-activation epilogues, prefetched during the tile's LAST K-step in the
    // layout the staged epilogue reads: [nt][mt][i] -> rows er+8i of sub-tile (mt,nt), cols ec..ec+3
    constexpr bool HAS_AUX_ANY = EPI == NT_EPI_DTANH || EPI == NT_EPI_DTANH_BN || EPI == NT_EPI_DRELU;
    constexpr bool HAS_AUX_K = HAS_AUX_ANY && NWM * NWN == 8;      // register budget: the 8-wave kernel only
    constexpr int NAUX = HAS_AUX_K ? 16 : 1;
    float4 auxr[NAUX];
    const bool vec_k = ((a.N | a.ldc) & 3) == 0 && (!a.aux || (a.ldaux & 3) == 0);
    auto aux_prefetch = [&]() {
        if (!HAS_AUX_K) return;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int row = row0 + wm * 64 + mt * 32 + (lane >> 3) + 8 * i;
                    const int col = n0 + wn * 64 + nt * 32 + (lane & 7) * 4;
                    float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < row_end) {
                        if (vec_k && col + 3 < a.N) {
                            x4 = *reinterpret_cast<const float4*>(a.aux + (size_t)row * a.ldaux + col);
                        } else {
                            if (col + 0 < a.N) x4.x = a.aux[(size_t)row * a.ldaux + col + 0];
                            if (col + 1 < a.N) x4.y = a.aux[(size_t)row * a.ldaux + col + 1];
                            if (col + 2 < a.N) x4.z = a.aux[(size_t)row * a.ldaux + col + 2];
                            if (col + 3 < a.N) x4.w = a.aux[(size_t)row * a.ldaux + col + 3];
                        }
                    }
                    auxr[HAS_AUX_K ? (nt * 2 + mt) * 4 + i : 0] = x4;
                }
    };

    if (tile >= total_tiles) return;
    tile_geom(tile, row0, row_end, n0, seg);
    make_ptrs(a_source(row0, row_end), n0, aptr, aval, wptr, wval);
    gload(0, aptr, aval, wptr, wval, seg);
    lstore(0);
    __syncthreads();
    zero_acc();
    int cur = 0;
    const int frag = (lane & 31) * LDS_LD + 4 * (lane >> 5);

    while (true) {
        // next tile's identity (prefetched during the last chunk of this one)
        const int ntile = tile + gridDim.x;
        int nrow0 = 0, nrow_end = 0, nn0 = 0, nseg = 0;
        const float* naptr = a.A; bool naval = false;
        const float* nwptr = a.W; bool nwval = false;
        int nsrc = -1;                                        // gather index of the next tile: issued a tile early
        if (ntile < total_tiles) {
            tile_geom(ntile, nrow0, nrow_end, nn0, nseg);
            nsrc = a_source(nrow0, nrow_end);
        }

        for (int kt = 0; kt < nk; kt++) {
            bool loaded = false;
            if (kt + 1 == nk) aux_prefetch();               // older than the prefetch below: its wait never covers it
            if (kt + 1 < nk) { gload((kt + 1) * BK, aptr, aval, wptr, wval, seg); loaded = true; }
            else if (ntile < total_tiles) {
                make_ptrs(nsrc, nn0, naptr, naval, nwptr, nwval);
                gload(0, naptr, naval, nwptr, nwval, nseg);
                loaded = true;
            }
            const float* As = smem + cur * BUF + wm * 64 * LDS_LD + frag;
            const float* Ws = smem + cur * BUF + (BM + wn * 64) * LDS_LD + frag;
            // fragments of k-block kk+1 are requested before the 16 MFMAs of block kk are issued
            float4 fa0 = *reinterpret_cast<const float4*>(As), fa1 = *reinterpret_cast<const float4*>(As + 32 * LDS_LD);
            float4 fb0 = *reinterpret_cast<const float4*>(Ws), fb1 = *reinterpret_cast<const float4*>(Ws + 32 * LDS_LD);
#pragma unroll
            for (int kk = 0; kk < BK / 8; kk++) {
                float4 na0 = fa0, na1 = fa1, nb0 = fb0, nb1 = fb1;
                if (kk + 1 < BK / 8) {
                    na0 = *reinterpret_cast<const float4*>(As + (kk + 1) * 8);
                    na1 = *reinterpret_cast<const float4*>(As + 32 * LDS_LD + (kk + 1) * 8);
                    nb0 = *reinterpret_cast<const float4*>(Ws + (kk + 1) * 8);
                    nb1 = *reinterpret_cast<const float4*>(Ws + 32 * LDS_LD + (kk + 1) * 8);
                }
                const float av0[4] = {fa0.x, fa0.y, fa0.z, fa0.w}, av1[4] = {fa1.x, fa1.y, fa1.z, fa1.w};
                const float bv0[4] = {fb0.x, fb0.y, fb0.z, fb0.w}, bv1[4] = {fb1.x, fb1.y, fb1.z, fb1.w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    acc[0][0] = mfma32(av0[r], bv0[r], acc[0][0]);
                    acc[0][1] = mfma32(av0[r], bv1[r], acc[0][1]);
                    acc[1][0] = mfma32(av1[r], bv0[r], acc[1][0]);
                    acc[1][1] = mfma32(av1[r], bv1[r], acc[1][1]);
                }
                fa0 = na0; fa1 = na1; fb0 = nb0; fb1 = nb1;
                // keep the request one whole block ahead of its use (the scheduler otherwise sinks the
                // reads to just before the MFMA that consumes them and exposes the LDS latency)
                if (kk + 1 < BK / 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
            if (loaded) lstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }

        // ---- epilogue of `tile`.  Stage `cur` now holds the next tile's first chunk; the other
        // stage is free: each wave transposes its 32x32 sub-tiles through a private 32x36 patch.
        float* free_stage = smem + (cur ^ 1) * BUF;
        float* stg = free_stage + w * (32 * PLD);
        const int er = lane >> 3, ec = (lane & 7) * 4;          // staged read: rows er+8i, cols ec..ec+3
        const bool vec = ((a.N | a.ldc) & 3) == 0 && (!a.aux || (a.ldaux & 3) == 0);
        constexpr bool HAS_AUX = EPI == NT_EPI_DTANH || EPI == NT_EPI_DTANH_BN || EPI == NT_EPI_DRELU;
        float cs1[2][4], cs2[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) { cs1[nt][q] = 0.f; cs2[nt][q] = 0.f; }
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int col = n0 + wn * 64 + nt * 32 + ec;
            float bias[4], es[4], eh[4], mu[4], is[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const bool cv = col + q < a.N;
                bias[q] = (cv && a.bias) ? a.bias[col + q] : 0.f;
                es[q] = eh[q] = mu[q] = is[q] = 0.f;
                if (cv && EPI == NT_EPI_DTANH_BN) {
                    es[q] = a.escale[(size_t)seg * a.N + col + q];
                    eh[q] = a.eshift[(size_t)seg * a.N + col + q];
                }
                if (cv && STATS == NT_STAT_BNBWD) {
                    mu[q] = a.mean[(size_t)seg * a.N + col + q];
                    is[q] = a.invstd[(size_t)seg * a.N + col + q];
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; mt++) {
#pragma unroll
                for (int reg = 0; reg < 16; reg++)
                    stg[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * PLD + (lane & 31)] = acc[mt][nt][reg];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int row = row0 + wm * 64 + mt * 32 + er + 8 * i;
                    const float4 v4 = *reinterpret_cast<const float4*>(&stg[(er + 8 * i) * PLD + ec]);
                    float v[4] = {v4.x, v4.y, v4.z, v4.w};
                    float ax[4] = {0.f, 0.f, 0.f, 0.f};
                    const bool rok = row < row_end;
                    if (HAS_AUX_K) {
                        const float4 x4 = auxr[HAS_AUX_K ? (nt * 2 + mt) * 4 + i : 0];
                        ax[0] = x4.x; ax[1] = x4.y; ax[2] = x4.z; ax[3] = x4.w;
                    } else if (rok && HAS_AUX) {
                        if (vec && col + 3 < a.N) {
                            const float4 x4 = *reinterpret_cast<const float4*>(a.aux + (size_t)row * a.ldaux + col);
                            ax[0] = x4.x; ax[1] = x4.y; ax[2] = x4.z; ax[3] = x4.w;
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; q++)
                                if (col + q < a.N) ax[q] = a.aux[(size_t)row * a.ldaux + col + q];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float x = v[q] + bias[q];
                        switch (EPI) {
                            case NT_EPI_TANH: x = fast_tanh(x); break;
                            case NT_EPI_RELU: x = x > 0.f ? x : 0.f; break;
                            case NT_EPI_DTANH: x = x * (1.f - ax[q] * ax[q]); break;
                            case NT_EPI_DTANH_BN: { const float s = fast_tanh(ax[q] * es[q] + eh[q]); x = x * (1.f - s * s); break; }
                            case NT_EPI_DRELU: x = ax[q] > 0.f ? x : 0.f; break;
                            default: break;
                        }
                        v[q] = x;
                        if (rok && col + q < a.N) {
                            if (STATS == NT_STAT_SUMSQ) {
                                const float wt = row == a.seg.wrow ? a.seg.wmult : 1.f;   // the row that stands for many
                                cs1[nt][q] += wt * x; cs2[nt][q] += wt * x * x;
                            }
                            else if (STATS == NT_STAT_BNBWD) { cs1[nt][q] += x; cs2[nt][q] += x * ((ax[q] - mu[q]) * is[q]); }
                        }
                    }
                    if (rok) {
                        if (vec && col + 3 < a.N) {
                            *reinterpret_cast<float4*>(a.C + (size_t)row * a.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; q++)
                                if (col + q < a.N) a.C[(size_t)row * a.ldc + col + q] = v[q];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (STATS != NT_STAT_NONE) {
            // fold the 8 row groups of a wave (lane>>3), then the two M-waves through LDS
            __syncthreads();                                   // every wave is done with its staging patch
            float* red = free_stage;                           // [2 stats][NWM][BN]
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float s1 = cs1[nt][q], s2 = cs2[nt][q];
#pragma unroll
                    for (int o = 8; o <= 32; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                    if (lane < 8) {
                        const int c = wn * 64 + nt * 32 + lane * 4 + q;
                        red[(0 * NWM + wm) * BN + c] = s1;
                        red[(1 * NWM + wm) * BN + c] = s2;
                    }
                }
            __syncthreads();
            const int tile_m = tile / ntn;
            for (int c = tid; c < BN; c += THREADS)
                if (n0 + c < a.N) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int m = 0; m < NWM; m++) { s1 += red[(0 * NWM + m) * BN + c]; s2 += red[(1 * NWM + m) * BN + c]; }
                    a.stat_sum[(size_t)tile_m * a.N + n0 + c] = s1;
                    a.stat_aux[(size_t)tile_m * a.N + n0 + c] = s2;
                }
        }
        if (ntile >= total_tiles) break;
        __syncthreads();                                       // free stage is written again by the next lstore
        zero_acc();
        tile = ntile; row0 = nrow0; row_end = nrow_end; n0 = nn0; seg = nseg;
        aptr = naptr; aval = naval; wptr = nwptr; wval = nwval;
    }
}

static SegInfo retile(const SegInfo& in, int tile_rows) {
    SegInfo si = in;
    int t = 0;
    for (int i = 0; i < PC_MAX_SEG; i++) {
        si.tile0[i] = t;
        t += (si.start[i + 1] - si.start[i] + tile_rows - 1) / tile_rows;
    }
    si.tile0[PC_MAX_SEG] = t;
    return si;
}

template <bool PRO, int EPI, int STATS>
static void launch_variant(const NtArgs& a, int ntm, hipStream_t st) {
    if (PRO || a.N > 128 || STATS != NT_STAT_NONE) {
        // 128 rows x 256 columns, 8 waves, one workgroup per CU: A is read once
        const int ntn = (a.N + 255) / 256, total = ntm * ntn;
        PC_LAUNCH((gemm_nt_kernel<2, 4, 32, 2, PRO, EPI, STATS>), dim3(total < 256 ? total : 256), dim3(512), 0, st, a, ntn,
                  total);
    } else if (ntm >= 192) {
        // N <= 128: 128x128 tiles, two independent 4-wave workgroups per CU (measured 89 vs 77 TF/s
        // for one 256x128 8-wave workgroup: the second workgroup fills the first one's epilogue)
        PC_LAUNCH((gemm_nt_kernel<2, 2, 32, 2, false, EPI, NT_STAT_NONE>), dim3(ntm < 512 ? ntm : 512), dim3(256), 0, st, a,
                  1, ntm);
    } else {
        // few rows (per-sample projections of the attention block, joint-step layers): 64-row
        // tiles of 2 waves reach 2x the CUs
        NtArgs b = a;
        b.seg = retile(a.seg, 64);
        const int total = gemm_nt_tiles(b.seg);
        PC_LAUNCH((gemm_nt_kernel<1, 2, 32, 2, false, EPI, NT_STAT_NONE>), dim3(total), dim3(128), 0, st, b, 1, total);
    }
}

int launch_gemm_nt(const NtArgs& a, hipStream_t st) {
    if (!a.A || !a.W || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return PC_EINVAL;
    if (a.K % 4 != 0 || a.lda % 4 != 0 || a.ldw % 4 != 0) return PC_ESHAPE;
    if (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.C) & 15) return PC_ESHAPE;
    if (a.aux && ((uintptr_t)a.aux & 15)) return PC_ESHAPE;
    const int ntm = gemm_nt_tiles(a.seg);
    if (ntm <= 0) return PC_EINVAL;
    if (a.prologue == NT_PRO_BNTANH && (!a.pscale || !a.pshift)) return PC_EINVAL;
    const bool needs_aux = a.epilogue == NT_EPI_DTANH || a.epilogue == NT_EPI_DTANH_BN || a.epilogue == NT_EPI_DRELU;
    if (needs_aux && !a.aux) return PC_EINVAL;
    if (a.stats != NT_STAT_NONE && (a.N > 256 || !a.stat_sum || !a.stat_aux)) return PC_ESHAPE;
    // the fusions the two hot paths use (any other combination is refused)
    const int key = a.prologue * 100 + a.epilogue * 10 + a.stats;
    const int pb = pc_prof_begin(PC_KIND_GEMM_NT, 2.0 * a.M * (double)a.N * a.K, st);
    switch (key) {
        case 0:   launch_variant<false, NT_EPI_NONE, NT_STAT_NONE>(a, ntm, st); break;      // plain Linear / dX
        case 1:   launch_variant<false, NT_EPI_NONE, NT_STAT_SUMSQ>(a, ntm, st); break;     // Linear0 + BN sums
        case 110: launch_variant<true, NT_EPI_TANH, NT_STAT_NONE>(a, ntm, st); break;       // BN+tanh -> Linear3 -> tanh
        case 10:  launch_variant<false, NT_EPI_TANH, NT_STAT_NONE>(a, ntm, st); break;
        case 20:  launch_variant<false, NT_EPI_RELU, NT_STAT_NONE>(a, ntm, st); break;
        case 30:  launch_variant<false, NT_EPI_DTANH, NT_STAT_NONE>(a, ntm, st); break;     // dZ2
        case 42:  launch_variant<false, NT_EPI_DTANH_BN, NT_STAT_BNBWD>(a, ntm, st); break; // dZ1 + BN-backward sums
        case 50:  launch_variant<false, NT_EPI_DRELU, NT_STAT_NONE>(a, ntm, st); break;
        default:  return PC_ESHAPE;
    }
    pc_prof_end(pb, st);
    return pc_launch_status();
}
