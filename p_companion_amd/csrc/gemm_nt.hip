// C[M,N] = prologue(A)[M,K] . W[N,K]^T (+bias) -> epilogue: fp32 in, fp32 accumulate, fp32-grade products on the
// BF16 matrix cores (six v_mfma_f32_32x32x16_bf16 over a three-way bf16 split of both operands, common.h split3);
// the few-row kernels further down stay on v_mfma_f32_32x32x2_f32.
//
// One kernel family serves every "activation x weight^T" product of the two hot paths
// (nn.Linear forward, and dX = dY . W through a pre-transposed W): product2vec.py:14-21
// (ffn), nn.MultiheadAttention's in/out projections (:23-28), type_transition.py:11-12,
// item_prediction.py:11-20, p_companion.py:60-63 (similarities).
//
// These are SKINNY products: M = hundreds of thousands of rows, N,K <= 256: ~43 FLOP per HBM byte, just under the
// ridge of the six-product bf16 rate (52 FLOP/B), so the kernel is built around the row stream, not around K:
//   * a workgroup owns 128 rows x the FULL N (up to 256 columns: 8 waves as 2 x 4, each 64x64
//     = 2x2 32x32 accumulator blocks) so A is read from HBM exactly once;
//   * operands go global -> LDS DIRECTLY (global_load_lds_dwordx4, no VGPR hop and no ds_write):
//     measured on this loop, 138 TFLOP/s steady state against 104 for register staging
//     (scripts/microbench/nt_staging.hip).  A wave instruction drops 1 KB = 256/BK whole rows
//     into an UNPADDED stage; bank conflicts are avoided by an XOR swizzle applied on the global
//     side (each lane picks which 16-B chunk of its row it fetches).  The DMA is issued from inline
//     asm and waited for by hand (s_waitcnt vmcnt(0) before the stage's barrier): through the
//     builtin, the compiler's conservative LDS-alias tracking parks a vmcnt(0) in front of every
//     ds_read that follows an issue, which serialises the prefetch with the MFMA stream;
//   * workgroups are PERSISTENT over row tiles and the stage stream runs across tile boundaries
//     (the first chunk of the next tile -- including its gather indices -- is in flight during
//     the last chunk of this one); two workgroups share a CU so one's epilogue overlaps the
//     other's MFMA stream;
//   * rows of A may be gathered by index straight from the feature table (BPG neighbour
//     gather, -1 = zero row), tiles never straddle a BatchNorm segment, A can be transformed
//     (BN-apply + tanh, in place in LDS) and the epilogue can emit per-tile BatchNorm partial sums;
//   * the epilogue goes through per-wave LDS patches so that C (and the aux operand of the
//     d-activation epilogues) move as 16 B per lane, 8 full 128-B row segments per instruction.
// Operands are fetched with a permuted k order (lanes 0-31 take k = 8j..8j+3, lanes 32-63
// k = 8j+4..8j+7, identically for A and W) so one ds_read_b128 feeds four MFMAs.
#include "common.h"
#include "mfma16.h"

#define PLD 36          // row stride (floats) of the per-wave epilogue transposition patch
#define PROWS 16        // rows per patch: half a 32x32 accumulator block

int gemm_nt_tiles(const SegInfo& si) { return si.tile0[PC_MAX_SEG]; }

__device__ __attribute__((aligned(64))) float pc_zero_chunk[16];   // source of every zero-filled 16-B chunk

typedef __attribute__((address_space(3))) void* lptr_t;

// 64 lanes x 16 B from per-lane global addresses to LDS [lds_addr + 16 * lane].  In-order with every
// other vector-memory operation of the wave (vmcnt), so compiler-placed waits stay correct (at worst
// they wait for this too); the data is visible after s_waitcnt vmcnt(0) + a workgroup barrier.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(const float* gsrc, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop

// C and aux are streams (written / read once per launch, far larger than the 4 MB L2 of an XCD): moved
// with the non-temporal hint they do not push W and the A rows in flight out of L2 (measured on the dZ2
// GEMM: 99 -> 85 us, scripts/microbench/nt_deferred2.hip)
__device__ __forceinline__ void store_stream(float* p, const float4& v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f*>(p));
}
// The same store from inline asm (no compiler-placed wait around it).  Loads and stores do NOT retire in order with respect
// to each other on this hardware: "DMA, store, s_waitcnt vmcnt(1)" let a K-step end with its DMA still in flight (run-to-run
// different results, measured).  The rows a K-step transforms are therefore stored at the TOP of the next K-step, ahead of
// its DMA: the step's closing vmcnt(0) then waits for a store that has had a whole K-step to complete.
__device__ __attribute__((aligned(16))) float pc_store_sink[4];
__device__ __forceinline__ void store_stream_asm(float* p, const float4& v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f x = {v.x, v.y, v.z, v.w};
    // (s_nop: a VALU write of the data registers right behind a store of more than 64 bits is a hazard the compiler's recogniser
    // would cover for its own stores -- it does not look inside inline asm; the 4-wave prologue loop computes its next chunk
    // into the same registers at once, and without the wait states the stored rows changed from run to run)
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ float4 load_stream(const float* p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// LDS-only hand-off between waves: no global-memory fence (a __syncthreads() would also wait for
// this wave's outstanding C stores and for the next stage's DMA)
__device__ __forceinline__ void lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// NWM x NWN waves of 64x64 each: BM = 64*NWM rows, BN = 64*NWN columns.
//   <2,4> 128x256, 8 waves   (N > 128, prologue or statistics)
//   <2,2> 128x128, 4 waves   (N <= 128, many rows)
//   <1,2>  64x128, 2 waves   (few rows: more, smaller tiles fill more CUs)
// WPS = resident waves per SIMD the register budget is set for.
// PRO / EPI / STATS are compile-time: a runtime switch per output element costs ~1200 scalar
// branches per tile and wave (measured: 12 us of a 33 us tile) and the unused fusions' registers.
#ifdef PC_NT_TIMING
__device__ unsigned long long pc_nt_timing[32 * 4];
extern "C" int pc_debug_nt_timing(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[32 * 4] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(pc_nt_timing), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pc_nt_timing), sizeof(unsigned long long) * 32 * 4);
}
#endif
// NBN: 32-column blocks per wave (2: 64 x 64 wave tiles; 4: 64 x 128 -- six operand splits and twelve fragment reads per 48 products
// instead of eight and sixteen)
template <int NWM, int NWN, int BK, int WPS, bool PRO, int EPI, int STATS, int NBN = 2>
__global__ __launch_bounds__(64 * NWM * NWN, WPS) void gemm_nt_kernel(NtArgs a, int ntn, int total_tiles) {
    constexpr int NW = NWM * NWN, THREADS = 64 * NW;
    constexpr int WNC = 32 * NBN;                                 // columns per wave
    constexpr int BM = 64 * NWM, BN = WNC * NWN, SR = BM + BN;    // stage rows: A image then W image
    constexpr int CPR = BK / 4;                   // 16-B chunks per stage row
    constexpr int RB = 64 / BK;                   // stage rows per 256 B (one pass over the 64 banks)
    constexpr int RPI = 256 / BK;                 // stage rows one wave instruction fills (1 KB)
    constexpr int NI = SR / RPI / NW;             // DMA instructions per wave and stage
    constexpr int NIA = BM / RPI / NW;            // ... the first NIA of them fetch rows of A
    static_assert(SR % (RPI * NW) == 0 && BM % (RPI * NW) == 0, "row blocks must split evenly over the waves");
    static_assert(!PRO || (BM * CPR) % THREADS == 0, "in-place prologue: whole 16-B chunks per thread");
    constexpr int TPT = PRO ? BM * CPR / THREADS : 1;     // chunks of the A image a thread transforms per K-step (8 waves: 1, 4 waves: 2)
    constexpr int PATCH = NW * PROWS * PLD;
    constexpr int STAGE = SR * BK;                // floats per stage
    __shared__ __attribute__((aligned(1024))) float stages[2 * STAGE];
    __shared__ __attribute__((aligned(16))) float patch[PATCH];
    __shared__ float red[STATS != NT_STAT_NONE ? 2 * NWM * BN : 1];                           // [2 stats][NWM][BN]
    // SUMSQ: the multiplicities of the tile's rows (a row of the unique-row layout stands for several), requested a whole tile
    // ahead and parked here: read from memory inside the epilogue they were a dependent load per row and half-block -- 11.7 of
    // Linear0's 67.4 us (round 5)
    constexpr bool RWT = STATS == NT_STAT_SUMSQ;
    __shared__ float row_wt[RWT ? BM : 1];
    __shared__ __attribute__((aligned(16))) float pro_ss[PRO ? 2 * PC_MAX_SEG * 256 : 4];   // [seg][scale|shift][K]
    // DTANH_BN: the per-column BN scale/shift of the epilogue are read from LDS at their use, not held in 8
    // registers per lane through the whole epilogue (the accumulators leave no room for them)
    constexpr bool EBN = EPI == NT_EPI_DTANH_BN;
    __shared__ __attribute__((aligned(16))) float epi_ss[EBN ? 2 * PC_MAX_SEG * 256 : 4];   // [seg][scale|shift][N]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w % NWM, wn = w / NWM;     // wave-uniform: SGPRs
    const int nk = (a.K + BK - 1) / BK;
    // loader geometry: this lane's row inside a wave instruction's row block, and the chunk of that
    // row it fetches (slot ^ swizzle; the swizzle of a row is (row / RB) % CPR, the same for every j)
    const int lrow = lane / CPR;
    const int lchunk = (lane % CPR) ^ ((((w * RPI) + lrow) / RB) % CPR);
    const float* const zsrc = pc_zero_chunk;

    if (EBN) {
        for (int i = tid; i < a.seg.nseg * a.N; i += THREADS) {
            const int s = i / a.N, n = i - s * a.N;
            epi_ss[(2 * s) * 256 + n] = a.escale[i];
            epi_ss[(2 * s + 1) * 256 + n] = a.eshift[i];
        }
    }
    if (PRO) {
        for (int i = tid; i < a.seg.nseg * a.K; i += THREADS) {
            const int s = i / a.K, k = i - s * a.K;
            pro_ss[(2 * s) * 256 + k] = a.pscale[i];
            pro_ss[(2 * s + 1) * 256 + k] = a.pshift[i];
        }
    }

    // ---- per-tile state ------------------------------------------------------------------
    int tile = blockIdx.x;
    if (tile >= total_tiles) return;
#ifdef PC_EXP_STAGGER
    // developer experiment: the co-resident workgroups of a CU (b, b + 256, b + 512) start PC_EXP_STAGGER us apart
    for (int i = 0; i < ((blockIdx.x >> 8) % 3) * PC_EXP_STAGGER * 4; i++) __builtin_amdgcn_s_sleep(8);    // ~0.25 us each
#endif
    int row0 = 0, row_end = 0, n0 = 0, seg = 0;
    auto tile_geom = [&](int t, int& r0, int& rend, int& nn0, int& sg) {
        int tm = t / ntn, tn = t % ntn;
        // Two column tiles per row tile: workgroups b and b + 8 -- the same XCD, blockIdx round-robins over the 8 XCDs
        // and the grid is a multiple of 16 -- take the two halves of one row tile, so the second fetch of its A rows is
        // an L2 hit.  The host pads the tile count to a multiple of 16; a row tile past the end has no rows.
        if (ntn == 2) { tm = ((t >> 4) << 3) + (t & 7); tn = (t >> 3) & 1; }
        sg = seg_of_tile(a.seg, tm);
        r0 = a.seg.start[sg] + (tm - a.seg.tile0[sg]) * BM;
        rend = a.seg.start[sg + 1];
        nn0 = tn * BN;
    };
    // source rows of this lane's A chunks: -1 = zero row (padding slot or past the segment end)
    auto a_sources = [&](int r0, int rend, int (&srow)[NIA]) {
#pragma unroll
        for (int j = 0; j < NIA; j++) {
            const int r = r0 + (w + NW * j) * RPI + lrow;
            srow[j] = r < rend ? (a.gather ? a.gather[r] : r) : -1;
        }
    };
    const float* src[NI];          // this lane's source of each DMA instruction (k0 = 0), or null for a zero row
    auto make_ptrs = [&](const int (&srow)[NIA], int nn0) {
        // (the chunk offset is made opaque here: otherwise the loop-invariant 64-bit sums a.A + 4 lchunk / a.W + 4 lchunk are
        // hoisted into VGPR pairs that live through the K loops and the epilogue -- the one spill of the Linear0 + statistics
        // variant; rebuilt per tile they cost two 64-bit adds)
        int lchunk_ = lchunk;
        asm volatile("" : "+v"(lchunk_));
#pragma unroll
        for (int j = 0; j < NIA; j++)
#ifdef PC_EXP_DMA_L2
            src[j] = srow[j] >= 0 ? a.A + (size_t)(srow[j] & 127) * a.lda + lchunk_ * 4 : nullptr;
#else
            src[j] = srow[j] >= 0 ? a.A + (size_t)srow[j] * a.lda + lchunk_ * 4 : nullptr;
#endif
#pragma unroll
        for (int j = NIA; j < NI; j++) {
            const int n = nn0 + (w + NW * j) * RPI - BM + lrow;
            src[j] = n < a.N ? a.W + (size_t)n * a.ldw + lchunk_ * 4 : nullptr;
        }
    };
    const unsigned lds_w = (unsigned)(uintptr_t)(lptr_t)&stages[0] + w * (RPI * BK * 4);   // this wave's first row block
    auto issue = [&](int st, int k0) {
        const bool kin = k0 + lchunk * 4 < a.K;           // K is a multiple of 4: a tail chunk is zero-filled
        const unsigned dst = lds_w + st * (STAGE * 4);
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const float* p = (src[j] && kin) ? src[j] + k0 : zsrc;
            dma16(p, dst + j * (NW * RPI * BK * 4));
        }
    };

    f32x16 acc[2][NBN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < NBN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    };

    // fragment addresses: stage row (lane & 31) of the wave's 64-row strip, chunk (2 kk + (lane >> 5)) ^ swizzle
    const int fr = lane & 31;

    // One K-step = one k group of 16: lane (row fr, half lane >> 5) holds k = 8 (lane >> 5) .. + 7 of its rows, i.e. the
    // two 16-B chunks 2 (lane >> 5), + 1 of the stage row.  A blocks are split once, each W block as it is used;
    // the six products of every accumulator are issued term by term across the four accumulators (no back-to-back
    // dependent MFMAs).
    static_assert(BK == 16 || BK == 32, "k groups of 16");
    const int gsw = (fr / RB) % CPR;                                        // the row's chunk swizzle
    auto compute = [&](const float* cur) {
#pragma unroll
        for (int g = 0; g < BK / 16; g++) {
            const int c0 = (((4 * g + 2 * (lane >> 5)) ^ gsw) << 2), c1 = (((4 * g + 2 * (lane >> 5) + 1) ^ gsw) << 2);
            const float* ar = cur + (wm * 64 + fr) * BK;
            const float* br = cur + (BM + wn * WNC + fr) * BK;
            // developer builds (scripts/dev/nt_decompose.sh; WRONG results, right instruction mix): what each part of the
            // loop costs.  -DPC_EXP_NO_SPLIT fragments used unsplit, -DPC_EXP_NO_LDSREAD fragments from loop-invariant
            // registers, -DPC_EXP_NO_MFMA products replaced by a register keep-alive, -DPC_EXP_NO_DMA only the first
            // stage is ever fetched, -DPC_EXP_DMA_L2 the A rows come from 128 hot rows, -DPC_EXP_STAGGER=us the three
            // co-resident workgroups of a CU start that many microseconds apart
#if defined(PC_EXP_NO_LDSREAD)
#define PC_FRAG(PTR, OFF) make_float4(__int_as_float(a.M), __int_as_float(a.N), __int_as_float(a.K), __int_as_float(a.lda))
#else
#define PC_FRAG(PTR, OFF) (*reinterpret_cast<const float4*>((PTR) + (OFF)))
#endif
#if defined(PC_EXP_NO_SPLIT) || defined(PC_EXP_NO_LDSREAD)
#define PC_SPLIT(LO, HI) Split3{__builtin_bit_cast(bf16x8, LO), __builtin_bit_cast(bf16x8, HI), __builtin_bit_cast(bf16x8, LO)}
#else
#define PC_SPLIT(LO, HI) split3(LO, HI)
#endif
#if defined(PC_EXP_NO_MFMA)
#define PC_MFMA(A, B, C) ([&]() { asm volatile("" ::"v"(A), "v"(B)); return C; }())
#else
#define PC_MFMA(A, B, C) mfma_bf16(A, B, C)
#endif
            Split3 sa[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const float4 lo = PC_FRAG(ar + i * 32 * BK, c0), hi = PC_FRAG(ar + i * 32 * BK, c1);
                sa[i] = PC_SPLIT(lo, hi);
            }
#pragma unroll
            for (int j = 0; j < NBN; j++) {
                const float4 wlo = PC_FRAG(br + j * 32 * BK, c0), whi = PC_FRAG(br + j * 32 * BK, c1);
                const Split3 sb = PC_SPLIT(wlo, whi);
#define PC_TERM(PA, PB)                                                   \
                acc[0][j] = PC_MFMA(sa[0].PA, sb.PB, acc[0][j]);          \
                acc[1][j] = PC_MFMA(sa[1].PA, sb.PB, acc[1][j]);
                PC_PRIO_MFMA(PC_PRIO_NT_COND, 1);
                PC_TERM(p2, p0) PC_TERM(p0, p2) PC_TERM(p1, p1) PC_TERM(p1, p0) PC_TERM(p0, p1) PC_TERM(p0, p0)
                PC_PRIO_MFMA(PC_PRIO_NT_COND, 0);
            }
#undef PC_TERM
        }
    };

    float* pend_p = pc_store_sink;                   // pro_out: the chunk transformed in the previous K-step, not yet stored
    float4 pend_v = make_float4(0.f, 0.f, 0.f, 0.f);
    // BN-apply + tanh on the A image of a landed stage, in place (one 16-B chunk per thread)
    auto transform = [&](float* cur, int k0, int sg) {
#pragma unroll
        for (int u = 0; u < TPT; u++) {
            const int c = tid + THREADS * u;
            const int row = c / CPR, slot = c % CPR;
            const int k = k0 + ((slot ^ ((row / RB) % CPR)) << 2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < a.K) {
                v = *reinterpret_cast<const float4*>(&cur[c * 4]);
                const float4 s = *reinterpret_cast<const float4*>(&pro_ss[(2 * sg) * 256 + k]);
                const float4 h = *reinterpret_cast<const float4*>(&pro_ss[(2 * sg + 1) * 256 + k]);
                v.x = fast_tanh(v.x * s.x + h.x);
                v.y = fast_tanh(v.y * s.y + h.y);
                v.z = fast_tanh(v.z * s.z + h.z);
                v.w = fast_tanh(v.w * s.w + h.w);
                *reinterpret_cast<float4*>(&cur[c * 4]) = v;
            }
            if (a.pro_out) {                           // (wave-uniform) the transformed rows, for the weight gradient that follows
                const bool live = k < a.K && row0 + row < row_end;
                if (TPT == 1) {
                    pend_p = live ? a.pro_out + (size_t)(row0 + row) * a.ldpo + k : pc_store_sink;
                    pend_v = v;
                } else if (live && n0 == 0) {
                    // 4-wave tiles: the two column tiles of a row tile each transform their own copy of the A image; the first
                    // one stores the rows, at once (the K-step's closing vmcnt(0) covers the store; with three workgroups per CU
                    // somebody else's MFMAs run meanwhile)
                    store_stream_asm(a.pro_out + (size_t)(row0 + row) * a.ldpo + k, v);
                }
            }
        }
    };

    int nsrow[NIA];
    int ntile = tile, nrow0 = 0, nrow_end = 0, nn0 = 0, nseg = 0;
    // end of a K-step: this wave's DMA has landed, then every wave is past its reads of the old stage
    // (developer builds, wrong results: -DPC_EXP_NO_BARRIER the K-steps of a workgroup's waves run unsynchronised,
    // -DPC_EXP_NO_VMWAIT no wave waits for its stage requests)
    auto stage_sync = [&]() {
#ifndef PC_EXP_NO_VMWAIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifndef PC_EXP_NO_BARRIER
        __syncthreads();
#endif
    };

    tile_geom(tile, row0, row_end, n0, seg);
    a_sources(row0, row_end, nsrow);
    make_ptrs(nsrow, n0);
    issue(0, 0);
    zero_acc();
    if (RWT && tid < BM) row_wt[tid] = row0 + tid < row_end ? row_multiplicity(a.seg, row0 + tid) : 0.f;
    stage_sync();
    int cur = 0;

#ifdef PC_NT_TIMING
    unsigned long long tk = 0, te = 0, tn = 0;
#endif
    while (true) {
#ifdef PC_NT_TIMING
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
        // next tile's identity; its gather indices are requested a whole tile early
        ntile = tile + gridDim.x;
        float next_wt = 0.f;
        if (ntile < total_tiles) {
            tile_geom(ntile, nrow0, nrow_end, nn0, nseg);
            a_sources(nrow0, nrow_end, nsrow);
            if (RWT && tid < BM && nrow0 + tid < nrow_end) next_wt = row_multiplicity(a.seg, nrow0 + tid);
        }
        // one K-step: start the DMA of the following chunk (this tile's, or the next tile's first),
        // then multiply the landed one
        for (int kt = 0; kt < nk; kt++) {
            if (PRO && TPT == 1 && a.pro_out && kt > 0) store_stream_asm(pend_p, pend_v);
#ifndef PC_EXP_NO_DMA
            if (kt + 1 < nk) issue(cur ^ 1, (kt + 1) * BK);
            else if (ntile < total_tiles) { make_ptrs(nsrow, nn0); issue(cur ^ 1, 0); }
#endif
            float* cs = stages + cur * STAGE;
            if (PRO) { transform(cs, kt * BK, seg); lds_sync(); }
            compute(cs);
            stage_sync();
            cur ^= 1;
        }
        if (PRO && TPT == 1 && a.pro_out) store_stream_asm(pend_p, pend_v);       // the last K-step's rows

#ifdef PC_NT_TIMING
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
        // ---- epilogue of `tile`: each wave transposes its accumulators, 16 rows x 32 columns at a
        // time, through a private patch (the stages are busy: the next tile's first chunk is landing)
        float* stg = patch + w * (PROWS * PLD);
        const int er = lane >> 3, ec = (lane & 7) * 4;          // staged read: rows er+8i, cols ec..ec+3
        const bool vec = ((a.N | a.ldc) & 3) == 0 && (!a.aux || (a.ldaux & 3) == 0);
        constexpr bool HAS_AUX = EPI == NT_EPI_DTANH || EPI == NT_EPI_DTANH_BN || EPI == NT_EPI_DRELU;
        // aux operand (d-activation epilogues): ADEPTH half-patches are in flight ahead of the one being
        // finished -- with two loads per lane and half-patch the wave would otherwise pay one full
        // memory latency per half-patch (measured: 19 us of a 32 us tile)
        // (three with the statistics epilogue: the fourth pair of aux registers was what spilled there -- 8 B of scratch)
        constexpr int ADEPTH = HAS_AUX ? (STATS != NT_STAT_NONE ? 3 : 4) : 1;
        const bool fast = vec && n0 + BN <= a.N;                 // whole tile inside N: unpredicated 16-B accesses
        float4 xq[ADEPTH][2];
        auto aux_load = [&](int h, float4 (&x4)[2]) {
            const int nt = h >> 2, mt = (h >> 1) & 1, half = h & 1;
            const int col = n0 + wn * WNC + nt * 32 + ec;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int row = row0 + wm * 64 + mt * 32 + half * 16 + er + 8 * i;
                x4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef PC_EXP_NO_AUX
                x4[i] = make_float4(0.25f, 0.5f, 0.125f, 0.75f);      // (developer experiment: no aux loads; WRONG results)
                continue;
#endif
                if (fast) {
                    const int rc = row < row_end ? row : row_end - 1;      // clamped: the value of a dead row is never used
                    x4[i] = load_stream(a.aux + (size_t)rc * a.ldaux + col);
                } else if (row < row_end) {
                    if (col + 0 < a.N) x4[i].x = a.aux[(size_t)row * a.ldaux + col + 0];
                    if (col + 1 < a.N) x4[i].y = a.aux[(size_t)row * a.ldaux + col + 1];
                    if (col + 2 < a.N) x4[i].z = a.aux[(size_t)row * a.ldaux + col + 2];
                    if (col + 3 < a.N) x4[i].w = a.aux[(size_t)row * a.ldaux + col + 3];
                }
            }
        };
        if (HAS_AUX) {
#pragma unroll
            for (int h = 0; h < ADEPTH; h++) aux_load(h, xq[h]);
        }
        float bias[4], cs1[4], cs2[4];
#pragma unroll
        for (int h = 0; h < 4 * NBN; h++) {
            const int nt = h >> 2, mt = (h >> 1) & 1, half = h & 1;
            const int col = n0 + wn * WNC + nt * 32 + ec;
            if ((h & 3) == 0) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool cv = col + q < a.N;
                    cs1[q] = cs2[q] = 0.f;
                    bias[q] = (!HAS_AUX && cv && a.bias) ? a.bias[col + q] : 0.f;     // the d-activation epilogues carry no bias
                }
            }
            const int rbase = row0 + wm * 64 + mt * 32 + half * 16 + er;
#pragma unroll
            for (int q8 = 0; q8 < 8; q8++)
                stg[((q8 & 3) + 8 * (q8 >> 2) + 4 * (lane >> 5)) * PLD + (lane & 31)] = acc[mt][nt][half * 8 + q8];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int row = rbase + 8 * i;
                const float4 v4 = *reinterpret_cast<const float4*>(&stg[(er + 8 * i) * PLD + ec]);
                float v[4] = {v4.x, v4.y, v4.z, v4.w};
                const float4 xa = xq[HAS_AUX ? h % ADEPTH : 0][i];
                const float ax[4] = {xa.x, xa.y, xa.z, xa.w};
                const bool rok = row < row_end;
                float es[4] = {0.f, 0.f, 0.f, 0.f}, eh[4] = {0.f, 0.f, 0.f, 0.f};
                if (EBN) {
                    int eo = (2 * seg) * 256 + (fast ? col : 0);          // (N <= 256 and the tile spans it when this epilogue runs)
                    asm volatile("" : "+v"(eo));                          // opaque: re-read here, do not cache in registers
                    const float4 s4 = *reinterpret_cast<const float4*>(&epi_ss[eo]);
                    const float4 h4 = *reinterpret_cast<const float4*>(&epi_ss[eo + 256]);
                    es[0] = s4.x; es[1] = s4.y; es[2] = s4.z; es[3] = s4.w;
                    eh[0] = h4.x; eh[1] = h4.y; eh[2] = h4.z; eh[3] = h4.w;
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
                            const float wt = row_wt[RWT ? row - row0 : 0];            // a row may stand for many
                            cs1[q] += wt * x; cs2[q] += wt * x * x;
                        }
                        else if (STATS == NT_STAT_BNBWD) { cs1[q] += x; cs2[q] += x * ax[q]; }      // raw moment: centred in fp64 by the finalize pass
                    }
                }
#ifdef PC_EXP_NO_CSTORE
                if (rok && v[0] == 12345.678f) {                      // (developer experiment: the C stores never execute; WRONG results)
#else
                if (rok) {
#endif
                    if (fast) {
                        store_stream(a.C + (size_t)row * a.ldc + col, make_float4(v[0], v[1], v[2], v[3]));
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            if (col + q < a.N) a.C[(size_t)row * a.ldc + col + q] = v[q];
                    }
                }
            }
            if (HAS_AUX && h + ADEPTH < 4 * NBN) aux_load(h + ADEPTH, xq[h % ADEPTH]);
            __builtin_amdgcn_wave_barrier();
            if (STATS != NT_STAT_NONE && (h & 3) == 3) {
                // fold the 8 row groups of the wave (lane >> 3); the M-waves meet in `red`
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float s1 = cs1[q], s2 = cs2[q];
#pragma unroll
                    for (int o = 8; o <= 32; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                    if (lane < 8) {
                        const int c = wn * WNC + nt * 32 + lane * 4 + q;
                        red[(0 * NWM + wm) * BN + c] = s1;
                        red[(1 * NWM + wm) * BN + c] = s2;
                    }
                }
            }
        }
        if (STATS != NT_STAT_NONE) {
            lds_sync();                                        // every wave has posted its column sums
            const int tile_m = ntn == 2 ? ((tile >> 4) << 3) + (tile & 7) : tile / ntn;
            const bool live = row0 < row_end;                  // (a padding tile of the paired numbering owns no statistics slot)
            for (int c = tid; c < BN; c += THREADS)
                if (live && n0 + c < a.N) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int m = 0; m < NWM; m++) { s1 += red[(0 * NWM + m) * BN + c]; s2 += red[(1 * NWM + m) * BN + c]; }
                    a.stat_sum[(size_t)tile_m * a.N + n0 + c] = s1;
                    a.stat_aux[(size_t)tile_m * a.N + n0 + c] = s2;
                }
            // (`red` is written again only after the next tile's K-steps and their barriers)
        }
#ifdef PC_NT_TIMING
        {
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
            tk += t1 - t0; te += t2 - t1; tn += 1;
            if (ntile >= total_tiles && NWM == 2 && tid == 0) {
                const int id = (EPI + 6 * (STATS != 0) + 12 * (PRO ? 1 : 0)) * 4;
                atomicAdd(&pc_nt_timing[id], tk); atomicAdd(&pc_nt_timing[id + 1], te); atomicAdd(&pc_nt_timing[id + 2], tn); atomicAdd(&pc_nt_timing[id + 3], 1ull);
            }
        }
#endif
        if (ntile >= total_tiles) break;
        zero_acc();
        if (RWT) {
            // (every wave has passed the lds_sync of the statistics fold, i.e. its last read of row_wt; the next tile's first
            // stage_sync makes the new values visible long before its epilogue)
            if (tid < BM) row_wt[tid] = next_wt;
        }
        tile = ntile; row0 = nrow0; row_end = nrow_end; n0 = nn0; seg = nseg;
        // the DMA source pointers (two registers each) are rebuilt from the row indices rather than kept alive
        // across the epilogue, which is where the register budget is tightest
#pragma unroll
        for (int j = 0; j < NIA; j++) asm volatile("" : "+v"(nsrow[j]));
        make_ptrs(nsrow, n0);
    }
}

#ifdef PC_CHAIN_TIMING
// developer build: shader-clock stamps of workgroup 0 / thread 0 at the phases of the few-row bodies (scripts/dev/chain_phase_times.py)
__device__ unsigned long long pc_chain_timing[64];
__device__ int pc_chain_slot;
extern "C" int pc_debug_chain_timing(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[64] = {}; int zs = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(pc_chain_slot), &zs, sizeof(int)); return (int)hipMemcpyToSymbol(HIP_SYMBOL(pc_chain_timing), z, sizeof(z)); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pc_chain_timing), sizeof(unsigned long long) * 64);
}
#define PC_CT() do { if (blockIdx.x == 0 && threadIdx.x == 0 && pc_chain_slot < 64) pc_chain_timing[pc_chain_slot++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PC_CT() do { } while (0)
#endif

// ---------------------------------------------------------------------------------------
// Few rows (the per-sample projections of the attention block: M = batch; every layer of the joint
// step): the product is a few hundred MFLOP and the persistent kernel's K pipeline is pure latency
// (4 dependent stage round trips for K = 128).  Here a 4-wave workgroup owns 32 rows x up to 128
// columns, requests the WHOLE K extent of its A rows and of W in one burst of LDS-DMA, waits once,
// and each wave multiplies its 32x32 block.  K <= 128 (256: two bursts), N <= 128; no prologue / statistics.
// Stage rows are KS = 32/64/128 floats (K rounded up), chunks XOR-swizzled as above.
// (EPI: a compile-time constant in the single-product kernel, the member's own value in the grouped one -- the
// kernel is latency-bound, a uniform switch per element costs nothing)
__device__ __forceinline__ void nt_small_body(const NtArgs& a, int ks_log2, int tile, const int EPI) {
    extern __shared__ __attribute__((aligned(1024))) float sm_small[];
    const int KS = 1 << ks_log2, CPR = KS >> 2;                  // floats / 16-B chunks per stage row
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sg = seg_of_tile(a.seg, tile);
    const int row0 = a.seg.start[sg] + (tile - a.seg.tile0[sg]) * 32, row_end = a.seg.start[sg + 1];
    auto swz = [&](int r) { return KS == 32 ? (r >> 1) & 7 : r & 15; };

    // ---- one burst per 128 k's (K <= 128: one): stage rows [0,32) = A rows of the tile, [32,160) = W rows; 1 KB per wave
    // instruction.  K in (128, 256] -- the per-head products of the attention block at D = 256 -- runs two bursts over the same
    // stage, the accumulators carry over
    const int rows_per_instr = 256 >> ks_log2, ninstr = (160 * KS) >> 8;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)&sm_small[0];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    const float* As = sm_small + fr * KS;
    const float* Ws = sm_small + (32 + w * 32 + fr) * KS;
    const int sa = swz(fr), sw = swz(32 + w * 32 + fr);
    for (int k0 = 0; k0 < a.K; k0 += 128) {
        if (k0) __syncthreads();                                 // every wave is past its reads of the previous 128 k's
        for (int g = w; g < ninstr; g += 4) {
            const int r = g * rows_per_instr + lane / CPR;       // stage row
            const int chunk = (lane % CPR) ^ swz(r);
            const float* p = pc_zero_chunk;
            if (k0 + chunk * 4 < a.K) {
                if (r < 32) {
                    const int gr = row0 + r;
                    const int srow = gr < row_end ? (a.gather ? a.gather[gr] : gr) : -1;
                    if (srow >= 0) p = a.A + (size_t)srow * a.lda + k0 + chunk * 4;
                } else if (r - 32 < a.N) {
                    p = a.W + (size_t)(r - 32) * a.ldw + k0 + chunk * 4;
                }
            }
            dma16(p, lds0 + g * 1024);
        }
        PC_CT();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        PC_CT();
        const int kleft = a.K - k0;
        const int nkk = ((kleft < 128 ? kleft : 128) + 7) >> 3;
        for (int kk = 0; kk < nkk; kk++) {
            const int c = 2 * kk + fh;
            const float4 fa4 = *reinterpret_cast<const float4*>(&As[(c ^ sa) << 2]);
            const float4 fb4 = *reinterpret_cast<const float4*>(&Ws[(c ^ sw) << 2]);
            acc = mfma32(fa4.x, fb4.x, acc);
            acc = mfma32(fa4.y, fb4.y, acc);
            acc = mfma32(fa4.z, fb4.z, acc);
            acc = mfma32(fa4.w, fb4.w, acc);
        }
    }
    __syncthreads();                                             // the stage becomes the epilogue patches
    PC_CT();

    float* stg = sm_small + w * (32 * PLD);
#pragma unroll
    for (int reg = 0; reg < 16; reg++)
        stg[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * PLD + (lane & 31)] = acc[reg];
    __builtin_amdgcn_wave_barrier();
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const int col = w * 32 + ec;
    const bool vec = ((a.N | a.ldc) & 3) == 0 && (!a.aux || (a.ldaux & 3) == 0);
    const bool HAS_AUX = EPI == NT_EPI_DTANH || EPI == NT_EPI_DRELU;
    float bias[4];
#pragma unroll
    for (int q = 0; q < 4; q++) bias[q] = (!HAS_AUX && a.bias && col + q < a.N) ? a.bias[col + q] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = row0 + er + 8 * i;
        if (row >= row_end) continue;
        const float4 v4 = *reinterpret_cast<const float4*>(&stg[(er + 8 * i) * PLD + ec]);
        float v[4] = {v4.x, v4.y, v4.z, v4.w}, ax[4] = {0.f, 0.f, 0.f, 0.f};
        if (HAS_AUX) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (col + q < a.N) ax[q] = a.aux[(size_t)row * a.ldaux + col + q];
        }
        const float rs = a.brs ? a.brs[(size_t)row * a.ldbrs] : 1.f;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float x = v[q] + (a.brs ? bias[q] * rs : bias[q]);
            switch (EPI) {
                case NT_EPI_TANH: x = fast_tanh(x); break;
                case NT_EPI_RELU: x = x > 0.f ? x : 0.f; break;
                case NT_EPI_DTANH: x = x * (1.f - ax[q] * ax[q]); break;
                case NT_EPI_DRELU: x = ax[q] > 0.f ? x : 0.f; break;
                default: break;
            }
            v[q] = x;
        }
        if (vec && col + 3 < a.N) {
            *reinterpret_cast<float4*>(a.C + (size_t)row * a.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (col + q < a.N) a.C[(size_t)row * a.ldc + col + q] = v[q];
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_small_kernel(NtArgs a, int ks_log2) {
    nt_small_body(a, ks_log2, blockIdx.x, EPI);
}

// Up to PC_NT_GROUP mutually independent few-row products in one launch (the joint step's encoder next to the item
// projection, dE_c next to the hidden-layer gradient): workgroups [block0[j], block0[j+1]) run product j.
__global__ __launch_bounds__(256) void gemm_nt_small_group_kernel(NtSmallGroup g) {
    const int b = blockIdx.x;
    int j = 0;
#pragma unroll
    for (int i = 1; i < PC_NT_GROUP; i++) j += (i < g.n && b >= g.block0[i]) ? 1 : 0;
    nt_small_body(g.a[j], g.ks_log2[j], b - g.block0[j], g.a[j].epilogue);
}

static SegInfo retile_plain(int M) {
    SegInfo si = make_seginfo(nullptr, M, 32);
    return si;
}

// ---------------------------------------------------------------------------------------
// Chains of few-row products over the SAME 32-row tile (the attention block of Product2Vec at D = 128, M = batch):
//   forward   q = e_a Wq^T + bq  ->  qt_h = q_h Wk_h            |  ctx_h = c_h Wv_h^T + bv_h sp_h  ->  out = ctx Wo^T + bo
//   backward  dctx = dout Wo     ->  dc_h = dctx_h Wv_h          |  dq_h = dqt_h Wk_h^T             ->  dquery = dq Wq
// Each pair was two launches of ~10 us of pure latency (plus the boundary between them); a workgroup owns its 32 rows in
// both products, so the second one follows in the same kernel: its A rows are the C rows this workgroup has just stored
// (s_waitcnt vmcnt(0) + barrier: the stores are complete, the rows come back through the CU's own L1 / L2).
// The per-head block products run one head per wave:
//   KHEAD  C[r][h*128 + d] = sum_{j<32} A[r][32h + j] W[d][32h + j]     (A 32 x 128, W 128 x 128 staged like a plain
//          product; wave h multiplies its head's 32 k's against all four column blocks; ldc = 512)
//   AHEAD  C[r][32h + j]   = sum_{d<128} A[r][h*128 + d] W[32h + j][d]  (+ bias[32h + j] * brs[r][h]; A is four staged
//          32 x 128 images, one per head; wave h takes image h and its own 32 output columns)
struct NtChain { NtArgs a[2]; int mode[2], epi[2], n, tiles; HingeMeanJob rider; LossPro loss; };

// The burst goes global -> registers -> LDS here (every lane up to 32 independent 16-B loads in flight, then the
// swizzled ds_write_b128s): measured with in-kernel clocks (scripts/dev/chain_phase_times.py), a 4-wave workgroup issuing its
// 20-32 LDS-DMA instructions per wave spent 4-12 us in the issue -- the DMA path holds only a few requests per wave in
// flight, which the persistent kernels hide behind twelve waves per CU and this one-shot burst cannot.
__device__ __forceinline__ void nt_head_body(const NtArgs& a, int tile, int mode) {
    extern __shared__ __attribute__((aligned(1024))) float sm_small[];
    constexpr int KS = 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = tile * 32, row_end = a.M;
    const int arows = mode == NT_MODE_AHEAD ? 128 : 32;          // staged A rows (AHEAD: image h at rows [32h, 32h + 32))
    {
        constexpr int MAXC = 32;                                 // 16-B chunks per thread: (arows + 128) * 32 / 256 = 20 or 32
        const int nchunk = (arows + 128) >> 3;
        float4 buf[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; i++) {
            buf[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nchunk) {
                const int id = tid + 256 * i, r = id >> 5, cc = id & 31;   // stage row, 16-B chunk of the row
                const float* p = nullptr;
                if (r < arows) {
                    const int gr = row0 + (r & 31);
                    if (gr < row_end) p = a.A + (size_t)gr * a.lda + (mode == NT_MODE_AHEAD ? (r >> 5) * 128 : 0) + cc * 4;
                } else {
                    p = a.W + (size_t)(r - arows) * a.ldw + cc * 4;
                }
                if (p) buf[i] = *reinterpret_cast<const float4*>(p);
            }
        }
        PC_CT();
#pragma unroll
        for (int i = 0; i < MAXC; i++)
            if (i < nchunk) {
                const int id = tid + 256 * i, r = id >> 5, cc = id & 31;
                *reinterpret_cast<float4*>(&sm_small[r * KS + ((cc ^ (r & 15)) << 2)]) = buf[i];
            }
    }
    __syncthreads();
    PC_CT();

    const int fr = lane & 31, fh = lane >> 5;
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const int nblk = mode == NT_MODE_KHEAD ? 4 : 1;
    f32x16 acc[4];
#pragma unroll
    for (int nb = 0; nb < 4; nb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[nb][r] = 0.f;
    if (mode == NT_MODE_KHEAD) {
        const float* As = sm_small + fr * KS;
        const int sa = fr & 15;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int c = 2 * (4 * w + kk) + fh;
            const float4 fa4 = *reinterpret_cast<const float4*>(&As[(c ^ sa) << 2]);
#pragma unroll
            for (int nb = 0; nb < 4; nb++) {
                const int wr = 32 + nb * 32 + fr;
                const float4 fb4 = *reinterpret_cast<const float4*>(&sm_small[wr * KS + ((c ^ (wr & 15)) << 2)]);
                acc[nb] = mfma32(fa4.x, fb4.x, acc[nb]);
                acc[nb] = mfma32(fa4.y, fb4.y, acc[nb]);
                acc[nb] = mfma32(fa4.z, fb4.z, acc[nb]);
                acc[nb] = mfma32(fa4.w, fb4.w, acc[nb]);
            }
        }
    } else {
        // PLAIN: every wave reads the one A image; AHEAD: wave h reads image h.  Two accumulators (even / odd k groups)
        // halve the dependent MFMA chain
        const int ar = (mode == NT_MODE_AHEAD ? w * 32 : 0) + fr, wr = arows + w * 32 + fr;
        const float* As = sm_small + ar * KS;
        const float* Ws = sm_small + wr * KS;
        const int sa = ar & 15, sw = wr & 15;
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            const int c = 2 * kk + fh;
            const float4 fa4 = *reinterpret_cast<const float4*>(&As[(c ^ sa) << 2]);
            const float4 fb4 = *reinterpret_cast<const float4*>(&Ws[(c ^ sw) << 2]);
            acc[kk & 1] = mfma32(fa4.x, fb4.x, acc[kk & 1]);
            acc[kk & 1] = mfma32(fa4.y, fb4.y, acc[kk & 1]);
            acc[kk & 1] = mfma32(fa4.z, fb4.z, acc[kk & 1]);
            acc[kk & 1] = mfma32(fa4.w, fb4.w, acc[kk & 1]);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) acc[0][r] += acc[1][r];
    }
    __syncthreads();                                             // the stage becomes the epilogue patches
    PC_CT();
    float* stg = sm_small + w * (32 * PLD);
#pragma unroll
    for (int nb = 0; nb < 4; nb++) {
        if (nb >= nblk) break;
#pragma unroll
        for (int reg = 0; reg < 16; reg++)
            stg[((reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) * PLD + (lane & 31)] = acc[nb][reg];
        __builtin_amdgcn_wave_barrier();
        const int col = mode == NT_MODE_KHEAD ? w * 128 + nb * 32 + ec : w * 32 + ec;
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        if (mode != NT_MODE_KHEAD && a.bias) {
#pragma unroll
            for (int q = 0; q < 4; q++) bias[q] = a.bias[col + q];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = row0 + er + 8 * i;
            if (row >= row_end) continue;
            float4 v4 = *reinterpret_cast<const float4*>(&stg[(er + 8 * i) * PLD + ec]);
            const float rs = (mode == NT_MODE_AHEAD && a.brs) ? a.brs[(size_t)row * a.ldbrs + w] : 1.f;
            v4.x += bias[0] * rs; v4.y += bias[1] * rs; v4.z += bias[2] * rs; v4.w += bias[3] * rs;
            *reinterpret_cast<float4*>(a.C + (size_t)row * a.ldc + col) = v4;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void gemm_nt_chain_kernel(NtChain c) {
    if ((int)blockIdx.x >= c.tiles) {                            // the rider's workgroup (see HingeMeanJob)
        extern __shared__ __attribute__((aligned(1024))) float sm_small[];
        hinge_mean_body(c.rider, sm_small);
        return;
    }
    PC_CT();
    for (int s = 0; s < c.n; s++) {
        PC_CT();
        if (s) {                                                 // this workgroup's C rows of stage s - 1 are stage s's A rows
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        PC_CT();
        nt_head_body(c.a[s], blockIdx.x, c.mode[s]);
    }
}

// The same two chains on 16-row tiles and v_mfma_f32_16x16x4_f32 (round 4): 256 workgroups at B = 4096 instead of 128 (the
// 32-row bodies above were bound by the matrix pipes of the half of the chip that held a tile: 2.7 us of MFMA per body),
// the weight fragments of BOTH stages requested at kernel entry straight from L2 into registers (no LDS image of W: 64 KB of
// staging per stage gone), and the first stage's result handed to the second through LDS (it is also stored: q, ctx, dctx and dq
// are read again later in the step) instead of through a store / vmcnt(0) / reload round trip.
//   AHEAD_FIRST = false:  PLAIN -> KHEAD   (q -> qt, dctx -> dc)
//   AHEAD_FIRST = true:   AHEAD -> PLAIN   (c -> ctx -> out, dqt -> dq -> dquery)
// LossPro (common.h): wave w of a 16-sample chain tile takes samples 4 w .. 4 w + 3, one per 16-lane group, all sixteen of the
// tile at once (triplet_sample16: the stand-alone kernel's definition -- the same bits).  (A first version gave every sample a
// whole wave, four in turn: the fused launch then lasted exactly as long as the two it replaced.)
#define LP_MAX_K PC_LOSS_MAX_K
__device__ __forceinline__ void loss_prologue16(const LossPro& lp, int row0, float* sA, int lda, int w, int lane) {
    const int g = lane >> 4, l16 = lane & 15, row = 4 * w + g, b = row0 + row;
    const bool live = b < lp.B;
    float ga[8];
    triplet_sample16(lp.emb, lp.pos, lp.neg, b, lp.B, lp.K, lp.margin, live, lp.d_pos, lp.d_neg, lp.dp, lp.dn, true, l16, ga);
    *reinterpret_cast<float4*>(&sA[row * lda + 8 * l16]) = make_float4(ga[0], ga[1], ga[2], ga[3]);
    *reinterpret_cast<float4*>(&sA[row * lda + 8 * l16 + 4]) = make_float4(ga[4], ga[5], ga[6], ga[7]);
    if (live) {
        *reinterpret_cast<float4*>(lp.demb + (size_t)b * 128 + 8 * l16) = make_float4(ga[0], ga[1], ga[2], ga[3]);
        *reinterpret_cast<float4*>(lp.demb + (size_t)b * 128 + 8 * l16 + 4) = make_float4(ga[4], ga[5], ga[6], ga[7]);
    }
}

#define CH_LDA 132          /* LDS row strides: width + 4 floats */
#define CH_LDA4 516
// one two-stage chain over the 16-row tile at row0 (sA: 16 x CH_LDA4 floats when AHEAD_FIRST, else 16 x CH_LDA; sS: 16 x CH_LDA)
template <bool AHEAD_FIRST>
__device__ __forceinline__ void chain16_body(const NtArgs& a0, const NtArgs& a1, const LossPro& loss, float* sA, float* sS, int row0) {
    const int tid = threadIdx.x, lane = tid & 63, ci = lane & 15, rh = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a0.M;
    auto lds_sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // a PLAIN stage over the 16 x 128 LDS tile `src`: wave w owns the column blocks 2 w, 2 w + 1
    // (biases and per-row bias scales are requested at entry with the weight fragments: a load in an epilogue is one more
    // L2 round trip on the critical path of a kernel that is nothing but such round trips)
    auto plain = [&](const NtArgs& a, const BFrag<8> (&f)[2], const float (&bias2)[2], const float* src, float* keep) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            mul_b<8, 1>(src, CH_LDA, 1, f[j], acc, lane);
            const int col = 16 * (2 * w + j) + ci;
            const float bias = bias2[j];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * rh + r;
                const float v = acc[0][r] + bias;
                if (keep) keep[row * CH_LDA + col] = v;
                if (row0 + row < M) a.C[(size_t)(row0 + row) * a.ldc + col] = v;
            }
        }
    };
    if (!AHEAD_FIRST) {
        BFrag<8> fp[2];
        BFrag<2> fk[8];
#pragma unroll
        for (int j = 0; j < 2; j++) fp[j] = load_b<128, false>(a0.W, a0.ldw, 16 * (2 * w + j), 128, lane);
#pragma unroll
        for (int nb = 0; nb < 8; nb++) fk[nb] = load_b<32, false>(a1.W + 32 * w, a1.ldw, 16 * nb, 128, lane);
        const float bp[2] = {a0.bias ? a0.bias[16 * (2 * w) + ci] : 0.f, a0.bias ? a0.bias[16 * (2 * w + 1) + ci] : 0.f};
        if (loss.emb) {
            // the tile's A rows are d(loss)/d(anchor embedding) of its 16 samples: formed here (LossPro), behind the weight requests
            loss_prologue16(loss, row0, sA, CH_LDA, w, lane);
        } else {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int e = tid + 256 * u, r = e >> 5, c4 = (e & 31) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row0 + r < M) v = *reinterpret_cast<const float4*>(a0.A + (size_t)(row0 + r) * a0.lda + c4);
                *reinterpret_cast<float4*>(&sA[r * CH_LDA + c4]) = v;
            }
        }
        lds_sync();
        plain(a0, fp, bp, sA, sS);
        lds_sync();
        // KHEAD: wave h multiplies its head's 32 k's against all eight column blocks; C[r][128 h + d]
#pragma unroll
        for (int nb = 0; nb < 8; nb++) {
            f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            mul_b<2, 1>(sS + 32 * w, CH_LDA, 1, fk[nb], acc, lane);
            const int col = 128 * w + 16 * nb + ci;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * rh + r;
                if (row0 + row < M) a1.C[(size_t)(row0 + row) * a1.ldc + col] = acc[0][r];
            }
        }
    } else {
        BFrag<8> fa[2], fp[2];
#pragma unroll
        for (int j = 0; j < 2; j++) fa[j] = load_b<128, false>(a0.W, a0.ldw, 32 * w + 16 * j, 128, lane);
#pragma unroll
        for (int j = 0; j < 2; j++) fp[j] = load_b<128, false>(a1.W, a1.ldw, 16 * (2 * w + j), 128, lane);
        const float bp[2] = {a1.bias ? a1.bias[16 * (2 * w) + ci] : 0.f, a1.bias ? a1.bias[16 * (2 * w + 1) + ci] : 0.f};
        const float ba[2] = {a0.bias ? a0.bias[32 * w + ci] : 0.f, a0.bias ? a0.bias[32 * w + 16 + ci] : 0.f};
        float rs4[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
            rs4[r] = (a0.brs && row0 + 4 * rh + r < M) ? a0.brs[(size_t)(row0 + 4 * rh + r) * a0.ldbrs + w] : 1.f;
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = tid + 256 * u, r = e >> 7, c4 = (e & 127) * 4;
            x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < M) x[u] = *reinterpret_cast<const float4*>(a0.A + (size_t)(row0 + r) * a0.lda + c4);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = tid + 256 * u, r = e >> 7, c4 = (e & 127) * 4;
            *reinterpret_cast<float4*>(&sA[r * CH_LDA4 + c4]) = x[u];
        }
        lds_sync();
        // AHEAD: wave h takes image h (columns [128 h, 128 h + 128) of the tile) and its own 32 output columns
#pragma unroll
        for (int j = 0; j < 2; j++) {
            f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            mul_b<8, 1>(sA + 128 * w, CH_LDA4, 1, fa[j], acc, lane);
            const int col = 32 * w + 16 * j + ci;
            const float bias = ba[j];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * rh + r;
                const bool live = row0 + row < M;
                const float v = acc[0][r] + bias * rs4[r];
                sS[row * CH_LDA + col] = v;
                if (live) a0.C[(size_t)(row0 + row) * a0.ldc + col] = v;
            }
        }
        lds_sync();
        plain(a1, fp, bp, sS, nullptr);
    }
}

template <bool AHEAD_FIRST>
__global__ __launch_bounds__(256) void gemm_nt_chain16_kernel(NtChain c) {
    __shared__ __attribute__((aligned(16))) float sA[16 * (AHEAD_FIRST ? CH_LDA4 : CH_LDA)];
    __shared__ __attribute__((aligned(16))) float sS[16 * CH_LDA];
    if ((int)blockIdx.x >= c.tiles) {                            // the rider's workgroup (see HingeMeanJob)
        __shared__ float red[256];
        hinge_mean_body(c.rider, red);
        return;
    }
    chain16_body<AHEAD_FIRST>(c.a[0], c.a[1], c.loss, sA, sS, blockIdx.x * 16);
}

// The out-projection's forward chain (ctx = c Wv^T ..., out = ctx Wo^T + bo), the triplet hinge and the backward chain
// (d_ctx = d_out Wo, d_c = d_ctx Wv_h) of the fused Product2Vec step as ONE launch (round 6): everything in them is per SAMPLE,
// so the 16 samples of a tile run all four products and their loss back to back -- the rows a stage needs from the one before
// it were written by this workgroup (global memory behind a workgroup-scope release / acquire, or the LDS tiles).
struct NtChainPair { NtArgs a[4]; int tiles; LossPro loss; };
__global__ __launch_bounds__(256) void gemm_nt_chain16_pair_kernel(NtChainPair c) {
    __shared__ __attribute__((aligned(16))) float sA[16 * CH_LDA4];
    __shared__ __attribute__((aligned(16))) float sS[16 * CH_LDA];
    const LossPro none = {};
    chain16_body<true>(c.a[0], c.a[1], none, sA, sS, blockIdx.x * 16);
    // the embedding rows of this tile (a[1].C) are the loss's input: written above by this workgroup's waves
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    chain16_body<false>(c.a[2], c.a[3], c.loss, sA, sS, blockIdx.x * 16);
}

// The same chains at PRODUCT_EMB_DIM = 256 (BASELINE configs[4]; head dim 64): the attention block's per-sample products ran as
// eight grouped few-row launches of ~20 us there.  A product's weight fragments (K = 256: 64 VGPRs per 16-column block) no
// longer fit the register file all at once, so a wave streams its column blocks from L2 one ahead of the block being multiplied;
// LDS: the 16 x 1024 input tile of the AHEAD stage (66 KB) + the 16 x 256 hand-over tile.
#define CH_LDB 260          /* 256-wide LDS rows */
#define CH_LDB4 1028        /* 1024-wide */
template <bool AHEAD_FIRST>
__global__ __launch_bounds__(256) void gemm_nt_chain16_d256_kernel(NtChain c) {
    extern __shared__ __attribute__((aligned(16))) float ch_lds[];
    float* sS = ch_lds;                                          // [16][CH_LDB]
    float* sA = ch_lds + 16 * CH_LDB;                            // [16][CH_LDB4] or [16][CH_LDB]
    if ((int)blockIdx.x >= c.tiles) {
        hinge_mean_body(c.rider, ch_lds);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, ci = lane & 15, rh = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const NtArgs& a0 = c.a[0];
    const NtArgs& a1 = c.a[1];
    const int row0 = blockIdx.x * 16, M = a0.M;
    auto lds_sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // PLAIN over the 16 x 256 LDS tile `src`: wave w owns the column blocks 4 w .. 4 w + 3, fragments one block ahead
    // (two blocks' fragments in flight: 128 VGPRs; the biases with them)
    auto plain = [&](const NtArgs& a, const float* src, float* keep) {
        BFrag<16> f[2];
        float bias4[4];
#pragma unroll
        for (int j = 0; j < 2; j++) f[j] = load_b<256, false>(a.W, a.ldw, 16 * (4 * w + j), 256, lane);
#pragma unroll
        for (int j = 0; j < 4; j++) bias4[j] = a.bias ? a.bias[16 * (4 * w + j) + ci] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int col = 16 * (4 * w + j) + ci;
            f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            mul_b<16, 1>(src, CH_LDB, 1, f[j & 1], acc, lane);
            if (j + 2 < 4) f[j & 1] = load_b<256, false>(a.W, a.ldw, 16 * (4 * w + j + 2), 256, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * rh + r;
                const float v = acc[0][r] + bias4[j];
                if (keep) keep[row * CH_LDB + col] = v;
                if (row0 + row < M) a.C[(size_t)(row0 + row) * a.ldc + col] = v;
            }
        }
    };
    if (!AHEAD_FIRST) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = tid + 256 * u, r = e >> 6, c4 = (e & 63) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < M) v = *reinterpret_cast<const float4*>(a0.A + (size_t)(row0 + r) * a0.lda + c4);
            *reinterpret_cast<float4*>(&sA[r * CH_LDB + c4]) = v;
        }
        lds_sync();
        plain(a0, sA, sS);
        lds_sync();
        // KHEAD: wave h multiplies its head's 64 k's against all sixteen column blocks; C[r][256 h + d]
        // (requested BEFORE the first stage's epilogue would be better still; here: groups of four blocks, two groups in flight)
        BFrag<4> fk[2][4];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int u = 0; u < 4; u++) fk[q][u] = load_b<64, false>(a1.W + 64 * w, a1.ldw, 16 * (4 * q + u), 256, lane);
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int nb = 4 * gq + u;
                f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}};
                mul_b<4, 1>(sS + 64 * w, CH_LDB, 1, fk[gq & 1][u], acc, lane);
                const int col = 256 * w + 16 * nb + ci;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 4 * rh + r;
                    if (row0 + row < M) a1.C[(size_t)(row0 + row) * a1.ldc + col] = acc[0][r];
                }
            }
            if (gq + 2 < 4) {
#pragma unroll
                for (int u = 0; u < 4; u++) fk[gq & 1][u] = load_b<64, false>(a1.W + 64 * w, a1.ldw, 16 * (4 * (gq + 2) + u), 256, lane);
            }
        }
    } else {
        float rs4[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
            rs4[r] = (a0.brs && row0 + 4 * rh + r < M) ? a0.brs[(size_t)(row0 + 4 * rh + r) * a0.ldbrs + w] : 1.f;
        BFrag<16> f[2];
#pragma unroll
        for (int j = 0; j < 2; j++) f[j] = load_b<256, false>(a0.W, a0.ldw, 64 * w + 16 * j, 256, lane);
        float ba4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) ba4[j] = a0.bias ? a0.bias[64 * w + 16 * j + ci] : 0.f;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = tid + 256 * (8 * half + u), r = e >> 8, c4 = (e & 255) * 4;
                x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row0 + r < M) x[u] = *reinterpret_cast<const float4*>(a0.A + (size_t)(row0 + r) * a0.lda + c4);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = tid + 256 * (8 * half + u), r = e >> 8, c4 = (e & 255) * 4;
                *reinterpret_cast<float4*>(&sA[r * CH_LDB4 + c4]) = x[u];
            }
        }
        lds_sync();
        // AHEAD: wave h takes image h (columns [256 h, 256 h + 256) of the tile) and its own 64 output columns
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int col = 64 * w + 16 * j + ci;
            f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            mul_b<16, 1>(sA + 256 * w, CH_LDB4, 1, f[j & 1], acc, lane);
            if (j + 2 < 4) f[j & 1] = load_b<256, false>(a0.W, a0.ldw, 64 * w + 16 * (j + 2), 256, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * rh + r;
                const float v = acc[0][r] + ba4[j] * rs4[r];
                sS[row * CH_LDB + col] = v;
                if (row0 + row < M) a0.C[(size_t)(row0 + row) * a0.ldc + col] = v;
            }
        }
        lds_sync();
        plain(a1, sS, nullptr);
    }
}

// The attention chains: every stage M rows (the same M); D = 128 or 256: K = N = D per plain stage / head structure above.
int launch_gemm_nt_chain(const NtArgs* args, const int* modes, int n, hipStream_t st, const HingeMeanJob* rider, const LossPro* loss) {
    if (!args || !modes || n < 1 || n > 2) return PC_EINVAL;
    NtChain c = {};
    if (loss) {
        // the loss prologue serves the attention backward's first chain at PRODUCT_EMB_DIM = 128: PLAIN (d_out -> d_ctx) then KHEAD,
        // 16-row tiles, the prologue's rows ARE the first product's A operand
        if (n != 2 || modes[0] != NT_MODE_PLAIN || modes[1] != NT_MODE_KHEAD || args[0].N != 128 || args[0].K != 128 || args[0].brs)
            return PC_ESHAPE;
        if (!loss->emb || !loss->pos || !loss->neg || !loss->d_pos || !loss->d_neg || !loss->dp || !loss->dn || !loss->demb) return PC_EINVAL;
        if (loss->B != args[0].M || loss->K < 1 || loss->K > LP_MAX_K || loss->demb != args[0].A || args[0].lda != 128) return PC_EINVAL;
        c.loss = *loss;
    }
    c.n = n;
    double flops = 0.0;
    for (int i = 0; i < n; i++) {
        const NtArgs& a = args[i];
        if (!a.A || !a.W || !a.C || a.M <= 0 || a.M != args[0].M) return PC_EINVAL;
        if (a.gather || a.prologue != NT_PRO_NONE || a.stats != NT_STAT_NONE || a.epilogue != NT_EPI_NONE) return PC_ESHAPE;
        if (a.lda % 4 || a.ldw % 4 || a.ldc % 4 || (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.C) & 15)) return PC_ESHAPE;
        if (args[0].N == 256 || args[0].K == 1024) {                 // the D = 256 chains (16-row tiles only)
            if (modes[i] == NT_MODE_PLAIN) { if (a.N != 256 || a.K != 256 || a.brs) return PC_ESHAPE; flops += 2.0 * a.M * 256 * 256; }
            else if (modes[i] == NT_MODE_KHEAD) { if (a.N != 1024 || a.K != 256 || a.bias) return PC_ESHAPE; flops += 2.0 * a.M * 1024 * 64; }
            else if (modes[i] == NT_MODE_AHEAD) { if (a.N != 256 || a.K != 1024) return PC_ESHAPE; flops += 2.0 * a.M * 256 * 256; }
            else return PC_EINVAL;
            c.a[i] = a;
            c.a[i].seg = retile_plain(a.M);
            c.mode[i] = modes[i];
            c.epi[i] = NT_EPI_NONE;
            continue;
        }
        if (modes[i] == NT_MODE_PLAIN) { if (a.N != 128 || a.K != 128 || a.brs) return PC_ESHAPE; flops += 2.0 * a.M * 128 * 128; }
        else if (modes[i] == NT_MODE_KHEAD) { if (a.N != 512 || a.K != 128 || a.bias) return PC_ESHAPE; flops += 2.0 * a.M * 512 * 32; }
        else if (modes[i] == NT_MODE_AHEAD) { if (a.N != 128 || a.K != 512) return PC_ESHAPE; flops += 2.0 * a.M * 128 * 128; }
        else return PC_EINVAL;
        c.a[i] = a;
        c.a[i].seg = retile_plain(a.M);
        c.mode[i] = modes[i];
        c.epi[i] = NT_EPI_NONE;
    }
    const int tiles = (args[0].M + 31) / 32;
    c.tiles = tiles;
    if (rider) {
        if (!rider->d_pos || !rider->d_neg || !rider->loss || rider->B <= 0) return PC_EINVAL;
        c.rider = *rider;
    }
    const bool two = n == 2 && ((modes[0] == NT_MODE_PLAIN && modes[1] == NT_MODE_KHEAD) || (modes[0] == NT_MODE_AHEAD && modes[1] == NT_MODE_PLAIN)) &&
                     !args[1].brs && !(modes[0] == NT_MODE_PLAIN && args[0].brs);
    if (args[0].N == 256 || args[0].K == 1024) {
        if (!two) return PC_ESHAPE;
        c.tiles = (args[0].M + 15) / 16;
        if (rider) {
            if (!rider->d_pos || !rider->d_neg || !rider->loss || rider->B <= 0) return PC_EINVAL;
            c.rider = *rider;
        }
        const size_t lds16 = (size_t)16 * (CH_LDB + (modes[0] == NT_MODE_AHEAD ? CH_LDB4 : CH_LDB)) * 4;
        static const hipError_t a16[2] = {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_chain16_d256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
            hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_chain16_d256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)};
        (void)a16;
        const int pbd = pc_prof_begin(PC_KIND_GEMM_NT_SMALL, flops, st);
        if (modes[0] == NT_MODE_PLAIN) PC_LAUNCH(gemm_nt_chain16_d256_kernel<false>, dim3(c.tiles + (rider ? 1 : 0)), dim3(256), lds16, st, c);
        else PC_LAUNCH(gemm_nt_chain16_d256_kernel<true>, dim3(c.tiles + (rider ? 1 : 0)), dim3(256), lds16, st, c);
        pc_prof_end(pbd, st);
        return pc_launch_status();
    }
    if (two) {
        // the two chains of the attention block: 16-row tiles
        c.tiles = (args[0].M + 15) / 16;
        const int pb16 = pc_prof_begin(PC_KIND_GEMM_NT_SMALL, flops, st);
        if (modes[0] == NT_MODE_PLAIN) PC_LAUNCH(gemm_nt_chain16_kernel<false>, dim3(c.tiles + (rider ? 1 : 0)), dim3(256), 0, st, c);
        else PC_LAUNCH(gemm_nt_chain16_kernel<true>, dim3(c.tiles + (rider ? 1 : 0)), dim3(256), 0, st, c);
        pc_prof_end(pb16, st);
        return pc_launch_status();
    }
    const size_t lds = (size_t)256 * 128 * 4;                    // AHEAD: 128 A-image rows + 128 W rows of 512 B
    static const hipError_t lds_attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_chain_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 256 * 128 * 4);
    (void)lds_attr;
    const int pb = pc_prof_begin(PC_KIND_GEMM_NT_SMALL, flops, st);
    PC_LAUNCH(gemm_nt_chain_kernel, dim3(tiles + (rider ? 1 : 0)), dim3(256), lds, st, c);
    pc_prof_end(pb, st);
    return pc_launch_status();
}

// fwd: {AHEAD (c -> ctx), PLAIN (ctx -> out)}; bwd: {PLAIN (d_out -> d_ctx), KHEAD (d_ctx -> d_c)}; loss->emb == fwd[1].C and
// loss->demb == bwd[0].A (gemm_nt_chain16_pair_kernel)
int launch_gemm_nt_chain_pair(const NtArgs* fwd, const NtArgs* bwd, const LossPro* loss, hipStream_t st) {
    if (!fwd || !bwd || !loss) return PC_EINVAL;
    NtChainPair c = {};
    double flops = 0.0;
    for (int i = 0; i < 4; i++) {
        const NtArgs& a = i < 2 ? fwd[i] : bwd[i - 2];
        if (!a.A || !a.W || !a.C || a.M <= 0 || a.M != fwd[0].M) return PC_EINVAL;
        if (a.gather || a.prologue != NT_PRO_NONE || a.stats != NT_STAT_NONE || a.epilogue != NT_EPI_NONE) return PC_ESHAPE;
        if (a.lda % 4 || a.ldw % 4 || a.ldc % 4 || (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.C) & 15)) return PC_ESHAPE;
        c.a[i] = a;
        c.a[i].seg = retile_plain(a.M);
    }
    if (fwd[0].N != 128 || fwd[0].K != 512 || fwd[1].N != 128 || fwd[1].K != 128 || fwd[1].brs) return PC_ESHAPE;      // AHEAD, PLAIN
    if (bwd[0].N != 128 || bwd[0].K != 128 || bwd[0].brs || bwd[1].N != 512 || bwd[1].K != 128 || bwd[1].bias || bwd[1].brs) return PC_ESHAPE;   // PLAIN, KHEAD
    if (!loss->emb || !loss->pos || !loss->neg || !loss->d_pos || !loss->d_neg || !loss->dp || !loss->dn || !loss->demb) return PC_EINVAL;
    if (loss->B != fwd[0].M || loss->K < 1 || loss->K > LP_MAX_K || loss->demb != bwd[0].A || bwd[0].lda != 128 ||
        loss->emb != fwd[1].C || fwd[1].ldc != 128)
        return PC_EINVAL;
    c.loss = *loss;
    c.tiles = (fwd[0].M + 15) / 16;
    flops = 2.0 * fwd[0].M * (3.0 * 128 * 128 + 512 * 32);
    const int pb = pc_prof_begin(PC_KIND_GEMM_NT_SMALL, flops, st);
    PC_LAUNCH(gemm_nt_chain16_pair_kernel, dim3(c.tiles), dim3(256), 0, st, c);
    pc_prof_end(pb, st);
    return pc_launch_status();
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
    if (a.N > 128 && a.N <= 256 && ntm >= 192 && (!PRO || (EPI == NT_EPI_TANH && STATS == NT_STAT_NONE))) {
        // many rows, 129..256 columns, no in-place prologue: 128x128 tiles of 4 waves, two column tiles per row tile,
        // THREE workgroups per CU.  A is fetched twice (the second time mostly from L2 / MALL), but one workgroup's
        // epilogue -- 30-54 % of a tile's time, during which its waves only move data -- overlaps the K loops of the
        // other two: dZ1 133 -> 115 us, dZ2 79 -> 70, Linear0 71 -> 67 against the 8-wave full-width tile (step 1.135 ->
        // 1.115 ms on the same box; two 4-wave workgroups with 256 registers each: no better than the 8-wave tile).
        if constexpr (PRO) {
            // Linear3 (BN + tanh prologue, light epilogue; until round 5 an 8-wave 128 x 256 tile, ONE workgroup per CU whose waves split,
            // multiplied and waited in step: 124.4 us): 128 x 256 tiles of four waves with 64 x 128 wave tiles (NBN = 4), two
            // workgroups per CU -- the whole N in one workgroup (A fetched and transformed once), six operand splits and twelve
            // fragment reads per 48 products instead of eight and sixteen: 110.1 us, against 114.6 as paired 128 x 128 tiles
            // (round 5, same box).  The epilogue-heavy products lose with two workgroups per CU (Linear0 68.5 -> 74.7, dZ2 70.8 ->
            // 83.4, dZ1 109.0 -> 124.1: one workgroup less to cover an epilogue) and keep the paired tiles.
            PC_LAUNCH((gemm_nt_kernel<2, 2, 16, 2, PRO, EPI, STATS, 4>), dim3(ntm < 512 ? ntm : 512), dim3(256), 0, st, a, 1, ntm);
        } else {
            const int total = (2 * ntm + 15) & ~15;                  // paired numbering (tile_geom): whole groups of 16
            PC_LAUNCH((gemm_nt_kernel<2, 2, 16, 3, PRO, EPI, STATS>), dim3(total < 768 ? total : 768), dim3(256), 0, st, a,
                      2, total);
        }
        return;
    }
    if (PRO || a.N > 128 || STATS != NT_STAT_NONE) {
        // 128 rows x 256 columns, 8 waves with 256 VGPRs, one workgroup per CU: A is read once
        const int ntn = (a.N + 255) / 256;
        const int total = ntn == 2 ? ((ntm * 2 + 15) & ~15) : ntm * ntn;      // (two column tiles: paired numbering, tile_geom)
        PC_LAUNCH((gemm_nt_kernel<2, 4, PRO ? 16 : 32, 2, PRO, EPI, STATS>), dim3(total < 256 ? total : 256), dim3(512), 0, st, a, ntn,
                  total);
    } else if (ntm >= 192) {
        // N <= 128: 128x128 tiles, 4 waves, three workgroups per CU
        PC_LAUNCH((gemm_nt_kernel<2, 2, 16, 3, false, EPI, NT_STAT_NONE>), dim3(ntm < 768 ? ntm : 768), dim3(256), 0, st, a,
                  1, ntm);
    } else if (a.K <= 256 && (EPI == NT_EPI_NONE || EPI == NT_EPI_TANH || EPI == NT_EPI_RELU || EPI == NT_EPI_DTANH ||
                              EPI == NT_EPI_DRELU)) {
        // few rows, short K: one LDS-DMA burst per 32-row tile and 128 k's, no K pipeline
        NtArgs b = a;
        b.seg = retile(a.seg, 32);
        const int total = gemm_nt_tiles(b.seg);
        const int ks_log2 = a.K <= 32 ? 5 : a.K <= 64 ? 6 : 7;
        constexpr int SEPI = (EPI == NT_EPI_DTANH_BN) ? NT_EPI_NONE : EPI;
        const size_t lds = (size_t)160 * (1 << ks_log2) * 4 > (size_t)4 * 32 * PLD * 4 ? (size_t)160 * (1 << ks_log2) * 4
                                                                                   : (size_t)4 * 32 * PLD * 4;
        static const hipError_t lds_attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_small_kernel<SEPI>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 128 * 4);
        (void)lds_attr;                                          // 80 KB of dynamic LDS at KS = 128 (> the 64 KB default cap)
        PC_LAUNCH((gemm_nt_small_kernel<SEPI>), dim3(total), dim3(256), lds, st, b, ks_log2);
    } else {
        // few rows, long K: 64-row tiles of 2 waves reach 2x the CUs
        NtArgs b = a;
        b.seg = retile(a.seg, 64);
        const int total = gemm_nt_tiles(b.seg);
        // (DTANH_BN never comes here -- it carries statistics and took the branch above; not instantiating it keeps the
        // build free of an occupancy remark about a kernel nobody launches)
        if constexpr (EPI != NT_EPI_DTANH_BN)
            PC_LAUNCH((gemm_nt_kernel<1, 2, 32, 2, false, EPI, NT_STAT_NONE>), dim3(total), dim3(128), 0, st, b, 1, total);
    }
}

static bool nt_small_ok(const NtArgs& a);
int launch_gemm_nt(const NtArgs& a, hipStream_t st) {
    if (!a.A || !a.W || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return PC_EINVAL;
    if (a.K % 4 != 0 || a.lda % 4 != 0 || a.ldw % 4 != 0) return PC_ESHAPE;
    if (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.C) & 15) return PC_ESHAPE;
    if (a.aux && ((uintptr_t)a.aux & 15)) return PC_ESHAPE;
    const int ntm = gemm_nt_tiles(a.seg);
    if (a.brs && !nt_small_ok(a)) return PC_ESHAPE;               // per-row bias scale: few-row kernel only
    if (ntm <= 0) return PC_EINVAL;
    if (a.prologue == NT_PRO_BNTANH && (!a.pscale || !a.pshift)) return PC_EINVAL;
    if (a.prologue == NT_PRO_BNTANH && a.K > 256) return PC_ESHAPE;      // scale/shift live in LDS
    const bool needs_aux = a.epilogue == NT_EPI_DTANH || a.epilogue == NT_EPI_DTANH_BN || a.epilogue == NT_EPI_DRELU;
    if (needs_aux && (!a.aux || a.bias)) return PC_EINVAL;
    if (a.epilogue == NT_EPI_DTANH_BN && (a.N > 256 || a.N % 4 || !a.escale || !a.eshift)) return PC_ESHAPE;   // scale/shift live in LDS
    if (a.stats != NT_STAT_NONE && (a.N > 256 || !a.stat_sum || !a.stat_aux)) return PC_ESHAPE;
    // the fusions the two hot paths use (any other combination is refused)
    const int key = a.prologue * 100 + a.epilogue * 10 + a.stats;
    // few-row launches (latency-bound, a few us each) are booked apart from the persistent kernel family
    const bool small_m = ntm < 192;                        // (M < 24.5 k rows: the attention block's per-sample products at any width)
    const int pb = pc_prof_begin(small_m ? PC_KIND_GEMM_NT_SMALL : PC_KIND_GEMM_NT, 2.0 * a.M * (double)a.N * a.K, st);
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


static bool nt_small_ok(const NtArgs& a) {
    if (!a.A || !a.W || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return false;
    if (a.K % 4 != 0 || a.lda % 4 != 0 || a.ldw % 4 != 0) return false;
    if (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.C) & 15) return false;
    if (a.aux && ((uintptr_t)a.aux & 15)) return false;
    if (a.prologue != NT_PRO_NONE || a.stats != NT_STAT_NONE || a.N > 128 || a.K > 256) return false;   // (K > 128: two bursts)
    const int e = a.epilogue;
    if (!(e == NT_EPI_NONE || e == NT_EPI_TANH || e == NT_EPI_RELU || e == NT_EPI_DTANH || e == NT_EPI_DRELU)) return false;
    if ((e == NT_EPI_DTANH || e == NT_EPI_DRELU) && (!a.aux || a.bias)) return false;
    return gemm_nt_tiles(a.seg) < 192;
}

// Products that do not fit the few-row kernel are launched on their own.
int launch_gemm_nt_group(const NtArgs* args, int n, hipStream_t st) {
    if (!args || n < 1 || n > PC_NT_GROUP) return PC_EINVAL;
    NtSmallGroup g = {};
    int blocks = 0;
    size_t lds = 0;
    double flops = 0.0;
    for (int i = 0; i < n; i++) {
        if (!nt_small_ok(args[i])) { PC_TRY(launch_gemm_nt(args[i], st)); continue; }
        const int k = g.n++;
        g.a[k] = args[i];
        g.a[k].seg = retile(args[i].seg, 32);
        g.ks_log2[k] = args[i].K <= 32 ? 5 : args[i].K <= 64 ? 6 : 7;
        g.block0[k] = blocks;
        blocks += gemm_nt_tiles(g.a[k].seg);
        const size_t need = (size_t)160 * (1 << g.ks_log2[k]) * 4 > (size_t)4 * 32 * PLD * 4 ? (size_t)160 * (1 << g.ks_log2[k]) * 4
                                                                                          : (size_t)4 * 32 * PLD * 4;
        if (need > lds) lds = need;
        flops += 2.0 * args[i].M * (double)args[i].N * args[i].K;
    }
    if (g.n == 0) return PC_OK;
    for (int k = g.n; k <= PC_NT_GROUP; k++) g.block0[k] = blocks;
    static const hipError_t lds_attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_small_group_kernel),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 128 * 4);
    (void)lds_attr;
    const int pb = pc_prof_begin(PC_KIND_GEMM_NT_SMALL, flops, st);
    PC_LAUNCH(gemm_nt_small_group_kernel, dim3(blocks), dim3(256), lds, st, g);
    pc_prof_end(pb, st);
    return pc_launch_status();
}
