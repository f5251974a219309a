// P-Companion joint step, fused (train.py:42-48): the whole per-sample part of
// PCompanion.forward + compute_loss + backward (p_companion.py:45-119, type_transition.py:15-20,
// item_prediction.py:22-40) is ONE kernel over 16-sample tiles; the weight / table gradients are one grouped
// "rows^T x rows" launch over the row buffers that kernel leaves behind; one more kernel sums their slabs in fixed
// order, forms the losses and (optionally) applies Adam.  Three launches per step instead of ~20 dependent few-us
// ones (the step moves ~11 MB: it is bound by kernel boundaries and ramps, not by bandwidth).
//
//   joint_tile_kernel   per 16-sample tile, 4 waves, every activation in LDS, weights streamed from L2:
//       t = E_q[qt], q = E_prod[qi]                                   (row gathers, p_companion.py:51,54)
//       h = dropout(relu(enc t)), c = dec h                           (type_transition.py:17-19)
//       sims = c E_c^T, top-K (ties -> lower index), e_k = E_c[top_k] (p_companion.py:60-65)
//       pi = itm q, tp_k = typ e_k, proj_k = pi * tp_k                (item_prediction.py:31-38)
//       both hinges and their gradients w.r.t. proj and the two touched similarity columns (p_companion.py:95-119)
//       dpi, dtp_k, dce_k = dtp_k typ_w, dc, dh = (dc dec_w) relu' dropout', dt = dh enc_w    (the whole dX chain)
//     products on v_mfma_f32_16x16x4_f32 (exact fp32 fma chains: 16-row tiles fill 256 workgroups at B = 4096)
//   joint_wgrad_kernel  d itm_w, d typ_w, d dec_w, d enc_w (+ biases) and both [T,64] table gradients as one-hot
//                       products over the same 16-sample subtiles, blocks sized to each product -- the type hinge's
//                       dE_c rows ride in the table product (no float atomics anywhere: the step is bitwise
//                       reproducible for T <= 512); one slab per workgroup
//   joint_finish_kernel slab sums (fixed order) -> .grad, losses, Adam
// Large tables (T > 512, e.g. config.py:27 NUM_TYPES = 34800): the similarity row depends on the query TYPE only, so
// it is formed once per distinct query type of the batch (present-type list, sims + per-chunk top-K in one kernel's
// epilogue, merge) instead of per sample, the [B,T] matrix never exists, and the table gradients fall back to
// hardware float atomics into the cleared dense gradient (row lists of B*(K+2) + B rows).
#include "common.h"
#include "mfma16.h"

#define LH (PC_L / 2)
#define FK 4          /* top-K capacity of the fused kernel (NUM_COMP_TYPES = 3, config.py:24) */
#define TS 16         /* samples per tile */
#define T_SMALL 512   /* largest table the per-tile similarity row (LDS) and the one-hot gradients serve */
#define T_WGRAD 128   /* largest table whose one-hot gradient blocks (8 per wave and table) the tile kernel carries itself */

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct FusedArgs {
    const float *table, *enc_w, *enc_b, *dec_w, *dec_b, *typ_w, *typ_b, *itm_w, *itm_b, *eq, *ec;
    const int32_t *query_idx, *query_types, *pos_types, *neg_types;
    const float *pos_items, *neg_items;
    int B, T, K, P;
    float margin, g_type, g_item;        // g_type = (1 - alpha) / B, g_item = alpha / (B K): the means' constants
    DropCfg drop;
    // regime L (T > T_SMALL): top-K per query TYPE, computed beforehand -- or, with hidden-layer dropout (c then differs from
    // sample to sample), per SAMPLE: topk_per_sample != 0, the same table indexed by the sample
    const int32_t* topk_by_type; int topk_per_sample;
    // outputs
    int32_t* topk;                        // [B,K]
    float *part_type, *part_item;         // [B] hinge values (summed by the finish kernel)
    float *h, *dpi, *dtp, *dc, *dh, *dt;  // row buffers for the gradient products
    float* ecsrc; int32_t* ecidx;         // [B (K + 2)][64] / [B (K + 2)]: rows added into dE_c[ecidx[r]]
    int32_t* cids;                        // [2][B] validated (clamped) query_idx / query_types for the kernels that follow
    int32_t* bad; int64_t* step_count;
    int32_t* run_counts;                  // (large tables, sorted gradients) the six run-list counters table_sort_kernel appends through: zeroed here
    float* slabs; int slab_floats;        // WGRAD: one gradient slab per workgroup (layout: wg_off_*)
    // PAIRS: the batch is built here from labelled pairs (data_loader.py:133-157, see pc_build_complementary_batch);
    // query_idx .. neg_items above are then OUTPUTS (the batch as the loader would have handed it), written on the way
    const int32_t* pairs; const float* features; const int32_t* type_idx; int n_types_mod; uint64_t bseed, bstep;
    int32_t *o_qidx, *o_qt, *o_pt, *o_nt; float *o_pos, *o_neg;
};

// ---- wave-wide reductions on the DPP network (6 VALU instructions; the xor-shuffle form is 6 dependent LDS-crossbar
// round trips, ~10x the latency -- and this kernel is a chain of latencies).  Fixed order: bitwise reproducible.
template <int CTRL, int RM>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RM, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = dpp_add<0x111, 0xf>(v);     // row_shr:1  } inclusive scan inside each row of 16 lanes: lane 15 of a row holds
    v = dpp_add<0x112, 0xf>(v);     // row_shr:2  } the row total
    v = dpp_add<0x114, 0xf>(v);     // row_shr:4
    v = dpp_add<0x118, 0xf>(v);     // row_shr:8
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1, 3: lane 31 = rows 0+1, lane 63 = rows 2+3
    v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2, 3: lane 63 = everything
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// inclusive prefix sum over the wave's 64 lanes on the same network (every lane keeps its prefix)
template <int CTRL, int RM>
__device__ __forceinline__ int dpp_addi(int v) { return v + __builtin_amdgcn_update_dpp(0, v, CTRL, RM, 0xf, true); }
__device__ __forceinline__ int wave_scan_incl(int v) {
    v = dpp_addi<0x111, 0xf>(v); v = dpp_addi<0x112, 0xf>(v); v = dpp_addi<0x114, 0xf>(v); v = dpp_addi<0x118, 0xf>(v);
    v = dpp_addi<0x142, 0xa>(v); v = dpp_addi<0x143, 0xc>(v);
    return v;
}
// ---- reductions inside a ROW of 16 lanes (four samples per wave side by side): an inclusive scan on the row_shr network
// leaves the row total in lane 15 of the row in a FIXED order; every lane of the row then takes that one value (a
// rotate-and-add all-reduce would give each lane its own association of the 16 terms: the hinge decisions of a sample
// must not differ between its lanes).
template <int CTRL>
__device__ __forceinline__ float dpp_row_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_bcast15(float v, int g) {
    const int x = __builtin_bit_cast(int, v);
    const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 15)), t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 31));
    const float t2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 47)), t3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 63));
    return g == 0 ? t0 : g == 1 ? t1 : g == 2 ? t2 : t3;
}
__device__ __forceinline__ float row_sum(float v, int g) {
    v = dpp_row_add<0x111>(v); v = dpp_row_add<0x112>(v); v = dpp_row_add<0x114>(v); v = dpp_row_add<0x118>(v);
    return row_bcast15(v, g);
}
// max of a 64-bit key (hi, lo) over the row: max is idempotent, so the rotate network (row_ror 1, 2, 4, 8) hands every
// lane of the row the same result directly
template <int CTRL>
__device__ __forceinline__ void dpp_row_maxkey(unsigned& hi, unsigned& lo) {
    const unsigned oh = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    const unsigned ol = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    const bool gt = oh > hi || (oh == hi && ol > lo);
    hi = gt ? oh : hi;
    lo = gt ? ol : lo;
}
__device__ __forceinline__ void row_maxkey(unsigned& hi, unsigned& lo) {
    dpp_row_maxkey<0x121>(hi, lo); dpp_row_maxkey<0x122>(hi, lo); dpp_row_maxkey<0x124>(hi, lo); dpp_row_maxkey<0x128>(hi, lo);
}
// the same over the whole wave (row_shr scan inside the rows, row_bcast 15 / 31 across them: lane 63 holds the maximum)
template <int CTRL, int RM>
__device__ __forceinline__ void dpp_maxkey(unsigned& hi, unsigned& lo) {
    const unsigned oh = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, RM, 0xf, false);
    const unsigned ol = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, RM, 0xf, false);
    const bool gt = oh > hi || (oh == hi && ol > lo);
    hi = gt ? oh : hi;
    lo = gt ? ol : lo;
}
__device__ __forceinline__ void wave_maxkey(unsigned& hi, unsigned& lo) {
    dpp_maxkey<0x111, 0xf>(hi, lo); dpp_maxkey<0x112, 0xf>(hi, lo); dpp_maxkey<0x114, 0xf>(hi, lo);
    dpp_maxkey<0x118, 0xf>(hi, lo); dpp_maxkey<0x142, 0xa>(hi, lo); dpp_maxkey<0x143, 0xc>(hi, lo);
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
}
// order-preserving map fp32 -> uint32 (larger float <=> larger unsigned); -0.0 < +0.0 here, torch.topk treats them as
// equal: a similarity of exactly -0.0 against +0.0 is the only case the tie rule could differ in
__device__ __forceinline__ unsigned ord_f32(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// top-K of one LDS row of T similarities by the 16 lanes of a row group (descending; ties -> the lower index, like
// torch.topk / pc_topk_rows): K rounds; in a round every lane scans its strided elements, skipping the earlier
// winners, and the row takes the maximum key (value, ~index).  out[r] is the same in all 16 lanes.
template <int KC>
__device__ __forceinline__ void row_topk(const float* row, int T, int K, int l16, int (&out)[FK]) {
#pragma unroll
    for (int r = 0; r < FK; r++) out[r] = -1;
#pragma unroll
    for (int r = 0; r < FK; r++) {
        if (r < (KC ? KC : K)) {
            unsigned bh = 0u, bl = 0u;                     // (0, 0) is below every real key
            for (int t = l16; t < T; t += 16) {
                const unsigned kh = ord_f32(row[t]), kl = ~(unsigned)t;
                const bool taken = t == out[0] || t == out[1] || t == out[2] || t == out[3];
                const bool gt = !taken && (kh > bh || (kh == bh && kl > bl));
                bh = gt ? kh : bh;
                bl = gt ? kl : bl;
            }
            row_maxkey(bh, bl);
            out[r] = (int)~bl;
        }
    }
}

// top-K of one LDS row of n values by the 16 lanes of a row group, ONE pass over the row: every lane keeps the best FK keys
// (value, ~index) of its strided elements in a sorted register list, then K rounds of a row maximum whose owner retires
// its head.  Same order as row_topk (descending, ties -> the lower index); (0, 0) = none (fewer than K values).
__device__ __forceinline__ void row_topk_ins(const float* row, int n, int K, int l16, unsigned (&oh)[FK], unsigned (&ol)[FK]) {
    unsigned kh[FK], kl[FK];
#pragma unroll
    for (int j = 0; j < FK; j++) { kh[j] = 0u; kl[j] = 0u; oh[j] = 0u; ol[j] = 0u; }
    for (int t = l16; t < n; t += 16) {
        unsigned h = ord_f32(row[t]), l = ~(unsigned)t;
#pragma unroll
        for (int j = 0; j < FK; j++) {
            const bool gt = h > kh[j] || (h == kh[j] && l > kl[j]);
            const unsigned th = gt ? kh[j] : h, tl = gt ? kl[j] : l;
            kh[j] = gt ? h : kh[j]; kl[j] = gt ? l : kl[j];
            h = th; l = tl;
        }
    }
#pragma unroll
    for (int r = 0; r < FK; r++) {
        if (r < K) {
            unsigned bh = kh[0], bl = kl[0];
            row_maxkey(bh, bl);
            if (kh[0] == bh && kl[0] == bl) {                      // (keys are distinct: exactly one lane owns the winner)
#pragma unroll
                for (int j = 0; j < FK - 1; j++) { kh[j] = kh[j + 1]; kl[j] = kl[j + 1]; }
                kh[FK - 1] = 0u; kl[FK - 1] = 0u;
            }
            oh[r] = bh; ol[r] = bl;
        }
    }
}

// The same selection at a fifth of the instructions (a first per-sample similarity kernel spent twice the matrix
// pipe's time in row_topk_ins: ~52 VALU instructions per element): ONE 32-bit key per element -- the order-preserving image
// of the value with its low 9 bits replaced by 511 - index -- so a lane keeps its best FOUR keys with v_max_u32 + three
// v_med3_u32 per element and the row's best four fall out of four single-register row maxima.  Truncated keys order
// elements exactly unless two of the first K + 1 agree in their upper 23 bits (values within 512 ulp of each other, or true
// ties): then -- and for K > 3 -- the caller falls back to row_topk_ins.  Proof of exactness otherwise: an element outside
// the four has a truncated value <= the fourth's < the third's, hence a true value below the third's.  n <= 512.
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// max over a row of 16 lanes, every lane gets it: four v_max_u32 with the rotated operand read through DPP (the builtin form
// compiles to a v_mov_dpp and a separate maximum each).  The wait states a DPP read of a VGPR the previous VALU instruction
// wrote needs are spelled out: the hazard recogniser does not look inside inline asm.
__device__ __forceinline__ unsigned row16_umax(unsigned v) {
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ bool row_topk_trunc(const float* row, int n, int K, int l16, unsigned (&oh)[FK], unsigned (&ol)[FK]) {
    unsigned a = 0u, b = 0u, c = 0u, d = 0u;
    for (int t = l16; t < n; t += 64) {                        // four reads in flight per round; a key of 0 changes nothing
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = t + 16 * j < n ? row[t + 16 * j] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned k = t + 16 * j < n ? ((ord_f32(x[j]) & ~511u) | (511u - (unsigned)(t + 16 * j))) : 0u;
            d = umed3(c, d, k); c = umed3(b, c, k); b = umed3(a, b, k); a = a > k ? a : k;
        }
    }
    unsigned top[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        top[r] = row16_umax(a);                                // row maximum on the rotate network: every lane gets it
        const bool own = a == top[r];                          // (keys carry their index: one owner retires its head)
        a = own ? b : a; b = own ? c : b; c = own ? d : c; d = own ? 0u : d;
    }
    bool exact = K <= 3;
#pragma unroll
    for (int r = 0; r < 3; r++)
        if (r < K && top[r + 1] != 0u && (top[r] >> 9) == (top[r + 1] >> 9)) exact = false;
#pragma unroll
    for (int r = 0; r < FK; r++) {
        oh[r] = 0u; ol[r] = 0u;
        if (r < 3 && r < K && top[r] != 0u) {
            const unsigned idx = 511u - (top[r] & 511u);
            oh[r] = ord_f32(row[idx]);                          // the exact value back from LDS
            ol[r] = ~idx;
        }
    }
    return exact;
}

// per-workgroup gradient slab: itm_w | itm_b | typ_w | typ_b | dec_w | dec_b | enc_w | enc_b | E_c | E_q
__host__ __device__ inline int wg_off_itm_w() { return 0; }
__host__ __device__ inline int wg_off_itm_b() { return PC_D * PC_D; }
__host__ __device__ inline int wg_off_typ_w() { return wg_off_itm_b() + PC_D; }
__host__ __device__ inline int wg_off_typ_b() { return wg_off_typ_w() + PC_D * PC_L; }
__host__ __device__ inline int wg_off_dec_w() { return wg_off_typ_b() + PC_D; }
__host__ __device__ inline int wg_off_dec_b() { return wg_off_dec_w() + PC_L * LH; }
__host__ __device__ inline int wg_off_enc_w() { return wg_off_dec_b() + PC_L; }
__host__ __device__ inline int wg_off_enc_b() { return wg_off_enc_w() + LH * PC_L; }
__host__ __device__ inline int wg_off_ec() { return wg_off_enc_b() + LH; }
__host__ __device__ inline int wg_off_eq(int T) { return wg_off_ec() + T * PC_L; }
__host__ __device__ inline int wg_slab_floats(int T) { return wg_off_eq(T) + T * PC_L; }

#define LD64 68       /* LDS row strides: row length + 4 floats (16-B reads of 16 rows spread over the banks) */
#define LD32 36
#define LD128 132

// LDS-only hand-off between the phases (no wait for this wave's outstanding global loads / stores)
__device__ __forceinline__ void phase_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

#ifdef PC_JOINT_TIMING
// developer build (scripts/joint_phase_times.py): 100 MHz wall-clock stamps of wave 0 of workgroups 0 and 128 at the
// phase boundaries of the last launch
__device__ unsigned long long pc_joint_timing[2 * 16];
extern "C" int pc_debug_joint_timing(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pc_joint_timing), sizeof(unsigned long long) * 32);
}
#define PC_STAMP(i)                                                                                      \
    do {                                                                                                 \
        if ((blockIdx.x == 0 || blockIdx.x == 128) && tid == 0) pc_joint_timing[(blockIdx.x ? 16 : 0) + (i)] = wall_clock64(); \
    } while (0)
#else
#define PC_STAMP(i) do { } while (0)
#endif

// SIMS_LOCAL: the similarity row and its top-K are computed here (T <= T_SMALL); else read per query type
// KC: compile-time NUM_COMP_TYPES (3 = config.py:24; the row-block loops then carry no branches and the compiler
// schedules the LDS reads of a whole phase ahead of its MFMAs), 0 = run-time K <= FK
// WGRAD: the tile's share of every weight gradient (and, with SIMS_LOCAL and T <= 128, of both table gradients as
// one-hot products) is formed HERE from the operands the phases left in LDS and written as the workgroup's slab: the
// row buffers never travel to HBM and back and the step is one launch shorter.  Samples of a product run in the order
// s = q + 4 h (lane group h of MFMA q): with row strides = 4 mod 64 banks the four rows of one MFMA sit 16 banks apart.
template <bool SIMS_LOCAL, int KC, bool WGRAD, bool PAIRS = false>
__global__ __launch_bounds__(256) void joint_tile_kernel(FusedArgs a, int ldsims) {
    constexpr bool TABLES = WGRAD && SIMS_LOCAL;       // table gradients in this kernel
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Tin = sm;                       // [16][LD64]   E_q rows
    float* Hs = Tin + TS * LD64;           // [16][LD32]   hidden (dropped)
    float* Cs = Hs + TS * LD32;            // [16][LD64]   complementary base
    float* Qs = Cs + TS * LD64;            // [16][LD128]  product rows
    float* PIs = Qs + TS * LD128;          // [16][LD128]  item projection
    float* ECs = PIs + TS * LD128;         // [16 FK][LD64] selected E_c rows
    float* TPs = ECs + TS * FK * LD64;     // [16 FK][LD128] type projection -> d(tp)
    float* DCs = TPs + TS * FK * LD128;    // [16][LD64]
    float* DHs = DCs + TS * LD64;          // [16][LD32]
    int* ints = reinterpret_cast<int*>(DHs + TS * LD32);      // [16][8]: qi, qt, pos, neg, topk[4]
    float* Sims = reinterpret_cast<float*>(ints + TS * 8);    // [16][ldsims]   (SIMS_LOCAL only)
    float* ES = Sims + (SIMS_LOCAL ? TS * ldsims : 0);        // TABLES: [16 (K + 2)][LD64] dE_c source rows (row block k: the
                                                              // selected types' rows, K: -> pos type, K + 1: -> neg type)
    float* DTs = ES + TS * (FK + 2) * LD64;                   // TABLES: [16][LD64] dE_q source rows (then the destination ids, 16 (FK + 3) ints)

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * TS;
    const int K = KC ? KC : a.K;
    constexpr int MBK = KC ? KC : FK;          // row blocks of the K-row products
    PC_STAMP(0);
    if (blockIdx.x == 0 && tid == 0 && a.step_count) *a.step_count += 1;     // Adam's step (read by the finish kernel)
    if (blockIdx.x == 0 && tid < 6 && a.run_counts) a.run_counts[tid] = 0;

    // ---- every weight fragment this wave will multiply by, requested now (see load_b)
    BFrag<4> f_h = {}, f_dh = {}, f_s0 = {}, f_s1 = {};
    BFrag<8> f_pa = {}, f_pb = {};
    if (w < 2) { f_h = load_b<PC_L, false>(a.enc_w, PC_L, 16 * w, LH, lane); f_dh = load_b<PC_L, true>(a.dec_w, LH, 16 * w, LH, lane); }
    else { f_pa = load_b<PC_D, false>(a.itm_w, PC_D, 32 * (w - 2), PC_D, lane); f_pb = load_b<PC_D, false>(a.itm_w, PC_D, 32 * (w - 2) + 16, PC_D, lane); }
    const BFrag<8> f_pc = load_b<PC_D, false>(a.itm_w, PC_D, 16 * (4 + w), PC_D, lane);
    const BFrag<2> f_c = load_b<LH, false>(a.dec_w, LH, 16 * w, PC_L, lane);
    const BFrag<4> f_t0 = load_b<PC_L, false>(a.typ_w, PC_L, 16 * w, PC_D, lane);
    const BFrag<4> f_t1 = load_b<PC_L, false>(a.typ_w, PC_L, 16 * (w + 4), PC_D, lane);
    const BFrag<8> f_dce = load_b<PC_D, true>(a.typ_w, PC_L, 16 * w, PC_L, lane);
    const BFrag<2> f_dt = load_b<LH, true>(a.enc_w, PC_L, 16 * w, PC_L, lane);
    if (SIMS_LOCAL) {                       // the first two of this wave's E_c column blocks (all of them for T <= 128)
        f_s0 = load_b<PC_L, false>(a.ec, PC_L, 16 * w, a.T, lane);
        f_s1 = load_b<PC_L, false>(a.ec, PC_L, 16 * (w + 4), a.T, lane);
    }
    const int ci = lane & 15, rh = lane >> 4;       // result column inside a block / row group
    const float bias_h = w < 2 ? a.enc_b[16 * w + ci] : 0.f, bias_c = a.dec_b[16 * w + ci];
    const float bias_pa = w >= 2 ? a.itm_b[32 * (w - 2) + ci] : 0.f, bias_pb = w >= 2 ? a.itm_b[32 * (w - 2) + 16 + ci] : 0.f;
    const float bias_pc = a.itm_b[16 * (4 + w) + ci], bias_t0 = a.typ_b[16 * w + ci], bias_t1 = a.typ_b[16 * (w + 4) + ci];
    // (the ids come AFTER the weight requests in program order: their two or three dependent round trips -- pairs -> type ids
    // -> LDS -- then run beside the ~60 fragment loads of the same wave instead of ahead of them)
    // ---- indices of the tile: validated (the reference raises for an id outside its table, p_companion.py:48-54; here
    // the offence is counted and the id clamped so that nothing is read or written out of bounds)
    if (tid < TS) {
        const int b = b0 + tid;
        int qi = 0, qt = 0, pt = 0, nt = 0, wrong = 0, tg = 0, lab = 1;
        if (b < a.B) {
            if (PAIRS) {
                qi = a.pairs[3 * b]; tg = a.pairs[3 * b + 1]; lab = a.pairs[3 * b + 2];
                if ((unsigned)qi >= (unsigned)a.P) { wrong++; qi = 0; }
                if ((unsigned)tg >= (unsigned)a.P) { wrong++; tg = 0; }
                const int tt = a.type_idx[tg];
                qt = a.type_idx[qi];
                pt = lab == 1 ? tt : 0;
                nt = lab == 1 ? (tt + 1) % a.n_types_mod : tt;
                a.o_qidx[b] = qi; a.o_qt[b] = qt; a.o_pt[b] = pt; a.o_nt[b] = nt;
            } else {
                qi = a.query_idx[b]; qt = a.query_types[b]; pt = a.pos_types[b]; nt = a.neg_types[b];
            }
            if ((unsigned)qi >= (unsigned)a.P) { wrong++; qi = 0; }
            if ((unsigned)qt >= (unsigned)a.T) { wrong++; qt = 0; }
            if ((unsigned)pt >= (unsigned)a.T) { wrong++; pt = 0; }
            if ((unsigned)nt >= (unsigned)a.T) { wrong++; nt = 0; }
            if (wrong && a.bad) atomicAdd(a.bad, wrong);
        }
        ints[tid * 8 + 0] = qi; ints[tid * 8 + 1] = qt; ints[tid * 8 + 2] = pt; ints[tid * 8 + 3] = nt;
        if (!TABLES && b < a.B) { a.cids[b] = qi; a.cids[a.B + b] = qt; }    // (gradient-product / scatter kernels gather by these)
        if (PAIRS) { ints[tid * 8 + 4] = tg; ints[tid * 8 + 5] = lab; }      // (the top-K slots: free until phase D)
    }
    phase_sync();
    PC_STAMP(1);
    // ---- row gathers: t = E_q[qt] (16 x 64), q = E_prod[qi] (16 x 128); rows past the batch are zero.  Also the rows of
    // the loss phase: there a row group of 16 lanes (g = lane >> 4) owns sample 4 w + g, lane l16 its item dims
    // [4 l16, 4 l16 + 4) and [64 + 4 l16, ...) and dims [4 l16, 4 l16 + 4) of the 64-wide type rows
    const int g4 = lane >> 4, l16 = lane & 15;
    const int sF = 4 * w + g4, bF = b0 + sF;
    const bool liveF = bF < a.B;
    float4 r_pos[2], r_neg[2], r_ep, r_en;
    {
        const int r = tid >> 4, c4 = (tid & 15) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b0 + r < a.B) v = *reinterpret_cast<const float4*>(a.eq + (size_t)ints[r * 8 + 1] * PC_L + c4);
        float4 x[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = tid + 256 * u, rr = e >> 5, cc = (e & 31) * 4;
            x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b0 + rr < a.B) x[u] = *reinterpret_cast<const float4*>(a.table + (size_t)ints[rr * 8 + 0] * PC_D + cc);
        }
        {
            const size_t bb = liveF ? bF : 0;
            if (PAIRS) {
                // positive / negative item rows: the target's feature row and an N(0,1) filler, by the pair's label
                const int tg = ints[sF * 8 + 4];
                const bool pos = ints[sF * 8 + 5] == 1;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const float4 f = *reinterpret_cast<const float4*>(a.features + (size_t)tg * PC_D + 64 * u + 4 * l16);
                    const float4 fill = pc_filler_chunk(a.bseed, a.bstep, (uint32_t)(bb * (PC_D / 4) + 16 * u + l16));
                    // (component-wise selects: `pos ? f : fill` on the structs becomes an indexed stack array)
                    r_pos[u] = make_float4(pos ? f.x : fill.x, pos ? f.y : fill.y, pos ? f.z : fill.z, pos ? f.w : fill.w);
                    r_neg[u] = make_float4(pos ? fill.x : f.x, pos ? fill.y : f.y, pos ? fill.z : f.z, pos ? fill.w : f.w);
                    if (liveF) {
                        *reinterpret_cast<float4*>(a.o_pos + bb * PC_D + 64 * u + 4 * l16) = r_pos[u];
                        *reinterpret_cast<float4*>(a.o_neg + bb * PC_D + 64 * u + 4 * l16) = r_neg[u];
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    r_pos[u] = *reinterpret_cast<const float4*>(a.pos_items + bb * PC_D + 64 * u + 4 * l16);
                    r_neg[u] = *reinterpret_cast<const float4*>(a.neg_items + bb * PC_D + 64 * u + 4 * l16);
                }
            }
            r_ep = *reinterpret_cast<const float4*>(a.ec + (size_t)ints[sF * 8 + 2] * PC_L + 4 * l16);
            r_en = *reinterpret_cast<const float4*>(a.ec + (size_t)ints[sF * 8 + 3] * PC_L + 4 * l16);
        }
        *reinterpret_cast<float4*>(&Tin[r * LD64 + c4]) = v;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = tid + 256 * u, rr = e >> 5, cc = (e & 31) * 4;
            *reinterpret_cast<float4*>(&Qs[rr * LD128 + cc]) = x[u];
        }
    }
    phase_sync();
    PC_STAMP(2);

    // ---- phase A: h = dropout(relu(enc t + b))  (waves 0, 1: one 16-column block each)   ||   pi blocks 0..3 (waves 2, 3)
    if (w < 2) {
        f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}};
        mul_b<4, 1>(Tin, LD64, 1, f_h, acc, lane);
        const int col = 16 * w + ci;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * rh + r;
            float x = acc[0][r] + bias_h;
            x = x > 0.f ? x : 0.f;
            if (a.drop.thr) {
                float m[4];
                pc_dropout_keep4(a.drop, (unsigned)((b0 + row) * (LH / 4) + (col >> 2)), PC_DROP_STREAM_HIDDEN, m);
                x *= m[col & 3];
            }
            Hs[row * LD32 + col] = x;
            if (!WGRAD && b0 + row < a.B) a.h[(size_t)(b0 + row) * LH + col] = x;
        }
    } else {
        f32x4v acc0[1] = {{0.f, 0.f, 0.f, 0.f}}, acc1[1] = {{0.f, 0.f, 0.f, 0.f}};
        mul_b<8, 1>(Qs, LD128, 1, f_pa, acc0, lane);
        mul_b<8, 1>(Qs, LD128, 1, f_pb, acc1, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            PIs[(4 * rh + r) * LD128 + 32 * (w - 2) + ci] = acc0[0][r] + bias_pa;
            PIs[(4 * rh + r) * LD128 + 32 * (w - 2) + 16 + ci] = acc1[0][r] + bias_pb;
        }
    }
    phase_sync();
    PC_STAMP(3);
    // ---- phase B: c = dec h + b (4 blocks, one per wave), then pi blocks 4..7
    {
        f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}}, accp[1] = {{0.f, 0.f, 0.f, 0.f}};
        mul_b<2, 1>(Hs, LD32, 1, f_c, acc, lane);
        mul_b<8, 1>(Qs, LD128, 1, f_pc, accp, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            Cs[(4 * rh + r) * LD64 + 16 * w + ci] = acc[0][r] + bias_c;
            PIs[(4 * rh + r) * LD128 + 16 * (4 + w) + ci] = accp[0][r] + bias_pc;
        }
    }
    phase_sync();
    PC_STAMP(4);
    // ---- phase C / D: similarities over all T types and their top-K -- or the per-type result of the dedup pass
    if (SIMS_LOCAL) {
        const int nblk = (a.T + 15) >> 4;
        {
            f32x4v acc0[1] = {{0.f, 0.f, 0.f, 0.f}}, acc1[1] = {{0.f, 0.f, 0.f, 0.f}};
            if (w < nblk) mul_b<4, 1>(Cs, LD64, 1, f_s0, acc0, lane);
            if (w + 4 < nblk) mul_b<4, 1>(Cs, LD64, 1, f_s1, acc1, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if (w < nblk) Sims[(4 * rh + r) * ldsims + 16 * w + ci] = acc0[0][r];
                if (w + 4 < nblk) Sims[(4 * rh + r) * ldsims + 16 * (w + 4) + ci] = acc1[0][r];
            }
        }
        for (int nb = w + 8; nb < nblk; nb += 4) {
            f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            block_product<PC_L, 1, false>(Cs, LD64, 1, a.ec, PC_L, 16 * nb, a.T, acc, lane);
#pragma unroll
            for (int r = 0; r < 4; r++) Sims[(4 * rh + r) * ldsims + 16 * nb + ci] = acc[0][r];
        }
        phase_sync();
        PC_STAMP(5);
        {
            // (one-word truncated keys; exact two-word fallback on near-ties and for K = 4: see row_topk_trunc)
            int idx[FK];
            {
                unsigned kh[FK], kl[FK];
                const bool exact = row_topk_trunc(Sims + sF * ldsims, a.T, K, l16, kh, kl);
                if (__ballot(!exact)) row_topk_ins(Sims + sF * ldsims, a.T, K, l16, kh, kl);
#pragma unroll
                for (int r = 0; r < FK; r++) idx[r] = (kh[r] | kl[r]) != 0u ? (int)~kl[r] : -1;
            }
            if (l16 == 0)
#pragma unroll
                for (int k = 0; k < FK; k++)
                    if (k < K) ints[sF * 8 + 4 + k] = idx[k];
        }
    } else {
        if (tid < TS * FK) {
            const int s = tid / FK, k = tid % FK;
            if (k < K) {
                // (a query type that was out of range and clamped may have no entry in the per-type table: whatever is read
                // there is forced into the table before it is used as a row id)
                const int t = b0 + s < a.B ? a.topk_by_type[(size_t)(a.topk_per_sample ? b0 + s : ints[s * 8 + 1]) * K + k] : 0;
                ints[s * 8 + 4 + k] = (unsigned)t < (unsigned)a.T ? t : 0;
            }
        }
    }
    phase_sync();
    PC_STAMP(6);
    // selected rows e_k = E_c[top_k]: 16 K rows of 64 floats; LDS row = k * 16 + s keeps one MFMA row block per k
    for (int e = tid; e < TS * K * 16; e += 256) {
        const int row = e >> 4, c4 = (e & 15) * 4;
        const int s = row & 15, k = row >> 4;
        const int t = ints[s * 8 + 4 + k];
        *reinterpret_cast<float4*>(&ECs[row * LD64 + c4]) = *reinterpret_cast<const float4*>(a.ec + (size_t)t * PC_L + c4);
        if (c4 == 0 && b0 + s < a.B) {
            a.topk[(size_t)(b0 + s) * K + k] = t;
            if (!TABLES) a.ecidx[(size_t)(b0 + s) * K + k] = t;
        }
    }
    phase_sync();
    PC_STAMP(7);
    // ---- phase E: tp_k = typ e_k + b: K row blocks x 8 column blocks; a wave takes column blocks w and w + 4
    {
        f32x4v acc0[MBK], acc1[MBK];
#pragma unroll
        for (int m = 0; m < MBK; m++) { acc0[m] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc1[m] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
        mul_b<4, MBK>(ECs, LD64, K, f_t0, acc0, lane);
        mul_b<4, MBK>(ECs, LD64, K, f_t1, acc1, lane);
#pragma unroll
        for (int m = 0; m < MBK; m++)
            if (m < K)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    TPs[(16 * m + 4 * rh + r) * LD128 + 16 * w + ci] = acc0[m][r] + bias_t0;
                    TPs[(16 * m + 4 * rh + r) * LD128 + 16 * (w + 4) + ci] = acc1[m][r] + bias_t1;
                }
    }
    phase_sync();
    PC_STAMP(8);
    // ---- phase F: per sample (a row group of 16 lanes each, four samples per wave side by side): proj_k = pi * tp_k, both
    // hinges, d(proj) -> d(pi), d(tp_k) (in place), the two-column type hinge -> dc and the two dE_c rows it touches
    {
        const int pt = ints[sF * 8 + 2], nt = ints[sF * 8 + 3];
        const float4 cb = *reinterpret_cast<const float4*>(&Cs[sF * LD64 + 4 * l16]);
        float sp, sn;
        if (SIMS_LOCAL) { sp = Sims[sF * ldsims + pt]; sn = Sims[sF * ldsims + nt]; }
        else {
            sp = row_sum(cb.x * r_ep.x + cb.y * r_ep.y + cb.z * r_ep.z + cb.w * r_ep.w, g4);
            sn = row_sum(cb.x * r_en.x + cb.y * r_en.y + cb.z * r_en.z + cb.w * r_en.w, g4);
        }
        const float lt = a.margin - sp + sn;
        const float gt = (liveF && lt > 0.f) ? a.g_type : 0.f;
        const float4 dcv = make_float4(gt * (r_en.x - r_ep.x), gt * (r_en.y - r_ep.y), gt * (r_en.z - r_ep.z), gt * (r_en.w - r_ep.w));
        *reinterpret_cast<float4*>(&DCs[sF * LD64 + 4 * l16]) = dcv;
        float4 pa[2];
#pragma unroll
        for (int u = 0; u < 2; u++) pa[u] = *reinterpret_cast<const float4*>(&PIs[sF * LD128 + 64 * u + 4 * l16]);
        float4 t[MBK][2], dp[MBK][2], dn[MBK][2];
        float np_[MBK], nn_[MBK];
#pragma unroll
        for (int k = 0; k < MBK; k++) {
            np_[k] = nn_[k] = 0.f;
            if (k < K) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    t[k][u] = *reinterpret_cast<const float4*>(&TPs[(16 * k + sF) * LD128 + 64 * u + 4 * l16]);
                    const float4 x = make_float4(pa[u].x * t[k][u].x, pa[u].y * t[k][u].y, pa[u].z * t[k][u].z, pa[u].w * t[k][u].w);
                    dp[k][u] = make_float4(x.x - r_pos[u].x, x.y - r_pos[u].y, x.z - r_pos[u].z, x.w - r_pos[u].w);
                    dn[k][u] = make_float4(x.x - r_neg[u].x, x.y - r_neg[u].y, x.z - r_neg[u].z, x.w - r_neg[u].w);
                    np_[k] += dp[k][u].x * dp[k][u].x + dp[k][u].y * dp[k][u].y + dp[k][u].z * dp[k][u].z + dp[k][u].w * dp[k][u].w;
                    nn_[k] += dn[k][u].x * dn[k][u].x + dn[k][u].y * dn[k][u].y + dn[k][u].z * dn[k][u].z + dn[k][u].w * dn[k][u].w;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < MBK; k++)
            if (k < K) { np_[k] = sqrtf(row_sum(np_[k], g4)); nn_[k] = sqrtf(row_sum(nn_[k], g4)); }
        float li = 0.f;
        float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
        for (int k = 0; k < MBK; k++)
            if (k < K) {
                const float l = a.margin - np_[k] + nn_[k];
                li += l > 0.f ? l : 0.f;
                const float g = (liveF && l > 0.f) ? a.g_item : 0.f;
                const float ip = np_[k] > 0.f ? g / np_[k] : 0.f, in = nn_[k] > 0.f ? g / nn_[k] : 0.f;   // torch.norm: subgradient 0 at 0
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const float4 d = make_float4(-dp[k][u].x * ip + dn[k][u].x * in, -dp[k][u].y * ip + dn[k][u].y * in,
                                                 -dp[k][u].z * ip + dn[k][u].z * in, -dp[k][u].w * ip + dn[k][u].w * in);
                    acc[u].x += d.x * t[k][u].x; acc[u].y += d.y * t[k][u].y; acc[u].z += d.z * t[k][u].z; acc[u].w += d.w * t[k][u].w;
                    const float4 dt4 = make_float4(d.x * pa[u].x, d.y * pa[u].y, d.z * pa[u].z, d.w * pa[u].w);
                    // (a dead row of the last tile carries g = 0: zero operands for the products downstream)
                    *reinterpret_cast<float4*>(&TPs[(16 * k + sF) * LD128 + 64 * u + 4 * l16]) = dt4;
                    if (!WGRAD && liveF) *reinterpret_cast<float4*>(a.dtp + ((size_t)bF * K + k) * PC_D + 64 * u + 4 * l16) = dt4;
                }
            }
        const float4 hp = make_float4(-gt * cb.x, -gt * cb.y, -gt * cb.z, -gt * cb.w);          // -> dE_c[pos]
        const float4 hn = make_float4(gt * cb.x, gt * cb.y, gt * cb.z, gt * cb.w);              // -> dE_c[neg]
        if (WGRAD) {                                   // d(pi) over pi (this lane's own elements); dead rows carry zeros
#pragma unroll
            for (int u = 0; u < 2; u++) *reinterpret_cast<float4*>(&PIs[sF * LD128 + 64 * u + 4 * l16]) = acc[u];
        }
        if (TABLES) {
            *reinterpret_cast<float4*>(&ES[(16 * K + sF) * LD64 + 4 * l16]) = hp;
            *reinterpret_cast<float4*>(&ES[(16 * (K + 1) + sF) * LD64 + 4 * l16]) = hn;
        }
        if (liveF) {
            if (!WGRAD) {
                *reinterpret_cast<float4*>(a.dc + (size_t)bF * PC_L + 4 * l16) = dcv;
#pragma unroll
                for (int u = 0; u < 2; u++) *reinterpret_cast<float4*>(a.dpi + (size_t)bF * PC_D + 64 * u + 4 * l16) = acc[u];
            }
            if (!TABLES) {
                *reinterpret_cast<float4*>(a.ecsrc + ((size_t)a.B * K + bF) * PC_L + 4 * l16) = hp;
                *reinterpret_cast<float4*>(a.ecsrc + ((size_t)a.B * (K + 1) + bF) * PC_L + 4 * l16) = hn;
            }
            if (l16 == 0) {
                a.part_type[bF] = lt > 0.f ? lt : 0.f;
                a.part_item[bF] = li;
                if (!TABLES) {
                    a.ecidx[(size_t)a.B * K + bF] = pt;
                    a.ecidx[(size_t)a.B * (K + 1) + bF] = nt;
                }
            }
        }
    }
    phase_sync();
    PC_STAMP(9);
    // ---- phase G: dce_k = dtp_k typ_w (K row blocks x 4 column blocks: wave w takes column block w) -> the dE_c rows of
    // the selected types; dh = (dc dec_w) relu' dropout' (waves 0, 1)
    {
        f32x4v acc[MBK];
#pragma unroll
        for (int m = 0; m < MBK; m++) acc[m] = f32x4v{0.f, 0.f, 0.f, 0.f};
        mul_b<8, MBK>(TPs, LD128, K, f_dce, acc, lane);
        const int col = 16 * w + ci;
#pragma unroll
        for (int m = 0; m < MBK; m++)
            if (m < K)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int s = 4 * rh + r;
                    if (TABLES) ES[(16 * m + s) * LD64 + col] = acc[m][r];
                    else if (b0 + s < a.B) a.ecsrc[((size_t)(b0 + s) * K + m) * PC_L + col] = acc[m][r];
                }
    }
    if (w < 2) {
        f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}};
        mul_b<4, 1>(DCs, LD64, 1, f_dh, acc, lane);
        const int col = 16 * w + ci;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * rh + r;
            // the saved hidden value is the DROPPED one: 0 where relu' = 0 or the unit was dropped; kept units carry 1/(1-p)
            float x = Hs[row * LD32 + col] > 0.f ? acc[0][r] : 0.f;
            if (a.drop.thr) x *= a.drop.scale;
            DHs[row * LD32 + col] = x;
            if (!WGRAD && b0 + row < a.B) a.dh[(size_t)(b0 + row) * LH + col] = x;
        }
    }
    phase_sync();
    PC_STAMP(10);
    // ---- phase H: dt = dh enc_w (4 column blocks, one per wave) -> the dE_q rows
    {
        f32x4v acc[1] = {{0.f, 0.f, 0.f, 0.f}};
        mul_b<2, 1>(DHs, LD32, 1, f_dt, acc, lane);
        const int col = 16 * w + ci;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * rh + r;
            if (TABLES) DTs[row * LD64 + col] = acc[0][r];
            else if (b0 + row < a.B) a.dt[(size_t)(b0 + row) * PC_L + col] = acc[0][r];
        }
    }
    PC_STAMP(11);
    if (WGRAD) {
        // ---- the tile's gradient products C[i][o] = sum_s X[s][i] Z[s][o] on 16 x 16 x 4 blocks: the lane's four results
        // are four consecutive i of one o -> one 16-B store at slab[o * Ni + i] (summed over the workgroups in fixed order by
        // the finish kernel).  Every operand is in LDS; rows of samples past the batch carry zero Z.
        phase_sync();
        float* slab = a.slabs + (size_t)blockIdx.x * a.slab_floats;
        const int i16 = lane & 15, h4 = lane >> 4;
        auto st4 = [&](float* dst, const f32x4v& v) { *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]); };
        {   // biases: column sums of d(pi), d(tp), dc, dh
            float bs = 0.f;
            if (tid < 128) {
#pragma unroll
                for (int r = 0; r < TS; r++) bs += PIs[r * LD128 + tid];
                slab[wg_off_itm_b() + tid] = bs;
            } else {
                for (int r = 0; r < TS * K; r++) bs += TPs[r * LD128 + tid - 128];
                slab[wg_off_typ_b() + tid - 128] = bs;
            }
            bs = 0.f;
            if (tid < 64) {
#pragma unroll
                for (int r = 0; r < TS; r++) bs += DCs[r * LD64 + tid];
                slab[wg_off_dec_b() + tid] = bs;
            } else if (tid < 96) {
#pragma unroll
                for (int r = 0; r < TS; r++) bs += DHs[r * LD32 + tid - 64];
                slab[wg_off_enc_b() + tid - 64] = bs;
            }
        }
        PC_STAMP(12);
        // d itm_w[o][i] = sum_s d(pi)[s][o] q[s][i]: wave w owns input blocks w and w + 4 x all 8 output blocks
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            const int ib = w + 4 * pass;
            f32x4v c[8];
#pragma unroll
            for (int ob = 0; ob < 8; ob++) c[ob] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int sr = q + 4 * h4;
                const float av = Qs[sr * LD128 + 16 * ib + i16];
#pragma unroll
                for (int ob = 0; ob < 8; ob++) c[ob] = mfma16(av, PIs[sr * LD128 + 16 * ob + i16], c[ob]);
            }
#pragma unroll
            for (int ob = 0; ob < 8; ob++) st4(slab + wg_off_itm_w() + (size_t)(16 * ob + i16) * PC_D + 16 * ib + 4 * h4, c[ob]);
        }
        PC_STAMP(13);
        // d typ_w[o][i] = sum_{s,k} d(tp)[s,k][o] e[s,k][i]: input block w (of 4) x 8 output blocks, 16 K rows
        {
            f32x4v c[8];
#pragma unroll
            for (int ob = 0; ob < 8; ob++) c[ob] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < MBK; k++) {
                if (k < K) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int row = 16 * k + q + 4 * h4;
                        const float av = ECs[row * LD64 + 16 * w + i16];
#pragma unroll
                        for (int ob = 0; ob < 8; ob++) c[ob] = mfma16(av, TPs[row * LD128 + 16 * ob + i16], c[ob]);
                    }
                }
            }
#pragma unroll
            for (int ob = 0; ob < 8; ob++) st4(slab + wg_off_typ_w() + (size_t)(16 * ob + i16) * PC_L + 16 * w + 4 * h4, c[ob]);
        }
        // d dec_w[o][i] = sum_s dc[s][o] h[s][i] ([64][32]: output block w x 2 input blocks);
        // d enc_w[o][i] = sum_s dh[s][o] t[s][i] ([32][64]: input block w x 2 output blocks)
        {
            f32x4v cd[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ce[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int sr = q + 4 * h4;
                const float zd = DCs[sr * LD64 + 16 * w + i16], xt = Tin[sr * LD64 + 16 * w + i16];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    cd[j] = mfma16(Hs[sr * LD32 + 16 * j + i16], zd, cd[j]);
                    ce[j] = mfma16(xt, DHs[sr * LD32 + 16 * j + i16], ce[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                st4(slab + wg_off_dec_w() + (size_t)(16 * w + i16) * LH + 16 * j + 4 * h4, cd[j]);
                st4(slab + wg_off_enc_w() + (size_t)(16 * j + i16) * PC_L + 16 * w + 4 * h4, ce[j]);
            }
        }
        PC_STAMP(14);
        if (TABLES) {
            // table gradients, transposed, as one-hot products C[j][t] = sum_r src[r][j] [dst[r] == t] on the BF16 matrix
            // cores, exactly: a source value is the sum of its three bf16 pieces (common.h split3), the one-hot operand is
            // 0 / 1, so every product is exact and the fp32 accumulator adds pieces of source rows in the matrix unit's
            // fixed order (bitwise reproducible).  One v_mfma_f32_32x32x16_bf16 covers 16 source rows -- one row block:
            // 5 + 1 k steps of 3 MFMAs per 32 x 32 block instead of 24 steps of v_mfma_f32_16x16x4_f32 per 16 x 16 block
            // (188 MFMAs per wave, 5.4 us -> 36 MFMAs).  Wave w: dims block w & 1, type blocks 2 (w >> 1), + 1 (T <= 128).
            // [Also measured for this phase: a run-time block count with the fp32 blocks -- every MFMA its own branch target,
            // 60 us; both slabs built in LDS by row-wise ds_add_f32, one wave per type, then copied out -- 10.8 us.]
            int* dsts = reinterpret_cast<int*>(DTs + TS * LD64);      // destination type of source row R = 16 rb + s; E_q's after them
            if (tid < TS * (FK + 2)) {
                const int rb = tid >> 4, sd = tid & 15;
                dsts[tid] = tid < TS * (K + 2) ? ints[sd * 8 + (rb < K ? 4 + rb : 2 + rb - K)] : -1;
            } else if (tid < TS * (FK + 3)) {
                dsts[tid] = ints[(tid - TS * (FK + 2)) * 8 + 1];
            }
            phase_sync();
            const int fr = lane & 31, fh = lane >> 5, jb = w & 1, tb0 = 2 * (w >> 1);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            // this lane's operand fragments of one k step: 8 consecutive source rows 8 fh .. + 7 of the row block
            auto src_frag = [&](const float* rows) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = rows[(8 * fh + q) * LD64 + 32 * jb + fr];
                return split3(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
            };
            auto onehot_frag = [&](const int* d, int t) {
                const int4 lo = *reinterpret_cast<const int4*>(d + 8 * fh), hi = *reinterpret_cast<const int4*>(d + 8 * fh + 4);
                u32x4 q;                                               // bf16 1.0 = 0x3f80; element 2i low half, 2i + 1 high half
                q[0] = (lo.x == t ? 0x3f80u : 0u) | (lo.y == t ? 0x3f800000u : 0u);
                q[1] = (lo.z == t ? 0x3f80u : 0u) | (lo.w == t ? 0x3f800000u : 0u);
                q[2] = (hi.x == t ? 0x3f80u : 0u) | (hi.y == t ? 0x3f800000u : 0u);
                q[3] = (hi.z == t ? 0x3f80u : 0u) | (hi.w == t ? 0x3f800000u : 0u);
                return __builtin_bit_cast(bf16x8, q);
            };
            auto store_blocks = [&](float* dst, const f32x16 (&c)[2]) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int t = 32 * (tb0 + u) + fr;
                    if (t < a.T)
#pragma unroll
                        for (int g = 0; g < 4; g++)                    // registers 4 g .. + 3: dims 8 g + 4 fh .. + 3 of the block
                            *reinterpret_cast<float4*>(dst + (size_t)t * PC_L + 32 * jb + 8 * g + 4 * fh) =
                                make_float4(c[u][4 * g], c[u][4 * g + 1], c[u][4 * g + 2], c[u][4 * g + 3]);
                }
            };
            f32x16 c[2];
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int r = 0; r < 16; r++) c[u][r] = 0.f;
#pragma unroll
            for (int rb = 0; rb < MBK + 2; rb++) {
                if (rb < K + 2) {
                    const Split3 sa = src_frag(ES + 16 * rb * LD64);
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const bf16x8 oh = onehot_frag(dsts + 16 * rb, 32 * (tb0 + u) + fr);
                        c[u] = mfma_bf16(sa.p2, oh, c[u]);
                        c[u] = mfma_bf16(sa.p1, oh, c[u]);
                        c[u] = mfma_bf16(sa.p0, oh, c[u]);
                    }
                }
            }
            store_blocks(slab + wg_off_ec(), c);
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int r = 0; r < 16; r++) c[u][r] = 0.f;
            {
                const Split3 sa = src_frag(DTs);
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const bf16x8 oh = onehot_frag(dsts + TS * (FK + 2), 32 * (tb0 + u) + fr);
                    c[u] = mfma_bf16(sa.p2, oh, c[u]);
                    c[u] = mfma_bf16(sa.p1, oh, c[u]);
                    c[u] = mfma_bf16(sa.p0, oh, c[u]);
                }
            }
            store_blocks(slab + wg_off_eq(a.T), c);
        }
        PC_STAMP(15);
    }
}

static size_t tile_lds_bytes(int T, bool sims_local, bool tables) {
    const int ldsims = sims_local ? ((T + 15) / 16 * 16 + 4) : 0;
    const size_t floats = (size_t)TS * LD64 + TS * LD32 + TS * LD64 + 2 * TS * LD128 + (size_t)TS * FK * LD64 +
                          (size_t)TS * FK * LD128 + TS * LD64 + TS * LD32 + TS * 8 + (size_t)TS * ldsims +
                          (tables ? (size_t)TS * (FK + 2) * LD64 + TS * LD64 + TS * (FK + 3) : 0);
    return floats * sizeof(float);
}

// ---------------------------------------------------------------------------------------------------------------
// Large tables: similarity row and top-K once per DISTINCT query type of the batch (dropout off: c is a function of
// the type alone).
//   present_types_kernel   bitmap of the batch's query types -> ascending list ulist[U] (one workgroup)
// then the three kernels of the per-sample form below with rows = the U listed types instead of the B samples
// (sample_hidden_kernel without a mask, sample_sims_max_kernel, sample_topk_refine_kernel -> topk_by_type[type][K]); U lives on
// the device, so the grids are sized for min(B, T) rows and the workgroups past U leave at once.
#define UT 64
#define PRESENT256_WORDS 2048                      /* LDS bitmap of the riding workgroup (joint_finish_kernel): num_types <= 65 536 */
#define PRESENT256_MAX_T (PRESENT256_WORDS * 32)
// inclusive prefix sum over the 1024 threads of a workgroup: the DPP scan inside each wave, the 16 wave totals through LDS -- two
// barriers (the Hillis-Steele form over LDS this replaces took twenty: 3 us of a 10 us single-workgroup kernel)
__device__ __forceinline__ int block_scan_1024(int v, unsigned* wsum /* [16] LDS */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_scan_incl(v);
    __syncthreads();                                    // (wsum may still be read from a previous call)
    if (lane == 63) wsum[w] = (unsigned)v;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) base += i < w ? (int)wsum[i] : 0;
    return v + base;
}

// pairs (optional): the batch is not built yet -- the query type of sample b is type_idx[pairs[3 b]] (data_loader.py:146)
__global__ __launch_bounds__(1024) void present_types_kernel(const int32_t* query_types, int B, int T, int32_t* ulist,
                                                             int32_t* n_u, const int32_t* pairs, const int32_t* type_idx,
                                                             int P) {
    extern __shared__ unsigned bits[];                  // [words] then scan scratch [1024]
    const int words = (T + 31) >> 5;
    unsigned* part = bits + words;
    for (int i = threadIdx.x; i < words; i += 1024) bits[i] = 0u;
    __syncthreads();
    for (int b0 = threadIdx.x; b0 < B; b0 += 4 * 1024) {          // four samples per thread and round: their two dependent loads overlap
        int t[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int b = b0 + 1024 * u;
            t[u] = b < B ? (pairs ? pairs[3 * b] : query_types[b]) : -1;
        }
        if (pairs) {
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (b0 + 1024 * u < B) t[u] = type_idx[(unsigned)t[u] < (unsigned)P ? t[u] : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if ((unsigned)t[u] < (unsigned)T) atomicOr(&bits[t[u] >> 5], 1u << (t[u] & 31));
    }
    __syncthreads();
    const int per = (words + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(words, lo + per);
    int cnt = 0;
    for (int i = lo; i < hi; i++) cnt += __popc(bits[i]);
    const int incl = block_scan_1024(cnt, part);
    int pos = incl - cnt;
    for (int i = lo; i < hi; i++) {
        unsigned m = bits[i];
        while (m) {
            const int bit = __ffs(m) - 1;
            m &= m - 1;
            ulist[pos++] = 32 * i + bit;
        }
    }
    if (threadIdx.x == 1023) *n_u = incl;
}

// The same list by ONE 256-thread workgroup riding in another launch (round 6): the list of step i + 1 depends on that step's
// labelled pairs alone, which an epoch call holds for every step in advance -- so it is formed during step i (an extra workgroup
// of sample_hidden_kernel, beside the G workgroups it has nothing to do with) into the other half of a double buffer, and step
// i + 1 starts with its hidden rows: the 10 us single-workgroup launch leaves the critical path of every step but an epoch's first.
// bits: LDS, (T + 31) / 32 words + 8.
__device__ __forceinline__ void present_types_body256(const int32_t* pairs, const int32_t* type_idx, int P, int B, int T,
                                                      int32_t* ulist, int32_t* n_u, unsigned* bits) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int words = (T + 31) >> 5;
    unsigned* wsum = bits + words;
    for (int i = tid; i < words; i += 256) bits[i] = 0u;
    __syncthreads();
    for (int b0 = tid; b0 < B; b0 += 8 * 256) {              // eight samples per thread and round: their two dependent loads overlap
        int t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int b = b0 + 256 * u; t[u] = b < B ? pairs[3 * b] : -1; }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (b0 + 256 * u < B) t[u] = type_idx[(unsigned)t[u] < (unsigned)P ? t[u] : 0];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if ((unsigned)t[u] < (unsigned)T) atomicOr(&bits[t[u] >> 5], 1u << (t[u] & 31));
    }
    __syncthreads();
    const int per = (words + 255) / 256;
    const int lo = tid * per, hi = min(words, lo + per);
    int cnt = 0;
    for (int i = lo; i < hi; i++) cnt += __popc(bits[i]);
    int incl = wave_scan_incl(cnt);
    if (lane == 63) wsum[w] = (unsigned)incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) base += i < w ? (int)wsum[i] : 0;
    incl += base;
    int pos = incl - cnt;
    for (int i = lo; i < hi; i++) {
        unsigned m = bits[i];
        while (m) {
            const int bit = __ffs(m) - 1;
            m &= m - 1;
            ulist[pos++] = 32 * i + bit;
        }
    }
    if (tid == 255) *n_u = incl;
}

// ---------------------------------------------------------------------------------------------------------------
// Large tables: the similarity row and its top-K per ROW -- a row is a distinct query type of the batch without dropout (above)
// and a SAMPLE with hidden-layer dropout (the reference as shipped: config.py:12 DROPOUT = 0.1, config.py:27 NUM_TYPES = 34800;
// type_transition.py:13-19): c = dec(mask_b (*) relu(enc t)) then differs from sample to sample, so the similarity row and its
// top-K exist per SAMPLE -- the [B,64] x [64,T] product of p_companion.py:60-63 (18 GFLOP at B = 4096), never written.
// The row is only used to SELECT the K types (the hinges read their two similarities from c and the E_c rows themselves), so
// it is formed through the 32-wide hidden layer instead of the 64-wide c:  sims[b][t] = E_c[t] . (dec_w hd_b + dec_b)
//   = G[t] . hd_b + g0[t]   with  G = E_c dec_w [T,32],  g0 = E_c dec_b [T]  (142 MFLOP, once per step), hd_b the dropped hidden
// row -- half the multiply-adds of c E_c^T.  (Another association of the same fp32 sums: like the reference's own BLAS order,
// it can only move a selection between two types whose similarities agree to rounding.)
//   sample_hidden_kernel     workgroups [0, nb_s): hd[b] for every sample (the step's own dropout mask: the tile kernel that
//                            follows regenerates the same bits) -> hd [B,32];  workgroups [nb_s, ...): G and g0
//   sample_sims_max_kernel   per chunk of 256 types (its G fragments resident in registers), walking tiles of SUT samples:
//                            sims = hd G[chunk]^T + g0 as fp32-grade products on the bf16 matrix cores, and of those only the MAXIMUM
//                            of every 64-type sub-chunk per sample -> cmax [B][T / 64] (no LDS image of the similarities)
//   sample_topk_refine_kernel  per sample: the K sub-chunks with the largest maxima hold the K best types; their 64 similarities
//                            each are formed again (lane = type) and selected exactly -> topk[b][K]
struct SampleHArgs {
    const float *enc_w, *enc_b, *dec_w, *dec_b, *eq, *ec;
    const int32_t *query_types, *pairs, *type_idx;
    int B, T, P, nb_s;
    DropCfg drop;
    float *hd, *G, *g0;
    const int32_t *ulist, *n_rows;          // rows = listed query types (dropout off): row b is type ulist[b], b < *n_rows; else NULL
    float* gnmax;                           // [T / 64 rounded up]: max |G[t]| per 64-type sub-chunk
};
static_assert(UT == 64, "a G workgroup of sample_hidden_kernel is one 64-type sub-chunk of sample_sims_max_kernel");

__global__ __launch_bounds__(256) void sample_hidden_kernel(SampleHArgs a) {
    __shared__ __attribute__((aligned(16))) float Tin[UT * LD64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ci = lane & 15, rh = lane >> 4;
    if ((int)blockIdx.x >= a.nb_s) {
        // ---- G[t][j] = sum_d E_c[t][d] dec_w[d][j], g0[t] = sum_d E_c[t][d] dec_b[d] for 64 types
        const int t0 = ((int)blockIdx.x - a.nb_s) * UT;
        const BFrag<4> f_g0 = load_b<PC_L, true>(a.dec_w, LH, 0, LH, lane), f_g1 = load_b<PC_L, true>(a.dec_w, LH, 16, LH, lane);
        for (int e = tid; e < UT * 16; e += 256) {
            const int r = e >> 4, c4 = (e & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t0 + r < a.T) v = *reinterpret_cast<const float4*>(a.ec + (size_t)(t0 + r) * PC_L + c4);
            *reinterpret_cast<float4*>(&Tin[r * LD64 + c4]) = v;
        }
        __syncthreads();
        const float* At = Tin + 16 * w * LD64;
        f32x4v a0[1] = {{0.f, 0.f, 0.f, 0.f}}, a1[1] = {{0.f, 0.f, 0.f, 0.f}};
        mul_b<4, 1>(At, LD64, 1, f_g0, a0, lane);
        mul_b<4, 1>(At, LD64, 1, f_g1, a1, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int t = t0 + 16 * w + 4 * rh + r;
            if (t < a.T) { a.G[(size_t)t * LH + ci] = a0[0][r]; a.G[(size_t)t * LH + 16 + ci] = a1[0][r]; }
        }
        float g0abs = 0.f;
        if (tid < UT && t0 + tid < a.T) {
            float s = 0.f;
            for (int d = 0; d < PC_L; d++) s += Tin[tid * LD64 + d] * a.dec_b[d];
            a.g0[t0 + tid] = s;
            g0abs = fabsf(s);
        }
        if (w == 0) {
            // the largest |g0[t]| of the sub-chunk, behind the norms (gnmax [nsub] | g0max [nsub]): pass 1 starts its accumulators
            // at g0 and pass 2 adds g0 last -- both round at the scale of |g0| + |G||hd|, which the |G||hd| bound alone does not
            // cover when |g0| is the larger of the two
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) g0abs = fmaxf(g0abs, __shfl_xor(g0abs, o, 64));
            if (lane == 0) a.gnmax[(a.T + UT - 1) / UT + t0 / UT] = g0abs;
        }
        // gnmax[sub] = the largest |G[t]|_2 of this workgroup's 64 types (= one sub-chunk of the similarity kernels): what bounds the
        // error of the two-piece products of sample_sims_max_kernel for that sub-chunk (Cauchy-Schwarz, see sample_topk_refine_kernel)
        float nmax = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) nmax = fmaxf(nmax, group16_sum(a0[0][r] * a0[0][r] + a1[0][r] * a1[0][r]));
        nmax = fmaxf(nmax, __shfl_xor(nmax, 16, 64));
        nmax = fmaxf(nmax, __shfl_xor(nmax, 32, 64));
        __syncthreads();                                   // (Tin is read no more: its first floats carry the four waves' maxima)
        if (lane == 0) Tin[w] = nmax;
        __syncthreads();
        if (tid == 0) a.gnmax[t0 / UT] = sqrtf(fmaxf(fmaxf(Tin[0], Tin[1]), fmaxf(Tin[2], Tin[3])));
        return;
    }
    const int b0 = blockIdx.x * UT;
    const int nrows = a.n_rows ? *a.n_rows : a.B;
    if (b0 >= nrows) return;
    const BFrag<4> f_e0 = load_b<PC_L, false>(a.enc_w, PC_L, 0, LH, lane), f_e1 = load_b<PC_L, false>(a.enc_w, PC_L, 16, LH, lane);
    const float bias_e0 = a.enc_b[ci], bias_e1 = a.enc_b[16 + ci];
    for (int e = tid; e < UT * 16; e += 256) {
        const int r = e >> 4, c4 = (e & 15) * 4, b = b0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < nrows) {
            // the query type as the tile kernel will validate it (an id outside its table is counted there and clamped to 0)
            int qt;
            if (a.ulist) qt = a.ulist[b];
            else if (a.pairs) { const int qi = a.pairs[3 * b]; qt = a.type_idx[(unsigned)qi < (unsigned)a.P ? qi : 0]; }
            else qt = a.query_types[b];
            if ((unsigned)qt >= (unsigned)a.T) qt = 0;
            v = *reinterpret_cast<const float4*>(a.eq + (size_t)qt * PC_L + c4);
        }
        *reinterpret_cast<float4*>(&Tin[r * LD64 + c4]) = v;
    }
    __syncthreads();
    // wave w owns samples [16 w, 16 w + 16) of the tile
    const float* At = Tin + 16 * w * LD64;
    f32x4v a0[1] = {{0.f, 0.f, 0.f, 0.f}}, a1[1] = {{0.f, 0.f, 0.f, 0.f}};
    mul_b<4, 1>(At, LD64, 1, f_e0, a0, lane);
    mul_b<4, 1>(At, LD64, 1, f_e1, a1, lane);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int b = b0 + 16 * w + 4 * rh + r;
        float x0 = a0[0][r] + bias_e0, x1 = a1[0][r] + bias_e1;
        x0 = x0 > 0.f ? x0 : 0.f;
        x1 = x1 > 0.f ? x1 : 0.f;
        if (a.drop.thr) {                                         // the mask of joint_tile_kernel's phase A, element for element
            float m[4];
            pc_dropout_keep4(a.drop, (unsigned)(b * (LH / 4) + (ci >> 2)), PC_DROP_STREAM_HIDDEN, m);
            x0 *= m[ci & 3];
            pc_dropout_keep4(a.drop, (unsigned)(b * (LH / 4) + ((16 + ci) >> 2)), PC_DROP_STREAM_HIDDEN, m);
            x1 *= m[ci & 3];
        }
        if (b < nrows) { a.hd[(size_t)b * LH + ci] = x0; a.hd[(size_t)b * LH + 16 + ci] = x1; }
    }
}

#define SUT 32        /* samples per tile of sample_sims_max_kernel */
#define PC_STC 256    /* types per chunk of sample_sims_max_kernel: four waves x one 64-type sub-chunk, < 128 VGPRs, FOUR workgroups per CU */
#define STC PC_STC
#define SWPS 4
#define HPL (SUT * LH)              /* bf16 elements of one piece plane of a tile's hd rows */
struct SampleSimsArgs {
    const float *hd, *G, *g0;
    int B, T, K, nchunks;
    float* part_val;                        // cmax [rows][4 nchunks]: the sub-chunk maxima
    const int32_t* n_rows;                  // device row count (listed query types), or NULL: B rows
    float* zero[2]; size_t nzero[2]; int zcols;      // rider: see TypeSimsArgs
};

// The product runs on the BF16 matrix cores: the contraction is LH = 32 wide, so ONE v_mfma_f32_16x16x32_bf16 covers a 16 x 16
// block's whole K.  Pass 1 only has to find the sub-chunks that CAN hold a row's best types, so it multiplies TWO bf16 pieces per
// operand (x = p0 + p1 + r, |r| < 2^-14 |x|: pieces by truncation, common.h split3's first two) in three products, p0 q1 + p1 q0 +
// p0 q0, smallest first: |sum - exact| <= 3 * 2^-14 |G[t]| |hd_r| (Cauchy-Schwarz over the 32 terms; the fp32 accumulation of the 96
// piece products adds < 2^-17 of the same), which pass 2 -- fp32 values -- turns into its candidate margin (PC_SS_EPS).  (The fp32-grade six-product form took
// 48 MFMAs per wave and tile, 23 us of the kernel's 42 at B = 4096, T = 34800; the first version's eight v_mfma_f32_16x16x4_f32 per
// block 8 x 32 clocks against 3 x 16.)  The hd rows of a tile are split ONCE, by the thread that fetched them (one float4 each), and
// staged as two bf16 piece planes; the chunk's G fragments are split once per workgroup and stay in registers.
#define PC_SS_G0 9.6e-7f       /* 8 * 2^-23: the roundings of pass 1's accumulation from g0 and of pass 2's final + g0, relative to the sub-chunk's largest |g0| */
#define PC_SS_EPS 2.5e-4f      /* > 3 * 2^-14 + 2^-17 + 2^-19 = 1.93e-4: bound of |pass-1 value - pass-2 value| / (|G[t]| |hd_r|), with a quarter to spare */
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16_bf16(const bf16x8& a, const bf16x8& b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// the first two pieces of four values (split3's arithmetic): piece p as two dwords = four bf16 in element order
__device__ __forceinline__ void split2_4(const float4& x, uint2 (&q)[2]) {
    const float v[4] = {x.x, x.y, x.z, x.w};
    unsigned u0[4], u1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        u0[i] = __float_as_uint(v[i]);
        u1[i] = __float_as_uint(v[i] - __uint_as_float(u0[i] & 0xffff0000u));
    }
    q[0] = make_uint2(__builtin_amdgcn_perm(u0[1], u0[0], 0x07060302u), __builtin_amdgcn_perm(u0[3], u0[2], 0x07060302u));
    q[1] = make_uint2(__builtin_amdgcn_perm(u1[1], u1[0], 0x07060302u), __builtin_amdgcn_perm(u1[3], u1[2], 0x07060302u));
}

// Pass 1 of 2: the MAXIMUM of every 64-type sub-chunk per sample -- cmax[b][sub] -- and nothing else.  The K best types of a sample
// lie in the K sub-chunks with the largest maxima (an element of the top K is >= the K-th best overall >= the K-th largest
// sub-chunk maximum, and so is the maximum of its own sub-chunk), so the exact selection -- indices, ties, two-word keys -- only has
// to look at K x 64 of the T similarities of a sample (pass 2, sample_topk_refine_kernel); the 142 M elements of a step each cost
// HALF a VALU instruction here (v_max3_f32 over the lane's four column blocks, then one 16-lane row maximum per sample) where keeping
// a sorted top-4 with indices per 16-lane row cost 15 per element all told (the first form of this kernel: 126 us at B = 4096,
// T = 34800, of which 87 the selection).  A wave owns the four column blocks of ONE sub-chunk (types [64 sub, 64 sub + 64)): its
// results never meet another wave's -- no LDS image of the similarities, no barrier between product and selection.  The accumulators
// start at g0[t] (-inf for the tail chunk's types >= T: zero G fragments leave it there).
__global__ __launch_bounds__(256, SWPS) void sample_sims_max_kernel(SampleSimsArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 Hp[2 * HPL];   // [2 pieces][SUT][LH] bf16: lane (i, h) of an A fragment reads
                                                                  // the 16 B at row i, k = 8 h of a plane (1 KB per 16 rows, dense)
    if ((int)blockIdx.x >= a.nchunks) {
        const size_t wg = ((size_t)blockIdx.x - a.nchunks) * gridDim.y + blockIdx.y, nwg = (size_t)a.zcols * gridDim.y;
        for (int i = 0; i < 2; i++) {
            if (!a.zero[i]) continue;
            const size_t n4 = a.nzero[i] / 4;
            for (size_t e = wg * 256 + threadIdx.x; e < n4; e += nwg * 256)
                reinterpret_cast<float4*>(a.zero[i])[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const int nrows = a.n_rows ? *a.n_rows : a.B;
    if ((int)blockIdx.y * SUT >= nrows) return;             // (rows = listed types: the grid is sized for their capacity)
    const int t0 = blockIdx.x * STC;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ci = lane & 15, rh = lane >> 4;
    constexpr int NBW = STC / 16 / 4;                       // column blocks per wave: NBW x 16 = 64 types = one sub-chunk
    static_assert(NBW == 4 && SUT == 32, "a wave owns one 64-type sub-chunk of a 32-sample tile: eight (sample) slots per lane");
    const int sub = 4 * blockIdx.x + w, nsub = 4 * a.nchunks;
    Split3 f_s[NBW];                                        // lane (j = ci, h = rh): G[64 sub + 16 q + j][8 h .. 8 h + 7] in pieces (the first two are used)
    float g0v[NBW];
#pragma unroll
    for (int q = 0; q < NBW; q++) {
        const int t = t0 + 64 * w + 16 * q + ci;
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        if (t < a.T) {
            lo = *reinterpret_cast<const float4*>(a.G + (size_t)t * LH + 8 * rh);
            hi = *reinterpret_cast<const float4*>(a.G + (size_t)t * LH + 8 * rh + 4);
        }
        f_s[q] = split3(lo, hi);
        g0v[q] = t < a.T ? a.g0[t] : -INFINITY;
    }
    // this thread's 16-B piece of a tile's hd rows (SUT x 32 floats = one float4 per thread), requested a tile ahead
    const int pr = tid >> 3, pc4 = (tid & 7) * 4;
    auto fetch = [&](int u0) {
        return u0 + pr < nrows ? *reinterpret_cast<const float4*>(a.hd + (size_t)(u0 + pr) * LH + pc4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const int ustep = gridDim.y * SUT;
    float4 nxt = fetch(blockIdx.y * SUT);
    for (int u0 = blockIdx.y * SUT; u0 < nrows; u0 += ustep) {
        __syncthreads();                                   // (the previous tile's fragments have been read)
        {
            uint2 pq[2];
            split2_4(nxt, pq);
#pragma unroll
            for (int p3 = 0; p3 < 2; p3++) *reinterpret_cast<uint2*>(Hp + p3 * HPL + pr * LH + pc4) = pq[p3];
        }
        if (u0 + ustep < nrows) nxt = fetch(u0 + ustep);
        __syncthreads();
        bf16x8 ap[SUT / 16][2];                            // lane (i = ci, h = rh): hd[16 m + i][8 h .. 8 h + 7], piece p
#pragma unroll
        for (int m = 0; m < SUT / 16; m++)
#pragma unroll
            for (int p3 = 0; p3 < 2; p3++)
                ap[m][p3] = *reinterpret_cast<const bf16x8*>(Hp + p3 * HPL + (16 * m + ci) * LH + 8 * rh);
        f32x4v acc[NBW][SUT / 16];
#pragma unroll
        for (int q = 0; q < NBW; q++)
#pragma unroll
            for (int m = 0; m < SUT / 16; m++) acc[q][m] = f32x4v{g0v[q], g0v[q], g0v[q], g0v[q]};
        // term by term over the NBW x SUT / 16 independent accumulators (no back-to-back dependent MFMAs), smallest first
#define PC_SS_TERM(PA, QB)                                                                              \
        _Pragma("unroll") for (int q = 0; q < NBW; q++)                                                 \
            _Pragma("unroll") for (int m = 0; m < SUT / 16; m++) acc[q][m] = mfma16_bf16(ap[m][PA], f_s[q].QB, acc[q][m]);
        PC_SS_TERM(1, p0) PC_SS_TERM(0, p1) PC_SS_TERM(0, p0)
#undef PC_SS_TERM
        // slot s = 4 m + r of a lane is sample 16 m + 4 rh + r: the maximum over the lane's four column blocks ...
        // (one asm block per row block, opened by the wait states a VALU read of a matrix-core result needs -- 11 after an 8-pass
        // MFMA: the hazard recogniser does not look inside inline asm, and without them the maxima were read before they were
        // written: a selection that changed from run to run)
        float v[8];
#pragma unroll
        for (int m = 0; m < SUT / 16; m++)
            asm("s_nop 15\n\t"
                "v_max3_f32 %0, %4, %8, %12\n\tv_max3_f32 %1, %5, %9, %13\n\tv_max3_f32 %2, %6, %10, %14\n\tv_max3_f32 %3, %7, %11, %15\n\t"
                "v_max_f32 %0, %0, %16\n\tv_max_f32 %1, %1, %17\n\tv_max_f32 %2, %2, %18\n\tv_max_f32 %3, %3, %19"
                : "=&v"(v[4 * m]), "=&v"(v[4 * m + 1]), "=&v"(v[4 * m + 2]), "=&v"(v[4 * m + 3])
                : "v"(acc[0][m][0]), "v"(acc[0][m][1]), "v"(acc[0][m][2]), "v"(acc[0][m][3]),
                  "v"(acc[1][m][0]), "v"(acc[1][m][1]), "v"(acc[1][m][2]), "v"(acc[1][m][3]),
                  "v"(acc[2][m][0]), "v"(acc[2][m][1]), "v"(acc[2][m][2]), "v"(acc[2][m][3]),
                  "v"(acc[3][m][0]), "v"(acc[3][m][1]), "v"(acc[3][m][2]), "v"(acc[3][m][3]));
        // ... and over the sixteen lanes of the row group (the sub-chunk's other 60 types): eight independent rotate-and-maximum
        // chains interleaved, so a DPP read never follows the write of its register by fewer than the two wait states it needs
#define PC_SS_ROR(CTRL)                                                                                                        \
        asm("s_nop 1\n\t"                                                                                                      \
            "v_max_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
            "v_max_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
            "v_max_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
            "v_max_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %7, %7, %7 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
            "s_nop 1"                                                                                                          \
            : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
        PC_SS_ROR("row_ror:1") PC_SS_ROR("row_ror:2") PC_SS_ROR("row_ror:4") PC_SS_ROR("row_ror:8")
#undef PC_SS_ROR
        // lane ci < 8 of a row group stores slot ci
        float pick = v[0];
#pragma unroll
        for (int sidx = 1; sidx < 8; sidx++) pick = ci == sidx ? v[sidx] : pick;
        const int smp = u0 + 16 * (ci >> 2) + 4 * rh + (ci & 3);
        if (ci < 8 && smp < nrows) a.part_val[(size_t)smp * nsub + sub] = pick;
    }
}

// Pass 2 of 2: one wave per row.  Pass 1's maximum of sub-chunk S is within e_S = PC_SS_EPS |hd_r| gnmax[S] + PC_SS_G0 g0max[S] of the true
// one (the second term: the roundings at the scale of g0, where the accumulation starts and ends), so with
// L_S = cmax_S - e_S and tau = the K-th largest L: K different sub-chunks hold an element >= tau, hence the K-th best similarity of the
// row is >= tau, and a sub-chunk with cmax_S + e_S < tau cannot hold one of the K best.  Every other sub-chunk (exactly K of them
// unless maxima come within the error bound of each other, or tie) has its 64 similarities formed again in fp32, and the exact
// selection -- two-word keys (value, ~index): descending, ties -> the lower index, like torch.topk / pc_topk_rows -- runs over those.
// 100 MFLOP and 100 MB of L2 reads per step at B = 4096, K = 3.
__global__ __launch_bounds__(256) void sample_topk_refine_kernel(const float* cmax, int nsub, const float* hd, const float* G,
                                                                 const float* g0, const float* gnmax, int B, int T, int K,
                                                                 int32_t* topk, const int32_t* ulist, const int32_t* n_rows) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= (n_rows ? *n_rows : B)) return;                 // wave-uniform
    const int orow = ulist ? ulist[b] : b;                   // rows = listed query types: the result is filed under the type
    const float* row = cmax + (size_t)b * nsub;
    const int kq = lane & 7;
    const float4 h4 = *reinterpret_cast<const float4*>(hd + (size_t)b * LH + 4 * kq);
    float hn = h4.x * h4.x + h4.y * h4.y + h4.z * h4.z + h4.w * h4.w;
    hn += __shfl_xor(hn, 1, 64); hn += __shfl_xor(hn, 2, 64); hn += __shfl_xor(hn, 4, 64);
    const float escale = PC_SS_EPS * sqrtf(hn);
    const int nsub_real = (T + 63) >> 6;                     // gnmax [nsub_real] norms, then [nsub_real] largest |g0|
    float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;      // the lane's four largest lower bounds, descending
    for (int i = lane; i < nsub; i += 64) {
        const float x = 64 * i < T ? row[i] - (escale * gnmax[i] + PC_SS_G0 * gnmax[nsub_real + i]) : -INFINITY;
        m3 = __builtin_amdgcn_fmed3f(m2, m3, x); m2 = __builtin_amdgcn_fmed3f(m1, m2, x); m1 = __builtin_amdgcn_fmed3f(m0, m1, x);
        m0 = fmaxf(m0, x);
    }
    float tau = -INFINITY;
    for (int r = 0; r < K; r++) {
        float m = m0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        tau = m;
        const bool own = m0 == m;                            // (equal bounds in two lanes retire together: tau can only come out lower)
        m0 = own ? m1 : m0; m1 = own ? m2 : m1; m2 = own ? m3 : m2; m3 = own ? -INFINITY : m3;
    }
    // A sub-chunk's 64 rows of G are 8 KB in a row: the wave reads them as eight dense 1 KB requests -- request j, lane l: the 16 B
    // at float4 index 64 j + l, i.e. columns [4 (l & 7), 4 (l & 7) + 4) of row 8 j + (l >> 3) -- instead of every lane walking its
    // own 128-B row (64 cache lines per request).  The eight lanes of a row add their four-column partials in a fixed tree.
    unsigned kh[FK], kl[FK];
#pragma unroll
    for (int j = 0; j < FK; j++) { kh[j] = 0u; kl[j] = 0u; }
    for (int i0 = 0; i0 < nsub; i0 += 64) {
        const int i = i0 + lane;
        const bool live = i < nsub && 64 * i < T;
        const float x = live ? row[i] + (escale * gnmax[i] + PC_SS_G0 * gnmax[nsub_real + i]) : -INFINITY;      // the sub-chunk's upper bound
        unsigned long long mask = __ballot(live && x >= tau);
        while (mask) {                                       // wave-uniform: K trips in all (more on tied maxima)
            const int j0 = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int tb = 64 * (i0 + j0);                   // first type of the sub-chunk
            float4 gv[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int t = tb + 8 * j + (lane >> 3);
                gv[j] = t < T ? *(reinterpret_cast<const float4*>(G + (size_t)tb * LH) + 64 * j + lane) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const int tmine = tb + 8 * kq + (lane >> 3);     // the row this lane enters into its list: request kq's
            const float g0m = tmine < T ? g0[tmine] : 0.f;
            float mine = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                float part = fmaf(gv[j].w, h4.w, fmaf(gv[j].z, h4.z, fmaf(gv[j].y, h4.y, gv[j].x * h4.x)));
                part = dpp_row_add<0xB1>(part);              // + the quad neighbour (lane ^ 1), on the DPP network
                part = dpp_row_add<0x4E>(part);              // + the other pair of the quad (lane ^ 2)
                part = dpp_row_add<0x141>(part);             // + the other quad of the eight (row_half_mirror)
                mine = kq == j ? part : mine;
            }
            if (tmine < T) {
                unsigned hk = ord_f32(mine + g0m), lk = ~(unsigned)tmine;
#pragma unroll
                for (int q = 0; q < FK; q++) {               // insertion into the sorted (descending) list
                    const bool gt = hk > kh[q] || (hk == kh[q] && lk > kl[q]);
                    const unsigned th = gt ? kh[q] : hk, tl = gt ? kl[q] : lk;
                    kh[q] = gt ? hk : kh[q]; kl[q] = gt ? lk : kl[q];
                    hk = th; lk = tl;
                }
            }
        }
    }
    for (int r = 0; r < K; r++) {
        unsigned bh = kh[0], bl = kl[0];
        wave_maxkey(bh, bl);
        if (kh[0] == bh && kl[0] == bl) {                    // pop the owner's head
#pragma unroll
            for (int j = 0; j < FK - 1; j++) { kh[j] = kh[j + 1]; kl[j] = kl[j + 1]; }
            kh[FK - 1] = 0u; kl[FK - 1] = 0u;
        }
        // (no candidate at all -- a row of NaN similarities, a diverged model -- must still leave an index inside the table: the
        // tile kernel gathers E_c rows by it)
        if (lane == 0) topk[(size_t)orow * K + r] = (bh | bl) ? (int)~bl : r;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Gradient products of the step, right-sized: every weight / table gradient is a "rows^T x rows" product over the
// row buffers the tile kernel left (d itm_w = dpi^T q, d typ_w = dtp^T e, d dec_w = dc^T h, d enc_w = dh^T t, and the two
// [T,64] tables as one-hot products, the type hinge's two rows per sample included).  The generic 128 x 128-tile kernel
// spends most of its MFMAs on zeros here (64 x 32, 32 x 64, T x 64 outputs); this one walks 16-sample subtiles with
// 16 x 16 x 4 blocks sized to each product, accumulates a workgroup's share of the batch in registers and writes ONE
// slab per workgroup (summed in fixed order by the finish kernel).
//   C[i][o] = sum_s X[s][i] Z[s][o]: the lane's four results are four consecutive i of one o -> one 16-B store at
//   slab[o * Ni + i].  Operands come from LDS row-major images (row stride = width + 16 floats: the four sample rows one
//   MFMA touches sit 16 banks apart).
static_assert(TS == 16, "one gradient slab per 16-sample tile");
#define WG_S 16            /* samples per workgroup (one 16-sample subtile: 256 workgroups at B = 4096; two subtiles per
                              workgroup halve the slab bytes but leave half the CUs idle: 34 vs 2x us, measured) */
#define WLD128 144
#define WLD64 80
#define WLD32 48
struct WgradArgs {
    const float *table, *eq, *ec;                         // gather sources: product rows, E_q, E_c
    const int32_t *query_idx, *query_types, *topk;        // [B], [B], [B,K]
    const float *dpi, *dtp, *dc, *h, *dh, *dt, *ecsrc;    // row buffers (see FusedArgs)
    const int32_t* ecidx;                                 // [B (K + 2)]
    int B, T, K;
    float* slabs; int slab_floats;                        // per workgroup: itm_w | itm_b | typ_w | typ_b | dec_w | dec_b | enc_w | enc_b | E_c | E_q
};
// NTW: 16-type column blocks of a table product per wave (16 serves T <= 512; T <= 128 never comes here)
template <int NTW, int KC>
__global__ __launch_bounds__(512) void joint_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* DPI = sm;                          // [16][WLD128]
    float* Q = DPI + 16 * WLD128;             // [16][WLD128]
    float* DTP = Q + 16 * WLD128;             // [16 FK][WLD128]   rows b * K + k
    float* E = DTP + 16 * FK * WLD128;        // [16 FK][WLD64]
    float* DC = E + 16 * FK * WLD64;          // [16][WLD64]
    float* Hh = DC + 16 * WLD64;              // [16][WLD32]
    float* DH = Hh + 16 * WLD32;              // [16][WLD32]
    float* Tq = DH + 16 * WLD32;              // [16][WLD64]
    float* ES = Tq + 16 * WLD64;              // [16 (FK + 2)][WLD64]  dE_c source rows: 16 K selected-type rows, 16 + 16 hinge rows
    float* DT = ES + 16 * (FK + 2) * WLD64;   // [16][WLD64]
    int* idx = reinterpret_cast<int*>(DT + 16 * WLD64);     // [16 (FK + 2)] E_c destinations, then [16] E_q destinations
    const int K = KC ? KC : a.K;
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, h = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntb = (a.T + 15) >> 4;                                    // 16-type column blocks of the tables
    constexpr int MBK = KC ? KC : FK;

    // One 16-sample subtile per workgroup (WG_S == 16): every product's accumulators live only from its own MFMA loop to its
    // slab store -- the widest group (a table product: NTW x 4 registers) instead of all 184 at once (which left 28-31 VGPRs
    // in scratch under the 256-register budget of two waves per SIMD).
    static_assert(WG_S == 16, "one subtile per workgroup: accumulators are stored right behind their product");
    float bsum = 0.f;                          // bias column owned by this thread (tid < 352)
    float* slab = a.slabs + (size_t)blockIdx.x * a.slab_floats;
    auto st4 = [&](float* dst, const f32x4v& v) { *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]); };
    const int b0 = blockIdx.x * WG_S;
    if (b0 >= a.B) return;                                              // workgroup-uniform (the grid is ceil(B / 16))
    {
        // ---- stage the subtile: rows past the batch are zero
        auto rowf4 = [&](float* dst, int ld, const float* src, int width4, int rows, int first_row, int row_limit) {
            for (int e = tid; e < rows * width4; e += 512) {
                const int r = e / width4, c4 = (e % width4) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (first_row + r < row_limit) v = *reinterpret_cast<const float4*>(src + (size_t)(first_row + r) * (width4 * 4) + c4);
                *reinterpret_cast<float4*>(&dst[r * ld + c4]) = v;
            }
        };
        rowf4(DPI, WLD128, a.dpi, 32, 16, b0, a.B);
        rowf4(DTP, WLD128, a.dtp, 32, 16 * K, b0 * K, a.B * K);
        rowf4(DC, WLD64, a.dc, 16, 16, b0, a.B);
        rowf4(Hh, WLD32, a.h, 8, 16, b0, a.B);
        rowf4(DH, WLD32, a.dh, 8, 16, b0, a.B);
        rowf4(DT, WLD64, a.dt, 16, 16, b0, a.B);
        rowf4(ES, WLD64, a.ecsrc, 16, 16 * K, b0 * K, a.B * K);
        rowf4(ES + 16 * K * WLD64, WLD64, a.ecsrc + (size_t)a.B * K * PC_L, 16, 16, b0, a.B);
        rowf4(ES + 16 * (K + 1) * WLD64, WLD64, a.ecsrc + (size_t)a.B * (K + 1) * PC_L, 16, 16, b0, a.B);
        for (int e = tid; e < 16 * 32; e += 512) {                     // q = E_prod[query_idx]
            const int r = e >> 5, c4 = (e & 31) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b0 + r < a.B) v = *reinterpret_cast<const float4*>(a.table + (size_t)a.query_idx[b0 + r] * PC_D + c4);
            *reinterpret_cast<float4*>(&Q[r * WLD128 + c4]) = v;
        }
        for (int e = tid; e < 16 * K * 16; e += 512) {                 // e = E_c[topk]
            const int r = e >> 4, c4 = (e & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b0 * K + r < a.B * K) v = *reinterpret_cast<const float4*>(a.ec + (size_t)a.topk[(size_t)b0 * K + r] * PC_L + c4);
            *reinterpret_cast<float4*>(&E[r * WLD64 + c4]) = v;
        }
        for (int e = tid; e < 16 * 16; e += 512) {                     // t = E_q[query_types]
            const int r = e >> 4, c4 = (e & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b0 + r < a.B) v = *reinterpret_cast<const float4*>(a.eq + (size_t)a.query_types[b0 + r] * PC_L + c4);
            *reinterpret_cast<float4*>(&Tq[r * WLD64 + c4]) = v;
        }
        if (tid < 16 * (K + 2)) {
            // destinations of the dE_c source rows, in the rows' order: [16 K selected | 16 pos | 16 neg]; -1 = no row
            int r = tid, v = -1;
            if (r < 16 * K) { if (b0 * K + r < a.B * K) v = a.ecidx[(size_t)b0 * K + r]; }
            else if (r < 16 * (K + 1)) { if (b0 + r - 16 * K < a.B) v = a.ecidx[(size_t)a.B * K + b0 + r - 16 * K]; }
            else if (b0 + r - 16 * (K + 1) < a.B) v = a.ecidx[(size_t)a.B * (K + 1) + b0 + r - 16 * (K + 1)];
            idx[r] = v;
        } else if (tid >= 448 && tid < 464) {
            const int r = tid - 448;
            idx[16 * (FK + 2) + r] = b0 + r < a.B ? a.query_types[b0 + r] : -1;
        }
        __syncthreads();
        // ---- biases: column sums of the Z images
        if (tid < 128) { for (int r = 0; r < 16; r++) bsum += DPI[r * WLD128 + tid]; }
        else if (tid < 256) { for (int r = 0; r < 16 * K; r++) bsum += DTP[r * WLD128 + tid - 128]; }
        else if (tid < 320) { for (int r = 0; r < 16; r++) bsum += DC[r * WLD64 + tid - 256]; }
        else if (tid < 352) { for (int r = 0; r < 16; r++) bsum += DH[r * WLD32 + tid - 320]; }
    }
    // result register r of a block = input index 4 h + r (+ block), column = output index
    {   // ---- d itm_w^T: wave w owns input block w (16 of the 128 q dims) x all 8 output blocks
        f32x4v c_itm[8];
#pragma unroll
        for (int i = 0; i < 8; i++) c_itm[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int q = 0; q < 4; q++) {
            const float av = Q[(4 * q + h) * WLD128 + 16 * w + i16];
#pragma unroll
            for (int ob = 0; ob < 8; ob++) c_itm[ob] = mfma16(av, DPI[(4 * q + h) * WLD128 + 16 * ob + i16], c_itm[ob]);
        }
#pragma unroll
        for (int ob = 0; ob < 8; ob++) st4(slab + wg_off_itm_w() + (size_t)(16 * ob + i16) * PC_D + 16 * w + 4 * h, c_itm[ob]);
    }
    {   // ---- d typ_w^T: input block w & 3 (of 4) x output blocks 4 (w >> 2) .. + 3; 16 K rows
        f32x4v c_typ[4];
#pragma unroll
        for (int i = 0; i < 4; i++) c_typ[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int q = 0; q < 4 * MBK; q++) {
            if (q < 4 * K) {
                const float av = E[(4 * q + h) * WLD64 + 16 * (w & 3) + i16];
#pragma unroll
                for (int ob = 0; ob < 4; ob++)
                    c_typ[ob] = mfma16(av, DTP[(4 * q + h) * WLD128 + 16 * (4 * (w >> 2) + ob) + i16], c_typ[ob]);
            }
        }
#pragma unroll
        for (int ob = 0; ob < 4; ob++)
            st4(slab + wg_off_typ_w() + (size_t)(16 * (4 * (w >> 2) + ob) + i16) * PC_L + 16 * (w & 3) + 4 * h, c_typ[ob]);
    }
    {   // ---- d dec_w^T [32 in][64 out]: input block w & 1, output block w >> 1;  d enc_w^T [64 in][32 out]: w & 3, w >> 2
        f32x4v c_dec = f32x4v{0.f, 0.f, 0.f, 0.f}, c_enc = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
            c_dec = mfma16(Hh[(4 * q + h) * WLD32 + 16 * (w & 1) + i16], DC[(4 * q + h) * WLD64 + 16 * (w >> 1) + i16], c_dec);
            c_enc = mfma16(Tq[(4 * q + h) * WLD64 + 16 * (w & 3) + i16], DH[(4 * q + h) * WLD32 + 16 * (w >> 2) + i16], c_enc);
        }
        st4(slab + wg_off_dec_w() + (size_t)(16 * (w >> 1) + i16) * LH + 16 * (w & 1) + 4 * h, c_dec);
        st4(slab + wg_off_enc_w() + (size_t)(16 * (w >> 2) + i16) * PC_L + 16 * (w & 3) + 4 * h, c_enc);
    }
    // ---- table gradients, transposed: C[j][t] = sum_r src[r][j] [idx[r] == t]; wave: dims block w & 3, type blocks
    // (w >> 2) + 2 n
    {
        f32x4v c_ec[NTW];
#pragma unroll
        for (int i = 0; i < NTW; i++) c_ec[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int q = 0; q < 4 * (MBK + 2); q++) {
            if (q < 4 * (K + 2)) {
                const float av = ES[(4 * q + h) * WLD64 + 16 * (w & 3) + i16];
                const int d = idx[4 * q + h];
#pragma unroll
                for (int n = 0; n < NTW; n++) {
                    const int tb = (w >> 2) + 2 * n;
                    if (tb < ntb) c_ec[n] = mfma16(av, d == 16 * tb + i16 ? 1.f : 0.f, c_ec[n]);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NTW; n++) {
            const int t = 16 * ((w >> 2) + 2 * n) + i16;
            if (t < a.T) st4(slab + wg_off_ec() + (size_t)t * PC_L + 16 * (w & 3) + 4 * h, c_ec[n]);
        }
    }
    {
        f32x4v c_eq[NTW];
#pragma unroll
        for (int i = 0; i < NTW; i++) c_eq[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int q = 0; q < 4; q++) {
            const float av = DT[(4 * q + h) * WLD64 + 16 * (w & 3) + i16];
            const int d = idx[16 * (FK + 2) + 4 * q + h];
#pragma unroll
            for (int n = 0; n < NTW; n++) {
                const int tb = (w >> 2) + 2 * n;
                if (tb < ntb) c_eq[n] = mfma16(av, d == 16 * tb + i16 ? 1.f : 0.f, c_eq[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < NTW; n++) {
            const int t = 16 * ((w >> 2) + 2 * n) + i16;
            if (t < a.T) st4(slab + wg_off_eq(a.T) + (size_t)t * PC_L + 16 * (w & 3) + 4 * h, c_eq[n]);
        }
    }
    if (tid < 128) slab[wg_off_itm_b() + tid] = bsum;
    else if (tid < 256) slab[wg_off_typ_b() + tid - 128] = bsum;
    else if (tid < 320) slab[wg_off_dec_b() + tid - 256] = bsum;
    else if (tid < 352) slab[wg_off_enc_b() + tid - 320] = bsum;
}

static size_t wgrad_lds_bytes() {
    return ((size_t)2 * 16 * WLD128 + 16 * FK * WLD128 + 16 * FK * WLD64 + 16 * WLD64 + 2 * 16 * WLD32 + 16 * WLD64 +
            16 * (FK + 2) * WLD64 + 16 * WLD64 + 16 * (FK + 2) + 16) * 4;
}

// ---------------------------------------------------------------------------------------------------------------
// Finish: per parameter tensor, the fixed-order sum of its gradient slabs (written by the grouped rows^T x rows launch)
// -> .grad; the two loss means; and -- when the caller handed its moments -- torch.optim.Adam's update on the element
// just reduced.  Jobs with nsplit == 0 have their gradient complete in `grad` already (atomics path of large tables).
#define FIN_JOBS 12
struct FinishJob { const float* slabs; size_t stride; int nsplit, n; float *grad, *param, *m, *v; };
struct FinishArgs {
    FinishJob job[FIN_JOBS]; int block0[FIN_JOBS + 1], njobs;
    const float *part_type, *part_item; int B, K; float alpha; float* losses;
    const int64_t* step_count; double lr, beta1, beta2, eps; int adam;
    int sl16;                                             // 16 slab lanes per output float4 (else 8): see joint_finish_kernel
    // look-ahead rider (workgroup 0 when next_pairs != NULL, the others shifted by one): the distinct-query-type list of the
    // NEXT step of an epoch call (present_types_body256)
    const int32_t *next_pairs, *type_idx; int P, next_B, T; int32_t *next_ulist, *next_n_u;
};

__global__ __launch_bounds__(256) void joint_finish_kernel(FinishArgs a) {
    __shared__ float r0[256], r1[256];
    __shared__ float scal[2];
    __shared__ unsigned pbits[PRESENT256_WORDS + 8];
    // (the rider is workgroup 0: dispatched first, it runs beside the whole launch -- as the LAST workgroup it started when the
    // others were done and the launch lasted its 10 us longer: 18 -> 27 us)
    if (a.next_pairs && blockIdx.x == 0) {
        present_types_body256(a.next_pairs, a.type_idx, a.P, a.next_B, a.T, a.next_ulist, a.next_n_u, pbits);
        return;
    }
    // (the hinge means come next, for the same reason: at num_types > 512 the grid is thousands of streaming workgroups and its
    // last one starts when the others are nearly done)
    const int b = (int)blockIdx.x - (a.next_pairs ? 2 : 1);
    if (b == -1) {                                      // the extra workgroup: the two hinge means
        float x = 0.f, y = 0.f;
        for (int i = threadIdx.x; i < a.B; i += 256) { x += a.part_type[i]; y += a.part_item[i]; }
        r0[threadIdx.x] = x; r1[threadIdx.x] = y;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if (threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float tl = r0[0] / (float)a.B, il = r1[0] / ((float)a.B * (float)a.K);
            a.losses[0] = a.alpha * il + (1.0f - a.alpha) * tl;
            a.losses[1] = tl;
            a.losses[2] = il;
        }
        return;
    }
    if (a.adam) {
        if (threadIdx.x == 0) {                         // torch's single-tensor path: scalars in fp64, rounded at use
            const int64_t t = *a.step_count;            // (already advanced by the tile kernel of this step)
            scal[0] = (float)(a.lr / (1.0 - pow(a.beta1, (double)t)));
            scal[1] = (float)sqrt(1.0 - pow(a.beta2, (double)t));
        }
        __syncthreads();
    }
    int j = 0;
#pragma unroll
    for (int i = 1; i < FIN_JOBS; i++) j += (i < a.njobs && b >= a.block0[i]) ? 1 : 0;
    const FinishJob& jb = a.job[j];
    // 8 or 16 slab lanes share one float4 of outputs: lane g sums slabs g, g + 8 (16), ...; a wave is 8 consecutive float4s (one
    // 128-byte line of every slab it reads) x 8 slab lanes; with 16 the other 8 slab lanes are the next wave; the partial sums fold
    // in a fixed order (xor over the wave's slab lanes, then the wave pair through LDS).  Sixteen where the slab jobs are the whole
    // kernel (T <= 512): the walk is bound by the cache lines a CU keeps in flight, so the kernel lasts as long as the CU with the
    // most workgroups -- 330 workgroups of eight slab lanes put two on 74 of the 256 CUs and one on the rest (0.0459 -> 0.0445 ms at
    // T = 100; beside the big tables' streaming jobs twice the workgroups cost 1.8 us instead).
    // (a job whose gradient is complete already -- nsplit == 0 -- gives every lane its own float4: pure streaming)
    __shared__ float4 xw[2][8];
    const bool direct = jb.nsplit == 0;
    const int tt = threadIdx.x, jl = tt & 7, gl = (tt >> 3) & 7, pair = tt >> 7, gh = a.sl16 ? (tt >> 6) & 1 : 0, sl = a.sl16 ? 16 : 8;
    const int j4 = direct ? (b - a.block0[j]) * 256 + tt
                          : a.sl16 ? (b - a.block0[j]) * 16 + pair * 8 + jl : (b - a.block0[j]) * 32 + (tt >> 6) * 8 + jl;
    const int g = direct ? 0 : gl + 8 * gh;
    const bool live = j4 * 4 < jb.n;
    if (direct && !live) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // Adam's operands are requested before the slab walk (they do not depend on it)
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), mv = pv, vv = pv;
    if (a.adam && g == 0 && live) {
        if (!direct) pv = *reinterpret_cast<const float4*>(jb.param + (size_t)j4 * 4);
        mv = *reinterpret_cast<const float4*>(jb.m + (size_t)j4 * 4);
        vv = *reinterpret_cast<const float4*>(jb.v + (size_t)j4 * 4);
    }
    if (jb.nsplit > 0) {
        // 16 slab reads in flight per lane: the walk is a chain of memory latencies (32 dependent-free loads per lane at 256
        // slabs; four at a time took 16 us for 43 MB that sit in the last-level cache), the order of the additions is fixed
        const float* src = jb.slabs + (size_t)j4 * 4;
        int k = live ? g : jb.nsplit;
        for (; k + sl * 15 < jb.nsplit; k += sl * 16) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(k + sl * u) * jb.stride);
#pragma unroll
            for (int u = 0; u < 16; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < jb.nsplit; k += sl) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * jb.stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
            s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
        }
        if (a.sl16) {                                    // (uniform over the launch)
            if (gh == 1 && gl == 0) xw[pair][jl] = s;
            __syncthreads();
            if (gh == 0 && gl == 0) { const float4 o4 = xw[pair][jl]; s.x += o4.x; s.y += o4.y; s.z += o4.z; s.w += o4.w; }
        }
    } else if (g == 0) {
        s = *reinterpret_cast<const float4*>(jb.grad + (size_t)j4 * 4);
    }
    if (g != 0 || !live) return;
    if (jb.nsplit > 0) *reinterpret_cast<float4*>(jb.grad + (size_t)j4 * 4) = s;
    if (a.adam && direct) {
        // A big table's elements that no batch has ever touched have g = m = v = 0, and torch.optim.Adam's update leaves such an
        // element exactly as it is (m' = 0, v' = 0, p' = p - step * (0 / (0 + eps)) = p): 12 bytes read instead of 28 moved -- at
        // config.py:27's NUM_TYPES = 34800 with the benchmark's 100 live types that is 99 % of the 4.45 M table parameters.  (The
        // dense update of rows that HAVE moments but no gradient this step -- p_companion.py:36-43 under train.py:24's Adam --
        // is applied as before.)
        if (pc_adam_dead(s, mv, vv)) return;
        pv = *reinterpret_cast<const float4*>(jb.param + (size_t)j4 * 4);
    }
    if (a.adam) {
        const float step_size = scal[0], bc2s = scal[1];
        const float omb1 = (float)(1.0 - a.beta1), beta2 = (float)a.beta2, omb2 = (float)(1.0 - a.beta2), eps = (float)a.eps;
#define ADAM1(c) pc_adam_update(pv.c, mv.c, vv.c, s.c, step_size, bc2s, omb1, beta2, omb2, eps);
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        *reinterpret_cast<float4*>(jb.param + (size_t)j4 * 4) = pv;
        *reinterpret_cast<float4*>(jb.m + (size_t)j4 * 4) = mv;
        *reinterpret_cast<float4*>(jb.v + (size_t)j4 * 4) = vv;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Large tables, DETERMINISTIC gradients (round 3; float atomics remain the path for more than TG_CAP distinct rows of a table:
// table_partials_kernel's own branch):
// the reference's NUM_TYPES = 34800 (config.py:27) is a big table of which a batch touches few rows (20 live types at the
// reference's catalogue, 100 at the benchmark's).
//   touched_types_kernel   one workgroup per list: bitmap of the destination rows of both lists -> ascending lists ulist_c / ulist_q
//                          and the inverse maps pos_c / pos_q (row -> index in its list)
//   table_partials_kernel  256 workgroups, each a contiguous chunk of the source rows: chunk staged in LDS, rows added IN
//                          SOURCE ORDER into an LDS table indexed by pos (destination u belongs to wave u % 4: one adder
//                          per destination), the table's n_u rows written as the workgroup's slab
//   table_reduce_kernel    table[ulist[u]] = sum of the 256 slabs in fixed order
// No float atomic anywhere: the step is bitwise reproducible at any T.  The ascending lists also ARE the row lists of the
// data-parallel exchange (pc_joint_fused_touched: SURVEY 8e-4, "reduce-scatter for the sparse grads").
#define TG_CAP 512          /* distinct touched rows per table on the deterministic path: 512 x 64 floats = 128 KB of LDS */
#define TG_WGS 256
#define TG_CHUNK 96         /* source rows a workgroup stages at a time (24 KB: a workgroup's whole share of either list at B = 4096, K = 3) */

__global__ __launch_bounds__(1024) void touched_types_kernel(const int32_t* idx_c, int n_c, const int32_t* idx_q, int n_q, int T,
                                                             int32_t* ulist_c, int32_t* pos_c, int32_t* ulist_q,
                                                             int32_t* pos_q, int32_t* n_touch) {
    extern __shared__ unsigned bits[];                  // [words] then scan scratch [1024]
    const int words = (T + 31) >> 5;
    unsigned* part = bits + words;
    {
        const int li = blockIdx.x;                          // one workgroup per list
        const int32_t* idx = li ? idx_q : idx_c;
        const int n = li ? n_q : n_c;
        int32_t* ulist = li ? ulist_q : ulist_c;
        int32_t* pos = li ? pos_q : pos_c;
        __syncthreads();
        for (int i = threadIdx.x; i < words; i += 1024) bits[i] = 0u;
        __syncthreads();
        for (int r0 = threadIdx.x; r0 < n; r0 += 8 * 1024) {     // (eight ids in flight per thread: the loop was a chain of L2 round trips)
            int t[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t[u] = r0 + 1024 * u < n ? idx[r0 + 1024 * u] : -1;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if ((unsigned)t[u] < (unsigned)T) atomicOr(&bits[t[u] >> 5], 1u << (t[u] & 31));      // (integer: the result is the set)
        }
        __syncthreads();
        const int per = (words + 1023) / 1024;
        const int lo = threadIdx.x * per, hi = min(words, lo + per);
        int cnt = 0;
        for (int i = lo; i < hi; i++) cnt += __popc(bits[i]);
        const int incl = block_scan_1024(cnt, part);
        int p = incl - cnt;
        for (int i = lo; i < hi; i++) {
            unsigned m = bits[i];
            while (m) {
                const int bit = __ffs(m) - 1;
                m &= m - 1;
                pos[32 * i + bit] = p;
                ulist[p++] = 32 * i + bit;
            }
        }
        if (threadIdx.x == 1023) n_touch[li] = incl;
    }
}

struct TableList { float* table; const int32_t* idx; const float* src; int rows; const int32_t* pos; float* slabs; };

__global__ __launch_bounds__(256) void table_partials_kernel(TableList l0, TableList l1, const int32_t* n_touch, int T) {
    extern __shared__ __attribute__((aligned(16))) float tg_lds[];
    float* tab = tg_lds;                                 // [TG_CAP][64]
    float* rowsb = tg_lds + TG_CAP * PC_L;               // [TG_CHUNK][64]
    int* dst = reinterpret_cast<int*>(rowsb + TG_CHUNK * PC_L);   // [TG_CHUNK]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nus[2] = {n_touch[0], n_touch[1]};         // (both counts in one round trip: the kernel is a chain of them)
#pragma unroll 1
    for (int li = 0; li < 2; li++) {
        const TableList& l = li ? l1 : l0;
        const int nu = nus[li];
        const int per = (l.rows + TG_WGS - 1) / TG_WGS;
        const int lo = blockIdx.x * per, hi = min(l.rows, lo + per);
        if (nu > TG_CAP) {
            // more distinct rows than the LDS table holds: the row adds go to the (cleared) table by float atomics
            // (eight rows' destinations and values requested together: one row per trip was a chain of dependent L2 round trips --
            // twenty per wave at B = 4096, K = 3 -- and this kernel's whole 30 us)
            for (int r0 = lo + w; r0 < hi; r0 += 32) {
                int d[8];
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int r = r0 + 4 * u;
                    d[u] = r < hi ? l.idx[r] : -1;
                    v[u] = r < hi ? l.src[(size_t)r * PC_L + lane] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if ((unsigned)d[u] < (unsigned)T) unsafeAtomicAdd(l.table + (size_t)d[u] * PC_L + lane, v[u]);
            }
            continue;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nu * PC_L; e += 256) tab[e] = 0.f;
        for (int c0 = lo; c0 < hi; c0 += TG_CHUNK) {
            const int nr = min(TG_CHUNK, hi - c0);
            // the chunk's rows and its destinations are requested together (rows into registers; idx -> pos is two dependent
            // loads): one wait instead of three in a row
            constexpr int RPT = TG_CHUNK * (PC_L / 4) / 256;            // float4 per thread
            float4 rv[RPT];
#pragma unroll
            for (int u = 0; u < RPT; u++) {
                const int e = threadIdx.x + 256 * u;
                rv[u] = e < nr * (PC_L / 4) ? *reinterpret_cast<const float4*>(l.src + (size_t)c0 * PC_L + e * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            int dv = -1;
            if ((int)threadIdx.x < nr) {
                const int d = l.idx[c0 + threadIdx.x];
                dv = (unsigned)d < (unsigned)T ? l.pos[d] : -1;
            }
            __syncthreads();                                         // (the previous chunk's adds have left rowsb / dst)
#pragma unroll
            for (int u = 0; u < RPT; u++) {
                const int e = threadIdx.x + 256 * u;
                if (e < nr * (PC_L / 4)) *reinterpret_cast<float4*>(&rowsb[e * 4]) = rv[u];
            }
            if ((int)threadIdx.x < nr) dst[threadIdx.x] = dv;
            __syncthreads();
            // destination u is wave (u % 4)'s: that wave adds u's rows in source order, nobody else touches tab[u]
            for (int i = 0; i < nr; i++) {
                const int u = dst[i];
                if (u >= 0 && (u & 3) == w) tab[u * PC_L + lane] += rowsb[i * PC_L + lane];
            }
        }
        __syncthreads();
        float* slab = l.slabs + (size_t)blockIdx.x * TG_CAP * PC_L;
        for (int e = threadIdx.x; e < nu * (PC_L / 4); e += 256)
            *reinterpret_cast<float4*>(slab + e * 4) = *reinterpret_cast<const float4*>(&tab[e * 4]);
    }
}

// table[ulist[u]][:] = sum_w slabs[w][u][:], w ascending in eight interleaved lanes then a fixed xor fold (as tn_reduce)
__global__ __launch_bounds__(256) void table_reduce_kernel(TableList l0, TableList l1, const int32_t* ulist0, const int32_t* ulist1,
                                                           const int32_t* n_touch) {
    const int per_list = TG_CAP * (PC_L / 4) * 8 / 256;          // workgroups per list
    const int li = (int)blockIdx.x >= per_list ? 1 : 0;
    const TableList& l = li ? l1 : l0;
    const int32_t* ulist = li ? ulist1 : ulist0;
    const int nu = n_touch[li];
    if (nu > TG_CAP) return;
    const int t = ((int)blockIdx.x - li * per_list) * 256 + threadIdx.x;
    const int j4 = t >> 3, g = t & 7;                     // float4 index inside [TG_CAP][64], slab lane
    const int u = j4 / (PC_L / 4);
    if (u >= nu) return;                                  // (whole 8-lane groups leave together)
    const float* src = l.slabs + (size_t)j4 * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int k = g; k < TG_WGS; k += 8) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * TG_CAP * PC_L);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
        s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
    }
    if (g == 0) *reinterpret_cast<float4*>(l.table + (size_t)ulist[u] * PC_L + (j4 % (PC_L / 4)) * 4) = s;
}

// ---------------------------------------------------------------------------------------------------------------
// Large tables, DETERMINISTIC gradients at ANY number of touched rows (round 5; replaces the LDS-table form above -- 512 distinct
// rows at most -- and its float-atomic branch wherever a list fits the sort kernel's LDS: n <= TS_MAXN source rows, T <= 65535,
// e.g. the reference's NUM_TYPES = 34800 (config.py:27) at B = 4096, K <= 4):
//   table_sort_kernel     TS_NR workgroups per list, each a contiguous range of the table's rows: a COUNTING sort of the source
//                         rows that point into its range, in LDS -- histogram (16-bit counts, two per word, integer atomics), scan
//                         over the range's bins (which also yields its distinct destinations, ascending, and where each one's run
//                         starts), placement through returning atomics.  Out: the order (`sorted`), the run table (one entry per
//                         destination; a run of up to four rows carries its rows), the lists of the runs of 5 .. TS_LONG rows and
//                         of the longer ones.  The placement order inside a run is whatever the atomics gave: every run is SORTED
//                         by source row by whoever consumes it
//   table_segsum_kernel   per destination: its source rows added in ascending source order, written to the (cleared) dense
//                         gradient -- sixteen short runs per workgroup, a wave per medium run, a workgroup per long one (four
//                         contiguous quarters, folded (w0 + w1) + (w2 + w3)); also the dense ascending list of touched rows
//                         (pc_joint_fused_touched)
// The order of every sum is a function of the index lists alone: bitwise reproducible whatever the number of touched rows --
// with DROPOUT = 0.1 (config.py:12) every sample selects its own K types and the complementary table has thousands.
// (Earlier forms, T = 34800, B = 4096, K = 3, both lists: a stable two-pass LSD radix sort with ballot ranking took 160 / 77 us;
// the counting sort as one workgroup per list 52 us, as four range workgroups that each histogrammed everything below their range
// 34 us, of which 60 % was one device atomic per run from the thread that scanned it; this form 12 us.  The consumer: 29 -> 21 us.)
#define TS_MAXN 24576       /* source rows per list (16-bit positions and row numbers) */
#define TS_LONG 64         /* a run of up to this many rows is summed by one wave, a longer one by a workgroup */
#define TS_GIANT 256       /* ... and a run longer than this by several workgroups (256 rows each), the last of which adds their sums */
struct SortList { const int32_t* idx; int n; int32_t* sorted; int4* seg; int2 *medium, *longl, *giant; int32_t *nruns, *gcnt; };
// entries of the giant list: ceil(rows / 256) per run of more than TS_GIANT rows -- at most n / 256 full ones and one more per run
static inline int ts_giant_cap(int n) { return n / 256 + n / (TS_GIANT + 1) + 4; }
// sorted [n]; seg [TS_NR ts_range_stride(T)] = {start of the run, its destination, its rows if it has up to four}, one stretch per
// range of the table's rows (see table_sort_kernel), nruns [TS_NR]: the runs in each; medium [n / 5]: the runs of 5 .. TS_LONG rows,
// longl [n / TS_LONG]: those of up to TS_GIANT ({start << 16 | rows, destination}, in no particular order); giant [ts_giant_cap(n)]:
// the longer ones, one entry per 256 rows ({start << 16 | rows, destination | part << 16}, a run's parts side by side; gcnt
// [first entry of the run]: parts done, zeroed here); n_touch [2 + 2 list], [3 + 2 list], [6 + list]: the lists' counts

#ifdef PC_SORT_TIMING
// developer build (scripts/dev/sort_phase_times.py): shader-clock stamps of every workgroup's thread 0 at the phases of the sort kernel
__device__ unsigned long long pc_sort_timing[1024];
extern "C" int pc_debug_sort_timing(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pc_sort_timing), sizeof(unsigned long long) * 1024);
}
#define PC_ST(i) do { if (threadIdx.x == 0) pc_sort_timing[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// table_segsum_kernel: the longest wave of each grid region (slots 1000 + region), waves with work (1008 + region), rows of the longest run (1016 + region)
#define PC_SEG_T0 const unsigned long long seg_t0 = __builtin_amdgcn_s_memtime()
#define PC_SEG_T1(region, rows) do { if ((threadIdx.x & 63) == 0) { atomicMax(&pc_sort_timing[1000 + (region)], __builtin_amdgcn_s_memtime() - seg_t0); \
        atomicAdd(&pc_sort_timing[1008 + (region)], 1ull); atomicMax(&pc_sort_timing[1016 + (region)], (unsigned long long)(rows)); } } while (0)
// the phases of a medium run of >= 48 rows (slots 900 ..: whichever such wave wrote last)
#define PC_SEG_P(i, rows) do { if ((rows) >= 48 && (threadIdx.x & 63) == 0) pc_sort_timing[900 + (i)] = __builtin_amdgcn_s_memtime() - seg_t0; } while (0)
#else
#define PC_ST(i) do { } while (0)
#define PC_SEG_P(i, rows) do { } while (0)
#define PC_SEG_T0 do { } while (0)
#define PC_SEG_T1(region, rows) do { } while (0)
#endif

typedef __attribute__((address_space(3))) unsigned short ts_l16;
typedef __attribute__((address_space(3))) unsigned ts_l32;
// inclusive prefix sums of a and of b over the 1024 threads of a workgroup and the workgroup's total of c (returned in c): the DPP
// scan inside each wave, the 3 x 16 wave totals through LDS (one barrier: ws is not in use before the call), where sixteen lanes
// of every wave scan them again
__device__ __forceinline__ void block_scan3_1024(int& a, int& b, int& c, ts_l32* ws /* [48] */) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    a = wave_scan_incl(a); b = wave_scan_incl(b); c = wave_scan_incl(c);
    if (lane == 63) { ws[w] = (unsigned)a; ws[16 + w] = (unsigned)b; ws[32 + w] = (unsigned)c; }
    __syncthreads();
    int ta = 0, tb = 0, tc = 0;
    if (lane < 16) { ta = (int)ws[lane]; tb = (int)ws[16 + lane]; tc = (int)ws[32 + lane]; }
    ta = wave_scan_incl(ta); tb = wave_scan_incl(tb); tc = wave_scan_incl(tc);
    // waves in front of this one: lane w - 1 of the inclusive scan
    const int ba = __builtin_amdgcn_readlane(ta, (w + 15) & 15), bb = __builtin_amdgcn_readlane(tb, (w + 15) & 15);
    a += w ? ba : 0; b += w ? bb : 0;
    c = __builtin_amdgcn_readlane(tc, 15);
}

// TS_NR workgroups per list, each owning a contiguous range of the table's rows (bins) and everything that follows from it: the
// counts of ITS bins, ITS stretch of the order (starting at the number of list entries that point below the range: counted
// while the list is read, no histogram needed) and ITS part of the run table (TS_NR stretches of ts_range_stride entries, each
// closed by {end, -1}; nruns[range] = runs in it: what lies in front of a range -- the position of a run in the dense touched-row
// list -- is summed by the consumer from those sixteen counts).  Every workgroup reads the whole list (80 KB, L2) and keeps its
// own entries; nothing passes between workgroups.  The chip clocks near 1 GHz under this step's load (clock64 against the
// kernel's duration): a phase that looks free at 2.4 GHz -- a barrier of sixteen waves, a round of shuffles -- is ~1 us here, so
// the kernel is built from as few of them as it can be.
#ifndef TS_NR
#define TS_NR 16
#endif
static_assert(TS_NR <= 16, "the consumer sums the ranges' run counts over sixteen lanes");
// a run's word in the `lists` counter of table_sort_kernel (see there)
__device__ __forceinline__ int ts_list_word(unsigned c) {
    return c > TS_GIANT ? (int)(((c + 255u) >> 8) << 23) : c > TS_LONG ? 1 << 13 : c > 4u ? 1 : 0;
}
__host__ __device__ inline int ts_range_words(int T) { return (((T + 1) >> 1) + TS_NR - 1) / TS_NR; }
__host__ __device__ inline int ts_range_stride(int T) { return 2 * ts_range_words(T) + 1; }       // runs of a range + its closing entry
__global__ __launch_bounds__(1024) void table_sort_kernel(SortList l0, SortList l1, int T, int32_t* n_touch) {
    extern __shared__ unsigned ts_lds[];
    PC_ST(0);
    const int li = (int)blockIdx.x / TS_NR, rg = (int)blockIdx.x % TS_NR;
    // (the list's fields by value: a reference chosen at run time between two kernel-argument structs puts both in scratch)
    SortList l;
    l.idx = li ? l1.idx : l0.idx; l.n = li ? l1.n : l0.n; l.sorted = li ? l1.sorted : l0.sorted; l.seg = li ? l1.seg : l0.seg;
    l.medium = li ? l1.medium : l0.medium; l.longl = li ? l1.longl : l0.longl; l.nruns = li ? l1.nruns : l0.nruns;
    l.giant = li ? l1.giant : l0.giant; l.gcnt = li ? l1.gcnt : l0.gcnt;
    const int n = l.n;
    const int words = (T + 1) >> 1;                          // two 16-bit bins per word: bin d = half (d & 1) of word d >> 1
    const int rw = ts_range_words(T), rs = ts_range_stride(T);
    const int rlo = min(words, rg * rw), nw = min(words, rlo + rw) - rlo;      // this workgroup's words: [rlo, rlo + nw)
    const int npad = ((n + 1023) >> 10) << 10;
    ts_l32* hist = (ts_l32*)ts_lds;                          // [rw] counts, then running positions
    ts_l16* out = (ts_l16*)(hist + rw);                      // [npad] this range's source rows by destination
    ts_l16* dst16 = out + npad;                              // [2 rw] the range's runs: their bins (relative to the range), ascending
    ts_l32* part = (ts_l32*)(dst16 + 2 * rw);                // [48] scan scratch
    const int tid = threadIdx.x;
    // a thread's list entries (e = tid + 1024 j): requested before anything else -- written by the previous kernel on other
    // XCDs, they come from memory while the bins are cleared -- and kept in registers for the placement below
    constexpr int EPT = TS_MAXN / 1024;
    int dreg[EPT];
#pragma unroll
    for (int j = 0; j < EPT; j++) { const int e = tid + 1024 * j; dreg[j] = e < n ? l.idx[e] : -1; }
    for (int i = tid; i < nw; i += 1024) hist[i] = 0u;
    __syncthreads();
    PC_ST(1);
    // ---- counts of this range's bins; entries that point below the range are counted (their rows come first in the order);
    // from here on dreg holds the bin RELATIVE to the range, -1 for every entry that is not this workgroup's
    int below = 0;
#pragma unroll
    for (int j = 0; j < EPT; j++) {
        const int d = dreg[j], w = (d >> 1) - rlo;
        const bool valid = (unsigned)d < (unsigned)T;
        below += (valid && w < 0) ? 1 : 0;
        const bool mine = valid && (unsigned)w < (unsigned)nw;
        dreg[j] = mine ? d - 2 * rlo : -1;
        if (mine) __hip_atomic_fetch_add(&hist[w], (d & 1) ? 0x10000u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    PC_ST(2);
    // ---- scan over the range's bins: thread t owns words [t wpt, ... + wpt) (one or two at the shipped sizes).
    // packed = rows (low half) | destinations (high half); lists = runs of 5 .. TS_LONG rows (bits 0 .. 12) | of up to TS_GIANT
    // (13 .. 22) | 256-row parts of the longer ones (23 .. 31) -- a list of 24 576 rows has at most 4 915 / 378 / 191 of them
    const int wpt = (nw + 1023) >> 10;
    const int w0 = min(nw, tid * wpt), w1 = min(nw, w0 + wpt);
    int packed = 0, lists = 0;
    for (int i = w0; i < w1; i++) {
        const unsigned c = hist[i], c0 = c & 0xffffu, c1 = c >> 16;
        packed += (int)(c0 + c1) + ((c0 ? 0x10000 : 0) + (c1 ? 0x10000 : 0));
        lists += ts_list_word(c0) + ts_list_word(c1);
    }
    int incl = packed, lincl = lists;
    PC_ST(3);
    block_scan3_1024(incl, lincl, below, part);
    PC_ST(4);
    const int row_base = below;                              // rows in front of this range's
    const int run0 = row_base + ((incl - packed) & 0xffff), pos0 = (incl - packed) >> 16;
    // ---- run lists: runs of more than four rows go on the consumer's lists (its workgroups take the short ones sixteen at a time
    // and would walk a cluster of long ones -- the hot types sit side by side -- one after the other).  The lists are shared by
    // the list's workgroups: ONE returning device atomic per workgroup and list reserves its entries (counters n_touch[2 + 2 list],
    // [3 + 2 list], zeroed by the tile kernel of the step) -- issued here, answered while the placement below runs (one per run
    // from the thread that scans it -- ten dependent round trips to L2 in a row -- was 60 % of an earlier form of this kernel)
    __shared__ int sh_lbase[3], sh_rows, sh_runs;
    int mres = 0, lres = 0, gres = 0;                        // (handed to the workgroup behind the placement: no wait here)
    if (tid == 1023) {
        const int nm = lincl & 0x1fff, nl = (lincl >> 13) & 0x3ff, ng = (int)((unsigned)lincl >> 23), rows = incl & 0xffff, runs = incl >> 16;
        if (nm) mres = atomicAdd(&n_touch[2 + 2 * li], nm);
        if (nl) lres = atomicAdd(&n_touch[3 + 2 * li], nl);
        if (ng) gres = atomicAdd(&n_touch[6 + li], ng);
        sh_rows = rows; sh_runs = runs;
        l.nruns[rg] = runs;
        l.seg[(size_t)rg * rs + runs] = make_int4(row_base + rows, -1, 0, 0);       // closes the range's last run
    }
    {
        int run = run0, pos = pos0;
        for (int i = w0; i < w1; i++) {
            const unsigned c = hist[i], c0 = c & 0xffffu, c1 = c >> 16;
            const unsigned s0 = (unsigned)run, s1 = s0 + c0;
            hist[i] = s0 | (s1 << 16);                       // running positions of the two bins (positions in the whole order)
            // (the run table goes out from LDS further down: written from here, a wave's store scatters over ~50 cache lines)
            if (c0) { dst16[pos] = (unsigned short)(2 * i); pos++; }
            if (c1) { dst16[pos] = (unsigned short)(2 * i + 1); pos++; }
            run += (int)(c0 + c1);
        }
    }
    PC_ST(5);
    __syncthreads();
    PC_ST(6);
    const int nrows = sh_rows;
    PC_ST(7);
    // ---- placement: the returning atomic hands every row of this range a slot of its destination's run (in no particular order)
#pragma unroll
    for (int j = 0; j < EPT; j++)
        if (dreg[j] >= 0) {
            const unsigned old = __hip_atomic_fetch_add(&hist[dreg[j] >> 1], (dreg[j] & 1) ? 0x10000u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            out[((dreg[j] & 1) ? old >> 16 : old & 0xffffu) - (unsigned)row_base] = (unsigned short)(tid + 1024 * j);
        }
    if (tid == 1023) { sh_lbase[0] = mres; sh_lbase[1] = lres; sh_lbase[2] = gres; }
    __syncthreads();
    PC_ST(8);
    // ---- the run table of this range, dense (a wave's store: sixteen cache lines): {start of the run, destination, its rows if
    // it has up to four (16 bits each, in placement order) -- the consumer of a short run needs nothing else}.  The placement left
    // every bin's END in its half word: a run starts where the one before it ends.
    for (int q = tid; q < sh_runs; q += 1024) {
        const unsigned d = dst16[q], h = hist[d >> 1];
        const unsigned end = (d & 1u) ? h >> 16 : h & 0xffffu;
        unsigned start = (unsigned)row_base;
        if (q) { const unsigned dp = dst16[q - 1], hp = hist[dp >> 1]; start = (dp & 1u) ? hp >> 16 : hp & 0xffffu; }
        const unsigned m = end - start, o = start - (unsigned)row_base;
        unsigned r01 = 0u, r23 = 0u;
        if (m <= 4u) {
            r01 = (unsigned)out[o] | (m > 1u ? (unsigned)out[o + 1] << 16 : 0u);
            r23 = (m > 2u ? (unsigned)out[o + 2] : 0u) | (m > 3u ? (unsigned)out[o + 3] << 16 : 0u);
        }
        l.seg[(size_t)rg * rs + q] = make_int4((int)start, 2 * rlo + (int)d, (int)r01, (int)r23);
    }
    // ---- the run lists' entries, self-contained ({start << 16 | rows, destination}: their consumers do not read the run table):
    // a second walk over the thread's words has the lengths again
    {
        const unsigned excl = (unsigned)(lincl - lists);
        int mpos = sh_lbase[0] + (int)(excl & 0x1fffu), lpos = sh_lbase[1] + (int)((excl >> 13) & 0x3ffu), gpos = sh_lbase[2] + (int)(excl >> 23);
        unsigned prev = (unsigned)run0;
        for (int i = w0; i < w1; i++) {
            const unsigned h = hist[i], e0 = h & 0xffffu, e1 = h >> 16;
            const unsigned cc[2] = {e0 - prev, e1 - e0}, ss[2] = {prev, e0};
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int2 ent = make_int2((int)((ss[u] << 16) | cc[u]), 2 * (rlo + i) + u);
                if (cc[u] > TS_GIANT) {                       // one entry per 256 rows of the run (ascending row order), side by side
                    l.gcnt[gpos] = 0;
                    for (unsigned pt = 0; pt < (cc[u] + 255u) >> 8; pt++) l.giant[gpos++] = make_int2(ent.x, ent.y | (int)(pt << 16));
                } else if (cc[u] > TS_LONG) l.longl[lpos++] = ent;
                else if (cc[u] > 4u) l.medium[mpos++] = ent;
            }
            prev = e1;
        }
    }
    PC_ST(9);
    // ---- the order out to memory, every run in placement order: its consumer sorts it (the rows are distinct numbers below n:
    // a bitmap in LDS, read back in ascending order)
    for (int i = tid; i < nrows; i += 1024) l.sorted[row_base + i] = (int)out[i];
    PC_ST(10);
}
// LDS: [a range's bins / 2] histogram | [2 npad] order | [a range's bins] 16-bit run destinations | scan scratch
static size_t table_sort_lds_bytes(int n, int T) {
    const size_t npad = (size_t)((n + 1023) >> 10) << 10;
    return ((size_t)2 * ts_range_words(T) + npad / 2 + 48) * 4;
}
static bool table_sort_fits(int n, int T) { return T <= 65535 && n <= TS_MAXN && table_sort_lds_bytes(n, T) <= 160 * 1024; }

// Up to 64 rows of src whose numbers stand in a wave's queue in LDS (q[0 .. mm), 16-bit, 16-byte aligned), added as float4: the
// wave's four 16-lane groups take sixteen consecutive rows each (group g: rows 16 g .. 16 g + 15 -- its 32 bytes of the queue
// are two LDS reads; handing the numbers round by lane shuffles, sixteen dependent LDS operations per block, cost more than the
// rows' own round trip), lane j of a group floats [4 j, 4 j + 4) of the row: sixteen 16-byte loads -- 64 rows -- in flight per
// lane.  The order is a function of the positions alone: every group ascending, the groups folded (g0 + g1) + (g2 + g3) by
// ts_fold_groups; ts_add_queue keeps two such blocks -- 32 loads -- in flight before the first add.
__device__ __forceinline__ void ts_load64(float4 (&v)[16], const float* src, const ts_l16* q, int mm, int lane) {
    const int g = lane >> 4, j = lane & 15;
    typedef unsigned ts_u4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const ts_u4 ts_l128;
    const ts_u4 lo = *reinterpret_cast<ts_l128*>(q + 16 * g), hi = *reinterpret_cast<ts_l128*>(q + 16 * g + 8);
    const unsigned wd[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const unsigned r = (wd[u >> 1] >> (16 * (u & 1))) & 0xffffu;
        v[u] = 16 * g + u < mm ? *reinterpret_cast<const float4*>(src + (size_t)r * PC_L + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ void ts_acc64(float4& acc, const float4 (&v)[16]) {
#pragma unroll
    for (int u = 0; u < 16; u++) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
}
__device__ __forceinline__ float4 ts_fold_groups(float4 a) {
    a.x += __shfl_xor(a.x, 16, 64); a.y += __shfl_xor(a.y, 16, 64); a.z += __shfl_xor(a.z, 16, 64); a.w += __shfl_xor(a.w, 16, 64);
    a.x += __shfl_xor(a.x, 32, 64); a.y += __shfl_xor(a.y, 32, 64); a.z += __shfl_xor(a.z, 32, 64); a.w += __shfl_xor(a.w, 32, 64);
    return a;
}
// rows qu[0 .. m) (a wave's queue in LDS, ascending; readable up to the next multiple of 64) of src added in that order, 128 at a
// time.  Every block of 64 has its own four group sums: folded into `acc` block by block, in order.
__device__ __forceinline__ void ts_add_queue(float4& acc, const float* src, const ts_l16* qu, int m, int lane) {
    for (int c = 0; c < m; c += 128) {
        const int mm = min(128, m - c);                      // wave-uniform
        float4 v0[16], v1[16];
        ts_load64(v0, src, qu + c, min(mm, 64), lane);
        if (mm > 64) ts_load64(v1, src, qu + c + 64, mm - 64, lane);
        ts_acc64(acc, v0);
        if (mm > 64) ts_acc64(acc, v1);
    }
}
// By the length m of a destination's run (its rows stand in `sorted` in the order the placement's atomics gave them):
//   m <= 4              (most of them: a type a few samples selected) sixteen runs per workgroup, one 16-lane group per run, four
//                       runs per wave instruction: the run's rows -- they came with the run table's entry -- sorted over four
//                       lanes, added as float4 per lane
//   m <= TS_LONG        (the list `medium`) one wave per run
//   m <= TS_GIANT       (the list `longl`) one workgroup per run, each wave 64 rows of its ascending order, the four sums folded
//                       (w0 + w1) + (w2 + w3)
//   m > TS_GIANT        (the list `giant`: a destination a good part of the batch points at) one workgroup per 256 rows of the run;
//                       a CU keeps only so many cache lines in flight (a row is two of them, ~16 clocks per row and CU at this
//                       step's memory latency: 1 900 rows by one workgroup were 15 us of this kernel's 20), eight of them do.  The
//                       64-row sums wait in memory; the workgroup that finishes last adds them in ascending order
// Medium and long runs are sorted through a bitmap: the rows are distinct numbers below n (24 576 at most: 3 KB of LDS), every
// row sets its bit, the bits read back in ascending order ARE the sorted run (~1 500 clocks whatever the length -- a bitonic
// network over the wave's registers needs 21 dependent shuffles for 64 rows and 144 for 256, one over LDS 40 000 clocks for 2 048).
// The chip clocks near 1 GHz in this step and a load that misses L2 (everything here was written by other XCDs a kernel ago)
// takes ~2 500 clocks: the paths are counted in such round trips -- short: run table, rows; medium / long: list entry + count,
// row numbers, rows (128 in flight per wave).
// Grid regions per list: [short][medium][long], the two lists one after the other (SegGrid).  The short-run workgroups also write
// the destination list (ulist: the touched rows of pc_joint_fused_touched, ascending).
struct SegList { float* table; const float* src; const int32_t* sorted; const int4* seg; int32_t* ulist; const int2 *medium, *longl;
                 const int32_t* nruns; int n; const int2* giant; int32_t* gcnt; float* gprt; };
// first workgroup of: short 0, medium 0, long 0, giant 0, short 1, medium 1, long 1, giant 1; total.  spr: short-run workgroups per
// range of the table (sixteen runs each), rs: ts_range_stride(T)
struct SegGrid { int start[9]; int spr[2]; int rs; };
#define TS_SPR 32          /* grid: short-run workgroups per range of the table and list (16 runs each; the workgroup walks on) */
#define TS_GMED 128        /* ... workgroups of four medium runs per list */
#define TS_GLONG 128       /* ... workgroups (long runs) per list */
#define TS_GGIANT 96        /* ... workgroups (256-row parts of giant runs) per list */
#define TS_QCAP 64         /* a wave's queue of row numbers */
__global__ __launch_bounds__(256) void table_segsum_kernel(SegList l0, SegList l1, int32_t* n_touch, SegGrid gr) {
    __shared__ float fold[4][PC_L];
    __shared__ unsigned bits[4 * (TS_MAXN / 32)];            // medium: one bitmap per wave; long: the first one, shared
    __shared__ __attribute__((aligned(16))) unsigned short queue[4 * TS_QCAP];
    __shared__ int sh_b[17], sh_d[16], sh_r[16][2];
    PC_SEG_T0;
    const int bid = blockIdx.x;
    int region = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) region += bid >= gr.start[i] ? 1 : 0;
    const int li = region >= 4 ? 1 : 0, kind = region - 4 * li, blk = bid - gr.start[region];
    const SegList& l = li ? l1 : l0;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (kind == 0) {
        const int spr = li ? gr.spr[1] : gr.spr[0];
        const int rg = blk / spr;
        int s0 = (blk - rg * spr) * 16;                           // runs [s0, s0 + 16) of range rg, then spr blocks further on, ...
        // (the run table is requested together with the counts it is checked against: one round trip instead of two)
        int4 e = make_int4(0, -1, 0, 0);
        if (tid < 17) e = l.seg[(size_t)rg * gr.rs + s0 + tid];   // (seg ends with 16 spare entries)
        const int nu = l.nruns[rg];
        int in_front = 0;
        if (w == 0 && (s0 < nu || blk == 0)) {
            // the ranges in front of this one: where its runs stand in the dense list of touched rows; the list's first
            // workgroup also files the total
            const int c = wave_scan_incl(lane < TS_NR ? l.nruns[lane] : 0);
            in_front = rg ? __builtin_amdgcn_readlane(c, (rg + 15) & 15) : 0;
            if (blk == 0 && lane == 15) n_touch[li] = c;
        }
        for (;;) {
        if (s0 >= nu) return;                                // workgroup-uniform
        if (tid < 17) {
            const bool live = s0 + tid <= nu;                // entry nu closes the last run
            sh_b[tid] = live ? e.x : 0;
            if (tid < 16) {
                sh_d[tid] = (live && s0 + tid < nu) ? e.y : -1;
                sh_r[tid][0] = e.z; sh_r[tid][1] = e.w;
                if (s0 + tid < nu) l.ulist[in_front + s0 + tid] = e.y;
            }
        }
        __syncthreads();
        // group g = tid >> 4, lane j of the group owns floats [4 j, 4 j + 4) of the row
        const int g = tid >> 4, j = tid & 15;
        const int dest = sh_d[g];
        const int m = dest >= 0 ? sh_b[g + 1] - sh_b[g] : 0;
        if (m >= 1 && m <= 4) {
            // sorted over lanes 0..3 of the group (lanes 4..15 hold +infinity)
            int x = j < m ? (int)(((unsigned)sh_r[g][(j >> 1) & 1] >> (16 * (j & 1))) & 0xffffu) : 0x7fffffff;
            {
                int o = __shfl_xor(x, 1, 64);
                bool up = (j & 2) == 0, lower = (j & 1) == 0;
                x = (lower == up) ? min(x, o) : max(x, o);
                o = __shfl_xor(x, 2, 64);
                lower = (j & 2) == 0;
                x = lower ? min(x, o) : max(x, o);
                o = __shfl_xor(x, 1, 64);
                lower = (j & 1) == 0;
                x = lower ? min(x, o) : max(x, o);
            }
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int r = __shfl(x, (lane & ~15) + i, 64);
                v[i] = i < m ? *reinterpret_cast<const float4*>(l.src + (size_t)r * PC_L + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float4 acc = v[0];
#pragma unroll
            for (int i = 1; i < 4; i++) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }      // ascending source row
            *reinterpret_cast<float4*>(l.table + (size_t)dest * PC_L + 4 * j) = acc;
        }
        PC_SEG_T1(region, 4);
        // (a range with more than 16 spr runs: rare -- the grid gives every range min(its capacity, TS_SPR) blocks)
        s0 += 16 * spr;
        if (s0 >= nu) return;
        __syncthreads();
        if (tid < 17) e = l.seg[(size_t)rg * gr.rs + s0 + tid];
        }
    }
    const int nwords = (l.n + 31) >> 5;                      // words of a bitmap over the list's row numbers
    if (kind == 1) {                                         // one wave per medium run (5 .. 64 rows)
        int h = blk * 4 + w;
        int2 ent = l.medium[h];                              // (the list is as long as this region has waves: requested with the count)
        const int nmed = n_touch[2 + 2 * li], hstep = 4 * (gr.start[region + 1] - gr.start[region]);
        for (; h < nmed; h += hstep, ent = l.medium[min(h, nmed - 1)]) {       // wave-uniform
        const int start = __builtin_amdgcn_readfirstlane((int)((unsigned)ent.x >> 16)), m = __builtin_amdgcn_readfirstlane(ent.x & 0xffff);
        const int dest = __builtin_amdgcn_readfirstlane(ent.y);
        PC_SEG_P(0, m);
        const int row = lane < m ? l.sorted[start + lane] : -1;
        ts_l32* bm = (ts_l32*)bits + w * (TS_MAXN / 32);
        ts_l16* qu = (ts_l16*)queue + w * TS_QCAP;
        for (int i = lane; i < nwords; i += 64) bm[i] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (row >= 0) __hip_atomic_fetch_or(&bm[row >> 5], 1u << (row & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_wave_barrier();
        PC_SEG_P(1, m);
        // lane i reads words [i wpl, (i + 1) wpl): ascending from lane to lane, so one scan of the lanes' counts places everything
        {
            constexpr int WPL = TS_MAXN / 32 / 64;
            const int wpl = (nwords + 63) >> 6;
            unsigned vv[WPL];
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < WPL; u++) {
                const int i = lane * wpl + u;
                vv[u] = (u < wpl && i < nwords) ? (unsigned)bm[i] : 0u;
                cnt += __popc(vv[u]);
            }
            int o = wave_scan_incl(cnt) - cnt;               // exclusive prefix: where this lane's rows go
#pragma unroll
            for (int u = 0; u < WPL; u++) {
                unsigned v = vv[u];
                const int base = 32 * (lane * wpl + u);
                while (v) {
                    qu[o++] = (unsigned short)(base + __ffs((int)v) - 1);
                    v &= v - 1u;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        PC_SEG_P(2, m);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        ts_add_queue(acc, l.src, qu, m, lane);
        PC_SEG_P(3, m);
        const float4 r = ts_fold_groups(acc);
        if (lane < 16) *reinterpret_cast<float4*>(l.table + (size_t)dest * PC_L + 4 * lane) = r;
        PC_SEG_T1(region, m);
        __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    // ---- long (65 .. TS_GIANT rows: the whole run) and giant runs (a 256-row part of one): a workgroup each.  All its threads
    // fill ONE bitmap with the run's rows and read it back, thread t words [t wpt, (t + 1) wpt), ascending from thread to thread;
    // the rows of ranks [64 w, 64 w + 64) (of the part) go to wave w's queue and are added by it -- one batch of loads.
    const bool giant = kind == 3;
    int2 ent = giant ? l.giant[blk] : l.longl[blk];
    const int nent = giant ? n_touch[6 + li] : n_touch[3 + 2 * li], estep = gr.start[region + 1] - gr.start[region];
    for (int hb = blk; hb < nent; hb += estep, ent = giant ? l.giant[min(hb, nent - 1)] : l.longl[min(hb, nent - 1)]) {
        const int start = __builtin_amdgcn_readfirstlane((int)((unsigned)ent.x >> 16)), m = __builtin_amdgcn_readfirstlane(ent.x & 0xffff);
        const int dest = __builtin_amdgcn_readfirstlane(ent.y & 0xffff), pt = __builtin_amdgcn_readfirstlane((int)((unsigned)ent.y >> 16));
        ts_l32* bm = (ts_l32*)bits;                          // one bitmap for the workgroup
        // the run's row numbers, eight per thread and round (requested before the bitmap is cleared)
        int rows[8];
#pragma unroll
        for (int u = 0; u < 8; u++) rows[u] = tid + 256 * u < m ? l.sorted[start + tid + 256 * u] : -1;
        for (int i = tid; i < nwords; i += 256) bm[i] = 0u;
        __syncthreads();
        for (int i0 = 0; i0 < m; i0 += 8 * 256) {
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (rows[u] >= 0) __hip_atomic_fetch_or(&bm[rows[u] >> 5], 1u << (rows[u] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (i0 + 8 * 256 < m) {
#pragma unroll
                for (int u = 0; u < 8; u++) { const int i = i0 + 8 * 256 + tid + 256 * u; rows[u] = i < m ? l.sorted[start + i] : -1; }
            }
        }
        __syncthreads();
        constexpr int WPT = TS_MAXN / 32 / 256;
        const int wpt = (nwords + 255) >> 8;
        unsigned vv[WPT];
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < WPT; u++) {
            const int i = tid * wpt + u;
            vv[u] = (u < wpt && i < nwords) ? (unsigned)bm[i] : 0u;
            cnt += __popc(vv[u]);
        }
        const int incl = wave_scan_incl(cnt);
        if (lane == 63) sh_b[w] = incl;
        __syncthreads();
        int o = incl - cnt - 256 * pt;                       // rank of this thread's first row, relative to the part
#pragma unroll
        for (int k = 0; k < 3; k++) o += k < w ? sh_b[k] : 0;
        ts_l16* qa = (ts_l16*)queue;                         // [4][64]: wave w's queue = ranks [64 w, 64 w + 64)
        if (o < 256 && o + cnt > 0) {
#pragma unroll
            for (int u = 0; u < WPT; u++) {
                unsigned v = vv[u];
                const int base = 32 * (tid * wpt + u);
                while (v) {
                    if ((unsigned)o < 256u) qa[o] = (unsigned short)(base + __ffs((int)v) - 1);
                    o++;
                    v &= v - 1u;
                }
            }
        }
        __syncthreads();
        const int mine = max(0, min(64, m - 256 * pt - 64 * w));      // rows of this wave
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        ts_add_queue(acc, l.src, qa + 64 * w, mine, lane);
        const float4 part = ts_fold_groups(acc);
        if (!giant) {
            if (lane < 16) *reinterpret_cast<float4*>(&fold[w][4 * lane]) = part;
            __syncthreads();
            if (w == 0) l.table[(size_t)dest * PC_L + lane] = (fold[0][lane] + fold[1][lane]) + (fold[2][lane] + fold[3][lane]);
            PC_SEG_T1(region, m);
            __syncthreads();
            continue;
        }
        // giant: the 64-row sums of all the run's parts wait in gprt ([4 x first entry of the run + 64-row block]); the workgroup
        // that finishes LAST (a device counter per run, zeroed by table_sort_kernel) adds them in ascending order.  The hand-off
        // across XCDs (private L2s): every storing wave drains its stores, one lane releases at agent scope (writes the XCD's dirty
        // lines back) and waits for that before it takes its ticket; the last arriver acquires at agent scope (drops stale lines)
        // and reads with plain vector loads
        const int e0 = hb - pt, nblk = (m + 63) >> 6, nparts = (m + 255) >> 8;
        if (mine && lane < 16) *reinterpret_cast<float4*>(l.gprt + ((size_t)4 * e0 + 4 * pt + w) * PC_L + 4 * lane) = part;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the fence's own wait is not to be relied on)
            sh_b[4] = __hip_atomic_fetch_add(&l.gcnt[e0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (sh_b[4] == nparts - 1 && w == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            float acc1 = 0.f;
            for (int k0 = 0; k0 < nblk; k0 += 16) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; u++) v[u] = k0 + u < nblk ? l.gprt[((size_t)4 * e0 + k0 + u) * PC_L + lane] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; u++) acc1 += v[u];
            }
            l.table[(size_t)dest * PC_L + lane] = acc1;
        }
        PC_SEG_T1(region, m);
        __syncthreads();
    }
}

int pc_opt_sorted_tables();     // (gemm_tn.hip: pc_set_option)

struct FusedWs {
    float *part, *h, *dpi, *dtp, *dc, *dh, *dt, *ecsrc;
    int32_t *ecidx, *cids, *ulist, *n_u, *topk_by_type;
    int32_t *tl_c, *tp_c, *tl_q, *tp_q, *n_touch;       // touched rows of the two big tables: ascending lists, row -> list position
    int32_t *srt_c, *srt_q; int4 *seg_c, *seg_q;        // sort path: source rows ordered by destination, the run table (see SortList)
    int2 *med_c, *med_q, *lng_c, *lng_q;                // ... the runs of 5 .. TS_LONG rows and the longer ones
    int2 *gnt_c, *gnt_q; int32_t *gct_c, *gct_q; float *gpr_c, *gpr_q;      // ... the giant ones' parts, counters and 64-row sums
    bool sorted_path;
    float *tslab_c, *tslab_q;                           // [TG_WGS][TG_CAP][64] each
    float* part_val;
    float *csamp, *gmat, *g0, *gnmax;                           // large tables with hidden-layer dropout: hd [B][32], G = E_c dec_w [T][32], g0 = E_c dec_b [T]
    float* wslabs; int wg_blocks, wslab_floats;      // joint_wgrad_kernel: one slab per workgroup
    int nchunks_s, ucap;               // chunks of the similarity kernels; capacity of the distinct-query-type list
    bool small;
    size_t total;
};

static FusedWs fused_ws_layout(void* base, int B, int T, int K) {
    FusedWs w;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? reinterpret_cast<char*>(base) + off : nullptr;
        off += align256(bytes);
        return p;
    };
    w.small = T <= T_SMALL;
    w.part = (float*)take((size_t)2 * B * 4);
    w.h = (float*)take((size_t)B * LH * 4);
    w.dpi = (float*)take((size_t)B * PC_D * 4);
    w.dtp = (float*)take((size_t)B * K * PC_D * 4);
    w.dc = (float*)take((size_t)B * PC_L * 4);
    w.dh = (float*)take((size_t)B * LH * 4);
    w.dt = (float*)take((size_t)B * PC_L * 4);
    w.ecsrc = (float*)take((size_t)B * (K + 2) * PC_L * 4);
    w.ecidx = (int32_t*)take((size_t)B * (K + 2) * 4);
    w.cids = (int32_t*)take((size_t)2 * B * 4);
    w.wg_blocks = (B + WG_S - 1) / WG_S;
    w.wslab_floats = wg_slab_floats(w.small ? T : 0);      // (large tables: their gradients go by row scatter-add)
    w.wslabs = (float*)take((size_t)w.wg_blocks * w.wslab_floats * 4);
    w.nchunks_s = w.ucap = 0;
    w.ulist = w.n_u = w.topk_by_type = nullptr;
    w.tl_c = w.tp_c = w.tl_q = w.tp_q = w.n_touch = nullptr;
    w.tslab_c = w.tslab_q = nullptr;
    w.srt_c = w.srt_q = nullptr;
    w.med_c = w.med_q = w.lng_c = w.lng_q = nullptr;
    w.gnt_c = w.gnt_q = nullptr; w.gct_c = w.gct_q = nullptr; w.gpr_c = w.gpr_q = nullptr;
    w.seg_c = w.seg_q = nullptr;
    w.sorted_path = false;
    w.part_val = nullptr;
    w.csamp = w.gmat = w.g0 = w.gnmax = nullptr;
    if (w.small) {
    } else {
        const int nc = B * (K + 2);
        w.tl_c = (int32_t*)take((size_t)(nc < T ? nc : T) * 4);
        w.tp_c = (int32_t*)take((size_t)T * 4);
        w.tl_q = (int32_t*)take((size_t)(B < T ? B : T) * 4);
        w.tp_q = (int32_t*)take((size_t)T * 4);
        w.n_touch = (int32_t*)take(256);
        w.sorted_path = table_sort_fits(nc, T);
        if (w.sorted_path) {
            w.srt_c = (int32_t*)take((size_t)nc * 4);
            w.srt_q = (int32_t*)take((size_t)B * 4);
            w.seg_c = (int4*)take((size_t)(TS_NR * ts_range_stride(T) + 16) * 16);     // (+ 16: a workgroup of the consumer requests 17 entries at once)
            w.seg_q = (int4*)take((size_t)(TS_NR * ts_range_stride(T) + 16) * 16);
            // (the run lists: one entry per wave / workgroup of the consumer's grid regions -- read before the count is known)
            w.med_c = (int2*)take((size_t)(nc / 5 + 8) * 8);
            w.med_q = (int2*)take((size_t)(B / 5 + 8) * 8);
            w.lng_c = (int2*)take((size_t)(nc / TS_LONG + 1) * 8);
            w.lng_q = (int2*)take((size_t)(B / TS_LONG + 1) * 8);
            w.gnt_c = (int2*)take((size_t)ts_giant_cap(nc) * 8); w.gnt_q = (int2*)take((size_t)ts_giant_cap(B) * 8);
            w.gct_c = (int32_t*)take((size_t)ts_giant_cap(nc) * 4); w.gct_q = (int32_t*)take((size_t)ts_giant_cap(B) * 4);
            w.gpr_c = (float*)take((size_t)ts_giant_cap(nc) * 4 * PC_L * 4); w.gpr_q = (float*)take((size_t)ts_giant_cap(B) * 4 * PC_L * 4);
        }
        w.tslab_c = (float*)take((size_t)TG_WGS * TG_CAP * PC_L * 4);      // (the LDS-table form: PC_OPT_SORTED_TABLE_GRADIENTS, below)
        w.tslab_q = (float*)take((size_t)TG_WGS * TG_CAP * PC_L * 4);
        w.nchunks_s = (T + PC_STC - 1) / PC_STC;
        w.ucap = B < T ? B : T;
        w.ulist = (int32_t*)take((size_t)2 * w.ucap * 4);          // two lists: this step's, and the next one's formed ahead (JointLookahead)
        w.n_u = (int32_t*)take(256);                                // n_u[0], n_u[1]
        // (sized for either regime: rows = distinct query types without dropout, = the B samples with it)
        w.topk_by_type = (int32_t*)take((size_t)(T > B ? T : B) * K * 4);
        w.part_val = (float*)take((size_t)B * 4 * w.nchunks_s * 4);   // cmax [rows <= B][4 nchunks_s]: the sub-chunk maxima of sample_sims_max_kernel
        w.csamp = (float*)take((size_t)B * LH * 4);                // hd: the dropped hidden rows
        w.gmat = (float*)take((size_t)T * LH * 4);
        w.g0 = (float*)take((size_t)T * 4);
        w.gnmax = (float*)take((size_t)2 * ((T + UT - 1) / UT) * 4);     // the sub-chunks' largest |G[t]|, then their largest |g0[t]|
    }
    w.total = off;
    return w;
}

extern "C" size_t pc_joint_fused_workspace_bytes(int batch, int num_types, int k) {
    if (batch <= 0 || num_types <= 0 || k <= 0) return 0;
    return fused_ws_layout(nullptr, batch, num_types, k).total;
}

// The touched rows of the two [T,64] tables after a fused step at T > 512 (pointers INTO the workspace; valid until the next
// step on it): ascending distinct row ids and their counts (device int32[2]: complementary table, query table).  This is the
// row list a data-parallel job exchanges instead of the dense tables.
extern "C" int pc_joint_fused_touched(void* ws, size_t ws_bytes, int batch, int num_types, int k, const int32_t** rows_comp,
                                      const int32_t** rows_query, const int32_t** n_touched) {
    if (!ws || batch <= 0 || num_types <= 0 || k <= 0 || !rows_comp || !rows_query || !n_touched) return PC_EINVAL;
    if (ws_bytes < pc_joint_fused_workspace_bytes(batch, num_types, k)) return PC_EWORKSPACE;
    FusedWs w = fused_ws_layout(ws, batch, num_types, k);
    if (w.small) return PC_ESHAPE;                       // T <= 512: the tables travel densely (51 KB at T = 100)
    *rows_comp = w.tl_c; *rows_query = w.tl_q; *n_touched = w.n_touch;
    return PC_OK;
}

extern "C" int pc_joint_fused_supported(int num_types, int k, float dropout_p) {
    if (k < 1 || k > FK || k > num_types || num_types < 1) return 0;
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return 0;
    // (T > T_SMALL with dropout: the similarity row per SAMPLE instead of per distinct query type, round 4)
    if (num_types > T_SMALL && (size_t)((num_types + 31) / 32 + 1024) * 4 > 160 * 1024) return 0;
    return 1;
}

static bool tensors_ok(const pc_joint_tensors* t, bool need_table) {
    return t && (!need_table || t->product_table) && t->enc_w && t->enc_b && t->dec_w && t->dec_b && t->typ_w && t->typ_b &&
           t->itm_w && t->itm_b && t->query_types && t->comp_types;
}

extern "C" int pc_build_complementary_batch(const int32_t* pairs, int batch, const float* features,
                                            const int32_t* type_idx, int n_types, uint64_t seed, uint64_t step,
                                            int32_t* query_idx, int32_t* query_types, int32_t* pos_types,
                                            int32_t* neg_types, float* pos_items, float* neg_items,
                                            float* target_features, void* stream);

struct PairsSrc { const int32_t* pairs; const float* features; const int32_t* type_idx; int n_types; uint64_t seed, step; };
// Across the steps of ONE epoch call (num_types > 512, no hidden-layer dropout): `have` -- the previous step of this call formed
// this step's distinct-query-type list into half `parity` of the double buffer; next_pairs -- the labelled pairs of the next step
// (same batch size: the workspace layout, hence the buffer, is the same), whose list this step forms into the other half.
struct JointLookahead { bool have; int parity; const int32_t* next_pairs; };

static int fused_step_impl(const pc_joint_tensors* p, const pc_joint_tensors* g, const pc_joint_tensors* exp_avg,
                           const pc_joint_tensors* exp_avg_sq, int64_t* step_count, double lr, double beta1,
                           double beta2, double eps, const PairsSrc* src, const int32_t* query_idx, const int32_t* query_types,
                           const int32_t* pos_types, const int32_t* neg_types, const float* pos_items,
                           const float* neg_items, int B, int T, int K, int num_products, float margin, float alpha,
                           float* losses, int32_t* topk, int32_t* bad_count, void* ws, size_t ws_bytes,
                           void* stream, JointLookahead* la = nullptr) {
    if (!tensors_ok(p, true) || !tensors_ok(g, false)) return PC_EINVAL;
    const bool adam = exp_avg != nullptr;
    if (adam && (!tensors_ok(exp_avg, false) || !tensors_ok(exp_avg_sq, false) || !step_count)) return PC_EINVAL;
    if (!query_idx || !query_types || !pos_types || !neg_types || !pos_items || !neg_items || !losses || !topk || !ws)
        return PC_EINVAL;
    if (B <= 0 || T <= 0 || num_products <= 0) return PC_EINVAL;
    if (!pc_joint_fused_supported(T, K, p->dropout.p)) return PC_ESHAPE;
    if (ws_bytes < pc_joint_fused_workspace_bytes(B, T, K)) return PC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    FusedWs w = fused_ws_layout(ws, B, T, K);
    // the batch from labelled pairs: inside the tile kernel (T <= 128, and T > 512 where present_types_kernel takes the query
    // types from the pairs as well); for 128 < T <= 512 by the builder's own launch, then the step as usual
    const bool pairs_in_tile = src && (!w.small || T <= T_WGRAD);
    if (src && !pairs_in_tile)
        PC_TRY(pc_build_complementary_batch(src->pairs, B, src->features, src->type_idx, src->n_types, src->seed, src->step,
                                            const_cast<int32_t*>(query_idx), const_cast<int32_t*>(query_types),
                                            const_cast<int32_t*>(pos_types), const_cast<int32_t*>(neg_types),
                                            const_cast<float*>(pos_items), const_cast<float*>(neg_items), nullptr, stream));

    const bool per_sample = !w.small && p->dropout.p > 0.f;      // hidden-layer dropout: c is a function of the SAMPLE
    const int32_t* look_pairs = nullptr; int32_t *look_ulist = nullptr, *look_n_u = nullptr;      // JointLookahead: see joint_finish_kernel
    if (!w.small) {
        // rows of the similarity product: the B samples (dropout), else the U distinct query types of the batch -- U is a device
        // scalar, the launches are sized for its capacity min(B, T) and the workgroups past U leave at once
        const int rows_cap = per_sample ? B : w.ucap;
        // the look-ahead serves the path whose query types come from the pairs (the epoch calls), lists that fit the riding
        // workgroup's LDS bitmap
        const bool la_ok = la && !per_sample && pairs_in_tile && T <= PRESENT256_MAX_T;
        const int par = la_ok ? (la->parity & 1) : 0;
        int32_t* ulist_w = per_sample ? nullptr : w.ulist + (size_t)par * w.ucap;
        int32_t* n_rows_w = per_sample ? nullptr : w.n_u + par;
        const int32_t* ulist = ulist_w;
        const int32_t* n_rows = n_rows_w;
        if (!per_sample && !(la_ok && la->have)) {
            const int words = (T + 31) / 32;
            PC_LAUNCH(present_types_kernel, dim3(1), dim3(1024), (size_t)(words + 1024) * 4, st, query_types, B, T, ulist_w, n_rows_w,
                      pairs_in_tile ? src->pairs : nullptr, pairs_in_tile ? src->type_idx : nullptr, num_products);
        }
        SampleHArgs ca = {p->enc_w, p->enc_b, p->dec_w, p->dec_b, p->query_types, p->comp_types, query_types,
                          pairs_in_tile ? src->pairs : nullptr, pairs_in_tile ? src->type_idx : nullptr, B, T, num_products,
                          (rows_cap + UT - 1) / UT, make_dropcfg(p->dropout), w.csamp, w.gmat, w.g0, ulist, n_rows, w.gnmax};
        if (la_ok && la->next_pairs) {                           // (the finish kernel of this step forms the next step's list)
            look_pairs = la->next_pairs;
            look_ulist = w.ulist + (size_t)(par ^ 1) * w.ucap; look_n_u = w.n_u + (par ^ 1);
        }
        if (la) { la->have = la_ok && la->next_pairs != nullptr; la->parity = par ^ 1; }
        PC_LAUNCH(sample_hidden_kernel, dim3(ca.nb_s + (T + UT - 1) / UT), dim3(256), 0, st, ca);
        SampleSimsArgs sa = {};
        sa.hd = w.csamp; sa.G = w.gmat; sa.g0 = w.g0; sa.B = B; sa.T = T; sa.K = K; sa.nchunks = w.nchunks_s;
        sa.part_val = w.part_val; sa.n_rows = n_rows;
        // the dense gradients of the two big tables hold zeros outside the touched rows (and receive float atomics beyond
        // TG_CAP touched rows): cleared by rider workgroups of this launch
        sa.zero[0] = g->query_types; sa.nzero[0] = (size_t)T * PC_L; sa.zero[1] = g->comp_types; sa.nzero[1] = (size_t)T * PC_L;
        sa.zcols = 8;
        const int tiles_s = (rows_cap + SUT - 1) / SUT;
        // SWPS workgroups per CU: y so that chunks x y fills the chip's slots (each workgroup then walks its share of the row tiles)
        int gy = (256 * SWPS - sa.zcols) / (sa.nchunks > 0 ? sa.nchunks : 1);
        gy = gy < 1 ? 1 : gy > tiles_s ? tiles_s : gy;
        // pass 1: the maximum of every 64-type sub-chunk per row; pass 2: the exact top K over each row's K best sub-chunks
        PC_LAUNCH(sample_sims_max_kernel, dim3(sa.nchunks + sa.zcols, gy), dim3(256), 0, st, sa);
        PC_LAUNCH(sample_topk_refine_kernel, dim3((rows_cap + 3) / 4), dim3(256), 0, st, w.part_val, 4 * sa.nchunks, w.csamp, w.gmat, w.g0,
                  w.gnmax, B, T, K, w.topk_by_type, ulist, n_rows);
        PC_TRY(pc_launch_status());
    }

    FusedArgs fa = {};
    fa.table = p->product_table; fa.enc_w = p->enc_w; fa.enc_b = p->enc_b; fa.dec_w = p->dec_w; fa.dec_b = p->dec_b;
    fa.typ_w = p->typ_w; fa.typ_b = p->typ_b; fa.itm_w = p->itm_w; fa.itm_b = p->itm_b; fa.eq = p->query_types;
    fa.ec = p->comp_types;
    fa.query_idx = query_idx; fa.query_types = query_types; fa.pos_types = pos_types; fa.neg_types = neg_types;
    fa.pos_items = pos_items; fa.neg_items = neg_items;
    fa.B = B; fa.T = T; fa.K = K; fa.P = num_products;
    fa.margin = margin; fa.g_type = (1.0f - alpha) / (float)B; fa.g_item = alpha / ((float)B * (float)K);
    fa.drop = make_dropcfg(p->dropout);
    fa.topk_by_type = w.topk_by_type; fa.topk_per_sample = per_sample ? 1 : 0;
    fa.topk = topk; fa.part_type = w.part; fa.part_item = w.part + B;
    fa.h = w.h; fa.dpi = w.dpi; fa.dtp = w.dtp; fa.dc = w.dc; fa.dh = w.dh; fa.dt = w.dt;
    fa.ecsrc = w.ecsrc; fa.ecidx = w.ecidx; fa.cids = w.cids; fa.bad = bad_count; fa.step_count = adam ? step_count : nullptr;
    fa.run_counts = (!w.small && w.sorted_path) ? w.n_touch + 2 : nullptr;
    fa.slabs = w.wslabs; fa.slab_floats = w.wslab_floats;
    if (pairs_in_tile) {
        fa.pairs = src->pairs; fa.features = src->features; fa.type_idx = src->type_idx; fa.n_types_mod = src->n_types;
        fa.bseed = src->seed; fa.bstep = src->step;
        fa.o_qidx = const_cast<int32_t*>(query_idx); fa.o_qt = const_cast<int32_t*>(query_types);
        fa.o_pt = const_cast<int32_t*>(pos_types); fa.o_nt = const_cast<int32_t*>(neg_types);
        fa.o_pos = const_cast<float*>(pos_items); fa.o_neg = const_cast<float*>(neg_items);
    }
    const int tiles = (B + TS - 1) / TS;                          // == w.wg_blocks (WG_S == TS): one slab per tile
    const int ldsims = w.small ? ((T + 15) / 16 * 16 + 4) : 0;
    // where the gradient products run: inside the tile kernel (T <= 128: tables included; T > 512: weights only, the table
    // rows go by scatter-add) or, for 128 < T <= 512, in joint_wgrad_kernel over the row buffers
    const bool wgrad_in_tile = !w.small || T <= T_WGRAD;
    const size_t lds = tile_lds_bytes(T, w.small, w.small && wgrad_in_tile);
    static const hipError_t attr[10] = {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_tile_kernel<false, 3, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_tile_kernel<false, 0, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_tile_kernel<true, 3, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_tile_kernel<true, 0, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_tile_kernel<true, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_tile_kernel<true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_tile_kernel<true, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_tile_kernel<true, 0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_tile_kernel<false, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
        hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_tile_kernel<false, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)};
    (void)attr;
    if (pairs_in_tile && w.small) {
        if (K == 3) PC_LAUNCH((joint_tile_kernel<true, 3, true, true>), dim3(tiles), dim3(256), lds, st, fa, ldsims);
        else PC_LAUNCH((joint_tile_kernel<true, 0, true, true>), dim3(tiles), dim3(256), lds, st, fa, ldsims);
    } else if (pairs_in_tile) {
        if (K == 3) PC_LAUNCH((joint_tile_kernel<false, 3, true, true>), dim3(tiles), dim3(256), lds, st, fa, ldsims);
        else PC_LAUNCH((joint_tile_kernel<false, 0, true, true>), dim3(tiles), dim3(256), lds, st, fa, ldsims);
    } else if (w.small && wgrad_in_tile) {
        if (K == 3) PC_LAUNCH((joint_tile_kernel<true, 3, true>), dim3(tiles), dim3(256), lds, st, fa, ldsims);
        else PC_LAUNCH((joint_tile_kernel<true, 0, true>), dim3(tiles), dim3(256), lds, st, fa, ldsims);
    } else if (w.small) {
        if (K == 3) PC_LAUNCH((joint_tile_kernel<true, 3, false>), dim3(tiles), dim3(256), lds, st, fa, ldsims);
        else PC_LAUNCH((joint_tile_kernel<true, 0, false>), dim3(tiles), dim3(256), lds, st, fa, ldsims);
    } else {
        if (K == 3) PC_LAUNCH((joint_tile_kernel<false, 3, true>), dim3(tiles), dim3(256), lds, st, fa, ldsims);
        else PC_LAUNCH((joint_tile_kernel<false, 0, true>), dim3(tiles), dim3(256), lds, st, fa, ldsims);
    }
    PC_TRY(pc_launch_status());

    // ---- gradient products over the row buffers: one launch, one slab per workgroup, summed by the finish kernel
    WgradArgs wa = {};
    wa.table = p->product_table; wa.eq = p->query_types; wa.ec = p->comp_types;
    wa.query_idx = w.cids; wa.query_types = w.cids + B; wa.topk = topk;      // (validated by the tile kernel)
    wa.dpi = w.dpi; wa.dtp = w.dtp; wa.dc = w.dc; wa.h = w.h; wa.dh = w.dh; wa.dt = w.dt; wa.ecsrc = w.ecsrc; wa.ecidx = w.ecidx;
    wa.B = B; wa.T = w.small ? T : 0; wa.K = K; wa.slabs = w.wslabs; wa.slab_floats = w.wslab_floats;
    if (!wgrad_in_tile) {
        // (only 128 < T <= 512 comes here: 16 type blocks per wave and table)
        static const hipError_t wattr[2] = {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_wgrad_kernel<16, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
            hipFuncSetAttribute(reinterpret_cast<const void*>(&joint_wgrad_kernel<16, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)};
        (void)wattr;
        const size_t wl = wgrad_lds_bytes();
        if (K == 3) PC_LAUNCH((joint_wgrad_kernel<16, 3>), dim3(w.wg_blocks), dim3(512), wl, st, wa);
        else PC_LAUNCH((joint_wgrad_kernel<16, 0>), dim3(w.wg_blocks), dim3(512), wl, st, wa);
        PC_TRY(pc_launch_status());
    }
    // which form sums the table gradients (PC_OPT_SORTED_TABLE_GRADIENTS in the header): the sorted one wherever the lists fit
    // its sort kernel (24 us at T = 34800, B = 4096 -- the three launches of the LDS-table form take 37, and that form is
    // reproducible only up to 512 touched rows per table); the option's value 0 keeps the LDS-table form for steps without
    // hidden-layer dropout
    const bool sorted_tables = !w.small && w.sorted_path && (per_sample || pc_opt_sorted_tables());
    if (sorted_tables) {
        // table gradients: source rows sorted by destination, then one wave per destination adds its run in ascending source order
        const int nc = B * (K + 2), cap_c = nc < T ? nc : T, cap_q = B < T ? B : T;
        // (n_touch: [0, 1] touched rows per table, [2 .. 6) run-list counters, [8 .. 8 + 2 TS_NR) runs per range of either list)
        const SortList sc = {w.ecidx, nc, w.srt_c, w.seg_c, w.med_c, w.lng_c, w.gnt_c, w.n_touch + 8, w.gct_c};
        const SortList sq = {w.cids + B, B, w.srt_q, w.seg_q, w.med_q, w.lng_q, w.gnt_q, w.n_touch + 8 + TS_NR, w.gct_q};
        static const hipError_t sattr = hipFuncSetAttribute(reinterpret_cast<const void*>(&table_sort_kernel),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)sattr;
        PC_LAUNCH(table_sort_kernel, dim3(2 * TS_NR), dim3(1024), table_sort_lds_bytes(nc, T), st, sc, sq, T, w.n_touch);
        const SegList gc = {g->comp_types, w.ecsrc, w.srt_c, w.seg_c, w.tl_c, w.med_c, w.lng_c, sc.nruns, sc.n, w.gnt_c, w.gct_c, w.gpr_c};
        const SegList gq = {g->query_types, w.dt, w.srt_q, w.seg_q, w.tl_q, w.med_q, w.lng_q, sq.nruns, sq.n, w.gnt_q, w.gct_q, w.gpr_q};
        // grid regions sized for the capacities (the counts live on the device: workgroups past them leave at once): runs of up to
        // four rows sixteen per workgroup, runs of 5 .. TS_LONG rows (at most n / 5 of them) a wave each, longer ones a workgroup each
        SegGrid gr;
        int at = 0;
        // (short runs: per range of the table's rows, at most min(its bins, the list's rows) of them)
        const int bins = 2 * ts_range_words(T);
        gr.rs = ts_range_stride(T);
        // (every region is bounded: its workgroups walk on by the size of the region while their list lasts -- a workgroup that
        // finds nothing to do still waits a round trip to memory for the count that tells it so, and five thousand of them, seven
        // to a CU at a time, were half of this kernel)
        auto upto = [](int v, int cap) { return v < cap ? v : cap; };
        gr.spr[0] = upto(((bins < cap_c ? bins : cap_c) + 15) / 16, TS_SPR); gr.spr[1] = upto(((bins < cap_q ? bins : cap_q) + 15) / 16, TS_SPR);
        const int sizes[8] = {TS_NR * gr.spr[0], upto((nc / 5 + 3) / 4, TS_GMED), upto(nc / TS_LONG + 1, TS_GLONG), upto(ts_giant_cap(nc), TS_GGIANT),
                              TS_NR * gr.spr[1], upto((B / 5 + 3) / 4, TS_GMED), upto(B / TS_LONG + 1, TS_GLONG), upto(ts_giant_cap(B), TS_GGIANT)};
        for (int i = 0; i < 8; i++) { gr.start[i] = at; at += sizes[i]; }
        gr.start[8] = at;
        PC_LAUNCH(table_segsum_kernel, dim3(at), dim3(256), 0, st, gc, gq, w.n_touch, gr);
        PC_TRY(pc_launch_status());
    } else if (!w.small) {
        // (lists too long for the sort kernel's LDS, or T > 65535) fixed-order sums over the touched rows while a table has <= TG_CAP
        // of them, float atomics beyond
        const int words = (T + 31) / 32;
        PC_LAUNCH(touched_types_kernel, dim3(2), dim3(1024), (size_t)(words + 1024) * 4, st, w.ecidx, B * (K + 2), w.cids + B, B, T,
                  w.tl_c, w.tp_c, w.tl_q, w.tp_q, w.n_touch);
        const TableList lc = {g->comp_types, w.ecidx, w.ecsrc, B * (K + 2), w.tp_c, w.tslab_c};
        const TableList lq = {g->query_types, w.cids + B, w.dt, B, w.tp_q, w.tslab_q};
        const size_t tlds = ((size_t)TG_CAP * PC_L + (size_t)TG_CHUNK * PC_L + TG_CHUNK) * 4;
        static const hipError_t tattr = hipFuncSetAttribute(reinterpret_cast<const void*>(&table_partials_kernel),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)tattr;
        PC_LAUNCH(table_partials_kernel, dim3(TG_WGS), dim3(256), tlds, st, lc, lq, w.n_touch, T);
        PC_LAUNCH(table_reduce_kernel, dim3(2 * (TG_CAP * (PC_L / 4) * 8 / 256)), dim3(256), 0, st, lc, lq, w.tl_c, w.tl_q, w.n_touch);
        PC_TRY(pc_launch_status());
    }

    // ---- finish: slab sums -> .grad, the losses, Adam
    FinishArgs fin = {};
    fin.sl16 = w.small ? 1 : 0;
    int blocks = 0, nj = 0;
    auto add = [&](const float* slabs, size_t stride, int nsplit, int n, float* grad, float* param, float* m, float* v) {
        FinishJob& j = fin.job[nj];
        j.slabs = slabs; j.stride = stride; j.nsplit = nsplit; j.n = n; j.grad = grad; j.param = param; j.m = m; j.v = v;
        fin.block0[nj++] = blocks;
        blocks += nsplit > 0 ? (fin.sl16 ? (n / 4 + 15) / 16 : (n / 4 + 31) / 32) : (n / 4 + 255) / 256;
    };
    {
        const int Tt = wa.T;
        const int off[10] = {wg_off_itm_w(), wg_off_itm_b(), wg_off_typ_w(), wg_off_typ_b(), wg_off_dec_w(), wg_off_dec_b(),
                             wg_off_enc_w(), wg_off_enc_b(), wg_off_ec(), wg_off_eq(Tt)};
        const int cnt[10] = {PC_D * PC_D, PC_D, PC_D * PC_L, PC_D, PC_L * LH, PC_L, LH * PC_L, LH, Tt * PC_L, Tt * PC_L};
        float* const gq[10] = {g->itm_w, g->itm_b, g->typ_w, g->typ_b, g->dec_w, g->dec_b, g->enc_w, g->enc_b, g->comp_types, g->query_types};
        float* const pq[10] = {p->itm_w, p->itm_b, p->typ_w, p->typ_b, p->dec_w, p->dec_b, p->enc_w, p->enc_b, p->comp_types, p->query_types};
        float *mq[10] = {}, *vq[10] = {};
        if (adam) {
            float* const m_[10] = {exp_avg->itm_w, exp_avg->itm_b, exp_avg->typ_w, exp_avg->typ_b, exp_avg->dec_w, exp_avg->dec_b,
                                   exp_avg->enc_w, exp_avg->enc_b, exp_avg->comp_types, exp_avg->query_types};
            float* const v_[10] = {exp_avg_sq->itm_w, exp_avg_sq->itm_b, exp_avg_sq->typ_w, exp_avg_sq->typ_b, exp_avg_sq->dec_w,
                                   exp_avg_sq->dec_b, exp_avg_sq->enc_w, exp_avg_sq->enc_b, exp_avg_sq->comp_types, exp_avg_sq->query_types};
            for (int i = 0; i < 10; i++) { mq[i] = m_[i]; vq[i] = v_[i]; }
        }
        for (int i = 0; i < 10; i++) {
            if (cnt[i] > 0) add(w.wslabs + off[i], (size_t)w.wslab_floats, w.wg_blocks, cnt[i], gq[i], pq[i], mq[i], vq[i]);
            else if (adam)                                             // big tables: gradient complete already, Adam only
                add(nullptr, 0, 0, T * PC_L, gq[i], pq[i], mq[i], vq[i]);
        }
    }
    fin.njobs = nj;
    for (int k = nj; k <= FIN_JOBS; k++) fin.block0[k] = blocks;
    fin.part_type = w.part; fin.part_item = w.part + B; fin.B = B; fin.K = K; fin.alpha = alpha; fin.losses = losses;
    fin.step_count = step_count; fin.lr = lr; fin.beta1 = beta1; fin.beta2 = beta2; fin.eps = eps; fin.adam = adam ? 1 : 0;
    if (look_pairs) {
        fin.next_pairs = look_pairs; fin.type_idx = src->type_idx; fin.P = num_products; fin.next_B = B; fin.T = T;
        fin.next_ulist = look_ulist; fin.next_n_u = look_n_u;
    }
    PC_LAUNCH(joint_finish_kernel, dim3(blocks + 1 + (look_pairs ? 1 : 0)), dim3(256), 0, st, fin);
    return pc_launch_status();
}

extern "C" int pc_joint_fused_step(const pc_joint_tensors* p, const pc_joint_tensors* g, const pc_joint_tensors* exp_avg,
                                   const pc_joint_tensors* exp_avg_sq, int64_t* step_count, double lr, double beta1,
                                   double beta2, double eps, const int32_t* query_idx, const int32_t* query_types,
                                   const int32_t* pos_types, const int32_t* neg_types, const float* pos_items,
                                   const float* neg_items, int B, int T, int K, int num_products, float margin, float alpha,
                                   float* losses, int32_t* topk, int32_t* bad_count, void* ws, size_t ws_bytes,
                                   void* stream) {
    return fused_step_impl(p, g, exp_avg, exp_avg_sq, step_count, lr, beta1, beta2, eps, nullptr, query_idx, query_types,
                           pos_types, neg_types, pos_items, neg_items, B, T, K, num_products, margin, alpha, losses, topk,
                           bad_count, ws, ws_bytes, stream);
}

extern "C" int pc_joint_fused_step_pairs(const pc_joint_tensors* p, const pc_joint_tensors* g, const pc_joint_tensors* exp_avg,
                                         const pc_joint_tensors* exp_avg_sq, int64_t* step_count, double lr, double beta1,
                                         double beta2, double eps, const int32_t* pairs, const float* features,
                                         const int32_t* type_idx, int n_types, uint64_t seed, uint64_t step,
                                         int32_t* query_idx, int32_t* query_types, int32_t* pos_types, int32_t* neg_types,
                                         float* pos_items, float* neg_items, int B, int T, int K, int num_products,
                                         float margin, float alpha, float* losses, int32_t* topk, int32_t* bad_count,
                                         void* ws, size_t ws_bytes, void* stream) {
    if (!pairs || !features || !type_idx || n_types <= 0) return PC_EINVAL;
    const PairsSrc src = {pairs, features, type_idx, n_types, seed, step};
    return fused_step_impl(p, g, exp_avg, exp_avg_sq, step_count, lr, beta1, beta2, eps, &src, query_idx, query_types,
                           pos_types, neg_types, pos_items, neg_items, B, T, K, num_products, margin, alpha, losses, topk,
                           bad_count, ws, ws_bytes, stream);
}

// train.py:36-57 (train_epoch: for batch in loader: forward, loss, zero_grad, backward, optimizer.step) for `n_pairs`
// labelled pairs already in epoch order on the device: full batches of `batch` pairs, then -- unless drop_last -- the
// ragged rest.  The host enqueues the launches back to back from this one call (a Python loop body costs more than the
// step's kernels run); losses_out[i] = {loss, type, item} of step i stays on the device (the reference's per-step
// loss.item() is a device round trip per batch: read losses_out once per epoch instead).  The batch buffers hold the
// LAST batch afterwards.  Dropout offsets and the loader's step counter advance by one per step from the given values.
extern "C" int pc_joint_train_epoch(const pc_joint_tensors* p, const pc_joint_tensors* g, const pc_joint_tensors* exp_avg,
                                    const pc_joint_tensors* exp_avg_sq, int64_t* step_count, double lr, double beta1,
                                    double beta2, double eps, const int32_t* pairs, int64_t n_pairs, const float* features,
                                    const int32_t* type_idx, int n_types, uint64_t seed, uint64_t first_step,
                                    int32_t* query_idx, int32_t* query_types, int32_t* pos_types, int32_t* neg_types,
                                    float* pos_items, float* neg_items, int B, int drop_last, int T, int K, int num_products,
                                    float margin, float alpha, float* losses_out, int32_t* topk, int32_t* bad_count,
                                    void* ws, size_t ws_bytes, void* stream) {
    if (!p || !pairs || !features || !type_idx || n_types <= 0 || n_pairs < 0 || B <= 0 || !losses_out) return PC_EINVAL;
    if (!exp_avg || !exp_avg_sq) return PC_EINVAL;               // an epoch without the optimizer step trains nothing
    pc_joint_tensors pl = *p;
    int64_t done = 0;
    JointLookahead la = {false, 0, nullptr};
    for (int64_t i = 0; done < n_pairs; i++) {
        const int64_t left = n_pairs - done;
        const int b = left >= B ? B : (int)left;
        if (b < B && drop_last) break;
        const PairsSrc src = {pairs + 3 * done, features, type_idx, n_types, seed, first_step + (uint64_t)i};
        pl.dropout.offset = p->dropout.offset + (uint64_t)i;
        // (the next step's list only when that step has this one's batch size: the workspace layout is a function of it)
        la.next_pairs = (left - b >= b && b == B) ? pairs + 3 * (done + b) : nullptr;
        PC_TRY(fused_step_impl(&pl, g, exp_avg, exp_avg_sq, step_count, lr, beta1, beta2, eps, &src, query_idx, query_types,
                               pos_types, neg_types, pos_items, neg_items, b, T, K, num_products, margin, alpha,
                               losses_out + 3 * i, topk, bad_count, ws, ws_bytes, stream, &la));
        done += b;
    }
    return PC_OK;
}

// The same epoch for a REPLICA of a data-parallel job (ABI 6): per step the fused step without its Adam (gradients only), the
// exchange slot -- pc_rccl_allreduce_mean on the library's own RCCL communicator, or whatever the caller supplies -- on the
// step's stream, then Adam over the flat buffers.  The gradient exchange sits where train.py:46-48 has nothing
// (loss.backward(); optimizer.step()): issued from this call, between two kernel launches, not from a host-language hook per
// step (the step is ~2 kernel latencies long).
extern "C" int pc_joint_train_epoch_plan(const pc_joint_tensors* p, const pc_joint_tensors* g, float* param_flat, float* grad_flat,
                                       float* exp_avg_flat, float* exp_avg_sq_flat, size_t n_flat, int64_t* step_count,
                                       int64_t t_first, float* adam_scalars, double lr, double beta1, double beta2, double eps,
                                       const pc_exchange_plan* plan, const int32_t* pairs, int64_t n_pairs,
                                       const float* features, const int32_t* type_idx, int n_types, uint64_t seed,
                                       uint64_t first_step, int32_t* query_idx, int32_t* query_types, int32_t* pos_types,
                                       int32_t* neg_types, float* pos_items, float* neg_items, int B, int drop_last, int T, int K,
                                       int num_products, float margin, float alpha, float* losses_out, int32_t* topk,
                                       int32_t* bad_count, void* ws, size_t ws_bytes, void* stream) {
    if (!p || !g || !pairs || !features || !type_idx || n_types <= 0 || n_pairs < 0 || B <= 0 || !losses_out) return PC_EINVAL;
    if (!param_flat || !grad_flat || !exp_avg_flat || !exp_avg_sq_flat || n_flat == 0 || t_first < 0) return PC_EINVAL;
    if (t_first == 0 && (!step_count || !adam_scalars)) return PC_EINVAL;
    // p / g are views into the flat buffers: what the exchange averages and Adam updates must be what the step reads and writes
    const float* const gp[10] = {g->itm_w, g->itm_b, g->typ_w, g->typ_b, g->dec_w, g->dec_b, g->enc_w, g->enc_b, g->comp_types, g->query_types};
    const float* const pp[10] = {p->itm_w, p->itm_b, p->typ_w, p->typ_b, p->dec_w, p->dec_b, p->enc_w, p->enc_b, p->comp_types, p->query_types};
    for (int i = 0; i < 10; i++) {
        if (!gp[i] || !pp[i]) return PC_EINVAL;
        if (gp[i] < grad_flat || gp[i] >= grad_flat + n_flat || pp[i] < param_flat || pp[i] >= param_flat + n_flat) return PC_EINVAL;
        if (gp[i] - grad_flat != pp[i] - param_flat) return PC_EINVAL;
    }
    pc_joint_tensors pl = *p;
    int64_t done = 0;
    JointLookahead la = {false, 0, nullptr};
    for (int64_t i = 0; done < n_pairs; i++) {
        const int64_t left = n_pairs - done;
        const int b = left >= B ? B : (int)left;
        if (b < B && drop_last) break;
        const PairsSrc src = {pairs + 3 * done, features, type_idx, n_types, seed, first_step + (uint64_t)i};
        pl.dropout.offset = p->dropout.offset + (uint64_t)i;
        la.next_pairs = (left - b >= b && b == B) ? pairs + 3 * (done + b) : nullptr;
        PC_TRY(fused_step_impl(&pl, g, nullptr, nullptr, nullptr, lr, beta1, beta2, eps, &src, query_idx, query_types, pos_types,
                               neg_types, pos_items, neg_items, b, T, K, num_products, margin, alpha, losses_out + 3 * i, topk,
                               bad_count, ws, ws_bytes, stream, &la));
        PC_TRY(pc_exchange_adam_plan(plan, param_flat, grad_flat, exp_avg_flat, exp_avg_sq_flat, n_flat, step_count,
                                     t_first > 0 ? t_first + i : 0, adam_scalars, lr, beta1, beta2, eps, stream));
        done += b;
    }
    return PC_OK;
}

extern "C" int pc_joint_train_epoch_dp(const pc_joint_tensors* p, const pc_joint_tensors* g, float* param_flat, float* grad_flat,
                                       float* exp_avg_flat, float* exp_avg_sq_flat, size_t n_flat, int64_t* step_count,
                                       int64_t t_first, float* adam_scalars, double lr, double beta1, double beta2, double eps,
                                       pc_exchange_fn exchange, void* exchange_ctx, const int32_t* pairs, int64_t n_pairs,
                                       const float* features, const int32_t* type_idx, int n_types, uint64_t seed,
                                       uint64_t first_step, int32_t* query_idx, int32_t* query_types, int32_t* pos_types,
                                       int32_t* neg_types, float* pos_items, float* neg_items, int B, int drop_last, int T, int K,
                                       int num_products, float margin, float alpha, float* losses_out, int32_t* topk,
                                       int32_t* bad_count, void* ws, size_t ws_bytes, void* stream) {
    const pc_exchange_plan plan = {exchange, nullptr, nullptr, exchange_ctx, 0, 1, 0};
    return pc_joint_train_epoch_plan(p, g, param_flat, grad_flat, exp_avg_flat, exp_avg_sq_flat, n_flat, step_count, t_first,
                                     adam_scalars, lr, beta1, beta2, eps, &plan, pairs, n_pairs, features, type_idx, n_types, seed,
                                     first_step, query_idx, query_types, pos_types, neg_types, pos_items, neg_items, B, drop_last, T,
                                     K, num_products, margin, alpha, losses_out, topk, bad_count, ws, ws_bytes, stream);
}
