// Shared device/host helpers for the gfx950 kernels of libpcompanion_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pcompanion_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define PC_WAVE 64

#define PC_HIP_TRY(expr)                          \
    do {                                          \
        hipError_t _e = (expr);                   \
        if (_e != hipSuccess) return (int)_e;     \
    } while (0)

#define PC_TRY(expr)                 \
    do {                             \
        int _r = (expr);             \
        if (_r != 0) return _r;      \
    } while (0)

// Kernel launch.  A stale (sticky) runtime error left by an unrelated earlier call on this thread is dropped first;
// the launch's own error, if any, is kept in a thread-local until pc_launch_status() collects it, so a sequence of
// launches checked once at its end still reports the FIRST failure (a later PC_LAUNCH does not wipe it).
extern thread_local int pc_tls_launch_err;
#define PC_LAUNCH(...)                                                           \
    do {                                                                         \
        (void)hipGetLastError();                                                 \
        hipLaunchKernelGGL(__VA_ARGS__);                                         \
        const hipError_t _le = hipGetLastError();                                \
        if (_le != hipSuccess && pc_tls_launch_err == 0) pc_tls_launch_err = (int)_le; \
    } while (0)

static inline int pc_launch_status() {
    const int e = pc_tls_launch_err;
    pc_tls_launch_err = 0;
    return e;
}

// Row segments (BatchNorm call groups) passed to kernels BY VALUE.
struct SegInfo {
    int nseg;
    int start[PC_MAX_SEG + 1];   // row starts, start[nseg] = total rows
    int tile0[PC_MAX_SEG + 1];   // first 128-row tile of each segment (tiles never straddle)
    int count[PC_MAX_SEG];       // logical rows per segment (what BatchNorm divides by)
    int wrow;                    // the one row that stands for `wmult` identical rows (-1: none)
    float wmult;
    const float* roww;           // multiplicities of rows [w0, w0 + wn) (null: none): unique-neighbour layout
    int w0, wn;
};

static inline SegInfo make_seginfo(const pc_segments* s, int rows, int tile_rows) {
    SegInfo si;
    si.nseg = 1;
    for (int i = 0; i <= PC_MAX_SEG; i++) { si.start[i] = rows; si.tile0[i] = 0; }
    si.start[0] = 0;
    si.wrow = -1;
    si.wmult = 1.f;
    si.roww = nullptr; si.w0 = 0; si.wn = 0;
    if (s) {
        si.nseg = s->nseg;
        for (int i = 0; i <= s->nseg; i++) si.start[i] = s->start[i];
        for (int i = s->nseg + 1; i <= PC_MAX_SEG; i++) si.start[i] = s->start[s->nseg];
        si.wrow = s->weighted_row;
        si.wmult = s->weight;
        if (s->row_weight && s->row_weight_rows > 0) { si.roww = s->row_weight; si.w0 = s->row_weight_start; si.wn = s->row_weight_rows; }
    }
    for (int i = 0; i < PC_MAX_SEG; i++) {
        const int phys = si.start[i + 1] - si.start[i];
        si.count[i] = (s && i < s->nseg && s->count[i] > 0) ? s->count[i] : phys;
    }
    int t = 0;
    for (int i = 0; i < PC_MAX_SEG; i++) {
        si.tile0[i] = t;
        int n = si.start[i + 1] - si.start[i];
        t += (n + tile_rows - 1) / tile_rows;
    }
    si.tile0[PC_MAX_SEG] = t;
    return si;
}

static inline int seg_valid(const pc_segments* s, int rows) {
    if (!s) return 1;
    if (s->nseg < 1 || s->nseg > PC_MAX_SEG || s->start[0] != 0 || s->start[s->nseg] != rows) return 0;
    for (int i = 0; i < s->nseg; i++)
        if (s->start[i + 1] < s->start[i] || s->count[i] < 0) return 0;
    if (s->weighted_row >= rows || (s->weighted_row >= 0 && s->weight < 0.f)) return 0;
    if (s->row_weight && (s->row_weight_start < 0 || s->row_weight_rows < 0 || s->row_weight_start + s->row_weight_rows > rows))
        return 0;
    return 1;
}

#ifdef __HIPCC__
__device__ __forceinline__ int seg_of_row(const SegInfo& si, int r) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < PC_MAX_SEG; i++) s += (i < si.nseg && r >= si.start[i]) ? 1 : 0;
    return s;
}

// how many identical logical rows physical row r stands for (1 unless the caller said otherwise)
__device__ __forceinline__ float row_multiplicity(const SegInfo& si, int r) {
#ifdef PC_EXP_NO_ROWMULT
    return 1.f;                                   // (developer experiment: what the per-row multiplicity loads cost; WRONG statistics)
#endif
    if (si.roww && (unsigned)(r - si.w0) < (unsigned)si.wn) return si.roww[r - si.w0];
    return r == si.wrow ? si.wmult : 1.f;
}

__device__ __forceinline__ int seg_of_tile(const SegInfo& si, int t) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < PC_MAX_SEG; i++) s += (i < si.nseg && t >= si.tile0[i]) ? 1 : 0;
    return s;
}

// tanh via one v_exp_f32 and one v_rcp_f32: tanh(x) = 1 - 2/(e^{2x}+1).
// |abs err| < 2e-7 over the real line (saturates cleanly: e^{2x}=inf -> 1, 0 -> -1).
__device__ __forceinline__ float fast_tanh(float x) {
    // v_exp_f32 takes a base-2 exponent: e^{2x} = 2^{2x*log2(e)}; v_rcp_f32 is a 1-ulp reciprocal
    // (__frcp_rn would expand to the ~10-instruction correctly-rounded division sequence)
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- fp32-grade products on the BF16 matrix cores (16x the fp32 MFMA rate): every fp32 operand value is split
// into three bf16 pieces, a = a0 + a1 + a2 (8 + 8 + 8 mantissa bits; the first two residuals are exact in fp32),
// and the six significant cross products a0b0, a0b1, a1b0, a1b1, a0b2, a2b0 are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16.  Measured error against an fp64 product: the same as v_mfma_f32_32x32x2_f32's
// (scripts/microbench/nt_bf16x6.hip: 7.9e-5 vs 9.4e-5 on K = 2048 sums of magnitude 58).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Split3 { bf16x8 p0, p1, p2; };
// Pieces by TRUNCATION (top 16 bits of the fp32 pattern): a0 = trunc(a), r1 = a - a0 (exact), a1 = trunc(r1),
// r2 = r1 - a1 (exact), a2 = trunc(r2): |a - (a0+a1+a2)| < 2^-24 |a|.  Only full-rate VALU instructions
// (v_and / v_sub / v_perm): the RNE conversion v_cvt_pk_bf16_f32 made the split, not the matrix pipe, the bound.
__device__ __forceinline__ Split3 split3(const float4& lo, const float4& hi) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned u0[8], u1[8], u2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u0[i] = __float_as_uint(v[i]);
        const float r1 = v[i] - __uint_as_float(u0[i] & 0xffff0000u);
        u1[i] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1[i] & 0xffff0000u);
        u2[i] = __float_as_uint(r2);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 4; j++) {            // dword j = {element 2j+1 (high half), element 2j (low half)}: the high halves of two words
        q0[j] = __builtin_amdgcn_perm(u0[2 * j + 1], u0[2 * j], 0x07060302u);
        q1[j] = __builtin_amdgcn_perm(u1[2 * j + 1], u1[2 * j], 0x07060302u);
        q2[j] = __builtin_amdgcn_perm(u2[2 * j + 1], u2[2 * j], 0x07060302u);
    }
    Split3 s;
    s.p0 = __builtin_bit_cast(bf16x8, q0);
    s.p1 = __builtin_bit_cast(bf16x8, q1);
    s.p2 = __builtin_bit_cast(bf16x8, q2);
    return s;
}
__device__ __forceinline__ f32x16 mfma_bf16(const bf16x8& a, const bf16x8& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// Wave priority 1 around a matrix-instruction cluster of a large product's K loop (cdna_hip_programming.md T5), per kernel
// variant.  Measured in A/B on two boxes (round 6, profiles/r06_setprio_ab.txt: alternating runs of the builds, three boxes): around the
// twelve-product clusters of the PLAIN weight-gradient kernel (gemm_tn8_kernel without a prologue: dW5, dW3's halves; 8 waves,
// two per SIMD) the step is 4.5 / 10 / 15 us shorter; in the in-place-prologue variants, held across a whole k group, or at level 3
// it gains less or nothing; in gemm_nt_kernel it gains nothing on Linear3's tile and LENGTHENS the three-workgroups-per-CU
// variants (dZ1 +2.6 %, dZ2 +2.4 %); priority 1 through gemm_nt_kernel's EPILOGUE (its memory streams against the
// neighbours' K loops) changes nothing (+-2 us).  The conditions are compile-time (developer builds: -DPC_PRIO_NT_COND=..., -DPC_PRIO_TN_COND=...).
#define PC_PRIO_MFMA(COND, x) do { if constexpr (COND) __builtin_amdgcn_s_setprio(x); } while (0)
#ifndef PC_PRIO_NT_COND
#define PC_PRIO_NT_COND false                     // gemm_nt_kernel: nowhere
#endif
#ifndef PC_PRIO_TN_COND
#define PC_PRIO_TN_COND (!APRO && !ZPRO)          // gemm_tn8_kernel: the plain weight gradients
#endif
#endif

// ---- training-mode dropout masks (pc_dropout in the header; restated for the tests by philox_oracle.dropout_mask)
struct DropCfg {
    unsigned thr;        // keep iff word >= thr; 0 = dropout off
    float scale;         // fp32 1 / (1 - p)
    unsigned k0, k1, c2, c3;
};
enum { PC_DROP_STREAM_ATTENTION = 0, PC_DROP_STREAM_HIDDEN = 1 };
static inline DropCfg make_dropcfg(const pc_dropout& d) {
    DropCfg c = {};
    if (d.p > 0.f) {
        double t = (double)d.p * 4294967296.0;
        c.thr = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
        c.scale = 1.0f / (1.0f - d.p);
        c.k0 = (unsigned)d.seed; c.k1 = (unsigned)(d.seed >> 32);
        c.c2 = (unsigned)d.offset; c.c3 = (unsigned)(d.offset >> 32);
    }
    return c;
}
#ifdef __HIPCC__
// the four multipliers (0 or scale) of elements 4 * group .. 4 * group + 3 of stream `stream`
__device__ __forceinline__ void pc_dropout_keep4(const DropCfg& c, unsigned group, unsigned stream, float (&m)[4]) {
    unsigned x0 = group, x1 = stream, x2 = c.c2, x3 = c.c3, k0 = c.k0, k1 = c.k1;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * x0, p1 = (unsigned long long)0xCD9E8D57u * x2;
        const unsigned y0 = (unsigned)(p1 >> 32) ^ x1 ^ k0, y1 = (unsigned)p1;
        const unsigned y2 = (unsigned)(p0 >> 32) ^ x3 ^ k1, y3 = (unsigned)p0;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    m[0] = x0 >= c.thr ? c.scale : 0.f; m[1] = x1 >= c.thr ? c.scale : 0.f;
    m[2] = x2 >= c.thr ? c.scale : 0.f; m[3] = x3 >= c.thr ? c.scale : 0.f;
}
// torch.optim.Adam's single-tensor update of one element (train.py:24 defaults; step_size = lr / (1 - beta1^t),
// bc2s = sqrt(1 - beta2^t), both rounded from fp64).  ONE definition with contraction off, so that the stand-alone Adam
// kernel and the fused step's finish kernel round identically whatever code surrounds them.
__device__ __forceinline__ void pc_adam_update(float& p, float& m, float& v, float g, float step_size, float bc2s,
                                               float omb1, float beta2, float omb2, float eps) {
#pragma clang fp contract(off)
    m = m + (g - m) * omb1;
    v = v * beta2 + (omb2 * g) * g;
    p = p - step_size * (m / (sqrtf(v) / bc2s + eps));
}
// Four elements whose gradient and both moments are exactly zero: torch.optim.Adam's update leaves them as they are (m' = 0, v' = 0,
// p' = p - step * (0 / (0 + eps)) = p), so a kernel may skip their stores -- the dense update of the [NUM_TYPES, 64] tables
// (p_companion.py:36-43) is mostly such elements: rows no batch has touched yet.
__device__ __forceinline__ bool pc_adam_dead(const float4& g, const float4& m, const float4& v) {
    return g.x == 0.f && g.y == 0.f && g.z == 0.f && g.w == 0.f && m.x == 0.f && m.y == 0.f && m.z == 0.f && m.w == 0.f &&
           v.x == 0.f && v.y == 0.f && v.z == 0.f && v.w == 0.f;
}
// The N(0,1) filler rows of the complementary batch (data_loader.py:148-151: torch.randn_like in the reference's worker
// -- input DATA): chunk t = 32 b + c (16-B chunk c of row b) takes the first Philox4x32-10 block of the stream
// (seed; sample t, step) through Box-Muller.  One definition for the batch builder and for the fused step that builds
// its batch itself, so that both produce the same bits.
__device__ __forceinline__ float2 pc_box_muller(uint32_t a, uint32_t b) {
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0,1)
    const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.283185307179586f * u2, &sn, &cs);
    return make_float2(r * cs, r * sn);
}
__device__ __forceinline__ float4 pc_filler_chunk(uint64_t seed, uint64_t step, uint32_t t) {
    uint32_t x0 = 0u, x1 = t, x2 = (uint32_t)step, x3 = (uint32_t)(step >> 32), k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
        const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1;
        const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float2 n0 = pc_box_muller(x0, x1), n1 = pc_box_muller(x2, x3);
    return make_float4(n0.x, n0.y, n1.x, n1.y);
}
#endif

#ifdef __HIPCC__
// counter-based Philox4x32-10 stream keyed by (seed; sample, step): the throughput sampler's and the generator's RNG
struct Philox {
    uint32_t c[4], k[2], out[4];
    int have;
    __device__ Philox(uint64_t seed, uint64_t step, uint32_t sample) {
        k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32);
        c[0] = 0; c[1] = sample; c[2] = (uint32_t)step; c[3] = (uint32_t)(step >> 32);
        have = 0;
    }
    __device__ void block() {
        uint32_t x0 = c[0], x1 = c[1], x2 = c[2], x3 = c[3], k0 = k[0], k1 = k[1];
#pragma unroll
        for (int r = 0; r < 10; r++) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
            const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1;
            const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
            x0 = y0; x1 = y1; x2 = y2; x3 = y3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = x0; out[1] = x1; out[2] = x2; out[3] = x3;
        c[0]++;                      // draw-block counter
        have = 4;
    }
    __device__ uint32_t next() {
        if (have == 0) block();
        return out[4 - have--];
    }
    // unbiased integer in [0, n): top bit_length(n) bits, redraw while >= n (the rule of
    // CPython's _randbelow_with_getrandbits, applied to this stream)
    __device__ uint32_t below(uint32_t n) {
        const int bits = 32 - __clz(n);
        uint32_t r = next() >> (32 - bits);
        while (r >= n) r = next() >> (32 - bits);
        return r;
    }
};
#endif

// ---- optional per-launch timing (bench.py roofline leg): HIP events recorded on the launch
// stream around every gemm_nt / gemm_tn launch of ONE C-ABI call.  The pointer is
// thread-local and only set for the duration of that call (no persistent global state).
struct pc_profile {
    hipEvent_t* ev;      // 2 per bracket: start, stop
    int* kind;           // per bracket
    double* flops;       // per bracket: algorithmic FLOPs of the launch
    int capacity, used;
    unsigned kinds;      // bit k set: brackets of kind k are recorded (default: all)
};
enum { PC_KIND_GEMM_NT = 0, PC_KIND_GEMM_TN = 1, PC_KIND_GEMM_NT_SMALL = 2 };   // small: gemm_nt_small_kernel / the 2-wave few-row variant
extern thread_local pc_profile* pc_tls_profile;
struct ProfileScope {
    explicit ProfileScope(pc_profile* p) { pc_tls_profile = p; }
    ~ProfileScope() { pc_tls_profile = nullptr; }
};
int pc_prof_begin(int kind, double flops, hipStream_t st);   // returns bracket index or -1
void pc_prof_end(int bracket, hipStream_t st);

// ---- internal launchers (defined in the .hip files) -------------------------------------
enum NtPrologue { NT_PRO_NONE = 0, NT_PRO_BNTANH = 1 };
enum NtEpilogue {
    NT_EPI_NONE = 0,       // C = acc (+bias)
    NT_EPI_TANH = 1,       // C = tanh(acc + bias)
    NT_EPI_RELU = 2,       // C = relu(acc + bias)
    NT_EPI_DTANH = 3,      // C = acc * (1 - S^2),  S = aux[m][n]
    NT_EPI_DTANH_BN = 4,   // C = acc * (1 - S^2),  S = tanh(aux[m][n]*escale[s][n] + eshift[s][n])
    NT_EPI_DRELU = 5       // C = acc * (aux[m][n] > 0)
};
enum NtStats {
    NT_STAT_NONE = 0,
    NT_STAT_SUMSQ = 1,     // per 128-row tile: sum_r C, sum_r C^2           (BatchNorm forward)
    NT_STAT_BNBWD = 2      // per tile: sum_r C, sum_r C*xhat, xhat=(aux-mean)*invstd (BN backward)
};

struct NtArgs {
    const float* A; int lda; const int32_t* gather;   // A row r = gather ? A[gather[r]] (or 0 if <0) : A[r]
    const float* W; int ldw;                          // [N,K] row-major
    const float* bias;                                // [N] or null
    float* C; int ldc;                                // [M,N]
    int M, N, K;
    SegInfo seg;
    int prologue; const float* pscale; const float* pshift;     // [nseg][K]
    float* pro_out; int ldpo;                                   // NT_PRO_BNTANH: the transformed A rows are also written here [M,K] (or null)
    int epilogue; const float* aux; int ldaux;
    const float* escale; const float* eshift;                    // [nseg][N]
    int stats; float* stat_sum; float* stat_aux;                 // [ntiles][N] each
    const float* mean; const float* invstd;                      // [nseg][N] for NT_STAT_BNBWD
    // few-row kernel only: the bias of row m is scaled by brs[m * ldbrs] (null: 1).  The value projection of the
    // absorbed attention: ctx_h = Wv_h c_h + bv_h * sum_n(p_n m_n), where the sum is 1 only without dropout
    const float* brs; int ldbrs;
};
int launch_gemm_nt(const NtArgs& a, hipStream_t st);
int gemm_nt_tiles(const SegInfo& si);
// independent few-row products (M < 192 x 128 rows, N <= 128, K <= 128, no prologue / statistics) as one launch
#define PC_NT_GROUP 8
struct NtSmallGroup { NtArgs a[PC_NT_GROUP]; int ks_log2[PC_NT_GROUP], block0[PC_NT_GROUP + 1], n; };
static_assert(sizeof(NtSmallGroup) <= 4096, "kernel argument segment");
int launch_gemm_nt_group(const NtArgs* args, int n, hipStream_t st);
// two few-row products over the same 32-row tiles in ONE launch (gemm_nt.hip: the attention block's projection chains)
enum { NT_MODE_PLAIN = 0, NT_MODE_KHEAD = 1, NT_MODE_AHEAD = 2 };
// rider (optional): loss = mean_b relu(margin - d+ + d-) (product2vec.py:154) by ONE extra workgroup of the chain launch that
// follows the triplet-loss kernel in the fused step (a launch of its own costs ~4.5 us of latency for 16 KB of work)
struct HingeMeanJob { const float* d_pos; const float* d_neg; int B; float margin; float* loss; };
#ifdef __HIPCC__
// One sample of the triplet hinge (product2vec.py:137-154) by a group of 16 lanes at PRODUCT_EMB_DIM = 128: lane l16 owns dims
// [8 l16, 8 l16 + 8); d+ = ||a - p + eps||, d-_j = ||a - n_j + eps||, d- = mean_j d-_j; gradients of mean_b relu(margin - d+ + d-)
// w.r.t. a, p and every n_j.  ONE definition for the stand-alone kernel (four samples per wave) and for the prologue of the
// attention backward's first chain (the tile's sixteen samples at once): the same bits either way.  Rows past the batch
// (live == false) are read from sample 0 and nothing of them is stored; ga[8] returns d(loss)/d(a) (zeros when not live).
#define PC_LOSS_MAX_K 8
#define PC_PAIR_EPS 1e-6f
__device__ __forceinline__ void triplet_sample16(const float* __restrict__ a, const float* __restrict__ p, const float* __restrict__ n,
                                                 int b, int B, int K, float margin, bool live, float* d_pos, float* d_neg,
                                                 float* dp, float* dn, bool grads, int l16, float (&ga)[8]) {
    constexpr int D = 128;
    const size_t bb = live ? (size_t)b : 0;
    float av[8], dpv[8], dnv[PC_LOSS_MAX_K][8], dn_j[PC_LOSS_MAX_K];
    {
        const float4 a0 = *reinterpret_cast<const float4*>(a + bb * D + 8 * l16), a1 = *reinterpret_cast<const float4*>(a + bb * D + 8 * l16 + 4);
        const float4 p0 = *reinterpret_cast<const float4*>(p + bb * D + 8 * l16), p1 = *reinterpret_cast<const float4*>(p + bb * D + 8 * l16 + 4);
        av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
        const float pv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
        for (int c = 0; c < 8; c++) dpv[c] = av[c] - pv[c] + PC_PAIR_EPS;
    }
#pragma unroll
    for (int j = 0; j < PC_LOSS_MAX_K; j++)
        if (j < K) {
            const float* r = n + (bb * K + j) * D + 8 * l16;
            const float4 n0 = *reinterpret_cast<const float4*>(r), n1 = *reinterpret_cast<const float4*>(r + 4);
            const float nv[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
#pragma unroll
            for (int c = 0; c < 8; c++) dnv[j][c] = av[c] - nv[c] + PC_PAIR_EPS;
        }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) s += dpv[c] * dpv[c];
    const float dpos = sqrtf(group16_sum(s));
    float dneg = 0.f;
#pragma unroll
    for (int j = 0; j < PC_LOSS_MAX_K; j++)
        if (j < K) {
            float t = 0.f;
#pragma unroll
            for (int c = 0; c < 8; c++) t += dnv[j][c] * dnv[j][c];
            dn_j[j] = sqrtf(group16_sum(t));
            dneg += dn_j[j];
        }
    dneg /= (float)K;
    if (live && l16 == 0) { d_pos[b] = dpos; d_neg[b] = dneg; }
#pragma unroll
    for (int c = 0; c < 8; c++) ga[c] = 0.f;
    if (!grads) return;
    const bool active = (margin - dpos + dneg) > 0.f;      // relu'(0) = 0 as in torch
    const float gs = (active && live) ? 1.0f / (float)B : 0.f;        // d(mean)/d(l_b)
    const float ip = gs / dpos;                            // d l / d d+ = -1 ; d l / d d-_j = 1/K
#pragma unroll
    for (int c = 0; c < 8; c++) ga[c] = -dpv[c] * ip;
    if (live) {
        *reinterpret_cast<float4*>(dp + bb * D + 8 * l16) = make_float4(dpv[0] * ip, dpv[1] * ip, dpv[2] * ip, dpv[3] * ip);
        *reinterpret_cast<float4*>(dp + bb * D + 8 * l16 + 4) = make_float4(dpv[4] * ip, dpv[5] * ip, dpv[6] * ip, dpv[7] * ip);
    }
#pragma unroll
    for (int j = 0; j < PC_LOSS_MAX_K; j++)
        if (j < K) {
            const float in = gs / ((float)K * dn_j[j]);
#pragma unroll
            for (int c = 0; c < 8; c++) ga[c] += dnv[j][c] * in;
            if (live) {
                float* r = dn + (bb * K + j) * D + 8 * l16;
                *reinterpret_cast<float4*>(r) = make_float4(-dnv[j][0] * in, -dnv[j][1] * in, -dnv[j][2] * in, -dnv[j][3] * in);
                *reinterpret_cast<float4*>(r + 4) = make_float4(-dnv[j][4] * in, -dnv[j][5] * in, -dnv[j][6] * in, -dnv[j][7] * in);
            }
        }
}
#endif

// The hinge of product2vec.py:137-154 (forward and all three input gradients) as the PROLOGUE of the attention backward's first
// chain (round 6): the 16 samples of a chain tile form their own rows of d(loss)/d(anchor embedding) -- the chain's A operand --
// straight into the tile's LDS image (and into `demb` for the out-projection's weight gradient), beside d_pos / d_neg and the
// positives' / negatives' gradient rows: triplet_sample16, the stand-alone kernel's own arithmetic, without its launch (10.5 us
// between two 11 us chains, each at the floor of a launch of dependent round trips).  emb == NULL: no prologue.
struct LossPro { const float *emb, *pos, *neg; int B, K; float margin; float *d_pos, *d_neg, *dp, *dn, *demb; };
int launch_gemm_nt_chain(const NtArgs* args, const int* modes, int n, hipStream_t st, const HingeMeanJob* rider = nullptr,
                         const LossPro* loss = nullptr);
int launch_gemm_nt_chain_pair(const NtArgs* fwd, const NtArgs* bwd, const LossPro* loss, hipStream_t st);
#ifdef __HIPCC__
// one workgroup of >= 256 threads (the first 256 add, in the same fixed order whatever the workgroup size; the others only
// take part in the barriers)
__device__ __forceinline__ void hinge_mean_body(const HingeMeanJob& j, float* red /* [256] LDS */) {
    const int t = threadIdx.x;
    float s = 0.f;
    if (t < 256)
        for (int b = t; b < j.B; b += 256) {
            const float l = j.margin - j.d_pos[b] + j.d_neg[b];
            s += l > 0.f ? l : 0.f;
        }
    if (t < 256) red[t] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) *j.loss = red[0] / (float)j.B;
}
#endif

struct TransposeJob { const float* in; float* out; int rows, cols; };   // out[c][r] = in[r][c]
#define PC_TRANSPOSE_JOBS 8
// zero / nzero (optional): floats the launch also clears (the key-bias gradient of the absorbed attention is exactly 0;
// a separate fill would be one more launch boundary)
struct TransposeBatch { TransposeJob job[PC_TRANSPOSE_JOBS]; int n; float* zero; int nzero; };
int launch_transpose_batch(const TransposeBatch& tb, hipStream_t st);
#ifdef __HIPCC__
// one 32 x 32 tile of one job, by a workgroup of NT threads that has this launch to itself or rides in another's (tile counts
// tiles_x x tiles_y per job); t: 32 x 33 floats of LDS
template <int NT>
__device__ __forceinline__ void transpose_tile_body(const TransposeBatch& tb, int tile, int tiles_x, int tiles_y, float (*t)[33], int tid) {
    const int jz = tile / (tiles_x * tiles_y), rem = tile % (tiles_x * tiles_y);
    const TransposeJob j = tb.job[jz];
    const int tx = tid & 31, ty = tid >> 5;
    if (tb.zero && tile == 0)
        for (int i = tid; i < tb.nzero; i += NT) tb.zero[i] = 0.f;
    const int bx = (rem % tiles_x) * 32, by = (rem / tiles_x) * 32;
    if (bx >= j.cols || by >= j.rows) return;                 // (uniform over the workgroup)
    for (int i = ty; i < 32; i += NT / 32) {
        const int r = by + i, cc = bx + tx;
        t[i][tx] = (r < j.rows && cc < j.cols) ? j.in[(size_t)r * j.cols + cc] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += NT / 32) {
        const int cc = bx + i, r = by + tx;
        if (r < j.rows && cc < j.cols) j.out[(size_t)cc * j.rows + r] = t[tx][i];
    }
}
#endif
static inline void transpose_batch_tiles(const TransposeBatch& tb, int* tiles_x, int* tiles_y) {
    int mx = 0, my = 0;
    for (int i = 0; i < tb.n; i++) { mx = tb.job[i].cols > mx ? tb.job[i].cols : mx; my = tb.job[i].rows > my ? tb.job[i].rows : my; }
    *tiles_x = (mx + 31) / 32; *tiles_y = (my + 31) / 32;
}

struct TnArgs {
    // dW[No,Ni] (+)= sum_r Z[r][o] * A[r][i];  db[o] (+)= sum_r Z[r][o]
    const float* Z; int ldz;
    const float* A; int lda; const int32_t* gather;
    int R, No, Ni;
    SegInfo seg;
    int prologue; const float* pscale; const float* pshift;      // bn-tanh on A, [nseg][Ni]
    // optional BatchNorm-backward transform of Z on load (Z = dZ1, zaux = H0, all [nseg][No]):
    //   Z' = scale_s * (Z - c1_s - (zaux - mean_s) * invstd_s * c2_s)
    const float* zaux; int ldzaux;
    const float *z_mean, *z_invstd, *z_scale, *z_c1, *z_c2;
    // Z implicit: row r of Z is the one-hot vector of z_onehot[r] (< 0: a zero row; Z itself is then unused).  The
    // product is a segmented row sum, dW[t] = sum of the A rows with z_onehot[r] == t: an nn.Embedding gradient over a
    // small table, through the matrix cores instead of float atomics (few-row kernel only)
    const int32_t* z_onehot;
    float* dW; int lddw; float* db;                              // db may be null
    int accumulate;                                              // 1: +=, 0: overwrite
    float* slabs; size_t slab_floats;                            // workspace
};
// Deferred slab sums: with a TnDefer the product launches append their reduce job to the list instead of launching
// tn_reduce themselves, and ONE tn_reduce_group launch folds every weight gradient of the step (the products' slab
// regions must then be distinct and stay untouched until launch_tn_reduce_deferred).  Five reduce launches of ~9 us
// per Product2Vec step became one.
struct TnDefer;
int launch_gemm_tn(const TnArgs& a, hipStream_t st, TnDefer* defer = nullptr);
int launch_gemm_tn_halves(const TnArgs& a, float* slabs0, float* slabs1, size_t slab_floats, hipStream_t st, TnDefer* defer);
size_t gemm_tn_workspace_floats(int R, int No, int Ni);
// up to PC_TN_GROUP small independent products (disjoint slab regions) as one launch + one reduce; the reduce can
// take PC_TN_EXTRA further slab sets that other kernels filled (the joint step's type-table scatter-adds)
#define PC_TN_GROUP 10
#define PC_TN_EXTRA 2
#define PC_TN_RGROUP (PC_TN_GROUP + PC_TN_EXTRA + 4)      // + the four large FFN gradients when a step's reduces are deferred
struct TnGroup { TnArgs a[PC_TN_GROUP]; int tiles_i[PC_TN_GROUP], nsplit[PC_TN_GROUP], rps[PC_TN_GROUP], block0[PC_TN_GROUP + 1], n; };
struct TnReduceJob { const float* slabs; int nsplit, n; float* out; int accumulate; };   // out[n] (+)= sum of nsplit slabs of n floats
// torch.optim.Adam riding in the step's LAST gradient launch (round 6; pc_p2v_train_step_unique_adam): every float4 of gradient the
// slab reduce forms is followed, in the same thread, by the update of its parameter and moments (the flat buffers share offsets),
// and rider workgroups behind the reduce blocks update the ranges no reduce job produces (biases and BatchNorm parameters other
// kernels of the step finished earlier).  Same expressions as adam_at_kernel: same bits as the separate launch.
#define PC_ADAM_REST (2 * (PC_TN_GROUP + PC_TN_EXTRA + 4) + 1)
struct AdamRider {
    float *p, *m, *v; const float* g; size_t n;       // flat parameter / moment / gradient buffers (g: what the reduce jobs' outputs point into)
    int64_t* step_count; long long t; double lr, beta1, beta2; float omb1, beta2f, omb2, eps;
    int block0;                                       // reduce blocks of the launch (the rider blocks come FIRST in the grid, the reduce blocks behind them)
    int n_rest, rest_lo[PC_ADAM_REST], rest_hi[PC_ADAM_REST], rest_block0[PC_ADAM_REST + 1];      // float offsets, multiples of 4
};
struct TnReduceGroup {
    const float* slabs[PC_TN_RGROUP]; float* dW[PC_TN_RGROUP]; float* db[PC_TN_RGROUP];
    int nsplit[PC_TN_RGROUP], n_w[PC_TN_RGROUP], n_b[PC_TN_RGROUP], accumulate[PC_TN_RGROUP], block0[PC_TN_RGROUP + 1], n;
    AdamRider ad;                                     // ad.p == NULL: no optimizer rides
};
// A second queue for the few launches of the fused Product2Vec step that nothing on the main queue waits for until much
// later: the attention block's ten few-row weight gradients (24 us, needed by the slab reduce at the end of the step) beside
// the key-row gradient product and dZ2, and the BatchNorm-backward finalize (12 us, 8 workgroups, needed by dW0) beside the two
// dW3 halves.  Fork = event on the main queue + wait on the side queue; join = the reverse.  The stream and its events belong
// to the library (one set per device and main queue, created on first use); work on the side queue only ever touches workspace buffers whose
// next reader on the main queue sits behind the join.
#define PC_FORK_EVENTS 3
struct PcFork { hipStream_t side; hipEvent_t fork[PC_FORK_EVENTS]; hipEvent_t join[PC_FORK_EVENTS]; int pending; };
PcFork* pc_fork_get(hipStream_t main_st);                    // null: no side queue (creation failed): everything stays on the main queue
int pc_fork_begin(PcFork* f, int i, hipStream_t main_st);    // the side queue continues behind everything enqueued on main so far
int pc_fork_mark(PcFork* f, int i);                          // a point on the side queue ...
int pc_fork_wait(PcFork* f, int i, hipStream_t main_st);     // ... behind which main continues (the side queue may go on)
int pc_fork_join(PcFork* f, int i, hipStream_t main_st);     // mark + wait, and nothing is pending afterwards
struct TnDefer { TnReduceGroup r; int rblocks; PcFork* fork; int bn_finalized; const pc_adam_fused* adam; };
static_assert(sizeof(TnGroup) <= 4096, "kernel argument segment");
int launch_gemm_tn_group(const TnArgs* args, int n, const TnReduceJob* extra, int n_extra, hipStream_t st,
                         TnDefer* defer = nullptr);
int launch_tn_reduce_deferred(TnDefer* d, hipStream_t st);
static inline void tn_defer_init(TnDefer* d) { d->r = TnReduceGroup{}; d->rblocks = 0; d->fork = nullptr; d->bn_finalized = 0; d->adam = nullptr; }
int scatter_add_slab_blocks(int table_rows, int rows, int width);
int launch_scatter_add_slabs(const int32_t* idx, int rows, int width, int table_rows, const float* src, float* slabs,
                             hipStream_t st);
