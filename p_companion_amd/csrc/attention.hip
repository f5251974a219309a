// P7: Product2Vec.apply_attention (product2vec.py:48-68) = nn.MultiheadAttention(embed 128,
// 4 heads, batch_first) with ONE query token per sample, keys == values == the FFN
// embeddings of the (zero-padded) neighbour rows, no key-padding mask, dropout 0.
//
//   q = Wq e_a + bq ; [k|v] = [Wk;Wv] e_nbr + [bk;bv]        (gemm_nt, packed in_proj rows)
//   per head h (32 dims): p = softmax_n((q_h/sqrt(32)) . k_{n,h}) ; ctx_h = sum_n p_n v_{n,h}
//   out = Wo ctx + bo
//
// The softmax core runs one 64-lane wavefront per sample: lane l owns dims (2l, 2l+1), a
// head is a 16-lane group, scores reduce with 4 xor-shuffles, K/V rows are read as one
// coalesced 512-B segment per wave.  N is small (<= a few dozen neighbours), so the scores
// of all heads sit in LDS between the two passes.
#include "common.h"

int launch_transpose(const float* in, int rows, int cols, float* out, hipStream_t st);

#define HEAD_DIM (PC_D / PC_HEADS)

// grid: ceil(B/4) blocks of 256; dynamic LDS: 4 waves * HEADS * N floats
// slot_row (optional): key/value row of neighbour slot (b,n) inside kv -- the compact layout of
// the index path, where every zero-padded slot points at ONE shared row (identical inputs give
// identical K/V); NULL = dense layout, slot (b,n) is row b*N+n.
__global__ __launch_bounds__(256) void attn_core_fwd_kernel(const float* q, const float* kv, int B, int N,
                                                            float* ctx, float* probs, const int32_t* slot_row) {
    extern __shared__ float sc_all[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 4;
    const int b = blockIdx.x * 4 + w;
    if (b >= B) return;                                  // wave-uniform
    float* sc = sc_all + (size_t)w * PC_HEADS * N;
    const float scale = 0.17677669529663687f;            // 1/sqrt(32), applied to q as torch does
    float2 qv = *reinterpret_cast<const float2*>(q + (size_t)b * PC_D + 2 * lane);
    qv.x *= scale; qv.y *= scale;
    const int32_t* sr = slot_row ? slot_row + (size_t)b * N : nullptr;
    const size_t base = (size_t)b * N;
    // slot -> row: one coalesced load per wave, broadcast by shuffle (a per-key index load would put
    // an extra dependent memory round trip in front of every K/V row)
    const bool sr_reg = sr && N <= 64;
    const int my_row = (sr_reg && lane < N) ? sr[lane] : 0;
    auto row_of = [&](int n) -> size_t {
        return sr_reg ? (size_t)__shfl(my_row, n, 64) : (sr ? (size_t)sr[n] : base + n);
    };
    // keys are taken AU at a time: the AU row loads of a group are in flight together (one wave per sample
    // and 16 waves per CU leave the latency of a load-use-load chain exposed otherwise)
    constexpr int AU = 8;
    float m = -INFINITY;
    int n = 0;
    for (; n + AU <= N; n += AU) {
        float2 k2[AU];
#pragma unroll
        for (int u = 0; u < AU; u++) k2[u] = *reinterpret_cast<const float2*>(kv + row_of(n + u) * (2 * PC_D) + 2 * lane);
#pragma unroll
        for (int u = 0; u < AU; u++) {
            const float s = group16_sum(qv.x * k2[u].x + qv.y * k2[u].y);
            if ((lane & 15) == 0) sc[h * N + n + u] = s;
            m = fmaxf(m, s);
        }
    }
    for (; n < N; n++) {
        const float2 k2 = *reinterpret_cast<const float2*>(kv + row_of(n) * (2 * PC_D) + 2 * lane);
        const float s = group16_sum(qv.x * k2.x + qv.y * k2.y);
        if ((lane & 15) == 0) sc[h * N + n] = s;
        m = fmaxf(m, s);
    }
    __builtin_amdgcn_wave_barrier();
    float sum = 0.f;
    float2 o = make_float2(0.f, 0.f);
    for (n = 0; n + AU <= N; n += AU) {
        float2 v2[AU];
#pragma unroll
        for (int u = 0; u < AU; u++) v2[u] = *reinterpret_cast<const float2*>(kv + row_of(n + u) * (2 * PC_D) + PC_D + 2 * lane);
#pragma unroll
        for (int u = 0; u < AU; u++) {
            const float e = expf(sc[h * N + n + u] - m);
            sum += e;
            o.x += e * v2[u].x;
            o.y += e * v2[u].y;
        }
    }
    for (; n < N; n++) {
        const float e = expf(sc[h * N + n] - m);
        const float2 v2 = *reinterpret_cast<const float2*>(kv + row_of(n) * (2 * PC_D) + PC_D + 2 * lane);
        sum += e;
        o.x += e * v2.x;
        o.y += e * v2.y;
    }
    const float inv = 1.0f / sum;
    o.x *= inv; o.y *= inv;
    *reinterpret_cast<float2*>(ctx + (size_t)b * PC_D + 2 * lane) = o;
    // probabilities [B][HEADS][N]: every lane of a head group knows (m, inv) of its head
    float* pb = probs + (size_t)b * PC_HEADS * N;
    for (int n = lane & 15; n < N; n += 16) pb[h * N + n] = expf(sc[h * N + n] - m) * inv;
}

// dctx[B,D] -> dq[B,D], dkv[B*N,2D]
// With a slot map, gradients of slots that share the padding row are summed per sample into
// dkv_pad[b] (their K/V -- hence p and ds -- are identical within a sample); a column sum over
// samples then forms the shared row's gradient (fixed order, no atomics).
__global__ __launch_bounds__(256) void attn_core_bwd_kernel(const float* dctx, const float* q, const float* kv,
                                                            const float* probs, int B, int N, float* dq,
                                                            float* dkv, const int32_t* slot_row, int pad_row,
                                                            float* dkv_pad, float* coef) {
    extern __shared__ float sc_all[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 4;
    const int b = blockIdx.x * 4 + w;
    if (b >= B) return;
    float* dps = sc_all + (size_t)w * PC_HEADS * N;
    const float scale = 0.17677669529663687f;
    const float2 g = *reinterpret_cast<const float2*>(dctx + (size_t)b * PC_D + 2 * lane);
    float2 qs = *reinterpret_cast<const float2*>(q + (size_t)b * PC_D + 2 * lane);
    qs.x *= scale; qs.y *= scale;
    const int32_t* sr = slot_row ? slot_row + (size_t)b * N : nullptr;
    const size_t base = (size_t)b * N;
    const bool sr_reg = sr && N <= 64;
    const int my_row = (sr_reg && lane < N) ? sr[lane] : 0;
    auto row_of = [&](int n) -> size_t {
        return sr_reg ? (size_t)__shfl(my_row, n, 64) : (sr ? (size_t)sr[n] : base + n);
    };
    const float* pb = probs + (size_t)b * PC_HEADS * N + h * N;
    constexpr int AU = 8;                                 // rows in flight per group (see the forward kernel)
    float dsum = 0.f;                                     // sum_n p_n * dp_n  (softmax backward)
    int n = 0;
    for (; n + AU <= N; n += AU) {
        float2 v2[AU];
#pragma unroll
        for (int u = 0; u < AU; u++) v2[u] = *reinterpret_cast<const float2*>(kv + row_of(n + u) * (2 * PC_D) + PC_D + 2 * lane);
#pragma unroll
        for (int u = 0; u < AU; u++) {
            const float dp = group16_sum(g.x * v2[u].x + g.y * v2[u].y);
            if ((lane & 15) == 0) dps[h * N + n + u] = dp;
            dsum += pb[n + u] * dp;
        }
    }
    for (; n < N; n++) {
        const float2 v2 = *reinterpret_cast<const float2*>(kv + row_of(n) * (2 * PC_D) + PC_D + 2 * lane);
        const float dp = group16_sum(g.x * v2.x + g.y * v2.y);
        if ((lane & 15) == 0) dps[h * N + n] = dp;
        dsum += pb[n] * dp;
    }
    __builtin_amdgcn_wave_barrier();
    float2 dqa = make_float2(0.f, 0.f), pk = make_float2(0.f, 0.f), pv = make_float2(0.f, 0.f);
    for (n = 0; n < N; n += AU) {
        float2 kk[AU];
#pragma unroll
        for (int u = 0; u < AU; u++)
            kk[u] = n + u < N ? *reinterpret_cast<const float2*>(kv + row_of(n + u) * (2 * PC_D) + 2 * lane) : make_float2(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < AU; u++) {
        if (n + u >= N) break;
        const float p = pb[n + u];
        const float ds = p * (dps[h * N + n + u] - dsum);
        const size_t row = row_of(n + u);
        const float2 k2 = kk[u];
        dqa.x += ds * k2.x;
        dqa.y += ds * k2.y;
        if (sr && (int)row == pad_row) {                   // wave-uniform: the slot map is per sample
            pk.x += ds * qs.x; pk.y += ds * qs.y;
            pv.x += p * g.x;   pv.y += p * g.y;
        } else if (coef) {
            // unique-neighbour layout: rows are shared between samples, so dK/dV are formed row by row afterwards
            // (attn_dkv_rows_kernel) from the per-slot softmax coefficients: [slot][ds per head | p per head]
            if ((lane & 15) == 0) {
                coef[(base + n + u) * 8 + h] = ds;
                coef[(base + n + u) * 8 + 4 + h] = p;
            }
        } else {
            *reinterpret_cast<float2*>(dkv + row * (2 * PC_D) + 2 * lane) = make_float2(ds * qs.x, ds * qs.y);
            *reinterpret_cast<float2*>(dkv + row * (2 * PC_D) + PC_D + 2 * lane) = make_float2(p * g.x, p * g.y);
        }
        }
    }
    if (dkv_pad) {
        *reinterpret_cast<float2*>(dkv_pad + (size_t)b * (2 * PC_D) + 2 * lane) = pk;
        *reinterpret_cast<float2*>(dkv_pad + (size_t)b * (2 * PC_D) + PC_D + 2 * lane) = pv;
    }
    *reinterpret_cast<float2*>(dq + (size_t)b * PC_D + 2 * lane) = make_float2(dqa.x * scale, dqa.y * scale);
}

// dK_r = sum over the slots (b,n) of row r of ds[b,h,n] * q_b / sqrt(32),  dV_r = sum p[b,h,n] * dctx_b : one wave
// per row; q / dctx rows of the batch are L2-resident.  The row's slot list arrives in arbitrary order: up to 64
// entries are sorted in the wave (bitonic, by shuffles) so the fp32 summation order is fixed; a longer list (a
// product that is a neighbour of more than 64 anchors of the batch) is accumulated in fp64, where the order no
// longer reaches the rounded fp32 result.
__global__ __launch_bounds__(256) void attn_dkv_rows_kernel(const float* coef, const float* q, const float* dctx,
                                                            const int32_t* ref_off, const int32_t* ref_slot, int N,
                                                            int rows, float* dkv) {
    const int lane = threadIdx.x & 63, h = lane >> 4;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float scale = 0.17677669529663687f;
    const int lo = ref_off[r], k = ref_off[r + 1] - lo;
    float2 dk = make_float2(0.f, 0.f), dv = make_float2(0.f, 0.f);
    if (k <= 64) {
        int mine = lane < k ? ref_slot[lo + lane] : 0x7fffffff;
        if (k > 1) {
#pragma unroll
            for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
                for (int stride = size >> 1; stride >= 1; stride >>= 1) {
                    const int other = __shfl_xor(mine, stride, 64);
                    const bool up = (lane & size) == 0, low = (lane & stride) == 0;
                    mine = (up == low) ? min(mine, other) : max(mine, other);
                }
        }
        for (int i = 0; i < k; i += 4) {
            float ds[4], p[4];
            float2 qq[4], gg[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int slot = __shfl(mine, min(i + u, k - 1), 64);
                const int b = slot / N;
                ds[u] = coef[(size_t)slot * 8 + h];
                p[u] = coef[(size_t)slot * 8 + 4 + h];
                qq[u] = *reinterpret_cast<const float2*>(q + (size_t)b * PC_D + 2 * lane);
                gg[u] = *reinterpret_cast<const float2*>(dctx + (size_t)b * PC_D + 2 * lane);
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (i + u < k) {
                    dk.x += ds[u] * (qq[u].x * scale); dk.y += ds[u] * (qq[u].y * scale);
                    dv.x += p[u] * gg[u].x;            dv.y += p[u] * gg[u].y;
                }
        }
    } else {
        double kx = 0.0, ky = 0.0, vx = 0.0, vy = 0.0;
        for (int i = 0; i < k; i++) {
            const int slot = ref_slot[lo + i];
            const int b = slot / N;
            const float ds = coef[(size_t)slot * 8 + h], p = coef[(size_t)slot * 8 + 4 + h];
            const float2 qq = *reinterpret_cast<const float2*>(q + (size_t)b * PC_D + 2 * lane);
            const float2 gg = *reinterpret_cast<const float2*>(dctx + (size_t)b * PC_D + 2 * lane);
            kx += (double)ds * (double)(qq.x * scale); ky += (double)ds * (double)(qq.y * scale);
            vx += (double)p * (double)gg.x;            vy += (double)p * (double)gg.y;
        }
        dk = make_float2((float)kx, (float)ky);
        dv = make_float2((float)vx, (float)vy);
    }
    *reinterpret_cast<float2*>(dkv + (size_t)r * (2 * PC_D) + 2 * lane) = dk;
    *reinterpret_cast<float2*>(dkv + (size_t)r * (2 * PC_D) + PC_D + 2 * lane) = dv;
}

// out[c] = sum_b x[b][c] over B rows of 2D = 256 columns, in two fixed-order stages: COLSUM_CHUNKS
// workgroups each fold a contiguous slice of rows (thread = column, coalesced 1-KB rows), then one
// workgroup folds the partials.
#define COLSUM_CHUNKS 64
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* x, int B, float* part) {
    const int c = threadIdx.x;
    const int per = (B + COLSUM_CHUNKS - 1) / COLSUM_CHUNKS;
    const int b0 = blockIdx.x * per, b1 = min(B, b0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0;
    for (; b + 3 < b1; b += 4) {
        s0 += x[(size_t)b * 256 + c]; s1 += x[(size_t)(b + 1) * 256 + c];
        s2 += x[(size_t)(b + 2) * 256 + c]; s3 += x[(size_t)(b + 3) * 256 + c];
    }
    for (; b < b1; b++) s0 += x[(size_t)b * 256 + c];
    part[blockIdx.x * 256 + c] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* part, float* out) {
    const int c = threadIdx.x;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < COLSUM_CHUNKS; i++) s += part[i * 256 + c];
    out[c] = s;
}

// ---------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct AttnWs {
    float *dctx, *dq, *dkv, *dkv_pad, *pad_part;   // backward intermediates
    float *coef;                                    // [B*N][8] per-slot softmax coefficients (unique-neighbour layout)
    float *wot, *wqt, *wkvt;      // transposed projections
    float *slabs; size_t slab_floats;
    float *slabs_o; size_t slab_o_floats;              // out_proj gradient: grouped with the q-projection gradient
    size_t total;
};

static AttnWs attn_ws_layout(void* base, int B, int key_rows, int n_slots_per_sample) {
    AttnWs w;
    size_t off = 0;
    auto take = [&](size_t floats) {
        float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
        off += align256(floats * sizeof(float));
        return p;
    };
    w.dctx = take((size_t)B * PC_D);
    w.dq = take((size_t)B * PC_D);
    w.dkv = take((size_t)key_rows * 2 * PC_D);
    w.dkv_pad = take((size_t)B * 2 * PC_D);
    w.pad_part = take((size_t)COLSUM_CHUNKS * 2 * PC_D);
    w.coef = take((size_t)B * n_slots_per_sample * 8);
    w.wot = take(PC_D * PC_D);
    w.wqt = take(PC_D * PC_D);
    w.wkvt = take(2 * PC_D * PC_D);
    size_t s1 = gemm_tn_workspace_floats(key_rows, 2 * PC_D, PC_D);
    size_t s2 = gemm_tn_workspace_floats(B, PC_D, PC_D);
    w.slab_floats = s1 > s2 ? s1 : s2;
    w.slabs = take(w.slab_floats);
    w.slab_o_floats = gemm_tn_workspace_floats(B, PC_D, PC_D);
    w.slabs_o = take(w.slab_o_floats);
    w.total = off;
    return w;
}

extern "C" size_t pc_p2v_attention_workspace_bytes(int batch, int n_keys) {
    if (batch <= 0 || n_keys <= 0) return 0;
    return attn_ws_layout(nullptr, batch, batch * n_keys, n_keys).total;     // dense layout bounds the compact one
}

static NtArgs nt_plain(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
                       int N, int K) {
    NtArgs a = {};
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.seg = make_seginfo(nullptr, M, 128);
    return a;
}

static int attn_check(const pc_p2v_tensors* p, int B, int N, int key_rows, const int32_t* slot_row,
                      const pc_attn_saved* sv, void* ws, size_t ws_bytes) {
    if (key_rows <= 0 || key_rows > B * N + 1 || (!slot_row && key_rows != B * N)) return PC_EINVAL;
    if (!p || !p->in_proj_w || !p->in_proj_b || !p->out_proj_w || !p->out_proj_b) return PC_EINVAL;
    if (B <= 0 || N <= 0 || !sv || !sv->q || !sv->kv || !sv->probs || !sv->ctx || !ws) return PC_EINVAL;
    if ((size_t)4 * PC_HEADS * N * sizeof(float) > 60000) return PC_ESHAPE;   // scores must fit LDS
    if (ws_bytes < pc_p2v_attention_workspace_bytes(B, N)) return PC_EWORKSPACE;
    return PC_OK;
}

// Compact form used by the fused step: keys[key_rows,D] holds each distinct neighbour row once and
// slot_row[B*N] maps slot (b,n) to its row (all padding slots -> one shared row).
int attention_forward_impl(const pc_p2v_tensors* p, const float* query, const float* keys, int B, int N,
                           int key_rows, const int32_t* slot_row, float* out, const pc_attn_saved* sv, void* ws,
                           size_t ws_bytes, void* stream) {
    PC_TRY(attn_check(p, B, N, key_rows, slot_row, sv, ws, ws_bytes));
    if (!query || !keys || !out) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // in_proj rows [0,D) = Wq, [D,3D) = [Wk;Wv]  (torch packs q,k,v in this order)
    PC_TRY(launch_gemm_nt(nt_plain(keys, PC_D, p->in_proj_w + PC_D * PC_D, PC_D, p->in_proj_b + PC_D, sv->kv,
                                   2 * PC_D, key_rows, 2 * PC_D, PC_D), st));
    PC_TRY(launch_gemm_nt(nt_plain(query, PC_D, p->in_proj_w, PC_D, p->in_proj_b, sv->q, PC_D, B, PC_D, PC_D), st));
    const size_t lds = (size_t)4 * PC_HEADS * N * sizeof(float);
    PC_LAUNCH(attn_core_fwd_kernel, dim3((B + 3) / 4), dim3(256), lds, st, sv->q, sv->kv, B, N, sv->ctx,
                       sv->probs, slot_row);
    PC_TRY(pc_launch_status());
    return launch_gemm_nt(nt_plain(sv->ctx, PC_D, p->out_proj_w, PC_D, p->out_proj_b, out, PC_D, B, PC_D, PC_D), st);
}

extern "C" int pc_p2v_attention_forward(const pc_p2v_tensors* p, const float* query, const float* keys, int B,
                                        int N, float* out, const pc_attn_saved* sv, void* ws, size_t ws_bytes,
                                        void* stream) {
    return attention_forward_impl(p, query, keys, B, N, B * N, nullptr, out, sv, ws, ws_bytes, stream);
}

// ref_off / ref_slot (optional, unique-neighbour layout): the slots of every key row; dK/dV are then formed row
// by row in a second pass from per-slot softmax coefficients (fixed summation order, no atomics)
int attention_backward_impl(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* query, const float* keys,
                            int B, int N, int key_rows, const int32_t* slot_row, int pad_row, const float* dout,
                            const pc_attn_saved* sv, float* dquery, float* dkeys, int accumulate, void* ws,
                            size_t ws_bytes, void* stream, const int32_t* ref_off, const int32_t* ref_slot) {
    PC_TRY(attn_check(p, B, N, key_rows, slot_row, sv, ws, ws_bytes));
    if (!g || !g->in_proj_w || !g->in_proj_b || !g->out_proj_w || !g->out_proj_b) return PC_EINVAL;
    if (!query || !keys || !dout || !dquery || !dkeys) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    AttnWs w = attn_ws_layout(ws, B, slot_row ? key_rows : B * N, N);
    const SegInfo si1 = make_seginfo(nullptr, B, 128), si2 = make_seginfo(nullptr, key_rows, 128);

    TransposeBatch tb = {};
    tb.job[0] = {p->out_proj_w, w.wot, PC_D, PC_D};
    tb.job[1] = {p->in_proj_w, w.wqt, PC_D, PC_D};
    tb.job[2] = {p->in_proj_w + PC_D * PC_D, w.wkvt, 2 * PC_D, PC_D};   // [2D,D] -> [D,2D]
    tb.n = 3;
    PC_TRY(launch_transpose_batch(tb, st));

    // out = ctx Wo^T + bo
    PC_TRY(launch_gemm_nt(nt_plain(dout, PC_D, w.wot, PC_D, nullptr, w.dctx, PC_D, B, PC_D, PC_D), st));
    TnArgs to = {};
    to.Z = dout; to.ldz = PC_D; to.A = sv->ctx; to.lda = PC_D; to.R = B; to.No = PC_D; to.Ni = PC_D; to.seg = si1;
    to.dW = g->out_proj_w; to.lddw = PC_D; to.db = g->out_proj_b; to.accumulate = accumulate; to.slabs = w.slabs_o;
    to.slab_floats = w.slab_o_floats;                  // (launched below, together with the q-projection gradient)

    const size_t lds = (size_t)4 * PC_HEADS * N * sizeof(float);
    const bool has_pad = slot_row && pad_row >= 0;
    const bool by_rows = ref_off && ref_slot && slot_row;
    PC_LAUNCH(attn_core_bwd_kernel, dim3((B + 3) / 4), dim3(256), lds, st, w.dctx, sv->q, sv->kv, sv->probs,
                       B, N, w.dq, w.dkv, slot_row, has_pad ? pad_row : -1, has_pad ? w.dkv_pad : nullptr,
                       by_rows ? w.coef : nullptr);
    PC_TRY(pc_launch_status());
    if (by_rows) {
        const int real_rows = has_pad ? key_rows - 1 : key_rows;
        if (real_rows > 0) {
            PC_LAUNCH(attn_dkv_rows_kernel, dim3((real_rows + 3) / 4), dim3(256), 0, st, w.coef, sv->q, w.dctx, ref_off,
                      ref_slot, N, real_rows, w.dkv);
            PC_TRY(pc_launch_status());
        }
    }
    if (has_pad) {
        PC_LAUNCH(colsum_partial_kernel, dim3(COLSUM_CHUNKS), dim3(256), 0, st, w.dkv_pad, B, w.pad_part);
        PC_TRY(pc_launch_status());
        PC_LAUNCH(colsum_final_kernel, dim3(1), dim3(256), 0, st, w.pad_part, w.dkv + (size_t)pad_row * 2 * PC_D);
        PC_TRY(pc_launch_status());
    }

    // q = query Wq^T + bq
    PC_TRY(launch_gemm_nt(nt_plain(w.dq, PC_D, w.wqt, PC_D, nullptr, dquery, PC_D, B, PC_D, PC_D), st));
    TnArgs tq = {};
    tq.Z = w.dq; tq.ldz = PC_D; tq.A = query; tq.lda = PC_D; tq.R = B; tq.No = PC_D; tq.Ni = PC_D; tq.seg = si1;
    tq.dW = g->in_proj_w; tq.lddw = PC_D; tq.db = g->in_proj_b; tq.accumulate = accumulate; tq.slabs = w.slabs;
    tq.slab_floats = w.slab_floats;
    {   // two few-row products (M = batch), each ~13 us of latency on a quarter of the chip: one launch, one reduce
        const TnArgs pair[2] = {to, tq};
        PC_TRY(launch_gemm_tn_group(pair, 2, nullptr, 0, st));
    }

    // [k|v] = keys [Wk;Wv]^T + [bk;bv]
    PC_TRY(launch_gemm_nt(nt_plain(w.dkv, 2 * PC_D, w.wkvt, 2 * PC_D, nullptr, dkeys, PC_D, key_rows, PC_D, 2 * PC_D), st));
    TnArgs tk = {};
    tk.Z = w.dkv; tk.ldz = 2 * PC_D; tk.A = keys; tk.lda = PC_D; tk.R = key_rows; tk.No = 2 * PC_D; tk.Ni = PC_D;
    tk.seg = si2;
    tk.dW = g->in_proj_w + PC_D * PC_D; tk.lddw = PC_D; tk.db = g->in_proj_b + PC_D; tk.accumulate = accumulate;
    tk.slabs = w.slabs; tk.slab_floats = w.slab_floats;
    return launch_gemm_tn(tk, st);
}

extern "C" int pc_p2v_attention_backward(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* query,
                                         const float* keys, int B, int N, const float* dout,
                                         const pc_attn_saved* sv, float* dquery, float* dkeys, int accumulate,
                                         void* ws, size_t ws_bytes, void* stream) {
    return attention_backward_impl(p, g, query, keys, B, N, B * N, nullptr, -1, dout, sv, dquery, dkeys, accumulate, ws,
                                   ws_bytes, stream, nullptr, nullptr);
}
