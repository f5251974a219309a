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
__global__ __launch_bounds__(256) void attn_core_fwd_kernel(const float* q, const float* kv, int B, int N,
                                                            float* ctx, float* probs) {
    extern __shared__ float sc_all[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 4;
    const int b = blockIdx.x * 4 + w;
    if (b >= B) return;                                  // wave-uniform
    float* sc = sc_all + (size_t)w * PC_HEADS * N;
    const float scale = 0.17677669529663687f;            // 1/sqrt(32), applied to q as torch does
    float2 qv = *reinterpret_cast<const float2*>(q + (size_t)b * PC_D + 2 * lane);
    qv.x *= scale; qv.y *= scale;
    const float* kvb = kv + (size_t)b * N * (2 * PC_D);
    float m = -INFINITY;
    for (int n = 0; n < N; n++) {
        const float2 k2 = *reinterpret_cast<const float2*>(kvb + (size_t)n * (2 * PC_D) + 2 * lane);
        const float s = group16_sum(qv.x * k2.x + qv.y * k2.y);
        if ((lane & 15) == 0) sc[h * N + n] = s;
        m = fmaxf(m, s);
    }
    __builtin_amdgcn_wave_barrier();
    float sum = 0.f;
    float2 o = make_float2(0.f, 0.f);
    for (int n = 0; n < N; n++) {
        const float e = expf(sc[h * N + n] - m);
        const float2 v2 = *reinterpret_cast<const float2*>(kvb + (size_t)n * (2 * PC_D) + PC_D + 2 * lane);
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
__global__ __launch_bounds__(256) void attn_core_bwd_kernel(const float* dctx, const float* q, const float* kv,
                                                            const float* probs, int B, int N, float* dq,
                                                            float* dkv) {
    extern __shared__ float sc_all[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, h = lane >> 4;
    const int b = blockIdx.x * 4 + w;
    if (b >= B) return;
    float* dps = sc_all + (size_t)w * PC_HEADS * N;
    const float scale = 0.17677669529663687f;
    const float2 g = *reinterpret_cast<const float2*>(dctx + (size_t)b * PC_D + 2 * lane);
    float2 qs = *reinterpret_cast<const float2*>(q + (size_t)b * PC_D + 2 * lane);
    qs.x *= scale; qs.y *= scale;
    const float* kvb = kv + (size_t)b * N * (2 * PC_D);
    float* dkvb = dkv + (size_t)b * N * (2 * PC_D);
    const float* pb = probs + (size_t)b * PC_HEADS * N + h * N;
    float dsum = 0.f;                                     // sum_n p_n * dp_n  (softmax backward)
    for (int n = 0; n < N; n++) {
        const float2 v2 = *reinterpret_cast<const float2*>(kvb + (size_t)n * (2 * PC_D) + PC_D + 2 * lane);
        const float dp = group16_sum(g.x * v2.x + g.y * v2.y);
        if ((lane & 15) == 0) dps[h * N + n] = dp;
        dsum += pb[n] * dp;
    }
    __builtin_amdgcn_wave_barrier();
    float2 dqa = make_float2(0.f, 0.f);
    for (int n = 0; n < N; n++) {
        const float p = pb[n];
        const float ds = p * (dps[h * N + n] - dsum);
        const float2 k2 = *reinterpret_cast<const float2*>(kvb + (size_t)n * (2 * PC_D) + 2 * lane);
        dqa.x += ds * k2.x;
        dqa.y += ds * k2.y;
        *reinterpret_cast<float2*>(dkvb + (size_t)n * (2 * PC_D) + 2 * lane) = make_float2(ds * qs.x, ds * qs.y);
        *reinterpret_cast<float2*>(dkvb + (size_t)n * (2 * PC_D) + PC_D + 2 * lane) = make_float2(p * g.x, p * g.y);
    }
    *reinterpret_cast<float2*>(dq + (size_t)b * PC_D + 2 * lane) = make_float2(dqa.x * scale, dqa.y * scale);
}

// ---------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct AttnWs {
    float *dctx, *dq, *dkv;       // backward intermediates
    float *wot, *wqt, *wkvt;      // transposed projections
    float *slabs; size_t slab_floats;
    size_t total;
};

static AttnWs attn_ws_layout(void* base, int B, int N) {
    AttnWs w;
    size_t off = 0;
    auto take = [&](size_t floats) {
        float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
        off += align256(floats * sizeof(float));
        return p;
    };
    w.dctx = take((size_t)B * PC_D);
    w.dq = take((size_t)B * PC_D);
    w.dkv = take((size_t)B * N * 2 * PC_D);
    w.wot = take(PC_D * PC_D);
    w.wqt = take(PC_D * PC_D);
    w.wkvt = take(2 * PC_D * PC_D);
    size_t s1 = gemm_tn_workspace_floats(B * N, 2 * PC_D, PC_D);
    size_t s2 = gemm_tn_workspace_floats(B, PC_D, PC_D);
    w.slab_floats = s1 > s2 ? s1 : s2;
    w.slabs = take(w.slab_floats);
    w.total = off;
    return w;
}

extern "C" size_t pc_p2v_attention_workspace_bytes(int batch, int n_keys) {
    if (batch <= 0 || n_keys <= 0) return 0;
    return attn_ws_layout(nullptr, batch, n_keys).total;
}

static NtArgs nt_plain(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
                       int N, int K) {
    NtArgs a = {};
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.seg = make_seginfo(nullptr, M, 128);
    return a;
}

static int attn_check(const pc_p2v_tensors* p, int B, int N, const pc_attn_saved* sv, void* ws, size_t ws_bytes) {
    if (!p || !p->in_proj_w || !p->in_proj_b || !p->out_proj_w || !p->out_proj_b) return PC_EINVAL;
    if (B <= 0 || N <= 0 || !sv || !sv->q || !sv->kv || !sv->probs || !sv->ctx || !ws) return PC_EINVAL;
    if ((size_t)4 * PC_HEADS * N * sizeof(float) > 60000) return PC_ESHAPE;   // scores must fit LDS
    if (ws_bytes < pc_p2v_attention_workspace_bytes(B, N)) return PC_EWORKSPACE;
    return PC_OK;
}

extern "C" int pc_p2v_attention_forward(const pc_p2v_tensors* p, const float* query, const float* keys, int B,
                                        int N, float* out, const pc_attn_saved* sv, void* ws, size_t ws_bytes,
                                        void* stream) {
    PC_TRY(attn_check(p, B, N, sv, ws, ws_bytes));
    if (!query || !keys || !out) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // in_proj rows [0,D) = Wq, [D,3D) = [Wk;Wv]  (torch packs q,k,v in this order)
    PC_TRY(launch_gemm_nt(nt_plain(keys, PC_D, p->in_proj_w + PC_D * PC_D, PC_D, p->in_proj_b + PC_D, sv->kv,
                                   2 * PC_D, B * N, 2 * PC_D, PC_D), st));
    PC_TRY(launch_gemm_nt(nt_plain(query, PC_D, p->in_proj_w, PC_D, p->in_proj_b, sv->q, PC_D, B, PC_D, PC_D), st));
    const size_t lds = (size_t)4 * PC_HEADS * N * sizeof(float);
    PC_LAUNCH(attn_core_fwd_kernel, dim3((B + 3) / 4), dim3(256), lds, st, sv->q, sv->kv, B, N, sv->ctx,
                       sv->probs);
    PC_TRY(pc_launch_status());
    return launch_gemm_nt(nt_plain(sv->ctx, PC_D, p->out_proj_w, PC_D, p->out_proj_b, out, PC_D, B, PC_D, PC_D), st);
}

extern "C" int pc_p2v_attention_backward(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* query,
                                         const float* keys, int B, int N, const float* dout,
                                         const pc_attn_saved* sv, float* dquery, float* dkeys, int accumulate,
                                         void* ws, size_t ws_bytes, void* stream) {
    PC_TRY(attn_check(p, B, N, sv, ws, ws_bytes));
    if (!g || !g->in_proj_w || !g->in_proj_b || !g->out_proj_w || !g->out_proj_b) return PC_EINVAL;
    if (!query || !keys || !dout || !dquery || !dkeys) return PC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    AttnWs w = attn_ws_layout(ws, B, N);
    const SegInfo si1 = make_seginfo(nullptr, B, 128), si2 = make_seginfo(nullptr, B * N, 128);

    PC_TRY(launch_transpose(p->out_proj_w, PC_D, PC_D, w.wot, st));
    PC_TRY(launch_transpose(p->in_proj_w, PC_D, PC_D, w.wqt, st));
    PC_TRY(launch_transpose(p->in_proj_w + PC_D * PC_D, 2 * PC_D, PC_D, w.wkvt, st));   // [2D,D] -> [D,2D]

    // out = ctx Wo^T + bo
    PC_TRY(launch_gemm_nt(nt_plain(dout, PC_D, w.wot, PC_D, nullptr, w.dctx, PC_D, B, PC_D, PC_D), st));
    TnArgs to = {};
    to.Z = dout; to.ldz = PC_D; to.A = sv->ctx; to.lda = PC_D; to.R = B; to.No = PC_D; to.Ni = PC_D; to.seg = si1;
    to.dW = g->out_proj_w; to.lddw = PC_D; to.db = g->out_proj_b; to.accumulate = accumulate; to.slabs = w.slabs;
    to.slab_floats = w.slab_floats;
    PC_TRY(launch_gemm_tn(to, st));

    const size_t lds = (size_t)4 * PC_HEADS * N * sizeof(float);
    PC_LAUNCH(attn_core_bwd_kernel, dim3((B + 3) / 4), dim3(256), lds, st, w.dctx, sv->q, sv->kv, sv->probs,
                       B, N, w.dq, w.dkv);
    PC_TRY(pc_launch_status());

    // q = query Wq^T + bq
    PC_TRY(launch_gemm_nt(nt_plain(w.dq, PC_D, w.wqt, PC_D, nullptr, dquery, PC_D, B, PC_D, PC_D), st));
    TnArgs tq = {};
    tq.Z = w.dq; tq.ldz = PC_D; tq.A = query; tq.lda = PC_D; tq.R = B; tq.No = PC_D; tq.Ni = PC_D; tq.seg = si1;
    tq.dW = g->in_proj_w; tq.lddw = PC_D; tq.db = g->in_proj_b; tq.accumulate = accumulate; tq.slabs = w.slabs;
    tq.slab_floats = w.slab_floats;
    PC_TRY(launch_gemm_tn(tq, st));

    // [k|v] = keys [Wk;Wv]^T + [bk;bv]
    PC_TRY(launch_gemm_nt(nt_plain(w.dkv, 2 * PC_D, w.wkvt, 2 * PC_D, nullptr, dkeys, PC_D, B * N, PC_D, 2 * PC_D), st));
    TnArgs tk = {};
    tk.Z = w.dkv; tk.ldz = 2 * PC_D; tk.A = keys; tk.lda = PC_D; tk.R = B * N; tk.No = 2 * PC_D; tk.Ni = PC_D;
    tk.seg = si2;
    tk.dW = g->in_proj_w + PC_D * PC_D; tk.lddw = PC_D; tk.db = g->in_proj_b + PC_D; tk.accumulate = accumulate;
    tk.slabs = w.slabs; tk.slab_floats = w.slab_floats;
    return launch_gemm_tn(tk, st);
}
