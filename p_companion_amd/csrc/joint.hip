// P-Companion joint step: PCompanion.forward and
// compute_loss (p_companion.py:45-119), ComplementaryTypeTransition (type_transition.py:15-20),
// ComplementaryItemPrediction (item_prediction.py:22-40), and their backward.
//
// The dense products go through the shared fp32-MFMA kernels (gemm_nt / gemm_tn) with the
// nn.Embedding lookups fused into the loaders as row gathers; the pieces the reference
// leaves to generic ATen ops are small wave-per-row kernels here: top-k over the type
// similarities, the Hadamard item projection with both distance norms and the hinge, and
// the two-column type hinge whose gradient stays SPARSE (the reference materialises a dense
// [B,T] zero gradient and a dense [T,B]x[B,L] product for it).
#include "common.h"

int launch_transpose(const float* in, int rows, int cols, float* out, hipStream_t st);
int launch_dropout(const float* x, size_t n, const pc_dropout& d, unsigned stream_id, float* y, hipStream_t st);
extern "C" int pc_scatter_add_rows_small(float* table, int table_rows, const int32_t* idx, int rows, int width,
                                         const float* src, void* stream);

#define JMAX_K 8   /* top-k capacity of topk_rows_kernel (NUM_COMP_TYPES = 3 in config.py:24) */
#define LH (PC_L / 2)

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static NtArgs nt_plain(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
                       int N, int K) {
    NtArgs a = {};
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.seg = make_seginfo(nullptr, M, 128);
    return a;
}

// ---------------------------------------------------------------------------------------
// Generic nn.Linear pieces (exported: the Python modules build their autograd on these).
//   y = act(x W^T + b), x row r = idx ? x[idx[r]] : x[r];  act: 0 none, 1 tanh, 2 relu
extern "C" int pc_linear_forward(const float* x, const int32_t* idx, int rows, int in_dim, const float* w,
                                 const float* b, int out_dim, int act, float* y, void* stream) {
    if (!x || !w || !y || rows <= 0 || in_dim <= 0 || out_dim <= 0) return PC_EINVAL;
    if (act < 0 || act > 2) return PC_EINVAL;
    NtArgs a = nt_plain(x, in_dim, w, in_dim, b, y, out_dim, rows, out_dim, in_dim);
    a.gather = idx;
    a.epilogue = act == 1 ? NT_EPI_TANH : act == 2 ? NT_EPI_RELU : NT_EPI_NONE;
    return launch_gemm_nt(a, (hipStream_t)stream);
}

// dx = (dy W) * act'(y):  wt = W^T [in_dim,out_dim] scratch (in_dim*out_dim floats)
extern "C" int pc_linear_backward_input(const float* dy, int rows, int out_dim, const float* w, int in_dim,
                                        int act, const float* y_saved, float* dx, float* wt_scratch,
                                        void* stream) {
    if (!dy || !w || !dx || !wt_scratch || rows <= 0 || in_dim <= 0 || out_dim <= 0) return PC_EINVAL;
    if (act != 0) return PC_ESHAPE;    // activation derivative is applied by the caller on dy
    (void)y_saved;
    hipStream_t st = (hipStream_t)stream;
    PC_TRY(launch_transpose(w, out_dim, in_dim, wt_scratch, st));
    return launch_gemm_nt(nt_plain(dy, out_dim, wt_scratch, out_dim, nullptr, dx, in_dim, rows, in_dim, out_dim), st);
}

extern "C" size_t pc_linear_backward_weight_workspace_bytes(int rows, int out_dim, int in_dim) {
    if (rows <= 0 || out_dim <= 0 || in_dim <= 0) return 0;
    return gemm_tn_workspace_floats(rows, out_dim, in_dim) * sizeof(float);
}

// dW[out,in] (+)= dy^T x, db (+)= sum_r dy
extern "C" int pc_linear_backward_weight(const float* dy, int rows, int out_dim, const float* x,
                                         const int32_t* idx, int in_dim, float* dw, float* db, int accumulate,
                                         void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !dw || !ws || rows <= 0) return PC_EINVAL;
    TnArgs t = {};
    t.Z = dy; t.ldz = out_dim; t.A = x; t.lda = in_dim; t.gather = idx; t.R = rows; t.No = out_dim; t.Ni = in_dim;
    t.seg = make_seginfo(nullptr, rows, 128);
    t.dW = dw; t.lddw = in_dim; t.db = db; t.accumulate = accumulate;
    t.slabs = (float*)ws; t.slab_floats = ws_bytes / sizeof(float);
    return launch_gemm_tn(t, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------
// torch.topk(sims, k, dim=1) (p_companion.py:64): one wavefront per row.  Each lane keeps
// the k best of its strided columns (sorted, descending; ties keep the lower index), then k
// rounds of a wave-wide arg-max pop the global winners.
// KC: compile-time K (1..4: the insertion is a fully unrolled chain of KC compare-exchanges; 0: run-time K up to JMAX_K).
// Inside a lane the columns arrive in increasing order, so an element never displaces an EQUAL earlier one: the
// in-lane test is a plain `>` (the cross-lane merge below keeps the full tie rule).  Rows whose length and address
// allow it are read as float4 (1 KB per wave instruction).  At T = 34800 (config.py:27) the generic form was
// VALU-bound: 8 predicated compare-exchanges per element, 444 us for 4096 rows against 90 us of HBM time.
template <int KC>
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* sims, int B, int T, int K, int32_t* idx_out,
                                                        float* val_out) {
    constexpr int KM = KC > 0 ? KC : JMAX_K;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    if (KC > 0) K = KC;
    const float* row = sims + (size_t)b * T;
    float v[KM];
    int ix[KM];
#pragma unroll
    for (int j = 0; j < KM; j++) { v[j] = -INFINITY; ix[j] = 0x7fffffff; }
    auto push = [&](float x, int xi) {
        if (KC > 0 && !(x > v[KM - 1]) && ix[KM - 1] != 0x7fffffff) return;    // (most elements: below the lane's K-th best)
#pragma unroll
        for (int j = 0; j < KM; j++) {
            if (KC > 0 || j < K) {
                const bool better = x > v[j] || ix[j] == 0x7fffffff;      // (an empty slot takes anything, -inf included)
                const float tv = better ? v[j] : x;
                const int ti = better ? ix[j] : xi;
                v[j] = better ? x : v[j];
                ix[j] = better ? xi : ix[j];
                x = tv; xi = ti;
            }
        }
    };
    int t0 = 0;
    if ((T & 3) == 0 && (((uintptr_t)row) & 15) == 0) {
        const int t4 = T >> 2;
        for (int q = lane; q < t4; q += 64) {
            const float4 x = *reinterpret_cast<const float4*>(row + 4 * q);
            push(x.x, 4 * q); push(x.y, 4 * q + 1); push(x.z, 4 * q + 2); push(x.w, 4 * q + 3);
        }
        t0 = T;
    }
    for (int t = t0 + lane; t < T; t += 64) push(row[t], t);
    for (int r = 0; r < K; r++) {
        float bv = v[0];
        int bi = ix[0];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (ix[0] == bi) {            // the owning lane pops its head
#pragma unroll
            for (int j = 0; j < KM - 1; j++) { v[j] = v[j + 1]; ix[j] = ix[j + 1]; }
            v[KM - 1] = -INFINITY; ix[KM - 1] = 0x7fffffff;
        }
        if (lane == 0) {
            idx_out[(size_t)b * K + r] = bi;
            if (val_out) val_out[(size_t)b * K + r] = bv;
        }
    }
}

extern "C" int pc_topk_rows(const float* sims, int batch, int num_types, int k, int32_t* idx_out, float* val_out,
                            void* stream) {
    if (!sims || !idx_out || batch <= 0 || num_types <= 0) return PC_EINVAL;
    if (k < 1 || k > JMAX_K || k > num_types) return PC_ESHAPE;
    const dim3 grid((batch + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (k) {
        case 1: PC_LAUNCH(topk_rows_kernel<1>, grid, block, 0, st, sims, batch, num_types, k, idx_out, val_out); break;
        case 2: PC_LAUNCH(topk_rows_kernel<2>, grid, block, 0, st, sims, batch, num_types, k, idx_out, val_out); break;
        case 3: PC_LAUNCH(topk_rows_kernel<3>, grid, block, 0, st, sims, batch, num_types, k, idx_out, val_out); break;
        case 4: PC_LAUNCH(topk_rows_kernel<4>, grid, block, 0, st, sims, batch, num_types, k, idx_out, val_out); break;
        default: PC_LAUNCH(topk_rows_kernel<0>, grid, block, 0, st, sims, batch, num_types, k, idx_out, val_out); break;
    }
    return pc_launch_status();
}

// proj[b,k,:] = pi[b,:] * tp[b*K+k,:]   (item_prediction.py:38)
__global__ void hadamard_fwd_kernel(const float* pi, const float* tp, int B, int K, int D, float* proj) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index
    const size_t total = (size_t)B * K * (D / 4);
    if (t >= total) return;
    const size_t row = t / (D / 4);
    const int c = (int)(t % (D / 4));
    const float4 a = *reinterpret_cast<const float4*>(pi + (row / K) * D + c * 4);
    const float4 x = *reinterpret_cast<const float4*>(tp + t * 4);
    *reinterpret_cast<float4*>(proj + t * 4) = make_float4(a.x * x.x, a.y * x.y, a.z * x.z, a.w * x.w);
}

// dpi[b] = sum_k dproj[b,k]*tp[b,k] ; dtp[b,k] = dproj[b,k]*pi[b] : one wave per sample
// (PRODUCT_EMB_DIM = 128 or 256: a lane owns dims (2 lane, 2 lane + 1) of every 128-dim chunk, NCH = D / 128 chunks)
template <int NCH>
__global__ __launch_bounds__(256) void hadamard_bwd_kernel(const float* dproj, const float* pi, const float* tp,
                                                           int B, int K, float* dpi, float* dtp) {
    constexpr int D = 128 * NCH;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float2 a[NCH], acc[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        a[ch] = *reinterpret_cast<const float2*>(pi + (size_t)b * D + 128 * ch + 2 * lane);
        acc[ch] = make_float2(0.f, 0.f);
    }
    for (int k = 0; k < K; k++) {
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            const size_t o = ((size_t)b * K + k) * D + 128 * ch + 2 * lane;
            const float2 g = *reinterpret_cast<const float2*>(dproj + o);
            const float2 x = *reinterpret_cast<const float2*>(tp + o);
            acc[ch].x += g.x * x.x; acc[ch].y += g.y * x.y;
            *reinterpret_cast<float2*>(dtp + o) = make_float2(g.x * a[ch].x, g.y * a[ch].y);
        }
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) *reinterpret_cast<float2*>(dpi + (size_t)b * D + 128 * ch + 2 * lane) = acc[ch];
}

extern "C" int pc_hadamard_forward_dim(const float* pi, const float* tp, int batch, int k, int dim, float* proj, void* stream) {
    if (!pi || !tp || !proj || batch <= 0 || k <= 0) return PC_EINVAL;
    if (dim != 128 && dim != 256) return PC_ESHAPE;
    const size_t total = (size_t)batch * k * (dim / 4);
    PC_LAUNCH(hadamard_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       pi, tp, batch, k, dim, proj);
    return pc_launch_status();
}
extern "C" int pc_hadamard_forward(const float* pi, const float* tp, int batch, int k, float* proj, void* stream) {
    return pc_hadamard_forward_dim(pi, tp, batch, k, PC_D, proj, stream);
}

extern "C" int pc_hadamard_backward_dim(const float* dproj, const float* pi, const float* tp, int batch, int k, int dim,
                                        float* dpi, float* dtp, void* stream) {
    if (!dproj || !pi || !tp || !dpi || !dtp || batch <= 0 || k <= 0) return PC_EINVAL;
    if (dim != 128 && dim != 256) return PC_ESHAPE;
    if (dim == 128) PC_LAUNCH(hadamard_bwd_kernel<1>, dim3((batch + 3) / 4), dim3(256), 0, (hipStream_t)stream, dproj, pi, tp,
                              batch, k, dpi, dtp);
    else PC_LAUNCH(hadamard_bwd_kernel<2>, dim3((batch + 3) / 4), dim3(256), 0, (hipStream_t)stream, dproj, pi, tp,
                   batch, k, dpi, dtp);
    return pc_launch_status();
}
extern "C" int pc_hadamard_backward(const float* dproj, const float* pi, const float* tp, int batch, int k,
                                    float* dpi, float* dtp, void* stream) {
    return pc_hadamard_backward_dim(dproj, pi, tp, batch, k, PC_D, dpi, dtp, stream);
}

// ---------------------------------------------------------------------------------------
// J6 (p_companion.py:79-119), one wave per sample, forward + backward:
//   type_b = clamp(margin - S[b,pos] + S[b,neg], 0)
//   item_bk = clamp(margin - ||proj_bk - pos_item_b|| + ||proj_bk - neg_item_b||, 0)   (torch.norm: no eps)
//   loss = alpha * mean_{b,k} item + (1-alpha) * mean_b type
template <int NCH>
__global__ __launch_bounds__(256) void joint_loss_kernel(const float* sims, const float* proj,
                                                         const int32_t* pos_t, const int32_t* neg_t,
                                                         const float* pos_items, const float* neg_items, int B,
                                                         int T, int K, float margin, float alpha, float* part_type,
                                                         float* part_item, float* dsims_val, float* dproj) {
    constexpr int D = 128 * NCH;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float lt = margin - sims[(size_t)b * T + pos_t[b]] + sims[(size_t)b * T + neg_t[b]];
    if (lane == 0) {
        part_type[b] = lt > 0.f ? lt : 0.f;
        if (dsims_val) {
            const float g = lt > 0.f ? (1.0f - alpha) / (float)B : 0.f;
            dsims_val[2 * b] = -g;
            dsims_val[2 * b + 1] = g;
        }
    }
    float2 pp[NCH], nn[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        pp[ch] = *reinterpret_cast<const float2*>(pos_items + (size_t)b * D + 128 * ch + 2 * lane);
        nn[ch] = *reinterpret_cast<const float2*>(neg_items + (size_t)b * D + 128 * ch + 2 * lane);
    }
    float li = 0.f;
    for (int k = 0; k < K; k++) {
        float2 dp[NCH], dn[NCH];
        float sp_ = 0.f, sn_ = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            const float2 x = *reinterpret_cast<const float2*>(proj + ((size_t)b * K + k) * D + 128 * ch + 2 * lane);
            dp[ch] = make_float2(x.x - pp[ch].x, x.y - pp[ch].y); dn[ch] = make_float2(x.x - nn[ch].x, x.y - nn[ch].y);
            sp_ += dp[ch].x * dp[ch].x + dp[ch].y * dp[ch].y; sn_ += dn[ch].x * dn[ch].x + dn[ch].y * dn[ch].y;
        }
        const float np_ = sqrtf(wave_sum(sp_));
        const float nn_ = sqrtf(wave_sum(sn_));
        const float l = margin - np_ + nn_;
        li += l > 0.f ? l : 0.f;
        if (dproj) {
            const float g = l > 0.f ? alpha / ((float)B * (float)K) : 0.f;
            const float ip = np_ > 0.f ? g / np_ : 0.f, in = nn_ > 0.f ? g / nn_ : 0.f;   // torch.norm: subgradient 0 at 0
#pragma unroll
            for (int ch = 0; ch < NCH; ch++)
                *reinterpret_cast<float2*>(dproj + ((size_t)b * K + k) * D + 128 * ch + 2 * lane) =
                    make_float2(-dp[ch].x * ip + dn[ch].x * in, -dp[ch].y * ip + dn[ch].y * in);
        }
    }
    if (lane == 0) part_item[b] = li;
}

// The fused training step's version of the four per-sample kernels between the projections and the dX chain --
// proj = pi * tp (item_prediction.py:38), both hinges (p_companion.py:95-119), d(proj) -> d(pi), d(tp), and the sparse
// type-hinge backward (dc, dE_c rows) -- as ONE pass, one wave per sample: the gradient scales 1/B and 1/(B K) are
// constants, so nothing here waits for the mean.  proj itself is never written (nothing else reads it in this step).
__global__ __launch_bounds__(256) void joint_rowwise_kernel(const float* sims, const float* pi, const float* tp,
                                                            const int32_t* pos_t, const int32_t* neg_t,
                                                            const float* pos_items, const float* neg_items,
                                                            const float* c, const float* ec, int B, int T, int K,
                                                            float margin, float alpha, float* part_type,
                                                            float* part_item, float* dpi, float* dtp, float* dc,
                                                            float* dec) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int p = pos_t[b], n = neg_t[b];
    const float lt = margin - sims[(size_t)b * T + p] + sims[(size_t)b * T + n];
    const float gt = lt > 0.f ? (1.0f - alpha) / (float)B : 0.f;
    const float v0 = -gt, v1 = gt;
    if (lane == 0) part_type[b] = lt > 0.f ? lt : 0.f;
    const float cb = c[(size_t)b * PC_L + lane];
    dc[(size_t)b * PC_L + lane] = v0 * ec[(size_t)p * PC_L + lane] + v1 * ec[(size_t)n * PC_L + lane];
    if (gt != 0.f) {
        unsafeAtomicAdd(dec + (size_t)p * PC_L + lane, v0 * cb);
        unsafeAtomicAdd(dec + (size_t)n * PC_L + lane, v1 * cb);
    }
    const float2 a = *reinterpret_cast<const float2*>(pi + (size_t)b * PC_D + 2 * lane);
    const float2 pp = *reinterpret_cast<const float2*>(pos_items + (size_t)b * PC_D + 2 * lane);
    const float2 nn = *reinterpret_cast<const float2*>(neg_items + (size_t)b * PC_D + 2 * lane);
    float li = 0.f;
    float2 acc = make_float2(0.f, 0.f);
    for (int k = 0; k < K; k++) {
        const size_t o = ((size_t)b * K + k) * PC_D + 2 * lane;
        const float2 t = *reinterpret_cast<const float2*>(tp + o);
        const float2 x = make_float2(a.x * t.x, a.y * t.y);
        const float2 dp = make_float2(x.x - pp.x, x.y - pp.y), dn = make_float2(x.x - nn.x, x.y - nn.y);
        const float np_ = sqrtf(wave_sum(dp.x * dp.x + dp.y * dp.y));
        const float nn_ = sqrtf(wave_sum(dn.x * dn.x + dn.y * dn.y));
        const float l = margin - np_ + nn_;
        li += l > 0.f ? l : 0.f;
        const float g = l > 0.f ? alpha / ((float)B * (float)K) : 0.f;
        const float ip = np_ > 0.f ? g / np_ : 0.f, in = nn_ > 0.f ? g / nn_ : 0.f;   // torch.norm: subgradient 0 at 0
        const float2 d = make_float2(-dp.x * ip + dn.x * in, -dp.y * ip + dn.y * in);
        acc.x += d.x * t.x; acc.y += d.y * t.y;
        *reinterpret_cast<float2*>(dtp + o) = make_float2(d.x * a.x, d.y * a.y);
    }
    *reinterpret_cast<float2*>(dpi + (size_t)b * PC_D + 2 * lane) = acc;
    if (lane == 0) part_item[b] = li;
}

__global__ void joint_loss_reduce_kernel(const float* part_type, const float* part_item, int B, int K, float alpha,
                                         float* losses) {
    __shared__ float r0[256], r1[256];
    float a = 0.f, c = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) { a += part_type[b]; c += part_item[b]; }
    r0[threadIdx.x] = a; r1[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float tl = r0[0] / (float)B, il = r1[0] / ((float)B * (float)K);
        losses[0] = alpha * il + (1.0f - alpha) * tl;
        losses[1] = tl;
        losses[2] = il;
    }
}

extern "C" int pc_joint_loss_dim(const float* sims, const float* proj, const int32_t* pos_types,
                                 const int32_t* neg_types, const float* pos_items, const float* neg_items, int batch,
                                 int num_types, int k, int dim, float margin, float alpha, float* losses, float* dsims_val,
                                 float* dproj, float* partials, void* stream) {
    if (!sims || !proj || !pos_types || !neg_types || !pos_items || !neg_items || !losses || !partials)
        return PC_EINVAL;
    if (batch <= 0 || num_types <= 0 || k <= 0) return PC_EINVAL;
    if (dim != 128 && dim != 256) return PC_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dim == 128) PC_LAUNCH(joint_loss_kernel<1>, dim3((batch + 3) / 4), dim3(256), 0, st, sims, proj, pos_types, neg_types,
                              pos_items, neg_items, batch, num_types, k, margin, alpha, partials, partials + batch, dsims_val, dproj);
    else PC_LAUNCH(joint_loss_kernel<2>, dim3((batch + 3) / 4), dim3(256), 0, st, sims, proj, pos_types, neg_types,
                   pos_items, neg_items, batch, num_types, k, margin, alpha, partials, partials + batch, dsims_val, dproj);
    PC_TRY(pc_launch_status());
    PC_LAUNCH(joint_loss_reduce_kernel, dim3(1), dim3(256), 0, st, partials, partials + batch, batch, k,
                       alpha, losses);
    return pc_launch_status();
}
extern "C" int pc_joint_loss(const float* sims, const float* proj, const int32_t* pos_types,
                             const int32_t* neg_types, const float* pos_items, const float* neg_items, int batch,
                             int num_types, int k, float margin, float alpha, float* losses, float* dsims_val,
                             float* dproj, float* partials, void* stream) {
    return pc_joint_loss_dim(sims, proj, pos_types, neg_types, pos_items, neg_items, batch, num_types, k, PC_D, margin, alpha,
                             losses, dsims_val, dproj, partials, stream);
}

// Sparse backward of sims = c E_c^T restricted to the two touched columns per row:
//   dc[b] = v0 E_c[pos_b] + v1 E_c[neg_b];  dE_c[pos_b] += v0 c[b];  dE_c[neg_b] += v1 c[b]
// one wave per sample, lane = one of the 64 type dims (256-B atomic rows: full-rate shape)
__global__ __launch_bounds__(256) void type_hinge_bwd_kernel(const float* dsims_val, const int32_t* pos_t,
                                                             const int32_t* neg_t, const float* c, const float* ec,
                                                             int B, float* dc, float* dec) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float v0 = dsims_val[2 * b], v1 = dsims_val[2 * b + 1];
    const int p = pos_t[b], n = neg_t[b];
    const float cb = c[(size_t)b * PC_L + lane];
    dc[(size_t)b * PC_L + lane] = v0 * ec[(size_t)p * PC_L + lane] + v1 * ec[(size_t)n * PC_L + lane];
    if (v0 != 0.f || v1 != 0.f) {
        unsafeAtomicAdd(dec + (size_t)p * PC_L + lane, v0 * cb);
        unsafeAtomicAdd(dec + (size_t)n * PC_L + lane, v1 * cb);
    }
}

// dense[b,:] = 0 except dense[b,pos_b] += v0, dense[b,neg_b] += v1 (autograd/module mode, where
// the caller's graph needs d(loss)/d(type_similarities) as a tensor)
__global__ void expand_type_grad_kernel(const float* dsims_val, const int32_t* pos_t, const int32_t* neg_t, int B,
                                        int T, float* dense) {
    const int b = blockIdx.x;
    float* row = dense + (size_t)b * T;
    for (int t = threadIdx.x; t < T; t += blockDim.x) row[t] = 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        row[pos_t[b]] += dsims_val[2 * b];
        row[neg_t[b]] += dsims_val[2 * b + 1];
    }
}

extern "C" int pc_expand_type_grad(const float* dsims_val, const int32_t* pos_types, const int32_t* neg_types,
                                   int batch, int num_types, float* dense, void* stream) {
    if (!dsims_val || !pos_types || !neg_types || !dense || batch <= 0 || num_types <= 0) return PC_EINVAL;
    PC_LAUNCH(expand_type_grad_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, dsims_val, pos_types,
                       neg_types, batch, num_types, dense);
    return pc_launch_status();
}

// ---------------------------------------------------------------------------------------
// Metrics.evaluate_model pieces (src/utils/metrics.py:62-117), one wave per row.
//   hit@k (metrics.py:7-27): row r hits iff its ground-truth column gt = r is among the k largest
//   of sims[r,:]; equivalently fewer than k entries beat sims[r,gt] (ties broken towards the
//   lower index, as pc_topk_rows does).  gt >= cols can never hit: the reference compares
//   arange(B*K) against only B columns (metrics.py:95-100) -- reproduced.
__global__ __launch_bounds__(256) void hit_rank_kernel(const float* sims, int rows, int cols, int32_t* rank) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    if (r >= cols) { if (lane == 0) rank[r] = 0x7fffffff; return; }
    const float* row = sims + (size_t)r * cols;
    const float g = row[r];
    int beat = 0;
    for (int c = lane; c < cols; c += 64) {
        const float v = row[c];
        beat += (v > g || (v == g && c < r)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) beat += __shfl_xor(beat, o, 64);
    if (lane == 0) rank[r] = beat;
}

extern "C" int pc_hit_rank(const float* sims, int rows, int cols, int32_t* rank, void* stream) {
    if (!sims || !rank || rows <= 0 || cols <= 0) return PC_EINVAL;
    PC_LAUNCH(hit_rank_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, sims, rows, cols, rank);
    return pc_launch_status();
}

//   mean_relevance (metrics.py:44-60): cos[b,k] = <x_bk, y_b> / (max(|x_bk|,eps) * max(|y_b|,eps)), eps 1e-8
template <int NCH>
__global__ __launch_bounds__(256) void cosine_rows_kernel(const float* x, const float* y, int B, int K, float* out) {
    constexpr int D = 128 * NCH;
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B * K) return;
    float d_ = 0.f, a_ = 0.f, b_ = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        const float2 a = *reinterpret_cast<const float2*>(x + (size_t)r * D + 128 * ch + 2 * lane);
        const float2 b = *reinterpret_cast<const float2*>(y + (size_t)(r / K) * D + 128 * ch + 2 * lane);
        d_ += a.x * b.x + a.y * b.y; a_ += a.x * a.x + a.y * a.y; b_ += b.x * b.x + b.y * b.y;
    }
    const float dot = wave_sum(d_);
    const float na = sqrtf(wave_sum(a_)), nb = sqrtf(wave_sum(b_));
    if (lane == 0) out[r] = dot / (fmaxf(na, 1e-8f) * fmaxf(nb, 1e-8f));
}

extern "C" int pc_cosine_rows_dim(const float* x, const float* y, int batch, int k, int dim, float* out, void* stream) {
    if (!x || !y || !out || batch <= 0 || k <= 0) return PC_EINVAL;
    if (dim != 128 && dim != 256) return PC_ESHAPE;
    if (dim == 128) PC_LAUNCH(cosine_rows_kernel<1>, dim3((batch * k + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, batch, k, out);
    else PC_LAUNCH(cosine_rows_kernel<2>, dim3((batch * k + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, batch, k, out);
    return pc_launch_status();
}
extern "C" int pc_cosine_rows(const float* x, const float* y, int batch, int k, float* out, void* stream) {
    return pc_cosine_rows_dim(x, y, batch, k, PC_D, out, stream);
}

// ---------------------------------------------------------------------------------------
struct JointWs {
    float *dpi, *dtp, *dce, *dc, *dh, *dt;
    float *typ_wt, *dec_wt, *enc_wt;
    float *sims, *proj, *dproj, *dsv, *partials;      // train_step scratch
    float *h, *c, *pi, *tp;                           // train_step saved activations
    float* slabs[6]; size_t slab_floats[6];            // one region per gradient product: they run as one grouped launch
    float* tslabs[2]; int tblocks[2];                  // per-workgroup private copies of the two type-table gradients
    int table_mode;                                    // type-table gradients: 2 one-hot products, 1 LDS slabs, 0 atomics
    size_t total;
};

static JointWs joint_ws_layout(void* base, int B, int T, int K) {
    JointWs w;
    size_t off = 0;
    auto take = [&](size_t floats) {
        float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
        off += align256(floats * sizeof(float));
        return p;
    };
    w.dpi = take((size_t)B * PC_D);
    w.dtp = take((size_t)B * K * PC_D);
    w.dce = take((size_t)B * K * PC_L);
    w.dc = take((size_t)B * PC_L);
    w.dh = take((size_t)B * LH);
    w.dt = take((size_t)B * PC_L);
    w.typ_wt = take(PC_L * PC_D);
    w.dec_wt = take(LH * PC_L);
    w.enc_wt = take(PC_L * LH);
    w.sims = take((size_t)B * T);
    w.proj = take((size_t)B * K * PC_D);
    w.dproj = take((size_t)B * K * PC_D);
    w.dsv = take((size_t)B * 2);
    w.partials = take((size_t)B * 2);
    w.h = take((size_t)B * LH);
    w.c = take((size_t)B * PC_L);
    w.pi = take((size_t)B * PC_D);
    w.tp = take((size_t)B * K * PC_D);
    const size_t sf[4] = {gemm_tn_workspace_floats(B, PC_D, PC_D), gemm_tn_workspace_floats(B * K, PC_D, PC_L),
                          gemm_tn_workspace_floats(B, PC_L, LH), gemm_tn_workspace_floats(B, LH, PC_L)};
    for (int i = 0; i < 4; i++) { w.slab_floats[i] = sf[i]; w.slabs[i] = take(sf[i]); }
    // nn.Embedding gradients of the two type tables (rows picked by top-k / by the query types).  Small tables: as
    // one-hot products dE = onehot(idx)^T x through the few-row TN kernel, in the same grouped launch as the weight
    // gradients (no atomics, fixed summation order); else per-workgroup LDS copies summed by the grouped reduce; else
    // (T = 34800: a dense [T,64] product would be 55 GFLOP) hardware float atomics.
    w.table_mode = (T <= 512 && T % 4 == 0) ? 2 : 0;
    w.tblocks[0] = w.tblocks[1] = 0;
    w.tslabs[0] = w.tslabs[1] = nullptr;
    w.slab_floats[4] = w.slab_floats[5] = 0;
    w.slabs[4] = w.slabs[5] = nullptr;
    if (w.table_mode == 2) {
        w.slab_floats[4] = gemm_tn_workspace_floats(B * K, T, PC_L);
        w.slab_floats[5] = gemm_tn_workspace_floats(B, T, PC_L);
        w.slabs[4] = take(w.slab_floats[4]);
        w.slabs[5] = take(w.slab_floats[5]);
    } else {
        w.tblocks[0] = scatter_add_slab_blocks(T, B * K, PC_L);
        w.tblocks[1] = scatter_add_slab_blocks(T, B, PC_L);
        if (w.tblocks[0] > 0 && w.tblocks[1] > 0) {
            w.table_mode = 1;
            for (int i = 0; i < 2; i++) w.tslabs[i] = take((size_t)w.tblocks[i] * T * PC_L);
        }
    }
    w.total = off;
    return w;
}

extern "C" size_t pc_joint_workspace_bytes(int batch, int num_types, int k) {
    if (batch <= 0 || num_types <= 0 || k <= 0) return 0;
    return joint_ws_layout(nullptr, batch, num_types, k).total;
}

static int joint_check(const pc_joint_tensors* p, int B, int T, int K) {
    if (!p || !p->product_table || !p->enc_w || !p->enc_b || !p->dec_w || !p->dec_b || !p->typ_w || !p->typ_b ||
        !p->itm_w || !p->itm_b || !p->query_types || !p->comp_types)
        return PC_EINVAL;
    if (B <= 0 || T <= 0) return PC_EINVAL;
    if (K < 1 || K > JMAX_K || K > T) return PC_ESHAPE;
    return PC_OK;
}

static int joint_forward_impl(const pc_joint_tensors* p, const int32_t* query_idx, const int32_t* query_types,
                              int B, int T, int K, float* sims, int32_t* topk, float* proj,
                              const pc_joint_saved* sv, void* ws, size_t ws_bytes, void* stream) {
    // proj == nullptr: the caller (the fused step) forms proj = pi * tp itself, inside joint_rowwise_kernel
    PC_TRY(joint_check(p, B, T, K));
    if (!query_idx || !query_types || !sims || !topk || !sv || !sv->h || !sv->c || !sv->pi || !sv->tp)
        return PC_EINVAL;
    (void)ws; (void)ws_bytes;
    hipStream_t st = (hipStream_t)stream;
    // h = relu(enc(E_q[query_types]))      type_transition.py:17 (dropout p = 0)
    // pi = item_projection(E_prod[query_idx])   item_prediction.py:31, p_companion.py:51
    // (both depend on the batch only: one grouped few-row launch)
    NtArgs first[2];
    NtArgs& e = first[0];
    e = nt_plain(p->query_types, PC_L, p->enc_w, PC_L, p->enc_b, sv->h, LH, B, LH, PC_L);
    e.gather = query_types; e.epilogue = NT_EPI_RELU;
    NtArgs& ip = first[1];
    ip = nt_plain(p->product_table, PC_D, p->itm_w, PC_D, p->itm_b, sv->pi, PC_D, B, PC_D, PC_D);
    ip.gather = query_idx;
    PC_TRY(launch_gemm_nt_group(first, 2, st));
    // h = dropout(h) in training mode        type_transition.py:17 (saved dropped: the decoder gradient needs it so)
    if (p->dropout.p > 0.f) PC_TRY(launch_dropout(sv->h, (size_t)B * LH, p->dropout, PC_DROP_STREAM_HIDDEN, sv->h, st));
    // c = dec(h)                            type_transition.py:19
    PC_TRY(launch_gemm_nt(nt_plain(sv->h, LH, p->dec_w, LH, p->dec_b, sv->c, PC_L, B, PC_L, LH), st));
    // sims = c E_c^T                        p_companion.py:60-63
    PC_TRY(launch_gemm_nt(nt_plain(sv->c, PC_L, p->comp_types, PC_L, nullptr, sims, T, B, T, PC_L), st));
    PC_TRY(pc_topk_rows(sims, B, T, K, topk, nullptr, stream));
    // tp = type_projection(E_c[topk])           item_prediction.py:35, p_companion.py:65
    NtArgs tpj = nt_plain(p->comp_types, PC_L, p->typ_w, PC_L, p->typ_b, sv->tp, PC_D, B * K, PC_D, PC_L);
    tpj.gather = topk;
    PC_TRY(launch_gemm_nt(tpj, st));
    return proj ? pc_hadamard_forward(sv->pi, sv->tp, B, K, proj, stream) : PC_OK;
}

extern "C" int pc_joint_forward(const pc_joint_tensors* p, const int32_t* query_idx, const int32_t* query_types,
                                int B, int T, int K, float* sims, int32_t* topk, float* proj,
                                const pc_joint_saved* sv, void* ws, size_t ws_bytes, void* stream) {
    if (!proj) return PC_EINVAL;
    return joint_forward_impl(p, query_idx, query_types, B, T, K, sims, topk, proj, sv, ws, ws_bytes, stream);
}

// rowwise_done: joint_rowwise_kernel has already produced dpi / dtp / dc and the type hinge's dE_c rows (into a
// cleared g->comp_types): the table clears, the Hadamard backward and the type-hinge backward are skipped here.
static int joint_backward_impl(const pc_joint_tensors* p, const pc_joint_tensors* g, const int32_t* query_idx,
                               const int32_t* query_types, const int32_t* pos_types, const int32_t* neg_types,
                               const int32_t* topk, int B, int T, int K, const float* dsims_val,
                               const float* dproj, const pc_joint_saved* sv, void* ws, size_t ws_bytes,
                               void* stream, bool rowwise_done) {
    PC_TRY(joint_check(p, B, T, K));
    if (!g || !g->enc_w || !g->enc_b || !g->dec_w || !g->dec_b || !g->typ_w || !g->typ_b || !g->itm_w ||
        !g->itm_b || !g->query_types || !g->comp_types)
        return PC_EINVAL;
    if (!query_idx || !query_types || !pos_types || !neg_types || !topk || !sv || !ws) return PC_EINVAL;
    if (!rowwise_done && (!dsims_val || !dproj)) return PC_EINVAL;
    if (ws_bytes < pc_joint_workspace_bytes(B, T, K)) return PC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    JointWs w = joint_ws_layout(ws, B, T, K);
    const SegInfo siB = make_seginfo(nullptr, B, 128), siBK = make_seginfo(nullptr, B * K, 128);

    // type-table gradients: small tables are summed deterministically from per-workgroup slabs by the grouped
    // reduce at the end (E_q: overwritten there, no clear needed; E_c also takes the type hinge's atomics)
    const bool tslab = w.table_mode == 1;
    if (!rowwise_done) {
        if (w.table_mode == 0) PC_HIP_TRY(hipMemsetAsync(g->query_types, 0, (size_t)T * PC_L * 4, st));
        PC_HIP_TRY(hipMemsetAsync(g->comp_types, 0, (size_t)T * PC_L * 4, st));
    }
    TransposeBatch tb = {};
    tb.n = 3;
    tb.job[0] = {p->typ_w, w.typ_wt, PC_D, PC_L};                    // [D,L] -> [L,D]
    tb.job[1] = {p->dec_w, w.dec_wt, PC_L, LH};                      // [L,L/2] -> [L/2,L]
    tb.job[2] = {p->enc_w, w.enc_wt, LH, PC_L};                      // [L/2,L] -> [L,L/2]
    PC_TRY(launch_transpose_batch(tb, st));
    // The four weight gradients depend only on buffers that stay untouched to the end of the backward pass: they are
    // collected here and run as ONE grouped launch + ONE grouped slab reduce after the dX chain (each alone is 64
    // workgroups and ~13 + 5 us of latency).
    TnArgs tn[6];

    // ---- item branch
    if (!rowwise_done) PC_TRY(pc_hadamard_backward(dproj, sv->pi, sv->tp, B, K, w.dpi, w.dtp, stream));
    TnArgs& ti = tn[0];
    ti = {};
    ti.Z = w.dpi; ti.ldz = PC_D; ti.A = p->product_table; ti.lda = PC_D; ti.gather = query_idx; ti.R = B;
    ti.No = PC_D; ti.Ni = PC_D; ti.seg = siB; ti.dW = g->itm_w; ti.lddw = PC_D; ti.db = g->itm_b;
    ti.slabs = w.slabs[0]; ti.slab_floats = w.slab_floats[0];
    TnArgs& tt = tn[1];
    tt = {};
    tt.Z = w.dtp; tt.ldz = PC_D; tt.A = p->comp_types; tt.lda = PC_L; tt.gather = topk; tt.R = B * K;
    tt.No = PC_D; tt.Ni = PC_L; tt.seg = siBK; tt.dW = g->typ_w; tt.lddw = PC_L; tt.db = g->typ_b;
    tt.slabs = w.slabs[1]; tt.slab_floats = w.slab_floats[1];
    // dE_c[topk] += dtp typ_w     (row-sparse: only the K selected rows per sample): launched below, grouped with dh
    NtArgs pair[2];
    pair[0] = nt_plain(w.dtp, PC_D, w.typ_wt, PC_D, nullptr, w.dce, PC_L, B * K, PC_L, PC_D);

    // ---- type branch (two touched similarity columns per row)
    if (!rowwise_done) {
        PC_LAUNCH(type_hinge_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, st, dsims_val, pos_types, neg_types,
                           sv->c, p->comp_types, B, w.dc, g->comp_types);
        PC_TRY(pc_launch_status());
    }
    TnArgs& td = tn[2];
    td = {};
    td.Z = w.dc; td.ldz = PC_L; td.A = sv->h; td.lda = LH; td.R = B; td.No = PC_L; td.Ni = LH; td.seg = siB;
    td.dW = g->dec_w; td.lddw = LH; td.db = g->dec_b; td.slabs = w.slabs[2]; td.slab_floats = w.slab_floats[2];
    NtArgs& dh = pair[1];
    dh = nt_plain(w.dc, PC_L, w.dec_wt, PC_L, nullptr, w.dh, LH, B, LH, PC_L);
    dh.epilogue = NT_EPI_DRELU; dh.aux = sv->h; dh.ldaux = LH;
    PC_TRY(launch_gemm_nt_group(pair, 2, st));
    // (the d-ReLU epilogue tested the saved, already dropped h: dropped units are 0 there; kept ones still need 1/(1-p))
    if (p->dropout.p > 0.f) PC_TRY(launch_dropout(w.dh, (size_t)B * LH, p->dropout, PC_DROP_STREAM_HIDDEN, w.dh, st));
    if (tslab) PC_TRY(launch_scatter_add_slabs(topk, B * K, PC_L, T, w.dce, w.tslabs[0], st));
    else if (w.table_mode == 0) PC_TRY(pc_scatter_add_rows_small(g->comp_types, T, topk, B * K, PC_L, w.dce, stream));
    TnArgs& te = tn[3];
    te = {};
    te.Z = w.dh; te.ldz = LH; te.A = p->query_types; te.lda = PC_L; te.gather = query_types; te.R = B;
    te.No = LH; te.Ni = PC_L; te.seg = siB; te.dW = g->enc_w; te.lddw = PC_L; te.db = g->enc_b;
    te.slabs = w.slabs[3]; te.slab_floats = w.slab_floats[3];
    PC_TRY(launch_gemm_nt(nt_plain(w.dh, LH, w.enc_wt, LH, nullptr, w.dt, PC_L, B, PC_L, LH), st));
    if (w.table_mode == 2) {
        const SegInfo siT0 = make_seginfo(nullptr, B * K, 128), siT1 = make_seginfo(nullptr, B, 128);
        TnArgs& tc = tn[4];
        tc = {};
        tc.z_onehot = topk; tc.A = w.dce; tc.lda = PC_L; tc.R = B * K; tc.No = T; tc.Ni = PC_L; tc.seg = siT0;
        tc.dW = g->comp_types; tc.lddw = PC_L; tc.accumulate = 1;      // (+ the type hinge's two rows per sample, above)
        tc.slabs = w.slabs[4]; tc.slab_floats = w.slab_floats[4];
        TnArgs& tqy = tn[5];
        tqy = {};
        tqy.z_onehot = query_types; tqy.A = w.dt; tqy.lda = PC_L; tqy.R = B; tqy.No = T; tqy.Ni = PC_L; tqy.seg = siT1;
        tqy.dW = g->query_types; tqy.lddw = PC_L; tqy.accumulate = 0;
        tqy.slabs = w.slabs[5]; tqy.slab_floats = w.slab_floats[5];
        return launch_gemm_tn_group(tn, 6, nullptr, 0, st);
    }
    if (!tslab) {
        PC_TRY(pc_scatter_add_rows_small(g->query_types, T, query_types, B, PC_L, w.dt, stream));
        return launch_gemm_tn_group(tn, 4, nullptr, 0, st);
    }
    PC_TRY(launch_scatter_add_slabs(query_types, B, PC_L, T, w.dt, w.tslabs[1], st));
    const TnReduceJob tj[2] = {{w.tslabs[0], w.tblocks[0], T * PC_L, g->comp_types, 1},
                               {w.tslabs[1], w.tblocks[1], T * PC_L, g->query_types, 0}};
    return launch_gemm_tn_group(tn, 4, tj, 2, st);
}

extern "C" int pc_joint_backward(const pc_joint_tensors* p, const pc_joint_tensors* g, const int32_t* query_idx,
                                 const int32_t* query_types, const int32_t* pos_types, const int32_t* neg_types,
                                 const int32_t* topk, int B, int T, int K, const float* dsims_val,
                                 const float* dproj, const pc_joint_saved* sv, void* ws, size_t ws_bytes,
                                 void* stream) {
    return joint_backward_impl(p, g, query_idx, query_types, pos_types, neg_types, topk, B, T, K, dsims_val, dproj, sv,
                               ws, ws_bytes, stream, false);
}

extern "C" int pc_joint_train_step(const pc_joint_tensors* p, const pc_joint_tensors* g, const int32_t* query_idx,
                                   const int32_t* query_types, const int32_t* pos_types, const int32_t* neg_types,
                                   const float* pos_items, const float* neg_items, int B, int T, int K,
                                   float margin, float alpha, float* losses, int32_t* topk, void* ws,
                                   size_t ws_bytes, void* stream) {
    PC_TRY(joint_check(p, B, T, K));
    if (!ws || !losses || !topk || !g || !g->comp_types || !g->query_types || !pos_types || !neg_types || !pos_items ||
        !neg_items)
        return PC_EINVAL;
    if (ws_bytes < pc_joint_workspace_bytes(B, T, K)) return PC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    JointWs w = joint_ws_layout(ws, B, T, K);
    pc_joint_saved sv;
    sv.h = w.h; sv.c = w.c; sv.pi = w.pi; sv.tp = w.tp;
    // the table-gradient clears come first: the row-wise pass below already adds the type hinge's rows
    if (w.table_mode == 0) PC_HIP_TRY(hipMemsetAsync(g->query_types, 0, (size_t)T * PC_L * 4, st));
    PC_HIP_TRY(hipMemsetAsync(g->comp_types, 0, (size_t)T * PC_L * 4, st));
    PC_TRY(joint_forward_impl(p, query_idx, query_types, B, T, K, w.sims, topk, nullptr, &sv, ws, ws_bytes, stream));
    PC_LAUNCH(joint_rowwise_kernel, dim3((B + 3) / 4), dim3(256), 0, st, w.sims, sv.pi, sv.tp, pos_types, neg_types,
              pos_items, neg_items, sv.c, p->comp_types, B, T, K, margin, alpha, w.partials, w.partials + B, w.dpi, w.dtp,
              w.dc, g->comp_types);
    PC_TRY(pc_launch_status());
    PC_LAUNCH(joint_loss_reduce_kernel, dim3(1), dim3(256), 0, st, w.partials, w.partials + B, B, K, alpha, losses);
    PC_TRY(pc_launch_status());
    return joint_backward_impl(p, g, query_idx, query_types, pos_types, neg_types, topk, B, T, K, nullptr, nullptr, &sv,
                               ws, ws_bytes, stream, true);
}

// ---------------------------------------------------------------------------------------
// Type-filtered retrieval (inference.py:90-118): for every predicted (query, complementary type)
// row r, score = proj[r] . features[c] over the products c of that type, top-n by score.  One
// wavefront per row: proj[r] sits in LDS (broadcast reads), each lane scores its strided candidates
// (rows of a 1000-product type are L2-resident) and keeps its n best sorted; n rounds of a wave-wide
// arg-max pop the winners (ties: lower product index, like topk_rows_kernel).
#define RMAX_N 16
__global__ __launch_bounds__(256) void retrieve_topk_kernel(const float* proj, const int32_t* types, int rows,
                                                            const int32_t* type_rowptr, const int32_t* type_col,
                                                            const float* table, int n_types, int n, int32_t* out_idx,
                                                            float* out_score, int D) {
    __shared__ __attribute__((aligned(16))) float q[4][256];          // PRODUCT_EMB_DIM 128 or 256
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + w;
    if (r < rows)
        for (int d = lane; d < D; d += 64) q[w][d] = proj[(size_t)r * D + d];
    __syncthreads();
    if (r >= rows) return;
    const int t = types[r];
    int c0 = 0, c1 = 0;
    if (t >= 0 && t < n_types) { c0 = type_rowptr[t]; c1 = type_rowptr[t + 1]; }
    float v[RMAX_N];
    int ix[RMAX_N];
#pragma unroll
    for (int j = 0; j < RMAX_N; j++) { v[j] = -INFINITY; ix[j] = 0x7fffffff; }
    for (int c = c0 + lane; c < c1; c += 64) {
        const int pid = type_col[c];
        const float4* f = reinterpret_cast<const float4*>(table + (size_t)pid * D);
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < D / 4; k++) {
            const float4 x = f[k];
            const float4 y = *reinterpret_cast<const float4*>(&q[w][4 * k]);
            s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
        float x = s;
        int xi = pid;
#pragma unroll
        for (int j = 0; j < RMAX_N; j++) {
            if (j < n) {
                const bool better = x > v[j] || (x == v[j] && xi < ix[j]);
                if (better) { const float tv = v[j]; const int ti = ix[j]; v[j] = x; ix[j] = xi; x = tv; xi = ti; }
            }
        }
    }
    for (int k = 0; k < n; k++) {
        float bv = v[0];
        int bi = ix[0];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (ix[0] == bi && bi != 0x7fffffff) {
#pragma unroll
            for (int j = 0; j < RMAX_N - 1; j++) { v[j] = v[j + 1]; ix[j] = ix[j + 1]; }
            v[RMAX_N - 1] = -INFINITY; ix[RMAX_N - 1] = 0x7fffffff;
        }
        if (lane == 0) {
            out_idx[(size_t)r * n + k] = bi == 0x7fffffff ? -1 : bi;
            out_score[(size_t)r * n + k] = bv;
        }
    }
}

extern "C" int pc_retrieve_topk_dim(const float* proj, const int32_t* types, int rows, const int32_t* type_rowptr,
                                    const int32_t* type_col, const float* table, int n_types, int n, int dim, int32_t* out_idx,
                                    float* out_score, void* stream) {
    if (!proj || !types || !type_rowptr || !type_col || !table || !out_idx || !out_score) return PC_EINVAL;
    if (rows <= 0 || n_types <= 0) return PC_EINVAL;
    if (n < 1 || n > RMAX_N || (dim != 128 && dim != 256)) return PC_ESHAPE;
    PC_LAUNCH(retrieve_topk_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, proj, types, rows,
              type_rowptr, type_col, table, n_types, n, out_idx, out_score, dim);
    return pc_launch_status();
}
extern "C" int pc_retrieve_topk(const float* proj, const int32_t* types, int rows, const int32_t* type_rowptr,
                                const int32_t* type_col, const float* table, int n_types, int n, int32_t* out_idx,
                                float* out_score, void* stream) {
    return pc_retrieve_topk_dim(proj, types, rows, type_rowptr, type_col, table, n_types, n, PC_D, out_idx, out_score, stream);
}
