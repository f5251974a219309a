// One iteration of Product2Vec.train_model's loop body (product2vec.py:126-159) in index
// form: the four FFN calls of :132-134 run as ONE segmented launch sequence over the
// concatenated rows [anchor | neighbours | positive | negatives] (BatchNorm statistics stay
// per call), then attention, loss and the whole backward.  Gradients overwrite `g`
// (= optimizer.zero_grad() + loss.backward()); the optimizer step is pc_adam_step.
#include "common.h"

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct StepWs {
    int32_t* idx_all;
    float *y, *h0, *a2, *a1, *bn;     // bn: mean, invstd, scale, shift  [4][MAX_SEG][H]
    float *q, *qt, *probs, *c, *sp, *ctx, *emb;
    float *dy, *demb, *dpos_tmp, *dneg_tmp;
    void* ffn_ws; size_t ffn_bytes;
    void* attn_ws; size_t attn_bytes;
    size_t total;
};

static StepWs step_ws_layout(void* base, int B, int N, int K, int D) {
    StepWs w;
    const size_t R = (size_t)B * (2 + N + K) + 1;          // +1: the shared padding row of the compact layout
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? reinterpret_cast<char*>(base) + off : nullptr;
        off += align256(bytes);
        return p;
    };
    w.idx_all = (int32_t*)take(R * 4);
    w.y = (float*)take(R * D * 4);
    w.h0 = (float*)take(R * PC_H * 4);
    w.a2 = (float*)take(R * PC_H * 4);
    w.a1 = (float*)take(R * PC_H * 4);
    w.bn = (float*)take(4 * PC_MAX_SEG * PC_H * 4);
    w.q = (float*)take((size_t)B * D * 4);
    w.qt = (float*)take((size_t)B * PC_HEADS * D * 4);
    w.c = (float*)take((size_t)B * PC_HEADS * D * 4);
    w.sp = (float*)take((size_t)B * PC_HEADS * 4);
    w.probs = (float*)take((size_t)B * PC_HEADS * (N > 0 ? N : 1) * 4);
    w.ctx = (float*)take((size_t)B * D * 4);
    w.emb = (float*)take((size_t)B * D * 4);
    w.dy = (float*)take(R * D * 4);
    w.demb = (float*)take((size_t)B * D * 4);
    w.dpos_tmp = (float*)take((size_t)B * 4);
    w.dneg_tmp = (float*)take((size_t)B * 4);
    w.ffn_bytes = pc_p2v_ffn_workspace_bytes((int)R);
    w.ffn_ws = take(w.ffn_bytes);
    w.attn_bytes = N > 0 ? pc_p2v_attention_workspace_bytes_dim(B, N, D) : 0;
    w.attn_ws = take(w.attn_bytes);
    w.total = off;
    return w;
}

extern "C" size_t pc_p2v_train_step_workspace_bytes_dim(int batch, int n_nbr, int k_neg, int dim) {
    if (batch <= 0 || n_nbr < 0 || k_neg <= 0 || (dim != 128 && dim != 256)) return 0;
    return step_ws_layout(nullptr, batch, n_nbr, k_neg, dim).total;
}
extern "C" size_t pc_p2v_train_step_workspace_bytes(int batch, int n_nbr, int k_neg) {
    return pc_p2v_train_step_workspace_bytes_dim(batch, n_nbr, k_neg, PC_D);
}
extern "C" size_t pc_p2v_attention_workspace_bytes_dim(int batch, int n_keys, int dim);
extern "C" int pc_p2v_triplet_loss_dim(const float* a, const float* p, const float* n, int batch, int k_neg, int dim,
                                       float margin, float* loss, float* d_pos, float* d_neg, float* da, float* dp,
                                       float* dn, void* stream);

__device__ __forceinline__ void concat_idx_body(const int32_t* a, int na, const int32_t* b, int nb, const int32_t* c, int nc,
                                                const int32_t* d, int nd, int32_t* out, int i) {
    if (i < na) out[i] = a[i];
    else if (i < na + nb) out[i] = b[i - na];
    else if (i < na + nb + nc) out[i] = c[i - na - nb];
    else if (i < na + nb + nc + nd) out[i] = d[i - na - nb - nc];
}
__global__ void concat_idx_kernel(const int32_t* a, int na, const int32_t* b, int nb, const int32_t* c, int nc,
                                  const int32_t* d, int nd, int32_t* out) {
    concat_idx_body(a, na, b, nb, c, nc, d, nd, out, blockIdx.x * blockDim.x + threadIdx.x);
}

// The loader's side of the unsplit step's first launch (round 6): the same concatenation, queued behind the batch builder on
// ITS stream, with the neighbour row count read from the device (the host learns it a step later).
__global__ void concat_rows_dev_kernel(const int32_t* a, int na, const int32_t* b, const int32_t* nb_dev, int nb_cap, const int32_t* c,
                                       int nc, const int32_t* d, int nd, int32_t* out) {
    int nb = *nb_dev + 1;                                    // the distinct neighbours and the -1 row
    nb = nb < 1 ? 1 : (nb > nb_cap ? nb_cap : nb);
    concat_idx_body(a, na, b, nb, c, nc, d, nd, out, blockIdx.x * blockDim.x + threadIdx.x);
}

extern "C" int pc_p2v_concat_step_rows(const int32_t* anchor_idx, const int32_t* positive_idx, const int32_t* negative_idx,
                                       const int32_t* nb_rows, const int32_t* n_unique_dev, int nb_capacity, int B, int K,
                                       int32_t* rows_out, int rows_capacity, void* stream) {
    if (!anchor_idx || !positive_idx || !negative_idx || !nb_rows || !n_unique_dev || !rows_out) return PC_EINVAL;
    if (B <= 0 || K <= 0 || nb_capacity < 1) return PC_EINVAL;
    const long long most = (long long)B * (2 + K) + nb_capacity;
    if (most > rows_capacity) return PC_ESHAPE;
    PC_LAUNCH(concat_rows_dev_kernel, dim3((unsigned)((most + 255) / 256)), dim3(256), 0, (hipStream_t)stream, anchor_idx, B, nb_rows,
              n_unique_dev, nb_capacity, positive_idx, B, negative_idx, B * K, rows_out);
    return pc_launch_status();
}

// The step's first launch: the row-index concatenation AND every transposed weight of the step (workgroups
// [0, concat_blocks) concatenate, the rest are 32 x 32 transpose tiles, tiles_x x tiles_y per job) -- the two were separate
// launches of ~5 us each, i.e. of pure launch latency.
__global__ __launch_bounds__(256) void p2v_prologue_kernel(const int32_t* a, int na, const int32_t* b, int nb, const int32_t* c, int nc,
                                                           const int32_t* d, int nd, int32_t* out, int concat_blocks,
                                                           TransposeBatch tb, int tiles_x, int tiles_y) {
    __shared__ float t[32][33];
    if ((int)blockIdx.x < concat_blocks) {
        concat_idx_body(a, na, b, nb, c, nc, d, nd, out, blockIdx.x * 256 + threadIdx.x);
        return;
    }
    transpose_tile_body<256>(tb, (int)blockIdx.x - concat_blocks, tiles_x, tiles_y, t, threadIdx.x);
}

int ffn_forward_part1(const pc_p2v_tensors* p, const float* table, const int32_t* idx, int rows, const pc_segments* seg,
                      const pc_ffn_saved* sv, double* local_sums, void* ws, size_t ws_bytes, void* stream);
int ffn_forward_part2(const pc_p2v_tensors* p, int rows, const pc_segments* seg, int update_running, float* y,
                      const pc_ffn_saved* sv, const double* global_sums, void* ws, size_t ws_bytes, void* stream,
                      const TransposeBatch* ride = nullptr);
int ffn_backward_part1(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* table, const int32_t* idx,
                       int rows, const pc_segments* seg, const float* dy, const pc_ffn_saved* sv, int with_dx,
                       int accumulate, double* local_sums, void* ws, size_t ws_bytes, void* stream, int transposed,
                       TnDefer* defer);
int ffn_backward_part2(const pc_p2v_tensors* g, const float* table, const int32_t* idx, int rows,
                       const pc_segments* seg, const pc_ffn_saved* sv, float* dx, int accumulate,
                       const double* local_sums, const double* global_sums, void* ws, size_t ws_bytes, void* stream,
                       TnDefer* defer);
int ffn_transposes(const pc_p2v_tensors* p, void* ws, int rows, int with_dx, TransposeBatch* tb);
int attention_transposes(const pc_p2v_tensors* p, void* ws, int B, int N, int key_rows, TransposeBatch* tb, float* zero_bk);
int attention_forward_impl(const pc_p2v_tensors* p, const float* query, const float* keys, int B, int N,
                           int key_rows, const int32_t* slot_row, float* out, const pc_attn_saved* sv, void* ws,
                           size_t ws_bytes, void* stream, int transposed, NtArgs* defer_out_chain);
int attention_backward_impl(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* query, const float* keys,
                            int B, int N, int key_rows, const int32_t* slot_row, int pad_row, const float* dout,
                            const pc_attn_saved* sv, float* dquery, float* dkeys, int accumulate, void* ws,
                            size_t ws_bytes, void* stream, const int32_t* ref_off, const int32_t* ref_slot,
                            int transposed, TnDefer* defer, const HingeMeanJob* rider, const LossPro* lossp,
                            const NtArgs* fwd_out_chain);
int pc_opt_fused_loss();       // (gemm_tn.hip: pc_set_option)
int pc_opt_fused_out_chain();
int triplet_loss_launch(const float* a, const float* p, const float* n, int batch, int k_neg, int dim, float margin,
                        float* loss, float* d_pos, float* d_neg, float* da, float* dp, float* dn, void* stream, int with_mean);

// nb_idx: neighbour rows of the step, nbc of them.  Dense layout: nbc = B*N slots in slot order
// (slot_row NULL).  Compact layout: the M real neighbours then one -1 row (nbc = M + 1), slot_row[B*N]
// maps every slot to its row and the -1 row carries the weight of all padding slots.
static int p2v_step_impl(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* table,
                         const int32_t* anchor_idx, const int32_t* positive_idx, const int32_t* negative_idx,
                         const int32_t* nb_idx, int nbc, const int32_t* slot_row, int B, int N, int K, float margin,
                         float* loss, float* d_pos, float* d_neg, float* anchor_emb, void* profile, void* ws,
                         size_t ws_bytes, void* stream, int phase = -1, double* fwd_sums = nullptr,
                         double* bwd_local = nullptr, const double* bwd_global = nullptr,
                         const float* nb_weight = nullptr, const int32_t* ref_off = nullptr,
                         const int32_t* ref_slot = nullptr, const pc_adam_fused* adam = nullptr,
                         const int32_t* rows_ready = nullptr) {
    // nb_weight (unique-neighbour layout): multiplicity of each of the nbc neighbour rows (its last entry = the
    // number of padding slots); replaces the single weighted row of the compact layout
    // phase -1: the whole step with this replica's BatchNorm statistics; 0/1/2: see pc_p2v_train_step_compact_sync
    const bool p0 = phase <= 0, p1 = phase == -1 || phase == 1, p2 = phase == -1 || phase == 2;
    if (adam && phase != -1) return PC_EINVAL;               // (the optimizer rides in the unsplit step's last launch only)
    // rows_ready: [anchor | neighbour rows | positive | negatives] as the loader concatenated them (pc_p2v_concat_step_rows).  The
    // step then has no launch of its own in front of Linear0: the transposed weights, which nothing needs before the attention,
    // ride in the BatchNorm finalize launch of the FFN forward.
    if (rows_ready && phase != -1) return PC_EINVAL;
    ProfileScope prof_scope((pc_profile*)profile);
    if (!p || !g || !table || !anchor_idx || !positive_idx || !negative_idx || !loss || !ws) return PC_EINVAL;
    if (B <= 0 || N < 0 || K <= 0 || (N > 0 && !nb_idx) || nbc < 0 || nbc > B * N + 1) return PC_EINVAL;
    // the anchor and positive calls are [B,128] BatchNorm inputs: the reference raises for a single row in training
    // mode (torch/nn/functional.py _verify_batch_size); so does a [1,5,128] negative block when K = 1
    if (B == 1) return PC_EBATCHNORM;
    if (p->dim != 0 && p->dim != 128 && p->dim != 256) return PC_ESHAPE;
    const int D = p->dim == 256 ? 256 : PC_D;
    if (ws_bytes < pc_p2v_train_step_workspace_bytes_dim(B, N, K, D)) return PC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    StepWs w = step_ws_layout(ws, B, N, K, D);
    const int32_t* rows = rows_ready ? rows_ready : w.idx_all;
    const int R = 2 * B + nbc + B * K;
    const int rA = 0, rN = B, rP = B + nbc, rG = rP + B;

    // call order of product2vec.py:132-134: anchor (:73), neighbours (:78), positive, negative
    pc_segments seg = {};
    seg.weighted_row = -1;
    seg.weight = 1.f;
    if (N > 0) {
        seg.nseg = 4; seg.start[0] = rA; seg.start[1] = rN; seg.start[2] = rP; seg.start[3] = rG; seg.start[4] = R;
        seg.count[1] = B * N;                                  // BatchNorm sees every padded slot
        if (nb_weight) { seg.row_weight = nb_weight; seg.row_weight_start = rN; seg.row_weight_rows = nbc; }
        else if (slot_row) { seg.weighted_row = rN + nbc - 1; seg.weight = (float)(B * N - (nbc - 1)); }
    } else {
        seg.nseg = 3; seg.start[0] = rA; seg.start[1] = rP; seg.start[2] = rG; seg.start[3] = R; seg.start[4] = R;
    }

    pc_ffn_saved sv;
    sv.h0 = w.h0; sv.a2 = w.a2; sv.a1 = w.a1;
    sv.bn_mean = w.bn; sv.bn_invstd = w.bn + PC_MAX_SEG * PC_H; sv.bn_scale = w.bn + 2 * PC_MAX_SEG * PC_H;
    sv.bn_shift = w.bn + 3 * PC_MAX_SEG * PC_H;
    // every transposed weight of the step (attention: Wo^T, Wq^T, [Wk;Wv]^T; FFN backward: W5^T, W3^T): one launch, which also
    // clears the key-bias gradient (exactly 0, see attention.hip); in the unsplit step it is the SAME launch as the row-index
    // concatenation
    TransposeBatch tb = {};
    if (p1) {
        PC_TRY(ffn_transposes(p, w.ffn_ws, R, 0, &tb));
        if (N > 0) PC_TRY(attention_transposes(p, w.attn_ws, B, N, slot_row ? nbc : B * N, &tb, g->in_proj_b + D));
    }
    if (p0) {
        if (rows_ready) {
            // (nothing to launch)
        } else if (phase == -1) {
            int tiles_x, tiles_y;
            transpose_batch_tiles(tb, &tiles_x, &tiles_y);
            const int cb = (R + 255) / 256;
            PC_LAUNCH(p2v_prologue_kernel, dim3(cb + tiles_x * tiles_y * tb.n), dim3(256), 0, st, anchor_idx, B, nb_idx, nbc,
                      positive_idx, B, negative_idx, B * K, w.idx_all, cb, tb, tiles_x, tiles_y);
        } else {
            PC_LAUNCH(concat_idx_kernel, dim3((R + 255) / 256), dim3(256), 0, st, anchor_idx, B, nb_idx, nbc, positive_idx, B,
                      negative_idx, B * K, w.idx_all);
        }
        PC_TRY(pc_launch_status());
        PC_TRY(ffn_forward_part1(p, table, rows, R, &seg, &sv, phase == 0 ? fwd_sums : nullptr, w.ffn_ws, w.ffn_bytes,
                                 stream));
        if (phase == 0) return PC_OK;
    }
    if (p2 && !p1)
        return ffn_backward_part2(g, table, rows, R, &seg, &sv, nullptr, 0, bwd_local, bwd_global, w.ffn_ws,
                                  w.ffn_bytes, stream, nullptr);
    if (phase == 1) PC_TRY(launch_transpose_batch(tb, st));      // (the split step: phase 0 ran the concatenation alone)
    PC_TRY(ffn_forward_part2(p, R, &seg, 1, w.y, &sv, phase == 1 ? fwd_sums : nullptr, w.ffn_ws, w.ffn_bytes, stream,
                             rows_ready ? &tb : nullptr));

    pc_attn_saved as;
    as.q = w.q; as.qt = w.qt; as.probs = w.probs; as.c = w.c; as.sp = w.sp; as.ctx = w.ctx;
    const float* emb = w.y;                     // anchor embedding = FFN output when there are no neighbours
    // (D = 128 with neighbours, the hinge riding: the forward's out-projection chain is deferred into the backward's first launch)
    const bool out_chain_rides = N > 0 && D == 128 && K <= 8 && pc_opt_fused_loss() && pc_opt_fused_out_chain();
    NtArgs fwd_out_chain[2];
    if (N > 0) {
        PC_TRY(attention_forward_impl(p, w.y + (size_t)rA * D, w.y + (size_t)rN * D, B, N, nbc, slot_row, w.emb,
                                      &as, w.attn_ws, w.attn_bytes, stream, 1, out_chain_rides ? fwd_out_chain : nullptr));
        emb = w.emb;
    }

    float* dp_out = d_pos ? d_pos : w.dpos_tmp;
    float* dn_out = d_neg ? d_neg : w.dneg_tmp;
    float* demb = N > 0 ? w.demb : w.dy + (size_t)rA * D;
    // the hinge mean rides on the first launch of the attention backward (D = 128 with neighbours); nothing in the step reads it
    const bool mean_rides = N > 0 && D == 128;
    // ... and so does the hinge itself (round 6): the prologue of the attention backward's first chain (LossPro, common.h) --
    // pc_set_option(PC_OPT_FUSED_LOSS, 0) keeps its own launch
    const bool loss_rides = mean_rides && K <= 8 && pc_opt_fused_loss();
    const LossPro lp = {emb, w.y + (size_t)rP * D, w.y + (size_t)rG * D, B, K, margin, dp_out, dn_out,
                        w.dy + (size_t)rP * D, w.dy + (size_t)rG * D, w.demb};
    if (!loss_rides)
        PC_TRY(triplet_loss_launch(emb, w.y + (size_t)rP * D, w.y + (size_t)rG * D, B, K, D, margin, loss, dp_out,
                                   dn_out, demb, w.dy + (size_t)rP * D, w.dy + (size_t)rG * D, stream, mean_rides ? 0 : 1));
    const HingeMeanJob hm = {dp_out, dn_out, B, margin, loss};
    if (anchor_emb && !out_chain_rides)
        PC_HIP_TRY(hipMemcpyAsync(anchor_emb, emb, (size_t)B * D * 4, hipMemcpyDeviceToDevice, st));

    // the slab sums of ALL weight gradients of the step (attention: 10 few-row products, FFN: dW5, dW3 x 2, dW0) fold in
    // one launch at the end of the call
    TnDefer df;
    tn_defer_init(&df);
    if (phase == -1) df.fork = pc_fork_get(st);              // (the unsplit step: two small launches leave the main queue, common.h PcFork)
    df.adam = adam;                                          // torch.optim.Adam inside the final slab reduce (pc_p2v_train_step_unique_adam)
    // an error return below must not leave work on the side queue that nothing orders before the caller's next use (or
    // free) of the workspace and gradient buffers: join it into the main queue on the way out (a no-op after the normal
    // end of the step, whose reduce launch has joined already)
    struct ForkGuard {
        PcFork* f; hipStream_t st;
        ~ForkGuard() { if (f && f->pending) (void)pc_fork_join(f, 1, st); }
    } fork_guard{df.fork, st};
    if (N > 0) {
        PC_TRY(attention_backward_impl(p, g, w.y + (size_t)rA * D, w.y + (size_t)rN * D, B, N, nbc, slot_row,
                                       slot_row ? nbc - 1 : -1, w.demb, &as, w.dy + (size_t)rA * D,
                                       w.dy + (size_t)rN * D, 0, w.attn_ws, w.attn_bytes, stream, ref_off, ref_slot, 1, &df,
                                       mean_rides ? &hm : nullptr, loss_rides ? &lp : nullptr,
                                       out_chain_rides ? fwd_out_chain : nullptr));
        if (anchor_emb && out_chain_rides)                    // (the embedding exists once the backward's first launch has run)
            PC_HIP_TRY(hipMemcpyAsync(anchor_emb, emb, (size_t)B * D * 4, hipMemcpyDeviceToDevice, st));
    } else {
        PC_HIP_TRY(hipMemsetAsync(g->in_proj_w, 0, 3 * D * D * 4, st));
        PC_HIP_TRY(hipMemsetAsync(g->in_proj_b, 0, 3 * D * 4, st));
        PC_HIP_TRY(hipMemsetAsync(g->out_proj_w, 0, D * D * 4, st));
        PC_HIP_TRY(hipMemsetAsync(g->out_proj_b, 0, D * 4, st));
    }
    PC_TRY(ffn_backward_part1(p, g, table, rows, R, &seg, w.dy, &sv, 0, 0, phase == 1 ? bwd_local : nullptr, w.ffn_ws,
                              w.ffn_bytes, stream, 1, &df));
    if (phase == 1) return launch_tn_reduce_deferred(&df, st);
    PC_TRY(ffn_backward_part2(g, table, rows, R, &seg, &sv, nullptr, 0, nullptr, nullptr, w.ffn_ws, w.ffn_bytes, stream, &df));
    return launch_tn_reduce_deferred(&df, st);
}

extern "C" int pc_p2v_train_step(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* table,
                                 const int32_t* anchor_idx, const int32_t* positive_idx,
                                 const int32_t* negative_idx, const int32_t* neighbor_idx, int B, int N, int K,
                                 float margin, float* loss, float* d_pos, float* d_neg, float* anchor_emb,
                                 void* profile, void* ws, size_t ws_bytes, void* stream) {
    return p2v_step_impl(p, g, table, anchor_idx, positive_idx, negative_idx, neighbor_idx, B * N, nullptr, B, N, K,
                         margin, loss, d_pos, d_neg, anchor_emb, profile, ws, ws_bytes, stream);
}

extern "C" int pc_p2v_train_step_compact(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* table,
                                         const int32_t* anchor_idx, const int32_t* positive_idx,
                                         const int32_t* negative_idx, const int32_t* nb_rows, int n_real,
                                         const int32_t* slot_row, int B, int N, int K, float margin, float* loss,
                                         float* d_pos, float* d_neg, float* anchor_emb, void* profile, void* ws,
                                         size_t ws_bytes, void* stream) {
    if (!slot_row || !nb_rows || N <= 0 || n_real < 0 || n_real > B * N) return PC_EINVAL;
    return p2v_step_impl(p, g, table, anchor_idx, positive_idx, negative_idx, nb_rows, n_real + 1, slot_row, B, N, K,
                         margin, loss, d_pos, d_neg, anchor_emb, profile, ws, ws_bytes, stream);
}

extern "C" int pc_p2v_train_step_compact_sync(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* table,
                                              const int32_t* anchor_idx, const int32_t* positive_idx,
                                              const int32_t* negative_idx, const int32_t* nb_rows, int n_real,
                                              const int32_t* slot_row, int B, int N, int K, float margin, float* loss,
                                              float* d_pos, float* d_neg, float* anchor_emb, int phase,
                                              double* fwd_sums, double* bwd_local, const double* bwd_global, void* ws,
                                              size_t ws_bytes, void* stream) {
    if (!slot_row || !nb_rows || N <= 0 || n_real < 0 || n_real > B * N) return PC_EINVAL;
    if (phase < 0 || phase > 2 || !fwd_sums || !bwd_local || (phase == 2 && !bwd_global)) return PC_EINVAL;
    if ((((uintptr_t)fwd_sums | (uintptr_t)bwd_local | (uintptr_t)bwd_global) & 7)) return PC_ESHAPE;
    return p2v_step_impl(p, g, table, anchor_idx, positive_idx, negative_idx, nb_rows, n_real + 1, slot_row, B, N, K,
                         margin, loss, d_pos, d_neg, anchor_emb, nullptr, ws, ws_bytes, stream, phase, fwd_sums,
                         bwd_local, bwd_global);
}

// Unique-neighbour layout (see pcompanion_hip.h): nb_rows[n_unique + 1] distinct neighbour products then -1,
// nb_weight[n_unique + 1] their multiplicities then the number of padding slots, slot_row[B*N] slot -> row,
// ref_off / ref_slot row -> slots.
extern "C" int pc_p2v_train_step_unique(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* table,
                                        const int32_t* anchor_idx, const int32_t* positive_idx,
                                        const int32_t* negative_idx, const int32_t* nb_rows, const float* nb_weight,
                                        int n_unique, const int32_t* slot_row, const int32_t* ref_off,
                                        const int32_t* ref_slot, int B, int N, int K, float margin,
                                        float* loss, float* d_pos, float* d_neg, float* anchor_emb, void* profile,
                                        int phase, double* fwd_sums, double* bwd_local, const double* bwd_global,
                                        void* ws, size_t ws_bytes, void* stream) {
    if (!slot_row || !nb_rows || !nb_weight || !ref_off || !ref_slot || N <= 0 || n_unique < 0 || n_unique > B * N)
        return PC_EINVAL;
    if (phase < -1 || phase > 2) return PC_EINVAL;
    if (phase >= 0 && (!fwd_sums || !bwd_local || (phase == 2 && !bwd_global))) return PC_EINVAL;
    return p2v_step_impl(p, g, table, anchor_idx, positive_idx, negative_idx, nb_rows, n_unique + 1, slot_row, B, N, K,
                         margin, loss, d_pos, d_neg, anchor_emb, profile, ws, ws_bytes, stream, phase, fwd_sums,
                         bwd_local, bwd_global, nb_weight, ref_off, ref_slot);
}

static int unique_adam_impl(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* table, const int32_t* anchor_idx,
                            const int32_t* positive_idx, const int32_t* negative_idx, const int32_t* nb_rows, const float* nb_weight,
                            int n_unique, const int32_t* slot_row, const int32_t* ref_off, const int32_t* ref_slot, int B, int N, int K,
                            float margin, float* loss, float* d_pos, float* d_neg, float* anchor_emb, void* profile, void* ws,
                            size_t ws_bytes, const pc_adam_fused* adam, void* stream, const int32_t* rows_ready) {
    if (!slot_row || !nb_rows || !nb_weight || !ref_off || !ref_slot || N <= 0 || n_unique < 0 || n_unique > B * N)
        return PC_EINVAL;
    if (adam && g) {
        // the gradient tensors must be views of the flat buffer the optimizer updates
        const float* gt[4] = {g->w0, g->w3, g->w5, g->out_proj_w};
        for (const float* x : gt)
            if (!x || x < adam->grad || x >= adam->grad + adam->n) return PC_EINVAL;
    }
    return p2v_step_impl(p, g, table, anchor_idx, positive_idx, negative_idx, nb_rows, n_unique + 1, slot_row, B, N, K,
                         margin, loss, d_pos, d_neg, anchor_emb, profile, ws, ws_bytes, stream, -1, nullptr, nullptr, nullptr,
                         nb_weight, ref_off, ref_slot, adam, rows_ready);
}

extern "C" int pc_p2v_train_step_unique_adam(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* table,
                                             const int32_t* anchor_idx, const int32_t* positive_idx,
                                             const int32_t* negative_idx, const int32_t* nb_rows, const float* nb_weight,
                                             int n_unique, const int32_t* slot_row, const int32_t* ref_off,
                                             const int32_t* ref_slot, int B, int N, int K, float margin, float* loss,
                                             float* d_pos, float* d_neg, float* anchor_emb, void* profile, void* ws,
                                             size_t ws_bytes, const pc_adam_fused* adam, void* stream) {
    return unique_adam_impl(p, g, table, anchor_idx, positive_idx, negative_idx, nb_rows, nb_weight, n_unique, slot_row, ref_off,
                            ref_slot, B, N, K, margin, loss, d_pos, d_neg, anchor_emb, profile, ws, ws_bytes, adam, stream, nullptr);
}

// ... with the step's row indices [anchor | nb_rows[0 .. n_unique] | positive | negatives] already concatenated
// (pc_p2v_concat_step_rows behind the loader's builder): the step's first launch is Linear0
extern "C" int pc_p2v_train_step_unique_rows(const pc_p2v_tensors* p, const pc_p2v_tensors* g, const float* table,
                                             const int32_t* step_rows, const int32_t* nb_rows, const float* nb_weight, int n_unique,
                                             const int32_t* slot_row, const int32_t* ref_off, const int32_t* ref_slot, int B, int N,
                                             int K, float margin, float* loss, float* d_pos, float* d_neg, float* anchor_emb,
                                             void* profile, void* ws, size_t ws_bytes, const pc_adam_fused* adam, void* stream) {
    if (!step_rows || B <= 0 || N <= 0 || n_unique < 0 || n_unique > B * N) return PC_EINVAL;
    const int32_t* a = step_rows;
    return unique_adam_impl(p, g, table, a, a + B + n_unique + 1, a + 2 * B + n_unique + 1, nb_rows, nb_weight, n_unique, slot_row,
                            ref_off, ref_slot, B, N, K, margin, loss, d_pos, d_neg, anchor_emb, profile, ws, ws_bytes, adam, stream,
                            step_rows);
}
