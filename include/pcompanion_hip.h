/* pcompanion_hip.h -- C ABI of libpcompanion_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the two embedding-learning hot paths of P-Companion.  The
 * reference is pure Python/PyTorch: its "FFI for this path" is the nn.Module surface of
 * src/models/{product2vec,type_transition,item_prediction,p_companion}.py plus the two
 * loop bodies (product2vec.py:126-164, train.py:34-52).  Each entry point below replaces
 * the ATen op sequence of one reference method (cited per function); the Python classes
 * in p_companion_amd/ keep the reference's names/signatures and call these through ctypes.
 *
 * Conventions:
 *   - raw DEVICE pointers + explicit sizes; fp32 data, int32 indices, row-major, dense;
 *   - `stream` is a hipStream_t passed as void* (the caller's current stream); every call
 *     is asynchronous w.r.t. the host and never synchronises, allocates or frees DEVICE MEMORY;
 *   - caller-allocated outputs and workspace (pc_*_workspace_bytes queries);
 *   - return: 0 ok, <0 invalid argument (PC_E*), >0 a hipError_t; nothing throws;
 *   - re-entrant across host threads, streams and devices: two threads may step two models on two
 *     streams of one device concurrently (tests/test_gpu_streams.py runs exactly that, bit-equal to serial).
 *   Exceptions: the host-side helpers (pc_mt_*, pc_rccl_unique_id) take HOST pointers; pc_rccl_comm_create / _destroy are
 *   RCCL's communicator setup (they block until every rank has arrived and RCCL allocates its own buffers).
 *
 * Library-owned device state (the ONE exception to "no global state"; ABI 5 states what ABI 4 did silently):
 *   the unsplit fused Product2Vec step -- pc_p2v_train_step, pc_p2v_train_step_compact, and
 *   pc_p2v_train_step_unique with phase = -1 -- moves three small launches that nothing on `stream` waits
 *   for until later in the step (the attention block's dq chain and its ten few-row weight gradients, the
 *   BatchNorm-backward finalize) onto a SIDE QUEUE: one non-blocking hipStream_t + six timing-disabled
 *   hipEvent_t per (device, calling stream), created on the first such call on that stream (a process-wide
 *   table of 32 entries behind a mutex; when it is full, or creation fails, the step stays on `stream`).
 *   Fork and join are events: when the call returns, everything it enqueued -- on either queue -- is ordered
 *   before whatever the caller enqueues on `stream` next, so a caller that synchronises `stream` or records an
 *   event on it sees no difference; results are bit-identical either way (tested).  On an error return the
 *   side queue is joined into `stream` first (no work is left behind that could outlive the caller's buffers).
 *   Not used: on a stream that is being captured into a graph; by any other entry point; when switched off.
 *     pc_set_option(PC_OPT_SIDE_QUEUE, 0 / 1)   process-wide switch (default 1); takes effect at the next step
 *     pc_release_device_state()                 destroys every side queue and its events (call with no pc_*
 *                                               call in flight and the streams drained, e.g. before
 *                                               hipDeviceReset or at interpreter exit); they are re-created on demand
 *   No environment variable is read anywhere in the library.
 */
#ifndef PCOMPANION_HIP_H
#define PCOMPANION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PC_OK 0
#define PC_EINVAL (-1)   /* null pointer / non-positive size */
#define PC_ESHAPE (-2)   /* dimension not supported by the kernels (see each function) */
#define PC_EWORKSPACE (-3)
#define PC_EBATCHNORM (-4) /* a BatchNorm call group of ONE row in training mode: nn.BatchNorm1d raises
                             "Expected more than 1 value per channel when training" (product2vec.py:17,39) */
#define PC_ECOMM (-5)    /* the RCCL library could not be loaded, or a communicator call failed (pc_rccl_last_error) */

#define PC_D 128         /* PRODUCT_EMB_DIM (config.py:8) */
#define PC_H 256         /* HIDDEN_SIZE (config.py:10) */
#define PC_HEADS 4       /* NUM_ATTENTION_HEADS (config.py:11) */
#define PC_L 64          /* TYPE_EMB_DIM (config.py:9) */
#define PC_MAX_SEG 4     /* BatchNorm call groups per launch (anchor, neighbours, positive, negative) */
/* doubles in one cross-replica BatchNorm exchange buffer: [seg][2][H] sums then [seg] row counts */
#define PC_BN_SYNC_DOUBLES (PC_MAX_SEG * 2 * 256 + PC_MAX_SEG)

/* 3: pc_ffn_saved gained the optional `a1` member (round 2; a caller built against version 2 passes a shorter struct);
 * 2: pc_p2v_tensors / pc_joint_tensors gained `dropout` (and `dim`). */
/* 5: pc_set_option / pc_get_option / pc_release_device_state (the side queue of the fused Product2Vec step is part of the
 *    contract; the PC_NO_FORK environment variable of version 4 is gone). */
/* 6: the data-parallel exchange slot (pc_exchange_fn, pc_exchange_adam, pc_joint_train_epoch_dp) and the library's own RCCL
 *    communicator behind it (pc_rccl_*): the gradient exchange of a replica is issued from the step's own call, on the step's
 *    stream, not from a host-language hook per step. */
/* 7: pc_rccl_alltoall / pc_rccl_allreduce_sum_f64 / pc_rccl_comm_stats: every collective of a step on the library's ONE
 *    communicator, chained across streams (the lookup all-to-all used to be torch.distributed's, on a second communicator). */
/* 8: the optimizer SHARDED over the replicas (pc_exchange_plan, pc_exchange_adam_plan, pc_joint_train_epoch_plan,
 *    pc_rccl_reduce_scatter_mean, pc_rccl_all_gather): reduce-scatter of the flat gradient, Adam on this rank's 1/world of it,
 *    all-gather of the updated parameters -- instead of an all-reduce and the full dense Adam on every rank (the [num_types,64]
 *    tables of src/models/p_companion.py:36-43 are 17.8 MB at config.py:27's num_types = 34800). */
#define PC_ABI_VERSION 8
int pc_abi_version(void);
/* Process-wide options.  PC_OPT_SIDE_QUEUE: 1 (default) = the unsplit fused Product2Vec step may use its side queue
 * (see "Library-owned device state" above), 0 = every launch stays on the caller's stream.
 * PC_OPT_SORTED_TABLE_GRADIENTS (the fused joint step at num_types > 512; src/models/p_companion.py:36-43): how the gradients of
 * the two [num_types,64] tables are summed.  1 (default) = the sorted form: the source rows are sorted by destination and every
 * destination's run is added in ascending source order -- bitwise reproducible at ANY number of touched rows (with hidden-layer
 * dropout every sample selects its own K types: thousands).  0 = the sorted form only with hidden-layer dropout; without it the
 * per-workgroup LDS tables of the touched rows, reproducible up to 512 touched rows per table (float atomics beyond) -- the
 * default of ABI 7's first builds, 13 us slower per step at num_types = 34800, batch 4096, kept for comparison.  Lists that do
 * not fit the sort kernel's LDS (num_types > 65535, more than 24576 source rows) take the LDS-table form either way.
 * PC_OPT_BN_FINALIZE_SIDE (ABI 8): where the fused Product2Vec step runs its BatchNorm-backward finalize (BatchNorm1d of
 * product2vec.py:16; dgamma / dbeta and the coefficients dW0's loader applies).  0 (default) = on the step's own queue between
 * dZ1 and dW3 (12-14 us there, no hops); 1 = on the side queue beside dW3 (rounds 3-5: the kernel hidden, two ~7 us cross-queue hops exposed), kept for
 * comparison.
 * PC_OPT_FUSED_LOSS (ABI 8): 1 (default) = the fused Product2Vec step at PRODUCT_EMB_DIM = 128 forms the triplet hinge of
 * product2vec.py:137-154 and its three input gradients inside the first launch of the attention backward (the 16 samples of a
 * tile compute their own rows: the same arithmetic, the same bits); 0 = as its own launch (rounds 1-5), kept for comparison.
 * PC_OPT_FUSED_OUT_CHAIN (ABI 8; needs PC_OPT_FUSED_LOSS): 1 (default) = the attention forward's last launch (ctx and the
 * out-projection, MultiheadAttention of product2vec.py:23-28,48-68) runs in front of that prologue in the same launch -- per
 * 16-sample tile: out-projection forward, hinge, out-projection backward; 0 = its own launch.  Same bits.
 * Unknown option / value: PC_EINVAL.  Thread-safe. */
enum { PC_OPT_SIDE_QUEUE = 1, PC_OPT_SORTED_TABLE_GRADIENTS = 2, PC_OPT_BN_FINALIZE_SIDE = 3, PC_OPT_FUSED_LOSS = 4,
       PC_OPT_FUSED_OUT_CHAIN = 5 };
int pc_set_option(int option, int value);
int pc_get_option(int option, int* value);
/* Destroys the library-owned device state (side queues and their events) of every device; 0 or a hipError_t. */
int pc_release_device_state(void);
/* Bitmask of the developer knobs this library was compiled with (0 = a production build).  PC_FLAG_EXP_*: a part of a
 * GEMM loop is compiled out to price it (scripts/dev/nt_decompose.sh) -- WRONG numbers by design; *_TIMING: in-kernel
 * clock reads.  Checked by tests/test_abi.py and __graft_entry__.build(). */
enum {
    PC_FLAG_EXP_NO_MFMA = 1, PC_FLAG_EXP_NO_SPLIT = 2, PC_FLAG_EXP_NO_LDSREAD = 4, PC_FLAG_EXP_NO_DMA = 8,
    PC_FLAG_EXP_NO_SLAB = 16, PC_FLAG_EXP_STAGGER = 32, PC_FLAG_EXP_DMA_L2 = 64, PC_FLAG_NT_TIMING = 128,
    PC_FLAG_JOINT_TIMING = 256, PC_FLAG_CHAIN_TIMING = 512, PC_FLAG_EXP_NO_BARRIER = 1024, PC_FLAG_EXP_NO_VMWAIT = 2048
};
unsigned pc_build_flags(void);

/* Training-mode dropout (config.py:12 DROPOUT = 0.1 is live in every reference training step: the attention
 * probabilities of product2vec.py:23-28 and the hidden layer of type_transition.py:13,17).  ATen's dropout stream
 * cannot be reproduced, so the mask is the build's own, counter-based and restated in oracle/philox_oracle.py:
 * element e of the dropped tensor belongs to group e >> 2, the four keep decisions of a group are the four words
 * of Philox4x32-10(counter = (group, stream, offset lo, offset hi), key = seed); word i keeps element 4*group + i
 * iff word >= floor(p * 2^32); kept values are multiplied by fp32 1/(1-p).  stream 0 = attention probabilities
 * [B,HEADS,N] (after the softmax, before the weighted sum: F.multi_head_attention_forward), 1 = hidden layer [B,32].
 * `offset`: the caller's step counter -- forward and backward of one step pass the same value.  p == 0: off. */
typedef struct {
    float p;
    uint64_t seed, offset;
} pc_dropout;

/* ---------------------------------------------------------------------------------
 * Product2Vec parameters, reference state_dict layout (product2vec.py:14-29):
 *   ffn.0 Linear(D->H) w0[H,D] b0[H]; ffn.1 BatchNorm1d(H) gamma/beta/running_*[H];
 *   ffn.3 Linear(H->H) w3[H,H] b3[H]; ffn.5 Linear(H->D) w5[D,H] b5[D];
 *   attention.in_proj_weight[3D,D] in_proj_bias[3D]; out_proj.weight[D,D] .bias[D].
 * The same struct type carries gradients (running_* / num_batches_tracked unused there).
 * --------------------------------------------------------------------------------- */
typedef struct {
    float *w0, *b0, *gamma, *beta, *w3, *b3, *w5, *b5;
    float *in_proj_w, *in_proj_b, *out_proj_w, *out_proj_b;
    float *running_mean, *running_var;
    int64_t *num_batches_tracked;
    pc_dropout dropout;   /* attention-weight dropout of nn.MultiheadAttention(dropout=config.DROPOUT) in
                           * training mode (product2vec.py:23-28); all-zero = off (eval, or DROPOUT = 0) */
    int dim;              /* D = config.PRODUCT_EMB_DIM (product2vec.py:14-29 take it from the config): 128, or 256
                           * (BASELINE configs[4]; head dim 64).  0 = 128.  Shapes above with that D; H stays 256. */
} pc_p2v_tensors;

/* Row groups ("segments") of one FFN launch.  The reference calls the FFN once per tensor
 * (anchor, neighbours, positive, negative: product2vec.py:132-134 via :73,:78) and
 * BatchNorm's batch statistics span exactly the rows of that call; a launch processes the
 * concatenation of up to PC_MAX_SEG such calls, segment s = rows [start[s], start[s+1]). */
typedef struct {
    int nseg;
    int start[PC_MAX_SEG + 1];
    /* Optional row multiplicity (index path): at most ONE row of the launch, `weighted_row`
     * (-1 = none), stands for `weight` identical rows -- the all-zero padding rows collate_fn
     * appends to short neighbour lists (data_loader.py:186-198) are bit-identical inputs, so
     * they are carried once.  count[s] (0 = physical row count) is the LOGICAL number of rows of
     * segment s, i.e. what BatchNorm divides by. */
    int count[PC_MAX_SEG];
    int weighted_row;
    float weight;
    /* General form of the same idea: rows [row_weight_start, row_weight_start + row_weight_rows) carry
     * the multiplicities row_weight[i] (device pointer, NULL = none) -- the unique-neighbour layout, where
     * every distinct product of the neighbour call is one row standing for all its slots. */
    const float *row_weight;
    int row_weight_start, row_weight_rows;
} pc_segments;

/* Saved-for-backward activations of one FFN launch over R rows (caller allocates). */
typedef struct {
    float *h0;        /* [R,H]  Linear0 output (pre-BatchNorm)                           */
    float *a2;        /* [R,H]  tanh(Linear3(...))                                       */
    float *bn_mean;   /* [nseg,H] batch mean per segment                                 */
    float *bn_invstd; /* [nseg,H] 1/sqrt(biased var + 1e-5)                              */
    float *bn_scale;  /* [nseg,H] gamma*invstd                                           */
    float *bn_shift;  /* [nseg,H] beta - mean*gamma*invstd                               */
    float *a1;        /* [R,H] tanh(BN(h0)), OPTIONAL (NULL: recomputed from h0 where needed): when given, the
                       * forward writes it on the way (its Linear3 forms it anyway) and the backward's dW3 reads it */
} pc_ffn_saved;

/* P6: Product2Vec.get_initial_embedding, training mode (product2vec.py:31-46; ffn :14-21).
 *   y[r] = W5 tanh(W3 tanh(BN_s(W0 x_r + b0)) + b3) + b5,  x_r = idx ? table[idx[r]] : table[r]
 * idx[r] == -1 gathers an all-zero row (collate_fn's zero padding, data_loader.py:186-198).
 * Updates running_mean/var once per segment in segment order (momentum 0.1, unbiased var)
 * and num_batches_tracked += nseg when `update_running` != 0.
 * D=128, H=256 only.  ws: pc_p2v_ffn_workspace_bytes(R). */
size_t pc_p2v_ffn_workspace_bytes(int rows);
int pc_p2v_ffn_forward_train(const pc_p2v_tensors *p, const float *table, const int32_t *idx,
                             int rows, const pc_segments *seg, int update_running,
                             float *y, const pc_ffn_saved *saved, void *ws, size_t ws_bytes,
                             void *stream);

/* P6 / P11 eval mode: BatchNorm uses running statistics (product2vec.py:83-111 runs the
 * model under self.eval()).  No saved activations. */
int pc_p2v_ffn_forward_eval(const pc_p2v_tensors *p, const float *table, const int32_t *idx,
                            int rows, float *y, void *ws, size_t ws_bytes, void *stream);

/* Backward of P6 (what autograd derives for ffn :14-21).  dy[R,D] -> g->{w0,b0,gamma,beta,
 * w3,b3,w5,b5}: overwritten when accumulate == 0, += otherwise.  dx (may be NULL) receives
 * d(loss)/d(input rows) [R,D] for dense-tensor callers whose autograd graph continues below
 * the features; the index path passes NULL (the feature table is frozen input data,
 * synthetic_data.py:50-58).  `dy` is not modified. */
int pc_p2v_ffn_backward(const pc_p2v_tensors *p, const pc_p2v_tensors *g, const float *table,
                        const int32_t *idx, int rows, const pc_segments *seg, const float *dy,
                        const pc_ffn_saved *saved, float *dx, int accumulate, void *ws,
                        size_t ws_bytes, void *stream);

/* P7: Product2Vec.apply_attention -> nn.MultiheadAttention(128, 4 heads), ONE query token
 * per sample, keys == values == neighbour embeddings, no key-padding mask, attention-weight dropout p->dropout
 * (product2vec.py:48-68).  query[B,D], keys[B*N,D] -> out[B,D].
 * With ONE query token the K and V projections of the key rows are absorbed into the per-sample side (exact algebra,
 * csrc/attention.hip): score = (Wk_h^T q_h / sqrt(hd)) . key + const, ctx_h = Wv_h (sum_n pm_n key_n) + bv_h sum_n pm_n;
 * no [B*N, 2D] K|V buffer exists.
 * Saved for backward: q[B,D] (projected, unscaled), qt[B,HEADS,D] (Wk_h^T q_h), probs[B,HEADS,N] (before dropout),
 * c[B,HEADS,D] (sum_n pm_n key_n), sp[B,HEADS] (sum_n pm_n), ctx[B,D] (pre-out_proj).
 * ws: pc_p2v_attention_workspace_bytes(B,N). */
typedef struct {
    float *q, *qt, *probs, *c, *sp, *ctx;
} pc_attn_saved;
size_t pc_p2v_attention_workspace_bytes(int batch, int n_keys);                       /* D = 128 */
size_t pc_p2v_attention_workspace_bytes_dim(int batch, int n_keys, int dim);          /* D = 128 or 256 */
int pc_p2v_attention_forward(const pc_p2v_tensors *p, const float *query, const float *keys,
                             int batch, int n_keys, float *out, const pc_attn_saved *saved,
                             void *ws, size_t ws_bytes, void *stream);
/* Backward of P7: dout[B,D] -> dquery[B,D], dkeys[B*N,D]; g->{in_proj_w,in_proj_b,
 * out_proj_w,out_proj_b} overwritten (accumulate == 0) or +=. */
int pc_p2v_attention_backward(const pc_p2v_tensors *p, const pc_p2v_tensors *g,
                              const float *query, const float *keys, int batch, int n_keys,
                              const float *dout, const pc_attn_saved *saved, float *dquery,
                              float *dkeys, int accumulate, void *ws, size_t ws_bytes,
                              void *stream);

/* P9: the loss of Product2Vec.train_model (product2vec.py:137-154), forward + backward:
 *   d+ = ||a - p + 1e-6||, d- = mean_j ||a - n_j + 1e-6||, loss = mean_b relu(margin - d+ + d-)
 * a[B,D], p[B,D], n[B*K,D] (row b*K+j).  Outputs: loss[1], d_pos[B], d_neg[B] and, when
 * non-NULL, da/dp/dn = d(loss)/d(.)  (already including the 1/B of the mean). */
int pc_p2v_triplet_loss(const float *a, const float *p, const float *n, int batch, int k_neg,
                        float margin, float *loss, float *d_pos, float *d_neg, float *da,
                        float *dp, float *dn, void *stream);                                   /* D = 128 */
int pc_p2v_triplet_loss_dim(const float *a, const float *p, const float *n, int batch, int k_neg, int dim,
                            float margin, float *loss, float *d_pos, float *d_neg, float *da,
                            float *dp, float *dn, void *stream);                               /* D = 128 or 256 */

/* P10: torch.optim.Adam (scripts/pretrain_product2vec.py:34, train.py:24; defaults beta
 * (0.9,0.999), eps 1e-8), no weight decay / amsgrad, bias-corrected, dense over `n` floats.
 * `step_count` is a DEVICE int64 incremented by the call (so a captured graph replays);
 * `scalars` is a DEVICE float[2] scratch.  All four arrays 16-byte aligned. */
int pc_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                 int64_t *step_count, float *scalars, double lr, double beta1, double beta2,
                 double eps, void *stream);
/* The same update as ONE launch when the caller knows the step number t (>= 1) on the host -- an optimizer that owns every
 * increment of its counter: the bias corrections are formed per workgroup by the same fp64 expressions (same bits as
 * pc_adam_step), *step_count (may be NULL) is left = t. */
int pc_adam_step_at(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, size_t n,
                    int64_t *step_count, int64_t t, double lr, double beta1, double beta2, double eps,
                    void *stream);

/* P9 whole: one iteration of Product2Vec.train_model's loop body (product2vec.py:126-159)
 * in index form: gather -> 4 FFN calls -> attention -> loss -> backward -> grads in `g`
 * (overwritten, i.e. zero_grad + backward).  The optimizer step is a separate call.
 *   anchor_idx[B], positive_idx[B], negative_idx[B*K], neighbor_idx[B*N] (-1 = zero row)
 * Outputs: loss[1] (+ d_pos/d_neg[B], anchor_emb[B,D] when non-NULL).  profile: NULL, or a
 * pc_profile_create handle (see below).
 * ws: pc_p2v_train_step_workspace_bytes(B,N,K). */
size_t pc_p2v_train_step_workspace_bytes(int batch, int n_nbr, int k_neg);                  /* D = 128 */
size_t pc_p2v_train_step_workspace_bytes_dim(int batch, int n_nbr, int k_neg, int dim);     /* D = 128 or 256 (p->dim) */
int pc_p2v_train_step(const pc_p2v_tensors *p, const pc_p2v_tensors *g, const float *table,
                      const int32_t *anchor_idx, const int32_t *positive_idx,
                      const int32_t *negative_idx, const int32_t *neighbor_idx, int batch,
                      int n_nbr, int k_neg, float margin, float *loss, float *d_pos,
                      float *d_neg, float *anchor_emb, void *profile, void *ws, size_t ws_bytes,
                      void *stream);

/* Same step with the neighbour rows COMPACTED: collate_fn pads short neighbour lists with all-zero
 * rows (data_loader.py:186-198) which the reference pushes through the FFN, BatchNorm and the
 * attention like any other row.  They are bit-identical inputs, hence bit-identical rows at every
 * layer, so the step carries them ONCE: nb_rows[n_real+1] = the real neighbours of the batch in slot
 * order followed by one -1 row; slot_row[B*N] = row of each slot (padding slots -> n_real).  The
 * shared row enters BatchNorm's statistics with weight B*N - n_real and collects the summed gradient
 * of the slots it stands for (the backward is linear in it): same numbers as the dense step up to
 * fp32 summation order, ~30% fewer rows at the benchmark's degree distribution.
 * Built by pc_build_similarity_batch_compact (row_off: device scratch int32[B+1]). */
int pc_p2v_train_step_compact(const pc_p2v_tensors *p, const pc_p2v_tensors *g, const float *table,
                              const int32_t *anchor_idx, const int32_t *positive_idx,
                              const int32_t *negative_idx, const int32_t *nb_rows, int n_real,
                              const int32_t *slot_row, int batch, int n_nbr, int k_neg, float margin,
                              float *loss, float *d_pos, float *d_neg, float *anchor_emb, void *profile,
                              void *ws, size_t ws_bytes, void *stream);

/* The same loop body (product2vec.py:126-159) in the unique-neighbour layout: the co-view neighbours
 * (bpg.get_neighbors, bpg.py:24-38, gathered per sample by data_loader.py:71-88) of a batch repeat (about a third of the real slots of a
 * 4096-anchor batch over 100 k products); identical table rows are identical FFN rows inside one BatchNorm
 * call, so -- exactly as for the zero-padding rows -- every DISTINCT neighbour product is carried once:
 *   nb_rows[n_unique + 1]    the distinct products (ascending) then -1 (the shared padding row)
 *   nb_weight[n_unique + 1]  how many slots each stands for; last entry = number of padding slots
 *   slot_row[B * N]          row of every (sample, slot)
 *   ref_off / ref_slot       the inverse: slots of every row, ascending (see pc_build_similarity_batch_unique)
 * (pc_build_similarity_batch_unique builds all of them.)  A row's multiplicity weights its BatchNorm sums and
 * the BatchNorm-backward correction; the gradients of the slots that share a row are summed: the attention
 * backward forms dK/dV row by row over the row's slots in ascending slot order (it sorts the row's list:
 * fixed summation order, no float atomics; a row with more than 64 slots is summed in fp64 instead).  Loss and gradients equal the dense step's up to fp32 summation order (tested).
 * phase: -1 = the whole step with this replica's BatchNorm statistics (fwd_sums / bwd_* ignored); 0,1,2 = the
 * cross-replica phases described at pc_p2v_train_step_compact_sync. */
int pc_p2v_train_step_unique(const pc_p2v_tensors *p, const pc_p2v_tensors *g, const float *table,
                             const int32_t *anchor_idx, const int32_t *positive_idx,
                             const int32_t *negative_idx, const int32_t *nb_rows, const float *nb_weight,
                             int n_unique, const int32_t *slot_row, const int32_t *ref_off,
                             const int32_t *ref_slot, int batch, int n_nbr, int k_neg,
                             float margin, float *loss, float *d_pos, float *d_neg, float *anchor_emb,
                             void *profile, int phase, double *fwd_sums, double *bwd_local,
                             const double *bwd_global, void *ws, size_t ws_bytes, void *stream);

/* The same loop body (product2vec.py:126-159; its four BatchNorm1d calls are ffn[1] of :14-21 reached through
 * :73,:78) with cross-replica BatchNorm statistics for data-parallel replicas ("all_reduce(sum) of
 * [2,256] sums + row count, once per BN call"): the same step, cut at the two points where BatchNorm needs
 * batch-wide sums.  Every replica runs
 *   phase 0   batch rows + Linear0 + per-segment sums          -> fwd_sums  (this replica)
 *   [host: all-reduce(SUM) fwd_sums over the replicas, in place]
 *   phase 1   statistics from fwd_sums (all replicas' rows), rest of the forward, loss, backward up to the
 *             BatchNorm-backward sums                          -> bwd_local (this replica)
 *   [host: bwd_global = all-reduce(SUM) of bwd_local, kept beside it]
 *   phase 2   dgamma/dbeta from bwd_local, BN backward with the means of bwd_global, dW0/db0
 * Buffers are PC_BN_SYNC_DOUBLES fp64 each ([seg][2][256] sums: forward sum x, sum x^2; backward sum dz,
 * sum dz*h0; then [seg] row counts), device memory.  The workspace carries the step's state between the
 * phases and must not be touched in between.  Gradients are those of THIS replica's mean loss under
 * batch-wide statistics: averaging them over replicas gives the single-device gradient of the concatenated
 * batch (tested).  Running statistics are updated from the batch-wide values. */
int pc_p2v_train_step_compact_sync(const pc_p2v_tensors *p, const pc_p2v_tensors *g, const float *table,
                                   const int32_t *anchor_idx, const int32_t *positive_idx,
                                   const int32_t *negative_idx, const int32_t *nb_rows, int n_real,
                                   const int32_t *slot_row, int batch, int n_nbr, int k_neg, float margin,
                                   float *loss, float *d_pos, float *d_neg, float *anchor_emb, int phase,
                                   double *fwd_sums, double *bwd_local, const double *bwd_global,
                                   void *ws, size_t ws_bytes, void *stream);
int pc_build_similarity_batch_compact(const int32_t *pair_ids, int batch, const int32_t *sim_pairs,
                                      const int32_t *cv_rowptr, const int32_t *cv_col,
                                      const int32_t *sim_rowptr, const int32_t *sim_col,
                                      int n_products, int n_pad, int k_neg, uint64_t seed,
                                      uint64_t step, int32_t *anchor_idx, int32_t *positive_idx,
                                      int32_t *negative_idx, int32_t *nb_rows, int32_t *slot_row,
                                      int32_t *row_off, void *stream);

/* Same batch in the unique-neighbour layout (see pc_p2v_train_step_unique).  n_real = number of real
 * neighbour slots (host-known: sum of the capped degrees), the upper bound of n_unique: nb_rows / nb_weight
 * hold n_real + 1 entries; *n_unique (device int) receives the number of distinct products.  Also the inverse
 * map, for the attention backward: ref_slot[n_real] = the real slots (sample * n_pad + position) grouped by
 * row -- in arrival order inside a row: the consumer sorts the few entries of a row before it sums --, and
 * ref_off[n_real + 2] = where each row's slots start (the padding row lists none).  `scratch`:
 * pc_build_similarity_batch_unique_scratch_bytes(n_products, batch * n_pad) bytes whose first 4 * n_products
 * bytes are ZERO-FILLED before the first call and left zeroed by every call (per-product occurrence
 * counters).  Integer work only: deterministic. */
size_t pc_build_similarity_batch_unique_scratch_bytes(int n_products, int max_slots /* batch * n_pad */);
int pc_build_similarity_batch_unique(const int32_t *pair_ids, int batch, const int32_t *sim_pairs,
                                     const int32_t *cv_rowptr, const int32_t *cv_col,
                                     const int32_t *sim_rowptr, const int32_t *sim_col, int n_products,
                                     int n_pad, int k_neg, uint64_t seed, uint64_t step, int n_real,
                                     int32_t *anchor_idx, int32_t *positive_idx, int32_t *negative_idx,
                                     int32_t *nb_rows, float *nb_weight, int32_t *slot_row,
                                     int32_t *ref_off, int32_t *ref_slot, int32_t *n_unique, void *scratch,
                                     size_t scratch_bytes, void *stream);

/* Optional measurement aid (bench.py's roofline leg; NULL everywhere else): a pool of HIP
 * events that pc_p2v_train_step records on its stream around each of its GEMM launches.
 * After synchronising the stream, pc_profile_summary totals launches / milliseconds /
 * algorithmic FLOPs per kernel kind (0 = gemm_nt_kernel, 1 = gemm_tn_kernel). */
int pc_profile_create(int capacity, void **out);
int pc_profile_destroy(void *profile);
/* Restrict the recorded brackets to the kinds whose bit is set (bit 0 gemm_nt_kernel, bit 1 gemm_tn*,
 * bit 2 gemm_nt_small_kernel); default all. */
int pc_profile_set_kinds(void *prof, unsigned kinds);
int pc_profile_reset(void *profile);
int pc_profile_summary(void *profile, int kind, int *launches, double *total_ms,
                       double *total_flops);

/* P1-P4 on device: build one index batch from the CSR graph (bpg.py:24-38 get_neighbors,
 * data_loader.py:27-40 negative sampling rules, :186-198 padding).  pair_ids[B] selects
 * (anchor, positive) = sim_pairs[pair]; neighbours = co-view CSR row of the anchor, right
 * padded with -1 to n_pad; negatives: k distinct uniform draws (Philox4x32-10 keyed by
 * seed/step/sample) rejecting the anchor, the anchor's positives (sim CSR) and repeats. */
int pc_build_similarity_batch(const int32_t *pair_ids, int batch, const int32_t *sim_pairs,
                              const int32_t *cv_rowptr, const int32_t *cv_col,
                              const int32_t *sim_rowptr, const int32_t *sim_col, int n_products,
                              int n_pad, int k_neg, uint64_t seed, uint64_t step,
                              int32_t *anchor_idx, int32_t *positive_idx, int32_t *negative_idx,
                              int32_t *neighbor_idx, void *stream);

/* J1 on device: ComplementaryDataset.__getitem__ + collate_fn (data_loader.py:133-157) for `batch`
 * labelled pairs[b] = (query, target, label in {+1,-1}) (int32 x3): query_idx/query_types/
 * pos_types/neg_types [B] by the reference's label rules (:148-151), pos_items/neg_items [B,D] =
 * feat(target) or an N(0,1) filler (torch.randn_like in the reference: input data; here Philox +
 * Box-Muller keyed by seed/step), target_features [B,D] optional. */
int pc_build_complementary_batch(const int32_t *pairs, int batch, const float *features,
                                 const int32_t *type_idx, int n_types, uint64_t seed, uint64_t step,
                                 int32_t *query_idx, int32_t *query_types, int32_t *pos_types,
                                 int32_t *neg_types, float *pos_items, float *neg_items,
                                 float *target_features, void *stream);

/* P2 exact (HOST pointers, host code): SimilarityDataset._get_negative_samples
 * (data_loader.py:27-40) on CPython's `random` stream: MT19937, random.seed(int) key
 * schedule, choice() = _randbelow_with_getrandbits.  Bit-exact negative indices.
 * `state` is an opaque buffer of pc_mt_state_bytes() bytes owned by the caller. */
size_t pc_mt_state_bytes(void);
int pc_mt_seed(void *state, uint64_t seed);
uint32_t pc_mt_getrandbits(void *state, int k);            /* 1 <= k <= 32 */
uint64_t pc_mt_randbelow(void *state, uint64_t n);
int pc_mt_shuffle(void *state, int64_t *perm, int64_t n);  /* random.shuffle of perm */
int pc_mt_negative_samples(void *state, int32_t n_products, const int32_t *sim_rowptr,
                           const int32_t *sim_col, const int32_t *anchors, int64_t n, int k,
                           int32_t *out);

/* ---------------------------------------------------------------------------------
 * P-Companion joint step.  Reference state_dict layout (p_companion.py:26-43,
 * type_transition.py:11-12, item_prediction.py:11-20).
 * --------------------------------------------------------------------------------- */
typedef struct {
    float *product_table;  /* [P,D] frozen (p_companion.py:26-29)                        */
    float *enc_w, *enc_b;  /* type_transition.encoder  [L/2,L], [L/2]                    */
    float *dec_w, *dec_b;  /* type_transition.decoder  [L,L/2], [L]                      */
    float *typ_w, *typ_b;  /* item_prediction.type_projection [D,L], [D]                 */
    float *itm_w, *itm_b;  /* item_prediction.item_projection [D,D], [D]                 */
    float *query_types;    /* query_type_embeddings.weight [T,L]                         */
    float *comp_types;     /* complementary_type_embeddings.weight [T,L]                 */
    pc_dropout dropout;    /* nn.Dropout on the hidden layer in training mode (type_transition.py:13,17);
                            * all-zero = off */
} pc_joint_tensors;

typedef struct {
    float *h;       /* [B,L/2] relu(encoder(E_q[query_types]))   (type_transition.py:17)  */
    float *c;       /* [B,L]   decoder(h)  (complementary base, type_transition.py:19)     */
    float *pi;      /* [B,D]   item_projection(E_prod[query_idx]) (item_prediction.py:31)  */
    float *tp;      /* [B*K,D] type_projection(E_c[topk])        (item_prediction.py:35)  */
} pc_joint_saved;

/* J3+J4+J5: PCompanion.forward (p_companion.py:45-77) with integer ids (the str->idx map
 * of :47-49 stays host-side).
 *   sims[B,T] = dec(relu(enc(E_q[query_types]))) . E_c^T ; topk[B,K] (int32, descending,
 *   ties -> lower index first); proj[B,K,D] = item_proj(E_prod[query_idx])[:,None,:] *
 *   type_proj(E_c[topk]).  D=128, L=64, K<=8, dropout 0.  The nn.Embedding lookups are
 *   fused into the GEMM loaders as row gathers. */
size_t pc_joint_workspace_bytes(int batch, int num_types, int k);
int pc_joint_forward(const pc_joint_tensors *p, const int32_t *query_idx,
                     const int32_t *query_types, int batch, int num_types, int k, float *sims,
                     int32_t *topk, float *proj, const pc_joint_saved *saved, void *ws,
                     size_t ws_bytes, void *stream);

/* J6: PCompanion.compute_loss (p_companion.py:79-119), forward + backward w.r.t. its two
 * differentiable inputs:  type = mean_b clamp(margin - S[b,pos_b] + S[b,neg_b], 0);
 * item = mean_{b,k} clamp(margin - ||proj-pos_item|| + ||proj-neg_item||, 0);
 * loss = alpha*item + (1-alpha)*type.  losses[3] = {loss, type, item}.
 * dsims is SPARSE: dsims_val[B,2] holds d(loss)/dS at columns (pos_b, neg_b) -- the only
 * non-zeros (the reference materialises a dense [B,T] zero gradient: SURVEY section 6).
 * dsims_val / dproj may be NULL (forward only).  partials: device scratch float[2*B]. */
int pc_joint_loss(const float *sims, const float *proj, const int32_t *pos_types,
                  const int32_t *neg_types, const float *pos_items, const float *neg_items,
                  int batch, int num_types, int k, float margin, float alpha, float *losses,
                  float *dsims_val, float *dproj, float *partials, void *stream);
int pc_joint_loss_dim(const float *sims, const float *proj, const int32_t *pos_types, const int32_t *neg_types,
                      const float *pos_items, const float *neg_items, int batch, int num_types, int k, int dim,
                      float margin, float alpha, float *losses, float *dsims_val, float *dproj, float *partials,
                      void *stream);                                                   /* PRODUCT_EMB_DIM 128 or 256 */

/* Dense form of the sparse type-hinge gradient, for callers whose autograd graph needs
 * d(loss)/d(type_similarities) as a [B,T] tensor (what p_companion.py:96-97's advanced
 * indexing produces in the reference): dense = 0; dense[b,pos_b] += v[b,0]; dense[b,neg_b] += v[b,1]. */
int pc_expand_type_grad(const float *dsims_val, const int32_t *pos_types,
                        const int32_t *neg_types, int batch, int num_types, float *dense,
                        void *stream);

/* Backward of J3-J5 from (dsims_val, dproj) -> g (overwritten = zero_grad + backward),
 * including the row-sparse scatter-add into the two [T,L] type tables (J7): only rows
 * query_types[b], topk[b,:], pos_types[b], neg_types[b] receive gradient. */
int pc_joint_backward(const pc_joint_tensors *p, const pc_joint_tensors *g,
                      const int32_t *query_idx, const int32_t *query_types,
                      const int32_t *pos_types, const int32_t *neg_types, const int32_t *topk,
                      int batch, int num_types, int k, const float *dsims_val,
                      const float *dproj, const pc_joint_saved *saved, void *ws,
                      size_t ws_bytes, void *stream);

/* J8 loop body (train.py:42-46): forward + loss + backward; Adam is pc_adam_step. */
int pc_joint_train_step(const pc_joint_tensors *p, const pc_joint_tensors *g,
                        const int32_t *query_idx, const int32_t *query_types,
                        const int32_t *pos_types, const int32_t *neg_types,
                        const float *pos_items, const float *neg_items, int batch,
                        int num_types, int k, float margin, float alpha, float *losses,
                        int32_t *topk, void *ws, size_t ws_bytes, void *stream);

/* J8 loop body, FUSED (train.py:42-48: forward, compute_loss, zero_grad, backward and -- optionally --
 * optimizer.step as TWO launches at num_types <= 128): one kernel over 16-sample tiles computes the whole per-sample
 * part (row gathers, type transition with dropout, similarities + top-K, item projection, both hinges, the dX chain)
 * and, from the operands it holds in LDS, the tile's share of the gradients of the four Linears and of both [T,64] tables
 * (one-hot products; the type hinge's two dE_c rows per sample ride in the same product: no float atomics, bitwise
 * reproducible) as one slab per tile; one kernel sums the slabs in fixed order into g, forms losses[3] = {loss, type,
 * item} and, when exp_avg / exp_avg_sq are given, applies torch.optim.Adam's update (defaults of train.py:24) to p in the
 * same pass (*step_count is advanced by one).  128 < num_types <= 512: the gradient products run as their own kernel
 * over row buffers (three launches).
 * exp_avg == NULL: gradients only (a data-parallel caller all-reduces g, then pc_adam_step).
 * T <= 512: as above.  T > 512 (config.py:27 NUM_TYPES = 34800): the similarity row and its top-K are formed once per
 * DISTINCT query type of the batch (it is a function of the type alone), the [B,T] matrix never exists; the table
 * gradients are fixed-order sums over the touched rows (per-workgroup partial tables in LDS, added in source order, then a
 * fixed-order fold: bitwise reproducible) while a table has <= 512 touched rows in the batch -- the reference's catalogue
 * has 20 live types, the benchmark's 100 -- and float atomics into the cleared dense g beyond that.  With hidden-layer
 * dropout (p > 0: config.py:12 ships 0.1) c differs from sample to sample: the similarity row and its top-K are then formed
 * per SAMPLE (the [B,64] x [64,T] product of p_companion.py:60-63 through the 32-wide hidden layer on the bf16 matrix cores;
 * still never written), everything else as above (ABI 5; version 4 refused this combination).  Either way the selection is
 * exact (descending, ties -> the lower index, like torch.topk): a first pass keeps the maximum of every 64-type sub-chunk of a
 * row with a rigorous error bound, a second re-forms in fp32 the sub-chunks that can hold one of the K best and selects there.
 * Ids outside their tables (query_idx vs num_products, the three type arrays vs num_types) are clamped, counted into
 * *bad_count (may be NULL) and never dereferenced out of bounds -- the reference raises at the lookup
 * (p_companion.py:48-54), the caller raises when it reads the counter.
 * pc_joint_fused_supported: 1 iff (num_types, k, dropout p) is served (k <= 4; see above), else use pc_joint_train_step. */
size_t pc_joint_fused_workspace_bytes(int batch, int num_types, int k);
int pc_joint_fused_supported(int num_types, int k, float dropout_p);
/* After a fused step at num_types > 512: the touched rows of the two [T,64] table gradients -- ascending distinct row ids
 * rows_comp / rows_query and their counts n_touched (device int32[2]) -- as pointers INTO `ws` (valid until the next step
 * on it).  The row list a data-parallel job exchanges instead of the dense tables (north_star: "reduce-scatter for the
 * sparse grads"; src/models/p_companion.py:36-43 + train.py:46-48): 17.8 MB of dense tables at T = 34800 against
 * 260 B per touched row. */
int pc_joint_fused_touched(void *ws, size_t ws_bytes, int batch, int num_types, int k, const int32_t **rows_comp,
                           const int32_t **rows_query, const int32_t **n_touched);
int pc_joint_fused_step(const pc_joint_tensors *p, const pc_joint_tensors *g, const pc_joint_tensors *exp_avg,
                        const pc_joint_tensors *exp_avg_sq, int64_t *step_count, double lr, double beta1,
                        double beta2, double eps, const int32_t *query_idx, const int32_t *query_types,
                        const int32_t *pos_types, const int32_t *neg_types, const float *pos_items,
                        const float *neg_items, int batch, int num_types, int k, int num_products, float margin,
                        float alpha, float *losses, int32_t *topk, int32_t *bad_count, void *ws, size_t ws_bytes,
                        void *stream);

/* The loader's batch construction (pc_build_complementary_batch, data_loader.py:133-157) and the fused step as ONE call
 * over `batch` labelled pairs[b] = (query, target, label): query_idx .. neg_items are OUTPUT buffers here -- after the call
 * they hold the batch exactly as pc_build_complementary_batch(pairs, ..., seed, step) writes it (same filler bits), for
 * whoever reads the batch afterwards (metrics, logging).  num_types <= 128 or > 512: the step's own kernels derive the ids
 * and the item rows (one launch and a 4 MB round trip less); 128 < num_types <= 512: the builder's launch, then the step.
 * `features` [num_products, 128] is the table the dataset serves item rows from (train.py:115: the exported Product2Vec
 * embeddings), `type_idx` [num_products] the products' type ids, `n_types` the dataset's modulus for the negative type. */
int pc_joint_fused_step_pairs(const pc_joint_tensors *p, const pc_joint_tensors *g, const pc_joint_tensors *exp_avg,
                              const pc_joint_tensors *exp_avg_sq, int64_t *step_count, double lr, double beta1,
                              double beta2, double eps, const int32_t *pairs, const float *features,
                              const int32_t *type_idx, int n_types, uint64_t seed, uint64_t step,
                              int32_t *query_idx, int32_t *query_types, int32_t *pos_types, int32_t *neg_types,
                              float *pos_items, float *neg_items, int batch, int num_types, int k, int num_products,
                              float margin, float alpha, float *losses, int32_t *topk, int32_t *bad_count, void *ws,
                              size_t ws_bytes, void *stream);

/* train.py:36-57 train_epoch (for batch in loader: forward, loss, zero_grad, backward, optimizer.step) over `n_pairs`
 * labelled pairs in epoch order on the device, as ONE call: full batches of `batch` pairs, then (unless drop_last) the
 * ragged rest, each through pc_joint_fused_step_pairs with the loader's step counter first_step + i and the dropout
 * offset p->dropout.offset + i.  losses_out[3 * i ..] = {loss, type, item} of step i, on the device (the reference's
 * loss.item() per step is one device round trip per batch).  The batch buffers (sized for `batch`) hold the last batch
 * afterwards; ws as for pc_joint_fused_step at `batch`.  exp_avg / exp_avg_sq are required. */
int pc_joint_train_epoch(const pc_joint_tensors *p, const pc_joint_tensors *g, const pc_joint_tensors *exp_avg,
                         const pc_joint_tensors *exp_avg_sq, int64_t *step_count, double lr, double beta1,
                         double beta2, double eps, const int32_t *pairs, int64_t n_pairs, const float *features,
                         const int32_t *type_idx, int n_types, uint64_t seed, uint64_t first_step,
                         int32_t *query_idx, int32_t *query_types, int32_t *pos_types, int32_t *neg_types,
                         float *pos_items, float *neg_items, int batch, int drop_last, int num_types, int k,
                         int num_products, float margin, float alpha, float *losses_out, int32_t *topk,
                         int32_t *bad_count, void *ws, size_t ws_bytes, void *stream);

/* --- ABI 8: the optimizer step inside the Product2Vec step's last gradient launch.
 * product2vec.py:156-158 is optimizer.zero_grad(); loss.backward(); optimizer.step(): the step functions above are the first two,
 * pc_adam_step[_at] the third -- one more launch over 0.8 MB behind a step whose last kernel (the slab reduce of the weight
 * gradients) has every gradient in hand.  With `adam` the reduce applies torch.optim.Adam's update to each parameter right
 * behind its gradient (and rider workgroups to the parameters whose gradients other kernels finished): the same expressions as
 * pc_adam_step_at(t), the same bits, one launch fewer.  adam->grad must be the flat buffer `g`'s tensors are views of,
 * param / exp_avg / exp_avg_sq the buffers with the same offsets (n floats, n % 4 == 0, 16-byte aligned); t >= 1 the step number
 * (the host's count: *step_count is set to t).  adam == NULL: pc_p2v_train_step_unique.  phase must be -1 (the unsplit step: a
 * replica that exchanges gradients or BatchNorm sums between the phases keeps its optimizer step apart). */
typedef struct pc_adam_fused {
    float *param, *grad, *exp_avg, *exp_avg_sq;
    size_t n;
    int64_t *step_count;
    int64_t t;
    double lr, beta1, beta2, eps;
} pc_adam_fused;
int pc_p2v_train_step_unique_adam(const pc_p2v_tensors *p, const pc_p2v_tensors *g, const float *table,
                                  const int32_t *anchor_idx, const int32_t *positive_idx,
                                  const int32_t *negative_idx, const int32_t *nb_rows, const float *nb_weight,
                                  int n_unique, const int32_t *slot_row, const int32_t *ref_off,
                                  const int32_t *ref_slot, int batch, int n_nbr, int k_neg, float margin,
                                  float *loss, float *d_pos, float *d_neg, float *anchor_emb, void *profile,
                                  void *ws, size_t ws_bytes, const pc_adam_fused *adam, void *stream);

/* The same step without a launch of its own in front of Linear0 (product2vec.py:73, the first nn.Linear of self.ffn).  The step
 * functions above begin by concatenating the four index arrays of product2vec.py:132-134 into the row list the segmented FFN
 * runs over; a loader that builds its batches on another stream (DeviceSimilarityLoader) does that there instead:
 * pc_p2v_concat_step_rows, queued behind pc_build_similarity_batch_unique, writes rows_out = [anchor_idx[B] | nb_rows[0 .. n_unique]
 * (n_unique read from the DEVICE: n_unique_dev, clamped to nb_capacity - 1) | positive_idx[B] | negative_idx[B*K]];
 * rows_capacity >= B * (2 + K) + nb_capacity entries.  pc_p2v_train_step_unique_rows then takes step_rows = rows_out in
 * place of the three index arrays (n_unique: the host's copy of the same count); the transposed weights of the step ride in
 * the FFN forward's BatchNorm finalize launch.  Same results, bit for bit, as pc_p2v_train_step_unique_adam. */
int pc_p2v_concat_step_rows(const int32_t *anchor_idx, const int32_t *positive_idx, const int32_t *negative_idx,
                            const int32_t *nb_rows, const int32_t *n_unique_dev, int nb_capacity, int batch, int k_neg,
                            int32_t *rows_out, int rows_capacity, void *stream);
int pc_p2v_train_step_unique_rows(const pc_p2v_tensors *p, const pc_p2v_tensors *g, const float *table,
                                  const int32_t *step_rows, const int32_t *nb_rows, const float *nb_weight, int n_unique,
                                  const int32_t *slot_row, const int32_t *ref_off, const int32_t *ref_slot, int batch,
                                  int n_nbr, int k_neg, float margin, float *loss, float *d_pos, float *d_neg,
                                  float *anchor_emb, void *profile, void *ws, size_t ws_bytes, const pc_adam_fused *adam,
                                  void *stream);

/* ---------------------------------------------------------------------------------
 * Data-parallel replicas (the reference is single-process: train.py:46-48 is loss.backward(); optimizer.step()
 * with nothing between -- a replica of a data-parallel job averages the gradients there).  ABI 6.
 *
 * pc_exchange_fn: called between a step's last gradient kernel and its Adam launch, ON THE STEP'S STREAM: it must leave in
 * grad[0 .. n) the mean over the replicas of what it found there, ordered on `stream` like a kernel (enqueue only; a host
 * implementation may block).  Returns 0, or an error code the step's call then returns.  Every replica must make the same
 * sequence of calls (same n): the entry points below call it exactly once per step.
 * pc_rccl_allreduce_mean is the native implementation: ncclAllReduce(ncclAvg) of RCCL over xGMI on the library's own
 * communicator, enqueued on `stream` -- no host round trip, no host-language call per step.
 * --------------------------------------------------------------------------------- */
typedef int (*pc_exchange_fn)(void *ctx, float *grad, size_t n, void *stream);

/* exchange(ctx, grad, n, stream) -- skipped when exchange is NULL -- then torch.optim.Adam's update over the flat
 * buffers as pc_adam_step_at(t) when the caller knows the step number (t >= 1), as pc_adam_step (device counter;
 * `scalars` required) when t == 0: loss.backward() is behind, optimizer.step() of a replica as ONE call
 * (scripts/pretrain_product2vec.py:34 + product2vec.py:157-158; train.py:47-48). */
int pc_exchange_adam(pc_exchange_fn exchange, void *exchange_ctx, float *param, float *grad, float *exp_avg,
                     float *exp_avg_sq, size_t n, int64_t *step_count, int64_t t, float *scalars, double lr,
                     double beta1, double beta2, double eps, void *stream);

/* --- ABI 8: the optimizer state sharded over the replicas ("ZeRO-1"; north_star: "reduce-scatter for the sparse grads").
 * The reference's optimizer is torch.optim.Adam over EVERY parameter of one process (train.py:24,46-48): dense moments and a
 * dense update for the two [num_types,64] tables (src/models/p_companion.py:36-43) whichever rows a batch touched.  G replicas
 * that all-reduce the flat gradient each repeat that whole update (7 passes over 17.8 MB per step at num_types = 34800).
 * Sharded: buf = world slices of n / world floats;
 *   reduce_scatter_mean(ctx, grad, n / world, stream)   leaves in slice `rank` of grad the mean over the ranks of that slice
 *                                                       (the other slices: unspecified), ordered on `stream`;
 *   Adam (pc_adam_step / pc_adam_step_at) over slice `rank` of param / grad / exp_avg / exp_avg_sq only;
 *   all_gather(ctx, param, n / world, stream)           every rank's slice `rank` of param -> all ranks, in place.
 * The same bytes on the wire as the all-reduce (which is these two phases back to back), 1 / world of the update's traffic and
 * of the moments' state that matters (this library keeps the moment buffers whole; only slice `rank` is ever read or written).
 * Adam is elementwise, so the parameters after a step are those of the all-reduce form whenever the two reductions sum in
 * the same order (always for world = 2; a ring all-reduce is this reduce-scatter + all-gather).
 * plan->shard_optimizer == 0 (or world == 1 with reduce_scatter_mean == NULL): the plain form -- all_reduce_mean (may be NULL:
 * no exchange) then Adam over all n.  n must be a multiple of world when sharded (the host pads the flat buffers), else PC_EINVAL.
 * pc_rccl_reduce_scatter_mean / pc_rccl_all_gather are the native members: ncclReduceScatter(ncclAvg) / ncclAllGather in place
 * on the library's communicator, chained like its other collectives. */
typedef int (*pc_shard_collective_fn)(void *ctx, float *buf, size_t n_per_rank, void *stream);
typedef struct pc_exchange_plan {
    pc_exchange_fn all_reduce_mean;            /* the plain form's exchange (NULL: none) */
    pc_shard_collective_fn reduce_scatter_mean;
    pc_shard_collective_fn all_gather;
    void *ctx;                                 /* handed to all three */
    int rank, world;
    int shard_optimizer;                       /* 1: reduce-scatter -> Adam on slice `rank` -> all-gather */
} pc_exchange_plan;
int pc_rccl_reduce_scatter_mean(void *comm, float *buf, size_t n_per_rank, void *stream);
int pc_rccl_all_gather(void *comm, float *buf, size_t n_per_rank, void *stream);
/* pc_exchange_adam with a plan (plan == NULL: no exchange, Adam over all n). */
int pc_exchange_adam_plan(const pc_exchange_plan *plan, float *param, float *grad, float *exp_avg, float *exp_avg_sq,
                          size_t n, int64_t *step_count, int64_t t, float *scalars, double lr, double beta1, double beta2,
                          double eps, void *stream);

/* pc_joint_train_epoch for a replica: every step is the fused step WITHOUT its Adam (gradients only), the exchange, then
 * Adam over the flat buffers -- train.py:36-57 with the replicas' mean gradient, all steps of the epoch enqueued by this one
 * call.  p / g must be views INTO param_flat / grad_flat (the [num_types,64] tables included: at num_types > 512 their dense
 * gradients hold zeros outside the touched rows, so the flat mean is the mean of the dense gradients the reference's autograd
 * would form).  t_first: the Adam step number of the epoch's first step (>= 1), or 0 = read the device counter
 * (adam_scalars: device float[2], required then).  Every replica must run the same number of steps (same n_pairs, batch,
 * drop_last): the exchange is a collective.  exchange == NULL: a single process (no exchange; same bits as
 * pc_joint_train_epoch, tested). */
int pc_joint_train_epoch_dp(const pc_joint_tensors *p, const pc_joint_tensors *g, float *param_flat, float *grad_flat,
                            float *exp_avg_flat, float *exp_avg_sq_flat, size_t n_flat, int64_t *step_count,
                            int64_t t_first, float *adam_scalars, double lr, double beta1, double beta2, double eps,
                            pc_exchange_fn exchange, void *exchange_ctx, const int32_t *pairs, int64_t n_pairs,
                            const float *features, const int32_t *type_idx, int n_types, uint64_t seed,
                            uint64_t first_step, int32_t *query_idx, int32_t *query_types, int32_t *pos_types,
                            int32_t *neg_types, float *pos_items, float *neg_items, int batch, int drop_last,
                            int num_types, int k, int num_products, float margin, float alpha, float *losses_out,
                            int32_t *topk, int32_t *bad_count, void *ws, size_t ws_bytes, void *stream);

/* pc_joint_train_epoch_dp with a plan in place of (exchange, exchange_ctx): per step the fused step without its Adam, then
 * pc_exchange_adam_plan -- at num_types > 512 p_companion_amd passes shard_optimizer = 1. */
int pc_joint_train_epoch_plan(const pc_joint_tensors *p, const pc_joint_tensors *g, float *param_flat, float *grad_flat,
                              float *exp_avg_flat, float *exp_avg_sq_flat, size_t n_flat, int64_t *step_count,
                              int64_t t_first, float *adam_scalars, double lr, double beta1, double beta2, double eps,
                              const pc_exchange_plan *plan, const int32_t *pairs, int64_t n_pairs,
                              const float *features, const int32_t *type_idx, int n_types, uint64_t seed,
                              uint64_t first_step, int32_t *query_idx, int32_t *query_types, int32_t *pos_types,
                              int32_t *neg_types, float *pos_items, float *neg_items, int batch, int drop_last,
                              int num_types, int k, int num_products, float margin, float alpha, float *losses_out,
                              int32_t *topk, int32_t *bad_count, void *ws, size_t ws_bytes, void *stream);

/* The library's own RCCL communicator (librccl.so.1 is resolved at run time with dlopen -- the copy the process has
 * loaded already, e.g. torch's, else the system's; the library has no link-time dependency on it).
 *   pc_rccl_available()          1 if the RCCL entry points could be resolved, else 0
 *   pc_rccl_unique_id(out)       ncclGetUniqueId into 128 HOST bytes (one rank calls it and hands the bytes to the others
 *                                by whatever channel the job has: torch.distributed.broadcast in p_companion_amd/distributed.py)
 *   pc_rccl_comm_create(...)     ncclCommInitRank on the calling thread's current HIP device: a collective over the
 *                                `world` ranks; *comm_out is the handle (the `ctx` of pc_rccl_allreduce_mean)
 *   pc_rccl_comm_destroy(comm)   ncclCommDestroy (streams drained by the caller)
 *   pc_rccl_allreduce_mean       a pc_exchange_fn: in-place ncclAllReduce(ncclFloat32, ncclAvg) of grad[0 .. n) on `stream`;
 *                                every rank receives the same bits
 *   pc_rccl_allreduce_sum_f64    in-place ncclAllReduce(ncclFloat64, ncclSum) of buf[0 .. n): the cross-replica BatchNorm sums
 *                                (ops.p2v_train_step(sync_reduce=...); replaces torch.distributed.all_reduce there)
 *   pc_rccl_alltoall             the lookup all-to-all of the row-sharded table (the reference keeps its table in one
 *                                process: src/models/p_companion.py:20-29, src/data/data_loader.py:45-55): bytes
 *                                [p * bytes_per_peer, (p + 1) * bytes_per_peer) of `send` go to rank p, which finds them in its
 *                                `recv` at [rank * bytes_per_peer, ...); one grouped ncclSend / ncclRecv per peer, constant splits
 *                                (send != recv; both device buffers of world * bytes_per_peer bytes)
 *   pc_rccl_comm_stats           {collectives issued on the communicator, cross-stream waits inserted}
 *   pc_rccl_last_error()         text of the calling thread's last PC_ECOMM (static storage; "" if none)
 * ORDER.  Every collective of a step goes through this ONE communicator, and the communicator chains them: a collective
 * enqueued on a stream other than its predecessor's first makes that stream wait for the predecessor.  Until the first change
 * of stream nothing is recorded (collectives that follow each other on one stream are ordered by it: the joint step, a
 * replicated table); at the first change the event is recorded at the tail of the predecessor's stream; from then on every
 * collective records the event right behind itself on its own stream, so a later change of stream waits for the collective
 * alone, not for whatever was enqueued behind it (the loader's look-ahead all-to-all is not serialised behind the step's
 * kernels).  Streams handed to the communicator must outlive it.
 * The replicas issue the same sequence of calls, so the device-side order of the collectives is the same on every rank
 * whichever streams carry them (the loader's side stream: pc_rccl_alltoall a few batches ahead; the step's stream:
 * pc_rccl_allreduce_mean) -- two collectives of one job never race for the links in different orders on different ranks.
 * One host thread at a time per communicator (calls are serialised by a mutex inside).
 * Returns PC_OK, PC_EINVAL, or PC_ECOMM. */
int pc_rccl_available(void);
int pc_rccl_unique_id(void *out_128_bytes);
int pc_rccl_comm_create(const void *unique_id_128_bytes, int rank, int world, void **comm_out);
int pc_rccl_comm_destroy(void *comm);
int pc_rccl_allreduce_mean(void *comm, float *grad, size_t n, void *stream);
int pc_rccl_allreduce_sum_f64(void *comm, double *buf, size_t n, void *stream);
int pc_rccl_alltoall(void *comm, const void *send, void *recv, size_t bytes_per_peer, void *stream);
int pc_rccl_comm_stats(void *comm, int64_t *issued, int64_t *chained);
const char *pc_rccl_last_error(void);

/* ---------------------------------------------------------------------------------
 * Building blocks the Python modules compose their autograd from (module / dense mode).
 * --------------------------------------------------------------------------------- */
/* nn.Linear forward (type_transition.py:11-12, item_prediction.py:11-20, and the
 * similarity product p_companion.py:60-63 with b = NULL):
 *   y[r] = act(W x_r + b), x_r = idx ? x[idx[r]] (zero row if idx[r] < 0) : x[r]
 * w[out_dim,in_dim]; act 0 none, 1 tanh, 2 relu; in_dim % 4 == 0. */
int pc_linear_forward(const float *x, const int32_t *idx, int rows, int in_dim, const float *w,
                      const float *b, int out_dim, int act, float *y, void *stream);
/* dx[rows,in_dim] = dy[rows,out_dim] W.  wt_scratch: in_dim*out_dim floats (holds W^T).
 * act must be 0 (callers fold act' into dy); y_saved unused. */
int pc_linear_backward_input(const float *dy, int rows, int out_dim, const float *w, int in_dim,
                             int act, const float *y_saved, float *dx, float *wt_scratch,
                             void *stream);
/* dW[out,in] (+)= dy^T x, db[out] (+)= sum_r dy[r] (db may be NULL); x rows gathered by idx
 * when non-NULL.  Fixed-order split-K reduction: bitwise reproducible. out_dim, in_dim % 4 == 0. */
size_t pc_linear_backward_weight_workspace_bytes(int rows, int out_dim, int in_dim);
int pc_linear_backward_weight(const float *dy, int rows, int out_dim, const float *x,
                              const int32_t *idx, int in_dim, float *dw, float *db,
                              int accumulate, void *ws, size_t ws_bytes, void *stream);
/* torch.topk(sims, k, dim=1) (p_companion.py:64; metrics.py:21): idx_out[B,k] int32,
 * val_out[B,k] (may be NULL); descending, ties -> lower index first; k <= 8. */
int pc_topk_rows(const float *sims, int batch, int num_types, int k, int32_t *idx_out,
                 float *val_out, void *stream);
/* Metrics.evaluate_model pieces (src/utils/metrics.py:62-117).
 * pc_hit_rank: rank[r] = number of entries of sims[r,:] that beat the ground-truth column
 *   gt = r (ties towards the lower index); hit@k <=> rank[r] < k.  Rows r >= cols can never
 *   hit (rank = INT_MAX): the reference compares arange(B*K) with B columns (metrics.py:95-100).
 * pc_cosine_rows: out[b*K+k] = cosine_similarity(x[b,k,:], y[b,:]) with torch's eps 1e-8
 *   (metrics.py:44-60).  D = 128. */
int pc_hit_rank(const float *sims, int rows, int cols, int32_t *rank, void *stream);
int pc_cosine_rows(const float *x, const float *y, int batch, int k, float *out, void *stream);
int pc_cosine_rows_dim(const float *x, const float *y, int batch, int k, int dim, float *out, void *stream);

/* item_prediction.py:38: proj[b,k,:] = pi[b,:] * tp[b*K+k,:] and its backward
 * (dpi[b] = sum_k dproj[b,k]*tp[b,k]; dtp[b,k] = dproj[b,k]*pi[b]).  D = 128. */
int pc_hadamard_forward(const float *pi, const float *tp, int batch, int k, float *proj,
                        void *stream);
int pc_hadamard_forward_dim(const float *pi, const float *tp, int batch, int k, int dim, float *proj, void *stream);   /* PRODUCT_EMB_DIM 128 or 256 (item_prediction.py:11-20 takes it from config) */
int pc_hadamard_backward(const float *dproj, const float *pi, const float *tp, int batch, int k,
                         float *dpi, float *dtp, void *stream);
int pc_hadamard_backward_dim(const float *dproj, const float *pi, const float *tp, int batch, int k, int dim,
                             float *dpi, float *dtp, void *stream);

/* ---------------------------------------------------------------------------------
 * Row movers used by the sharded-table exchange and the modules.
 * --------------------------------------------------------------------------------- */
/* out[r] = idx[r] >= 0 ? table[idx[r]] : 0   (nn.Embedding lookup p_companion.py:51,54,65;
 * feature gather of data_loader.py:50-55 in index form).  width % 4 == 0. */
int pc_gather_rows(const float *table, const int32_t *idx, int rows, int width, float *out,
                   void *stream);
/* table[idx[r]] += src[r] (row-sparse embedding gradient, J7), deterministic per row order
 * is NOT guaranteed (float atomics); idx < 0 skipped. */
int pc_scatter_add_rows(float *table, const int32_t *idx, int rows, int width, const float *src,
                        void *stream);
/* Same for a SMALL destination table (table_rows*width*4 <= 64 KB, e.g. the [T,64] type tables
 * at T ~ 100): source rows are first summed into a per-workgroup LDS copy of the table. Falls
 * back to pc_scatter_add_rows for larger tables. */
int pc_scatter_add_rows_small(float *table, int table_rows, const int32_t *idx, int rows, int width,
                              const float *src, void *stream);

/* out[idx[r]] = src[r] (row assignment, idx < 0 skipped): writes the attention-updated rows
 * back into the embedding table in the batched generate_all_embeddings pass
 * (product2vec.py:108-109). width % 4 == 0. */
int pc_scatter_rows(float *out, const int32_t *idx, int rows, int width, const float *src,
                    void *stream);
/* dx = dy * act'(y) for a stand-alone activation: act 1 = tanh (1 - y^2), 2 = relu (y > 0)
 * (F.relu of type_transition.py:17 in module mode). */
int pc_act_backward(const float *dy, const float *y, size_t n, int act, float *dx, void *stream);

/* Zipf(s = 1) negative sampling (BASELINE configs[4]; an extension: the reference draws its negatives uniformly,
 * data_loader.py:34).  Replaces negative_idx[batch,k_neg] of an already built batch: candidate = perm[rank - 1]
 * (perm NULL: product id = rank - 1) with P(rank) ~ 1 / rank over ranks 1..n_products, under the reference's rejection
 * rules (data_loader.py:33-38: not the anchor, not one of its positives, no repeats).  Integer-only: an octave
 * [2^j, 2^(j+1)) is chosen by the first j with draw <= octave_cum[j] (n_octaves = floor(log2(n_products)) + 1 cumulative
 * 32-bit thresholds of the octaves' probability masses, last = 0xFFFFFFFF), a rank in it uniformly (j bits), accepted
 * with probability 2^j / rank.  Philox4x32-10 keyed by (seed ^ "ZIPF"; sample, step). */
int pc_sample_negatives_zipf(const int32_t *pair_ids, int batch, const int32_t *sim_pairs,
                             const int32_t *sim_rowptr, const int32_t *sim_col, int n_products, int k_neg,
                             uint64_t seed, uint64_t step, const uint32_t *octave_cum, int n_octaves,
                             const int32_t *perm, int32_t *negative_idx, int32_t *failed, void *stream);
/* (`failed`, optional device int32: += 1 for every sample that ran out of its 4096 proposals -- an anchor whose positives
 * cover the head of the popularity order, or one with fewer than k_neg eligible products; such a sample's remaining
 * negatives are the first eligible products in rank order, -1 when none is left.  Every wave terminates.) */

/* ---- The synthetic catalogue generated ON THE DEVICE (csrc/generator.hip): SyntheticDataGenerator
 * (src/data/synthetic_data.py:11-153) restated per source node like data.generate_scaled_bpg, but as kernels that write
 * straight into HBM -- BASELINE configs[3]/[4] (10 M / 100 M products) cannot exist as host numpy arrays.  Every array is a
 * pure function of (seed, product id) through Philox4x32-10, so a rank generates exactly its rows of the feature table
 * (rows first, first + stride, ...: the cyclic shard r % world of SURVEY 8e) and the replicated graph arrays are identical
 * on every rank.  Distributions: type uniform (synthetic_data.py:35-46); features N(0,1)^dim + 1.0 on the category's 20-dim
 * block (:50-52); co-view out-degree Poisson(2 mean (1 - u)) capped, targets uniform, different-category targets kept with
 * probability 2/3 (:107-108), distinct per row; similarity = co-view & purchase-after-view (0.2) & not co-purchase (0.075
 * same category / 0.15) (:110-121); complementary = Poisson(comp_mean) targets, same-category kept 0.5, not co-viewed
 * (:122-127).  Call order: pc_gen_degrees -> pc_exclusive_scan_i32 (cv_rowptr) -> pc_gen_coview -> scan (sim_rowptr) ->
 * pc_gen_similarity; pc_gen_complementary(count) -> scan -> pc_gen_complementary(pairs). */
int pc_gen_types(int64_t n_products, int num_types, uint64_t seed, int32_t *type_idx, void *stream);
/* features[k][0:dim] = the feature row of product first + k * stride, k < n_local; dim >= 100, dim % 4 == 0 */
int pc_gen_features(int64_t first, int64_t stride, int64_t n_local, int dim, int num_types, uint64_t seed,
                    float *features, void *stream);
/* deg[P]: co-view out-degrees; comp_cand[P] (optional): complementary candidates per product.  degree_cap <= 64 */
int pc_gen_degrees(int64_t n_products, double mean_degree, int degree_cap, double comp_mean, uint64_t seed,
                   int32_t *deg, int32_t *comp_cand, void *stream);
size_t pc_scan_scratch_bytes(int64_t n);
/* out[i] = sum_{j<i} in[j] for i = 0..n (out has n + 1 entries); total_out (optional): device int64 = out[n] unrounded
 * (check it against 2^31 on the host: the offsets are int32) */
int pc_exclusive_scan_i32(const int32_t *in, int64_t n, int32_t *out, int64_t *total_out, void *scratch,
                          size_t scratch_bytes, void *stream);
/* cv_col[cv_rowptr[i] + j] = j-th co-view target of product i, bit 31 = "this edge is a similarity pair" (cleared by
 * pc_gen_similarity); sim_count[i] = number of such edges */
int pc_gen_coview(int64_t n_products, int num_types, uint64_t seed, const int32_t *cv_rowptr, int32_t *cv_col,
                  int32_t *sim_count, void *stream);
/* sim_pairs[S][2] in source order, sim_col[S] (= the positives' CSR with sim_rowptr), pair_deg[S] (optional): co-view
 * degree of each pair's anchor (what the loader pads to, data_loader.py:186-198) */
int pc_gen_similarity(int64_t n_products, const int32_t *cv_rowptr, int32_t *cv_col, const int32_t *sim_rowptr,
                      int32_t *sim_pairs, int32_t *sim_col, int32_t *pair_deg, void *stream);
/* count != NULL: count[i] = complementary pairs of product i; comp_pairs != NULL: pairs written at comp_rowptr[i] */
int pc_gen_complementary(int64_t n_products, int num_types, uint64_t seed, const int32_t *comp_cand,
                         const int32_t *cv_rowptr, const int32_t *cv_col, int32_t *count,
                         const int32_t *comp_rowptr, int32_t *comp_pairs, void *stream);

/* The epoch order of DataLoader(shuffle=True) (scripts/pretrain_product2vec.py:24-30, train.py:115-121) as a keyed
 * bijection: out[i] = perm_{seed,epoch}(i), i in [0, n) -- a six-round balanced Feistel network over the even number of
 * bits covering n, cycle-walked into [0, n).  No sort, no storage, deterministic in (seed, epoch); restated in
 * oracle/philox_oracle.py::epoch_permutation.  (The reference's order comes from torch's CPU generator and cannot be
 * bit-matched; parity mode replays CPython's random.shuffle on the host instead: pc_mt_shuffle.) */
int pc_epoch_permutation(int n, uint64_t seed, uint64_t epoch, int32_t *out, void *stream);
/* out[i][0:width] = rows[perm_{seed,epoch}(i)][0:width] (int32 rows, e.g. the labelled pairs [n,3] of
 * ComplementaryDataset, data_loader.py:113-126): one epoch's shuffled order in one launch.  out != rows. */
int pc_shuffle_rows_i32(const int32_t *rows, int n, int width, uint64_t seed, uint64_t epoch, int32_t *out,
                        void *stream);
/* collate_fn pads every neighbour list to the batch maximum (data_loader.py:186-198), so the host needs two integers per
 * batch to size it: plan[b] = (max, sum) of deg[order[i]] over positions i in [b*batch, (b+1)*batch), i < n
 * (order NULL: the identity).  plan: device int64 [n_batches][2]. */
int pc_epoch_plan(const int32_t *order, const int32_t *deg, int n, int batch, int n_batches, int64_t *plan,
                  void *stream);

/* Row-sharded feature table, device-resident request bucketing (north_star: "embedding table row-shards across up
 * to 8 MI355X with RCCL all-to-all for cross-shard lookups"; the reference's table is one in-process dict,
 * src/data/bpg.py:4-22, data_loader.py:50-55).  Product r lives on rank r % world as local row r / world.
 * Up to `count` <= 4 id arrays ids[a][n[a]] (global product ids, < 0 = padding; only the first *n_dev[a] + n_dev_add[a]
 * entries are live when n_dev[a] != NULL -- the unique-neighbour list's length exists on the device only) become
 *   send_ids[world][capacity]   per-owner request lists (local row indices; unused slots -1), and
 *   remap_out[a][n[a]]          the same arrays as indices into the [world][capacity][D] buffer the exchange returns
 *                               (row = owner * capacity + slot; padding and non-live entries -> -1).
 * counts[world] receives the bucket sizes; *overflow is incremented for every id that did not fit its bucket (that id
 * maps to -1).  Nothing is read back to the host: shapes are constant, the two all-to-all rounds need no size exchange. */
int pc_shard_bucket(const int32_t *const *ids, const int *n, const int32_t *const *n_dev, const int *n_dev_add,
                    int32_t *const *remap_out, int count, int world, int capacity, int32_t *counts,
                    int32_t *send_ids, int32_t *overflow, void *stream);
/* The same with a REPLICATED HOT SET (ABI 8; BASELINE configs[4]: "Zipf-skewed negative sampling + hot-row cache"; the
 * reference draws negatives uniformly, src/data/data_loader.py:27-40, and keeps its table in one process): the hot_rows most
 * popular products live on EVERY rank as rows [world * capacity, world * capacity + hot_rows) of the table the step reads
 * (the caller keeps that replica behind the exchange buffer), and an id of the set maps there instead of taking a request slot.
 * hot_ids: the set as ascending product ids [hot_rows] (device; NULL = the ids [0, hot_rows): popularity rank = product id, the
 * Zipf sampler's default); hot slot = position in the set.  *hot_served (device, may be NULL) is increased by the number of
 * live entries served from the replica.  hot_rows = 0: pc_shard_bucket.  Lookups stay constant-shape. */
int pc_shard_bucket_hot(const int32_t *const *ids, const int *n, const int32_t *const *n_dev, const int *n_dev_add,
                        int32_t *const *remap_out, int count, int world, int capacity, const int32_t *hot_ids,
                        int hot_rows, int32_t *counts, int32_t *send_ids, int32_t *overflow, int32_t *hot_served,
                        void *stream);

/* nn.Dropout of ComplementaryTypeTransition (type_transition.py:13,17) in training mode on the hidden activations
 * x[n] (n % 4 == 0, rows of 32): y = x * mask, mask = stream 1 of pc_dropout (0 or 1/(1-p)).  The same call with the
 * same (seed, offset) is its backward (dx = dy * mask).  In place (y == x) allowed.  0 < p < 1. */
int pc_dropout_hidden(const float *x, size_t n, const pc_dropout *d, float *y, void *stream);

/* Index validation.  The reference's lookups raise for an id outside its table (nn.Embedding: IndexError;
 * product_to_idx: KeyError -- p_companion.py:48-54); here up to `count` <= 4 int32 index arrays idx[a][n[a]] are
 * checked against [0, hi[a]) (or [-1, hi[a]) where allow_pad[a] != 0: -1 is the collate padding sentinel) in one
 * launch, and the number of offending entries is ADDED to the device counter *bad.  The caller reads the counter
 * when it synchronises anyway (PCompanion.raise_index_errors) and raises IndexError. */
int pc_check_indices(const int32_t *const *idx, const int *n, const int *hi, const int *allow_pad, int count,
                     int32_t *bad, void *stream);

/* ---------------------------------------------------------------------------------
 * Serving: type-filtered top-n retrieval -- the candidate search of
 * PCompanionInference.recommend (inference.py:90-118): for row r (one predicted complementary
 * type of one query), scores = proj[r] . features[c] over the products c of type types[r]
 * (CSR type_rowptr[n_types+1] / type_col, products in node order = bpg.get_products_by_type),
 * torch.topk(scores, min(n, count)).  out_idx / out_score [rows, n]; missing entries (fewer than
 * n products of that type, or a type outside [0, n_types)) are -1 / -inf.  1 <= n <= 16.
 * --------------------------------------------------------------------------------- */
int pc_retrieve_topk(const float *proj, const int32_t *types, int rows, const int32_t *type_rowptr,
                     const int32_t *type_col, const float *table, int n_types, int n, int32_t *out_idx,
                     float *out_score, void *stream);
int pc_retrieve_topk_dim(const float *proj, const int32_t *types, int rows, const int32_t *type_rowptr,
                        const int32_t *type_col, const float *table, int n_types, int n, int dim, int32_t *out_idx,
                        float *out_score, void *stream);

#ifdef __cplusplus
}
#endif
#endif
