// Data-parallel replicas (ABI 6): the gradient exchange of a step issued from the step's own call.
//   pc_exchange_adam        exchange slot + torch.optim.Adam over the flat buffers: optimizer.step() of a replica as one call
//   pc_rccl_*               the library's own RCCL communicator: ncclAllReduce(ncclAvg) as the native pc_exchange_fn,
//                           pc_rccl_alltoall (grouped ncclSend / ncclRecv, constant splits) for the row-sharded table's lookup
//                           rounds, pc_rccl_allreduce_sum_f64 for cross-replica BatchNorm sums -- EVERY collective of a step on
//                           ONE communicator, chained by an event when two of them sit on different streams (see PcComm)
// The reference is single-process (train.py:46-48: loss.backward(); optimizer.step()); a replica averages the gradients
// between the two.  RCCL is resolved at run time (dlopen of the copy the process has loaded already -- torch's -- else the
// system's): the library has no link-time dependency on it and loads on a box without it.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllReduce) all_reduce = nullptr;
    decltype(&ncclReduceScatter) reduce_scatter = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclSend) send = nullptr;
    decltype(&ncclRecv) recv = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    bool ok = false;
};

thread_local char g_last_error[256] = "";

void set_error(const char* what, const char* detail) {
    std::snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, detail ? detail : "?");
}

const RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // the copy already mapped into the process first (RTLD_NOLOAD: torch links its own librccl.so with this soname -- two
        // copies of a collective library in one process would each bring their own device state), then the loader's search path
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (api.handle) break;
        }
        for (int i = 0; !api.handle && i < 2; i++) api.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!api.handle) return;
        api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(dlsym(api.handle, "ncclGetUniqueId"));
        api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(dlsym(api.handle, "ncclCommInitRank"));
        api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(api.handle, "ncclCommDestroy"));
        api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(dlsym(api.handle, "ncclAllReduce"));
        api.reduce_scatter = reinterpret_cast<decltype(api.reduce_scatter)>(dlsym(api.handle, "ncclReduceScatter"));
        api.all_gather = reinterpret_cast<decltype(api.all_gather)>(dlsym(api.handle, "ncclAllGather"));
        api.error_string = reinterpret_cast<decltype(api.error_string)>(dlsym(api.handle, "ncclGetErrorString"));
        api.send = reinterpret_cast<decltype(api.send)>(dlsym(api.handle, "ncclSend"));
        api.recv = reinterpret_cast<decltype(api.recv)>(dlsym(api.handle, "ncclRecv"));
        api.group_start = reinterpret_cast<decltype(api.group_start)>(dlsym(api.handle, "ncclGroupStart"));
        api.group_end = reinterpret_cast<decltype(api.group_end)>(dlsym(api.handle, "ncclGroupEnd"));
        api.ok = api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_reduce && api.error_string && api.send &&
                 api.recv && api.group_start && api.group_end && api.reduce_scatter && api.all_gather;
    });
    return api;
}

int rccl_fail(const RcclApi& api, const char* what, ncclResult_t r) {
    set_error(what, api.error_string ? api.error_string(r) : "ncclResult_t != ncclSuccess");
    return PC_ECOMM;
}

// ONE cross-rank launch order for everything a step exchanges.  The replicas run the same program, so every rank ISSUES its
// collectives in the same host order -- but a step has two streams (the loader's side stream carries the lookup all-to-all of a
// batch a few steps ahead, the step's stream the gradient all-reduce), and two collectives that are in flight on two streams at
// once may start in different orders on different ranks: the classic way to deadlock a ring.  So a collective enqueued on a
// stream OTHER than its predecessor's first makes its stream wait for the predecessor.  The device-side order of the
// communicator's collectives is then their host issue order on every rank, whatever the streams do.
// Collectives that all sit on one stream -- the joint step, a replicated table -- never touch the event: stream order is the
// chain, and the step's stream carries no extra packet.  At the FIRST change of stream the event is recorded, then, at the tail
// of the predecessor's stream (everything enqueued there so far: a one-time over-wait); from then on (`multi`) every collective
// records the event right behind itself, on its own stream, and a change of stream waits for exactly that -- the loader's
// look-ahead all-to-all is not serialised behind the step kernels queued after the gradient all-reduce, and no handle of a
// stream the caller may since have destroyed is touched.  One host thread at a time per communicator: `mu`.
struct PcComm {
    ncclComm_t comm;
    int rank, world;
    hipEvent_t done = nullptr;             // behind the latest collective (multi), or recorded at the first change of stream
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    bool multi = false;                    // a change of stream has been seen: every collective records `done` behind itself
    std::mutex mu;
    long long chained = 0;                 // cross-stream waits inserted so far (pc_rccl_comm_stats)
    long long issued = 0;
};

// before enqueuing on `st`: order it behind the communicator's latest collective
int comm_order(PcComm* c, hipStream_t st) {
    if (c->has_last && st != c->last_stream) {
        if (!c->done && hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
            set_error("hipEventCreateWithFlags", "could not create the communicator's ordering event");
            return PC_ECOMM;
        }
        if (!c->multi) {                   // the first change: nothing was recorded behind the predecessor -- its stream's tail, now
            if (hipEventRecord(c->done, c->last_stream) != hipSuccess) { set_error("hipEventRecord", "ordering event"); return PC_ECOMM; }
            c->multi = true;
        }
        if (hipStreamWaitEvent(st, c->done, 0) != hipSuccess) { set_error("hipStreamWaitEvent", "ordering event"); return PC_ECOMM; }
        c->chained++;
    }
    return PC_OK;
}
// after enqueuing on `st`
int comm_issued(PcComm* c, hipStream_t st) {
    c->issued++;
    c->last_stream = st;
    c->has_last = true;
    if (c->multi && hipEventRecord(c->done, st) != hipSuccess) { set_error("hipEventRecord", "ordering event"); return PC_ECOMM; }
    return PC_OK;
}

}  // namespace

extern "C" int pc_rccl_available(void) { return rccl_api().ok ? 1 : 0; }

extern "C" const char* pc_rccl_last_error(void) { return g_last_error; }

extern "C" int pc_rccl_unique_id(void* out) {
    static_assert(sizeof(ncclUniqueId) == 128, "pcompanion_hip.h states 128 bytes");
    if (!out) return PC_EINVAL;
    const RcclApi& api = rccl_api();
    if (!api.ok) { set_error("pc_rccl_unique_id", "librccl.so.1 could not be loaded"); return PC_ECOMM; }
    ncclUniqueId id;
    const ncclResult_t r = api.get_unique_id(&id);
    if (r != ncclSuccess) return rccl_fail(api, "ncclGetUniqueId", r);
    std::memcpy(out, &id, sizeof(id));
    return PC_OK;
}

extern "C" int pc_rccl_comm_create(const void* unique_id, int rank, int world, void** comm_out) {
    if (!unique_id || !comm_out || world < 1 || rank < 0 || rank >= world) return PC_EINVAL;
    const RcclApi& api = rccl_api();
    if (!api.ok) { set_error("pc_rccl_comm_create", "librccl.so.1 could not be loaded"); return PC_ECOMM; }
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = api.comm_init_rank(&comm, world, id, rank);
    if (r != ncclSuccess) return rccl_fail(api, "ncclCommInitRank", r);
    PcComm* c = new PcComm;
    c->comm = comm; c->rank = rank; c->world = world;
    *comm_out = c;
    return PC_OK;
}

extern "C" int pc_rccl_comm_destroy(void* comm) {
    if (!comm) return PC_EINVAL;
    PcComm* c = static_cast<PcComm*>(comm);
    const RcclApi& api = rccl_api();
    const ncclResult_t r = api.ok ? api.comm_destroy(c->comm) : ncclSuccess;
    if (c->done) (void)hipEventDestroy(c->done);
    delete c;
    return r == ncclSuccess ? PC_OK : rccl_fail(api, "ncclCommDestroy", r);
}

// pc_exchange_fn: grad <- mean over the ranks, in place, ordered on `stream` like a kernel.  (ncclAvg: the sum is scaled inside
// the collective -- no separate scaling launch; a ring / tree all-reduce leaves the same bits on every rank.)
extern "C" int pc_rccl_allreduce_mean(void* comm, float* grad, size_t n, void* stream) {
    if (!comm || !grad || n == 0) return PC_EINVAL;
    const RcclApi& api = rccl_api();
    if (!api.ok) { set_error("pc_rccl_allreduce_mean", "librccl.so.1 could not be loaded"); return PC_ECOMM; }
    PcComm* c = static_cast<PcComm*>(comm);
    std::lock_guard<std::mutex> lk(c->mu);
    PC_TRY(comm_order(c, (hipStream_t)stream));
    const ncclResult_t r = api.all_reduce(grad, grad, n, ncclFloat32, ncclAvg, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return rccl_fail(api, "ncclAllReduce", r);
    return comm_issued(c, (hipStream_t)stream);
}

// In-place SUM of n doubles over the ranks (cross-replica BatchNorm: per-segment sums, sums of squares and row counts,
// PC_BN_SYNC_DOUBLES): the same communicator and the same chain as the gradient exchange.
extern "C" int pc_rccl_allreduce_sum_f64(void* comm, double* buf, size_t n, void* stream) {
    if (!comm || !buf || n == 0) return PC_EINVAL;
    const RcclApi& api = rccl_api();
    if (!api.ok) { set_error("pc_rccl_allreduce_sum_f64", "librccl.so.1 could not be loaded"); return PC_ECOMM; }
    PcComm* c = static_cast<PcComm*>(comm);
    std::lock_guard<std::mutex> lk(c->mu);
    PC_TRY(comm_order(c, (hipStream_t)stream));
    const ncclResult_t r = api.all_reduce(buf, buf, n, ncclFloat64, ncclSum, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return rccl_fail(api, "ncclAllReduce(f64)", r);
    return comm_issued(c, (hipStream_t)stream);
}

// The lookup all-to-all of the row-sharded table (north_star: "RCCL all-to-all over xGMI for cross-shard lookups"):
// rank r's bytes [p * bytes_per_peer, (p + 1) * bytes_per_peer) of `send` land in rank p's `recv` at [r * bytes_per_peer, ...).
// Constant splits (the request capacity is agreed once, at construction), so no size exchange precedes it.  One grouped
// ncclSend / ncclRecv per peer: xGMI is point to point, every pair has its own link, and RCCL runs the group as one launch.
extern "C" int pc_rccl_alltoall(void* comm, const void* send, void* recv, size_t bytes_per_peer, void* stream) {
    if (!comm || !send || !recv || bytes_per_peer == 0 || send == recv) return PC_EINVAL;
    const RcclApi& api = rccl_api();
    if (!api.ok) { set_error("pc_rccl_alltoall", "librccl.so.1 could not be loaded"); return PC_ECOMM; }
    PcComm* c = static_cast<PcComm*>(comm);
    std::lock_guard<std::mutex> lk(c->mu);
    PC_TRY(comm_order(c, (hipStream_t)stream));
    ncclResult_t r = api.group_start();
    if (r != ncclSuccess) return rccl_fail(api, "ncclGroupStart", r);
    ncclResult_t bad = ncclSuccess;
    for (int p = 0; p < c->world; p++) {
        r = api.send(static_cast<const char*>(send) + (size_t)p * bytes_per_peer, bytes_per_peer, ncclInt8, p, c->comm, (hipStream_t)stream);
        if (r != ncclSuccess && bad == ncclSuccess) bad = r;
        r = api.recv(static_cast<char*>(recv) + (size_t)p * bytes_per_peer, bytes_per_peer, ncclInt8, p, c->comm, (hipStream_t)stream);
        if (r != ncclSuccess && bad == ncclSuccess) bad = r;
    }
    r = api.group_end();                                         // (always closed: an open group would swallow every later call)
    if (bad != ncclSuccess) return rccl_fail(api, "ncclSend/ncclRecv", bad);
    if (r != ncclSuccess) return rccl_fail(api, "ncclGroupEnd", r);
    return comm_issued(c, (hipStream_t)stream);
}

// ABI 8: the two halves of the sharded optimizer's exchange, in place over a buffer of world * n_per_rank floats.
// ncclReduceScatter is in place when recvbuff == sendbuff + rank * recvcount, ncclAllGather when sendbuff == recvbuff + rank * sendcount.
extern "C" int pc_rccl_reduce_scatter_mean(void* comm, float* buf, size_t n_per_rank, void* stream) {
    if (!comm || !buf || n_per_rank == 0) return PC_EINVAL;
    const RcclApi& api = rccl_api();
    if (!api.ok) { set_error("pc_rccl_reduce_scatter_mean", "librccl.so.1 could not be loaded"); return PC_ECOMM; }
    PcComm* c = static_cast<PcComm*>(comm);
    std::lock_guard<std::mutex> lk(c->mu);
    PC_TRY(comm_order(c, (hipStream_t)stream));
    const ncclResult_t r = api.reduce_scatter(buf, buf + (size_t)c->rank * n_per_rank, n_per_rank, ncclFloat32, ncclAvg, c->comm,
                                              (hipStream_t)stream);
    if (r != ncclSuccess) return rccl_fail(api, "ncclReduceScatter", r);
    return comm_issued(c, (hipStream_t)stream);
}

extern "C" int pc_rccl_all_gather(void* comm, float* buf, size_t n_per_rank, void* stream) {
    if (!comm || !buf || n_per_rank == 0) return PC_EINVAL;
    const RcclApi& api = rccl_api();
    if (!api.ok) { set_error("pc_rccl_all_gather", "librccl.so.1 could not be loaded"); return PC_ECOMM; }
    PcComm* c = static_cast<PcComm*>(comm);
    std::lock_guard<std::mutex> lk(c->mu);
    PC_TRY(comm_order(c, (hipStream_t)stream));
    const ncclResult_t r = api.all_gather(buf + (size_t)c->rank * n_per_rank, buf, n_per_rank, ncclFloat32, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return rccl_fail(api, "ncclAllGather", r);
    return comm_issued(c, (hipStream_t)stream);
}

// {collectives issued, cross-stream waits inserted}: what the tests read to see the chain at work
extern "C" int pc_rccl_comm_stats(void* comm, int64_t* issued, int64_t* chained) {
    if (!comm) return PC_EINVAL;
    PcComm* c = static_cast<PcComm*>(comm);
    std::lock_guard<std::mutex> lk(c->mu);
    if (issued) *issued = c->issued;
    if (chained) *chained = c->chained;
    return PC_OK;
}

extern "C" int pc_exchange_adam_plan(const pc_exchange_plan* plan, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                                     size_t n, int64_t* step_count, int64_t t, float* scalars, double lr, double beta1,
                                     double beta2, double eps, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n == 0 || t < 0) return PC_EINVAL;
    if (t == 0 && (!step_count || !scalars)) return PC_EINVAL;
    size_t lo = 0, len = n;
    const bool shard = plan && plan->shard_optimizer && (plan->world > 1 || plan->reduce_scatter_mean);
    if (shard) {
        if (plan->world < 1 || plan->rank < 0 || plan->rank >= plan->world || n % (size_t)plan->world) return PC_EINVAL;
        if (!plan->reduce_scatter_mean || !plan->all_gather) return PC_EINVAL;
        len = n / (size_t)plan->world;
        lo = (size_t)plan->rank * len;
        PC_TRY(plan->reduce_scatter_mean(plan->ctx, grad, len, stream));
    } else if (plan && plan->all_reduce_mean) {
        PC_TRY(plan->all_reduce_mean(plan->ctx, grad, n, stream));
    }
    if (t > 0) PC_TRY(pc_adam_step_at(param + lo, grad + lo, exp_avg + lo, exp_avg_sq + lo, len, step_count, t, lr, beta1, beta2, eps, stream));
    else PC_TRY(pc_adam_step(param + lo, grad + lo, exp_avg + lo, exp_avg_sq + lo, len, step_count, scalars, lr, beta1, beta2, eps, stream));
    if (shard) PC_TRY(plan->all_gather(plan->ctx, param, len, stream));
    return PC_OK;
}

extern "C" int pc_exchange_adam(pc_exchange_fn exchange, void* exchange_ctx, float* param, float* grad, float* exp_avg,
                                float* exp_avg_sq, size_t n, int64_t* step_count, int64_t t, float* scalars, double lr,
                                double beta1, double beta2, double eps, void* stream) {
    const pc_exchange_plan plan = {exchange, nullptr, nullptr, exchange_ctx, 0, 1, 0};
    return pc_exchange_adam_plan(&plan, param, grad, exp_avg, exp_avg_sq, n, step_count, t, scalars, lr, beta1, beta2, eps, stream);
}
