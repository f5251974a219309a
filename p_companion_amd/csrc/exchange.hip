// Data-parallel replicas (ABI 6): the gradient exchange of a step issued from the step's own call.
//   pc_exchange_adam        exchange slot + torch.optim.Adam over the flat buffers: optimizer.step() of a replica as one call
//   pc_rccl_*               the library's own RCCL communicator and ncclAllReduce(ncclAvg) as the native pc_exchange_fn
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
        api.error_string = reinterpret_cast<decltype(api.error_string)>(dlsym(api.handle, "ncclGetErrorString"));
        api.ok = api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_reduce && api.error_string;
    });
    return api;
}

int rccl_fail(const RcclApi& api, const char* what, ncclResult_t r) {
    set_error(what, api.error_string ? api.error_string(r) : "ncclResult_t != ncclSuccess");
    return PC_ECOMM;
}

struct PcComm {
    ncclComm_t comm;
    int rank, world;
};

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
    *comm_out = new PcComm{comm, rank, world};
    return PC_OK;
}

extern "C" int pc_rccl_comm_destroy(void* comm) {
    if (!comm) return PC_EINVAL;
    PcComm* c = static_cast<PcComm*>(comm);
    const RcclApi& api = rccl_api();
    const ncclResult_t r = api.ok ? api.comm_destroy(c->comm) : ncclSuccess;
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
    const ncclResult_t r = api.all_reduce(grad, grad, n, ncclFloat32, ncclAvg, c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? PC_OK : rccl_fail(api, "ncclAllReduce", r);
}

extern "C" int pc_exchange_adam(pc_exchange_fn exchange, void* exchange_ctx, float* param, float* grad, float* exp_avg,
                                float* exp_avg_sq, size_t n, int64_t* step_count, int64_t t, float* scalars, double lr,
                                double beta1, double beta2, double eps, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n == 0 || t < 0) return PC_EINVAL;
    if (t == 0 && (!step_count || !scalars)) return PC_EINVAL;
    if (exchange) PC_TRY(exchange(exchange_ctx, grad, n, stream));
    if (t > 0) return pc_adam_step_at(param, grad, exp_avg, exp_avg_sq, n, step_count, t, lr, beta1, beta2, eps, stream);
    return pc_adam_step(param, grad, exp_avg, exp_avg_sq, n, step_count, scalars, lr, beta1, beta2, eps, stream);
}
