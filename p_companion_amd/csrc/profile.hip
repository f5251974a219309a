// Optional HIP-event brackets around the GEMM launches of one C-ABI call (see common.h).
#include "common.h"

#include <stdlib.h>

thread_local pc_profile* pc_tls_profile = nullptr;
thread_local int pc_tls_launch_err = 0;          // first failed launch since the last pc_launch_status() (common.h)

int pc_prof_begin(int kind, double flops, hipStream_t st) {
    pc_profile* p = pc_tls_profile;
    if (!p || p->used >= p->capacity || !(p->kinds & (1u << kind))) return -1;
    const int b = p->used++;
    p->kind[b] = kind;
    p->flops[b] = flops;
    (void)hipEventRecord(p->ev[2 * b], st);
    return b;
}

void pc_prof_end(int bracket, hipStream_t st) {
    pc_profile* p = pc_tls_profile;
    if (!p || bracket < 0) return;
    (void)hipEventRecord(p->ev[2 * bracket + 1], st);
}

extern "C" int pc_profile_create(int capacity, void** out) {
    if (capacity <= 0 || !out) return PC_EINVAL;
    pc_profile* p = (pc_profile*)calloc(1, sizeof(pc_profile));
    if (!p) return PC_EINVAL;
    p->ev = (hipEvent_t*)calloc((size_t)2 * capacity, sizeof(hipEvent_t));
    p->kind = (int*)calloc(capacity, sizeof(int));
    p->flops = (double*)calloc(capacity, sizeof(double));
    p->capacity = capacity;
    p->kinds = ~0u;
    for (int i = 0; i < 2 * capacity; i++) PC_HIP_TRY(hipEventCreate(&p->ev[i]));
    *out = p;
    return PC_OK;
}

extern "C" int pc_profile_destroy(void* prof) {
    pc_profile* p = (pc_profile*)prof;
    if (!p) return PC_EINVAL;
    for (int i = 0; i < 2 * p->capacity; i++) (void)hipEventDestroy(p->ev[i]);
    free(p->ev); free(p->kind); free(p->flops); free(p);
    return PC_OK;
}

// Every bracket is two event packets between kernels (a few us each, and the bracketed kernel cannot overlap its
// neighbours' tails): a caller that only needs one kernel family restricts the brackets to it.
extern "C" int pc_profile_set_kinds(void* prof, unsigned kinds) {
    if (!prof) return PC_EINVAL;
    ((pc_profile*)prof)->kinds = kinds;
    return PC_OK;
}

extern "C" int pc_profile_reset(void* prof) {
    if (!prof) return PC_EINVAL;
    ((pc_profile*)prof)->used = 0;
    return PC_OK;
}

// After the stream has been synchronised: totals over the recorded brackets of `kind`.
extern "C" int pc_profile_summary(void* prof, int kind, int* launches, double* total_ms, double* total_flops) {
    pc_profile* p = (pc_profile*)prof;
    if (!p || !launches || !total_ms || !total_flops) return PC_EINVAL;
    int n = 0;
    double ms = 0.0, fl = 0.0;
    for (int b = 0; b < p->used; b++) {
        if (p->kind[b] != kind) continue;
        float t = 0.f;
        PC_HIP_TRY(hipEventElapsedTime(&t, p->ev[2 * b], p->ev[2 * b + 1]));
        ms += t; fl += p->flops[b]; n++;
    }
    *launches = n; *total_ms = ms; *total_flops = fl;
    return PC_OK;
}
