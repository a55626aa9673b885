// common.hip -- library identification.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <string.h>
#include "savp_hip.h"
#include "opts.h"

extern "C" const char* savp_version(void) { return "savp_hip 0.1 gfx950"; }

// ---- kernel-only timing of one instrumented launch (bench.py) -----------------------------------------------
// savp_prof_arm(start, stop) hands an event pair to the NEXT ring-kernel launch of the calling thread: the launch goes through
// hipExtLaunchKernelGGL, which stamps the events with the dispatch's own begin / end (what rocprofv3's kernel trace reports),
// instead of hipEventRecord markers in front of and behind the dispatch, whose interval also holds the command processor's
// hand-over between packets (10-15 us on a 30 us kernel).
thread_local hipEvent_t g_savp_prof_start = nullptr;
thread_local hipEvent_t g_savp_prof_stop = nullptr;

extern "C" int savp_prof_event_create(void** ev) {
    if (!ev) return SAVP_EINVAL;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return SAVP_ELAUNCH;
    *ev = (void*)e;
    return SAVP_OK;
}
extern "C" int savp_prof_event_destroy(void* ev) { return (ev && hipEventDestroy((hipEvent_t)ev) == hipSuccess) ? SAVP_OK : SAVP_EINVAL; }
extern "C" int savp_prof_arm(void* start, void* stop) {
    if ((start == nullptr) != (stop == nullptr)) return SAVP_EINVAL;
    g_savp_prof_start = (hipEvent_t)start; g_savp_prof_stop = (hipEvent_t)stop;
    return SAVP_OK;
}
extern "C" int savp_prof_armed(void) { return g_savp_prof_start != nullptr; }
extern "C" int savp_prof_elapsed_us(void* start, void* stop, float* us) {
    if (!start || !stop || !us) return SAVP_EINVAL;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return SAVP_ELAUNCH;
    *us = ms * 1e3f;
    return SAVP_OK;
}

// ---- options (opts.h) ----------------------------------------------------------------------------------------
static struct { const char* name; int value; } g_opts[OPT_COUNT] = {
    {"conv_ring", 0}, {"s2dgrad", 1}, {"thin", 1}, {"wgp_cfg", 0}, {"wgp_split", 0}, {"inorm_min_hw", 64}, {"colsum_2stage", 1},
    {"dense_legacy", 0}, {"cdna_legacy", 0}, {"lstm_fused", 1}, {"ring_dma", 1}, {"lstm_q", 0}, {"ring_wwarm", 1},
};
int savp_opt(int id) { return g_opts[id].value; }
extern "C" int savp_set_option(const char* name, int value) {
    if (!name) return SAVP_EINVAL;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(g_opts[i].name, name)) { g_opts[i].value = value; return SAVP_OK; }
    return SAVP_EINVAL;
}
extern "C" int savp_get_option(const char* name, int* value) {
    if (!name || !value) return SAVP_EINVAL;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(g_opts[i].name, name)) { *value = g_opts[i].value; return SAVP_OK; }
    return SAVP_EINVAL;
}

// ---- gradient bucket all-reduce over RCCL (tf_utils.py:450-480 allreduce_grads) ----------------------------------
// The communicator belongs to the caller (rendezvous / ncclCommInitRank happen in the host framework: torch.distributed in this
// repository's own runners).  librccl.so is resolved at the first call so that the kernel library carries no link-time dependency
// on it (single-GPU users never load RCCL).
typedef int (*rccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
extern "C" int savp_allreduce_bucket(void* comm, void* stream, void* buf, int64_t count) {
    if (!comm || !buf || count < 0) return SAVP_EINVAL;
    if (count == 0) return SAVP_OK;
    static rccl_allreduce_fn fn = nullptr;
    if (!fn) {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return SAVP_ELAUNCH;
        fn = (rccl_allreduce_fn)dlsym(h, "ncclAllReduce");
        if (!fn) return SAVP_ELAUNCH;
    }
    // in place, fp32 (ncclFloat32 = 7), sum (ncclSum = 0); the 1/K of average=True is folded into savp_adam's gscale
    return fn(buf, buf, (size_t)count, 7, 0, comm, (hipStream_t)stream) == 0 ? SAVP_OK : SAVP_ELAUNCH;
}
