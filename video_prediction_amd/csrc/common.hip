// common.hip -- library identification.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <string.h>
#include "savp_hip.h"
#include "opts.h"
#include "conv_common.h"

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
    {"dense_legacy", 0}, {"cdna_legacy", 0}, {"lstm_fused", 1}, {"ring_dma", 1}, {"lstm_q", 0}, {"ring_wwarm", 1}, {"wgp_dma", 1}, {"ring_early", 1}, {"gate_kernel", 1}, {"gate_alt", 0}, {"gate_cell", 1}, {"gate_wwarm", 1}, {"splitk_reduced", 0},
};
int savp_opt(int id) { return g_opts[id].value; }
void savp_opt_count(int id) { ++g_opts[id].value; }
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

// ---- deterministic split-K (conv_common.h) --------------------------------------------------------------------------------------
int splitk_fit(const SavpConvArgs* a, int want, long long block_elems) {
    if (want <= 1) return 1;
    if (!a->ws || a->ws_bytes <= 0 || block_elems <= 0 || (((uintptr_t)a->ws) & 15)) { savp_opt_count(OPT_SPLITK_REDUCED); return 1; }
    const long long fit = a->ws_bytes / (block_elems * (long long)sizeof(float));
    if (fit < want) savp_opt_count(OPT_SPLITK_REDUCED);          // the scratch holds fewer slices than asked for: visible in the option table
    if (fit < 2) return 1;
    return want < fit ? want : (int)fit;
}

__global__ __launch_bounds__(256) void splitk_fold_kernel(float* __restrict__ out, const float* __restrict__ part, int splitk, long long n4, long long n,
                                                          int beta, int Cd, int gap_at, int gap) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 s = beta ? reinterpret_cast<const float4*>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < splitk; ++k) {                       // fixed order: the result does not depend on which split finished first
            const float4 v = reinterpret_cast<const float4*>(part + (long long)k * n)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (gap) {                                               // the gap's channels were computed by nobody (SavpConvArgs.dst_gap)
            const int c = (int)((i * 4) % Cd);
            float* e = &s.x;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c + j >= gap_at && c + j < gap_at + gap) e[j] = beta ? reinterpret_cast<const float*>(out)[i * 4 + j] : 0.f;
        }
        reinterpret_cast<float4*>(out)[i] = s;
    }
}

__global__ __launch_bounds__(256) void splitk_fold_scalar_kernel(float* __restrict__ out, const float* __restrict__ part, int splitk, long long n, int beta,
                                                                 int Cd, int gap_at, int gap) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const int c = (int)(i % Cd);
        if (gap && c >= gap_at && c < gap_at + gap) { if (!beta) out[i] = 0.f; continue; }
        float s = beta ? out[i] : 0.f;
        for (int k = 0; k < splitk; ++k) s += part[(long long)k * n + i];
        out[i] = s;
    }
}

void splitk_fold(float* out, const float* part, int splitk, long long n, int beta, int Cd, int gap_at, int gap, hipStream_t st) {
    const bool vec = (n % 4 == 0) && (Cd % 4 == 0) && ((((uintptr_t)out) & 15) == 0) && ((((uintptr_t)part) & 15) == 0);
    const long long work = vec ? n / 4 : n;
    long long blocks = (work + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (vec) hipLaunchKernelGGL(splitk_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, part, splitk, n / 4, n, beta, Cd, gap_at, gap);
    else hipLaunchKernelGGL(splitk_fold_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, part, splitk, n, beta, Cd, gap_at, gap);
}

// ---- deterministic weight gradients (conv_common.h) ------------------------------------------------------------------------------------
// dW[i] += part[0][i] + part[1][i] + ... in split order: the pixel splits of a weight-gradient launch leave their tiles in slices of the
// caller's scratch (plain stores) instead of meeting in dW with fp32 atomics, whose arrival order changed the sum's rounding from run to run
// (round 5: two identical bench runs forked after a few steps through the bf16 rounding of the updated weights).  Eight slices are
// requested before the first is added: 64 slices of a 1 MB ConvLSTM kernel are 64 MB to read, and a thread that waited for every load
// before issuing the next would be latency-bound.
typedef float fold_f4 __attribute__((ext_vector_type(4)));
template <bool VEC>
__global__ __launch_bounds__(256) void wgrad_fold_kernel(float* __restrict__ out, const float* __restrict__ part, int nsplit, long long n, long long slice) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if constexpr (VEC) {
        fold_f4 s = reinterpret_cast<const fold_f4*>(out)[i];
        const fold_f4* __restrict__ q = reinterpret_cast<const fold_f4*>(part) + i;
        const long long st4 = slice >> 2;
        int k = 0;
        for (; k + 8 <= nsplit; k += 8) {
            fold_f4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(q + (long long)(k + j) * st4);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];               // split order: ((s + v0) + v1) + ...
        }
        for (; k < nsplit; ++k) s += __builtin_nontemporal_load(q + (long long)k * st4);
        reinterpret_cast<fold_f4*>(out)[i] = s;
    } else {
        float s = out[i];
        for (int k = 0; k < nsplit; ++k) s += part[(long long)k * slice + i];
        out[i] = s;
    }
}

void wgrad_fold(float* out, const float* part, int nsplit, long long n, hipStream_t st, long long slice) {
    if (nsplit < 1 || n <= 0) return;
    if (slice <= 0) slice = n;
    const bool vec = (n % 4 == 0) && (slice % 4 == 0) && ((((uintptr_t)out) & 15) == 0) && ((((uintptr_t)part) & 15) == 0);
    const long long work = vec ? n / 4 : n;
    const unsigned blocks = (unsigned)((work + 255) / 256);
    if (vec) hipLaunchKernelGGL(wgrad_fold_kernel<true>, dim3(blocks), dim3(256), 0, st, out, part, nsplit, work, slice);
    else hipLaunchKernelGGL(wgrad_fold_kernel<false>, dim3(blocks), dim3(256), 0, st, out, part, nsplit, work, slice);
}
