// zero_fill.h -- clearing device memory from inside a launch sequence.
//
// Every launch sequence of this library may be captured into a hipGraph (the train step and the generator unroll are replayed that way,
// models/savp_model.py), and hipMemsetAsync must not be part of one: on this ROCm build (7.2, MI355X) a memset NODE of a replayed graph
// is not ordered with the kernel nodes around it.  tests/tools/ab_calls/graph_memset_probe.py -- a captured chain of
// { hipMemsetAsync(y, 0) ; y += x ; z += y } -- leaves z wrong in 29 of 30 replays (a quarter to three quarters of the elements, for
// 16 KB and for 4 MB buffers alike), while the same chain with a fill KERNEL in place of the memset, and a chain with hipMemcpyAsync
// device-to-device nodes, are exact in 30 of 30.  In the engine this showed as NaNs in the variables after ~8 replayed steps of a small
// fp32 model (split-K outputs and statistics workspaces cleared late: a negative variance is enough), gone with AMD_SERIALIZE_KERNEL=3.
// So: a kernel.  (Eager hipMemsetAsync is a blit kernel on the stream as well; nothing is lost.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__global__ __launch_bounds__(256) void savp_zero_kernel(float* __restrict__ p, unsigned long long n, int vec) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (vec) {                                               // p is 16-byte aligned
        const unsigned long long nv = n >> 2;
        float4* q = reinterpret_cast<float4*>(p);
        for (unsigned long long j = i; j < nv; j += stride) q[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (unsigned long long j = (nv << 2) + i; j < n; j += stride) p[j] = 0.f;
    } else {
        for (unsigned long long j = i; j < n; j += stride) p[j] = 0.f;
    }
}

// bytes: a multiple of 4 (every caller clears float / int32 arrays); p: 4-byte aligned
inline void savp_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (!p || bytes < 4) return;
    const unsigned long long n = bytes >> 2;
    const int vec = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    const unsigned long long work = vec ? (n + 3) / 4 : n;
    unsigned long long blocks = (work + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(savp_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<float*>(p), n, vec);
}

}  // namespace
