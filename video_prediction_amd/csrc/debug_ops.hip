// debug_ops.hip -- developer / soak-test aids: poison what a correct launch sequence must never read.
//
// The parity suite once failed on ONE box with the same binary that was green on four others (round 5).  Two things differ between boxes that a
// kernel could wrongly depend on: what a never-written byte of VRAM or LDS holds, and timing.  These entries make the first one adversarial:
//   savp_debug_poison_lds   every CU's LDS is filled with a pattern (0xFFFFFFFF = NaN as fp32, as two bf16 and as half of an fp64): a kernel that
//                           reads an LDS byte it has not written (a zero slot it assumed, a halo it did not stage) now reads NaN
//   savp_debug_fill_u32     the same for device memory (caller-owned scratch, the allocator's free blocks)
// No product path calls them; tests/test_gpu_soak.py and the SAVP_POISON mode of tests/conftest.py do (video_prediction_amd/debug.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "savp_hip.h"

__global__ __launch_bounds__(256) void debug_lds_poison_kernel(unsigned pattern, int words, unsigned* __restrict__ sink, int spin) {
    extern __shared__ unsigned lds_words[];
    for (int i = threadIdx.x; i < words; i += 256) lds_words[i] = pattern;
    __syncthreads();
    // keep the workgroup (and with it the CU's whole LDS allocation) resident for a moment, so that the `grid` workgroups spread over every CU
    // instead of one CU retiring them back to back
    unsigned long long t0 = __builtin_readcyclecounter();
    while ((long long)(__builtin_readcyclecounter() - t0) < spin) __builtin_amdgcn_s_sleep(8);
    // a read the compiler cannot drop keeps the stores alive
    if (sink && lds_words[(threadIdx.x * 97u + blockIdx.x) % (unsigned)words] != pattern) atomicAdd(sink, 1u);
}

extern "C" int savp_debug_poison_lds(void* stream, uint32_t pattern, void* sink) {
    constexpr int LDS_BYTES = 160 * 1024;                        // all of a CU's LDS: one workgroup per CU at a time
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)debug_lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return SAVP_ELAUNCH;
        attr = true;
    }
    // 4 x 256 workgroups, each holding its CU for ~20 us: whatever the dispatcher's placement, every CU is visited
    hipLaunchKernelGGL(debug_lds_poison_kernel, dim3(1024), dim3(256), LDS_BYTES, (hipStream_t)stream, pattern, LDS_BYTES / 4, (unsigned*)sink, 40000);
    return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
}

__global__ __launch_bounds__(256) void debug_fill_u32_kernel(unsigned* __restrict__ p, unsigned long long n, unsigned pattern) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = pattern;
}

extern "C" int savp_debug_fill_u32(void* stream, void* p, int64_t words, uint32_t pattern) {
    if (!p || words < 0 || (((uintptr_t)p) & 3)) return SAVP_EINVAL;
    if (words == 0) return SAVP_OK;
    long long blocks = (words + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(debug_fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned*)p, (unsigned long long)words, pattern);
    return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
}

// Reads LDS WITHOUT writing it and counts the words that hold `pattern`: the check that savp_debug_poison_lds reaches what the next kernel
// finds (a CU's LDS is not cleared between workgroups).  out[0] += words equal to the pattern, out[1] += words read.
__global__ __launch_bounds__(256) void debug_lds_probe_kernel(unsigned pattern, int words, unsigned long long* __restrict__ out) {
    extern __shared__ unsigned lds_words[];
    unsigned hit = 0, seen = 0;
    for (int i = threadIdx.x; i < words; i += 256) { hit += (lds_words[i] == pattern) ? 1u : 0u; ++seen; }
    atomicAdd(out, (unsigned long long)hit);
    atomicAdd(out + 1, (unsigned long long)seen);
}

extern "C" int savp_debug_probe_lds(void* stream, uint32_t pattern, void* out2) {
    constexpr int LDS_BYTES = 64 * 1024;
    if (!out2) return SAVP_EINVAL;
    hipLaunchKernelGGL(debug_lds_probe_kernel, dim3(512), dim3(256), LDS_BYTES, (hipStream_t)stream, pattern, LDS_BYTES / 4, (unsigned long long*)out2);
    return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
}
