// Micro-benchmark (developer tool): how long does a wave take to ISSUE K LDS-DMA instructions (global_load_lds_dwordx4, 1 KB each), and how long
// until they have landed -- with M0 saved / written / restored around every instruction (the ring kernel's helper), with M0 written once per
// instruction, with M0 written once per FOUR instructions + immediate offsets, and with plain global_load_dwordx4 for reference.
//   hipcc --offload-arch=gfx950 -O3 tests/tools/micro/dma_issue.hip -o /tmp/dma_issue && /tmp/dma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define K 12
__global__ __launch_bounds__(512) void k_dma(const char* src, unsigned long long* out, int variant, int stride_kb) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(uintptr_t)lds) + wave * K * 1024;
    const char* g = src + ((size_t)blockIdx.x * 8 + wave) * K * 1024 * stride_kb + lane * 16;
    float4 sink[K];
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (variant == 0) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g + (size_t)i * 1024 * stride_kb), "s"(lbase + i * 1024) : "memory");
        }
    } else if (variant == 1) {
#pragma unroll
        for (int i = 0; i < K; ++i)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g + (size_t)i * 1024 * stride_kb), "s"(lbase + i * 1024) : "memory");
    } else if (variant == 2) {
#pragma unroll
        for (int i = 0; i < K; i += 4) {
            asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %0, off\n\t"
                         "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %2, off offset:2048\n\t"
                         "global_load_lds_dwordx4 %3, off offset:3072"
                         :: "v"(g + (size_t)i * 1024 * stride_kb), "v"(g + (size_t)(i + 1) * 1024 * stride_kb - 1024), "v"(g + (size_t)(i + 2) * 1024 * stride_kb - 2048),
                            "v"(g + (size_t)(i + 3) * 1024 * stride_kb - 3072), "s"(lbase + i * 1024) : "memory");
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sink[i]) : "v"(g + (size_t)i * 1024 * stride_kb) : "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (variant == 3) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) s += sink[i].x;
        if (s == 123.456f) out[999] = 1;
    }
    if (lane == 0 && blockIdx.x < 4) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}

int main() {
    const size_t bytes = (size_t)256 * 8 * K * 1024 * 4 + 65536;
    char* src; unsigned long long* out;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&out, 8192 * 8); hipMemset(out, 0, 8192 * 8);
    hipFuncSetAttribute((const void*)k_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * K * 1024);
    const char* names[4] = {"M0 save/write/restore per DMA (ring helper)", "M0 written once per DMA", "M0 once per 4 DMAs + immediate offsets", "plain global_load_dwordx4"};
    for (int stride_kb = 1; stride_kb <= 4; stride_kb += 3)
        for (int nblk = 1; nblk <= 256; nblk *= 256)
            for (int v = 0; v < 4; ++v) {
                unsigned long long h[64];
                for (int rep = 0; rep < 3; ++rep) {
                    hipLaunchKernelGGL(k_dma, dim3(nblk), dim3(512), 8 * K * 1024, 0, src, out, v, stride_kb);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
                printf("blocks %3d stride %d KB  %-46s issue %6llu cycles (%5llu per DMA)  landed %6llu   [wave 7: %llu / %llu]\n", nblk, stride_kb, names[v], h[0], h[0] / K, h[1], h[14], h[15]);
            }
    return 0;
}
