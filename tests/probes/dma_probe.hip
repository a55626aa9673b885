// Probe of the gfx950 LDS-DMA path used by conv_ring.hip: global_load_lds_dwordx4 issued from inline asm (M0 = wave-uniform LDS
// byte address, per-lane global source), counted vmcnt, raw s_barrier, LDS destinations above 64 KB.
//   test 0: lane l of wave w copies src[perm(l)] (16 B) to LDS slot base + (w*64 + l)*16  -> checks lane-linear destination,
//           per-lane source, M0 base > 64 KB
//   test 1: three-deep ring with vmcnt(1) / barrier: every wave streams NIT slabs and sums what the OTHER waves wrote
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(256) void probe0(const uint4* src, uint4* out, unsigned lds_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned base = (unsigned)(uintptr_t)smem + lds_off;
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int perm = (lane * 7 + 3) & 63;                        // per-lane source permutation
    dma16(src + wave * 64 + perm, base + wave * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const uint4* l = reinterpret_cast<const uint4*>(smem + lds_off);
    out[threadIdx.x] = l[threadIdx.x];
}

template <int LW>
__global__ __launch_bounds__(256) void probe1(const uint4* src, unsigned long long* out, int nit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned base = (unsigned)(uintptr_t)smem;
    const int lane = threadIdx.x & 63;
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr unsigned SLAB = 4 * LW * 1024;                    // 4 waves x LW instructions x 1 KB
    auto issue = [&](int it) {
        const unsigned buf = (unsigned)(it % 3);
#pragma unroll
        for (int q = 0; q < LW; ++q)
            dma16(src + (size_t)it * (SLAB / 16) + (wave * LW + q) * 64 + lane, base + buf * SLAB + (wave * LW + q) * 1024);
    };
    issue(0);
    if (nit > 1) issue(1);
    unsigned long long acc = 0;
    for (int it = 0; it < nit; ++it) {
        if (it + 1 < nit) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + 2 < nit) issue(it + 2);
        const uint4* l = reinterpret_cast<const uint4*>(smem + (it % 3) * SLAB);
        // read slots written by the other waves
        for (int q = 0; q < LW; ++q) {
            const uint4 v = l[(((wave + 1) & 3) * LW + q) * 64 + lane];
            acc += v.x + v.y + v.z + v.w;
        }
    }
    out[threadIdx.x] = acc;
}

int main() {
    const int N = 1 << 20;
    std::vector<uint4> h(N);
    for (int i = 0; i < N; ++i) h[i] = make_uint4(4 * i, 4 * i + 1, 4 * i + 2, 4 * i + 3);
    uint4 *d_src, *d_out; unsigned long long* d_acc;
    hipMalloc(&d_src, N * sizeof(uint4)); hipMalloc(&d_out, 256 * sizeof(uint4)); hipMalloc(&d_acc, 256 * 8);
    hipMemcpy(d_src, h.data(), N * sizeof(uint4), hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)probe0, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int bad_total = 0;
    for (unsigned off : {0u, 60u * 1024, 100u * 1024, 150u * 1024}) {
        hipMemset(d_out, 0, 256 * sizeof(uint4));
        probe0<<<1, 256, 160 * 1024>>>(d_src, d_out, off);
        uint4 o[256];
        hipError_t e = hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 256; ++t) {
            const int w = t >> 6, l = t & 63, perm = (l * 7 + 3) & 63;
            const unsigned exp = 4u * (w * 64 + perm);
            if (o[t].x != exp || o[t].w != exp + 3) ++bad;
        }
        printf("probe0 lds_off=%6u err=%d bad=%d  (slot0 = %u %u %u %u)\n", off, (int)e, bad, o[0].x, o[0].y, o[0].z, o[0].w);
        bad_total += bad;
    }
    hipFuncSetAttribute((const void*)probe1<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int nit : {1, 2, 3, 7, 50}) {
        probe1<3><<<1, 256, 3 * 12 * 1024>>>(d_src, d_acc, nit);
        unsigned long long a[256];
        hipMemcpy(a, d_acc, sizeof(a), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 256; ++t) {
            const int w = t >> 6, l = t & 63;
            unsigned long long exp = 0;
            for (int it = 0; it < nit; ++it)
                for (int q = 0; q < 3; ++q) {
                    const unsigned long long i = (unsigned long long)it * (12 * 64) + (((w + 1) & 3) * 3 + q) * 64 + l;
                    exp += (unsigned long long)(unsigned)(4 * i) + (unsigned)(4 * i + 1) + (unsigned)(4 * i + 2) + (unsigned)(4 * i + 3);
                }
            if (a[t] != exp) ++bad;
        }
        printf("probe1 nit=%2d bad=%d\n", nit, bad);
        bad_total += bad;
    }
    printf(bad_total ? "DMA PROBE FAILED\n" : "DMA PROBE OK\n");
    return bad_total != 0;
}
