// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds u16 value = element index; lane l reads at byte address
// addr[l] (host supplied); prints the 4 u16 each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4_t;
__global__ void probe(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    auto p = (__attribute__((address_space(3))) bf16x4_t*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]);
    bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
    unsigned short r[4];
    __builtin_memcpy(r, &v, 8);
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}
int main() {
    int h_addr[64]; unsigned short h_out[256];
    int *d_addr; unsigned short* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int mode = 0; mode < 3; ++mode) {
        for (int l = 0; l < 64; ++l) {
            if (mode == 0) h_addr[l] = l * 8;                                   // lane-linear 8 B
            else if (mode == 1) h_addr[l] = ((l & 15) >> 2) * 200 + (l & 3) * 8 + (l >> 4) * 1000;   // 4 rows (stride 200 B) x 4 quads
            else h_addr[l] = (l & 3) * 200 + ((l & 15) >> 2) * 8 + (l >> 4) * 1000;                  // rows by l&3
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 32; ++l) printf("lane %2d addr %4d(elem %4d): %4d %4d %4d %4d\n", l, h_addr[l], h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
    }
    return 0;
}
