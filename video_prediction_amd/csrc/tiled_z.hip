// tiled_z.hip -- gradient of the tiled latent of a ConvLSTM gate convolution's input WITHOUT its channels in the data gradient.
//
// The cell input is [x | tile(z) | h] (rnn_ops.py:144-146 via tile_concat, savp_model.py:436-444): the nz latent channels hold the
// same value at every pixel of a sample, so their gradient is the sum over pixels of the convolution's data gradient -- 8 output
// columns of every per-timestep DGRAD that push its column count over a tile boundary (72 / 136 / 264 -> 128 / 192 / 384 computed
// columns).  With SAME zero padding a tap (u, v) sees the tiled z only where the shifted pixel lies inside the image, i.e. on a
// sub-rectangle of the gate gradient; all those rectangles are unions of 25 fixed REGIONS (row class x column class, classes
// {0, 1, middle, n-2, n-1}), hence
//
//     dz[n, c] = sum_{regions r} sum_k R[n, r, k] * Weff[r, k, c],      R = per-region sums of the gate gradient dy[n, :, :, k],
//     Weff[r, k, c] = sum of W[u, v, z0 + c, k] over the taps (u, v) whose valid rectangle contains region r
//
// (identity pinned on the CPU against autograd: tests/test_tiled_z_gradient_algebra.py).  The gate gradients of ALL timesteps are
// resident ([T-1, N, H, W, 4F], the batched weight gradient reads them anyway), and dz is only needed after BPTT (z is an input of
// the unroll, not a recurrent state), so ONE launch per layer walks the whole history: a workgroup owns an image, streams its
// plane once (full 16-byte pieces on consecutive lanes), keeps 5 row-class sums per lane, folds them into the 25 x 64 region sums
// of its 64-channel chunk in LDS and multiplies those with Weff on the spot (one workgroup per image and chunk; a second, tiny launch
// adds the chunks' partial dz in chunk order).  Every reduction is a fixed-order tree: deterministic.
// The per-timestep DGRAD then leaves the z channels out (SavpConvArgs.dst_gap).  HBM-bound: reads H*W*4F*2 B per image once.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "savp_hip.h"

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH)
#define TZ_NZ 8            // padded row of Weff / of a partial dz vector for nz <= 8 (BAIR: 8) ...
#define TZ_NZW 32          // ... and for 8 < nz <= 32 (KTH: 32)
static inline int tz_pad(int nz) { return nz <= TZ_NZ ? TZ_NZ : TZ_NZW; }

// class of coordinate y in [0, n): 0, 1, 2 = middle, 3 = n-2, 4 = n-1   (n >= 4: the classes are disjoint)
__device__ __host__ __forceinline__ int tz_class(int y, int n) { return y < 2 ? y : (y >= n - 2 ? y - n + 5 : 2); }

// is tap offset `off` (= tap index - pad) inside the image for every coordinate of class `cls`?  (|off| <= 2)
__device__ __forceinline__ bool tz_tap_valid(int cls, int off) {
    // representative coordinates in an image of extent 8: class 0 -> 0, 1 -> 1, 2 -> 3, 3 -> 6, 4 -> 7
    const int y = cls == 0 ? 0 : cls == 1 ? 1 : cls == 2 ? 3 : cls == 3 ? 6 : 7;
    return y + off >= 0 && y + off < 8;
}

// Weff[r = ry*5 + rx][k][c] (c padded to TZ_NZ) from the HWIO kernel W[kh][kw][Cin][Cout], z channels z0 .. z0+nz-1.
// One thread per (r, c, k): up to 25 independent loads (invalid taps read a valid address and are multiplied by zero, so that all of
// them are in flight together), coalesced over k.
__global__ void tiled_z_weff_kernel(const float* __restrict__ w, int kh, int kw, int ph, int pw, int Cin, int Cout, int z0, int nz,
                                    float* __restrict__ weff, int nzp) {
    const int r = blockIdx.y, ry = r / 5, rx = r - ry * 5;
    const int c = blockIdx.z;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Cout) return;
    float s = 0.f;
    if (c < nz) {
        const float* base = w + (long long)(z0 + c) * Cout + k;
        for (int u = 0; u < kh; ++u) {
            const float mu = tz_tap_valid(ry, u - ph) ? 1.f : 0.f;
#pragma unroll 5
            for (int v = 0; v < kw; ++v) {
                const float m = tz_tap_valid(rx, v - pw) ? mu : 0.f;
                s = fmaf(base[(long long)(u * kw + v) * Cin * Cout], m, s);
            }
        }
    }
    weff[((long long)r * Cout + k) * nzp + c] = s;
}

extern "C" int savp_tiled_z_weff(void* stream, const float* w, int32_t kh, int32_t kw, int32_t ph, int32_t pw, int32_t Cin,
                                 int32_t Cout, int32_t z0, int32_t nz, float* weff) {
    if (!w || !weff || nz < 1 || nz > TZ_NZW || z0 < 0 || z0 + nz > Cin || Cout < 1) return SAVP_EINVAL;
    // the class construction needs every tap offset within +-2 of the pixel
    if (kh < 1 || kw < 1 || ph < 0 || pw < 0 || ph > 2 || pw > 2 || kh - 1 - ph > 2 || kw - 1 - pw > 2) return SAVP_EINVAL;
    dim3 grid((unsigned)((Cout + 63) / 64), 25, (unsigned)tz_pad(nz));
    hipLaunchKernelGGL(tiled_z_weff_kernel, grid, dim3(64), 0, (hipStream_t)stream, w, kh, kw, ph, pw, Cin, Cout, z0, nz, weff, tz_pad(nz));
    return LAUNCH_OK();
}

// One workgroup (256 threads = 32 pixel slots x 8 lanes of 8 channels) per image.  W divides 32, so a thread's column (and its
// column class) is fixed: it keeps one sum per ROW class (5 x 8 channels).
// (grid.y = the 64-channel chunk: C / 64 times the workgroups of an image-only grid, every one with a quarter to an eighth of the
// serial walk -- 113 -> measured below; the chunk partials are folded in chunk order by tiled_z_reduce_kernel: deterministic)
template <bool BF16, int NZP>
__global__ __launch_bounds__(256) void tiled_z_grad_kernel(const void* __restrict__ dy_, int H, int W, int C, const float* __restrict__ weff,
                                                           int nz, float* __restrict__ part_out) {
    __shared__ float part[32][5][64];          // per pixel slot: row-class sums of the chunk's 64 channels
    __shared__ float red[4][NZP];
    const long long img = blockIdx.x;
    const int tid = threadIdx.x, lane8 = tid & 7, slot = tid >> 3;
    const int HW = H * W;
    const int wsh = 31 - __builtin_clz((unsigned)W);     // W is a power of two
    const int xcls = tz_class(slot & (W - 1), W);
    float dzp[NZP];
#pragma unroll
    for (int c = 0; c < NZP; ++c) dzp[c] = 0.f;
    {
        const int c0 = blockIdx.y * 64;
        float acc[5][8];
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[a][j] = 0.f;
        const int ch = c0 + lane8 * 8;
        // U pixels per trip, all U loads issued before the first is used (one load per trip would make the walk a chain of HBM round
        // trips: 131 us per launch on MI355X for 243 MB = 1.8 TB/s); trips past the end re-read the last pixel with weight zero
        constexpr int U = 4;
        for (int p0 = slot; p0 < HW; p0 += 32 * U) {
            uint4 q[U];
            float4 a0[U], a1[U];
            float live[U];
            int ycls[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + 32 * u;
                live[u] = p < HW ? 1.f : 0.f;
                const int pc = min(p, HW - 1);
                ycls[u] = tz_class(pc >> wsh, H);
                const long long off = (img * HW + pc) * (long long)C + ch;
                if (BF16) q[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(dy_) + off);
                else {
                    a0[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy_) + off);
                    a1[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy_) + off + 4);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float v[8];
                if (BF16) {
                    const unsigned w4[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w4[j] << 16); v[2 * j + 1] = __uint_as_float(w4[j] & 0xffff0000u); }
                } else {
                    v[0] = a0[u].x; v[1] = a0[u].y; v[2] = a0[u].z; v[3] = a0[u].w; v[4] = a1[u].x; v[5] = a1[u].y; v[6] = a1[u].z; v[7] = a1[u].w;
                }
#pragma unroll
                for (int a = 0; a < 5; ++a) {
                    const float m = (a == ycls[u]) ? live[u] : 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[a][j] = fmaf(v[j], m, acc[a][j]);
                }
            }
        }
        __syncthreads();                                 // previous chunk's readers are done
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int j = 0; j < 8; ++j) part[slot][a][lane8 * 8 + j] = acc[a][j];
        __syncthreads();
        // 25 x 64 region sums of the chunk, each folded over the slots of its column class in slot order, times Weff
        for (int o = tid; o < 25 * 64; o += 256) {
            const int r = o >> 6, k = o & 63, ry = r / 5, rx = r - ry * 5;
            float R = 0.f;
            for (int s = 0; s < 32; ++s)
                if (tz_class(s & (W - 1), W) == rx && s < HW) R += part[s][ry][k];
            const float* wr = weff + ((long long)r * C + c0 + k) * NZP;
#pragma unroll
            for (int q4 = 0; q4 < NZP / 4; ++q4) {
                const float4 wv = *reinterpret_cast<const float4*>(wr + 4 * q4);
                dzp[4 * q4] = fmaf(R, wv.x, dzp[4 * q4]); dzp[4 * q4 + 1] = fmaf(R, wv.y, dzp[4 * q4 + 1]);
                dzp[4 * q4 + 2] = fmaf(R, wv.z, dzp[4 * q4 + 2]); dzp[4 * q4 + 3] = fmaf(R, wv.w, dzp[4 * q4 + 3]);
            }
        }
    }
    (void)xcls;
    // fixed-order reduction of the 256 partial dz vectors: butterfly inside the wave, then the four waves in order
#pragma unroll
    for (int c = 0; c < NZP; ++c) {
        float v = dzp[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63) == 0) red[tid >> 6][c] = v;
    }
    __syncthreads();
    if (tid < NZP) part_out[(img * gridDim.y + blockIdx.y) * NZP + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

__global__ void tiled_z_reduce_kernel(const float* __restrict__ part, long long nimg, int nchunk, int nz, float* __restrict__ dz, int beta, int nzp) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= nimg * nz) return;
    const long long img = i / nz;
    const int c = (int)(i - img * nz);
    float s = 0.f;
    for (int k = 0; k < nchunk; ++k) s += part[(img * nchunk + k) * nzp + c];
    dz[i] = beta ? dz[i] + s : s;
}

extern "C" int64_t savp_tiled_z_workspace_bytes(int64_t nimg, int32_t C) { return nimg * (int64_t)((C + 63) / 64) * TZ_NZW * 4; }      // (the wider of the two row paddings)

extern "C" int savp_tiled_z_grad(void* stream, const void* dy, int32_t dy_bf16, int64_t nimg, int32_t H, int32_t W, int32_t C,
                                 const float* weff, int32_t nz, float* dz, int32_t beta, void* ws, int64_t ws_bytes) {
    if (!dy || !weff || !dz || nimg < 0 || nz < 1 || nz > TZ_NZW) return SAVP_EINVAL;
    if (!ws || ws_bytes < savp_tiled_z_workspace_bytes(nimg, C)) return SAVP_EINVAL;
    // W a power of two dividing 32 (a thread's column is fixed), >= 4 rows and columns (disjoint classes), whole 64-channel chunks
    if (H < 4 || W < 4 || W > 32 || (W & (W - 1)) || (C % 64) || nimg >= (1ll << 31)) return SAVP_EINVAL;
    if ((((uintptr_t)dy) & 15) || (((uintptr_t)weff) & 15)) return SAVP_EINVAL;
    if (nimg == 0) return SAVP_OK;
    const int nchunk = C / 64;
    if (nchunk > 65535) return SAVP_EINVAL;
    const dim3 grid((unsigned)nimg, (unsigned)nchunk);
    const int nzp = tz_pad(nz);
    hipStream_t st = (hipStream_t)stream;
    if (nzp == TZ_NZ) {
        if (dy_bf16) hipLaunchKernelGGL((tiled_z_grad_kernel<true, TZ_NZ>), grid, dim3(256), 0, st, dy, H, W, C, weff, nz, (float*)ws);
        else hipLaunchKernelGGL((tiled_z_grad_kernel<false, TZ_NZ>), grid, dim3(256), 0, st, dy, H, W, C, weff, nz, (float*)ws);
    } else {
        if (dy_bf16) hipLaunchKernelGGL((tiled_z_grad_kernel<true, TZ_NZW>), grid, dim3(256), 0, st, dy, H, W, C, weff, nz, (float*)ws);
        else hipLaunchKernelGGL((tiled_z_grad_kernel<false, TZ_NZW>), grid, dim3(256), 0, st, dy, H, W, C, weff, nz, (float*)ws);
    }
    hipLaunchKernelGGL(tiled_z_reduce_kernel, dim3((unsigned)((nimg * nz + 255) / 256)), dim3(256), 0, st, (const float*)ws, (long long)nimg,
                       nchunk, nz, dz, beta, nzp);
    return LAUNCH_OK();
}
