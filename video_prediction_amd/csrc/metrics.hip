// metrics.hip -- evaluation metrics and the best-of-N sampling fold of the reference:
//   frame mse / psnr   : video_prediction/metrics.py:5-10 (tf.image.psnr with max_val 1)
//   frame ssim         : metrics.py:13-14 (tf.image.ssim: 11x11 Gaussian window sigma 1.5, k1 0.01, k2 0.03, 'VALID')
//   eval_accumulate    : base_model.py:176-190 -- running min / sum / max of a [T, B] metric, chosen per batch element by the
//                        mean over time (sort_criterion :173-174)
//   select_batch       : base_model.py:170-171,191-196 -- where_axis1 on time-major tensors / running sum of the samples
// All tensors are time-major [T, B, ...] with explicit (time, batch) strides so that one batch half of the generator's
// [T, 2B, ...] buffer can be used in place.  HBM-bound, tiny next to the generator unroll; one launch each.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "savp_hip.h"
#include "zero_fill.h"

#define NT 256
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH)

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float block_sum1(float v, float* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    float s = wsum(v);
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    return t;
}

// one workgroup per frame (t, b): mse = mean((a-b)^2) over the frame; psnr = -10 log10(mse)
__global__ __launch_bounds__(NT) void frame_mse_kernel(const float* a, long long a_st, long long a_sb, const float* b, long long b_st,
                                                       long long b_sb, int B, int inner, float* mse, float* psnr) {
    __shared__ float sh[4];
    const int t = blockIdx.x / B, bb = blockIdx.x % B;
    const float* pa = a + t * a_st + bb * a_sb;
    const float* pb = b + t * b_st + bb * b_sb;
    float acc = 0.f;
    for (int i = threadIdx.x; i < inner; i += NT) { const float d = pa[i] - pb[i]; acc += d * d; }
    const float s = block_sum1(acc, sh);
    if (threadIdx.x == 0) {
        const float m = s / (float)inner;
        if (mse) mse[blockIdx.x] = m;
        if (psnr) psnr[blockIdx.x] = -10.f * log10f(m);
    }
}

// one workgroup per (frame, channel): both planes in LDS, every thread evaluates a strip of window positions
#define SSIM_K 11
__global__ __launch_bounds__(NT) void frame_ssim_kernel(const float* a, long long a_st, long long a_sb, const float* b, long long b_st,
                                                        long long b_sb, int B, int H, int W, int C, float* out) {
    extern __shared__ float plane[];                   // [2][H*W]
    __shared__ float sh[4];
    __shared__ float g[SSIM_K];
    const int f = blockIdx.x / C, c = blockIdx.x % C;
    const int t = f / B, bb = f % B;
    const float* pa = a + t * a_st + bb * a_sb + c;
    const float* pb = b + t * b_st + bb * b_sb + c;
    float* xa = plane;
    float* xb = plane + H * W;
    for (int i = threadIdx.x; i < H * W; i += NT) { xa[i] = pa[(long long)i * C]; xb[i] = pb[(long long)i * C]; }
    if (threadIdx.x == 0) {                            // _fspecial_gauss: the 2-D softmax factorises into normalised 1-D windows
        float s = 0.f, w[SSIM_K];
        for (int i = 0; i < SSIM_K; ++i) { const float d = (float)i - 0.5f * (SSIM_K - 1); w[i] = expf(-d * d / (2.f * 1.5f * 1.5f)); s += w[i]; }
        for (int i = 0; i < SSIM_K; ++i) g[i] = w[i] / s;
    }
    __syncthreads();
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    const int Ho = H - SSIM_K + 1, Wo = W - SSIM_K + 1;
    float acc = 0.f;
    for (int o = threadIdx.x; o < Ho * Wo; o += NT) {
        const int oy = o / Wo, ox = o % Wo;
        float m0 = 0.f, m1 = 0.f, sxy = 0.f, sqq = 0.f;
        for (int u = 0; u < SSIM_K; ++u) {
            float r0 = 0.f, r1 = 0.f, rxy = 0.f, rqq = 0.f;
            const float* ra = xa + (oy + u) * W + ox;
            const float* rb = xb + (oy + u) * W + ox;
#pragma unroll
            for (int v = 0; v < SSIM_K; ++v) {
                const float x = ra[v], y = rb[v], w = g[v];
                r0 += w * x; r1 += w * y; rxy += w * x * y; rqq += w * (x * x + y * y);
            }
            m0 += g[u] * r0; m1 += g[u] * r1; sxy += g[u] * rxy; sqq += g[u] * rqq;
        }
        const float num0 = 2.f * m0 * m1, den0 = m0 * m0 + m1 * m1;
        const float lum = (num0 + c1) / (den0 + c1);
        const float cs = (2.f * sxy - num0 + c2) / (sqq - den0 + c2);
        acc += lum * cs;
    }
    const float s = block_sum1(acc, sh);
    if (threadIdx.x == 0) unsafeAtomicAdd(out + f, s / ((float)(Ho * Wo) * (float)C));
}

// single workgroup: per batch element b compare mean_t metric with mean_t vmin / vmax, update min / sum / max
__global__ __launch_bounds__(NT) void eval_accumulate_kernel(const float* metric, float* vmin, float* vsum, float* vmax, int* cmin, int* cmax,
                                                            int T, int B) {
    for (int b = threadIdx.x; b < B; b += NT) {
        float sm = 0.f, smin = 0.f, smax = 0.f;
        for (int t = 0; t < T; ++t) { sm += metric[t * B + b]; smin += vmin[t * B + b]; smax += vmax[t * B + b]; }
        const bool lo = sm / (float)T < smin / (float)T, hi = sm / (float)T > smax / (float)T;
        for (int t = 0; t < T; ++t) {
            const float m = metric[t * B + b];
            if (lo) vmin[t * B + b] = m;
            if (hi) vmax[t * B + b] = m;
            vsum[t * B + b] += m;
        }
        cmin[b] = lo ? 1 : 0; cmax[b] = hi ? 1 : 0;
    }
}

// out[t, b, :] = cond[b] ? x[t, b, :] : out[t, b, :]   (mode 0)   |   out[t, b, :] += x[t, b, :]   (mode 1)
__global__ __launch_bounds__(NT) void select_batch_kernel(const int* cond, const float* x, long long x_st, long long x_sb, float* out,
                                                         long long o_st, long long o_sb, int B, int inner, int mode) {
    const int t = blockIdx.y, bb = blockIdx.z;
    if (mode == 0 && !cond[bb]) return;
    const float* px = x + t * x_st + bb * x_sb;
    float* po = out + t * o_st + bb * o_sb;
    for (int i = blockIdx.x * NT + threadIdx.x; i < inner; i += gridDim.x * NT) po[i] = mode ? po[i] + px[i] : px[i];
}

extern "C" int savp_frame_mse_psnr(void* stream, const float* a, int64_t a_st, int64_t a_sb, const float* b, int64_t b_st, int64_t b_sb,
                                   int32_t T, int32_t B, int32_t inner, float* mse, float* psnr) {
    if (!a || !b || T < 1 || B < 1 || inner < 1 || (!mse && !psnr)) return SAVP_EINVAL;
    hipLaunchKernelGGL(frame_mse_kernel, dim3((unsigned)(T * B)), dim3(NT), 0, (hipStream_t)stream, a, (long long)a_st, (long long)a_sb, b,
                       (long long)b_st, (long long)b_sb, B, inner, mse, psnr);
    return LAUNCH_OK();
}

extern "C" int savp_frame_ssim(void* stream, const float* a, int64_t a_st, int64_t a_sb, const float* b, int64_t b_st, int64_t b_sb,
                               int32_t T, int32_t B, int32_t H, int32_t W, int32_t C, float* out) {
    if (!a || !b || !out || T < 1 || B < 1 || H < SSIM_K || W < SSIM_K || C < 1) return SAVP_EINVAL;
    const size_t lds = (size_t)2 * H * W * sizeof(float);
    if (lds > 64 * 1024) return SAVP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    savp_zero_async(out, (size_t)T * B * sizeof(float), st);
    hipLaunchKernelGGL(frame_ssim_kernel, dim3((unsigned)(T * B * C)), dim3(NT), lds, st, a, (long long)a_st, (long long)a_sb, b,
                       (long long)b_st, (long long)b_sb, B, H, W, C, out);
    return LAUNCH_OK();
}

extern "C" int savp_eval_accumulate(void* stream, const float* metric, float* vmin, float* vsum, float* vmax, int32_t* cond_min,
                                    int32_t* cond_max, int32_t T, int32_t B) {
    if (!metric || !vmin || !vsum || !vmax || !cond_min || !cond_max || T < 1 || B < 1) return SAVP_EINVAL;
    hipLaunchKernelGGL(eval_accumulate_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, metric, vmin, vsum, vmax, cond_min, cond_max, T, B);
    return LAUNCH_OK();
}

extern "C" int savp_select_batch(void* stream, const int32_t* cond, const float* x, int64_t x_st, int64_t x_sb, float* out, int64_t o_st,
                                 int64_t o_sb, int32_t T, int32_t B, int32_t inner, int32_t mode) {
    if (!x || !out || T < 1 || B < 1 || inner < 1 || (mode == 0 && !cond)) return SAVP_EINVAL;
    unsigned gx = (unsigned)((inner + NT * 4 - 1) / (NT * 4));
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(select_batch_kernel, dim3(gx, (unsigned)T, (unsigned)B), dim3(NT), 0, (hipStream_t)stream, cond, x, (long long)x_st,
                       (long long)x_sb, out, (long long)o_st, (long long)o_sb, B, inner, mode);
    return LAUNCH_OK();
}
