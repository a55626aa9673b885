// cdna_composite.hip -- the CDNA transformation head and the softmax-mask compositing of SAVPCell.call.
//
//   cdna_kernels_fwd/bwd : `kernels + identity; relu(k - 1e-12) + 1e-12; k /= sum_{5x5} k`  (savp_model.py:551,556-559)
//   cdna_apply_fwd/bwd   : apply_cdna_kernels (savp_model.py:893-923): SYMMETRIC pad, per-sample kernels applied to
//                          every colour channel.  The reference goes through 3 transposes + depthwise_conv2d; here
//                          the mirrored gather is done in registers, one thread per output pixel.
//   composite_fwd/bwd    : masks = softmax(logits); gen = sum_k mask_k * transformed_k  (savp_model.py:634-646)
//                          fused, so neither the masks nor the per-layer products touch HBM unless asked for.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "savp_hip.h"
#include "zero_fill.h"
#include "opts.h"

#define NT 256
#define RELU_SHIFT 1e-12f
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH)

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// identity_kernel (savp_model.py:968-980): odd k -> 1 at the centre; even k -> 1/4 on the central 2x2
__device__ __forceinline__ float ident_at(int u, int v, int kh, int kw) {
    float fu, fv;
    if (kh & 1) fu = (u == kh / 2) ? 1.f : 0.f; else fu = (u == kh / 2 - 1 || u == kh / 2) ? 0.5f : 0.f;
    if (kw & 1) fv = (v == kw / 2) ? 1.f : 0.f; else fv = (v == kw / 2 - 1 || v == kw / 2) ? 0.5f : 0.f;
    return fu * fv;
}

// raw [N, kh*kw*K] (index (u*kw+v)*K + k) -> normalised kern, same layout.  One workgroup per sample, one thread per
// (tap, k) element (kh*kw*K <= 512): every element is loaded once, all loads in flight together; the per-k sums over the taps go
// through LDS.  (The first version walked the taps serially in one thread per (n, k): 75 dependent loads, 11 / 23 us for 3 KB.)
#define CK_NT 512
__device__ __forceinline__ float cdna_tap_sum(float* sh, float v, int e, int taps, int K, bool live) {
    // sum over the taps of v for this thread's k; sh [taps*K] scratch, result broadcast to every thread of the same k
    if (live) sh[e] = v;
    __syncthreads();
    float s = 0.f;
    if (live) {
        const int k = e % K;
        for (int t = 0; t < taps; ++t) s += sh[t * K + k];
    }
    __syncthreads();
    return s;
}

__global__ __launch_bounds__(CK_NT) void cdna_kernels_fwd_kernel(const float* __restrict__ raw, float* __restrict__ kern, int kh, int kw, int K) {
    __shared__ float sh[CK_NT];
    const int n = blockIdx.x, e = threadIdx.x, taps = kh * kw;
    const bool live = e < taps * K;
    const long long base = (long long)n * taps * K;
    float v = 0.f;
    if (live) {
        const int t = e / K;
        v = fmaxf(raw[base + e] + ident_at(t / kw, t % kw, kh, kw) - RELU_SHIFT, 0.f) + RELU_SHIFT;
    }
    const float s = cdna_tap_sum(sh, v, e, taps, K, live);
    if (live) kern[base + e] = v * (1.f / s);
}

// draw = ((dkern - sum(dkern*kern)) / s) * [raw + ident - shift > 0]
__global__ __launch_bounds__(CK_NT) void cdna_kernels_bwd_kernel(const float* __restrict__ raw, const double* __restrict__ dkern,
                                                                 float* __restrict__ draw, int kh, int kw, int K) {
    __shared__ float sh[CK_NT];
    const int n = blockIdx.x, e = threadIdx.x, taps = kh * kw;
    const bool live = e < taps * K;
    const long long base = (long long)n * taps * K;
    float pre = 0.f, v = 0.f, d = 0.f;
    if (live) {
        const int t = e / K;
        pre = raw[base + e] + ident_at(t / kw, t % kw, kh, kw) - RELU_SHIFT;
        d = (float)dkern[base + e];
        v = fmaxf(pre, 0.f) + RELU_SHIFT;
    }
    const float s = cdna_tap_sum(sh, v, e, taps, K, live);
    const float inv = live ? 1.f / s : 0.f;
    const float dot = cdna_tap_sum(sh, d * v * inv, e, taps, K, live);
    if (live) draw[base + e] = pre > 0.f ? (d - dot) * inv : 0.f;
}

extern "C" int savp_cdna_kernels_fwd(void* stream, const float* raw, float* kern, int32_t N, int32_t kh, int32_t kw, int32_t K) {
    if (!raw || !kern || N < 1 || kh * kw * K > CK_NT) return SAVP_EINVAL;
    hipLaunchKernelGGL(cdna_kernels_fwd_kernel, dim3(N), dim3(CK_NT), 0, (hipStream_t)stream, raw, kern, kh, kw, K);
    return LAUNCH_OK();
}
extern "C" int savp_cdna_kernels_bwd(void* stream, const float* raw, const double* dkern, float* draw, int32_t N, int32_t kh,
                                     int32_t kw, int32_t K) {
    if (!raw || !dkern || !draw || N < 1 || kh * kw * K > CK_NT) return SAVP_EINVAL;
    hipLaunchKernelGGL(cdna_kernels_bwd_kernel, dim3(N), dim3(CK_NT), 0, (hipStream_t)stream, raw, dkern, draw, kh, kw, K);
    return LAUNCH_OK();
}

// tf.pad SYMMETRIC index: padded coordinate q in [-pad, n+pad) -> source index
__device__ __forceinline__ int sym(int q, int n) { return q < 0 ? -q - 1 : (q >= n ? 2 * n - 1 - q : q); }

#define MAXTAPS 49
#define MAXK 8
#define MAXC 4

struct CdnaP {
    int N, H, W, C, K, kh, kw, pt, pl;           // pt/pl = SAME pad before (kh-1)/2
    const float* img; long long i_sn, i_sp;
    const float* kern;                           // [N, kh*kw, K]
    float* out; long long o_sn, o_sp;            // [N,H,W,K*C]  channel index k*C + c
    // bwd
    const float* dout; long long do_sn, do_sp;
    float* dimg; long long di_sn, di_sp; int dimg_beta;
    double* dkern;                               // [N, kh*kw, K] float64 (overwritten): the tiles' partial sums meet here through float64 atomics -- exact, order-independent
};

__global__ __launch_bounds__(NT) void cdna_apply_fwd_kernel(CdnaP p) {
    __shared__ float sk[MAXTAPS * MAXK];
    const int n = blockIdx.y;
    const int taps = p.kh * p.kw;
    for (int i = threadIdx.x; i < taps * p.K; i += NT) sk[i] = p.kern[(long long)n * taps * p.K + i];
    __syncthreads();
    const int px = blockIdx.x * NT + threadIdx.x;
    if (px >= p.H * p.W) return;
    const int y = px / p.W, x = px % p.W;
    float acc[MAXK][MAXC];
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) acc[k][c] = 0.f;
    const float* im = p.img + (long long)n * p.i_sn;
    for (int u = 0; u < p.kh; ++u) {
        const int sy = sym(y + u - p.pt, p.H);
        for (int v = 0; v < p.kw; ++v) {
            const int sx = sym(x + v - p.pl, p.W);
            const float* q = im + (long long)(sy * p.W + sx) * p.i_sp;
            float pix[MAXC];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) pix[c] = c < p.C ? q[c] : 0.f;
            const float* kk = sk + (u * p.kw + v) * p.K;
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if (k < p.K) {
                    const float w = kk[k];
#pragma unroll
                    for (int c = 0; c < MAXC; ++c) acc[k][c] += pix[c] * w;
                }
        }
    }
    float* o = p.out + (long long)n * p.o_sn + (long long)px * p.o_sp;
    for (int k = 0; k < p.K; ++k)
        for (int c = 0; c < p.C; ++c) o[k * p.C + c] = acc[k][c];
}

// d_img in gather form: every source pixel collects from the <=2x2 padded positions that mirror onto it.
__global__ __launch_bounds__(NT) void cdna_apply_bwd_img_kernel(CdnaP p) {
    __shared__ float sk[MAXTAPS * MAXK];
    const int n = blockIdx.y;
    const int taps = p.kh * p.kw;
    for (int i = threadIdx.x; i < taps * p.K; i += NT) sk[i] = p.kern[(long long)n * taps * p.K + i];
    __syncthreads();
    const int px = blockIdx.x * NT + threadIdx.x;
    if (px >= p.H * p.W) return;
    const int sy = px / p.W, sx = px % p.W;
    const int pb = p.kh - 1 - p.pt, pr = p.kw - 1 - p.pl;     // pad after
    // padded rows (in un-padded coordinates q, i.e. padded index - pt) that map onto sy
    int qy[3], nqy = 0, qx[3], nqx = 0;
    qy[nqy++] = sy;
    if (-sy - 1 >= -p.pt) qy[nqy++] = -sy - 1;
    if (2 * p.H - 1 - sy < p.H + pb && 2 * p.H - 1 - sy >= p.H) qy[nqy++] = 2 * p.H - 1 - sy;
    qx[nqx++] = sx;
    if (-sx - 1 >= -p.pl) qx[nqx++] = -sx - 1;
    if (2 * p.W - 1 - sx < p.W + pr && 2 * p.W - 1 - sx >= p.W) qx[nqx++] = 2 * p.W - 1 - sx;
    float acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
    const float* dout = p.dout + (long long)n * p.do_sn;
    for (int a = 0; a < nqy; ++a)
        for (int b = 0; b < nqx; ++b) {
            // padded position (qy[a], qx[b]) is read by output (y, x) with tap (u, v): qy = y + u - pt
            for (int u = 0; u < p.kh; ++u) {
                const int y = qy[a] - u + p.pt;
                if (y < 0 || y >= p.H) continue;
                for (int v = 0; v < p.kw; ++v) {
                    const int x = qx[b] - v + p.pl;
                    if (x < 0 || x >= p.W) continue;
                    const float* d = dout + (long long)(y * p.W + x) * p.do_sp;
                    const float* kk = sk + (u * p.kw + v) * p.K;
                    for (int k = 0; k < p.K; ++k) {
                        const float w = kk[k];
#pragma unroll
                        for (int c = 0; c < MAXC; ++c)
                            if (c < p.C) acc[c] += d[k * p.C + c] * w;
                    }
                }
            }
        }
    float* di = p.dimg + (long long)n * p.di_sn + (long long)px * p.di_sp;
    for (int c = 0; c < p.C; ++c) di[c] = p.dimg_beta ? di[c] + acc[c] : acc[c];
}

// dkern[n,u,v,k] = sum_{y,x,c} img_sym[y+u-pt, x+v-pl, c] * dout[y,x,k*C+c].  grid (K, N); each WG reduces all taps
// for one (n,k) with per-thread tap accumulators.
__global__ __launch_bounds__(NT) void cdna_apply_bwd_kern_kernel(CdnaP p) {
    __shared__ float sh[4 * MAXTAPS];
    const int k = blockIdx.x, n = blockIdx.y;
    const int taps = p.kh * p.kw;
    float acc[MAXTAPS];
#pragma unroll
    for (int t = 0; t < MAXTAPS; ++t) acc[t] = 0.f;
    const float* im = p.img + (long long)n * p.i_sn;
    const float* dout = p.dout + (long long)n * p.do_sn;
    for (int px = threadIdx.x; px < p.H * p.W; px += NT) {
        const int y = px / p.W, x = px % p.W;
        float d[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) d[c] = c < p.C ? dout[(long long)px * p.do_sp + k * p.C + c] : 0.f;
#pragma unroll
        for (int t = 0; t < MAXTAPS; ++t) {
            if (t < taps) {
                const int u = t / p.kw, v = t % p.kw;
                const float* q = im + (long long)(sym(y + u - p.pt, p.H) * p.W + sym(x + v - p.pl, p.W)) * p.i_sp;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < p.C) s += q[c] * d[c];
                acc[t] += s;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < MAXTAPS; ++t) {
        if (t < taps) {
            float s = wsum(acc[t]);
            if (lane == 0) sh[wave * MAXTAPS + t] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < taps) {
        const int t = threadIdx.x;
        p.dkern[((long long)n * taps + t) * p.K + k] = (double)(sh[t] + sh[MAXTAPS + t] + sh[2 * MAXTAPS + t] + sh[3 * MAXTAPS + t]);
    }
}

// ---- specialised (compile-time kh,kw,K,C) backward kernels: fully unrolled, vector loads of the K*C gradient row -----
template <int KH, int KW, int TK, int TC>
__global__ __launch_bounds__(NT) void cdna_bwd_img_fast_kernel(CdnaP p) {
    __shared__ float sk[KH * KW * TK];
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < KH * KW * TK; i += NT) sk[i] = p.kern[(long long)n * KH * KW * TK + i];
    __syncthreads();
    const int px = blockIdx.x * NT + threadIdx.x;
    if (px >= p.H * p.W) return;
    const int sy = px / p.W, sx = px % p.W;
    constexpr int PT = (KH - 1) / 2, PL = (KW - 1) / 2, PB = KH - 1 - PT, PR = KW - 1 - PL;
    int qy[3], nqy = 0, qx[3], nqx = 0;
    qy[nqy++] = sy;
    if (-sy - 1 >= -PT) qy[nqy++] = -sy - 1;
    if (2 * p.H - 1 - sy < p.H + PB && 2 * p.H - 1 - sy >= p.H) qy[nqy++] = 2 * p.H - 1 - sy;
    qx[nqx++] = sx;
    if (-sx - 1 >= -PL) qx[nqx++] = -sx - 1;
    if (2 * p.W - 1 - sx < p.W + PR && 2 * p.W - 1 - sx >= p.W) qx[nqx++] = 2 * p.W - 1 - sx;
    float acc[TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[c] = 0.f;
    const float* dout = p.dout + (long long)n * p.do_sn;
    for (int a = 0; a < nqy; ++a)
        for (int b = 0; b < nqx; ++b) {
#pragma unroll
            for (int u = 0; u < KH; ++u) {
                const int y = qy[a] - u + PT;
                if (y < 0 || y >= p.H) continue;
#pragma unroll
                for (int v = 0; v < KW; ++v) {
                    const int x = qx[b] - v + PL;
                    if (x < 0 || x >= p.W) continue;
                    const float* d = dout + (long long)(y * p.W + x) * p.do_sp;
                    float dv[TK * TC];
                    if ((TK * TC) % 4 == 0) {
#pragma unroll
                        for (int q = 0; q < TK * TC / 4; ++q) {
                            float4 t = *reinterpret_cast<const float4*>(d + 4 * q);
                            dv[4 * q] = t.x; dv[4 * q + 1] = t.y; dv[4 * q + 2] = t.z; dv[4 * q + 3] = t.w;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < TK * TC; ++q) dv[q] = d[q];
                    }
#pragma unroll
                    for (int k = 0; k < TK; ++k) {
                        const float w = sk[(u * KW + v) * TK + k];
#pragma unroll
                        for (int c = 0; c < TC; ++c) acc[c] += dv[k * TC + c] * w;
                    }
                }
            }
        }
    float* di = p.dimg + (long long)n * p.di_sn + (long long)px * p.di_sp;
#pragma unroll
    for (int c = 0; c < TC; ++c) di[c] = p.dimg_beta ? di[c] + acc[c] : acc[c];
}

// grid (pixel chunks, N): every thread accumulates all taps x K for its pixels, block-reduces and atomically adds.
template <int KH, int KW, int TK, int TC>
__global__ __launch_bounds__(NT) void cdna_bwd_kern_fast_kernel(CdnaP p, int chunk) {
    constexpr int NV = KH * KW * TK;
    __shared__ float sh[4 * NV];
    const int n = blockIdx.y;
    constexpr int PT = (KH - 1) / 2, PL = (KW - 1) / 2;
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    const float* im = p.img + (long long)n * p.i_sn;
    const float* dout = p.dout + (long long)n * p.do_sn;
    const int p0 = blockIdx.x * chunk, p1 = min(p.H * p.W, p0 + chunk);
    for (int px = p0 + threadIdx.x; px < p1; px += NT) {
        const int y = px / p.W, x = px % p.W;
        float dv[TK * TC];
        const float* d = dout + (long long)px * p.do_sp;
        if ((TK * TC) % 4 == 0) {
#pragma unroll
            for (int q = 0; q < TK * TC / 4; ++q) {
                float4 t = *reinterpret_cast<const float4*>(d + 4 * q);
                dv[4 * q] = t.x; dv[4 * q + 1] = t.y; dv[4 * q + 2] = t.z; dv[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < TK * TC; ++q) dv[q] = d[q];
        }
#pragma unroll
        for (int u = 0; u < KH; ++u) {
            const int yy = sym(y + u - PT, p.H);
#pragma unroll
            for (int v = 0; v < KW; ++v) {
                const float* q = im + (long long)(yy * p.W + sym(x + v - PL, p.W)) * p.i_sp;
                float pix[TC];
#pragma unroll
                for (int c = 0; c < TC; ++c) pix[c] = q[c];
#pragma unroll
                for (int k = 0; k < TK; ++k) {
                    float s = 0.f;
#pragma unroll
                    for (int c = 0; c < TC; ++c) s += pix[c] * dv[k * TC + c];
                    acc[(u * KW + v) * TK + k] += s;
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = wsum(acc[i]);
        if (lane == 0) sh[wave * NV + i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NV; i += NT)
        unsafeAtomicAdd(p.dkern + (long long)n * NV + i, (double)(sh[i] + sh[NV + i] + sh[2 * NV + i] + sh[3 * NV + i]));
}

// ---- LDS-tiled 5x5 kernels (compile-time K, C): a workgroup owns a 16x16 pixel tile --------------------------------------------
// The one-thread-per-pixel kernels above read every tap straight from global memory: 75 scalar loads per thread at a pixel
// stride of 64 B (forward; the image lives in the 16-channel input buffer of h0) or 75 float4 loads at a stride of 224 B
// (backward; the gradient lives in the 56-channel mask-conv input row) -- a wave instruction touches 32-64 different 128-B
// lines, so the kernels were bound by the texture-address path at ~10 x their HBM time.  Here the (tile + 2-pixel halo) of the
// image (SYMMETRIC padding resolved while staging) or of the gradient is parked in LDS once per workgroup; taps are LDS reads.
#define CT_TS 16
#define CT_HS 20

__device__ __forceinline__ int sym_clamped(int q, int n) { const int s = sym(q, n); return min(max(s, 0), n - 1); }

// image halo tile -> LDS as one float4 per pixel (channels >= TC are 0)
template <int TC>
__device__ __forceinline__ void stage_img_halo(const CdnaP& p, int n, int ty0, int tx0, float* img) {
    const float* im = p.img + (long long)n * p.i_sn;
    for (int i = threadIdx.x; i < CT_HS * CT_HS; i += NT) {
        const int yy = i / CT_HS, xx = i - yy * CT_HS;
        const int sy = sym_clamped(ty0 + yy - 2, p.H), sx = sym_clamped(tx0 + xx - 2, p.W);
        const float* q = im + (long long)(sy * p.W + sx) * p.i_sp;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = q[0];
        if (TC > 1) v.y = q[1];
        if (TC > 2) v.z = q[2];
        if (TC > 3) v.w = q[3];
        *reinterpret_cast<float4*>(img + 4 * i) = v;
    }
}

template <int TK, int TC>
__global__ __launch_bounds__(NT) void cdna_apply_fwd_tiled_kernel(CdnaP p, int tiles_x, int vec) {
    __shared__ __attribute__((aligned(16))) float img[CT_HS * CT_HS * 4];
    __shared__ __attribute__((aligned(16))) float sk[25 * TK];
    const int n = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * CT_TS, tx0 = (blockIdx.x % tiles_x) * CT_TS;
    for (int i = threadIdx.x; i < 25 * TK; i += NT) sk[i] = p.kern[(long long)n * 25 * TK + i];
    stage_img_halo<TC>(p, n, ty0, tx0, img);
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int y = ty0 + ty, x = tx0 + tx;
    if (y >= p.H || x >= p.W) return;
    float acc[TK][TC];
#pragma unroll
    for (int k = 0; k < TK; ++k)
#pragma unroll
        for (int c = 0; c < TC; ++c) acc[k][c] = 0.f;
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int v = 0; v < 5; ++v) {
            const float4 pv = *reinterpret_cast<const float4*>(img + ((ty + u) * CT_HS + tx + v) * 4);
            const float pix[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
            for (int k = 0; k < TK; ++k) {
                const float w = sk[(u * 5 + v) * TK + k];
#pragma unroll
                for (int c = 0; c < TC; ++c) acc[k][c] += pix[c] * w;
            }
        }
    float* o = p.out + (long long)n * p.o_sn + (long long)(y * p.W + x) * p.o_sp;
    if (vec && (TK * TC) % 4 == 0) {
#pragma unroll
        for (int q = 0; q < TK * TC / 4; ++q) {
            float4 t;
            t.x = acc[(4 * q) / TC][(4 * q) % TC]; t.y = acc[(4 * q + 1) / TC][(4 * q + 1) % TC];
            t.z = acc[(4 * q + 2) / TC][(4 * q + 2) % TC]; t.w = acc[(4 * q + 3) / TC][(4 * q + 3) % TC];
            *reinterpret_cast<float4*>(o + 4 * q) = t;
        }
    } else {
#pragma unroll
        for (int k = 0; k < TK; ++k)
#pragma unroll
            for (int c = 0; c < TC; ++c) o[k * TC + c] = acc[k][c];
    }
}

// gradient (K*C floats per pixel) of the pixels (ty0 - 2 .. ty0 + 17) x (tx0 - 2 .. tx0 + 17) -> LDS; outside the image: zeros
template <int TK, int TC>
__device__ __forceinline__ void stage_dout_halo(const CdnaP& p, int n, int ty0, int tx0, int vec, float* dts) {
    constexpr int KC = TK * TC;
    const float* dout = p.dout + (long long)n * p.do_sn;
    for (int i = threadIdx.x; i < CT_HS * CT_HS; i += NT) {
        const int yy = i / CT_HS, xx = i - yy * CT_HS;
        const int y = ty0 + yy - 2, x = tx0 + xx - 2;
        const bool ok = y >= 0 && y < p.H && x >= 0 && x < p.W;
        const float* d = dout + (long long)((ok ? y : 0) * p.W + (ok ? x : 0)) * p.do_sp;
        float* dst = dts + i * KC;
        if (vec && KC % 4 == 0) {
#pragma unroll
            for (int q = 0; q < KC / 4; ++q) {
                float4 t = *reinterpret_cast<const float4*>(d + 4 * q);
                if (!ok) t = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(dst + 4 * q) = t;
            }
        } else {
#pragma unroll
            for (int q = 0; q < KC; ++q) dst[q] = ok ? d[q] : 0.f;
        }
    }
}

template <int TK, int TC>
__global__ __launch_bounds__(NT) void cdna_bwd_img_tiled_kernel(CdnaP p, int tiles_x, int vec) {
    constexpr int KC = TK * TC, PT = 2, PL = 2, PB = 2, PR = 2;
    __shared__ __attribute__((aligned(16))) float dts[CT_HS * CT_HS * KC];
    __shared__ __attribute__((aligned(16))) float sk[25 * TK];
    const int n = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * CT_TS, tx0 = (blockIdx.x % tiles_x) * CT_TS;
    for (int i = threadIdx.x; i < 25 * TK; i += NT) sk[i] = p.kern[(long long)n * 25 * TK + i];
    // vec bit 1: clear this sample's kernel-gradient accumulator for the cdna_bwd_kern launch that follows on the stream (it
    // adds with atomics; saves the launcher a 12 KB memset per timestep)
    if ((vec & 2) && blockIdx.x == 0)
        for (int i = threadIdx.x; i < 25 * TK; i += NT) p.dkern[(long long)n * 25 * TK + i] = 0.0;
    vec &= 1;
    stage_dout_halo<TK, TC>(p, n, ty0, tx0, vec, dts);
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    const int sy = ty0 + ty, sx = tx0 + tx;
    if (sy >= p.H || sx >= p.W) return;
    // padded positions (un-padded coordinates) that mirror onto (sy, sx): itself and, next to a border, its reflections
    int qy[3], nqy = 0, qx[3], nqx = 0;
    qy[nqy++] = sy;
    if (-sy - 1 >= -PT) qy[nqy++] = -sy - 1;
    if (2 * p.H - 1 - sy < p.H + PB && 2 * p.H - 1 - sy >= p.H) qy[nqy++] = 2 * p.H - 1 - sy;
    qx[nqx++] = sx;
    if (-sx - 1 >= -PL) qx[nqx++] = -sx - 1;
    if (2 * p.W - 1 - sx < p.W + PR && 2 * p.W - 1 - sx >= p.W) qx[nqx++] = 2 * p.W - 1 - sx;
    float acc[TC];
#pragma unroll
    for (int c = 0; c < TC; ++c) acc[c] = 0.f;
    for (int a = 0; a < nqy; ++a)
        for (int b = 0; b < nqx; ++b) {
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int y = qy[a] - u + PT;
                if (y < 0 || y >= p.H) continue;
#pragma unroll
                for (int v = 0; v < 5; ++v) {
                    const int x = qx[b] - v + PL;
                    if (x < 0 || x >= p.W) continue;
                    // every (y, x) reached here lies within 2 pixels of (sy, sx): inside the staged halo
                    const float* d = dts + ((y - ty0 + 2) * CT_HS + (x - tx0 + 2)) * KC;
                    float dv[KC];
                    if (KC % 4 == 0) {
#pragma unroll
                        for (int q = 0; q < KC / 4; ++q) {
                            const float4 t = *reinterpret_cast<const float4*>(d + 4 * q);
                            dv[4 * q] = t.x; dv[4 * q + 1] = t.y; dv[4 * q + 2] = t.z; dv[4 * q + 3] = t.w;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < KC; ++q) dv[q] = d[q];
                    }
#pragma unroll
                    for (int k = 0; k < TK; ++k) {
                        const float w = sk[(u * 5 + v) * TK + k];
#pragma unroll
                        for (int c = 0; c < TC; ++c) acc[c] += dv[k * TC + c] * w;
                    }
                }
            }
        }
    float* di = p.dimg + (long long)n * p.di_sn + (long long)(sy * p.W + sx) * p.di_sp;
#pragma unroll
    for (int c = 0; c < TC; ++c) di[c] = p.dimg_beta ? di[c] + acc[c] : acc[c];
}

// dkern[n, tap, k] += the contribution of one 16x16 tile.  Wave w owns the taps t_lo(w) .. (7 + 6 + 6 + 6 = 25), lane l the
// pixels l, l + 64, l + 128, l + 192 of the tile: 28 accumulators per thread instead of 100, so the wave reduction at the end (one
// butterfly per accumulator) stays comparable to the tile work, and every workgroup needs exactly one staging round trip
// (a strip of tiles per workgroup serialised four of them: 23 us).  100 global atomics per workgroup.
template <int TK, int TC>
__global__ __launch_bounds__(NT) void cdna_bwd_kern_tiled_kernel(CdnaP p, int tiles_x, int vec) {
    constexpr int KC = TK * TC;
    __shared__ __attribute__((aligned(16))) float img[CT_HS * CT_HS * 4];
    __shared__ __attribute__((aligned(16))) float dts[NT * KC];
    const int n = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * CT_TS, tx0 = (blockIdx.x % tiles_x) * CT_TS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t_lo = wave == 0 ? 0 : 1 + 6 * wave, nt = wave == 0 ? 7 : 6;
    float acc[7][TK];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int k = 0; k < TK; ++k) acc[t][k] = 0.f;
    const float* dout = p.dout + (long long)n * p.do_sn;
    {
        const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
        const int y = ty0 + ty, x = tx0 + tx;
        const bool ok = y < p.H && x < p.W;
        const float* d = dout + (long long)((ok ? y : 0) * p.W + (ok ? x : 0)) * p.do_sp;
        float* dst = dts + threadIdx.x * KC;
        if (vec && KC % 4 == 0) {
#pragma unroll
            for (int q = 0; q < KC / 4; ++q) {
                float4 t = *reinterpret_cast<const float4*>(d + 4 * q);
                if (!ok) t = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(dst + 4 * q) = t;
            }
        } else {
#pragma unroll
            for (int q = 0; q < KC; ++q) dst[q] = ok ? d[q] : 0.f;
        }
    }
    stage_img_halo<TC>(p, n, ty0, tx0, img);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pxl = lane + 64 * j;
        const int ty = pxl >> 4, tx = pxl & 15;
        float dv[KC];
        if (KC % 4 == 0) {
#pragma unroll
            for (int q = 0; q < KC / 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(dts + pxl * KC + 4 * q);
                dv[4 * q] = t.x; dv[4 * q + 1] = t.y; dv[4 * q + 2] = t.z; dv[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < KC; ++q) dv[q] = dts[pxl * KC + q];
        }
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            if (t < nt) {
                const int tap = t_lo + t;
                const int u = (tap * 13) >> 6, v = tap - 5 * u;          // tap / 5, tap % 5 for tap < 25
                const float4 pv = *reinterpret_cast<const float4*>(img + ((ty + u) * CT_HS + tx + v) * 4);
                const float pix[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
                for (int k = 0; k < TK; ++k) {
                    float s = 0.f;
#pragma unroll
                    for (int c = 0; c < TC; ++c) s += pix[c] * dv[k * TC + c];
                    acc[t][k] += s;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        if (t < nt) {
#pragma unroll
            for (int k = 0; k < TK; ++k) {
                const float s = wsum(acc[t][k]);
                if (lane == 0) unsafeAtomicAdd(p.dkern + ((long long)n * 25 + t_lo + t) * TK + k, (double)s);
            }
        }
    }
}

static int fill_cdna(CdnaP& p, const SavpCdnaArgs* a) {
    if (!a || a->kh * a->kw > MAXTAPS || a->K > MAXK || a->C > MAXC || a->K < 1 || a->C < 1) return SAVP_EINVAL;
    p.N = a->N; p.H = a->H; p.W = a->W; p.C = a->C; p.K = a->K; p.kh = a->kh; p.kw = a->kw;
    p.pt = (a->kh - 1) / 2; p.pl = (a->kw - 1) / 2;
    p.img = (const float*)a->img.p; p.i_sn = a->img.sn; p.i_sp = a->img.sp;
    p.kern = a->kern;
    p.out = (float*)a->out.p; p.o_sn = a->out.sn; p.o_sp = a->out.sp;
    p.dout = (const float*)a->dout.p; p.do_sn = a->dout.sn; p.do_sp = a->dout.sp;
    p.dimg = (float*)a->dimg.p; p.di_sn = a->dimg.sn; p.di_sp = a->dimg.sp; p.dimg_beta = a->dimg_beta;
    p.dkern = (double*)a->dkern;
    if (p.dkern && (((uintptr_t)p.dkern) & 7)) return SAVP_EINVAL;
    return SAVP_OK;
}

// SAVP_CDNA_LEGACY=1: the one-thread-per-pixel global-memory kernels (developer A/B switch)
static bool cdna_legacy() {
    return savp_opt(OPT_CDNA_LEGACY) != 0;
}
// 3 / 1: the tiled 5x5, K = 4 kernels for C = 3 / 1 apply; 0: generic
static int cdna_tiled_kind(const SavpCdnaArgs* a) {
    if (cdna_legacy() || a->kh != 5 || a->kw != 5 || a->K != 4 || a->H < 3 || a->W < 3) return 0;
    return a->C == 3 ? 3 : (a->C == 1 ? 1 : 0);
}

extern "C" int savp_cdna_apply_fwd(void* stream, const SavpCdnaArgs* a) {
    CdnaP p;
    int rc = fill_cdna(p, a);
    if (rc) return rc;
    const int kind = cdna_tiled_kind(a);
    if (kind) {
        const int tiles_x = (a->W + CT_TS - 1) / CT_TS, tiles_y = (a->H + CT_TS - 1) / CT_TS;
        const int vec = ((uintptr_t)a->out.p % 16 == 0) && (a->out.sn % 4 == 0) && (a->out.sp % 4 == 0);
        dim3 grid(tiles_x * tiles_y, a->N);
        if (kind == 3) hipLaunchKernelGGL((cdna_apply_fwd_tiled_kernel<4, 3>), grid, dim3(NT), 0, (hipStream_t)stream, p, tiles_x, vec);
        else hipLaunchKernelGGL((cdna_apply_fwd_tiled_kernel<4, 1>), grid, dim3(NT), 0, (hipStream_t)stream, p, tiles_x, vec);
        return LAUNCH_OK();
    }
    hipLaunchKernelGGL(cdna_apply_fwd_kernel, dim3((a->H * a->W + NT - 1) / NT, a->N), dim3(NT), 0, (hipStream_t)stream, p);
    return LAUNCH_OK();
}

extern "C" int savp_cdna_apply_bwd(void* stream, const SavpCdnaArgs* a) {
    CdnaP p;
    int rc = fill_cdna(p, a);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool al = ((uintptr_t)a->dout.p % 16 == 0) && (a->dout.sn % 4 == 0) && (a->dout.sp % 4 == 0);
    const int kind = cdna_tiled_kind(a);
    if (kind) {
        const int tiles_x = (a->W + CT_TS - 1) / CT_TS, tiles_y = (a->H + CT_TS - 1) / CT_TS;
        const int vec = al ? 1 : 0;
        if (p.dimg) {
            dim3 grid(tiles_x * tiles_y, a->N);
            const int v2 = vec | (p.dkern ? 2 : 0);
            if (kind == 3) hipLaunchKernelGGL((cdna_bwd_img_tiled_kernel<4, 3>), grid, dim3(NT), 0, st, p, tiles_x, v2);
            else hipLaunchKernelGGL((cdna_bwd_img_tiled_kernel<4, 1>), grid, dim3(NT), 0, st, p, tiles_x, v2);
        }
        if (p.dkern) {
            if (!p.dimg) savp_zero_async(p.dkern, (size_t)a->N * 25 * 4 * sizeof(double), st);
            dim3 grid(tiles_x * tiles_y, a->N);
            if (kind == 3) hipLaunchKernelGGL((cdna_bwd_kern_tiled_kernel<4, 3>), grid, dim3(NT), 0, st, p, tiles_x, vec);
            else hipLaunchKernelGGL((cdna_bwd_kern_tiled_kernel<4, 1>), grid, dim3(NT), 0, st, p, tiles_x, vec);
        }
        return LAUNCH_OK();
    }
    const int fast = (a->kh == 5 && a->kw == 5 && a->K == 4 && a->C == 3 && al) ? 3 : ((a->kh == 5 && a->kw == 5 && a->K == 4 && a->C == 1 && al) ? 1 : 0);
    dim3 gimg((a->H * a->W + NT - 1) / NT, a->N);
    if (p.dimg) {
        if (fast == 3) hipLaunchKernelGGL((cdna_bwd_img_fast_kernel<5, 5, 4, 3>), gimg, dim3(NT), 0, st, p);
        else if (fast == 1) hipLaunchKernelGGL((cdna_bwd_img_fast_kernel<5, 5, 4, 1>), gimg, dim3(NT), 0, st, p);
        else hipLaunchKernelGGL(cdna_apply_bwd_img_kernel, gimg, dim3(NT), 0, st, p);
    }
    if (p.dkern) {
        if (fast) {
            const int chunk = 512;
            savp_zero_async(p.dkern, (size_t)a->N * 25 * 4 * sizeof(double), st);
            dim3 gk((a->H * a->W + chunk - 1) / chunk, a->N);
            if (fast == 3) hipLaunchKernelGGL((cdna_bwd_kern_fast_kernel<5, 5, 4, 3>), gk, dim3(NT), 0, st, p, chunk);
            else hipLaunchKernelGGL((cdna_bwd_kern_fast_kernel<5, 5, 4, 1>), gk, dim3(NT), 0, st, p, chunk);
        } else {
            hipLaunchKernelGGL(cdna_apply_bwd_kern_kernel, dim3(a->K, a->N), dim3(NT), 0, st, p);
        }
    }
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// softmax over M mask logits + composite.  logits [N*HW, M] contiguous; timgs view [N,HW,M*C] (channel m*C+c)
// ---------------------------------------------------------------------------------------------------------------
#define MAXM 16
struct CompP {
    int N, HW, M, C;
    const float* logits; int ls;           // [N*HW, ls] (ls >= M: row stride, padded for aligned conv stores)
    const float* timgs; long long t_sn, t_sp;
    float* gen; long long g_sn, g_sp;
    float* masks;                          // optional [N*HW, M]
    const float* dgen; long long dg_sn, dg_sp;
    float* dlogits;                        // [N*HW, ls] (pad columns written as 0)
    float* drow; long long dr_sn, dr_sp;   // gradient row of the whole mask-conv input buffer [N,HW,rowc]
    int toff, rowc;                        // timgs live at channels [toff, toff+M*C); everything else is written as 0
    int nnext; float* next[2]; long long n_sn[2], n_sp[2];      // fwd: the next step's input image slots (SavpCompositeArgs.next)
    const int* gt_mask; const float* gt_img; long long gi_sn, gi_sp;
};

template <int TM, int TC>
__global__ void composite_fwd_kernel(CompP p) {
    const int M = TM ? TM : p.M, C = TC ? TC : p.C;
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)p.N * p.HW) return;
    const int n = (int)(i / p.HW), px = (int)(i % p.HW);
    const float* lg = p.logits + i * p.ls;
    float m[TM ? TM : MAXM];
    float mx = -3.4e38f;
#pragma unroll
    for (int k = 0; k < M; ++k) { m[k] = lg[k]; mx = fmaxf(mx, m[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < M; ++k) { m[k] = __expf(m[k] - mx); s += m[k]; }
    const float inv = 1.f / s;
    const float* t = p.timgs + (long long)n * p.t_sn + (long long)px * p.t_sp;
    float g[TC ? TC : MAXC];
#pragma unroll
    for (int c = 0; c < C; ++c) g[c] = 0.f;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        m[k] *= inv;
#pragma unroll
        for (int c = 0; c < C; ++c) g[c] += m[k] * t[k * C + c];
    }
    float* go = p.gen + (long long)n * p.g_sn + (long long)px * p.g_sp;
#pragma unroll
    for (int c = 0; c < C; ++c) go[c] = g[c];
    if (p.nnext) {        // image of step t+1 = ground truth where scheduled sampling says so, this step's prediction elsewhere
        if (p.gt_mask[n]) {
            const float* gi = p.gt_img + (long long)n * p.gi_sn + (long long)px * p.gi_sp;
#pragma unroll
            for (int c = 0; c < C; ++c) g[c] = gi[c];
        }
        for (int k = 0; k < p.nnext; ++k) {
            float* o = p.next[k] + (long long)n * p.n_sn[k] + (long long)px * p.n_sp[k];
#pragma unroll
            for (int c = 0; c < C; ++c) o[c] = g[c];
        }
    }
    if (p.masks) {
#pragma unroll
        for (int k = 0; k < M; ++k) p.masks[i * M + k] = m[k];
    }
}

template <int TM, int TC>
__global__ void composite_bwd_kernel(CompP p) {
    const int M = TM ? TM : p.M, C = TC ? TC : p.C;
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)p.N * p.HW) return;
    const int n = (int)(i / p.HW), px = (int)(i % p.HW);
    const float* lg = p.logits + i * p.ls;
    float m[TM ? TM : MAXM], sk[TM ? TM : MAXM];
    float mx = -3.4e38f;
#pragma unroll
    for (int k = 0; k < M; ++k) { m[k] = lg[k]; mx = fmaxf(mx, m[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < M; ++k) { m[k] = __expf(m[k] - mx); s += m[k]; }
    const float inv = 1.f / s;
    const float* t = p.timgs + (long long)n * p.t_sn + (long long)px * p.t_sp;
    const float* dg = p.dgen + (long long)n * p.dg_sn + (long long)px * p.dg_sp;
    float* dr = p.drow + (long long)n * p.dr_sn + (long long)px * p.dr_sp;
    float d[TC ? TC : MAXC];
#pragma unroll
    for (int c = 0; c < C; ++c) d[c] = dg[c];
    for (int c = 0; c < p.toff; ++c) dr[c] = 0.f;
    for (int c = p.toff + M * C; c < p.rowc; ++c) dr[c] = 0.f;
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        m[k] *= inv;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            a += d[c] * t[k * C + c];
            dr[p.toff + k * C + c] = m[k] * d[c];
        }
        sk[k] = a;
        dot += m[k] * a;
    }
    float* dl = p.dlogits + i * p.ls;
#pragma unroll
    for (int k = 0; k < M; ++k) dl[k] = m[k] * (sk[k] - dot);
    for (int k = M; k < p.ls; ++k) dl[k] = 0.f;
}

// Coalesced form of composite_bwd for rows that are contiguous over all pixels (the mask-conv input buffer [N,HW,rowc] and its
// gradient twin): the thread-per-pixel kernel above stores its 56-float gradient row with 56 scalar stores at a lane stride of
// 224 B (every wave instruction touches 64 lines) and gathers the 21 transformed-image channels the same way.  Here a workgroup
// (256 pixels) moves both through LDS: 16-byte pieces are loaded / stored with consecutive lanes on consecutive addresses, each
// thread picks up / deposits the pieces of its own pixel in between.  Requires toff % 4 == 0, rowc % 4 == 0, 16-byte aligned bases.
template <int TM, int TC>
__global__ __launch_bounds__(NT) void composite_bwd_tiled_kernel(CompP p) {
    constexpr int MC = TM * TC;
    constexpr int NPF = (MC + 3) / 4;                       // 16-byte pieces holding the transformed images of one pixel
    __shared__ float4 tile[NT * NPF];
    const long long tot = (long long)p.N * p.HW;
    const long long i0 = blockIdx.x * (long long)NT;
    const int tid = threadIdx.x;
    // ---- transformed images of the 256 pixels -> LDS (timgs points at channel toff of the value buffer; rows contiguous) ----
    for (int f = tid; f < NT * NPF; f += NT) {
        const int pl = f / NPF, q = f - pl * NPF;
        const long long gi = i0 + pl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gi < tot) v = *reinterpret_cast<const float4*>(p.timgs + gi * p.t_sp + 4 * q);
        tile[f] = v;
    }
    __syncthreads();
    const long long i = i0 + tid;
    if (i < tot) {
        float t[NPF * 4];
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const float4 v = tile[tid * NPF + q];
            t[4 * q] = v.x; t[4 * q + 1] = v.y; t[4 * q + 2] = v.z; t[4 * q + 3] = v.w;
        }
        const float* lg = p.logits + i * p.ls;
        float m[TM], sk[TM];
        float mx = -3.4e38f;
#pragma unroll
        for (int k = 0; k < TM; ++k) { m[k] = lg[k]; mx = fmaxf(mx, m[k]); }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < TM; ++k) { m[k] = __expf(m[k] - mx); s += m[k]; }
        const float inv = 1.f / s;
        const int n = (int)(i / p.HW), px = (int)(i - (long long)n * p.HW);
        const float* dg = p.dgen + (long long)n * p.dg_sn + (long long)px * p.dg_sp;
        float d[TC];
#pragma unroll
        for (int c = 0; c < TC; ++c) d[c] = dg[c];
        float o[NPF * 4];
#pragma unroll
        for (int j = 0; j < NPF * 4; ++j) o[j] = 0.f;
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < TM; ++k) {
            m[k] *= inv;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < TC; ++c) {
                a += d[c] * t[k * TC + c];
                o[k * TC + c] = m[k] * d[c];
            }
            sk[k] = a;
            dot += m[k] * a;
        }
#pragma unroll
        for (int q = 0; q < NPF; ++q) tile[tid * NPF + q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);   // own row only
        float* dl = p.dlogits + i * p.ls;
#pragma unroll
        for (int k = 0; k < TM; ++k) dl[k] = m[k] * (sk[k] - dot);
        for (int k = TM; k < p.ls; ++k) dl[k] = 0.f;
    }
    __syncthreads();
    // ---- gradient rows out: rowc / 4 pieces per pixel, zeros outside [toff, toff + 4 NPF) ----
    const int rp4 = p.rowc >> 2, q0 = p.toff >> 2;
    const int npix = (int)min((long long)NT, tot - i0);
    for (int f = tid; f < npix * rp4; f += NT) {
        const int pl = f / rp4, q = f - pl * rp4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q >= q0 && q < q0 + NPF) v = tile[pl * NPF + q - q0];
        *reinterpret_cast<float4*>(p.drow + (i0 + pl) * p.dr_sp + 4 * q) = v;
    }
}

static int fill_comp(CompP& p, const SavpCompositeArgs* a) {
    if (!a || a->M > MAXM || a->C > MAXC || a->M < 1 || a->C < 1 || a->logits_stride < a->M) return SAVP_EINVAL;
    p.N = a->N; p.HW = a->HW; p.M = a->M; p.C = a->C;
    p.logits = a->logits; p.ls = a->logits_stride;
    p.timgs = (const float*)a->timgs.p; p.t_sn = a->timgs.sn; p.t_sp = a->timgs.sp;
    p.gen = (float*)a->gen.p; p.g_sn = a->gen.sn; p.g_sp = a->gen.sp;
    p.masks = a->masks;
    p.dgen = (const float*)a->dgen.p; p.dg_sn = a->dgen.sn; p.dg_sp = a->dgen.sp;
    p.dlogits = a->dlogits;
    p.drow = (float*)a->drow.p; p.dr_sn = a->drow.sn; p.dr_sp = a->drow.sp;
    p.toff = a->timgs_offset; p.rowc = a->row_channels;
    p.nnext = a->nnext;
    if (a->nnext < 0 || a->nnext > 2 || (a->nnext && (!a->gt_mask || !a->gt_img.p))) return SAVP_EINVAL;
    for (int k = 0; k < a->nnext; ++k) {
        if (!a->next[k].p) return SAVP_EINVAL;
        p.next[k] = (float*)a->next[k].p; p.n_sn[k] = a->next[k].sn; p.n_sp[k] = a->next[k].sp;
    }
    p.gt_mask = a->gt_mask; p.gt_img = (const float*)a->gt_img.p; p.gi_sn = a->gt_img.sn; p.gi_sp = a->gt_img.sp;
    return SAVP_OK;
}

extern "C" int savp_composite_fwd(void* stream, const SavpCompositeArgs* a) {
    CompP p;
    int rc = fill_comp(p, a);
    if (rc) return rc;
    long long tot = (long long)a->N * a->HW;
    dim3 grid((unsigned)((tot + NT - 1) / NT));
    if (a->M == 7 && a->C == 3) hipLaunchKernelGGL((composite_fwd_kernel<7, 3>), grid, dim3(NT), 0, (hipStream_t)stream, p);
    else if (a->M == 7 && a->C == 1) hipLaunchKernelGGL((composite_fwd_kernel<7, 1>), grid, dim3(NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((composite_fwd_kernel<0, 0>), grid, dim3(NT), 0, (hipStream_t)stream, p);
    return LAUNCH_OK();
}

extern "C" int savp_composite_bwd(void* stream, const SavpCompositeArgs* a) {
    CompP p;
    int rc = fill_comp(p, a);
    if (rc) return rc;
    if (!p.drow || !p.dlogits || a->timgs_offset + a->M * a->C > a->row_channels) return SAVP_EINVAL;
    long long tot = (long long)a->N * a->HW;
    dim3 grid((unsigned)((tot + NT - 1) / NT));
    // coalesced form: value rows (timgs) and gradient rows contiguous over all pixels, 16-byte pieces
    const int npf4 = ((a->M * a->C + 3) / 4) * 4;
    const bool tiled = !cdna_legacy() && a->M == 7 && (a->C == 3 || a->C == 1) && (a->timgs_offset % 4 == 0) && (a->row_channels % 4 == 0) &&
                       a->timgs_offset + npf4 <= a->row_channels && p.dr_sp == a->row_channels && p.dr_sn == (long long)a->HW * p.dr_sp &&
                       p.t_sp == a->row_channels && p.t_sn == (long long)a->HW * p.t_sp && ((uintptr_t)p.drow % 16 == 0) &&
                       ((uintptr_t)p.timgs % 16 == 0);
    if (tiled && a->C == 3) hipLaunchKernelGGL((composite_bwd_tiled_kernel<7, 3>), grid, dim3(NT), 0, (hipStream_t)stream, p);
    else if (tiled && a->C == 1) hipLaunchKernelGGL((composite_bwd_tiled_kernel<7, 1>), grid, dim3(NT), 0, (hipStream_t)stream, p);
    else if (a->M == 7 && a->C == 3) hipLaunchKernelGGL((composite_bwd_kernel<7, 3>), grid, dim3(NT), 0, (hipStream_t)stream, p);
    else if (a->M == 7 && a->C == 1) hipLaunchKernelGGL((composite_bwd_kernel<7, 1>), grid, dim3(NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((composite_bwd_kernel<0, 0>), grid, dim3(NT), 0, (hipStream_t)stream, p);
    return LAUNCH_OK();
}
