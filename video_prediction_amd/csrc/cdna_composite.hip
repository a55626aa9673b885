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
#include "savp_hip.h"

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

// raw [N, kh*kw*K] (index (u*kw+v)*K + k) -> normalised kern, same layout.  one thread per (n, k)
__global__ void cdna_kernels_fwd_kernel(const float* __restrict__ raw, float* __restrict__ kern, int N, int kh, int kw, int K) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * K) return;
    int n = i / K, k = i % K;
    const float* r = raw + (long long)n * kh * kw * K + k;
    float* o = kern + (long long)n * kh * kw * K + k;
    float s = 0.f;
    for (int t = 0; t < kh * kw; ++t) {
        float v = fmaxf(r[t * K] + ident_at(t / kw, t % kw, kh, kw) - RELU_SHIFT, 0.f) + RELU_SHIFT;
        s += v;
    }
    float inv = 1.f / s;
    for (int t = 0; t < kh * kw; ++t) {
        float v = fmaxf(r[t * K] + ident_at(t / kw, t % kw, kh, kw) - RELU_SHIFT, 0.f) + RELU_SHIFT;
        o[t * K] = v * inv;
    }
}

// draw = ((dkern - sum(dkern*kern)) / s) * [raw + ident - shift > 0]
__global__ void cdna_kernels_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ dkern, float* __restrict__ draw,
                                        int N, int kh, int kw, int K) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * K) return;
    int n = i / K, k = i % K;
    const long long base = (long long)n * kh * kw * K + k;
    float s = 0.f;
    for (int t = 0; t < kh * kw; ++t) s += fmaxf(raw[base + t * K] + ident_at(t / kw, t % kw, kh, kw) - RELU_SHIFT, 0.f) + RELU_SHIFT;
    float inv = 1.f / s;
    float dot = 0.f;
    for (int t = 0; t < kh * kw; ++t) {
        float v = fmaxf(raw[base + t * K] + ident_at(t / kw, t % kw, kh, kw) - RELU_SHIFT, 0.f) + RELU_SHIFT;
        dot += dkern[base + t * K] * v * inv;
    }
    for (int t = 0; t < kh * kw; ++t) {
        float pre = raw[base + t * K] + ident_at(t / kw, t % kw, kh, kw) - RELU_SHIFT;
        draw[base + t * K] = pre > 0.f ? (dkern[base + t * K] - dot) * inv : 0.f;
    }
}

extern "C" int savp_cdna_kernels_fwd(void* stream, const float* raw, float* kern, int32_t N, int32_t kh, int32_t kw, int32_t K) {
    if (!raw || !kern) return SAVP_EINVAL;
    hipLaunchKernelGGL(cdna_kernels_fwd_kernel, dim3((N * K + 63) / 64), dim3(64), 0, (hipStream_t)stream, raw, kern, N, kh, kw, K);
    return LAUNCH_OK();
}
extern "C" int savp_cdna_kernels_bwd(void* stream, const float* raw, const float* dkern, float* draw, int32_t N, int32_t kh,
                                     int32_t kw, int32_t K) {
    if (!raw || !dkern || !draw) return SAVP_EINVAL;
    hipLaunchKernelGGL(cdna_kernels_bwd_kernel, dim3((N * K + 63) / 64), dim3(64), 0, (hipStream_t)stream, raw, dkern, draw, N, kh, kw, K);
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
    float* dkern;                                // [N, kh*kw, K]  (overwritten)
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
        p.dkern[((long long)n * taps + t) * p.K + k] = sh[t] + sh[MAXTAPS + t] + sh[2 * MAXTAPS + t] + sh[3 * MAXTAPS + t];
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
        unsafeAtomicAdd(p.dkern + (long long)n * NV + i, sh[i] + sh[NV + i] + sh[2 * NV + i] + sh[3 * NV + i]);
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
    p.dkern = a->dkern;
    return SAVP_OK;
}

extern "C" int savp_cdna_apply_fwd(void* stream, const SavpCdnaArgs* a) {
    CdnaP p;
    int rc = fill_cdna(p, a);
    if (rc) return rc;
    hipLaunchKernelGGL(cdna_apply_fwd_kernel, dim3((a->H * a->W + NT - 1) / NT, a->N), dim3(NT), 0, (hipStream_t)stream, p);
    return LAUNCH_OK();
}

extern "C" int savp_cdna_apply_bwd(void* stream, const SavpCdnaArgs* a) {
    CdnaP p;
    int rc = fill_cdna(p, a);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool al = ((uintptr_t)a->dout.p % 16 == 0) && (a->dout.sn % 4 == 0) && (a->dout.sp % 4 == 0);
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
            hipMemsetAsync(p.dkern, 0, (size_t)a->N * 25 * 4 * sizeof(float), st);
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
    if (a->M == 7 && a->C == 3) hipLaunchKernelGGL((composite_bwd_kernel<7, 3>), grid, dim3(NT), 0, (hipStream_t)stream, p);
    else if (a->M == 7 && a->C == 1) hipLaunchKernelGGL((composite_bwd_kernel<7, 1>), grid, dim3(NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((composite_bwd_kernel<0, 0>), grid, dim3(NT), 0, (hipStream_t)stream, p);
    return LAUNCH_OK();
}
