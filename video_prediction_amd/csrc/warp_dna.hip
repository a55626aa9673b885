// warp_dna.hip -- the two alternative pixel transformations of SAVPCell (hparams.transformation):
//   image_warp fwd/bwd : flow_ops.image_warp (flow_ops.py:4-79) + apply_flows (savp_model.py:955-965): backward
//                        bilinear warp with clamped gathers; gradient w.r.t. the flow (through the bilinear weights,
//                        floor() has zero gradient) and w.r.t. the image (scatter-add of the 4 corner weights).
//   dna_apply fwd/bwd  : per-pixel 5x5 kernels (savp_model.py:541-544,556-559,858-890): identity added, relu-shift,
//                        normalised over the taps, applied to the SYMMETRIC-padded image.  The reference goes through
//                        extract_image_patches + batched matmul; here one thread per pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "savp_hip.h"
#include "zero_fill.h"

#define NT 256
#define RELU_SHIFT 1e-12f
#define MAXC 4
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH)

struct WarpP {
    int N, H, W, C, K;
    const float* img; long long i_sn, i_sp;
    const float* flows;                        // [N,H,W,2K] contiguous, channel comp*K + k (comp 0 = x, 1 = y)
    float* out; long long o_sn, o_sp;          // [N,H,W,K*C] channel k*C + c
    const float* dout; long long do_sn, do_sp;
    float* dflows;                             // [N,H,W,2K]
    float* dimg;                               // [N,H,W,C] contiguous, pre-zeroed (atomics); may be null
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <bool BWD>
__global__ void image_warp_kernel(WarpP p) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long total = (long long)p.N * p.H * p.W * p.K;
    if (i >= total) return;
    const int k = (int)(i % p.K);
    long long r = i / p.K;
    const int px = (int)(r % (p.H * p.W));
    const int n = (int)(r / (p.H * p.W));
    const int y = px / p.W, x = px % p.W;
    const float* fl = p.flows + ((long long)n * p.H * p.W + px) * 2 * p.K;
    const float fx = fl[k], fy = fl[p.K + k];
    const float ffx = floorf(fx), ffy = floorf(fy);
    const float xw = fx - ffx, yw = fy - ffy;
    const int x0 = clampi(x + (int)ffx, 0, p.W - 1), x1 = clampi(x + (int)ffx + 1, 0, p.W - 1);
    const int y0 = clampi(y + (int)ffy, 0, p.H - 1), y1 = clampi(y + (int)ffy + 1, 0, p.H - 1);
    const float* im = p.img + (long long)n * p.i_sn;
    const float* Ia = im + (long long)(y0 * p.W + x0) * p.i_sp;      // top left
    const float* Ib = im + (long long)(y1 * p.W + x0) * p.i_sp;      // bottom left
    const float* Ic = im + (long long)(y0 * p.W + x1) * p.i_sp;      // top right
    const float* Id = im + (long long)(y1 * p.W + x1) * p.i_sp;      // bottom right
    const float wa = (1.f - xw) * (1.f - yw), wb = (1.f - xw) * yw, wc = xw * (1.f - yw), wd = xw * yw;
    if (!BWD) {
        float* o = p.out + (long long)n * p.o_sn + (long long)px * p.o_sp + k * p.C;
        for (int c = 0; c < p.C; ++c) o[c] = wa * Ia[c] + wb * Ib[c] + wc * Ic[c] + wd * Id[c];
    } else {
        const float* d = p.dout + (long long)n * p.do_sn + (long long)px * p.do_sp + k * p.C;
        float gx = 0.f, gy = 0.f;
        for (int c = 0; c < p.C; ++c) {
            const float a = Ia[c], b = Ib[c], cc = Ic[c], dd = Id[c], g = d[c];
            gx += g * (-(1.f - yw) * a - yw * b + (1.f - yw) * cc + yw * dd);
            gy += g * (-(1.f - xw) * a + (1.f - xw) * b - xw * cc + xw * dd);
            if (p.dimg) {
                float* di = p.dimg + (long long)n * p.H * p.W * p.C;
                unsafeAtomicAdd(di + (long long)(y0 * p.W + x0) * p.C + c, wa * g);
                unsafeAtomicAdd(di + (long long)(y1 * p.W + x0) * p.C + c, wb * g);
                unsafeAtomicAdd(di + (long long)(y0 * p.W + x1) * p.C + c, wc * g);
                unsafeAtomicAdd(di + (long long)(y1 * p.W + x1) * p.C + c, wd * g);
            }
        }
        float* df = p.dflows + ((long long)n * p.H * p.W + px) * 2 * p.K;
        df[k] = gx; df[p.K + k] = gy;
    }
}

static int fill_warp(WarpP& p, const SavpWarpArgs* a) {
    if (!a || a->C < 1 || a->K < 1 || !a->img.p || !a->flows) return SAVP_EINVAL;
    p.N = a->N; p.H = a->H; p.W = a->W; p.C = a->C; p.K = a->K;
    p.img = (const float*)a->img.p; p.i_sn = a->img.sn; p.i_sp = a->img.sp;
    p.flows = a->flows;
    p.out = (float*)a->out.p; p.o_sn = a->out.sn; p.o_sp = a->out.sp;
    p.dout = (const float*)a->dout.p; p.do_sn = a->dout.sn; p.do_sp = a->dout.sp;
    p.dflows = a->dflows; p.dimg = a->dimg;
    return SAVP_OK;
}

extern "C" int savp_image_warp_fwd(void* stream, const SavpWarpArgs* a) {
    WarpP p;
    int rc = fill_warp(p, a);
    if (rc || !p.out) return SAVP_EINVAL;
    long long total = (long long)a->N * a->H * a->W * a->K;
    hipLaunchKernelGGL((image_warp_kernel<false>), dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, p);
    return LAUNCH_OK();
}

extern "C" int savp_image_warp_bwd(void* stream, const SavpWarpArgs* a) {
    WarpP p;
    int rc = fill_warp(p, a);
    if (rc || !p.dout || !p.dflows) return SAVP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (p.dimg) savp_zero_async(p.dimg, (size_t)a->N * a->H * a->W * a->C * sizeof(float), st);
    long long total = (long long)a->N * a->H * a->W * a->K;
    hipLaunchKernelGGL((image_warp_kernel<true>), dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, st, p);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// DNA
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int symi(int q, int n) { return q < 0 ? -q - 1 : (q >= n ? 2 * n - 1 - q : q); }
__device__ __forceinline__ float ident5(int u, int v, int kh, int kw) {
    float fu, fv;
    if (kh & 1) fu = (u == kh / 2) ? 1.f : 0.f; else fu = (u == kh / 2 - 1 || u == kh / 2) ? 0.5f : 0.f;
    if (kw & 1) fv = (v == kw / 2) ? 1.f : 0.f; else fv = (v == kw / 2 - 1 || v == kw / 2) ? 0.5f : 0.f;
    return fu * fv;
}

struct DnaP {
    int N, H, W, C, K, kh, kw;
    const float* img; long long i_sn, i_sp;
    const float* raw;                          // [N,HW,taps*K] channel t*K + k (conv output)
    float* kern;                               // [N,HW,taps*K] normalised kernels (saved by fwd, read by bwd)
    float* out; long long o_sn, o_sp;
    const float* dout; long long do_sn, do_sp;
    float* draw;                               // [N,HW,taps*K]
    float* dimg; long long di_sn, di_sp; int dimg_beta;
};

__global__ void dna_fwd_kernel(DnaP p) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)p.N * p.H * p.W) return;
    const int n = (int)(i / (p.H * p.W)), px = (int)(i % (p.H * p.W));
    const int y = px / p.W, x = px % p.W;
    const int taps = p.kh * p.kw, pt = (p.kh - 1) / 2, pl = (p.kw - 1) / 2;
    const float* r = p.raw + i * taps * p.K;
    float* kn = p.kern + i * taps * p.K;
    const float* im = p.img + (long long)n * p.i_sn;
    float* o = p.out + (long long)n * p.o_sn + (long long)px * p.o_sp;
    for (int k = 0; k < p.K; ++k) {
        float s = 0.f;
        for (int t = 0; t < taps; ++t) s += fmaxf(r[t * p.K + k] + ident5(t / p.kw, t % p.kw, p.kh, p.kw) - RELU_SHIFT, 0.f) + RELU_SHIFT;
        const float inv = 1.f / s;
        float acc[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
        for (int t = 0; t < taps; ++t) {
            const float w = (fmaxf(r[t * p.K + k] + ident5(t / p.kw, t % p.kw, p.kh, p.kw) - RELU_SHIFT, 0.f) + RELU_SHIFT) * inv;
            kn[t * p.K + k] = w;
            const float* q = im + (long long)(symi(y + t / p.kw - pt, p.H) * p.W + symi(x + t % p.kw - pl, p.W)) * p.i_sp;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < p.C) acc[c] += q[c] * w;
        }
        for (int c = 0; c < p.C; ++c) o[k * p.C + c] = acc[c];
    }
}

// per-pixel kernel gradient + normalisation backward
__global__ void dna_bwd_kern_kernel(DnaP p) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)p.N * p.H * p.W) return;
    const int n = (int)(i / (p.H * p.W)), px = (int)(i % (p.H * p.W));
    const int y = px / p.W, x = px % p.W;
    const int taps = p.kh * p.kw, pt = (p.kh - 1) / 2, pl = (p.kw - 1) / 2;
    const float* r = p.raw + i * taps * p.K;
    const float* kn = p.kern + i * taps * p.K;
    float* dr = p.draw + i * taps * p.K;
    const float* im = p.img + (long long)n * p.i_sn;
    const float* d = p.dout + (long long)n * p.do_sn + (long long)px * p.do_sp;
    for (int k = 0; k < p.K; ++k) {
        float s = 0.f;
        for (int t = 0; t < taps; ++t) s += fmaxf(r[t * p.K + k] + ident5(t / p.kw, t % p.kw, p.kh, p.kw) - RELU_SHIFT, 0.f) + RELU_SHIFT;
        const float inv = 1.f / s;
        float dot = 0.f;
        // first pass: dkern (stored temporarily in draw) and its dot with the normalised kernel
        for (int t = 0; t < taps; ++t) {
            const float* q = im + (long long)(symi(y + t / p.kw - pt, p.H) * p.W + symi(x + t % p.kw - pl, p.W)) * p.i_sp;
            float g = 0.f;
            for (int c = 0; c < p.C; ++c) g += q[c] * d[k * p.C + c];
            dr[t * p.K + k] = g;
            dot += g * kn[t * p.K + k];
        }
        for (int t = 0; t < taps; ++t) {
            const float pre = r[t * p.K + k] + ident5(t / p.kw, t % p.kw, p.kh, p.kw) - RELU_SHIFT;
            dr[t * p.K + k] = pre > 0.f ? (dr[t * p.K + k] - dot) * inv : 0.f;
        }
    }
}

// image gradient in gather form (mirrored positions as in cdna_composite.hip)
__global__ void dna_bwd_img_kernel(DnaP p) {
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= (long long)p.N * p.H * p.W) return;
    const int n = (int)(i / (p.H * p.W)), px = (int)(i % (p.H * p.W));
    const int sy = px / p.W, sx = px % p.W;
    const int taps = p.kh * p.kw, pt = (p.kh - 1) / 2, pl = (p.kw - 1) / 2;
    const int pb = p.kh - 1 - pt, pr = p.kw - 1 - pl;
    int qy[3], nqy = 0, qx[3], nqx = 0;
    qy[nqy++] = sy;
    if (-sy - 1 >= -pt) qy[nqy++] = -sy - 1;
    if (2 * p.H - 1 - sy < p.H + pb && 2 * p.H - 1 - sy >= p.H) qy[nqy++] = 2 * p.H - 1 - sy;
    qx[nqx++] = sx;
    if (-sx - 1 >= -pl) qx[nqx++] = -sx - 1;
    if (2 * p.W - 1 - sx < p.W + pr && 2 * p.W - 1 - sx >= p.W) qx[nqx++] = 2 * p.W - 1 - sx;
    float acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
    const float* dbase = p.dout + (long long)n * p.do_sn;
    const float* kbase = p.kern + (long long)n * p.H * p.W * taps * p.K;
    for (int a = 0; a < nqy; ++a)
        for (int b = 0; b < nqx; ++b)
            for (int u = 0; u < p.kh; ++u) {
                const int y = qy[a] - u + pt;
                if (y < 0 || y >= p.H) continue;
                for (int v = 0; v < p.kw; ++v) {
                    const int x = qx[b] - v + pl;
                    if (x < 0 || x >= p.W) continue;
                    const long long op = (long long)y * p.W + x;
                    const float* d = dbase + op * p.do_sp;
                    const float* kk = kbase + op * taps * p.K + (u * p.kw + v) * p.K;
                    for (int k = 0; k < p.K; ++k) {
                        const float w = kk[k];
#pragma unroll
                        for (int c = 0; c < MAXC; ++c)
                            if (c < p.C) acc[c] += d[k * p.C + c] * w;
                    }
                }
            }
    float* di = p.dimg + (long long)n * p.di_sn + (long long)px * p.di_sp;
    for (int c = 0; c < p.C; ++c) di[c] = p.dimg_beta ? di[c] + acc[c] : acc[c];
}

static int fill_dna(DnaP& p, const SavpDnaArgs* a) {
    if (!a || a->C < 1 || a->C > MAXC || a->K < 1 || !a->img.p || !a->raw || !a->kern) return SAVP_EINVAL;
    p.N = a->N; p.H = a->H; p.W = a->W; p.C = a->C; p.K = a->K; p.kh = a->kh; p.kw = a->kw;
    p.img = (const float*)a->img.p; p.i_sn = a->img.sn; p.i_sp = a->img.sp;
    p.raw = a->raw; p.kern = a->kern;
    p.out = (float*)a->out.p; p.o_sn = a->out.sn; p.o_sp = a->out.sp;
    p.dout = (const float*)a->dout.p; p.do_sn = a->dout.sn; p.do_sp = a->dout.sp;
    p.draw = a->draw;
    p.dimg = (float*)a->dimg.p; p.di_sn = a->dimg.sn; p.di_sp = a->dimg.sp; p.dimg_beta = a->dimg_beta;
    return SAVP_OK;
}

extern "C" int savp_dna_apply_fwd(void* stream, const SavpDnaArgs* a) {
    DnaP p;
    int rc = fill_dna(p, a);
    if (rc || !p.out) return SAVP_EINVAL;
    long long total = (long long)a->N * a->H * a->W;
    hipLaunchKernelGGL(dna_fwd_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, p);
    return LAUNCH_OK();
}

extern "C" int savp_dna_apply_bwd(void* stream, const SavpDnaArgs* a) {
    DnaP p;
    int rc = fill_dna(p, a);
    if (rc || !p.dout || !p.draw) return SAVP_EINVAL;
    long long total = (long long)a->N * a->H * a->W;
    dim3 grid((unsigned)((total + NT - 1) / NT));
    hipLaunchKernelGGL(dna_bwd_kern_kernel, grid, dim3(NT), 0, (hipStream_t)stream, p);
    if (p.dimg) hipLaunchKernelGGL(dna_bwd_img_kernel, grid, dim3(NT), 0, (hipStream_t)stream, p);
    return LAUNCH_OK();
}
