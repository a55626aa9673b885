// weight_prep.hip -- per-step weight preparation (the reference does these as TF graph algebra on every run):
//   pack_weights     : HWIO master weights -> the k-contiguous layouts the implicit-GEMM kernel streams
//                      (WT[Cy][taps*Cx] for FPROP, WD[Cx][taps*Cy] for DGRAD), optionally scaled by a device scalar
//                      (1/sigma of spectral normalisation).
//   fold_pool        : conv_pool2d's avg-pool folding of the kernel (ops.py:838-842) and its adjoint.
//   fold_bilinear    : upsample_conv2d's bilinear folding (ops.py:697-704) and its adjoint.
//   sn_*             : spectral_normed_weight (ops.py:1020-1049), one power iteration, forward and the full
//                      backward (gradients flow through sigma, u', v -- the reference has no stop_gradient).
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

// src [T, Cx, Cy]
__global__ void pack_weights_kernel(const float* __restrict__ src, long long T, int Cx, int Cy, const float* scale, float* wt,
                                    float* wd, __bf16* wt16, __bf16* wd16) {
    const long long total = T * Cx * Cy;
    const float s = scale ? *scale : 1.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cy = (int)(i % Cy);
        long long r = i / Cy;
        int cx = (int)(r % Cx);
        long long t = r / Cx;
        float v = src[i] * s;
        if (wt) wt[(long long)cy * (T * Cx) + t * Cx + cx] = v;
        if (wd) wd[(long long)cx * (T * Cy) + t * Cy + cy] = v;
        if (wt16) wt16[(long long)cy * (T * Cx) + t * Cx + cx] = (__bf16)v;
        if (wd16) wd16[(long long)cx * (T * Cy) + t * Cy + cy] = (__bf16)v;
    }
}

extern "C" int savp_pack_weights(void* stream, const float* src, int64_t T, int32_t Cx, int32_t Cy, const float* scale,
                                 float* wt, float* wd, void* wt_bf16, void* wd_bf16) {
    if (!src || (!wt && !wd && !wt_bf16 && !wd_bf16)) return SAVP_EINVAL;
    long long total = (long long)T * Cx * Cy;
    unsigned nb = (unsigned)((total + NT - 1) / NT);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, src, (long long)T, Cx, Cy, scale, wt, wd,
                       (__bf16*)wt_bf16, (__bf16*)wd_bf16);
    return LAUNCH_OK();
}

// Several packs in ONE launch (a network's layers after their optimiser step): blockIdx.y = layer, grid-stride over its
// elements.  52 single launches of ~11 us each per train step otherwise.
#define PACK_MAX 32
struct PackBatch {
    const float* src[PACK_MAX]; const float* scale[PACK_MAX];
    float* wt[PACK_MAX]; float* wd[PACK_MAX]; __bf16* wt16[PACK_MAX]; __bf16* wd16[PACK_MAX];
    long long T[PACK_MAX]; int Cx[PACK_MAX], Cy[PACK_MAX];
};

// A = src viewed as [R = T * Cx rows][Cy]: wt = A^T (the transposed copies go through a 64 x 64 LDS tile so that BOTH sides are
// coalesced -- written straight from the element loop they were 4- / 2-byte scatters with a stride of R and made the pack cost as
// much as 0.6 ms of a 66 ms step), wd = the rows regrouped by input channel (runs of Cy, coalesced as they are).
#define PK_T 64
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(PackBatch b) {
    __shared__ float tile[PK_T][PK_T + 1];
    const int it = blockIdx.y;
    const float* __restrict__ src = b.src[it];
    const long long T = b.T[it];
    const int Cx = b.Cx[it], Cy = b.Cy[it];
    float* __restrict__ wt = b.wt[it]; float* __restrict__ wd = b.wd[it];
    __bf16* __restrict__ wt16 = b.wt16[it]; __bf16* __restrict__ wd16 = b.wd16[it];
    const long long R = T * Cx;
    const float s = b.scale[it] ? *b.scale[it] : 1.f;
    const int tc = (Cy + PK_T - 1) / PK_T;
    const long long tiles = ((R + PK_T - 1) / PK_T) * tc;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;            // 64 x 4 threads
    for (long long tl = blockIdx.x; tl < tiles; tl += gridDim.x) {
        const long long r0 = (tl / tc) * PK_T;
        const int c0 = (int)(tl % tc) * PK_T;
        __syncthreads();
#pragma unroll 4
        for (int j = ty; j < PK_T; j += 4) {                           // row r0 + j, column c0 + tx: coalesced along Cy
            const long long r = r0 + j;
            const int cy = c0 + tx;
            if (r < R && cy < Cy) {
                const float v = src[r * Cy + cy] * s;
                tile[j][tx] = v;
                const long long t = r / Cx;
                const int cx = (int)(r - t * Cx);
                const long long o = (long long)cx * (T * Cy) + t * Cy + cy;
                if (wd) wd[o] = v;
                if (wd16) wd16[o] = (__bf16)v;
            }
        }
        __syncthreads();
        if (wt || wt16) {
#pragma unroll 4
            for (int j = ty; j < PK_T; j += 4) {                       // column c0 + j of A = row of wt, element r0 + tx: coalesced along R
                const int cy = c0 + j;
                const long long r = r0 + tx;
                if (r < R && cy < Cy) {
                    const float v = tile[tx][j];
                    if (wt) wt[(long long)cy * R + r] = v;
                    if (wt16) wt16[(long long)cy * R + r] = (__bf16)v;
                }
            }
        }
    }
}

extern "C" int savp_pack_weights_batch(void* stream, int32_t n, const SavpPackItem* items) {
    if (!items || n < 1 || n > PACK_MAX) return SAVP_EINVAL;
    PackBatch b;
    long long most = 0;
    for (int i = 0; i < n; ++i) {
        const SavpPackItem& q = items[i];
        if (!q.src || (!q.wt && !q.wd && !q.wt_bf16 && !q.wd_bf16) || q.T < 1 || q.Cx < 1 || q.Cy < 1) return SAVP_EINVAL;
        b.src[i] = q.src; b.scale[i] = q.scale; b.wt[i] = q.wt; b.wd[i] = q.wd;
        b.wt16[i] = (__bf16*)q.wt_bf16; b.wd16[i] = (__bf16*)q.wd_bf16;
        b.T[i] = q.T; b.Cx[i] = q.Cx; b.Cy[i] = q.Cy;
        const long long tiles = (((long long)q.T * q.Cx + PK_T - 1) / PK_T) * ((q.Cy + PK_T - 1) / PK_T);
        if (tiles > most) most = tiles;
    }
    long long nb = most;                                          // 64 x 64 tiles of the largest layer (smaller layers: idle blocks exit)
    if (nb > 512) nb = 512;
    hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)nb, (unsigned)n), dim3(256), 0, (hipStream_t)stream, b);
    return LAUNCH_OK();
}

// fold_pool: src [k,k,C] -> dst [k+1,k+1,C];  adjoint: dsrc[k,k,C] += from ddst
__global__ void fold_pool_kernel(const float* __restrict__ in, float* __restrict__ out, int k, long long C, int adjoint) {
    const int ko = k + 1;
    const long long total = adjoint ? (long long)k * k * C : (long long)ko * ko * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long c = i % C;
        long long r = i / C;
        float s = 0.f;
        if (!adjoint) {
            int b = (int)(r % ko), a = (int)(r / ko);
            for (int di = 0; di < 2; ++di)
                for (int dj = 0; dj < 2; ++dj) {
                    int u = a - di, v = b - dj;
                    if (u >= 0 && u < k && v >= 0 && v < k) s += in[((long long)u * k + v) * C + c];
                }
            out[i] = 0.25f * s;
        } else {
            int v = (int)(r % k), u = (int)(r / k);
            for (int di = 0; di < 2; ++di)
                for (int dj = 0; dj < 2; ++dj) s += in[((long long)(u + di) * ko + (v + dj)) * C + c];
            out[i] += 0.25f * s;
        }
    }
}

extern "C" int savp_fold_pool(void* stream, const float* in, float* out, int32_t k, int64_t C, int32_t adjoint) {
    if (!in || !out || k < 1) return SAVP_EINVAL;
    long long total = (long long)(k + 1) * (k + 1) * C;
    unsigned nb = (unsigned)((total + NT - 1) / NT);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(fold_pool_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, in, out, k, (long long)C, adjoint);
    return LAUNCH_OK();
}

// fold_bilinear (stride 2): W [k,k,Cin,F] -> Kup [k+3,k+3,F,Cin];  adjoint: dW += from dKup
__device__ __forceinline__ float bil1(int i) { return (i == 0 || i == 3) ? 0.25f : ((i == 1 || i == 2) ? 0.75f : 0.f); }

__global__ void fold_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int k, int Cin, int F, int adjoint) {
    const int ko = k + 3;
    const long long total = adjoint ? (long long)k * k * Cin * F : (long long)ko * ko * Cin * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        if (!adjoint) {
            // out index: ((a*ko + b)*F + f)*Cin + ci
            int ci = (int)(i % Cin); long long r = i / Cin;
            int f = (int)(r % F); r /= F;
            int b = (int)(r % ko), a = (int)(r / ko);
            float s = 0.f;
            for (int u = 0; u < k; ++u) {
                float wu = bil1(a + u - (k - 1));
                if (wu == 0.f) continue;
                for (int v = 0; v < k; ++v) {
                    float wv = bil1(b + v - (k - 1));
                    if (wv == 0.f) continue;
                    s += wu * wv * in[(((long long)u * k + v) * Cin + ci) * F + f];
                }
            }
            out[i] = s;
        } else {
            // out (dW) index: ((u*k + v)*Cin + ci)*F + f ; in = dKup
            int f = (int)(i % F); long long r = i / F;
            int ci = (int)(r % Cin); r /= Cin;
            int v = (int)(r % k), u = (int)(r / k);
            float s = 0.f;
            for (int a = 0; a < ko; ++a) {
                float wu = bil1(a + u - (k - 1));
                if (wu == 0.f) continue;
                for (int b = 0; b < ko; ++b) {
                    float wv = bil1(b + v - (k - 1));
                    if (wv == 0.f) continue;
                    s += wu * wv * in[(((long long)a * ko + b) * F + f) * Cin + ci];
                }
            }
            out[i] += s;
        }
    }
}

extern "C" int savp_fold_bilinear(void* stream, const float* in, float* out, int32_t k, int32_t Cin, int32_t F, int32_t adjoint) {
    if (!in || !out || k < 1) return SAVP_EINVAL;
    long long total = (long long)(k + 3) * (k + 3) * Cin * F;
    unsigned nb = (unsigned)((total + NT - 1) / NT);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(fold_bilinear_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, in, out, k, Cin, F, adjoint);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// spectral norm.  W [K, C] (K = prod(kernel dims, Cin), C = Cout), u [C].
//   a = W u ; v = a/(|a|+eps) ; b = W^T v ; u' = b/(|b|+eps) ; sigma = v^T W u' = |b|^2/(|b|+eps)
// workspace ws (floats): [0]=sigma [1]=1/sigma [2]=|a| [3]=|b| [4]=kappa [5]=<G,W> [6]=a.gv [7]=(unused)
//                        [8 .. 8+C) = b ; [8+C .. 8+2C) = u' ; [8+2C .. 8+2C+K) = a ; [.. +K) = gv/ga ;
//                        then, 8-byte aligned, FLOAT64 accumulators: [0] = |a|^2, [1 .. 1+C) = W^T a -- the forward sums that many
//                        workgroups add to atomically.  float64 because a sum of fp32 partials is exact there: sigma (and with it every
//                        weight of the discriminator) does not depend on the workgroups' arrival order.
//                        Total: 8 + 2C + 2K + 2 (C + 2) floats; ws 8-byte aligned.
// ---------------------------------------------------------------------------------------------------------------
#define SN_EPS 1e-12f
__host__ __device__ __forceinline__ double* sn_acc64(float* ws, long long K, int C) { return reinterpret_cast<double*>(ws + 8 + 2 * C + 2 * K); }

// y[k] = sum_c W[k,c] x[c]; one wave per row; optionally accumulates sum_k y[k]^2 into *sq and sum_k y[k]*z[k] into *dotz
__device__ __forceinline__ void sn_rows_body(const float* __restrict__ W, long long K, int C, const float* __restrict__ x,
                                             float xscale, float* __restrict__ y, double* sq, const float* z, double* dotz, int bx, int gx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float sqacc = 0.f, dzacc = 0.f;
    for (long long k = bx * 4LL + wave; k < K; k += (long long)gx * 4) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += W[k * C + c] * x[c];
        s = wsum(s) * xscale;
        if (lane == 0) {
            y[k] = s;
            sqacc += s * s;
            if (z) dzacc += s * z[k];
        }
    }
    if (lane == 0) {
        if (sq) unsafeAtomicAdd(sq, (double)sqacc);
        if (dotz) unsafeAtomicAdd(dotz, (double)dzacc);
    }
}

__global__ __launch_bounds__(NT) void sn_gemv_rows_kernel(const float* __restrict__ W, long long K, int C, const float* __restrict__ x,
                                                          float xscale, float* __restrict__ y, double* sq, const float* z,
                                                          double* dotz) {
    sn_rows_body(W, K, C, x, xscale, y, sq, z, dotz, blockIdx.x, gridDim.x);
}

// Same product for C = 4 * LPR with LPR a power of two <= 64 (every spectrally normalised conv of the model: 32..256 output
// channels): LPR lanes x float4 cover one row, a wave covers 64/LPR consecutive rows per pass = 1 KB of contiguous memory,
// two passes in flight.  (The one-element-per-lane kernel above spends its time in shuffles and exposed load latency.)
__device__ __forceinline__ void sn_rows_vec_body(const float* __restrict__ W, long long K, int C, const float* __restrict__ x,
                                                 float xscale, float* __restrict__ y, double* sq, const float* z, double* dotz, int bx, int gx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lpr = C >> 2, rpw = 64 / lpr;
    const int c4 = lane & (lpr - 1), sub = lane / lpr;
    const float4 xv = *reinterpret_cast<const float4*>(x + c4 * 4);
    float sqacc = 0.f, dzacc = 0.f;
    const long long stride = (long long)gx * 4 * rpw;
    for (long long k0 = ((long long)bx * 4 + wave) * rpw; k0 < K; k0 += 2 * stride) {
        const long long ka = k0 + sub, kb = k0 + stride + sub;
        float4 wa = make_float4(0.f, 0.f, 0.f, 0.f), wb = wa;
        if (ka < K) wa = *reinterpret_cast<const float4*>(W + ka * C + c4 * 4);
        if (kb < K) wb = *reinterpret_cast<const float4*>(W + kb * C + c4 * 4);
        float sa = wa.x * xv.x + wa.y * xv.y + wa.z * xv.z + wa.w * xv.w;
        float sb = wb.x * xv.x + wb.y * xv.y + wb.z * xv.z + wb.w * xv.w;
        for (int o = lpr >> 1; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
        if (c4 == 0) {
            sa *= xscale; sb *= xscale;
            if (ka < K) { y[ka] = sa; sqacc += sa * sa; if (z) dzacc += sa * z[ka]; }
            if (kb < K) { y[kb] = sb; sqacc += sb * sb; if (z) dzacc += sb * z[kb]; }
        }
    }
    // one float64 atomic per WORKGROUP and accumulator: every launch of this body adds into ONE address per matrix, and same-address
    // atomics retire one at a time (measured: the batched launch over a discriminator's eight matrices took 62 us with one atomic per
    // wave of 1024 workgroups -- 20 MB of weights are 5 us of HBM time)
    __shared__ float sn_part[2][4];
    sqacc = wsum(sqacc); dzacc = wsum(dzacc);
    if (lane == 0) { sn_part[0][wave] = sqacc; sn_part[1][wave] = dzacc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (sq) unsafeAtomicAdd(sq, (double)(sn_part[0][0] + sn_part[0][1]) + (double)(sn_part[0][2] + sn_part[0][3]));
        if (dotz) unsafeAtomicAdd(dotz, (double)(sn_part[1][0] + sn_part[1][1]) + (double)(sn_part[1][2] + sn_part[1][3]));
    }
}

__global__ __launch_bounds__(NT) void sn_gemv_rows_vec_kernel(const float* __restrict__ W, long long K, int C, const float* __restrict__ x,
                                                              float xscale, float* __restrict__ y, double* sq, const float* z,
                                                              double* dotz) {
    sn_rows_vec_body(W, K, C, x, xscale, y, sq, z, dotz, blockIdx.x, gridDim.x);
}

static bool sn_vec_ok(const float* W, const float* x, int C) {
    const int lpr = C >> 2;
    return (C % 4 == 0) && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && ((((uintptr_t)W) | ((uintptr_t)x)) & 15) == 0;
}

// narrow matrices (C <= 16, e.g. the discriminators' final linear [65536, 1]): one THREAD per row
__device__ __forceinline__ void sn_rows_narrow_body(const float* __restrict__ W, long long K, int C, const float* __restrict__ x,
                                                    float xscale, float* __restrict__ y, double* sq, const float* z, double* dotz, int bx, int gx,
                                                    float* sh) {
    float sqacc = 0.f, dzacc = 0.f;
    for (long long k = bx * (long long)NT + threadIdx.x; k < K; k += (long long)gx * NT) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += W[k * C + c] * x[c];
        s *= xscale;
        y[k] = s;
        sqacc += s * s;
        if (z) dzacc += s * z[k];
    }
    float t = block_sum1(sqacc, sh);
    if (threadIdx.x == 0 && sq) unsafeAtomicAdd(sq, (double)t);
    if (dotz) {
        float t2 = block_sum1(dzacc, sh);
        if (threadIdx.x == 0) unsafeAtomicAdd(dotz, (double)t2);
    }
}

__global__ __launch_bounds__(NT) void sn_gemv_rows_narrow_kernel(const float* __restrict__ W, long long K, int C, const float* __restrict__ x,
                                                                 float xscale, float* __restrict__ y, double* sq, const float* z,
                                                                 double* dotz) {
    __shared__ float sh[4];
    sn_rows_narrow_body(W, K, C, x, xscale, y, sq, z, dotz, blockIdx.x, gridDim.x, sh);
}

// y[c] += sum_k W[k,c] x[k]   (atomic; y zeroed by the caller)
__device__ __forceinline__ void sn_cols_body(const float* __restrict__ W, long long K, int C, const float* __restrict__ x,
                                             double* __restrict__ y, int rows_per_block, int bx) {
    const long long k0 = (long long)bx * rows_per_block;
    const long long k1 = min(K, k0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += NT) {
        float s = 0.f;
        for (long long k = k0; k < k1; ++k) s += W[k * C + c] * x[k];
        unsafeAtomicAdd(y + c, (double)s);
    }
}

__global__ __launch_bounds__(NT) void sn_gemv_cols_kernel(const float* __restrict__ W, long long K, int C, const float* __restrict__ x,
                                                          double* __restrict__ y, int rows_per_block) {
    sn_cols_body(W, K, C, x, y, rows_per_block, blockIdx.x);
}

// narrow matrices (C <= 16: the discriminators' final linear layer, [65536, 1]): every thread walks rows, the block reduces per column
__device__ __forceinline__ void sn_cols_narrow_body(const float* __restrict__ W, long long K, int C, const float* __restrict__ x,
                                                    double* __restrict__ y, int rows_per_block, int bx, float* sh4) {
    const long long k0 = (long long)bx * rows_per_block;
    const long long k1 = min(K, k0 + rows_per_block);
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    for (long long k = k0 + threadIdx.x; k < k1; k += NT) {
        const float xv = x[k];
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < C) acc[c] += W[k * C + c] * xv;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c < C) {                                       // C is uniform: every thread takes the same branches (barriers inside block_sum1)
            const float t = block_sum1(acc[c], sh4);
            if (threadIdx.x == 0) unsafeAtomicAdd(y + c, (double)t);
        }
    }
}

// vectorised variant (C = 4 * LPR, LPR a power of two <= 64): thread = (column quad, row slot), float4 loads of full rows,
// LDS reduction over the row slots, one atomic per column and block
__device__ __forceinline__ void sn_cols_vec_body(const float* __restrict__ W, long long K, int C, const float* __restrict__ x,
                                                 double* __restrict__ y, int rows_per_block, int bx, float4* sh) {
    const int lpr = C >> 2, slots = NT / lpr;
    const int c4 = threadIdx.x & (lpr - 1), sub = threadIdx.x / lpr;
    const long long k0 = (long long)bx * rows_per_block;
    const long long k1 = min(K, k0 + rows_per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    long long k = k0 + sub;
    for (; k + 3 * slots < k1; k += 4 * slots) {            // four rows in flight per thread (the loop is latency-bound otherwise)
        float4 w[4];
        float xv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { w[j] = *reinterpret_cast<const float4*>(W + (k + j * slots) * C + c4 * 4); xv[j] = x[k + j * slots]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc.x += w[j].x * xv[j]; acc.y += w[j].y * xv[j]; acc.z += w[j].z * xv[j]; acc.w += w[j].w * xv[j]; }
    }
    for (; k < k1; k += slots) {
        const float4 w = *reinterpret_cast<const float4*>(W + k * C + c4 * 4);
        const float xv = x[k];
        acc.x += w.x * xv; acc.y += w.y * xv; acc.z += w.z * xv; acc.w += w.w * xv;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < lpr) {
        float4 t = sh[threadIdx.x];
        for (int r = 1; r < slots; ++r) { const float4 v = sh[r * lpr + threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        unsafeAtomicAdd(y + c4 * 4, (double)t.x); unsafeAtomicAdd(y + c4 * 4 + 1, (double)t.y);
        unsafeAtomicAdd(y + c4 * 4 + 2, (double)t.z); unsafeAtomicAdd(y + c4 * 4 + 3, (double)t.w);
    }
}

__global__ __launch_bounds__(NT) void sn_gemv_cols_vec_kernel(const float* __restrict__ W, long long K, int C, const float* __restrict__ x,
                                                              double* __restrict__ y, int rows_per_block) {
    __shared__ float4 sh[NT];
    sn_cols_vec_body(W, K, C, x, y, rows_per_block, blockIdx.x, sh);
}

// single workgroup: finish the forward scalars. bt = W^T a (unnormalised)
__device__ __forceinline__ void sn_finalize_body(float* ws, long long K, int C, float* u_new, float* sh) {
    const double* acc64 = sn_acc64(ws, K, C);             // [0] = |a|^2, [1 .. 1 + C) = W^T a (exact float64 sums)
    const float na = sqrtf((float)acc64[0]);
    const float s = na + SN_EPS;
    float* b = ws + 8;
    float* up = ws + 8 + C;
    float acc = 0.f;
    for (int c = threadIdx.x; c < C; c += NT) { float v = (float)acc64[1 + c] / s; b[c] = v; acc += v * v; }
    const float nb2 = block_sum1(acc, sh);
    const float nb = sqrtf(nb2);
    for (int c = threadIdx.x; c < C; c += NT) { float v = b[c] / (nb + SN_EPS); up[c] = v; if (u_new) u_new[c] = v; }
    if (threadIdx.x == 0) {
        const float sigma = nb2 / (nb + SN_EPS);
        ws[0] = sigma; ws[1] = 1.f / sigma; ws[2] = na; ws[3] = nb;
        ws[4] = (nb + 2.f * SN_EPS) / ((nb + SN_EPS) * (nb + SN_EPS));   // kappa: dsigma/db = kappa*b
    }
}

__global__ __launch_bounds__(NT) void sn_finalize_kernel(float* ws, long long K, int C, float* u_new) {
    __shared__ float sh[4];
    sn_finalize_body(ws, K, C, u_new, sh);
}

extern "C" int savp_sn_fwd(void* stream, const float* W, int64_t K, int32_t C, const float* u, float* ws, float* u_new) {
    // ws must hold 8 + 2C + 2K + 2 (C + 2) floats, 8-byte aligned.  On return ws[1] = 1/sigma (device scalar for pack_weights), u_new = u_final.
    if (!W || !u || !ws || K < 1 || C < 1 || (((uintptr_t)ws) & 7)) return SAVP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    double* acc64 = sn_acc64(ws, K, C);
    savp_zero_async(acc64, (size_t)(C + 1) * sizeof(double), st);
    float* a = ws + 8 + 2 * C;
    unsigned nb = (unsigned)((K + 3) / 4);
    if (nb > 2048) nb = 2048;
    if (C <= 16) {
        unsigned nbn = (unsigned)((K + NT - 1) / NT);
        if (nbn > 1024) nbn = 1024;
        hipLaunchKernelGGL(sn_gemv_rows_narrow_kernel, dim3(nbn), dim3(NT), 0, st, W, (long long)K, C, u, 1.f, a, acc64,
                           (const float*)nullptr, (double*)nullptr);
    } else if (sn_vec_ok(W, u, C)) {
        const long long passes = (K + (256 / (C >> 2)) * 2 - 1) / ((256 / (C >> 2)) * 2);     // rows per block pass pair
        unsigned nbv = (unsigned)(passes < 1024 ? passes : 1024);
        hipLaunchKernelGGL(sn_gemv_rows_vec_kernel, dim3(nbv), dim3(NT), 0, st, W, (long long)K, C, u, 1.f, a, acc64,
                           (const float*)nullptr, (double*)nullptr);
    } else {
        hipLaunchKernelGGL(sn_gemv_rows_kernel, dim3(nb), dim3(NT), 0, st, W, (long long)K, C, u, 1.f, a, acc64, (const float*)nullptr,
                           (double*)nullptr);
    }
    if (sn_vec_ok(W, W, C)) {
        int rpb = (int)((K + 511) / 512);                    // ~512 blocks, at least one pass of the row slots
        const int slots = NT / (C >> 2);
        if (rpb < 4 * slots) rpb = 4 * slots;
        hipLaunchKernelGGL(sn_gemv_cols_vec_kernel, dim3((unsigned)((K + rpb - 1) / rpb)), dim3(NT), 0, st, W, (long long)K, C,
                           (const float*)a, acc64 + 1, rpb);
    } else {
        int rpb = 64;
        hipLaunchKernelGGL(sn_gemv_cols_kernel, dim3((unsigned)((K + rpb - 1) / rpb)), dim3(NT), 0, st, W, (long long)K, C,
                           (const float*)a, acc64 + 1, rpb);
    }
    hipLaunchKernelGGL(sn_finalize_kernel, dim3(1), dim3(NT), 0, st, ws, (long long)K, C, u_new);
    return LAUNCH_OK();
}

// <G, W> -> the backward's float64 accumulator [0] (round 6: the two dot products of the backward are summed in float64 like the forward's
// sums -- exact, so the discriminator's weight gradients do not depend on the workgroups' arrival order)
__global__ __launch_bounds__(NT) void sn_dot_kernel(const float* __restrict__ G, const float* __restrict__ W, long long n, double* out) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) acc += G[i] * W[i];
    float t = block_sum1(acc, sh);
    if (threadIdx.x == 0) unsafeAtomicAdd(out, (double)t);
}

// dW[k,c] (=|+=) G/sigma + alpha*(kappa * v_k * b_c + ga_k * u_c)
//   v = a/s ; ga = gv/s - a*(a.gv)/(na*s^2) with gv = kappa * (W b) stored in ws (unscaled W b; kappa applied here)
__global__ __launch_bounds__(NT) void sn_bwd_apply_kernel(const float* __restrict__ G, long long K, int C, const float* __restrict__ u,
                                                          const float* __restrict__ ws, float* __restrict__ dW, int beta) {
    const float sigma = ws[0], na = ws[2], kappa = ws[4];
    const float s = na + SN_EPS;
    const double* bacc = sn_acc64(const_cast<float*>(ws), K, C);   // backward accumulators: [0] = <G, W>, [1] = a . (W b)
    const float alpha = -(float)bacc[0] / (sigma * sigma);       // dL/dsigma = -<G,W>/sigma^2
    const float adotgv = (float)bacc[1] * kappa;                 // a . gv
    const float* b = ws + 8;
    const float* a = ws + 8 + 2 * C;
    const float* wb = a + K;                                     // W b
    const long long total = K * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long k = i / C;
        const int c = (int)(i % C);
        const float ak = a[k];
        const float gvk = kappa * wb[k];
        const float gak = gvk / s - (na > 0.f ? ak * adotgv / (na * s * s) : 0.f);
        float v = G[i] / sigma + alpha * (kappa * (ak / s) * b[c] + gak * u[c]);
        dW[i] = beta ? dW[i] + v : v;
    }
}

extern "C" int savp_sn_bwd(void* stream, const float* W, int64_t K, int32_t C, const float* u, float* ws, const float* G, float* dW,
                           int32_t beta) {
    // ws as left by savp_sn_fwd for the same (W, u).  G = dL/dW_bar [K,C]; dW = dL/dW.
    if (!W || !u || !ws || !G || !dW) return SAVP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    double* bacc = sn_acc64(ws, K, C);                           // the forward's accumulators are dead: [0] = <G, W>, [1] = a . (W b)
    savp_zero_async(bacc, 2 * sizeof(double), st);
    long long n = (long long)K * C;
    unsigned nb = (unsigned)((n + NT - 1) / NT);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(sn_dot_kernel, dim3(nb), dim3(NT), 0, st, G, W, n, bacc);
    float* a = ws + 8 + 2 * C;
    unsigned nr = (unsigned)((K + 3) / 4);
    if (nr > 2048) nr = 2048;
    // wb = W b ; ws[6] = a . wb
    if (C <= 16) {
        unsigned nbn = (unsigned)((K + NT - 1) / NT);
        if (nbn > 1024) nbn = 1024;
        hipLaunchKernelGGL(sn_gemv_rows_narrow_kernel, dim3(nbn), dim3(NT), 0, st, W, (long long)K, C, (const float*)(ws + 8), 1.f,
                           a + K, (double*)nullptr, (const float*)a, bacc + 1);
    } else if (sn_vec_ok(W, ws + 8, C)) {
        const long long passes = (K + (256 / (C >> 2)) * 2 - 1) / ((256 / (C >> 2)) * 2);
        unsigned nbv = (unsigned)(passes < 1024 ? passes : 1024);
        hipLaunchKernelGGL(sn_gemv_rows_vec_kernel, dim3(nbv), dim3(NT), 0, st, W, (long long)K, C, (const float*)(ws + 8), 1.f, a + K,
                           (double*)nullptr, (const float*)a, bacc + 1);
    } else {
        hipLaunchKernelGGL(sn_gemv_rows_kernel, dim3(nr), dim3(NT), 0, st, W, (long long)K, C, (const float*)(ws + 8), 1.f, a + K,
                           (double*)nullptr, (const float*)a, bacc + 1);
    }
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(nb), dim3(NT), 0, st, G, (long long)K, C, u, (const float*)ws, dW, beta);
    return LAUNCH_OK();
}


// ---------------------------------------------------------------------------------------------------------------
// Batched spectral norm: the same arithmetic for up to SN_MAXB weight tensors per launch (blockIdx.y = tensor).  One discriminator
// has 8 spectrally normalised layers, each needing memset + 3 launches forward and memset + 3 backward, twice per step for the
// forward: ~190 launches of 5-20 us per train step for tensors of a few MB.  The batch entries run the 8 layers of a discriminator
// in 4 launches each way; a layer's blocks beyond what its size needs exit immediately.
// ---------------------------------------------------------------------------------------------------------------
#define SN_MAXB 16
struct SnB { const float* W; long long K; int C; const float* u; float* ws; float* u_new; const float* G; float* dW; int beta; int vec; };
struct SnBatch { int n; SnB it[SN_MAXB]; };
// Workgroups per matrix of the batched products.  Each workgroup ends in float64 atomics on the matrix's accumulators (one address for
// |W u|^2, C addresses for W^T a): few workgroups with several rows in flight each, not one workgroup per handful of rows.
#ifndef SNB_ROW_BLOCKS
#define SNB_ROW_BLOCKS 256
#endif
#ifndef SNB_COL_BLOCKS
#define SNB_COL_BLOCKS 128
#endif

__global__ __launch_bounds__(NT) void snb_zero_kernel(SnBatch b, int off, int count_plus_c) {
    const SnB& t = b.it[blockIdx.y];
    const int n = count_plus_c < 0 ? -count_plus_c : count_plus_c + t.C;          // negative: fixed count; else count + C floats
    if (count_plus_c < 0) {                                                        // backward: its two float64 accumulators (the forward's are dead)
        double* bacc = sn_acc64(t.ws, t.K, t.C);
        if (threadIdx.x < 2) bacc[threadIdx.x] = 0.0;
        return;
    }
    for (int i = threadIdx.x; i < n; i += NT) t.ws[off + i] = 0.f;
    if (count_plus_c >= 0) {                                                       // forward: the float64 accumulators as well
        double* acc64 = sn_acc64(t.ws, t.K, t.C);
        for (int i = threadIdx.x; i < t.C + 1; i += NT) acc64[i] = 0.0;
    }
}

// phase 0: a = W u (+ |a|^2 -> ws[7]);  phase 1 (backward): wb = W b (+ a . wb -> ws[6])
__global__ __launch_bounds__(NT) void snb_rows_kernel(SnBatch b, int phase) {
    __shared__ float sh[4];
    const SnB& t = b.it[blockIdx.y];
    float* a = t.ws + 8 + 2 * t.C;
    const float* x = phase == 0 ? t.u : (const float*)(t.ws + 8);
    float* y = phase == 0 ? a : a + t.K;
    double* sq = phase == 0 ? sn_acc64(t.ws, t.K, t.C) : nullptr;
    const float* z = phase == 0 ? nullptr : (const float*)a;
    double* dotz = phase == 0 ? nullptr : sn_acc64(t.ws, t.K, t.C) + 1;
    if (t.C <= 16) {
        const int need = (int)min((t.K + NT - 1) / NT, (long long)SNB_ROW_BLOCKS);
        if ((int)blockIdx.x >= need) return;
        sn_rows_narrow_body(t.W, t.K, t.C, x, 1.f, y, sq, z, dotz, blockIdx.x, need, sh);
    } else if (t.vec) {
        const int rp = (256 / (t.C >> 2)) * 2;
        const int need = (int)min((t.K + rp - 1) / rp, (long long)SNB_ROW_BLOCKS);
        if ((int)blockIdx.x >= need) return;
        sn_rows_vec_body(t.W, t.K, t.C, x, 1.f, y, sq, z, dotz, blockIdx.x, need);
    } else {
        const int need = (int)min((t.K + 3) / 4, (long long)SNB_ROW_BLOCKS);
        if ((int)blockIdx.x >= need) return;
        sn_rows_body(t.W, t.K, t.C, x, 1.f, y, sq, z, dotz, blockIdx.x, need);
    }
}

// b = W^T a  (ws + 8, zeroed before)
__global__ __launch_bounds__(NT) void snb_cols_kernel(SnBatch b) {
    __shared__ float4 sh[NT];
    const SnB& t = b.it[blockIdx.y];
    const float* a = t.ws + 8 + 2 * t.C;
    if (t.vec) {
        int rpb = (int)((t.K + SNB_COL_BLOCKS - 1) / SNB_COL_BLOCKS);
        const int slots = NT / (t.C >> 2);
        if (rpb < 4 * slots) rpb = 4 * slots;
        if ((long long)blockIdx.x * rpb >= t.K) return;
        sn_cols_vec_body(t.W, t.K, t.C, a, sn_acc64(t.ws, t.K, t.C) + 1, rpb, blockIdx.x, sh);
    } else if (t.C <= 16) {
        const int rpb = (int)max((long long)NT, (t.K + SNB_COL_BLOCKS - 1) / SNB_COL_BLOCKS);
        if ((long long)blockIdx.x * rpb >= t.K) return;
        sn_cols_narrow_body(t.W, t.K, t.C, a, sn_acc64(t.ws, t.K, t.C) + 1, rpb, blockIdx.x, reinterpret_cast<float*>(sh));
    } else {
        const int rpb = (int)max(64ll, (t.K + SNB_COL_BLOCKS - 1) / SNB_COL_BLOCKS);
        if ((long long)blockIdx.x * rpb >= t.K) return;
        sn_cols_body(t.W, t.K, t.C, a, sn_acc64(t.ws, t.K, t.C) + 1, rpb, blockIdx.x);
    }
}

__global__ __launch_bounds__(NT) void snb_finalize_kernel(SnBatch b) {
    __shared__ float sh[4];
    const SnB& t = b.it[blockIdx.y];
    sn_finalize_body(t.ws, t.K, t.C, t.u_new, sh);
}

__global__ __launch_bounds__(NT) void snb_dot_kernel(SnBatch b) {
    __shared__ float sh[4];
    const SnB& t = b.it[blockIdx.y];
    const long long n = t.K * t.C;
    if ((long long)blockIdx.x * NT >= n) return;
    float acc = 0.f;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) acc += t.G[i] * t.W[i];
    const float s = block_sum1(acc, sh);
    if (threadIdx.x == 0) unsafeAtomicAdd(sn_acc64(t.ws, t.K, t.C), (double)s);
}

__global__ __launch_bounds__(NT) void snb_apply_kernel(SnBatch b) {
    const SnB& t = b.it[blockIdx.y];
    const float* ws = t.ws;
    const long long K = t.K;
    const int C = t.C;
    const long long total = K * C;
    if ((long long)blockIdx.x * NT >= total) return;
    const float sigma = ws[0], na = ws[2], kappa = ws[4];
    const float s = na + SN_EPS;
    const double* bacc = sn_acc64(t.ws, K, C);
    const float alpha = -(float)bacc[0] / (sigma * sigma);
    const float adotgv = (float)bacc[1] * kappa;
    const float* bv = ws + 8;
    const float* a = ws + 8 + 2 * C;
    const float* wb = a + K;
    for (long long i = blockIdx.x * (long long)NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const long long k = i / C;
        const int c = (int)(i % C);
        const float ak = a[k];
        const float gvk = kappa * wb[k];
        const float gak = gvk / s - (na > 0.f ? ak * adotgv / (na * s * s) : 0.f);
        const float v = t.G[i] / sigma + alpha * (kappa * (ak / s) * bv[c] + gak * t.u[c]);
        t.dW[i] = t.beta ? t.dW[i] + v : v;
    }
}

static int snb_fill(SnBatch& b, int32_t n, const SavpSnItem* items, bool bwd) {
    if (!items || n < 1 || n > SN_MAXB) return SAVP_EINVAL;
    b.n = n;
    for (int i = 0; i < n; ++i) {
        const SavpSnItem& s = items[i];
        if (!s.W || !s.u || !s.ws || s.K < 1 || s.C < 1 || (bwd && (!s.G || !s.dW)) || (((uintptr_t)s.ws) & 7)) return SAVP_EINVAL;
        SnB& t = b.it[i];
        t.W = s.W; t.K = s.K; t.C = s.C; t.u = s.u; t.ws = s.ws; t.u_new = s.u_new; t.G = s.G; t.dW = s.dW; t.beta = s.beta;
        t.vec = (s.C > 16 && sn_vec_ok(s.W, s.u, s.C) && sn_vec_ok(s.W, s.ws + 8, s.C)) ? 1 : 0;
    }
    return SAVP_OK;
}

extern "C" int savp_sn_fwd_batch(void* stream, int32_t n, const SavpSnItem* items) {
    SnBatch b;
    int rc = snb_fill(b, n, items, false);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(snb_zero_kernel, dim3(1, n), dim3(NT), 0, st, b, 0, 8);            // ws[0 .. 8 + C)
    hipLaunchKernelGGL(snb_rows_kernel, dim3(SNB_ROW_BLOCKS, n), dim3(NT), 0, st, b, 0);
    hipLaunchKernelGGL(snb_cols_kernel, dim3(SNB_COL_BLOCKS, n), dim3(NT), 0, st, b);
    hipLaunchKernelGGL(snb_finalize_kernel, dim3(1, n), dim3(NT), 0, st, b);
    return LAUNCH_OK();
}

extern "C" int savp_sn_bwd_batch(void* stream, int32_t n, const SavpSnItem* items) {
    SnBatch b;
    int rc = snb_fill(b, n, items, true);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(snb_zero_kernel, dim3(1, n), dim3(NT), 0, st, b, 5, -2);           // ws[5], ws[6]
    hipLaunchKernelGGL(snb_dot_kernel, dim3(SNB_ROW_BLOCKS, n), dim3(NT), 0, st, b);
    hipLaunchKernelGGL(snb_rows_kernel, dim3(SNB_ROW_BLOCKS, n), dim3(NT), 0, st, b, 1);
    hipLaunchKernelGGL(snb_apply_kernel, dim3(2048, n), dim3(NT), 0, st, b);
    return LAUNCH_OK();
}
