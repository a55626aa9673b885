// norm_lstm.hip -- fused instance-norm(+activation) and the fused ConvLSTM gate block, forward and backward.
//
// Replaces, per call, what the reference spends 2 transposes + nn.fused_batch_norm + activation on
// (layers/normalization.py:146-170, savp_model.py:463-464,499-500) and, for the ConvLSTM cell
// (rnn_ops.py:148-165), IN(4F) + split + sigmoid/tanh + Hadamard + IN(F) + tanh*sigmoid -- about 14 TF ops --
// with ONE kernel each way.  One workgroup owns (sample n, 4 consecutive channels) over the whole H*W plane,
// so both per-sample reductions are workgroup-local (no grid sync, no atomics except the per-channel
// gamma/beta gradients which are summed over samples and timesteps).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include "savp_hip.h"
#include "zero_fill.h"
extern thread_local hipEvent_t g_savp_prof_start, g_savp_prof_stop;      // common.hip: savp_prof_arm
#include "opts.h"

#define NT 256

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// sum NV per-thread values over the 256-thread block; result broadcast to all threads. `sh` >= 4*NV floats.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = wave_sum(v[i]);
        if (lane == 0) sh[wave * NV + i] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = sh[i] + sh[NV + i] + sh[2 * NV + i] + sh[3 * NV + i];
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // tanh via exp; exact enough in fp32 (|err| ~ 1e-7) and saturates correctly
    float e = __expf(-2.f * fabsf(x));
    float t = (1.f - e) / (1.f + e);
    return copysignf(t, x);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
#ifndef INORM_U
#define INORM_U 2            // pixel rows per trip (measured in the step: 1 -> 46.39 ms, 2 -> 45.66, 4 -> 45.76, 8 -> 45.94: more rows cost occupancy); of the instance-norm streaming loops (loads of a trip in flight together)
#endif
// four consecutive elements at element index idx of a tensor that holds fp32 or (is16) bf16
__device__ __forceinline__ float4 ld4x(const float* base, long long idx, int is16) {
    if (is16) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u));
    }
    return ld4(base + idx);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// four consecutive elements at element index idx of a tensor that holds fp32 or (is16) bf16 (round to nearest even, as every
// bf16 operand of the convolutions is rounded when it is staged: a bf16 destination feeds ONLY convolutions)
__device__ __forceinline__ void st4x(float* base, long long idx, float4 v, int is16) {
    if (is16) {
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        const bf16x4_t o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        *reinterpret_cast<bf16x4_t*>(reinterpret_cast<unsigned short*>(base) + idx) = o;
    } else {
        st4(base + idx, v);
    }
}

__device__ __forceinline__ float act_fwd(float v, int act, float alpha) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return fmaxf(v, alpha * v);
    return v;
}
__device__ __forceinline__ float act_grad_from_out(float y, int act, float alpha) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 2) return y > 0.f ? 1.f : alpha;
    return 1.f;
}

// ------------------------------------------------------------------------------------------------------------
// instance norm + activation, forward.  grid = N * C/4
// ------------------------------------------------------------------------------------------------------------
struct InormP {
    int N, HW, C;
    int chunk;                      // pixels per workgroup on the coalesced (two-kernel) path
    const float* x; long long x_sn, x_sp;
    const float* gamma; const float* beta;
    float eps; int act; float alpha;
    int nout; float* out[4]; long long o_sn[4], o_sp[4];
    int o_c0[4], o_c1[4];           // output k receives channels [o_c0, o_c1) of x (multiples of 4; default: all C)
    int o16[4];                     // output k is a bf16 tensor (strides in bf16 elements)
    float* mean; float* rstd;       // [N, C]
    // backward
    int ndy; const float* dy[4]; long long dy_sn[4], dy_sp[4];
    int dy_c0[4], dy_c1[4];         // gradient k covers channels [dy_c0, dy_c1) of the output (default: all C)
    float* dx; long long dx_sn, dx_sp; int dx_beta;
    int dx16;                       // dx is a bf16 tensor (strides in bf16 elements; no dx_beta)
    double* dgamma; double* dbeta;     // float64 accumulators (savp_hip.h SavpInormArgs): sums of fp32 partials are exact there
};

__global__ __launch_bounds__(NT) void inorm_fwd_kernel(InormP p) {
    __shared__ float sh[4 * 4];
    const int cg = p.C / 4;
    const int n = blockIdx.x / cg, c0 = (blockIdx.x % cg) * 4;
    const float* x = p.x + (long long)n * p.x_sn + c0;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int px = threadIdx.x; px < p.HW; px += NT) {
        float4 v = ld4(x + (long long)px * p.x_sp);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
    }
    block_sum<4>(s, sh);
    const float inv = 1.f / (float)p.HW;
    const float m0 = s[0] * inv, m1 = s[1] * inv, m2 = s[2] * inv, m3 = s[3] * inv;
    float q[4] = {0.f, 0.f, 0.f, 0.f};
    for (int px = threadIdx.x; px < p.HW; px += NT) {
        float4 v = ld4(x + (long long)px * p.x_sp);
        q[0] += (v.x - m0) * (v.x - m0); q[1] += (v.y - m1) * (v.y - m1);
        q[2] += (v.z - m2) * (v.z - m2); q[3] += (v.w - m3) * (v.w - m3);
    }
    block_sum<4>(q, sh);
    const float r0 = rsqrtf(q[0] * inv + p.eps), r1 = rsqrtf(q[1] * inv + p.eps), r2 = rsqrtf(q[2] * inv + p.eps),
                r3 = rsqrtf(q[3] * inv + p.eps);
    if (threadIdx.x == 0) {
        float* mp = p.mean + (long long)n * p.C + c0; float* rp = p.rstd + (long long)n * p.C + c0;
        mp[0] = m0; mp[1] = m1; mp[2] = m2; mp[3] = m3;
        rp[0] = r0; rp[1] = r1; rp[2] = r2; rp[3] = r3;
    }
    const float4 g = ld4(p.gamma + c0), b = ld4(p.beta + c0);
    for (int px = threadIdx.x; px < p.HW; px += NT) {
        float4 v = ld4(x + (long long)px * p.x_sp);
        float4 o;
        o.x = act_fwd((v.x - m0) * r0 * g.x + b.x, p.act, p.alpha);
        o.y = act_fwd((v.y - m1) * r1 * g.y + b.y, p.act, p.alpha);
        o.z = act_fwd((v.z - m2) * r2 * g.z + b.z, p.act, p.alpha);
        o.w = act_fwd((v.w - m3) * r3 * g.w + b.w, p.act, p.alpha);
        for (int k = 0; k < p.nout; ++k)
            if (c0 >= p.o_c0[k] && c0 < p.o_c1[k])
                st4x(p.out[k], (long long)n * p.o_sn[k] + (long long)px * p.o_sp[k] + (c0 - p.o_c0[k]), o, p.o16[k]);
    }
}

__global__ __launch_bounds__(NT) void inorm_bwd_kernel(InormP p) {
    __shared__ float sh[4 * 8];
    const int cg = p.C / 4;
    const int n = blockIdx.x / cg, c0 = (blockIdx.x % cg) * 4;
    const float* x = p.x + (long long)n * p.x_sn + c0;
    const float4 m = ld4(p.mean + (long long)n * p.C + c0), r = ld4(p.rstd + (long long)n * p.C + c0);
    const float4 g = ld4(p.gamma + c0), bt = ld4(p.beta + c0);
    // the activation mask is recomputed from the pre-activation z = xh * gamma + beta (y > 0 <=> z > 0 for relu / lrelu): the saved
    // activation output is not read back
    auto load_dz = [&](int px, float4& xh) -> float4 {
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < p.ndy; ++k) {
            if (c0 < p.dy_c0[k] || c0 >= p.dy_c1[k]) continue;
            float4 t = ld4(p.dy[k] + (long long)n * p.dy_sn[k] + (long long)px * p.dy_sp[k] + (c0 - p.dy_c0[k]));
            d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
        }
        float4 v = ld4(x + (long long)px * p.x_sp);
        xh.x = (v.x - m.x) * r.x; xh.y = (v.y - m.y) * r.y; xh.z = (v.z - m.z) * r.z; xh.w = (v.w - m.w) * r.w;
        d.x *= act_grad_from_out((v.x - m.x) * r.x * g.x + bt.x, p.act, p.alpha); d.y *= act_grad_from_out((v.y - m.y) * r.y * g.y + bt.y, p.act, p.alpha);
        d.z *= act_grad_from_out((v.z - m.z) * r.z * g.z + bt.z, p.act, p.alpha); d.w *= act_grad_from_out((v.w - m.w) * r.w * g.w + bt.w, p.act, p.alpha);
        return d;
    };
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int px = threadIdx.x; px < p.HW; px += NT) {
        float4 xh; float4 d = load_dz(px, xh);
        s[0] += d.x; s[1] += d.y; s[2] += d.z; s[3] += d.w;
        s[4] += d.x * xh.x; s[5] += d.y * xh.y; s[6] += d.z * xh.z; s[7] += d.w * xh.w;
    }
    block_sum<8>(s, sh);
    if (threadIdx.x < 4) {
        unsafeAtomicAdd(p.dbeta + c0 + threadIdx.x, s[threadIdx.x]);
        unsafeAtomicAdd(p.dgamma + c0 + threadIdx.x, s[4 + threadIdx.x]);
    }
    const float inv = 1.f / (float)p.HW;
    float* dx = p.dx + (p.dx16 ? 0 : (long long)n * p.dx_sn + c0);
    for (int px = threadIdx.x; px < p.HW; px += NT) {
        float4 xh; float4 d = load_dz(px, xh);
        float4 o;
        o.x = g.x * r.x * (d.x - s[0] * inv - xh.x * s[4] * inv);
        o.y = g.y * r.y * (d.y - s[1] * inv - xh.y * s[5] * inv);
        o.z = g.z * r.z * (d.z - s[2] * inv - xh.z * s[6] * inv);
        o.w = g.w * r.w * (d.w - s[3] * inv - xh.w * s[7] * inv);
        if (p.dx16) { st4x(dx, (long long)n * p.dx_sn + (long long)px * p.dx_sp + c0, o, 1); continue; }
        float* q = dx + (long long)px * p.dx_sp;
        if (p.dx_beta) { float4 t = ld4(q); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
        st4(q, o);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Planes of H*W >= 256 pixels (SAVP_INORM_MIN_HW overrides; measured: 1.3 ms/step faster than the one-workgroup-per-
// (n, 4 channels) kernels down to 16x16): statistics by a fully coalesced reduction
// with per-(n,c) atomics, then an elementwise apply pass.  Every lane reads 16 B of a full pixel row, so HBM/L2
// lines are used completely (the one-workgroup-per-(n,4ch) kernels above use 16 B of each 128-B line).
// Sums are taken around the first pixel's value (shifted variance) to keep E[x^2]-E[x]^2 well conditioned.
//   ws layout per call: [N][C][2] floats, zeroed by the launcher.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void inorm_stats_kernel(InormP p, double* ws) {
    extern __shared__ float sh[];                 // [rows][2*C]
    const int n = blockIdx.y, C = p.C, C4 = C / 4;
    const int c4 = threadIdx.x % C4, prow = threadIdx.x / C4, rows = NT / C4;
    const float* x = p.x + (long long)n * p.x_sn;
    const float4 k = ld4(x + c4 * 4);             // shift = pixel 0
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    const int p0 = blockIdx.x * p.chunk, p1 = min(p.HW, p0 + p.chunk);
    if (prow < rows)
        for (int px = p0 + prow; px < p1; px += rows) {
            float4 v = ld4(x + (long long)px * p.x_sp + c4 * 4);
            v.x -= k.x; v.y -= k.y; v.z -= k.z; v.w -= k.w;
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
        }
    if (prow < rows) {
        float* d = sh + prow * 2 * C + c4 * 4;
        d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w;
        d[C] = q.x; d[C + 1] = q.y; d[C + 2] = q.z; d[C + 3] = q.w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += NT) {
        float t = 0.f;
        for (int r = 0; r < rows; ++r) t += sh[r * 2 * C + i];
        const int c = i % C, which = i / C;
        // float64 accumulator: a sum of fp32 partials is exact there, so the result does not depend on the workgroups' arrival order
        unsafeAtomicAdd(ws + ((long long)n * C + c) * 2 + which, (double)t);
    }
}

__global__ __launch_bounds__(NT) void inorm_apply_kernel(InormP p, const double* ws, int unshifted, const float* shift) {
    const int n = blockIdx.y, C = p.C, C4 = C / 4;
    const int c4 = threadIdx.x % C4, prow = threadIdx.x / C4, rows = NT / C4;
    if (prow >= rows) return;
    const float* x = p.x + (long long)n * p.x_sn;
    // unshifted: the sums came out of the producing convolution's epilogue (SavpInormArgs.stats_ready), taken around that convolution's
    // bias (`shift`, per channel; NULL = around 0); otherwise around the sample's first pixel
    const float4 k = unshifted ? (shift ? ld4(shift + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f)) : ld4(x + c4 * 4);
    const float inv = 1.f / (float)p.HW;
    float m[4], r[4];
    const float kk[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double* w = ws + ((long long)n * C + c4 * 4 + e) * 2;
        const double ms = w[0] * (double)inv;
        const float var = fmaxf((float)(w[1] * (double)inv - ms * ms), 0.f);
        m[e] = kk[e] + (float)ms; r[e] = rsqrtf(var + p.eps);
    }
    if (blockIdx.x == 0 && prow == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { p.mean[(long long)n * C + c4 * 4 + e] = m[e]; p.rstd[(long long)n * C + c4 * 4 + e] = r[e]; }
    }
    const float4 g = ld4(p.gamma + c4 * 4), b = ld4(p.beta + c4 * 4);
    const int p0 = blockIdx.x * p.chunk, p1 = min(p.HW, p0 + p.chunk);
    // INORM_U pixel rows per trip, every load of the trip issued before the first store (the destinations may alias the source as far as the
    // compiler knows, so a one-pixel loop exposes a full memory round trip per pixel)
    for (int px0 = p0 + prow; px0 < p1; px0 += INORM_U * rows) {
        float4 v[INORM_U];
#pragma unroll
        for (int j = 0; j < INORM_U; ++j) {
            const int px = px0 + j * rows;
            v[j] = ld4(x + (long long)(px < p1 ? px : px0) * p.x_sp + c4 * 4);
        }
#pragma unroll
        for (int j = 0; j < INORM_U; ++j) {
            const int px = px0 + j * rows;
            if (px >= p1) break;
            float4 o;
            o.x = act_fwd((v[j].x - m[0]) * r[0] * g.x + b.x, p.act, p.alpha);
            o.y = act_fwd((v[j].y - m[1]) * r[1] * g.y + b.y, p.act, p.alpha);
            o.z = act_fwd((v[j].z - m[2]) * r[2] * g.z + b.z, p.act, p.alpha);
            o.w = act_fwd((v[j].w - m[3]) * r[3] * g.w + b.w, p.act, p.alpha);
            for (int kq = 0; kq < p.nout; ++kq)
                if (c4 * 4 >= p.o_c0[kq] && c4 * 4 < p.o_c1[kq])
                    st4x(p.out[kq], (long long)n * p.o_sn[kq] + (long long)px * p.o_sp[kq] + (c4 * 4 - p.o_c0[kq]), o, p.o16[kq]);
        }
    }
}

// dz = dL/d(normalised pre-activation) of INORM_U pixel rows px0, px0 + rows, ... of this thread's four channels (rows past p1 re-read px0 and
// are ignored by the caller): the sum of the gradient views that cover the channels, times the activation's derivative (the mask is recomputed from
// the pre-activation: y > 0 <=> z > 0), and xhat.  All loads of the trip are issued before anything is computed.
__device__ __forceinline__ void inorm_dz_batch(const InormP& p, int n, int px0, int rows, int p1, int c0, const float* x, const float m[4],
                                               const float r[4], const float4 g, const float4 bt, float4 (&d)[INORM_U], float4 (&xh)[INORM_U]) {
    float4 v[INORM_U];
    int pxs[INORM_U];
#pragma unroll
    for (int j = 0; j < INORM_U; ++j) {
        pxs[j] = px0 + j * rows < p1 ? px0 + j * rows : px0;
        v[j] = ld4(x + (long long)pxs[j] * p.x_sp + c0);
        d[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int k = 0; k < p.ndy; ++k) {
        if (c0 < p.dy_c0[k] || c0 >= p.dy_c1[k]) continue;
        const float* base = p.dy[k] + (long long)n * p.dy_sn[k] + (c0 - p.dy_c0[k]);
        float4 t[INORM_U];
#pragma unroll
        for (int j = 0; j < INORM_U; ++j) t[j] = ld4(base + (long long)pxs[j] * p.dy_sp[k]);
#pragma unroll
        for (int j = 0; j < INORM_U; ++j) { d[j].x += t[j].x; d[j].y += t[j].y; d[j].z += t[j].z; d[j].w += t[j].w; }
    }
#pragma unroll
    for (int j = 0; j < INORM_U; ++j) {
        xh[j].x = (v[j].x - m[0]) * r[0]; xh[j].y = (v[j].y - m[1]) * r[1]; xh[j].z = (v[j].z - m[2]) * r[2]; xh[j].w = (v[j].w - m[3]) * r[3];
        d[j].x *= act_grad_from_out((v[j].x - m[0]) * r[0] * g.x + bt.x, p.act, p.alpha);
        d[j].y *= act_grad_from_out((v[j].y - m[1]) * r[1] * g.y + bt.y, p.act, p.alpha);
        d[j].z *= act_grad_from_out((v[j].z - m[2]) * r[2] * g.z + bt.z, p.act, p.alpha);
        d[j].w *= act_grad_from_out((v[j].w - m[3]) * r[3] * g.w + bt.w, p.act, p.alpha);
    }
}

__global__ __launch_bounds__(NT) void inorm_bwd_stats_kernel(InormP p, double* ws) {
    extern __shared__ float sh[];
    const int n = blockIdx.y, C = p.C, C4 = C / 4;
    const int c4 = threadIdx.x % C4, prow = threadIdx.x / C4, rows = NT / C4;
    const float* x = p.x + (long long)n * p.x_sn;
    float m[4], r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { m[e] = p.mean[(long long)n * C + c4 * 4 + e]; r[e] = p.rstd[(long long)n * C + c4 * 4 + e]; }
    const float4 gm = ld4(p.gamma + c4 * 4), bt = ld4(p.beta + c4 * 4);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    const int p0 = blockIdx.x * p.chunk, p1 = min(p.HW, p0 + p.chunk);
    if (prow < rows)
        for (int px0 = p0 + prow; px0 < p1; px0 += INORM_U * rows) {
            float4 xh[INORM_U], d[INORM_U];
            inorm_dz_batch(p, n, px0, rows, p1, c4 * 4, x, m, r, gm, bt, d, xh);
#pragma unroll
            for (int j = 0; j < INORM_U; ++j) {           // same per-thread order of additions as a one-pixel loop
                if (px0 + j * rows >= p1) break;
                s.x += d[j].x; s.y += d[j].y; s.z += d[j].z; s.w += d[j].w;
                q.x += d[j].x * xh[j].x; q.y += d[j].y * xh[j].y; q.z += d[j].z * xh[j].z; q.w += d[j].w * xh[j].w;
            }
        }
    if (prow < rows) {
        float* d = sh + prow * 2 * C + c4 * 4;
        d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w;
        d[C] = q.x; d[C + 1] = q.y; d[C + 2] = q.z; d[C + 3] = q.w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += NT) {
        float t = 0.f;
        for (int rr = 0; rr < rows; ++rr) t += sh[rr * 2 * C + i];
        const int c = i % C, which = i / C;
        unsafeAtomicAdd(ws + ((long long)n * C + c) * 2 + which, (double)t);
    }
}

__global__ __launch_bounds__(NT) void inorm_bwd_apply_kernel(InormP p, const double* ws) {
    const int n = blockIdx.y, C = p.C, C4 = C / 4;
    const int c4 = threadIdx.x % C4, prow = threadIdx.x / C4, rows = NT / C4;
    if (prow >= rows) return;
    const float* x = p.x + (long long)n * p.x_sn;
    float m[4], r[4], s1[4], s2[4];
    const float inv = 1.f / (float)p.HW;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        m[e] = p.mean[(long long)n * C + c4 * 4 + e]; r[e] = p.rstd[(long long)n * C + c4 * 4 + e];
        s1[e] = (float)ws[((long long)n * C + c4 * 4 + e) * 2]; s2[e] = (float)ws[((long long)n * C + c4 * 4 + e) * 2 + 1];
    }
    // dbeta / dgamma = the per-sample sums added up over samples: one atomic per (sample, channel) here instead of one per
    // (workgroup, channel) in the statistics pass (512 workgroups hammering C addresses made that pass latency-bound)
    if (blockIdx.x == 0 && prow == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { unsafeAtomicAdd(p.dbeta + c4 * 4 + e, s1[e]); unsafeAtomicAdd(p.dgamma + c4 * 4 + e, s2[e]); }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[e] *= inv; s2[e] *= inv; }
    const float4 g = ld4(p.gamma + c4 * 4), bt = ld4(p.beta + c4 * 4);
    float* dx = p.dx + (p.dx16 ? 0 : (long long)n * p.dx_sn + c4 * 4);
    const int p0 = blockIdx.x * p.chunk, p1 = min(p.HW, p0 + p.chunk);
    for (int px0 = p0 + prow; px0 < p1; px0 += INORM_U * rows) {
        float4 xh[INORM_U], d[INORM_U];
        inorm_dz_batch(p, n, px0, rows, p1, c4 * 4, x, m, r, g, bt, d, xh);
#pragma unroll
        for (int j = 0; j < INORM_U; ++j) {
            const int px = px0 + j * rows;
            if (px >= p1) break;
            float4 o;
            o.x = g.x * r[0] * (d[j].x - s1[0] - xh[j].x * s2[0]);
            o.y = g.y * r[1] * (d[j].y - s1[1] - xh[j].y * s2[1]);
            o.z = g.z * r[2] * (d[j].z - s1[2] - xh[j].z * s2[2]);
            o.w = g.w * r[3] * (d[j].w - s1[3] - xh[j].w * s2[3]);
            if (p.dx16) { st4x(dx, (long long)n * p.dx_sn + (long long)px * p.dx_sp + c4 * 4, o, 1); continue; }
            float* qq = dx + (long long)px * p.dx_sp;
            if (p.dx_beta) { float4 t = ld4(qq); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
            st4(qq, o);
        }
    }
}

static int inorm_min_hw() {
    return savp_opt(OPT_INORM_MIN_HW);
}
static bool use_large_plane_path(const SavpInormArgs* a) { return a->ws && a->HW >= inorm_min_hw() && a->C % 4 == 0 && a->C <= 256 && (NT % (a->C / 4) == 0); }
// pixels per workgroup: ~512 workgroups per launch, at least one pass of the block's pixel rows, at most 256
static int inorm_chunk(const SavpInormArgs* a) {
    const int rows = NT / (a->C / 4);
    long long c = ((long long)a->HW * a->N + 511) / 512;
    if (c < rows) c = rows;
    if (c > 256) c = 256;
    return (int)c;
}

extern "C" int savp_instnorm_act_fwd(void* stream, const SavpInormArgs* a) {
    if (!a || a->C % 4 || a->nout < 1 || a->nout > 4) return SAVP_EINVAL;
    InormP p;
    p.N = a->N; p.HW = a->HW; p.C = a->C;
    p.x = (const float*)a->x.p; p.x_sn = a->x.sn; p.x_sp = a->x.sp;
    p.gamma = a->gamma; p.beta = a->beta; p.eps = a->eps; p.act = a->act; p.alpha = a->alpha;
    p.nout = a->nout;
    for (int i = 0; i < a->nout; ++i) {
        p.out[i] = (float*)a->out[i].p; p.o_sn[i] = a->out[i].sn; p.o_sp[i] = a->out[i].sp;
        p.o_c0[i] = a->out_c0[i]; p.o_c1[i] = a->out_nc[i] > 0 ? a->out_c0[i] + a->out_nc[i] : a->C;
        p.o16[i] = (a->out_bf16 >> i) & 1;
        if ((p.o_c0[i] & 3) || (p.o_c1[i] & 3) || p.o_c0[i] < 0 || p.o_c1[i] > a->C) return SAVP_EINVAL;
    }
    p.mean = a->mean; p.rstd = a->rstd;
    if (a->stats_ready) {                      // statistics from the producing convolution: the apply pass alone, any plane size
        if (!a->ws || a->C > 256 || (NT % (a->C / 4)) != 0) return SAVP_EINVAL;
        hipStream_t st = (hipStream_t)stream;
        p.chunk = inorm_chunk(a);
        dim3 grid((a->HW + p.chunk - 1) / p.chunk, a->N);
        hipLaunchKernelGGL(inorm_apply_kernel, grid, dim3(NT), 0, st, p, (const double*)a->ws, 1, a->stats_shift);
        return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
    }
    if (use_large_plane_path(a)) {
        hipStream_t st = (hipStream_t)stream;
        if (!a->ws_clean) savp_zero_async(a->ws, (size_t)a->N * a->C * 2 * sizeof(double), st);
        p.chunk = inorm_chunk(a);
        dim3 grid((a->HW + p.chunk - 1) / p.chunk, a->N);
        size_t lds = (size_t)(NT / (a->C / 4)) * 2 * a->C * sizeof(float);
        hipLaunchKernelGGL(inorm_stats_kernel, grid, dim3(NT), lds, st, p, (double*)a->ws);
        hipLaunchKernelGGL(inorm_apply_kernel, grid, dim3(NT), 0, st, p, (const double*)a->ws, 0, (const float*)nullptr);
        return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
    }
    hipLaunchKernelGGL(inorm_fwd_kernel, dim3(a->N * (a->C / 4)), dim3(NT), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
}

extern "C" int savp_instnorm_act_bwd(void* stream, const SavpInormArgs* a) {
    if (!a || a->C % 4 || a->ndy < 1 || a->ndy > 4) return SAVP_EINVAL;
    InormP p;
    p.N = a->N; p.HW = a->HW; p.C = a->C;
    p.x = (const float*)a->x.p; p.x_sn = a->x.sn; p.x_sp = a->x.sp;
    p.gamma = a->gamma; p.beta = a->beta; p.eps = a->eps; p.act = a->act; p.alpha = a->alpha;
    p.mean = a->mean; p.rstd = a->rstd;
    p.ndy = a->ndy;
    for (int i = 0; i < a->ndy; ++i) {
        p.dy[i] = (const float*)a->dy[i].p; p.dy_sn[i] = a->dy[i].sn; p.dy_sp[i] = a->dy[i].sp;
        p.dy_c0[i] = a->dy_c0[i]; p.dy_c1[i] = a->dy_nc[i] > 0 ? a->dy_c0[i] + a->dy_nc[i] : a->C;
        if ((p.dy_c0[i] & 3) || (p.dy_c1[i] & 3) || p.dy_c0[i] < 0 || p.dy_c1[i] > a->C) return SAVP_EINVAL;
    }
    p.dx = (float*)a->dx.p; p.dx_sn = a->dx.sn; p.dx_sp = a->dx.sp; p.dx_beta = a->dx_beta;
    p.dx16 = a->dx_bf16 ? 1 : 0;
    if (p.dx16 && (a->dx_beta || (a->dx.sn & 3) || (a->dx.sp & 3) || (((uintptr_t)a->dx.p) & 7))) return SAVP_EINVAL;
    p.dgamma = a->dgamma; p.dbeta = a->dbeta;
    if (a->stats_ready) {                      // sum(dy'), sum(dy' * xhat) from the convolution that produced dy: the apply pass alone
        if (!a->ws || a->C > 256 || (NT % (a->C / 4)) != 0) return SAVP_EINVAL;
        hipStream_t st = (hipStream_t)stream;
        p.chunk = inorm_chunk(a);
        dim3 grid((a->HW + p.chunk - 1) / p.chunk, a->N);
        hipLaunchKernelGGL(inorm_bwd_apply_kernel, grid, dim3(NT), 0, st, p, (const double*)a->ws);
        return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
    }
    if (use_large_plane_path(a)) {
        hipStream_t st = (hipStream_t)stream;
        if (!a->ws_clean) savp_zero_async(a->ws, (size_t)a->N * a->C * 2 * sizeof(double), st);
        p.chunk = inorm_chunk(a);
        dim3 grid((a->HW + p.chunk - 1) / p.chunk, a->N);
        size_t lds = (size_t)(NT / (a->C / 4)) * 2 * a->C * sizeof(float);
        hipLaunchKernelGGL(inorm_bwd_stats_kernel, grid, dim3(NT), lds, st, p, (double*)a->ws);
        hipLaunchKernelGGL(inorm_bwd_apply_kernel, grid, dim3(NT), 0, st, p, (const double*)a->ws);
        return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
    }
    hipLaunchKernelGGL(inorm_bwd_kernel, dim3(a->N * (a->C / 4)), dim3(NT), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
}

// ------------------------------------------------------------------------------------------------------------
// fused ConvLSTM gate block.  gates_pre [N, HW, 4F] contiguous (order i, j, f, o as in rnn_ops.py:150).
// One workgroup = (n, 4 channels); HW <= 1024 (every ConvLSTM plane of the reference is <= 32x32).
// ------------------------------------------------------------------------------------------------------------
#define MAXPPT 4
struct LstmP {
    int N, HW, F;
    const float* gates;                                   // fp32, or bf16 when gates16 (coalesced kernels only)
    int gates16, stats1_ready;                            // stats1_ready: ws s1 holds UNSHIFTED sums (conv epilogue), pass 1 is skipped
    const float* c_prev; long long cp_sn, cp_sp;          // may be null (zero state)
    const float *g1, *b1, *g2, *b2;
    float eps, forget_bias;
    float* c_new;                                         // [N, HW, F] contiguous
    int nh; float* h[4]; long long h_sn[4], h_sp[4];
    int h16[4];                                           // destination k of h' is a bf16 tensor (strides in bf16 elements)
    float *mean1, *rstd1, *mean2, *rstd2;                 // [N,4F], [N,F]
    // backward
    int ndh; const float* dh[4]; long long dh_sn[4], dh_sp[4];
    const float* dc_new;                                  // [N,HW,F] contiguous or null
    float* dgates;                                        // [N,HW,4F]; bf16 when dgates16 (coalesced kernels only)
    float* draw;                                          // fp32 [N,HW,4F] scratch of the raw gate gradients between the passes
    int dgates16;                                         //   (= dgates itself unless dgates16)
    float* dc_prev;                                       // [N,HW,F] contiguous or null
    double *dg1, *db1, *dg2, *db2;     // float64 accumulators (savp_hip.h SavpLstmArgs)
};

__global__ __launch_bounds__(NT) void lstm_fwd_kernel(LstmP p) {
    __shared__ float sh[4 * 16];
    const int cg = p.F / 4;
    const int n = blockIdx.x / cg, c0 = (blockIdx.x % cg) * 4;
    const int F = p.F;
    const float* gp = p.gates + (long long)n * p.HW * 4 * F + c0;
    float4 gi[MAXPPT], gj[MAXPPT], gf[MAXPPT], go[MAXPPT];
    float s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            const float* q = gp + (long long)px * 4 * F;
            gi[t] = ld4(q); gj[t] = ld4(q + F); gf[t] = ld4(q + 2 * F); go[t] = ld4(q + 3 * F);
            s[0] += gi[t].x; s[1] += gi[t].y; s[2] += gi[t].z; s[3] += gi[t].w;
            s[4] += gj[t].x; s[5] += gj[t].y; s[6] += gj[t].z; s[7] += gj[t].w;
            s[8] += gf[t].x; s[9] += gf[t].y; s[10] += gf[t].z; s[11] += gf[t].w;
            s[12] += go[t].x; s[13] += go[t].y; s[14] += go[t].z; s[15] += go[t].w;
        }
    }
    block_sum<16>(s, sh);
    const float inv = 1.f / (float)p.HW;
    float mu[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mu[i] = s[i] * inv;
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
#define SQ(a, m) (((a) - (m)) * ((a) - (m)))
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            s[0] += SQ(gi[t].x, mu[0]); s[1] += SQ(gi[t].y, mu[1]); s[2] += SQ(gi[t].z, mu[2]); s[3] += SQ(gi[t].w, mu[3]);
            s[4] += SQ(gj[t].x, mu[4]); s[5] += SQ(gj[t].y, mu[5]); s[6] += SQ(gj[t].z, mu[6]); s[7] += SQ(gj[t].w, mu[7]);
            s[8] += SQ(gf[t].x, mu[8]); s[9] += SQ(gf[t].y, mu[9]); s[10] += SQ(gf[t].z, mu[10]); s[11] += SQ(gf[t].w, mu[11]);
            s[12] += SQ(go[t].x, mu[12]); s[13] += SQ(go[t].y, mu[13]); s[14] += SQ(go[t].z, mu[14]); s[15] += SQ(go[t].w, mu[15]);
        }
    }
    block_sum<16>(s, sh);
    float rs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) rs[i] = rsqrtf(s[i] * inv + p.eps);
    if (threadIdx.x < 16) {
        const int q = threadIdx.x >> 2, c = threadIdx.x & 3;
        p.mean1[(long long)n * 4 * F + q * F + c0 + c] = mu[threadIdx.x];
        p.rstd1[(long long)n * 4 * F + q * F + c0 + c] = rs[threadIdx.x];
    }
    float ga[16], be[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) { ga[q * 4 + c] = p.g1[q * F + c0 + c]; be[q * 4 + c] = p.b1[q * F + c0 + c]; }

    // normalised gates -> c_pre ; keep sigmoid(o) and c_pre in registers
    float cpre[MAXPPT][4], so[MAXPPT][4];
    float s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.c_prev) cp = ld4(p.c_prev + (long long)n * p.cp_sn + (long long)px * p.cp_sp + c0);
            const float iv[4] = {gi[t].x, gi[t].y, gi[t].z, gi[t].w}, jv[4] = {gj[t].x, gj[t].y, gj[t].z, gj[t].w};
            const float fv[4] = {gf[t].x, gf[t].y, gf[t].z, gf[t].w}, ov[4] = {go[t].x, go[t].y, go[t].z, go[t].w};
            const float cpv[4] = {cp.x, cp.y, cp.z, cp.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float in_ = (iv[c] - mu[c]) * rs[c] * ga[c] + be[c];
                float jn = (jv[c] - mu[4 + c]) * rs[4 + c] * ga[4 + c] + be[4 + c];
                float fn = (fv[c] - mu[8 + c]) * rs[8 + c] * ga[8 + c] + be[8 + c];
                float on = (ov[c] - mu[12 + c]) * rs[12 + c] * ga[12 + c] + be[12 + c];
                float cn = cpv[c] * sigmoidf_(fn + p.forget_bias) + sigmoidf_(in_) * tanhf_(jn);
                cpre[t][c] = cn; so[t][c] = sigmoidf_(on);
                s2[c] += cn;
            }
        }
    }
    block_sum<4>(s2, sh);
    float mu2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { mu2[c] = s2[c] * inv; s2[c] = 0.f; }
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s2[c] += SQ(cpre[t][c], mu2[c]);
        }
    }
    block_sum<4>(s2, sh);
    float rs2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) rs2[c] = rsqrtf(s2[c] * inv + p.eps);
    if (threadIdx.x < 4) {
        p.mean2[(long long)n * F + c0 + threadIdx.x] = mu2[threadIdx.x];
        p.rstd2[(long long)n * F + c0 + threadIdx.x] = rs2[threadIdx.x];
    }
    const float4 g2 = ld4(p.g2 + c0), b2 = ld4(p.b2 + c0);
    const float g2v[4] = {g2.x, g2.y, g2.z, g2.w}, b2v[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            float cn[4], hv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                cn[c] = (cpre[t][c] - mu2[c]) * rs2[c] * g2v[c] + b2v[c];
                hv[c] = tanhf_(cn[c]) * so[t][c];
            }
            st4(p.c_new + ((long long)n * p.HW + px) * F + c0, make_float4(cn[0], cn[1], cn[2], cn[3]));
            const float4 h4 = make_float4(hv[0], hv[1], hv[2], hv[3]);
            for (int k = 0; k < p.nh; ++k) st4x(p.h[k], (long long)n * p.h_sn[k] + (long long)px * p.h_sp[k] + c0, h4, p.h16[k]);
        }
    }
}

__global__ __launch_bounds__(NT) void lstm_bwd_kernel(LstmP p) {
    __shared__ float sh[4 * 32];
    const int cg = p.F / 4;
    const int n = blockIdx.x / cg, c0 = (blockIdx.x % cg) * 4;
    const int F = p.F;
    const float* gp = p.gates + (long long)n * p.HW * 4 * F + c0;
    float mu[16], rs[16], ga[16], be[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long o = (long long)n * 4 * F + q * F + c0 + c;
            mu[q * 4 + c] = p.mean1[o]; rs[q * 4 + c] = p.rstd1[o];
            ga[q * 4 + c] = p.g1[q * F + c0 + c]; be[q * 4 + c] = p.b1[q * F + c0 + c];
        }
    float mu2[4], rs2[4], g2v[4], b2v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        mu2[c] = p.mean2[(long long)n * F + c0 + c]; rs2[c] = p.rstd2[(long long)n * F + c0 + c];
        g2v[c] = p.g2[c0 + c]; b2v[c] = p.b2[c0 + c];
    }
    // per-pixel state kept in registers
    float xh[MAXPPT][16];     // normalised (pre-affine) gates
    float dz2[MAXPPT][4];     // d c_new (total)
    float xh2[MAXPPT][4];     // normalised c_pre
    float don[MAXPPT][4];     // d o_n (post-affine)
    float cpv[MAXPPT][4];
    float r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r2[i] = 0.f;
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            const float* q = gp + (long long)px * 4 * F;
            const float4 a0 = ld4(q), a1 = ld4(q + F), a2 = ld4(q + 2 * F), a3 = ld4(q + 3 * F);
            const float raw[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
#pragma unroll
            for (int i = 0; i < 16; ++i) xh[t][i] = (raw[i] - mu[i]) * rs[i];
            float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.c_prev) cp = ld4(p.c_prev + (long long)n * p.cp_sn + (long long)px * p.cp_sp + c0);
            cpv[t][0] = cp.x; cpv[t][1] = cp.y; cpv[t][2] = cp.z; cpv[t][3] = cp.w;
            float4 dh = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < p.ndh; ++k) {
                float4 tt = ld4(p.dh[k] + (long long)n * p.dh_sn[k] + (long long)px * p.dh_sp[k] + c0);
                dh.x += tt.x; dh.y += tt.y; dh.z += tt.z; dh.w += tt.w;
            }
            float4 dcn = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.dc_new) dcn = ld4(p.dc_new + ((long long)n * p.HW + px) * F + c0);
            const float dhv[4] = {dh.x, dh.y, dh.z, dh.w}, dcnv[4] = {dcn.x, dcn.y, dcn.z, dcn.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float in_ = xh[t][c] * ga[c] + be[c];
                float jn = xh[t][4 + c] * ga[4 + c] + be[4 + c];
                float fn = xh[t][8 + c] * ga[8 + c] + be[8 + c];
                float on = xh[t][12 + c] * ga[12 + c] + be[12 + c];
                float cpre = cpv[t][c] * sigmoidf_(fn + p.forget_bias) + sigmoidf_(in_) * tanhf_(jn);
                float x2 = (cpre - mu2[c]) * rs2[c];
                float cn = x2 * g2v[c] + b2v[c];
                float th = tanhf_(cn), so = sigmoidf_(on);
                xh2[t][c] = x2;
                dz2[t][c] = dhv[c] * so * (1.f - th * th) + dcnv[c];
                don[t][c] = dhv[c] * th * so * (1.f - so);
                r2[c] += dz2[t][c]; r2[4 + c] += dz2[t][c] * x2;
            }
        }
    }
    block_sum<8>(r2, sh);
    if (threadIdx.x < 4) {
        unsafeAtomicAdd(p.db2 + c0 + threadIdx.x, r2[threadIdx.x]);
        unsafeAtomicAdd(p.dg2 + c0 + threadIdx.x, r2[4 + threadIdx.x]);
    }
    const float inv = 1.f / (float)p.HW;
    float dg[MAXPPT][16];     // d (post-affine normalised gate)
    float r1[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) r1[i] = 0.f;
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            float dcp[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float dcpre = g2v[c] * rs2[c] * (dz2[t][c] - r2[c] * inv - xh2[t][c] * r2[4 + c] * inv);
                float in_ = xh[t][c] * ga[c] + be[c];
                float jn = xh[t][4 + c] * ga[4 + c] + be[4 + c];
                float fn = xh[t][8 + c] * ga[8 + c] + be[8 + c];
                float si = sigmoidf_(in_), tj = tanhf_(jn), sf = sigmoidf_(fn + p.forget_bias);
                dcp[c] = dcpre * sf;
                dg[t][c] = dcpre * tj * si * (1.f - si);
                dg[t][4 + c] = dcpre * si * (1.f - tj * tj);
                dg[t][8 + c] = dcpre * cpv[t][c] * sf * (1.f - sf);
                dg[t][12 + c] = don[t][c];
            }
            if (p.dc_prev) st4(p.dc_prev + ((long long)n * p.HW + px) * F + c0, make_float4(dcp[0], dcp[1], dcp[2], dcp[3]));
#pragma unroll
            for (int i = 0; i < 16; ++i) { r1[i] += dg[t][i]; r1[16 + i] += dg[t][i] * xh[t][i]; }
        }
    }
    block_sum<32>(r1, sh);
    if (threadIdx.x < 16) {
        const int q = threadIdx.x >> 2, c = threadIdx.x & 3;
        unsafeAtomicAdd(p.db1 + q * F + c0 + c, r1[threadIdx.x]);
        unsafeAtomicAdd(p.dg1 + q * F + c0 + c, r1[16 + threadIdx.x]);
    }
    float* dgp = p.dgates + (long long)n * p.HW * 4 * F + c0;
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            float o[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] = ga[i] * rs[i] * (dg[t][i] - r1[i] * inv - xh[t][i] * r1[16 + i] * inv);
            float* q = dgp + (long long)px * 4 * F;
            st4(q, make_float4(o[0], o[1], o[2], o[3]));
            st4(q + F, make_float4(o[4], o[5], o[6], o[7]));
            st4(q + 2 * F, make_float4(o[8], o[9], o[10], o[11]));
            st4(q + 3 * F, make_float4(o[12], o[13], o[14], o[15]));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Coalesced three-pass forward of the ConvLSTM gate block (used when the caller supplies a workspace).
// The fused kernel above assigns (sample, 4 channels) to a workgroup, i.e. every lane reads 16 B of a different
// 128-B line (the gate tensor is pixel-major with 4F channels per pixel) -- measured ~1 TB/s.  Here a workgroup takes
// a chunk of pixels x ALL channels, lanes run along the channel axis (full lines per wave instruction) and the two
// per-sample reductions go through small per-(n, c) workspaces:
//   pass 1  inorm_stats_kernel on the gate tensor (C = 4F)                    -> ws1 [N][4F][2]  shifted sums
//   pass 2  lstm_cell_kernel: normalise gates, c_pre, sigmoid(o); c_pre statistics -> ws2 [N][F][2], k2 [N][F]
//   pass 3  lstm_out_kernel: normalise c_pre, h = tanh(c) * sigmoid(o), write c_new and the h destinations
// Shifted sums (shift = the value at pixel 0) keep E[x^2] - E[x]^2 well conditioned, exactly as the instance-norm path.
// ------------------------------------------------------------------------------------------------------------
struct LstmWs { double* s1; double* s2; float* k2; float* so; };     // s1, s2: float64 sums (exact, order-independent: inorm_stats_kernel)

__global__ __launch_bounds__(NT) void lstm_cell_kernel(LstmP p, LstmWs w, int chunk) {
    extern __shared__ float sh[];                     // [rows][2*F]
    const int n = blockIdx.y, F = p.F, F4 = F / 4;
    const int fq = threadIdx.x % F4, prow = threadIdx.x / F4, rows = NT / F4;
    const int c0 = fq * 4;
    const long long g0 = (long long)n * p.HW * 4 * F;     // element index of this sample's gates
    const float inv = 1.f / (float)p.HW;
    float mu[16], rs[16], ga[16], be[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 k = make_float4(0.f, 0.f, 0.f, 0.f);       // shift of pass 1 = pixel 0 (none for the conv epilogue's sums)
        if (!p.stats1_ready) k = ld4x(p.gates, g0 + q * F + c0, p.gates16);
        const float kk[4] = {k.x, k.y, k.z, k.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double* s = w.s1 + ((long long)n * 4 * F + q * F + c0 + c) * 2;
            const double ms = s[0] * (double)inv;
            const float var = fmaxf((float)(s[1] * (double)inv - ms * ms), 0.f);
            mu[q * 4 + c] = kk[c] + (float)ms; rs[q * 4 + c] = rsqrtf(var + p.eps);
            ga[q * 4 + c] = p.g1[q * F + c0 + c]; be[q * 4 + c] = p.b1[q * F + c0 + c];
        }
    }
    if (blockIdx.x == 0 && prow == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            p.mean1[(long long)n * 4 * F + (i >> 2) * F + c0 + (i & 3)] = mu[i];
            p.rstd1[(long long)n * 4 * F + (i >> 2) * F + c0 + (i & 3)] = rs[i];
        }
    }
    auto cell = [&](int px, float (&cn)[4], float (&so)[4]) {
        const long long q = g0 + (long long)px * 4 * F + c0;
        const float4 gi = ld4x(p.gates, q, p.gates16), gj = ld4x(p.gates, q + F, p.gates16), gf = ld4x(p.gates, q + 2 * F, p.gates16),
                     go = ld4x(p.gates, q + 3 * F, p.gates16);
        float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.c_prev) cp = ld4(p.c_prev + (long long)n * p.cp_sn + (long long)px * p.cp_sp + c0);
        const float iv[4] = {gi.x, gi.y, gi.z, gi.w}, jv[4] = {gj.x, gj.y, gj.z, gj.w};
        const float fv[4] = {gf.x, gf.y, gf.z, gf.w}, ov[4] = {go.x, go.y, go.z, go.w};
        const float cpv[4] = {cp.x, cp.y, cp.z, cp.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float in_ = (iv[c] - mu[c]) * rs[c] * ga[c] + be[c];
            const float jn = (jv[c] - mu[4 + c]) * rs[4 + c] * ga[4 + c] + be[4 + c];
            const float fn = (fv[c] - mu[8 + c]) * rs[8 + c] * ga[8 + c] + be[8 + c];
            const float on = (ov[c] - mu[12 + c]) * rs[12 + c] * ga[12 + c] + be[12 + c];
            cn[c] = cpv[c] * sigmoidf_(fn + p.forget_bias) + sigmoidf_(in_) * tanhf_(jn);
            so[c] = sigmoidf_(on);
        }
    };
    float k2[4], dummy[4];
    cell(0, k2, dummy);                               // shift of the second reduction = c_pre at pixel 0 (same in every block)
    if (blockIdx.x == 0 && prow == 0) st4(w.k2 + (long long)n * F + c0, make_float4(k2[0], k2[1], k2[2], k2[3]));
    float s[4] = {0.f, 0.f, 0.f, 0.f}, qq[4] = {0.f, 0.f, 0.f, 0.f};
    const int p0 = blockIdx.x * chunk, p1 = min(p.HW, p0 + chunk);
    for (int px = p0 + prow; px < p1; px += rows) {
        float cn[4], so[4];
        cell(px, cn, so);
        st4(p.c_new + ((long long)n * p.HW + px) * F + c0, make_float4(cn[0], cn[1], cn[2], cn[3]));
        st4(w.so + ((long long)n * p.HW + px) * F + c0, make_float4(so[0], so[1], so[2], so[3]));
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float d = cn[c] - k2[c]; s[c] += d; qq[c] += d * d; }
    }
    float* d = sh + prow * 2 * F + c0;
#pragma unroll
    for (int c = 0; c < 4; ++c) { d[c] = s[c]; d[F + c] = qq[c]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * F; i += NT) {
        float t = 0.f;
        for (int r = 0; r < rows; ++r) t += sh[r * 2 * F + i];
        unsafeAtomicAdd(w.s2 + ((long long)n * F + (i % F)) * 2 + (i / F), (double)t);
    }
}

__global__ __launch_bounds__(NT) void lstm_out_kernel(LstmP p, LstmWs w, int chunk) {
    const int n = blockIdx.y, F = p.F, F4 = F / 4;
    const int fq = threadIdx.x % F4, prow = threadIdx.x / F4, rows = NT / F4;
    const int c0 = fq * 4;
    const float inv = 1.f / (float)p.HW;
    const float4 k2 = ld4(w.k2 + (long long)n * F + c0);
    const float kk[4] = {k2.x, k2.y, k2.z, k2.w};
    float m2[4], r2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double* s = w.s2 + ((long long)n * F + c0 + c) * 2;
        const double ms = s[0] * (double)inv;
        const float var = fmaxf((float)(s[1] * (double)inv - ms * ms), 0.f);
        m2[c] = kk[c] + (float)ms; r2[c] = rsqrtf(var + p.eps);
    }
    if (blockIdx.x == 0 && prow == 0) {
        st4(p.mean2 + (long long)n * F + c0, make_float4(m2[0], m2[1], m2[2], m2[3]));
        st4(p.rstd2 + (long long)n * F + c0, make_float4(r2[0], r2[1], r2[2], r2[3]));
    }
    const float4 g2 = ld4(p.g2 + c0), b2 = ld4(p.b2 + c0);
    const float g2v[4] = {g2.x, g2.y, g2.z, g2.w}, b2v[4] = {b2.x, b2.y, b2.z, b2.w};
    const int p0 = blockIdx.x * chunk, p1 = min(p.HW, p0 + chunk);
    for (int px = p0 + prow; px < p1; px += rows) {
        float* cq = p.c_new + ((long long)n * p.HW + px) * F + c0;
        const float4 cpre = ld4(cq), so = ld4(w.so + ((long long)n * p.HW + px) * F + c0);
        const float cv[4] = {cpre.x, cpre.y, cpre.z, cpre.w}, sv[4] = {so.x, so.y, so.z, so.w};
        float cn[4], hv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            cn[c] = (cv[c] - m2[c]) * r2[c] * g2v[c] + b2v[c];
            hv[c] = tanhf_(cn[c]) * sv[c];
        }
        st4(cq, make_float4(cn[0], cn[1], cn[2], cn[3]));
        const float4 h4 = make_float4(hv[0], hv[1], hv[2], hv[3]);
        for (int k = 0; k < p.nh; ++k) st4x(p.h[k], (long long)n * p.h_sn[k] + (long long)px * p.h_sp[k] + c0, h4, p.h16[k]);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Coalesced three-pass backward (same thread mapping as lstm_cell_kernel; workspace r2 [N][F][2], r1 [N][4F][2],
// dz2 [N][HW][F]; the raw gate gradients are parked in the dgates output between passes 2 and 3):
//   pass 1  dz2 = d c_new (total), d o_n ; r2 = sum dz2, sum dz2 * x2 over the plane
//   pass 2  d c_pre through the second norm, d c_prev, raw d(i, j, f) ; r1 = sum dg, sum dg * xh ; dgamma2 / dbeta2
//   pass 3  dgates through the first norm ; dgamma1 / dbeta1
// ------------------------------------------------------------------------------------------------------------
struct LstmBws { double* r2; double* r1; float* dz2; };            // r2, r1: float64 sums (exact, order-independent)

struct LstmLane {            // per-thread constants of the (sample, 4 channels) column this thread owns
    float mu[16], rs[16], ga[16], be[16], mu2[4], rs2[4], g2[4], b2[4];
};

__device__ __forceinline__ void lstm_lane_load(const LstmP& p, int n, int c0, LstmLane& L) {
    const int F = p.F;
    // 20 float4 loads (was 80 scalar ones in front of every backward pass): c0 and F are multiples of 4, the tables 16-byte aligned
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const long long o = (long long)n * 4 * F + q * F + c0;
        const float4 m = ld4(p.mean1 + o), r = ld4(p.rstd1 + o), g = ld4(p.g1 + q * F + c0), b = ld4(p.b1 + q * F + c0);
        L.mu[q * 4] = m.x; L.mu[q * 4 + 1] = m.y; L.mu[q * 4 + 2] = m.z; L.mu[q * 4 + 3] = m.w;
        L.rs[q * 4] = r.x; L.rs[q * 4 + 1] = r.y; L.rs[q * 4 + 2] = r.z; L.rs[q * 4 + 3] = r.w;
        L.ga[q * 4] = g.x; L.ga[q * 4 + 1] = g.y; L.ga[q * 4 + 2] = g.z; L.ga[q * 4 + 3] = g.w;
        L.be[q * 4] = b.x; L.be[q * 4 + 1] = b.y; L.be[q * 4 + 2] = b.z; L.be[q * 4 + 3] = b.w;
    }
    const float4 m2 = ld4(p.mean2 + (long long)n * F + c0), r2 = ld4(p.rstd2 + (long long)n * F + c0), g2 = ld4(p.g2 + c0), b2 = ld4(p.b2 + c0);
    L.mu2[0] = m2.x; L.mu2[1] = m2.y; L.mu2[2] = m2.z; L.mu2[3] = m2.w;
    L.rs2[0] = r2.x; L.rs2[1] = r2.y; L.rs2[2] = r2.z; L.rs2[3] = r2.w;
    L.g2[0] = g2.x; L.g2[1] = g2.y; L.g2[2] = g2.z; L.g2[3] = g2.w;
    L.b2[0] = b2.x; L.b2[1] = b2.y; L.b2[2] = b2.z; L.b2[3] = b2.w;
}

// normalised gates xh[16] and previous cell state of pixel px
__device__ __forceinline__ void lstm_load_px(const LstmP& p, const LstmLane& L, int n, int px, int c0, float (&xh)[16], float (&cp)[4]) {
    const int F = p.F;
    const long long q = ((long long)n * p.HW + px) * 4 * F + c0;
    const float4 a0 = ld4x(p.gates, q, p.gates16), a1 = ld4x(p.gates, q + F, p.gates16), a2 = ld4x(p.gates, q + 2 * F, p.gates16),
                 a3 = ld4x(p.gates, q + 3 * F, p.gates16);
    const float raw[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) xh[i] = (raw[i] - L.mu[i]) * L.rs[i];
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.c_prev) c = ld4(p.c_prev + (long long)n * p.cp_sn + (long long)px * p.cp_sp + c0);
    cp[0] = c.x; cp[1] = c.y; cp[2] = c.z; cp[3] = c.w;
}

__global__ __launch_bounds__(NT) void lstm_bwd1_kernel(LstmP p, LstmBws w, int chunk) {
    extern __shared__ float sh[];                     // [rows][2*F]
    const int n = blockIdx.y, F = p.F, F4 = F / 4;
    const int fq = threadIdx.x % F4, prow = threadIdx.x / F4, rows = NT / F4;
    const int c0 = fq * 4;
    LstmLane L;
    lstm_lane_load(p, n, c0, L);
    float ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
    const int p0 = blockIdx.x * chunk, p1 = min(p.HW, p0 + chunk);
    for (int px = p0 + prow; px < p1; px += rows) {
        float xh[16], cp[4];
        lstm_load_px(p, L, n, px, c0, xh, cp);
        float4 dh = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < p.ndh; ++k) {
            const float4 t = ld4(p.dh[k] + (long long)n * p.dh_sn[k] + (long long)px * p.dh_sp[k] + c0);
            dh.x += t.x; dh.y += t.y; dh.z += t.z; dh.w += t.w;
        }
        float4 dcn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.dc_new) dcn = ld4(p.dc_new + ((long long)n * p.HW + px) * F + c0);
        const float dhv[4] = {dh.x, dh.y, dh.z, dh.w}, dcnv[4] = {dcn.x, dcn.y, dcn.z, dcn.w};
        float dz[4], don[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float in_ = xh[c] * L.ga[c] + L.be[c];
            const float jn = xh[4 + c] * L.ga[4 + c] + L.be[4 + c];
            const float fn = xh[8 + c] * L.ga[8 + c] + L.be[8 + c];
            const float on = xh[12 + c] * L.ga[12 + c] + L.be[12 + c];
            const float cpre = cp[c] * sigmoidf_(fn + p.forget_bias) + sigmoidf_(in_) * tanhf_(jn);
            const float x2 = (cpre - L.mu2[c]) * L.rs2[c];
            const float th = tanhf_(x2 * L.g2[c] + L.b2[c]), so = sigmoidf_(on);
            dz[c] = dhv[c] * so * (1.f - th * th) + dcnv[c];
            don[c] = dhv[c] * th * so * (1.f - so);
            ra[c] += dz[c]; rb[c] += dz[c] * x2;
        }
        st4(w.dz2 + ((long long)n * p.HW + px) * F + c0, make_float4(dz[0], dz[1], dz[2], dz[3]));
        st4(p.draw + ((long long)n * p.HW + px) * 4 * F + 3 * F + c0, make_float4(don[0], don[1], don[2], don[3]));
    }
    float* d = sh + prow * 2 * F + c0;
#pragma unroll
    for (int c = 0; c < 4; ++c) { d[c] = ra[c]; d[F + c] = rb[c]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * F; i += NT) {
        float t = 0.f;
        for (int r = 0; r < rows; ++r) t += sh[r * 2 * F + i];
        unsafeAtomicAdd(w.r2 + ((long long)n * F + (i % F)) * 2 + (i / F), (double)t);
    }
}

__global__ __launch_bounds__(NT) void lstm_bwd2_kernel(LstmP p, LstmBws w, int chunk) {
    extern __shared__ float sh[];                     // [rows][8*F]
    const int n = blockIdx.y, F = p.F, F4 = F / 4;
    const int fq = threadIdx.x % F4, prow = threadIdx.x / F4, rows = NT / F4;
    const int c0 = fq * 4;
    LstmLane L;
    lstm_lane_load(p, n, c0, L);
    const float inv = 1.f / (float)p.HW;
    float r2a[4], r2b[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        r2a[c] = (float)w.r2[((long long)n * F + c0 + c) * 2]; r2b[c] = (float)w.r2[((long long)n * F + c0 + c) * 2 + 1];
    }
    if (blockIdx.x == 0 && prow == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { unsafeAtomicAdd(p.db2 + c0 + c, r2a[c]); unsafeAtomicAdd(p.dg2 + c0 + c, r2b[c]); }
    }
    float r1[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) r1[i] = 0.f;
    const int p0 = blockIdx.x * chunk, p1 = min(p.HW, p0 + chunk);
    for (int px = p0 + prow; px < p1; px += rows) {
        float xh[16], cp[4];
        lstm_load_px(p, L, n, px, c0, xh, cp);
        const float4 dz4 = ld4(w.dz2 + ((long long)n * p.HW + px) * F + c0);
        const float dz[4] = {dz4.x, dz4.y, dz4.z, dz4.w};
        float* gq = p.draw + ((long long)n * p.HW + px) * 4 * F + c0;
        const float4 don4 = ld4(gq + 3 * F);
        const float don[4] = {don4.x, don4.y, don4.z, don4.w};
        float dg[16], dcp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float in_ = xh[c] * L.ga[c] + L.be[c];
            const float jn = xh[4 + c] * L.ga[4 + c] + L.be[4 + c];
            const float fn = xh[8 + c] * L.ga[8 + c] + L.be[8 + c];
            const float si = sigmoidf_(in_), tj = tanhf_(jn), sf = sigmoidf_(fn + p.forget_bias);
            const float cpre = cp[c] * sf + si * tj;
            const float x2 = (cpre - L.mu2[c]) * L.rs2[c];
            const float dcpre = L.g2[c] * L.rs2[c] * (dz[c] - r2a[c] * inv - x2 * r2b[c] * inv);
            dcp[c] = dcpre * sf;
            dg[c] = dcpre * tj * si * (1.f - si);
            dg[4 + c] = dcpre * si * (1.f - tj * tj);
            dg[8 + c] = dcpre * cp[c] * sf * (1.f - sf);
            dg[12 + c] = don[c];
        }
        if (p.dc_prev) st4(p.dc_prev + ((long long)n * p.HW + px) * F + c0, make_float4(dcp[0], dcp[1], dcp[2], dcp[3]));
        st4(gq, make_float4(dg[0], dg[1], dg[2], dg[3]));
        st4(gq + F, make_float4(dg[4], dg[5], dg[6], dg[7]));
        st4(gq + 2 * F, make_float4(dg[8], dg[9], dg[10], dg[11]));
#pragma unroll
        for (int i = 0; i < 16; ++i) { r1[i] += dg[i]; r1[16 + i] += dg[i] * xh[i]; }
    }
    // sh[prow][which][q*F + c]
    float* d = sh + prow * 8 * F;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) { d[q * F + c0 + c] = r1[q * 4 + c]; d[4 * F + q * F + c0 + c] = r1[16 + q * 4 + c]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 8 * F; i += NT) {
        float t = 0.f;
        for (int r = 0; r < rows; ++r) t += sh[r * 8 * F + i];
        const int which = i / (4 * F), ch = i % (4 * F);
        unsafeAtomicAdd(w.r1 + ((long long)n * 4 * F + ch) * 2 + which, (double)t);
    }
}

__global__ __launch_bounds__(NT) void lstm_bwd3_kernel(LstmP p, LstmBws w, int chunk) {
    const int n = blockIdx.y, F = p.F, F4 = F / 4;
    const int fq = threadIdx.x % F4, prow = threadIdx.x / F4, rows = NT / F4;
    const int c0 = fq * 4;
    const float inv = 1.f / (float)p.HW;
    float mu[16], rs[16], ga[16], s1[16], s2[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long o = (long long)n * 4 * F + q * F + c0 + c;
            mu[q * 4 + c] = p.mean1[o]; rs[q * 4 + c] = p.rstd1[o]; ga[q * 4 + c] = p.g1[q * F + c0 + c];
            s1[q * 4 + c] = (float)w.r1[o * 2]; s2[q * 4 + c] = (float)w.r1[o * 2 + 1];
        }
    if (blockIdx.x == 0 && prow == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsafeAtomicAdd(p.db1 + q * F + c0 + c, s1[q * 4 + c]);
                unsafeAtomicAdd(p.dg1 + q * F + c0 + c, s2[q * 4 + c]);
            }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { s1[i] *= inv; s2[i] *= inv; }
    const int p0 = blockIdx.x * chunk, p1 = min(p.HW, p0 + chunk);
    for (int px = p0 + prow; px < p1; px += rows) {
        const long long q = ((long long)n * p.HW + px) * 4 * F + c0;
        const float* gq = p.draw + q;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 raw = ld4x(p.gates, q + g * F, p.gates16), dgv = ld4(gq + g * F);
            const float rv[4] = {raw.x, raw.y, raw.z, raw.w}, dv[4] = {dgv.x, dgv.y, dgv.z, dgv.w};
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = g * 4 + c;
                const float xh = (rv[c] - mu[i]) * rs[i];
                o[c] = ga[i] * rs[i] * (dv[c] - s1[i] - xh * s2[i]);
            }
            st4x(p.dgates, q + g * F, make_float4(o[0], o[1], o[2], o[3]), p.dgates16);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// One-launch ConvLSTM gate block, forward and backward (option "lstm_fused", default on; round 3).
// Ownership as in the first version of this file -- one workgroup = (sample, a slab of 4*Q channels) over the WHOLE plane, so both
// per-sample reductions are workgroup-local and the three passes (and their two round trips of c_pre / sigmoid(o) / the raw gate
// gradients through HBM) collapse into one launch -- with what made that version slow removed:
//  * every global load of a thread (PPT pixel items x (4 gate quads + c_prev [+ dh, dc'])) is issued unconditionally, on clamped
//    addresses, before the first use (under `if (px < HW) load` hipcc emits load / s_waitcnt vmcnt(0) pairs: 20 serialised L2 round
//    trips per thread);
//  * Q threads side by side cover a slab's 4*Q channels of one pixel (8..64 contiguous bytes per gate and pixel), Q chosen as large
//    as still leaves one workgroup per CU;
//  * the workgroups of one sample sit on ONE XCD (blockIdx -> (xcd, slab, sample)): the 4F-wide pixel rows they all take their
//    slices from are fetched into that XCD's L2 once;
//  * IN(4F) statistics come from the gate convolution's epilogue (stats1_ready) or are reduced locally from registers.
// ------------------------------------------------------------------------------------------------------------
// Sums NV per-thread values over the threads with the same (tid % Q); result broadcast to them.  Through LDS in a fixed order
// (deterministic, unlike atomics; and ~10x cheaper than 6 * NV serialised ds_bpermute + wait pairs of a shuffle tree at NV = 32):
// every thread parks its values, thread j sums value (j % NV) over NV rows of one q, NV * Q threads fold the NT / (NV * Q) partial
// sums.  `sh` holds LSTM_SUM_FLOATS(NV, Q) floats.
#define LSTM_SUM_FLOATS(NV_, Q_) (NT * ((NV_) + 1) + NT + (NV_) * (Q_))
template <int NV, int Q>
__device__ __forceinline__ void block_sum_q(float (&v)[NV], float* sh) {
    static_assert(NT % (NV * Q) == 0, "NV * Q divides the workgroup");
    float* part = sh + NT * (NV + 1);
    float* fin = part + NT;
    const int tid = threadIdx.x;
    __syncthreads();                                    // earlier readers of sh are done
#pragma unroll
    for (int i = 0; i < NV; ++i) sh[tid * (NV + 1) + i] = v[i];
    __syncthreads();
    {
        const int i = tid % NV, rest = tid / NV, q = rest % Q, pt = rest / Q;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) s += sh[((pt * NV + k) * Q + q) * (NV + 1) + i];
        part[tid] = s;                                  // tid == (pt * Q + q) * NV + i
    }
    __syncthreads();
    constexpr int NP = NT / (NV * Q);
    if (tid < NV * Q) {
        const int i = tid % NV, q = tid / NV;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) s += part[(k * Q + q) * NV + i];
        fin[q * NV + i] = s;
    }
    __syncthreads();
    const int q = tid & (Q - 1);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = fin[q * NV + i];
}

// Developer build (SAVP_EXTRA_FLAGS=-DSAVP_LSTM_STAMPS, tests/tools/lstm_stamps.py): s_memtime stamps of three workgroups' first
// wave at the phase boundaries of the one-launch kernels.  Not compiled into the shipped library.
#ifdef SAVP_LSTM_STAMPS
__device__ unsigned long long g_lstm_t[3][8];
#define LT(i) do { if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x - 1)) \
    g_lstm_t[blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 2 : 1)][i] = __builtin_readcyclecounter(); } while (0)
#define LT_WAITVM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
extern "C" int savp_debug_lstm_times(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lstm_t), sizeof(g_lstm_t)) == hipSuccess ? 0 : -1;
}
#else
#define LT(i) do {} while (0)
#define LT_WAITVM() do {} while (0)
#endif

template <bool G16> struct GateQuad;
template <> struct GateQuad<true> {
    uint2 u;
    __device__ __forceinline__ void load(const float* base, long long idx) {
        u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
    }
    __device__ __forceinline__ void get(float* o) const {
        o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
        o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
    }
};
template <> struct GateQuad<false> {
    float4 v;
    __device__ __forceinline__ void load(const float* base, long long idx) { v = ld4(base + idx); }
    __device__ __forceinline__ void get(float* o) const { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
};

__device__ __forceinline__ void lstm_block_owner(int nslab, int xcd_map, int& n, int& slab) {
    const int b = blockIdx.x;
    if (xcd_map) { const int r = b >> 3; slab = r % nslab; n = (b & 7) + 8 * (r / nslab); }
    else { n = b / nslab; slab = b % nslab; }
}

template <int Q, int PPT, bool G16>
__global__ __launch_bounds__(NT) void lstm_fused_fwd_kernel(LstmP p, const double* __restrict__ s1, int nslab, int xcd_map) {
    __shared__ float sh[LSTM_SUM_FLOATS(16, Q)];
    LT(0);
    constexpr int ROWS = NT / Q;
    int n, slab;
    lstm_block_owner(nslab, xcd_map, n, slab);
    const int q = threadIdx.x & (Q - 1), prow = threadIdx.x / Q;
    const int F = p.F, HW = p.HW, c0 = (slab * Q + q) * 4;
    const long long g0 = (long long)n * HW * 4 * F + c0;
    // ---- every load of this thread, issued before anything is used ---------------------------------------------
    GateQuad<G16> gq[PPT][4];
    float4 cpq[PPT];
    int pxs[PPT];
    bool ok[PPT];
#pragma unroll
    for (int t = 0; t < PPT; ++t) {
        const int px = prow + t * ROWS;
        ok[t] = px < HW;
        pxs[t] = ok[t] ? px : HW - 1;
        const long long idx = g0 + (long long)pxs[t] * 4 * F;
#pragma unroll
        for (int g = 0; g < 4; ++g) gq[t][g].load(p.gates, idx + g * F);
    }
    if (p.c_prev) {
#pragma unroll
        for (int t = 0; t < PPT; ++t) cpq[t] = ld4(p.c_prev + (long long)n * p.cp_sn + (long long)pxs[t] * p.cp_sp + c0);
    } else {
#pragma unroll
        for (int t = 0; t < PPT; ++t) cpq[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 g1q[4], b1q[4];
    double2 sd[4][4];                                 // [gate][channel] {sum, sum of squares}: float64 from the conv epilogue (exact sums)
#pragma unroll
    for (int g = 0; g < 4; ++g) { g1q[g] = ld4(p.g1 + g * F + c0); b1q[g] = ld4(p.b1 + g * F + c0); }
    if (s1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const double2* s = reinterpret_cast<const double2*>(s1 + ((long long)n * 4 * F + g * F + c0) * 2);
#pragma unroll
            for (int c = 0; c < 4; ++c) sd[g][c] = s[c];
        }
    }
    const float4 g2q = ld4(p.g2 + c0), b2q = ld4(p.b2 + c0);
    LT(1);
    LT_WAITVM();
    LT(2);
    // ---- IN(4F) -----------------------------------------------------------------------------------------------
    const float inv = 1.f / (float)HW;
    float x[PPT][16];
#pragma unroll
    for (int t = 0; t < PPT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) gq[t][g].get(&x[t][g * 4]);
    float mu[16], rs[16];
    if (s1) {          // unshifted sums of the fp32 accumulators (conv epilogue): [sum, sumsq] pairs per channel
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double m = sd[g][c].x * (double)inv;
                mu[g * 4 + c] = (float)m;
                rs[g * 4 + c] = rsqrtf(fmaxf((float)(sd[g][c].y * (double)inv - m * m), 0.f) + p.eps);
            }
        }
    } else {
        float s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int t = 0; t < PPT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] += ok[t] ? x[t][i] : 0.f;
        block_sum_q<16, Q>(s, sh);
#pragma unroll
        for (int i = 0; i < 16; ++i) { mu[i] = s[i] * inv; s[i] = 0.f; }
#pragma unroll
        for (int t = 0; t < PPT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { const float d = x[t][i] - mu[i]; s[i] += ok[t] ? d * d : 0.f; }
        block_sum_q<16, Q>(s, sh);
#pragma unroll
        for (int i = 0; i < 16; ++i) rs[i] = rsqrtf(s[i] * inv + p.eps);
    }
    if (prow == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const long long o = (long long)n * 4 * F + g * F + c0;
            st4(p.mean1 + o, make_float4(mu[g * 4], mu[g * 4 + 1], mu[g * 4 + 2], mu[g * 4 + 3]));
            st4(p.rstd1 + o, make_float4(rs[g * 4], rs[g * 4 + 1], rs[g * 4 + 2], rs[g * 4 + 3]));
        }
    }
    float ga[16], be[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        ga[g * 4] = g1q[g].x; ga[g * 4 + 1] = g1q[g].y; ga[g * 4 + 2] = g1q[g].z; ga[g * 4 + 3] = g1q[g].w;
        be[g * 4] = b1q[g].x; be[g * 4 + 1] = b1q[g].y; be[g * 4 + 2] = b1q[g].z; be[g * 4 + 3] = b1q[g].w;
    }
    // ---- gates -> c_pre, sigmoid(o) ; IN(F) of c_pre -------------------------------------------------------------
    float cpre[PPT][4], so[PPT][4];
    float s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < PPT; ++t) {
        const float cpv[4] = {cpq[t].x, cpq[t].y, cpq[t].z, cpq[t].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float in_ = (x[t][c] - mu[c]) * rs[c] * ga[c] + be[c];
            const float jn = (x[t][4 + c] - mu[4 + c]) * rs[4 + c] * ga[4 + c] + be[4 + c];
            const float fn = (x[t][8 + c] - mu[8 + c]) * rs[8 + c] * ga[8 + c] + be[8 + c];
            const float on = (x[t][12 + c] - mu[12 + c]) * rs[12 + c] * ga[12 + c] + be[12 + c];
            cpre[t][c] = cpv[c] * sigmoidf_(fn + p.forget_bias) + sigmoidf_(in_) * tanhf_(jn);
            so[t][c] = sigmoidf_(on);
            s2[c] += ok[t] ? cpre[t][c] : 0.f;
        }
    }
    LT(3);
    block_sum_q<4, Q>(s2, sh);
    LT(4);
    float mu2[4], rs2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { mu2[c] = s2[c] * inv; s2[c] = 0.f; }
#pragma unroll
    for (int t = 0; t < PPT; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float d = cpre[t][c] - mu2[c]; s2[c] += ok[t] ? d * d : 0.f; }
    block_sum_q<4, Q>(s2, sh);
#pragma unroll
    for (int c = 0; c < 4; ++c) rs2[c] = rsqrtf(s2[c] * inv + p.eps);
    if (prow == 0) {
        st4(p.mean2 + (long long)n * F + c0, make_float4(mu2[0], mu2[1], mu2[2], mu2[3]));
        st4(p.rstd2 + (long long)n * F + c0, make_float4(rs2[0], rs2[1], rs2[2], rs2[3]));
    }
    const float g2v[4] = {g2q.x, g2q.y, g2q.z, g2q.w}, b2v[4] = {b2q.x, b2q.y, b2q.z, b2q.w};
    LT(5);
#pragma unroll
    for (int t = 0; t < PPT; ++t) {
        if (!ok[t]) continue;
        float cn[4], hv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            cn[c] = (cpre[t][c] - mu2[c]) * rs2[c] * g2v[c] + b2v[c];
            hv[c] = tanhf_(cn[c]) * so[t][c];
        }
        const int px = pxs[t];
        st4(p.c_new + ((long long)n * HW + px) * F + c0, make_float4(cn[0], cn[1], cn[2], cn[3]));
        const float4 h4 = make_float4(hv[0], hv[1], hv[2], hv[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < p.nh) st4x(p.h[k], (long long)n * p.h_sn[k] + (long long)px * p.h_sp[k] + c0, h4, p.h16[k]);
    }
    LT(6);
    LT_WAITVM();
    LT(7);
}

template <int Q, int PPT, bool G16>
__global__ __launch_bounds__(NT) void lstm_fused_bwd_kernel(LstmP p, int nslab, int xcd_map) {
    __shared__ float sh[LSTM_SUM_FLOATS(32, Q)];
    LT(0);
    constexpr int ROWS = NT / Q;
    int n, slab;
    lstm_block_owner(nslab, xcd_map, n, slab);
    const int q = threadIdx.x & (Q - 1), prow = threadIdx.x / Q;
    const int F = p.F, HW = p.HW, c0 = (slab * Q + q) * 4;
    const long long g0 = (long long)n * HW * 4 * F + c0;
    // ---- every load of this thread up front ----------------------------------------------------------------------
    GateQuad<G16> gq[PPT][4];
    float4 cpq[PPT], dhq[PPT], dcq[PPT];
    int pxs[PPT];
    float okf[PPT];
#pragma unroll
    for (int t = 0; t < PPT; ++t) {
        const int px = prow + t * ROWS;
        okf[t] = px < HW ? 1.f : 0.f;
        pxs[t] = px < HW ? px : HW - 1;
        const long long idx = g0 + (long long)pxs[t] * 4 * F;
#pragma unroll
        for (int g = 0; g < 4; ++g) gq[t][g].load(p.gates, idx + g * F);
        dhq[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        cpq[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        dcq[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (p.c_prev) {
#pragma unroll
        for (int t = 0; t < PPT; ++t) cpq[t] = ld4(p.c_prev + (long long)n * p.cp_sn + (long long)pxs[t] * p.cp_sp + c0);
    }
    if (p.dc_new) {
#pragma unroll
        for (int t = 0; t < PPT; ++t) dcq[t] = ld4(p.dc_new + ((long long)n * HW + pxs[t]) * F + c0);
    }
    // up to three gradient sources of h' (consumers of this step + the next step's gate conv); a fourth is rare.  Loads stay
    // unconditional (source min(k, ndh-1), weight 0 beyond ndh): a load under `if (k < ndh)` is waited for inside its branch
    float4 dhs[3][PPT];
    {
        const int nd = p.ndh > 0 ? p.ndh : 1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int kk = k < nd ? k : nd - 1;
            const float* src = p.ndh > 0 ? p.dh[kk] : p.gates;          // ndh == 0: any valid address, weight 0
            const long long sn = p.ndh > 0 ? p.dh_sn[kk] : 0, sp = p.ndh > 0 ? p.dh_sp[kk] : 0;
#pragma unroll
            for (int t = 0; t < PPT; ++t) dhs[k][t] = ld4(src + (long long)n * sn + (long long)pxs[t] * sp + (p.ndh > 0 ? c0 : 0));
        }
    }
    if (p.ndh > 3) {
#pragma unroll
        for (int t = 0; t < PPT; ++t) dhq[t] = ld4(p.dh[3] + (long long)n * p.dh_sn[3] + (long long)pxs[t] * p.dh_sp[3] + c0);
    }
    float mu[16], rs[16], ga[16], be[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const long long o = (long long)n * 4 * F + g * F + c0;
        const float4 m = ld4(p.mean1 + o), r = ld4(p.rstd1 + o), gg = ld4(p.g1 + g * F + c0), bb = ld4(p.b1 + g * F + c0);
        mu[g * 4] = m.x; mu[g * 4 + 1] = m.y; mu[g * 4 + 2] = m.z; mu[g * 4 + 3] = m.w;
        rs[g * 4] = r.x; rs[g * 4 + 1] = r.y; rs[g * 4 + 2] = r.z; rs[g * 4 + 3] = r.w;
        ga[g * 4] = gg.x; ga[g * 4 + 1] = gg.y; ga[g * 4 + 2] = gg.z; ga[g * 4 + 3] = gg.w;
        be[g * 4] = bb.x; be[g * 4 + 1] = bb.y; be[g * 4 + 2] = bb.z; be[g * 4 + 3] = bb.w;
    }
    const float4 m2q = ld4(p.mean2 + (long long)n * F + c0), r2q = ld4(p.rstd2 + (long long)n * F + c0), g2q = ld4(p.g2 + c0), b2q = ld4(p.b2 + c0);
    const float mu2[4] = {m2q.x, m2q.y, m2q.z, m2q.w}, rs2[4] = {r2q.x, r2q.y, r2q.z, r2q.w};
    const float g2v[4] = {g2q.x, g2q.y, g2q.z, g2q.w}, b2v[4] = {b2q.x, b2q.y, b2q.z, b2q.w};
    const float inv = 1.f / (float)HW;
    LT(1);
    LT_WAITVM();
    LT(2);
    // ---- phase 1: d c_new (total), d o_n ; sums of the second norm's backward ---------------------------------------
    float xh[PPT][16];          // normalised (pre-affine) gates
    float dg[PPT][16];          // gradients of the post-affine normalised gates (o first, the rest in phase 2)
    float dz[PPT][4];
    float r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r2[i] = 0.f;
#pragma unroll
    for (int t = 0; t < PPT; ++t) {
        float raw[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) gq[t][g].get(&raw[g * 4]);
#pragma unroll
        for (int i = 0; i < 16; ++i) xh[t][i] = (raw[i] - mu[i]) * rs[i];
        const float cpv[4] = {cpq[t].x, cpq[t].y, cpq[t].z, cpq[t].w};
        float dhv[4] = {dhq[t].x, dhq[t].y, dhq[t].z, dhq[t].w};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float wk = k < p.ndh ? 1.f : 0.f;
            dhv[0] += wk * dhs[k][t].x; dhv[1] += wk * dhs[k][t].y; dhv[2] += wk * dhs[k][t].z; dhv[3] += wk * dhs[k][t].w;
        }
        const float dcnv[4] = {dcq[t].x, dcq[t].y, dcq[t].z, dcq[t].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float in_ = xh[t][c] * ga[c] + be[c];
            const float jn = xh[t][4 + c] * ga[4 + c] + be[4 + c];
            const float fn = xh[t][8 + c] * ga[8 + c] + be[8 + c];
            const float on = xh[t][12 + c] * ga[12 + c] + be[12 + c];
            const float cpre = cpv[c] * sigmoidf_(fn + p.forget_bias) + sigmoidf_(in_) * tanhf_(jn);
            const float x2 = (cpre - mu2[c]) * rs2[c];
            const float th = tanhf_(x2 * g2v[c] + b2v[c]), so = sigmoidf_(on);
            dz[t][c] = (dhv[c] * so * (1.f - th * th) + dcnv[c]) * okf[t];
            dg[t][12 + c] = dhv[c] * th * so * (1.f - so) * okf[t];
            r2[c] += dz[t][c]; r2[4 + c] += dz[t][c] * x2;
        }
    }
    LT(3);
    block_sum_q<8, Q>(r2, sh);
    LT(4);
    if (prow == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { unsafeAtomicAdd(p.db2 + c0 + c, r2[c]); unsafeAtomicAdd(p.dg2 + c0 + c, r2[4 + c]); }
    }
    // ---- phase 2: through the second norm and the gates ; sums of the first norm's backward ------------------------------
    float r1[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) r1[i] = 0.f;
#pragma unroll
    for (int t = 0; t < PPT; ++t) {
        const float cpv[4] = {cpq[t].x, cpq[t].y, cpq[t].z, cpq[t].w};
        float dcp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float in_ = xh[t][c] * ga[c] + be[c];
            const float jn = xh[t][4 + c] * ga[4 + c] + be[4 + c];
            const float fn = xh[t][8 + c] * ga[8 + c] + be[8 + c];
            const float si = sigmoidf_(in_), tj = tanhf_(jn), sf = sigmoidf_(fn + p.forget_bias);
            const float x2 = (cpv[c] * sf + si * tj - mu2[c]) * rs2[c];
            const float dcpre = g2v[c] * rs2[c] * (dz[t][c] - r2[c] * inv - x2 * r2[4 + c] * inv) * okf[t];
            dcp[c] = dcpre * sf;
            dg[t][c] = dcpre * tj * si * (1.f - si);
            dg[t][4 + c] = dcpre * si * (1.f - tj * tj);
            dg[t][8 + c] = dcpre * cpv[c] * sf * (1.f - sf);
        }
        if (p.dc_prev && okf[t] != 0.f) st4(p.dc_prev + ((long long)n * HW + pxs[t]) * F + c0, make_float4(dcp[0], dcp[1], dcp[2], dcp[3]));
#pragma unroll
        for (int i = 0; i < 16; ++i) { r1[i] += dg[t][i]; r1[16 + i] += dg[t][i] * xh[t][i]; }
    }
    LT(5);
    block_sum_q<32, Q>(r1, sh);
    LT(6);
    if (prow == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                unsafeAtomicAdd(p.db1 + g * F + c0 + c, r1[g * 4 + c]);
                unsafeAtomicAdd(p.dg1 + g * F + c0 + c, r1[16 + g * 4 + c]);
            }
    }
    // ---- phase 3: through the first norm --------------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < PPT; ++t) {
        if (okf[t] == 0.f) continue;
        const long long idx = g0 + (long long)pxs[t] * 4 * F;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = g * 4 + c;
                o[c] = ga[i] * rs[i] * (dg[t][i] - r1[i] * inv - xh[t][i] * r1[16 + i] * inv);
            }
            st4x(p.dgates, idx + g * F, make_float4(o[0], o[1], o[2], o[3]), p.dgates16);
        }
    }
    LT_WAITVM();
    LT(7);
}

// configuration of the one-launch kernels for a call: Q threads per pixel, PPT pixel items per thread; false = not applicable
static bool lstm_fused_cfg(const SavpLstmArgs* a, bool fwd, int& Q, int& PPT, int& nslab, int& xcd_map) {
    if (!savp_opt(OPT_LSTM_FUSED) || a->F % 4 || a->HW < 1 || a->HW > 4 * NT || a->N < 1) return false;
    if (fwd && a->gates_bf16 && !a->stats1_ready) return false;       // bf16 gates come with their statistics from the conv epilogue
    if (fwd && a->stats1_ready && !a->ws_stats) return false;
    Q = 1;
    for (int c = 4; c > 1; c >>= 1)
        if (a->F % (4 * c) == 0 && a->HW <= 4 * (NT / c) && (long long)a->N * (a->F / (4 * c)) >= 256) { Q = c; break; }
    const int fq = savp_opt(OPT_LSTM_Q);
    if ((fq == 1 || fq == 2 || fq == 4) && a->F % (4 * fq) == 0 && a->HW <= 4 * (NT / fq)) Q = fq;
    const int rows = NT / Q;
    PPT = a->HW <= rows ? 1 : (a->HW <= 2 * rows ? 2 : 4);
    nslab = a->F / (4 * Q);
    xcd_map = (a->N % 8 == 0) ? 1 : 0;
    return true;
}

#define LSTM_FUSED_DISPATCH(KERNEL, ...)                                                                                   \
    do {                                                                                                                   \
        const dim3 grid((unsigned)(a->N * nslab));                                                                         \
        if (a->gates_bf16) {                                                                                               \
            if (Q == 1) { if (PPT == 1) KERNEL(1, 1, true, __VA_ARGS__); else if (PPT == 2) KERNEL(1, 2, true, __VA_ARGS__); else KERNEL(1, 4, true, __VA_ARGS__); } \
            else if (Q == 2) { if (PPT == 1) KERNEL(2, 1, true, __VA_ARGS__); else if (PPT == 2) KERNEL(2, 2, true, __VA_ARGS__); else KERNEL(2, 4, true, __VA_ARGS__); } \
            else { if (PPT == 1) KERNEL(4, 1, true, __VA_ARGS__); else if (PPT == 2) KERNEL(4, 2, true, __VA_ARGS__); else KERNEL(4, 4, true, __VA_ARGS__); } \
        } else {                                                                                                           \
            if (Q == 1) { if (PPT == 1) KERNEL(1, 1, false, __VA_ARGS__); else if (PPT == 2) KERNEL(1, 2, false, __VA_ARGS__); else KERNEL(1, 4, false, __VA_ARGS__); } \
            else if (Q == 2) { if (PPT == 1) KERNEL(2, 1, false, __VA_ARGS__); else if (PPT == 2) KERNEL(2, 2, false, __VA_ARGS__); else KERNEL(2, 4, false, __VA_ARGS__); } \
            else { if (PPT == 1) KERNEL(4, 1, false, __VA_ARGS__); else if (PPT == 2) KERNEL(4, 2, false, __VA_ARGS__); else KERNEL(4, 4, false, __VA_ARGS__); } \
        }                                                                                                                  \
    } while (0)
// (bench.py's roofline_cell: a launch armed with savp_prof_arm stamps its own begin / end, like the ring convolution's)
#define LSTM_FWD_LAUNCH(Q_, P_, G_, ...)                                                                                      \
    do {                                                                                                                   \
        if (g_savp_prof_start) {                                                                                           \
            hipExtLaunchKernelGGL((lstm_fused_fwd_kernel<Q_, P_, G_>), grid, dim3(NT), 0, st, g_savp_prof_start, g_savp_prof_stop, 0, __VA_ARGS__); \
            g_savp_prof_start = g_savp_prof_stop = nullptr;                                                                \
        } else hipLaunchKernelGGL((lstm_fused_fwd_kernel<Q_, P_, G_>), grid, dim3(NT), 0, st, __VA_ARGS__);                \
    } while (0)
#define LSTM_BWD_LAUNCH(Q_, P_, G_, ...) hipLaunchKernelGGL((lstm_fused_bwd_kernel<Q_, P_, G_>), grid, dim3(NT), 0, st, __VA_ARGS__)

static int fill_lstm(LstmP& p, const SavpLstmArgs* a) {
    if (!a || a->F % 4 || a->HW < 1) return SAVP_EINVAL;
    p.N = a->N; p.HW = a->HW; p.F = a->F;
    p.gates = (const float*)a->gates; p.gates16 = a->gates_bf16 ? 1 : 0; p.stats1_ready = a->stats1_ready ? 1 : 0;
    p.c_prev = (const float*)a->c_prev.p; p.cp_sn = a->c_prev.sn; p.cp_sp = a->c_prev.sp;
    p.g1 = a->gamma1; p.b1 = a->beta1; p.g2 = a->gamma2; p.b2 = a->beta2;
    p.eps = a->eps; p.forget_bias = a->forget_bias;
    p.c_new = a->c_new;
    p.nh = a->nh;
    if (a->nh < 0 || a->nh > 4 || a->ndh < 0 || a->ndh > 4) return SAVP_EINVAL;
    for (int i = 0; i < a->nh; ++i) { p.h[i] = (float*)a->h[i].p; p.h_sn[i] = a->h[i].sn; p.h_sp[i] = a->h[i].sp; p.h16[i] = (a->h_bf16 >> i) & 1; }
    p.mean1 = a->mean1; p.rstd1 = a->rstd1; p.mean2 = a->mean2; p.rstd2 = a->rstd2;
    p.ndh = a->ndh;
    for (int i = 0; i < a->ndh; ++i) { p.dh[i] = (const float*)a->dh[i].p; p.dh_sn[i] = a->dh[i].sn; p.dh_sp[i] = a->dh[i].sp; }
    p.dc_new = a->dc_new; p.dgates = a->dgates; p.dc_prev = a->dc_prev;
    p.dgates16 = a->dgates_bf16 ? 1 : 0;
    p.draw = a->dgates_bf16 ? a->dgates_raw : a->dgates;
    p.dg1 = a->dgamma1; p.db1 = a->dbeta1; p.dg2 = a->dgamma2; p.db2 = a->dbeta2;
    return SAVP_OK;
}

// workspace floats of the coalesced forward: ws1 [N][4F][2] + ws2 [N][F][2] + k2 [N][F] + sigmoid(o) [N][HW][F]
// reduction workspace: float64 sums -- forward s1 [N][4F][2] + s2 [N][F][2] (+ k2 [N][F] fp32), backward r2 [N][F][2] + r1 [N][4F][2]: N*F*22 floats
#define LSTM_RED_FLOATS 22
static long long lstm_ws_floats(const SavpLstmArgs* a) { return (long long)a->N * a->F * ((a->ws_stats ? 0 : LSTM_RED_FLOATS) + (long long)a->HW); }
static bool lstm_coalesced_ok(const SavpLstmArgs* a) {
    const int F = a->F;
    return a->ws && a->ws_floats >= lstm_ws_floats(a) && F >= 16 && F <= 256 && (F & (F - 1)) == 0 && a->HW >= 16;
}

// ------------------------------------------------------------------------------------------------------------
// The cell without a normaliser (SavpLstmArgs.no_norm): pointwise.  A thread owns 4 channels of one pixel; float4 accesses.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void lstm_plain_fwd_kernel(LstmP p) {
    const int F4 = p.F >> 2;
    const long long total = (long long)p.N * p.HW * F4;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int c0 = (int)(i % F4) * 4;
        const long long px = i / F4;                       // n * HW + pixel
        const int n = (int)(px / p.HW), q = (int)(px - (long long)n * p.HW);
        const float* g = p.gates + px * 4 * p.F + c0;
        const float4 gi = ld4(g), gj = ld4(g + p.F), gf = ld4(g + 2 * p.F), go = ld4(g + 3 * p.F);
        float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.c_prev) cp = ld4(p.c_prev + (long long)n * p.cp_sn + (long long)q * p.cp_sp + c0);
        const float iv[4] = {gi.x, gi.y, gi.z, gi.w}, jv[4] = {gj.x, gj.y, gj.z, gj.w}, fv[4] = {gf.x, gf.y, gf.z, gf.w},
                    ov[4] = {go.x, go.y, go.z, go.w}, cv[4] = {cp.x, cp.y, cp.z, cp.w};
        float cn[4], h[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            cn[c] = cv[c] * sigmoidf_(fv[c] + p.forget_bias) + sigmoidf_(iv[c]) * tanhf_(jv[c]);
            h[c] = tanhf_(cn[c]) * sigmoidf_(ov[c]);
        }
        st4(p.c_new + px * p.F + c0, make_float4(cn[0], cn[1], cn[2], cn[3]));
        for (int k = 0; k < p.nh; ++k)
            st4x(p.h[k], (long long)n * p.h_sn[k] + (long long)q * p.h_sp[k] + c0, make_float4(h[0], h[1], h[2], h[3]), p.h16[k]);
    }
}

__global__ __launch_bounds__(NT) void lstm_plain_bwd_kernel(LstmP p) {
    const int F4 = p.F >> 2;
    const long long total = (long long)p.N * p.HW * F4;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
        const int c0 = (int)(i % F4) * 4;
        const long long px = i / F4;
        const int n = (int)(px / p.HW), q = (int)(px - (long long)n * p.HW);
        const float* g = p.gates + px * 4 * p.F + c0;
        const float4 gi = ld4(g), gj = ld4(g + p.F), gf = ld4(g + 2 * p.F), go = ld4(g + 3 * p.F);
        float4 cp = make_float4(0.f, 0.f, 0.f, 0.f), dcn = cp, dh4 = cp;
        if (p.c_prev) cp = ld4(p.c_prev + (long long)n * p.cp_sn + (long long)q * p.cp_sp + c0);
        if (p.dc_new) dcn = ld4(p.dc_new + px * p.F + c0);
        for (int k = 0; k < p.ndh; ++k) {
            const float4 t = ld4(p.dh[k] + (long long)n * p.dh_sn[k] + (long long)q * p.dh_sp[k] + c0);
            dh4.x += t.x; dh4.y += t.y; dh4.z += t.z; dh4.w += t.w;
        }
        const float iv[4] = {gi.x, gi.y, gi.z, gi.w}, jv[4] = {gj.x, gj.y, gj.z, gj.w}, fv[4] = {gf.x, gf.y, gf.z, gf.w},
                    ov[4] = {go.x, go.y, go.z, go.w}, cv[4] = {cp.x, cp.y, cp.z, cp.w}, dhv[4] = {dh4.x, dh4.y, dh4.z, dh4.w},
                    dcv[4] = {dcn.x, dcn.y, dcn.z, dcn.w};
        float di[4], dj[4], df[4], dO[4], dcp[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float si = sigmoidf_(iv[c]), tj = tanhf_(jv[c]), sf = sigmoidf_(fv[c] + p.forget_bias), so = sigmoidf_(ov[c]);
            const float tc = tanhf_(cv[c] * sf + si * tj);
            const float dc = dhv[c] * so * (1.f - tc * tc) + dcv[c];
            di[c] = dc * tj * si * (1.f - si);
            dj[c] = dc * si * (1.f - tj * tj);
            df[c] = dc * cv[c] * sf * (1.f - sf);
            dO[c] = dhv[c] * tc * so * (1.f - so);
            dcp[c] = dc * sf;
        }
        float* d = p.dgates + px * 4 * p.F + c0;
        st4(d, make_float4(di[0], di[1], di[2], di[3])); st4(d + p.F, make_float4(dj[0], dj[1], dj[2], dj[3]));
        st4(d + 2 * p.F, make_float4(df[0], df[1], df[2], df[3])); st4(d + 3 * p.F, make_float4(dO[0], dO[1], dO[2], dO[3]));
        if (p.dc_prev) st4(p.dc_prev + px * p.F + c0, make_float4(dcp[0], dcp[1], dcp[2], dcp[3]));
    }
}

static int lstm_plain(void* stream, const SavpLstmArgs* a, const LstmP& p, bool fwd) {
    if (a->gates_bf16 || a->dgates_bf16 || a->stats1_ready || !a->gates || (((uintptr_t)a->gates) & 15)) return SAVP_EINVAL;
    if (fwd ? !a->c_new : !a->dgates) return SAVP_EINVAL;
    const long long total = (long long)a->N * a->HW * (a->F / 4);
    long long blocks = (total + NT - 1) / NT;
    if (blocks > 4096) blocks = 4096;
    if (fwd) hipLaunchKernelGGL(lstm_plain_fwd_kernel, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(lstm_plain_bwd_kernel, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
}

extern "C" int savp_convlstm_gates_fwd(void* stream, const SavpLstmArgs* a) {
    LstmP p;
    int rc = fill_lstm(p, a);
    if (rc) return rc;
    if (a->no_norm) return lstm_plain(stream, a, p, true);
    hipStream_t st = (hipStream_t)stream;
    {
        int Q, PPT, nslab, xcd_map;
        if (lstm_fused_cfg(a, true, Q, PPT, nslab, xcd_map)) {
            const double* s1 = a->stats1_ready ? (const double*)a->ws_stats : nullptr;
            LSTM_FUSED_DISPATCH(LSTM_FWD_LAUNCH, p, s1, nslab, xcd_map);
            return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
        }
    }
    if (lstm_coalesced_ok(a)) {
        const int N = a->N, F = a->F, HW = a->HW;
        LstmWs w;
        float* red = a->ws_stats ? a->ws_stats : a->ws;
        if (((uintptr_t)red) & 7) return SAVP_EINVAL;
        w.s1 = reinterpret_cast<double*>(red);
        w.s2 = w.s1 + (size_t)N * 4 * F * 2; w.k2 = reinterpret_cast<float*>(w.s2 + (size_t)N * F * 2);
        w.so = a->ws_stats ? a->ws : red + (size_t)N * F * LSTM_RED_FLOATS;
        if (a->stats1_ready && !a->ws_stats) return SAVP_EINVAL;
        if (!a->stats1_ready) {
        if (!(a->ws_stats && a->ws_stats_clean)) savp_zero_async(w.s1, (size_t)N * F * 20 * sizeof(float), st);
        if (a->gates_bf16) return SAVP_EINVAL;             // bf16 gates come with their statistics from the conv epilogue
        // pass 1: shifted sums of the gate tensor, the instance-norm statistics kernel with C = 4F
        InormP q;
        q.N = N; q.HW = HW; q.C = 4 * F;
        q.x = (const float*)a->gates; q.x_sn = (long long)HW * 4 * F; q.x_sp = 4 * F;
        const int rows1 = NT / F;                                   // C/4 = F float4 per pixel
        long long c1 = ((long long)HW * N + 511) / 512;
        if (c1 < rows1) c1 = rows1;
        if (c1 > 256) c1 = 256;
        q.chunk = (int)c1;
        hipLaunchKernelGGL(inorm_stats_kernel, dim3((HW + q.chunk - 1) / q.chunk, N), dim3(NT), (size_t)rows1 * 2 * 4 * F * sizeof(float), st,
                           q, w.s1);
        }
        const int rows2 = NT / (F / 4);
        long long c2 = ((long long)HW * N + 511) / 512;
        if (c2 < rows2) c2 = rows2;
        if (c2 > 256) c2 = 256;
        dim3 grid((HW + (int)c2 - 1) / (int)c2, N);
        hipLaunchKernelGGL(lstm_cell_kernel, grid, dim3(NT), (size_t)rows2 * 2 * F * sizeof(float), st, p, w, (int)c2);
        hipLaunchKernelGGL(lstm_out_kernel, grid, dim3(NT), 0, st, p, w, (int)c2);
        return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
    }
    if (a->HW > MAXPPT * NT || a->gates_bf16 || a->stats1_ready) return SAVP_EINVAL;
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(a->N * (a->F / 4)), dim3(NT), 0, st, p);
    return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
}

extern "C" int savp_convlstm_gates_bwd(void* stream, const SavpLstmArgs* a) {
    LstmP p;
    int rc = fill_lstm(p, a);
    if (rc) return rc;
    if (a->no_norm) return lstm_plain(stream, a, p, false);
    {
        int Q, PPT, nslab, xcd_map;
        if (lstm_fused_cfg(a, false, Q, PPT, nslab, xcd_map)) {
            hipStream_t st = (hipStream_t)stream;
            LSTM_FUSED_DISPATCH(LSTM_BWD_LAUNCH, p, nslab, xcd_map);
            return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
        }
    }
    if (a->dgates_bf16 && !a->dgates_raw) return SAVP_EINVAL;
    if (lstm_coalesced_ok(a)) {
        hipStream_t st = (hipStream_t)stream;
        const int N = a->N, F = a->F, HW = a->HW;
        LstmBws w;
        float* red = a->ws_stats ? a->ws_stats : a->ws;
        if (((uintptr_t)red) & 7) return SAVP_EINVAL;
        w.r2 = reinterpret_cast<double*>(red);
        w.r1 = w.r2 + (size_t)N * F * 2;
        w.dz2 = a->ws_stats ? a->ws : red + (size_t)N * F * LSTM_RED_FLOATS;
        if (!(a->ws_stats && a->ws_stats_clean)) savp_zero_async(w.r2, (size_t)N * F * 20 * sizeof(float), st);
        const int rows = NT / (F / 4);
        long long c = ((long long)HW * N + 511) / 512;
        if (c < rows) c = rows;
        if (c > 256) c = 256;
        dim3 grid((HW + (int)c - 1) / (int)c, N);
        hipLaunchKernelGGL(lstm_bwd1_kernel, grid, dim3(NT), (size_t)rows * 2 * F * sizeof(float), st, p, w, (int)c);
        hipLaunchKernelGGL(lstm_bwd2_kernel, grid, dim3(NT), (size_t)rows * 8 * F * sizeof(float), st, p, w, (int)c);
        hipLaunchKernelGGL(lstm_bwd3_kernel, grid, dim3(NT), 0, st, p, w, (int)c);
        return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
    }
    if (a->HW > MAXPPT * NT || a->gates_bf16 || a->dgates_bf16) return SAVP_EINVAL;
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(a->N * (a->F / 4)), dim3(NT), 0, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
}
