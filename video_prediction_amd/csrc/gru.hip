// gru.hip -- fused gate blocks of Conv2DGRUCell (rnn_ops.py:234-267, normaliser fused_instance_norm,
// separate_norms=False), forward and backward.  One workgroup = (sample n, 4 channels) over the whole plane
// (H*W <= 1024), so the instance-norm reductions are workgroup-local, like the ConvLSTM block in norm_lstm.hip.
//
//   gates stage : [r|u] = sigmoid(IN_{2F}(conv5x5([x, h])))            -> u saved, r*h written into the candidate conv's
//                                                                         input slot (the reference's quirk: the candidate
//                                                                         conv sees [x, h, r*h], rnn_ops.py:242,258)
//   output stage: c = tanh(IN_F(conv5x5([x, h, r*h]))) ; h' = u*h + (1-u)*c
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "savp_hip.h"

#define NT 256
#define MAXPPT 4
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
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
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) {
    float e = __expf(-2.f * fabsf(x));
    return copysignf((1.f - e) / (1.f + e), x);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

struct GruP {
    int N, HW, F;
    float eps;
    const float* pre;                                    // gates stage: [N,HW,2F]; output stage: [N,HW,F]
    const float* h; long long h_sn, h_sp;                // h_prev view (never null: zero state is a zero buffer)
    const float *gamma, *beta;
    float *mean, *rstd;                                  // [N,2F] / [N,F]
    float* u;                                            // [N,HW,F] contiguous (gates: written, output: read)
    float* rh; long long rh_sn, rh_sp;                   // gates fwd: r*h destination view
    int nout; float* out[4]; long long o_sn[4], o_sp[4]; // output fwd: h' destinations
    const float* hnew; long long hn_sn, hn_sp;           // output bwd: saved h' (any destination)
    // backward
    int ndy; const float* dy[4]; long long dy_sn[4], dy_sp[4];
    float* dpre;                                         // [N,HW,F] or [N,HW,2F]
    float* du;                                           // output bwd: d u (post-sigmoid) [N,HW,F]; gates bwd reads it
    float* dh; long long dh_sn, dh_sp;                   // accumulated (+=) gradient of h_prev
    const float* drh; long long drh_sn, drh_sp;          // gates bwd: gradient of the r*h slot
    double *dgamma, *dbeta;            // float64 accumulators (savp_hip.h SavpGruArgs)
};

// ---- gates stage forward: IN over 2F channels (this WG: channels c0..c0+3 of r and of u) ---------------------------
__global__ __launch_bounds__(NT) void gru_gates_fwd_kernel(GruP p) {
    __shared__ float sh[4 * 8];
    const int cg = p.F / 4, n = blockIdx.x / cg, c0 = (blockIdx.x % cg) * 4, F = p.F;
    const float* gp = p.pre + (long long)n * p.HW * 2 * F + c0;
    float4 gr[MAXPPT], gu[MAXPPT];
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            gr[t] = ld4(gp + (long long)px * 2 * F); gu[t] = ld4(gp + (long long)px * 2 * F + F);
            s[0] += gr[t].x; s[1] += gr[t].y; s[2] += gr[t].z; s[3] += gr[t].w;
            s[4] += gu[t].x; s[5] += gu[t].y; s[6] += gu[t].z; s[7] += gu[t].w;
        }
    }
    block_sum<8>(s, sh);
    const float inv = 1.f / (float)p.HW;
    float mu[8], rs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { mu[i] = s[i] * inv; s[i] = 0.f; }
#define SQ(a, m) (((a) - (m)) * ((a) - (m)))
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            s[0] += SQ(gr[t].x, mu[0]); s[1] += SQ(gr[t].y, mu[1]); s[2] += SQ(gr[t].z, mu[2]); s[3] += SQ(gr[t].w, mu[3]);
            s[4] += SQ(gu[t].x, mu[4]); s[5] += SQ(gu[t].y, mu[5]); s[6] += SQ(gu[t].z, mu[6]); s[7] += SQ(gu[t].w, mu[7]);
        }
    }
    block_sum<8>(s, sh);
#pragma unroll
    for (int i = 0; i < 8; ++i) rs[i] = rsqrtf(s[i] * inv + p.eps);
    if (threadIdx.x < 8) {
        const int q = threadIdx.x >> 2, c = threadIdx.x & 3;
        p.mean[(long long)n * 2 * F + q * F + c0 + c] = mu[threadIdx.x];
        p.rstd[(long long)n * 2 * F + q * F + c0 + c] = rs[threadIdx.x];
    }
    float ga[8], be[8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) { ga[q * 4 + c] = p.gamma[q * F + c0 + c]; be[q * 4 + c] = p.beta[q * F + c0 + c]; }
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            const float rv[4] = {gr[t].x, gr[t].y, gr[t].z, gr[t].w}, uv[4] = {gu[t].x, gu[t].y, gu[t].z, gu[t].w};
            const float4 h4 = ld4(p.h + (long long)n * p.h_sn + (long long)px * p.h_sp + c0);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w};
            float r[4], u[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                r[c] = sigm((rv[c] - mu[c]) * rs[c] * ga[c] + be[c]) * hv[c];
                u[c] = sigm((uv[c] - mu[4 + c]) * rs[4 + c] * ga[4 + c] + be[4 + c]);
            }
            st4(p.rh + (long long)n * p.rh_sn + (long long)px * p.rh_sp + c0, make_float4(r[0], r[1], r[2], r[3]));
            st4(p.u + ((long long)n * p.HW + px) * F + c0, make_float4(u[0], u[1], u[2], u[3]));
        }
    }
}

// ---- output stage forward ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gru_out_fwd_kernel(GruP p) {
    __shared__ float sh[4 * 4];
    const int cg = p.F / 4, n = blockIdx.x / cg, c0 = (blockIdx.x % cg) * 4, F = p.F;
    const float* gp = p.pre + (long long)n * p.HW * F + c0;
    float4 g[MAXPPT];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) { g[t] = ld4(gp + (long long)px * F); s[0] += g[t].x; s[1] += g[t].y; s[2] += g[t].z; s[3] += g[t].w; }
    }
    block_sum<4>(s, sh);
    const float inv = 1.f / (float)p.HW;
    float mu[4], rs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { mu[i] = s[i] * inv; s[i] = 0.f; }
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) { s[0] += SQ(g[t].x, mu[0]); s[1] += SQ(g[t].y, mu[1]); s[2] += SQ(g[t].z, mu[2]); s[3] += SQ(g[t].w, mu[3]); }
    }
    block_sum<4>(s, sh);
#pragma unroll
    for (int i = 0; i < 4; ++i) rs[i] = rsqrtf(s[i] * inv + p.eps);
    if (threadIdx.x < 4) {
        p.mean[(long long)n * F + c0 + threadIdx.x] = mu[threadIdx.x];
        p.rstd[(long long)n * F + c0 + threadIdx.x] = rs[threadIdx.x];
    }
    const float4 ga = ld4(p.gamma + c0), be = ld4(p.beta + c0);
    const float gav[4] = {ga.x, ga.y, ga.z, ga.w}, bev[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            const float gv[4] = {g[t].x, g[t].y, g[t].z, g[t].w};
            const float4 h4 = ld4(p.h + (long long)n * p.h_sn + (long long)px * p.h_sp + c0);
            const float4 u4 = ld4(p.u + ((long long)n * p.HW + px) * F + c0);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w}, uv[4] = {u4.x, u4.y, u4.z, u4.w};
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float cnd = tanh_((gv[c] - mu[c]) * rs[c] * gav[c] + bev[c]);
                o[c] = uv[c] * hv[c] + (1.f - uv[c]) * cnd;
            }
            const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
            for (int k = 0; k < p.nout; ++k) st4(p.out[k] + (long long)n * p.o_sn[k] + (long long)px * p.o_sp[k] + c0, o4);
        }
    }
}

// ---- output stage backward: dpre (candidate conv output), du, dh += u*dh' -------------------------------------------------
__global__ __launch_bounds__(NT) void gru_out_bwd_kernel(GruP p) {
    __shared__ float sh[4 * 8];
    const int cg = p.F / 4, n = blockIdx.x / cg, c0 = (blockIdx.x % cg) * 4, F = p.F;
    const float* gp = p.pre + (long long)n * p.HW * F + c0;
    float mu[4], rs[4], gav[4], bev[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        mu[c] = p.mean[(long long)n * F + c0 + c]; rs[c] = p.rstd[(long long)n * F + c0 + c];
        gav[c] = p.gamma[c0 + c]; bev[c] = p.beta[c0 + c];
    }
    float xh[MAXPPT][4], dz[MAXPPT][4];
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = 0.f;
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            const float4 g4 = ld4(gp + (long long)px * F);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
            const float4 h4 = ld4(p.h + (long long)n * p.h_sn + (long long)px * p.h_sp + c0);
            const float4 u4 = ld4(p.u + ((long long)n * p.HW + px) * F + c0);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w}, uv[4] = {u4.x, u4.y, u4.z, u4.w};
            float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < p.ndy; ++k) {
                float4 tt = ld4(p.dy[k] + (long long)n * p.dy_sn[k] + (long long)px * p.dy_sp[k] + c0);
                d4.x += tt.x; d4.y += tt.y; d4.z += tt.z; d4.w += tt.w;
            }
            const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
            float duv[4], dhv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float x = (gv[c] - mu[c]) * rs[c];
                const float cnd = tanh_(x * gav[c] + bev[c]);
                xh[t][c] = x;
                dz[t][c] = dv[c] * (1.f - uv[c]) * (1.f - cnd * cnd);
                duv[c] = dv[c] * (hv[c] - cnd);
                dhv[c] = dv[c] * uv[c];
                r[c] += dz[t][c]; r[4 + c] += dz[t][c] * x;
            }
            st4(p.du + ((long long)n * p.HW + px) * F + c0, make_float4(duv[0], duv[1], duv[2], duv[3]));
            // out_bwd OVERWRITES its dh destination (u * dh'); gates_bwd accumulates into its own
            st4(p.dh + (long long)n * p.dh_sn + (long long)px * p.dh_sp + c0, make_float4(dhv[0], dhv[1], dhv[2], dhv[3]));
        }
    }
    block_sum<8>(r, sh);
    if (threadIdx.x < 4) {
        unsafeAtomicAdd(p.dbeta + c0 + threadIdx.x, r[threadIdx.x]);
        unsafeAtomicAdd(p.dgamma + c0 + threadIdx.x, r[4 + threadIdx.x]);
    }
    const float inv = 1.f / (float)p.HW;
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = gav[c] * rs[c] * (dz[t][c] - r[c] * inv - xh[t][c] * r[4 + c] * inv);
            st4(p.dpre + ((long long)n * p.HW + px) * F + c0, make_float4(o[0], o[1], o[2], o[3]));
        }
    }
}

// ---- gates stage backward: dpre [N,HW,2F]; dh += d(rh)*r -----------------------------------------------------------------
__global__ __launch_bounds__(NT) void gru_gates_bwd_kernel(GruP p) {
    __shared__ float sh[4 * 16];
    const int cg = p.F / 4, n = blockIdx.x / cg, c0 = (blockIdx.x % cg) * 4, F = p.F;
    const float* gp = p.pre + (long long)n * p.HW * 2 * F + c0;
    float mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long o = (long long)n * 2 * F + q * F + c0 + c;
            mu[q * 4 + c] = p.mean[o]; rs[q * 4 + c] = p.rstd[o];
            ga[q * 4 + c] = p.gamma[q * F + c0 + c]; be[q * 4 + c] = p.beta[q * F + c0 + c];
        }
    float xh[MAXPPT][8], dg[MAXPPT][8];
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = 0.f;
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            const float4 a0 = ld4(gp + (long long)px * 2 * F), a1 = ld4(gp + (long long)px * 2 * F + F);
            const float raw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            const float4 h4 = ld4(p.h + (long long)n * p.h_sn + (long long)px * p.h_sp + c0);
            const float4 d4 = ld4(p.drh + (long long)n * p.drh_sn + (long long)px * p.drh_sp + c0);
            const float4 du4 = ld4(p.du + ((long long)n * p.HW + px) * F + c0);
            const float hv[4] = {h4.x, h4.y, h4.z, h4.w}, drhv[4] = {d4.x, d4.y, d4.z, d4.w}, duv[4] = {du4.x, du4.y, du4.z, du4.w};
            float dhv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                xh[t][c] = (raw[c] - mu[c]) * rs[c];
                xh[t][4 + c] = (raw[4 + c] - mu[4 + c]) * rs[4 + c];
                const float rr = sigm(xh[t][c] * ga[c] + be[c]);
                const float uu = sigm(xh[t][4 + c] * ga[4 + c] + be[4 + c]);
                dg[t][c] = drhv[c] * hv[c] * rr * (1.f - rr);
                dg[t][4 + c] = duv[c] * uu * (1.f - uu);
                dhv[c] = drhv[c] * rr;
            }
            float* dh = p.dh + (long long)n * p.dh_sn + (long long)px * p.dh_sp + c0;
            float4 o = ld4(dh);
            o.x += dhv[0]; o.y += dhv[1]; o.z += dhv[2]; o.w += dhv[3];
            st4(dh, o);
#pragma unroll
            for (int i = 0; i < 8; ++i) { r[i] += dg[t][i]; r[8 + i] += dg[t][i] * xh[t][i]; }
        }
    }
    block_sum<16>(r, sh);
    if (threadIdx.x < 8) {
        const int q = threadIdx.x >> 2, c = threadIdx.x & 3;
        unsafeAtomicAdd(p.dbeta + q * F + c0 + c, r[threadIdx.x]);
        unsafeAtomicAdd(p.dgamma + q * F + c0 + c, r[8 + threadIdx.x]);
    }
    const float inv = 1.f / (float)p.HW;
    float* dp = p.dpre + (long long)n * p.HW * 2 * F + c0;
#pragma unroll
    for (int t = 0; t < MAXPPT; ++t) {
        const int px = threadIdx.x + t * NT;
        if (px < p.HW) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = ga[i] * rs[i] * (dg[t][i] - r[i] * inv - xh[t][i] * r[8 + i] * inv);
            st4(dp + (long long)px * 2 * F, make_float4(o[0], o[1], o[2], o[3]));
            st4(dp + (long long)px * 2 * F + F, make_float4(o[4], o[5], o[6], o[7]));
        }
    }
}

static int fill_gru(GruP& p, const SavpGruArgs* a) {
    if (!a || a->F % 4 || a->HW < 1 || a->HW > MAXPPT * NT || !a->pre || !a->h.p) return SAVP_EINVAL;
    if (a->nout < 0 || a->nout > 4 || a->ndy < 0 || a->ndy > 4) return SAVP_EINVAL;
    p.N = a->N; p.HW = a->HW; p.F = a->F; p.eps = a->eps;
    p.pre = a->pre;
    p.h = (const float*)a->h.p; p.h_sn = a->h.sn; p.h_sp = a->h.sp;
    p.gamma = a->gamma; p.beta = a->beta; p.mean = a->mean; p.rstd = a->rstd;
    p.u = a->u;
    p.rh = (float*)a->rh.p; p.rh_sn = a->rh.sn; p.rh_sp = a->rh.sp;
    p.nout = a->nout;
    for (int i = 0; i < a->nout; ++i) { p.out[i] = (float*)a->out[i].p; p.o_sn[i] = a->out[i].sn; p.o_sp[i] = a->out[i].sp; }
    p.ndy = a->ndy;
    for (int i = 0; i < a->ndy; ++i) { p.dy[i] = (const float*)a->dy[i].p; p.dy_sn[i] = a->dy[i].sn; p.dy_sp[i] = a->dy[i].sp; }
    p.dpre = a->dpre; p.du = a->du;
    p.dh = (float*)a->dh.p; p.dh_sn = a->dh.sn; p.dh_sp = a->dh.sp;
    p.drh = (const float*)a->drh.p; p.drh_sn = a->drh.sn; p.drh_sp = a->drh.sp;
    p.dgamma = a->dgamma; p.dbeta = a->dbeta;
    return SAVP_OK;
}

#define GRU_ENTRY(name, kernel, cond)                                                                              \
    extern "C" int name(void* stream, const SavpGruArgs* a) {                                                      \
        GruP p;                                                                                                     \
        int rc = fill_gru(p, a);                                                                                    \
        if (rc) return rc;                                                                                          \
        if (!(cond)) return SAVP_EINVAL;                                                                            \
        hipLaunchKernelGGL(kernel, dim3(a->N * (a->F / 4)), dim3(NT), 0, (hipStream_t)stream, p);                   \
        return LAUNCH_OK();                                                                                         \
    }
GRU_ENTRY(savp_convgru_gates_fwd, gru_gates_fwd_kernel, p.u && p.rh && p.mean && p.rstd)
GRU_ENTRY(savp_convgru_out_fwd, gru_out_fwd_kernel, p.u && p.nout >= 1 && p.mean && p.rstd)
GRU_ENTRY(savp_convgru_out_bwd, gru_out_bwd_kernel, p.u && p.du && p.dpre && p.dh && p.ndy >= 1 && p.dgamma && p.dbeta)
GRU_ENTRY(savp_convgru_gates_bwd, gru_gates_bwd_kernel, p.du && p.dpre && p.dh && p.drh && p.dgamma && p.dbeta)
