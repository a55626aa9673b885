// small_ops.hip -- the tiny-tensor ends of the SAVP graph, each fused to one launch:
//   lstm_z fwd/bwd   : tf.nn.rnn_cell.LSTMCell on z for ALL timesteps in one kernel (savp_model.py:354-362,426-432);
//                      the recurrence only involves z, so it is hoisted out of the per-frame loop.
//   reparam fwd/bwd  : clip(log_sigma_sq,-10,10); z = mu + sqrt(exp(ls))*eps; KL(mu,ls) (savp_model.py:45-49,711-712,
//                      losses.py:57-60) and their gradients.
//   l1/l2 loss       : losses.py:6-11 value + gradient in one pass.
//   lsgan loss       : losses.py:41-44 on the [B,1] logits.
//   cosine distance  : losses.py:14-22 feature-matching term, value + gradient w.r.t. the first argument.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "savp_hip.h"

#define NT 256
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH)

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) {
    float e = __expf(-2.f * fabsf(x));
    return copysignf((1.f - e) / (1.f + e), x);
}
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

// ---------------------------------------------------------------------------------------------------------------
// lstm_z.  zs [T,B,nz]; W [2nz,4nz] (rows: z then h; gate order i,j,f,o); b [4nz].  One workgroup per batch row,
// 4nz threads.  Saves pre-activation gates [T,B,4nz] and cell states c [T,B,nz].
// ---------------------------------------------------------------------------------------------------------------
#define MAXNZ 64
__global__ void lstm_z_fwd_kernel(const float* __restrict__ zs, const float* __restrict__ W, const float* __restrict__ bias,
                                  float* __restrict__ hout, float* __restrict__ gates, float* __restrict__ cs, int T, int B,
                                  int nz, float forget_bias, const float* __restrict__ c0, const float* __restrict__ h0) {
    __shared__ float sh_h[MAXNZ], sh_z[MAXNZ], sh_g[4 * MAXNZ];
    const int b = blockIdx.x, j = threadIdx.x;
    // initial state: zero, or the learned per-unit vectors c0 / h0 [nz] tiled over the batch (learn_initial_state, savp_model.py:295-307,344-352)
    float c = (c0 && j < nz) ? c0[j] : 0.f;
    if (j < nz) sh_h[j] = h0 ? h0[j] : 0.f;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if (j < nz) sh_z[j] = zs[((long long)t * B + b) * nz + j];
        __syncthreads();
        float g = bias[j];
        for (int i = 0; i < nz; ++i) g += sh_z[i] * W[i * 4 * nz + j] + sh_h[i] * W[(nz + i) * 4 * nz + j];
        sh_g[j] = g;
        gates[((long long)t * B + b) * 4 * nz + j] = g;
        __syncthreads();
        if (j < nz) {
            float gi = sh_g[j], gj = sh_g[nz + j], gf = sh_g[2 * nz + j], go = sh_g[3 * nz + j];
            c = sigm(gf + forget_bias) * c + sigm(gi) * tanh_(gj);
            float h = sigm(go) * tanh_(c);
            cs[((long long)t * B + b) * nz + j] = c;
            hout[((long long)t * B + b) * nz + j] = h;
            sh_h[j] = h;
        }
        __syncthreads();
    }
}

__global__ void lstm_z_bwd_kernel(const float* __restrict__ zs, const float* __restrict__ W, const float* __restrict__ hout,
                                  const float* __restrict__ gates, const float* __restrict__ cs, const float* __restrict__ dh_out,
                                  float* __restrict__ dzs, double* __restrict__ dW, double* __restrict__ db, int T, int B, int nz,
                                  float forget_bias, const float* __restrict__ c0, const float* __restrict__ h0, double* __restrict__ dc0,
                                  double* __restrict__ dh0) {
    __shared__ float sh_dg[4 * MAXNZ], sh_x[2 * MAXNZ], sh_dh[MAXNZ];
    const int b = blockIdx.x, j = threadIdx.x;
    float dWcol[2 * MAXNZ];
#pragma unroll
    for (int i = 0; i < 2 * MAXNZ; ++i) dWcol[i] = 0.f;
    float dbj = 0.f;
    float dc_next = 0.f;
    if (j < nz) sh_dh[j] = 0.f;           // dh carried from t+1
    // W in LDS (rows padded by one float: thread j reads row j), staged ONCE: inside the time loop the 4 nz global loads per step and thread were
    // a serial chain of L1 round trips -- 278 us for this one launch in the step (profiles/r06_kernel_stats.csv).  nz <= 32; larger: global.
    __shared__ float wsh[64 * 129];
    const bool w_lds = nz <= 32;
    const int wp = 4 * nz + 1;
    if (w_lds)
        for (int i = j; i < 2 * nz * 4 * nz; i += 4 * nz) wsh[(i / (4 * nz)) * wp + (i % (4 * nz))] = W[i];
    __syncthreads();
    // every global operand of step t is independent of the recurrence: the loads of step t - 1 are issued before step t's arithmetic and its
    // three barriers (one exposed L2 round trip per step was most of this launch: 228 us for 29 steps of a 32-wide cell)
    struct StepIn { float gi, gj, gf, go, c, cprev, dh, z, hprev; };
    auto load_step = [&](int t) {
        StepIn r = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (j < nz && t >= 0) {
            const long long o = (long long)t * B + b;
            r.gi = gates[o * 4 * nz + j]; r.gj = gates[o * 4 * nz + nz + j]; r.gf = gates[o * 4 * nz + 2 * nz + j];
            r.go = gates[o * 4 * nz + 3 * nz + j];
            r.c = cs[o * nz + j];
            r.cprev = t > 0 ? cs[(o - B) * nz + j] : (c0 ? c0[j] : 0.f);
            r.dh = dh_out[o * nz + j];
            r.z = zs[o * nz + j];
            r.hprev = t > 0 ? hout[(o - B) * nz + j] : (h0 ? h0[j] : 0.f);
        }
        return r;
    };
    StepIn cur = load_step(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        const long long o = (long long)t * B + b;
        const StepIn nxt = load_step(t - 1);
        if (j < nz) {
            float gi = cur.gi, gj = cur.gj, gf = cur.gf, go = cur.go;
            float c = cur.c;
            float cprev = cur.cprev;
            float dh = cur.dh + sh_dh[j];
            float so = sigm(go), tc = tanh_(c);
            float dc = dh * so * (1.f - tc * tc) + dc_next;
            float si = sigm(gi), tj = tanh_(gj), sf = sigm(gf + forget_bias);
            sh_dg[j] = dc * tj * si * (1.f - si);
            sh_dg[nz + j] = dc * si * (1.f - tj * tj);
            sh_dg[2 * nz + j] = dc * cprev * sf * (1.f - sf);
            sh_dg[3 * nz + j] = dh * tc * so * (1.f - so);
            dc_next = dc * sf;
            sh_x[j] = cur.z;
            sh_x[nz + j] = cur.hprev;
        }
        cur = nxt;
        __syncthreads();
        const float dgj = sh_dg[j];
        dbj += dgj;
#pragma unroll
        for (int i = 0; i < 2 * MAXNZ; ++i)
            if (i < 2 * nz) dWcol[i] += sh_x[i] * dgj;
        __syncthreads();
        // dx = W dgate : thread i < 2nz computes sum_j W[i][j] * dg[j]
        if (j < 2 * nz) {
            float s = 0.f;
            if (w_lds) { for (int q = 0; q < 4 * nz; ++q) s += wsh[j * wp + q] * sh_dg[q]; }
            else { for (int q = 0; q < 4 * nz; ++q) s += W[j * 4 * nz + q] * sh_dg[q]; }
            if (j < nz) dzs[o * nz + j] = s;
            else sh_dh[j - nz] = s;
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2 * MAXNZ; ++i)
        if (i < 2 * nz) unsafeAtomicAdd(dW + i * 4 * nz + j, dWcol[i]);
    unsafeAtomicAdd(db + j, dbj);
    // gradients of the learned initial state: what step 0 hands back, summed over the batch (the variables are tiled over it)
    if (j < nz) {
        if (dc0) unsafeAtomicAdd(dc0 + j, dc_next);
        if (dh0) unsafeAtomicAdd(dh0 + j, sh_dh[j]);
    }
}

extern "C" int savp_lstm_z_fwd(void* stream, const float* zs, const float* W, const float* bias, float* hout, float* gates,
                               float* cs, int32_t T, int32_t B, int32_t nz, float forget_bias) {
    if (!zs || !W || !bias || !hout || !gates || !cs || nz < 1 || nz > MAXNZ) return SAVP_EINVAL;
    hipLaunchKernelGGL(lstm_z_fwd_kernel, dim3(B), dim3(4 * nz), 0, (hipStream_t)stream, zs, W, bias, hout, gates, cs, T, B, nz,
                       forget_bias, (const float*)nullptr, (const float*)nullptr);
    return LAUNCH_OK();
}

extern "C" int savp_lstm_z_fwd_init(void* stream, const float* zs, const float* W, const float* bias, float* hout, float* gates,
                                    float* cs, int32_t T, int32_t B, int32_t nz, float forget_bias, const float* c0, const float* h0) {
    if (!zs || !W || !bias || !hout || !gates || !cs || nz < 1 || nz > MAXNZ) return SAVP_EINVAL;
    hipLaunchKernelGGL(lstm_z_fwd_kernel, dim3(B), dim3(4 * nz), 0, (hipStream_t)stream, zs, W, bias, hout, gates, cs, T, B, nz,
                       forget_bias, c0, h0);
    return LAUNCH_OK();
}

extern "C" int savp_lstm_z_bwd(void* stream, const float* zs, const float* W, const float* hout, const float* gates,
                               const float* cs, const float* dh_out, float* dzs, double* dW, double* db, int32_t T, int32_t B,
                               int32_t nz, float forget_bias) {
    if (!zs || !W || !hout || !gates || !cs || !dh_out || !dzs || !dW || !db || nz < 1 || nz > MAXNZ) return SAVP_EINVAL;
    hipLaunchKernelGGL(lstm_z_bwd_kernel, dim3(B), dim3(4 * nz), 0, (hipStream_t)stream, zs, W, hout, gates, cs, dh_out, dzs, dW,
                       db, T, B, nz, forget_bias, (const float*)nullptr, (const float*)nullptr, (double*)nullptr, (double*)nullptr);
    return LAUNCH_OK();
}

extern "C" int savp_lstm_z_bwd_init(void* stream, const float* zs, const float* W, const float* hout, const float* gates,
                                    const float* cs, const float* dh_out, float* dzs, double* dW, double* db, int32_t T, int32_t B,
                                    int32_t nz, float forget_bias, const float* c0, const float* h0, double* dc0, double* dh0) {
    if (!zs || !W || !hout || !gates || !cs || !dh_out || !dzs || !dW || !db || nz < 1 || nz > MAXNZ) return SAVP_EINVAL;
    hipLaunchKernelGGL(lstm_z_bwd_kernel, dim3(B), dim3(4 * nz), 0, (hipStream_t)stream, zs, W, hout, gates, cs, dh_out, dzs, dW,
                       db, T, B, nz, forget_bias, c0, h0, dc0, dh0);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// lstm_seq: tf.contrib.rnn.BasicLSTMCell under dynamic_rnn for ALL timesteps in one launch -- the recurrent encoder of
// posterior_fn (use_e_rnn, savp_model.py:31-43) and prior_fn (:66-76).  A [T,B,I+U]: columns [0,I) hold the inputs x_t
// (filled by the caller), columns [I,I+U) receive h_{t-1} (written here; zeros at t = 0), so that afterwards A is exactly
// the matrix whose transpose times dG is the kernel gradient.  W [I+U,4U] (rows: x then h; gate order i,j,f,o), bias [4U].
// One workgroup per batch row, 4U threads: thread j owns gate column j.  Saves the pre-activation gates [T,B,4U] and the
// cell states [T,B,U].
// bwd: dh_out [T,B,U] -> dG [T,B,4U] (gradient of the pre-activation gates) and dA [T,B,I+U] whose first I columns are
// dL/dx_t (the last U columns are scratch: dL/dh_{t-1} before the carry).  Weight / bias gradients are a GEMM over all
// (t,b) rows afterwards (A^T dG, column sums of dG): the caller runs them as one WGRAD launch.
// ---------------------------------------------------------------------------------------------------------------
#define MAXU 256
__global__ void lstm_seq_fwd_kernel(float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                    float* __restrict__ hout, float* __restrict__ gates, float* __restrict__ cs, int T, int B,
                                    int I, int U, float forget_bias) {
    extern __shared__ float sh_seq[];
    const int K = I + U, G = 4 * U;
    float* sh_a = sh_seq;              // [K]  current [x_t | h_{t-1}]
    float* sh_g = sh_seq + K;          // [G]
    const int b = blockIdx.x, j = threadIdx.x;
    float c = 0.f;
    for (int i = j; i < U; i += G) sh_a[I + i] = 0.f;
    for (int t = 0; t < T; ++t) {
        const long long o = (long long)t * B + b;
        for (int i = j; i < I; i += G) sh_a[i] = A[o * K + i];
        __syncthreads();
        float g0 = bias[j], g1 = 0.f, g2 = 0.f, g3 = 0.f;
        int i = 0;
        for (; i + 4 <= K; i += 4) {
            g0 += sh_a[i] * W[(long long)i * G + j];
            g1 += sh_a[i + 1] * W[(long long)(i + 1) * G + j];
            g2 += sh_a[i + 2] * W[(long long)(i + 2) * G + j];
            g3 += sh_a[i + 3] * W[(long long)(i + 3) * G + j];
        }
        for (; i < K; ++i) g0 += sh_a[i] * W[(long long)i * G + j];
        const float g = (g0 + g1) + (g2 + g3);
        sh_g[j] = g;
        gates[o * G + j] = g;
        __syncthreads();
        if (j < U) {
            const float gi = sh_g[j], gj = sh_g[U + j], gf = sh_g[2 * U + j], go = sh_g[3 * U + j];
            c = sigm(gf + forget_bias) * c + sigm(gi) * tanh_(gj);
            const float h = sigm(go) * tanh_(c);
            cs[o * U + j] = c;
            hout[o * U + j] = h;
            A[o * K + I + j] = sh_a[I + j];            // h_{t-1}: the recurrent half of row t
            sh_a[I + j] = h;
        }
        __syncthreads();
    }
}

__global__ void lstm_seq_bwd_kernel(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ gates,
                                    const float* __restrict__ cs, const float* __restrict__ dh_out, float* __restrict__ dG,
                                    float* __restrict__ dA, int T, int B, int I, int U, float forget_bias) {
    extern __shared__ float sh_seq[];
    const int K = I + U, G = 4 * U;
    float* sh_dg = sh_seq;             // [G]
    float* sh_dh = sh_seq + G;         // [U]  dL/dh_t carried from step t+1
    const int b = blockIdx.x, j = threadIdx.x;
    const int lane = j & 63, wave = j >> 6, nwaves = G >> 6;
    float dc_next = 0.f;
    if (j < U) sh_dh[j] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const long long o = (long long)t * B + b;
        if (j < U) {
            const float gi = gates[o * G + j], gj = gates[o * G + U + j], gf = gates[o * G + 2 * U + j], go = gates[o * G + 3 * U + j];
            const float c = cs[o * U + j];
            const float cprev = t > 0 ? cs[(o - B) * U + j] : 0.f;
            const float dh = dh_out[o * U + j] + sh_dh[j];
            const float so = sigm(go), tc = tanh_(c);
            const float dc = dh * so * (1.f - tc * tc) + dc_next;
            const float si = sigm(gi), tj = tanh_(gj), sf = sigm(gf + forget_bias);
            const float d0 = dc * tj * si * (1.f - si), d1 = dc * si * (1.f - tj * tj), d2 = dc * cprev * sf * (1.f - sf),
                        d3 = dh * tc * so * (1.f - so);
            sh_dg[j] = d0; sh_dg[U + j] = d1; sh_dg[2 * U + j] = d2; sh_dg[3 * U + j] = d3;
            dG[o * G + j] = d0; dG[o * G + U + j] = d1; dG[o * G + 2 * U + j] = d2; dG[o * G + 3 * U + j] = d3;
            dc_next = dc * sf;
        }
        __syncthreads();
        // dA[i] = sum_q W[i][q] dg[q]: one wave per row i, lanes stride the 4U columns (coalesced), butterfly reduction
        for (int i = wave; i < K; i += nwaves) {
            float s = 0.f;
            for (int q = lane; q < G; q += 64) s += W[(long long)i * G + q] * sh_dg[q];
            s = wsum(s);
            if (lane == 0) {
                dA[o * K + i] = s;
                if (i >= I) sh_dh[i - I] = s;
            }
        }
        __syncthreads();
    }
}

extern "C" int savp_lstm_seq_fwd(void* stream, float* A, const float* W, const float* bias, float* hout, float* gates, float* cs,
                                 int32_t T, int32_t B, int32_t I, int32_t U, float forget_bias) {
    if (!A || !W || !bias || !hout || !gates || !cs || U < 16 || U > MAXU || (U & 15) || I < 1 || I + U > 4096) return SAVP_EINVAL;
    hipLaunchKernelGGL(lstm_seq_fwd_kernel, dim3(B), dim3(4 * U), (size_t)(I + U + 4 * U) * sizeof(float), (hipStream_t)stream, A, W,
                       bias, hout, gates, cs, T, B, I, U, forget_bias);
    return LAUNCH_OK();
}

extern "C" int savp_lstm_seq_bwd(void* stream, const float* A, const float* W, const float* gates, const float* cs,
                                 const float* dh_out, float* dG, float* dA, int32_t T, int32_t B, int32_t I, int32_t U,
                                 float forget_bias) {
    if (!A || !W || !gates || !cs || !dh_out || !dG || !dA || U < 16 || U > MAXU || (U & 15) || I < 1 || I + U > 4096) return SAVP_EINVAL;
    hipLaunchKernelGGL(lstm_seq_bwd_kernel, dim3(B), dim3(4 * U), (size_t)(5 * U) * sizeof(float), (hipStream_t)stream, A, W, gates,
                       cs, dh_out, dG, dA, T, B, I, U, forget_bias);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// tf.contrib.rnn.GRUCell over all timesteps (rnn = 'gru': the latent's cell savp_model.py:358-359 and the encoders' recurrent tail :38-41):
//   [r, u] = sigmoid([x, h] Wg + bg) (r first);  c = tanh([x, r*h] Wc + bc);  h' = u*h + (1 - u)*c;  zero initial state.
// A [T,B,I+U]: x_t in columns [0,I) (caller), h_{t-1} in [I,I+U) (written by fwd).  fwd also leaves A2 = [x | r*h_{t-1}] (the candidate
// GEMM's input, needed for its weight gradient), ru [T,B,2U] (gate values), cand [T,B,U], hout [T,B,U].  One workgroup per batch row,
// 2U threads.  bwd produces dGg [T,B,2U] / dGc [T,B,U] (gradients of the two pre-activations) and dA [T,B,I+U] (first I columns =
// dL/dx); dWg = A^T dGg, dWc = A2^T dGc and the bias column sums are the caller's GEMMs (like savp_lstm_seq_*).
// ---------------------------------------------------------------------------------------------------------------
__global__ void gru_seq_fwd_kernel(float* __restrict__ A, float* __restrict__ A2, const float* __restrict__ Wg, const float* __restrict__ bg,
                                   const float* __restrict__ Wc, const float* __restrict__ bc, float* __restrict__ hout,
                                   float* __restrict__ ru, float* __restrict__ cand, int T, int B, int I, int U,
                                   const float* __restrict__ h0) {
    extern __shared__ float sh_gru[];
    const int K = I + U;
    float* sh_a = sh_gru;               // [K]  [x_t | h_{t-1}]
    float* sh_ru = sh_gru + K;          // [2U]
    float* sh_rh = sh_ru + 2 * U;       // [U]
    const int b = blockIdx.x, j = threadIdx.x;
    if (j < U) sh_a[I + j] = h0 ? h0[j] : 0.f;       // learn_initial_state: the variable, tiled over the batch (savp_model.py:344-352)
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const long long o = (long long)t * B + b;
        for (int i = j; i < I; i += 2 * U) sh_a[i] = A[o * K + i];
        __syncthreads();
        {
            float g = bg[j];
            for (int i = 0; i < K; ++i) g += sh_a[i] * Wg[(long long)i * 2 * U + j];
            const float v = sigm(g);
            sh_ru[j] = v;
            ru[o * 2 * U + j] = v;
        }
        __syncthreads();
        if (j < U) sh_rh[j] = sh_ru[j] * sh_a[I + j];
        for (int i = j; i < I; i += 2 * U) A2[o * K + i] = sh_a[i];
        __syncthreads();
        if (j < U) {
            float g = bc[j];
            for (int i = 0; i < I; ++i) g += sh_a[i] * Wc[(long long)i * U + j];
            for (int i = 0; i < U; ++i) g += sh_rh[i] * Wc[(long long)(I + i) * U + j];
            const float c = tanh_(g), u = sh_ru[U + j], hp = sh_a[I + j];
            const float h = u * hp + (1.f - u) * c;
            cand[o * U + j] = c;
            hout[o * U + j] = h;
            A[o * K + I + j] = hp;
            A2[o * K + I + j] = sh_rh[j];
        }
        __syncthreads();
        if (j < U) sh_a[I + j] = hout[o * U + j];
        __syncthreads();
    }
}

__global__ void gru_seq_bwd_kernel(const float* __restrict__ A, const float* __restrict__ Wg, const float* __restrict__ Wc,
                                   const float* __restrict__ ru, const float* __restrict__ cand, const float* __restrict__ dh_out,
                                   float* __restrict__ dGg, float* __restrict__ dGc, float* __restrict__ dA, int T, int B, int I, int U,
                                   double* __restrict__ dh0) {
    extern __shared__ float sh_gru[];
    const int K = I + U;
    float* sh_dgc = sh_gru;             // [U]
    float* sh_dgg = sh_dgc + U;         // [2U]
    float* sh_dxc = sh_dgg + 2 * U;     // [I]  dL/dx through the candidate
    float* sh_drh = sh_dxc + I;         // [U]  dL/d(r * h_prev)
    float* sh_dh = sh_drh + U;          // [U]  dL/dh_t carried from step t+1
    float* sh_dhp = sh_dh + U;          // [U]  direct part of dL/dh_{t-1}
    const int b = blockIdx.x, j = threadIdx.x;
    const int lane = j & 63, wave = j >> 6, nwaves = (2 * U + 63) >> 6;
    if (j < U) sh_dh[j] = 0.f;
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        const long long o = (long long)t * B + b;
        float hp = 0.f, r = 0.f, u = 0.f, du = 0.f, dhp = 0.f;
        if (j < U) {
            hp = A[o * K + I + j]; r = ru[o * 2 * U + j]; u = ru[o * 2 * U + U + j];
            const float c = cand[o * U + j];
            const float dh = dh_out[o * U + j] + sh_dh[j];
            du = dh * (hp - c);
            dhp = dh * u;
            const float dgc = dh * (1.f - u) * (1.f - c * c);
            sh_dgc[j] = dgc;
            dGc[o * U + j] = dgc;
        }
        __syncthreads();
        for (int i = wave; i < K; i += nwaves) {                    // [dx_c | d(r h)] = Wc dgc: one wave per row
            float s = 0.f;
            for (int q = lane; q < U; q += 64) s += Wc[(long long)i * U + q] * sh_dgc[q];
            s = wsum(s);
            if (lane == 0) { if (i < I) sh_dxc[i] = s; else sh_drh[i - I] = s; }
        }
        __syncthreads();
        if (j < U) {
            const float drh = sh_drh[j];
            const float dgr = drh * hp * r * (1.f - r), dgu = du * u * (1.f - u);
            sh_dgg[j] = dgr; sh_dgg[U + j] = dgu;
            dGg[o * 2 * U + j] = dgr; dGg[o * 2 * U + U + j] = dgu;
            sh_dhp[j] = dhp + drh * r;
        }
        __syncthreads();
        for (int i = wave; i < K; i += nwaves) {                    // [dx_g | dh_g] = Wg dgg
            float s = 0.f;
            for (int q = lane; q < 2 * U; q += 64) s += Wg[(long long)i * 2 * U + q] * sh_dgg[q];
            s = wsum(s);
            if (lane == 0) {
                if (i < I) dA[o * K + i] = s + sh_dxc[i];
                else { const float v = s + sh_dhp[i - I]; dA[o * K + i] = v; sh_dh[i - I] = v; }
            }
        }
        __syncthreads();
    }
    // gradient of the learned initial state: what step 0 hands back, summed over the batch (float64: one workgroup per sample adds)
    if (dh0 && j < U) unsafeAtomicAdd(dh0 + j, (double)sh_dh[j]);
}

extern "C" int savp_gru_seq_fwd(void* stream, float* A, float* A2, const float* Wg, const float* bg, const float* Wc, const float* bc,
                                float* hout, float* ru, float* cand, int32_t T, int32_t B, int32_t I, int32_t U) {
    if (!A || !A2 || !Wg || !bg || !Wc || !bc || !hout || !ru || !cand || U < 1 || U > 512 || I < 1 || I + U > 4096) return SAVP_EINVAL;
    const int nt = ((2 * U + 63) / 64) * 64;
    if (nt > 1024) return SAVP_EINVAL;
    // (threads beyond 2U would index past the gate arrays: the launch uses exactly 2U threads -- any count is legal, waves are padded)
    hipLaunchKernelGGL(gru_seq_fwd_kernel, dim3(B), dim3(2 * U), (size_t)(I + U + 3 * U) * sizeof(float), (hipStream_t)stream, A, A2, Wg, bg,
                       Wc, bc, hout, ru, cand, T, B, I, U, (const float*)nullptr);
    return LAUNCH_OK();
}

extern "C" int savp_gru_seq_fwd_init(void* stream, float* A, float* A2, const float* Wg, const float* bg, const float* Wc, const float* bc,
                                     float* hout, float* ru, float* cand, int32_t T, int32_t B, int32_t I, int32_t U, const float* h0) {
    if (!A || !A2 || !Wg || !bg || !Wc || !bc || !hout || !ru || !cand || !h0 || U < 1 || U > 512 || I < 1 || I + U > 4096) return SAVP_EINVAL;
    hipLaunchKernelGGL(gru_seq_fwd_kernel, dim3(B), dim3(2 * U), (size_t)(I + U + 3 * U) * sizeof(float), (hipStream_t)stream, A, A2, Wg, bg,
                       Wc, bc, hout, ru, cand, T, B, I, U, h0);
    return LAUNCH_OK();
}

extern "C" int savp_gru_seq_bwd(void* stream, const float* A, const float* Wg, const float* Wc, const float* ru, const float* cand,
                                const float* dh_out, float* dGg, float* dGc, float* dA, int32_t T, int32_t B, int32_t I, int32_t U) {
    if (!A || !Wg || !Wc || !ru || !cand || !dh_out || !dGg || !dGc || !dA || U < 1 || U > 512 || I < 1 || I + U > 4096) return SAVP_EINVAL;
    hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3(B), dim3(2 * U), (size_t)(6 * U + I) * sizeof(float), (hipStream_t)stream, A, Wg, Wc, ru,
                       cand, dh_out, dGg, dGc, dA, T, B, I, U, (double*)nullptr);
    return LAUNCH_OK();
}

extern "C" int savp_gru_seq_bwd_init(void* stream, const float* A, const float* Wg, const float* Wc, const float* ru, const float* cand,
                                     const float* dh_out, float* dGg, float* dGc, float* dA, int32_t T, int32_t B, int32_t I, int32_t U,
                                     double* dh0) {
    if (!A || !Wg || !Wc || !ru || !cand || !dh_out || !dGg || !dGc || !dA || !dh0 || U < 1 || U > 512 || I < 1 || I + U > 4096) return SAVP_EINVAL;
    hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3(B), dim3(2 * U), (size_t)(6 * U + I) * sizeof(float), (hipStream_t)stream, A, Wg, Wc, ru,
                       cand, dh_out, dGg, dGc, dA, T, B, I, U, dh0);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// KL between two diagonal Gaussians (losses.py:61-67; learn_prior: posterior (1) against the learned prior (2)):
//   kl = mean_rows sum_z [ (l2 - l1)/2 + (exp(l1) + (m1 - m2)^2) / (2 exp(l2)) - 1/2 ],   l = clip(ls_raw, -10, 10).
// kl_out (optional) += value; with dmu1 != null the gradient, times klw (host value or *klw_dev), is ADDED to the four
// gradient tensors (the clip passes no gradient outside [-10, 10], like tf.clip_by_value).
// ---------------------------------------------------------------------------------------------------------------
__global__ void kl_gauss_kernel(long long n, int rows, const float* mu1, const float* ls1_raw, const float* mu2, const float* ls2_raw,
                                double* kl_out, float klw_host, const float* klw_dev, float* dmu1, float* dls1, float* dmu2,
                                float* dls2) {
    __shared__ float sh[4];
    const float klw = klw_dev ? *klw_dev : klw_host;
    const float inv = 1.f / (float)rows;
    float acc = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float r1 = ls1_raw[i], r2 = ls2_raw[i];
        const float l1 = fminf(fmaxf(r1, -10.f), 10.f), l2 = fminf(fmaxf(r2, -10.f), 10.f);
        const float d = mu1[i] - mu2[i];
        // with x = l1 - l2:  (l2-l1)/2 + (e^l1 + d^2) / (2 e^l2) - 1/2  =  [ (e^x - 1 - x) + d^2 e^-l2 ] / 2.  Early in training the two
        // Gaussians nearly coincide (x ~ 1e-3): e^x - 1 - x is evaluated by its series there instead of by cancellation
        const float x = l1 - l2, ie2 = __expf(-l2);
        const float em1 = expm1f(x);
        const float g = fabsf(x) < 0.5f
            ? x * x * (0.5f + x * (1.f / 6.f + x * (1.f / 24.f + x * (1.f / 120.f + x * (1.f / 720.f + x * (1.f / 5040.f))))))
            : em1 - x;
        acc += 0.5f * (g + d * d * ie2);
        if (dmu1) {
            const float s = klw * inv;
            dmu1[i] += s * d * ie2;
            dmu2[i] -= s * d * ie2;
            if (r1 >= -10.f && r1 <= 10.f) dls1[i] += s * 0.5f * em1;
            if (r2 >= -10.f && r2 <= 10.f) dls2[i] -= s * 0.5f * (em1 + d * d * ie2);
        }
    }
    const float t = block_sum1(acc, sh);
    if (threadIdx.x == 0 && kl_out) unsafeAtomicAdd(kl_out, t * inv);
}

extern "C" int savp_kl_gauss(void* stream, int64_t n, int32_t rows, const float* mu1, const float* ls1_raw, const float* mu2,
                             const float* ls2_raw, double* kl_out, float klw, const float* klw_dev, float* dmu1, float* dls1,
                             float* dmu2, float* dls2) {
    if (!mu1 || !ls1_raw || !mu2 || !ls2_raw || rows < 1) return SAVP_EINVAL;
    if (dmu1 && (!dls1 || !dmu2 || !dls2)) return SAVP_EINVAL;
    unsigned nb = (unsigned)((n + NT - 1) / NT);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(kl_gauss_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, (long long)n, rows, mu1, ls1_raw, mu2, ls2_raw,
                       kl_out, klw, klw_dev, dmu1, dls1, dmu2, dls2);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// reparameterisation + KL.  n = T*B*nz elements; rows = T*B (the KL mean is over rows).
// fwd: ls = clip(ls_raw); z = mu + exp(0.5 ls)*eps; kl_out += -0.5*sum(1+ls-mu^2-exp(ls))/rows
// bwd: dmu = dz + klw*mu/rows ; dls_raw = [ls_raw in [-10,10]] * (dz*eps*0.5*exp(0.5 ls) - 0.5*klw*(1-exp(ls))/rows)
// ---------------------------------------------------------------------------------------------------------------
__global__ void reparam_fwd_kernel(long long n, int rows, const float* mu, const float* ls_raw, const float* eps, float* ls,
                                   float* z, double* kl_out) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float l = fminf(fmaxf(ls_raw[i], -10.f), 10.f);
        float m = mu[i];
        ls[i] = l;
        z[i] = m + __expf(0.5f * l) * eps[i];
        acc += 1.f + l - m * m - __expf(l);
    }
    float t = block_sum1(acc, sh);
    if (threadIdx.x == 0 && kl_out) unsafeAtomicAdd(kl_out, -0.5f * t / (float)rows);
}

__global__ void reparam_bwd_kernel(long long n, int rows, const float* mu, const float* ls_raw, const float* eps, const float* dz,
                                   float klw_host, float* dmu, float* dls_raw, const float* klw_dev) {
    const float klw = klw_dev ? *klw_dev : klw_host;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float lr = ls_raw[i];
        float l = fminf(fmaxf(lr, -10.f), 10.f);
        float g = dz ? dz[i] : 0.f;
        dmu[i] = g + klw * mu[i] / (float)rows;
        float d = g * eps[i] * 0.5f * __expf(0.5f * l) - 0.5f * klw * (1.f - __expf(l)) / (float)rows;
        dls_raw[i] = (lr >= -10.f && lr <= 10.f) ? d : 0.f;
    }
}

extern "C" int savp_reparam_fwd(void* stream, int64_t n, int32_t rows, const float* mu, const float* ls_raw, const float* eps,
                                float* ls, float* z, double* kl_out) {
    if (!mu || !ls_raw || !eps || !ls || !z) return SAVP_EINVAL;
    unsigned nb = (unsigned)((n + NT - 1) / NT);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, (long long)n, rows, mu, ls_raw, eps, ls, z,
                       kl_out);
    return LAUNCH_OK();
}

extern "C" int savp_reparam_bwd(void* stream, int64_t n, int32_t rows, const float* mu, const float* ls_raw, const float* eps,
                                const float* dz, float klw, float* dmu, float* dls_raw, const float* klw_dev) {
    if (!mu || !ls_raw || !eps || !dmu || !dls_raw) return SAVP_EINVAL;
    unsigned nb = (unsigned)((n + NT - 1) / NT);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, (long long)n, rows, mu, ls_raw, eps, dz,
                       klw, dmu, dls_raw, klw_dev);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// l1 / l2 image loss: loss_out += mean(|t-p|) or mean((t-p)^2); dpred += weight * dloss/dpred  (dpred may be null)
// ---------------------------------------------------------------------------------------------------------------
__global__ void lp_loss_kernel(long long rows, long long row_len, long long p_rs, long long t_rs, int p2, const float* pred,
                               const float* target, float weight, double* loss_out, float* dpred) {
    __shared__ float sh[4];
    float acc = 0.f;
    const long long n = rows * row_len;
    const float invn = 1.f / (float)n;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / row_len, e = i - r * row_len;
        float d = pred[r * p_rs + e] - target[r * t_rs + e];
        if (p2) {
            acc += d * d;
            if (dpred) dpred[r * p_rs + e] += weight * 2.f * d * invn;
        } else {
            acc += fabsf(d);
            // tf.abs gradient is sign(x) (0 at 0)
            if (dpred) dpred[r * p_rs + e] += weight * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * invn;
        }
    }
    float t = block_sum1(acc, sh);
    if (threadIdx.x == 0 && loss_out) unsafeAtomicAdd(loss_out, t * invn);
}

extern "C" int savp_lp_loss(void* stream, int64_t rows, int64_t row_len, int64_t pred_row_stride, int64_t target_row_stride,
                            int32_t p2, const float* pred, const float* target, float weight, double* loss_out, float* dpred) {
    // pred/dpred addressed as [rows][row_len] with row stride pred_row_stride (a half of a [T,2B,...] buffer)
    if (!pred || !target || rows < 1 || row_len < 1) return SAVP_EINVAL;
    long long n = (long long)rows * row_len;
    unsigned nb = (unsigned)((n + NT - 1) / NT);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(lp_loss_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, (long long)rows, (long long)row_len,
                       (long long)pred_row_stride, (long long)target_row_stride, p2, pred, target, weight, loss_out, dpred);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// Total variation of the predicted flows (base_model.py:763-769, tv_weight; transformation = 'flow'):
//   loss = mean_{t,b,y<H-1,x} sum_c |f[y+1,x,c] - f[y,x,c]| + mean_{t,b,y,x<W-1} sum_c |f[y,x+1,c] - f[y,x,c]|,  c over the 2 * nk flow channels.
// One launch per time step (the flow gradient of a step is complete only inside BPTT): s1 / s2 = 1 / (T B (H-1) W), 1 / (T B H (W-1)) of the WHOLE
// sequence; loss_out (float64) += this step's share; dflows += weight * d loss / d f  (tf.abs: sign(0) = 0).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgn_(float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); }
__global__ void tv_loss_kernel(const float* __restrict__ f, int n_img, int H, int W, int C, long long img_stride, long long px_stride, float s1,
                               float s2, float weight, double* loss_out, float* __restrict__ df) {
    __shared__ float sh[4];
    float acc = 0.f;
    const long long total = (long long)n_img * H * W * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const long long n = r / H;
        const float* q = f + n * img_stride + ((long long)y * W + x) * px_stride + c;
        const float v = *q;
        float g = 0.f;
        if (y + 1 < H) { const float d = q[(long long)W * px_stride] - v; acc += fabsf(d) * s1; g -= sgn_(d) * s1; }
        if (y > 0) g += sgn_(v - q[-(long long)W * px_stride]) * s1;
        if (x + 1 < W) { const float d = q[px_stride] - v; acc += fabsf(d) * s2; g -= sgn_(d) * s2; }
        if (x > 0) g += sgn_(v - q[-px_stride]) * s2;
        if (df) df[n * img_stride + ((long long)y * W + x) * px_stride + c] += weight * g;
    }
    const float t = block_sum1(acc, sh);
    if (threadIdx.x == 0 && loss_out) unsafeAtomicAdd(loss_out, (double)t);
}

extern "C" int savp_tv_loss(void* stream, const float* flows, int32_t n_img, int32_t H, int32_t W, int32_t C, int64_t img_stride, int64_t px_stride,
                            float s1, float s2, float weight, double* loss_out, float* dflows) {
    if (!flows || n_img < 1 || H < 2 || W < 2 || C < 1) return SAVP_EINVAL;
    const long long total = (long long)n_img * H * W * C;
    unsigned nb = (unsigned)((total + NT - 1) / NT);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(tv_loss_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, flows, n_img, H, W, C, (long long)img_stride,
                       (long long)px_stride, s1, s2, weight, loss_out, dflows);
    return LAUNCH_OK();
}

// GAN losses on logits [n] (losses.py:29-54).  type 0 LSGAN: mean((l-label)^2); 1 GAN: mean sigmoid cross-entropy with
// constant labels; 2 SNGAN: mean softplus(l) for label 0, mean softplus(-l) for label 1.
// loss_out += loss ; dlogits (=|+=) weight * dloss/dlogits
__device__ __forceinline__ float softplus_(float x) { return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x))); }
__global__ void gan_loss_kernel(int n, int type, const float* logits, float label, float weight, double* loss_out, float* dlogits,
                                int beta) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float l = logits[i];
        float g;
        if (type == 0) {
            const float d = l - label;
            acc += d * d;
            g = 2.f * d;
        } else if (type == 1) {
            // max(l,0) - l*z + log(1+exp(-|l|))   (tf.nn.sigmoid_cross_entropy_with_logits)
            acc += softplus_(l) - l * label;
            g = 1.f / (1.f + __expf(-l)) - label;
        } else {
            acc += (label == 0.f) ? softplus_(l) : softplus_(-l);
            g = 1.f / (1.f + __expf(-l)) - ((label == 0.f) ? 0.f : 1.f);
        }
        if (dlogits) {
            g *= weight / (float)n;
            dlogits[i] = beta ? dlogits[i] + g : g;
        }
    }
    float t = block_sum1(acc, sh);
    if (threadIdx.x == 0 && loss_out) unsafeAtomicAdd(loss_out, t / (float)n);
}

extern "C" int savp_gan_loss(void* stream, int32_t n, int32_t type, const float* logits, float label, float weight, double* loss_out,
                             float* dlogits, int32_t beta) {
    if (!logits || n < 1 || type < 0 || type > 2) return SAVP_EINVAL;
    if (type == 2 && label != 0.f && label != 1.f) return SAVP_EINVAL;
    hipLaunchKernelGGL(gan_loss_kernel, dim3(1), dim3(NT), 0, (hipStream_t)stream, n, type, logits, label, weight, loss_out, dlogits,
                       beta);
    return LAUNCH_OK();
}

extern "C" int savp_lsgan_loss(void* stream, int32_t n, const float* logits, float label, float weight, double* loss_out,
                               float* dlogits, int32_t beta) {
    return savp_gan_loss(stream, n, 0, logits, label, weight, loss_out, dlogits, beta);
}

// ---------------------------------------------------------------------------------------------------------------
// cosine feature distance over the channel axis: f0,f1 [P,C] contiguous.
// loss_out += mean_P( 0.5*|a-b|^2 ), a = f0/(|f0|+eps), b = f1/(|f1|+eps);  df0 (=|+=) weight * dloss/df0
// one wave per position.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void cosine_kernel(long long P, int C, const float* __restrict__ f0, const float* __restrict__ f1,
                                                    float weight, float eps, double* loss_out, float* df0, int beta) {
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float lacc = 0.f;
    for (long long pos = blockIdx.x * 4LL + wave; pos < P; pos += (long long)gridDim.x * 4) {
        const float* a = f0 + pos * C;
        const float* b = f1 + pos * C;
        float n0 = 0.f, n1 = 0.f;
        for (int c = lane; c < C; c += 64) { n0 += a[c] * a[c]; n1 += b[c] * b[c]; }
        n0 = sqrtf(wsum(n0)); n1 = sqrtf(wsum(n1));
        const float s0 = n0 + eps, s1 = n1 + eps;
        float dist = 0.f, dotag = 0.f;
        for (int c = lane; c < C; c += 64) {
            float d = a[c] / s0 - b[c] / s1;
            dist += d * d;
            dotag += a[c] * d;
        }
        dist = wsum(dist); dotag = wsum(dotag);
        if (lane == 0) lacc += 0.5f * dist;
        if (df0) {
            // g = (a_n - b_n) * weight / P ; df0 = g/s0 - f0 * (f0.g) / (n0 * s0^2)
            const float wp = weight / (float)P;
            const float coef = n0 > 0.f ? dotag / (n0 * s0 * s0) : 0.f;
            for (int c = lane; c < C; c += 64) {
                float d = a[c] / s0 - b[c] / s1;
                float g = wp * (d / s0 - a[c] * coef);
                df0[pos * C + c] = beta ? df0[pos * C + c] + g : g;
            }
        }
    }
    float t = block_sum1(lacc, sh);
    if (threadIdx.x == 0 && loss_out) unsafeAtomicAdd(loss_out, t / (float)P);
}

// Same computation with G = C/4 lanes per position (one float4 of each feature row per lane, rows read exactly once, reductions
// by xor-shuffles inside the lane group): the one-wave-per-position kernel above uses 32 of its 64 lanes on the widest feature map
// (C = 32, 655k positions) and re-reads the rows three times through dependent loops -- 224 us for 250 MB.
template <int G>
__global__ __launch_bounds__(NT) void cosine_vec_kernel(long long P, const float* __restrict__ f0, const float* __restrict__ f1,
                                                        float weight, float eps, double* loss_out, float* __restrict__ df0, int beta) {
    __shared__ float sh[4];
    constexpr int C = 4 * G, PW = 64 / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % G, grp = lane / G;
    float lacc = 0.f;
    const float wp = weight / (float)P;
    for (long long pos = (blockIdx.x * 4LL + wave) * PW + grp; pos < P; pos += (long long)gridDim.x * 4 * PW) {
        const float4 a = *reinterpret_cast<const float4*>(f0 + pos * C + 4 * sub);
        const float4 b = *reinterpret_cast<const float4*>(f1 + pos * C + 4 * sub);
        float n0 = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w, n1 = b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) { n0 += __shfl_xor(n0, o); n1 += __shfl_xor(n1, o); }
        n0 = sqrtf(n0); n1 = sqrtf(n1);
        const float s0 = n0 + eps, s1 = n1 + eps;
        const float4 d = make_float4(a.x / s0 - b.x / s1, a.y / s0 - b.y / s1, a.z / s0 - b.z / s1, a.w / s0 - b.w / s1);
        float dist = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w, dotag = a.x * d.x + a.y * d.y + a.z * d.z + a.w * d.w;
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) { dist += __shfl_xor(dist, o); dotag += __shfl_xor(dotag, o); }
        if (sub == 0) lacc += 0.5f * dist;
        if (df0) {
            const float coef = n0 > 0.f ? dotag / (n0 * s0 * s0) : 0.f;
            float4 g = make_float4(wp * (d.x / s0 - a.x * coef), wp * (d.y / s0 - a.y * coef), wp * (d.z / s0 - a.z * coef),
                                   wp * (d.w / s0 - a.w * coef));
            float* q = df0 + pos * C + 4 * sub;
            if (beta) { const float4 t = *reinterpret_cast<const float4*>(q); g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
            *reinterpret_cast<float4*>(q) = g;
        }
    }
    float t = block_sum1(lacc, sh);
    if (threadIdx.x == 0 && loss_out) unsafeAtomicAdd(loss_out, t / (float)P);
}

template <int G>
static void launch_cosine_vec(hipStream_t st, long long P, const float* f0, const float* f1, float weight, float eps, double* loss_out,
                              float* df0, int beta) {
    const long long per = 4 * (64 / G);
    unsigned nb = (unsigned)((P + per - 1) / per);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(cosine_vec_kernel<G>, dim3(nb), dim3(NT), 0, st, P, f0, f1, weight, eps, loss_out, df0, beta);
}

extern "C" int savp_cosine_distance(void* stream, int64_t P, int32_t C, const float* f0, const float* f1, float weight, float eps,
                                    double* loss_out, float* df0, int32_t beta) {
    if (!f0 || !f1 || P < 1 || C < 1) return SAVP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const bool al = (((uintptr_t)f0 | (uintptr_t)f1 | (uintptr_t)df0) & 15) == 0;
    if (al && C == 32) { launch_cosine_vec<8>(st, P, f0, f1, weight, eps, loss_out, df0, beta); return LAUNCH_OK(); }
    if (al && C == 64) { launch_cosine_vec<16>(st, P, f0, f1, weight, eps, loss_out, df0, beta); return LAUNCH_OK(); }
    if (al && C == 128) { launch_cosine_vec<32>(st, P, f0, f1, weight, eps, loss_out, df0, beta); return LAUNCH_OK(); }
    if (al && C == 256) { launch_cosine_vec<64>(st, P, f0, f1, weight, eps, loss_out, df0, beta); return LAUNCH_OK(); }
    unsigned nb = (unsigned)((P + 3) / 4);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(cosine_kernel, dim3(nb), dim3(NT), 0, st, (long long)P, C, f0, f1, weight, eps, loss_out,
                       df0, beta);
    return LAUNCH_OK();
}
