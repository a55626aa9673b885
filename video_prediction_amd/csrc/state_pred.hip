// state_pred.hip -- the robot-state recurrence of the action / state-conditioned SAVP cell (savp_model.py:411-422, 655-658, 684-685):
//
//     state_t     = ground_truth[t] ? states_in[t] : gen_state_{t-1}            (gen_state_{-1} = 0: zero_state)
//     gen_state_t = [actions_t | state_t] . W + b                              ('state_pred/dense')
//
// gen_state depends on the actions and states only -- not on the images -- so the recurrence is hoisted out of the per-frame loop: ONE
// launch runs all T steps (one thread per sample; the vectors are a handful of floats), like the latent's LSTMCell (small_ops.hip).
// The cell's convolutions read [actions_t | stop_gradient(state_t)] beside the latent: `sa` [T, N, na + ns] is that block (the caller
// copies it into its tile source); the backward therefore only has the state-loss gradient to carry (base_model.py:758-762).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "savp_hip.h"

#define SP_MAX 32           // na + ns and ns are bounded by this (registers of one thread)
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH)

__global__ void state_pred_fwd_kernel(int T, int N, int na, int ns, const float* __restrict__ actions, const float* __restrict__ states_in,
                                      const int* __restrict__ gt, const float* __restrict__ W, const float* __restrict__ b,
                                      float* __restrict__ sa, float* __restrict__ gen) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int K = na + ns;
    float prev[SP_MAX];
#pragma unroll
    for (int j = 0; j < SP_MAX; ++j) prev[j] = 0.f;
    for (int t = 0; t < T; ++t) {
        const long long r = (long long)t * N + n;
        float in[SP_MAX];
        const bool g = gt[r] != 0;
#pragma unroll
        for (int i = 0; i < SP_MAX; ++i) {
            float v = 0.f;
            if (i < na) v = actions[r * na + i];
            else if (i < K) v = g ? states_in[r * ns + (i - na)] : prev[(i - na) & (SP_MAX - 1)];
            in[i] = v;
            if (i < K) sa[r * K + i] = v;
        }
#pragma unroll
        for (int j = 0; j < SP_MAX; ++j) {
            if (j < ns) {
                float acc = b[j];
                for (int i = 0; i < K; ++i) acc = fmaf(in[i], W[i * ns + j], acc);
                prev[j] = acc;
                gen[r * ns + j] = acc;
            }
        }
    }
}

// dgen [T, N, ns]: in = dL/dgen_state_t of the loss, out = the total gradient (with what the next step's state hands back when it took
// the prediction).  dW [(na+ns), ns], db [ns]: float64, added to by ONE workgroup in a fixed order (thread per entry, serial sums).
__global__ void state_pred_bwd_kernel(int T, int N, int na, int ns, const int* __restrict__ gt, const float* __restrict__ W,
                                      const float* __restrict__ sa, float* __restrict__ dgen, double* __restrict__ dW,
                                      double* __restrict__ db) {
    const int K = na + ns;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float carry[SP_MAX];                                // dL/dstate_{t+1} where step t+1 took gen_state_t
#pragma unroll
        for (int j = 0; j < SP_MAX; ++j) carry[j] = 0.f;
        for (int t = T - 1; t >= 0; --t) {
            const long long r = (long long)t * N + n;
            float g[SP_MAX];
#pragma unroll
            for (int j = 0; j < SP_MAX; ++j) {
                g[j] = 0.f;
                if (j < ns) {
                    g[j] = dgen[r * ns + j] + carry[j];
                    dgen[r * ns + j] = g[j];
                }
            }
            const bool took_pred = gt[r] == 0;              // state_t = gen_state_{t-1}: its gradient goes one step back
#pragma unroll
            for (int i = 0; i < SP_MAX; ++i) {
                float d = 0.f;
                if (i < ns && took_pred)
                    for (int j = 0; j < ns; ++j) d = fmaf(W[(na + i) * ns + j], g[j], d);
                carry[i] = d;
            }
        }
    }
    __syncthreads();
    const long long R = (long long)T * N;
    for (int e = threadIdx.x; e < K * ns + ns; e += blockDim.x) {
        double acc = 0.0;
        if (e < K * ns) {
            const int i = e / ns, j = e - i * ns;
            for (long long r = 0; r < R; ++r) acc += (double)(sa[r * K + i] * dgen[r * ns + j]);
            dW[e] += acc;
        } else {
            const int j = e - K * ns;
            for (long long r = 0; r < R; ++r) acc += (double)dgen[r * ns + j];
            db[j] += acc;
        }
    }
}

extern "C" int savp_state_pred_fwd(void* stream, int32_t T, int32_t N, int32_t na, int32_t ns, const float* actions, const float* states_in,
                                   const int32_t* gt, const float* W, const float* b, float* sa, float* gen) {
    if (T < 1 || N < 1 || na < 0 || ns < 1 || na + ns > SP_MAX || (na && !actions) || !states_in || !gt || !W || !b || !sa || !gen)
        return SAVP_EINVAL;
    hipLaunchKernelGGL(state_pred_fwd_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, T, N, na, ns, actions, states_in, gt,
                       W, b, sa, gen);
    return LAUNCH_OK();
}

extern "C" int savp_state_pred_bwd(void* stream, int32_t T, int32_t N, int32_t na, int32_t ns, const int32_t* gt, const float* W,
                                   const float* sa, float* dgen, double* dW, double* db) {
    if (T < 1 || N < 1 || na < 0 || ns < 1 || na + ns > SP_MAX || !gt || !W || !sa || !dgen || !dW || !db) return SAVP_EINVAL;
    hipLaunchKernelGGL(state_pred_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, T, N, na, ns, gt, W, sa, dgen, dW, db);
    return LAUNCH_OK();
}
