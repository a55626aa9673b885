/*
 * savp_hip.h -- C ABI of libsavp_hip.so, the MI355X (gfx950) kernel library behind the SAVP hot path.
 *
 * The reference (alexlee-gk/video_prediction) has no native plug-in point: its arithmetic is TensorFlow kernels
 * reached through Python.  Each entry below replaces the TF kernel call(s) of the cited reference lines.  All
 * entry points are stream-ordered, allocate nothing, keep no global state, and return 0 on success or a negative
 * SAVP_E* code.  Tensors are fp32, channels-last, addressed by raw device pointers + element strides
 * (channel stride is always 1), so channel-slice views of wider "concat" buffers are first-class.
 */
#ifndef SAVP_HIP_H
#define SAVP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAVP_OK 0
#define SAVP_EINVAL (-1)
#define SAVP_ELAUNCH (-2)

/* library / build info: returns a static string "savp_hip <version> gfx950" */
const char* savp_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32), 2-D and 3-D, NHWC / NDHWC.
 *
 * One geometric description of a forward cross-correlation  F: x[N,D,H,W,Cx] -> y[N,Do,Ho,Wo,Cy]
 *     y[n,od,oy,ox,cy] = sum_{a,u,v,cx} x[n, od*sd-pd+a, oy*sh-ph+u, ox*sw-pw+v, cx] * W[a,u,v,cx,cy]
 * and three modes over it:
 *   SAVP_CONV_FPROP : y  = F(x)            w = WT[Cy][kd*kh*kw*Cx]   (k-contiguous rows)
 *   SAVP_CONV_DGRAD : x  = F^T(y)          w = WD[Cx][kd*kh*kw*Cy]
 *   SAVP_CONV_WGRAD : dW += x (*) y        w = dW[kd*kh*kw*Cx][Cy]   (HWIO flattening, atomically accumulated)
 * Replaces: tf.nn.conv2d (ops.py:528, rnn_ops.py:121), tf.nn.conv2d_transpose (ops.py:584) = DGRAD mode,
 * tf.nn.conv3d (ops.py:773), tf.matmul in ops.dense (ops.py:12) = 1x1 FPROP, and their tf.gradients.
 * Epilogue (FPROP/DGRAD): v = acc + bias[c] + (beta ? old : 0);
 *   act 0: v; 1: lrelu(v, alpha); 2: sigmoid(v); 3: v * (aux>0 ? 1 : alpha)  [= backprop through lrelu/relu
 *   given the saved activation output `aux`, same view as the output].
 * ------------------------------------------------------------------------------------------------------------ */
enum { SAVP_CONV_FPROP = 0, SAVP_CONV_DGRAD = 1, SAVP_CONV_WGRAD = 2 };
enum { SAVP_ACT_NONE = 0, SAVP_ACT_LRELU = 1, SAVP_ACT_SIGMOID = 2, SAVP_ACT_DLRELU_FROM_OUT = 3 };

typedef struct SavpConvArgs {
    int32_t mode;
    int32_t N, D, H, W, Cx;        /* x-side tensor (input of F) */
    int32_t Do, Ho, Wo, Cy;        /* y-side tensor (output of F) */
    int32_t kd, kh, kw;
    int32_t sd, sh, sw;
    int32_t pd, ph, pw;            /* padding before (TF SAME: pad_total/2) */
    int32_t beta;                  /* FPROP/DGRAD: accumulate into the destination */
    int32_t act;
    float alpha;
    int32_t splitk;                /* WGRAD: number of K splits (>=1); 0 = pick automatically */
    int32_t tile;                  /* 0 = auto; else (WM<<4)|WN with tile = 64*WM x 64*WN */
    void* x; int64_t x_sn, x_sd, x_sh, x_sw;
    void* y; int64_t y_sn, y_sd, y_sh, y_sw;
    void* w;
    const float* bias;             /* per destination channel, or NULL */
    const float* aux;              /* act==3: saved activation, addressed like the destination */
} SavpConvArgs;

int savp_conv(void* stream, const SavpConvArgs* args);


/* A channels-last activation view addressed as p[n*sn + pixel*sp + c] (pixel = flattened D*H*W index; valid for
 * channel-slice views of contiguous buffers). */
typedef struct SavpView { void* p; int64_t sn; int64_t sp; } SavpView;

/* ------------------------------------------------------------------------------------------------------------
 * fused_instance_norm (+ ReLU / LeakyReLU), eps 1e-6, biased variance  (layers/normalization.py:146-170;
 * call sites savp_model.py:463-464,499-500,564-565,627-628; networks.py:26-27).
 * fwd: out[k] = act(gamma*(x-mean)/sqrt(var+eps)+beta) for k < nout (multi-destination so that concat buffers are
 *      filled without copy kernels); saves mean/rstd [N,C].
 * bwd: dy = sum_k dy[k]; masks by the saved activation output out[0]; writes dx (accumulates if dx_beta) and
 *      atomically accumulates dgamma/dbeta.   act: 0 none, 1 relu, 2 lrelu(alpha).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct SavpInormArgs {
    int32_t N, HW, C;
    int32_t act; float alpha; float eps;
    SavpView x;
    const float* gamma; const float* beta;
    int32_t nout; SavpView out[4];
    float* mean; float* rstd;
    int32_t ndy; SavpView dy[4];
    SavpView dx; int32_t dx_beta;
    float* dgamma; float* dbeta;
} SavpInormArgs;
int savp_instnorm_act_fwd(void* stream, const SavpInormArgs* a);
int savp_instnorm_act_bwd(void* stream, const SavpInormArgs* a);

/* ------------------------------------------------------------------------------------------------------------
 * Fused ConvLSTM gate block = everything in BasicConv2DLSTMCell.call after the convolution (rnn_ops.py:148-165,
 * normalizer fused_instance_norm, separate_norms=False, forget_bias 1.0):
 *   g = IN_{4F}(gates); i,j,f,o = split(g); c' = IN_F(c*sigmoid(f+fb) + sigmoid(i)*tanh(j)); h' = tanh(c')*sigmoid(o)
 * gates [N,HW,4F] contiguous; c_prev view (p may be NULL = zero state); c_new [N,HW,F] contiguous; h' is written to
 * nh destinations.  bwd consumes dh = sum_k dh[k] and dc_new (NULL = 0) and produces dgates, dc_prev (NULL = skip)
 * and atomically accumulated dgamma/dbeta of both norms.  HW <= 1024.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct SavpLstmArgs {
    int32_t N, HW, F;
    float eps, forget_bias;
    const float* gates;
    SavpView c_prev;
    const float *gamma1, *beta1, *gamma2, *beta2;
    float* c_new;
    int32_t nh; SavpView h[4];
    float *mean1, *rstd1, *mean2, *rstd2;
    int32_t ndh; SavpView dh[4];
    const float* dc_new;
    float* dgates;
    float* dc_prev;
    float *dgamma1, *dbeta1, *dgamma2, *dbeta2;
} SavpLstmArgs;
int savp_convlstm_gates_fwd(void* stream, const SavpLstmArgs* a);
int savp_convlstm_gates_bwd(void* stream, const SavpLstmArgs* a);

#ifdef __cplusplus
}
#endif
#endif /* SAVP_HIP_H */
