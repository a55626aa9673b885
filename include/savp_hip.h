/*
 * savp_hip.h -- C ABI of libsavp_hip.so, the MI355X (gfx950) kernel library behind the SAVP hot path.
 *
 * The reference (alexlee-gk/video_prediction) has no native plug-in point: its arithmetic is TensorFlow kernels
 * reached through Python.  Each entry below replaces the TF kernel call(s) of the cited reference lines.  All
 * entry points are stream-ordered, allocate no device memory (every scratch buffer is an argument the caller owns), never read
 * the environment, and return 0 on success or a negative SAVP_E* code.  The only process-wide state is the option table behind
 * savp_set_option (kernel-selection switches; defaults = the shipped configuration), the event pair of savp_prof_arm, and
 * one-time per-kernel attribute flags (hipFuncSetAttribute for > 64 KB of LDS).  Tensors are fp32, channels-last, addressed by
 * raw device pointers + element strides (channel stride is always 1), so channel-slice views of wider "concat" buffers are
 * first-class.
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

/* Kernel-selection switches (process-wide; SAVP_EINVAL for an unknown name).  Names and defaults: "conv_ring" 0 (auto algorithm
 * prefers the LDS-DMA ring kernel), "s2dgrad" 1, "thin" 1, "lstm_fused" 1, "ring_dma" 1 (problem-specific kernels / LDS-DMA patch staging of bf16 sources on), "ring_wwarm" 1 (the ring kernel's
 * workgroups pull their column tile's weight block into the XCD's L2 first), "wgp_dma" 1 (weight gradient of two bf16 operands: LDS-DMA
 * staging), "ring_early" 1 (ring kernel: the first patch is requested at the top of the prologue), "gate_kernel" 1 (the ConvLSTM gate convolution's own kernel
 * when SavpConvArgs.w_frag is given), "gate_cell" 1 (savp_convlstm_cell_fwd: the whole cell in one launch where a tile holds whole images), "gate_wwarm" 1 (its workgroups touch their column tile's weight block into the XCD's L2 first), "colsum_2stage" 1,
 * "inorm_min_hw" 64, and the developer overrides "wgp_cfg", "wgp_split", "lstm_q", "dense_legacy", "cdna_legacy", "gate_alt" (0).
 * "splitk_reduced" is a counter, not a switch: the number of convolution calls so far whose requested split-K count was cut (or dropped) because
 * SavpConvArgs.ws was absent or too small (savp_conv_workspace_bytes says how much a call can use); set it to 0 to reset. */
int savp_set_option(const char* name, int32_t value);
int savp_get_option(const char* name, int32_t* value);

/* Sum `count` fp32 elements of a flat gradient bucket in place over all replicas of `comm` (an ncclComm_t of RCCL, created by
 * the caller), ordered on `stream`: one collective per optimiser-group chunk instead of tf.contrib.nccl.all_sum per tensor
 * (utils/tf_utils.py:450-480, base_model.py:590-592,614-616).  The mean's 1/K goes into savp_adam's gscale.  librccl.so is
 * resolved on first use (SAVP_ELAUNCH if it cannot be loaded).  The repository's own runners hand their buckets to
 * torch.distributed (ProcessGroupNCCL = the same ncclAllReduce on the same pointer) because it also owns the rendezvous;
 * see INTEGRATION.md. */
int savp_allreduce_bucket(void* comm, void* stream, void* buf, int64_t count);

/* Measurement aid (bench.py's roofline leg; no reference counterpart): time ONE kernel by itself.  savp_prof_arm hands an event
 * pair to the next LDS-ring convolution launch of the calling thread (savp_conv, bf16 datapath); that launch stamps the events
 * with the dispatch's own begin / end -- the duration rocprofv3's kernel trace reports -- and disarms.  savp_prof_armed() == 1
 * afterwards means the convolution took another kernel and the pair is still waiting (disarm with savp_prof_arm(NULL, NULL)).
 * savp_prof_elapsed_us needs both events complete (synchronise the stream first). */
int savp_prof_event_create(void** ev);
int savp_prof_event_destroy(void* ev);
int savp_prof_arm(void* start, void* stop);
int savp_prof_armed(void);
int savp_prof_elapsed_us(void* start, void* stop, float* us);

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
enum { SAVP_PREC_F32 = 0, SAVP_PREC_BF16 = 1 };   /* multiply precision; accumulation is always fp32 */
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
    int32_t tile;                  /* 0 = auto; low byte (WM<<4)|WN with tile = 64*WM x 64*WN; bits 8-9 pick the FPROP/DGRAD
                                      algorithm: 0 auto, 1 generic gather kernel, 2 LDS patch kernel, 3 LDS-DMA ring kernel (EINVAL if
                                      not applicable); bit 10: patch / ring kernel with 8 waves; bits 12-13: LDS budget of the patch
                                      kernel (0 = 160 KB, 1 = 64 KB, 2 = 96 KB); ring kernel: bit 12 = wide weight slabs (8 / 9 k-steps per
                                      entry, tile 0x11 only) */
    int32_t precision;             /* SAVP_PREC_F32: exact fp32 MFMA; SAVP_PREC_BF16: operands rounded to bf16 in LDS */
    void* x; int64_t x_sn, x_sd, x_sh, x_sw;
    void* y; int64_t y_sn, y_sd, y_sh, y_sw;
    void* w;
    const float* bias;             /* FPROP/DGRAD: per destination channel, or NULL.  WGRAD: if not NULL the bias gradient is
                                      accumulated as well: bias[c] += sum over all pixels of y[..., c] (y pixel-contiguous) */
    const float* aux;              /* act==3: saved activation, addressed like the destination */
    const void* w_bf16;            /* optional bf16 copy of w (FPROP/DGRAD, SAVP_PREC_BF16): halves the weight stream */
    /* bf16 activations (SAVP_PREC_BF16, FPROP/DGRAD, ring kernel; EINVAL where it does not apply -- no silent fp32 fallback): */
    int32_t src_bf16;              /* the source tensor (x for FPROP and WGRAD, y for DGRAD) holds bf16; its strides count bf16 elements */
    int32_t out_bf16;              /* the destination receives bf16 (strides in bf16 elements; no bias / act / beta / split-K); WGRAD: the
                                      y (output-gradient) operand holds bf16.
                                      This is the ConvLSTM gate convolution (rnn_ops.py:121): the gate tensor makes its round trip to
                                      the gate kernels in half the bytes */
    double* stats;                 /* FPROP / DGRAD: [N][C_dst][2] FLOAT64, atomically accumulated sum / sum of squares over the pixels
                                      (per-workgroup fp32 partials folded in a fixed order inside the workgroup, then ONE float64 atomic per
                                      (sample, channel, workgroup): a float64 sum of fp32 partials is exact, so the result does not depend on
                                      the workgroups' arrival order -- two runs give the same bits)
                                      of every (sample, channel) of the destination's distance from the bias (y - bias = the fp32 accumulators,
                                      before any rounding: a large bias does not enter the one-pass variance; SavpInormArgs.stats_shift)
                                      = the statistics of the instance norm that follows (rnn_ops.py:148-149, normalization.py:146-170);
                                      caller zeroes; may be NULL.  bf16 precision only (the ring kernel); an fp32 destination additionally
                                      needs whole tiles (savp_conv_stats_ok() tells), no activation, no beta; SAVP_EINVAL otherwise */
    void* ws; int64_t ws_bytes;    /* optional caller-owned scratch (16-byte aligned; written before it is read, so one buffer can serve
                                      every call on a stream).  savp_conv_workspace_bytes() says how much a call can use; without it
                                      the call takes a kernel that needs none.  FPROP / DGRAD: split-K needs it -- split s stores its share of
                                      the destination block in slice s and one fold launch adds the slices in split order (deterministic; the
                                      splits used to meet in the destination through float atomics); without scratch the call runs unsplit,
                                      with too little it runs with as many splits as fit */
    int32_t dst_gap_at, dst_gap;   /* FPROP / DGRAD, SAVP_PREC_BF16 (ring kernel; SAVP_EINVAL elsewhere): the destination channel count
                                      (Cy for FPROP, Cx for DGRAD) counts LOGICAL channels; logical channel c >= dst_gap_at is physical
                                      channel c + dst_gap of the destination tensor, of the bias and of the packed weights (whose row
                                      count is the logical count + dst_gap); the gap's channels are not computed (they keep their contents, or are cleared
                                      with the rest of the block when the launch splits K).  dst_gap == 0: off.  This is the ConvLSTM gate convolution's
                                      data gradient without the tiled-z channels of its input [x | z | h] (rnn_ops.py:144-146,
                                      savp_model.py:436-444): their gradient is a per-sample sum (savp_tiled_z_grad) and leaving them
                                      out keeps the column count on a tile boundary (72 / 136 / 264 -> 64 / 128 / 256) */
    /* FPROP / DGRAD, SAVP_PREC_BF16, fp32 destination, unit strides (ring kernel; savp_conv_stats_ok() with nb_ws set tells whether the
       tiling can honour it, SAVP_EINVAL otherwise): the destination's logical channels [nb_c0, nb_c0 + nb_nc) are the output gradient
       dy of a fused_instance_norm + activation whose input is nb_x (layers/normalization.py:146-170 differentiated; the ConvLSTM layer's
       conv_pool / upsample convolution in front of the cell input's x slice, savp_model.py:449-464).  The epilogue then also leaves the
       two per-(sample, channel) sums that norm's backward needs -- sum(dy') and sum(dy' * xhat), dy' = dy * act'(gamma * xhat + beta),
       xhat = (x - mean) * rstd -- in nb_ws [N][nb_nc][2] (float64, caller zeroes, atomically accumulated: exact, order-independent), so that
       savp_instnorm_act_bwd(stats_ready) runs its apply pass alone. */
    const float* nb_x; int64_t nb_x_sn, nb_x_sp;     /* the norm's input [N][pixels][nb_nc], addressed like a SavpView (pixel = y * W + x) */
    const float *nb_mean, *nb_rstd;                  /* [N][nb_nc] saved by the forward pass */
    const float *nb_gamma, *nb_beta;                 /* [nb_nc] */
    double* nb_ws;                                   /* [N][nb_nc][2] FLOAT64 (see `stats`) */
    int32_t nb_c0, nb_nc, nb_act; float nb_alpha;    /* nb_act: 0 none, 1 relu, 2 leaky relu (nb_alpha) */
    /* Round 6, the ConvLSTM gate convolution's own kernel (conv_gate.hip; rnn_ops.py:115-126,143): the weights once more, in MFMA B-fragment
       order (savp_pack_gate_weights; savp_gate_weights_bytes of them, 16-byte aligned).  Given it, a 2-D 5x5 stride-1 SAME FPROP between dense
       bf16 tensors with `stats` (bias / act / beta / aux none, Cy % 128 == 0, H == W and (H, Cx) one of the instantiated shapes) takes that
       kernel whatever `tile` says (option "gate_kernel" 1); NULL or any other problem: the general kernels, as before. */
    const void* w_frag;
    /* ... and in the INTERLEAVED column order (savp_pack_gate_weights(interleave = 1): local column l of a 32-column block = gate l & 3 of channel
       l >> 2).  Only savp_convlstm_cell_fwd reads it: given it, the 16 x 16 and 8 x 8 layers' whole cell -- convolution, IN(4F), gates, IN(F), h --
       is ONE launch (option "gate_cell" 1); savp_conv ignores it. */
    const void* w_frag_il;
} SavpConvArgs;

int savp_conv(void* stream, const SavpConvArgs* args);

/* Gradient of a latent vector that is TILED over the plane and concatenated into a stride-1 SAME convolution's input -- the nz
 * channels [z0, z0 + nz) of the ConvLSTM gate convolution's input [x | tile(z) | h] (rnn_ops.py:144-146 through tile_concat,
 * savp_model.py:436-444) -- from the convolution's OUTPUT gradient alone, so that the per-pixel data gradient can leave those
 * channels out (SavpConvArgs.dst_gap):  dz[img, c] = sum_p (F^T dy)[img, p, z0 + c]  computed as
 * sum_{25 regions r} sum_k R[img, r, k] * weff[r, k, c]  with R = region sums of dy (row class x column class, classes
 * {0, 1, middle, n-2, n-1}) -- tests/test_tiled_z_gradient_algebra.py pins the identity against autograd.
 *   savp_tiled_z_weff: weff [25][Cout][P] fp32 from the HWIO kernel w [kh][kw][Cin][Cout] (kh, kw <= 5, pads <= 2; P = 8 for nz <= 8, 32 for
 *                      8 < nz <= 32; once per step and layer).
 *   savp_tiled_z_grad: dy [nimg][H][W][C] contiguous (bf16 if dy_bf16, else fp32; 16-byte aligned), one launch for the gate
 *                      gradients of all timesteps; dz [nimg][nz] fp32 is overwritten (beta 0) or accumulated into (beta 1).
 *                      H >= 4, W in {4, 8, 16, 32}, C % 64 == 0; SAVP_EINVAL otherwise.  Two launches (per-chunk partials into ws, then
 *                      their sum in chunk order).  Deterministic (fixed-order reductions). */
int savp_tiled_z_weff(void* stream, const float* w, int32_t kh, int32_t kw, int32_t ph, int32_t pw, int32_t Cin, int32_t Cout,
                      int32_t z0, int32_t nz, float* weff);
int savp_tiled_z_grad(void* stream, const void* dy, int32_t dy_bf16, int64_t nimg, int32_t H, int32_t W, int32_t C,
                      const float* weff, int32_t nz, float* dz, int32_t beta, void* ws, int64_t ws_bytes);
/* bytes of the caller-owned scratch `ws` of savp_tiled_z_grad (per-(image, 64-channel chunk) partial sums; written before it is read) */
int64_t savp_tiled_z_workspace_bytes(int64_t nimg, int32_t C);
/* 1: savp_conv would honour args->stats (any non-NULL value) for this problem; 0: it would return SAVP_EINVAL -- the caller then
 * leaves stats NULL and lets the instance norm take its own statistics.  No launch, no device access. */
int savp_conv_stats_ok(const SavpConvArgs* args);
/* bytes of args->ws this call would use (0: none): the RGB-side weight gradient's per-workgroup partial sums; FPROP / DGRAD: the split-K
 * slices (args->splitk > 1: exactly that many; 0 = automatic: an upper bound of what the library's heuristic can choose) */
int64_t savp_conv_workspace_bytes(const SavpConvArgs* args);
/* 1 when, with tile bits 8-9 == 0 (automatic algorithm), a problem-specific kernel takes this call and tile / splitk are not
 * looked at (a tuner can skip its search); 0 otherwise */
int savp_conv_special(const SavpConvArgs* args);


/* A channels-last activation view addressed as p[n*sn + pixel*sp + c] (pixel = flattened D*H*W index; valid for
 * channel-slice views of contiguous buffers). */
typedef struct SavpView { void* p; int64_t sn; int64_t sp; } SavpView;

/* ------------------------------------------------------------------------------------------------------------
 * fused_instance_norm (+ ReLU / LeakyReLU), eps 1e-6, biased variance  (layers/normalization.py:146-170;
 * call sites savp_model.py:463-464,499-500,564-565,627-628; networks.py:26-27).
 * fwd: out[k] = act(gamma*(x-mean)/sqrt(var+eps)+beta) for k < nout (multi-destination so that concat buffers are
 *      filled without copy kernels); saves mean/rstd [N,C].
 * bwd: dy = sum_k dy[k]; masks by the saved activation output out[0]; writes dx (accumulates if dx_beta) and
 *      atomically accumulates dgamma/dbeta.   act: 0 none, 1 relu, 2 lrelu(alpha).  The activation mask is recomputed from
 *      x, mean, rstd, gamma, beta (y > 0 <=> z > 0): bwd does not read the saved output (`out` is ignored).
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
    double* dgamma; double* dbeta; /* FLOAT64 accumulators [C] (8-byte aligned), atomically added to: a sum of fp32 partials is exact in float64, so the
                                      parameter gradients do not depend on the workgroups' arrival order (round 6); the caller folds them into its fp32 gradients */
    void* ws;                      /* optional scratch [N*C*2] FLOAT64 (8-byte aligned; sums of fp32 partials are exact there, so the statistics do not
                                      depend on the workgroups' arrival order): selects the coalesced two-kernel path for planes of >= "inorm_min_hw" (64) pixels */
    int32_t ws_clean;              /* 1: the caller guarantees ws is all zero (e.g. a slice of an arena cleared once per step),
                                      0: the library clears it with a memset per call */
    int32_t out_c0[4], out_nc[4];  /* fwd: output k receives channels [out_c0, out_c0 + out_nc) of the normalised tensor, stored from
                                      channel 0 of its view (multiples of 4; out_nc == 0: all C channels) -- one launch can normalise
                                      the concatenated output of two convolutions that feed different consumers */
    int32_t dy_c0[4], dy_nc[4];    /* bwd: gradient k covers channels [dy_c0, dy_c0 + dy_nc) (dy_nc == 0: all C) */
    int32_t out_bf16;              /* fwd: bit k set = output view k is a bf16 tensor (its strides count bf16 elements): a destination
                                      that only feeds convolutions of the bf16 datapath (they round to bf16 anyway) in half the bytes */
    int32_t stats_ready;           /* fwd: ws already holds the per-(sample, channel) sum / sum of squares of x, UNSHIFTED (savp_conv's
                                      `stats` epilogue wrote them while it produced x): the statistics pass is skipped -> one launch.
                                      bwd: ws already holds sum(dy'), sum(dy' * xhat) per (sample, channel) (savp_conv's nb_ws epilogue wrote
                                      them while it produced dy): the statistics pass is skipped -> one launch */
    const float* stats_shift;      /* fwd with stats_ready: per-channel value the producer took its sums around -- savp_conv's `stats` epilogue
                                      sums the accumulators WITHOUT the bias, so this is that convolution's bias (NULL: sums around 0) */
    int32_t dx_bf16;               /* bwd: dx is a bf16 tensor (strides in bf16 elements, multiples of 4; dx_beta must be 0): the gradient of
                                      a convolution's output, whose only readers are that convolution's DGRAD / WGRAD on the bf16
                                      datapath (they round it to bf16 when they stage it anyway) -- half the bytes, identical numbers */
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
    const void* gates;
    SavpView c_prev;
    const float *gamma1, *beta1, *gamma2, *beta2;
    float* c_new;
    int32_t nh; SavpView h[4];
    float *mean1, *rstd1, *mean2, *rstd2;
    int32_t ndh; SavpView dh[4];
    const float* dc_new;
    float* dgates;
    float* dc_prev;
    double *dgamma1, *dbeta1, *dgamma2, *dbeta2;   /* FLOAT64 accumulators [4F], [4F], [F], [F] (see SavpInormArgs.dgamma) */
    float* ws; int64_t ws_floats;  /* optional workspace, >= N*F*(22 + HW) floats (N*F*HW if ws_stats is given; 8-byte aligned): selects the
                                      coalesced three-pass kernels (F a power of two in [16, 256]); NULL = single fused
                                      kernel (HW <= 1024) */
    float* ws_stats;               /* optional separate reduction workspace, N*F*22 floats, 8-byte aligned: float64 sums (forward: [N][4F][2] of the
                                      gate tensor, then [N][F][2] of c_pre, then N*F fp32 shifts; backward: [N][F][2], [N][4F][2]) */
    int32_t ws_stats_clean;        /* 1: the caller guarantees ws_stats is all zero (see SavpInormArgs.ws_clean) */
    int32_t gates_bf16;            /* `gates` holds bf16 (written by savp_conv with out_bf16); coalesced kernels only */
    int32_t stats1_ready;          /* fwd: the first N*4F*2 float64 of ws_stats already hold the per-(sample, gate channel) sum / sum of
                                      squares of the gate tensor (savp_conv's `stats` epilogue; the rest zero): the
                                      statistics pass over the gates is skipped -> conv + 2 launches per ConvLSTM cell */
    int32_t h_bf16;                /* fwd: bit k set = destination k of h' is a bf16 tensor (strides in bf16 elements) */
    int32_t dgates_bf16;           /* bwd: `dgates` receives bf16 (its readers are the gate convolution's DGRAD / WGRAD, which round to
                                      bf16 anyway); the raw gate gradients between the passes then live in dgates_raw.  Coalesced kernels only */
    float* dgates_raw;             /* bwd, with dgates_bf16: fp32 scratch [N,HW,4F] (three-pass kernels only) */
    int32_t no_norm;               /* 1: the cell WITHOUT a normaliser (conv_rnn_norm_layer = 'none' / ablation_conv_rnn_norm, rnn_ops.py:122-125,
                                      148-165 with normalizer_fn = None): `gates` holds conv + bias (fp32), i, j, f, o = split(gates);
                                      c' = c*sigmoid(f+fb) + sigmoid(i)*tanh(j); h' = tanh(c')*sigmoid(o) -- pointwise, no statistics, the
                                      norm parameters / mean / rstd arguments are not read */
} SavpLstmArgs;
int savp_convlstm_gates_fwd(void* stream, const SavpLstmArgs* a);
int savp_convlstm_gates_bwd(void* stream, const SavpLstmArgs* a);

/* ------------------------------------------------------------------------------------------------------------
 * HBM-bound glue (util_ops.hip).  See the file header for the reference lines each one replaces.
 * ------------------------------------------------------------------------------------------------------------ */
/* out[r,p,c] (=|+=) scale*z[r,c]  : tile_concat broadcast (ops.py:968-1006) / backward of global average pool */
int savp_tile_channels(void* stream, const float* z, int64_t R, int32_t HW, int32_t C, float scale, SavpView out, int32_t beta);
/* the same into a bf16 view (strides in bf16 elements; overwrite) */
int savp_tile_channels_bf16(void* stream, const float* z, int64_t R, int32_t HW, int32_t C, float scale, SavpView out);
/* out[(r,)c] += scale*sum_p in[r,p,c] (atomic accumulate; per_row keeps r) : bias grads, d(tile_concat), avg pool.
 * ws: optional caller-owned scratch of >= SAVP_COLSUM_WS_FLOATS floats (16-byte aligned, written before read): large all-pixel sums
 * then go through partial rows + one reduce launch instead of ~10^4 workgroups of atomics onto a few cache lines */
#define SAVP_COLSUM_WS_FLOATS (1024 * 4 * 256)
int savp_colsum(void* stream, SavpView in, int64_t R, int32_t HW, int32_t C, float scale, float* out, int32_t per_row,
                float* ws, int64_t ws_floats);
/* out_k = mask[n] ? a : b   (scheduled sampling tf.where, savp_model.py:406); b.p may be NULL (zeros) */
int savp_select(void* stream, int32_t N, int32_t HW, int32_t C, const int32_t* mask, SavpView a, SavpView b, int32_t nout,
                const SavpView* outs);
/* db += mask[n] ? 0 : sum_k din_k */
int savp_select_bwd(void* stream, int32_t N, int32_t HW, int32_t C, const int32_t* mask, int32_t nin, const SavpView* dins,
                    SavpView db);
/* dst[b,i,:] = src[(t_start[b]+i)*src_t_stride + b*E + :] (tf.gather_nd, savp_model.py:97-102); adjoint: src += dst */
int savp_gather_clips(void* stream, float* src, float* dst, const int32_t* t_start, int32_t B, int32_t clip, int64_t E,
                      int64_t src_t_stride, int32_t adjoint);
/* out = dy * y * (1-y) (contiguous out): backward of the sigmoid fused into a conv epilogue (savp_model.py:572) */
int savp_sigmoid_bwd(void* stream, SavpView dy, SavpView y, float* out, int64_t N, int32_t HW, int32_t C);
/* ops.dense for few rows (M <= 64), split over K: out[M,C] = scale*x[M,K] W[K,C] + bias (out contiguous).
 * ws (optional, ws_floats >= M*C; best 64*M*C): per-slice partial sums + a reduction launch instead of float atomics on `out` */
int savp_dense_fwd(void* stream, const float* x, int64_t x_row_stride, int32_t M, int64_t K, int32_t C, const float* W,
                   const float* bias, const float* scale, float* out, float* ws, int64_t ws_floats);
int savp_axpby(void* stream, int64_t n, float a, const float* x, float b, const float* y, float* out);
int savp_fill_view(void* stream, SavpView out, int64_t R, int32_t HW, int32_t C, float value);
/* tf.train.AdamOptimizer on a flat arena (base_model.py:486-487); lr_t = lr*sqrt(1-b2^t)/(1-b1^t) from the host */
int savp_adam(void* stream, int64_t n, float* p, const float* g, float* m, float* v, float lr_t, float beta1, float beta2,
              float eps, float gscale, const float* lr_t_dev);   /* lr_t_dev != NULL: lr_t is read from device memory
                                                                    (hipGraph replays of the step with a changing rate) */

/* ------------------------------------------------------------------------------------------------------------
 * CDNA head + mask compositing (cdna_composite.hip): savp_model.py:551-559, 893-923, 634-646.
 * ------------------------------------------------------------------------------------------------------------ */
int savp_cdna_kernels_fwd(void* stream, const float* raw, float* kern, int32_t N, int32_t kh, int32_t kw, int32_t K);
int savp_cdna_kernels_bwd(void* stream, const float* raw, const double* dkern, float* draw, int32_t N, int32_t kh, int32_t kw,
                          int32_t K);
typedef struct SavpCdnaArgs {
    int32_t N, H, W, C, K, kh, kw;
    SavpView img;                  /* [N,H,W,C] */
    const float* kern;             /* normalised kernels [N, kh*kw, K] */
    SavpView out;                  /* [N,H,W,K*C], channel k*C+c */
    SavpView dout;                 /* bwd: gradient of out */
    SavpView dimg; int32_t dimg_beta;   /* bwd: p may be NULL */
    double* dkern;                 /* bwd: [N, kh*kw, K] FLOAT64, overwritten; may be NULL.  (The image tiles' partial sums meet here through
                                      float64 atomics: exact, so the result does not depend on the tiles' arrival order) */
} SavpCdnaArgs;
int savp_cdna_apply_fwd(void* stream, const SavpCdnaArgs* a);
int savp_cdna_apply_bwd(void* stream, const SavpCdnaArgs* a);
typedef struct SavpCompositeArgs {
    int32_t N, HW, M, C;
    const float* logits;           /* [N*HW, logits_stride], first M columns are the mask logits */
    int32_t logits_stride;
    SavpView timgs;                /* [N,HW,M*C], channel m*C+c */
    SavpView gen;                  /* [N,HW,C] */
    float* masks;                  /* optional [N*HW, M] */
    SavpView dgen;
    float* dlogits;                /* bwd: [N*HW, logits_stride] (pad columns zeroed) */
    SavpView drow;                 /* bwd: gradient of the WHOLE mask-conv input row [N,HW,row_channels]: channels
                                      [timgs_offset, timgs_offset+M*C) get mask_k*dgen, all others are written as 0 */
    int32_t timgs_offset, row_channels;
    /* fwd, optional (nnext > 0): the NEXT time step's input image, image = tf.where(ground_truth[t+1], inputs['images'][t+1], gen_image)
       (savp_model.py:406), written by the kernel that produces gen_image: next[k] [N,HW,C] receives gt_img where gt_mask[n] != 0 and
       the composited image elsewhere -- the separate select launch of every autoregressive step disappears */
    int32_t nnext; SavpView next[2];
    const int32_t* gt_mask;        /* [N] */
    SavpView gt_img;               /* [N,HW,C] */
} SavpCompositeArgs;
int savp_composite_fwd(void* stream, const SavpCompositeArgs* a);
int savp_composite_bwd(void* stream, const SavpCompositeArgs* a);

/* ------------------------------------------------------------------------------------------------------------
 * One entry per FUSED operator of the cell: the launches above with their workspace hand-overs checked
 * inside the library instead of being a caller convention (csrc/fused_ops.hip).  Each is two launches on the stream
 * (convolution, then the per-sample pass that needs whole-plane sums) and ONE host call.
 *   savp_convlstm_cell_fwd  BasicConv2DLSTMCell.call (rnn_ops.py:137-171): conv.mode FPROP with y = gates.gates; conv.stats (if
 *                           set) must be the head of gates.ws_stats and gates.stats1_ready must say so.
 *   savp_convlstm_cell_bwd  the gate block's backward, then conv (mode DGRAD, y = gates.dgates; dst_gap / nb_* as set).
 *   savp_conv_in_act_fwd    conv_pool2d / upsample_conv2d / conv2d + fused_instance_norm + activation (savp_model.py:449-464,
 *                           486-500,562-567,625-631): conv FPROP (or DGRAD mode for upsample_conv2d) whose destination is norm.x;
 *                           conv.stats (if set) == norm.ws with norm.stats_ready and norm.stats_shift == conv.bias.
 *   savp_conv_in_act_bwd    the norm's backward, then the convolution's data gradient reading norm.dx.
 * SAVP_EINVAL when the two halves do not fit together; otherwise the first failing half's code.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct SavpConvLstmCellArgs { SavpConvArgs conv; SavpLstmArgs gates; } SavpConvLstmCellArgs;
typedef struct SavpConvNormArgs { SavpConvArgs conv; SavpInormArgs norm; } SavpConvNormArgs;
int savp_convlstm_cell_fwd(void* stream, const SavpConvLstmCellArgs* a);
int savp_convlstm_cell_bwd(void* stream, const SavpConvLstmCellArgs* a);
int savp_conv_in_act_fwd(void* stream, const SavpConvNormArgs* a);
int savp_conv_in_act_bwd(void* stream, const SavpConvNormArgs* a);

/* ------------------------------------------------------------------------------------------------------------
 * small_ops.hip: z-LSTM over all timesteps (savp_model.py:354-362), reparameterisation + KL
 * (savp_model.py:45-49,711-712; losses.py:57-60), image / GAN / feature-matching losses (losses.py:6-54).
 * Loss entries accumulate the (unweighted) loss value into *loss_out and the weighted gradient into d*.
 * Round 6: every value that several workgroups add to -- *loss_out, *kl_out, the z-LSTM's dW / db / dc0 / dh0 -- is a FLOAT64 accumulator
 * (8-byte aligned): fp32 partials summed in float64 are exact, so the result does not depend on the workgroups' arrival order.
 * ------------------------------------------------------------------------------------------------------------ */
int savp_lstm_z_fwd(void* stream, const float* zs, const float* W, const float* bias, float* hout, float* gates, float* cs,
                    int32_t T, int32_t B, int32_t nz, float forget_bias);
int savp_lstm_z_bwd(void* stream, const float* zs, const float* W, const float* hout, const float* gates, const float* cs,
                    const float* dh_out, float* dzs, double* dW, double* db, int32_t T, int32_t B, int32_t nz, float forget_bias);
/* The same with a learned initial state (learn_initial_state, savp_model.py:295-307,344-352): c0 / h0 [nz] are tiled over the batch (NULL =
 * zero); bwd ADDS their gradients (what step 0 hands back, summed over the batch) to dc0 / dh0 [nz] (NULL = not wanted). */
int savp_lstm_z_fwd_init(void* stream, const float* zs, const float* W, const float* bias, float* hout, float* gates, float* cs,
                         int32_t T, int32_t B, int32_t nz, float forget_bias, const float* c0, const float* h0);
int savp_lstm_z_bwd_init(void* stream, const float* zs, const float* W, const float* hout, const float* gates, const float* cs,
                         const float* dh_out, float* dzs, double* dW, double* db, int32_t T, int32_t B, int32_t nz, float forget_bias,
                         const float* c0, const float* h0, double* dc0, double* dh0);
/* BasicLSTMCell over all timesteps (recurrent encoder of posterior_fn / prior_fn, savp_model.py:31-43,66-76).
 * A [T,B,I+U]: x_t in columns [0,I) (caller), h_{t-1} in [I,I+U) (written by fwd); W [I+U,4U], gate order i,j,f,o.
 * bwd produces dG [T,B,4U] and dA [T,B,I+U] (first I columns = dL/dx); dW = A^T dG and db = colsum(dG) are the caller's GEMM. */
int savp_lstm_seq_fwd(void* stream, float* A, const float* W, const float* bias, float* hout, float* gates, float* cs,
                      int32_t T, int32_t B, int32_t I, int32_t U, float forget_bias);
int savp_lstm_seq_bwd(void* stream, const float* A, const float* W, const float* gates, const float* cs, const float* dh_out,
                      float* dG, float* dA, int32_t T, int32_t B, int32_t I, int32_t U, float forget_bias);
/* tf.contrib.rnn.GRUCell over all timesteps (rnn = 'gru': the latent's cell, savp_model.py:358-359, and the encoders' recurrent tail, :38-41):
 * [r, u] = sigmoid([x, h] Wg + bg), c = tanh([x, r*h] Wc + bc), h' = u*h + (1-u)*c, zero initial state.  A [T,B,I+U]: x_t in columns [0,I)
 * (caller), h_{t-1} in [I,I+U) (written by fwd); Wg [I+U,2U] (r first), Wc [I+U,U].  fwd also fills A2 = [x | r*h_{t-1}] (the candidate GEMM's
 * input), ru [T,B,2U], cand [T,B,U], hout [T,B,U].  bwd produces dGg [T,B,2U], dGc [T,B,U] (gradients of the two pre-activations) and
 * dA [T,B,I+U] (first I columns = dL/dx); dWg = A^T dGg, dWc = A2^T dGc and the bias column sums are the caller's GEMMs.  U <= 512. */
int savp_gru_seq_fwd(void* stream, float* A, float* A2, const float* Wg, const float* bg, const float* Wc, const float* bc, float* hout,
                     float* ru, float* cand, int32_t T, int32_t B, int32_t I, int32_t U);
int savp_gru_seq_bwd(void* stream, const float* A, const float* Wg, const float* Wc, const float* ru, const float* cand, const float* dh_out,
                     float* dGg, float* dGc, float* dA, int32_t T, int32_t B, int32_t I, int32_t U);
/* ... from a caller-supplied initial state h0 [U] (learn_initial_state with the GRU latent cell, savp_model.py:288-291,344-352) and its
 * gradient dh0 [U] (float64, += : one workgroup per sample adds what step 0 hands back). */
int savp_gru_seq_fwd_init(void* stream, float* A, float* A2, const float* Wg, const float* bg, const float* Wc, const float* bc, float* hout,
                          float* ru, float* cand, int32_t T, int32_t B, int32_t I, int32_t U, const float* h0);
int savp_gru_seq_bwd_init(void* stream, const float* A, const float* Wg, const float* Wc, const float* ru, const float* cand, const float* dh_out,
                          float* dGg, float* dGc, float* dA, int32_t T, int32_t B, int32_t I, int32_t U, double* dh0);
/* KL between two diagonal Gaussians (losses.py:61-67; learn_prior): value into *kl_out (optional), klw-weighted gradient ADDED
 * to dmu1 / dls1 / dmu2 / dls2 (all four or none); ls*_raw are the unclipped log-variances. */
int savp_kl_gauss(void* stream, int64_t n, int32_t rows, const float* mu1, const float* ls1_raw, const float* mu2,
                  const float* ls2_raw, double* kl_out, float klw, const float* klw_dev, float* dmu1, float* dls1, float* dmu2,
                  float* dls2);
int savp_reparam_fwd(void* stream, int64_t n, int32_t rows, const float* mu, const float* ls_raw, const float* eps, float* ls,
                     float* z, double* kl_out);
int savp_reparam_bwd(void* stream, int64_t n, int32_t rows, const float* mu, const float* ls_raw, const float* eps,
                     const float* dz, float klw, float* dmu, float* dls_raw, const float* klw_dev);   /* klw_dev: as lr_t_dev */
int savp_lp_loss(void* stream, int64_t rows, int64_t row_len, int64_t pred_row_stride, int64_t target_row_stride, int32_t p2,
                 const float* pred, const float* target, float weight, double* loss_out, float* dpred);
/* Total variation of the predicted flows (base_model.py:763-769, tv_weight with transformation = 'flow'): flows = n_img images [H, W, C = 2 nk]
 * (element strides img_stride / px_stride); loss_out (float64) += s1 sum |f[y+1] - f[y]| + s2 sum |f[x+1] - f[x]| over these images (s1, s2: the
 * means' 1 / count over the WHOLE sequence); dflows (same addressing, may be NULL) += weight * gradient.  One call per time step inside BPTT. */
int savp_tv_loss(void* stream, const float* flows, int32_t n_img, int32_t H, int32_t W, int32_t C, int64_t img_stride, int64_t px_stride,
                 float s1, float s2, float weight, double* loss_out, float* dflows);
/* type 0 LSGAN, 1 GAN (sigmoid cross-entropy), 2 SNGAN (softplus hinge-free form), losses.py:29-54 */
int savp_gan_loss(void* stream, int32_t n, int32_t type, const float* logits, float label, float weight, double* loss_out,
                  float* dlogits, int32_t beta);
int savp_lsgan_loss(void* stream, int32_t n, const float* logits, float label, float weight, double* loss_out, float* dlogits,
                    int32_t beta);
int savp_cosine_distance(void* stream, int64_t P, int32_t C, const float* f0, const float* f1, float weight, float eps,
                         double* loss_out, float* df0, int32_t beta);

/* ------------------------------------------------------------------------------------------------------------
 * weight_prep.hip: packing for the conv kernel, conv_pool2d / upsample_conv2d kernel folding (ops.py:838-842,
 * 697-704) and spectral normalisation with its full gradient (ops.py:1020-1049).
 * ------------------------------------------------------------------------------------------------------------ */
/* wt_bf16 / wd_bf16: optional bf16 copies (same layouts) consumed by savp_conv in SAVP_PREC_BF16 mode */
int savp_pack_weights(void* stream, const float* src, int64_t T, int32_t Cx, int32_t Cy, const float* scale, float* wt, float* wd,
                      void* wt_bf16, void* wd_bf16);

/* The same for up to 32 layers in one launch (a network's layers after their optimiser step).  items is a HOST array. */
typedef struct {
    const float* src; const float* scale;          /* as savp_pack_weights; scale may be NULL */
    float* wt; float* wd; void* wt_bf16; void* wd_bf16;
    int64_t T; int32_t Cx, Cy;
} SavpPackItem;
int savp_pack_weights_batch(void* stream, int32_t n, const SavpPackItem* items);
int savp_fold_pool(void* stream, const float* in, float* out, int32_t k, int64_t C, int32_t adjoint);
int savp_fold_bilinear(void* stream, const float* in, float* out, int32_t k, int32_t Cin, int32_t F, int32_t adjoint);
/* ws: 8 + 2C + 2K + 2 (C + 2) floats, 8-byte aligned (the tail holds float64 accumulators of the forward sums: exact, order-independent); after fwd ws[0]=sigma, ws[1]=1/sigma; u_new receives u_final */
int savp_sn_fwd(void* stream, const float* W, int64_t K, int32_t C, const float* u, float* ws, float* u_new);
int savp_sn_bwd(void* stream, const float* W, int64_t K, int32_t C, const float* u, float* ws, const float* G, float* dW,
                int32_t beta);
/* The same for up to 16 weight tensors per call (e.g. the 8 spectrally normalised layers of one discriminator): 4 launches
 * instead of 4 per tensor.  items: HOST array.  fwd uses W, K, C, u, ws, u_new (may be NULL); bwd additionally G, dW, beta. */
typedef struct SavpSnItem {
    const float* W; int64_t K; int32_t C; const float* u; float* ws; float* u_new; const float* G; float* dW; int32_t beta;
} SavpSnItem;
int savp_sn_fwd_batch(void* stream, int32_t n, const SavpSnItem* items);
int savp_sn_bwd_batch(void* stream, int32_t n, const SavpSnItem* items);

/* ------------------------------------------------------------------------------------------------------------
 * warp_dna.hip: the alternative pixel transformations (hparams.transformation = 'flow' / 'dna').
 * image_warp: flow_ops.image_warp (flow_ops.py:4-79) for K flows at once; flows [N,H,W,2K] (x components then y).
 * dna_apply : per-pixel kernels incl. identity / relu-shift / normalisation (savp_model.py:541-544,556-559,858-890);
 *             raw [N,H*W,kh*kw*K] is the conv output, kern receives the normalised kernels (needed by bwd).
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct SavpWarpArgs {
    int32_t N, H, W, C, K;
    SavpView img;
    const float* flows;
    SavpView out;                  /* [N,H,W,K*C] */
    SavpView dout;
    float* dflows;                 /* bwd: [N,H,W,2K] overwritten */
    float* dimg;                   /* bwd: [N,H,W,C] contiguous, overwritten (zeroed + atomics); may be NULL */
} SavpWarpArgs;
int savp_image_warp_fwd(void* stream, const SavpWarpArgs* a);
int savp_image_warp_bwd(void* stream, const SavpWarpArgs* a);
typedef struct SavpDnaArgs {
    int32_t N, H, W, C, K, kh, kw;
    SavpView img;
    const float* raw;
    float* kern;
    SavpView out;
    SavpView dout;
    float* draw;
    SavpView dimg; int32_t dimg_beta;
} SavpDnaArgs;
int savp_dna_apply_fwd(void* stream, const SavpDnaArgs* a);
int savp_dna_apply_bwd(void* stream, const SavpDnaArgs* a);

/* ------------------------------------------------------------------------------------------------------------
 * gru.hip: fused gate blocks of Conv2DGRUCell (rnn_ops.py:234-267, instance norm, separate_norms=False).
 *  gates stage : pre [N,HW,2F] -> r,u = sigmoid(IN(pre)); writes u [N,HW,F] and r*h into the candidate conv's input slot
 *  output stage: pre [N,HW,F]  -> c = tanh(IN(pre)); h' = u*h + (1-u)*c written to nout destinations
 *  backward    : out_bwd gives dpre(F), du and dh = u*dh' (overwrites); gates_bwd gives dpre(2F) and dh += d(rh)*r (accumulates)
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct SavpGruArgs {
    int32_t N, HW, F;
    float eps;
    const float* pre;
    SavpView h;
    const float *gamma, *beta;
    float *mean, *rstd;
    float* u;
    SavpView rh;
    int32_t nout; SavpView out[4];
    int32_t ndy; SavpView dy[4];
    float* dpre;
    float* du;
    SavpView dh;
    SavpView drh;
    double *dgamma, *dbeta;        /* FLOAT64 accumulators (see SavpInormArgs.dgamma) */
} SavpGruArgs;
int savp_convgru_gates_fwd(void* stream, const SavpGruArgs* a);
int savp_convgru_out_fwd(void* stream, const SavpGruArgs* a);
int savp_convgru_out_bwd(void* stream, const SavpGruArgs* a);
int savp_convgru_gates_bwd(void* stream, const SavpGruArgs* a);

/* uint8 frames [B, T, frame] -> float32 time-major [T, B, frame] * (1/255): tf.image.convert_image_dtype (base_dataset.py:187)
 * + transpose_batch_time (tf_utils.py:118-122) for batches delivered by libsavp_io.so (include/savp_io.h); frame % 4 == 0 */
int savp_u8_frames_to_f32(void* stream, const uint8_t* in, float* out, int32_t B, int32_t T, int64_t frame);

/* ------------------------------------------------------------------------------------------------------------
 * Evaluation metrics and the best-of-N sampling fold (metrics.hip).  Time-major [T, B, ...] tensors with
 * explicit element strides (x_st = time, x_sb = batch); frames are contiguous (H*W*C floats).
 * ------------------------------------------------------------------------------------------------------------ */
/* metrics.py:5-10: mse[t,b] = mean((a-b)^2), psnr[t,b] = -10 log10(mse) (tf.image.psnr, max_val 1); either output may be NULL */
int savp_frame_mse_psnr(void* stream, const float* a, int64_t a_st, int64_t a_sb, const float* b, int64_t b_st, int64_t b_sb,
                        int32_t T, int32_t B, int32_t inner, float* mse, float* psnr);
/* metrics.py:13-14: tf.image.ssim(a, b, 1.0) per frame -> out[t,b] (11x11 Gaussian sigma 1.5, k1 .01, k2 .03, VALID) */
int savp_frame_ssim(void* stream, const float* a, int64_t a_st, int64_t a_sb, const float* b, int64_t b_st, int64_t b_sb,
                    int32_t T, int32_t B, int32_t H, int32_t W, int32_t C, float* out);
/* base_model.py:176-190: per batch element, if mean_t(metric) < mean_t(vmin): vmin <- metric (cond_min = 1); likewise max;
 * vsum += metric.  All [T, B] contiguous. */
int savp_eval_accumulate(void* stream, const float* metric, float* vmin, float* vsum, float* vmax, int32_t* cond_min,
                         int32_t* cond_max, int32_t T, int32_t B);
/* base_model.py:170-171: mode 0: out[t,b,:] = cond[b] ? x[t,b,:] : out[t,b,:] ; mode 1: out[t,b,:] += x[t,b,:] */
int savp_select_batch(void* stream, const int32_t* cond, const float* x, int64_t x_st, int64_t x_sb, float* out, int64_t o_st,
                      int64_t o_sb, int32_t T, int32_t B, int32_t inner, int32_t mode);

/* Fold float64 accumulators into fp32 gradients (round 6): dst[i] += (float) src[i] ; src[i] = 0 for i in idx[0 .. n) (idx NULL: i = 0 .. n-1).
 * The parameter gradients that many workgroups add to are accumulated in a float64 twin of the gradient arena (SavpInormArgs.dgamma, ...)
 * and rounded to fp32 once, here, before the optimiser (base_model.py:486-510) or the gradient exchange reads them. */
int savp_fold_f64(void* stream, const int32_t* idx, int64_t n, double* src, float* dst);

/* The robot-state recurrence of the action / state-conditioned cell (savp_model.py:411-422, 655-658, 684-685), all T steps in one launch
 * (state_pred.hip; it involves the actions and states only, so it is hoisted out of the per-frame loop like the latent's LSTMCell):
 *     state_t = gt[t, n] ? states_in[t, n] : gen_{t-1, n}  (gen_{-1} = 0) ;   gen_t = [actions_t | state_t] . W + b
 * actions [T, N, na] (NULL when na == 0), states_in [T, N, ns], gt int32 [T, N], W [(na + ns), ns], b [ns]; writes sa [T, N, na + ns] =
 * [actions_t | state_t] (what the cell tiles beside the latent -- under stop_gradient, :421-422) and gen [T, N, ns].  na + ns <= 32.
 * _bwd: dgen [T, N, ns] holds dL/dgen of the state loss (base_model.py:758-762) and is replaced by the total gradient (the step after
 * hands its state's gradient back where it took the prediction); dW / db: float64, += , one workgroup in a fixed order. */
int savp_state_pred_fwd(void* stream, int32_t T, int32_t N, int32_t na, int32_t ns, const float* actions, const float* states_in,
                        const int32_t* gt, const float* W, const float* b, float* sa, float* gen);
int savp_state_pred_bwd(void* stream, int32_t T, int32_t N, int32_t na, int32_t ns, const int32_t* gt, const float* W, const float* sa,
                        float* dgen, double* dW, double* db);

/* Weights of a gate convolution in MFMA B-fragment order (conv_gate.hip): src = the HWIO fp32 master [taps][Cx][Cy] (Cx % 8 == 0, Cy % 32 == 0);
 * out[cb][ks][lane][j] (bf16) = src[tap][ch8 * 8 + j][cb * 32 + (lane & 31)] for chunk 2 ks + (lane >> 5) = tap * (Cx / 8) + ch8, zero past the last
 * chunk, followed by 8 KB of zeros (the kernel's look-ahead reads past the last column block); savp_gate_weights_bytes(taps, Cx, Cy) bytes in all
 * (0: shape not supported), 16-byte aligned.  Once per optimiser step, like savp_pack_weights. */
int64_t savp_gate_weights_bytes(int32_t taps, int32_t Cx, int32_t Cy);
int savp_pack_gate_weights(void* stream, const float* src, int32_t taps, int32_t Cx, int32_t Cy, void* out, int32_t interleave);   /* interleave: see SavpConvArgs.w_frag_il */

/* ------------------------------------------------------------------------------------------------------------
 * Developer / soak-test aids (debug_ops.hip; no reference counterpart, no product caller): make what a correct launch sequence must never
 * read adversarial.  savp_debug_poison_lds fills all 160 KB of LDS of every CU with `pattern` (0xFFFFFFFF = NaN as fp32, bf16 and fp64);
 * `sink` (one uint32 of device memory, or NULL) counts words that did not read back.  savp_debug_fill_u32 fills `words` 32-bit words of
 * device memory (caller-owned scratch, free allocator blocks).  tests/test_gpu_soak.py, tests/conftest.py (SAVP_POISON=1).
 * ------------------------------------------------------------------------------------------------------------ */
int savp_debug_poison_lds(void* stream, uint32_t pattern, void* sink);
int savp_debug_fill_u32(void* stream, void* p, int64_t words, uint32_t pattern);
/* reads 64 KB of LDS per workgroup WITHOUT writing it: out2[0] += words equal to `pattern`, out2[1] += words read (two uint64 of device memory) */
int savp_debug_probe_lds(void* stream, uint32_t pattern, void* out2);

#ifdef __cplusplus
}
#endif
#endif /* SAVP_HIP_H */
