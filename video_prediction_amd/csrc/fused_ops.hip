// fused_ops.hip -- one C entry point per fused operator of the SAVP cell ("one per fused op, each with a _bwd twin").
//
// The kernels behind them are the library's own entry points (savp_conv, savp_convlstm_gates_*, savp_instnorm_act_*); what these
// entries add is the ORCHESTRATION a binder would otherwise have to know: which launch hands which workspace to the next one
// (the gate convolution's statistics epilogue -> the gate block's stats1; a conv_pool / upsample convolution's statistics -> the
// instance norm's stats_ready; the norm-backward sums of a data gradient -> the norm's backward), checked here instead of being a
// convention in prose.  A fused operator is still two launches on the stream -- conv, then the per-sample pass:
// the second needs per-(sample, channel) sums over the whole plane, i.e. a grid-wide dependency -- but ONE call
// from the host: the Python engine issues ~640 fewer ctypes calls per train step through these.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "savp_hip.h"

bool conv_gate_cell_try(const SavpConvLstmCellArgs* c, hipStream_t st, int* rc);      // conv_gate.hip

// BasicConv2DLSTMCell.call (rnn_ops.py:137-171), forward: gates = conv2d([x | z | h], W); IN(4F); i, j, f, o; c', h'.
extern "C" int savp_convlstm_cell_fwd(void* stream, const SavpConvLstmCellArgs* a) {
    if (!a || a->conv.mode != SAVP_CONV_FPROP) return SAVP_EINVAL;
    const SavpConvArgs& c = a->conv;
    const SavpLstmArgs& g = a->gates;
    if (c.y != g.gates || (c.out_bf16 != 0) != (g.gates_bf16 != 0)) return SAVP_EINVAL;       // the gate block reads what the conv wrote
    if (c.Cy != 4 * g.F || c.N != g.N || (long long)c.Do * c.Ho * c.Wo != g.HW) return SAVP_EINVAL;
    // statistics hand-over: the conv's epilogue fills the head of the gate block's workspace, and the gate block is told so
    if ((c.stats != nullptr) != (g.stats1_ready != 0)) return SAVP_EINVAL;
    if (c.stats && (void*)c.stats != g.ws_stats) return SAVP_EINVAL;
    // the `stats` epilogue sums the accumulators WITHOUT the bias and the gate block reads them as sums around 0 (rnn_ops.py:122-125: the gate
    // convolution has no bias when a normaliser follows): a biased convolution here would silently shift the mean
    if (c.stats && c.bias) return SAVP_EINVAL;
    {   // the whole cell as ONE kernel where the gate convolution's tile holds whole images (conv_gate.hip, SavpConvArgs.w_frag_il): both instance
        // norms' statistics are then workgroup-local -- no grid-wide dependency, no second launch
        int rc1 = SAVP_OK;
        if (conv_gate_cell_try(a, (hipStream_t)stream, &rc1)) return rc1;
    }
    int rc = savp_conv(stream, &c);
    if (rc != SAVP_OK) return rc;
    return savp_convlstm_gates_fwd(stream, &g);
}

// ... backward: the gate block's gradients (dgates, dc_prev, norm parameters), then the data gradient of the convolution into
// d[x | z | h] (a->conv.mode == SAVP_CONV_DGRAD with y = the gate gradient; dst_gap / nb_* as the caller set them).
extern "C" int savp_convlstm_cell_bwd(void* stream, const SavpConvLstmCellArgs* a) {
    if (!a || a->conv.mode != SAVP_CONV_DGRAD) return SAVP_EINVAL;
    const SavpConvArgs& c = a->conv;
    const SavpLstmArgs& g = a->gates;
    if (c.y != (void*)g.dgates || (c.src_bf16 != 0) != (g.dgates_bf16 != 0)) return SAVP_EINVAL;  // the DGRAD reads what the gate block wrote
    if (c.Cy != 4 * g.F || c.N != g.N || (long long)c.Do * c.Ho * c.Wo != g.HW) return SAVP_EINVAL;
    int rc = savp_convlstm_gates_bwd(stream, &g);
    if (rc != SAVP_OK) return rc;
    return savp_conv(stream, &c);
}

// conv_pool2d / upsample_conv2d / conv2d followed by fused_instance_norm + activation (savp_model.py:449-464,486-500,562-567,
// 625-631), forward.  The convolution is FPROP (x -> y = the norm's input) or, for upsample_conv2d, the DGRAD mode of its stride-2
// forward description (y -> x = the norm's input).
extern "C" int savp_conv_in_act_fwd(void* stream, const SavpConvNormArgs* a) {
    if (!a || (a->conv.mode != SAVP_CONV_FPROP && a->conv.mode != SAVP_CONV_DGRAD)) return SAVP_EINVAL;
    const SavpConvArgs& c = a->conv;
    const SavpInormArgs& n = a->norm;
    const void* dst = c.mode == SAVP_CONV_FPROP ? c.y : c.x;
    const int cdst = c.mode == SAVP_CONV_FPROP ? c.Cy : c.Cx;
    if (dst != n.x.p || cdst != n.C || c.N != n.N) return SAVP_EINVAL;                            // the norm reads what the conv wrote
    if ((c.stats != nullptr) != (n.stats_ready != 0)) return SAVP_EINVAL;                        // statistics hand-over, both or neither
    if (c.stats && (c.stats != n.ws || (c.bias != nullptr) != (n.stats_shift != nullptr) || (c.bias && c.bias != n.stats_shift))) return SAVP_EINVAL;
    int rc = savp_conv(stream, &c);
    if (rc != SAVP_OK) return rc;
    return savp_instnorm_act_fwd(stream, &n);
}

// ... backward: the norm's input gradient, then the convolution's data gradient from it.
extern "C" int savp_conv_in_act_bwd(void* stream, const SavpConvNormArgs* a) {
    if (!a || (a->conv.mode != SAVP_CONV_FPROP && a->conv.mode != SAVP_CONV_DGRAD)) return SAVP_EINVAL;
    const SavpConvArgs& c = a->conv;
    const SavpInormArgs& n = a->norm;
    const void* src = c.mode == SAVP_CONV_DGRAD ? c.y : c.x;                                       // the operand the data gradient reads
    if (src != n.dx.p || (c.src_bf16 != 0) != (n.dx_bf16 != 0) || c.N != n.N) return SAVP_EINVAL;
    int rc = savp_instnorm_act_bwd(stream, &n);
    if (rc != SAVP_OK) return rc;
    return savp_conv(stream, &c);
}
