// opts.h -- the library's ONLY process-wide switches: a table of named integers behind savp_set_option / savp_get_option
// (include/savp_hip.h).  The library never reads the environment; the host (video_prediction_amd/lib.py) may forward SAVP_*
// variables through savp_set_option when it loads the library.  Defaults are the shipped configuration.
#pragma once

enum SavpOptId {
    OPT_CONV_RING = 0,     // auto algorithm choice prefers the LDS-DMA ring kernel over the patch kernel (0)
    OPT_S2DGRAD,           // all-phase stride-(1,2,2) data-gradient kernel (1)
    OPT_THIN,              // RGB-side convolution kernels (1)
    OPT_WGP_CFG,           // developer: force a configuration of the LDS-patch weight gradient (0 = auto)
    OPT_WGP_SPLIT,         // developer: force its number of workgroups (0 = auto)
    OPT_INORM_MIN_HW,      // smallest plane that takes the coalesced two-kernel instance norm (64: measured 60.28 / 60.47 / 60.92 / 61.83 ms per step at 64 / 256 / 512 / 2048)
    OPT_COLSUM_2STAGE,     // partial rows + reduce launch for large column sums when a workspace is supplied (1)
    OPT_DENSE_LEGACY,      // developer: pre-round-2 few-row dense kernel (0)
    OPT_CDNA_LEGACY,       // developer: pre-round-2 CDNA kernels (0)
    OPT_LSTM_FUSED,        // one-launch ConvLSTM gate block, forward and backward (1)
    OPT_RING_DMA,          // bf16 sources of the ring kernel are staged into the LDS patch by LDS-DMA (1)
    OPT_LSTM_Q,            // developer: force the threads per pixel (channel quads per slab) of the one-launch gate kernels (0 = auto)
    OPT_RING_WWARM,        // ring kernel: workgroups of a column tile pull its weight block into their XCD's L2 first (1)
    OPT_WGP_DMA,           // LDS-patch weight gradient with both operands bf16: patch and dy tile staged by LDS-DMA (1)
    OPT_RING_EARLY,        // ring kernel: the first slab group's DMA-staged patch is requested at the top of the prologue (1)
    OPT_GATE_KERNEL,       // the ConvLSTM gate convolution takes conv_gate.hip when SavpConvArgs.w_frag is given (1)
    OPT_GATE_ALT,          // developer: conv_gate.hip's alternative tile instantiations (0)
    OPT_GATE_CELL,         // savp_convlstm_cell_fwd runs the whole cell in ONE launch where conv_gate.hip's tile holds whole images (1)
    OPT_GATE_WWARM,        // conv_gate.hip: workgroups of a column tile touch its weight block into their XCD's L2 first (1)
    OPT_SPLITK_REDUCED,    // counter, not a switch: calls whose requested split-K count was cut (or dropped) because the caller's scratch was too small
    OPT_COUNT
};

int savp_opt(int id);
void savp_opt_count(int id);       // += 1 (counters among the options: readable / resettable through savp_get_option / savp_set_option)

#ifdef __HIPCC__
// Touch every 64-byte line of the kernel-argument segment with one scalar load each and wait for all of them once.  The
// argument block of a launch is always cold (the command processor has just written it) and a miss is an HBM-latency round trip
// (~2 k cycles); hipcc loads a 500-byte struct field group by field group as the code reaches them, so a long prologue pays
// that latency three or four times in series (cycle stamps of conv_ring_kernel: ~3 k cycles before the first use of the
// geometry, 1.2 k more for the next group, ...).  After this call every later s_load of the struct hits the scalar cache.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
    constexpr int LINES = (BYTES + 63) / 64;
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned t[LINES];
#pragma unroll
    for (int i = 0; i < LINES; ++i) asm volatile("s_load_dword %0, %1, %2" : "=&s"(t[i]) : "s"(ka), "n"(i * 64) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < LINES; ++i) asm volatile("" :: "s"(t[i]));
}
#endif
