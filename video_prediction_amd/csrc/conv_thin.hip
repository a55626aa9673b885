// conv_thin.hip -- first layer of the spectral-norm discriminators (networks.py:35-108 of the reference: conv3d / conv2d, 3x3(x3),
// stride 1, SAME) in bf16 mode: FPROP and WGRAD of a convolution whose input is the RGB / grey clip itself (Cx <= 4, Cy = 32).
//
// Why its own kernels.  On the BAIR workload the layer reads 16 MB and writes (or, in WGRAD, reads) 168 MB of fp32 activations
// for 6.8 GFLOP: it is bound by HBM (~45 us at 4 TB/s), but the general kernels need 245 us (FPROP, implicit-GEMM gather with
// K = 81 of a 32-wide tile) and 420 + 95 us (generic WGRAD + a separate bias column sum): their tiles are built for hundreds
// of input channels, and the LDS-patch kernels want Cx % 4 == 0 (a spectrally normalised weight cannot be zero-padded).
//
// Layout used by both kernels: an input pixel is ONE 8-byte LDS word of 4 bf16 channels (the 4th, and for grey clips the
// 2nd..4th, zero).  The GEMM K / M index that runs over (tap, channel) is therefore "slot" s = tap (27 or 9 of them) times 4
// channels, and a slot of a pixel is one aligned ds_read_b64 at a tap-shifted address of the staged patch.
//
//   FPROP  y[px][co]   = sum_s  patch[px + shift(s)][0..3] * W[s][0..3][co]
//          A (rows = 32 consecutive output pixels of one image row) k-step of 16 = 4 slots = two ds_read_b64 per lane;
//          B (32 output channels) = the whole weight matrix, 7 (or 3) fragments held in registers for the kernel's lifetime;
//          v_mfma_f32_32x32x16_bf16; epilogue bias + LeakyReLU, each store instruction writes two full 128-byte pixel rows.
//   WGRAD  dW[s][c][co] += sum_px patch[px + shift(s)][c] * dy[px][co]          (K = pixels)
//          both operands are pixel-major in LDS while the MFMA wants k (= pixel) contiguous fragments: ds_read_b64_tr_b16, the
//          gfx950 transpose read (see conv_wgrad_patch.hip), with the four 4-channel chunks of a 16-row group mapped to four TAPS
//          (lane t of 16 supplies [pixel t>>2][tap t&3, 4 channels] and receives row (tap t>>2, channel t&3) of 4 pixels).
//          The bias gradient is summed in fp32 from the dy values as they are staged.  Each workgroup leaves its partial dW / db
//          in a workspace row; a second small launch sums the rows into dW / db.
#include "conv_common.h"
#include <stdlib.h>

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4v;
#define LDS_AS __attribute__((address_space(3)))

struct ThinP {
    const float* x; long long x_sn, x_sd, x_sh, x_sw;
    float* y; long long y_sn, y_sd, y_sh, y_sw;        // FPROP: destination; WGRAD: dy (read only)
    const float* w;                                    // FPROP: packed WT [32][taps * Cx] fp32
    float* dw; float* db;                              // WGRAD: dW [taps][Cx][32] (+=), db [32] (+=, may be null)
    float* ws;                                         // WGRAD: [workgroups][taps * Cx * 32 + 32] partial sums
    const float* bias; int act; float alpha;
    int N, D, H, W, Cx;
    int tilesX, tilesY, items, per_wg;
    int flip;                                          // FPROP kernel used as DGRAD: weight tap TAPS - 1 - s (the transposed conv's mirror)
    const float* zero;                                 // 16 bytes of zeros in global memory (the source of out-of-range pixels)
    const unsigned short* w16; int Cy;                 // wide -> thin FPROP: packed bf16 weights [Cy][9 * Cx], output channels
};

// one input pixel (CX <= 4 channels, zero outside the tensor) as raw floats: converted when it is parked in LDS, so that the loads
// of the NEXT work item stay in flight across the MFMAs of the current one.  Out-of-range pixels LOAD from a zero word instead of
// being zeroed after the load: with `if (inside) load` per channel hipcc sank half of the loads behind the MFMA block and waited
// for each group separately, and with a select on the loaded value it waits for the prefetch right where it is issued -- either
// way two or three exposed memory round trips per work item.
__device__ float4 g_thin_zero[1] = {{0.f, 0.f, 0.f, 0.f}};     // reached through ThinP.zero (a kernel-argument pointer = global
                                                               // address space; the symbol itself would turn the loads into FLAT ones,
                                                               // which also count on lgkmcnt and so cannot stay in flight across LDS reads)

template <int CX>
__device__ __forceinline__ float4 thin_fetch_pixel(const ThinP& p, int n, int iz, int iy, int ix) {
    const bool ok = iz >= 0 && iz < p.D && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    const float* __restrict__ s = ok ? p.x + ((long long)n * p.x_sn + (long long)iz * p.x_sd + (long long)iy * p.x_sh + (long long)ix * p.x_sw)
                                     : p.zero;
    float4 v;
    v.x = s[0];
    v.y = CX > 1 ? s[1] : 0.f;
    v.z = CX > 2 ? s[2] : 0.f;
    v.w = CX > 3 ? s[3] : 0.f;
    return v;
}

struct ThinItem { int n, z, y0, x0; };
__device__ __forceinline__ ThinItem thin_item(const ThinP& p, int it, int tile_r, int tile_c) {
    const int tx = it % p.tilesX, t1 = it / p.tilesX;
    const int ty = t1 % p.tilesY, t2 = t1 / p.tilesY;
    ThinItem q;
    q.z = t2 % p.D; q.n = t2 / p.D; q.y0 = ty * tile_r; q.x0 = tx * tile_c;
    return q;
}

// the (plane, row, column) of patch slot s for a patch of PR x PC pixels per plane
#define THIN_FETCH_SLOTS(NSL, PR, PC, KD_, pv, q)                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < NSL; ++i_) {                                                       \
        const int s_ = tid + 256 * i_;                                                                        \
        const int a_ = s_ / ((PR) * (PC)), rem_ = s_ - a_ * ((PR) * (PC));                                    \
        const int r_ = rem_ / (PC), c_ = rem_ - r_ * (PC);                                                    \
        /* slots past the patch (last round of the 256-thread sweep) read pixel (-1, ..) = out of range = zero, never stored */ \
        pv[i_] = thin_fetch_pixel<CX>(p, q.n, s_ < (KD_) * (PR) * (PC) ? q.z + a_ - (KD_) / 2 : -1, q.y0 + r_ - 1, q.x0 + c_ - 1); \
    }
#define THIN_STAGE_SLOTS(NSL, PR, PC, KD_, pv)                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < NSL; ++i_) {                                                       \
        const int s_ = tid + 256 * i_;                                                                        \
        if (s_ < (KD_) * (PR) * (PC)) patch[s_] = bf16x4v{(__bf16)pv[i_].x, (__bf16)pv[i_].y, (__bf16)pv[i_].z, (__bf16)pv[i_].w}; \
    }

// Pin prefetched registers behind the compute block: left alone, hipcc hoists the bf16 conversion of the NEXT item's pixels (the
// first thing the next loop iteration does) above this item's MFMAs and waits for the prefetch right after issuing it.
#define THIN_PIN4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))

// ------------------------------------------------------------------------------------------------------------
// FPROP: 8 x 32 output pixels of one (sample, plane) per work item; wave w owns rows 2w, 2w+1
// ------------------------------------------------------------------------------------------------------------
#define TF_R 8
#define TF_C 32
#define TF_PR (TF_R + 2)
#define TF_PC (TF_C + 2)

template <int KD, int CX>
__global__ __launch_bounds__(256) void thin_fprop_kernel(ThinP p) {
    constexpr int TAPS = 9 * KD, NS = (TAPS + 3) / 4;          // slots; k-steps of 4 slots (16 k)
    constexpr int PPL = TF_PR * TF_PC, ZP = KD * PPL;           // pixels per patch plane; index of the all-zero pixel
    __shared__ __attribute__((aligned(16))) bf16x4v patch[ZP + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    // weights: lane = output channel l31, k = 8 half + j  <->  slot 4 ks + 2 half + (j >> 2), channel j & 3
    bf16x8 bw[NS];
    int aoff[NS][2];                                            // byte offset of the lane's two slots per k-step (-1: zero pixel)
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = 4 * ks + 2 * half + (j >> 2), c = j & 3;
            const bool live = s < TAPS && c < CX;               // unconditional (clamped) load + select: 56 loads in flight at once
            const float v = p.w[live ? l31 * TAPS * CX + (p.flip ? TAPS - 1 - s : s) * CX + c : 0];
            bw[ks][j] = (__bf16)(live ? v : 0.f);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int s = 4 * ks + 2 * half + jj;
            const int a = s / 9, u = (s % 9) / 3, v = s % 3;
            aoff[ks][jj] = (s < TAPS) ? ((a * TF_PR + u) * TF_PC + v) * 8 : -1;
        }
    }
    float bias = 0.f;
    if (p.bias) bias = p.bias[l31];
    if (tid == 0) patch[ZP] = bf16x4v{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    constexpr int NSL = (KD * PPL + 255) / 256;
    float4 pv[NSL];
    const int it_begin = blockIdx.x * p.per_wg, it_end = min(p.items, (int)(blockIdx.x + 1) * p.per_wg);
    {
        const ThinItem q0 = thin_item(p, it_begin, TF_R, TF_C);
        THIN_FETCH_SLOTS(NSL, TF_PR, TF_PC, KD, pv, q0)
    }
    for (int it = it_begin; it < it_end; ++it) {
        const ThinItem q = thin_item(p, it, TF_R, TF_C);
        const int z = q.z, n = q.n, y0 = q.y0, x0 = q.x0;
        __syncthreads();                                       // the previous item's reads are done
        THIN_STAGE_SLOTS(NSL, TF_PR, TF_PC, KD, pv)
        __syncthreads();
        {   // next item's pixels fly while this one is multiplied and stored.  Unconditional (the last item is fetched twice): under
            // `if (it + 1 < it_end)` the loop-carried registers become phis that hipcc resolves with copies of the loaded values
            // right behind the loads, i.e. with a full wait for the prefetch
            const ThinItem qn = thin_item(p, min(it + 1, it_end - 1), TF_R, TF_C);
            THIN_FETCH_SLOTS(NSL, TF_PR, TF_PC, KD, pv, qn)
        }
        const LDS_AS char* pl = (const LDS_AS char*)patch;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * wave + rr, oy = y0 + r;
            const int base = (r * TF_PC + l31) * 8;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NS; ++ks) {
                const int ad0 = aoff[ks][0] < 0 ? ZP * 8 : base + aoff[ks][0];
                const int ad1 = aoff[ks][1] < 0 ? ZP * 8 : base + aoff[ks][1];
                const bf16x4v a0 = *(const LDS_AS bf16x4v*)(pl + ad0);
                const bf16x4v a1 = *(const LDS_AS bf16x4v*)(pl + ad1);
                const bf16x8 af = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bw[ks], acc, 0, 0, 0);
            }
            if (oy < p.H) {
                float* __restrict__ dst = p.y + (long long)n * p.y_sn + (long long)z * p.y_sd + (long long)oy * p.y_sh + l31;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ox = x0 + (i & 3) + 8 * (i >> 2) + 4 * half;    // accumulator row = output pixel of the 32-wide row tile
                    if (ox < p.W) {
                        float v = acc[i] + bias;
                        if (p.act == SAVP_ACT_LRELU) v = v > 0.f ? v : v * p.alpha;
                        dst[(long long)ox * p.y_sw] = v;
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                     // ... and keep the pins themselves below the MFMAs / stores
#pragma unroll
        for (int i = 0; i < NSL; ++i) THIN_PIN4(pv[i]);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Thin -> wide with 8 thin channels (round 5): the data gradient of the mask convolution (56 <- 8, accumulated into the gradient of its
// input: beta = 1) -- K = 72, 29 MB written (+ 29 MB read back for the accumulation) per launch, 36 us on the ring kernel.  As
// thin_fprop_kernel with a pixel = ONE 16-byte LDS word of 8 bf16 channels: a k-step of 16 is two taps (lane half h supplies tap 2 ks + h),
// up to 64 wide channels as two 32-column blocks, weights (5 k-steps x 2 blocks) in registers from the packed bf16 copy, every store
// instruction writes two 128-byte runs.  flip = the transposed convolution's mirrored taps (DGRAD).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void thin8_wide_kernel(ThinP p) {
    constexpr int NS = 5;                                       // k-steps: taps 0 .. 9 (tap 9 = zeros)
    constexpr int PPL = TF_PR * TF_PC, ZP = PPL;                // patch pixels; index of the all-zero pixel
    __shared__ __attribute__((aligned(16))) uint4 patch8[PPL + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    bf16x8 bw[NS][2];
    int aoff[NS];                                               // byte offset of the lane's tap per k-step (-1: zero pixel)
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
        const int s = 2 * ks + half;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = 32 * b + l31;
            const bool live = s < 9 && col < p.Cy;              // p.Cy = wide channels here
            const uint4 v = *reinterpret_cast<const uint4*>(p.w16 + (live ? (long long)col * 72 + (p.flip ? 8 - s : s) * 8 : 0ll));
            bw[ks][b] = __builtin_bit_cast(bf16x8, live ? v : make_uint4(0u, 0u, 0u, 0u));
        }
        aoff[ks] = s < 9 ? ((s / 3) * TF_PC + s % 3) * 16 : -1;
    }
    if (tid == 0) patch8[ZP] = make_uint4(0u, 0u, 0u, 0u);
    constexpr int NSL = (PPL * 2 + 255) / 256;                  // float4 slots: two per pixel
    float4 pv[NSL];
    auto fetch = [&](const ThinItem& q) {
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int s = tid + 256 * i;
            const int px = s >> 1, c4 = s & 1;
            const int r = px / TF_PC, c = px - r * TF_PC;
            const int iy = q.y0 + r - 1, ix = q.x0 + c - 1;
            const bool ok = px < PPL && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const float* __restrict__ src = ok ? p.x + ((long long)q.n * p.x_sn + (long long)iy * p.x_sh + (long long)ix * p.x_sw + c4 * 4) : p.zero;
            pv[i] = *reinterpret_cast<const float4*>(src);
        }
    };
    const int it_begin = blockIdx.x * p.per_wg, it_end = min(p.items, (int)(blockIdx.x + 1) * p.per_wg);
    if (it_begin >= it_end) return;
    fetch(thin_item(p, it_begin, TF_R, TF_C));
    for (int it = it_begin; it < it_end; ++it) {
        const ThinItem q = thin_item(p, it, TF_R, TF_C);
        __syncthreads();                                       // the previous item's reads are done
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int s = tid + 256 * i;
            if ((s >> 1) < PPL)
                reinterpret_cast<bf16x4v*>(patch8)[s] = bf16x4v{(__bf16)pv[i].x, (__bf16)pv[i].y, (__bf16)pv[i].z, (__bf16)pv[i].w};
        }
        __syncthreads();
        if (it + 1 < it_end) fetch(thin_item(p, it + 1, TF_R, TF_C));
        const LDS_AS char* pl = (const LDS_AS char*)patch8;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * wave + rr, oy = q.y0 + r;
            if (oy >= p.H) continue;                           // wave-uniform
            float* __restrict__ dst = p.y + (long long)q.n * p.y_sn + (long long)oy * p.y_sh + l31;
            // the accumulation's old values: requested before the MFMAs, consumed behind them
            float old[2][16];
            if (p.flip > 1) {                                  // (flip bit 1 = beta)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int ox = min(q.x0 + (i & 3) + 8 * (i >> 2) + 4 * half, p.W - 1), col = min(32 * b + l31, p.Cy - 1);
                        old[b][i] = dst[(long long)ox * p.y_sw + (col - l31)];
                    }
            }
            const int base = (r * TF_PC + l31) * 16;
            f32x16 acc[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NS; ++ks) {
                const bf16x8 af = *(const LDS_AS bf16x8*)(pl + (aoff[ks] < 0 ? ZP * 16 : base + aoff[ks]));
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bw[ks][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bw[ks][1], acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = 32 * b + l31;
                if (col >= p.Cy) continue;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int ox = q.x0 + (i & 3) + 8 * (i >> 2) + 4 * half;
                    if (ox < p.W) dst[(long long)ox * p.y_sw + 32 * b] = acc[b][i] + (p.flip > 1 ? old[b][i] : 0.f);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Wide -> thin FPROP (round 5): a 3x3 stride-1 SAME convolution from a feature tensor (Cx a multiple of 8, <= 64) to a FEW channels
// (Cy <= 32 computed, only the first Cy stored) -- the generator's scratch-image head (32 -> 3 or 4, sigmoid, written into a channel
// slice of the mask convolution's input) and the mask convolution itself (56 -> 8).  0.3 - 0.6 GFLOP over 17 - 29 MB: HBM-bound
// (~6 us), but the general kernels spend 23 - 26 us on it per time step (per-workgroup prologue + patch staging for 9 - 18 (tap,
// slab) entries of work, a 64-column tile for 4 - 8 real columns).  Same organisation as thin_fprop_kernel: a work item is 4 x 32 output
// pixels (a row per wave), the next item's pixels are in flight while this one multiplies, the WHOLE weight matrix sits in registers (9 taps x NKT
// k-steps of 16 channels, one 16-byte load each from the packed bf16 copy), A fragments are tap-shifted ds_read_b128 of the bf16 patch
// (pixel stride an odd multiple of 16 bytes: conflict-free).
// ------------------------------------------------------------------------------------------------------------
#define WF_R 4                                                 // output rows per work item: one per wave
#define WF_PR (WF_R + 2)
template <int NKT>
__global__ __launch_bounds__(256, 2) void wthin_fprop_kernel(ThinP p) {
    constexpr int SPP = 4 * NKT;                               // float4 slots per patch pixel (16 NKT channels, zero beyond Cx)
    constexpr int PSTR = 32 * NKT + 16;                        // LDS bytes per pixel: 80 / 144 = 5 / 9 x 16
    constexpr int PPL = WF_PR * TF_PC;                         // patch pixels (SAME padding is staged as zeros)
    constexpr int NSL = (PPL * SPP + 255) / 256;
    constexpr int NE = 9 * NKT;                                // (tap, k-step) entries
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    // weights: lane = output channel l31 (zero beyond Cy), k = 8 half + j  <->  channel 16 ks + 8 half + j of tap t.  In registers for the
    // lifetime of the kernel up to 32 input channels (72 VGPRs); beyond (144) they are parked in LDS as ready-made B fragments
    // [entry][lane] -- with them in registers the kernel needs > 256 VGPRs, i.e. ONE 4-wave workgroup per CU and nothing to hide its latencies
    constexpr bool WLDS = NKT > 2;
    char* wl = wsm + PPL * PSTR;                               // [NE][64] x 16 bytes
    bf16x8 bw[WLDS ? 1 : NE];
    auto w_of = [&](int e, int l31_, int half_) -> uint4 {
        const int t = e / NKT, c0 = 16 * (e % NKT) + 8 * half_;
        const bool live = l31_ < p.Cy && c0 < p.Cx;            // unconditional (clamped) load + select: all of them in flight at once
        const uint4 v = *reinterpret_cast<const uint4*>(p.w16 + (live ? (long long)l31_ * 9 * p.Cx + t * p.Cx + c0 : 0ll));
        return live ? v : make_uint4(0u, 0u, 0u, 0u);
    };
    if constexpr (WLDS) {
        constexpr int NWL = (NE * 64 + 255) / 256;
        uint4 wv[NWL];
#pragma unroll
        for (int i = 0; i < NWL; ++i) { const int idx = min(tid + 256 * i, NE * 64 - 1); wv[i] = w_of(idx >> 6, idx & 31, (idx >> 5) & 1); }
#pragma unroll
        for (int i = 0; i < NWL; ++i) { const int idx = tid + 256 * i; if (idx < NE * 64) *reinterpret_cast<uint4*>(wl + idx * 16) = wv[i]; }
    } else {
#pragma unroll
        for (int e = 0; e < NE; ++e) bw[e] = __builtin_bit_cast(bf16x8, w_of(e, l31, half));
    }
    float bias = 0.f;
    if (p.bias && l31 < p.Cy) bias = p.bias[l31];
    float4 pv[NSL];
    auto fetch = [&](const ThinItem& q) {
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int s = tid + 256 * i;
            const int px = s / SPP, c4 = s - px * SPP;
            const int r = px / TF_PC, c = px - r * TF_PC;
            const int iy = q.y0 + r - 1, ix = q.x0 + c - 1;
            const bool ok = px < PPL && c4 * 4 < p.Cx && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const float* __restrict__ src = ok ? p.x + ((long long)q.n * p.x_sn + (long long)iy * p.x_sh + (long long)ix * p.x_sw + c4 * 4) : p.zero;
            pv[i] = *reinterpret_cast<const float4*>(src);
        }
    };
    const int it_begin = blockIdx.x * p.per_wg, it_end = min(p.items, (int)(blockIdx.x + 1) * p.per_wg);
    if (it_begin >= it_end) return;
    fetch(thin_item(p, it_begin, WF_R, TF_C));
    for (int it = it_begin; it < it_end; ++it) {
        const ThinItem q = thin_item(p, it, WF_R, TF_C);
        __syncthreads();                                       // the previous item's reads are done
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int s = tid + 256 * i;
            const int px = s / SPP, c4 = s - px * SPP;
            if (px < PPL)
                *reinterpret_cast<bf16x4v*>(wsm + px * PSTR + c4 * 8) = bf16x4v{(__bf16)pv[i].x, (__bf16)pv[i].y, (__bf16)pv[i].z, (__bf16)pv[i].w};
        }
        __syncthreads();
        if (it + 1 < it_end) fetch(thin_item(p, it + 1, WF_R, TF_C));
        const LDS_AS char* pl = (const LDS_AS char*)wsm + (wave * TF_PC + l31) * PSTR + half * 16;
        const int oy = q.y0 + wave;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        // fragment reads run three entries ahead of their MFMA (left alone, hipcc reads all 9 NKT fragments first: 72 - 144 more registers)
        auto a_of = [&](int e) -> bf16x8 { const int t = e / NKT; return *(const LDS_AS bf16x8*)(pl + ((t / 3) * TF_PC + t % 3) * PSTR + (e % NKT) * 32); };
        constexpr int AD = 3;
        bf16x8 af[AD + 1], bf[WLDS ? AD + 1 : 1];
        const LDS_AS char* wll = (const LDS_AS char*)wl + lane * 16;
        auto b_of = [&](int e) -> bf16x8 { return *(const LDS_AS bf16x8*)(wll + e * 1024); };
#pragma unroll
        for (int e = 0; e < AD; ++e) { af[e] = a_of(e); if constexpr (WLDS) bf[e] = b_of(e); }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (e + AD < NE) { af[(e + AD) % (AD + 1)] = a_of(e + AD); if constexpr (WLDS) bf[(e + AD) % (AD + 1)] = b_of(e + AD); }
            if constexpr (WLDS) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[e % (AD + 1)], bf[e % (AD + 1)], acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[e % (AD + 1)], bw[e], acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, WLDS ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if (oy < p.H && l31 < p.Cy) {
            float* __restrict__ dst = p.y + (long long)q.n * p.y_sn + (long long)oy * p.y_sh + l31;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ox = q.x0 + (i & 3) + 8 * (i >> 2) + 4 * half;    // accumulator row = output pixel of the 32-wide row tile
                if (ox < p.W) {
                    float v = acc[i] + bias;
                    if (p.act == SAVP_ACT_LRELU) v = v > 0.f ? v : v * p.alpha;
                    else if (p.act == SAVP_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                    dst[(long long)ox * p.y_sw] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// WGRAD: 4 x 64 output pixels of one (sample, plane) per work item; wave w reduces over row w (a larger item would need more
// prefetch registers than two resident workgroups per CU allow)
// ------------------------------------------------------------------------------------------------------------
#define TW_R 4
#define TW_C 64
#define TW_PR (TW_R + 2)
#define TW_PC (TW_C + 2)

template <int KD, int CX>
__global__ __launch_bounds__(256, 2) void thin_wgrad_kernel(ThinP p) {
    constexpr int TAPS = 9 * KD, NG = (TAPS + 3) / 4, NT32 = (NG + 1) / 2;   // 16-row groups (4 taps x 4 channels); 32-row tiles
    constexpr int PPL = TW_PR * TW_PC, ZP = KD * PPL;
    constexpr int PATCH_BYTES = ((ZP + 1) * 8 + 15) & ~15;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16x4v* patch = reinterpret_cast<bf16x4v*>(smem);
    char* dyt = smem + PATCH_BYTES;                               // [256 pixels][32 channels] bf16, 64 bytes per pixel (>= 16 KB: reused by the final reduction)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, g = (lane >> 4) & 1, r4 = (lane & 15) >> 2, q = lane & 3;
    // A: this lane feeds tap 4 (2 i + g) + q of row tile i at k-pixel 8 h + 4 j + r4 (j = 0, 1: the two transpose reads of a k-step)
    int toff[NT32];
#pragma unroll
    for (int i = 0; i < NT32; ++i) {
        const int tap = 4 * (2 * i + g) + q;
        const int a = tap / 9, u = (tap % 9) / 3, v = tap % 3;
        toff[i] = (tap < TAPS) ? ((a * TW_PR + u) * TW_PC + v) * 8 : -1;
    }
    const int a_lane = (8 * h + r4) * 8;                           // bytes
    const int b_lane = (8 * h + r4) * 64 + (16 * g + 4 * q) * 2;
    f32x16 acc[NT32];
#pragma unroll
    for (int i = 0; i < NT32; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid == 0) patch[ZP] = bf16x4v{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    constexpr int NSL = (KD * PPL + 255) / 256;
    float4 pv[NSL], dv[8];
    auto fetch = [&](int it) {
        const ThinItem q = thin_item(p, it, TW_R, TW_C);
        THIN_FETCH_SLOTS(NSL, TW_PR, TW_PC, KD, pv, q)
        const float* __restrict__ dyb = p.y + (long long)q.n * p.y_sn + (long long)q.z * p.y_sd;
#pragma unroll
        for (int i = 0; i < 8; ++i) {                              // 256 pixels x 8 channel quads; this thread's quad is tid & 7
            const int slot = tid + 256 * i, px = slot >> 3, cq = slot & 7;
            const int oy = q.y0 + (px >> 6), ox = q.x0 + (px & 63);
            const bool ok = oy < p.H && ox < p.W;
            dv[i] = ldg4(ok ? dyb + (long long)oy * p.y_sh + (long long)ox * p.y_sw + 4 * cq : p.zero);
        }
    };
    const int it_begin = blockIdx.x * p.per_wg, it_end = min(p.items, (int)(blockIdx.x + 1) * p.per_wg);
    fetch(it_begin);
    for (int it = it_begin; it < it_end; ++it) {
        __syncthreads();
        THIN_STAGE_SLOTS(NSL, TW_PR, TW_PC, KD, pv)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int slot = tid + 256 * i, px = slot >> 3, cq = slot & 7;
            const float4 v = dv[i];
            bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
            *reinterpret_cast<bf16x4v*>(dyt + px * 64 + cq * 8) = bf16x4v{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        }
        __syncthreads();
        fetch(min(it + 1, it_end - 1));                            // next item's operands fly across this item's MFMAs (unconditional, see FPROP)
        const LDS_AS char* pa = (const LDS_AS char*)patch;
        const LDS_AS char* pb = (const LDS_AS char*)dyt + b_lane;
        {
            const int r = wave;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const LDS_AS char* b0p = pb + (r * 64 + 16 * ks) * 64;
                const bf16x4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4v*)b0p);
                const bf16x4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4v*)(b0p + 4 * 64));
                const bf16x8 bf = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                const int abase = (r * TW_PC + 16 * ks) * 8 + a_lane;
#pragma unroll
                for (int i = 0; i < NT32; ++i) {
                    const int ad = toff[i] < 0 ? ZP * 8 : abase + toff[i];
                    const int ad1 = toff[i] < 0 ? ZP * 8 : ad + 4 * 8;
                    const bf16x4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4v*)(pa + ad));
                    const bf16x4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4v*)(pa + ad1));
                    const bf16x8 af = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[i], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NSL; ++i) THIN_PIN4(pv[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) THIN_PIN4(dv[i]);
    }
    // ---- this workgroup's partial dW / db goes to its own row of a workspace (plain stores); thin_wgrad_reduce_kernel sums the
    //      rows.  Adding to dW directly costs 2.6 k atomics per workgroup onto the SAME 82 cache lines from 512 workgroups that all
    //      finish together: ~16 k serialised read-modify-writes per line, 90 of the kernel's 139 us in the first version.
    __syncthreads();
    float* red = reinterpret_cast<float*>(dyt);                    // [NT32][32 rows][32 columns] fp32 (<= 16 KB = the dy tile)
    float* bred = reinterpret_cast<float*>(patch);                 // [32] fp32 (the patch is dead)
    const int l31 = lane & 31;
    if (tid < 32) bred[tid] = 0.f;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NT32; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * h;  // accumulator row of the 32-row tile
                    float* e = red + (i * 32 + m) * 32 + l31;
                    if (w == 0) *e = acc[i][r]; else *e += acc[i][r];
                }
        }
        __syncthreads();
    }
    // bias gradient: all of a thread's dy slots are channel quad tid & 7
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) {
        bsum.x += __shfl_xor(bsum.x, m); bsum.y += __shfl_xor(bsum.y, m);
        bsum.z += __shfl_xor(bsum.z, m); bsum.w += __shfl_xor(bsum.w, m);
    }
    for (int w = 0; w < 4; ++w) {                                  // the four waves in wave order (an LDS atomic's order is arrival order)
        if (wave == w && lane < 8) {
            bred[4 * lane] += bsum.x; bred[4 * lane + 1] += bsum.y; bred[4 * lane + 2] += bsum.z; bred[4 * lane + 3] += bsum.w;
        }
        __syncthreads();
    }
    constexpr int ROW = TAPS * CX * 32 + 32;                       // floats per workspace row: dW then db
    float* __restrict__ wsr = p.ws + (long long)blockIdx.x * ROW;
    for (int e = tid; e < NT32 * 1024; e += 256) {
        const int co = e & 31, m = (e >> 5) & 31, i = e >> 10;
        const int tap = 4 * (2 * i + (m >> 4)) + ((m & 15) >> 2), c = m & 3;
        if (tap < TAPS && c < CX) wsr[(tap * CX + c) * 32 + co] = red[e];
    }
    if (tid < 32) wsr[TAPS * CX * 32 + tid] = bred[tid];
}

// out[e] += sum over the workspace rows, in a fixed order: a block owns 32 consecutive elements; thread (g = tid >> 5, tid & 31) sums rows
// g, g + 8, ... (eight loads in flight), the eight row groups are added in group order.  One writer per element: no atomics, and the
// same bits whatever order the workgroups of the gradient kernel finished in.
__global__ __launch_bounds__(256) void thin_wgrad_reduce_kernel(const float* __restrict__ ws, int rows, int row, int ndw, float* dw, float* db) {
    __shared__ float part[8][32];
    const int e = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
    float s = 0.f;
    if (e < row) {
        int r = g;
        for (; r + 56 < rows; r += 64) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ws[(long long)(r + 8 * j) * row + e];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; r < rows; r += 8) s += ws[(long long)r * row + e];
    }
    part[g][threadIdx.x & 31] = s;
    __syncthreads();
    if (g == 0 && e < row) {
        float t = part[0][threadIdx.x];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += part[j][threadIdx.x];
        if (e < ndw) dw[e] += t;
        else if (db) db[e - ndw] += t;
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
static bool thin_enabled() { return savp_opt(OPT_THIN) == 1; }

static bool thin_geometry_ok(const SavpConvArgs* a) {
    // DGRAD of a 32 -> (1 | 3 | 4)-channel convolution is the same problem with the tensors' roles swapped and the taps mirrored
    const bool dg = a->mode == SAVP_CONV_DGRAD;
    const int cthin = dg ? a->Cy : a->Cx, cwide = dg ? a->Cx : a->Cy;
    return a->precision == SAVP_PREC_BF16 && (cthin == 1 || cthin == 3 || cthin == 4) && cwide == 32 && a->kh == 3 && a->kw == 3 && a->ph == 1 &&
           a->pw == 1 && (a->kd == 1 || a->kd == 3) && a->pd == a->kd / 2 && a->sd == 1 && a->sh == 1 && a->sw == 1 && a->Do == a->D &&
           a->Ho == a->H && a->Wo == a->W && !a->src_bf16 && !a->out_bf16 && !a->stats && a->N >= 1 && a->H >= 1 && a->W >= 1;
}

static void thin_fill(ThinP& p, const SavpConvArgs* a, int tile_r, int tile_c, int target_wgs) {
    const bool dg = a->mode == SAVP_CONV_DGRAD;             // ThinP.x = the thin (source) tensor, ThinP.y = the 32-channel one
    if (dg) {
        p.x = (const float*)a->y; p.x_sn = a->y_sn; p.x_sd = a->y_sd; p.x_sh = a->y_sh; p.x_sw = a->y_sw;
        p.y = (float*)a->x; p.y_sn = a->x_sn; p.y_sd = a->x_sd; p.y_sh = a->x_sh; p.y_sw = a->x_sw;
    } else {
        p.x = (const float*)a->x; p.x_sn = a->x_sn; p.x_sd = a->x_sd; p.x_sh = a->x_sh; p.x_sw = a->x_sw;
        p.y = (float*)a->y; p.y_sn = a->y_sn; p.y_sd = a->y_sd; p.y_sh = a->y_sh; p.y_sw = a->y_sw;
    }
    p.w = nullptr; p.dw = nullptr; p.db = nullptr; p.ws = nullptr; p.bias = nullptr; p.act = 0; p.alpha = 0.f;
    p.flip = dg ? 1 : 0;
    p.N = a->N; p.D = a->D; p.H = a->H; p.W = a->W; p.Cx = dg ? a->Cy : a->Cx;
    p.tilesX = (a->W + tile_c - 1) / tile_c; p.tilesY = (a->H + tile_r - 1) / tile_r;
    p.items = a->N * a->D * p.tilesY * p.tilesX;
    p.per_wg = (p.items + target_wgs - 1) / target_wgs;
    if (p.per_wg < 1) p.per_wg = 1;
}

// Floats of partial-sum workspace the WGRAD needs (one row of dW + db per workgroup); the caller owns it (SavpConvArgs.ws).
static long long thin_wgrad_ws_floats(const SavpConvArgs* a) {
    ThinP p;
    thin_fill(p, a, TW_R, TW_C, 512);
    const int nwg = (p.items + p.per_wg - 1) / p.per_wg;
    return (long long)nwg * (9 * a->kd * a->Cx * 32 + 32);
}

// wide -> thin FPROP (wthin_fprop_kernel): 2-D 3x3 stride-1 SAME, fp32 tensors, Cx % 8 == 0 <= 64, Cy <= 8, packed bf16 weights at hand
static bool wthin_applies(const SavpConvArgs* a) {
    if (!thin_enabled() || a->mode != SAVP_CONV_FPROP || a->precision != SAVP_PREC_BF16) return false;
    if (!(a->kd == 1 && a->kh == 3 && a->kw == 3 && a->pd == 0 && a->ph == 1 && a->pw == 1 && a->sd == 1 && a->sh == 1 && a->sw == 1 && a->D == 1 &&
          a->Do == 1 && a->Ho == a->H && a->Wo == a->W && a->N >= 1 && a->H >= 1 && a->W >= 1))
        return false;
    if (a->Cx % 8 || a->Cx < 16 || a->Cx > 64 || a->Cy < 1 || a->Cy > 8) return false;
    if (a->src_bf16 || a->out_bf16 || a->stats || a->beta || a->aux || !a->w_bf16 || !aligned16(a->w_bf16)) return false;
    if (a->act != SAVP_ACT_NONE && a->act != SAVP_ACT_LRELU && a->act != SAVP_ACT_SIGMOID) return false;
    if ((a->x_sn % 4) || (a->x_sh % 4) || (a->x_sw % 4) || !aligned16(a->x)) return false;
    return (long long)a->N * a->H * a->W < (1ll << 31) / 64;
}

// thin -> wide DGRAD with 8 thin channels (thin8_wide_kernel): 2-D 3x3 stride-1 SAME, fp32 tensors, Cy == 8, Cx <= 64, beta 0 / 1, packed bf16 weights
static bool thin8_applies(const SavpConvArgs* a) {
    if (!thin_enabled() || a->mode != SAVP_CONV_DGRAD || a->precision != SAVP_PREC_BF16) return false;
    if (!(a->kd == 1 && a->kh == 3 && a->kw == 3 && a->pd == 0 && a->ph == 1 && a->pw == 1 && a->sd == 1 && a->sh == 1 && a->sw == 1 && a->D == 1 &&
          a->Do == 1 && a->Ho == a->H && a->Wo == a->W && a->N >= 1 && a->H >= 1 && a->W >= 1))
        return false;
    if (a->Cy != 8 || a->Cx < 1 || a->Cx > 64) return false;
    if (a->src_bf16 || a->out_bf16 || a->stats || a->aux || a->bias || a->act != SAVP_ACT_NONE || !a->w_bf16 || !aligned16(a->w_bf16)) return false;
    if ((a->y_sn % 4) || (a->y_sh % 4) || (a->y_sw % 4) || !aligned16(a->y)) return false;
    return (long long)a->N * a->H * a->W < (1ll << 31) / 64;
}

// Is this call the kernel's problem (everything except the workspace)?
static bool thin_applies_geom(const SavpConvArgs* a) {
    if (!thin_enabled() || !thin_geometry_ok(a)) return false;
    const long long px = (long long)a->N * a->D * a->H * a->W;
    if (px >= (1ll << 31) / 64) return false;
    if (a->mode == SAVP_CONV_FPROP || a->mode == SAVP_CONV_DGRAD)
        return !(a->beta || a->aux || (a->act != SAVP_ACT_NONE && a->act != SAVP_ACT_LRELU));
    if (a->mode == SAVP_CONV_WGRAD)
        return !((a->y_sn % 4) || (a->y_sd % 4) || (a->y_sh % 4) || (a->y_sw % 4) || !aligned16(a->y));
    return false;
}

long long conv_thin_workspace_bytes(const SavpConvArgs* a) {
    return (thin_applies_geom(a) && a->mode == SAVP_CONV_WGRAD) ? thin_wgrad_ws_floats(a) * (long long)sizeof(float) : 0;
}

bool conv_thin_applies(const SavpConvArgs* a) {
    if (wthin_applies(a) || thin8_applies(a)) return true;
    if (!thin_applies_geom(a)) return false;
    // the weight gradient leaves one partial dW per workgroup in caller-owned scratch; without it the general kernel runs
    return a->mode != SAVP_CONV_WGRAD || (a->ws && aligned16(a->ws) && a->ws_bytes >= conv_thin_workspace_bytes(a));
}

// Returns true when the call was handled (rc set); false = not this kernel's problem, the caller goes on to the general kernels.
bool conv_thin_try(const SavpConvArgs* a, hipStream_t st, int* rc) {
    if (!conv_thin_applies(a)) return false;
    static const float* zero_of[64] = {nullptr};               // per device ordinal: a device symbol has one address per device
    int dev_ord = 0;
    if (hipGetDevice(&dev_ord) != hipSuccess || dev_ord < 0 || dev_ord >= 64) dev_ord = 0;
    if (!zero_of[dev_ord] && hipGetSymbolAddress((void**)&zero_of[dev_ord], HIP_SYMBOL(g_thin_zero)) != hipSuccess) { *rc = SAVP_ELAUNCH; return true; }
    ThinP p;
    p.zero = zero_of[dev_ord];
    if (thin8_applies(a)) {
        thin_fill(p, a, TF_R, TF_C, 512);                     // ThinP.x = dy (8 channels), ThinP.y = dx (wide); two resident workgroups per CU
        p.w16 = (const unsigned short*)a->w_bf16; p.Cy = a->Cx; p.flip = 1 | (a->beta ? 2 : 0);
        const dim3 grid((unsigned)((p.items + p.per_wg - 1) / p.per_wg));
        hipLaunchKernelGGL(thin8_wide_kernel, grid, dim3(256), 0, st, p);
        *rc = hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
        return true;
    }
    if (wthin_applies(a)) {
        thin_fill(p, a, WF_R, TF_C, 1024);                    // four rows x 32 pixels per item; two to four resident workgroups per CU
        p.w16 = (const unsigned short*)a->w_bf16; p.Cy = a->Cy; p.bias = a->bias; p.act = a->act; p.alpha = a->alpha;
        const dim3 grid((unsigned)((p.items + p.per_wg - 1) / p.per_wg));
        const int nkt = a->Cx <= 32 ? 2 : 4;
        const size_t lds = (size_t)(WF_PR * TF_PC) * (32 * nkt + 16) + (nkt > 2 ? (size_t)9 * nkt * 64 * 16 : 0);
        if (nkt == 2) hipLaunchKernelGGL((wthin_fprop_kernel<2>), grid, dim3(256), lds, st, p);
        else {
            static bool attr = false;
            if (!attr) { hipFuncSetAttribute((const void*)wthin_fprop_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
            hipLaunchKernelGGL((wthin_fprop_kernel<4>), grid, dim3(256), lds, st, p);
        }
        *rc = hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
        return true;
    }
    if (a->mode == SAVP_CONV_FPROP || a->mode == SAVP_CONV_DGRAD) {
        thin_fill(p, a, TF_R, TF_C, 1024);                    // four resident workgroups per CU: one full wave of them
        p.w = (const float*)a->w; p.bias = a->bias; p.act = a->act; p.alpha = a->alpha;
        const dim3 grid((unsigned)((p.items + p.per_wg - 1) / p.per_wg));
#define THIN_F(KD_, CX_) hipLaunchKernelGGL((thin_fprop_kernel<KD_, CX_>), grid, dim3(256), 0, st, p)
        if (a->kd == 3) { if (p.Cx == 3) THIN_F(3, 3); else if (p.Cx == 1) THIN_F(3, 1); else THIN_F(3, 4); }
        else { if (p.Cx == 3) THIN_F(1, 3); else if (p.Cx == 1) THIN_F(1, 1); else THIN_F(1, 4); }
#undef THIN_F
    } else if (a->mode == SAVP_CONV_WGRAD) {
        thin_fill(p, a, TW_R, TW_C, 512);                     // two resident workgroups per CU (VGPRs): one full wave of them
        p.dw = (float*)a->w; p.db = (float*)a->bias;
        const int nwg = (p.items + p.per_wg - 1) / p.per_wg;
        const int ndw = 9 * a->kd * a->Cx * 32, row = ndw + 32;
        // partial-sum workspace: caller-owned scratch (SavpConvArgs.ws; 5.4 MB for the BAIR layer), fully written before it is read
        float* ws = (float*)a->ws;
        p.ws = ws;
        const dim3 grid((unsigned)nwg);
        const size_t lds = (size_t)((((a->kd * TW_PR * TW_PC + 1) * 8 + 15) & ~15) + 256 * 64);
#define THIN_W(KD_, CX_) hipLaunchKernelGGL((thin_wgrad_kernel<KD_, CX_>), grid, dim3(256), lds, st, p)
        if (a->kd == 3) { if (a->Cx == 3) THIN_W(3, 3); else if (a->Cx == 1) THIN_W(3, 1); else THIN_W(3, 4); }
        else { if (a->Cx == 3) THIN_W(1, 3); else if (a->Cx == 1) THIN_W(1, 1); else THIN_W(1, 4); }
#undef THIN_W
        hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((unsigned)((row + 31) / 32)), dim3(256), 0, st, (const float*)ws, nwg, row, ndw, p.dw, p.db);
    } else {
        return false;
    }
    *rc = hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
    return true;
}
