// conv_s2dgrad.hip -- data gradient of a 4x4 stride-2 (spatial), stride-1 (depth) convolution whose INPUT has 32 channels:
// the second layer of the video / image discriminators (networks.py:35-108: conv k4 s(1,2,2) 32 -> 64), i.e. the gradient that
// flows into the first layer's 64x64x32 activation (bf16 mode).
//
// Why its own kernel.  dx is 168 MB (+168 MB of saved activation for the LeakyReLU backward, +75 MB of dy) for 77 GFLOP: bound by
// HBM at ~110-160 us, but the general kernels take 520-620 us.  They run the four output phases of the transposed convolution as
// four independent stride-1 problems (each re-stages the same dy patch) and pad the 32 output channels to a 64-wide tile.
// Here ONE workgroup owns a 16 x 32 tile of dx of one (sample, plane): it stages the 10 x 18 dy pixels under the tile once per depth
// tap, and each of its four waves computes one output PHASE (row parity, column parity) -- 8 x 16 pixels = four 32-row MFMA tiles
// that all use the same 2 x 2 taps, so a wave's B fragments (its taps' weights, 32 output channels = one tile, no padding) are
// loaded once per plane and reused by the four tiles.  A fragments are aligned ds_read_b128 at tap-shifted addresses of the staged
// plane (pixel stride 16 B x odd).  Epilogue: + old value (beta), x LeakyReLU' from the saved activation, full 128-byte pixel rows.
//
//   dx[n, z, Y, X, c] = sum_{a, u, v, co} dy[n, z + pd - a, (Y + 1 - u) / 2, (X + 1 - v) / 2, co] * W[a, u, v, c, co]
//   (u, v restricted to the parity that makes the divisions exact; pd = depth pad, spatial pad 1)
#include "conv_common.h"
#include <stdlib.h>

struct S2P {
    const float* dy; long long y_sn, y_sd, y_sh, y_sw;     // [N, Do, Ho, Wo, Cy]
    float* dx; long long x_sn, x_sd, x_sh, x_sw;           // [N, D, H, W, 32]
    const unsigned short* w16;                             // WD bf16 [32][kd * 16 * Cy]  (row c, then (a, u, v), then co)
    const float* bias; const float* aux;
    int beta, act; float alpha;
    int N, D, H, W, Do, Ho, Wo, Cy, kd, pd;
    int tilesX, tilesY;
    const float* zero;                                     // 16 bytes of zeros in global memory (source of out-of-range pixels)
};

__device__ float4 g_s2_zero[1] = {{0.f, 0.f, 0.f, 0.f}};     // reached through S2P.zero (see conv_thin.hip for why not directly)

#define S2_TR 16                      // dx tile rows / columns
#define S2_TC 32
#define S2_PR (S2_TR / 2 + 2)         // dy patch rows / columns under the tile
#define S2_PC (S2_TC / 2 + 2)

// CQ = Cy / 4 (float4 quads per dy pixel): 16 for the 64-channel layer.  KS = Cy / 16 k-steps per tap.
template <int CY>
__global__ __launch_bounds__(256, 2) void s2dgrad_kernel(S2P p) {
    constexpr int CQ = CY / 4, KS = CY / 16;
    constexpr int PSTR = CY * 2 + 16;                          // bytes per staged pixel: 16 x odd -> conflict-free b128 reads
    constexpr int NPX = S2_PR * S2_PC;                         // 180 pixels per plane
    constexpr int NSL = (NPX * CQ + 255) / 256;                // float4 slots per thread
    __shared__ __attribute__((aligned(16))) char patch[NPX * PSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int py = wave >> 1, px = wave & 1;                   // this wave's output phase
    int it = blockIdx.x;
    const int tx = it % p.tilesX; it /= p.tilesX;
    const int ty = it % p.tilesY; it /= p.tilesY;
    const int z = it % p.D, n = it / p.D;
    const int Y0 = ty * S2_TR, X0 = tx * S2_TC;
    const int oy0 = Y0 / 2 - 1, ox0 = X0 / 2 - 1;              // dy coordinates of patch pixel (0, 0)

    // taps of this phase: u_j = py ? 2 j : 2 j + 1, patch-row shift (py + 1 - u_j) / 2 + 1 (and the same for columns)
    int ut[2], vt[2], shr[2], shc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        ut[j] = py ? 2 * j : 2 * j + 1; shr[j] = (py + 1 - ut[j]) / 2 + 1;
        vt[j] = px ? 2 * j : 2 * j + 1; shc[j] = (px + 1 - vt[j]) / 2 + 1;
    }
    // A: lane row m = l31 -> pixel (ry = 2 t + (m >> 4), cx = m & 15) of row tile t; k = 8 h + e -> channel 16 ks + 8 h + e
    const int a_lane = (((l31 >> 4) * S2_PC) + (l31 & 15)) * PSTR + 16 * h;

    float4 pv[NSL];
    auto fetch = [&](int od) {                                 // dy plane od of sample n -> registers (zero outside the tensor)
        const bool plane_ok = od >= 0 && od < p.Do;
        const float* __restrict__ src = p.dy + (long long)n * p.y_sn + (long long)(plane_ok ? od : 0) * p.y_sd;
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int s = tid + 256 * i;
            const int pix = s / CQ, cq = s - pix * CQ;
            const int lr = pix / S2_PC, lc = pix - lr * S2_PC;
            const int oy = oy0 + lr, ox = ox0 + lc;
            const bool ok = plane_ok && s < NPX * CQ && oy >= 0 && oy < p.Ho && ox >= 0 && ox < p.Wo;
            pv[i] = ldg4(ok ? src + (long long)oy * p.y_sh + (long long)ox * p.y_sw + 4 * cq : p.zero);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int s = tid + 256 * i;
            const int pix = s / CQ, cq = s - pix * CQ;
            if (s < NPX * CQ)
                *reinterpret_cast<bf16x4*>(patch + pix * PSTR + cq * 8) = bf16x4{(__bf16)pv[i].x, (__bf16)pv[i].y, (__bf16)pv[i].z, (__bf16)pv[i].w};
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int ldw = p.kd * 16 * CY;                            // weight row length (elements)
    const unsigned short* __restrict__ wrow = p.w16 + (long long)l31 * ldw + 8 * h;
    for (int a = 0; a < p.kd; ++a) {
        const int od = z + p.pd - a;
        if (od < 0 || od >= p.Do) continue;                    // uniform: this depth tap falls outside dy
        // This phase's 4 taps x KS k-steps of weights for depth tap a are 16-byte loads that hit L2.  The first row of taps goes out
        // in front of the dy plane and is waited for together with it; the second row is issued when the plane's registers are
        // free again (after the staging) and arrives under the first row's MFMAs.  (All 16 fragments + 64 accumulators + the 48
        // plane registers at once spill; a register prefetch of the next plane across the MFMAs spilled 58 VGPRs -- the second
        // resident workgroup of the CU covers the plane's round trip instead.)
        bf16x8 bwa[2 * KS], bwb[2 * KS];
#pragma unroll
        for (int jv = 0; jv < 2; ++jv)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                bwa[jv * KS + ks] = *reinterpret_cast<const bf16x8*>(wrow + ((a * 4 + ut[0]) * 4 + vt[jv]) * CY + 16 * ks);
        fetch(od);
        __syncthreads();                                       // the previous plane's reads are done
        stage();
#pragma unroll
        for (int jv = 0; jv < 2; ++jv)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                bwb[jv * KS + ks] = *reinterpret_cast<const bf16x8*>(wrow + ((a * 4 + ut[1]) * 4 + vt[jv]) * CY + 16 * ks);
        __syncthreads();
#pragma unroll
        for (int ju = 0; ju < 2; ++ju)
#pragma unroll
            for (int jv = 0; jv < 2; ++jv)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int base = a_lane + (shr[ju] * S2_PC + shc[jv]) * PSTR + 32 * ks;
                    const bf16x8 bf = ju ? bwb[jv * KS + ks] : bwa[jv * KS + ks];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(patch + base + 2 * t * S2_PC * PSTR);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[t], 0, 0, 0);
                    }
                }
    }

    // ---- epilogue: accumulator row m of tile t -> pixel (Y0 + py + 2 (2 t + (m >> 4)), X0 + px + 2 (m & 15)), column = channel l31.
    //      The old values (beta) and the saved activations (LeakyReLU') of a tile are loaded as two batches from clamped addresses
    //      (out-of-range rows read the plane's first pixel and are not stored): per-element `if (beta) load` serialises 32 round trips.
    float bias = 0.f;
    if (p.bias) bias = p.bias[l31];
    const long long plane = (long long)n * p.x_sn + (long long)z * p.x_sd + l31;
    float* __restrict__ dxp = p.dx + plane;
    const float* __restrict__ auxp = p.aux ? p.aux + plane : nullptr;
    const bool use_old = p.beta != 0, use_aux = p.act == SAVP_ACT_DLRELU_FROM_OUT;
    const bool full = Y0 + S2_TR <= p.H && X0 + S2_TC <= p.W;      // whole tile inside the plane (uniform): unconditional stores
#pragma unroll
    for (int tp = 0; tp < 2; ++tp) {                               // two row tiles per batch: 64 loads in flight, then 32 stores
        int off[32];                                               // element offsets inside the (sample, plane): < 2^31 (launcher)
        bool ok[32];
        float old[32], ax[32], vout[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int t = 2 * tp + (i >> 4), r = i & 15;
            const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int Y = Y0 + py + 2 * (2 * t + (m >> 4)), X = X0 + px + 2 * (m & 15);
            ok[i] = Y < p.H && X < p.W;
            off[i] = ok[i] ? (int)(Y * p.x_sh + X * p.x_sw) : 0;
        }
        if (use_old) {
#pragma unroll
            for (int i = 0; i < 32; ++i) old[i] = dxp[off[i]];
        }
        if (use_aux) {
#pragma unroll
            for (int i = 0; i < 32; ++i) ax[i] = auxp[off[i]];
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            float v = acc[2 * tp + (i >> 4)][i & 15] + bias;
            if (use_old) v += old[i];
            if (use_aux) v *= (ax[i] > 0.f ? 1.f : p.alpha);
            vout[i] = v;
        }
        // values first, stores after: a store inside `if (ok)` next to the use of a loaded value makes hipcc wait for vmcnt(0) --
        // i.e. for the previous store's acknowledgement -- in front of every single store
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(vout[i]));
        if (full) {
#pragma unroll
            for (int i = 0; i < 32; ++i) dxp[off[i]] = vout[i];
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (ok[i]) dxp[off[i]] = vout[i];
        }
    }
}

// Is this call the kernel's problem?  (option "s2dgrad" = 0 switches the kernel off: A/B against the general kernels)
bool conv_s2dgrad_applies(const SavpConvArgs* a) {
    if (!savp_opt(OPT_S2DGRAD) || a->mode != SAVP_CONV_DGRAD || a->precision != SAVP_PREC_BF16 || !a->w_bf16) return false;
    if (!(a->Cx == 32 && (a->Cy == 64 || a->Cy == 32) && a->kh == 4 && a->kw == 4 && a->sh == 2 && a->sw == 2 && a->ph == 1 && a->pw == 1 &&
          a->sd == 1 && a->kd >= 1 && a->kd <= 4 && a->H == 2 * a->Ho && a->W == 2 * a->Wo && a->Do == a->D + 2 * a->pd - a->kd + 1 &&
          !a->src_bf16 && !a->out_bf16 && !a->stats && (a->beta == 0 || a->beta == 1) &&
          (a->act == SAVP_ACT_NONE || (a->act == SAVP_ACT_DLRELU_FROM_OUT && a->aux))))
        return false;
    if ((a->y_sn % 4) || (a->y_sd % 4) || (a->y_sh % 4) || (a->y_sw % 4) || !aligned16(a->y) || !aligned16(a->w_bf16)) return false;
    const long long items = (long long)a->N * a->D * ((a->H + S2_TR - 1) / S2_TR) * ((a->W + S2_TC - 1) / S2_TC);
    if (items < 1 || items >= (1ll << 31)) return false;
    if ((long long)a->H * a->x_sh + (long long)a->W * a->x_sw >= (1ll << 31)) return false;      // 32-bit offsets inside a plane
    return true;
}

// Returns true when the call was handled (rc set); false = not this kernel's problem.
bool conv_s2dgrad_try(const SavpConvArgs* a, hipStream_t st, int* rc) {
    if (!conv_s2dgrad_applies(a)) return false;
    static const float* zero = nullptr;
    if (!zero && hipGetSymbolAddress((void**)&zero, HIP_SYMBOL(g_s2_zero)) != hipSuccess) { *rc = SAVP_ELAUNCH; return true; }
    S2P p;
    p.dy = (const float*)a->y; p.y_sn = a->y_sn; p.y_sd = a->y_sd; p.y_sh = a->y_sh; p.y_sw = a->y_sw;
    p.dx = (float*)a->x; p.x_sn = a->x_sn; p.x_sd = a->x_sd; p.x_sh = a->x_sh; p.x_sw = a->x_sw;
    p.w16 = (const unsigned short*)a->w_bf16; p.bias = a->bias; p.aux = a->aux;
    p.beta = a->beta; p.act = a->act; p.alpha = a->alpha;
    p.N = a->N; p.D = a->D; p.H = a->H; p.W = a->W; p.Do = a->Do; p.Ho = a->Ho; p.Wo = a->Wo; p.Cy = a->Cy; p.kd = a->kd; p.pd = a->pd;
    p.tilesX = (a->W + S2_TC - 1) / S2_TC; p.tilesY = (a->H + S2_TR - 1) / S2_TR;
    p.zero = zero;
    const long long items = (long long)a->N * a->D * p.tilesY * p.tilesX;
    if (a->Cy == 64) hipLaunchKernelGGL(s2dgrad_kernel<64>, dim3((unsigned)items), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(s2dgrad_kernel<32>, dim3((unsigned)items), dim3(256), 0, st, p);
    *rc = hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
    return true;
}
