// conv_s2fprop.hip -- FPROP of a 4x4 stride-2 (spatial), stride-1 (depth) convolution with 32 input channels: the second layer of the
// video / image discriminators (networks.py:35-108: conv k4 s(1,2,2) 32 -> 64), bf16 mode.  Companion of conv_s2dgrad.hip.
//
// EXPERIMENTAL: written at the end of round 2 with no GPU time left to run it; off unless SAVP_S2FPROP=1 (the general kernels
// then never see the problem), tests behind SAVP_TEST_EXPERIMENTAL=1.
//
// The layer reads the first layer's 168 MB activation and writes 75 MB for 77 GFLOP: bound by HBM at ~65-80 us; the generic
// implicit-GEMM kernel (the winner of the sweep for this shape) takes 228 us -- every 64 x 64 output tile gathers its K = 2048
// operand rows from L2 through 4-byte-granular im2col addressing.  Here a workgroup owns 8 x 16 output pixels x 64 channels of one
// (sample, plane): per depth tap it parks the 18 x 34 input pixels under the tile in LDS as bf16 (pixel stride 16 B x odd), each
// wave multiplies two 32-pixel row tiles by one 32-channel column tile, A fragments are aligned ds_read_b128 at tap-shifted
// addresses of the patch (stride-2 pixel step), B fragments are 16-byte loads of the packed weights (L2 hits), one tap row (4 taps
// x 2 k-steps) at a time in two alternating register sets.  Epilogue: bias + LeakyReLU, full 128-byte channel rows.
//
//   y[n, od, oy, ox, co] = sum_{a, u, v, c} x[n, od - pd + a, 2 oy - 1 + u, 2 ox - 1 + v, c] * W[a, u, v, c, co]
#include "conv_common.h"
#include <stdlib.h>

struct S2F {
    const float* x; long long x_sn, x_sd, x_sh, x_sw;      // [N, D, H, W, 32]
    float* y; long long y_sn, y_sd, y_sh, y_sw;            // [N, Do, Ho, Wo, 64]
    const unsigned short* w16;                             // WT bf16 [64][kd * 16 * 32]  (row co, then (a, u, v), then c)
    const float* bias;
    int act; float alpha;
    int N, D, H, W, Do, Ho, Wo, kd, pd;
    int tilesX, tilesY;
    const float* zero;
};

__device__ float4 g_s2f_zero[1] = {{0.f, 0.f, 0.f, 0.f}};

#define SF_TR 8                       // output tile rows / columns
#define SF_TC 16
#define SF_PR (2 * SF_TR + 2)         // input patch rows / columns under the tile
#define SF_PC (2 * SF_TC + 2)
#define SF_PSTR 80                    // bytes per staged pixel: 32 bf16 channels + 16 (16 B x odd)

__global__ __launch_bounds__(256, 2) void s2fprop_kernel(S2F p) {
    constexpr int NPX = SF_PR * SF_PC;                         // 612 pixels per plane
    constexpr int NSL = (NPX * 8 + 255) / 256;                 // float4 slots per thread (8 channel quads per pixel)
    __shared__ __attribute__((aligned(16))) char patch[NPX * SF_PSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int mt0 = (wave >> 1) * 2, nt = wave & 1;            // this wave: row tiles mt0, mt0 + 1 (2 output rows each), column tile nt
    int it = blockIdx.x;
    const int tx = it % p.tilesX; it /= p.tilesX;
    const int ty = it % p.tilesY; it /= p.tilesY;
    const int od = it % p.Do, n = it / p.Do;
    const int oy0 = ty * SF_TR, ox0 = tx * SF_TC;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;            // input coordinates of patch pixel (0, 0)

    // A: lane row m = l31 -> output pixel (row 2 mt + (m >> 4), column m & 15) -> patch pixel (2 row + u, 2 column + v)
    const int a_lane = ((2 * (l31 >> 4)) * SF_PC + 2 * (l31 & 15)) * SF_PSTR + 16 * h;
    const int ldw = p.kd * 16 * 32;                            // weight row length (elements)
    const unsigned short* __restrict__ wrow = p.w16 + (long long)(32 * nt + l31) * ldw + 8 * h;

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int a = 0; a < p.kd; ++a) {
        const int iz = od - p.pd + a;
        if (iz < 0 || iz >= p.D) continue;                     // uniform: this depth tap falls outside x
        // first tap row's weights go out in front of the plane and are waited for together with it (vmcnt counts in order)
        bf16x8 bw[2][8];
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                bw[0][v * 2 + ks] = *reinterpret_cast<const bf16x8*>(wrow + ((a * 4 + 0) * 4 + v) * 32 + 16 * ks);
        {
            const float* __restrict__ src = p.x + (long long)n * p.x_sn + (long long)iz * p.x_sd;
            float4 pv[NSL];
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int s = tid + 256 * i;
                const int pix = s >> 3, cq = s & 7;
                const int lr = pix / SF_PC, lc = pix - lr * SF_PC;
                const int iy = iy0 + lr, ix = ix0 + lc;
                const bool ok = s < NPX * 8 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                pv[i] = ldg4(ok ? src + (long long)iy * p.x_sh + (long long)ix * p.x_sw + 4 * cq : p.zero);
            }
            __syncthreads();                                   // the previous plane's reads are done
#pragma unroll
            for (int i = 0; i < NSL; ++i) {
                const int s = tid + 256 * i;
                if (s < NPX * 8)
                    *reinterpret_cast<bf16x4*>(patch + (s >> 3) * SF_PSTR + (s & 7) * 8) =
                        bf16x4{(__bf16)pv[i].x, (__bf16)pv[i].y, (__bf16)pv[i].z, (__bf16)pv[i].w};
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u + 1 < 4) {                                   // next tap row's weights fly under this row's MFMAs
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        bw[(u + 1) & 1][v * 2 + ks] = *reinterpret_cast<const bf16x8*>(wrow + ((a * 4 + u + 1) * 4 + v) * 32 + 16 * ks);
            }
            __builtin_amdgcn_sched_barrier(0);                 // all eight loads first: left alone hipcc issues each one ~4 MFMAs ahead of its use
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int base = a_lane + (u * SF_PC + v) * SF_PSTR + 32 * ks;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(patch + base + 4 * (mt0 + t) * SF_PC * SF_PSTR);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bw[u & 1][v * 2 + ks], acc[t], 0, 0, 0);
                    }
                }
        }
    }

    // ---- epilogue: accumulator row m of tile t -> output pixel (oy0 + 2 (mt0 + t) + (m >> 4), ox0 + (m & 15)), column = channel
    const int co = 32 * nt + l31;
    float bias = 0.f;
    if (p.bias) bias = p.bias[co];
    float* __restrict__ yp = p.y + (long long)n * p.y_sn + (long long)od * p.y_sd + co;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int oy = oy0 + 2 * (mt0 + t) + (m >> 4), ox = ox0 + (m & 15);
            if (oy < p.Ho && ox < p.Wo) {
                float v = acc[t][r] + bias;
                if (p.act == SAVP_ACT_LRELU) v = fmaxf(v, p.alpha * v);
                yp[(long long)oy * p.y_sh + (long long)ox * p.y_sw] = v;
            }
        }
}

static int s2f_mode() {                // experimental: on only with SAVP_S2FPROP=1
    static int v = -1;
    if (v < 0) { const char* e = getenv("SAVP_S2FPROP"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}

// Returns true when the call was handled (rc set); false = not this kernel's problem.
bool conv_s2fprop_try(const SavpConvArgs* a, hipStream_t st, int* rc) {
    if (!s2f_mode() || a->mode != SAVP_CONV_FPROP || a->precision != SAVP_PREC_BF16 || !a->w_bf16) return false;
    if (!(a->Cx == 32 && a->Cy == 64 && a->kh == 4 && a->kw == 4 && a->sh == 2 && a->sw == 2 && a->ph == 1 && a->pw == 1 &&
          a->sd == 1 && a->kd >= 1 && a->kd <= 4 && a->H == 2 * a->Ho && a->W == 2 * a->Wo && a->Do == a->D + 2 * a->pd - a->kd + 1 &&
          !a->src_bf16 && !a->out_bf16 && !a->stats && !a->beta && !a->aux &&
          (a->act == SAVP_ACT_NONE || a->act == SAVP_ACT_LRELU)))
        return false;
    if ((a->x_sn % 4) || (a->x_sd % 4) || (a->x_sh % 4) || (a->x_sw % 4) || !aligned16(a->x) || !aligned16(a->w_bf16)) return false;
    static const float* zero = nullptr;
    if (!zero && hipGetSymbolAddress((void**)&zero, HIP_SYMBOL(g_s2f_zero)) != hipSuccess) { *rc = SAVP_ELAUNCH; return true; }
    S2F p;
    p.x = (const float*)a->x; p.x_sn = a->x_sn; p.x_sd = a->x_sd; p.x_sh = a->x_sh; p.x_sw = a->x_sw;
    p.y = (float*)a->y; p.y_sn = a->y_sn; p.y_sd = a->y_sd; p.y_sh = a->y_sh; p.y_sw = a->y_sw;
    p.w16 = (const unsigned short*)a->w_bf16; p.bias = a->bias; p.act = a->act; p.alpha = a->alpha;
    p.N = a->N; p.D = a->D; p.H = a->H; p.W = a->W; p.Do = a->Do; p.Ho = a->Ho; p.Wo = a->Wo; p.kd = a->kd; p.pd = a->pd;
    p.tilesX = (a->Wo + SF_TC - 1) / SF_TC; p.tilesY = (a->Ho + SF_TR - 1) / SF_TR;
    p.zero = zero;
    const long long items = (long long)a->N * a->Do * p.tilesY * p.tilesX;
    if (items < 1 || items >= (1ll << 31)) return false;
    hipLaunchKernelGGL(s2fprop_kernel, dim3((unsigned)items), dim3(256), 0, st, p);
    *rc = hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
    return true;
}
