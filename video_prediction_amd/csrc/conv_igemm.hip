// conv_igemm.hip -- implicit-GEMM convolution for gfx950 on the fp32 MFMA pipe.
//
// One kernel family covers every dense contraction of the SAVP hot path (see include/savp_hip.h):
// ConvLSTM 5x5 gate convs (rnn_ops.py:121), conv_pool2d stride-2 convs (ops.py:844), upsample_conv2d
// (= conv2d_transpose, ops.py:707 -> DGRAD mode), 3x3 heads (ops.py:528), encoder 4x4 s2 (networks.py:18-25),
// the discriminators' conv2d/conv3d ladders (networks.py:45-102), dense layers as 1x1 convs (ops.py:12), and
// the data-/weight-gradients of all of them.
//
// Design (MI355X-first, not a port of anything):
//   * 256-thread workgroups = 4 wave64, 2x2 wave grid, each wave owns WMxWN tiles of 32x32 accumulated by
//     v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = the fp32 peak of the chip).
//   * the im2col gather happens on the fly while staging global -> registers -> LDS; the next K-tile's global
//     loads are issued before the MFMA loop of the current tile (register prefetch + double-buffered LDS, one
//     barrier per K-tile of 32).
//   * FPROP/DGRAD: LDS rows are [row][k] with k contiguous (+4 pad -> conflict-free ds_read_b128); each lane
//     reads 4 consecutive k for its row and feeds 4 MFMA k-steps (A and B use the same k permutation).
//   * DGRAD of strided convs is phase-decomposed (blockIdx.z = output phase) so no MAC is spent on the zeros
//     of the transposed convolution.
//   * WGRAD: K = all output pixels (time and batch folded in), split-K over blockIdx.z with fp32 atomics;
//     LDS is K-major so both operands are staged with float4 along their contiguous channel axis.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "savp_hip.h"
#include "zero_fill.h"

#include "conv_common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------------------
// FPROP / DGRAD kernel.  GEMM: C[M = grid pixels][N = dst channels] = A[M][K=(taps,Cred)] * B[K][N]
// ------------------------------------------------------------------------------------------------------------
// BF16 = true: operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while staging into LDS and multiplied on
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate); K-tile 64.  BF16 = false: exact fp32 on v_mfma_f32_32x32x2_f32; K-tile 32.
// WB16 = true (bf16 mode only): the weight operand is read from a pre-packed bf16 copy (16-byte loads of 8 k values,
// no conversion, half the L2 traffic and staging instructions of the fp32 stream).
template <int WM, int WN, bool VEC, bool BF16, bool WB16>
__global__ __launch_bounds__(NTHREADS) void conv_fd_kernel(ConvP p) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int BKT = BF16 ? 64 : 32;                // K-tile
    constexpr int KV = BKT / 4;                        // float4 columns per row
    constexpr int RP = NTHREADS / KV;                  // rows covered per pass
    constexpr int RA = BM / RP, RB = BN / RP;          // float4 rows fetched per thread for A / B
    constexpr int ROWB = BF16 ? (BKT + 8) * 2 : BKP * 4;   // LDS row stride in bytes (padded: conflict-free b128 reads)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = reinterpret_cast<char*>(smem);          // [2][BM] rows
    char* Bs = As + 2 * BM * ROWB;                     // [2][BN] rows

    const bool dgrad = (p.mode == SAVP_CONV_DGRAD);
    // blockIdx.z = phase * splitk + split ; phase decode (DGRAD only) -> (fd, fh, fw)
    const int split = blockIdx.z % p.splitk;
    int fz = blockIdx.z / p.splitk;
    int fw = dgrad ? fz % p.sw : 0; fz = dgrad ? fz / p.sw : 0;
    int fh = dgrad ? fz % p.sh : 0; fz = dgrad ? fz / p.sh : 0;
    int fd = fz;
    const DimGeom gd = make_geom(dgrad, fd, p.D, p.Do, p.kd, p.sd, p.pd);
    const DimGeom gh = make_geom(dgrad, fh, p.H, p.Ho, p.kh, p.sh, p.ph);
    const DimGeom gw = make_geom(dgrad, fw, p.W, p.Wo, p.kw, p.sw, p.pw);

    const int Cred = dgrad ? p.Cy : p.Cx;             // reduction channels
    const int Nout = dgrad ? p.Cx : p.Cy;             // destination channels
    const int ntaps = gd.nt * gh.nt * gw.nt;
    const int K = ntaps * Cred;
    const int ldb = p.kd * p.kh * p.kw * Cred;        // packed weight row length
    const int Mtot = p.N * gd.Mdim * gh.Mdim * gw.Mdim;
    // logical tile id: n-tile major so that the tiles sharing a weight block are consecutive (-> same XCD L2)
    const int tlog = xcd_logical(blockIdx.x, p.tm * p.tn);
    const int m0 = (tlog % p.tm) * BM;
    const int n0 = (tlog / p.tm) * BN;
    if (m0 >= Mtot) return;                            // uniform per workgroup (phase with fewer pixels)

    const float* __restrict__ src = dgrad ? p.y : p.x;
    const long long s_sn = dgrad ? p.y_sn : p.x_sn;
    const int s_sd = (int)(dgrad ? p.y_sd : p.x_sd), s_sh = (int)(dgrad ? p.y_sh : p.x_sh),
              s_sw = (int)(dgrad ? p.y_sw : p.x_sw);
    const float* __restrict__ wt = p.w;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int kv = tid % KV, r0 = tid / KV;

    // ---- per-thread A rows --------------------------------------------------------------------------------
    long long a_base[RA];
    int a_cd[RA], a_ch[RA], a_cw[RA];
    bool a_ok[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        int m = m0 + r0 + RP * j;
        a_ok[j] = m < Mtot;
        int mm = a_ok[j] ? m : 0;
        int qw = mm % gw.Mdim; mm /= gw.Mdim;
        int qh = mm % gh.Mdim; mm /= gh.Mdim;
        int qd = mm % gd.Mdim; int n = mm / gd.Mdim;
        a_base[j] = (long long)n * s_sn;
        a_cd[j] = gd.base + qd * gd.mstep;
        a_ch[j] = gh.base + qh * gh.mstep;
        a_cw[j] = gw.base + qw * gw.mstep;
    }
    // ---- per-thread B rows --------------------------------------------------------------------------------
    bool b_ok[RB];
    long long b_base[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        int n = n0 + r0 + RP * j;
        b_ok[j] = n < Nout;
        b_base[j] = (long long)(b_ok[j] ? n : 0) * ldb;
    }
    // ---- K state (vector path): this thread's float4 sits at k = kt*32 + kv*4 -------------------------------
    // split-K: this workgroup reduces K-tiles [kt_begin, kt_end)
    const int nk_all = (K + BKT - 1) / BKT;
    const int kt_per = (nk_all + p.splitk - 1) / p.splitk;
    const int kt_begin = split * kt_per;
    const int kt_end = min(nk_all, kt_begin + kt_per);
    int kc = 0, jd = 0, jh = 0, jw = 0;     // channel offset, reduced tap indices
    if (VEC) {
        int k = kt_begin * BKT + kv * 4;
        kc = k % Cred; int tap = k / Cred;
        jw = tap % max(gw.nt, 1); tap /= max(gw.nt, 1);
        jh = tap % max(gh.nt, 1); jd = tap / max(gh.nt, 1);
    }

    float4 ra[RA], rb[RB];
    // bf16 weight stream: thread -> (row r0b + 32*j, 8-wide k chunk kvb)
    constexpr int RB16 = BN / 32;
    const int kvb = tid & 7, r0b = tid >> 3;
    uint4 rb16[WB16 ? RB16 : 1];
    long long bb16[WB16 ? RB16 : 1];
    bool bok16[WB16 ? RB16 : 1];
    int kcb = 0, jdb = 0, jhb = 0, jwb = 0;
    if (WB16) {
#pragma unroll
        for (int j = 0; j < RB16; ++j) {
            int n = n0 + r0b + 32 * j;
            bok16[j] = n < Nout;
            bb16[j] = (long long)(bok16[j] ? n : 0) * ldb;
        }
        int k = kt_begin * BKT + kvb * 8;
        kcb = k % Cred; int tap = k / Cred;
        jwb = tap % max(gw.nt, 1); tap /= max(gw.nt, 1);
        jhb = tap % max(gh.nt, 1); jdb = tap / max(gh.nt, 1);
    }

    auto fetch = [&](int kt) {
        if (WB16) {
            const bool kokb = (jdb < gd.nt) && (ntaps > 0);
            const int ta = gd.t0 + jdb * gd.tstep, tu = gh.t0 + jhb * gh.tstep, tv = gw.t0 + jwb * gw.tstep;
            const long long woff = (long long)((ta * p.kh + tu) * p.kw + tv) * Cred + kcb;
#pragma unroll
            for (int j = 0; j < RB16; ++j) {
                const bool ok = bok16[j] && kokb;
                uint4 v = *reinterpret_cast<const uint4*>(p.w16 + (ok ? bb16[j] + woff : 0ll));
                rb16[j] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
            }
            kcb += BKT;
            while (kcb >= Cred) {
                kcb -= Cred;
                if (++jwb >= gw.nt) { jwb = 0; if (++jhb >= gh.nt) { jhb = 0; ++jdb; } }
            }
        }
        if (VEC) {
            const bool kok = (jd < gd.nt) && (ntaps > 0);
            const int zd0 = jd * gd.jstep, zh0 = jh * gh.jstep, zw0 = jw * gw.jstep;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                int zd = a_cd[j] + zd0, zh = a_ch[j] + zh0, zw = a_cw[j] + zw0;
                bool ok = a_ok[j] && kok && (unsigned)zd < (unsigned)gd.srcN && (unsigned)zh < (unsigned)gh.srcN &&
                          (unsigned)zw < (unsigned)gw.srcN;
                // branch-free: out-of-image taps read a safe address and are zeroed by a select, so all gathers of a K-tile
                // are issued back to back (a branch per load serialised them and cost an s_cbranch each)
                const long long off = ok ? (a_base[j] + (long long)zd * s_sd + (long long)zh * s_sh + (long long)zw * s_sw + kc) : 0ll;
                float4 v = ldg4(src + off);
                ra[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const int ta = gd.t0 + jd * gd.tstep, tu = gh.t0 + jh * gh.tstep, tv = gw.t0 + jw * gw.tstep;
            const long long woff = (long long)((ta * p.kh + tu) * p.kw + tv) * Cred + kc;
            if (!WB16) {
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const bool ok = b_ok[j] && kok;
                    float4 v = ldg4(wt + (ok ? b_base[j] + woff : 0ll));
                    rb[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            // advance by the K-tile
            kc += BKT;
            while (kc >= Cred) {
                kc -= Cred;
                if (++jw >= gw.nt) { jw = 0; if (++jh >= gh.nt) { jh = 0; ++jd; } }
            }
        } else {
            // scalar path: arbitrary Cred (first layers with 3/6/14 channels, the 53-channel mask conv)
            float av[RA][4], bv[RB][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int k = kt * BKT + kv * 4 + e;
                bool kok = k < K;
                int kk = kok ? k : 0;
                int c = kk % Cred; int tap = kk / Cred;
                int tw_ = tap % max(gw.nt, 1); tap /= max(gw.nt, 1);
                int th_ = tap % max(gh.nt, 1); int td_ = tap / max(gh.nt, 1);
#pragma unroll
                for (int j = 0; j < RA; ++j) {
                    int zd = a_cd[j] + td_ * gd.jstep, zh = a_ch[j] + th_ * gh.jstep, zw = a_cw[j] + tw_ * gw.jstep;
                    bool ok = a_ok[j] && kok && (unsigned)zd < (unsigned)gd.srcN && (unsigned)zh < (unsigned)gh.srcN &&
                              (unsigned)zw < (unsigned)gw.srcN;
                    av[j][e] = ok ? src[a_base[j] + (long long)zd * s_sd + (long long)zh * s_sh + (long long)zw * s_sw + c] : 0.f;
                }
                int ta = gd.t0 + td_ * gd.tstep, tu = gh.t0 + th_ * gh.tstep, tv = gw.t0 + tw_ * gw.tstep;
                long long woff = (long long)((ta * p.kh + tu) * p.kw + tv) * Cred + c;
#pragma unroll
                for (int j = 0; j < RB; ++j) bv[j][e] = (b_ok[j] && kok) ? wt[b_base[j] + woff] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < RA; ++j) ra[j] = make_float4(av[j][0], av[j][1], av[j][2], av[j][3]);
#pragma unroll
            for (int j = 0; j < RB; ++j) rb[j] = make_float4(bv[j][0], bv[j][1], bv[j][2], bv[j][3]);
        }
    };
    auto stage = [&](int buf) {
        char* a = As + buf * BM * ROWB;
        char* b = Bs + buf * BN * ROWB;
        if (BF16) {
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                bf16x4 v = {(__bf16)ra[j].x, (__bf16)ra[j].y, (__bf16)ra[j].z, (__bf16)ra[j].w};
                *reinterpret_cast<bf16x4*>(a + (r0 + RP * j) * ROWB + kv * 8) = v;
            }
            if (WB16) {
#pragma unroll
                for (int j = 0; j < RB16; ++j) *reinterpret_cast<uint4*>(b + (r0b + 32 * j) * ROWB + kvb * 16) = rb16[j];
            } else {
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    bf16x4 v = {(__bf16)rb[j].x, (__bf16)rb[j].y, (__bf16)rb[j].z, (__bf16)rb[j].w};
                    *reinterpret_cast<bf16x4*>(b + (r0 + RP * j) * ROWB + kv * 8) = v;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < RA; ++j) *reinterpret_cast<float4*>(a + (r0 + RP * j) * ROWB + kv * 16) = ra[j];
#pragma unroll
            for (int j = 0; j < RB; ++j) *reinterpret_cast<float4*>(b + (r0 + RP * j) * ROWB + kv * 16) = rb[j];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave >> 1) * 32 * WM, wn0 = (wave & 1) * 32 * WN;
    const int l31 = lane & 31, khalf = lane >> 5;
    // software pipeline: LDS holds tile kt (cur) while the registers hold tile kt+1 whose global loads were issued one
    // whole iteration earlier (right after the previous stage), so they fly under a barrier + a full MFMA block.
    if (kt_begin < kt_end) {
        fetch(kt_begin);
        stage(0);
        if (kt_begin + 1 < kt_end) fetch(kt_begin + 1);
    }
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const char* a = As + cur * BM * ROWB;
        const char* b = Bs + cur * BN * ROWB;
        if (BF16) {
#pragma unroll
            for (int ks = 0; ks < BKT / 16; ++ks) {
                bf16x8 af[WM], bf[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    af[i] = *reinterpret_cast<const bf16x8*>(a + (wm0 + i * 32 + l31) * ROWB + (ks * 16 + khalf * 8) * 2);
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    bf[j] = *reinterpret_cast<const bf16x8*>(b + (wn0 + j * 32 + l31) * ROWB + (ks * 16 + khalf * 8) * 2);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int c8 = 0; c8 < BKT / 8; ++c8) {
                float4 af[WM], bf[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    af[i] = *reinterpret_cast<const float4*>(a + (wm0 + i * 32 + l31) * ROWB + (c8 * 8 + khalf * 4) * 4);
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    bf[j] = *reinterpret_cast<const float4*>(b + (wn0 + j * 32 + l31) * ROWB + (c8 * 8 + khalf * 4) * 4);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }
        if (kt + 1 < kt_end) {
            stage(cur ^ 1);
            if (kt + 2 < kt_end) fetch(kt + 2);
        }
        __syncthreads();
    }

    // ---- epilogue: destination row offsets through LDS -------------------------------------------------------
    long long* rowoff = reinterpret_cast<long long*>(smem);     // [BM], -1 = invalid (safe: last barrier of the loop passed)
    float* __restrict__ dst = p.out;
    const long long d_sn = dgrad ? p.x_sn : p.y_sn;
    const long long d_sd = dgrad ? p.x_sd : p.y_sd, d_sh = dgrad ? p.x_sh : p.y_sh, d_sw = dgrad ? p.x_sw : p.y_sw;
    if (tid < BM) {
        int m = m0 + tid;
        long long off = -1;
        if (m < Mtot) {
            int mm = m;
            int qw = mm % gw.Mdim; mm /= gw.Mdim;
            int qh = mm % gh.Mdim; mm /= gh.Mdim;
            int qd = mm % gd.Mdim; int n = mm / gd.Mdim;
            off = (long long)n * d_sn + (long long)(gd.ob + qd * gd.os) * d_sd + (long long)(gh.ob + qh * gh.os) * d_sh +
                  (long long)(gw.ob + qw * gw.os) * d_sw;
        }
        rowoff[tid] = off;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = n0 + wn0 + j * 32 + l31;
        if (col >= Nout) continue;
        const float bias = (p.bias && split == 0) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const long long off = rowoff[row];
                if (off < 0) continue;
                float v = acc[i][j][r] + bias;
                float* q = dst + off + col;
                if (p.splitk > 1) {               // this split's share (conv_common.h: deterministic split-K)
                    (p.part + (long long)split * p.part_sz + (dst - p.out))[off + col] = v;
                    continue;
                }
                if (p.beta) v += *q;
                if (p.act == SAVP_ACT_LRELU) v = fmaxf(v, p.alpha * v);
                else if (p.act == SAVP_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                else if (p.act == SAVP_ACT_DLRELU_FROM_OUT) v *= (p.aux[off + col] > 0.f ? 1.f : p.alpha);
                *q = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// WGRAD kernel.  GEMM: dW[M=(tap,cx)][N=cy] += A[M][K=pixels] * B[K][N];  split-K over blockIdx.z.
// ------------------------------------------------------------------------------------------------------------
template <int WM, int WN, bool VECA, bool VECB>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_kernel(ConvP p) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int BMP = BM + 4, BNP = BN + 4;
    constexpr int AV = BM / 4, BV = BN / 4;          // float4 columns per k row
    constexpr int AK = NTHREADS / AV, BKK = NTHREADS / BV;   // k rows covered per pass
    constexpr int AP = BK / AK, BP = BK / BKK;       // passes
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                  // [2][BK][BMP]
    float* Bs = smem + 2 * BK * BMP;                   // [2][BK][BNP]

    const int M = p.kd * p.kh * p.kw * p.Cx;
    const int Nn = p.Cy;
    const int HWo = p.Ho * p.Wo, DHWo = p.Do * HWo;
    const int Ktot = p.N * DHWo;
    // 1-D grid: logical id = split * (tm*tn) + tile, so the tiles of one K split (same pixels) share an XCD L2
    const int llog = xcd_logical(blockIdx.x, p.tm * p.tn * p.splitk);
    const int zsplit = llog / (p.tm * p.tn), tl = llog % (p.tm * p.tn);
    const int m0 = (tl % p.tm) * BM, n0 = (tl / p.tm) * BN;
    // K range of this split (multiple of BK)
    const int ktiles = (Ktot + BK - 1) / BK;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt_begin = zsplit * per;
    const int kt_end = min(ktiles, kt_begin + per);
    if (kt_begin >= kt_end) return;

    const float* __restrict__ X = p.x;
    const float* __restrict__ Y = p.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // A mapping: fixed m-vector per thread
    const int amv = tid % AV, akk = tid / AV;
    int a_m = m0 + amv * 4;
    int a_c[4], a_td[4], a_th[4], a_tw[4]; bool a_mok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int m = a_m + e;
        a_mok[e] = m < M;
        int mm = a_mok[e] ? m : 0;
        a_c[e] = mm % p.Cx; int tap = mm / p.Cx;
        a_tw[e] = tap % p.kw; tap /= p.kw;
        a_th[e] = tap % p.kh; a_td[e] = tap / p.kh;
    }
    const int bnv = tid % BV, bkk = tid / BV;
    const int b_n = n0 + bnv * 4;

    float4 ra[AP], rb[BP];

    auto decode = [&](int pix, int& n, int& od, int& oy, int& ox) {
        unsigned up = (unsigned)pix;
        n = (int)fastdiv(up, p.magDHW); unsigned r = up - (unsigned)n * DHWo;
        od = (int)fastdiv(r, p.magHW); r -= (unsigned)od * HWo;
        oy = (int)fastdiv(r, p.magW); ox = (int)(r - (unsigned)oy * p.Wo);
    };
    auto fetch = [&](int kt) {
#pragma unroll
        for (int j = 0; j < AP; ++j) {
            int pix = kt * BK + akk + j * AK;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pix < Ktot) {
                int n, od, oy, ox; decode(pix, n, od, oy, ox);
                const long long nb = (long long)n * p.x_sn;
                if (VECA) {
                    int zd = od * p.sd - p.pd + a_td[0], zh = oy * p.sh - p.ph + a_th[0], zw = ox * p.sw - p.pw + a_tw[0];
                    if (a_mok[0] && (unsigned)zd < (unsigned)p.D && (unsigned)zh < (unsigned)p.H && (unsigned)zw < (unsigned)p.W)
                        v = ldg4(X + nb + zd * p.x_sd + zh * p.x_sh + zw * p.x_sw + a_c[0]);
                } else {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int zd = od * p.sd - p.pd + a_td[e], zh = oy * p.sh - p.ph + a_th[e], zw = ox * p.sw - p.pw + a_tw[e];
                        bool ok = a_mok[e] && (unsigned)zd < (unsigned)p.D && (unsigned)zh < (unsigned)p.H && (unsigned)zw < (unsigned)p.W;
                        t[e] = ok ? X[nb + zd * p.x_sd + zh * p.x_sh + zw * p.x_sw + a_c[e]] : 0.f;
                    }
                    v = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            int pix = kt * BK + bkk + j * BKK;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pix < Ktot) {
                int n, od, oy, ox; decode(pix, n, od, oy, ox);
                const float* q = Y + (long long)n * p.y_sn + od * p.y_sd + oy * p.y_sh + ox * p.y_sw;
                if (VECB) {
                    if (b_n < Nn) v = ldg4(q + b_n);
                } else {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = (b_n + e < Nn) ? q[b_n + e] : 0.f;
                    v = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
            rb[j] = v;
        }
    };
    auto stage = [&](int buf) {
        float* a = As + buf * BK * BMP;
        float* b = Bs + buf * BK * BNP;
#pragma unroll
        for (int j = 0; j < AP; ++j) *reinterpret_cast<float4*>(a + (akk + j * AK) * BMP + amv * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < BP; ++j) *reinterpret_cast<float4*>(b + (bkk + j * BKK) * BNP + bnv * 4) = rb[j];
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave >> 1) * 32 * WM, wn0 = (wave & 1) * 32 * WN;
    const int l31 = lane & 31, khalf = lane >> 5;

    fetch(kt_begin);
    stage(0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) fetch(kt + 1);
        const float* a = As + cur * BK * BMP;
        const float* b = Bs + cur * BK * BNP;
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2) {
            float af[WM], bf[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) af[i] = a[(k2 * 2 + khalf) * BMP + wm0 + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[j] = b[(k2 * 2 + khalf) * BNP + wn0 + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < kt_end) stage(cur ^ 1);
        __syncthreads();
    }

    // deterministic weight gradient (p.part): this split's tile goes to its own slice of the caller's scratch with plain stores and
    // wgrad_fold adds the slices to dW in split order; without scratch (a C-ABI caller that passed none) the splits meet in dW atomically
    const bool sliced = p.part != nullptr;
    float* __restrict__ dW = sliced ? p.part + (long long)zsplit * p.part_sz : p.out;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = n0 + wn0 + j * 32 + l31;
        if (col >= Nn) continue;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (row < M) {
                    if (sliced) dW[(long long)row * Nn + col] = acc[i][j][r];
                    else unsafeAtomicAdd(dW + (long long)row * Nn + col, acc[i][j][r]);
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// WGRAD, bf16 operands (fp32 accumulate).  Same GEMM as above, K-tile = 64 pixels.  Both operands are pixel-major in
// HBM, the MFMA wants k(=pixel)-contiguous fragments: each thread loads a 4(pixel) x 4(channel) fp32 block (float4 per
// pixel), transposes it in registers, rounds to bf16 and writes four 8-byte rows into the row-major-K LDS tile
// (lanes run along k first -> conflict-free ds_write_b64).  Requires Cx % 4 == 0 and Cy % 4 == 0.
// ------------------------------------------------------------------------------------------------------------
template <int WM, int WN>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_bf16_kernel(ConvP p) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int BKT = 64;
    constexpr int ROWB = (BKT + 8) * 2;
    constexpr int PA = BM / 64, PB = BN / 64;          // passes of 16 row-quads
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* As = reinterpret_cast<char*>(smem);
    char* Bs = As + 2 * BM * ROWB;

    const int M = p.kd * p.kh * p.kw * p.Cx;
    const int Nn = p.Cy;
    const int HWo = p.Ho * p.Wo, DHWo = p.Do * HWo;
    const int Ktot = p.N * DHWo;
    const int llog = xcd_logical(blockIdx.x, p.tm * p.tn * p.splitk);
    const int zsplit = llog / (p.tm * p.tn), tl = llog % (p.tm * p.tn);
    const int m0 = (tl % p.tm) * BM, n0 = (tl / p.tm) * BN;
    const int ktiles = (Ktot + BKT - 1) / BKT;
    const int per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt_begin = zsplit * per;
    const int kt_end = min(ktiles, kt_begin + per);
    if (kt_begin >= kt_end) return;

    const float* __restrict__ X = p.x;
    const float* __restrict__ Y = p.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = tid & 15, rq0 = tid >> 4;           // k-quad (4 pixels), row-quad within a pass

    int a_c[PA], a_td[PA], a_th[PA], a_tw[PA]; bool a_ok[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        int m = m0 + 4 * (rq0 + 16 * j);
        a_ok[j] = m < M;
        int mm = a_ok[j] ? m : 0;
        a_c[j] = mm % p.Cx; int tap = mm / p.Cx;
        a_tw[j] = tap % p.kw; tap /= p.kw;
        a_th[j] = tap % p.kh; a_td[j] = tap / p.kh;
    }
    int b_n[PB]; bool b_ok[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) { b_n[j] = n0 + 4 * (rq0 + 16 * j); b_ok[j] = b_n[j] < Nn; }

    float4 ra[PA][4], rb[PB][4];

    auto fetch = [&](int kt) {
        if (ABL(1)) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pix = kt * BKT + kq * 4 + i;
            const bool pok = pix < Ktot;
            unsigned up = (unsigned)(pok ? pix : 0);
            int n = (int)fastdiv(up, p.magDHW); unsigned r = up - (unsigned)n * DHWo;
            int od = (int)fastdiv(r, p.magHW); r -= (unsigned)od * HWo;
            int oy = (int)fastdiv(r, p.magW); int ox = (int)(r - (unsigned)oy * p.Wo);
            const long long nb = (long long)n * p.x_sn;
#pragma unroll
            for (int j = 0; j < PA; ++j) {
                int zd = od * p.sd - p.pd + a_td[j], zh = oy * p.sh - p.ph + a_th[j], zw = ox * p.sw - p.pw + a_tw[j];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pok && a_ok[j] && (unsigned)zd < (unsigned)p.D && (unsigned)zh < (unsigned)p.H && (unsigned)zw < (unsigned)p.W)
                    v = ldg4(X + nb + zd * p.x_sd + zh * p.x_sh + zw * p.x_sw + a_c[j]);
                ra[j][i] = v;
            }
            const float* q = Y + (long long)n * p.y_sn + od * p.y_sd + oy * p.y_sh + ox * p.y_sw;
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pok && b_ok[j]) v = ldg4(q + b_n[j]);
                rb[j][i] = v;
            }
        }
    };
    auto stage = [&](int buf) {
        if (ABL(64)) return;
        char* a = As + buf * BM * ROWB;
        char* b = Bs + buf * BN * ROWB;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            char* base = a + (4 * (rq0 + 16 * j)) * ROWB + kq * 8;
            bf16x4 v0 = {(__bf16)ra[j][0].x, (__bf16)ra[j][1].x, (__bf16)ra[j][2].x, (__bf16)ra[j][3].x};
            bf16x4 v1 = {(__bf16)ra[j][0].y, (__bf16)ra[j][1].y, (__bf16)ra[j][2].y, (__bf16)ra[j][3].y};
            bf16x4 v2 = {(__bf16)ra[j][0].z, (__bf16)ra[j][1].z, (__bf16)ra[j][2].z, (__bf16)ra[j][3].z};
            bf16x4 v3 = {(__bf16)ra[j][0].w, (__bf16)ra[j][1].w, (__bf16)ra[j][2].w, (__bf16)ra[j][3].w};
            *reinterpret_cast<bf16x4*>(base) = v0;
            *reinterpret_cast<bf16x4*>(base + ROWB) = v1;
            *reinterpret_cast<bf16x4*>(base + 2 * ROWB) = v2;
            *reinterpret_cast<bf16x4*>(base + 3 * ROWB) = v3;
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            char* base = b + (4 * (rq0 + 16 * j)) * ROWB + kq * 8;
            bf16x4 v0 = {(__bf16)rb[j][0].x, (__bf16)rb[j][1].x, (__bf16)rb[j][2].x, (__bf16)rb[j][3].x};
            bf16x4 v1 = {(__bf16)rb[j][0].y, (__bf16)rb[j][1].y, (__bf16)rb[j][2].y, (__bf16)rb[j][3].y};
            bf16x4 v2 = {(__bf16)rb[j][0].z, (__bf16)rb[j][1].z, (__bf16)rb[j][2].z, (__bf16)rb[j][3].z};
            bf16x4 v3 = {(__bf16)rb[j][0].w, (__bf16)rb[j][1].w, (__bf16)rb[j][2].w, (__bf16)rb[j][3].w};
            *reinterpret_cast<bf16x4*>(base) = v0;
            *reinterpret_cast<bf16x4*>(base + ROWB) = v1;
            *reinterpret_cast<bf16x4*>(base + 2 * ROWB) = v2;
            *reinterpret_cast<bf16x4*>(base + 3 * ROWB) = v3;
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave >> 1) * 32 * WM, wn0 = (wave & 1) * 32 * WN;
    const int l31 = lane & 31, khalf = lane >> 5;

    fetch(kt_begin);
    stage(0);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        if (kt + 1 < kt_end) fetch(kt + 1);
        const char* a = As + cur * BM * ROWB;
        const char* b = Bs + cur * BN * ROWB;
#pragma unroll
        for (int ks = 0; ks < (ABL(2) ? 0 : BKT / 16); ++ks) {
            bf16x8 af[WM], bf[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(a + (wm0 + i * 32 + l31) * ROWB + (ks * 16 + khalf * 8) * 2);
#pragma unroll
            for (int j = 0; j < WN; ++j)
                bf[j] = *reinterpret_cast<const bf16x8*>(b + (wn0 + j * 32 + l31) * ROWB + (ks * 16 + khalf * 8) * 2);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < kt_end) stage(cur ^ 1);
        __syncthreads();
    }

    // deterministic weight gradient (p.part): this split's tile goes to its own slice of the caller's scratch with plain stores and
    // wgrad_fold adds the slices to dW in split order; without scratch (a C-ABI caller that passed none) the splits meet in dW atomically
    const bool sliced = p.part != nullptr;
    float* __restrict__ dW = sliced ? p.part + (long long)zsplit * p.part_sz : p.out;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int col = n0 + wn0 + j * 32 + l31;
        if (col >= Nn) continue;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                if (row < M) {
                    if (sliced) dW[(long long)row * Nn + col] = acc[i][j][r];
                    else unsafeAtomicAdd(dW + (long long)row * Nn + col, acc[i][j][r]);
                }
            }
    }
}

template <int WM, int WN>
static hipError_t launch_wg_bf16(const ConvP& p, dim3 grid, hipStream_t st) {
    size_t lds = (size_t)2 * (64 * WM + 64 * WN) * (64 + 8) * 2;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)conv_wgrad_bf16_kernel<WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_wgrad_bf16_kernel<WM, WN>), grid, dim3(NTHREADS), lds, st, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// host launcher
// ------------------------------------------------------------------------------------------------------------

template <int WM, int WN, bool VEC, bool BF16, bool WB16 = false>
static hipError_t launch_fd1(const ConvP& p, dim3 grid, hipStream_t st) {
    constexpr int BKT = BF16 ? 64 : 32;
    constexpr size_t rowb = BF16 ? (BKT + 8) * 2 : BKP * 4;
    size_t lds = (size_t)2 * (64 * WM + 64 * WN) * rowb;
    if (lds < (size_t)64 * WM * sizeof(long long)) lds = (size_t)64 * WM * sizeof(long long);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)conv_fd_kernel<WM, WN, VEC, BF16, WB16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_fd_kernel<WM, WN, VEC, BF16, WB16>), grid, dim3(NTHREADS), lds, st, p);
    return hipGetLastError();
}

template <int WM, int WN>
static hipError_t launch_fd(const ConvP& p, bool vec, dim3 grid, hipStream_t st) {
    if (p.bf16 && vec && p.w16) return launch_fd1<WM, WN, true, true, true>(p, grid, st);
    if (p.bf16) return vec ? launch_fd1<WM, WN, true, true>(p, grid, st) : launch_fd1<WM, WN, false, true>(p, grid, st);
    return vec ? launch_fd1<WM, WN, true, false>(p, grid, st) : launch_fd1<WM, WN, false, false>(p, grid, st);
}

template <int WM, int WN>
static hipError_t launch_wg(const ConvP& p, bool va, bool vb, dim3 grid, hipStream_t st) {
    size_t lds = (size_t)2 * BK * ((64 * WM + 4) + (64 * WN + 4)) * sizeof(float);
#define WG_CASE(A, B)                                                                                              \
    {                                                                                                              \
        hipFuncSetAttribute((const void*)conv_wgrad_kernel<WM, WN, A, B>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds);                                                                             \
        hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, A, B>), grid, dim3(NTHREADS), lds, st, p);                   \
    }
    if (va && vb) WG_CASE(true, true)
    else if (va) WG_CASE(true, false)
    else if (vb) WG_CASE(false, true)
    else WG_CASE(false, false)
#undef WG_CASE
    return hipGetLastError();
}


static void pick_tile(long long M, long long N, int& wm, int& wn) {
    // cost model: rounds over the 256 CUs x tile area x a re-read penalty for small tiles
    const int opts[4][2] = {{2, 2}, {2, 1}, {1, 2}, {1, 1}};
    double best = 1e300;
    for (int i = 0; i < 4; ++i) {
        long long bm = 64 * opts[i][0], bn = 64 * opts[i][1];
        long long tm = (M + bm - 1) / bm, tn = (N + bn - 1) / bn;
        double wgs = (double)tm * tn;
        double rounds = wgs <= 256.0 ? 1.0 : wgs / 256.0;
        double eff = (bm * bn == 128 * 128) ? 1.0 : (bm * bn == 64 * 64 ? 1.3 : 1.12);
        double cost = rounds * (double)(bm * bn) * eff;
        if (cost < best) { best = cost; wm = opts[i][0]; wn = opts[i][1]; }
    }
}

// auto algorithm choice between the two LDS-patch kernels when the caller gives no tile: SAVP_CONV_RING=1 prefers conv_ring.hip
static bool ring_default() {
    return savp_opt(OPT_CONV_RING) != 0;
}

// Generic (im2col GEMM) weight gradient: tile shape and K splits -- ONE planner for the launcher and savp_conv_workspace_bytes.
// `splitk` never leaves a split without K tiles (every split's slice of the scratch is then written in full).
struct WgPlan { int wm, wn, splitk; long long M, ktiles, part_sz; bool va, vb, wbf16; };
static bool wgrad_generic_plan(const SavpConvArgs* a, bool bf16, int wm, int wn, WgPlan& g) {
    const bool xs4 = (a->x_sn % 4 == 0) && (a->x_sd % 4 == 0) && (a->x_sh % 4 == 0) && (a->x_sw % 4 == 0) && aligned16(a->x);
    const bool ys4 = (a->y_sn % 4 == 0) && (a->y_sd % 4 == 0) && (a->y_sh % 4 == 0) && (a->y_sw % 4 == 0) && aligned16(a->y);
    g.M = (long long)a->kd * a->kh * a->kw * a->Cx;
    const long long Ktot = (long long)a->N * a->Do * a->Ho * a->Wo;
    // fastdiv exactness domain: p * d < 2^40 for every (pixel index p, divisor d)
    if (Ktot <= 0 || (double)Ktot * (double)((long long)a->Do * a->Ho * a->Wo) >= 1099511627776.0) return false;
    g.va = (a->Cx % 4 == 0) && xs4;
    g.vb = (a->Cy % 4 == 0) && ys4;
    if (!wm) {
        wm = (g.M > 64) ? 2 : 1;
        wn = (a->Cy > 64) ? 2 : 1;
    }
    g.wm = wm; g.wn = wn;
    const int BM = 64 * wm, BN = 64 * wn;
    const long long tiles = ((g.M + BM - 1) / BM) * ((a->Cy + BN - 1) / BN);
    g.wbf16 = bf16 && g.va && g.vb;
    const int bkt = g.wbf16 ? 64 : BK;
    g.ktiles = (Ktot + bkt - 1) / bkt;
    long long splitk = a->splitk;
    if (splitk <= 0) {
        long long want = (512 + tiles - 1) / tiles;               // ~2 workgroups per CU
        long long maxs = g.ktiles / 8 > 0 ? g.ktiles / 8 : 1;     // at least 8 K-tiles per split
        splitk = want < maxs ? want : maxs;
    }
    if (splitk > g.ktiles) splitk = g.ktiles;
    if (splitk < 1) splitk = 1;
    const long long per = (g.ktiles + splitk - 1) / splitk;
    g.splitk = (int)((g.ktiles + per - 1) / per);                 // no empty split
    g.part_sz = g.M * a->Cy;
    return true;
}

extern "C" int64_t savp_conv_workspace_bytes(const SavpConvArgs* a) {
    if (!a) return 0;
    if (a->mode == SAVP_CONV_FPROP || a->mode == SAVP_CONV_DGRAD) {
        // deterministic split-K (conv_common.h): one slice of the dense destination block per split.  An explicit split count is taken as
        // it is; for the automatic one (0) the bound of every kernel's heuristic: min(16, 512 / tiles) with the largest tile (256 x 128)
        if (a->act != SAVP_ACT_NONE || a->out_bf16 || a->stats || a->splitk == 1) return 0;
        const bool dg = a->mode == SAVP_CONV_DGRAD;
        const long long dD = dg ? a->D : a->Do, dH = dg ? a->H : a->Ho, dW_ = dg ? a->W : a->Wo;
        const long long Cd = (dg ? a->Cx : a->Cy) + (a->dst_gap > 0 ? a->dst_gap : 0);
        const long long block = (long long)a->N * dD * dH * dW_ * Cd;
        long long S = a->splitk;
        if (S <= 0) {
            const long long tiles = ((block / Cd + 255) / 256) * ((Cd + 127) / 128);
            if (tiles > 192) return 0;
            S = 512 / tiles;
            if (S > 16) S = 16;
        }
        return S > 1 ? S * block * (long long)sizeof(float) : 0;
    }
    if (a->mode != SAVP_CONV_WGRAD) return 0;
    // weight gradient: the per-split slices of the deterministic accumulation (the planner of whichever kernel savp_conv will take), the
    // RGB-side kernel's partial sums, the separate bias-gradient pass
    const int algo = (a->tile >> 8) & 3;
    long long need = 0;
    const long long thin = algo == 0 ? conv_thin_workspace_bytes(a) : 0;     // geometry only: the RGB-side kernel takes the call when it gets this much scratch
    if (thin) need = thin;
    else {
        int wm = 0, wn = 0;
        if (a->tile & 0xff) { wm = (a->tile >> 4) & 15; wn = a->tile & 15; if (wm < 1 || wm > 2 || wn < 1 || wn > 2) return 0; }
        ConvP p;
        p.bf16 = (a->precision == SAVP_PREC_BF16) ? 1 : 0;
        int rc = SAVP_OK;
        long long patch_bytes = 0;
        WgPlan g;
        if (algo != 1 && conv_wgrad_patch_try(p, a, nullptr, &rc, &patch_bytes)) need = patch_bytes;
        else if (algo != 2 && wgrad_generic_plan(a, p.bf16 != 0, wm, wn, g) && g.splitk > 1) need = (long long)g.splitk * g.part_sz * (long long)sizeof(float);
        const long long bias = a->bias ? (long long)SAVP_COLSUM_WS_FLOATS * 4 : 0;      // the separate bias-gradient pass (savp_colsum) of the generic path
        if (bias > need) need = bias;
    }
    return need;
}

extern "C" int savp_conv_special(const SavpConvArgs* a) {
    if (a && conv_gate_applies(a)) return 1;                     // whatever `tile` says (SavpConvArgs.w_frag)
    if (!a || ((a->tile >> 8) & 3) != 0) return 0;
    return (conv_thin_applies(a) || conv_s2dgrad_applies(a) || conv_gate_applies(a)) ? 1 : 0;
}

extern "C" int savp_conv_stats_ok(const SavpConvArgs* a) {
    if (!a || (a->mode != SAVP_CONV_FPROP && a->mode != SAVP_CONV_DGRAD) || a->precision != SAVP_PREC_BF16) return 0;
    if (((a->tile >> 8) & 3) == 0 && (conv_thin_applies(a) || conv_s2dgrad_applies(a))) return 0;     // those kernels take no statistics
    const bool dg = a->mode == SAVP_CONV_DGRAD;
    const int Cred = dg ? a->Cy : a->Cx;
    if (!(a->w_bf16 && (Cred % 8 == 0) && aligned16(a->w_bf16))) return 0;
    ConvP p;
    p.bf16 = 1; p.w16 = (const unsigned short*)a->w_bf16; p.src16 = a->src_bf16 ? 1 : 0; p.splitk = 1; p.tm = p.tn = 1;
    p.gap_at = a->dst_gap ? a->dst_gap_at : 0x7fffffff; p.gap = a->dst_gap;
    p.nb_ws = (double*)a->nb_ws; p.nb_c0 = a->nb_c0; p.nb_nc = a->nb_nc; p.part = nullptr; p.part_sz = 0;
    if (a->dst_gap && !a->nb_ws) return 0;                      // forward statistics of a gapped destination: not offered
    SavpConvArgs b = *a;
    if (!b.stats && !b.nb_ws) b.stats = (double*)(uintptr_t)16;   // any non-NULL value: only the plan is made
    int wm = 0, wn = 0;
    if (b.tile & 0xff) { wm = (b.tile >> 4) & 15; wn = b.tile & 15; if (wm < 1 || wm > 2 || wn < 1 || wn > 2) return 0; }
    int rc = SAVP_OK;
    return conv_ring_try(p, &b, wm, wn, nullptr, &rc, true) ? 1 : 0;
}

extern "C" int savp_conv(void* stream, const SavpConvArgs* a) {
    if (!a || !a->x || !a->y || !a->w) return SAVP_EINVAL;
    if (a->sd < 1 || a->sh < 1 || a->sw < 1 || a->kd < 1 || a->kh < 1 || a->kw < 1) return SAVP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    ConvP p;
    p.mode = a->mode;
    p.N = a->N; p.D = a->D; p.H = a->H; p.W = a->W; p.Cx = a->Cx;
    p.Do = a->Do; p.Ho = a->Ho; p.Wo = a->Wo; p.Cy = a->Cy;
    p.kd = a->kd; p.kh = a->kh; p.kw = a->kw; p.sd = a->sd; p.sh = a->sh; p.sw = a->sw;
    p.pd = a->pd; p.ph = a->ph; p.pw = a->pw;
    p.beta = a->beta; p.act = a->act; p.alpha = a->alpha;
    p.x = (const float*)a->x; p.x_sn = a->x_sn; p.x_sd = a->x_sd; p.x_sh = a->x_sh; p.x_sw = a->x_sw;
    p.y = (const float*)a->y; p.y_sn = a->y_sn; p.y_sd = a->y_sd; p.y_sh = a->y_sh; p.y_sw = a->y_sw;
    p.w = (const float*)a->w;
    p.w16 = nullptr;
    p.bias = a->bias; p.aux = a->aux;
    p.splitk = 1; p.tm = p.tn = 1; p.part = nullptr; p.part_sz = 0;
    p.src16 = a->src_bf16 ? 1 : 0; p.cell = 0; p.stats = nullptr;
    p.bf16 = (a->precision == SAVP_PREC_BF16) ? 1 : 0;
    const bool gapped = a->dst_gap != 0;
    if (gapped && (a->dst_gap < 0 || a->dst_gap_at < 0 || a->mode == SAVP_CONV_WGRAD)) return SAVP_EINVAL;
    p.gap_at = gapped ? a->dst_gap_at : 0x7fffffff; p.gap = gapped ? a->dst_gap : 0;
    p.nb_ws = (double*)a->nb_ws;
    if (a->nb_ws) {
        if (a->mode == SAVP_CONV_WGRAD || !a->nb_x || !a->nb_mean || !a->nb_rstd || !a->nb_gamma || !a->nb_beta || a->nb_c0 < 0 || a->nb_nc < 1 ||
            a->nb_act < 0 || a->nb_act > 2)
            return SAVP_EINVAL;
        p.nb_x = a->nb_x; p.nb_x_sn = a->nb_x_sn; p.nb_x_sp = a->nb_x_sp; p.nb_mean = a->nb_mean; p.nb_rstd = a->nb_rstd;
        p.nb_gamma = a->nb_gamma; p.nb_beta = a->nb_beta; p.nb_c0 = a->nb_c0; p.nb_nc = a->nb_nc; p.nb_act = a->nb_act; p.nb_alpha = a->nb_alpha;
    }
    p.magW = magic40(a->Wo); p.magHW = magic40(a->Ho * a->Wo); p.magDHW = magic40(a->Do * a->Ho * a->Wo);
    int wm = 0, wn = 0;
    const int algo = (a->tile >> 8) & 3;               // 0 = auto, 1 = generic gather kernel, 2 = LDS patch kernel, 3 = LDS-DMA ring kernel
    if (a->tile & 0xff) { wm = (a->tile >> 4) & 15; wn = a->tile & 15; if (wm < 1 || wm > 2 || wn < 1 || wn > 2) return SAVP_EINVAL; }
    hipError_t err;
    ablate_init();
    const bool xs4 = (a->x_sn % 4 == 0) && (a->x_sd % 4 == 0) && (a->x_sh % 4 == 0) && (a->x_sw % 4 == 0) && aligned16(a->x);
    const bool ys4 = (a->y_sn % 4 == 0) && (a->y_sd % 4 == 0) && (a->y_sh % 4 == 0) && (a->y_sw % 4 == 0) && aligned16(a->y);
    // problem-specific kernels, taken only when the caller leaves the algorithm to the library (tile bits 8-9 == 0): a forced
    // algorithm gets exactly that kernel or EINVAL
    if (algo == 0 && !gapped && !a->nb_ws) {   // convolutions between an RGB / grey image and 32 feature channels (conv_thin.hip): the discriminators' first
        // layer (FPROP, WGRAD) and the data gradient of the generator's scratch-image head
        int rc = SAVP_OK;
        if (conv_thin_try(a, st, &rc)) return rc;
    }
    if (algo == 0 && !gapped && !a->nb_ws && a->mode == SAVP_CONV_DGRAD) {     // 4x4 stride-2 data gradient into a 32-channel activation (conv_s2dgrad.hip)
        int rc = SAVP_OK;
        if (conv_s2dgrad_try(a, st, &rc)) return rc;
    }
    if (a->mode == SAVP_CONV_FPROP) {                  // the ConvLSTM gate convolution's own kernel (conv_gate.hip), given its weight pack
        int rc = SAVP_OK;
        if (conv_gate_try(a, st, &rc)) return rc;
    }
    if (a->mode == SAVP_CONV_FPROP || a->mode == SAVP_CONV_DGRAD) {
        const bool dg = a->mode == SAVP_CONV_DGRAD;
        p.out = (float*)(dg ? a->x : a->y);
        const int Cred = dg ? a->Cy : a->Cx;
        const int Nout = dg ? a->Cx : a->Cy;
        const bool vec = (Cred % 4 == 0) && (dg ? ys4 : xs4) && aligned16(a->w);
        if (a->w_bf16 && (Cred % 8 == 0) && aligned16(a->w_bf16)) p.w16 = (const unsigned short*)a->w_bf16;
        long long Mmax;
        int phases = 1;
        if (dg) {
            phases = a->sd * a->sh * a->sw;
            Mmax = (long long)a->N * ((a->D + a->sd - 1) / a->sd) * ((a->H + a->sh - 1) / a->sh) * ((a->W + a->sw - 1) / a->sw);
        } else {
            Mmax = (long long)a->N * a->Do * a->Ho * a->Wo;
        }
        if (Mmax <= 0 || Nout <= 0) return SAVP_EINVAL;
        // ---- LDS patch kernel (conv_patch.hip): 2-D stride-1 convs in bf16 with pre-packed bf16 weights --------------
        const bool needs_ring = a->out_bf16 || a->src_bf16 || a->stats || gapped || a->nb_ws;   // only the ring kernel reads / writes bf16 activations / skips destination channels
        if (algo == 3 || needs_ring || (algo == 0 && ring_default())) {
            int rc = SAVP_OK;
            if (conv_ring_try(p, a, wm, wn, st, &rc)) return rc;
            if (algo == 3 || needs_ring) return SAVP_EINVAL;
        }
        if (algo != 1) {
            int rc = SAVP_OK;
            if (conv_patch_try(p, a, wm, wn, algo == 2, st, &rc)) return rc;
            if (algo == 2) return SAVP_EINVAL;
        }
        if (!wm) pick_tile(Mmax * phases, Nout, wm, wn);
        const int BM = 64 * wm, BN = 64 * wn;
        // split-K (plain epilogue only): fills the chip when M*N is small and K is long (8x8 / 16x16 ConvLSTM layers)
        int splitk = a->splitk;
        const long long tiles = ((Mmax + BM - 1) / BM) * ((Nout + BN - 1) / BN) * phases;
        const long long taps_max = (long long)a->kd * a->kh * a->kw / (dg ? phases : 1);
        const long long nkt = (taps_max * Cred + (p.bf16 ? 63 : 31)) / (p.bf16 ? 64 : 32);
        const bool can_split = (a->act == SAVP_ACT_NONE);
        if (!can_split) splitk = 1;
        else if (splitk <= 0) {
            splitk = 1;
            if (tiles <= 192 && nkt >= 16) {
                long long s1 = 512 / tiles, s2 = nkt / 8;
                splitk = (int)(s1 < s2 ? s1 : s2);
                if (splitk < 1) splitk = 1;
                if (splitk > 16) splitk = 16;
            }
        }
        if (splitk > 1) {
            // the destination must be one dense block: split s stores its share at the same offsets of its slice of the scratch
            const long long dD = dg ? a->D : a->Do, dH = dg ? a->H : a->Ho, dW_ = dg ? a->W : a->Wo;
            const long long s_n = dg ? a->x_sn : a->y_sn, s_d = dg ? a->x_sd : a->y_sd, s_h = dg ? a->x_sh : a->y_sh,
                            s_w = dg ? a->x_sw : a->y_sw;
            const bool dense = (s_w == Nout) && (s_h == dW_ * Nout) && (dD == 1 || s_d == dH * dW_ * Nout) &&
                               (s_n == dD * dH * dW_ * Nout);
            if (!dense) splitk = 1;
            else {
                p.part_sz = (long long)a->N * dD * dH * dW_ * Nout;
                splitk = splitk_fit(a, splitk, p.part_sz);
                p.part = (float*)a->ws;
            }
        }
        p.splitk = splitk;
        p.tm = (int)((Mmax + BM - 1) / BM); p.tn = (int)((Nout + BN - 1) / BN);
        dim3 grid((unsigned)(p.tm * p.tn), 1, (unsigned)(phases * splitk));
        if (wm == 2 && wn == 2) err = launch_fd<2, 2>(p, vec, grid, st);
        else if (wm == 2 && wn == 1) err = launch_fd<2, 1>(p, vec, grid, st);
        else if (wm == 1 && wn == 2) err = launch_fd<1, 2>(p, vec, grid, st);
        else err = launch_fd<1, 1>(p, vec, grid, st);
        if (splitk > 1 && err == hipSuccess) {
            splitk_fold(p.out, p.part, splitk, p.part_sz, a->beta, Nout, 0, 0, st);
            err = hipGetLastError();
        }
    } else if (a->mode == SAVP_CONV_WGRAD) {
        p.out = (float*)a->w;
        if (algo != 1) {                                   // LDS patch WGRAD (conv_wgrad_patch.hip)
            int rc = SAVP_OK;
            if (conv_wgrad_patch_try(p, a, st, &rc)) return rc;
            if (algo == 2) return SAVP_EINVAL;
        }
        if (a->src_bf16 || a->out_bf16) return SAVP_EINVAL;    // bf16 operand tensors: only the LDS-patch kernel reads them
        WgPlan g;
        if (!wgrad_generic_plan(a, p.bf16 != 0, wm, wn, g)) return SAVP_EINVAL;
        wm = g.wm; wn = g.wn;
        const long long M = g.M, ktiles = g.ktiles;
        const bool va = g.va, vb = g.vb, wbf16 = g.wbf16;
        const int BM = 64 * wm, BN = 64 * wn;
        int splitk = g.splitk;
        // deterministic accumulation: one slice of dW per split in the caller's scratch, folded in split order (fewer splits if it is small;
        // no scratch at all: the splits add atomically, in arrival order)
        p.part = nullptr; p.part_sz = g.part_sz;
        if (splitk > 1) {
            const int fit = splitk_fit(a, splitk, p.part_sz);
            if (fit > 1) {
                const long long per = (ktiles + fit - 1) / fit;
                splitk = (int)((ktiles + per - 1) / per);
                p.part = (float*)a->ws;
            }
        }
        p.splitk = splitk;
        p.tm = (int)((M + BM - 1) / BM); p.tn = (int)((a->Cy + BN - 1) / BN);
        dim3 grid((unsigned)(p.tm * p.tn * splitk), 1, 1);
        if (wbf16) {
            if (wm == 2 && wn == 2) err = launch_wg_bf16<2, 2>(p, grid, st);
            else if (wm == 2 && wn == 1) err = launch_wg_bf16<2, 1>(p, grid, st);
            else if (wm == 1 && wn == 2) err = launch_wg_bf16<1, 2>(p, grid, st);
            else err = launch_wg_bf16<1, 1>(p, grid, st);
        } else if (wm == 2 && wn == 2) err = launch_wg<2, 2>(p, va, vb, grid, st);
        else if (wm == 2 && wn == 1) err = launch_wg<2, 1>(p, va, vb, grid, st);
        else if (wm == 1 && wn == 2) err = launch_wg<1, 2>(p, va, vb, grid, st);
        else err = launch_wg<1, 1>(p, va, vb, grid, st);
        if (p.part && err == hipSuccess) {
            wgrad_fold(p.out, p.part, splitk, p.part_sz, st);
            err = hipGetLastError();
        }
        if (a->bias && err == hipSuccess) {                // bias gradient: column sums of y (separate pass on this path)
            const long long px = (long long)a->Do * a->Ho * a->Wo;
            const bool joint = (a->y_sh == a->Wo * a->y_sw) && (a->Do == 1 || a->y_sd == a->Ho * a->y_sh);
            if (!joint) return SAVP_EINVAL;
            SavpView yv; yv.p = (void*)a->y; yv.sn = a->y_sn; yv.sp = a->y_sw;
            int rc2 = savp_colsum(stream, yv, a->N, (int32_t)px, a->Cy, 1.f, (float*)a->bias, 0, (float*)a->ws, a->ws ? a->ws_bytes / 4 : 0);
            if (rc2 != SAVP_OK) return rc2;
        }
    } else {
        return SAVP_EINVAL;
    }
    return err == hipSuccess ? SAVP_OK : SAVP_ELAUNCH;
}
