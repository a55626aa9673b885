// conv_patch.hip -- LDS-patch convolution kernel for gfx950: stride-1 2-D FPROP / DGRAD on the bf16 MFMA pipe
// (ConvLSTM 5x5 gate convs rnn_ops.py:121, the 3x3 heads ops.py:528, and their data gradients -- the bulk of the
// SAVP step's FLOPs).
//
// The generic implicit-GEMM kernel (conv_igemm.hip) re-gathers every (pixel, tap) operand from global memory, i.e. the
// address/bounds arithmetic of an im2col row per 16 B loaded.  Here a workgroup owns a TH x 8 block of output pixels of
// one image and ALL reduction channels: the (TH+kh-1) x (8+kw-1) input patch is converted to bf16 and parked in LDS
// once, and every tap's A fragment is that patch read at a shifted LDS address.  Only the weight slab of the current
// (tap, channel chunk) streams through a double-buffered LDS stage (pre-packed bf16, 16-byte loads, no conversion).
//
// Measured lessons baked into the structure (MI355X, one or two workgroups per CU => latency is NOT hidden by other
// waves, every instruction of the loop is on the critical path):
//   * all per-thread weight-slab offsets are computed once; an iteration issues Q loads off a scalar base pointer;
//   * the slab width is a template parameter (NKS k-steps of 16): the MFMA block is straight-line code, all ds_reads
//     of an iteration are issued up front and retired with partial lgkmcnt waits;
//   * the epilogue addresses rows arithmetically (no LDS row table, no per-element branches on the fast path).
// LDS layouts: pixel stride CP = Cpad + 8 elements (16 B x odd), patch row pitch = 8 (mod 16) 16-byte slots, weight
// rows 16 B x odd -> every ds_read_b128 lane group of the MFMA fragments is bank-conflict-free.
#include "conv_common.h"
#include <type_traits>


// NW waves per workgroup in an (NW/2) x 2 grid, each wave owns a 32 WM x 32 WN block of the BM x BN tile.
template <int NW, int WM, int WN, int NKS>
__global__ __launch_bounds__(64 * NW) void conv_patch_kernel(ConvP p) {
    constexpr int NT = 64 * NW;
    constexpr int BM = 16 * NW * WM, BN = 64 * WN, TW = 8, TH = BM / TW;
    constexpr int CKB = 16 * NKS, BROW = CKB + 8;
    constexpr int SLOTS = BN * 2 * NKS;                    // 16-byte slots of one weight slab
    constexpr int Q = (SLOTS + NT - 1) / NT;
    constexpr bool ALLIN = (SLOTS % NT) == 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const bool dgrad = (p.mode == SAVP_CONV_DGRAD);
    // blockIdx.y = output phase of a strided DGRAD (conv2d_transpose): each phase is a dense stride-1 problem over its
    // own taps t0 + j*s, so no MAC is spent on the zeros of the transposed convolution
    const int fh = dgrad ? (int)blockIdx.y / p.sw : 0, fw = dgrad ? (int)blockIdx.y % p.sw : 0;
    const DimGeom gh = make_geom(dgrad, fh, p.H, p.Ho, p.kh, p.sh, p.ph);
    const DimGeom gw = make_geom(dgrad, fw, p.W, p.Wo, p.kw, p.sw, p.pw);
    const int Cred = dgrad ? p.Cy : p.Cx;
    const int Nout = dgrad ? p.Cx : p.Cy;
    const int kh = gh.nt, kw = gw.nt;                      // (reduced) taps of this phase
    const int ldb = p.kh * p.kw * Cred;
    const int Hm = gh.Mdim, Wm = gw.Mdim;
    const int tW = p.s1_tw, tH = p.s1_th;                  // tiles per image (of the largest phase)
    const int PW = p.s1_pw;                                // patch columns: (TW-1)*mstep + (nt-1)*|jstep| + 1 (max over phases)
    const int PH = p.s1_ph;
    const int nch = p.s1_nch, CP = p.s1_cp, pitch = p.s1_pitch;
    const int Cpad = nch * CKB;
    __bf16* patch = reinterpret_cast<__bf16*>(smem);
    __bf16* Bs = patch + PH * pitch;                       // [2][BN][BROW]

    if (ABL(16)) return;
    const int split = blockIdx.z;
    const int tlog = xcd_logical(blockIdx.x, p.tm * p.tn);
    const int mt = tlog % p.tm;
    const int n0 = (tlog / p.tm) * BN;
    const int img = mt / (tH * tW);
    const int trem = mt - img * (tH * tW);
    const int oy0 = (trem / tW) * TH, ox0 = (trem % tW) * TW;
    if (oy0 >= Hm || ox0 >= Wm || kh <= 0 || kw <= 0) return;     // smaller phase / phase without taps (uniform)
    // patch origin in source coordinates: smallest tap displacement
    const int org_h = gh.base + oy0 * gh.mstep + (gh.jstep > 0 ? 0 : (kh - 1) * gh.jstep);
    const int org_w = gw.base + ox0 * gw.mstep + (gw.jstep > 0 ? 0 : (kw - 1) * gw.jstep);

    const float* __restrict__ src = (dgrad ? p.y : p.x) + (long long)img * (dgrad ? p.y_sn : p.x_sn);
    const int s_sh = (int)(dgrad ? p.y_sh : p.x_sh), s_sw = (int)(dgrad ? p.y_sw : p.x_sw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- split-K range over the (tap, slab) iteration list ----------------------------------------------------
    const int it_all = kh * kw * nch;
    const int it_per = (it_all + p.splitk - 1) / p.splitk;
    const int it_begin = split * it_per;
    const int it_end = ABL(32) ? it_begin : min(it_all, it_begin + it_per);

    // ---- weight slab fetch: per-thread offsets computed once ---------------------------------------------------
    unsigned goffF[Q], goffL[Q];
    int loff[Q];
    bool okL[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int slot = tid + NT * q;
        const int r = slot / (2 * NKS), k8 = slot % (2 * NKS);
        const int row = min(n0 + r, Nout - 1);             // columns >= Nout are computed on valid data, never stored
        okL[q] = (nch - 1) * CKB + k8 * 8 < Cred;          // channel padding exists only in the last slab of a tap
        goffF[q] = (unsigned)(row * ldb + k8 * 8);
        goffL[q] = (unsigned)(row * ldb + (okL[q] ? k8 * 8 : 0));
        loff[q] = (ALLIN || slot < SLOTS) ? r * BROW + k8 * 8 : -1;
    }
    uint4 rb[Q];
    int f_cc = it_begin % nch;
    int f_jh = (it_begin / nch) / kw, f_jw = (it_begin / nch) % kw;
    auto fetch = [&]() {
        const int f_tap = (gh.t0 + f_jh * gh.tstep) * p.kw + (gw.t0 + f_jw * gw.tstep);    // full weight tap
        const unsigned short* wp = p.w16 + (f_tap * Cred + f_cc * CKB);
        if (ABL(1)) return;
        if (f_cc == nch - 1) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                uint4 v = *reinterpret_cast<const uint4*>(wp + goffL[q]);
                rb[q] = okL[q] ? v : make_uint4(0u, 0u, 0u, 0u);
            }
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) rb[q] = *reinterpret_cast<const uint4*>(wp + goffF[q]);
        }
        if (++f_cc == nch) {
            f_cc = 0;
            if (++f_jw == kw) { f_jw = 0; ++f_jh; }
        }
    };
    auto stage = [&](auto curc) {
        constexpr int cur = decltype(curc)::value;
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (ALLIN || loff[q] >= 0) *reinterpret_cast<uint4*>(Bs + cur * BN * BROW + loff[q]) = rb[q];
    };

    if (it_begin < it_end) fetch();                        // first slab in flight while the patch is staged

    // ---- stage the input patch (all channels; zero outside the image and in the channel padding) ---------------
    {
        const int c4n = Cpad >> 2;
        const int total = PH * PW * c4n;
#pragma unroll 16
        for (int idx = tid; idx < (ABL(4) ? 0 : total); idx += NT) {
            const int pix = (int)fastdiv((unsigned)idx, p.s1_magC4);
            const int c = (idx - pix * c4n) << 2;
            const int pyy = (int)fastdiv((unsigned)pix, p.s1_magPW);
            const int pxx = pix - pyy * PW;
            const int iy = org_h + pyy, ix = org_w + pxx;
            const bool ok = (unsigned)iy < (unsigned)gh.srcN && (unsigned)ix < (unsigned)gw.srcN && c < Cred;
            float4 v = ldg4(src + (ok ? iy * s_sh + ix * s_sw + c : 0));
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
            *reinterpret_cast<bf16x4*>(patch + pyy * pitch + pxx * CP + c) = o;
        }
    }

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave >> 1) * 32 * WM, wn0 = (wave & 1) * 32 * WN;
    const int l31 = lane & 31, khalf = lane >> 5;
    int arow[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = wm0 + i * 32 + l31;
        arow[i] = (row >> 3) * gh.mstep * pitch + (row & 7) * gw.mstep * CP + khalf * 8;
    }
    const int brow0 = (wn0 + l31) * BROW + khalf * 8;

    int c_cc = it_begin % nch, c_tap = it_begin / nch;
    int c_jh = c_tap / kw, c_jw = c_tap - (c_tap / kw) * kw;
    auto compute = [&](auto curc) {
        constexpr int cur = decltype(curc)::value;
        const int pu = gh.jstep > 0 ? c_jh * gh.jstep : (kh - 1 - c_jh) * -gh.jstep;
        const int pv = gw.jstep > 0 ? c_jw * gw.jstep : (kw - 1 - c_jw) * -gw.jstep;
        const __bf16* a = patch + (pu * pitch + pv * CP + c_cc * CKB);
        const __bf16* b = Bs + cur * BN * BROW + brow0;
        bf16x8 af[NKS][WM], bf[NKS][WN];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
            for (int i = 0; i < WM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8*>(a + arow[i] + ks * 16);
#pragma unroll
            for (int j = 0; j < WN; ++j) bf[ks][j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * BROW + ks * 16);
        }
#pragma unroll
        for (int ks = 0; ks < (ABL(2) ? 0 : NKS); ++ks)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
        if (++c_cc == nch) {
            c_cc = 0;
            if (++c_jw == kw) { c_jw = 0; ++c_jh; }
        }
    };

    if (it_begin < it_end) {
        stage(std::integral_constant<int, 0>{});
        if (it_begin + 1 < it_end) fetch();
    }
    __syncthreads();

    // software pipeline, unrolled by two so that the LDS buffer parity is a compile-time constant: LDS holds slab `it`,
    // the registers hold slab it+1 (its global loads were issued one whole iteration earlier).  Pairs first, odd tail
    // after the loop (a break inside the pair makes the compiler shuffle all accumulators between two AGPR sets).
    int it = it_begin;
    for (; it + 1 < it_end; it += 2) {
        compute(std::integral_constant<int, 0>{});
        stage(std::integral_constant<int, 1>{});
        if (it + 2 < it_end) fetch();
        __syncthreads();
        compute(std::integral_constant<int, 1>{});
        if (it + 2 < it_end) {
            stage(std::integral_constant<int, 0>{});
            if (it + 3 < it_end) fetch();
        }
        __syncthreads();
    }
    if (it < it_end) compute(std::integral_constant<int, 0>{});

    // ---- epilogue: accumulator (i, j, r) of lane (l31, khalf) is pixel row wm0 + 32 i + (r&3) + 8 (r>>2) + 4 khalf of
    // the tile, i.e. tile pixel (py, px) = (wm0/8 + 4 i + (r>>2), (r&3) + 4 khalf), column n0 + wn0 + 32 j + l31 ---------
    if (ABL(8) && acc[0][0][0] != 123.f) return;
    const long long d_sn = dgrad ? p.x_sn : p.y_sn;
    const int d_sh = (int)(dgrad ? p.x_sh : p.y_sh), d_sw = (int)(dgrad ? p.x_sw : p.y_sw);
    const int py0 = oy0 + (wm0 >> 3), px0 = ox0 + 4 * khalf;       // M-grid coordinates of this lane's first pixel
    const int col0 = n0 + wn0 + l31;
    float* __restrict__ dst = p.out + (long long)img * d_sn + (long long)(gh.ob + py0 * gh.os) * d_sh +
                              (long long)(gw.ob + px0 * gw.os) * d_sw + col0;
    const int e_sh = d_sh * gh.os, e_sw = d_sw * gw.os;            // destination strides of one M-grid step
    const bool full = (oy0 + TH <= Hm) && (ox0 + TW <= Wm);
    const bool plain = (p.splitk == 1) && !p.beta && (p.act == SAVP_ACT_NONE);
    if (plain && full) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if (col0 + 32 * j >= Nout) continue;
            const float bias = p.bias ? p.bias[col0 + 32 * j] : 0.f;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dst[(4 * i + (r >> 2)) * e_sh + (r & 3) * e_sw + 32 * j] = acc[i][j][r] + bias;
        }
        return;
    }
    const float* __restrict__ aux = p.aux ? p.aux + (dst - p.out) : nullptr;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        if (col0 + 32 * j >= Nout) continue;
        const float bias = (p.bias && split == 0) ? p.bias[col0 + 32 * j] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!full && (py0 + 4 * i + (r >> 2) >= Hm || px0 + (r & 3) >= Wm)) continue;
                const int off = (4 * i + (r >> 2)) * e_sh + (r & 3) * e_sw + 32 * j;
                float v = acc[i][j][r] + bias;
                if (p.splitk > 1) {
                    unsafeAtomicAdd(dst + off, v);
                    continue;
                }
                if (p.beta) v += dst[off];
                if (p.act == SAVP_ACT_LRELU) v = fmaxf(v, p.alpha * v);
                else if (p.act == SAVP_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                else if (p.act == SAVP_ACT_DLRELU_FROM_OUT) v *= (aux[off] > 0.f ? 1.f : p.alpha);
                dst[off] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
template <int NW, int WM, int WN, int NKS>
static hipError_t launch_patch(const ConvP& p, dim3 grid, size_t lds, hipStream_t st) {
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipFuncSetAttribute((const void*)conv_patch_kernel<NW, WM, WN, NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL((conv_patch_kernel<NW, WM, WN, NKS>), grid, dim3(64 * NW), lds, st, p);
    return hipGetLastError();
}

template <int NW, int WM, int WN>
static hipError_t launch_patch_nks(const ConvP& p, int nks, dim3 grid, size_t lds, hipStream_t st) {
    switch (nks) {
        case 1: return launch_patch<NW, WM, WN, 1>(p, grid, lds, st);
        case 2: return launch_patch<NW, WM, WN, 2>(p, grid, lds, st);
        case 3: return launch_patch<NW, WM, WN, 3>(p, grid, lds, st);
        case 4: return launch_patch<NW, WM, WN, 4>(p, grid, lds, st);
        case 5: return launch_patch<NW, WM, WN, 5>(p, grid, lds, st);
        default: return launch_patch<NW, WM, WN, 6>(p, grid, lds, st);
    }
}

template <int NW>
static hipError_t launch_patch_tile(const ConvP& p, int wm, int wn, int nks, dim3 grid, size_t lds, hipStream_t st) {
    if (wm == 2 && wn == 2) return launch_patch_nks<NW, 2, 2>(p, nks, grid, lds, st);
    if (wm == 2 && wn == 1) return launch_patch_nks<NW, 2, 1>(p, nks, grid, lds, st);
    if (wm == 1 && wn == 2) return launch_patch_nks<NW, 1, 2>(p, nks, grid, lds, st);
    return launch_patch_nks<NW, 1, 1>(p, nks, grid, lds, st);
}

bool conv_patch_try(ConvP& p, const SavpConvArgs* a, int wm, int wn, bool forced, hipStream_t st, int* rc) {
    (void)forced;
    int nw = (a->tile & 0x400) ? 8 : 4;
    const bool dg = a->mode == SAVP_CONV_DGRAD;
    const int Cred = dg ? a->Cy : a->Cx, Nout = dg ? a->Cx : a->Cy;
    const long long ssn = dg ? a->y_sn : a->x_sn, ssh = dg ? a->y_sh : a->x_sh, ssw = dg ? a->y_sw : a->x_sw;
    const void* sptr = dg ? a->y : a->x;
    const bool src4 = (ssn % 4 == 0) && (ssh % 4 == 0) && (ssw % 4 == 0) && aligned16(sptr);
    if (!(p.bf16 && p.w16 && a->D == 1 && a->Do == 1 && a->kd == 1 && a->sd == 1 && a->sh <= 4 && a->sw <= 4 &&
          a->kh >= a->sh && a->kw >= a->sw && (Cred % 8 == 0) && src4))
        return false;
    // M-grid of the (largest) output phase; strided DGRAD runs sh*sw phases as blockIdx.y
    const int phases = dg ? a->sh * a->sw : 1;
    const int Hm = dg ? (a->H + a->sh - 1) / a->sh : a->Ho, Wm = dg ? (a->W + a->sw - 1) / a->sw : a->Wo;
    const long long dH = Hm, dW_ = Wm;
    const long long d_sn = dg ? a->x_sn : a->y_sn, d_sh = dg ? a->x_sh : a->y_sh, d_sw = dg ? a->x_sw : a->y_sw;
    // 32-bit in-image offsets (source and destination) and weight offsets
    if (ssh * (a->H + a->kh) >= (1ll << 30) || d_sh * (dH + 16) >= (1ll << 30) || (long long)Nout * a->kh * a->kw * Cred >= (1ll << 31))
        return false;
    const int tW = (Wm + 7) / 8;
    if (!wm) {
        wn = Nout > 64 ? 2 : 1;
        const long long t16 = (long long)a->N * ((Hm + 15) / 16) * tW * ((Nout + 64 * wn - 1) / (64 * wn));
        wm = (t16 >= 256 && Hm >= 16) ? 2 : 1;
        nw = 4;
    }
    // channel chunking: nch slabs of NKS*16 channels per tap, minimising (k-steps + per-iteration overhead)
    const int Cp16 = (Cred + 15) & ~15;
    int nch = 0, nks = 0;
    double best = 1e30;
    for (int c = (Cp16 + 95) / 96; c <= (Cp16 + 95) / 96 + 2; ++c) {
        const int k = (Cp16 / 16 + c - 1) / c;
        if (k < 1 || k > 6) continue;
        const double cost = c * (k + 1.5);
        if (cost < best) { best = cost; nch = c; nks = k; }
    }
    if (!nch) return false;
    const int Cpad = nch * nks * 16;
    // patch extent per dim: (tile-1)*mstep + (taps-1)*|jstep| + 1 ; FPROP: mstep = stride, jstep = 1 ; DGRAD: mstep = 1,
    // taps = ceil(k / stride) per phase, jstep = -1
    const int TH = 2 * nw * wm;
    const int PH = dg ? TH + (a->kh + a->sh - 1) / a->sh - 1 : (TH - 1) * a->sh + a->kh;
    const int PW = dg ? 8 + (a->kw + a->sw - 1) / a->sw - 1 : 7 * a->sw + a->kw;
    p.s1_ph = PH; p.s1_pw = PW; p.s1_th = (Hm + TH - 1) / TH; p.s1_tw = tW;
    const int CP = Cpad + 8;
    const int x = (8 - (PW * (CP / 8)) % 16 + 16) % 16;          // row pitch = 8 (mod 16) 16-byte slots
    const int pitch = PW * CP + 8 * x;
    const size_t lds = (size_t)PH * pitch * 2 + (size_t)2 * 64 * wn * (nks * 16 + 8) * 2;
    if (lds > 160 * 1024) return false;
    p.s1_cp = CP; p.s1_pitch = pitch; p.s1_nch = nch;
    p.s1_magC4 = magic40(Cpad / 4); p.s1_magPW = magic40(PW);
    const int BN = 64 * wn;
    p.tm = a->N * ((Hm + TH - 1) / TH) * tW; p.tn = (Nout + BN - 1) / BN;
    const long long tiles = (long long)p.tm * p.tn;
    const long long iters = (long long)(dg ? (a->kh / a->sh) * (a->kw / a->sw) : a->kh * a->kw) * nch;
    int splitk = a->splitk;
    if (a->act != SAVP_ACT_NONE) splitk = 1;
    else if (splitk <= 0) {
        splitk = 1;
        if (tiles <= 192 && iters >= 16) {
            long long s1 = 512 / tiles, s2 = iters / 8;
            splitk = (int)(s1 < s2 ? s1 : s2);
            if (splitk < 1) splitk = 1;
            if (splitk > 16) splitk = 16;
        }
    }
    if (splitk > iters) splitk = (int)iters;
    if (splitk > 1 && !a->beta) {
        const bool dense = (d_sw == Nout) && (d_sh == dW_ * Nout) && (d_sn == dH * dW_ * Nout);
        if (dense) hipMemsetAsync(p.out, 0, (size_t)a->N * dH * dW_ * Nout * sizeof(float), st);
        else splitk = 1;
    }
    p.splitk = splitk;
    dim3 grid((unsigned)(p.tm * p.tn), (unsigned)phases, (unsigned)splitk);
    hipError_t err;
    ablate_init();
    err = (nw == 8) ? launch_patch_tile<8>(p, wm, wn, nks, grid, lds, st) : launch_patch_tile<4>(p, wm, wn, nks, grid, lds, st);
    *rc = (err == hipSuccess) ? SAVP_OK : SAVP_ELAUNCH;
    return true;
}
