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
#include "zero_fill.h"
#include <type_traits>


// NW waves per workgroup in an (NW/2) x 2 grid, each wave owns a 32 WM x 32 WN block of the BM x BN tile.
// The BM = 16 NW WM tile rows are NI images x TIH rows x 8 columns of output pixels (TIH = s1_tih, a power of two >= 4 chosen by
// the launcher: 8x8 planes put several images into one tile so that the weight stream is shared by more rows).
// The reduction channels are cut into slabs of CKB = 16 NKS; the LDS patch holds a GROUP of s1_spp slabs (all of them when they
// fit -> one staging pass; fewer for wide layers, e.g. 512-channel DGRADs, whose patch is then re-staged per group).  Inside a
// group the (tap, slab) weight slabs stream through the double-buffered LDS stage.
template <int NW, int WM, int WN, int NKS>
__global__ __launch_bounds__(64 * NW) void conv_patch_kernel(ConvP p) {
    constexpr int NT = 64 * NW;
    constexpr int BM = 16 * NW * WM, BN = 64 * WN, TW = 8;
    constexpr int CKB = 16 * NKS, BROW = CKB + 8;
    constexpr int SLOTS = BN * 2 * NKS;                    // 16-byte slots of one weight slab
    constexpr int Q = (SLOTS + NT - 1) / NT;
    constexpr bool ALLIN = (SLOTS % NT) == 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const bool dgrad = (p.mode == SAVP_CONV_DGRAD);
    // blockIdx.y = output phase of a strided DGRAD (conv2d_transpose): each phase is a dense stride-1 problem over its
    // own taps t0 + j*s, so no MAC is spent on the zeros of the transposed convolution
    const int fh = dgrad ? (int)blockIdx.y / p.sw : 0, fw = dgrad ? (int)blockIdx.y % p.sw : 0;
    const DimGeom gd = make_geom(dgrad, 0, p.D, p.Do, p.kd, 1, p.pd);      // depth: stride 1 only (3-D discriminator convs)
    const DimGeom gh = make_geom(dgrad, fh, p.H, p.Ho, p.kh, p.sh, p.ph);
    const DimGeom gw = make_geom(dgrad, fw, p.W, p.Wo, p.kw, p.sw, p.pw);
    const int Cred = dgrad ? p.Cy : p.Cx;
    const int Nout = dgrad ? p.Cx : p.Cy;
    const int kh = gh.nt, kw = gw.nt;                      // (reduced) taps of this phase
    const int ntaps = kh * kw;
    const int ldb = p.kd * p.kh * p.kw * Cred;
    const int Hm = gh.Mdim, Wm = gw.Mdim, Dm = gd.Mdim;
    const int nimg = p.N * Dm;                             // "images" = (sample, output depth) pairs
    const int tW = p.s1_tw, tH = p.s1_th;                  // tiles per image (of the largest phase)
    const int PW = p.s1_pw, PH = p.s1_ph;                  // patch extent of ONE image (max over phases)
    const int tih = p.s1_tih;                              // tile rows per image
    const int ni = (BM / TW) / tih;                        // images per tile
    const int rpi = tih * TW;                              // tile rows (GEMM rows) per image: a multiple of 32
    const int nch = p.s1_nch, pitch = p.s1_pitch;
    const int spp = p.s1_spp;                              // slabs per patch group
    const int CP = spp * CKB + 8;                          // patch pixel stride (16 B x odd)
    const int pimg = PH * pitch;                           // LDS elements of one image's patch
    __bf16* patch = reinterpret_cast<__bf16*>(smem);       // [ni][PH][pitch]
    __bf16* Bs = patch + ni * pimg + 8;                    // [2][BN][BROW] (8 elements behind the patch = stage_patch's dummy slot)

    if (ABL(16)) return;
    const int split = blockIdx.z;
    const int tlog = xcd_logical(blockIdx.x, p.tm * p.tn);
    const int mt = tlog % p.tm;
    const int n0 = (tlog / p.tm) * BN;
    const int ig = mt / (tH * tW);                         // image group
    const int trem = mt - ig * (tH * tW);
    const int oy0 = (trem / tW) * tih, ox0 = (trem % tW) * TW;
    const int img0 = ig * ni;
    if (oy0 >= Hm || ox0 >= Wm || kh <= 0 || kw <= 0) return;     // smaller phase / phase without taps (uniform)
    // patch origin in source coordinates: smallest tap displacement
    const int org_h = gh.base + oy0 * gh.mstep + (gh.jstep > 0 ? 0 : (kh - 1) * gh.jstep);
    const int org_w = gw.base + ox0 * gw.mstep + (gw.jstep > 0 ? 0 : (kw - 1) * gw.jstep);

    const float* __restrict__ src = dgrad ? p.y : p.x;
    const long long s_sn = dgrad ? p.y_sn : p.x_sn, s_sd = dgrad ? p.y_sd : p.x_sd;
    const int s_sh = (int)(dgrad ? p.y_sh : p.x_sh), s_sw = (int)(dgrad ? p.y_sw : p.x_sw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // ---- split-K range over the (group, tap, slab-in-group) iteration list ---------------------------------------------
    const int it_dep = ntaps * nch;                        // entries of one depth tap
    const int it_all = gd.nt * it_dep;
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
        okL[q] = (nch - 1) * CKB + k8 * 8 < Cred;          // channel padding exists only in the last slab
        goffF[q] = (unsigned)(row * ldb + k8 * 8);
        goffL[q] = (unsigned)(row * ldb + (okL[q] ? k8 * 8 : 0));
        loff[q] = (ALLIN || slot < SLOTS) ? r * BROW + k8 * 8 : -1;
    }
    uint4 rb[Q];
    int f_cc = 0, f_sl = 0, f_jh = 0, f_jw = 0, g_slabs = 1, g_first = 0, g_jd = 0;      // set per group
    auto fetch = [&]() {
        const int f_tap = ((gd.t0 + g_jd * gd.tstep) * p.kh + (gh.t0 + f_jh * gh.tstep)) * p.kw + (gw.t0 + f_jw * gw.tstep);   // full weight tap
        f_cc = g_first + f_sl;
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
        if (++f_sl == g_slabs) {
            f_sl = 0;
            if (++f_jw == kw) { f_jw = 0; ++f_jh; }
        }
    };
    auto stage = [&](auto curc) {
        constexpr int cur = decltype(curc)::value;
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (ALLIN || loff[q] >= 0) *reinterpret_cast<uint4*>(Bs + cur * BN * BROW + loff[q]) = rb[q];
    };

    // ---- input patch of one channel slab (zero outside the image, beyond Cred and for images >= N) ---------------
    auto stage_patch = [&](int cfirst) {                   // channels [cfirst*CKB, (cfirst + spp)*CKB)
        const int c4n = (spp * CKB) >> 2;
        const int per_img = PH * PW * c4n;
        const int total = ni * per_img;
        // batches of U loads per thread, all issued before the first is consumed (one guarded load per loop trip made hipcc wait
        // vmcnt(0) after each: ~10 serial HBM round trips per workgroup, the "patch staging 6 us" of the r01 ablation); trips past
        // the end are redirected to a dummy slot behind the patch so that no load is left unconsumed
        constexpr int U = 8;
        for (int base = tid; base < (ABL(4) ? 0 : total); base += NT * U) {
            float4 v[U];
            int dsto[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = min(base + u * NT, total - 1);
                const int im = (int)fastdiv((unsigned)idx, p.s1_magPI);
                const int rem = idx - im * per_img;
                const int pix = (int)fastdiv((unsigned)rem, p.s1_magC4);
                const int c = (rem - pix * c4n) << 2;
                const int pyy = (int)fastdiv((unsigned)pix, p.s1_magPW);
                const int pxx = pix - pyy * PW;
                const int iy = org_h + pyy, ix = org_w + pxx;
                const int cg = cfirst * CKB + c;
                const int gi = img0 + im;                          // (sample, depth) index
                const int n = (int)fastdiv((unsigned)gi, p.s1_magDm);
                const int dz = gd.base + (gi - n * Dm) * gd.mstep + g_jd * gd.jstep;      // source plane of this depth tap
                const bool ok = (unsigned)iy < (unsigned)gh.srcN && (unsigned)ix < (unsigned)gw.srcN && cg < Cred && gi < nimg &&
                                (unsigned)dz < (unsigned)gd.srcN;
                v[u] = ldg4(src + (ok ? (long long)n * s_sn + (long long)dz * s_sd + iy * s_sh + ix * s_sw + cg : 0ll));
                const int d = im * pimg + pyy * pitch + pxx * CP + c;
                dsto[u] = (base + u * NT < total) ? (ok ? d : (d | (int)0x40000000)) : (ni * pimg) | (int)0x40000000;   // bit 30: zeros
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float4 t = v[u];
                if (dsto[u] & 0x40000000) t = make_float4(0.f, 0.f, 0.f, 0.f);
                bf16x4 o = {(__bf16)t.x, (__bf16)t.y, (__bf16)t.z, (__bf16)t.w};
                *reinterpret_cast<bf16x4*>(patch + (dsto[u] & 0x3fffffff)) = o;
            }
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
    int arow[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = wm0 + i * 32 + l31;
        const int im = row / rpi, rr = row - im * rpi;
        arow[i] = im * pimg + (rr >> 3) * gh.mstep * pitch + (rr & 7) * gw.mstep * CP + khalf * 8;
    }
    const int brow0 = (wn0 + l31) * BROW + khalf * 8;

    int c_jh = 0, c_jw = 0, c_sl = 0;
    auto compute = [&](auto curc) {
        constexpr int cur = decltype(curc)::value;
        const int pu = gh.jstep > 0 ? c_jh * gh.jstep : (kh - 1 - c_jh) * -gh.jstep;
        const int pv = gw.jstep > 0 ? c_jw * gw.jstep : (kw - 1 - c_jw) * -gw.jstep;
        const __bf16* a = patch + (pu * pitch + pv * CP + c_sl * CKB);
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
        if (++c_sl == g_slabs) {
            c_sl = 0;
            if (++c_jw == kw) { c_jw = 0; ++c_jh; }
        }
    };

    // ---- group-outer loop; inside a group the weight slabs of its (tap, slab) entries run through a two-stage software
    // pipeline (LDS holds entry e, registers hold entry e+1 whose global loads were issued one whole iteration earlier), unrolled
    // by two so that the LDS buffer parity is a compile-time constant ------------------------------------------------------
    const int gsz = ntaps * spp;                           // entries of a full slab group
    const int ngs = (nch + spp - 1) / spp;                 // slab groups per depth tap
    for (int gg = (it_begin / it_dep) * ngs + (it_begin % it_dep) / gsz; gg < gd.nt * ngs; ++gg) {
        g_jd = gg / ngs;
        const int g = gg - g_jd * ngs;
        g_first = g * spp;
        g_slabs = min(spp, nch - g_first);
        const int e_lo = g_jd * it_dep + g * gsz;
        if (e_lo >= it_end) break;
        const int t_begin = max(it_begin, e_lo) - e_lo;
        const int t_end = min(it_end, e_lo + ntaps * g_slabs) - e_lo;
        if (t_begin >= t_end) continue;
        const int tap0 = t_begin / g_slabs;
        f_sl = t_begin - tap0 * g_slabs; f_jh = tap0 / kw; f_jw = tap0 - f_jh * kw;
        c_sl = f_sl; c_jh = f_jh; c_jw = f_jw;
        fetch();                                           // first weight slab in flight while the patch is staged
        __syncthreads();                                   // every wave is done with the previous group's patch and weights
        stage_patch(g_first);
        stage(std::integral_constant<int, 0>{});
        if (t_begin + 1 < t_end) fetch();
        __syncthreads();
        int it = t_begin;
        for (; it + 1 < t_end; it += 2) {
            compute(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            if (it + 2 < t_end) fetch();
            __syncthreads();
            compute(std::integral_constant<int, 1>{});
            if (it + 2 < t_end) {
                stage(std::integral_constant<int, 0>{});
                if (it + 3 < t_end) fetch();
            }
            __syncthreads();
        }
        if (it < t_end) compute(std::integral_constant<int, 0>{});
    }

    // ---- epilogue: accumulator (i, j, r) of lane (l31, khalf) is tile row wm0 + 32 i + (r&3) + 8 (r>>2) + 4 khalf; every
    // 32-row block lies inside one image (rows per image are a multiple of 32) --------------------------------------------
    if (ABL(8) && acc[0][0][0] != 123.f) return;
    const long long d_sn = dgrad ? p.x_sn : p.y_sn, d_sd = dgrad ? p.x_sd : p.y_sd;
    const int d_sh = (int)(dgrad ? p.x_sh : p.y_sh), d_sw = (int)(dgrad ? p.x_sw : p.y_sw);
    const int e_sh = d_sh * gh.os, e_sw = d_sw * gw.os;            // destination strides of one M-grid step
    const int col0 = n0 + wn0 + l31;
    const int px0 = ox0 + 4 * khalf;
    const bool plain = (p.splitk == 1) && !p.beta && (p.act == SAVP_ACT_NONE);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int rowb = wm0 + i * 32;
        const int im = rowb / rpi;
        const int py0 = oy0 + ((rowb - im * rpi) >> 3);            // M-grid row of this block's first pixel row
        const int gi = img0 + im;
        if (gi >= nimg) continue;
        const int n = (int)fastdiv((unsigned)gi, p.s1_magDm);
        float* __restrict__ dst = p.out + (long long)n * d_sn + (long long)(gd.ob + (gi - n * Dm) * gd.os) * d_sd +
                                  (long long)(gh.ob + py0 * gh.os) * d_sh +
                                  (long long)(gw.ob + px0 * gw.os) * d_sw + col0;
        const bool full = (py0 + 4 <= Hm) && (ox0 + TW <= Wm);
        if (plain && full) {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (col0 + 32 * j >= Nout) continue;
                const float bias = p.bias ? p.bias[col0 + 32 * j] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(r >> 2) * e_sh + (r & 3) * e_sw + 32 * j] = acc[i][j][r] + bias;
            }
            continue;
        }
        const float* __restrict__ aux = p.aux ? p.aux + (dst - p.out) : nullptr;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if (col0 + 32 * j >= Nout) continue;
            const float bias = (p.bias && split == 0) ? p.bias[col0 + 32 * j] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!full && (py0 + (r >> 2) >= Hm || px0 + (r & 3) >= Wm)) continue;
                const int off = (r >> 2) * e_sh + (r & 3) * e_sw + 32 * j;
                float v = acc[i][j][r] + bias;
                if (p.splitk > 1) {                            // this split's share (conv_common.h: deterministic split-K)
                    (p.part + (long long)split * p.part_sz + (dst - p.out))[off] = v;
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
    const long long ssd = dg ? a->y_sd : a->x_sd;
    if (!(p.bf16 && p.w16 && a->sd == 1 && a->sh <= 4 && a->sw <= 4 && a->kh >= a->sh && a->kw >= a->sw && (Cred % 8 == 0) &&
          src4 && ssd % 4 == 0))
        return false;
    const int Dm = dg ? a->D : a->Do;                           // depth of the output grid (1 for 2-D problems)
    // M-grid of the (largest) output phase; strided DGRAD runs sh*sw phases as blockIdx.y
    const int phases = dg ? a->sh * a->sw : 1;
    const int Hm = dg ? (a->H + a->sh - 1) / a->sh : a->Ho, Wm = dg ? (a->W + a->sw - 1) / a->sw : a->Wo;
    const long long dH = dg ? a->H : a->Ho, dW_ = dg ? a->W : a->Wo;      // full destination extents (all phases)
    const long long d_sn = dg ? a->x_sn : a->y_sn, d_sh = dg ? a->x_sh : a->y_sh, d_sw = dg ? a->x_sw : a->y_sw;
    const long long d_sd = dg ? a->x_sd : a->y_sd;
    // 32-bit in-image offsets (source and destination) and weight offsets
    if (ssh * (a->H + a->kh) >= (1ll << 30) || d_sh * (dH + 16) >= (1ll << 30) || (long long)Nout * a->kh * a->kw * Cred >= (1ll << 31))
        return false;
    if ((long long)Hm * Wm < 16) return false;                   // dense-like problems: the generic kernel
    const int tW = (Wm + 7) / 8;
    if (!wm) {
        wn = Nout > 64 ? 2 : 1;
        const long long t16 = (long long)a->N * ((Hm + 15) / 16) * tW * ((Nout + 64 * wn - 1) / (64 * wn));
        wm = (t16 >= 256 && Hm >= 16) ? 2 : 1;
        nw = 4;
    }
    // channel slabs: nch slabs of NKS*16 channels, minimising (k-steps + per-iteration overhead)
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
    // tile rows: TH = BM/8 rows of 8 pixels, split into NI images x TIH rows (TIH a power of two, 4 <= TIH <= TH): small planes
    // put several images into one tile
    const int TH = 2 * nw * wm;
    int tih = 4;
    while (tih < TH && tih < Hm) tih *= 2;
    const int ni = TH / tih;
    // patch extent per dim: (tile-1)*mstep + (taps-1)*|jstep| + 1 ; FPROP: mstep = stride, jstep = 1 ; DGRAD: mstep = 1,
    // taps = ceil(k / stride) per phase, jstep = -1
    const int PH = dg ? tih + (a->kh + a->sh - 1) / a->sh - 1 : (tih - 1) * a->sh + a->kh;
    const int PW = dg ? 8 + (a->kw + a->sw - 1) / a->sw - 1 : 7 * a->sw + a->kw;
    // slabs per patch group: all of them if the patch fits into LDS beside the weight stage (one staging pass), otherwise as
    // many as fit (every further group costs a barrier + an exposed patch load)
    // (bits 12-13 of `tile` cap the LDS budget at 64 / 96 KB instead: more workgroups per CU can beat fewer staging passes; the
    // autotuner tries all three)
    const int cls = (a->tile >> 12) & 3;
    const size_t budget = cls == 1 ? 64 * 1024 : (cls == 2 ? 96 * 1024 : 160 * 1024);
    int spp = nch, pitch = 0;
    size_t lds = 0;
    for (; spp >= 1; --spp) {
        const int CP = spp * nks * 16 + 8;
        const int x = (8 - (PW * (CP / 8)) % 16 + 16) % 16;      // row pitch = 8 (mod 16) 16-byte slots
        pitch = PW * CP + 8 * x;
        lds = (size_t)ni * PH * pitch * 2 + 16 + (size_t)2 * 64 * wn * (nks * 16 + 8) * 2;
        if (lds <= budget || (spp == 1 && lds <= 160 * 1024)) break;
    }
    if (spp < 1) return false;
    p.s1_ph = PH; p.s1_pw = PW; p.s1_th = (Hm + tih - 1) / tih; p.s1_tw = tW; p.s1_tih = tih;
    p.s1_pitch = pitch; p.s1_nch = nch; p.s1_spp = spp;
    p.s1_magPI = magic40(PH * PW * spp * nks * 4); p.s1_magPW = magic40(PW); p.s1_magC4 = magic40(spp * nks * 4);
    p.s1_magDm = magic40(Dm);
    if ((long long)a->N * Dm >= (1 << 24)) return false;
    if ((long long)ni * PH * PW * spp * nks * 4 >= (1 << 24)) return false;
    const int BN = 64 * wn;
    p.tm = (int)(((long long)a->N * Dm + ni - 1) / ni) * p.s1_th * tW; p.tn = (Nout + BN - 1) / BN;
    const long long tiles = (long long)p.tm * p.tn;
    const long long iters = (long long)(dg ? (a->kh / a->sh) * (a->kw / a->sw) : a->kh * a->kw) * nch * a->kd;
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
    if (splitk > 1) {
        const long long dD = Dm;
        const bool dense = (d_sw == Nout) && (d_sh == dW_ * Nout) && (dD == 1 || d_sd == dH * dW_ * Nout) &&
                           (d_sn == dD * dH * dW_ * Nout);
        if (!dense) splitk = 1;
        else {
            p.part_sz = (long long)a->N * dD * dH * dW_ * Nout;
            splitk = splitk_fit(a, splitk, p.part_sz);
            p.part = (float*)a->ws;
        }
    }
    p.splitk = splitk;
    dim3 grid((unsigned)(p.tm * p.tn), (unsigned)phases, (unsigned)splitk);
    hipError_t err;
    ablate_init();
    err = (nw == 8) ? launch_patch_tile<8>(p, wm, wn, nks, grid, lds, st) : launch_patch_tile<4>(p, wm, wn, nks, grid, lds, st);
    if (splitk > 1 && err == hipSuccess) {
        splitk_fold(p.out, p.part, splitk, p.part_sz, a->beta, Nout, 0, 0, st);
        err = hipGetLastError();
    }
    *rc = (err == hipSuccess) ? SAVP_OK : SAVP_ELAUNCH;
    return true;
}
