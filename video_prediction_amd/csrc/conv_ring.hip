// conv_ring.hip -- LDS-patch convolution with an LDS-DMA weight ring and a fused "cell" epilogue, gfx950, bf16 MFMA.
//
// Same tiling idea as conv_patch.hip (a workgroup owns NI images x TIH rows x 8 columns of output pixels, the input patch of a
// group of channel slabs is parked in LDS once, every tap's A fragment is that patch read at a tap-shifted address) with the two
// parts of that kernel that its own ablation found on the critical path rebuilt:
//
//   * weight stream: the (tap, slab) weight slabs no longer travel global -> VGPR -> ds_write behind a barrier per slab.  Every
//     wave issues `global_load_lds_dwordx4` (LDS-DMA, inline asm so that hipcc neither counts nor drains it) straight into a
//     FOUR-deep LDS ring; slabs i+2 and i+3 are in flight while slab i is multiplied, the loop waits with a counted `s_waitcnt vmcnt(LW)`
//     and a raw `s_barrier` (DMA requests stay in flight across it).  The LDS image keeps the conflict-free row pitch of
//     conv_patch.hip (16 B x odd): the DMA destination is lane-linear, so the pad slot of every row is simply one more 16-byte
//     lane whose source address is a duplicate.  No VGPRs, no ds_write, no exposed L2 latency per slab.
//   * epilogue "cell" mode (the ConvLSTM gate convolution, rnn_ops.py:121,148-149): the accumulators are (a) reduced to the
//     per-(sample, channel) sum / sum of squares the instance norm over the 4F gate pre-activations needs (registers -> LDS ->
//     ONE global atomic per (image, channel) per workgroup; the separate statistics pass over the gate tensor disappears) and
//     (b) rounded to bf16, transposed through LDS and stored as full 16-byte pieces of contiguous pixel rows (the fp32 epilogue
//     of conv_patch.hip stores 16 strided dwords per lane).  The gate tensor makes its HBM round trip in half the bytes.
//
// Applies to stride-1 and strided 2-D / 3-D FPROP and DGRAD problems in SAVP_PREC_BF16 exactly like conv_patch.hip (same ConvP
// geometry fields, filled by conv_ring_try); the plain fp32 epilogue (bias / LeakyReLU / sigmoid / beta / split-K) is kept.
#include "conv_common.h"
#include <hip/hip_ext.h>
#include <type_traits>

// one LDS-DMA instruction: lane l copies 16 bytes from its own global address to LDS byte address lds_dst + 16 l
// (MI355X guide 5.7: M0 is written in the same statement that reads it; hipcc does not count this load)
__device__ __forceinline__ void ring_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ float dpp_xor1(float v) {        // value of lane ^ 1 (quad_perm [1,0,3,2])
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

// Issue order of one pipeline step: the R ds_reads of the NEXT entry's fragments are spread between the M MFMAs of the current
// one (both waves of a SIMD share its matrix pipe and all waves leave the barrier together: with the reads issued as a block every
// wave reads while the matrix pipe idles, then every wave multiplies while the LDS idles).
template <int R, int M, int I>
__device__ __forceinline__ void sched_interleave() {
    if constexpr (I < M) {
        constexpr int r = (R * (I + 1)) / M - (R * I) / M;     // reads in front of MFMA I
        if constexpr (r > 0) __builtin_amdgcn_sched_group_barrier(0x100, r, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        sched_interleave<R, M, I + 1>();
    }
}

// developer builds: -DSAVP_RING_STAMPS = the cycle stamps alone (the shipped instruction stream plus ~12 s_memtime reads of one wave);
// -DSAVP_CONV_ABLATE = stamps + the ablation switches, whose branches change the loop's schedule (that build ran 2x slower)
#if defined(SAVP_CONV_ABLATE) || defined(SAVP_RING_STAMPS)
__device__ unsigned long long g_ring_t[16];       // developer build: s_memtime stamps of workgroup 0, wave 0 (savp_debug_ring_times)
__constant__ int g_ring_blk = 0;                  // which workgroup stamps
__constant__ int g_ring_wv = 0;                   // which wave of it
#define RT(i) do { if (blockIdx.x == g_ring_blk && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == g_ring_wv * 64) g_ring_t[i] = __builtin_readcyclecounter(); } while (0)
__device__ unsigned long long g_ring_w[2][8];     // per-wave stamps of workgroup 0: kernel entry, arrival at the first barrier
#define RTW(k) do { if (blockIdx.x == g_ring_blk && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0) g_ring_w[k][threadIdx.x >> 6] = __builtin_readcyclecounter(); } while (0)
__constant__ int g_ring_kwarm = 1;                // developer A/B of kernarg_warm
#define savp_kwarm() g_ring_kwarm
extern "C" int savp_debug_ring_times(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ring_t), sizeof(g_ring_t)) == hipSuccess ? 0 : -1;
}
extern "C" int savp_debug_ring_wave_times(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ring_w), sizeof(g_ring_w)) == hipSuccess ? 0 : -1;
}
extern "C" int savp_debug_ring_block(int b) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_blk), &b, sizeof(int)) == hipSuccess ? 0 : -1;
}
extern "C" int savp_debug_ring_wave(int w) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_wv), &w, sizeof(int)) == hipSuccess ? 0 : -1;
}
#ifdef SAVP_CONV_ABLATE
extern "C" int savp_debug_ring_ablate(int bits) {          // same bits as SAVP_ABLATE, changeable between launches
    ablate_init();
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &bits, sizeof(int)) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int savp_debug_ring_kwarm(int on) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ring_kwarm), &on, sizeof(int)) == hipSuccess ? 0 : -1;
}
#else
#define RT(i) do {} while (0)
#define RTW(k) do {} while (0)
#define savp_kwarm() true
#endif

// Minimum waves per SIMD the register allocator has to leave room for (__launch_bounds__' second argument).  The 8-wave instantiations whose
// LDS footprint lets two workgroups share a CU get the hint (128 VGPRs): <8, 1, 1, <= 3> and <8, 2, 1, 2> (<8, 1, 1, 4> would spill).  E.g. <8, 2, 1, 2> sits at 128 VGPRs = two resident 8-wave workgroups per CU, and its small-K problems (the 64x64 layers) run two per CU -- two more
// registers (130 -> 136 allocated) halve that: 21 -> 28 us per launch in the step (round 5, profiles/r05_ab_calls.md).
template <int NW, int WM, int WN, int NKS> constexpr int ring_min_waves() {
    return (NW == 8 && ((WM == 2 && WN == 1 && NKS == 2) || (WM == 1 && WN == 1 && NKS <= 3))) ? 4 : 1;
}

template <int NW, int WM, int WN, int NKS>
__global__ __launch_bounds__(64 * NW, (ring_min_waves<NW, WM, WN, NKS>())) void conv_ring_kernel(ConvP p) {
    RT(0);
    RTW(0);
    if (savp_kwarm()) kernarg_warm<sizeof(ConvP)>();
    constexpr int NT = 64 * NW;
    constexpr int BM = 16 * NW * WM, BN = 64 * WN, TW = 8;
    constexpr int CKB = 16 * NKS;
    constexpr int RS = 2 * NKS + 1;                        // 16-byte slots per weight row (last one = pad)
    constexpr int BROW = RS * 8;                           // row pitch in elements
    constexpr int SLOTS = BN * RS;
    constexpr int LW = (SLOTS + NT - 1) / NT;              // DMA instructions per wave and slab
    constexpr int SLABB = LW * NT * 16;                    // bytes of one ring buffer
    constexpr int RING = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const bool dgrad = (p.mode == SAVP_CONV_DGRAD);
    // p.pre: the launcher worked the geometry and the index-arithmetic constants out (ConvP); otherwise (strided DGRAD: the geometry
    // depends on blockIdx.y's output phase) they are derived here with real divisions
    const bool pre = p.pre != 0;
    DimGeom gd, gh, gw;
    if (pre) { gd = p.gD; gh = p.gH; gw = p.gW; }
    else {
        const int fh = dgrad ? (int)blockIdx.y / p.sw : 0, fw = dgrad ? (int)blockIdx.y % p.sw : 0;
        gd = make_geom(dgrad, 0, p.D, p.Do, p.kd, 1, p.pd);
        gh = make_geom(dgrad, fh, p.H, p.Ho, p.kh, p.sh, p.ph);
        gw = make_geom(dgrad, fw, p.W, p.Wo, p.kw, p.sw, p.pw);
    }
    auto divq = [&](int x, int d, unsigned long long mag) -> int { return pre ? (int)fastdiv((unsigned)x, mag) : x / d; };
    const int Cred = dgrad ? p.Cy : p.Cx;
    const int Nout = dgrad ? p.Cx : p.Cy;
    const int kh = gh.nt, kw = gw.nt;
    const int ntaps = kh * kw;
    const int ldb = p.kd * p.kh * p.kw * Cred;
    const int Hm = gh.Mdim, Wm = gw.Mdim, Dm = gd.Mdim;
    const int nimg = p.N * Dm;
    const int tW = p.s1_tw, tH = p.s1_th;
    const int PW = p.s1_pw, PH = p.s1_ph;
    const int tih = p.s1_tih;
    const int rsh = p.s1_tih_sh + 3;                       // log2(rows of one image in the tile): tih is a power of two
    const int ni = (BM / TW) >> p.s1_tih_sh;
    const int nch = p.s1_nch, pitch = p.s1_pitch;
    const int spp = p.s1_spp;
    const int CP = spp * CKB + 8;
    const int pimg = PH * pitch;
    unsigned char* ring = reinterpret_cast<unsigned char*>(smem);                     // [RING][SLABB]
    __bf16* patch = reinterpret_cast<__bf16*>(ring + RING * SLABB);                    // [ni][PH][pitch]
    const unsigned ring_lds = (unsigned)(uintptr_t)ring;                               // LDS byte address of the ring

    RT(7);
    if (ABL(16)) return;
    const int split = blockIdx.z;
    const int tlog = xcd_logical(blockIdx.x, p.tm * p.tn);
    const int nt_ = divq(tlog, p.tm, p.s1_magTm);
    const int mt = tlog - nt_ * p.tm;
    const int n0 = nt_ * BN;
    const int ig = divq(mt, tH * tW, p.s1_magTHW);
    const int trem = mt - ig * (tH * tW);
    const int trow = divq(trem, tW, p.s1_magTW);
    const int oy0 = trow * tih, ox0 = (trem - trow * tW) * TW;
    const int img0 = ig * ni;
    if (oy0 >= Hm || ox0 >= Wm || kh <= 0 || kw <= 0) return;
    const int org_h = gh.base + oy0 * gh.mstep + (gh.jstep > 0 ? 0 : (kh - 1) * gh.jstep);
    const int org_w = gw.base + ox0 * gw.mstep + (gw.jstep > 0 ? 0 : (kw - 1) * gw.jstep);

    const long long s_sn = dgrad ? p.y_sn : p.x_sn, s_sd = dgrad ? p.y_sd : p.x_sd;
    const int s_sh = (int)(dgrad ? p.y_sh : p.x_sh), s_sw = (int)(dgrad ? p.y_sw : p.x_sw);
    const float* __restrict__ src = dgrad ? p.y : p.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int it_dep = ntaps * nch;
    const int it_all = gd.nt * it_dep;
    const int it_per = pre ? p.s1_itper : (it_all + p.splitk - 1) / p.splitk;
    const int it_begin = split * it_per;
    const int it_end = min(it_all, it_begin + it_per);

    // ---- weight slab DMA: per-lane source offsets (bytes) computed once --------------------------------------------------
    unsigned goffF[LW], goffL[LW];
    int g_slabs = 1, g_first = 0, g_jd = 0;
    // Per-entry offsets come from a small LDS table filled once per group (below): the walk over (tap, slab) entries then costs
    // no scalar arithmetic.  (The CU has ONE scalar unit for all its waves: the tap / slab state machine of conv_patch.hip, run
    // by 8 waves, was ~40 % of that kernel's main loop.)
    uint2* etab = reinterpret_cast<uint2*>(ring + RING * SLABB + (size_t)ni * pimg * 2 + 16);      // [entries] {weight byte offset | last-slab flag, patch byte offset}
    const unsigned dma_lds = ring_lds + (unsigned)(wave * LW) * 1024u;
    auto issue = [&](uint2 te, auto bufc) {                    // DMA of one (tap, slab) entry into ring buffer BUF
        constexpr int BUF = decltype(bufc)::value;
        if (ABL(1)) return;
        const unsigned char* wp = reinterpret_cast<const unsigned char*>(p.w16) + (te.x & 0x7fffffffu);
        const bool last = (te.x >> 31) != 0;
#ifdef SAVP_RING_DMA_ABLATE          // developer timing builds only (wrong results): issue at most this many of the LW slab DMAs per wave and entry
        constexpr int NQ = SAVP_RING_DMA_ABLATE < LW ? SAVP_RING_DMA_ABLATE : LW;
#else
        constexpr int NQ = LW;
#endif
#pragma unroll
        for (int q = 0; q < NQ; ++q) ring_dma16(wp + (last ? goffL[q] : goffF[q]), dma_lds + (unsigned)(BUF * SLABB + q * 1024));
    };

    // Per-GROUP arguments (index-arithmetic constants of the entry table and of the patch staging, the staging switches) are read from the
    // kernel-argument segment at the head of every slab group through a pointer laundered there (round 5, as the epilogue's): ~25 scalar
    // registers that otherwise stay live across the main loop.
    typedef const ConvP __attribute__((address_space(4))) ConvPK;
    ConvPK* pgk = (ConvPK*)__builtin_amdgcn_kernarg_segment_ptr();
    // ---- input patch of one slab group (fp32 or bf16 source; zero outside the image, beyond Cred and for images >= N) ----
    auto stage_patch = [&](int cfirst, auto s16c) {
        constexpr bool S16 = decltype(s16c)::value;        // compile-time: a runtime branch inside the loop serialises the loads
        // (values, not references: a load through pgk behind an asm with a memory clobber would be repeated after every such asm)
        const unsigned long long magPI = (*pgk).s1_magPI, magC4 = (*pgk).s1_magC4, magPW = (*pgk).s1_magPW, magDm = (*pgk).s1_magDm;
        const int c4n = (spp * CKB) >> 2;
        const int per_img = PH * PW * c4n;
        const int total = ni * per_img;
        // batches of U loads per thread: every load of a batch is issued before the first one is consumed (a loop with one guarded
        // load per trip makes hipcc wait vmcnt(0) after each -- ten serial HBM round trips per workgroup in conv_patch.hip)
        constexpr int U = 8;
        for (int base = tid; base < (ABL(4) ? 0 : total); base += NT * U) {
            float4 v[U];
            uint2 w[U];
            int dsto[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = min(base + u * NT, total - 1);
                const int im = (int)fastdiv((unsigned)idx, magPI);
                const int rem = idx - im * per_img;
                const int pix = (int)fastdiv((unsigned)rem, magC4);
                const int c = (rem - pix * c4n) << 2;
                const int pyy = (int)fastdiv((unsigned)pix, magPW);
                const int pxx = pix - pyy * PW;
                const int iy = org_h + pyy, ix = org_w + pxx;
                const int cg = cfirst * CKB + c;
                const int gi = img0 + im;
                const int n = (int)fastdiv((unsigned)gi, magDm);
                const int dz = gd.base + (gi - n * Dm) * gd.mstep + g_jd * gd.jstep;
                const bool ok = (unsigned)iy < (unsigned)gh.srcN && (unsigned)ix < (unsigned)gw.srcN && cg < Cred && gi < nimg &&
                                (unsigned)dz < (unsigned)gd.srcN;
                const long long off = ok ? (long long)n * s_sn + (long long)dz * s_sd + iy * s_sh + ix * s_sw + cg : 0ll;
                if constexpr (S16) w[u] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(src) + off);
                else v[u] = ldg4(src + off);
                const int d = im * pimg + pyy * pitch + pxx * CP + c;
                // bit 30: store zeros.  Trips past the end write zeros to a dummy slot behind the patch: EVERY load is consumed
                // unconditionally, so hipcc's scoreboard is empty when the DMA loop starts (a skipped consumer leaves the load
                // "pending" and the compiler then drains vmcnt -- and with it the DMA ring -- inside the main loop)
                dsto[u] = (base + u * NT < total) ? (ok ? d : (d | (int)0x40000000)) : (ni * pimg) | (int)0x40000000;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool zero = (dsto[u] & 0x40000000) != 0;
                bf16x4 o;
                if constexpr (S16) {
                    uint2 t = w[u];
                    if (zero) t = make_uint2(0u, 0u);
                    o = __builtin_bit_cast(bf16x4, t);
                } else {
                    float4 t = v[u];
                    if (zero) t = make_float4(0.f, 0.f, 0.f, 0.f);
                    o = bf16x4{(__bf16)t.x, (__bf16)t.y, (__bf16)t.z, (__bf16)t.w};
                }
                *reinterpret_cast<bf16x4*>(patch + (dsto[u] & 0x3fffffff)) = o;
            }
        }
    };

    // ---- the same patch filled by LDS-DMA (bf16 source, round 3): the patch is a run of 16-byte slots -- [image][patch row][pixel]
    // [8-channel chunk], row pitch and pixel pitch both multiples of 16 bytes -- so a wave instruction of `global_load_lds_dwordx4`
    // fills 64 consecutive slots, each lane from its own source address: the pixel's 8 channels, or 16 zero bytes in global memory
    // for halo pixels, channels beyond Cred, images beyond N and the pad slots.  No VGPR round trip, no conversion, no ds_write (the
    // VGPR path above is instruction-bound: 12-22 k cycles of a 62-105 k cycle gate convolution).  The DMAs are drained (vmcnt 0)
    // before the barrier that publishes the patch, so the weight ring's counted waits never see them.
    auto stage_patch_dma = [&](int cfirst, auto drainc) {
        // kernel arguments of this group as VALUES: read through pgk inside the loops they would be re-loaded (s_load + wait, in a divergent
        // branch for the zero slot) after every DMA instruction, whose asm carries a memory clobber -- measured: ~400 cycles per DMA instruction
        const unsigned long long magDm = (*pgk).s1_magDm, magC8 = (*pgk).s1_magC8, magPI8 = (*pgk).s1_magPI8, magP8 = (*pgk).s1_magP8;
        const unsigned long long zero16 = (unsigned long long)(uintptr_t)(*pgk).zero16;
        if constexpr (NKS >= 3) {
            // Row-wise (round 5; the instantiations with >= 3 k-steps per slab -- the ConvLSTM gate convolutions and the other wide-channel
            // layers): a wave takes whole patch rows (image, patch row = wave-uniform), a row is ceil(P8 / 64) DMA instructions.  Everything
            // that depends on the lane -- which pixel / 8-channel chunk of the row its slot is, whether that slot is data or padding, its
            // offset inside a source row -- is the same for every row and worked out ONCE per group; per row only scalar arithmetic (row
            // validity, the row's base offset) and one add + select per instruction remain.  The slot-linear walk below decomposes every
            // slot with four 64-bit multiply-high divisions per lane: cycle stamps put its staging at 8.1 / 13.6 k cycles of the 45 / 58
            // k-cycle gate convolutions at 16x16 / 8x8 (now 4.7 / 4.7 k: the memory round trip of the burst; 32x32 was there already).
            // In the step (rocprofv3, profiles/r05_ab_calls.md): <8,1,1,3> 24.1 -> 22.0 us, <8,2,1,4> 42.4 -> 36.8, <4,1,1,6> 31.9 ->
            // 28.7, <8,2,1,5> 37.9 -> 35.3, <4,1,1,8> 36.2 -> 34.1.  NOT for NKS <= 2: the per-lane offset array costs ~15 VGPRs, and the
            // small-K instantiations live on two to five resident workgroups per CU (<8,2,1,2> 21 -> 28 us, <4,2,1,1> 38 -> 63 us with it;
            // a register-lean incremental walk instead of the array lost as much on <8,1,1,4>: per-row scalar work behind spilled SGPRs).
            const int C8 = CP >> 3, P8 = pitch >> 3;
            const int used8 = (spp * CKB) >> 3;               // chunks of a pixel that are read (the last one of CP is pitch padding)
            const unsigned patch_lds = ring_lds + (unsigned)(RING * SLABB);
            const unsigned char* src_b = reinterpret_cast<const unsigned char*>(src);
            constexpr int NJMAX = 8;                          // launcher: P8 <= 512 slots per patch row (ring_plan), else slot-linear
            const int nj = (P8 + 63) >> 6;
            int rel[NJMAX];                                   // element offset inside a source row, or -1: a zero slot in every row
#pragma unroll
            for (int j = 0; j < NJMAX; ++j) {
                const int sl = j * 64 + lane;
                const int pxx = (int)fastdiv((unsigned)sl, magC8);
                const int ch8 = sl - pxx * C8;
                const int ix = org_w + pxx;
                const int cg = cfirst * CKB + ch8 * 8;
                const bool ok = sl < P8 && pxx < PW && ch8 < used8 && (unsigned)ix < (unsigned)gw.srcN && cg < Cred;
                rel[j] = ok ? ix * s_sw + cg : -1;
            }
            const int rows = ni * PH;
            int im = 0, pyy = wave;                           // row = im * PH + pyy, waves take rows round-robin
            while (pyy >= PH) { pyy -= PH; ++im; }
            for (int row = wave; row < (ABL(4) ? 0 : rows); row += NW) {
                const int gi = img0 + im;
                const int n = (int)fastdiv((unsigned)gi, magDm);
                const int dz = gd.base + (gi - n * Dm) * gd.mstep + g_jd * gd.jstep;
                const int iy = org_h + pyy;
                const bool row_ok = (unsigned)iy < (unsigned)gh.srcN && gi < nimg && (unsigned)dz < (unsigned)gd.srcN;
                const long long row_base = (long long)n * s_sn + (long long)dz * s_sd + (long long)iy * s_sh;
                const unsigned char* rb = src_b + row_base * 2;
                const unsigned lds_row = patch_lds + (unsigned)(row * P8) * 16u;
                // (Measured and not kept, round 5: zeroing the slots that hold no data with a store and masking their lanes out of the DMA
                // instruction instead of gathering them from the 16 zero bytes -- the patch is complete at the same cycle.  What the staging
                // costs is its instruction stream: ~650 instructions per wave for ~12 DMA instructions, two waves per SIMD.)
#pragma unroll
                for (int j = 0; j < NJMAX; ++j) {
                    if (j < nj && j * 64 + lane < P8) {
                        const unsigned long long g = (row_ok && rel[j] >= 0) ? (unsigned long long)(uintptr_t)(rb + (long long)rel[j] * 2) : zero16;
                        ring_dma16(reinterpret_cast<const void*>((uintptr_t)g), lds_row + (unsigned)(j * 1024));
                    }
                }
                pyy += NW;
                while (pyy >= PH) { pyy -= PH; ++im; }
            }
        } else {
            const int C8 = CP >> 3, P8 = pitch >> 3;
            const int used8 = (spp * CKB) >> 3;                   // chunks of a pixel that are read (the last one of CP is pitch padding)
            const int per_img8 = PH * P8;
            const int total = ni * per_img8;
            const unsigned patch_lds = ring_lds + (unsigned)(RING * SLABB);
            const unsigned char* src_b = reinterpret_cast<const unsigned char*>(src);
            for (int base = wave * 64; base < (ABL(4) ? 0 : total); base += NT) {
                const int slot = base + lane;
                if (slot < total) {
                    const int im = (int)fastdiv((unsigned)slot, magPI8);
                    const int rem = slot - im * per_img8;
                    const int pyy = (int)fastdiv((unsigned)rem, magP8);
                    const int r = rem - pyy * P8;
                    const int pxx = (int)fastdiv((unsigned)r, magC8);
                    const int ch8 = r - pxx * C8;
                    const int iy = org_h + pyy, ix = org_w + pxx;
                    const int cg = cfirst * CKB + ch8 * 8;
                    const int gi = img0 + im;
                    const int n = (int)fastdiv((unsigned)gi, magDm);
                    const int dz = gd.base + (gi - n * Dm) * gd.mstep + g_jd * gd.jstep;
                    const bool ok = pxx < PW && ch8 < used8 && (unsigned)iy < (unsigned)gh.srcN && (unsigned)ix < (unsigned)gw.srcN &&
                                    cg < Cred && gi < nimg && (unsigned)dz < (unsigned)gd.srcN;
                    const long long off = (long long)n * s_sn + (long long)dz * s_sd + iy * s_sh + ix * s_sw + cg;
                    const unsigned long long g = ok ? (unsigned long long)(uintptr_t)(src_b + off * 2) : zero16;
                    ring_dma16(reinterpret_cast<const void*>((uintptr_t)g), patch_lds + (unsigned)(base * 16));
                }
            }

        }
        if constexpr (decltype(drainc)::value) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // (Round 5, measured and NOT kept: the first group's patch requested here, in front of the per-lane weight offsets / accumulator clears /
    // entry table -- ~3 k cycles of VALU issue that could run underneath the patch's memory round trip.  In the step every ring instantiation
    // got 0.5 - 1 us SLOWER, 53.15 -> 53.5 ms per step (profiles/r05_ab_calls.md): the patch burst then coincides with the weight warm-up's.)
    const int gsz = ntaps * spp;
    const int ngs = pre ? p.s1_ngs : (nch + spp - 1) / spp;
    const int gg0 = (split == 0) ? 0 : (it_begin / it_dep) * ngs + (it_begin % it_dep) / gsz;
    // ---- first group's patch requested NOW (option ring_early, round 5): the patch is what the first barrier waits for -- a memory round trip
    // of 5 - 6 k cycles that used to start behind the whole prologue (7 k cycles) -- so its LDS-DMA requests go out before the weight
    // warm-up, the per-lane slab offsets, the accumulator clears and the entry table; they are drained where the patch used to be staged.
    // (Round 5's first attempt issued them BEHIND the warm-up and was slower in the step: the two bursts then queue behind each other.)
    bool early_patch = false;
    if (p.early && p.dma_patch && gg0 < gd.nt * ngs) {
        g_jd = (gd.nt == 1) ? 0 : gg0 / ngs;
        asm volatile("" : "+s"(pgk));
        stage_patch_dma((gg0 - g_jd * ngs) * spp, std::false_type{});
        early_patch = true;
    }
    // ---- L2 warm-up of this column tile's weight block ------------------------------------------------------------------
    // The block (BN rows x ldb bf16, contiguous in the packed layout) is streamed by every workgroup of the tile through a
    // four-deep DMA ring: three slabs (~1.5 k cycles) of look-ahead.  Inside the train step the weights are cold (each layer's
    // are touched once per time step, tens of MB of other traffic in between) and an L2 miss costs more than the look-ahead:
    // the in-step gate convolutions ran 5-15 us above their back-to-back time (tests/tools/insitu_tune.py).  All workgroups of
    // a column tile on one XCD (consecutive logical ids, one private L2) therefore split the block between them and pull their
    // slices in with LDS-DMA instructions issued before anything else -- bandwidth-bound and overlapped with the prologue and
    // the patch staging.  The data lands in this wave's own slots of the (still unused) ring; the real slab DMAs of the same
    // wave are ordered behind it.  (Not with split-K: a split reads a tap range of every row, not a contiguous block.)
    if (p.wwarm && p.splitk == 1) {
        const int nwg = p.tm * p.tn, qx = nwg >> 3, rx = nwg & 7, xcd = (int)blockIdx.x & 7;
        const int first = xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx;
        const int cnt = qx + (xcd < rx ? 1 : 0);
        const int l_lo = max(first, nt_ * p.tm), l_hi = min(first + cnt, (nt_ + 1) * p.tm);
        const int share = max(l_hi - l_lo, 1), mine = min(max(tlog - l_lo, 0), share - 1);
        // (with a column gap the physical rows of the tile's first .. last logical column are warmed, gap rows included)
        const int r_lo = n0 + (n0 >= p.gap_at ? p.gap : 0), c_hi = min(n0 + BN, Nout) - 1, r_hi = c_hi + (c_hi >= p.gap_at ? p.gap : 0);
        const int blk = (r_hi - r_lo + 1) * ldb * 2;                        // bytes of the block (launcher: < 2^31)
        const int chunk = ((blk + share - 1) / share + 1023) & ~1023;       // bytes per workgroup, whole wave instructions
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w16) + (size_t)r_lo * ldb * 2;
        const int lo = mine * chunk;
        int slot = 0;
        for (int off = wave * 1024; off < chunk && lo + off < blk; off += NW * 1024) {
            const int a = min(lo + off + lane * 16, blk - 16);
            ring_dma16(wb + a, ring_lds + (unsigned)((wave * LW + slot % LW) * 1024 + (slot / LW % RING) * SLABB));
            ++slot;
        }
    }

#pragma unroll
    for (int q = 0; q < LW; ++q) {
        const int slot = (wave * LW + q) * 64 + lane;
        const int r = min(slot / RS, BN - 1);                  // slots >= SLOTS land in the buffer's tail, never read
        int j = slot % RS;
        if (j == 2 * NKS) j = 0;                               // pad slot of the row: any valid address
        int row = min(n0 + r, Nout - 1);                       // columns >= Nout are computed on valid data, never stored
        if (row >= p.gap_at) row += p.gap;                     // logical output column -> physical weight row (ConvP::gap)
        const bool okL = (nch - 1) * CKB + j * 8 < Cred;       // beyond Cred the patch holds zeros: any FINITE weights do
        goffF[q] = (unsigned)(row * ldb + j * 8) * 2u;
        goffL[q] = (unsigned)(row * ldb + (okL ? j * 8 : 0)) * 2u;
    }
    RT(8);
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
        const int im = row >> rsh, rr = row - (im << rsh);
        arow[i] = im * pimg + (rr >> 3) * gh.mstep * pitch + (rr & 7) * gw.mstep * CP + khalf * 8;
    }
    const int brow0 = (wn0 + l31) * BROW + khalf * 8;

    // ---- fragments of one (tap, slab) entry; two register sets: the ds_reads of entry e+1 are issued BEFORE the MFMAs of entry e --
    struct Frags { bf16x8 a[NKS][WM]; bf16x8 b[NKS][WN]; };
    const unsigned char* patch_b = reinterpret_cast<const unsigned char*>(patch);
    int arow_b[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) arow_b[i] = arow[i] * 2;
    auto load_a = [&](Frags& f, uint2 te) {
        const unsigned char* a = patch_b + te.y;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int i = 0; i < WM; ++i) f.a[ks][i] = *reinterpret_cast<const bf16x8*>(a + arow_b[i] + ks * 32);
    };
    auto load_b = [&](Frags& f, auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
        const __bf16* b = reinterpret_cast<const __bf16*>(ring + BUF * SLABB) + brow0;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int j = 0; j < WN; ++j) f.b[ks][j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * BROW + ks * 16);
    };
    auto mma = [&](const Frags& f, int k0, int k1) {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks < k0 || ks >= k1) continue;
            if (ABL(2)) {
#pragma unroll
                for (int i = 0; i < WM; ++i) asm volatile("" :: "v"(f.a[ks][i]));
#pragma unroll
                for (int j = 0; j < WN; ++j) asm volatile("" :: "v"(f.b[ks][j]));
                continue;
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ks][i], f.b[ks][j], acc[i][j], 0, 0, 0);
        }
    };

    RT(9);
    // ---- group-outer loop; inside a group the weight slabs stream through the four-deep DMA ring -----------------------------
    // entry e: its slab is DMAed three iterations ahead, its fragments are read one iteration ahead, its MFMAs run in iteration e.
    Frags F0, F1;
    bool first_group = true;
    using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>;
    using B2 = std::integral_constant<int, 2>; using B3 = std::integral_constant<int, 3>;
    for (int gg = gg0; gg < gd.nt * ngs; ++gg) {
        RT(11);
        g_jd = (gd.nt == 1) ? 0 : gg / ngs;
        const int g = gg - g_jd * ngs;
        g_first = g * spp;
        g_slabs = min(spp, nch - g_first);
        const int e_lo = g_jd * it_dep + g * gsz;
        if (e_lo >= it_end) break;
        const int t_begin = max(it_begin, e_lo) - e_lo;
        const int t_end = ABL(32) ? 0 : min(it_end, e_lo + ntaps * g_slabs) - e_lo;
        if (t_begin >= t_end) continue;
        const int len = t_end - t_begin;
        if (gg != gg0) early_patch = false;                   // (the early request was for group gg0 only)
        RT(12);
        RTW(1);
        // previous group's patch, ring and table are dead (no DMA in flight here).  Not for the first group: nothing to protect, and
        // __syncthreads() = s_waitcnt vmcnt(0) + s_barrier would put the whole latency of the weight warm-up DMAs (issued a few
        // hundred cycles ago) in front of the table fill and the patch staging
        if (!first_group) __syncthreads();
        first_group = false;
        RT(10);
        asm volatile("" : "+s"(pgk));
        ConvPK& pg = *pgk;
        auto entry_of = [&](int ent) {                     // (tap, slab) entry -> {weight byte offset | last-slab flag, patch byte offset}
            const int tap = divq(ent, g_slabs, g_slabs == spp ? pg.s1_magSpp : pg.s1_magTail), sl = ent - tap * g_slabs;
            const int jh = divq(tap, kw, pg.s1_magKw), jw = tap - jh * kw;
            const int f_tap = ((gd.t0 + g_jd * gd.tstep) * pg.kh + (gh.t0 + jh * gh.tstep)) * pg.kw + (gw.t0 + jw * gw.tstep);
            const int f_cc = g_first + sl;
            const int pu = gh.jstep > 0 ? jh * gh.jstep : (kh - 1 - jh) * -gh.jstep;
            const int pv = gw.jstep > 0 ? jw * gw.jstep : (kw - 1 - jw) * -gw.jstep;
            return make_uint2((unsigned)((f_tap * Cred + f_cc * CKB) * 2) | (f_cc == nch - 1 ? 0x80000000u : 0u),
                              (unsigned)((pu * pitch + pv * CP + sl * CKB) * 2));
        };
        for (int e = tid; e < len; e += NT) etab[e] = entry_of(t_begin + e);      // entry table of this group
        RT(1);
        // (Requesting the first three weight slabs right behind the DMA-staged patch's requests -- their L2 round trips overlapped instead
        // of back to back -- was built and measured in the step: 54.57 / 54.37 ms against 53.93 / 53.91 for this order, two builds of the
        // same source in one call.  The slab requests queue behind ~40 patch requests per workgroup either way; issued early they only
        // delay the patch, which everything waits for.  Removed.)
        {
            if (early_patch) { early_patch = false; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }      // requested in the prologue
            else if (pg.dma_patch) stage_patch_dma(g_first, std::true_type{});
            else if (pg.src16) stage_patch(g_first, std::true_type{});
            else stage_patch(g_first, std::false_type{});
            __syncthreads();                                   // table + patch visible
            RT(2);
            issue(etab[0], B0{});
            if (len > 1) issue(etab[1], B1{});
            if (len > 2) issue(etab[2], B2{});
            // slab 0 of every wave has landed (up to two more stay in flight)
            if (len > 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * LW) : "memory");
            else if (len > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        load_a(F0, etab[0]);
        load_b(F0, B0{});
        // step e (cur holds the fragments of entry e): wait for slab e+1, barrier, DMA slab e+3 into the buffer slab e-1 just left,
        // read the fragments of entry e+1, then the MFMAs of entry e
        // The reads of entry e+1 are issued in two batches around the first MFMAs of entry e (pinned with sched_barrier): with all
        // of them ahead of the MFMAs more than 15 LDS operations are outstanding, lgkmcnt cannot count that far and hipcc falls
        // back to lgkmcnt(0) in front of the second MFMA -- the serialisation the second register set is there to remove.
        constexpr int KH = (NKS + 1) / 2;
        // `steady`: entries e+1 .. e+3 exist -- no branch in the step (a conditional load / DMA merges hipcc's wait-count states
        // at the join and it falls back to lgkmcnt(0) in front of the MFMAs again)
        // Table entries travel one step ahead of their use in a three-deep register queue (tq0 = entry e+1: fragments read in step e,
        // tq2 = entry e+3: its slab DMA is issued in step e): the one table read of a step (entry e+4) is issued behind the barrier
        // and consumed a whole step later, so neither the DMA address nor the A-fragment address waits for an LDS round trip there.
        //
        // Where the slab DMA of a step is issued (round 4, measured in the step on MI355X with three builds side by side in one call,
        // twice each): right behind the barrier, before the entry's MFMAs (this order) 53.38 / 53.52 ms; between the two halves of the
        // MFMAs (-DSAVP_RING_DMA_MID) 54.26 / 54.36; behind the MFMAs (-DSAVP_RING_LATE_DMA) 54.01 / 54.01.  A slab requested late
        // lands late: the three-slab look-ahead is worth more than the MFMA issue slots the request sequence occupies.
#ifdef SAVP_RING_LATE_DMA
        constexpr bool LATE = true;
#else
        constexpr bool LATE = false;
#endif
        // What bounds this loop (round 5, stamps-only builds, profiles/r05_ring_loop_findings.md): the LDS ARRAY.  With the slab DMAs ablated
        // the 16x16 gate convolution's loop takes 19.9 k cycles, with them 29.9 k (32x32: 18.5 -> 22.9 k, 8x8: 19.5 -> 30.0 k): every
        // 1 KB LDS-DMA piece lands through the LDS write path (~64 B/clk) and takes ~16 array cycles from the fragment reads, which already
        // need 2 KB per MFMA with the 32 x 32 wave tile (1.5 KB with 32 x 64).  It is NOT the DMA latency: an eight-deep ring (seven slabs of
        // look-ahead, built and measured) made the same loop 5 % slower, and wide slabs give the same 45 % of the matrix pipe per tile.
        // (the table entries are wave-uniform: kept in scalar registers)
        auto sld = [&](int i) { const uint2 t = etab[i]; return make_uint2((unsigned)__builtin_amdgcn_readfirstlane((int)t.x), (unsigned)__builtin_amdgcn_readfirstlane((int)t.y)); };
        {
            uint2 tq0 = sld(min(1, len - 1)), tq1 = sld(min(2, len - 1)), tq2 = sld(min(3, len - 1));
            auto step = [&](int e, Frags& cur, Frags& nxt, auto bn, auto bd, auto steadyc) {
                constexpr bool STEADY = decltype(steadyc)::value;
                const bool more = STEADY || e + 1 < len;
                const bool dma = STEADY || e + 3 < len;
                uint2 td = tq2;
                if (more) {
                    const uint2 tn = tq0;
                    if (STEADY || e + 2 < len) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    const uint2 t4 = sld(min(e + 4, len - 1));
                    if (!LATE && dma) issue(td, bd);
                    load_a(nxt, tn);
                    tq0 = tq1; tq1 = tq2; tq2 = t4;
                }
                if constexpr (STEADY) {
                    load_b(nxt, bn);
#ifdef SAVP_RING_DMA_MID
                    // developer A/B: the slab DMA between the two halves of the entry's MFMAs (fragment reads around the first half)
                    mma(cur, 0, KH);
                    sched_interleave<NKS * (WM + WN), KH * WM * WN, 0>();
                    __builtin_amdgcn_sched_barrier(0);
                    if (dma) issue(td, bd);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(cur, KH, NKS);
#else
                    mma(cur, 0, NKS);
                    sched_interleave<NKS * (WM + WN), NKS * WM * WN, 0>();
#endif
                } else {
                    __builtin_amdgcn_sched_barrier(0);
                    mma(cur, 0, KH);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) load_b(nxt, bn);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(cur, KH, NKS);
                }
#ifdef SAVP_RING_DMA_MID
                if (LATE && more && dma && !STEADY) {
#else
                if (LATE && more && dma) {
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    issue(td, bd);
                }
            };
            RT(3);
            int e = 0;
            for (; e + 6 < len; e += 4) {
                step(e, F0, F1, B1{}, B3{}, std::true_type{});
                step(e + 1, F1, F0, B2{}, B0{}, std::true_type{});
                step(e + 2, F0, F1, B3{}, B1{}, std::true_type{});
                step(e + 3, F1, F0, B0{}, B2{}, std::true_type{});
            }
            for (; e < len; e += 4) {
                step(e, F0, F1, B1{}, B3{}, std::false_type{});
                if (e + 1 < len) step(e + 1, F1, F0, B2{}, B0{}, std::false_type{});
                if (e + 2 < len) step(e + 2, F0, F1, B3{}, B1{}, std::false_type{});
                if (e + 3 < len) step(e + 3, F1, F0, B0{}, B2{}, std::false_type{});
            }
        }
        RT(4);
    }

    // ---- the epilogue reads its arguments from the kernel-argument segment AGAIN, behind the main loop (round 5): the destination's strides
    // and pointers, bias / activation / statistics / norm-backward arguments are ~55 scalar registers that nothing in front of this point
    // needs; loaded with the prologue's s_loads they stayed live through the whole kernel and the prologue spilled and restored ~400 SGPRs
    // through VGPR lanes (v_writelane / v_readlane: a quarter of its ~1 700 instructions).  The pointer is laundered through an empty asm
    // so that the loads cannot be merged with / hoisted to the prologue's; the segment's lines are hot (every workgroup of the launch reads
    // the same 550 bytes).
    ConvPK* pek = (ConvPK*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pek));
    ConvPK& pe = *pek;
    const long long d_sn = dgrad ? pe.x_sn : pe.y_sn, d_sd = dgrad ? pe.x_sd : pe.y_sd;
    const int d_sh = (int)(dgrad ? pe.x_sh : pe.y_sh), d_sw = (int)(dgrad ? pe.x_sw : pe.y_sw);

    RT(5);
    if (ABL(8) && acc[0][0][0] != 123.f) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    if (pe.cell) {
        // ---- "cell" epilogue: per-(image, channel) sum / sum of squares + bf16 rows through LDS ---------------------------------
        // (launcher guarantees: full tiles, Nout % BN == 0, nimg % ni == 0, split-K 1, no bias / activation / beta)
        constexpr int TP = BN / 2 + 4;                        // dwords per tile row (bf16 pairs; 16-byte aligned rows)
        unsigned* T = reinterpret_cast<unsigned*>(smem);      // [BM][TP]
        // statistics: one slot per (32-row block of the tile, column), written by exactly one lane -- no LDS atomics; the blocks of an image
        // are folded in a fixed order below and leave as ONE float64 atomic per (image, channel, workgroup).  A sum of fp32 partials in
        // float64 is exact (no rounding unless the partials span more than 2^29 in magnitude), hence independent of the order in which the
        // workgroups arrive: two runs of the step produce the same bits.
        float* stat = reinterpret_cast<float*>(T + BM * TP);  // [BM / 32][BN][2]
        __syncthreads();                                      // ring and patch are dead
        const bool odd = lane & 1;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int rowb = wm0 + i * 32;
            const int im = rowb >> rsh;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s += v; q += v * v; }
                s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
                if (khalf == 0) {
                    float* d = stat + ((rowb >> 5) * BN + wn0 + 32 * j + l31) * 2;
                    d[0] = s; d[1] = q;
                }
                const int cp = (wn0 + 32 * j + (l31 & ~1)) >> 1;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const float e = acc[i][j][2 * m], o = acc[i][j][2 * m + 1];
                    const float en = dpp_xor1(e), on = dpp_xor1(o);
                    // even lane: row of register 2m, columns (own, neighbour); odd lane: row of register 2m+1, (neighbour, own)
                    const int r = 2 * m + (odd ? 1 : 0);
                    const int row = rowb + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    T[row * TP + cp] = odd ? pack_bf16x2(on, o) : pack_bf16x2(e, en);
                }
            }
        }
        __syncthreads();
        unsigned short* out16 = reinterpret_cast<unsigned short*>(pe.out);
        constexpr int CH = BN / 8;                            // 16-byte pieces per tile row
        for (int idx = tid; idx < BM * CH; idx += NT) {
            const int row = idx / CH, c8 = idx - row * CH;
            const int im = row >> rsh, rr = row - (im << rsh);
            const int gi = img0 + im;
            const int n = (int)fastdiv((unsigned)gi, pe.s1_magDm);
            const int py = oy0 + (rr >> 3), px = ox0 + (rr & 7);
            const uint4 v = *reinterpret_cast<const uint4*>(T + row * TP + c8 * 4);
            unsigned short* dst = out16 + (long long)n * d_sn + (long long)(gd.ob + (gi - n * Dm) * gd.os) * d_sd +
                                  (long long)(gh.ob + py * gh.os) * d_sh + (long long)(gw.ob + px * gw.os) * d_sw + n0 + c8 * 8;
            *reinterpret_cast<uint4*>(dst) = v;
        }
        if (pe.stats) {
            const int bpi = 1 << (rsh - 5);                   // 32-row blocks per image of the tile
            for (int i = tid; i < ni * BN * 2; i += NT) {
                const int im = i / (BN * 2), rem = i - im * (BN * 2);
                const int gi = img0 + im;
                const int n = (int)fastdiv((unsigned)gi, pe.s1_magDm);
                float t = 0.f;
                for (int b = 0; b < bpi; ++b) t += stat[(im * bpi + b) * (BN * 2) + rem];
                unsafeAtomicAdd(pe.stats + ((long long)n * Nout + n0 + (rem >> 1)) * 2 + (rem & 1), (double)t);
            }
        }
        RT(6);
        return;
    }

    // ---- plain epilogue (as conv_patch.hip) -------------------------------------------------------------------------------------
    const int e_sh = d_sh * gh.os, e_sw = d_sw * gw.os;
    const int col0 = n0 + wn0 + l31;
    int pcol[WN];                                             // physical destination channel of this lane's column j (ConvP::gap)
#pragma unroll
    for (int j = 0; j < WN; ++j) pcol[j] = col0 + 32 * j + ((col0 + 32 * j) >= pe.gap_at ? pe.gap : 0);
    const int px0 = ox0 + 4 * khalf;
    const bool plain = (pe.splitk == 1) && !pe.beta && (pe.act == SAVP_ACT_NONE);
    bool biased = false;                                      // the bias is already in the accumulators
    if (pe.stats) {
        // ---- statistics of an fp32 destination (the instance norm behind a generator convolution, normalization.py:146-170): the
        // per-(image, channel) sum / sum of squares of conv + bias leave with this kernel and the norm's own statistics pass (one more
        // launch that re-reads the tensor) disappears.  Launcher guarantees: full tiles, every tile row block inside one image,
        // split-K 1, no activation / beta; one global atomic per (image, channel, workgroup) as in the cell epilogue.
        float* stat = reinterpret_cast<float*>(smem);         // [BM / 32][BN][2]: one slot per (32-row block, column), see the cell epilogue
        __syncthreads();                                      // ring and patch are dead
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int blk = (wm0 + i * 32) >> 5;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = col0 + 32 * j;
                const float bias = (pe.bias && col < Nout) ? pe.bias[col] : 0.f;
                // the sums are taken AROUND THE BIAS (of the accumulators alone): sum(y - b), sum((y - b)^2).  One-pass variance
                // E[v^2] - E[v]^2 cancels when |mean| >> std; a large bias -- the usual reason for a large mean -- no longer enters it
                // (savp_instnorm_act_fwd(stats_ready, stats_shift = this bias) adds it back to the mean)
                float sm = 0.f, q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; sm += v; q += v * v; acc[i][j][r] = v + bias; }
                sm += __shfl_xor(sm, 32); q += __shfl_xor(q, 32);
                if (khalf == 0 && col < Nout) {
                    float* d = stat + (blk * BN + wn0 + 32 * j + l31) * 2;
                    d[0] = sm; d[1] = q;
                }
            }
        }
        biased = true;
        __syncthreads();
        const int bpi = 1 << (rsh - 5);
        for (int i = tid; i < ni * BN * 2; i += NT) {
            const int im = i / (BN * 2), rem = i - im * (BN * 2);
            const int gi = img0 + im;
            const int n = (int)fastdiv((unsigned)gi, pe.s1_magDm);
            if (n0 + (rem >> 1) < Nout) {
                float t = 0.f;
                for (int b = 0; b < bpi; ++b) t += stat[(im * bpi + b) * (BN * 2) + rem];
                unsafeAtomicAdd(pe.stats + ((long long)n * Nout + n0 + (rem >> 1)) * 2 + (rem & 1), (double)t);
            }
        }
    }
    if (pe.nb_ws) {
        // ---- backward statistics of the instance norm whose OUTPUT gradient this kernel produces (SavpConvArgs.nb_*): the destination's
        // logical channels [nb_c0, nb_c0 + nb_nc) are dy of y = act(gamma * xhat + beta), xhat = (x - mean) * rstd; the two sums that norm's
        // backward needs, sum(dy') and sum(dy' * xhat) with dy' = dy * act'(gamma * xhat + beta), leave with the accumulators -- the norm's
        // statistics launch (one more pass over x and dy) disappears.  The mask is the forward apply pass's expression, bit for bit.
        // Launcher guarantees: whole tiles, a row block inside one image, unit destination strides per pixel, split-K 1, no act / beta.
        float* stat = reinterpret_cast<float*>(smem);         // [BM / 32][BN][2]: one slot per (32-row block, column), see the cell epilogue
        __syncthreads();                                      // ring and patch are dead
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int rowb = wm0 + i * 32;
            const int im = rowb >> rsh;
            const int gi = img0 + im;
            const int n = (int)fastdiv((unsigned)gi, pe.s1_magDm);
            const int py0 = oy0 + ((rowb - (im << rsh)) >> 3);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int cc = col0 + 32 * j - pe.nb_c0;         // channel of the norm
                const bool in = cc >= 0 && cc < pe.nb_nc && col0 + 32 * j < Nout;
                const int cq = in ? cc : 0;
                const float mu = pe.nb_mean[(long long)n * pe.nb_nc + cq], rs = pe.nb_rstd[(long long)n * pe.nb_nc + cq];
                const float ga = pe.nb_gamma[cq], be = pe.nb_beta[cq];
                const float* xp = pe.nb_x + (long long)n * pe.nb_x_sn + ((long long)py0 * Wm + px0) * pe.nb_x_sp + cq;
                float xv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) xv[r] = xp[((r >> 2) * Wm + (r & 3)) * pe.nb_x_sp];      // all 16 loads in flight
                float sm = 0.f, q = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float xh = (xv[r] - mu) * rs;
                    const float y = xh * ga + be;
                    const float gr = pe.nb_act == 1 ? (y > 0.f ? 1.f : 0.f) : (pe.nb_act == 2 ? (y > 0.f ? 1.f : pe.nb_alpha) : 1.f);
                    const float d = acc[i][j][r] * gr;
                    sm += d; q += d * xh;
                }
                sm += __shfl_xor(sm, 32); q += __shfl_xor(q, 32);
                if (khalf == 0 && in) {
                    float* d = stat + ((rowb >> 5) * BN + wn0 + 32 * j + l31) * 2;
                    d[0] = sm; d[1] = q;
                }
            }
        }
        __syncthreads();
        const int bpi = 1 << (rsh - 5);
        for (int i = tid; i < ni * BN * 2; i += NT) {
            const int im = i / (BN * 2), rem = i - im * (BN * 2);
            const int gi = img0 + im;
            const int n = (int)fastdiv((unsigned)gi, pe.s1_magDm);
            const int cc = n0 + (rem >> 1) - pe.nb_c0;
            if (cc >= 0 && cc < pe.nb_nc && n0 + (rem >> 1) < Nout) {
                float t = 0.f;
                for (int b = 0; b < bpi; ++b) t += stat[(im * bpi + b) * (BN * 2) + rem];
                unsafeAtomicAdd(pe.nb_ws + ((long long)n * pe.nb_nc + cc) * 2 + (rem & 1), (double)t);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int rowb = wm0 + i * 32;
        const int im = rowb >> rsh;
        const int py0 = oy0 + ((rowb - (im << rsh)) >> 3);
        const int gi = img0 + im;
        if (gi >= nimg) continue;
        const int n = (int)fastdiv((unsigned)gi, pe.s1_magDm);
        float* __restrict__ dst = pe.out + (long long)n * d_sn + (long long)(gd.ob + (gi - n * Dm) * gd.os) * d_sd +
                                  (long long)(gh.ob + py0 * gh.os) * d_sh +
                                  (long long)(gw.ob + px0 * gw.os) * d_sw;
        const bool full = (py0 + 4 <= Hm) && (ox0 + TW <= Wm);
        if (plain && full) {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (col0 + 32 * j >= Nout) continue;
                const float bias = (pe.bias && !biased) ? pe.bias[pcol[j]] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(r >> 2) * e_sh + (r & 3) * e_sw + pcol[j]] = acc[i][j][r] + bias;
            }
            continue;
        }
        const float* __restrict__ aux = pe.aux ? pe.aux + (dst - pe.out) : nullptr;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            if (col0 + 32 * j >= Nout) continue;
            const float bias = (pe.bias && split == 0 && !biased) ? pe.bias[pcol[j]] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (!full && (py0 + (r >> 2) >= Hm || px0 + (r & 3) >= Wm)) continue;
                const int off = (r >> 2) * e_sh + (r & 3) * e_sw + pcol[j];
                float v = acc[i][j][r] + bias;
                if (pe.splitk > 1) {                            // this split's share: its own slice of the scratch, folded in split order afterwards
                    (pe.part + (long long)split * pe.part_sz + (dst - pe.out))[off] = v;
                    continue;
                }
                if (pe.beta) v += dst[off];
                if (pe.act == SAVP_ACT_LRELU) v = fmaxf(v, pe.alpha * v);
                else if (pe.act == SAVP_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                else if (pe.act == SAVP_ACT_DLRELU_FROM_OUT) v *= (aux[off] > 0.f ? 1.f : pe.alpha);
                dst[off] = v;
            }
        }
    }
    RT(6);
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned g_ring_zero[4] = {0u, 0u, 0u, 0u};      // source of the DMA-staged patch's zero slots

template <int NW, int WM, int WN, int NKS>
static hipError_t launch_ring(const ConvP& p, dim3 grid, size_t lds, hipStream_t st) {
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipFuncSetAttribute((const void*)conv_ring_kernel<NW, WM, WN, NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    if (g_savp_prof_start) {                                     // bench.py: kernel-only timing of this one launch (common.hip)
        hipExtLaunchKernelGGL((conv_ring_kernel<NW, WM, WN, NKS>), grid, dim3(64 * NW), lds, st, g_savp_prof_start, g_savp_prof_stop, 0, p);
        g_savp_prof_start = g_savp_prof_stop = nullptr;
    } else {
        hipLaunchKernelGGL((conv_ring_kernel<NW, WM, WN, NKS>), grid, dim3(64 * NW), lds, st, p);
    }
    return hipGetLastError();
}

template <int NW, int WM, int WN>
static hipError_t launch_ring_nks(const ConvP& p, int nks, dim3 grid, size_t lds, hipStream_t st) {
    switch (nks) {
        case 1: return launch_ring<NW, WM, WN, 1>(p, grid, lds, st);
        case 2: return launch_ring<NW, WM, WN, 2>(p, grid, lds, st);
        case 3: return launch_ring<NW, WM, WN, 3>(p, grid, lds, st);
        case 4: return launch_ring<NW, WM, WN, 4>(p, grid, lds, st);
        case 5: return launch_ring<NW, WM, WN, 5>(p, grid, lds, st);
        default: return launch_ring<NW, WM, WN, 6>(p, grid, lds, st);
    }
}

template <int NW>
static hipError_t launch_ring_tile(const ConvP& p, int wm, int wn, int nks, dim3 grid, size_t lds, hipStream_t st) {
    if (wm == 2 && wn == 2) return launch_ring_nks<NW, 2, 2>(p, nks, grid, lds, st);
    if (wm == 2 && wn == 1) return launch_ring_nks<NW, 2, 1>(p, nks, grid, lds, st);
    if (wm == 1 && wn == 2) return launch_ring_nks<NW, 1, 2>(p, nks, grid, lds, st);
    // wide slabs (tile bit 0x1000): 8 / 9 k-steps per entry, instantiated for the 32 x 32 wave tile only (two fragment sets of a wider
    // wave tile would not fit 256 VGPRs)
    if (nks == 8) return launch_ring<NW, 1, 1, 8>(p, grid, lds, st);
    if (nks == 9) return launch_ring<NW, 1, 1, 9>(p, grid, lds, st);
    return launch_ring_nks<NW, 1, 1>(p, nks, grid, lds, st);
}

struct RingPlan { int nw, wm, wn, nks; size_t lds; dim3 grid; };

// Geometry of one (nw, wm, wn) choice; false = this choice cannot run the problem (LDS, cell-mode tiling constraints, ...).
static bool ring_plan(ConvP& p, const SavpConvArgs* a, int nw, int wm, int wn, RingPlan& pl) {
    const bool dg = a->mode == SAVP_CONV_DGRAD;
    const int Cred = dg ? a->Cy : a->Cx, Nout = dg ? a->Cx : a->Cy;
    const int Dm = dg ? a->D : a->Do;
    const int phases = dg ? a->sh * a->sw : 1;
    const int Hm = dg ? (a->H + a->sh - 1) / a->sh : a->Ho, Wm = dg ? (a->W + a->sw - 1) / a->sw : a->Wo;
    const long long dH = dg ? a->H : a->Ho, dW_ = dg ? a->W : a->Wo;
    const long long d_sn = dg ? a->x_sn : a->y_sn, d_sh = dg ? a->x_sh : a->y_sh, d_sw = dg ? a->x_sw : a->y_sw;
    const long long d_sd = dg ? a->x_sd : a->y_sd;
    const int tW = (Wm + 7) / 8;
    // channel slabs: with the DMA ring a slab costs (k-steps + ~0.5), so exact fits beat fewer, wider slabs
    const int Cp16 = (Cred + 15) & ~15;
    // Wide slabs (tile bit 0x1000, 32 x 32 wave tile): up to 9 k-steps per (tap, slab) entry -- 144 channels are ONE slab (25 entries
    // instead of 75 for the 16x16 gate convolution), 256 / 512 channels split 2 x 8 / 4 x 8 without padding.  The per-entry costs that
    // do not scale with the slab (barrier, table, address arithmetic) are paid a third as often; the ring needs more LDS per slot.
    const int kmax = ((a->tile & 0x1000) && wm == 1 && wn == 1) ? 9 : 6;
    int nch = 0, nks = 0;
    double best = 1e30;
    for (int c = (Cp16 + 16 * kmax - 1) / (16 * kmax); c <= (Cp16 + 95) / 96 + 3; ++c) {
        const int k = (Cp16 / 16 + c - 1) / c;
        if (k < 1 || k > kmax || k == 7) continue;
        const double cost = c * (k + 0.5);
        if (cost < best) { best = cost; nch = c; nks = k; }
    }
    if (!nch) return false;
    if ((a->tile & 0x1000) && nks <= 6) return false;          // nothing wide to offer: let the tuner's plain candidate stand for it
    const int TH = 2 * nw * wm;
    int tih = 4;
    while (tih < TH && tih < Hm) tih *= 2;
    const int ni = TH / tih;
    const int PH = dg ? tih + (a->kh + a->sh - 1) / a->sh - 1 : (tih - 1) * a->sh + a->kh;
    const int PW = dg ? 8 + (a->kw + a->sw - 1) / a->sw - 1 : 7 * a->sw + a->kw;
    const int BM = 16 * nw * wm, BN = 64 * wn, NT = 64 * nw;
    const int RS = 2 * nks + 1;
    const int LW = (BN * RS + NT - 1) / NT;
    const size_t ringb = (size_t)4 * LW * NT * 16;
    const size_t budget = 160 * 1024;
    int spp = nch, pitch = 0;
    size_t lds = 0;
    for (; spp >= 1; --spp) {
        const int CP = spp * nks * 16 + 8;
        const int x = (8 - (PW * (CP / 8)) % 16 + 16) % 16;
        pitch = PW * CP + 8 * x;
        // + dummy slot of stage_patch + entry table of one group (8 bytes per (tap, slab) entry)
        lds = ringb + (size_t)ni * PH * pitch * 2 + 16 + (size_t)8 * (dg ? ((a->kh + a->sh - 1) / a->sh) * ((a->kw + a->sw - 1) / a->sw) : a->kh * a->kw) * spp;
        if (lds <= budget) break;
    }
    if (spp < 1) return false;
    const bool cell = a->out_bf16 != 0;
    if (cell) {
        const size_t epi = (size_t)BM * (BN / 2 + 4) * 4 + (size_t)(BM / 32) * BN * 2 * 4;
        if (epi > lds) lds = epi;
        if (lds > budget) return false;
        const long long nimg = (long long)a->N * Dm;
        if (a->bias || a->act != SAVP_ACT_NONE || a->beta || Hm % tih || Wm % 8 || Nout % BN || nimg % ni || (d_sw % 8) || (d_sh % 8) ||
            (d_sn % 8) || (d_sd % 8) || ((((uintptr_t)(dg ? a->x : a->y)) & 15) != 0) || phases != 1)
            return false;
    }
    if ((long long)a->N * Dm >= (1 << 24)) return false;
    if ((long long)ni * PH * PW * spp * nks * 4 >= (1 << 24)) return false;
    p.cell = cell ? 1 : 0;
    p.stats = (double*)a->stats;
    if (a->stats && !cell) {
        // statistics of an fp32 destination: whole tiles only (every accumulator is a real output), no split-K, nothing after the bias
        const long long nimg = (long long)a->N * Dm;
        const bool even = !dg || (a->H % a->sh == 0 && a->W % a->sw == 0);       // every output phase has the same extent
        if (a->act != SAVP_ACT_NONE || a->beta || Hm % tih || Wm % 8 || nimg % ni || !even || (a->splitk > 1) ||
            (size_t)(BM / 32) * BN * 2 * 4 > lds)
            return false;
    }
    if (a->nb_ws) {
        // norm-backward statistics: whole tiles (every accumulator is a real output of one image), unit strides (destination pixel (y, x)
        // is pixel y * Wm + x of nb_x), depth 1, no split-K, nothing after the accumulators, an fp32 destination
        const long long nimg = (long long)a->N * Dm;
        // (split-K is fine: both sums are linear in the accumulators -- the mask depends on x only -- so every split adds its share)
        if (cell || a->stats || a->act != SAVP_ACT_NONE || a->beta || Hm % tih || Wm % 8 || nimg % ni || phases != 1 || Dm != 1 ||
            a->sh != 1 || a->sw != 1 || a->nb_c0 + a->nb_nc > Nout || (size_t)(BM / 32) * BN * 2 * 4 > lds)
            return false;
    }
    p.s1_ph = PH; p.s1_pw = PW; p.s1_th = (Hm + tih - 1) / tih; p.s1_tw = tW; p.s1_tih = tih;
    p.s1_pitch = pitch; p.s1_nch = nch; p.s1_spp = spp;
    p.s1_magPI = magic40(PH * PW * spp * nks * 4); p.s1_magPW = magic40(PW); p.s1_magC4 = magic40(spp * nks * 4);
    p.s1_magDm = magic40(Dm);
    // LDS-DMA patch staging: bf16 source whose pixel rows are 16-byte aligned runs (option "ring_dma" = 0: the VGPR path, for A/B)
    {
        const long long ssn = dg ? a->y_sn : a->x_sn, ssd = dg ? a->y_sd : a->x_sd, ssh = dg ? a->y_sh : a->x_sh, ssw = dg ? a->y_sw : a->x_sw;
        const void* sptr = dg ? a->y : a->x;
        const int CP = spp * nks * 16 + 8;
        p.dma_patch = (savp_opt(OPT_RING_DMA) && a->src_bf16 && (ssn % 8 == 0) && (ssd % 8 == 0) && (ssh % 8 == 0) && (ssw % 8 == 0) &&
                       ((((uintptr_t)sptr) & 15) == 0) && p.zero16 && (long long)ni * PH * (pitch / 8) < (1 << 24) && (long long)PH * (pitch / 8) < 65536 && (pitch % 8 == 0) &&
                       (nks <= 2 || (pitch / 8 <= 512 && ssw * (long long)(a->W + a->kw) + Cred < (1ll << 30)))) ? 1 : 0;       // row-wise staging (nks >= 3): <= 8 DMA instructions per patch row, 32-bit in-row offsets
        p.s1_magPI8 = magic40(PH * (pitch / 8)); p.s1_magP8 = magic40(pitch / 8); p.s1_magC8 = magic40(CP / 8);
    }
    p.tm = (int)(((long long)a->N * Dm + ni - 1) / ni) * p.s1_th * tW; p.tn = (Nout + BN - 1) / BN;
    const long long tiles = (long long)p.tm * p.tn;
    const long long iters = (long long)(dg ? (a->kh / a->sh) * (a->kw / a->sw) : a->kh * a->kw) * nch * a->kd;
    int splitk = a->splitk;
    if (a->act != SAVP_ACT_NONE || cell || a->stats) splitk = 1;
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
        // the destination must be one dense block: split s stores its share at the same offsets of its slice of the scratch (a column
        // gap's channels are cleared by the fold: nobody reads them)
        const long long Cd = Nout + p.gap;
        const bool dense = (d_sw == Cd) && (d_sh == dW_ * Cd) && (dD == 1 || d_sd == dH * dW_ * Cd) &&
                           (d_sn == dD * dH * dW_ * Cd);
        if (!dense) splitk = 1;
        else {
            p.part_sz = (long long)a->N * dD * dH * dW_ * Cd;
            splitk = splitk_fit(a, splitk, p.part_sz);
            p.part = (float*)a->ws;
        }
    }
    p.splitk = splitk;
    patch_launch_constants(p, a, phases, tih, nch, spp, tW, splitk);
    p.wwarm = savp_opt(OPT_RING_WWARM) ? 1 : 0;
    p.early = savp_opt(OPT_RING_EARLY) ? 1 : 0;
    pl.nw = nw; pl.wm = wm; pl.wn = wn; pl.nks = nks; pl.lds = lds;
    pl.grid = dim3((unsigned)(p.tm * p.tn), (unsigned)phases, (unsigned)splitk);
    return true;
}

// Returns true when the ring kernel handled the call (*rc = status); false = not applicable (the caller falls back).
// SavpConvArgs.out_bf16 selects the cell epilogue: bf16 destination + optional statistics; with an automatic tile the first
// (waves, tile) choice whose tiling can honour it is taken, a forced tile that cannot is refused (false).
bool conv_ring_try(ConvP& p, const SavpConvArgs* a, int wm, int wn, hipStream_t st, int* rc, bool dry) {
    const bool dg = a->mode == SAVP_CONV_DGRAD;
    const int Cred = dg ? a->Cy : a->Cx, Nout = dg ? a->Cx : a->Cy;
    const long long ssn = dg ? a->y_sn : a->x_sn, ssh = dg ? a->y_sh : a->x_sh, ssw = dg ? a->y_sw : a->x_sw;
    const void* sptr = dg ? a->y : a->x;
    const int sal = p.src16 ? 8 : 4;                             // source alignment in elements (8 / 16 bytes per load)
    const bool src_al = (ssn % sal == 0) && (ssh % sal == 0) && (ssw % sal == 0) && ((((uintptr_t)sptr) & (p.src16 ? 7 : 15)) == 0);
    const long long ssd = dg ? a->y_sd : a->x_sd;
    if (!(p.bf16 && p.w16 && a->sd == 1 && a->sh <= 4 && a->sw <= 4 && a->kh >= a->sh && a->kw >= a->sw && (Cred % 8 == 0) &&
          src_al && ssd % sal == 0))
        return false;
    const int Hm = dg ? (a->H + a->sh - 1) / a->sh : a->Ho, Wm = dg ? (a->W + a->sw - 1) / a->sw : a->Wo;
    const long long dH = dg ? a->H : a->Ho;
    const long long d_sh = dg ? a->x_sh : a->y_sh;
    if (ssh * (a->H + a->kh) >= (1ll << 30) || d_sh * (dH + 16) >= (1ll << 30) || (long long)(Nout + p.gap) * a->kh * a->kw * a->kd * Cred >= (1ll << 30))
        return false;
    if (p.gap && (a->out_bf16 || a->stats)) return false;        // the gap lives in the plain epilogue only
    if ((long long)Hm * Wm < 16) return false;
    // address of g_ring_zero: a device symbol has one address PER DEVICE, so the cache is indexed by the current device's ordinal (a
    // process that drives several GPUs -- SAVP_DIST_BACKEND=gloo with differing device indices, library users outside the
    // one-process-per-GPU runners -- must not hand device 0's pointer to a kernel on device 1)
    static const void* zero16_of[64] = {nullptr};
    int dev_ord = 0;
    if (hipGetDevice(&dev_ord) != hipSuccess || dev_ord < 0 || dev_ord >= 64) dev_ord = 0;
    if (!dry && !zero16_of[dev_ord] && hipGetSymbolAddress((void**)&zero16_of[dev_ord], HIP_SYMBOL(g_ring_zero)) != hipSuccess) zero16_of[dev_ord] = nullptr;
    p.zero16 = zero16_of[dev_ord];
    RingPlan pl;
    bool ok = false;
    if (wm) {
        ok = ring_plan(p, a, (a->tile & 0x400) ? 8 : 4, wm, wn, pl);
    } else {
        const int tW = (Wm + 7) / 8;
        const int wn0 = Nout > 64 ? 2 : 1;
        const long long t16 = (long long)a->N * ((Hm + 15) / 16) * tW * ((Nout + 64 * wn0 - 1) / (64 * wn0));
        const int wm0 = (t16 >= 256 && Hm >= 16) ? 2 : 1;
        const int cand[5][3] = {{8, wm0, wn0}, {8, 1, wn0}, {4, 1, wn0}, {8, 1, 1}, {4, 1, 1}};
        for (int i = 0; i < 5 && !ok; ++i) ok = ring_plan(p, a, cand[i][0], cand[i][1], cand[i][2], pl);
    }
    if (!ok) return false;
    if (dry) return true;                                        // savp_conv_stats_ok: the plan exists, nothing is launched
    ablate_init();
    hipError_t err = (pl.nw == 8) ? launch_ring_tile<8>(p, pl.wm, pl.wn, pl.nks, pl.grid, pl.lds, st)
                                  : launch_ring_tile<4>(p, pl.wm, pl.wn, pl.nks, pl.grid, pl.lds, st);
    if (p.splitk > 1 && err == hipSuccess) {
        splitk_fold(p.out, p.part, p.splitk, p.part_sz, a->beta, Nout + p.gap, p.gap ? p.gap_at : 0, p.gap, st);
        err = hipGetLastError();
    }
    *rc = (err == hipSuccess) ? SAVP_OK : SAVP_ELAUNCH;
    return true;
}
