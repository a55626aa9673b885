// conv_common.h -- shared declarations of the convolution kernels (conv_igemm.hip, conv_patch.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "savp_hip.h"
#include "opts.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BK 32
#define BKP 36          // padded K row (floats) for the row-major-K LDS layout
#define NTHREADS 256

struct DimGeom {          // one spatial dimension of the (possibly phase-restricted) problem
    int Mdim;             // extent of the M-grid along this dim
    int base, mstep;      // source coord = base + m*mstep + j*jstep
    int jstep;
    int nt;               // number of (reduced) taps
    int t0, tstep;        // full weight tap index = t0 + j*tstep
    int ob, os;           // destination coord = ob + m*os
    int srcN;             // source extent (bounds)
};

__host__ __device__ __forceinline__ DimGeom make_geom(bool dgrad, int f, int In, int Out, int k, int s, int p) {
    DimGeom g;
    if (!dgrad) {
        g.Mdim = Out; g.base = -p; g.mstep = s; g.jstep = 1; g.nt = k; g.t0 = 0; g.tstep = 1;
        g.ob = 0; g.os = 1; g.srcN = In;
    } else {
        int u0 = (f + p) % s;
        g.nt = (k > u0) ? (k - u0 + s - 1) / s : 0;
        g.base = (f + p - u0) / s; g.mstep = 1; g.jstep = -1;
        g.t0 = u0; g.tstep = s;
        g.Mdim = (In > f) ? (In - f + s - 1) / s : 0;
        g.ob = f; g.os = s; g.srcN = Out;
    }
    return g;
}

struct ConvP {
    int mode;
    int N, D, H, W, Cx;
    int Do, Ho, Wo, Cy;
    int kd, kh, kw, sd, sh, sw, pd, ph, pw;
    int beta, act;
    float alpha;
    const float* x; long long x_sn, x_sd, x_sh, x_sw;
    const float* y; long long y_sn, y_sd, y_sh, y_sw;
    const float* w;
    const unsigned short* w16;   // optional bf16 copy of the packed weights (same layout)
    float* out;          // destination (y for FPROP, x for DGRAD, dW for WGRAD)
    const float* bias;
    const float* aux;
    int splitk;
    float* part; long long part_sz;   // split-K > 1: split s writes its share of the (dense) destination block to part + s * part_sz; splitk_fold adds them in split order
    int bf16;
    int tm, tn;          // tile counts (1-D XCD-aware launch grids)
    unsigned long long magW, magHW, magDHW;   // WGRAD fast division by Wo, Ho*Wo, Do*Ho*Wo
    // patch kernel (conv_patch.hip): LDS geometry chosen by the launcher
    int s1_pitch, s1_nch, s1_spp;             // patch row pitch (bf16 elements), channel slabs, slabs per patch group
    int s1_ph, s1_pw, s1_th, s1_tw, s1_tih;   // patch rows / columns per image, tiles per image (rows, columns), tile rows per image
    unsigned long long s1_magDm;              // fastdiv by the depth of the output grid
    unsigned long long s1_magPI, s1_magPW, s1_magC4;   // fastdiv by the float4 count of one image's patch, the patch width, float4 per pixel
    // ring kernel (conv_ring.hip)
    int src16;                                // source activations are bf16 (strides in elements)
    int cell;                                 // bf16 destination through LDS + per-(sample, channel) statistics
    double* stats;                            // [N][Nout][2] sum / sum of squares, float64, atomically accumulated: exact, hence order-independent (may be null)
    // bf16 source staged by LDS-DMA (src16 && dma_patch): the patch as a sequence of 16-byte slots
    int dma_patch;                            // 1: stage_patch_dma (16-byte aligned bf16 source)
    const void* zero16;                       // 16 zero bytes in global memory (source of halo / padding slots)
    unsigned long long s1_magPI8, s1_magP8, s1_magC8;   // fastdiv by slots per image (PH * pitch / 8), per patch row (pitch / 8), per pixel (CP / 8)
    // launch constants of conv_ring_kernel, worked out by the launcher (pre = 1: FPROP, or DGRAD with unit H/W strides -- every
    // workgroup then sees the same geometry; 0: the kernel derives them from blockIdx.y's output phase).  Fifteen scalar integer
    // divisions leave the kernel's prologue this way (1-2 k of its 8-10 k cycles, profiles/r03_ring_prologue_stamps.log).  The same
    // change made conv_patch_kernel 10-19 % SLOWER in the step (its main loop's register allocation) and was not kept there.
    int pre;
    // conv_ring_kernel: backward statistics of the instance norm whose output gradient this convolution produces (SavpConvArgs.nb_*)
    const float* nb_x; long long nb_x_sn, nb_x_sp; const float *nb_mean, *nb_rstd, *nb_gamma, *nb_beta; double* nb_ws;
    int nb_c0, nb_nc, nb_act; float nb_alpha;
    int gap_at, gap;                          // conv_ring_kernel: logical output column c >= gap_at is physical weight row / destination channel c + gap (SavpConvArgs.dst_gap)
    int wwarm;                                // conv_ring_kernel: warm the L2 with the column tile's weight block first (option ring_wwarm)
    int early;                                // conv_ring_kernel: request the first group's DMA-staged patch at the top of the prologue (option ring_early)
    DimGeom gD, gH, gW;
    int s1_tih_sh;                            // log2(s1_tih): tile rows per image are a power of two
    int s1_ngs, s1_itper;                     // slab groups per depth tap, (tap, slab) entries per K split
    unsigned long long s1_magTm, s1_magTHW, s1_magTW;    // fastdiv by tm, tiles per image (th * tw), tile columns
    unsigned long long s1_magKw, s1_magSpp, s1_magTail;  // fastdiv by the reduced tap columns, slabs per group, slabs of the last group
};

__device__ __forceinline__ unsigned fastdiv(unsigned p, unsigned long long magic) {
    // floor(p / d) for p < 2^24, d < 2^16 with magic = ceil(2^40 / d)
    return (unsigned)(((unsigned long long)p * magic) >> 40);
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// XCD-aware workgroup -> logical id map (MI355X: hardware block b runs on XCD b % 8, each XCD has a private L2).
// Consecutive LOGICAL ids land on the same XCD, so tiles that share an operand (same weight column block, neighbouring
// pixel rows, same K split) hit one L2 instead of being fetched once per XCD.  Bijective for any n (speed only).
__device__ __forceinline__ int xcd_logical(int b, int n) {
    const int q = n >> 3, r = n & 7, xcd = b & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}


// ceil(2^40 / d): magic number of fastdiv()
static inline unsigned long long magic40(int d) {
    if (d <= 0) d = 1;
    return ((1ULL << 40) + (unsigned long long)d - 1ULL) / (unsigned long long)d;
}

// Launch constants of conv_ring_kernel (ConvP::pre and friends) from the launcher's tiling: identical for every workgroup
// unless a strided DGRAD splits the output into phases (then pre = 0 and the kernel derives them from blockIdx.y).
static inline void patch_launch_constants(ConvP& p, const SavpConvArgs* a, int phases, int tih, int nch, int spp, int tW, int splitk) {
    const bool dg = a->mode == SAVP_CONV_DGRAD;
    int sh_ = 0;
    while ((1 << sh_) < tih) ++sh_;
    p.s1_tih_sh = sh_;
    const int ngs = (nch + spp - 1) / spp;
    p.s1_ngs = ngs;
    p.pre = 0;
    if (phases != 1 || (dg && (a->sh != 1 || a->sw != 1))) return;
    p.gD = make_geom(dg, 0, a->D, a->Do, a->kd, 1, a->pd);
    p.gH = make_geom(dg, 0, a->H, a->Ho, a->kh, a->sh, a->ph);
    p.gW = make_geom(dg, 0, a->W, a->Wo, a->kw, a->sw, a->pw);
    const long long it_all = (long long)p.gD.nt * p.gH.nt * p.gW.nt * nch;
    p.s1_itper = (int)((it_all + splitk - 1) / splitk);
    const int thw = p.s1_th * tW;
    p.s1_magTm = magic40(p.tm); p.s1_magTHW = magic40(thw); p.s1_magTW = magic40(tW);
    p.s1_magKw = magic40(p.gW.nt); p.s1_magSpp = magic40(spp); p.s1_magTail = magic40(nch - (ngs - 1) * spp);
    // fastdiv: numerators < 2^24, divisors < 2^16
    if ((long long)p.tm * p.tn < (1 << 24) && p.tm < 65536 && thw < 65536 && it_all < (1 << 24) && p.gW.nt >= 1 && p.gH.nt >= 1) p.pre = 1;
}

// conv_thin.hip: FPROP / WGRAD of a 3x3(x3) stride-1 convolution with Cx <= 4, Cy = 32 (bf16 mode).  true = handled.
bool conv_thin_try(const SavpConvArgs* a, hipStream_t st, int* rc);
bool conv_thin_applies(const SavpConvArgs* a);
long long conv_thin_workspace_bytes(const SavpConvArgs* a);

// conv_s2dgrad.hip: DGRAD of a 4x4 stride-(1,2,2) convolution with 32 input channels, all four output phases per workgroup.
bool conv_s2dgrad_try(const SavpConvArgs* a, hipStream_t st, int* rc);
bool conv_s2dgrad_applies(const SavpConvArgs* a);

extern thread_local hipEvent_t g_savp_prof_start, g_savp_prof_stop;      // common.hip: savp_prof_arm

// Deterministic split-K (common.hip).  The K splits of a FPROP / DGRAD launch used to meet in the destination through float atomics, in
// arrival order: two runs of one step differed in the last bit, and in the bf16 datapath that bit decides roundings downstream.  Now split
// s stores its share of the (dense) destination block in its own slice of the caller's scratch (SavpConvArgs.ws) and one fold launch adds
// the slices in split order -- where the old path launched a fill kernel to clear the block, so the launch count is unchanged.
// splitk_fit: the largest split count <= want whose slices fit the scratch (1 = no scratch: the call runs unsplit).
int splitk_fit(const SavpConvArgs* a, int want, long long block_elems);
// out[i] = (beta ? out[i] : 0) + sum_s part[s * n + i]; channels [gap_at, gap_at + gap) of every Cd-wide pixel are cleared instead (beta 0)
void splitk_fold(float* out, const float* part, int splitk, long long n, int beta, int Cd, int gap_at, int gap, hipStream_t st);

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Developer aid: build with SAVP_EXTRA_FLAGS=-DSAVP_CONV_ABLATE and set SAVP_ABLATE=<bits> to switch off parts of a kernel
// (1 global loads, 2 MFMAs, 4 patch staging, 8 epilogue, 16 everything, 32 main loop, 64 LDS staging) when attributing
// its time.  Not compiled into the shipped library.
#ifdef SAVP_CONV_ABLATE
#include <stdlib.h>
static __constant__ int g_ablate;
#define ABL(bit) (g_ablate & (bit))
static inline void ablate_init() {
    static bool done = false;
    if (!done) { const char* e = getenv("SAVP_ABLATE"); int v = e ? atoi(e) : 0; hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &v, sizeof(int)); done = true; }
}
#else
#define ABL(bit) false
static inline void ablate_init() {}
#endif

// conv_patch.hip: LDS patch kernel for 2-D stride-1 FPROP/DGRAD in bf16.  Returns true when it handled the call
// (*rc = SAVP_* status); false = not applicable, the caller falls back to the generic kernel.
bool conv_patch_try(ConvP& p, const SavpConvArgs* a, int wm, int wn, bool forced, hipStream_t st, int* rc);
// conv_wgrad_patch.hip: LDS patch WGRAD (2-D stride-1, bf16); same contract.
bool conv_wgrad_patch_try(ConvP& p, const SavpConvArgs* a, hipStream_t st, int* rc, long long* plan_bytes = nullptr);   // plan_bytes: only say how much scratch the call would use
// common.hip: out[i] += part[0 * slice + i] + part[1 * slice + i] + ... (split order; slice = n when 0): the deterministic weight gradient
void wgrad_fold(float* out, const float* part, int nsplit, long long n, hipStream_t st, long long slice = 0);
// conv_ring.hip: LDS patch + LDS-DMA weight ring (+ fused bf16 / statistics epilogue); same contract.
bool conv_ring_try(ConvP& p, const SavpConvArgs* a, int wm, int wn, hipStream_t st, int* rc, bool dry = false);
// conv_gate.hip: the ConvLSTM gate convolution's own kernel (weights in B-fragment order, SavpConvArgs.w_frag); same contract.
bool conv_gate_applies(const SavpConvArgs* a);
bool conv_gate_try(const SavpConvArgs* a, hipStream_t st, int* rc);
struct SavpConvLstmCellArgs;
bool conv_gate_cell_try(const SavpConvLstmCellArgs* c, hipStream_t st, int* rc);   // the whole cell forward in one launch (w_frag_il)
