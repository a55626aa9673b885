// conv_wgrad_patch.hip -- weight gradient of 2-D / 3-D convolutions (strides 1 - 2, depth stride 1) on the bf16 MFMA pipe, LDS-patch formulation.
//
//   dW[u, v, cx, cy] = sum over (n, oy, ox) of x[n, oy + u - ph, ox + v - pw, cx] * dy[n, oy, ox, cy]
//
// The generic WGRAD kernel (conv_igemm.hip) is an im2col GEMM whose 128x128 output tiles each re-gather their operands
// from L2: measured on the ConvLSTM layers it spends 72 % of its time in those loads (14 GB of L2->CU traffic for a
// 0.76 GB problem).  Here the reduction runs over 8x8 pixel tiles of one image at a time: the x patch (tile + halo) and
// the dy tile are converted to bf16 and parked in LDS ONCE, and every (tap, 16-channel group) row block of dW reads its A
// operand from that patch at a tap-shifted address.  Both operands are pixel-major in LDS while the MFMA wants
// k(=pixel)-contiguous fragments: ds_read_b64_tr_b16 (the gfx950 transpose read: within 16 lanes, lane t supplies the
// address of [pixel t>>2][4 channels t&3] and receives channel t of the 4 pixels) delivers exactly the
// v_mfma_f32_32x32x16_bf16 operand layout with no register shuffles.
//
// Work split: M' rows of dW are 16-channel groups ordered [cx16][tap]; a workgroup (4 waves x MTW row tiles of 32) owns a
// chunk of channel groups x ALL taps x 32 output channels and accumulates in AGPRs (MTW = 16 -> the full 256) over its
// share of the pixel tiles; the result is added to dW with fp32 atomics.  LDS pixel strides are 64 (mod 256) bytes so
// that the 8 row segments of a 32-lane transpose read fall into distinct bank groups.
#include "conv_common.h"
#include <stdlib.h>

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4v;
#ifdef SAVP_CONV_ABLATE
__device__ unsigned long long g_wgp_t[8];         // developer build: cycles of workgroup 0 / wave 0 per phase (savp_debug_wgp_times)
extern "C" int savp_debug_wgp_times(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgp_t), sizeof(g_wgp_t)) == hipSuccess ? 0 : -1;
}
#define WT_DECL unsigned long long wt_[6] = {0, 0, 0, 0, 0, 0}, wt_last = __builtin_readcyclecounter();
#define WT(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); wt_[i] += n_ - wt_last; wt_last = n_; } while (0)
#define WT_FLUSH(ntiles) do { if (blockIdx.x == 0 && threadIdx.x == 0) { for (int i_ = 0; i_ < 6; ++i_) g_wgp_t[i_] = wt_[i_]; g_wgp_t[6] = (ntiles); } } while (0)
#else
#define WT_DECL
#define WT(i) do {} while (0)
#define WT_FLUSH(n) do {} while (0)
#endif
#define LDS_AS __attribute__((address_space(3)))

// LDS-DMA staging (round 5, both operands bf16): lane l of a wave instruction copies 16 bytes from its own global address to LDS byte
// address lds_dst + 16 l (M0 written in the statement that reads it, as conv_ring.hip's ring_dma16); halo / padding slots read 16 zero bytes
__device__ __attribute__((aligned(16))) unsigned g_wgp_zero[4] = {0u, 0u, 0u, 0u};
__device__ __forceinline__ void wgp_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct WgP {
    const float* x; const float* y; float* dw; float* db;
    long long x_sn, y_sn;
    int x_sh, x_sw, y_sh, y_sw;
    int H, W, Ho, Wo, Cx, Cy, ph, pw, kw, sh, sw;
    int D, Do, kd, pd, khw, sd;        // depth (3-D convs): input / output planes, depth taps, pad, kh*kw, depth stride (the launcher admits 1: the
                                       // 4x4x4 stride-2 layers of the video discriminator need more patch than the prefetch registers / 8 DMA
                                       // instructions per wave hold, so nothing exercises 2)
    long long x_sd, y_sd;
    int taps, G16, CG, S, NB, MC;      // taps, 16-channel groups of Cx, groups per chunk, pixel splits, column blocks, channel chunks
    int PH, PW, CP, pitch;             // patch geometry (bf16 elements)
    int tHW, tW, PT;                   // 8x8 tiles per image (count, columns), total tiles
    unsigned long long magC4, magPW, magTaps, magTHW, magTW, magPP, magDo, magKHW;
    unsigned long long magCG, magCGl;  // divide by the channel groups of a full chunk / of the last chunk (row-group order [tap][group])
    // deterministic accumulation (round 6): pixel split sp leaves its dW tiles in part + sp * part_sz (plain stores; the launcher adds the slices
    // to dW in split order, wgrad_fold) and wave w its bias-gradient share in bpart + (sp * NW + w) * Cy.  part == nullptr (a C-ABI caller that
    // passed no scratch): the splits meet in dW / db with fp32 atomics, in arrival order.
    float* part; long long part_sz; float* bpart;
};

// issue order of the MFMA block: in front of MFMA J go the two transpose reads of MFMA J + PD (four when it opens a k-step: + B)
template <int MTW, int PD, int J>
__device__ __forceinline__ void wgp_sched() {
    if constexpr (J < 4 * MTW) {
        if constexpr (J + PD < 4 * MTW) __builtin_amdgcn_sched_group_barrier(0x100, ((J + PD) % MTW == 0) ? 4 : 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        wgp_sched<MTW, PD, J + 1>();
    }
}

// NW waves x MTW row tiles (32 rows of dW each) per workgroup.  NPF = float4 prefetch registers per thread for the patch.
// X16 / Y16: the x / dy tensor holds bf16 (strides in bf16 elements): 8-byte loads, no conversion when parked in LDS.
// DMA (both bf16, no bias gradient, channel counts and strides multiples of 8): the patch and the dy tile of the NEXT pixel tile are
// filled by LDS-DMA (NPF = wave instructions per wave for the patch) while the current one multiplies -- no prefetch registers, no
// conversion, no ds_write, ~6 VALU instructions per 1 KB.  Cycle stamps of the register path at the step's operands (928 images, every
// ConvLSTM layer alike, profiles/r05_wgrad_stamps.log): per pixel tile 940 cycles waiting for the loads + ds_write, 1 340 at the barrier,
// 1 890 ISSUING the next tile's 9 loads (~35 VALU instructions each, two waves per SIMD), 1 590 in the 32 MFMAs -- 5 760 for 2 048
// cycles of matrix work.
template <int NW, int MTW, int NPF, bool X16 = false, bool Y16 = false, bool DMA = false>
__global__ __launch_bounds__(64 * NW, (NW == 4 && MTW == 4) ? 2 : 1) void wgrad_patch_kernel(WgP q) {      // 4 x 4: two workgroups per SIMD set (256 registers)
    static_assert(!DMA || (X16 && Y16), "LDS-DMA staging copies bf16 operands as they are");
    constexpr int NT = 64 * NW;
    constexpr int XES = X16 ? 2 : 4;                           // bytes per x element
    constexpr int NPD = (64 * 8 + NT - 1) / NT;                // float4 prefetch registers for the dy tile (64 px x 32 ch)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware order: the NB column blocks (and MC channel chunks) of one pixel split are consecutive LOGICAL ids, i.e.
    // they run at the same time on CUs of ONE XCD and read the same x patches / dy tiles through one L2.  With the split index
    // fastest (the first version) the four column blocks of a tile ran hundreds of workgroups apart: every x patch came from HBM
    // four times (2.9 GB of traffic for the 0.76 GB 32x32 ConvLSTM problem).
    const int logical = xcd_logical((int)blockIdx.x, q.S * q.NB * q.MC);
    const int nb = logical % q.NB, mc = (logical / q.NB) % q.MC, sp = logical / (q.NB * q.MC);
    const int ca = mc * q.CG;                                  // first 16-channel group of this chunk
    const int cgc = min(q.CG, q.G16 - ca);                     // groups in this chunk
    const int ngroups = cgc * q.taps;                          // (tap, cx16) row groups of this workgroup
    // Row groups are ordered [tap][channel group]: the two 16-row groups of an MFMA row tile (lanes 0-15 / 16-31 of a transpose read) are
    // then two channel groups of ONE tap, 32 bytes apart in the patch, and with the pixel stride a multiple of 64 bytes the eight 32-byte
    // segments of a 32-lane read cover all 64 banks once.  In the first order ([group][tap]) the two halves were consecutive TAPS, one
    // pixel stride apart: three of their four pixels coincide (broadcast) and the fourth shares its banks with the other half's first --
    // SQ_LDS_BANK_CONFLICT was 42 % of SQ_LDS_IDX_ACTIVE on the ConvLSTM layers (profiles/r05_wgrad_counters.log).  With an odd number of
    // groups per chunk one tile in `cgc` still pairs two taps.
    const unsigned long long mag_cgc = (cgc == q.CG) ? q.magCG : q.magCGl;
    const int cy0 = nb * 32;
    const int pplane = q.PH * q.pitch;                         // one depth plane of the patch
    // the kd input planes under one output plane (DMA: rounded up to whole 1 KB wave instructions, the tail slots receive zeros)
    const int patch_elems = DMA ? ((q.kd * pplane + 511) & ~511) : q.kd * pplane;
    __bf16* patch = reinterpret_cast<__bf16*>(smem);           // [2][kd][PH * pitch]
    constexpr int NBUF = 2;
    __bf16* dyt = patch + NBUF * patch_elems;                  // [NBUF][64 * 32]

    // pixel tiles of this split
    const int per = (q.PT + q.S - 1) / q.S;
    const int t_begin = sp * per, t_end = min(q.PT, t_begin + per);
    if (t_begin >= t_end) return;

    // ---- staging maps, three registers per slot: LDS offset | valid << 31 ; py | px << 8 | plane << 16 ; element offset of the
    // slot relative to the patch origin.  Everything that does not depend on the tile is folded in here once: per tile a slot
    // then costs one mask test and one add (the first version re-derived channel / row / column / plane bounds and the address
    // from the packed coordinates for every slot of every tile: ~35 VALU instructions per load, more than the MFMA work of a tile).
    unsigned pinfo[NPF], pcoord[NPF], poff[NPF];
    // DMA: wave instruction i of this wave fills the 64 consecutive 16-byte slots (i * NW + wave) * 64 ... of the patch [plane][row][pixel]
    // [CP / 8 slots]; q.magC4 / q.magPP divide by CP / 8 and PH * PW * CP / 8 here (launcher).  A slot is 8 channels of one pixel; the
    // pad slots of a pixel, channels beyond Cx and the tail of the last instruction read the 16 zero bytes.  Per slot the lane keeps the
    // 64-bit address of the slot under a patch whose origin is element 0 of the tensor, and the slot's (row | plane, column) as ONE bit
    // each of a 64-bit word: per tile a slot then costs a mask test (and, and, compare), a 64-bit add of the tile's delta and a select.
    const int dma_instr = DMA ? (patch_elems >> 9) : 0;        // wave instructions per patch
    unsigned long long laddr[DMA ? NPF : 1], lbits[DMA ? NPF : 1];
    unsigned long long yaddr = 0ull; unsigned ybits = 0u;      // dy tile: this lane's slot (waves NW-4 ..)
    if constexpr (DMA) {
        const int c8n = q.CP >> 3;
        const int per_plane = q.PH * q.PW * c8n;
        const int ptotal = q.kd * per_plane;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = (i * NW + wave) * 64 + lane;
            const int plane = (int)fastdiv((unsigned)idx, q.magPP);
            const int rem = idx - plane * per_plane;
            const int pix = (int)fastdiv((unsigned)rem, q.magC4);
            const int c8 = rem - pix * c8n;
            const int pyy = (int)fastdiv((unsigned)pix, q.magPW);
            const int pxx = pix - pyy * q.PW;
            const bool ok = idx < ptotal && c8 < 2 * q.CG && ca * 16 + c8 * 8 < q.Cx;
            laddr[i] = (unsigned long long)(uintptr_t)q.x +
                       (unsigned long long)(((long long)plane * q.x_sd + (long long)pyy * q.x_sh + (long long)pxx * q.x_sw + ca * 16 + c8 * 8) * 2);
            // low word: row bit (0 .. 21) | plane bit (22 .. 29) | bit 31 = never valid; high word: column bit
            lbits[i] = ok ? ((1ull << pyy) | (1ull << (22 + plane)) | (1ull << (32 + pxx))) : (1ull << 31);
        }
        const int sl = (wave - (NW - 4)) * 64 + lane;          // (garbage for the waves that do not stage dy: never used there)
        const int px = (sl >> 2) & 63, c = (sl & 3) << 3;
        yaddr = (unsigned long long)(uintptr_t)q.y + (unsigned long long)(((long long)(px >> 3) * q.y_sh + (long long)(px & 7) * q.y_sw + cy0 + c) * 2);
        ybits = (cy0 + c < q.Cy) ? ((1u << (px >> 3)) | (1u << (8 + (px & 7)))) : (1u << 31);
        for (int i = 0; i < NPF; ++i) { pinfo[i] = 0u; pcoord[i] = 0u; poff[i] = 0u; }
    } else {
        const int c4n = q.CG * 4;
        const int per_plane = q.PH * q.PW * c4n;
        const int ptotal = q.kd * per_plane;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int idx = tid + NT * i;
            const int plane = (int)fastdiv((unsigned)idx, q.magPP);
            const int rem = idx - plane * per_plane;
            const int pix = (int)fastdiv((unsigned)rem, q.magC4);
            const int c4 = rem - pix * c4n;
            const int pyy = (int)fastdiv((unsigned)pix, q.magPW);
            const int pxx = pix - pyy * q.PW;
            const bool ok = idx < ptotal;
            const bool chan_ok = ca * 16 + c4 * 4 < q.Cx;         // channels beyond Cx: staged as zeros (slot stays valid)
            pinfo[i] = ok ? ((unsigned)(plane * pplane + pyy * q.pitch + pxx * q.CP + c4 * 4) | (1u << 31)) : 0u;
            pcoord[i] = (unsigned)pyy | ((unsigned)pxx << 8) | ((unsigned)plane << 16) | ((ok && chan_ok) ? (1u << 31) : 0u);
            // byte offset of the slot from the patch origin; < 2^31 (checked by the launcher)
            poff[i] = (unsigned)(((long long)plane * q.x_sd + pyy * q.x_sh + pxx * q.x_sw + c4 * 4) * XES);
        }
    }
    // ---- DMA: per-tile scalars from a table in LDS ----------------------------------------------------------------------------------
    // Address deltas and validity masks of a pixel tile come from a table that the workgroup's threads fill for TCH tiles at a time, one
    // tile per thread: decoding a tile index costs ~150 scalar instructions, and with every wave doing that for every tile the scalar
    // stream WAS the fetch phase (stamps: 1 430 cycles per tile to issue 4 - 5 DMA instructions per wave; a wave now reads 32 bytes).
    constexpr int TCH = 256;
    constexpr int YES = Y16 ? 2 : 4;                           // bytes per dy element
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(uintptr_t)smem);
    uint4* dtab = reinterpret_cast<uint4*>(reinterpret_cast<char*>(smem) + (size_t)(NBUF * patch_elems + NBUF * 64 * 32) * 2);     // [TCH][2]
    auto range_mask = [](int lo, int hi) -> unsigned {               // bits lo .. hi-1, clamped to [0, 32)
        lo = max(lo, 0); hi = min(hi, 32);
        if (hi <= lo) return 0u;
        return (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
    };
    // a0 / a1: byte deltas of the tile's patch origin / dy tile origin; m0: patch rows | planes << 22, m1: patch columns, m2: dy rows | dy
    // columns << 8.  (The register-staged path keeps its scalar decode: with the table it was SLOWER -- fp32 operands, 700 -> 770 us on the
    // 16x16 ConvLSTM layer: its fetch phase is bound by the 9 x 16-byte loads per lane, and the table read sits in front of them.)
    struct TileD { unsigned long long a0, a1; unsigned m0, m1, m2, m3; };
    auto fill_desc = [&](int t0) {
        const int t = t0 + tid;
        if (tid < TCH && t < t_end) {
            const int gi = (int)fastdiv((unsigned)t, q.magTHW);          // (sample, output plane)
            const int r = t - gi * q.tHW;
            const int img = (int)fastdiv((unsigned)gi, q.magDo);
            const int dout = gi - img * q.Do;
            const int ty = (int)fastdiv((unsigned)r, q.magTW);
            const int oy0 = ty * 8, ox0 = (r - ty * q.tW) * 8;
            const int iy0 = oy0 * q.sh - q.ph, ix0 = ox0 * q.sw - q.pw, dz0 = dout * q.sd - q.pd;     // patch origin in the input
            const unsigned rowmask = range_mask(-iy0, q.H - iy0), colmask = range_mask(-ix0, q.W - ix0), plmask = range_mask(-dz0, q.D - dz0);
            const unsigned ym = (range_mask(0, q.Ho - oy0) & 0xffu) | ((range_mask(0, q.Wo - ox0) & 0xffu) << 8);
            const long long yo = (long long)img * q.y_sn + (long long)dout * q.y_sd + (long long)oy0 * q.y_sh + (long long)ox0 * q.y_sw;
            const unsigned long long a0 = (unsigned long long)(((long long)img * q.x_sn + (long long)dz0 * q.x_sd + (long long)iy0 * q.x_sh + (long long)ix0 * q.x_sw) * 2);
            const unsigned long long a1 = (unsigned long long)(yo * 2);
            const unsigned m0 = (rowmask & 0x3fffffu) | ((plmask & 0xffu) << 22), m1 = colmask, m2 = ym, m3 = 0u;
            dtab[2 * tid] = make_uint4((unsigned)a0, (unsigned)(a0 >> 32), (unsigned)a1, (unsigned)(a1 >> 32));
            dtab[2 * tid + 1] = make_uint4(m0, m1, m2, m3);
        }
    };
    auto read_desc = [&](int k) -> TileD {                     // wave-uniform address: one broadcast read, then to scalar registers
        const uint4 a = dtab[2 * k], b = dtab[2 * k + 1];
        auto sc = [](unsigned v) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
        TileD d;
        d.a0 = (unsigned long long)sc(a.x) | ((unsigned long long)sc(a.y) << 32);
        d.a1 = (unsigned long long)sc(a.z) | ((unsigned long long)sc(a.w) << 32);
        d.m0 = sc(b.x); d.m1 = sc(b.y); d.m2 = sc(b.z); d.m3 = sc(b.w);
        return d;
    };
    float4 pf[NPF], pd[NPD];
    // bias gradient = column sums of dy, taken from the tiles as they stream by (fp32, before the bf16 rounding); one M chunk only
    const bool do_db = (q.db != nullptr) && (mc == 0);
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned fmask = 0u, dmask = 0u;                           // validity of the elements held in pf / pd
    auto fetch = [&](int t) {
        fmask = 0u; dmask = 0u;
        const int gi = (int)fastdiv((unsigned)t, q.magTHW);          // (sample, output plane)
        const int r = t - gi * q.tHW;
        const int img = (int)fastdiv((unsigned)gi, q.magDo);
        const int dout = gi - img * q.Do;
        const int ty = (int)fastdiv((unsigned)r, q.magTW);
        const int oy0 = ty * 8, ox0 = (r - ty * q.tW) * 8;
        const int iy0 = oy0 * q.sh - q.ph, ix0 = ox0 * q.sw - q.pw;       // patch origin in the input plane
        const int dz0 = dout * q.sd - q.pd;                                // input plane under patch plane 0
        // wave-uniform validity masks of the patch rows / columns / planes of this tile (PH, PW <= 22, kd <= 8)
        auto range_mask = [](int lo, int hi) -> unsigned {               // bits lo .. hi-1, clamped to [0, 32)
            lo = max(lo, 0); hi = min(hi, 32);
            if (hi <= lo) return 0u;
            return (hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        };
        const unsigned rowmask = range_mask(-iy0, q.H - iy0), colmask = range_mask(-ix0, q.W - ix0), plmask = range_mask(-dz0, q.D - dz0);
        // The patch origin may lie outside the tensor (border tiles).  Loads are addressed from the first VALID element of the
        // patch (wave-uniform base, inside the tensor) plus a non-negative 32-bit byte offset per slot; masked slots read offset 0.
        const int lo_y = max(0, -iy0), lo_x = max(0, -ix0), lo_z = max(0, -dz0);
        const unsigned adj = (unsigned)(((long long)lo_z * q.x_sd + lo_y * q.x_sh + lo_x * q.x_sw) * XES);
        const char* __restrict__ base = reinterpret_cast<const char*>(q.x) +
            ((long long)img * q.x_sn + (long long)(dz0 + lo_z) * q.x_sd + (long long)(iy0 + lo_y) * q.x_sh +
             (long long)(ix0 + lo_x) * q.x_sw + ca * 16) * XES;
        const bool flat = q.kd == 1;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            unsigned co = pcoord[i], po = poff[i];
            asm volatile("" : "+v"(co), "+v"(po));             // keep the unpacking inside the loop (register pressure)
            const unsigned pyy = co & 255u, pxx = (co >> 8) & 255u, pl = (co >> 16) & 127u;
            unsigned bit = (rowmask >> pyy) & (colmask >> pxx);
            if (!flat) bit &= plmask >> pl;
            const bool ok = (co >> 31) && (bit & 1u);
            // unconditional load from a clamped address, zeroed at stage() time through the mask: a branch around the load (or
            // a select right behind it) makes hipcc wait for every element here instead of behind the MFMAs of the current tile
            const unsigned offb = ok ? po - adj : 0u;           // valid slots lie at or behind the first valid element
            if constexpr (X16) {                                // 4 bf16 = 8 bytes, kept as raw bits in .x / .y
                const uint2 u = *reinterpret_cast<const uint2*>(base + offb);
                pf[i] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
            } else {
                pf[i] = *reinterpret_cast<const float4*>(base + offb);
            }
            fmask |= (ok ? 1u : 0u) << i;
        }
        const long long yoff = (long long)img * q.y_sn + (long long)dout * q.y_sd + (long long)oy0 * q.y_sh + (long long)ox0 * q.y_sw + cy0;
#pragma unroll
        for (int i = 0; i < NPD; ++i) {
            const int idx = tid + NT * i;                      // 64 pixels x 8 float4
            const int px = idx >> 3, c = (idx & 7) << 2;
            const bool ok = idx < 512 && oy0 + (px >> 3) < q.Ho && ox0 + (px & 7) < q.Wo && cy0 + c < q.Cy;
            if constexpr (Y16) {
                const unsigned short* y16 = reinterpret_cast<const unsigned short*>(q.y);
                const uint2 u = *reinterpret_cast<const uint2*>(ok ? y16 + yoff + (px >> 3) * q.y_sh + (px & 7) * q.y_sw + c : y16);
                pd[i] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
            } else {
                pd[i] = ldg4(ok ? q.y + yoff + (px >> 3) * q.y_sh + (px & 7) * q.y_sw + c : q.y);
            }
            dmask |= (ok ? 1u : 0u) << i;
        }
    };
    auto fetch_dma = [&](const TileD& d, int buf) {
        const unsigned long long zero = (unsigned long long)(uintptr_t)g_wgp_zero;
        const unsigned pbuf = lds0 + (unsigned)(buf * patch_elems * 2);
        const unsigned long long dm = (unsigned long long)d.m0 | ((unsigned long long)d.m1 << 32);
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            if (i * NW + wave < dma_instr) {                    // wave-uniform
                const bool ok = (dm & lbits[i]) == lbits[i];
                const unsigned long long g = ok ? laddr[i] + d.a0 : zero;
                wgp_dma16(reinterpret_cast<const void*>((uintptr_t)g), pbuf + (unsigned)((i * NW + wave) * 1024));
            }
        }
        // dy tile: 64 pixels x 4 slots = 4 wave instructions, taken by the LAST four waves (the first ones carry the patch's remainder)
        if (wave >= NW - 4) {
            const bool ok = (d.m2 & ybits) == ybits;
            const unsigned long long g = ok ? yaddr + d.a1 : zero;
            wgp_dma16(reinterpret_cast<const void*>((uintptr_t)g), lds0 + (unsigned)((NBUF * patch_elems + buf * 64 * 32) * 2 + (wave - (NW - 4)) * 1024));
        }
    };
    auto stage = [&](int buf) {
        __bf16* pa = patch + buf * patch_elems;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            unsigned inf = pinfo[i];
            asm volatile("" : "+v"(inf));
            if (inf >> 31) {
                float4 v = pf[i];
                if (!((fmask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);      // all-zero bits are zero in both formats
                if constexpr (X16) {
                    *reinterpret_cast<uint2*>(pa + (inf & 0x7fffffffu)) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
                } else {
                    bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                    *reinterpret_cast<bf16x4*>(pa + (inf & 0x7fffffffu)) = o;
                }
            }
        }
        __bf16* pb = dyt + buf * 64 * 32;
#pragma unroll
        for (int i = 0; i < NPD; ++i) {
            const int idx = tid + NT * i;
            if (idx < 512) {
                float4 v = pd[i];
                if (!((dmask >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (Y16) {
                    const unsigned u0 = __float_as_uint(v.x), u1 = __float_as_uint(v.y);
                    if (do_db) {
                        bsum.x += __uint_as_float(u0 << 16); bsum.y += __uint_as_float(u0 & 0xffff0000u);
                        bsum.z += __uint_as_float(u1 << 16); bsum.w += __uint_as_float(u1 & 0xffff0000u);
                    }
                    *reinterpret_cast<uint2*>(pb + (idx >> 3) * 32 + ((idx & 7) << 2)) = make_uint2(u0, u1);
                } else {
                    if (do_db) { bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w; }
                    bf16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                    *reinterpret_cast<bf16x4*>(pb + (idx >> 3) * 32 + ((idx & 7) << 2)) = o;
                }
            }
        }
    };

    // ---- per-lane operand addressing ----------------------------------------------------------------------------
    // k-step s covers tile pixels 16 s .. 16 s + 15; this lane's transpose read j (0/1) touches pixel
    //   k = 8 (lane>>5) + 4 j + ((lane&15)>>2)  ->  tile row 2 s + (lane>>5), tile column 4 j + ((lane&15)>>2)
    const int h = lane >> 5, g = (lane >> 4) & 1, r4 = (lane & 15) >> 2, c4 = (lane & 3) << 2;
    const int a_lane = h * q.sh * q.pitch + r4 * q.sw * q.CP + c4;   // elements (output pixel -> input pixel: x stride)
    const int b_lane = (8 * h + r4) * 32 + 16 * g + c4;
    int a_tile[MTW];                                           // tap shift + local channel offset of this lane's row group
    const int nt = min(MTW, max(0, (ngroups + 1) / 2 - wave * MTW));   // valid row tiles of this wave
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int lg = 2 * (wave * MTW + i) + g;               // local row group
        const int lgc = min(lg, ngroups - 1);
        const int tap = (int)fastdiv((unsigned)lgc, mag_cgc);
        const int cl = lgc - tap * cgc;
        const int jd = (int)fastdiv((unsigned)tap, q.magKHW);
        const int t2 = tap - jd * q.khw;
        const int u = t2 / q.kw, v = t2 - u * q.kw;
        a_tile[i] = (a_lane + jd * pplane + u * q.pitch + v * q.CP + cl * 16) * 2;   // bytes
    }

    f32x16 acc[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    WT_DECL
    int tc0 = t_begin;                                         // DMA: first tile of the descriptor table's current chunk
    // (Round 5, measured and NOT kept, profiles/r05_ab_calls.md: the DMA pieces spread between the MFMAs -- the MFMA block then takes 3 000
    // instead of 1 900 + 650 cycles; requests two tiles ahead into three buffers with the two waves of a SIMD requesting at opposite ends of
    // an iteration, with 3 / 4 / 6 MFMAs of fragment read-ahead -- a wave's 32 MFMAs take ~1 950 cycles whether or not its SIMD partner
    // multiplies at the same time, so de-phasing buys nothing and the third buffer costs 3 - 7 %.)
    if constexpr (DMA) {
        fill_desc(tc0);
        __syncthreads();
        fetch_dma(read_desc(0), 0);
    } else fetch(t_begin);
    WT(0);
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        TileD nd = {0ull, 0ull, 0u, 0u, 0u, 0u};
        const bool more = t + 1 < t_end, refill = more && (t + 1 - tc0 == TCH);
        if constexpr (DMA) {
            if (more && !refill) nd = read_desc(t + 1 - tc0);  // lands while this wave waits for its DMAs and at the barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else stage(buf);
        WT(1);
        __syncthreads();
        WT(2);
        auto mma_block = [&]() {
            const LDS_AS char* pa = (const LDS_AS char*)(patch + buf * patch_elems);
            const LDS_AS char* pb = (const LDS_AS char*)(dyt + buf * 64 * 32) + b_lane * 2;
            // 4 k-steps x MTW row tiles = one straight-line block of 4 MTW MFMAs with the transpose reads of MFMA j + PD issued in
            // front of MFMA j (fragment ring of PD + 1 register sets).  The first version looped with a wave-uniform `break` on the
            // valid tile count: hipcc then emitted {2 reads, lgkmcnt(0), MFMA} per tile into ONE register pair -- every MFMA waited for
            // a full LDS round trip (13.4k of 18.8k cycles per tile on the 32x32 ConvLSTM layer, MFMA pipe 15 % busy).  Row tiles past
            // the wave's valid count recompute the last valid group (a_tile is clamped) and are dropped in the epilogue.
            if (nt > 0) {
                constexpr int TOT = 4 * MTW, PD = (MTW >= 4 ? 3 : 2);
                const int rstep = 4 * q.sh * q.pitch, cstep = 8 * q.sw * q.CP;      // bytes: two output rows / four output pixels
                bf16x8 af[PD + 1], bfr[2];
                auto load_a = [&](int j) {
                    const int ks = j / MTW, i = j % MTW;
                    const LDS_AS char* p0 = pa + ks * rstep + a_tile[i];
                    const bf16x4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4v*)p0);
                    const bf16x4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4v*)(p0 + cstep));
                    af[j % (PD + 1)] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                };
                auto load_b = [&](int ks) {
                    const bf16x4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4v*)(pb + ks * 1024));
                    const bf16x4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS bf16x4v*)(pb + ks * 1024 + 256));
                    bfr[ks & 1] = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                };
                load_b(0);
#pragma unroll
                for (int j = 0; j < PD; ++j) load_a(j);
#pragma unroll
                for (int j = 0; j < TOT; ++j) {
                    if (j + PD < TOT) {
                        if ((j + PD) % MTW == 0) load_b((j + PD) / MTW);
                        load_a(j + PD);
                    }
                    acc[j % MTW] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[j % (PD + 1)], bfr[(j / MTW) & 1], acc[j % MTW], 0, 0, 0);
                }
                // pin the issue order: the reads of MFMA j + PD go out in front of MFMA j (left alone, hipcc folds the ring back to
                // a distance of one MFMA to save registers)
                __builtin_amdgcn_sched_group_barrier(0x100, 2 + 2 * PD, 0);
                wgp_sched<MTW, PD, 0>();
            }
        };
        if constexpr (DMA) {
            if (refill) {                                      // every wave read its last descriptor of the old chunk before the barrier above
                tc0 = t + 1;
                fill_desc(tc0);
                __syncthreads();
                nd = read_desc(0);
            }
            if (more) fetch_dma(nd, buf ^ 1);
        } else if (more) fetch(t + 1);
        WT(3);
        mma_block();
        WT(4);
    }
    WT_FLUSH(t_end - t_begin);

    if (do_db) {                                               // every slot of this thread is channel quad (tid & 7)
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {
            bsum.x += __shfl_xor(bsum.x, m); bsum.y += __shfl_xor(bsum.y, m);
            bsum.z += __shfl_xor(bsum.z, m); bsum.w += __shfl_xor(bsum.w, m);
        }
        const int c = cy0 + (lane << 2);
        if (lane < 8) {
            if (q.bpart) {                                     // this wave's own row of the split's bias slice
                float* __restrict__ bp = q.bpart + ((long long)sp * NW + wave) * q.Cy;
                if (c < q.Cy) bp[c] = bsum.x;
                if (c + 1 < q.Cy) bp[c + 1] = bsum.y;
                if (c + 2 < q.Cy) bp[c + 2] = bsum.z;
                if (c + 3 < q.Cy) bp[c + 3] = bsum.w;
            } else {
                if (c < q.Cy) unsafeAtomicAdd(q.db + c, bsum.x);
                if (c + 1 < q.Cy) unsafeAtomicAdd(q.db + c + 1, bsum.y);
                if (c + 2 < q.Cy) unsafeAtomicAdd(q.db + c + 2, bsum.z);
                if (c + 3 < q.Cy) unsafeAtomicAdd(q.db + c + 3, bsum.w);
            }
        }
    }
    // ---- epilogue: acc[i][r] of lane (l31 = column cy, khalf) is row (r&3) + 8 (r>>2) + 4 khalf of row tile i ----------
    const int l31 = lane & 31, khalf = lane >> 5;
    const int cy = cy0 + l31;
    if (cy >= q.Cy) return;
    const bool sliced = q.part != nullptr;
    float* __restrict__ dW = sliced ? q.part + (long long)sp * q.part_sz : q.dw;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        if (i >= nt) break;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {                       // the two 16-row groups of the tile
            const int lg = 2 * (wave * MTW + i) + hh;
            if (lg >= ngroups) continue;
            const int tap = (int)fastdiv((unsigned)lg, mag_cgc);
            const int cl = lg - tap * cgc;
            float* __restrict__ base = dW + ((long long)tap * q.Cx + (ca + cl) * 16) * q.Cy + cy;
#pragma unroll
            for (int r = 8 * hh; r < 8 * hh + 8; ++r) {
                const int m = (r & 3) + 8 * ((r >> 2) & 1) + 4 * khalf;   // row within the 16-row group
                if ((ca + cl) * 16 + m < q.Cx) {
                    if (sliced) base[(long long)m * q.Cy] = acc[i][r];
                    else unsafeAtomicAdd(base + (long long)m * q.Cy, acc[i][r]);
                }
            }
        }
    }
}

template <int NW, int MTW, int NPF, bool X16 = false, bool Y16 = false, bool DMA = false>
static hipError_t launch_wgp(const WgP& q, dim3 grid, size_t lds, hipStream_t st) {
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        hipFuncSetAttribute((const void*)wgrad_patch_kernel<NW, MTW, NPF, X16, Y16, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL((wgrad_patch_kernel<NW, MTW, NPF, X16, Y16, DMA>), grid, dim3(64 * NW), lds, st, q);
    return hipGetLastError();
}

// operand dtypes picked at run time: both fp32, both bf16, or dy alone in bf16
template <int NW, int MTW, int NPF>
static hipError_t launch_wgp_dt(const WgP& q, dim3 grid, size_t lds, hipStream_t st, bool x16, bool y16) {
    if (x16 && y16) return launch_wgp<NW, MTW, NPF, true, true>(q, grid, lds, st);
    if (y16) return launch_wgp<NW, MTW, NPF, false, true>(q, grid, lds, st);
    return launch_wgp<NW, MTW, NPF, false, false>(q, grid, lds, st);
}

// Returns true when the call was handled (2-D / 3-D, strides <= 2 in the plane and 1 in depth, bf16 precision, channel counts % 4, <= 8 taps per side,
// >= 64 output pixels per plane, a patch that fits the prefetch registers).
bool conv_wgrad_patch_try(ConvP& p, const SavpConvArgs* a, hipStream_t st, int* rc, long long* plan_bytes) {
    const bool xs4 = (a->x_sn % 4 == 0) && (a->x_sh % 4 == 0) && (a->x_sw % 4 == 0) && aligned16(a->x);
    const bool ys4 = (a->y_sn % 4 == 0) && (a->y_sh % 4 == 0) && (a->y_sw % 4 == 0) && aligned16(a->y);
    if (!(p.bf16 && a->sd == 1 && a->x_sd % 4 == 0 && a->y_sd % 4 == 0 && a->sh <= 2 && a->sw <= 2 && a->Cx % 4 == 0 &&
          a->Cy % 4 == 0 && xs4 && ys4 && a->kh <= 8 && a->kw <= 8 && a->Ho * a->Wo >= 64))
        return false;
    if (a->x_sh * (long long)(a->H + 8) >= (1ll << 31) || a->y_sh * (long long)(a->Ho + 8) >= (1ll << 31)) return false;
    // byte offsets inside one patch (kd planes x PH rows) are 32-bit
    if (((long long)a->kd * a->x_sd + (long long)(7 * a->sh + a->kh + 1) * a->x_sh) * 4 >= (1ll << 31)) return false;
    WgP q;
    q.x = (const float*)a->x; q.y = (const float*)a->y; q.dw = (float*)a->w; q.db = (float*)a->bias;
    q.x_sn = a->x_sn; q.y_sn = a->y_sn;
    q.x_sh = (int)a->x_sh; q.x_sw = (int)a->x_sw; q.y_sh = (int)a->y_sh; q.y_sw = (int)a->y_sw;
    q.H = a->H; q.W = a->W; q.Ho = a->Ho; q.Wo = a->Wo; q.Cx = a->Cx; q.Cy = a->Cy; q.ph = a->ph; q.pw = a->pw; q.kw = a->kw; q.sh = a->sh; q.sw = a->sw;
    q.D = a->D; q.Do = a->Do; q.kd = a->kd; q.pd = a->pd; q.sd = a->sd; q.khw = a->kh * a->kw; q.x_sd = a->x_sd; q.y_sd = a->y_sd;
    q.taps = a->kd * a->kh * a->kw;
    q.G16 = (a->Cx + 15) / 16;
    q.PH = 7 * a->sh + a->kh; q.PW = 7 * a->sw + a->kw;           // input rows / columns under an 8x8 output tile
    // workgroup shape
    const int groups_all = q.taps * q.G16;
    int nw, mtw;
    // 8 waves; the fewest row tiles per wave that still cover every (tap, channel group) row block in one chunk (measured on
    // the 3x3 heads: 8x2 305 us, 8x4 424 us, 4x4 476 us, 4x8 831 us -- fewer tiles per wave = better balance and occupancy)
    nw = 8;
    mtw = groups_all <= 32 ? 2 : (groups_all <= 64 ? 4 : 8);
    // few row groups (3x3 heads: 18): these problems are bound by streaming x / dy, and 4-wave workgroups (three per CU, each at its
    // own phase of fetch / stage / multiply) overlap that better than one barrier-coupled 8-wave group: 490 -> 379 us on 64x64x32->32
    if (groups_all <= 32) { nw = 4; mtw = 4; }
    {   // developer override: SAVP_WGP_CFG=<nw><mtw> (e.g. 84)
        const int ov = savp_opt(OPT_WGP_CFG);
        if (ov) { nw = ov / 10; mtw = ov % 10; if (2 * nw * mtw < groups_all && 2 * nw * mtw < q.taps) return false; }
    }
    const int nthreads = 64 * nw;
    const int npf_max = (nw == 8) ? 8 : 16;
    const int cg_pf = (npf_max * nthreads) / (4 * a->kd * q.PH * q.PW);  // prefetch-register bound on the channel groups
    int cg = (2 * nw * mtw) / q.taps;                            // row groups per workgroup / taps
    if (cg > q.G16) cg = q.G16;
    if (cg > cg_pf) cg = cg_pf;
    if (cg < 1) return false;
    q.CG = cg;
    const int MC = (q.G16 + cg - 1) / cg, NB = (a->Cy + 31) / 32;
    // pixel stride: CP / 16 = 2 or 6 (mod 8) -> the 8 row segments of a 32-lane transpose read hit distinct bank groups
    int cpu = cg;
    while ((cpu % 8) != 2 && (cpu % 8) != 6) ++cpu;
    q.CP = cpu * 16;
    q.pitch = q.PW * q.CP;
    if (q.PH > 31 || q.PW > 31 || a->kd > 31) return false;      // per-tile validity masks are 32 bits; packed coordinates 8 / 8 / 7 bits
    const int tH = (a->Ho + 7) / 8;
    q.tW = (a->Wo + 7) / 8; q.tHW = tH * q.tW;
    q.PT = a->N * a->Do * q.tHW;
    // Total workgroups: ONE round with eight waves per CU -- 256 eight-wave or 512 four-wave workgroups.  Measured INSIDE the train
    // step (round 3, `wgp_split` sweep 128 ... 1536 with per-kernel traces): every 8-wave instantiation is fastest at 256 (e.g. the
    // ConvLSTM gate layers 648 -> 580 us, the 3x3 heads 395 -> 214 us against the former 768), the 4-wave ones at 512 (622 -> 575 us;
    // 831 us at 256), anything that is not a whole round pays a tail (384 / 640: slower than either neighbour).  In the step the
    // activations of all 928 images are HBM-cold and a workgroup's fixed costs (prologue, the atomic dW epilogue of its split) count;
    // back to back on warm caches round 2 had measured the opposite (768 best), which is how 768 got here.
    long long s = (256LL * (8 / nw)) / ((long long)NB * MC);
    if (s < 1) s = 1;
    {
        const int ovs = savp_opt(OPT_WGP_SPLIT);      // developer override: total workgroups
        if (ovs > 0) s = ovs / ((long long)NB * MC) > 0 ? ovs / ((long long)NB * MC) : 1;
    }
    if (s > q.PT) s = q.PT;
    {   // no empty split: with per = ceil(PT / S) tiles each, ceil(PT / per) splits have work -- every slice below is written in full
        const long long per = (q.PT + s - 1) / s;
        s = (q.PT + per - 1) / per;
    }
    q.S = (int)s;
    q.magC4 = magic40(cg * 4); q.magPW = magic40(q.PW); q.magTaps = magic40(q.taps);
    q.magCG = magic40(cg); q.magCGl = magic40(q.G16 - (MC - 1) * cg);
    q.magTHW = magic40(q.tHW); q.magTW = magic40(q.tW);
    q.magPP = magic40(q.PH * q.PW * cg * 4); q.magDo = magic40(a->Do); q.magKHW = magic40(q.khw);
    if ((double)q.PT * q.tHW >= 1099511627776.0) return false;
    size_t lds = (size_t)2 * a->kd * q.PH * q.pitch * 2 + (size_t)2 * 64 * 32 * 2;
    if (lds > 160 * 1024) return false;
    // deterministic accumulation: S slices of dW (+ S x NW rows of the bias gradient) in the caller's scratch; too little scratch for two
    // slices (or none): atomics.  The bound handed to savp_conv_workspace_bytes is this very computation.
    const long long nW = (long long)q.taps * a->Cx * a->Cy;
    const long long nB = a->bias ? (long long)nw * a->Cy : 0;
    if (plan_bytes) {
        *plan_bytes = q.S > 1 || nB ? (long long)q.S * (nW + nB) * (long long)sizeof(float) : 0;
        return true;
    }
    q.part = nullptr; q.part_sz = nW; q.bpart = nullptr;
    if ((q.S > 1 || nB) && a->ws && !(((uintptr_t)a->ws) & 15)) {
        const long long fit = a->ws_bytes / ((nW + nB) * (long long)sizeof(float));
        if (fit >= q.S) {
            q.part = (float*)a->ws;
            if (nB) q.bpart = q.part + (long long)q.S * nW;
        }
    }
    // LDS-DMA staging: both operands bf16, no bias gradient, every slot of 8 channels whole and 16-byte aligned in both tensors
    const long long dma_slots = (((long long)a->kd * q.PH * q.pitch + 511) & ~511LL) / 8;
    const size_t lds_dma = (size_t)dma_slots * 8 * 2 * 2 + (size_t)2 * 64 * 32 * 2 + (size_t)256 * 32;     // two buffers + the tile descriptor table
    const bool s8 = a->x_sn % 8 == 0 && a->x_sd % 8 == 0 && a->x_sh % 8 == 0 && a->x_sw % 8 == 0 && a->y_sn % 8 == 0 && a->y_sd % 8 == 0 &&
                    a->y_sh % 8 == 0 && a->y_sw % 8 == 0 && a->Cx % 8 == 0 && a->Cy % 8 == 0;
    const bool dma = savp_opt(OPT_WGP_DMA) && a->src_bf16 && a->out_bf16 && !a->bias && s8 && dma_slots <= 8LL * 64 * nw && lds_dma <= 160 * 1024 && q.PH <= 22 && q.PW <= 32 && a->kd <= 8;
    if (dma) {
        lds = lds_dma;
        q.magC4 = magic40(q.CP / 8); q.magPP = magic40(q.PH * q.PW * (q.CP / 8));
    }
    const int npf = (a->kd * q.PH * q.PW * cg * 4 + nthreads - 1) / nthreads;
    q.NB = NB; q.MC = MC;
    dim3 grid((unsigned)(q.S * NB * MC), 1u, 1u);
    hipError_t err;
    const bool x16 = a->src_bf16 != 0, y16 = a->out_bf16 != 0;     // WGRAD: src = x, "out" = the dy operand
    // bf16 operand tensors (round 3: the ConvLSTM gate convolutions; round 4: every generator convolution whose input / output
    // gradient is stored in bf16): both operands bf16, or dy alone (layer 0: the input image stays fp32), for every workgroup shape;
    // x alone only for the 8 x 8 shape.  A combination that is not instantiated is refused, never read as fp32.
    if (x16 && !y16 && !(nw == 8 && mtw == 8)) { *rc = SAVP_EINVAL; return true; }
    if (dma) {
        if (nw == 8 && mtw == 8) err = launch_wgp<8, 8, 8, true, true, true>(q, grid, lds, st);
        else if (nw == 8 && mtw == 4) err = launch_wgp<8, 4, 8, true, true, true>(q, grid, lds, st);
        else if (nw == 8) err = launch_wgp<8, 2, 8, true, true, true>(q, grid, lds, st);
        else if (mtw == 8) err = launch_wgp<4, 8, 8, true, true, true>(q, grid, lds, st);
        else err = launch_wgp<4, 4, 8, true, true, true>(q, grid, lds, st);
        if (err == hipSuccess && q.part) { wgrad_fold(q.dw, q.part, q.S, nW, st); err = hipGetLastError(); }
        *rc = (err == hipSuccess) ? SAVP_OK : SAVP_ELAUNCH;
        return true;
    }
    if (nw == 8 && mtw == 4) err = (npf <= 4) ? launch_wgp_dt<8, 4, 4>(q, grid, lds, st, x16, y16) : launch_wgp_dt<8, 4, 8>(q, grid, lds, st, x16, y16);
    else if (nw == 8 && mtw == 2) err = (npf <= 4) ? launch_wgp_dt<8, 2, 4>(q, grid, lds, st, x16, y16) : launch_wgp_dt<8, 2, 8>(q, grid, lds, st, x16, y16);
    else if (nw == 8) {
        if (x16 && !y16) err = (npf <= 4) ? launch_wgp<8, 8, 4, true, false>(q, grid, lds, st) : launch_wgp<8, 8, 8, true, false>(q, grid, lds, st);
        else err = (npf <= 4) ? launch_wgp_dt<8, 8, 4>(q, grid, lds, st, x16, y16) : launch_wgp_dt<8, 8, 8>(q, grid, lds, st, x16, y16);
    }
    else if (mtw == 8) err = (npf <= 8) ? launch_wgp_dt<4, 8, 8>(q, grid, lds, st, x16, y16) : launch_wgp_dt<4, 8, 16>(q, grid, lds, st, x16, y16);
    else err = (npf <= 8) ? launch_wgp_dt<4, 4, 8>(q, grid, lds, st, x16, y16) : launch_wgp_dt<4, 4, 16>(q, grid, lds, st, x16, y16);
    if (err == hipSuccess && q.part) {
        wgrad_fold(q.dw, q.part, q.S, nW, st);
        if (q.bpart) wgrad_fold(q.db, q.bpart, q.S * nw, a->Cy, st);
        err = hipGetLastError();
    }
    *rc = (err == hipSuccess) ? SAVP_OK : SAVP_ELAUNCH;
    return true;
}
