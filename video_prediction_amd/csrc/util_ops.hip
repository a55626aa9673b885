// util_ops.hip -- HBM-bound glue kernels of the SAVP path (all fp32, channels-last views).
//
//   tile_channels : tile_concat's broadcast of a [R,c] vector over the pixels of a concat-buffer slice
//                   (ops.py:968-1006, savp_model.py:456-459,467-469,492-494,503-505); also the backward of the
//                   global average pool.
//   colsum        : sum over pixels (and optionally rows): bias gradients, gradient of tile_concat, global
//                   average pool (networks.py:30).
//   select        : scheduled-sampling tf.where(ground_truth[t], images, gen_image) (savp_model.py:406) and its
//                   gradient routing.
//   gather_clips  : the discriminator's random clip gather tf.gather_nd (savp_model.py:97-102) and its adjoint.
//   axpby / fill  : flat elementwise helpers (z concatenation savp_model.py:725, gradient accumulation).
//   adam          : tf.train.AdamOptimizer update on a flat parameter arena (base_model.py:486-487).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "savp_hip.h"
#include "zero_fill.h"
#include "opts.h"

#define NT 256

static inline unsigned nblocks(long long n, int per = NT) {
    long long b = (n + per - 1) / per;
    if (b < 1) b = 1;
    return (unsigned)b;
}
#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? SAVP_OK : SAVP_ELAUNCH)

// ---------------------------------------------------------------------------------------------------------------
__global__ void tile_channels_kernel(const float* __restrict__ z, long long R, int HW, int C, float scale, float* out,
                                     long long sn, long long sp, int beta) {
    long long total = R * HW * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long long rp = i / C;
        int p = (int)(rp % HW);
        long long r = rp / HW;
        float v = z[r * C + c] * scale;
        float* q = out + r * sn + (long long)p * sp + c;
        *q = beta ? *q + v : v;
    }
}

// bf16 destination (a slice of a buffer that only feeds convolutions of the bf16 datapath): overwrite only
__global__ void tile_channels_bf16_kernel(const float* __restrict__ z, long long R, int HW, int C, float scale, unsigned short* out,
                                          long long sn, long long sp) {
    long long total = R * HW * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long long rp = i / C;
        int p = (int)(rp % HW);
        long long r = rp / HW;
        const __bf16 v = (__bf16)(z[r * C + c] * scale);
        out[r * sn + (long long)p * sp + c] = __builtin_bit_cast(unsigned short, v);
    }
}

extern "C" int savp_tile_channels_bf16(void* stream, const float* z, int64_t R, int32_t HW, int32_t C, float scale, SavpView out) {
    if (!z || !out.p || R < 1 || HW < 1 || C < 1) return SAVP_EINVAL;
    long long total = (long long)R * HW * C;
    unsigned nb = nblocks(total);
    if (nb > 65535u * 8) nb = 65535u * 8;
    hipLaunchKernelGGL(tile_channels_bf16_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, z, (long long)R, HW, C, scale,
                       (unsigned short*)out.p, (long long)out.sn, (long long)out.sp);
    return LAUNCH_OK();
}

extern "C" int savp_tile_channels(void* stream, const float* z, int64_t R, int32_t HW, int32_t C, float scale, SavpView out,
                                  int32_t beta) {
    if (!z || !out.p || R < 1 || HW < 1 || C < 1) return SAVP_EINVAL;
    long long total = (long long)R * HW * C;
    unsigned nb = nblocks(total);
    if (nb > 65535u * 8) nb = 65535u * 8;
    hipLaunchKernelGGL(tile_channels_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, z, (long long)R, HW, C, scale,
                       (float*)out.p, (long long)out.sn, (long long)out.sp, beta);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// colsum: in view [R, HW, C]; per_row: out[r*C+c] (=|+=) scale * sum_p ; else out[c] += scale * sum_{r,p} (atomic).
// grid = (R, pixel chunks, channel blocks of 64)
__global__ __launch_bounds__(NT) void colsum_kernel(const float* __restrict__ in, long long sn, long long sp, int HW, int C,
                                                    float scale, float* out, int per_row, int chunk, float* part) {
    __shared__ float sh[NT];
    const long long r = blockIdx.x;
    const int cb = blockIdx.z * 64;
    const int c = cb + (threadIdx.x & 63);
    const int pl = threadIdx.x >> 6;                 // 4 pixel lanes
    const int p0 = blockIdx.y * chunk, p1 = min(HW, p0 + chunk);
    float s = 0.f;
    if (c < C)
        for (int p = p0 + pl; p < p1; p += 4) s += in[r * sn + (long long)p * sp + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64 && c < C) {
        float t = (sh[threadIdx.x] + sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192]) * scale;
        if (part) part[r * C + c] = t;                // deterministic all-row sum: this row's partial, added up in row order by colsum_reduce_kernel
        else if (per_row) unsafeAtomicAdd(out + r * C + c, t);
        else unsafeAtomicAdd(out + c, t);
    }
}

// Narrow channel slices (C <= 16: the nz tiled-z channels inside the 16..144-channel gradient rows, d(tile_concat)): the kernel above
// maps its 64 lanes to channels, so a C = 8 slice keeps 8 of 64 lanes busy with 4-byte loads (51 us per call, 16 calls per step).
// Here a pixel is covered by C / V lanes with V-wide loads (V = 4 / 2 / 1 by alignment), a workgroup sums NT * V / C pixels per pass
// and reduces over its pixel lanes in LDS.
template <int V>
__global__ __launch_bounds__(NT) void colsum_narrow_kernel(const float* __restrict__ in, long long sn, long long sp, int HW, int C,
                                                           float scale, float* out, int per_row, int chunk) {
    __shared__ float sh[NT * V];
    const int q = C / V;                              // lanes per pixel (power of two)
    const int cv = threadIdx.x & (q - 1), pl = threadIdx.x / q, npl = NT / q;
    const long long r = blockIdx.x;
    const int p0 = blockIdx.y * chunk, p1 = min(HW, p0 + chunk);
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    const float* base = in + r * sn + cv * V;
#pragma unroll 4
    for (int p = p0 + pl; p < p1; p += npl) {
        const float* a = base + (long long)p * sp;
        if constexpr (V == 4) { const float4 t = *reinterpret_cast<const float4*>(a); acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w; }
        else if constexpr (V == 2) { const float2 t = *reinterpret_cast<const float2*>(a); acc[0] += t.x; acc[1] += t.y; }
        else acc[0] += a[0];
    }
#pragma unroll
    for (int v = 0; v < V; ++v) sh[threadIdx.x * V + v] = acc[v];
    __syncthreads();
    for (int off = npl >> 1; off > 0; off >>= 1) {
        if (pl < off) {
#pragma unroll
            for (int v = 0; v < V; ++v) sh[threadIdx.x * V + v] += sh[(threadIdx.x + off * q) * V + v];
        }
        __syncthreads();
    }
    if (pl == 0) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float t = sh[threadIdx.x * V + v] * scale;
            const int c = cv * V + v;
            if (per_row) unsafeAtomicAdd(out + r * C + c, t);
            else unsafeAtomicAdd(out + c, t);
        }
    }
}

// Sum over ALL pixels of a pixel-linear tensor [P = R * HW][C] (bias gradients of the 'up' layers and of the discriminators):
// the kernels above end every workgroup in C atomics onto the same 1..8 cache lines, and with thousands of workgroups those
// serialise (64x64x32 over 928 images: 14.8 k workgroups, 270 us, atomic-bound).  Here ~1 k workgroups each sum a contiguous
// range of pixels with C / 4 lanes per pixel and float4 loads (4 in flight per thread), leave ONE partial row in a workspace, and
// a second small launch adds the rows up (16 atomics per output element instead of thousands).
__global__ __launch_bounds__(NT) void colsum_part_kernel(const float* __restrict__ in, long long sp, long long P, int C, long long chunk,
                                                         float* __restrict__ part) {
    __shared__ float sh[NT * 4];
    const int q = C >> 2;                             // lanes per pixel (power of two, <= NT)
    const int cv = threadIdx.x & (q - 1), pl = threadIdx.x / q, npl = NT / q;
    const long long p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* base = in + cv * 4;
#pragma unroll 4
    for (long long p = p0 + pl; p < p1; p += npl) {
        const float4 t = *reinterpret_cast<const float4*>(base + p * sp);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    *reinterpret_cast<float4*>(sh + threadIdx.x * 4) = acc;
    __syncthreads();
    for (int off = npl >> 1; off > 0; off >>= 1) {
        if (pl < off) {
#pragma unroll
            for (int v = 0; v < 4; ++v) sh[threadIdx.x * 4 + v] += sh[(threadIdx.x + off * q) * 4 + v];
        }
        __syncthreads();
    }
    if (pl == 0) *reinterpret_cast<float4*>(part + (long long)blockIdx.x * C + cv * 4) = *reinterpret_cast<const float4*>(sh + threadIdx.x * 4);
}

// out[c] += scale * sum over rows of part[row][c], in a fixed order (round 6): a block owns 32 columns; thread (g = tid >> 5, tid & 31) sums
// rows g, g + 8, ... with eight loads in flight, the eight row groups are added in group order and ONE thread writes the column -- no
// atomics, the same bits whatever order the partial rows were produced in.
__global__ __launch_bounds__(NT) void colsum_reduce_kernel(const float* __restrict__ part, int rows, int C, float scale, float* out) {
    __shared__ float sh[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
    float s = 0.f;
    if (c < C) {
        int r = g;
        for (; r + 56 < rows; r += 64) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(long long)(r + 8 * j) * C + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; r < rows; r += 8) s += part[(long long)r * C + c];
    }
    sh[g][threadIdx.x & 31] = s;
    __syncthreads();
    if (g == 0 && c < C) {
        float t = sh[0][threadIdx.x];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += sh[j][threadIdx.x];
        out[c] += t * scale;
    }
}

extern "C" int savp_colsum(void* stream, SavpView in, int64_t R, int32_t HW, int32_t C, float scale, float* out,
                           int32_t per_row, float* ws, int64_t ws_floats) {
    // NOTE: always accumulates (atomically) into `out`; zero it first for an overwrite.
    if (!in.p || !out || R < 1 || HW < 1 || C < 1) return SAVP_EINVAL;
    // Narrow power-of-two slices: C / V lanes per pixel with V-wide loads.  (Tried for the discriminators' 32..256-channel bias
    // gradients too, with ~1024 workgroups: 0.8 ms per step SLOWER than the kernel below -- every workgroup ends in C atomics onto
    // the same one to eight cache lines, and those serialise.)
    const uintptr_t al = (uintptr_t)in.p | (uintptr_t)(in.sn * 4) | (uintptr_t)(in.sp * 4);
    // (C = 32 per row: KTH's 32 tiled-z channels -- eleven launches per step that the 64-lanes-to-channels kernel below ran at half occupancy with
    //  scalar loads, one pixel per trip: 1.06 ms per step at 1.4 TB/s)
    if ((C <= 16 || (C == 32 && per_row)) && (C & (C - 1)) == 0 && HW >= 64) {
        const int V = (C >= 4 && (al & 15) == 0) ? 4 : ((C >= 2 && (al & 7) == 0) ? 2 : 1);
        const int per_pass = NT * V / C;
        const long long per_row_wgs = per_row ? 1 : 4;      // per_row: ONE workgroup per row -- a single writer per output, nothing to order
        int chunk = (int)(((HW + per_row_wgs - 1) / per_row_wgs + per_pass - 1) / per_pass * per_pass);   // whole passes
        if (chunk < per_pass) chunk = per_pass;
        dim3 grid((unsigned)R, (unsigned)((HW + chunk - 1) / chunk), 1u);
        hipStream_t st = (hipStream_t)stream;
        if (V == 4) hipLaunchKernelGGL(colsum_narrow_kernel<4>, grid, dim3(NT), 0, st, (const float*)in.p, (long long)in.sn, (long long)in.sp, HW, C, scale, out, per_row, chunk);
        else if (V == 2) hipLaunchKernelGGL(colsum_narrow_kernel<2>, grid, dim3(NT), 0, st, (const float*)in.p, (long long)in.sn, (long long)in.sp, HW, C, scale, out, per_row, chunk);
        else hipLaunchKernelGGL(colsum_narrow_kernel<1>, grid, dim3(NT), 0, st, (const float*)in.p, (long long)in.sn, (long long)in.sp, HW, C, scale, out, per_row, chunk);
        return LAUNCH_OK();
    }
    // all-pixel sums of large pixel-linear tensors: partial rows in caller-owned scratch + reduce launch (see colsum_part_kernel);
    // needs SAVP_COLSUM_WS_FLOATS floats of `ws` (written before read), without them the atomic kernel below runs
    const int two_stage = savp_opt(OPT_COLSUM_2STAGE) && ws && ws_floats >= SAVP_COLSUM_WS_FLOATS && (((uintptr_t)ws) & 15) == 0;
    const long long P = (long long)R * HW;
    const int q4 = C >> 2;
    if (two_stage && !per_row && (C & 3) == 0 && q4 >= 1 && q4 <= NT && (q4 & (q4 - 1)) == 0 && (al & 15) == 0 && in.sn == (long long)HW * in.sp &&
        P >= 32768) {
        const int npl = NT / q4;
        const int nwg = 1024;
        long long chunk2 = (P + nwg - 1) / nwg;
        chunk2 = (chunk2 + 4 * npl - 1) / (4 * npl) * (4 * npl);                       // whole unrolled passes
        const int rows = (int)((P + chunk2 - 1) / chunk2);
        static_assert(SAVP_COLSUM_WS_FLOATS >= 1024 * 4 * NT, "workspace covers nwg rows of up to 4*NT channels");
        hipStream_t st = (hipStream_t)stream;
        hipLaunchKernelGGL(colsum_part_kernel, dim3((unsigned)rows), dim3(NT), 0, st, (const float*)in.p, (long long)in.sp, P, C, chunk2, ws);
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((C + 31) / 32)), dim3(NT), 0, st, (const float*)ws, rows, C, scale, out);
        return LAUNCH_OK();
    }
    // all-row sums of everything else (dense-layer bias gradients: HW = 1, C = 8 / 100 / ...): with caller scratch, one partial row per input row
    // (one workgroup per (row, 64 channels): plain stores) + the fixed-order reduce -- no atomics, the same bits every run; without scratch the
    // workgroups add to `out` atomically, in arrival order
    const bool rows_det = !per_row && ws && (((uintptr_t)ws) & 15) == 0 && (long long)R * C <= ws_floats && R <= 65536;
    int chunk = (per_row || rows_det) ? HW : 256;          // per_row: one workgroup per (row, 64 channels) -- a single writer per output
    dim3 grid((unsigned)R, (unsigned)((HW + chunk - 1) / chunk), (unsigned)((C + 63) / 64));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(NT), 0, (hipStream_t)stream, (const float*)in.p, (long long)in.sn,
                       (long long)in.sp, HW, C, scale, out, per_row, chunk, rows_det ? ws : (float*)nullptr);
    if (rows_det)
        hipLaunchKernelGGL(colsum_reduce_kernel, dim3((unsigned)((C + 31) / 32)), dim3(NT), 0, (hipStream_t)stream, (const float*)ws, (int)R, C, 1.f, out);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
struct SelP {
    int N, HW, C;
    const int* mask;                         // [N] nonzero -> take a
    const float* a; long long a_sn, a_sp;
    const float* b; long long b_sn, b_sp;    // may be null (treated as zeros)
    int nout; float* out[4]; long long o_sn[4], o_sp[4];
};

__global__ void select_kernel(SelP p) {
    long long total = (long long)p.N * p.HW * p.C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % p.C);
        long long np = i / p.C;
        int px = (int)(np % p.HW);
        int n = (int)(np / p.HW);
        float v;
        if (p.mask[n]) v = p.a[n * p.a_sn + px * p.a_sp + c];
        else v = p.b ? p.b[n * p.b_sn + px * p.b_sp + c] : 0.f;
        for (int k = 0; k < p.nout; ++k) p.out[k][n * p.o_sn[k] + px * p.o_sp[k] + c] = v;
    }
}

// One thread per pixel, C compile-time (1 / 3 colour channels): the element-per-thread kernel above pays two runtime integer
// divisions per float and writes 4-byte pieces; here a pixel's channels are one 12-byte run and the sample index comes from the
// grid (blockIdx.y), 350 -> ~100 us for the 29-step first-frame fill.
template <int C>
__global__ __launch_bounds__(NT) void select_px_kernel(SelP p) {
    const int n = blockIdx.y;
    const bool take_a = p.mask[n] != 0;
    const float* __restrict__ src = take_a ? p.a + (long long)n * p.a_sn : (p.b ? p.b + (long long)n * p.b_sn : nullptr);
    const long long s_sp = take_a ? p.a_sp : p.b_sp;
    for (int px = blockIdx.x * NT + threadIdx.x; px < p.HW; px += gridDim.x * NT) {
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = src ? src[(long long)px * s_sp + c] : 0.f;
        for (int k = 0; k < p.nout; ++k) {
            float* __restrict__ o = p.out[k] + (long long)n * p.o_sn[k] + (long long)px * p.o_sp[k];
#pragma unroll
            for (int c = 0; c < C; ++c) o[c] = v[c];
        }
    }
}

extern "C" int savp_select(void* stream, int32_t N, int32_t HW, int32_t C, const int32_t* mask, SavpView a, SavpView b,
                           int32_t nout, const SavpView* outs) {
    if (!mask || !a.p || nout < 1 || nout > 4 || !outs) return SAVP_EINVAL;
    SelP p;
    p.N = N; p.HW = HW; p.C = C; p.mask = mask;
    p.a = (const float*)a.p; p.a_sn = a.sn; p.a_sp = a.sp;
    p.b = (const float*)b.p; p.b_sn = b.sn; p.b_sp = b.sp;
    p.nout = nout;
    for (int i = 0; i < nout; ++i) { p.out[i] = (float*)outs[i].p; p.o_sn[i] = outs[i].sn; p.o_sp[i] = outs[i].sp; }
    if ((C == 1 || C == 3) && N <= 65535) {
        unsigned bx = (unsigned)((HW + NT - 1) / NT);
        if (bx > 1024) bx = 1024;
        if (C == 3) hipLaunchKernelGGL(select_px_kernel<3>, dim3(bx, (unsigned)N), dim3(NT), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(select_px_kernel<1>, dim3(bx, (unsigned)N), dim3(NT), 0, (hipStream_t)stream, p);
        return LAUNCH_OK();
    }
    hipLaunchKernelGGL(select_kernel, dim3(nblocks((long long)N * HW * C)), dim3(NT), 0, (hipStream_t)stream, p);
    return LAUNCH_OK();
}

// db[n] += (mask[n] ? 0 : sum_k din_k[n])   (gradient of the not-ground-truth branch)
struct SelBP {
    int N, HW, C;
    const int* mask;
    int nin; const float* din[4]; long long i_sn[4], i_sp[4];
    float* db; long long b_sn, b_sp;
};

__global__ void select_bwd_kernel(SelBP p) {
    long long total = (long long)p.N * p.HW * p.C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % p.C);
        long long np = i / p.C;
        int px = (int)(np % p.HW);
        int n = (int)(np / p.HW);
        if (p.mask[n]) continue;
        float v = 0.f;
        for (int k = 0; k < p.nin; ++k) v += p.din[k][n * p.i_sn[k] + px * p.i_sp[k] + c];
        p.db[n * p.b_sn + px * p.b_sp + c] += v;
    }
}

extern "C" int savp_select_bwd(void* stream, int32_t N, int32_t HW, int32_t C, const int32_t* mask, int32_t nin,
                               const SavpView* dins, SavpView db) {
    if (!mask || !db.p || nin < 1 || nin > 4 || !dins) return SAVP_EINVAL;
    SelBP p;
    p.N = N; p.HW = HW; p.C = C; p.mask = mask; p.nin = nin;
    for (int i = 0; i < nin; ++i) { p.din[i] = (const float*)dins[i].p; p.i_sn[i] = dins[i].sn; p.i_sp[i] = dins[i].sp; }
    p.db = (float*)db.p; p.b_sn = db.sn; p.b_sp = db.sp;
    hipLaunchKernelGGL(select_bwd_kernel, dim3(nblocks((long long)N * HW * C)), dim3(NT), 0, (hipStream_t)stream, p);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// gather_clips: src time-major [L, B, E] (E = H*W*C contiguous), dst batch-major [B, clip, E]:
//   dst[b,i,:] = src[(t_start[b]+i)*src_ts + b*E + :]   ; adjoint: src[...] += dst[b,i,:]
//   (src_ts = element stride between timesteps, so one half of a [T,2B,...] buffer can be addressed)
__global__ void gather_clips_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ t_start,
                                    int B, int clip, long long E, long long src_ts, int adjoint, float* srcw, const float* dstr) {
    long long total = (long long)B * clip * (E / 4);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long e4 = i % (E / 4);
        long long bi = i / (E / 4);
        int ci = (int)(bi % clip);
        int b = (int)(bi / clip);
        long long so = (long long)(t_start[b] + ci) * src_ts + (long long)b * E + e4 * 4;
        long long d_o = ((long long)b * clip + ci) * E + e4 * 4;
        if (!adjoint) {
            *reinterpret_cast<float4*>(dst + d_o) = *reinterpret_cast<const float4*>(src + so);
        } else {
            float4 g = *reinterpret_cast<const float4*>(dstr + d_o);
            float4 o = *reinterpret_cast<float4*>(srcw + so);
            o.x += g.x; o.y += g.y; o.z += g.z; o.w += g.w;
            *reinterpret_cast<float4*>(srcw + so) = o;
        }
    }
}

extern "C" int savp_gather_clips(void* stream, float* src, float* dst, const int32_t* t_start, int32_t B, int32_t clip,
                                 int64_t E, int64_t src_ts, int32_t adjoint) {
    if (!src || !dst || !t_start || E % 4 || src_ts % 4) return SAVP_EINVAL;
    long long total = (long long)B * clip * (E / 4);
    unsigned nb = nblocks(total);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(gather_clips_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, (const float*)src, dst, t_start, B,
                       clip, (long long)E, (long long)src_ts, adjoint, src, (const float*)dst);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void axpby_kernel(long long n, float a, const float* __restrict__ x, float b, const float* y, float* out) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = a * x[i];
        if (y) v += b * y[i];
        out[i] = v;
    }
}

extern "C" int savp_axpby(void* stream, int64_t n, float a, const float* x, float b, const float* y, float* out) {
    if (!x || !out || n < 0) return SAVP_EINVAL;
    if (n == 0) return SAVP_OK;
    unsigned nb = nblocks(n);
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(axpby_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, (long long)n, a, x, b, y, out);
    return LAUNCH_OK();
}

__global__ void fill_view_kernel(float* out, long long sn, long long sp, long long R, int HW, int C, float value) {
    long long total = R * HW * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long long rp = i / C;
        int p = (int)(rp % HW);
        long long r = rp / HW;
        out[r * sn + (long long)p * sp + c] = value;
    }
}

extern "C" int savp_fill_view(void* stream, SavpView out, int64_t R, int32_t HW, int32_t C, float value) {
    if (!out.p) return SAVP_EINVAL;
    unsigned nb = nblocks((long long)R * HW * C);
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(fill_view_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, (float*)out.p, (long long)out.sn,
                       (long long)out.sp, (long long)R, HW, C, value);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// TF Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed on the host and passed in;
//   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2 ; p -= lr_t * m / (sqrt(v) + eps)       (epsilon outside the sqrt)
// gscale multiplies the gradient first (1/world_size after a sum all-reduce).
__global__ void adam_kernel(long long n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float lr_host, float b1, float b2, float eps, float gscale, const float* lr_dev) {
    const float lr_t = lr_dev ? *lr_dev : lr_host;
    long long n4 = n / 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
#define ADAM1(f)                                       \
    {                                                  \
        float gr = gg.f * gscale;                      \
        mm.f = b1 * mm.f + (1.f - b1) * gr;            \
        vv.f = b2 * vv.f + (1.f - b2) * gr * gr;       \
        pp.f -= lr_t * mm.f / (sqrtf(vv.f) + eps);     \
    }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (long long i = n4 * 4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gr = g[i] * gscale;
        float mi = b1 * m[i] + (1.f - b1) * gr;
        float vi = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mi; v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

extern "C" int savp_adam(void* stream, int64_t n, float* p, const float* g, float* m, float* v, float lr_t, float beta1,
                         float beta2, float eps, float gscale, const float* lr_t_dev) {
    if (!p || !g || !m || !v || n < 1) return SAVP_EINVAL;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return SAVP_EINVAL;
    unsigned nb = nblocks(n / 4 + 1);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, (long long)n, p, g, m, v, lr_t, beta1, beta2,
                       eps, gscale, lr_t_dev);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// dpre = dy * y * (1 - y) : backward of the sigmoid fused into the scratch-image conv epilogue (savp_model.py:572)
__global__ void sigmoid_bwd_kernel(const float* dy, long long dy_sn, long long dy_sp, const float* y, long long y_sn, long long y_sp,
                                   float* out, long long N, int HW, int C) {
    long long total = N * HW * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long long np = i / C;
        int px = (int)(np % HW);
        long long n = np / HW;
        float yy = y[n * y_sn + px * y_sp + c];
        out[i] = dy[n * dy_sn + px * dy_sp + c] * yy * (1.f - yy);
    }
}

extern "C" int savp_sigmoid_bwd(void* stream, SavpView dy, SavpView y, float* out, int64_t N, int32_t HW, int32_t C) {
    if (!dy.p || !y.p || !out) return SAVP_EINVAL;
    unsigned nb = nblocks((long long)N * HW * C);
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, (const float*)dy.p, (long long)dy.sn,
                       (long long)dy.sp, (const float*)y.p, (long long)y.sn, (long long)y.sp, out, (long long)N, HW, C);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// dense layer with few rows (M <= 64): out[m,c] = scale * sum_k x[m,k] W[k,c] + bias[c]    (ops.dense, ops.py:5-16)
// split over K across workgroups (the implicit-GEMM kernel would put the whole K loop into a single workgroup).
// Used for the CDNA kernel head (8192 -> 100) and the discriminators' final linear (65536 -> 1).
// one workgroup = one K-chunk of 64 rows of W: x-chunk [M][64] and W-chunk [64][C] are staged in LDS, thread (c, g)
// accumulates the outputs of column c for the samples m = g, g+G, ... ; partial sums are atomically added to out.
#define DKC 64
__global__ __launch_bounds__(NT) void dense_smallm_kernel(const float* __restrict__ x, long long xs, int M, long long Kd, int C,
                                                          const float* __restrict__ W, const float* bias, const float* scale,
                                                          float* out, int nsub) {
    extern __shared__ float dsm[];
    float* xsh = dsm;                 // [M][DKC]
    float* wsh = dsm + M * DKC;       // [DKC][C]
    float* osh = wsh + DKC * C;       // [M][C] partial outputs of this workgroup (nsub K-chunks): one atomic per output and
                                      // workgroup -- with one chunk per workgroup the 128-way atomic contention on the
                                      // 3200 outputs of the CDNA head was the whole cost of the layer
    for (int i = threadIdx.x; i < M * C; i += NT) osh[i] = 0.f;
    // thread -> (column c, sample group g); CT columns per pass
    const int CT = C >= NT ? NT : C;
    const int G = NT / CT;                     // sample groups
    const int c0 = threadIdx.x % CT, g = threadIdx.x / CT;
    for (int sub = 0; sub < nsub; ++sub) {
        const long long k0 = ((long long)blockIdx.x * nsub + sub) * DKC;
        if (k0 >= Kd) break;                   // uniform
        const int kn = (int)min((long long)DKC, Kd - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < M * DKC; i += NT) {
            int m = i / DKC, k = i % DKC;
            xsh[i] = k < kn ? x[(long long)m * xs + k0 + k] : 0.f;
        }
        for (int i = threadIdx.x; i < DKC * C; i += NT) {
            int k = i / C;
            wsh[i] = k < kn ? W[(k0 + k) * C + (i % C)] : 0.f;
        }
        __syncthreads();
        if (g < G) {
            for (int c = c0; c < C; c += CT) {
                for (int mb = g; mb < M; mb += G * 8) {
                    float acc[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
                    for (int k = 0; k < DKC; ++k) {
                        const float w = wsh[k * C + c];
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int m = mb + i * G;
                            if (m < M) acc[i] += xsh[m * DKC + k] * w;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int m = mb + i * G;
                        if (m < M) osh[m * C + c] += acc[i];          // (m, c) is owned by exactly one thread
                    }
                }
            }
        }
    }
    __syncthreads();
    const float sc = scale ? *scale : 1.f;
    for (int i = threadIdx.x; i < M * C; i += NT) {
        float v = osh[i] * sc;
        if (blockIdx.x == 0 && bias) v += bias[i % C];
        unsafeAtomicAdd(out + i, v);
    }
}

// Two-launch form used when the caller provides a workspace: partial products of S K-slices written with plain stores, then one
// small reduction kernel (scale, bias).  The atomic form above ends every workgroup with M*C device-scope float atomics on the
// same M*C words (~7 per ns chip-wide: 80 % of its 62 us on the CDNA head 8192 -> 100 at M = 32).
// Thread (c4, kq): 4 output columns x all rows of a 32-row block, k rows kq, kq + KQ, ... of the slice; x is staged transposed in
// LDS ([k][m], rows padded to 36 floats) and read as broadcast float4.
#define DP_KT 64
template <bool VEC>          // C % 4 == 0 (compile time: a runtime branch around the row loads serialises them)
__global__ __launch_bounds__(NT) void dense_partial_kernel(const float* __restrict__ x, long long xs, int M, long long Kd, int C,
                                                           const float* __restrict__ W, float* __restrict__ part, long long kslice) {
    __shared__ __attribute__((aligned(16))) float xl[DP_KT * 36];
    const int C4 = (C + 3) >> 2;
    const int KQ = NT / C4;
    const int c4 = threadIdx.x % C4, kq = threadIdx.x / C4;
    const bool active = kq < KQ;
    const long long k_lo = (long long)blockIdx.x * kslice, k_hi = min(Kd, k_lo + kslice);
    for (int m0 = 0; m0 < M; m0 += 32) {
        const int mb = min(32, M - m0);
        float acc[32][4];
#pragma unroll
        for (int m = 0; m < 32; ++m) { acc[m][0] = 0.f; acc[m][1] = 0.f; acc[m][2] = 0.f; acc[m][3] = 0.f; }
        for (long long k0 = k_lo; k0 < k_hi; k0 += DP_KT) {
            const int kn = (int)min((long long)DP_KT, k_hi - k0);
            __syncthreads();
            for (int i = threadIdx.x; i < 32 * DP_KT; i += NT) {
                const int m = i / DP_KT, kk = i - m * DP_KT;
                xl[kk * 36 + m] = (m < mb && kk < kn) ? x[(long long)(m0 + m) * xs + k0 + kk] : 0.f;
            }
            __syncthreads();
            if (active) {
                // all weight rows of this sub-chunk are requested before the first is used (W is streamed from HBM once per call:
                // one load per loop trip would expose a full memory round trip per row)
                constexpr int RMAX = DP_KT / 4;                      // KQ >= 4 (C <= 256)
                float4 wv[RMAX];
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    const int kk = kq + r * KQ;
                    const float* wr = W + (k0 + min(kk, kn - 1)) * C + c4 * 4;
                    if constexpr (VEC) wv[r] = *reinterpret_cast<const float4*>(wr);
                    else {
                        wv[r].x = wr[0];
                        wv[r].y = c4 * 4 + 1 < C ? wr[1] : 0.f; wv[r].z = c4 * 4 + 2 < C ? wr[2] : 0.f; wv[r].w = c4 * 4 + 3 < C ? wr[3] : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    const int kk = kq + r * KQ;
                    if (kk >= kn) break;
                    const float4 w = wv[r];
                    const float4* xr = reinterpret_cast<const float4*>(xl + kk * 36);
#pragma unroll
                    for (int mq = 0; mq < 8; ++mq) {
                        const float4 xv = xr[mq];
                        const float xm[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[mq * 4 + j][0] += xm[j] * w.x; acc[mq * 4 + j][1] += xm[j] * w.y;
                            acc[mq * 4 + j][2] += xm[j] * w.z; acc[mq * 4 + j][3] += xm[j] * w.w;
                        }
                    }
                }
            }
        }
        // reduce the KQ partial sums of this workgroup through LDS (one [32][4 C4] block), then one plain store per output
        __syncthreads();
        float* red = xl;                                   // needs 32 * 4 * C4 floats <= DP_KT * 36 (C <= 64 ... handled in passes)
        const int cols = C4 * 4;
        for (int pass = 0; pass * 8 < 32; ++pass) {        // 8 rows per pass: 8 * cols <= 8 * 256 = 2048 floats < 2304
            for (int i = threadIdx.x; i < 8 * cols; i += NT) red[i] = 0.f;
            __syncthreads();
            if (active) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) unsafeAtomicAdd(&red[r * cols + c4 * 4 + j], acc[pass * 8 + r][j]);   // ds_add_f32
            }
            __syncthreads();
            for (int i = threadIdx.x; i < 8 * cols; i += NT) {
                const int r = i / cols, c = i - r * cols, m = pass * 8 + r;
                if (m < mb && c < C) part[((long long)blockIdx.x * M + m0 + m) * C + c] = red[i];
            }
            __syncthreads();
        }
    }
}

// MFMA form of the K-sliced partial products (the default): exact fp32 on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain per
// output, MI355X guide 3).  A workgroup (4 waves) owns one K slice; its waves take 32-row groups of the slice round-robin.  A wave
// holds its 32 x 32 k block of x in registers (lane (m, half): 16 consecutive k of row m) and reads the matching W rows straight
// from global memory -- no LDS staging, no barrier in the K loop.  Column mapping: lane l of tile t owns output column NTL*l + t,
// so the NTL tiles' B operands of one k row are ONE NTL-wide vector load per lane (dwordx4 for the 100-column CDNA head) and the
// 32 lanes of a half-wave read one contiguous row piece.  EVERY load of a k group is issued before the first MFMA (sched_barrier):
// the kernel's time is a single memory round trip plus 16 NTL MFMAs; with the loads interleaved between dependent MFMAs, as hipcc
// schedules them by default, a group paid 6-7 serial round trips (52 us for the 3.3 MB CDNA head).
// MFMA i of a group contracts k = base + 16*half + i: any bijection of the group's 32 k onto (instruction, half) is a valid order.
// The waves' accumulators meet in LDS in wave order (deterministic), one plain store per output of the slice.
typedef float dm_f32x16 __attribute__((ext_vector_type(16)));
template <int NTL, bool VEC>   // 32-column tiles: NTL * 32 >= C; VEC: W rows allow aligned NTL-wide vector loads (C % NTL == 0)
__global__ __launch_bounds__(256) void dense_mfma_partial_kernel(const float* __restrict__ x, long long xs, int M, long long Kd, int C,
                                                                 const float* __restrict__ W, float* __restrict__ part, long long kslice) {
    __shared__ float red[32 * 32 * NTL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const long long k_lo = (long long)blockIdx.x * kslice, k_hi = min(Kd, k_lo + kslice);
    const bool xvec = ((xs & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
    // first column of this lane's NTL-wide piece; pieces beyond C are clamped onto the last full piece (results never stored)
    const int cbase = min(NTL * l31, (C - 1) / NTL * NTL);
    for (int m0 = 0; m0 < M; m0 += 32) {
        const int mb = min(32, M - m0);
        dm_f32x16 acc[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        const float* xrow = x + (long long)(m0 + min(l31, mb - 1)) * xs;
        for (long long k0 = k_lo + wave * 32; k0 < k_hi; k0 += 4 * 32) {
            const long long kb = k0 + half * 16;
            float a[16];
            float b[16][NTL];
            if (xvec && kb + 16 <= k_hi) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4*>(xrow + kb + 4 * q);
                    a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = (kb + i < k_hi) ? xrow[kb + i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float* wr = W + min(kb + i, Kd - 1) * C + cbase;                       // rows >= k_hi meet a == 0
                if constexpr (VEC && NTL == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(wr);
                    b[i][0] = v.x; b[i][1] = v.y; b[i][2] = v.z; b[i][3] = v.w;
                } else if constexpr (VEC && NTL == 8) {
                    const float4 v = *reinterpret_cast<const float4*>(wr), w = *reinterpret_cast<const float4*>(wr + 4);
                    b[i][0] = v.x; b[i][1] = v.y; b[i][2] = v.z; b[i][3] = v.w; b[i][4] = w.x; b[i][5] = w.y; b[i][6] = w.z; b[i][7] = w.w;
                } else if constexpr (VEC && NTL == 2) {
                    const float2 v = *reinterpret_cast<const float2*>(wr);
                    b[i][0] = v.x; b[i][1] = v.y;
                } else {
#pragma unroll
                    for (int t = 0; t < NTL; ++t) b[i][t] = wr[min(t, C - 1 - cbase)];
                }
            }
            __builtin_amdgcn_sched_barrier(0);          // all loads of the group are in flight before the first MFMA waits
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int t = 0; t < NTL; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i][t], acc[t], 0, 0, 0);
        }
        // the four waves' accumulators meet in LDS in wave order, one wave per round (a wave covers every element of the block exactly
        // once): a fixed summation order -- ds_add_f32 from four racing waves made the sum depend on their arrival order, and this
        // layer's output feeds bf16 roundings downstream
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int t = 0; t < NTL; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;                    // C/D layout of the 32x32 MFMAs
                        float* d = &red[row * (32 * NTL) + NTL * l31 + t];
                        *d = (w == 0 ? 0.f : *d) + acc[t][r];
                    }
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < 32 * 32 * NTL; i += 256) {
            const int row = i / (32 * NTL), c = i - row * (32 * NTL);
            if (row < mb && c < C) part[((long long)blockIdx.x * M + m0 + row) * C + c] = red[i];
        }
        __syncthreads();
    }
}

template <int NTL>
static void launch_dense_mfma(hipStream_t st, unsigned S, const float* x, long long xs, int M, long long K, int C, const float* W, float* ws,
                              long long kslice) {
    const bool vec = (C % NTL == 0) && ((((uintptr_t)W) & 15) == 0) && NTL > 1;
    if (vec) hipLaunchKernelGGL((dense_mfma_partial_kernel<NTL, true>), dim3(S), dim3(256), 0, st, x, xs, M, K, C, W, ws, kslice);
    else hipLaunchKernelGGL((dense_mfma_partial_kernel<NTL, false>), dim3(S), dim3(256), 0, st, x, xs, M, K, C, W, ws, kslice);
}

// 32 outputs x 8 slice groups per workgroup: lane = 8 * output + group; every thread adds S/8 partials, then a shuffle tree
__global__ __launch_bounds__(NT) void dense_reduce_kernel(const float* __restrict__ part, int S, int MC, int C, const float* bias,
                                                          const float* scale, float* __restrict__ out) {
    const int g = threadIdx.x & 7;
    const int i = blockIdx.x * (NT / 8) + (threadIdx.x >> 3);
    float s = 0.f;
    if (i < MC) {
#pragma unroll 8
        for (int k = g; k < S; k += 8) s += part[(long long)k * MC + i];
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (i >= MC || g) return;
    s *= scale ? *scale : 1.f;
    if (bias) s += bias[i % C];
    out[i] = s;
}

// SAVP_DENSE_LEGACY=1: the VALU / LDS-broadcast partial kernel (developer A/B switch)
static bool dense_legacy() {
    return savp_opt(OPT_DENSE_LEGACY) != 0;
}

extern "C" int savp_dense_fwd(void* stream, const float* x, int64_t x_row_stride, int32_t M, int64_t K, int32_t C, const float* W,
                              const float* bias, const float* scale, float* out, float* ws, int64_t ws_floats) {
    if (!x || !W || !out || M < 1 || K < 1 || C < 1) return SAVP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (M > 64 || C > 256) return SAVP_EINVAL;
    if (ws && ws_floats >= (int64_t)M * C) {
        // K slices: multiples of 128 rows (4 waves x 32), as many as the workspace holds, at most 64 (the reduction kernel reads them all)
        long long S = ws_floats / ((long long)M * C);
        if (S > 64) S = 64;
        const bool legacy = dense_legacy();
        const long long unit = legacy ? DP_KT : 128;
        long long kslice = ((K + S - 1) / S + unit - 1) / unit * unit;
        S = (K + kslice - 1) / kslice;
        if (!legacy) {
            if (C <= 32) launch_dense_mfma<1>(st, (unsigned)S, x, (long long)x_row_stride, M, (long long)K, C, W, ws, kslice);
            else if (C <= 64) launch_dense_mfma<2>(st, (unsigned)S, x, (long long)x_row_stride, M, (long long)K, C, W, ws, kslice);
            else if (C <= 128) launch_dense_mfma<4>(st, (unsigned)S, x, (long long)x_row_stride, M, (long long)K, C, W, ws, kslice);
            else launch_dense_mfma<8>(st, (unsigned)S, x, (long long)x_row_stride, M, (long long)K, C, W, ws, kslice);
        } else if ((C & 3) == 0 && ((uintptr_t)W & 15) == 0)
            hipLaunchKernelGGL(dense_partial_kernel<true>, dim3((unsigned)S), dim3(NT), 0, st, x, (long long)x_row_stride, M, (long long)K, C,
                               W, ws, kslice);
        else
            hipLaunchKernelGGL(dense_partial_kernel<false>, dim3((unsigned)S), dim3(NT), 0, st, x, (long long)x_row_stride, M, (long long)K, C,
                               W, ws, kslice);
        hipLaunchKernelGGL(dense_reduce_kernel, dim3((unsigned)((M * C + NT / 8 - 1) / (NT / 8))), dim3(NT), 0, st, ws, (int)S, M * C, C, bias, scale,
                           out);
        return LAUNCH_OK();
    }
    savp_zero_async(out, (size_t)M * C * sizeof(float), st);
    size_t lds = (size_t)(M * DKC + DKC * C + M * C) * sizeof(float);
    const long long chunks = (K + DKC - 1) / DKC;
    int nsub = (int)((chunks + 255) / 256);                    // at most ~256 workgroups (fewer atomics for very long K)
    if (nsub < 1) nsub = 1;
    hipLaunchKernelGGL(dense_smallm_kernel, dim3((unsigned)((chunks + nsub - 1) / nsub)), dim3(NT), lds, st, x, (long long)x_row_stride, M,
                       (long long)K, C, W, bias, scale, out, nsub);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// uint8 frames [B, T, F] (batch-major, as the input pipeline delivers them) -> float32 [T, B, F] * (1/255):
// tf.image.convert_image_dtype(uint8 -> float32) (base_dataset.py:187) fused with transpose_batch_time
// (base_model.py:283 / tf_utils.py:118-122).  4 bytes per thread per step.
// ---------------------------------------------------------------------------------------------------------------
__global__ void u8_frames_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int B, int T, long long F) {
    const int t = blockIdx.y, b = blockIdx.z;
    const uint8_t* src = in + ((long long)b * T + t) * F;
    float* dst = out + ((long long)t * B + b) * F;
    const float scale = (float)(1.0 / 255.0);
    const long long F4 = F / 4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < F4; i += (long long)gridDim.x * blockDim.x) {
        const uchar4 v = reinterpret_cast<const uchar4*>(src)[i];
        reinterpret_cast<float4*>(dst)[i] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    }
    for (long long i = F4 * 4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < F; i += (long long)gridDim.x * blockDim.x)
        dst[i] = src[i] * scale;
}

extern "C" int savp_u8_frames_to_f32(void* stream, const uint8_t* in, float* out, int32_t B, int32_t T, int64_t frame) {
    // frame = H*W*C values per frame; requires frame % 4 == 0 and 4- / 16-byte aligned buffers (true for every dataset shape)
    if (!in || !out || B < 1 || T < 1 || frame < 4 || (frame % 4) != 0) return SAVP_EINVAL;
    if ((((uintptr_t)in) & 3) != 0 || (((uintptr_t)out) & 15) != 0) return SAVP_EINVAL;
    unsigned gx = (unsigned)((frame / 4 + NT - 1) / NT);
    if (gx > 16) gx = 16;
    hipLaunchKernelGGL(u8_frames_kernel, dim3(gx, (unsigned)T, (unsigned)B), dim3(NT), 0, (hipStream_t)stream, in, out, B, T, (long long)frame);
    return LAUNCH_OK();
}

// ---------------------------------------------------------------------------------------------------------------
// Fold float64 accumulators into fp32 gradients (round 6): dst[i] += (float) src[i] ; src[i] = 0, for the elements listed in idx (NULL: the
// first n elements).  The parameter gradients that many workgroups add to (norm gamma / beta, biases, the z-LSTM) are accumulated in a
// float64 twin of the gradient arena -- exact, hence independent of arrival order -- and rounded to fp32 ONCE, here.
__global__ __launch_bounds__(NT) void fold_f64_kernel(const int* __restrict__ idx, long long n, double* __restrict__ src, float* __restrict__ dst) {
    const long long t = (long long)blockIdx.x * NT + threadIdx.x;
    if (t >= n) return;
    const long long i = idx ? (long long)idx[t] : t;
    dst[i] += (float)src[i];
    src[i] = 0.0;
}

extern "C" int savp_fold_f64(void* stream, const int32_t* idx, int64_t n, double* src, float* dst) {
    if (!src || !dst || n < 0) return SAVP_EINVAL;
    if (n == 0) return SAVP_OK;
    hipLaunchKernelGGL(fold_f64_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, (const int*)idx, (long long)n, src, dst);
    return LAUNCH_OK();
}
