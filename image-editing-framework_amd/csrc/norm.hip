// HBM-bound normalisation kernels (fp16 storage, fp32 statistics), 16-byte vector accesses.
//
//   GroupNorm(32) [+ SiLU] over NHWC      ResnetBlock2D.norm1/norm2 + nonlinearity
//                                          (/root/reference/pnp/model/register.py:105-110,149-158)
//   LayerNorm over C                       BasicTransformerBlock.norm1/2/3 [ext diffusers]
//   GEGLU                                  FeedForward.net[0] [ext diffusers]
//
// GroupNorm is two launches: (1) per-(batch, pixel-split) partial sums per group, written to a
// scratch slab (no atomics: deterministic, nothing to zero), (2) normalise + affine + SiLU.
// A thread always owns the same 8-channel chunk, so group membership is resolved once.
#include "ief_common.h"
#include "ief_params.h"

#define GN_MAX_GROUPS 64
#define GN_PART_FLOATS 4096   // LDS partials of the statistics pass: PY * C <= 4096

#include <stdlib.h>
// A/B switches read once from the environment (same-box comparisons of two forms of one kernel; not part of the ABI)
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// pixel splits per image for the statistics pass: enough blocks to cover the chip, at least 16 pixels each
static inline int gn_splits_host(int HW) {
    int s = HW / 16;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}
extern "C" int ief_gn_splits(int HW) { return gn_splits_host(HW); }

__device__ __forceinline__ half8 load_cat8(const half_t* x, const half_t* x2, int C1, int C2, long long pix, int c) {
    // channel chunk c..c+7 of pixel `pix` of the virtual concat [C1 | C2]; C1 % 8 == 0
    if (c < C1) return *(const half8*)(x + pix * C1 + c);
    return *(const half8*)(x2 + pix * C2 + (c - C1));
}

// Thread layout of both kernels: blockDim.x = C8 * PY (C8 = C/8 chunks, PY pixel lanes), so a thread
// keeps ONE 8-channel chunk for its whole life: group ids, affine and statistics are resolved once
// and the inner loop is load - 8 FMAs - store with no integer division.
// grid (splits, B)
__global__ void gn_stats_kernel(const half_t* __restrict__ x, const half_t* __restrict__ x2, int C1, int C2,
                                float* __restrict__ partial, int HW, int groups, int splits, int PY) {
    const int C = C1 + C2, C8 = C >> 3, cpg = C / groups;
    const int b = blockIdx.y, sp = blockIdx.x;
    const int cx = threadIdx.x % C8, py = threadIdx.x / C8;
    // per-(pixel lane, channel) partials go to LDS and each group is then summed by ONE thread in a fixed order:
    // no atomics, so the statistics (and everything downstream) are bit-reproducible run to run
    __shared__ float ps[GN_PART_FLOATS], pq[GN_PART_FLOATS];
    const int per = (HW + splits - 1) / splits;
    const int p0 = sp * per, p1 = min(HW, p0 + per);
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    for (int p = p0 + py; py < PY && p < p1; p += PY) {  // py >= PY: padding threads of a <64-wide layout
        const half8 v = load_cat8(x, x2, C1, C2, (long long)b * HW + p, cx * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[e] += f; q[e] += f * f; }
    }
    if (py < PY) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { ps[py * C + cx * 8 + e] = s[e]; pq[py * C + cx * 8 + e] = q[e]; }
    }
    __syncthreads();
    if (threadIdx.x < groups) {
        const int g = threadIdx.x;
        float ts = 0.f, tq = 0.f;
        for (int r = 0; r < PY; ++r)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) { ts += ps[r * C + c]; tq += pq[r * C + c]; }
        float* o = partial + (((long long)b * splits + sp) * groups + g) * 2;
        o[0] = ts; o[1] = tq;
    }
}

// grid (B * groups); block 64: one wave sums the <= 64 split partials of one (batch, group) in parallel
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats,
                                                         int HW, int groups, int splits, int cpg, float eps) {
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int sp = threadIdx.x;
    float s = 0.f, q = 0.f;
    if (sp < splits) {
        const float* o = partial + (((long long)b * splits + sp) * groups + g) * 2;
        s = o[0]; q = o[1];
    }
    s = wave_sum(s); q = wave_sum(q);
    if (sp == 0) {
        const float inv = 1.0f / ((float)cpg * (float)HW);
        const float m = s * inv;
        const float var = fmaxf(q * inv - m * m, 0.f);
        stats[((long long)b * groups + g) * 2] = m;
        stats[((long long)b * groups + g) * 2 + 1] = rsqrtf(var + eps);
    }
}

// grid (pixel blocks, B)
__global__ void gn_apply_kernel(const half_t* __restrict__ x, const half_t* __restrict__ x2, int C1, int C2,
                                half_t* __restrict__ out, const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ partial, int HW, int groups, int splits, float eps,
                                int apply_silu, int pix_per_block, int PY) {
    const int C = C1 + C2, C8 = C >> 3, cpg = C / groups;
    const int b = blockIdx.y;
    const float* st = partial + (long long)gridDim.y * splits * groups * 2 + (long long)b * groups * 2;  // finalized stats
    const int cx = threadIdx.x % C8, py = threadIdx.x / C8;
    const int c = cx * 8;
    float sc[8], sh[8];  // y = x * sc + sh
    {
        const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
        const f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (c + e) / cpg;
            const float ga = e < 4 ? g0[e] : g1[e - 4], be = e < 4 ? b0[e] : b1[e - 4];
            sc[e] = st[2 * g + 1] * ga;
            sh[e] = be - st[2 * g] * sc[e];
        }
    }
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(HW, p0 + pix_per_block);
    for (int p = p0 + py; py < PY && p < p1; p += PY) {
        const long long pix = (long long)b * HW + p;
        const half8 v = load_cat8(x, x2, C1, C2, pix, c);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (float)v[e] * sc[e] + sh[e];
            if (apply_silu) y = silu_f(y);
            o[e] = (half_t)y;
        }
        *(half8*)(out + pix * C + c) = o;
    }
}

// ---------------------------------------------------------------------------------------------------
// Single-launch GroupNorm: one workgroup per (batch, group).  The group's slab (HW pixels x cpg channels)
// is read twice by the SAME workgroup (second pass from L2), so statistics need no cross-workgroup
// reduction, no scratch and no extra launches.  Accesses are 4-byte (2 channels): a group is cpg*2 bytes
// per pixel (20 .. 160 B), i.e. part of a cache line; groups that share lines are placed on the same XCD
// (block ids congruent mod 8) so one L2 merges them.  Thread (py, j) keeps channel pair j for all its pixels.
__global__ __launch_bounds__(512) void gn_fused_kernel(const half_t* __restrict__ x, const half_t* __restrict__ x2,
                                                       int C1, int C2, half_t* __restrict__ out,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ stats_out,
                                                       int HW, int groups, float eps, int apply_silu, int PY, int KS) {
    const int C = C1 + C2, cpg = C / groups, cp2 = cpg >> 1;
    // KS workgroups share one (batch, group): each streams the whole slab for the statistics (the same sums in the same
    // order: identical mean / rstd in all of them, no hand-off) and normalises every KS-th run of pixels — the second,
    // heavier pass (SiLU + stores) is split KS ways and 128 x KS workgroups fill the chip
    // (the KS workgroups of a (batch, group) are gridDim.x / KS block ids apart: the same XCD, one L2 copy of the slab)
    const int nbg = gridDim.x / KS;
    const int ks = blockIdx.x / nbg;
    // 4 consecutive groups per XCD: id = xcd + 8k  ->  group = 4*xcd + k%4 (needs groups % 32 == 0, else linear)
    const int id = blockIdx.x - ks * nbg;
    int b, g;
    if ((groups & 31) == 0) {
        const int per_b = groups, local = id % per_b;
        b = id / per_b;
        const int chunk = local / 32, l32 = local % 32;
        g = chunk * 32 + (l32 & 7) * 4 + (l32 >> 3);
    } else {
        b = id / groups; g = id % groups;
    }
    const int j = threadIdx.x % cp2, py = threadIdx.x / cp2;
    const bool live = py < PY;
    const int c = g * cpg + 2 * j;
    const half_t* src = c < C1 ? x : x2;
    const int cs = c < C1 ? C1 : C2, cc = c < C1 ? c : c - C1;
    const half_t* base = src + (long long)b * HW * cs + cc;
    float s = 0.f, q = 0.f;
    constexpr int GU = 8;        // pixels in flight per thread: loads first, arithmetic after (memory-level parallelism)
    if (live) {
        for (int p0 = py; p0 < HW; p0 += PY * GU) {
            half2_t v[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int p = p0 + u * PY;
                v[u] = p < HW ? *(const half2_t*)(base + (long long)p * cs) : (half2_t){0, 0};
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {      // same order as a pixel-at-a-time loop: bit-identical sums
                const float a0 = (float)v[u][0], a1 = (float)v[u][1];
                s += a0 + a1; q += a0 * a0 + a1 * a1;
            }
        }
    }
    __shared__ float red[2][8];
    s = wave_sum(s); q = wave_sum(q);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = s; red[1][wave] = q; }
    __syncthreads();
    float ts = 0.f, tq = 0.f;
    for (int w = 0; w < nw; ++w) { ts += red[0][w]; tq += red[1][w]; }
    const float inv = 1.0f / ((float)cpg * (float)HW);
    const float mean = ts * inv;
    const float rstd = rsqrtf(fmaxf(tq * inv - mean * mean, 0.f) + eps);
    if (threadIdx.x == 0 && ks == 0) { stats_out[((long long)b * groups + g) * 2] = mean; stats_out[((long long)b * groups + g) * 2 + 1] = rstd; }
    if (!live) return;
    const float sc0 = rstd * gamma[c], sc1 = rstd * gamma[c + 1];
    const float sh0 = beta[c] - mean * sc0, sh1 = beta[c + 1] - mean * sc1;
    half_t* ob = out + (long long)b * HW * C + c;
    for (int p0 = py + ks * PY * GU; p0 < HW; p0 += PY * GU * KS) {
        half2_t v[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int p = p0 + u * PY;
            v[u] = p < HW ? *(const half2_t*)(base + (long long)p * cs) : (half2_t){0, 0};
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int p = p0 + u * PY;
            if (p < HW) {
                float y0 = (float)v[u][0] * sc0 + sh0, y1 = (float)v[u][1] * sc1 + sh1;
                if (apply_silu) { y0 = silu_f(y0); y1 = silu_f(y1); }
                half2_t o = {(half_t)y0, (half_t)y1};
                *(half2_t*)(ob + (long long)p * C) = o;
            }
        }
    }
}

extern "C" int ief_groupnorm_silu_f16(const ief_half* x, const ief_half* x2, int C1, int C2, ief_half* out,
                                      const float* gamma, const float* beta, float* partial,
                                      int B, int HW, int groups, float eps, int apply_silu, void* stream) {
    if (!x || !out || !gamma || !beta || !partial) return IEF_EINVAL;
    if (C2 > 0 && !x2) return IEF_EINVAL;
    const int C = C1 + C2;
    if (B <= 0 || HW <= 0 || groups <= 0 || groups > GN_MAX_GROUPS) return IEF_ESHAPE;
    if ((C1 & 7) || (C2 & 7) || (C % groups) || C > 8 * 1024) return IEF_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int cpg_ = C / groups;
    // single launch when one (batch, group) slab is small enough for one workgroup to stream twice quickly
    if (!(cpg_ & 1) && !(C1 & 1) && (cpg_ >> 1) <= 256 && (long long)HW * cpg_ * 2 <= 48 * 1024) {
        const int cp2 = cpg_ >> 1;
        int PY = 512 / cp2;
        if (PY > HW) PY = HW;
        int threads = ((cp2 * PY + 63) / 64) * 64;
        if (threads > 512) { PY -= 1; threads = ((cp2 * PY + 63) / 64) * 64; }
        // workgroups per (batch, group): enough to put >= 2 workgroups on every CU, each keeping >= one full round of pixels
        const int rounds = (HW + PY * 8 - 1) / (PY * 8);       // 8 pixels in flight per thread and round
        static const int ks_max = env_int("IEF_GN_KS_MAX", 2);   // same-box A/B: 1 -> 7.45, 8 -> 7.42, 2 -> 7.39 ms per step
        int KS = 1;
        while (KS * 2 <= rounds && KS * 2 <= ks_max && B * groups * KS < 512) KS *= 2;
        // (mean, rstd) land where the three-launch path leaves them: after the split partials
        hipLaunchKernelGGL(gn_fused_kernel, dim3(B * groups * KS), dim3(threads), 0, st, x, x2, C1, C2, out, gamma, beta,
                           partial + (long long)B * gn_splits_host(HW) * groups * 2, HW, groups, eps, apply_silu, PY, KS);
        IEF_LAUNCH_CHECK();
        return IEF_OK;
    }
    const int splits = gn_splits_host(HW);
    const int C8 = C / 8;
    // statistics: up to 1024 threads (C8 * PY), PY bounded by the pixels a block owns
    int per = (HW + splits - 1) / splits;
    if (C > GN_PART_FLOATS) return IEF_ESHAPE;
    int PYs = 256 / C8;                    // ~256-thread blocks, several pixels per thread
    if (PYs > per) PYs = per;
    if (PYs > GN_PART_FLOATS / C) PYs = GN_PART_FLOATS / C;
    if (PYs < 1) PYs = 1;
    int ts = C8 * PYs;
    if (ts < 64) ts = 64;  // the LDS zero-fill / final write use the first 64 threads
    hipLaunchKernelGGL(gn_stats_kernel, dim3(splits, B), dim3(ts), 0, st, x, x2, C1, C2, partial, HW, groups, splits, PYs);
    IEF_LAUNCH_CHECK();
    float* stats = partial + (long long)B * splits * groups * 2;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * groups), dim3(64), 0, st, partial, stats, HW, groups, splits, C / groups, eps);
    IEF_LAUNCH_CHECK();
    // apply: ~256..512-thread blocks, ~4 pixels per thread
    int PYa = 256 / C8;
    if (PYa < 1) PYa = 1;
    if (PYa > HW) PYa = HW;
    int ppb = PYa * 4;
    if (ppb > HW) ppb = HW;
    int ta = C8 * PYa;
    if (ta < 64) ta = 64;
    const int gx = (HW + ppb - 1) / ppb;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gx, B), dim3(ta), 0, st, x, x2, C1, C2, out, gamma, beta, partial, HW,
                       groups, splits, eps, apply_silu, ppb, PYa);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm from the producer's per-tile column statistics (IefGemmParams.cstat_out): one wave per (batch, group) folds
// (tiles of the image) x (channels of the group) partial pairs in a fixed order; a group of a channel-concat input may
// take channels from both sources, each with its own tile height.
__global__ __launch_bounds__(64) void gn_finalize_cstat_kernel(const float* __restrict__ cs1, int bm1, int C1,
                                                               const float* __restrict__ cs2, int bm2, int C2,
                                                               float* __restrict__ stats, int HW, int groups, float eps) {
    const int C = C1 + C2, cpg = C / groups;
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int t1 = HW / bm1, t2 = C2 > 0 ? HW / bm2 : 0;
    float s = 0.f, q = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        const bool first = c < C1;
        const float* cs = first ? cs1 : cs2;
        const int Cs = first ? C1 : C2, cc = first ? c : c - C1, nt = first ? t1 : t2;
        for (int t = threadIdx.x; t < nt; t += 64) {
            const float* o = cs + ((long long)(b * nt + t) * Cs + cc) * 2;
            s += o[0]; q += o[1];
        }
    }
    s = wave_sum(s); q = wave_sum(q);
    if (threadIdx.x == 0) {
        const float inv = 1.0f / ((float)cpg * (float)HW);
        const float m = s * inv;
        stats[((long long)b * groups + g) * 2] = m;
        stats[((long long)b * groups + g) * 2 + 1] = rsqrtf(fmaxf(q * inv - m * m, 0.f) + eps);
    }
}

// ONE launch on the producer's statistics: a workgroup owns a channel slice [c_lo, c_hi) (whole groups, whole 8-channel
// chunks) of a run of pixels of one image.  It first folds the statistics of ITS channels only — a thread per channel sums
// that channel's per-tile pairs in tile order, then a thread per group sums the group's channels in channel order (fixed
// orders: bit reproducible) — and then normalises its pixels.  The fold reads nt * slice * 8 bytes, so this form is used
// while that stays small (the 32x32 / 16x16 levels: 2-8 tiles per image); above, the two-launch form below.
#define GNS_MAXC 512
__global__ __launch_bounds__(256) void gn_cstat_fused_kernel(const half_t* __restrict__ x, const half_t* __restrict__ x2, int C1, int C2,
                                                             half_t* __restrict__ out, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ cs1, int bm1,
                                                             const float* __restrict__ cs2, int bm2, int HW, int groups, float eps,
                                                             int apply_silu, int CW, int ppb) {
    __shared__ float ch_s[GNS_MAXC], ch_q[GNS_MAXC], g_mu[GNS_MAXC / 2], g_rs[GNS_MAXC / 2];
    const int C = C1 + C2, cpg = C / groups;
    const int b = blockIdx.z;
    const int c_lo = blockIdx.y * CW, c_hi = min(C, c_lo + CW), cw = c_hi - c_lo;
    const int t1 = HW / bm1, t2 = C2 > 0 ? HW / bm2 : 0;
    for (int i = threadIdx.x; i < cw; i += 256) {
        const int c = c_lo + i;
        const bool first = c < C1;
        const float* cs = first ? cs1 : cs2;
        const int Cs = first ? C1 : C2, cc = first ? c : c - C1, nt = first ? t1 : t2;
        float s = 0.f, q = 0.f;
        for (int t = 0; t < nt; ++t) {
            const float* o = cs + ((long long)(b * nt + t) * Cs + cc) * 2;
            s += o[0]; q += o[1];
        }
        ch_s[i] = s; ch_q[i] = q;
    }
    __syncthreads();
    const int ng = cw / cpg;
    for (int g = threadIdx.x; g < ng; g += 256) {
        float s = 0.f, q = 0.f;
        for (int j = 0; j < cpg; ++j) { s += ch_s[g * cpg + j]; q += ch_q[g * cpg + j]; }
        const float inv = 1.0f / ((float)cpg * (float)HW);
        const float m = s * inv;
        g_mu[g] = m;
        g_rs[g] = rsqrtf(fmaxf(q * inv - m * m, 0.f) + eps);
    }
    __syncthreads();
    const int C8s = cw >> 3;                       // 8-channel chunks of the slice
    const int PY = 256 / C8s;                      // pixel lanes
    const int cx = threadIdx.x % C8s, py = threadIdx.x / C8s;
    if (py >= PY) return;
    const int c = c_lo + cx * 8;
    float sc[8], sh[8];
    {
        const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
        const f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (cx * 8 + e) / cpg;
            const float ga = e < 4 ? g0[e] : g1[e - 4], be = e < 4 ? b0[e] : b1[e - 4];
            sc[e] = g_rs[g] * ga;
            sh[e] = be - g_mu[g] * sc[e];
        }
    }
    const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
    for (int p = p0 + py; p < p1; p += PY) {
        const long long pix = (long long)b * HW + p;
        const half8 v = load_cat8(x, x2, C1, C2, pix, c);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float y = (float)v[e] * sc[e] + sh[e];
            if (apply_silu) y = silu_f(y);
            o[e] = (half_t)y;
        }
        *(half8*)(out + pix * C + c) = o;
    }
}

static int lcm_i(int a, int b) {
    int x = a, y = b;
    while (y) { const int t = x % y; x = y; y = t; }
    return a / x * b;
}

extern "C" int ief_groupnorm_cstat_f16(const ief_half* x, const ief_half* x2, int C1, int C2, ief_half* out,
                                       const float* gamma, const float* beta, const float* cstat1, int bm1,
                                       const float* cstat2, int bm2, float* stats, int B, int HW, int groups, float eps,
                                       int apply_silu, void* stream) {
    if (!x || !out || !gamma || !beta || !cstat1 || !stats) return IEF_EINVAL;
    if (C2 > 0 && (!x2 || !cstat2)) return IEF_EINVAL;
    const int C = C1 + C2;
    if (B <= 0 || HW <= 0 || groups <= 0 || groups > GN_MAX_GROUPS) return IEF_ESHAPE;
    if ((C1 & 7) || (C2 & 7) || (C % groups) || C > 8 * 1024) return IEF_ESHAPE;
    if (bm1 <= 0 || (HW % bm1) || (C2 > 0 && (bm2 <= 0 || (HW % bm2)))) return IEF_ESHAPE;
    hipStream_t st = (hipStream_t)stream;
    // one launch when a workgroup's share of the fold is small: slices of whole groups and whole 8-channel chunks
    {
        const int cpg = C / groups, unit = lcm_i(cpg, 8);
        const int nt1 = HW / bm1, nt2 = C2 > 0 ? HW / bm2 : 0;
        const int nt = nt1 > nt2 ? nt1 : nt2;
        int CW = unit * ((192 + unit - 1) / unit);            // ~192 channels per slice
        if (CW > C) CW = C;
        static const int fused_ok = env_int("IEF_GN_CSTAT_FUSED", 1);
        static const int fused_kb = env_int("IEF_GN_CSTAT_FUSED_KB", 16);
        if (fused_ok && unit <= 256 && CW <= GNS_MAXC && CW / 8 <= 256 && (long long)nt * CW * 8 <= fused_kb * 1024 && B <= 65535) {
            const int PY = 256 / (CW / 8);
            int ppb = PY * 4;                                  // ~4 pixels per thread
            if (ppb > HW) ppb = HW;
            dim3 grid((HW + ppb - 1) / ppb, (C + CW - 1) / CW, B);
            hipLaunchKernelGGL(gn_cstat_fused_kernel, grid, dim3(256), 0, st, x, x2, C1, C2, out, gamma, beta, cstat1, bm1,
                               cstat2, bm2, HW, groups, eps, apply_silu, CW, ppb);
            IEF_LAUNCH_CHECK();
            return IEF_OK;
        }
    }
    hipLaunchKernelGGL(gn_finalize_cstat_kernel, dim3(B * groups), dim3(64), 0, st, cstat1, bm1, C1, cstat2, bm2, C2, stats,
                       HW, groups, eps);
    IEF_LAUNCH_CHECK();
    const int C8 = C / 8;
    int PYa = 256 / C8;
    if (PYa < 1) PYa = 1;
    if (PYa > HW) PYa = HW;
    int ppb = PYa * 4;
    if (ppb > HW) ppb = HW;
    int ta = C8 * PYa;
    if (ta < 64) ta = 64;
    const int gx = (HW + ppb - 1) / ppb;
    // splits = 0: the apply kernel then reads the finalized statistics at the start of the buffer it is handed
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gx, B), dim3(ta), 0, st, x, x2, C1, C2, out, gamma, beta, stats, HW, groups, 0,
                       eps, apply_silu, ppb, PYa);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// one wave per row; row kept in registers (C <= 8 * 64 * LN_MAXCH)
#define LN_MAXCH 4
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, half_t* __restrict__ out,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, C8 = C >> 3;
    half8 v[LN_MAXCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < C8) {
            v[i] = *(const half8*)(x + (long long)row * C + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[i][e];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < C8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXCH; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < C8) {
            const int c = c8 * 8;
            const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
            const f32x4 b0 = *(const f32x4*)(beta + c), b1 = *(const f32x4*)(beta + c + 4);
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ga = e < 4 ? g0[e] : g1[e - 4], be = e < 4 ? b0[e] : b1[e - 4];
                o[e] = (half_t)(((float)v[i][e] - mean) * rstd * ga + be);
            }
            *(half8*)(out + (long long)row * C + c) = o;
        }
    }
}

extern "C" int ief_layernorm_f16(const ief_half* x, ief_half* out, const float* gamma, const float* beta,
                                 int rows, int C, float eps, void* stream) {
    if (!x || !out || !gamma || !beta) return IEF_EINVAL;
    if (rows <= 0 || C <= 0 || (C & 7) || C > 8 * 64 * LN_MAXCH) return IEF_ESHAPE;
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, out, gamma, beta,
                       rows, C, eps);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

__global__ __launch_bounds__(256) void geglu_kernel(const half_t* __restrict__ in, half_t* __restrict__ out,
                                                    long long rows, int Ch) {
    const int C8 = Ch >> 3;
    const long long total = rows * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / C8;
        const int c = (int)(i % C8) * 8;
        const half8 h = *(const half8*)(in + r * 2 * Ch + c);
        const half8 g = *(const half8*)(in + r * 2 * Ch + Ch + c);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)h[e] * gelu_f((float)g[e]));
        *(half8*)(out + r * Ch + c) = o;
    }
}

extern "C" int ief_geglu_f16(const ief_half* in, ief_half* out, int rows, int Ch, void* stream) {
    if (!in || !out) return IEF_EINVAL;
    if (rows <= 0 || Ch <= 0 || (Ch & 7)) return IEF_ESHAPE;
    const long long total = (long long)rows * (Ch / 8);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(geglu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, out, (long long)rows, Ch);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
