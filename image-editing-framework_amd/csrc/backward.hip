// Activation-gradient kernels of the null-text-inversion path (everything that is not a GEMM / conv /
// attention): GroupNorm(+SiLU), LayerNorm, GEGLU, the resampling adjoints, conv_out's data gradient,
// the NTI objective and its Adam step.  Reference: /root/reference/p2p/inversion/nti.py:15-33 (torch
// autograd + torch.optim.Adam); the formulas are the textbook adjoints of the forward kernels in
// norm.hip / conv_io.hip / elementwise.hip.  fp16 storage, fp32 arithmetic, HBM/L2-bound.
#include "ief_common.h"
#include "ief_params.h"

__device__ __forceinline__ half_t sat16(float v) { return (half_t)fminf(fmaxf(v, -65504.f), 65504.f); }

// block-wide sum of two values (blockDim.x <= 1024), result broadcast to every thread
__device__ __forceinline__ void block_sum2(float& a, float& b, float (*red)[16]) {
    a = wave_sum(a); b = wave_sum(b);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();                      // previous use of `red` is over
    if ((threadIdx.x & 63) == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    float ta = 0.f, tb = 0.f;
    for (int w = 0; w < nw; ++w) { ta += red[0][w]; tb += red[1][w]; }
    a = ta; b = tb;
}

__device__ __forceinline__ float dsilu_f(float z) {
    const float s = 1.0f / (1.0f + __expf(-z));
    return s * (1.0f + z * (1.0f - s));
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm (+SiLU) backward: one workgroup per (batch, group) streams its slab three times (statistics,
// the two gradient moments, the result); the slab (<= a few hundred KB) stays in this XCD's L2.
//   xh = (x - mean) rstd,  z = xh gamma + beta,  dz = dy silu'(z),  g = dz gamma
//   dx = rstd (g - mean_grp(g) - xh mean_grp(g xh)) + add
// Thread (py, j) keeps channel pair j of the group for all its pixels (4-byte accesses, as gn_fused_kernel).
__global__ __launch_bounds__(512) void gn_bwd_kernel(const half_t* __restrict__ x, const half_t* __restrict__ x2, int C1,
                                                     int C2, const half_t* __restrict__ dy, const half_t* __restrict__ add,
                                                     half_t* __restrict__ dx, half_t* __restrict__ dx2,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int HW, int groups, float eps, int apply_silu, int PY) {
    const int C = C1 + C2, cpg = C / groups, cp2 = cpg >> 1;
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int j = threadIdx.x % cp2, py = threadIdx.x / cp2;
    const bool live = py < PY;
    const int c = g * cpg + 2 * j;
    const bool first = c < C1;
    const half_t* src = first ? x : x2;
    const int cs = first ? C1 : C2, cc = first ? c : c - C1;
    const half_t* base = src + (long long)b * HW * cs + cc;
    const half_t* gy = dy + (long long)b * HW * C + c;
    __shared__ float red[2][16];
    float s = 0.f, q = 0.f;
    if (live)
        for (int p = py; p < HW; p += PY) {
            const half2_t v = *(const half2_t*)(base + (long long)p * cs);
            const float a0 = (float)v[0], a1 = (float)v[1];
            s += a0 + a1; q += a0 * a0 + a1 * a1;
        }
    block_sum2(s, q, red);
    const float inv = 1.0f / ((float)cpg * (float)HW);
    const float mean = s * inv;
    const float rstd = rsqrtf(fmaxf(q * inv - mean * mean, 0.f) + eps);
    const float ga0 = gamma[c], ga1 = gamma[c + 1], be0 = beta[c], be1 = beta[c + 1];
    float s1 = 0.f, s2 = 0.f;
    if (live)
        for (int p = py; p < HW; p += PY) {
            const half2_t v = *(const half2_t*)(base + (long long)p * cs);
            const half2_t d = *(const half2_t*)(gy + (long long)p * C);
            const float h0 = ((float)v[0] - mean) * rstd, h1 = ((float)v[1] - mean) * rstd;
            float g0 = (float)d[0], g1 = (float)d[1];
            if (apply_silu) { g0 *= dsilu_f(h0 * ga0 + be0); g1 *= dsilu_f(h1 * ga1 + be1); }
            g0 *= ga0; g1 *= ga1;
            s1 += g0 + g1; s2 += g0 * h0 + g1 * h1;
        }
    block_sum2(s1, s2, red);
    if (!live) return;
    const float m1 = s1 * inv, m2 = s2 * inv;
    half_t* ob = first ? dx + (long long)b * HW * C1 + cc : dx2 + (long long)b * HW * C2 + cc;
    const half_t* ad = add ? add + (long long)b * HW * C + c : nullptr;
    for (int p = py; p < HW; p += PY) {
        const half2_t v = *(const half2_t*)(base + (long long)p * cs);
        const half2_t d = *(const half2_t*)(gy + (long long)p * C);
        const float h0 = ((float)v[0] - mean) * rstd, h1 = ((float)v[1] - mean) * rstd;
        float g0 = (float)d[0], g1 = (float)d[1];
        if (apply_silu) { g0 *= dsilu_f(h0 * ga0 + be0); g1 *= dsilu_f(h1 * ga1 + be1); }
        g0 *= ga0; g1 *= ga1;
        float r0 = rstd * (g0 - m1 - h0 * m2), r1 = rstd * (g1 - m1 - h1 * m2);
        if (ad) { const half2_t a = *(const half2_t*)(ad + (long long)p * C); r0 += (float)a[0]; r1 += (float)a[1]; }
        half2_t o = {sat16(r0), sat16(r1)};
        *(half2_t*)(ob + (long long)p * cs) = o;
    }
}

// Large slabs: two launches over (pixel split, batch) with 16-byte accesses, like gn_stats / gn_apply in norm.hip.
// A thread owns ONE 8-channel chunk for all its pixels; (mean, rstd) come from the forward.
#define GNB_MAX_GROUPS 64
#define GNB_PART_FLOATS 4096
__device__ __forceinline__ half8 load_cat8b(const half_t* x, const half_t* x2, int C1, int C2, long long pix, int c) {
    if (c < C1) return *(const half8*)(x + pix * C1 + c);
    return *(const half8*)(x2 + pix * C2 + (c - C1));
}

// grid (splits, B): per-(batch, split, group) partial sums of g and g * xhat
__global__ void gn_bwd_partial_kernel(const half_t* __restrict__ x, const half_t* __restrict__ x2, int C1, int C2,
                                      const half_t* __restrict__ dy, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, const float* __restrict__ stats,
                                      float* __restrict__ partial, int HW, int groups, int splits, int apply_silu, int PY) {
    const int C = C1 + C2, C8 = C >> 3, cpg = C / groups;
    const int b = blockIdx.y, sp = blockIdx.x;
    const int cx = threadIdx.x % C8, py = threadIdx.x / C8;
    const int c = cx * 8;
    __shared__ float ps[GNB_PART_FLOATS], pq[GNB_PART_FLOATS];   // per-(pixel lane, channel) partials, summed in a fixed order
    float mean[8], rstd[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c + e) / cpg;
        mean[e] = stats[((long long)b * groups + g) * 2];
        rstd[e] = stats[((long long)b * groups + g) * 2 + 1];
        ga[e] = gamma[c + e]; be[e] = beta[c + e];
        s1[e] = 0.f; s2[e] = 0.f;
    }
    const int per = (HW + splits - 1) / splits;
    const int p0 = sp * per, p1 = min(HW, p0 + per);
    for (int p = p0 + py; py < PY && p < p1; p += PY) {
        const long long pix = (long long)b * HW + p;
        const half8 v = load_cat8b(x, x2, C1, C2, pix, c);
        const half8 d = *(const half8*)(dy + pix * C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float h = ((float)v[e] - mean[e]) * rstd[e];
            float g = (float)d[e];
            if (apply_silu) g *= dsilu_f(h * ga[e] + be[e]);
            g *= ga[e];
            s1[e] += g; s2[e] += g * h;
        }
    }
    if (py < PY) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { ps[py * C + c + e] = s1[e]; pq[py * C + c + e] = s2[e]; }
    }
    __syncthreads();
    if (threadIdx.x < groups) {
        const int g = threadIdx.x;
        float t1 = 0.f, t2 = 0.f;
        for (int r = 0; r < PY; ++r)
            for (int cc = g * cpg; cc < (g + 1) * cpg; ++cc) { t1 += ps[r * C + cc]; t2 += pq[r * C + cc]; }
        float* o = partial + (((long long)b * splits + sp) * groups + g) * 2;
        o[0] = t1; o[1] = t2;
    }
}

// grid (pixel blocks, B): sums the split partials (one wave per group, in LDS), then writes dx
__global__ void gn_bwd_apply_kernel(const half_t* __restrict__ x, const half_t* __restrict__ x2, int C1, int C2,
                                    const half_t* __restrict__ dy, const half_t* __restrict__ add, half_t* __restrict__ dx,
                                    half_t* __restrict__ dx2, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ stats, const float* __restrict__ partial, int HW, int groups,
                                    int splits, int apply_silu, int pix_per_block, int PY) {
    const int C = C1 + C2, C8 = C >> 3, cpg = C / groups;
    const int b = blockIdx.y;
    __shared__ float m1s[GNB_MAX_GROUPS], m2s[GNB_MAX_GROUPS];
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
        const float inv = 1.0f / ((float)cpg * (float)HW);
        for (int g = wave; g < groups; g += nw) {
            float a = 0.f, q = 0.f;
            if (lane < splits) {
                const float* o = partial + (((long long)b * splits + lane) * groups + g) * 2;
                a = o[0]; q = o[1];
            }
            a = wave_sum(a); q = wave_sum(q);
            if (lane == 0) { m1s[g] = a * inv; m2s[g] = q * inv; }
        }
    }
    __syncthreads();
    const int cx = threadIdx.x % C8, py = threadIdx.x / C8;
    const int c = cx * 8;
    float mean[8], rstd[8], ga[8], be[8], m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c + e) / cpg;
        mean[e] = stats[((long long)b * groups + g) * 2];
        rstd[e] = stats[((long long)b * groups + g) * 2 + 1];
        ga[e] = gamma[c + e]; be[e] = beta[c + e];
        m1[e] = m1s[g]; m2[e] = m2s[g];
    }
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(HW, p0 + pix_per_block);
    for (int p = p0 + py; py < PY && p < p1; p += PY) {
        const long long pix = (long long)b * HW + p;
        const half8 v = load_cat8b(x, x2, C1, C2, pix, c);
        const half8 d = *(const half8*)(dy + pix * C + c);
        half8 a = {0, 0, 0, 0, 0, 0, 0, 0};
        if (add) a = *(const half8*)(add + pix * C + c);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float h = ((float)v[e] - mean[e]) * rstd[e];
            float g = (float)d[e];
            if (apply_silu) g *= dsilu_f(h * ga[e] + be[e]);
            g *= ga[e];
            o[e] = sat16(rstd[e] * (g - m1[e] - h * m2[e]) + (float)a[e]);
        }
        if (c < C1) *(half8*)(dx + pix * C1 + c) = o;
        else *(half8*)(dx2 + pix * C2 + (c - C1)) = o;
    }
}

extern "C" int ief_groupnorm_bwd_f16(const ief_half* x, const ief_half* x2, int C1, int C2, const ief_half* dy,
                                     const ief_half* add, ief_half* dx, ief_half* dx2, const float* gamma, const float* beta,
                                     const float* stats, float* partial, int B, int HW, int groups, float eps,
                                     int apply_silu, void* stream) {
    if (!x || !dy || !dx || !gamma || !beta) return IEF_EINVAL;
    if (C2 > 0 && (!x2 || !dx2)) return IEF_EINVAL;
    const int C = C1 + C2;
    if (B <= 0 || HW <= 0 || groups <= 0 || groups > GNB_MAX_GROUPS || C1 <= 0 || C2 < 0 || (C % groups)) return IEF_ESHAPE;
    const int cpg = C / groups;
    hipStream_t st = (hipStream_t)stream;
    const bool big = (long long)HW * cpg * 2 > 48 * 1024;
    if (big && stats && partial && !(C1 & 7) && !(C2 & 7) && C <= GNB_PART_FLOATS) {
        const int splits = ief_gn_splits(HW);
        const int C8 = C / 8;
        const int per = (HW + splits - 1) / splits;
        int PYs = 256 / C8;
        if (PYs > per) PYs = per;
        if (PYs > GNB_PART_FLOATS / C) PYs = GNB_PART_FLOATS / C;
        if (PYs < 1) PYs = 1;
        int ts = ((C8 * PYs + 63) / 64) * 64;   // whole waves: the group fold below uses every lane of a wave
        hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(splits, B), dim3(ts), 0, st, x, x2, C1, C2, dy, gamma, beta, stats,
                           partial, HW, groups, splits, apply_silu, PYs);
        IEF_LAUNCH_CHECK();
        int PYa = 256 / C8;
        if (PYa < 1) PYa = 1;
        if (PYa > HW) PYa = HW;
        int ppb = PYa * 4;
        if (ppb > HW) ppb = HW;
        int ta = ((C8 * PYa + 63) / 64) * 64;   // whole waves (wave_sum over the split partials)
        hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3((HW + ppb - 1) / ppb, B), dim3(ta), 0, st, x, x2, C1, C2, dy, add, dx, dx2,
                           gamma, beta, stats, partial, HW, groups, splits, apply_silu, ppb, PYa);
        IEF_LAUNCH_CHECK();
        return IEF_OK;
    }
    if ((cpg & 1) || (C1 & 1) || (cpg >> 1) > 512) return IEF_ESHAPE;
    const int cp2 = cpg >> 1;
    int PY = 512 / cp2;
    if (PY > HW) PY = HW;
    int threads = ((cp2 * PY + 63) / 64) * 64;
    if (threads > 512) { PY -= 1; threads = ((cp2 * PY + 63) / 64) * 64; }
    if (PY < 1) return IEF_ESHAPE;
    hipLaunchKernelGGL(gn_bwd_kernel, dim3(B * groups), dim3(threads), 0, st, x, x2, C1, C2, dy, add, dx, dx2, gamma, beta,
                       HW, groups, eps, apply_silu, PY);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row, the row and its gradient in registers (C <= 8 * 64 * 4)
#define LNB_MAXCH 4
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const half_t* __restrict__ x, const half_t* __restrict__ dy,
                                                            const half_t* __restrict__ add, half_t* __restrict__ dx,
                                                            const float* __restrict__ gamma, int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, C8 = C >> 3;
    half8 v[LNB_MAXCH], d[LNB_MAXCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXCH; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < C8) {
            v[i] = *(const half8*)(x + (long long)row * C + c8 * 8);
            d[i] = *(const half8*)(dy + (long long)row * C + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[i][e];
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LNB_MAXCH; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < C8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float t = (float)v[i][e] - mean; q += t * t; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
    float gq[LNB_MAXCH][8];
#pragma unroll
    for (int i = 0; i < LNB_MAXCH; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < C8) {
            const int c = c8 * 8;
            const f32x4 g0 = *(const f32x4*)(gamma + c), g1 = *(const f32x4*)(gamma + c + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ga = e < 4 ? g0[e] : g1[e - 4];
                const float g = (float)d[i][e] * ga, h = ((float)v[i][e] - mean) * rstd;
                gq[i][e] = g;
                s1 += g; s2 += g * h;
            }
        }
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < LNB_MAXCH; ++i) {
        const int c8 = lane + 64 * i;
        if (c8 < C8) {
            const int c = c8 * 8;
            half8 a = {0, 0, 0, 0, 0, 0, 0, 0};
            if (add) a = *(const half8*)(add + (long long)row * C + c);
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float h = ((float)v[i][e] - mean) * rstd;
                o[e] = sat16(rstd * (gq[i][e] - m1 - h * m2) + (float)a[e]);
            }
            *(half8*)(dx + (long long)row * C + c) = o;
        }
    }
}

extern "C" int ief_layernorm_bwd_f16(const ief_half* x, const ief_half* dy, const ief_half* add, ief_half* dx,
                                     const float* gamma, int rows, int C, float eps, void* stream) {
    if (!x || !dy || !dx || !gamma) return IEF_EINVAL;
    if (rows <= 0 || C <= 0 || (C & 7) || C > 8 * 64 * LNB_MAXCH) return IEF_ESHAPE;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, dy, add, dx, gamma,
                       rows, C, eps);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// GEGLU on the interleaved projection ([8 hidden | 8 gate] column groups, see unet.GEGLU)
__global__ __launch_bounds__(256) void geglu_il_kernel(const half_t* __restrict__ pre, half_t* __restrict__ out,
                                                       long long rows, int Ch) {
    const int C8 = Ch >> 3;
    const long long total = rows * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / C8;
        const int g8 = (int)(i % C8);
        const half8 hh = *(const half8*)(pre + r * 2 * Ch + 16 * g8);
        const half8 gg = *(const half8*)(pre + r * 2 * Ch + 16 * g8 + 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)hh[e] * gelu_f((float)gg[e]));
        *(half8*)(out + r * Ch + 8 * g8) = o;
    }
}

__global__ __launch_bounds__(256) void geglu_il_bwd_kernel(const half_t* __restrict__ pre, const half_t* __restrict__ dy,
                                                           half_t* __restrict__ dpre, long long rows, int Ch) {
    const int C8 = Ch >> 3;
    const long long total = rows * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / C8;
        const int g8 = (int)(i % C8);
        const half8 hh = *(const half8*)(pre + r * 2 * Ch + 16 * g8);
        const half8 gg = *(const half8*)(pre + r * 2 * Ch + 16 * g8 + 8);
        const half8 d = *(const half8*)(dy + r * Ch + 8 * g8);
        half8 dh, dg;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gv = (float)gg[e], dv = (float)d[e];
            const float cdf = 0.5f * (1.0f + erff(gv * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * gv * gv);
            dh[e] = sat16(dv * gv * cdf);
            dg[e] = sat16(dv * (float)hh[e] * (cdf + gv * pdf));
        }
        *(half8*)(dpre + r * 2 * Ch + 16 * g8) = dh;
        *(half8*)(dpre + r * 2 * Ch + 16 * g8 + 8) = dg;
    }
}

static inline int ew_grid(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

extern "C" int ief_geglu_il_f16(const ief_half* pre, ief_half* out, int rows, int Ch, void* stream) {
    if (!pre || !out) return IEF_EINVAL;
    if (rows <= 0 || Ch <= 0 || (Ch & 7)) return IEF_ESHAPE;
    hipLaunchKernelGGL(geglu_il_kernel, dim3(ew_grid((long long)rows * (Ch / 8))), dim3(256), 0, (hipStream_t)stream, pre, out,
                       (long long)rows, Ch);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

extern "C" int ief_geglu_il_bwd_f16(const ief_half* pre, const ief_half* dy, ief_half* dpre, int rows, int Ch, void* stream) {
    if (!pre || !dy || !dpre) return IEF_EINVAL;
    if (rows <= 0 || Ch <= 0 || (Ch & 7)) return IEF_ESHAPE;
    hipLaunchKernelGGL(geglu_il_bwd_kernel, dim3(ew_grid((long long)rows * (Ch / 8))), dim3(256), 0, (hipStream_t)stream, pre,
                       dy, dpre, (long long)rows, Ch);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// resampling adjoints
__global__ __launch_bounds__(256) void zero_insert2x_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int B,
                                                            int H, int W, int C) {
    const int C8 = C >> 3;
    const long long total = (long long)B * 2 * H * 2 * W * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        long long pix = i / C8;
        const int xo = (int)(pix % (2 * W)); pix /= 2 * W;
        const int yo = (int)(pix % (2 * H));
        const int b = (int)(pix / (2 * H));
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (!(xo & 1) && !(yo & 1)) v = *(const half8*)(in + (((long long)b * H + (yo >> 1)) * W + (xo >> 1)) * C + c8 * 8);
        *(half8*)(out + i * 8) = v;
    }
}

__global__ __launch_bounds__(256) void pool2x2_sum_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int B,
                                                          int H, int W, int C) {
    const int C8 = C >> 3;
    const long long total = (long long)B * H * W * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        long long pix = i / C8;
        const int xo = (int)(pix % W); pix /= W;
        const int yo = (int)(pix % H);
        const int b = (int)(pix / H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
            for (int dxx = 0; dxx < 2; ++dxx) {
                const half8 v = *(const half8*)(in + (((long long)b * 2 * H + 2 * yo + dyy) * 2 * W + 2 * xo + dxx) * C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sat16(acc[e]);
        *(half8*)(out + i * 8) = o;
    }
}

extern "C" int ief_zero_insert2x_f16(const ief_half* in, ief_half* out, int B, int H, int W, int C, void* stream) {
    if (!in || !out) return IEF_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return IEF_ESHAPE;
    hipLaunchKernelGGL(zero_insert2x_kernel, dim3(ew_grid((long long)B * 4 * H * W * (C / 8))), dim3(256), 0,
                       (hipStream_t)stream, in, out, B, H, W, C);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

extern "C" int ief_pool2x2_sum_f16(const ief_half* in, ief_half* out, int B, int H, int W, int C, void* stream) {
    if (!in || !out) return IEF_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return IEF_ESHAPE;
    hipLaunchKernelGGL(pool2x2_sum_kernel, dim3(ew_grid((long long)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, in,
                       out, B, H, W, C);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// conv_out data gradient: dh[b][y][x][c] = sum_{co,ky,kx} d_eps[b][co][y+1-ky][x+1-kx] w[co][ky][kx][c]
__global__ __launch_bounds__(256) void conv_out_bwd_kernel(const float* __restrict__ de, const half_t* __restrict__ w,
                                                           half_t* __restrict__ dh, int B, int C, int H, int W, int Cout) {
    const int C8 = C >> 3;
    const long long total = (long long)B * H * W * C8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        long long pix = i / C8;
        const int xo = (int)(pix % W); pix /= W;
        const int yo = (int)(pix % H);
        const int b = (int)(pix / H);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int co = 0; co < Cout; ++co)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int ys = yo + 1 - ky;
                if (ys < 0 || ys >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int xs = xo + 1 - kx;
                    if (xs < 0 || xs >= W) continue;
                    const float g = de[(((long long)b * Cout + co) * H + ys) * W + xs];
                    const half8 wv = *(const half8*)(w + (((long long)co * 3 + ky) * 3 + kx) * C + c8 * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += g * (float)wv[e];
                }
            }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = sat16(acc[e]);
        *(half8*)(dh + i * 8) = o;
    }
}

extern "C" int ief_conv_out_bwd_f32(const float* d_eps, const ief_half* w, ief_half* dh, int B, int C, int H, int W, int Cout,
                                    void* stream) {
    if (!d_eps || !w || !dh) return IEF_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || Cout <= 0 || Cout > 16) return IEF_ESHAPE;
    hipLaunchKernelGGL(conv_out_bwd_kernel, dim3(ew_grid((long long)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       d_eps, w, dh, B, C, H, W, Cout);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// NTI objective and its gradient w.r.t. eps_u, normalised to max |.| = grad_scale for the fp16 backward pass
__global__ __launch_bounds__(1024) void nti_loss_grad_kernel(const float* __restrict__ eu, const float* __restrict__ ec,
                                                             const float* __restrict__ x, const float* __restrict__ target,
                                                             const float* __restrict__ coef, float* __restrict__ d_eps,
                                                             float* __restrict__ stats, int n, float grad_scale) {
    // same operation order as cfg_ddim_kernel (elementwise.hip): x0 = (x - sqrt(1-a_f) e) / sqrt(a_f); rec = sqrt(a_t) x0 + sqrt(1-a_t) e
    const float a_f = coef[0], a_t = coef[1], g = coef[2];
    const float sb_f = sqrtf(1.0f - a_f), sa_f = sqrtf(a_f), sa_t = sqrtf(a_t), sb_t = sqrtf(1.0f - a_t);
    const float drec_de = sb_t - sa_t * sb_f / sa_f;
    __shared__ float red[2][16];
    float sq = 0.f, mx = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float u = eu[i];
        const float e = u + g * (ec[i] - u);
        const float d = (sa_t * ((x[i] - sb_f * e) / sa_f) + sb_t * e) - target[i];
        sq += d * d;
        mx = fmaxf(mx, fabsf(d));
    }
    sq = wave_sum(sq); mx = wave_max(mx);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = sq; red[1][wave] = mx; }
    __syncthreads();
    float tsq = 0.f, tmx = 0.f;
    for (int w = 0; w < 16; ++w) { tsq += red[0][w]; tmx = fmaxf(tmx, red[1][w]); }
    const float inv = tmx > 0.f ? grad_scale / tmx : 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float u = eu[i];
        const float e = u + g * (ec[i] - u);
        const float d = (sa_t * ((x[i] - sb_f * e) / sa_f) + sb_t * e) - target[i];
        d_eps[i] = d * inv;
    }
    if (threadIdx.x == 0) {
        stats[0] = tsq / (float)n;
        stats[1] = tmx > 0.f ? (2.0f * drec_de * (1.0f - g) / (float)n) * (tmx / grad_scale) : 0.f;
    }
}

extern "C" int ief_nti_loss_grad_f32(const float* eps_u, const float* eps_c, const float* x, const float* target,
                                     const float* coef, float* d_eps, float* stats, int n, float grad_scale, void* stream) {
    if (!eps_u || !eps_c || !x || !target || !coef || !d_eps || !stats) return IEF_EINVAL;
    if (n <= 0 || n > (1 << 20) || !(grad_scale > 0.f)) return IEF_ESHAPE;
    hipLaunchKernelGGL(nti_loss_grad_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, eps_u, eps_c, x, target, coef,
                       d_eps, stats, n, grad_scale);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// torch.optim.Adam (single-tensor form, no weight decay / amsgrad), t = step[0] + 1
__global__ __launch_bounds__(256) void nti_adam_kernel(float* __restrict__ param, float* __restrict__ m, float* __restrict__ v,
                                                       const half_t* __restrict__ grad16, const float* __restrict__ stats,
                                                       const float* __restrict__ hyper, const int* __restrict__ step,
                                                       half_t* __restrict__ param16, int n) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
    const float t = (float)(step[0] + 1);
    const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
    const float step_size = lr / bc1, bc2s = sqrtf(bc2);
    const float factor = stats[1];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float g = (float)grad16[i] * factor;
        const float mi = b1 * m[i] + (1.0f - b1) * g;
        const float vi = b2 * v[i] + (1.0f - b2) * (g * g);
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        const float pn = param[i] - step_size * (mi / denom);
        param[i] = pn;
        param16[i] = (half_t)pn;
    }
}
__global__ void nti_step_inc_kernel(int* step) { step[0] += 1; }

extern "C" int ief_nti_adam_f32(float* param, float* m, float* v, const ief_half* grad16, const float* stats,
                                const float* hyper, int* step, ief_half* param16, int n, void* stream) {
    if (!param || !m || !v || !grad16 || !stats || !hyper || !step || !param16) return IEF_EINVAL;
    if (n <= 0) return IEF_ESHAPE;
    hipLaunchKernelGGL(nti_adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, param, m, v, grad16, stats, hyper,
                       step, param16, n);
    IEF_LAUNCH_CHECK();
    hipLaunchKernelGGL(nti_step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
