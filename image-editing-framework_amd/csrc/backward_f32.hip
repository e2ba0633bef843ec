// Activation-gradient kernels of the fp32-storage modes (precision "f32" / "f16x3"): the reverse pass of null-text inversion
// and of Pix2Pix-zero at the reference's precision.
//
// The reference differentiates the UNet with torch autograd in fp32 (`/root/reference/p2p/inversion/nti.py:15-33`,
// `/root/reference/pix2pix-zero/model/sd_utils.py:160-174`; `dtype = torch.float32`, `p2p/edit_real.py:45`).  Every weight is
// frozen, so the pass is a chain of activation gradients (grad.py): linear / convolution data gradients are the forward
// GEMM kernels on re-packed fp32 weights (exact_f32.hip / split_x3.hip); this file holds the adjoints that are not GEMMs, in
// fp32 with the formulas of csrc/backward.hip, and the pieces of an attention backward on MATERIALISED fp32 maps:
//   P = softmax(scale q k^T) (scores GEMM + row softmax), dP = dO V^T (GEMM), dS = scale P o (dP - rowsum(dP o P))
//   (`softmax_bwd_rows_f32`), dQ = dS K, dK = dS^T Q, dV = P^T dO (GEMMs on `transpose_batched_f32` copies of the maps).
// Simple kernels: this mode is for parity with the fp32 reference, the fp16 path (backward.hip, attention_bwd.hip) for speed.
#include "ief_common.h"
#include "ief_params.h"

__device__ __forceinline__ float dsilu_x(float z) {
    const float s = 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
}

static inline int ewf_grid(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// one value summed over the workgroup (256 threads), the same total in every thread, fixed order
__device__ __forceinline__ float wg_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm (+SiLU) backward: one workgroup per (batch, group); two-pass statistics (as the fp32 forward), then the two
// gradient moments, then the result.  xh = (x - mean) rstd, z = xh gamma + beta, g = dy silu'(z) gamma,
// dx = rstd (g - mean(g) - xh mean(g xh)) + add; a channel-concat input gets its two gradients separately.
__global__ __launch_bounds__(256) void gn_bwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ x2, int C1, int C2,
                                                         const float* __restrict__ dy, const float* __restrict__ add,
                                                         float* __restrict__ dx, float* __restrict__ dx2,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int HW,
                                                         int groups, float eps, int apply_silu) {
    __shared__ float red[4];
    const int C = C1 + C2, cpg = C / groups;
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int c0 = g * cpg;
    const long long n = (long long)HW * cpg;
    auto xat = [&](long long pix, int c) -> float {
        return c < C1 ? x[((long long)b * HW + pix) * C1 + c] : x2[((long long)b * HW + pix) * C2 + (c - C1)];
    };
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) { const long long pix = i / cpg; s += xat(pix, c0 + (int)(i - pix * cpg)); }
    const float mean = wg_sum(s, red) / (float)n;
    float q = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const long long pix = i / cpg;
        const float d = xat(pix, c0 + (int)(i - pix * cpg)) - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wg_sum(q, red) / (float)n + eps);
    auto grad_at = [&](long long pix, int c, float& h) -> float {
        h = (xat(pix, c) - mean) * rstd;
        float gv = dy[((long long)b * HW + pix) * C + c];
        if (apply_silu) gv *= dsilu_x(h * gamma[c] + beta[c]);
        return gv * gamma[c];
    };
    float s1 = 0.f, s2 = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const long long pix = i / cpg;
        float h;
        const float gv = grad_at(pix, c0 + (int)(i - pix * cpg), h);
        s1 += gv; s2 += gv * h;
    }
    const float m1 = wg_sum(s1, red) / (float)n;
    const float m2 = wg_sum(s2, red) / (float)n;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const long long pix = i / cpg;
        const int c = c0 + (int)(i - pix * cpg);
        float h;
        const float gv = grad_at(pix, c, h);
        float r = rstd * (gv - m1 - h * m2);
        if (add) r += add[((long long)b * HW + pix) * C + c];
        if (c < C1) dx[((long long)b * HW + pix) * C1 + c] = r;
        else dx2[((long long)b * HW + pix) * C2 + (c - C1)] = r;
    }
}
extern "C" int ief_groupnorm_bwd_f32(const float* x, const float* x2, int C1, int C2, const float* dy, const float* add, float* dx,
                                     float* dx2, const float* gamma, const float* beta, int B, int HW, int groups, float eps,
                                     int silu, void* stream) {
    if (!x || !dy || !dx || !gamma || !beta || (C2 > 0 && (!x2 || !dx2))) return IEF_EINVAL;
    if (B <= 0 || HW <= 0 || groups <= 0 || C1 <= 0 || C2 < 0 || (C1 + C2) % groups) return IEF_ESHAPE;
    hipLaunchKernelGGL(gn_bwd_f32_kernel, dim3(B * groups), dim3(256), 0, (hipStream_t)stream, x, x2, C1, C2, dy, add, dx, dx2,
                       gamma, beta, HW, groups, eps, silu);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm backward, one wave per row: g = dy gamma, dx = rstd (g - mean(g) - xh mean(g xh)) + add
__global__ __launch_bounds__(256) void layernorm_bwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                const float* __restrict__ add, float* __restrict__ dx,
                                                                const float* __restrict__ gamma, long long rows, int C, float eps) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* r = x + row * C;
    const float* d = dy + row * C;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int i = lane; i < C; i += 64) s += r[i];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int i = lane; i < C; i += 64) { const float t = r[i] - mean; q += t * t; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < C; i += 64) { const float g = d[i] * gamma[i]; s1 += g; s2 += g * (r[i] - mean) * rstd; }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
    for (int i = lane; i < C; i += 64) {
        const float h = (r[i] - mean) * rstd;
        float o = rstd * (d[i] * gamma[i] - m1 - h * m2);
        if (add) o += add[row * C + i];
        dx[row * C + i] = o;
    }
}
extern "C" int ief_layernorm_bwd_f32(const float* x, const float* dy, const float* add, float* dx, const float* gamma,
                                     long long rows, int C, float eps, void* stream) {
    if (!x || !dy || !dx || !gamma) return IEF_EINVAL;
    if (rows <= 0 || C <= 0) return IEF_ESHAPE;
    hipLaunchKernelGGL(layernorm_bwd_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, dy, add,
                       dx, gamma, rows, C, eps);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// GEGLU backward on the interleaved FF1 layout ([8 hidden | 8 gate] groups): d h = dy gelu(g), d g = dy h (Phi(g) + g phi(g))
__global__ __launch_bounds__(256) void geglu_il_bwd_f32_kernel(const float* __restrict__ pre, const float* __restrict__ dy,
                                                               float* __restrict__ dpre, long long n, int Ch) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long row = i / Ch;
        const int c = (int)(i - row * Ch);
        const long long o = row * 2 * Ch + (c >> 3) * 16 + (c & 7);
        const float hv = pre[o], gv = pre[o + 8], dv = dy[i];
        const float cdf = 0.5f * (1.0f + erff(gv * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * expf(-0.5f * gv * gv);
        dpre[o] = dv * gv * cdf;
        dpre[o + 8] = dv * hv * (cdf + gv * pdf);
    }
}
extern "C" int ief_geglu_il_bwd_f32(const float* pre, const float* dy, float* dpre, long long rows, int Ch, void* stream) {
    if (!pre || !dy || !dpre) return IEF_EINVAL;
    if (rows <= 0 || Ch <= 0 || (Ch & 7)) return IEF_ESHAPE;
    hipLaunchKernelGGL(geglu_il_bwd_f32_kernel, dim3(ewf_grid(rows * Ch)), dim3(256), 0, (hipStream_t)stream, pre, dy, dpre,
                       rows * Ch, Ch);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// stride-2 convolution data gradient helper: [B,H,W,C] -> [B,2H,2W,C], x at the even positions;
// nearest-2x upsample backward: [B,2H,2W,C] -> [B,H,W,C], sum of each 2x2 block
__global__ __launch_bounds__(256) void zero_insert2x_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H,
                                                                int W, int C4) {
    const long long total = (long long)B * 2 * H * 2 * W * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long long pix = i / C4;
        const int xo = (int)(pix % (2 * W)); pix /= 2 * W;
        const int yo = (int)(pix % (2 * H));
        const int b = (int)(pix / (2 * H));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!(xo & 1) && !(yo & 1)) v = ((const f32x4*)in)[(((long long)b * H + (yo >> 1)) * W + (xo >> 1)) * C4 + c];
        ((f32x4*)out)[i] = v;
    }
}
__global__ __launch_bounds__(256) void pool2x2_sum_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int H,
                                                              int W, int C4) {
    const long long total = (long long)B * H * W * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C4);
        long long pix = i / C4;
        const int xo = (int)(pix % W); pix /= W;
        const int yo = (int)(pix % H);
        const int b = (int)(pix / H);
        const f32x4* base = (const f32x4*)in + (((long long)b * 2 * H + 2 * yo) * 2 * W + 2 * xo) * C4 + c;
        ((f32x4*)out)[i] = (base[0] + base[C4]) + (base[2ll * W * C4] + base[2ll * W * C4 + C4]);
    }
}
extern "C" int ief_zero_insert2x_f32(const float* in, float* out, int B, int H, int W, int C, void* stream) {
    if (!in || !out) return IEF_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return IEF_ESHAPE;
    hipLaunchKernelGGL(zero_insert2x_f32_kernel, dim3(ewf_grid((long long)B * 4 * H * W * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, in, out, B, H, W, C / 4);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
extern "C" int ief_pool2x2_sum_f32(const float* in, float* out, int B, int H, int W, int C, void* stream) {
    if (!in || !out) return IEF_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return IEF_ESHAPE;
    hipLaunchKernelGGL(pool2x2_sum_f32_kernel, dim3(ewf_grid((long long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, in,
                       out, B, H, W, C / 4);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// conv_out data gradient with fp32 weights: dh[b][y][x][c] = sum_{co,ky,kx} d_eps[b][co][y+1-ky][x+1-kx] w[co][ky][kx][c]
__global__ __launch_bounds__(256) void conv_out_bwd_f32w_kernel(const float* __restrict__ de, const float* __restrict__ w,
                                                                float* __restrict__ dh, int B, int C, int H, int W, int Cout) {
    const int C4 = C >> 2;
    const long long total = (long long)B * H * W * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        long long pix = i / C4;
        const int xo = (int)(pix % W); pix /= W;
        const int yo = (int)(pix % H);
        const int b = (int)(pix / H);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
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
                    acc += g * *(const f32x4*)(w + (((long long)co * 3 + ky) * 3 + kx) * C + c4 * 4);
                }
            }
        ((f32x4*)dh)[i] = acc;
    }
}
extern "C" int ief_conv_out_bwd_f32w(const float* d_eps, const float* w, float* dh, int B, int C, int H, int W, int Cout,
                                     void* stream) {
    if (!d_eps || !w || !dh) return IEF_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || Cout <= 0 || Cout > 16) return IEF_ESHAPE;
    hipLaunchKernelGGL(conv_out_bwd_f32w_kernel, dim3(ewf_grid((long long)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       d_eps, w, dh, B, C, H, W, Cout);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// softmax backward on materialised maps, in place on dP: dS = scale P o (dP - sum_j dP_j P_j); one wave per row
__global__ __launch_bounds__(256) void softmax_bwd_rows_f32_kernel(const float* __restrict__ P, float* __restrict__ dP, long long rows,
                                                                   int L, float scale) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = P + row * L;
    float* d = dP + row * L;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int i = lane; i < L; i += 64) s += p[i] * d[i];
    s = wave_sum(s);
    for (int i = lane; i < L; i += 64) d[i] = scale * p[i] * (d[i] - s);
}
extern "C" int ief_softmax_bwd_rows_f32(const float* P, float* dP, long long rows, int L, float scale, void* stream) {
    if (!P || !dP) return IEF_EINVAL;
    if (rows <= 0 || L <= 0) return IEF_ESHAPE;
    hipLaunchKernelGGL(softmax_bwd_rows_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, dP, rows,
                       L, scale);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// [R][N][L] -> [R][L][N] through a 32 x 33 LDS tile (both sides coalesced)
__global__ __launch_bounds__(256) void transpose_batched_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int L) {
    __shared__ float tile[32][33];
    const long long r = blockIdx.z;
    const int n0 = blockIdx.y * 32, l0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + r * N * L;
    float* dst = out + r * N * L;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = n0 + ty + 8 * k, l = l0 + tx;
        if (n < N && l < L) tile[ty + 8 * k][tx] = src[(long long)n * L + l];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int l = l0 + ty + 8 * k, n = n0 + tx;
        if (l < L && n < N) dst[(long long)l * N + n] = tile[tx][ty + 8 * k];
    }
}
extern "C" int ief_transpose_batched_f32(const float* in, float* out, int R, int N, int L, void* stream) {
    if (!in || !out) return IEF_EINVAL;
    if (R <= 0 || N <= 0 || L <= 0 || R > 65535) return IEF_ESHAPE;
    hipLaunchKernelGGL(transpose_batched_f32_kernel, dim3((L + 31) / 32, (N + 31) / 32, R), dim3(256), 0, (hipStream_t)stream, in, out,
                       N, L);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// Pix2Pix-zero map objective on materialised maps (`/root/reference/pix2pix-zero/model/sd_utils.py:166-173`):
//   e = P - ref,  dP = gcoef e,  loss partial of this workgroup = loss_coef sum e^2   (fixed order: bit-reproducible)
#define MLR_ROWS 64          // map rows per workgroup
__global__ __launch_bounds__(256) void map_loss_rows_f32_kernel(const float* __restrict__ P, const float* __restrict__ ref,
                                                                float* __restrict__ dP, float* __restrict__ loss, long long rows,
                                                                int L, float gcoef, float loss_coef) {
    __shared__ float red[4];
    const long long r0 = (long long)blockIdx.x * MLR_ROWS;
    const long long n = (min(rows, r0 + MLR_ROWS) - r0) * L;
    const long long base = r0 * L;
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const float e = P[base + i] - ref[base + i];
        dP[base + i] = gcoef * e;
        s += e * e;
    }
    const float t = wg_sum(s, red);
    if (threadIdx.x == 0 && loss) loss[blockIdx.x] = loss_coef * t;
}
extern "C" int ief_map_loss_rows_blocks(long long rows) { return (int)((rows + MLR_ROWS - 1) / MLR_ROWS); }
extern "C" int ief_map_loss_rows_f32(const float* P, const float* ref, float* dP, float* loss, long long rows, int L, float gcoef,
                                     float loss_coef, void* stream) {
    if (!P || !ref || !dP) return IEF_EINVAL;
    if (rows <= 0 || L <= 0) return IEF_ESHAPE;
    hipLaunchKernelGGL(map_loss_rows_f32_kernel, dim3(ief_map_loss_rows_blocks(rows)), dim3(256), 0, (hipStream_t)stream, P, ref, dP,
                       loss, rows, L, gcoef, loss_coef);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---------------------------------------------------------------------------------------------------
// torch.optim.Adam (single-tensor form, no weight decay / amsgrad) on an fp32 gradient; g = grad * stats[1]; t = step[0] + 1
__global__ __launch_bounds__(256) void nti_adam_f32g_kernel(float* __restrict__ param, float* __restrict__ m, float* __restrict__ v,
                                                            const float* __restrict__ grad, const float* __restrict__ stats,
                                                            const float* __restrict__ hyper, const int* __restrict__ step, int n) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
    const float t = (float)(step[0] + 1);
    const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
    const float step_size = lr / bc1, bc2s = sqrtf(bc2);
    const float factor = stats[1];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float g = grad[i] * factor;
        const float mi = b1 * m[i] + (1.0f - b1) * g;
        const float vi = b2 * v[i] + (1.0f - b2) * (g * g);
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2s + eps;
        param[i] = param[i] - step_size * (mi / denom);
    }
}
__global__ void nti_step_inc_f32g_kernel(int* step) { step[0] += 1; }
extern "C" int ief_nti_adam_f32g(float* param, float* m, float* v, const float* grad, const float* stats, const float* hyper,
                                 int* step, int n, void* stream) {
    if (!param || !m || !v || !grad || !stats || !hyper || !step) return IEF_EINVAL;
    if (n <= 0) return IEF_ESHAPE;
    hipLaunchKernelGGL(nti_adam_f32g_kernel, dim3(ewf_grid(n)), dim3(256), 0, (hipStream_t)stream, param, m, v, grad, stats, hyper,
                       step, n);
    IEF_LAUNCH_CHECK();
    hipLaunchKernelGGL(nti_step_inc_f32g_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
