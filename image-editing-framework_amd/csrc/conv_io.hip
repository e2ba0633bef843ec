// Boundary convolutions of the UNet: they also carry the layout / dtype change between the
// sampler's fp32 NCHW latents and the network's fp16 NHWC activations, so no separate
// transpose or cast kernel runs at either end.
//   conv_in : [B,Cin<=8,H,W] fp32 NCHW -> [B,H,W,Cout] fp16        (K = 9*Cin, too thin for MFMA)
//   conv_out: [B,H,W,C] fp16 -> [B,Cout<=8,H,W] fp32 NCHW           (N = Cout, too thin for MFMA)
// Both are small (0.4 GFLOP at 64x64, B=4) and bandwidth/latency-bound.
#include "ief_common.h"
#include "ief_params.h"

#define CIN_MAX 8
#define COUT_MAX 8

// One block = CI_PIX consecutive output pixels x all Cout channels.  Weights ([9*Cin][Cout] fp16 -> fp32) and the im2col'ed input rows ([CI_PIX][9*Cin] fp32) are staged in LDS once; each thread
// then produces 8 output channels for several pixels: 9*Cin broadcast reads + 8-wide LDS weight reads.
#define CI_PIX 32
#define CI_CMAX 256
__global__ __launch_bounds__(256) void conv_in_kernel(const float* __restrict__ x, const half_t* __restrict__ w,
                                                      const float* __restrict__ bias, half_t* __restrict__ out,
                                                      int B, int Cin, int H, int Wd, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float smem_ci[];
    const int K = 9 * Cin;
    // output channels are processed in slices of Cc (grid.y) so the fp32 weight slice fits LDS for any Cout
    const int Cfull = Cout;
    const int Cc = Cfull / gridDim.y, co0 = blockIdx.y * Cc;   // host picks gridDim.y | Cout with Cc % 8 == 0, Cc <= CI_CMAX
    Cout = Cc;
    float* wl = smem_ci;                 // [K][Cc]
    float* xin = smem_ci + K * Cc;       // [CI_PIX][K]
    for (int i = threadIdx.x; i < K * (Cout >> 3); i += 256) {   // w is [3][3][Cin][Cfull]: k-major; 8 halves per load (Cc % 8 == 0)
        const int k = i / (Cout >> 3), c = (i - k * (Cout >> 3)) << 3;
        const half8 v = *(const half8*)(w + (long long)k * Cfull + co0 + c);
        float* d = wl + k * Cc + c;
        *(f32x4*)d = (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        *(f32x4*)(d + 4) = (f32x4){(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    }
    const long long total = (long long)B * H * Wd;
    const long long pix0 = (long long)blockIdx.x * CI_PIX;
    for (int i = threadIdx.x; i < CI_PIX * K; i += 256) {
        const int pl = i / K, k = i - pl * K;
        const long long pix = pix0 + pl;
        float v = 0.f;
        if (pix < total) {
            const int b = (int)(pix / (H * Wd));
            const int rem = (int)(pix - (long long)b * H * Wd);
            const int oy = rem / Wd, ox = rem - oy * Wd;
            const int tap = k / Cin, ci = k - tap * Cin;
            const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd)
                v = x[(((long long)b * Cin + ci) * H + iy) * Wd + ix];
        }
        xin[i] = v;
    }
    __syncthreads();
    const int C8 = Cout >> 3;
    for (int i = threadIdx.x; i < CI_PIX * C8; i += 256) {
        const int pl = i / C8, c8 = i - pl * C8;
        const long long pix = pix0 + pl;
        if (pix >= total) continue;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[co0 + c8 * 8 + e] : 0.f;
        const float* xr = xin + pl * K;
        for (int k = 0; k < K; ++k) {
            const float v = xr[k];
            const f32x4 w0 = *(const f32x4*)(wl + k * Cc + c8 * 8), w1 = *(const f32x4*)(wl + k * Cc + c8 * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += v * w0[e]; acc[4 + e] += v * w1[e]; }
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)acc[e];
        *(half8*)(out + pix * Cfull + co0 + c8 * 8) = o;
    }
}

extern "C" int ief_conv_in_f32(const float* x, const ief_half* w, const float* bias, ief_half* out,
                               int B, int Cin, int H, int Wd, int Cout, void* stream) {
    if (!x || !w || !out) return IEF_EINVAL;
    if (B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || Cin > CIN_MAX || Cout <= 0 || (Cout & 7) || Cout > 2048) return IEF_ESHAPE;
    int nsl = (Cout + CI_CMAX - 1) / CI_CMAX;   // fewest equal slices of <= CI_CMAX channels, each a multiple of 8
    while (nsl <= Cout / 8 && (Cout % nsl || ((Cout / nsl) & 7))) ++nsl;
    if (nsl > Cout / 8) return IEF_ESHAPE;
    const int Cc = Cout / nsl;
    const size_t lds = ((size_t)9 * Cin * Cc + (size_t)CI_PIX * 9 * Cin) * sizeof(float);
    if (lds > 64 * 1024) return IEF_ESHAPE;
    const long long total = (long long)B * H * Wd;
    hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)((total + CI_PIX - 1) / CI_PIX), Cout / Cc), dim3(256), lds,
                       (hipStream_t)stream, x, w, bias, out, B, Cin, H, Wd, Cout);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// 16 lanes per output pixel split the 9*C reduction (8-channel chunks round-robin), shuffle-reduce.  A workgroup stages the
// weights ([Cout][9][C] fp16, 23 KB for SD's 320 -> 4) in LDS once and walks CO_PPG pixels per 16-lane group; per pixel the
// activation chunks of a whole tap ROW (3 taps x up to CO_CH chunks) are requested before the first is used — the earlier
// form issued one dependent load per 8 channels, 27 memory round trips per pixel (49 us for 0.4 GFLOP) — and the products
// run on v_dot2_f32_f16 (fp16 pairs, fp32 accumulation).
typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
template <int CH>            // chunks of 8 channels per lane and tap: C <= 16 * 8 * CH
__global__ __launch_bounds__(256) void conv_out_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int B, int C, int H, int Wd, int Cout) {
    extern __shared__ __attribute__((aligned(16))) half_t wl[];          // [Cout][9][C]
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long long total = (long long)B * H * Wd;
    const int C8 = C >> 3;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long pix = (long long)blockIdx.x * 16 + grp;
    const bool live = pix < total;
    const long long pc = live ? pix : 0;
    const int b = (int)(pc / (H * Wd));
    const int rem = (int)(pc - (long long)b * H * Wd);
    const int oy = rem / Wd, ox = rem - oy * Wd;
    // every activation chunk this lane needs (9 taps x CH chunks) is requested before the weights are staged
    half8 v[9][CH];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
        const bool ok = live && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)Wd;
        const half_t* xp = x + (((long long)b * H + (ok ? iy : 0)) * Wd + (ok ? ix : 0)) * C;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c8 = sub + 16 * j;
            v[tap][j] = (ok && c8 < C8) ? *(const half8*)(xp + c8 * 8) : zero8;
        }
    }
    const int nw8 = (Cout * 9 * C) >> 3;
    for (int i = threadIdx.x; i < nw8; i += 256) ((half8*)wl)[i] = ((const half8*)w)[i];
    __syncthreads();
    float acc[COUT_MAX];
#pragma unroll
    for (int o = 0; o < COUT_MAX; ++o) acc[o] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c8 = sub + 16 * j;
            if (c8 < C8) {
#pragma unroll
                for (int o = 0; o < COUT_MAX; ++o) {
                    if (o < Cout) {
                        const half8 wv = *(const half8*)(wl + ((o * 9 + tap) * C + c8 * 8));
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[o] = __builtin_amdgcn_fdot2((h2_t){v[tap][j][2 * e], v[tap][j][2 * e + 1]},
                                                            (h2_t){wv[2 * e], wv[2 * e + 1]}, acc[o], false);
                    }
                }
            }
        }
#pragma unroll
    for (int o = 0; o < COUT_MAX; ++o) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc[o] += __shfl_xor(acc[o], off);
    }
    if (live && sub == 0) {
        for (int o = 0; o < Cout; ++o)
            out[(((long long)b * Cout + o) * H + oy) * Wd + ox] = acc[o] + (bias ? bias[o] : 0.f);
    }
}

// fallback for wide inputs (C > 512) or weight sets past the LDS budget: the plain loop, one chunk at a time
__global__ __launch_bounds__(256) void conv_out_wide_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int B, int C, int H, int Wd, int Cout) {
    const int sub = threadIdx.x & 15;
    const long long pix = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long long total = (long long)B * H * Wd;
    const bool live = pix < total;
    const long long pc = live ? pix : 0;
    const int b = (int)(pc / (H * Wd));
    const int rem = (int)(pc - (long long)b * H * Wd);
    const int oy = rem / Wd, ox = rem - oy * Wd;
    const int C8 = C >> 3;
    float acc[COUT_MAX];
#pragma unroll
    for (int o = 0; o < COUT_MAX; ++o) acc[o] = 0.f;
    if (live) {
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)Wd) continue;
            const half_t* xp = x + (((long long)b * H + iy) * Wd + ix) * C;
            for (int c8 = sub; c8 < C8; c8 += 16) {
                const half8 v = *(const half8*)(xp + c8 * 8);
#pragma unroll
                for (int o = 0; o < COUT_MAX; ++o) {
                    if (o < Cout) {
                        const half8 wv = *(const half8*)(w + ((long long)o * 9 + tap) * C + c8 * 8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[o] += (float)v[e] * (float)wv[e];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < COUT_MAX; ++o) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc[o] += __shfl_xor(acc[o], off);
    }
    if (live && sub == 0) {
        for (int o = 0; o < Cout; ++o)
            out[(((long long)b * Cout + o) * H + oy) * Wd + ox] = acc[o] + (bias ? bias[o] : 0.f);
    }
}

extern "C" int ief_conv_out_f32(const ief_half* x, const ief_half* w, const float* bias, float* out,
                                int B, int C, int H, int Wd, int Cout, void* stream) {
    if (!x || !w || !out) return IEF_EINVAL;
    if (B <= 0 || H <= 0 || Wd <= 0 || C <= 0 || (C & 7) || Cout <= 0 || Cout > COUT_MAX) return IEF_ESHAPE;
    const long long total = (long long)B * H * Wd;
    const size_t lds = (size_t)Cout * 9 * C * sizeof(half_t);
    const dim3 grid((unsigned)((total + 15) / 16));
    if (C <= 128 && lds <= 48 * 1024)
        hipLaunchKernelGGL(conv_out_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias, out, B, C, H, Wd, Cout);
    else if (C <= 256 && lds <= 48 * 1024)
        hipLaunchKernelGGL(conv_out_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias, out, B, C, H, Wd, Cout);
    else if (C <= 384 && lds <= 48 * 1024)
        hipLaunchKernelGGL(conv_out_kernel<3>, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias, out, B, C, H, Wd, Cout);
    else
        hipLaunchKernelGGL(conv_out_wide_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                           out, B, C, H, Wd, Cout);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
