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
    for (int i = threadIdx.x; i < K * Cout; i += 256) {   // w is [3][3][Cin][Cfull]: k-major
        const int k = i / Cout, c = i - k * Cout;
        wl[k * Cc + c] = (float)w[(long long)k * Cfull + co0 + c];
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

// 16 lanes per output pixel split the 9*C reduction (8-channel chunks round-robin), shuffle-reduce.
__global__ __launch_bounds__(256) void conv_out_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
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
    hipLaunchKernelGGL(conv_out_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                       out, B, C, H, Wd, Cout);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
