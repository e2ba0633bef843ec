// Small latency-bound kernels of the sampler loop: CFG + DDIM update, timestep embedding,
// SiLU, casts, and the per-step table select that lets ONE captured graph serve all 50 steps.
#include "ief_common.h"
#include "ief_params.h"

// eps = eps_u + g (eps_c - eps_u);  x0 = (x - sqrt(1-a_f) eps) / sqrt(a_f);  x' = sqrt(a_t) x0 + sqrt(1-a_t) eps
// Same operation order as the reference's scheduler.step / ddim_reverse so fp32 results stay
// within an ulp or two of the eager formula (/root/reference/p2p/model/sd_utils.py:74-76,
// /root/reference/p2p/inversion/ddim.py:14-17).
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const float* __restrict__ eu, const float* __restrict__ ec,
                                                       const float* __restrict__ x, float* __restrict__ xo,
                                                       float* __restrict__ x0o, const float* __restrict__ coef,
                                                       long long n) {
    const float a_f = coef[0], a_t = coef[1], g = coef[2];
    const float sb_f = sqrtf(1.0f - a_f), sa_f = sqrtf(a_f), sa_t = sqrtf(a_t), sb_t = sqrtf(1.0f - a_t);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float e = ec[i];
        if (eu) { const float u = eu[i]; e = u + g * (e - u); }
        const float x0 = (x[i] - sb_f * e) / sa_f;
        if (x0o) x0o[i] = x0;
        xo[i] = sa_t * x0 + sb_t * e;
    }
}

extern "C" int ief_cfg_ddim_step_f32(const float* eps_u, const float* eps_c, const float* x, float* x_out,
                                     float* x0_out, const float* coef, long long n, void* stream) {
    if (!eps_c || !x || !x_out || !coef) return IEF_EINVAL;
    if (n <= 0) return IEF_ESHAPE;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, eps_u, eps_c, x, x_out, x0_out,
                       coef, n);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) [ext]: [cos | sin]
__global__ void timestep_embedding_kernel(const float* __restrict__ t, half_t* __restrict__ out, int B, int dim) {
    const int half_dim = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half_dim) return;
    const int b = i / half_dim, k = i - b * half_dim;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half_dim);  // ln(10000)
    const float a = t[b] * freq;
    out[(long long)b * dim + k] = (half_t)cosf(a);
    out[(long long)b * dim + half_dim + k] = (half_t)sinf(a);
}

extern "C" int ief_timestep_embedding_f16(const float* t, ief_half* out, int B, int dim, void* stream) {
    if (!t || !out) return IEF_EINVAL;
    if (B <= 0 || dim <= 0 || (dim & 1)) return IEF_ESHAPE;
    const int n = B * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, B, dim);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

__global__ __launch_bounds__(256) void silu_kernel(const half_t* __restrict__ x, half_t* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (half_t)silu_f((float)x[i]);
}
extern "C" int ief_silu_f16(const ief_half* x, ief_half* out, long long n, void* stream) {
    if (!x || !out) return IEF_EINVAL;
    if (n <= 0) return IEF_ESHAPE;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(silu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, n);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// one block per row; row kept in registers (L <= 256 * 8 * SM_MAXCH)
#define SM_MAXCH 8
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* __restrict__ x, int L) {
    half_t* row = x + (long long)blockIdx.x * L;
    const int L8 = L >> 3;
    half8 v[SM_MAXCH];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < SM_MAXCH; ++i) {
        const int c8 = threadIdx.x + 256 * i;
        if (c8 < L8) {
            v[i] = *(const half8*)(row + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[i][e]);
        }
    }
    __shared__ float red[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    float ev[SM_MAXCH][8];
#pragma unroll
    for (int i = 0; i < SM_MAXCH; ++i) {
        const int c8 = threadIdx.x + 256 * i;
        if (c8 < L8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { ev[i][e] = __expf((float)v[i][e] - mx); s += ev[i][e]; }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int i = 0; i < SM_MAXCH; ++i) {
        const int c8 = threadIdx.x + 256 * i;
        if (c8 < L8) {
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(ev[i][e] * inv);
            *(half8*)(row + c8 * 8) = o;
        }
    }
}
extern "C" int ief_softmax_rows_f16(ief_half* x, int rows, int L, void* stream) {
    if (!x) return IEF_EINVAL;
    if (rows <= 0 || L <= 0 || (L & 7) || L > 256 * 8 * SM_MAXCH) return IEF_ESHAPE;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, L);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// 32x32 tiles through LDS (padded): coalesced reads and writes
__global__ __launch_bounds__(256) void transpose_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int R, int C) {
    __shared__ half_t tile[32][34];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) tile[j][tx] = in[(long long)(r0 + j) * C + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) out[(long long)(c0 + j) * R + r0 + tx] = tile[tx][j];
}
extern "C" int ief_transpose_f16(const ief_half* in, ief_half* out, int R, int C, void* stream) {
    if (!in || !out) return IEF_EINVAL;
    if (R <= 0 || C <= 0) return IEF_ESHAPE;
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, out, R, C);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

__global__ __launch_bounds__(256) void pointwise_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int B, int Cin, int Cout, int HW) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * HW) return;
    const int b = (int)(i / HW), p = (int)(i - (long long)b * HW);
    float xin[8];
    for (int c = 0; c < Cin; ++c) xin[c] = x[((long long)b * Cin + c) * HW + p];
    for (int o = 0; o < Cout; ++o) {
        float a = bias ? bias[o] : 0.f;
        for (int c = 0; c < Cin; ++c) a += w[o * Cin + c] * xin[c];
        out[((long long)b * Cout + o) * HW + p] = a;
    }
}
extern "C" int ief_pointwise_f32(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout,
                                 int HW, void* stream) {
    if (!x || !w || !out) return IEF_EINVAL;
    if (B <= 0 || HW <= 0 || Cin <= 0 || Cin > 8 || Cout <= 0 || Cout > 8) return IEF_ESHAPE;
    const long long n = (long long)B * HW;
    hipLaunchKernelGGL(pointwise_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                       out, B, Cin, Cout, HW);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

__global__ __launch_bounds__(256) void add_f16_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                      half_t* __restrict__ o, long long n8, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const half8 x = ((const half8*)a)[i], y = ((const half8*)b)[i];
        half8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (half_t)((float)x[e] + (float)y[e]);
        ((half8*)o)[i] = r;
    }
    if (blockIdx.x == 0) for (long long i = n8 * 8 + threadIdx.x; i < n; i += 256) o[i] = (half_t)((float)a[i] + (float)b[i]);
}
extern "C" int ief_add_f16(const ief_half* a, const ief_half* b, ief_half* out, long long n, void* stream) {
    if (!a || !b || !out) return IEF_EINVAL;
    if (n <= 0) return IEF_ESHAPE;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) return IEF_EALIGN;
    const long long n8 = n / 8;
    int grid = (int)((n8 + 255) / 256);
    if (grid < 1) grid = 1;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(add_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, out, n8, n);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// out[b][:] = in[src[b]][:]  (batch-row gather; Plug-and-Play's feature injection copies the source image's
// ResnetBlock2D features over the edited rows: /root/reference/pnp/model/register.py:161-166)
__global__ __launch_bounds__(256) void gather_rows_kernel(const half_t* __restrict__ in, half_t* __restrict__ out,
                                                          const int* __restrict__ src, int B, long long row8) {
    const long long total = (long long)B * row8;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / row8);
        const long long c = i - (long long)b * row8;
        ((half8*)out)[i] = ((const half8*)in)[(long long)src[b] * row8 + c];
    }
}
extern "C" int ief_gather_rows_f16(const ief_half* in, ief_half* out, const int* src, int B, long long row_elems, void* stream) {
    if (!in || !out || !src) return IEF_EINVAL;
    if (B <= 0 || row_elems <= 0 || (row_elems & 7)) return IEF_ESHAPE;
    if (((uintptr_t)in | (uintptr_t)out) & 15) return IEF_EALIGN;
    const long long total = (long long)B * (row_elems / 8);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, out, src, B, row_elems / 8);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float* __restrict__ x, half_t* __restrict__ o, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) o[i] = (half_t)x[i];
}
__global__ __launch_bounds__(256) void cast_f16_f32_kernel(const half_t* __restrict__ x, float* __restrict__ o, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) o[i] = (float)x[i];
}
extern "C" int ief_cast_f32_to_f16(const float* x, ief_half* out, long long n, void* stream) {
    if (!x || !out) return IEF_EINVAL;
    if (n <= 0) return IEF_ESHAPE;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, n);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
extern "C" int ief_cast_f16_to_f32(const ief_half* x, float* out, long long n, void* stream) {
    if (!x || !out) return IEF_EINVAL;
    if (n <= 0) return IEF_ESHAPE;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, n);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// out[:] = table[step[0]][:]   (4-byte words).  Step-dependent inputs of the UNet (time-embedding
// projections, P2P gate coefficients, self-replace source maps, DDIM alphas) live in per-step
// tables; this copy runs INSIDE the captured graph and reads the step index from device memory.
// The row index is clamped to [0, n_rows - 1]: a replay past the end of a schedule re-reads the last row (tables keep an
// identity / final row there) instead of reading beyond the allocation.
__global__ __launch_bounds__(256) void select_step_kernel(const uint32_t* __restrict__ table, uint32_t* __restrict__ out,
                                                          const int* __restrict__ step, long long words, int n_rows) {
    int row = step[0];
    row = row < 0 ? 0 : (row >= n_rows ? n_rows - 1 : row);
    const long long base = (long long)row * words;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < words; i += (long long)gridDim.x * 256)
        out[i] = table[base + i];
}
extern "C" int ief_select_step(const void* table, void* out, const int* step, long long bytes_per_step, int n_rows,
                               void* stream) {
    if (!table || !out || !step) return IEF_EINVAL;
    if (n_rows <= 0) return IEF_ESHAPE;
    if (bytes_per_step <= 0 || (bytes_per_step & 3)) return IEF_EALIGN;
    const long long words = bytes_per_step / 4;
    int grid = (int)((words + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(select_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)table,
                       (uint32_t*)out, step, words, n_rows);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

__global__ void advance_step_kernel(int* step) { if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1; }
extern "C" int ief_advance_step(int* step, void* stream) {
    if (!step) return IEF_EINVAL;
    hipLaunchKernelGGL(advance_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

extern "C" int ief_abi_version(void) { return IEF_ABI_VERSION; }
// sizeof of the parameter structs as compiled, so a binding can verify its own layout (0: gemm, 1: attn, 2: cross)
extern "C" int ief_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(IefGemmParams);
        case 1: return (int)sizeof(IefAttnParams);
        case 2: return (int)sizeof(IefCrossParams);
        case 3: return (int)sizeof(IefAttnBwdParams);
        case 4: return (int)sizeof(IefMapLossParams);
        case 5: return (int)sizeof(IefGemmF32Params);
        case 6: return (int)sizeof(IefAttnF32Params);
        case 7: return (int)sizeof(IefGemmX3pParams);
        default: return -1;
    }
}
extern "C" const char* ief_target_arch(void) { return "gfx950"; }
