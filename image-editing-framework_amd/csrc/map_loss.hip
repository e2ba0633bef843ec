// Pix2Pix-zero cross-attention guidance for gfx950: objective and its gradient w.r.t. the queries.
//
// Reference (/root/reference/pix2pix-zero/model/sd_utils.py:166-173): per cross-attention module
//     loss += ((curr - ref) ** 2).sum((1, 2)).mean(0)        curr, ref: softmax maps [B*heads, N, 77]
// then `loss.backward()` down to the UNet INPUT.  Keys / values come from the (constant) text context, so the only
// path from this term into the network is through the queries:
//     P = softmax(scale q K^T)      e = P - ref      dP = 2 e / (B heads)
//     dS = P * (dP - sum_j dP_j P_j)                  dq = scale * dS K
// With <= 96 keys a query row is tiny: G = D / DC adjacent LANES own one query of one (batch, head), each a DC-wide slice of
// the head dim (DC = 40 or 32: G = 1, 2, 4 for D = 40, 80, 160 — the per-lane work is the same for every head dim, and
// the low-resolution levels, where D is large and queries are few, still spread over enough waves); K of that
// (batch, head) sits in LDS and is read as a broadcast, the scores are completed by a butterfly over the G lanes, every
// lane keeps the 77 scores / dS in a private LDS column (S[j][lane]: conflict free), and writes its own slice of dq.
// No map is written anywhere; the reference maps are the fp16 maps `ief_attn_probs_f16` recorded during the reference pass.  dq is ADDED to the gradient already sitting in dQ
// (the softmax-backward of the value path) when `accumulate` is set.  Loss partials: one float per workgroup, each summed
// in a fixed order (deterministic); the host adds them.
#include "ief_common.h"
#include "ief_params.h"

template <int D>
__global__ __launch_bounds__(256) void attn_map_loss_bwd_kernel(const IefMapLossParams p) {
    constexpr int DC = (D % 40 == 0) ? 40 : 32;      // register chunk of the head dim
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    half_t* Ks = (half_t*)smem_raw;                                   // [L][D]
    float* S = (float*)(smem_raw + (((size_t)p.L * D * 2 + 15) & ~(size_t)15));   // [L][256]
    __shared__ float red[4];
    constexpr int G = D / DC;                         // lanes per query
    const int tid = threadIdx.x;
    const int head = blockIdx.y, b = blockIdx.z;
    const int n = (blockIdx.x * 256 + tid) / G;
    const int part_id = tid % G;                      // which DC-wide slice of the head dim this lane owns
    const bool live = n < p.N;
    // K of this (batch, head): L x D halves, 16-byte pieces
    constexpr int D8 = D / 8;
    for (int c = tid; c < p.L * D8; c += 256) {
        const int j = c / D8, c8 = c - j * D8;
        *(half8*)(Ks + j * D + c8 * 8) = *(const half8*)(p.K + ((long long)b * p.L + j) * p.ldk + head * D + c8 * 8);
    }
    __syncthreads();
    float part = 0.f;
    {
        const int nq = live ? n : p.N - 1;            // dead lanes shadow the last query (they take part in the butterfly)
        const int c0 = part_id * DC;
        const half_t* q = p.Q + ((long long)b * p.N + nq) * p.ldq + head * D + c0;
        float* s = S + tid;
        float qv[DC];
#pragma unroll
        for (int c = 0; c < DC; c += 8) {
            const half8 h = *(const half8*)(q + c);
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[c + e] = (float)h[e];
        }
        float mx = -3.0e38f;
        for (int j = 0; j < p.L; ++j) {
            const half_t* kr = Ks + j * D + c0;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < DC; c += 8) {
                const half8 h = *(const half8*)(kr + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) a += qv[c + e] * (float)h[e];
            }
            if (G >= 2) a += __shfl_xor(a, 1);
            if (G >= 4) a += __shfl_xor(a, 2);
            s[j * 256] = a;
            mx = fmaxf(mx, a);
        }
        float sum = 0.f;
        for (int j = 0; j < p.L; ++j) {
            const float e = __expf((s[j * 256] - mx) * p.scale);
            s[j * 256] = e;
            sum += e;
        }
        const float inv = 1.f / sum;
        const half_t* rf = p.ref + (((long long)b * p.heads + head) * p.N + nq) * p.L;
        float dot = 0.f;
        for (int j = 0; j < p.L; ++j) {
            const float P = s[j * 256] * inv;
            const float e = P - (float)rf[j];
            part += e * e;
            dot += e * P;
            s[j * 256] = P;
        }
        for (int j = 0; j < p.L; ++j) {
            const float P = s[j * 256];
            const float e = P - (float)rf[j];
            s[j * 256] = P * (e - dot);                 // dS up to the common factor
        }
        if (!live || part_id != 0) part = 0.f;          // one lane per query carries the objective
        // this lane's slice of dq = gcoef * scale * dS K
        float acc[DC];
#pragma unroll
        for (int c = 0; c < DC; ++c) acc[c] = 0.f;
        for (int j = 0; j < p.L; ++j) {
            const float w = s[j * 256];
            const half_t* kr = Ks + j * D + c0;
#pragma unroll
            for (int c = 0; c < DC; c += 8) {
                const half8 h = *(const half8*)(kr + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[c + e] += w * (float)h[e];
            }
        }
        if (live) {
            half_t* dq = p.dQ + ((long long)b * p.N + n) * p.lddq + head * D + c0;
            const float g = p.gcoef * p.scale;
#pragma unroll
            for (int c = 0; c < DC; c += 8) {
                half8 o;
                if (p.accumulate) {
                    const half8 old = *(const half8*)(dq + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)fminf(fmaxf((float)old[e] + g * acc[c + e], -65504.f), 65504.f);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)fminf(fmaxf(g * acc[c + e], -65504.f), 65504.f);
                }
                *(half8*)(dq + c) = o;
            }
        }
    }
    // loss partial of this workgroup, fixed order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0 && p.loss)
        p.loss[((long long)b * p.heads + head) * gridDim.x + blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * p.loss_coef;
}

extern "C" int ief_map_loss_blocks(int N, int d) {
    if (N <= 0 || d <= 0) return 0;
    const int G = d / ((d % 40 == 0) ? 40 : 32);
    return (int)(((long long)N * G + 255) / 256);
}

extern "C" int ief_attn_map_loss_bwd_f16(const IefMapLossParams* pp, void* stream) {
    if (!pp) return IEF_EINVAL;
    const IefMapLossParams p = *pp;
    if (!p.Q || !p.K || !p.ref || !p.dQ) return IEF_EINVAL;
    if (p.B <= 0 || p.heads <= 0 || p.N <= 0 || p.L <= 0 || p.L > 96) return IEF_ESHAPE;
    if ((p.ldq & 7) || (p.ldk & 7) || (p.lddq & 7)) return IEF_EALIGN;
    if (p.heads * p.d > p.ldq || p.heads * p.d > p.ldk || p.heads * p.d > p.lddq) return IEF_ESHAPE;
    const size_t lds = (((size_t)p.L * p.d * 2 + 15) & ~(size_t)15) + (size_t)p.L * 256 * 4;
    const int G = p.d / ((p.d % 40 == 0) ? 40 : 32);          // lanes per query, as in the kernel
    dim3 grid(((long long)p.N * G + 255) / 256, p.heads, p.B);
    hipStream_t st = (hipStream_t)stream;
#define ML_LAUNCH(DD)                                                                                              \
    {                                                                                                              \
        hipError_t e = hipFuncSetAttribute((const void*)attn_map_loss_bwd_kernel<DD>,                              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
        if (e != hipSuccess) return (int)e;                                                                        \
        hipLaunchKernelGGL((attn_map_loss_bwd_kernel<DD>), grid, dim3(256), lds, st, p);                           \
    }
    switch (p.d) {
        case 32: ML_LAUNCH(32); break;
        case 40: ML_LAUNCH(40); break;
        case 64: ML_LAUNCH(64); break;
        case 80: ML_LAUNCH(80); break;
        case 160: ML_LAUNCH(160); break;
        default: return IEF_ESHAPE;
    }
#undef ML_LAUNCH
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// y[i] += a * x[i]  (fp32): the plain SGD step on the UNet input (sd_utils.py:160,174), a = -lr / gradient scale
__global__ __launch_bounds__(256) void axpy_f32_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] += a * x[i];
}
extern "C" int ief_axpy_f32(float* y, const float* x, float a, long long n, void* stream) {
    if (!y || !x) return IEF_EINVAL;
    if (n <= 0) return IEF_ESHAPE;
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(axpy_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
