// Shared device helpers for the gfx950 kernels of the attention-controlled denoising path.
// fp16 storage, fp32 accumulation (MFMA f32 <- f16 x f16), 64-wide wavefronts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define IEF_OK 0
#define IEF_EINVAL (-1)
#define IEF_ESHAPE (-2)
#define IEF_EALIGN (-3)

#define IEF_LAUNCH_CHECK()                          \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU, as torch.nn.functional.gelu default
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// erf without branches: both ranges evaluated, one select (the device library's erff branches per lane and costs ~3x the
// instructions).  |a| > 475/512: 1 - exp(p(|a|)), below: a + a q(a^2); both minimax fits are within 1 ulp of erf, the exp is the
// hardware exp2 (its error enters times exp(p) <= 0.4): max abs error 7.6e-8 over [-6, 6] (checked against math.erf on 3e5 points)
__device__ __forceinline__ float erf_nb(float a) {
    const float t = fminf(fabsf(a), 4.0f), s = a * a;
    float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
    const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
    r = fmaf(r, t * t, u);
    r = fmaf(r, t, -1.06777877e-1f);
    r = fmaf(r, t, -6.34846687e-1f);
    r = fmaf(r, t, -1.28717512e-1f);
    r = fmaf(r, t, -t);
    const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.4426950408889634f), a);
    float q = -5.96761703e-4f;
    q = fmaf(q, s, 4.99119423e-3f);
    q = fmaf(q, s, -2.67681349e-2f);
    q = fmaf(q, s, 1.12819925e-1f);
    q = fmaf(q, s, -3.76125336e-1f);
    q = fmaf(q, s, 1.28379166e-1f);
    q = fmaf(q, a, a);
    return t > 0.927734375f ? big : q;
}
__device__ __forceinline__ float gelu_nb(float x) { return 0.5f * x * (1.0f + erf_nb(x * 0.70710678118654752f)); }

// XCD-aware remap of a linear block id: blocks b and b+8 share an XCD (and its L2), so hand
// each XCD a CONTIGUOUS run of logical ids.  Bijective for any grid size (guide §5 T1).
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}
