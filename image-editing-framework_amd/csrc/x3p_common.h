// Shared pieces of the planes kernels (gemm_x3p.hip, conv_halo_x3p.hip): LDS-DMA piece, LDS swizzle, the epilogue of one block.
#pragma once
#include "ief_common.h"
#include "ief_params.h"
#include "x3_common.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
__device__ __forceinline__ void glds16(const char* g, half_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

#define XP_BK 32
#ifndef XP_ABL
#define XP_ABL 0        // timing-only ablation builds (outputs wrong): 1 no LDS-DMA in the loop, 2 no fragment reads, 4 one MFMA per block, 8 no barrier
#endif
#define XP_GROUP_M 8
#if XP_ABL & 8
#define XP_BARRIER() do {} while (0)
#else
#define XP_BARRIER() asm volatile("s_barrier" ::: "memory")
#endif
// LDS rows are 64 B (one 32-deep K tile of one plane); the slot (16-byte unit) of chunk c of row r is c ^ xp_swz(r).  With
// xp_swz(r) = 2 * bit 2 of r the ds_read_b128 fragments of 16 CONSECUTIVE rows starting at ANY row are free of bank conflicts
// (brute-forced over the instruction's four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; the tap shift of the
// halo convolution moves the start).  An LDS-DMA piece is lane-linear (lane l -> row l >> 2 of 16, slot l & 3), so the swizzle
// is applied to the SOURCE chunk a lane fetches: XP_LANE_CHUNK(l), the same for every piece (pieces start at multiples of 16 rows)
__device__ __forceinline__ int xp_swz(int row) { return (row >> 1) & 2; }
#define XP_LANE_CHUNK(lane) (((lane) & 3) ^ (((lane) >> 3) & 2))

// the epilogue of one 16 x 16 block held as f32x4 per lane (row m, columns n .. n + 3): bias / row vector / residual, fp32
// and / or plane stores
// returns the value as stored (what the consumer will read): the row / column statistics are taken on it
__device__ __forceinline__ f32x4 xp_store(const IefGemmX3pParams& p, f32x4 v, int m, int n) {
    if (p.bias) v += *(const f32x4*)(p.bias + n);
    if (p.rowvec) v += *(const f32x4*)(p.rowvec + (long long)(m / p.rows_per_batch) * p.N + n);
    if (p.residual) v += *(const f32x4*)(p.residual + (long long)m * p.ldr + n);
    v = v * p.out_scale;
    if (p.Out) *(f32x4*)(p.Out + (long long)m * p.ldo + n) = v;
    if (p.OutP) {
        half4 h, l;
        split4(v, 1.0f, h, l);
        half_t* o = p.OutP + (long long)m * p.ldp + n;
        *(half4*)o = h;
        *(half4*)(o + p.planeO) = l;
    }
    return v;
}

// ---- LayerNorm folded into the consumer GEMM (IefGemmX3pParams.rstat_*): (mean, rstd) of row m of the launch's A operand from
// the producer's per-slice (mean, M2) partials, merged Chan-style in slice order (deterministic)
__device__ __forceinline__ void xp_ln_row(const IefGemmX3pParams& p, int m, float& mean, float& rstd) {
    const float* rs = p.rstat_in + (long long)m * p.rstat_slots * 2;
    const float cnt = (float)p.rstat_cnt;
    float n = cnt, mu = rs[0], m2 = rs[1];
    for (int t = 1; t < p.rstat_slots; ++t) {
        const float mb = rs[2 * t], qb = rs[2 * t + 1];
        const float nn = n + cnt, d = mb - mu;
        mu += d * (cnt / nn);
        m2 += qb + d * d * (n * cnt / nn);
        n = nn;
    }
    mean = mu;
    rstd = 1.0f / sqrtf(m2 / n + p.ln_eps);
}
// (mean, M2) of the NV x 4 values one lane holds of its row, then merged over the four lanes (lane >> 4 = 0..3) that hold the
// same row of a 16-row block: equal counts at every level, fixed order
template <int NV>
__device__ __forceinline__ void xp_row_stats(const f32x4 (&v)[NV], float& mean, float& m2) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    float mu = s * (1.0f / (4 * NV)), q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const f32x4 d = v[j] - mu;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    float n = 4.f * NV;
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const float mb = __shfl_xor(mu, off), qb = __shfl_xor(q, off);
        const float d = mb - mu;
        q = q + qb + d * d * (n * 0.5f);
        mu = 0.5f * (mu + mb);
        n *= 2.f;
    }
    mean = mu; m2 = q;
}

