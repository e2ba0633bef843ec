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

// ---- column order of the implicit GEMM's weight block (igemm_x3p_kernel).  The MFMA hands a lane FOUR consecutive output columns
// of a 16-column block (rows 4 fq .. 4 fq + 3 of its A operand = the weight block).  Which weight row feeds which MFMA row is the
// staging's choice, so blocks are PAIRED: LDS row rho of block b (0 / 1) of pair P of a wave's WN-column slice holds
//   plain:  slice column 32 P + 8 (rho >> 2) + 4 b + (rho & 3)   -> a lane's two accumulators are EIGHT consecutive columns: fp32
//           results leave as 32 contiguous bytes per lane (a row's 128 bytes from four lanes), planes as 16-byte stores -- measured
//           with 4-column lanes: FeedForward.net[0] with planes out 123 us vs 99 us with fp32 out of twice the bytes
//   GEGLU:  weight rows come interleaved [8 hidden | 8 gate] per 16 (host layout); slice row 32 P + 16 (rho >> 3) + 8 b + (rho & 7):
//           block 0 = the pair's 16 hidden columns, block 1 = their 16 gates -- hidden and gate of a column meet in ONE lane (no
//           cross-lane exchange, every lane evaluates gelu, a lane stores four consecutive outputs)
// An odd last block of the slice (WN = 80: the fifth) keeps the identity order.
template <int WN>
__device__ __forceinline__ int xp_perm_col(int q, bool geglu) {
    constexpr int TN = WN / 16, NP2 = TN / 2;
    const int w = q / WN, lq = q - w * WN, jb = lq >> 4, rho = lq & 15;
    if (jb >= 2 * NP2) return q;
    const int P = jb >> 1, b = jb & 1;
    const int off = geglu ? 32 * P + 16 * (rho >> 3) + 8 * b + (rho & 7) : 32 * P + 8 * (rho >> 2) + 4 * b + (rho & 3);
    return w * WN + off;
}

// eight consecutive columns n .. n + 7 of row m (both halves inside N): the pair form of xp_store
__device__ __forceinline__ void xp_store8(const IefGemmX3pParams& p, f32x4& v0, f32x4& v1, int m, int n) {
    if (p.bias) { v0 += *(const f32x4*)(p.bias + n); v1 += *(const f32x4*)(p.bias + n + 4); }
    if (p.rowvec) {
        const float* rv = p.rowvec + (long long)(m / p.rows_per_batch) * p.N + n;
        v0 += *(const f32x4*)rv; v1 += *(const f32x4*)(rv + 4);
    }
    if (p.residual) {
        const float* rr = p.residual + (long long)m * p.ldr + n;
        v0 += *(const f32x4*)rr; v1 += *(const f32x4*)(rr + 4);
    }
    v0 = v0 * p.out_scale; v1 = v1 * p.out_scale;
    if (p.Out) {
        float* o = p.Out + (long long)m * p.ldo + n;
        *(f32x4*)o = v0; *(f32x4*)(o + 4) = v1;
    }
    if (p.OutP) {
        half4 h0, l0, h1, l1;
        split4(v0, 1.0f, h0, l0);
        split4(v1, 1.0f, h1, l1);
        half_t* o = p.OutP + (long long)m * p.ldp + n;
        if ((((uintptr_t)o | (uintptr_t)(p.planeO * 2)) & 15) == 0) {
            *(half8*)o = half8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            *(half8*)(o + p.planeO) = half8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        } else {
            *(half4*)o = h0; *(half4*)(o + 4) = h1;
            *(half4*)(o + p.planeO) = l0; *(half4*)(o + p.planeO + 4) = l1;
        }
    }
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

