// Cross-attention (<= 96 keys) with the Prompt-to-Prompt map edit FUSED, split-operand arithmetic ("f16x3" mode).
//
// The fp32-storage modes ran an edited cross-attention layer as four launches over materialised maps [B*heads][N][77]:
// scores GEMM, row softmax, `p2p_cross_edit_f32_kernel` (P' = c1 (P_src M) + c2 P_tgt), apply GEMM: 81 us per layer at the
// 64x64 level, 16 layers per step.  Here one workgroup owns 128 queries of one (batch row, head) -- a wave 32 of them -- and
// keeps every map in registers, in the layout of attn_flash_x3_kernel (S^T = K Q^T on v_mfma_f32_32x32x16_f16: a lane owns
// one query COLUMN, register r of key tile t <-> key 32 t + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)):
//   1. a row the controller edits (edit_src[b] >= 0): the SOURCE row's maps P_src = softmax(scale Q_src K_src^T), all <= 96
//      keys at once (three 32-key tiles, exact softmax: no running rescale), recomputed here rather than read from HBM;
//      T^T = MT P_src^T with the P registers as the B operand and the slot's table MT[w][v] = M[v][w] (split into fp16
//      planes while staged) as the A operand -- the k order of a P register block is the one the V^T fragments of the fused
//      attention use, so the table's fragments are two 8-byte LDS reads like theirs;
//   2. this row's own maps P = softmax(scale Q K^T);  P'[w] = c1[w] T[w] + c2[w] P[w]  (`attention_base.py:118-121`,
//      `attention_control.py:15-46` lowered to (M, c1, c2) by the host's plan);
//   3. O^T = V^T P'^T as in the fused attention.
// Every product is Ah Bh + Al Bh + Ah Bl on fp16 MFMAs (scales: q / k / v 2^2, maps 2^14, table 2^8), softmax and the mix in
// fp32.  K (all <= 96 keys), the table and V^T take turns in ONE LDS region, each loaded into registers a phase ahead and
// staged whole (two barriers per phase): the kernel is latency bound and small (5 MFLOP per workgroup); what it removes is
// three launches and the HBM round trips of the maps.
#include "ief_common.h"
#include "ief_params.h"
#include "x3_common.h"

template <int D>
__global__ __launch_bounds__(256, (D > 80) ? 1 : 2) void attn_cross_p2p_x3_kernel(const IefAttnF32Params p, const int* __restrict__ edit_src,
                                                                                  const int* __restrict__ edit_slot,
                                                                                  const float* __restrict__ MT,
                                                                                  const float* __restrict__ coef) {
    constexpr int DG = (D + 15) / 16;            // 16-deep groups of the score product
    constexpr int DT = (D + 31) / 32;            // 32-row tiles of O^T
    constexpr int KLD = DG * 16 + 8;             // halves per K row
    constexpr int VLD = 100;                     // halves per V^T row (all <= 96 keys + 4)
    constexpr int MLD = 100;                     // halves per table row (96 source tokens + 4)
    constexpr float SQ = 1.f, SK = 1.f, SV = 1.f, SP = 16384.f, SM = 256.f;   // activations: scale 1; source maps (<= 1): 2^14; table: 2^8
    const float SPE = p.p_scale > 0.f ? p.p_scale : SP;      // the EDITED maps P' = c1 T + c2 P can exceed 1 (AttentionReweight): host-chosen scale
    // ONE region, re-used phase by phase (K of the source row | the table | K of this row | V^T), each image complete (all
    // <= 96 keys) so that a phase costs two barriers, not two per 32-key tile.  Whatever a phase leaves behind is finite fp16
    // data, and the region is zeroed once at the start: padding columns of K (d >= D) multiply the zero padding of q.
    constexpr int RK = 2 * 96 * KLD, RV = 2 * DT * 32 * VLD, RM = 2 * 96 * MLD;
    constexpr int RSZ = RK > RV ? (RK > RM ? RK : RM) : (RV > RM ? RV : RM);
    __shared__ __attribute__((aligned(16))) half_t smem_c[RSZ];
    __shared__ float cf[2 * 96];
    half_t* Kh = smem_c;
    half_t* Kl = Kh + 96 * KLD;
    half_t* Vh = smem_c;                         // [DT*32][VLD]
    half_t* Vl = Vh + DT * 32 * VLD;
    half_t* Mh = smem_c;                         // [96][MLD]
    half_t* Ml = Mh + 96 * MLD;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int nqb = gridDim.x;
    const int lid = xcd_remap(blockIdx.x + nqb * blockIdx.y, nqb * gridDim.y);
    const int bh = lid / nqb, qb = lid - bh * nqb, b = bh / p.heads, h = bh - b * p.heads;
    const int src = edit_src ? edit_src[b] : -1;
    const int qi = qb * 128 + wid * 32 + li;
    const int nt = (p.L + 31) / 32;              // <= 3
    const float sc2 = p.scale * 1.44269504088896341f / (SQ * SK);      // scores in log2 units
    // zero the LDS once: padding columns of K (d >= D) and padding rows of V^T stay zero (staging never writes them)
    for (int c = tid; c < (int)(sizeof(smem_c) / 16); c += 256) ((f32x4*)smem_c)[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    constexpr int KCH = 32 * (D / 4);            // 16-byte chunks of one K (or V) tile
    constexpr int NLD = (KCH + 255) / 256;
    // normalised maps of batch row `br` for this wave's 32 queries: P[t][r] = softmax over ALL keys, keys >= L exactly 0
    auto probs = [&](int br, f32x16 (&P)[3]) {
        const float* Q = p.Q + (long long)br * p.sQb + (long long)h * D;
        const float* Kp = p.K + (long long)br * p.sKb + (long long)h * D;
        half8_t qh[DG], ql[DG];
#pragma unroll
        for (int g = 0; g < DG; ++g) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const int d0 = g * 16 + 8 * lh + 4 * c2;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (qi < p.N && d0 < D) v = *(const f32x4*)(Q + (long long)qi * p.ldq + d0);
                half4 hh, ll;
                split4(v, SQ, hh, ll);
#pragma unroll
                for (int j = 0; j < 4; ++j) { qh[g][4 * c2 + j] = hh[j]; ql[g][4 * c2 + j] = ll[j]; }
            }
        }
        // all <= 3 key tiles' loads are issued before the first use (one exposed memory latency per phase, not one per tile)
        f32x4 rk[3][NLD];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int c = tid + 256 * i;
                const int row = c / (D / 4), ch = c - row * (D / 4);
                rk[t][i] = (t < nt && c < KCH && t * 32 + row < p.L) ? *(const f32x4*)(Kp + (long long)(t * 32 + row) * p.ldk + ch * 4)
                                                                     : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        __syncthreads();                          // the previous phase's readers are done
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int c = tid + 256 * i;
                if (t < nt && c < KCH) {
                    const int row = c / (D / 4), ch = c - row * (D / 4);
                    half4 hh, ll;
                    split4(rk[t][i], SK, hh, ll);
                    *(half4*)(Kh + (t * 32 + row) * KLD + ch * 4) = hh;
                    *(half4*)(Kl + (t * 32 + row) * KLD + ch * 4) = ll;
                }
            }
        if (DG * 16 > D) {                        // padding columns: zeros (the region may hold another phase's data)
            constexpr int PADC = (DG * 16 - D) / 4;
            for (int c = tid; c < 96 * PADC; c += 256) {
                const int row = c / PADC, ch = c - row * PADC;
                *(half4*)(Kh + row * KLD + D + ch * 4) = half4{0, 0, 0, 0};
                *(half4*)(Kl + row * KLD + D + ch * 4) = half4{0, 0, 0, 0};
            }
        }
        __syncthreads();
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) P[t][r] = -INFINITY;
            if (t < nt) {
                f32x16 sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
                for (int g = 0; g < DG; ++g) {
                    const half8_t kh = *(const half8_t*)(Kh + (t * 32 + li) * KLD + g * 16 + 8 * lh);
                    const half8_t kl = *(const half8_t*)(Kl + (t * 32 + li) * KLD + g * 16 + 8 * lh);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[g], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[g], sacc, 0, 0, 0);
                    sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[g], sacc, 0, 0, 0);
                }
                if ((t + 1) * 32 > p.L) {         // wave-uniform: only the tile that crosses L masks
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        P[t][r] = key < p.L ? sacc[r] * sc2 : -INFINITY;
                        mx = fmaxf(mx, P[t][r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { P[t][r] = sacc[r] * sc2; mx = fmaxf(mx, P[t][r]); }
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float ls = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { P[t][r] = __builtin_amdgcn_exp2f(P[t][r] - mx); ls += P[t][r]; }      // bare v_exp_f32; exp2(-inf) = 0
        ls += __shfl_xor(ls, 32);
        const float inv = 1.0f / ls;
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) P[t][r] *= inv;
    };
    // registers 8 s .. 8 s + 7 of a map tile -> the hi / lo fp16 B operand of one 16-deep k step
    auto split_p = [&](const f32x16& P, int s, half8_t& ph, half8_t& pl, float sp) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const f32x4 v = {P[8 * s + 4 * c2], P[8 * s + 4 * c2 + 1], P[8 * s + 4 * c2 + 2], P[8 * s + 4 * c2 + 3]};
            half4 hh, ll;
            split4(v, sp, hh, ll);
#pragma unroll
            for (int j = 0; j < 4; ++j) { ph[4 * c2 + j] = hh[j]; pl[4 * c2 + j] = ll[j]; }
        }
    };

    f32x16 T[3];
    if (src >= 0) {                               // ---- 1. the source row's maps through the slot's table (uniform branch)
        const int slot = edit_slot[b];
        // table: 96 x 96 floats = 2304 chunks of 4, split with scale 2^8; coefficients
        const float* mt = MT + (long long)slot * 96 * 96;
        f32x4 rm[9];                              // 96 * 24 = 9 * 256 chunks; the loads travel under the source row's maps
#pragma unroll
        for (int i = 0; i < 9; ++i) rm[i] = *(const f32x4*)(mt + (tid + 256 * i) * 4);
        if (tid < 192) cf[tid] = coef[(long long)slot * 192 + tid];
        f32x16 Ps[3];
        probs(src, Ps);
        __syncthreads();                          // K of the source row is dead: the table takes the region
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int c = tid + 256 * i;
            const int w = c / 24, ch = c - w * 24;
            half4 hh, ll;
            split4(rm[i], SM, hh, ll);
            *(half4*)(Mh + w * MLD + ch * 4) = hh;
            *(half4*)(Ml + w * MLD + ch * 4) = ll;
        }
        __syncthreads();
        half8_t ph[6], pl[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) split_p(Ps[s >> 1], s & 1, ph[s], pl[s], SP);
#pragma unroll
        for (int wt = 0; wt < 3; ++wt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T[wt][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const half_t* mr = Mh + (wt * 32 + li) * MLD + 16 * s + 4 * lh;
                const half_t* mq = Ml + (wt * 32 + li) * MLD + 16 * s + 4 * lh;
                half8_t mh, ml;
                const half4 a0 = *(const half4*)mr, a1 = *(const half4*)(mr + 8);
                const half4 b0 = *(const half4*)mq, b1 = *(const half4*)(mq + 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) { mh[j] = a0[j]; mh[4 + j] = a1[j]; ml[j] = b0[j]; ml[4 + j] = b1[j]; }
                T[wt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ml, ph[s], T[wt], 0, 0, 0);
                T[wt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh, pl[s], T[wt], 0, 0, 0);
                T[wt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(mh, ph[s], T[wt], 0, 0, 0);
            }
        }
    }
    // ---- 2. this row's own maps, mixed (the V tiles' loads travel under it)
    const float* Vp = p.V + (long long)b * p.sVb + (long long)h * D;
    f32x4 rv[3][NLD];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 256 * i;
            const int row = c / (D / 4), ch = c - row * (D / 4);
            rv[t][i] = (t < nt && c < KCH && t * 32 + row < p.L) ? *(const f32x4*)(Vp + (long long)(t * 32 + row) * p.ldv + ch * 4)
                                                                 : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    f32x16 P[3];
    probs(b, P);
    if (src >= 0) {
        const float invt = 1.0f / (SM * SP);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int w = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                P[t][r] = cf[w] * (T[t][r] * invt) + cf[96 + w] * P[t][r];
            }
    }
    // ---- 3. O^T = V^T P'^T
    f32x16 o[DT];
#pragma unroll
    for (int tt = 0; tt < DT; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[tt][r] = 0.f;
    __syncthreads();                              // K of this row is dead: V^T takes the region (rows d >= D stay whatever they
                                                  // are: an A-operand row only reaches its own output row, which is not stored)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 256 * i;
            if (t < nt && c < KCH) {
                const int row = c / (D / 4), ch = c - row * (D / 4);
                half4 hh, ll;
                split4(rv[t][i], SV, hh, ll);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Vh[(ch * 4 + j) * VLD + t * 32 + row] = hh[j];
                    Vl[(ch * 4 + j) * VLD + t * 32 + row] = ll[j];
                }
            }
        }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (t < nt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8_t ph, pl;
                split_p(P[t], s, ph, pl, SPE);
#pragma unroll
                for (int tt = 0; tt < DT; ++tt) {
                    const half_t* vr = Vh + (tt * 32 + li) * VLD + t * 32 + 16 * s + 4 * lh;
                    const half_t* vq = Vl + (tt * 32 + li) * VLD + t * 32 + 16 * s + 4 * lh;
                    half8_t vh, vl;
                    const half4 a0 = *(const half4*)vr, a1 = *(const half4*)(vr + 8);
                    const half4 b0 = *(const half4*)vq, b1 = *(const half4*)(vq + 8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { vh[j] = a0[j]; vh[4 + j] = a1[j]; vl[j] = b0[j]; vl[4 + j] = b1[j]; }
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[tt], 0, 0, 0);
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[tt], 0, 0, 0);
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[tt], 0, 0, 0);
                }
            }
        }
    }
    const float inv = 1.0f / (SV * SPE);
    float* O = p.Out ? p.Out + (long long)b * p.sOb + (long long)h * D : nullptr;
    half_t* OP = p.OutP ? p.OutP + (long long)b * p.sOPb + (long long)h * D : nullptr;
    if (qi < p.N) {
#pragma unroll
        for (int tt = 0; tt < DT; ++tt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = tt * 32 + 8 * g + 4 * lh;       // registers 4g .. 4g+3 <-> d .. d+3
                if (d < D) {
                    const f32x4 v = {o[tt][4 * g] * inv, o[tt][4 * g + 1] * inv, o[tt][4 * g + 2] * inv, o[tt][4 * g + 3] * inv};
                    if (O) *(f32x4*)(O + (long long)qi * p.ldo + d) = v;
                    if (OP) {
                        half4 hh, ll;
                        split4(v, 1.0f, hh, ll);
                        half_t* op = OP + (long long)qi * p.ldp + d;
                        *(half4*)op = hh;
                        *(half4*)(op + p.planeO) = ll;
                    }
                }
            }
    }
}

// Cross-attention with the fused map edit on fp32 operands, split-operand arithmetic.  p: as ief_attn_flash_f32 (no batch-row
// indirection: q_src / k_src / v_src must be NULL; L <= 96; d in {40, 64, 80, 160}; 16-byte aligned rows); edit_src / edit_slot /
// MT / coef as ief_p2p_cross_edit_f32 (edit_src NULL: plain attention).
extern "C" int ief_attn_cross_p2p_f32(const IefAttnF32Params* pp, const int* edit_src, const int* edit_slot, const float* MT,
                                      const float* coef, void* stream) {
    if (!pp) return IEF_EINVAL;
    const IefAttnF32Params& p = *pp;
    if (!p.Q || !p.K || !p.V || (!p.Out && !p.OutP) || p.q_src || p.k_src || p.v_src) return IEF_EINVAL;
    if (p.OutP && ((p.ldp & 3) || (p.sOPb & 3) || (p.planeO & 3) || ((uintptr_t)p.OutP & 7))) return IEF_EALIGN;
    if (p.p_scale != 0.f && !(p.p_scale >= 1.f && p.p_scale <= 16384.f)) return IEF_EINVAL;
    if (edit_src && (!edit_slot || !MT || !coef)) return IEF_EINVAL;
    if (p.B <= 0 || p.heads <= 0 || p.N <= 0 || p.L <= 0 || p.L > 96) return IEF_ESHAPE;
    if ((p.ldq & 3) || (p.ldk & 3) || (p.ldv & 3) || (p.ldo & 3) || (p.sQb & 3) || (p.sKb & 3) || (p.sVb & 3) || (p.sOb & 3) ||
        (((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V | (uintptr_t)(p.Out ? p.Out : p.Q) | (uintptr_t)(MT ? MT : p.Q)) & 15)) return IEF_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((p.N + 127) / 128, p.B * p.heads);
    switch (p.d) {
        case 40: hipLaunchKernelGGL(attn_cross_p2p_x3_kernel<40>, grid, dim3(256), 0, st, p, edit_src, edit_slot, MT, coef); break;
        case 64: hipLaunchKernelGGL(attn_cross_p2p_x3_kernel<64>, grid, dim3(256), 0, st, p, edit_src, edit_slot, MT, coef); break;
        case 80: hipLaunchKernelGGL(attn_cross_p2p_x3_kernel<80>, grid, dim3(256), 0, st, p, edit_src, edit_slot, MT, coef); break;
        case 160: hipLaunchKernelGGL(attn_cross_p2p_x3_kernel<160>, grid, dim3(256), 0, st, p, edit_src, edit_slot, MT, coef); break;
        default: return IEF_ESHAPE;
    }
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
