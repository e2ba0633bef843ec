// Fused attention kernels for gfx950 (fp16 operands, fp32 softmax and accumulation).
//
// All four kernels share one tiling.  A workgroup = 4 waves = 128 query rows of one
// (batch, head); each wave owns 32 queries.  Scores are computed TRANSPOSED,
//     S^T[kv][q] = K[kv][:] . Q[q][:]        v_mfma_f32_32x32x16_f16(A = K rows, B = Q rows)
// so that a lane holds one query COLUMN: 16 scores per 32-key sub-tile, the other 16 in lane^32.
// Row max / row sum are then in-lane reductions plus one cross-half shuffle, the online-softmax
// rescale of the output is a per-lane scalar, and the probability registers ARE the B operand of
//     O^T[d][q] += V^T[d][kv] . P^T[kv][q]   (A = V^T rows read from LDS, B = P^T from registers)
// with the k-order permutation of an accumulator-as-operand (element j of lane half h is key
// 16s + 8(j>>2) + 4h + (j&3)) applied to the V^T fragment reads.  No probability ever touches LDS
// or HBM in the fused kernels.
//
//   attn_flash      softmax(QK^T)V with per-batch Q/K/V source indirection (self-attention; P2P
//                   self-replace and MasaCtrl mutual attention are pure index remaps)
//   attn_cross_p2p  L <= 96 keys; Prompt-to-Prompt cross-map edit fused between softmax and PV
//   attn_probs / attn_apply   materialised maps for the generic Python-controller path
//
// Reference call sites: /root/reference/p2p/model/register.py:47-51 (scores, controller, bmm),
// /root/reference/p2p/model/attention_base.py:113-136 (edit), attention_control.py:15-46.
#include "ief_common.h"
#include "ief_params.h"

#define LOG2E 1.4426950408889634f

template <int D>
struct AttnCfg {
    static constexpr int D16 = (D + 15) / 16;  // k-steps of the score MFMA
    static constexpr int DP = D16 * 16;        // padded head dim of the K tile
    static constexpr int DT = (D + 31) / 32;   // 32-row output tiles of O^T
    static constexpr int KS = DP + 8;          // K row stride (halves): odd number of 16-B slots
    static constexpr int CPR = D / 8;          // 16-B chunks per K/V row
    // D not a multiple of 32 leaves padding rows in the last O^T tile: row D of V^T is set to all ones, so
    // the PV MFMA itself accumulates the softmax row sum (O^T[D][q] = sum_kv P[q][kv]) at no VALU cost.
    static constexpr bool ONES_ROW = (D % 32) != 0;
    // D not a multiple of 16 leaves padding columns in the K tile / Q fragments (k = D .. DP-1 of the score MFMA):
    // K[key][D] is set to 1 and Q^T[D][q] to -m_run[q], so the MFMA itself subtracts the running row maximum and the
    // accumulators can start from the constant 0 (no per-lane 16-register init vector, no per-element subtraction).
    static constexpr bool FOLD_MAX = (D % 16) != 0;
    static constexpr int PAD_S = D / 16, PAD_H = (D % 16) / 8, PAD_E = (D % 16) % 8;   // k-step / lane half / element of column D
};

template <bool B>
struct BoolTag { static constexpr bool value = B; };

__device__ __forceinline__ half8 pack8(const f32x16& p, int base) {
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)p[base + j];
    return o;
}

// A-operand fragment with the accumulator k-permutation: 4 + 4 consecutive halves of one LDS row
__device__ __forceinline__ half8 read_perm_frag(const half_t* row_ptr, int kbase, int h) {
    const half4 lo = *(const half4*)(row_ptr + kbase + 4 * h);
    const half4 hi = *(const half4*)(row_ptr + kbase + 8 + 4 * h);
    half8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return o;
}
// natural k order: 8 consecutive halves starting at kbase + 8h (8-byte aligned rows)
__device__ __forceinline__ half8 read_nat_frag(const half_t* row_ptr, int kbase, int h) {
    const half4 lo = *(const half4*)(row_ptr + kbase + 8 * h);
    const half4 hi = *(const half4*)(row_ptr + kbase + 8 * h + 4);
    half8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return o;
}

template <int D>
__device__ __forceinline__ void load_q_frags(half8 (&qf)[AttnCfg<D>::D16], const half_t* Q, long long row_off,
                                             bool row_ok, int h) {
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < AttnCfg<D>::D16; ++s) {
        const int dc = 16 * s + 8 * h;
        qf[s] = (row_ok && dc < D) ? *(const half8*)(Q + row_off + dc) : zero8;
    }
}

// ------------------------------------------------------------------------------------------
// flash attention with source indirection
// ------------------------------------------------------------------------------------------
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4_t lds_fp16x4_t;

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of halves, delivered column-major
// (lane i of the group gets column i of the 4 rows).  Every lane passes the address of row (L>>2),
// columns 4(L&3).. of ITS group's block (L = lane & 15).  EXEC must be all ones.
__device__ __forceinline__ half4 lds_tr_read(const half_t* p) {
    const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4_t*)p);
    return __builtin_bit_cast(half4, v);
}

template <int D>
__global__ __launch_bounds__(256) void attn_flash_kernel(const IefAttnParams p) {
    using C = AttnCfg<D>;
    // V tile kept ROW-major [key][d]; the PV A-operand (V^T fragment) comes from transposing reads.  Row stride
    // = 64 or 192 (mod 256) bytes so the four rows of a 32-lane half land in distinct 64-B bank slots.
    constexpr int VRS = D <= 32 ? 32 : (D <= 96 ? 96 : 160);   // halves per V row in LDS
    constexpr int NCH = (64 * C::CPR + 255) / 256;
    constexpr int KBUF = 64 * C::KS, VBUF = 64 * VRS;
    __shared__ __attribute__((aligned(16))) half_t Ks[2 * KBUF];
    __shared__ __attribute__((aligned(16))) half_t Vs[2 * VBUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // 1-D grid, XCD-aware: the query blocks of one (batch, head) get consecutive LOGICAL ids, which xcd_remap
    // places on one XCD — its L2 then serves that head's K/V to all of them (measured before: every XCD
    // fetched every head's K/V, 4x the algorithmic read traffic)
    const int qblocks = (p.N + 127) / 128;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = lid % qblocks, head = (lid / qblocks) % p.heads, b = lid / (qblocks * p.heads);
    const int qs = p.q_src ? p.q_src[b] : b;
    const int ks = p.k_src ? p.k_src[b] : b;
    const int vs = p.v_src ? p.v_src[b] : b;
    const int q0 = qblk * 128 + wave * 32;
    const bool q_ok = q0 + r < p.N;

    // zero both buffers once: K pad columns (d..DP) and V pad columns (d..32*DT) are never rewritten
    for (int i = tid; i < 2 * KBUF / 8; i += 256) ((half8*)Ks)[i] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < 2 * VBUF / 8; i += 256) ((half8*)Vs)[i] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (C::ONES_ROW) {  // column D of V = 1: the PV MFMA then accumulates the softmax row sum in O^T[D][q]
        __syncthreads();
        if (tid < 128) Vs[(tid >> 6) * VBUF + (tid & 63) * VRS + D] = (half_t)1.0f;
        if constexpr (C::FOLD_MAX)
            if (tid < 128) Ks[(tid >> 6) * KBUF + (tid & 63) * C::KS + D] = (half_t)1.0f;
    } else if constexpr (C::FOLD_MAX) {
        __syncthreads();
        if (tid < 128) Ks[(tid >> 6) * KBUF + (tid & 63) * C::KS + D] = (half_t)1.0f;
    }

    // Q fragments pre-multiplied by scale*log2(e): scores come out of the MFMA in log2 units, so the softmax
    // needs no per-element multiply (one extra fp16 rounding of q; error << the fp16 rounding of P)
    half8 qf[C::D16];
    load_q_frags<D>(qf, p.Q, ((long long)qs * p.N + q0 + r) * p.ldq + head * D, q_ok, h);
    {
        const float sc = p.scale * LOG2E;
#pragma unroll
        for (int s = 0; s < C::D16; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = (half_t)((float)qf[s][e] * sc);
    }

    const half_t* Kb = p.K + (long long)ks * p.L * p.ldk + head * D;
    const half_t* Vb = p.V + (long long)vs * p.L * p.ldv + head * D;
    // staging map of this thread (fixed for the whole kernel).  Loads are unconditional: a row past the last key is
    // clamped to the last key (its scores are masked to -inf, so neither its K nor its V reaches the output), which
    // keeps the tile loop free of selects; the row pointers advance by one scalar per tile.
    int st_k[NCH], st_v[NCH];
    int st_row[NCH], st_ch[NCH];
    bool st_ok[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + 256 * i;
        const int row = c / C::CPR, ch = c - row * C::CPR;
        st_ok[i] = c < 64 * C::CPR;
        st_row[i] = st_ok[i] ? row : 0;
        st_ch[i] = st_ok[i] ? ch * 8 : 0;
        st_k[i] = row * C::KS + ch * 8;
        st_v[i] = row * VRS + ch * 8;
    }
    half8 kreg[NCH], vreg[NCH];
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int row = min(kv0 + st_row[i], p.L - 1);
            kreg[i] = *(const half8*)(Kb + (long long)row * p.ldk + st_ch[i]);
            vreg[i] = *(const half8*)(Vb + (long long)row * p.ldv + st_ch[i]);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (st_ok[i]) {
                *(half8*)(Ks + buf * KBUF + st_k[i]) = kreg[i];
                *(half8*)(Vs + buf * VBUF + st_v[i]) = vreg[i];
            }
        }
    };

    f32x16 o[C::DT];
#pragma unroll
    for (int t = 0; t < C::DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    // Online softmax with the running row maximum folded INTO the score MFMA: the accumulators start at
    // -m_run (a per-lane row constant, kept in `minit`), so S' = s - m_run leaves the MFMA chain and the common
    // case is p = 2^S' with no subtraction at all.  m_run moves only when a tile's maximum exceeds it by more than
    // RESCALE_THR (then O and the pending S' are corrected exactly once, before S' is exponentiated); P stays
    // <= 2^RESCALE_THR, far inside fp16 range.
    constexpr float RESCALE_THR = 5.0f;
    float m_run = 0.f, l_run = 0.f;
    bool first = true;
    f32x16 minit;
#pragma unroll
    for (int i = 0; i < 16; ++i) minit[i] = 0.f;

    // per-lane bases of the fragment reads
    const int k_lane = r * C::KS + 8 * h;                                   // K rows (A operand of S^T)
    const int L16 = lane & 15;
    const int v_lane = (4 * h + (L16 >> 2)) * VRS + 16 * ((lane >> 4) & 1) + 4 * (L16 & 3);   // transposing V reads

    // one 64-key tile: S^T = K Q^T, online softmax, O^T += V^T P^T.  `masked` is a compile-time tag so the
    // full tiles carry no bounds code at all.
    auto tile_body = [&](int kv0, const half_t* Kc, const half_t* Vc, auto masked) -> bool {
        f32x16 s0, s1;
        if constexpr (C::FOLD_MAX) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
        } else {
            s0 = minit; s1 = minit;
        }
#pragma unroll
        for (int s = 0; s < C::D16; ++s) {
            const half8 k0 = *(const half8*)(Kc + k_lane + 16 * s);
            const half8 k1 = *(const half8*)(Kc + k_lane + 32 * C::KS + 16 * s);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[s], s1, 0, 0, 0);
        }
        if constexpr (decltype(masked)::value) {   // only the last, partial key tile carries the mask code
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kvl = (i & 3) + 8 * (i >> 2) + 4 * h;
                if (kv0 + kvl >= p.L) s0[i] = -INFINITY;
                if (kv0 + 32 + kvl >= p.L) s1[i] = -INFINITY;
            }
        }
        // this lane's 32 scores (two independent max3 chains); the other half of the row lives in lane^32, but the
        // common case only needs to know that NOBODY exceeds the threshold, which __any answers without a shuffle
        float mxa = fmaxf(s0[0], s1[0]), mxb = fmaxf(s0[8], s1[8]);
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, s0[i]), s1[i]);
            mxb = __builtin_fmaxf(__builtin_fmaxf(mxb, s0[8 + i]), s1[8 + i]);
        }
        float mx = fmaxf(mxa, mxb);
        if (__builtin_expect(first || __any(mx > RESCALE_THR), 0)) {   // wave-uniform; rare after the first tile
            // move the reference maximum, rescale what was accumulated under the old one, and REDO this tile's
            // scores against the new reference (nothing of this tile has been consumed yet).  Keeping the
            // correction off the fall-through path keeps the accumulators in place in the hot loop.
            mx = fmaxf(mx, __shfl_xor(mx, 32));          // this row's maximum, relative to m_run
            float delta = first ? mx : fmaxf(mx, 0.f);
            if constexpr (C::FOLD_MAX) {
                // the reference lives in an fp16 operand: move it by an amount that keeps it exactly representable
                const float m_new = (float)(half_t)(m_run + delta);
                delta = m_new - m_run;
                m_run = m_new;
                if (h == C::PAD_H) qf[C::PAD_S][C::PAD_E] = (half_t)(-m_new);
            } else {
                m_run += delta;
#pragma unroll
                for (int i = 0; i < 16; ++i) minit[i] = -m_run;
            }
            const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
            if constexpr (!C::ONES_ROW) l_run *= alpha;
#pragma unroll
            for (int t = 0; t < C::DT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
            first = false;
            return false;
        }
        float ls = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s0[i] = __builtin_amdgcn_exp2f(s0[i]);
            s1[i] = __builtin_amdgcn_exp2f(s1[i]);
            if constexpr (!C::ONES_ROW) ls += s0[i] + s1[i];
        }
        if constexpr (!C::ONES_ROW) l_run += ls;
        const half8 pb[4] = {pack8(s0, 0), pack8(s0, 8), pack8(s1, 0), pack8(s1, 8)};
#pragma unroll
        for (int t = 0; t < C::DT; ++t) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {   // 16 keys per MFMA; fragment = keys {4h..4h+3} and {8+4h..8+4h+3} of the step
                const half_t* vp = Vc + v_lane + (16 * kk) * VRS + 32 * t;
                const half4 lo = lds_tr_read(vp), hi = lds_tr_read(vp + 8 * VRS);
                const half8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[kk], o[t], 0, 0, 0);
            }
        }
        return true;
    };

    const int nt = (p.L + 63) / 64;
    __syncthreads();  // zero fill (and ones column) done
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int pf = 0;   // highest tile whose global loads were issued (a redone tile must not issue them twice)
    for (int j = 0; j < nt;) {
        const int kv0 = j * 64, cur = j & 1;
        if (j + 1 < nt && pf < j + 1) { load_tile(kv0 + 64); pf = j + 1; }   // in flight during the MFMA/softmax work
        const half_t* Kc = Ks + cur * KBUF;
        const half_t* Vc = Vs + cur * VBUF;
        const bool done = (kv0 + 64 > p.L) ? tile_body(kv0, Kc, Vc, BoolTag<true>{}) : tile_body(kv0, Kc, Vc, BoolTag<false>{});
        if (!done) continue;                            // reference maximum moved: same tile again
        if (j + 1 < nt) store_tile(cur ^ 1);            // other buffer: last read one barrier ago
        __syncthreads();
        ++j;
    }
    float l_tot;
    if constexpr (C::ONES_ROW) {
        // row D of O^T: tile D/32, row i = D%32 = (reg&3) + 8(reg>>2) + 4h  ->  D%32 is a multiple of 8: reg = (D%32)/2, h = 0
        constexpr int LT = D / 32, LR = (D % 32) / 2;
        const float mine = o[LT][LR];
        const float other = __shfl_xor(mine, 32);
        l_tot = h ? other : mine;
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32);
    }
    const float inv = 1.0f / l_tot;
    // row log-sum-exp in log2 units of the PRE-SCALED scores (q * scale * log2e): what the backward kernels
    // (attention_bwd.hip) subtract to rebuild P without a second online softmax
    if (p.lse && q_ok && h == 0) p.lse[((long long)b * p.heads + head) * p.N + q0 + r] = m_run + __log2f(l_tot);
    if (q_ok) {
        half_t* orow = p.Out + ((long long)b * p.N + q0 + r) * p.ldo + head * D;
#pragma unroll
        for (int t = 0; t < C::DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int db = t * 32 + 8 * g + 4 * h;
                if (db < D) {
                    half4 v = {(half_t)(o[t][4 * g] * inv), (half_t)(o[t][4 * g + 1] * inv),
                               (half_t)(o[t][4 * g + 2] * inv), (half_t)(o[t][4 * g + 3] * inv)};
                    *(half4*)(orow + db) = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// flash attention, ping-pong form (long sequences, grids that fill the chip)
// ------------------------------------------------------------------------------------------
// attn_flash_kernel is bound by instruction ISSUE, not by a pipe: per 64-key tile a wave spends ~450 cycles of MFMA
// and ~550 cycles of VALU (32 v_exp at quarter rate, max, pack) one after the other, and with two free-running waves
// per SIMD the two pipes were busy 31 % / 56 % of the time without overlapping (rocprofv3 PMC).
// Here a workgroup is 8 waves = 256 queries; waves w and w+4 share a SIMD and run the SAME per-tile program half a
// tile apart, in lockstep phases separated by s_barrier:
//      phase:        0        1        2        3        4      ...
//      waves 0-3:   S(0)   soft(0)  PV(0)+S(1) soft(1) PV(1)+S(2)        <- matrix / vector / matrix / ...
//      waves 4-7:    -      S(0)    soft(0)  PV(0)+S(1) soft(1)
// so in every phase one wave of each SIMD feeds the matrix pipe (PV of the previous tile + scores of the next) while
// its partner does the softmax VALU work of its own tile.  K tiles are double- and V tiles triple-buffered in LDS;
// the vector half of phases t = 1, 2 (mod 4) stores the tile it loaded four phases earlier and issues the loads of
// the one after (global loads stay in flight across the raw barriers).  The online-softmax correction needs no tile
// redo here: the scores of tile j are still in registers when its maximum is known, so they are shifted in place.
template <int D>
__global__ __launch_bounds__(512) void attn_flash_pp_kernel(const IefAttnParams p) {
    using C = AttnCfg<D>;
    constexpr int VRS = D <= 32 ? 32 : (D <= 96 ? 96 : 160);   // halves per V row in LDS
    constexpr int NCH = (64 * C::CPR + 255) / 256;             // staging chunks per thread of one half
    constexpr int KBUF = 64 * C::KS, VBUF = 64 * VRS;
    __shared__ __attribute__((aligned(16))) half_t Ks[2 * KBUF];
    __shared__ __attribute__((aligned(16))) half_t Vs[3 * VBUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = wave >> 2;              // 0: waves 0..3, 1: waves 4..7 (the SIMD partners, half a tile late)
    const int th = tid & 255;             // thread index inside the half (staging map)
    const int r = lane & 31, h = lane >> 5;
    const int qblocks = (p.N + 255) / 256;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = lid % qblocks, head = (lid / qblocks) % p.heads, b = lid / (qblocks * p.heads);
    const int qs = p.q_src ? p.q_src[b] : b;
    const int ks = p.k_src ? p.k_src[b] : b;
    const int vs = p.v_src ? p.v_src[b] : b;
    const int q0 = qblk * 256 + wave * 32;
    const bool q_ok = q0 + r < p.N;

    for (int i = tid; i < 2 * KBUF / 8; i += 512) ((half8*)Ks)[i] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < 3 * VBUF / 8; i += 512) ((half8*)Vs)[i] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();
    if constexpr (C::ONES_ROW)
        if (tid < 192) Vs[(tid >> 6) * VBUF + (tid & 63) * VRS + D] = (half_t)1.0f;
    if constexpr (C::FOLD_MAX)
        if (tid >= 256 && tid < 384) Ks[((tid - 256) >> 6) * KBUF + (tid & 63) * C::KS + D] = (half_t)1.0f;

    half8 qf[C::D16];
    load_q_frags<D>(qf, p.Q, ((long long)qs * p.N + q0 + r) * p.ldq + head * D, q_ok, h);
    {
        const float sc = p.scale * LOG2E;
#pragma unroll
        for (int s = 0; s < C::D16; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = (half_t)((float)qf[s][e] * sc);
    }

    const half_t* Kb = p.K + (long long)ks * p.L * p.ldk + head * D;
    const half_t* Vb = p.V + (long long)vs * p.L * p.ldv + head * D;
    int st_k[NCH], st_v[NCH], st_row[NCH], st_ch[NCH];
    bool st_ok[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = th + 256 * i;
        const int row = c / C::CPR, ch = c - row * C::CPR;
        st_ok[i] = c < 64 * C::CPR;
        st_row[i] = st_ok[i] ? row : 0;
        st_ch[i] = st_ok[i] ? ch * 8 : 0;
        st_k[i] = row * C::KS + ch * 8;
        st_v[i] = row * VRS + ch * 8;
    }
    half8 kreg[NCH], vreg[NCH];
    auto load_tile = [&](int kv0) {      // rows past the last key are clamped (their scores are masked)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int row = min(kv0 + st_row[i], p.L - 1);
            kreg[i] = *(const half8*)(Kb + (long long)row * p.ldk + st_ch[i]);
            vreg[i] = *(const half8*)(Vb + (long long)row * p.ldv + st_ch[i]);
        }
    };
    auto store_tile = [&](int kslot, int vslot) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (st_ok[i]) {
                *(half8*)(Ks + kslot * KBUF + st_k[i]) = kreg[i];
                *(half8*)(Vs + vslot * VBUF + st_v[i]) = vreg[i];
            }
        }
    };

    f32x16 o[C::DT];
#pragma unroll
    for (int t = 0; t < C::DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    constexpr float RESCALE_THR = 5.0f;
    float m_run = 0.f, l_run = 0.f;
    bool first = true;
    f32x16 minit, s0, s1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { minit[i] = 0.f; s0[i] = 0.f; s1[i] = 0.f; }
    half8 pb[4] = {};

    const int k_lane = r * C::KS + 8 * h;
    const int L16 = lane & 15;
    const int v_lane = (4 * h + (L16 >> 2)) * VRS + 16 * ((lane >> 4) & 1) + 4 * (L16 & 3);
    const int nt = (p.L + 63) / 64;

    auto scores = [&](const half_t* Kc) {
        if constexpr (C::FOLD_MAX) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
        } else {
            s0 = minit; s1 = minit;
        }
#pragma unroll
        for (int s = 0; s < C::D16; ++s) {
            const half8 k0 = *(const half8*)(Kc + k_lane + 16 * s);
            const half8 k1 = *(const half8*)(Kc + k_lane + 32 * C::KS + 16 * s);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[s], s1, 0, 0, 0);
        }
    };
    auto pv = [&](const half_t* Vc) {
#pragma unroll
        for (int t = 0; t < C::DT; ++t) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const half_t* vp = Vc + v_lane + (16 * kk) * VRS + 32 * t;
                const half4 lo = lds_tr_read(vp), hi = lds_tr_read(vp + 8 * VRS);
                const half8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[kk], o[t], 0, 0, 0);
            }
        }
    };
    auto softmax = [&](int kv0) {
        if (kv0 + 64 > p.L) {                    // last, partial key tile
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kvl = (i & 3) + 8 * (i >> 2) + 4 * h;
                if (kv0 + kvl >= p.L) s0[i] = -INFINITY;
                if (kv0 + 32 + kvl >= p.L) s1[i] = -INFINITY;
            }
        }
        float mxa = fmaxf(s0[0], s1[0]), mxb = fmaxf(s0[8], s1[8]);
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, s0[i]), s1[i]);
            mxb = __builtin_fmaxf(__builtin_fmaxf(mxb, s0[8 + i]), s1[8 + i]);
        }
        float mx = fmaxf(mxa, mxb);
        if (__builtin_expect(first || __any(mx > RESCALE_THR), 0)) {   // wave-uniform; rare after the first tile
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float delta = first ? mx : fmaxf(mx, 0.f);
            if constexpr (C::FOLD_MAX) {
                const float m_new = (float)(half_t)(m_run + delta);     // the reference lives in an fp16 operand
                delta = m_new - m_run;
                m_run = m_new;
                if (h == C::PAD_H) qf[C::PAD_S][C::PAD_E] = (half_t)(-m_new);
            } else {
                m_run += delta;
#pragma unroll
                for (int i = 0; i < 16; ++i) minit[i] = -m_run;
            }
            const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
            if constexpr (!C::ONES_ROW) l_run *= alpha;
#pragma unroll
            for (int t = 0; t < C::DT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s0[i] -= delta; s1[i] -= delta; }   // this tile's scores, still in registers
            first = false;
        }
        float ls = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s0[i] = __builtin_amdgcn_exp2f(s0[i]);
            s1[i] = __builtin_amdgcn_exp2f(s1[i]);
            if constexpr (!C::ONES_ROW) ls += s0[i] + s1[i];
        }
        if constexpr (!C::ONES_ROW) l_run += ls;
        pb[0] = pack8(s0, 0); pb[1] = pack8(s0, 8); pb[2] = pack8(s1, 0); pb[3] = pack8(s1, 8);
    };
    auto phase_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    __syncthreads();                       // zero fill and the constant columns are in place
    if (H == 0) {
        load_tile(0);
        store_tile(0, 0);
        if (nt > 1) load_tile(64);         // tile 1: stored in phase 1
    } else if (nt > 2) {
        load_tile(128);                    // tile 2: stored in phase 2
    }
    phase_barrier();
    // Both halves run the SAME straight-line program per tile — [PV(j-1), S(j)] | barrier | [staging, softmax(j)] |
    // barrier — with waves 4..7 one phase late (one extra barrier in front, waves 0..3 one behind): phase t is the
    // matrix phase of tile (t - H) / 2 for half H.  The vector phases of even tiles carry the staging turn:
    // tile j + 1 + H is stored (loaded four phases earlier) and tile j + 3 + H requested.
    auto vector_phase = [&](int j) {
        if (!(j & 1)) {
            const int js = j + 1 + H;
            if (js < nt) {
                store_tile(js & 1, js % 3);
                if (js + 2 < nt) load_tile((js + 2) * 64);
            }
        }
        softmax(j * 64);
    };
    // the wave in its matrix phase gets issue priority: an MFMA occupies the issue port for 8 of its 32 cycles, the
    // partner's VALU stream fills the rest; at equal priority the (older) VALU stream starves the MFMAs instead
    if (H) phase_barrier();
    __builtin_amdgcn_s_setprio(1);
    scores(Ks);                                      // tile 0
    __builtin_amdgcn_s_setprio(0);
    phase_barrier();
    vector_phase(0);
    phase_barrier();
    for (int j = 1; j < nt; ++j) {
        __builtin_amdgcn_s_setprio(1);
        pv(Vs + ((j - 1) % 3) * VBUF);
        scores(Ks + (j & 1) * KBUF);
        __builtin_amdgcn_s_setprio(0);
        phase_barrier();
        vector_phase(j);
        phase_barrier();
    }
    __builtin_amdgcn_s_setprio(1);
    pv(Vs + ((nt - 1) % 3) * VBUF);
    __builtin_amdgcn_s_setprio(0);
    if (!H) phase_barrier();

    float l_tot;
    if constexpr (C::ONES_ROW) {
        constexpr int LT = D / 32, LR = (D % 32) / 2;
        const float mine = o[LT][LR];
        const float other = __shfl_xor(mine, 32);
        l_tot = h ? other : mine;
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32);
    }
    const float inv = 1.0f / l_tot;
    if (p.lse && q_ok && h == 0) p.lse[((long long)b * p.heads + head) * p.N + q0 + r] = m_run + __log2f(l_tot);
    if (q_ok) {
        half_t* orow = p.Out + ((long long)b * p.N + q0 + r) * p.ldo + head * D;
#pragma unroll
        for (int t = 0; t < C::DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int db = t * 32 + 8 * g + 4 * h;
                if (db < D) {
                    half4 v = {(half_t)(o[t][4 * g] * inv), (half_t)(o[t][4 * g + 1] * inv),
                               (half_t)(o[t][4 * g + 2] * inv), (half_t)(o[t][4 * g + 3] * inv)};
                    *(half4*)(orow + db) = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// flash attention, software-pipelined form: softmax of tile j in the MFMA shadows of PV(j-1) and S(j+1)
// ------------------------------------------------------------------------------------------
// Same tiling as attn_flash_kernel (4 waves x 32 queries), but one loop iteration holds, in ONE basic block, the
// matrix work of two other tiles — O += P(j-1) V(j-1) and S(j+1) = K(j+1) Q — next to the exp / pack VALU work of
// tile j, so the compiler can issue the VALU stream between the MFMAs of the same wave (an MFMA keeps the issue port
// for 8 of its 32 cycles).  The running maximum is checked BEFORE that block; when it moves, O, the packed P(j-1) and
// the pending scores are corrected first.  K is double-, V quadruple-buffered (V(j-1) .. V(j+2) are live).
template <int D>
__global__ __launch_bounds__(256, (D > 32 && D <= 48) ? 3 : 1) void attn_flash_sp_kernel(const IefAttnParams p) {
    using C = AttnCfg<D>;
    // d = 40: 128-B V rows (the transposing reads are then 2-way conflicted, LDS is not what bounds the kernel) bring a workgroup
    // to 46 KB and, with 166 registers, three workgroups onto a CU instead of two
    constexpr int VRS = D <= 32 ? 32 : (D <= 48 ? 64 : (D <= 96 ? 96 : 160));
    constexpr int NCH = (64 * C::CPR + 255) / 256;
    constexpr int KBUF = 64 * C::KS, VBUF = 64 * VRS;
    __shared__ __attribute__((aligned(16))) half_t Ks[2 * KBUF];
    __shared__ __attribute__((aligned(16))) half_t Vs[4 * VBUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int qblocks = (p.N + 127) / 128;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = lid % qblocks, head = (lid / qblocks) % p.heads, b = lid / (qblocks * p.heads);
    const int qs = p.q_src ? p.q_src[b] : b;
    const int ks = p.k_src ? p.k_src[b] : b;
    const int vs = p.v_src ? p.v_src[b] : b;
    const int q0 = qblk * 128 + wave * 32;
    const bool q_ok = q0 + r < p.N;

    for (int i = tid; i < 2 * KBUF / 8; i += 256) ((half8*)Ks)[i] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < 4 * VBUF / 8; i += 256) ((half8*)Vs)[i] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();
    if constexpr (C::ONES_ROW) Vs[(tid >> 6) * VBUF + (tid & 63) * VRS + D] = (half_t)1.0f;
    if constexpr (C::FOLD_MAX)
        if (tid < 128) Ks[(tid >> 6) * KBUF + (tid & 63) * C::KS + D] = (half_t)1.0f;

    half8 qf[C::D16];
    load_q_frags<D>(qf, p.Q, ((long long)qs * p.N + q0 + r) * p.ldq + head * D, q_ok, h);
    {
        const float sc = p.scale * LOG2E;
#pragma unroll
        for (int s = 0; s < C::D16; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = (half_t)((float)qf[s][e] * sc);
    }
    const half_t* Kb = p.K + (long long)ks * p.L * p.ldk + head * D;
    const half_t* Vb = p.V + (long long)vs * p.L * p.ldv + head * D;
    int st_k[NCH], st_v[NCH], st_row[NCH], st_ch[NCH];
    bool st_ok[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + 256 * i;
        const int row = c / C::CPR, ch = c - row * C::CPR;
        st_ok[i] = c < 64 * C::CPR;
        st_row[i] = st_ok[i] ? row : 0;
        st_ch[i] = st_ok[i] ? ch * 8 : 0;
        st_k[i] = row * C::KS + ch * 8;
        st_v[i] = row * VRS + ch * 8;
    }
    half8 kreg[NCH], vreg[NCH];
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int row = min(kv0 + st_row[i], p.L - 1);
            kreg[i] = *(const half8*)(Kb + (long long)row * p.ldk + st_ch[i]);
            vreg[i] = *(const half8*)(Vb + (long long)row * p.ldv + st_ch[i]);
        }
    };
    auto store_tile = [&](int kslot, int vslot) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (st_ok[i]) {
                *(half8*)(Ks + kslot * KBUF + st_k[i]) = kreg[i];
                *(half8*)(Vs + vslot * VBUF + st_v[i]) = vreg[i];
            }
        }
    };

    f32x16 o[C::DT];
#pragma unroll
    for (int t = 0; t < C::DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    constexpr float RESCALE_THR = 5.0f;
    float m_run = 0.f, l_run = 0.f;
    bool first = true;
    f32x16 minit, sa0, sa1, sb0, sb1;      // two score sets: the loop alternates them instead of copying "next" into "current"
#pragma unroll
    for (int i = 0; i < 16; ++i) { minit[i] = 0.f; sa0[i] = 0.f; sa1[i] = 0.f; sb0[i] = 0.f; sb1[i] = 0.f; }
    half8 pb[4] = {};

    const int k_lane = r * C::KS + 8 * h;
    const int L16 = lane & 15;
    const int v_lane = (4 * h + (L16 >> 2)) * VRS + 16 * ((lane >> 4) & 1) + 4 * (L16 & 3);
    const int nt = (p.L + 63) / 64;

    // scores of one tile into (a0, a1)
    auto scores = [&](const half_t* Kc, f32x16& a0, f32x16& a1) {
        if constexpr (C::FOLD_MAX) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
        } else {
            a0 = minit; a1 = minit;
        }
#pragma unroll
        for (int s = 0; s < C::D16; ++s) {
            const half8 k0 = *(const half8*)(Kc + k_lane + 16 * s);
            const half8 k1 = *(const half8*)(Kc + k_lane + 32 * C::KS + 16 * s);
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[s], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[s], a1, 0, 0, 0);
        }
    };
    auto pv = [&](const half_t* Vc) {
#pragma unroll
        for (int t = 0; t < C::DT; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const half_t* vp = Vc + v_lane + (16 * kk) * VRS + 32 * t;
                const half4 lo = lds_tr_read(vp), hi = lds_tr_read(vp + 8 * VRS);
                const half8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[kk], o[t], 0, 0, 0);
            }
    };
    // mask (last tile) + running-maximum maintenance for the scores in (s0, s1); touches o, pb when the maximum moves
    auto settle_max = [&](int kv0, f32x16& s0, f32x16& s1) {
        if (kv0 + 64 > p.L) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kvl = (i & 3) + 8 * (i >> 2) + 4 * h;
                if (kv0 + kvl >= p.L) s0[i] = -INFINITY;
                if (kv0 + 32 + kvl >= p.L) s1[i] = -INFINITY;
            }
        }
        float mxa = fmaxf(s0[0], s1[0]), mxb = fmaxf(s0[8], s1[8]);
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, s0[i]), s1[i]);
            mxb = __builtin_fmaxf(__builtin_fmaxf(mxb, s0[8 + i]), s1[8 + i]);
        }
        float mx = fmaxf(mxa, mxb);
        if (__builtin_expect(first || __any(mx > RESCALE_THR), 0)) {
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float delta = first ? mx : fmaxf(mx, 0.f);
            if constexpr (C::FOLD_MAX) {
                const float m_new = (float)(half_t)(m_run + delta);
                delta = m_new - m_run;
                m_run = m_new;
                if (h == C::PAD_H) qf[C::PAD_S][C::PAD_E] = (half_t)(-m_new);
            } else {
                m_run += delta;
#pragma unroll
                for (int i = 0; i < 16; ++i) minit[i] = -m_run;
            }
            const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
            if constexpr (!C::ONES_ROW) l_run *= alpha;
#pragma unroll
            for (int t = 0; t < C::DT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[t][i] *= alpha;
            const half_t ah = (half_t)alpha;                 // P(j-1), packed and not yet multiplied into O
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[kk][e] = pb[kk][e] * ah;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s0[i] -= delta; s1[i] -= delta; }
            first = false;
        }
    };
    auto exp_pack = [&](half8 (&out)[4], f32x16& s0, f32x16& s1) {
        float ls = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s0[i] = __builtin_amdgcn_exp2f(s0[i]);
            s1[i] = __builtin_amdgcn_exp2f(s1[i]);
            if constexpr (!C::ONES_ROW) ls += s0[i] + s1[i];
        }
        if constexpr (!C::ONES_ROW) l_run += ls;
        out[0] = pack8(s0, 0); out[1] = pack8(s0, 8); out[2] = pack8(s1, 0); out[3] = pack8(s1, 8);
    };

    // prologue: tiles 0 and 1 in LDS, tile 2 in flight; P(0) packed, S(1) pending
    load_tile(0);
    store_tile(0, 0);
    if (nt > 1) { load_tile(64); store_tile(1, 1); }
    if (nt > 2) load_tile(128);
    __syncthreads();
    scores(Ks, sa0, sa1);
    settle_max(0, sa0, sa1);
    exp_pack(pb, sa0, sa1);
    if (nt > 1) scores(Ks + KBUF, sa0, sa1);
    if (nt > 2) { store_tile(0, 2); if (nt > 3) load_tile(192); }
    __syncthreads();
    auto one_tile = [&](int j, f32x16& s0, f32x16& s1, f32x16& n0, f32x16& n1) {
        settle_max(j * 64, s0, s1);
        // one basic block: MFMAs of tiles j-1 and j+1 beside the exp / pack of tile j
        half8 pn[4];
        pv(Vs + ((j - 1) & 3) * VBUF);
        scores(Ks + ((j + 1) & 1) * KBUF, n0, n1);         // tile j+1 (a stale slot when j+1 == nt: result unused)
        exp_pack(pn, s0, s1);
        // interleave: per MFMA two fragment reads and a handful of VALU ops (the exp / pack stream) in its shadow
#pragma unroll
        for (int g = 0; g < 8 + 2 * C::D16; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        // opaque use: keeps the exp / pack work in THIS block (LLVM would sink it behind the staging branch)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            typedef int v4i_t __attribute__((ext_vector_type(4)));
            v4i_t tmp = __builtin_bit_cast(v4i_t, pn[kk]);
            asm volatile("" : "+v"(tmp));
            pb[kk] = __builtin_bit_cast(half8, tmp);
        }
        if (j + 2 < nt) {
            store_tile(j & 1, (j + 2) & 3);
            if (j + 3 < nt) load_tile((j + 3) * 64);
        }
        __syncthreads();
    };
    for (int j = 1; j < nt; j += 2) {          // two tiles per trip: the score sets swap roles, no register copies
        one_tile(j, sa0, sa1, sb0, sb1);
        if (j + 1 < nt) one_tile(j + 1, sb0, sb1, sa0, sa1);
    }
    pv(Vs + ((nt - 1) & 3) * VBUF);

    float l_tot;
    if constexpr (C::ONES_ROW) {
        constexpr int LT = D / 32, LR = (D % 32) / 2;
        const float mine = o[LT][LR];
        const float other = __shfl_xor(mine, 32);
        l_tot = h ? other : mine;
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32);
    }
    const float inv = 1.0f / l_tot;
    if (p.lse && q_ok && h == 0) p.lse[((long long)b * p.heads + head) * p.N + q0 + r] = m_run + __log2f(l_tot);
    if (q_ok) {
        half_t* orow = p.Out + ((long long)b * p.N + q0 + r) * p.ldo + head * D;
#pragma unroll
        for (int t = 0; t < C::DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int db = t * 32 + 8 * g + 4 * h;
                if (db < D) {
                    half4 v = {(half_t)(o[t][4 * g] * inv), (half_t)(o[t][4 * g + 1] * inv),
                               (half_t)(o[t][4 * g + 2] * inv), (half_t)(o[t][4 * g + 3] * inv)};
                    *(half4*)(orow + db) = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// cross attention (<= 96 keys) with the P2P edit
// ------------------------------------------------------------------------------------------
#define XL 96   // padded key count: 3 sub-tiles of 32
#define XS 100  // row stride (halves) of the [*, 96]-wide LDS images (V^T, M^T): 200 B

template <int D>
__device__ __forceinline__ void cross_scores(f32x16 (&s)[3], const half_t* Ks, const half8 (&qf)[AttnCfg<D>::D16],
                                             int r, int h, float sc, int L) {
    using C = AttnCfg<D>;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[u][i] = 0.f;
#pragma unroll
    for (int st = 0; st < C::D16; ++st) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const half8 k = *(const half8*)(Ks + (32 * u + r) * C::KS + 16 * st + 8 * h);
            s[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k, qf[st], s[u], 0, 0, 0);
        }
    }
    // normalised softmax over the L valid keys (row = this lane's query; other half in lane^32)
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kv = 32 * u + (i & 3) + 8 * (i >> 2) + 4 * h;
            const float a = kv < L ? s[u][i] * sc : -INFINITY;
            s[u][i] = a;
            mx = fmaxf(mx, a);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float ls = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[u][i] = exp2f(s[u][i] - mx);
            ls += s[u][i];
        }
    ls += __shfl_xor(ls, 32);
    const float inv = 1.0f / ls;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[u][i] *= inv;
}

template <int D>
__global__ __launch_bounds__(256) void attn_cross_p2p_kernel(const IefCrossParams p) {
    using C = AttnCfg<D>;
    // everything the workgroup reads is staged in ONE phase (own K, the edit source's K, V^T, the mapper, the gate rows)
    // and both query fragments are fetched meanwhile: one global-memory latency on the critical path instead of four
    __shared__ __attribute__((aligned(16))) half_t Ks[XL * C::KS];
    __shared__ __attribute__((aligned(16))) half_t Ks2[XL * C::KS];     // K of the edit source row
    __shared__ __attribute__((aligned(16))) half_t Vt[C::DT * 32 * XS];
    __shared__ __attribute__((aligned(16))) half_t Ms[XL * XS];
    __shared__ float coef_s[2 * XL];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int es = p.edit_src ? p.edit_src[b] : -1;
    const int slot = (es >= 0 && p.edit_slot) ? p.edit_slot[b] : 0;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool q_ok = q0 + r < p.N;
    const float sc = p.scale * LOG2E;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // query fragments straight from global memory (no dependence on the staging below)
    half8 qf[C::D16], qs[C::D16];
    load_q_frags<D>(qf, p.Q, ((long long)b * p.N + q0 + r) * p.ldq + head * D, q_ok, h);
    if (es >= 0) load_q_frags<D>(qs, p.Q, ((long long)es * p.N + q0 + r) * p.ldq + head * D, q_ok, h);

    // ALL global loads of the staging are issued before anything waits on one of them (a load -> LDS-store loop exposes
    // one memory latency per iteration: ~27 of them at d = 160)
    constexpr int NCH = (XL * C::CPR + 255) / 256;      // 16-byte chunks of a K / V image per thread
    constexpr int NM = (XL * XL / 8 + 255) / 256;       // ... of the mapper
    half8 kreg[NCH], vreg[NCH], k2reg[NCH], mreg[NM];
    float creg = 0.f;
    {
        const half_t* Kb = p.K + (long long)b * p.L * p.ldk + head * D;
        const half_t* Vb = p.V + (long long)b * p.L * p.ldv + head * D;
        const half_t* K2b = p.K + (long long)(es >= 0 ? es : b) * p.L * p.ldk + head * D;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + 256 * i, row = c / C::CPR, ch = c - row * C::CPR;
            const bool ok = c < XL * C::CPR && row < p.L;
            kreg[i] = ok ? *(const half8*)(Kb + (long long)row * p.ldk + ch * 8) : zero8;
            vreg[i] = ok ? *(const half8*)(Vb + (long long)row * p.ldv + ch * 8) : zero8;
            k2reg[i] = (ok && es >= 0) ? *(const half8*)(K2b + (long long)row * p.ldk + ch * 8) : zero8;
        }
        if (es >= 0) {
            const half_t* Mg = p.MT + (long long)slot * XL * XL;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                const int c = tid + 256 * i;
                mreg[i] = c < XL * XL / 8 ? *(const half8*)(Mg + c * 8) : zero8;
            }
            if (tid < 2 * XL) creg = p.coef[(long long)slot * 2 * XL + tid];
        }
    }

    for (int i = tid; i < XL * C::KS / 8; i += 256) { ((half8*)Ks)[i] = zero8; ((half8*)Ks2)[i] = zero8; }
    for (int i = tid; i < C::DT * 32 * XS / 4; i += 256) ((half4*)Vt)[i] = (half4){0, 0, 0, 0};
    __syncthreads();

#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + 256 * i, row = c / C::CPR, ch = c - row * C::CPR;
        if (c < XL * C::CPR && row < p.L) {
            *(half8*)(Ks + row * C::KS + ch * 8) = kreg[i];
            if (es >= 0) *(half8*)(Ks2 + row * C::KS + ch * 8) = k2reg[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[(ch * 8 + e) * XS + row] = vreg[i][e];
        }
    }
    if (es >= 0) {  // block-uniform
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int c = tid + 256 * i;
            if (c < XL * XL / 8) {
                const int row = c / (XL / 8), ch = c - row * (XL / 8);
                const half8 v = mreg[i];
                *(half4*)(Ms + row * XS + ch * 8) = (half4){v[0], v[1], v[2], v[3]};
                *(half4*)(Ms + row * XS + ch * 8 + 4) = (half4){v[4], v[5], v[6], v[7]};
            }
        }
        if (tid < 2 * XL) coef_s[tid] = creg;
    }
    __syncthreads();

    f32x16 pm[3];
    if (es >= 0) {
        f32x16 ps[3];
        cross_scores<D>(ps, Ks2, qs, r, h, sc, p.L);
        // PM^T[n][q] = sum_w M^T[n][w] P_src^T[w][q]
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) pm[u][i] = 0.f;
#pragma unroll
        for (int u2 = 0; u2 < 3; ++u2) {
            const half8 pb0 = pack8(ps[u2], 0), pb1 = pack8(ps[u2], 8);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const half_t* mrow = Ms + (32 * u + r) * XS;
                pm[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(read_perm_frag(mrow, 32 * u2, h), pb0, pm[u], 0, 0, 0);
                pm[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(read_perm_frag(mrow, 32 * u2 + 16, h), pb1, pm[u], 0, 0, 0);
            }
        }
    }
    f32x16 ps[3];
    cross_scores<D>(ps, Ks, qf, r, h, sc, p.L);
    if (es >= 0) {
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = 32 * u + (i & 3) + 8 * (i >> 2) + 4 * h;
                ps[u][i] = coef_s[n] * pm[u][i] + coef_s[XL + n] * ps[u][i];
            }
    }
    f32x16 o[C::DT];
#pragma unroll
    for (int t = 0; t < C::DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const half8 pb0 = pack8(ps[u], 0), pb1 = pack8(ps[u], 8);
#pragma unroll
        for (int t = 0; t < C::DT; ++t) {
            const half_t* vrow = Vt + (t * 32 + r) * XS;
            o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(read_perm_frag(vrow, 32 * u, h), pb0, o[t], 0, 0, 0);
            o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(read_perm_frag(vrow, 32 * u + 16, h), pb1, o[t], 0, 0, 0);
        }
    }
    if (q_ok) {
        half_t* orow = p.Out + ((long long)b * p.N + q0 + r) * p.ldo + head * D;
#pragma unroll
        for (int t = 0; t < C::DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int db = t * 32 + 8 * g + 4 * h;
                if (db < D) {
                    half4 v = {(half_t)o[t][4 * g], (half_t)o[t][4 * g + 1], (half_t)o[t][4 * g + 2], (half_t)o[t][4 * g + 3]};
                    *(half4*)(orow + db) = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// generic path: materialised probabilities, and probabilities x V
// ------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_probs_kernel(const IefAttnParams p, half_t* __restrict__ probs) {
    using C = AttnCfg<D>;
    __shared__ __attribute__((aligned(16))) half_t Ks[64 * C::KS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool q_ok = q0 + r < p.N;
    const float sc = p.scale * LOG2E;
    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < 64 * C::KS / 8; i += 256) ((half8*)Ks)[i] = zero8;
    half8 qf[C::D16];
    load_q_frags<D>(qf, p.Q, ((long long)b * p.N + q0 + r) * p.ldq + head * D, q_ok, h);
    const half_t* Kb = p.K + (long long)b * p.L * p.ldk + head * D;
    half_t* prow = probs + (((long long)b * p.heads + head) * p.N + q0 + r) * p.L;
    const int nt = (p.L + 63) / 64;
    float m_run = -INFINITY, l_run = 0.f;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        const float inv = pass ? 1.0f / (l_run + __shfl_xor(l_run, 32)) : 0.f;
        for (int j = 0; j < nt; ++j) {
            const int kv0 = j * 64;
            for (int c = tid; c < 64 * C::CPR; c += 256) {
                const int row = c / C::CPR, ch = c - row * C::CPR;
                *(half8*)(Ks + row * C::KS + ch * 8) =
                    kv0 + row < p.L ? *(const half8*)(Kb + (long long)(kv0 + row) * p.ldk + ch * 8) : zero8;
            }
            __syncthreads();
            f32x16 s0, s1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
            for (int s = 0; s < C::D16; ++s) {
                const half8 k0 = *(const half8*)(Ks + r * C::KS + 16 * s + 8 * h);
                const half8 k1 = *(const half8*)(Ks + (32 + r) * C::KS + 16 * s + 8 * h);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[s], s1, 0, 0, 0);
            }
            if (pass == 0) {
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int kvl = (i & 3) + 8 * (i >> 2) + 4 * h;
                    s0[i] = kv0 + kvl < p.L ? s0[i] * sc : -INFINITY;
                    s1[i] = kv0 + 32 + kvl < p.L ? s1[i] * sc : -INFINITY;
                    mx = fmaxf(mx, fmaxf(s0[i], s1[i]));
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                const float m_new = fmaxf(m_run, mx);
                float ls = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) ls += exp2f(s0[i] - m_new) + exp2f(s1[i] - m_new);
                l_run = l_run * exp2f(m_run - m_new) + ls;
                m_run = m_new;
            } else if (q_ok) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int kv = kv0 + 32 * u + 8 * g + 4 * h + e;
                            const float sv = u ? s1[4 * g + e] : s0[4 * g + e];
                            if (kv < p.L) prow[kv] = (half_t)(exp2f(sv * sc - m_run) * inv);
                        }
            }
            __syncthreads();
        }
    }
}

template <int D>
__global__ __launch_bounds__(256) void attn_apply_kernel(const IefAttnParams p, const half_t* __restrict__ probs) {
    using C = AttnCfg<D>;
    constexpr int VS = 68;
    __shared__ __attribute__((aligned(16))) half_t Vt[C::DT * 32 * VS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bool q_ok = q0 + r < p.N;
    for (int i = tid; i < C::DT * 32 * VS / 4; i += 256) ((half4*)Vt)[i] = (half4){0, 0, 0, 0};
    const half_t* Vb = p.V + (long long)b * p.L * p.ldv + head * D;
    const half_t* prow = probs + (((long long)b * p.heads + head) * p.N + (q_ok ? q0 + r : 0)) * p.L;
    f32x16 o[C::DT];
#pragma unroll
    for (int t = 0; t < C::DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[t][i] = 0.f;
    const int nt = (p.L + 63) / 64;
    __syncthreads();
    for (int j = 0; j < nt; ++j) {
        const int kv0 = j * 64;
        for (int c = tid; c < 64 * C::CPR; c += 256) {
            const int row = c / C::CPR, ch = c - row * C::CPR;
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kv0 + row < p.L) v = *(const half8*)(Vb + (long long)(kv0 + row) * p.ldv + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) Vt[(ch * 8 + e) * VS + row] = v[e];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            half8 pb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kv = kv0 + 16 * s + 8 * h + e;
                pb[e] = (q_ok && kv < p.L) ? prow[kv] : (half_t)0;
            }
#pragma unroll
            for (int t = 0; t < C::DT; ++t)
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(read_nat_frag(Vt + (t * 32 + r) * VS, 16 * s, h), pb, o[t], 0, 0, 0);
        }
        __syncthreads();
    }
    if (q_ok) {
        half_t* orow = p.Out + ((long long)b * p.N + q0 + r) * p.ldo + head * D;
#pragma unroll
        for (int t = 0; t < C::DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int db = t * 32 + 8 * g + 4 * h;
                if (db < D) {
                    half4 v = {(half_t)o[t][4 * g], (half_t)o[t][4 * g + 1], (half_t)o[t][4 * g + 2], (half_t)o[t][4 * g + 3]};
                    *(half4*)(orow + db) = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------
static int check_attn(int B, int heads, int N, int L, int d, int ldq, int ldk, int ldv, int ldo, const void* Q,
                      const void* K, const void* V, const void* O) {
    if (!Q || !K || !V || !O) return IEF_EINVAL;
    if (B <= 0 || heads <= 0 || N <= 0 || L <= 0) return IEF_ESHAPE;
    if (d != 32 && d != 40 && d != 64 && d != 80 && d != 160) return IEF_ESHAPE;
    if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3)) return IEF_EALIGN;
    if (heads * d > ldq || heads * d > ldk || heads * d > ldv || heads * d > ldo) return IEF_ESHAPE;
    return IEF_OK;
}

#define DISPATCH_D(d, CALL)                   \
    switch (d) {                              \
        case 32: { constexpr int DD = 32; CALL; } break;   \
        case 40: { constexpr int DD = 40; CALL; } break;   \
        case 64: { constexpr int DD = 64; CALL; } break;   \
        case 80: { constexpr int DD = 80; CALL; } break;   \
        case 160: { constexpr int DD = 160; CALL; } break; \
        default: return IEF_ESHAPE;           \
    }

extern "C" int ief_attn_flash_f16(const IefAttnParams* pp, void* stream) {
    if (!pp) return IEF_EINVAL;
    const IefAttnParams p = *pp;
    int rc = check_attn(p.B, p.heads, p.N, p.L, p.d, p.ldq, p.ldk, p.ldv, p.ldo, p.Q, p.K, p.V, p.Out);
    if (rc) return rc;
    // IefAttnParams.variant: 0 = default, the software-pipelined kernel (fastest on every shape of the three workloads,
    // DESIGN.md section 3); 1 forces the plain 4-wave kernel, 2 the 8-wave ping-pong kernel, 3 = 0.  All three produce
    // the same softmax within fp16 rounding and are held to the same oracle by tests/test_gpu_ops.py.
    if (p.variant == 2) {
        dim3 gridp((unsigned)((long long)((p.N + 255) / 256) * p.heads * p.B));
        DISPATCH_D(p.d, hipLaunchKernelGGL((attn_flash_pp_kernel<DD>), gridp, dim3(512), 0, (hipStream_t)stream, p));
        IEF_LAUNCH_CHECK();
        return IEF_OK;
    }
    dim3 grid(((p.N + 127) / 128) * p.heads * p.B);
    if (p.variant == 1) {
        DISPATCH_D(p.d, hipLaunchKernelGGL((attn_flash_kernel<DD>), grid, dim3(256), 0, (hipStream_t)stream, p));
        IEF_LAUNCH_CHECK();
        return IEF_OK;
    }
    if (p.variant != 0 && p.variant != 3) return IEF_EINVAL;
    DISPATCH_D(p.d, hipLaunchKernelGGL((attn_flash_sp_kernel<DD>), grid, dim3(256), 0, (hipStream_t)stream, p));
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// diagnostics: resident workgroups per CU the runtime grants attn_flash_kernel<d> (occupancy of the hot attention kernel)
extern "C" int ief_attn_flash_occupancy(int d) {
    int n = -1;
    hipError_t e = hipErrorInvalidValue;
    if (d == 40) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_flash_kernel<40>, 256, 0);
    else if (d == 80) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_flash_kernel<80>, 256, 0);
    else if (d == 160) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_flash_kernel<160>, 256, 0);
    return e == hipSuccess ? n : -(int)e;
}
extern "C" int ief_attn_flash_attrs(int* out) {   // {numRegs, sharedSizeBytes, maxThreadsPerBlock, localSizeBytes, device LDS per CU}
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, (const void*)attn_flash_kernel<40>) != hipSuccess) return -1;
    out[0] = a.numRegs; out[1] = (int)a.sharedSizeBytes; out[2] = a.maxThreadsPerBlock; out[3] = (int)a.localSizeBytes;
    int v = 0; hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0); out[4] = v;
    for (int dyn = 0; dyn < 6; ++dyn) {
        int n = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, attn_flash_kernel<40>, 256, dyn * 8192);
        out[5 + dyn] = n;
    }
    return 0;
}

extern "C" int ief_attn_cross_p2p_f16(const IefCrossParams* pp, void* stream) {
    if (!pp) return IEF_EINVAL;
    const IefCrossParams p = *pp;
    int rc = check_attn(p.B, p.heads, p.N, p.L, p.d, p.ldq, p.ldk, p.ldv, p.ldo, p.Q, p.K, p.V, p.Out);
    if (rc) return rc;
    if (p.L > XL) return IEF_ESHAPE;
    if (p.edit_src && (!p.MT || !p.coef)) return IEF_EINVAL;
    dim3 grid((p.N + 127) / 128, p.heads, p.B);
    DISPATCH_D(p.d, hipLaunchKernelGGL((attn_cross_p2p_kernel<DD>), grid, dim3(256), 0, (hipStream_t)stream, p));
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

extern "C" int ief_attn_probs_f16(const IefAttnParams* pp, ief_half* probs, void* stream) {
    if (!pp || !probs) return IEF_EINVAL;
    const IefAttnParams p = *pp;
    int rc = check_attn(p.B, p.heads, p.N, p.L, p.d, p.ldq, p.ldk, p.ldk, p.ldq, p.Q, p.K, p.K, probs);
    if (rc) return rc;
    dim3 grid((p.N + 127) / 128, p.heads, p.B);
    DISPATCH_D(p.d, hipLaunchKernelGGL((attn_probs_kernel<DD>), grid, dim3(256), 0, (hipStream_t)stream, p, probs));
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

extern "C" int ief_attn_apply_f16(const IefAttnParams* pp, const ief_half* probs, void* stream) {
    if (!pp || !probs) return IEF_EINVAL;
    const IefAttnParams p = *pp;
    int rc = check_attn(p.B, p.heads, p.N, p.L, p.d, p.ldv, p.ldv, p.ldv, p.ldo, probs, probs, p.V, p.Out);
    if (rc) return rc;
    dim3 grid((p.N + 127) / 128, p.heads, p.B);
    DISPATCH_D(p.d, hipLaunchKernelGGL((attn_apply_kernel<DD>), grid, dim3(256), 0, (hipStream_t)stream, p, probs));
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
