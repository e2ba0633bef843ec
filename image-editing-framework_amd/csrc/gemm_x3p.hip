// Split-operand contractions on PRE-SPLIT PLANES ("x3p"): the f16x3 mode's GEMM / implicit-GEMM convolution with BOTH operands
// staged by LDS-DMA and no vector work in the K loop.
//
// split_x3.hip splits every fp32 activation element into (hi, lo) fp16 halves INSIDE the GEMM, while its tile is staged: global
// -> registers -> 10 VALU per four elements -> ds_write, redone for every column tile and every 3x3 tap (36 times per element
// of a convolution input), and the phases of a wave (fragment reads, 30 MFMAs, wait, split, LDS writes, barrier) run one after
// the other: the matrix pipe was busy 49 % (convolutions) / 22 % (K = C projections) of the time.  Here the PRODUCER of an
// activation writes the two planes (GroupNorm / LayerNorm apply, GEGLU epilogue, attention epilogues, this kernel's own
// epilogue; the same 4 bytes per element as fp32), weights are split once per tensor, and the K loop is the fp16 kernel's
// (gemm_conv.hip): `global_load_lds` pieces into an NS-deep ring, counted vmcnt + raw s_barrier, fragments by ds_read_b128,
// three v_mfma_f32_16x16x32_f16 per fragment pair (Wl Ah + Wh Al + Wh Ah into one fp32 accumulator).
//
//   K tile = 32 halves per plane (64-byte LDS rows: two planes of both operands make a 128 x 160 tile 36 KiB per ring slot);
//   a ring slot is [A hi | A lo | W hi | W lo] as 16-row PIECES of 1 KiB = one LDS-DMA wave-instruction (lane l -> row l >> 2,
//   16-byte slot l & 3); the slot of chunk c of row r is c ^ (2 * bit 2 of r): conflict-free ds_read_b128
//   fragments of 16 consecutive rows (checked against the instruction's four 16-lane groups), applied to the per-lane SOURCE
//   address (the LDS image of a piece is lane-linear).  Pieces are dealt round-robin to the staging waves (all waves, or NL
//   loader waves); a wave whose share is short issues filler pieces from the zero page into a scratch kilobyte so that every
//   vmcnt count is a compile-time constant.
//   The MFMA's A operand is the WEIGHT block, its B operand the activation block: a lane owns one output row and four
//   consecutive columns of a 16-column block; blocks are PAIRED by the staging's choice of weight rows (xp_perm_col, x3p_common.h),
//   so a lane's two accumulators of a pair are EIGHT consecutive columns -- fp32 results leave as 32 contiguous bytes per lane,
//   planes as 16-byte stores (GEGLU: hidden and gate of a column in one lane); no LDS pass in the epilogue.
//   Tiles (IefGemmX3pParams.tile): 128 x 160 on 8 waves (+ 4 loader waves; or 4 waves, two workgroups per CU), 128 x 80, 256 x 160,
//   64 x 160, 128 x 64, and 256 x 320 (8 waves of 64 x 160, weight fragments one block at a time: the fewest staged bytes per
//   FLOP -- the per-CU LDS-DMA fill rate is what bounds these kernels).
//
// Replaces on the reference path: the same call sites as gemm_conv.hip (linears /root/reference/p2p/model/register.py:33-54,
// ResnetBlock2D convolutions /root/reference/pnp/model/register.py:139-175).
#include "ief_common.h"
#include "ief_params.h"
#include "x3_common.h"

#include "x3p_common.h"

template <int BM, int BN, int WAVES_M, int WAVES_N, int NS, int NL, bool CONV>
__global__ __launch_bounds__(64 * (WAVES_M * WAVES_N + NL)) void igemm_x3p_kernel(const IefGemmX3pParams p) {
    constexpr int BK = XP_BK;
    constexpr int NWC = WAVES_M * WAVES_N;          // compute waves
    constexpr int NSW = NL > 0 ? NL : NWC;          // staging waves
    constexpr int PA = BM / 16, PB = BN / 16;       // 16-row pieces per plane of A / W
    constexpr int P = 2 * (PA + PB);                // pieces per K tile: [A hi][A lo][W hi][W lo]
    constexpr int G = (P + NSW - 1) / NSW;          // LDS-DMA instructions a staging wave issues per K tile
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
    constexpr int SLOT = P * 16 * BK;               // halves per ring slot
    constexpr int DUMP = NS * SLOT;                 // where filler pieces land
    static_assert(BM % 16 == 0 && BN % 16 == 0 && WM % 16 == 0 && WN % 16 == 0, "tile");
    static_assert(NS >= 2 && NS <= 4, "ring depth");
    static_assert((NS * SLOT + 512) * 2 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) half_t smem[NS * SLOT + 512];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const bool stages_ = NL == 0 || wave >= NWC;
    const bool computes = NL == 0 || wave < NWC;
    const int swave = NL > 0 ? wave - NWC : wave;
    // tile order: each XCD takes a contiguous run of logical ids; inside a run, groups of XP_GROUP_M row blocks are walked
    // column by column (the ~32-64 workgroups resident on an XCD cover a near-square patch of the output: an activation row
    // block and a weight column block are each fetched into that L2 once per patch)
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, ntm * ntn);
    const int grp = bid / (XP_GROUP_M * ntn), within = bid - grp * (XP_GROUP_M * ntn);
    const int gsz = min(XP_GROUP_M, ntm - grp * XP_GROUP_M);
    const int tn = within / gsz, tm = grp * XP_GROUP_M + (within - tn * gsz);
    const int m0 = tm * BM, n0 = tn * BN;
    const char* __restrict__ zp = (const char*)p.zeros;

    const int nk_all = p.K / BK;
    int kt_lo = 0, nk = nk_all;
    if (p.splits > 1) {
        const int per = (nk_all + p.splits - 1) / p.splits;
        kt_lo = min(nk_all, (int)blockIdx.y * per);
        nk = min(nk_all, kt_lo + per);
    }

    // ---- staging state: one 64-bit source pointer per piece (plane offset included), advanced by one K tile (64 B) per stage.
    // Out-of-range rows (M / N tails, padded taps) and filler pieces point at the zero page and do not advance.
    const int prow = lane >> 2;                                                     // row of this lane inside a piece
    const unsigned kcb = (unsigned)(XP_LANE_CHUNK(lane) * 16);                         // chunk FETCHED for LDS slot lane & 3
    const char* ptr[G];
    unsigned step[G];
    bool exists[G], isA[G];          // wave-uniform
    int plane[G];
    // conv: per A piece the output pixel of this lane's row
    bool a_ok[G], a_live[G];
    int a_y[G], a_x[G], a_b[G];
    unsigned a_m[G], r1[G], r2[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int q = swave + NSW * j;
        exists[j] = stages_ && q < P;
        isA[j] = q < 2 * PA;
        int blk;
        if (isA[j]) { plane[j] = q / PA; blk = q - plane[j] * PA; }
        else { const int q2 = q - 2 * PA; plane[j] = q2 / PB; blk = q2 - plane[j] * PB; }
        ptr[j] = zp; step[j] = 0;
        a_ok[j] = a_live[j] = false; a_y[j] = a_x[j] = a_b[j] = 0; a_m[j] = r1[j] = r2[j] = 0;
        if (!exists[j]) continue;
        if (isA[j]) {
            const int m = m0 + 16 * blk + prow;
            a_ok[j] = m < p.M;
            a_m[j] = (unsigned)m;
            if constexpr (CONV) {
                const int hw = p.Ho * p.Wo;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                a_b[j] = b;
                a_y[j] = oy * p.stride - (p.pad_hi_only ? 0 : 1);
                a_x[j] = ox * p.stride - (p.pad_hi_only ? 0 : 1);
            } else if (a_ok[j]) {
                ptr[j] = (const char*)p.A + ((long long)plane[j] * p.planeA + (long long)m * p.lda) * 2 + kcb + (long long)kt_lo * (BK * 2);
                step[j] = BK * 2;
            }
        } else {
            const int n = n0 + xp_perm_col<WN>(16 * blk + prow, p.geglu != 0);     // column order: x3p_common.h
            if (n < p.N) {
                ptr[j] = (const char*)p.W + ((long long)plane[j] * p.planeW + (long long)n * p.ldw) * 2 + kcb + (long long)kt_lo * (BK * 2);
                step[j] = BK * 2;
            }
        }
    }

    unsigned stepmask = 0;            // bit j: piece j advances (TN > 5, linear: replaces step[])
#pragma unroll
    for (int j = 0; j < G; ++j) stepmask |= (step[j] != 0 ? 1u : 0u) << j;
    // ---- conv K iterator: (tap, channel offset) advanced by one K tile per stage; tap 9 = the fused 1x1 range
    const int Ctot = p.C1 + p.C2;
    const int Hp = p.H >> p.ups, Wp = p.Wd >> p.ups;
    int it_tap = 0, it_c = 0;
    auto set_tap = [&](int tap) {
        if (tap < 9) {
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (!(exists[j] && isA[j])) continue;
                const int iy = a_y[j] + ky, ix = a_x[j] + kx;
                a_live[j] = a_ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                const unsigned pix = (unsigned)((a_b[j] * Hp + (iy >> p.ups)) * Wp + (ix >> p.ups));
                r1[j] = pix * (unsigned)p.C1 * 2u + kcb;
                r2[j] = pix * (unsigned)p.C2 * 2u + kcb;
            }
        } else {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (!(exists[j] && isA[j])) continue;
                a_live[j] = a_ok[j];
                r1[j] = a_m[j] * (unsigned)p.CE1 * 2u + kcb;
                r2[j] = a_m[j] * (unsigned)p.CE2 * 2u + kcb;
            }
        }
    };
    auto set_src = [&]() {
        const bool tapm = it_tap < 9;
        const int c1 = tapm ? p.C1 : p.CE1;
        const bool first = it_c < c1;
        const char* src = (const char*)(tapm ? (first ? p.A : p.A2) : (first ? p.E1 : p.E2));
        const long long pl = tapm ? (first ? p.planeA : p.planeA2) : (first ? p.planeE1 : p.planeE2);
        const unsigned cc = (unsigned)(first ? it_c : it_c - c1) * 2u;
#pragma unroll
        for (int j = 0; j < G; ++j) {
            if (!(exists[j] && isA[j])) continue;
            ptr[j] = a_live[j] ? src + (long long)plane[j] * pl * 2 + ((first ? r1[j] : r2[j]) + cc) : zp;
            step[j] = a_live[j] ? BK * 2 : 0;
        }
    };

    bool abl_prologue = true;
    auto stage_tile = [&](int buf) {
        if (!stages_) return;
        half_t* base = smem + buf * SLOT + swave * (16 * BK);
#pragma unroll
        for (int j = 0; j < G; ++j) {
            half_t* dst = (swave + NSW * j < P) ? base + NSW * j * (16 * BK) : smem + DUMP;        // wave-uniform
            if (!(XP_ABL & 1) || abl_prologue) glds16(ptr[j], dst);
            if constexpr (TN > 5 && !CONV) ptr[j] += ((stepmask >> j) & 1u) ? BK * 2 : 0;      // the wide tile has no register for step[]
            else ptr[j] += step[j];
        }
        if constexpr (CONV) {
            it_c += BK;
            if (it_tap < 9) {
                if (it_c >= Ctot) { it_c = 0; ++it_tap; set_tap(it_tap); set_src(); }
                else if (it_c == p.C1) set_src();
            } else if (it_c == p.CE1 && p.CE2 > 0) {
                set_src();
            }
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (CONV) {
        const int k0 = kt_lo * BK;
        if (k0 < 9 * Ctot) { it_tap = k0 / Ctot; it_c = k0 - it_tap * Ctot; }
        else { it_tap = 9; it_c = k0 - 9 * Ctot; }
        set_tap(it_tap);
        set_src();
    }
    const int fr = lane & 15, fq = lane >> 4;
    int aoff[TM], boff[TN];          // halves inside a ring slot, hi plane; the lo plane lies BM (BN) rows further
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wr * WM + i * 16 + fr;
        aoff[i] = row * BK + ((fq ^ xp_swz(row)) << 3);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wc * WN + j * 16 + fr;
        boff[j] = 2 * BM * BK + row * BK + ((fq ^ xp_swz(row)) << 3);
    }
    half8 ah[TM], al[TM], bh[TN], bl[TN];
    bool abl_first = true;
    auto read_frags = [&](const half_t* S) {
        if ((XP_ABL & 2) && !abl_first) return;
        abl_first = false;
#pragma unroll
        for (int i = 0; i < TM; ++i) { ah[i] = *(const half8*)(S + aoff[i]); al[i] = *(const half8*)(S + aoff[i] + BM * BK); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { bh[j] = *(const half8*)(S + boff[j]); bl[j] = *(const half8*)(S + boff[j] + BN * BK); }
    };
    auto mfma_frags = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (!(XP_ABL & 4)) {
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[i], acc[j][i], 0, 0, 0);
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[i], acc[j][i], 0, 0, 0);
                }
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[i], acc[j][i], 0, 0, 0);
            }
    };
    // wide wave tiles (TN > 5: the 256 x 320 tile): the weight fragments of ONE 16-column block at a time (8 registers) instead of all TN
    // (80): 160 accumulator registers leave no room for more
    auto read_mfma_wide = [&](const half_t* S) {
#pragma unroll
        for (int i = 0; i < TM; ++i) { ah[i] = *(const half8*)(S + aoff[i]); al[i] = *(const half8*)(S + aoff[i] + BM * BK); }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const half8 wh = *(const half8*)(S + boff[j]), wl = *(const half8*)(S + boff[j] + BN * BK);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[i], acc[j][i], 0, 0, 0);
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[i], acc[j][i], 0, 0, 0);
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[i], acc[j][i], 0, 0, 0);
            }
        }
    };
    // NS-deep ring, as igemm_f16_kernel: wait for this wave's share of tile k (counted vmcnt), s_barrier (everybody's has
    // landed; everybody is done reading the slot tile k - 1 used), refill that slot with tile k + NS - 1, multiply tile k.
    // 8 compute waves, NS >= 3: waves 4..7 run half a tile late (they multiply tile k - 1 from fragments read before the
    // barrier while waves 0..3 issue DMA and read tile k), so each SIMD has one wave on the matrix pipe while its partner
    // issues LDS-DMA / fragment reads.
    constexpr bool PINGPONG = NWC == 8 && NS >= 3;
    const bool late = PINGPONG && wave >= 4 && wave < NWC;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (kt_lo + s < nk) stage_tile(s);
    abl_prologue = false;
    if (NL > 0 && !computes) {
        for (int kt = kt_lo; kt < nk; kt += NS) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int k = kt + s;
                if (k < nk) {
                    if (nk - 1 - k >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (NS - 2)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    XP_BARRIER();
                    if (k + NS - 1 < nk) stage_tile((s + NS - 1) % NS);
                }
            }
        }
    } else if (!late) {
        for (int kt = kt_lo; kt < nk; kt += NS) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int k = kt + s;
                if (k < nk) {
                    if constexpr (NL == 0) {
                        if (nk - 1 - k >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (NS - 2)) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    XP_BARRIER();
                    if constexpr (NL == 0) { if (k + NS - 1 < nk) stage_tile((s + NS - 1) % NS); }
                    if constexpr (TN > 5) {
                        read_mfma_wide(smem + s * SLOT);
                    } else {
                        read_frags(smem + s * SLOT);
                        mfma_frags();
                    }
                }
            }
        }
    } else {
        for (int kt = kt_lo; kt < nk; kt += NS) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int k = kt + s;
                if (k < nk) {
                    if constexpr (NL == 0) {
                        if (nk - 1 - k >= NS - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G * (NS - 2)) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    XP_BARRIER();
                    if (k > kt_lo) mfma_frags();
                    if constexpr (NL == 0) { if (k + NS - 1 < nk) stage_tile((s + NS - 1) % NS); }
                    read_frags(smem + s * SLOT);
                }
            }
        }
        if (nk > kt_lo) mfma_frags();
    }
    if (!computes) return;

    // ---------------- epilogue: blocks are paired (xp_perm_col): a lane holds row m and, of pair P, the EIGHT columns
    // cb + 32 P + 8 fq .. + 7 (accumulator 2 P: the first four, 2 P + 1: the last four); of an odd last block columns 16 j + 4 fq .. + 3
    constexpr int NP2 = TN / 2;
    const int cb = n0 + wc * WN;
    const float inv = p.inv_scale;
    if (p.splits > 1) {
        float* slab = p.ws + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wr * WM + i * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = j < 2 * NP2 ? cb + 32 * (j >> 1) + 8 * fq + 4 * (j & 1) : cb + j * 16 + 4 * fq;
                if (n < p.N) *(f32x4*)(slab + (long long)m * p.N + n) = acc[j][i] * inv;
            }
        }
        return;
    }
    if (p.geglu) {
        // host layout of the weight rows: [8 hidden | 8 gate] per 16.  Pairs (xp_perm_col): accumulator 2 P = hidden pre-activations of
        // output columns cb / 2 + 16 P + 4 fq .. + 3, accumulator 2 P + 1 = their gates, in the same lane: out = h * gelu(g), no exchange.
        // Odd last block: lanes 0-31 hold its 8 hidden columns (4 per lane), lanes 32-63 the gates; each half finishes two of the
        // partner pair's four outputs (one exchange of two values with lane ^ 32)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wr * WM + i * 16 + fr;
            float mu = 0.f, rs = 1.f;
            if (p.rstat_in && m < p.M) xp_ln_row(p, m, mu, rs);
#pragma unroll
            for (int P = 0; P < NP2; ++P) {
                const int nh = cb + 32 * P + 16 * (fq >> 1) + 4 * (fq & 1), ng = nh + 8;       // weight rows of this lane's hidden / gate columns
                if (m >= p.M || nh >= p.N) continue;
                f32x4 hv = acc[2 * P][i] * inv, gv = acc[2 * P + 1][i] * inv;
                if (p.rstat_in) {
                    hv = (hv - *(const f32x4*)(p.colsum + nh) * mu) * rs;
                    gv = (gv - *(const f32x4*)(p.colsum + ng) * mu) * rs;
                }
                if (p.bias) { hv += *(const f32x4*)(p.bias + nh); gv += *(const f32x4*)(p.bias + ng); }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = hv[e] * gelu_nb(gv[e]) * p.out_scale;
                const int no = cb / 2 + 16 * P + 4 * fq;
                if (p.Out) *(f32x4*)(p.Out + (long long)m * p.ldo + no) = o;
                if (p.OutP) {
                    half4 h, l;
                    split4(o, 1.0f, h, l);
                    half_t* op = p.OutP + (long long)m * p.ldp + no;
                    *(half4*)op = h;
                    *(half4*)(op + p.planeO) = l;
                }
            }
            if constexpr (TN & 1) {
                constexpr int j = TN - 1;
                const int nb = cb + j * 16, n = nb + 4 * fq;
                const bool lo_half = fq < 2;
                f32x4 v = acc[j][i] * inv;
                if (p.rstat_in && n < p.N) v = (v - *(const f32x4*)(p.colsum + n) * mu) * rs;
                if (p.bias && n < p.N) v += *(const f32x4*)(p.bias + n);
                // low lane: hidden 4 fq .. + 3, finishes columns + 0, + 1 (needs gates 0, 1 of lane + 32); high lane: gates, finishes
                // columns + 2, + 3 of its partner (needs the partner's hidden 2, 3)
                const float x0 = __shfl_xor(lo_half ? v[2] : v[0], 32), x1 = __shfl_xor(lo_half ? v[3] : v[1], 32);
                const float o0 = (lo_half ? v[0] * gelu_nb(x0) : x0 * gelu_nb(v[2])) * p.out_scale;
                const float o1 = (lo_half ? v[1] * gelu_nb(x1) : x1 * gelu_nb(v[3])) * p.out_scale;
                const int no = nb / 2 + (lo_half ? 4 * fq : 4 * (fq - 2) + 2);
                if (m < p.M && nb < p.N) {
                    if (p.Out) *(f32x2*)(p.Out + (long long)m * p.ldo + no) = f32x2{o0, o1};
                    if (p.OutP) {
                        const half2_t h = __builtin_convertvector(f32x2{o0, o1}, half2_t);
                        float r0, r1;
                        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(o0), "v"(h));
                        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(o1), "v"(h));
                        half_t* op = p.OutP + (long long)m * p.ldp + no;
                        *(half2_t*)op = h;
                        *(half2_t*)(op + p.planeO) = __builtin_convertvector(f32x2{r0, r1}, half2_t);
                    }
                }
            }
        }
        return;
    }
    const int slots_out = (p.N + WN - 1) / WN;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wr * WM + i * 16 + fr;
        const bool mok = m < p.M;
        float mu = 0.f, rs = 1.f;
        if (p.rstat_in && mok) xp_ln_row(p, m, mu, rs);
        f32x4 vv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) vv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int P = 0; P < NP2; ++P) {
            const int n = cb + 32 * P + 8 * fq;
            if (!mok || n >= p.N) continue;
            f32x4 v0 = acc[2 * P][i] * inv, v1 = acc[2 * P + 1][i] * inv;
            if (n + 4 < p.N) {
                if (p.rstat_in) {                                                          // LayerNorm folded
                    v0 = (v0 - *(const f32x4*)(p.colsum + n) * mu) * rs;
                    v1 = (v1 - *(const f32x4*)(p.colsum + n + 4) * mu) * rs;
                }
                xp_store8(p, v0, v1, m, n);
                vv[2 * P] = v0; vv[2 * P + 1] = v1;
            } else {
                if (p.rstat_in) v0 = (v0 - *(const f32x4*)(p.colsum + n) * mu) * rs;
                vv[2 * P] = xp_store(p, v0, m, n);
            }
        }
        if constexpr (TN & 1) {
            constexpr int j = TN - 1;
            const int n = cb + j * 16 + 4 * fq;
            if (mok && n < p.N) {
                f32x4 v = acc[j][i] * inv;
                if (p.rstat_in) v = (v - *(const f32x4*)(p.colsum + n) * mu) * rs;
                vv[j] = xp_store(p, v, m, n);
            }
        }
        if (p.rstat_out) {        // (mean, M2) of this wave's WN-column slice of row m (host-checked: N % WN == 0): every lane shuffles
            float sm, sq;
            xp_row_stats<TN>(vv, sm, sq);
            if (fq == 0 && mok && cb < p.N) {
                float* ro = p.rstat_out + ((long long)m * slots_out + cb / WN) * 2;
                ro[0] = sm; ro[1] = sq;
            }
        }
    }
}

// sums the split-K slabs in slab order and applies the epilogue (four columns per thread)
__global__ __launch_bounds__(256) void x3p_reduce_kernel(const IefGemmX3pParams p) {
    const int N4 = p.N >> 2;
    const long long total = (long long)p.M * N4, slab = (long long)p.M * p.N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / N4), n = (int)(i - (long long)m * N4) * 4;
        const float* w = p.ws + (long long)m * p.N + n;
        f32x4 a = *(const f32x4*)w;
        int s = 1;
        for (; s + 3 < p.splits; s += 4) {           // four slabs' loads in flight, added in slab order
            f32x4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *(const f32x4*)(w + (s + u) * slab);
#pragma unroll
            for (int u = 0; u < 4; ++u) a += t[u];
        }
        for (; s < p.splits; ++s) a += *(const f32x4*)(w + s * slab);
        xp_store(p, a, m, n);
    }
}

// fp32 -> planes (activation scale `scale`); C % 4 == 0
__global__ __launch_bounds__(256) void x3_split_act_kernel(const float* __restrict__ x, half_t* __restrict__ planes, long long plane,
                                                           long long rows, int C4, int ldx, int ldp, float scale) {
    const long long total = rows * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / C4;
        const int c = (int)(i - r * C4) * 4;
        half4 h, l;
        split4(*(const f32x4*)(x + r * ldx + c), scale, h, l);
        half_t* o = planes + r * ldp + c;
        *(half4*)o = h;
        *(half4*)(o + plane) = l;
    }
}
extern "C" int ief_x3_split_act(const float* x, ief_half* planes, long long plane, long long rows, int C, int ldx, int ldp, float scale,
                                void* stream) {
    if (!x || !planes) return IEF_EINVAL;
    if (rows <= 0 || C <= 0 || (C & 3) || (ldx & 3) || (ldp & 3) || (plane & 3) || !(scale > 0.f)) return IEF_ESHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)planes & 7)) return IEF_EALIGN;
    long long grid = (rows * (C / 4) + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(x3_split_act_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, (half_t*)planes, plane, rows,
                       C / 4, ldx, ldp, scale);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

extern "C" int ief_gemm_x3p_tile_bm(int tile) {
    switch (tile) {
        case 1: case 2: case 3: case 7: return 128;
        case 4: case 8: case 11: case 12: return 256;
        case 5: return 64;
        case 6: return 128;
        default: return 0;
    }
}
extern "C" int ief_gemm_x3p_tile_wn(int tile) { return tile == 6 ? 64 : tile == 8 ? 160 : 80; }
extern "C" int ief_gemm_x3p_tile_bn(int tile) {
    switch (tile) {
        case 1: case 2: case 4: case 5: case 7: return 160;
        case 3: case 11: case 12: return 80;
        case 6: return 64;
        case 8: return 320;
        default: return 0;
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NS, int NL>
static int launch_x3p(const IefGemmX3pParams& p, hipStream_t st) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int splits = p.splits > 1 ? p.splits : 1;
    constexpr int NT = 64 * (WAVES_M * WAVES_N + NL);
    if (p.conv) hipLaunchKernelGGL((igemm_x3p_kernel<BM, BN, WAVES_M, WAVES_N, NS, NL, true>), dim3(tiles, splits), dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((igemm_x3p_kernel<BM, BN, WAVES_M, WAVES_N, NS, NL, false>), dim3(tiles, splits), dim3(NT), 0, st, p);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

int ief_conv_halo_x3p_dispatch(const IefGemmX3pParams& p, hipStream_t st);      // conv_halo_x3p.hip

extern "C" int ief_gemm_x3p(const IefGemmX3pParams* pp, void* stream) {
    if (!pp || !pp->A || !pp->W || !pp->zeros || (!pp->Out && !pp->OutP)) return IEF_EINVAL;
    IefGemmX3pParams p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K % XP_BK) || (p.N & 3)) return IEF_ESHAPE;
    if (p.rowvec && p.rows_per_batch <= 0) return IEF_ESHAPE;
    if (p.splits > 1 && (!p.ws || p.splits > 64 || p.geglu || p.rstat_out || p.cstat_out)) return IEF_EINVAL;
    if (!(p.inv_scale > 0.f)) return IEF_EINVAL;
    if ((p.ldw & 7) || (p.planeW & 7) || ((uintptr_t)p.W & 15)) return IEF_EALIGN;
    if (p.Out && ((p.ldo & 3) || ((uintptr_t)p.Out & 15))) return IEF_EALIGN;
    if (p.OutP && ((p.ldp & 3) || (p.planeO & 3) || ((uintptr_t)p.OutP & 7))) return IEF_EALIGN;
    if ((p.bias && ((uintptr_t)p.bias & 15)) || (p.rowvec && ((uintptr_t)p.rowvec & 15)) ||
        (p.residual && (((uintptr_t)p.residual & 15) || (p.ldr & 3)))) return IEF_EALIGN;
    if (p.geglu && ((p.N & 15) || p.residual || p.rowvec)) return IEF_EINVAL;
    if (p.rstat_out && (p.splits > 1 || p.geglu || p.tile == 11 || p.tile == 12 || (p.N % ief_gemm_x3p_tile_wn(p.tile)))) return IEF_EINVAL;
    if (p.rstat_in && (!p.colsum || p.rstat_slots <= 0 || p.rstat_cnt <= 0 || p.rstat_slots * p.rstat_cnt != p.K || p.splits > 1 ||
                       p.conv || !(p.ln_eps > 0.f) || ((uintptr_t)p.colsum & 15))) return IEF_EINVAL;
    if (p.conv) {
        if (p.C1 <= 0 || p.C2 < 0 || (p.C1 % XP_BK) || (p.C2 % XP_BK) || (p.CE1 % XP_BK) || (p.CE2 % XP_BK)) return IEF_ESHAPE;
        if ((p.C2 > 0 && !p.A2) || (p.CE1 > 0 && !p.E1) || (p.CE2 > 0 && !p.E2) || (p.CE2 > 0 && p.CE1 == 0)) return IEF_EINVAL;
        if (p.stride != 1 && p.stride != 2) return IEF_ESHAPE;
        if (p.ups != 0 && p.ups != 1) return IEF_ESHAPE;
        if (p.ups && ((p.H | p.Wd) & 1)) return IEF_ESHAPE;
        if ((p.CE1 + p.CE2) > 0 && (p.stride != 1 || p.ups)) return IEF_ESHAPE;
        if (p.K != 9 * (p.C1 + p.C2) + p.CE1 + p.CE2 || p.batch_images <= 0 || p.M != p.batch_images * p.Ho * p.Wo) return IEF_ESHAPE;
        const long long in_pix = (long long)p.batch_images * (p.H >> p.ups) * (p.Wd >> p.ups);
        const int cmax = p.C1 > p.C2 ? p.C1 : p.C2, emax = p.CE1 > p.CE2 ? p.CE1 : p.CE2;
        if (in_pix * cmax * 2 >= (1ll << 32) || (long long)p.M * emax * 2 >= (1ll << 32)) return IEF_ESHAPE;   // 32-bit pixel offsets
        if (((uintptr_t)p.A & 15) || (p.planeA & 7) || (p.A2 && (((uintptr_t)p.A2 & 15) || (p.planeA2 & 7))) ||
            (p.E1 && (((uintptr_t)p.E1 & 15) || (p.planeE1 & 7))) || (p.E2 && (((uintptr_t)p.E2 & 15) || (p.planeE2 & 7)))) return IEF_EALIGN;
    } else {
        if ((p.lda & 7) || (p.planeA & 7) || ((uintptr_t)p.A & 15)) return IEF_EALIGN;
    }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (p.tile) {
        case 2: rc = launch_x3p<128, 160, 4, 2, 4, 4>(p, st); break;
        case 3: rc = launch_x3p<128, 80, 4, 1, 3, 0>(p, st); break;
        case 4: rc = launch_x3p<256, 160, 4, 2, 3, 0>(p, st); break;
        case 5: rc = launch_x3p<64, 160, 2, 2, 4, 0>(p, st); break;
        case 6: rc = launch_x3p<128, 64, 4, 1, 3, 0>(p, st); break;
        case 7: rc = launch_x3p<128, 160, 2, 2, 2, 0>(p, st); break;       // 4 waves (64 x 80 each), 72 KiB of LDS: two workgroups per CU
        case 8: rc = launch_x3p<256, 320, 4, 2, 2, 0>(p, st); break;       // 8 waves of 64 x 160: the fewest staged bytes per FLOP (wide N only)
        case 11: case 12: rc = ief_conv_halo_x3p_dispatch(p, st); break;
        case 1: default: rc = launch_x3p<128, 160, 4, 2, 4, 0>(p, st); break;
    }
    if (rc) return rc;
    if (p.splits > 1) {
        const long long total = (long long)p.M * (p.N / 4);
        int grid = (int)((total + 255) / 256);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(x3p_reduce_kernel, dim3(grid), dim3(256), 0, st, p);
        IEF_LAUNCH_CHECK();
    }
    return IEF_OK;
}
