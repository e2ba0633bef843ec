// MFMA GEMM and NHWC implicit-GEMM 3x3 convolution for gfx950 (fp16 in, fp32 accumulate).
//
//   Out[m][n] = epilogue( sum_k A(m,k) * W[n][k] )
//
// DENSE : A is [M][K] row-major (linear / 1x1-conv over tokens-major activations).
// CONV  : A(m,k) is gathered on the fly from an NHWC activation: m = (b, oy, ox),
//         k = (ky, kx, c); zero padding 1; stride 1 or 2; optional nearest-2x upsample of the
//         input fused into the gather; optional channel concat of two sources (the UNet's
//         skip connections) fused as two K ranges; optional extra K range = fused 1x1 convolution
//         over further sources (ResnetBlock2D.conv_shortcut).  Nothing is materialised (no im2col).
//         W is [Cout][3][3][C1+C2](+extra) so both operands are K-major, what MFMA fragments want.
//
// One kernel template, a family of tiles <BM, BN, WAVES_M, WAVES_N> x BK = 64:
//   - operand tiles are staged by LDS-DMA (global_load_lds, 16 B per lane, asynchronous) into an
//     NS-deep ring (2..4 K tiles, counted vmcnt + raw s_barrier so loads stay in flight across barriers); LDS rows are 128 B with the 16-B chunk index XOR-ed by (row>>1)&7 — applied to
//     the per-lane SOURCE address, the LDS image of a wave-instruction being lane-linear — which
//     makes the ds_read_b128 fragment reads conflict-free;
//   - conv addressing is incremental and 32-bit: per-row pixel offsets / padding masks are recomputed
//     only when the 3x3 tap changes (every Ctot/64 K tiles), a K tile adds one scalar; padded taps,
//     M / N / K tails read a zero page;
//   - v_mfma_f32_16x16x32_f16, (BM/WAVES_M/16) x (BN/WAVES_N/16) accumulator tiles per wave;
//   - epilogue through LDS in 64-row passes: + bias[n] + rowvec[batch(m)][n] (time embedding)
//     + residual[m][n], * scale, one 16-B fp16 store per 8 outputs;
//   - split-K (grid.y) with fp32 partial slabs for layers whose M x N gives too few tiles for 256 CUs (the 16x16 /
//     8x8 UNet levels), combined INSIDE the launch by the last-arriving workgroup of each tile (agent-scope release /
//     ticket / acquire; IefGemmParams.cnt) or, without counters, by a reducer kernel;
//   - XCD-aware block remap: consecutive logical tiles (same A panel) share an XCD's L2.
// BN = 160 tiles exist because SD's channel widths are multiples of 320: N = 320 is 2 x 160 exactly,
// while 128-wide tiles pad it to 384 and leave 1.5 blocks per CU.
//
// Replaces on the reference path (all executed by diffusers/PyTorch eager there):
//   ResnetBlock2D.conv1/conv2/conv_shortcut + temb add + skip add   /root/reference/pnp/model/register.py:139-175
//   Attention.to_q/to_k/to_v/to_out, proj_in/out, FF linears        /root/reference/p2p/model/register.py:33-54
#include "ief_common.h"
#include "ief_params.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
// one wave-instruction: 64 lanes x 16 B from per-lane global addresses -> 1 KiB of LDS at a wave-uniform base
__device__ __forceinline__ void glds16(const char* g, half_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

// NL > 0: NL extra LOADER waves stage both operands (all LDS-DMA issues, the address bookkeeping and the vmcnt waits are
// theirs); the WAVES_M x WAVES_N compute waves only read fragments and multiply.  An LDS-DMA instruction costs a wave that
// also multiplies ~180 cycles of its in-order stream and a wave that does nothing else ~25-60 (measured on the halo kernel
// below, where the split took 8-12 % off).
template <int BM, int BN, int WAVES_M, int WAVES_N, int NS, int NL, bool CONV>
__global__ __launch_bounds__(64 * (WAVES_M * WAVES_N + NL)) void igemm_f16_kernel(const IefGemmParams p) {
    constexpr int BK = 64;
    constexpr int NWC = WAVES_M * WAVES_N;          // compute waves
    constexpr int NT = 64 * (NWC + NL);             // threads
    constexpr int NST = NL > 0 ? 64 * NL : NT;      // threads that stage
    constexpr int RP = NST / 8;                     // tile rows staged per pass (8 lanes x 16 B = one 128-B row)
    constexpr int NA = (BM + RP - 1) / RP, NB = (BN + RP - 1) / RP;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
    constexpr int ROWS_A = NA * RP, ROWS_B = NB * RP;   // LDS rows incl. staging overshoot
    static_assert(WM % 16 == 0 && WN % 16 == 0 && WM <= 64, "wave tile");
    static_assert(NS >= 2 && NS <= 4, "LDS ring depth");
    constexpr int STAGE_HALFS = NS * (ROWS_A + ROWS_B) * BK;
    constexpr int G = NA + NB;                      // LDS-DMA instructions a wave issues per K tile
    constexpr int EPI_HALFS = (64 * (BN + 4) + 64 * (BN / 8) * 2 + 128 + 4) * 2;   // fp32 [64][BN+4] + row-moment partials [64][BN/8][2] + row mean / rstd [2][64] + the split-K "last arriver" word
    __shared__ __attribute__((aligned(16))) half_t smem[STAGE_HALFS > EPI_HALFS ? STAGE_HALFS : EPI_HALFS];
    half_t* As = smem;
    half_t* Bs = smem + NS * ROWS_A * BK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA bases (m0) stay on the scalar unit
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const bool stages_ = NL == 0 || wave >= NWC;                 // this wave stages operands
    const bool computes = NL == 0 || wave < NWC;                 // this wave multiplies
    const int stid = NL > 0 ? tid - 64 * NWC : tid;              // index among the staging threads (negative: not one)
    const int swave = NL > 0 ? wave - NWC : wave;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    // consecutive logical tiles run on one XCD and share its L2: walk N first (they share the A panel) where the activations are
    // the larger operand, M first (they share the weight tile) where the weights are — the 16x16 / 8x8 levels, whose weights
    // every XCD would otherwise fetch in full
    const int tiles_m = (p.M + BM - 1) / BM;
    const bool m_first = p.N > p.M;
    const int m0 = (m_first ? lid % tiles_m : lid / tiles_n) * BM, n0 = (m_first ? lid / tiles_m : lid % tiles_n) * BN;
    const long long z = blockIdx.z;
    const char* __restrict__ A = (const char*)(p.A + z * p.strideA);
    const char* __restrict__ Wt = (const char*)(p.W + z * p.strideW);
    const char* __restrict__ zp = (const char*)p.zeros;

    const int rbase = (stid >> 3) & (RP - 1);   // row (+RP*i) this thread stages
    // the LDS slot (tid&7) of a lane is fixed by LDS-DMA; the swizzle is applied to the chunk FETCHED
    const unsigned kcb = (unsigned)(((tid & 7) ^ ((rbase >> 1) & 7)) * 16);   // byte offset of that chunk in the K row

    const int nk_all = (p.K + BK - 1) / BK;
    int kt_lo = 0, nk = nk_all;
    if (p.splits > 1) {  // this block's K slice
        const int per = (nk_all + p.splits - 1) / p.splits;
        kt_lo = blockIdx.y * per;
        nk = min(nk_all, kt_lo + per);
    }
    // K % 64 != 0 (never on the UNet's shapes): chunks of the LAST K tile that lie past K read the zero page
    const bool ktail = (p.K & (BK - 1)) != 0;
    const bool tail_zero = (unsigned)(nk_all - 1) * (BK * 2) + kcb >= (unsigned)p.K * 2u;

    // ---- staging state: one 64-bit source pointer per LDS-DMA piece, advanced by one K tile (128 B) per stage.
    // Rows that are out of range (M / N tails, conv padding) point at the zero page and do not advance, so the
    // per-tile address work is one 64-bit add per piece; the pointers are rebuilt only when the conv tap or the
    // concat source changes.
    const char* pa[NA];
    unsigned sa[NA];
    const char* pw[NB];
    unsigned sw[NB];
    bool a_ok[NA];
    int a_y[NA], a_x[NA], a_b[NA];
    unsigned a_m[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int r = rbase + RP * i;
        const int m = m0 + r;
        a_ok[i] = r < BM && m < p.M;
        a_m[i] = (unsigned)m;
        if constexpr (CONV) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_b[i] = b;
            a_y[i] = oy * p.stride - (p.pad_hi_only ? 0 : 1);
            a_x[i] = ox * p.stride - (p.pad_hi_only ? 0 : 1);
            pa[i] = zp; sa[i] = 0;
        } else {
            a_b[i] = a_y[i] = a_x[i] = 0;
            pa[i] = a_ok[i] ? A + ((unsigned long long)(unsigned)m * (unsigned)p.lda * 2ull + kcb + (unsigned long long)kt_lo * (BK * 2)) : zp;
            sa[i] = a_ok[i] ? BK * 2 : 0;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int r = rbase + RP * i;
        const int n = n0 + r;
        const bool ok = r < BN && n < p.N;
        pw[i] = ok ? Wt + ((unsigned long long)(unsigned)n * (unsigned)p.ldw * 2ull + kcb + (unsigned long long)kt_lo * (BK * 2)) : zp;
        sw[i] = ok ? BK * 2 : 0;
    }

    // ---- conv K iterator: (tap, channel offset) advanced by one K tile per stage
    const int Ctot = p.C1 + p.C2;
    const int Hp = p.H >> p.ups, Wp = p.Wd >> p.ups;
    int it_tap = 0, it_c = 0;           // tap 0..8 = 3x3 taps, 9 = fused-1x1 extra range
    bool a_live[NA];
    unsigned r1[NA], r2[NA];            // byte offsets of the row's pixel in source 1 / source 2 (+ chunk)
    auto set_tap = [&](int tap) {
        if (tap < 9) {
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                a_live[i] = a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                const unsigned pix = (unsigned)((a_b[i] * Hp + (iy >> p.ups)) * Wp + (ix >> p.ups));
                r1[i] = pix * (unsigned)p.C1 * 2u + kcb;
                r2[i] = pix * (unsigned)p.C2 * 2u + kcb;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                a_live[i] = a_ok[i];
                r1[i] = a_m[i] * (unsigned)p.CE1 * 2u + kcb;
                r2[i] = a_m[i] * (unsigned)p.CE2 * 2u + kcb;
            }
        }
    };
    // pointers of the current (tap, channel offset): source 1 below c1 channels, source 2 above
    auto set_src = [&]() {
        const char* s1 = it_tap < 9 ? A : (const char*)p.E1;
        const char* s2 = it_tap < 9 ? (const char*)p.A2 : (const char*)p.E2;
        const int c1 = it_tap < 9 ? p.C1 : p.CE1;
        const bool first = it_c < c1;
        const char* src = first ? s1 : s2;
        const unsigned cc = (unsigned)(first ? it_c : it_c - c1) * 2u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            pa[i] = a_live[i] ? src + ((first ? r1[i] : r2[i]) + cc) : zp;
            sa[i] = a_live[i] ? BK * 2 : 0;
        }
    };

    auto stage_tile = [&](int buf, int kt) {
        if (!stages_) return;
        half_t* la = As + buf * ROWS_A * BK + (swave * 8) * BK;
        half_t* lb = Bs + buf * ROWS_B * BK + (swave * 8) * BK;
        if (ktail && kt == nk_all - 1) {   // wave-uniform and never taken on the UNet's shapes (K % 64 == 0): keeps the per-lane
            if (tail_zero) {               // zero-page select out of every other K tile
#pragma unroll
                for (int i = 0; i < NA; ++i) pa[i] = zp;
#pragma unroll
                for (int i = 0; i < NB; ++i) pw[i] = zp;
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            glds16(pa[i], la + RP * i * BK);
            pa[i] += sa[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            glds16(pw[i], lb + RP * i * BK);
            pw[i] += sw[i];
        }
        if constexpr (CONV) {   // advance the iterator by one K tile; rebuild the pointers at tap / source boundaries
            it_c += BK;
            if (it_tap < 9) {
                if (it_c >= Ctot) { it_c = 0; ++it_tap; set_tap(it_tap); set_src(); }
                else if (it_c == p.C1) set_src();
            } else if (it_c == p.CE1 && p.CE2 > 0) {
                set_src();
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The residual rows this thread will add in the epilogue are requested NOW: they do not depend on the product, and a
    // K = C projection is otherwise a chain of memory round trips (first tile, each K tile, then the residual).  Only where it
    // costs few registers (<= 4 chunks of 8 halves per thread).
    constexpr int EP_ROWS = BM < 64 ? BM : 64, EP_CH = BN / 8, EP_NPASS = (BM + 63) / 64;
    constexpr int EP_IPT = (EP_ROWS * EP_CH + NT - 1) / NT;                       // output chunks per thread and pass
    constexpr bool RES_PREFETCH = EP_IPT * EP_NPASS <= 4;
    half8 res_pf[RES_PREFETCH ? EP_IPT * EP_NPASS : 1];
    const bool res_pf_on = RES_PREFETCH && p.residual && p.splits <= 1 && !(p.flags & 2);
    if (res_pf_on) {
#pragma unroll
        for (int ps = 0; ps < EP_NPASS; ++ps)
#pragma unroll
            for (int k = 0; k < EP_IPT; ++k) {
                const int c = tid + k * NT, row = c / EP_CH, nc = c - row * EP_CH;
                const int m = m0 + ps * 64 + row, n = n0 + nc * 8;
                res_pf[RES_PREFETCH ? ps * EP_IPT + k : 0] = (c < EP_ROWS * EP_CH && m < p.M && n < p.N)
                    ? *(const half8*)(p.residual + z * p.strideR + (long long)m * p.ldr + n) : (half8){0, 0, 0, 0, 0, 0, 0, 0};
            }
    }

    if constexpr (CONV) {
        const int k0 = kt_lo * BK;
        if (k0 < 9 * Ctot) { it_tap = k0 / Ctot; it_c = k0 - it_tap * Ctot; }
        else { it_tap = 9; it_c = k0 - 9 * Ctot; }
        set_tap(it_tap);
        set_src();
    }
    const int fr = lane & 15, fq = lane >> 4;
    // fragment offsets (halves) inside one ring slot: row * 64 + swizzled chunk; k-step 1 flips chunk bit 2
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wr * WM + i * 16 + fr;
        aoff[i] = row * BK + ((fq ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wc * WN + j * 16 + fr;
        boff[j] = row * BK + ((fq ^ ((row >> 1) & 7)) << 3);
    }
    // One K tile: both k-steps' fragments are requested up front, so the second set's LDS latency hides behind the
    // first set's MFMAs (the compiler turns the two uses into counted lgkmcnt waits).
    half8 af0[TM], bf0[TN], af1[TM], bf1[TN];
    auto read_frags = [&](const half_t* Ac, const half_t* Bc) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af0[i] = *(const half8*)(Ac + aoff[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf0[j] = *(const half8*)(Bc + boff[j]);
#pragma unroll
        for (int i = 0; i < TM; ++i) af1[i] = *(const half8*)(Ac + (aoff[i] ^ 32));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf1[j] = *(const half8*)(Bc + (boff[j] ^ 32));
    };
    auto mfma_frags = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af0[i], bf0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af1[i], bf1[j], acc[i][j], 0, 0, 0);
    };
    // NS-deep LDS ring: K tiles kt .. kt+NS-2 are in flight while tile kt is multiplied.  Order per tile (interval
    // between two barriers):
    //   counted vmcnt (this wave's share of tile kt has landed)  ->  s_barrier (everybody's has; everybody is done
    //   reading the slot tile kt-1 used)  ->  issue tile kt+NS-1 into that slot  ->  fragments + MFMAs of tile kt.
    // Raw s_barrier + counted waits: __syncthreads() would drain every LDS-DMA in flight.  The loop is unrolled NS
    // times so every slot index is a compile-time constant.
    // 8-wave tiles, NS >= 3: the two waves of a SIMD would run that sequence in lockstep (both issuing LDS-DMA, then
    // both multiplying).  Waves 4..7 therefore run HALF A TILE LATE: in interval kt they first multiply tile kt-1 from
    // fragments they read (and waited for) before the barrier, then stage and read tile kt — so each SIMD always has one
    // wave on the matrix pipe while its partner issues DMA / LDS reads.  Same barriers, same slots, same sums.
    constexpr bool PINGPONG = (WAVES_M * WAVES_N == 8) && NS >= 3;
    const bool late = PINGPONG && wave >= 4 && wave < NWC;
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (kt_lo + s < nk) stage_tile(s, kt_lo + s);
    if (NL > 0 && !computes) {
        // loader waves: their share of tile k has landed -> barrier -> refill the slot tile k-1 used
        for (int kt = kt_lo; kt < nk; kt += NS) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int k = kt + s;
                if (k < nk) {
                    if (nk - 1 - k >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (NS - 2)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_barrier" ::: "memory");
                    if (k + NS - 1 < nk) stage_tile((s + NS - 1) % NS, k + NS - 1);
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
                    asm volatile("s_barrier" ::: "memory");
                    if constexpr (NL == 0) { if (k + NS - 1 < nk) stage_tile((s + NS - 1) % NS, k + NS - 1); }
                    read_frags(As + s * ROWS_A * BK, Bs + s * ROWS_B * BK);
                    mfma_frags();
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
                    asm volatile("s_barrier" ::: "memory");
                    if (k > kt_lo) mfma_frags();                                   // tile k-1, fragments already in registers
                    if constexpr (NL == 0) { if (k + NS - 1 < nk) stage_tile((s + NS - 1) % NS, k + NS - 1); }
                    read_frags(As + s * ROWS_A * BK, Bs + s * ROWS_B * BK);         // tile k: landed (barrier above)
                }
            }
        }
        if (nk > kt_lo) mfma_frags();
    }
    __syncthreads();   // all fragment reads done before the staging buffers become the epilogue's scratch

    // ---------------- epilogue through LDS: 64 output rows per pass
    constexpr int LDS_N = BN + 4;
    constexpr int NPASS = (BM + 63) / 64;
    constexpr int CH = BN / 8;
    float* stage = (float*)smem;
    float* rpart = stage + 64 * LDS_N;               // [64][CH][2] partial (sum, sumsq) per output chunk
    float* row_mu = rpart + 64 * CH * 2;             // folded LayerNorm: mean and rstd of the 64 rows of a pass
    float* row_rs = row_mu + 64;
    half_t* __restrict__ Out = p.Out + z * p.strideO;
    float ccs = 0.f, ccq = 0.f;                      // cstat_out: thread tid < BN owns output column n0 + tid
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (p.rstat_in && tid < 64) {
            const int m = m0 + pass * 64 + tid;
            float s1 = 0.f, s2 = 0.f;
            if (m < p.M) {
                const float* rs = p.rstat_in + (long long)m * p.rstat_slots * 2;
                for (int t = 0; t < p.rstat_slots; ++t) { s1 += rs[2 * t]; s2 += rs[2 * t + 1]; }
            }
            const float mu = s1 / (float)p.K;
            row_mu[tid] = mu;
            row_rs[tid] = rsqrtf(fmaxf(s2 / (float)p.K - mu * mu, 0.f) + p.ln_eps);
        }
        if ((wr * WM) / 64 == pass) {
            const int rb0 = wr * WM - pass * 64;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stage[(rb0 + i * 16 + fq * 4 + r) * LDS_N + wc * WN + j * 16 + fr] = acc[i][j][r];
        }
        __syncthreads();
        constexpr int PROWS = BM < 64 ? BM : 64;
        if (p.flags & 2) {
            // GEGLU epilogue (FeedForward.net[0]): the weight rows were interleaved in groups of 8 at pack time, so
            // column chunks alternate [8 hidden | 8 gate]; out[m][j] = hidden_j * gelu(gate_j), Out has N/2 columns.
            for (int c = tid; c < PROWS * (CH / 2); c += NT) {
                const int row = c / (CH / 2), pc = c - row * (CH / 2);
                const int m = m0 + pass * 64 + row, n = n0 + pc * 16;
                if (m < p.M && n < p.N) {
                    const float* sp = stage + row * LDS_N + pc * 16;
                    float hv[8], gv[8];
                    const float rs_ = p.rstat_in ? row_rs[row] : 1.f, mu_ = p.rstat_in ? row_mu[row] : 0.f;
#pragma unroll
                    for (int q4 = 0; q4 < 2; ++q4) {
                        f32x4 a = *(const f32x4*)(sp + 4 * q4), g = *(const f32x4*)(sp + 8 + 4 * q4);
                        const f32x4 ba = *(const f32x4*)(p.bias + n + 4 * q4), bg = *(const f32x4*)(p.bias + n + 8 + 4 * q4);
                        if (p.rstat_in) {
                            const f32x4 ca = *(const f32x4*)(p.colsum + n + 4 * q4), cg = *(const f32x4*)(p.colsum + n + 8 + 4 * q4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) { a[e] = rs_ * (a[e] - mu_ * ca[e]); g[e] = rs_ * (g[e] - mu_ * cg[e]); }
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { hv[4 * q4 + e] = a[e] + ba[e]; gv[4 * q4 + e] = g[e] + bg[e]; }
                    }
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)(hv[e] * gelu_f(gv[e]));
                    *(half8*)(Out + (long long)m * p.ldo + (n >> 1)) = o;
                }
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int kq = 0; kq < EP_IPT; ++kq) {
            const int c = tid + kq * NT;
            if (c >= PROWS * CH) break;
            const int row = c / CH, nc = c - row * CH;
            const int m = m0 + pass * 64 + row, n = n0 + nc * 8;
            if (m < p.M && n < p.N) {
                const f32x4 s0 = *(const f32x4*)(stage + row * LDS_N + nc * 8);
                const f32x4 s1 = *(const f32x4*)(stage + row * LDS_N + nc * 8 + 4);
                if (p.splits > 1) {  // raw partial sums; the reducer applies the epilogue
                    float* w = p.ws + ((long long)blockIdx.y * p.M + m) * p.N + n;
                    *(f32x4*)w = s0;
                    *(f32x4*)(w + 4) = s1;
                    continue;
                }
                float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
                if (p.rstat_in) {
                    const float rs_ = row_rs[row], mu_ = row_mu[row];
                    const f32x4 c0 = *(const f32x4*)(p.colsum + n), c1 = *(const f32x4*)(p.colsum + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = rs_ * (v[e] - mu_ * c0[e]); v[4 + e] = rs_ * (v[4 + e] - mu_ * c1[e]); }
                }
                if (p.bias) {
                    const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                }
                if (p.rowvec) {
                    const float* rv = p.rowvec + (long long)(m / p.rows_per_batch) * p.N + n;
                    const f32x4 b0 = *(const f32x4*)rv, b1 = *(const f32x4*)(rv + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                }
                if (p.residual) {
                    half8 rs;
                    if constexpr (RES_PREFETCH) {
                        rs = res_pf_on ? res_pf[pass * EP_IPT + kq] : *(const half8*)(p.residual + z * p.strideR + (long long)m * p.ldr + n);
                    } else {
                        rs = *(const half8*)(p.residual + z * p.strideR + (long long)m * p.ldr + n);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rs[e];
                }
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)(v[e] * p.out_scale);
                *(half8*)(Out + (long long)m * p.ldo + n) = o;
                if (p.cstat_out) {   // what the consumer will read goes back into the stage for the column sums below
                    float* sp = stage + row * LDS_N + nc * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) sp[e] = (float)o[e];
                }
                if (p.rstat_out) {   // moments of what the consumer will read (the fp16-rounded values)
                    float a1 = 0.f, a2 = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float f = (float)o[e]; a1 += f; a2 += f * f; }
                    rpart[(row * CH + nc) * 2] = a1;
                    rpart[(row * CH + nc) * 2 + 1] = a2;
                }
            } else {
                if (p.rstat_out) {
                    rpart[(row * CH + nc) * 2] = 0.f;
                    rpart[(row * CH + nc) * 2 + 1] = 0.f;
                }
                if (p.cstat_out) {
                    float* sp = stage + row * LDS_N + nc * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) sp[e] = 0.f;
                }
            }
        }
        __syncthreads();
        if (p.cstat_out && p.splits <= 1) {   // column sums over this pass's rows, fixed order: deterministic
            if (tid < BN) {
                for (int r = 0; r < PROWS; ++r) {
                    const float v = stage[r * LDS_N + tid];
                    ccs += v;
                    ccq += v * v;
                }
            }
            __syncthreads();         // the next pass overwrites the stage
        }
        if (p.rstat_out && tid < PROWS) {   // fixed summation order: deterministic
            const int m = m0 + pass * 64 + tid;
            if (m < p.M) {
                float a1 = 0.f, a2 = 0.f;
                for (int c = 0; c < CH; ++c) { a1 += rpart[(tid * CH + c) * 2]; a2 += rpart[(tid * CH + c) * 2 + 1]; }
                float* ro = p.rstat_out + ((long long)m * tiles_n + (n0 / BN)) * 2;
                ro[0] = a1; ro[1] = a2;
            }
        }
    }
    if (p.splits > 1 && p.cnt) {
        // ---- split-K combined inside the launch (guide: "In-launch split-K reduction"): this workgroup's slab tile was
        // written with plain 16-B stores above.  Every wave drains its stores, the workgroup meets, ONE lane releases at
        // agent scope (write-back of this XCD's L2) and draws a ticket; whoever draws the last one acquires (drops this
        // CU's stale lines) and combines.  Correct for any placement of a tile's slices over XCDs / CUs.
        int* last_word = (int*)(row_rs + 64);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the fence's own wait can be dropped by the compiler
            const int ticket = __hip_atomic_fetch_add(p.cnt + lid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == p.splits - 1;
            if (last) {
                __hip_atomic_store(p.cnt + lid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *last_word = last;
        }
        __syncthreads();
        if (!*last_word) return;
        const long long slab = (long long)p.M * p.N;
        constexpr int PROWS = BM < 64 ? BM : 64;
        for (int pass = 0; pass < NPASS; ++pass) {
            for (int c = tid; c < PROWS * CH; c += NT) {
                const int row = c / CH, nc = c - row * CH;
                const int m = m0 + pass * 64 + row, n = n0 + nc * 8;
                float* sp = stage + row * LDS_N + nc * 8;
                if (m < p.M && n < p.N) {
                    const float* w = p.ws + (long long)m * p.N + n;
                    f32x4 a0 = *(const f32x4*)w, a1 = *(const f32x4*)(w + 4);
                    int sl = 1;
                    for (; sl + 3 < p.splits; sl += 4) {          // four slabs' loads in flight, added in slab order
                        f32x4 t0[4], t1[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { t0[u] = *(const f32x4*)(w + (sl + u) * slab); t1[u] = *(const f32x4*)(w + (sl + u) * slab + 4); }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { a0 += t0[u]; a1 += t1[u]; }
                    }
                    for (; sl < p.splits; ++sl) { a0 += *(const f32x4*)(w + sl * slab); a1 += *(const f32x4*)(w + sl * slab + 4); }
                    float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    if (p.bias) {
                        const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                    }
                    if (p.rowvec) {
                        const float* rv = p.rowvec + (long long)(m / p.rows_per_batch) * p.N + n;
                        const f32x4 b0 = *(const f32x4*)rv, b1 = *(const f32x4*)(rv + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                    }
                    if (p.residual) {
                        const half8 rs = *(const half8*)(p.residual + (long long)m * p.ldr + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)rs[e];
                    }
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)(v[e] * p.out_scale);
                    *(half8*)(Out + (long long)m * p.ldo + n) = o;
                    if (p.cstat_out) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) sp[e] = (float)o[e];
                    }
                } else if (p.cstat_out) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sp[e] = 0.f;
                }
            }
            if (p.cstat_out) {
                __syncthreads();
                if (tid < BN) {
                    for (int r = 0; r < PROWS; ++r) {
                        const float v = stage[r * LDS_N + tid];
                        ccs += v;
                        ccq += v * v;
                    }
                }
                __syncthreads();
            }
        }
    }
    if (p.cstat_out && tid < BN && n0 + tid < p.N) {
        float* co = p.cstat_out + ((long long)(m0 / BM) * p.N + n0 + tid) * 2;
        co[0] = ccs; co[1] = ccq;
    }
}

// sums the split-K slabs and applies the same epilogue as the single-pass kernel
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const IefGemmParams p) {
    const int N8 = p.N >> 3;
    const long long total = (long long)p.M * N8;
    const long long slab = (long long)p.M * p.N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / N8), n = (int)(i - (long long)m * N8) * 8;
        const float* w = p.ws + (long long)m * p.N + n;
        f32x4 a0 = *(const f32x4*)w, a1 = *(const f32x4*)(w + 4);
        int s = 1;
        for (; s + 3 < p.splits; s += 4) {      // four slabs' loads in flight, added in slab order (bit-identical sums)
            f32x4 t0[4], t1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { t0[u] = *(const f32x4*)(w + (s + u) * slab); t1[u] = *(const f32x4*)(w + (s + u) * slab + 4); }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a0 += t0[u]; a1 += t1[u]; }
        }
        for (; s < p.splits; ++s) {
            a0 += *(const f32x4*)(w + s * slab);
            a1 += *(const f32x4*)(w + s * slab + 4);
        }
        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        if (p.bias) {
            const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
        }
        if (p.rowvec) {
            const float* rv = p.rowvec + (long long)(m / p.rows_per_batch) * p.N + n;
            const f32x4 b0 = *(const f32x4*)rv, b1 = *(const f32x4*)(rv + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
        }
        if (p.residual) {
            const half8 rs = *(const half8*)(p.residual + (long long)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rs[e];
        }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)(v[e] * p.out_scale);
        *(half8*)(p.Out + (long long)m * p.ldo + n) = o;
    }
}


// ------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with the input tile RESIDENT in LDS across the nine taps
// ------------------------------------------------------------------------------------------
// The implicit GEMM above re-stages the A operand once per tap: nine LDS-DMA passes over (nearly) the same pixels.
// Measured on it: the K loop runs at the rate its LDS-DMA pieces can be issued and landed (36 one-KiB pieces per
// 128x160x64 K tile; pointing them at one L1-hot page changes little, removing them takes a third off), so the
// lever is FEWER PIECES PER FLOP, not a smarter pipeline.  Here a workgroup owns BM consecutive output pixels
// m0 .. m0+BM-1 of the flattened [B*H*W] pixel axis and, per 64-channel block, stages the pixel range
// m0-(W+1) .. m0+BM+W ("super-tile", BM + 2W + 2 rows of 128 B) ONCE; tap (ky, kx) of output row r is then super-tile
// row r + ky*W + kx, and lanes whose tap falls outside the image (left / right / top / bottom edge) read a zero row
// instead.  The loop nest is channel block (outer) x tap (inner); only the weight tile changes per tap.  With
// BM = 256, BN = 80 a K tile costs 10 weight pieces + 5.4 input pieces instead of 20 + 16 for the same 2.6 MFLOP.
// The M tile stays a contiguous range of pixels, so the epilogue, split-K slabs and GroupNorm column statistics
// are exactly those of igemm_f16_kernel.
//
// Pipeline: weight tiles in a 4-slot ring, three steps ahead; the next channel block's super-tile (double buffered)
// is fetched one piece per wave per step during taps 0..6 of the current block.  Each wave counts the LDS-DMA
// instructions it issued in the last two steps and waits with vmcnt(that count): everything older — this step's
// weight tile and, at tap 0, the whole super-tile — has then landed; the s_barrier publishes it to the other waves.
template <int N>
struct IntTag { static constexpr int value = N; };

template <int N>
__device__ __forceinline__ void wait_vmcnt_le(int n) {   // n wave-uniform; waits until at most min(n, N) LDS-DMA are in flight
    if constexpr (N == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (n >= N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
        else wait_vmcnt_le<N - 1>(n);
    }
}

// VAR (schedule switches kept for A/B; measured on the SD1.5 batch-4 shapes, same box, interleaved rounds — 64x64x320 conv in us):
//   bit 0: waves NW/2.. issue their LDS-DMA group at the END of the step            off 33.8  on 33.2
//   bit 1: one lgkmcnt(0) BEFORE each half's reads instead of after its MFMAs        off 33.8  on 32.7   (both: 33.1, best on the concat shapes)
//   bit 2: sched_group_barrier pinning "all reads, then all MFMAs" inside a half     on 34.2-34.9: worse, the compiler's own interleave stays
//   NL: loader waves.  NL = 0: every wave issues its share of the LDS-DMA (the schedule below).  NL = 4: waves NW .. NW+3 do
//   nothing but issue LDS-DMA and count it (they take the vmcnt waits), the NW compute waves only read fragments and
//   multiply.  Measured on the 64x64x320 convolution (ablation builds, us): whole kernel 34.4, without the LDS-DMA
//   instructions 26.6, without the fragment reads 29.2, without either 23.1 — an LDS-DMA instruction costs a COMPUTE wave
//   ~180 cycles of its in-order stream (address VALU queued behind MFMAs, M0 write, issue), a wave that does nothing else
//   ~25 (guide: ldsdma-fill).
//   UPS: the nearest-2x upsample of Upsample2D fused in (the source has H/2 x W/2 pixels): tap (ky, kx) of output pixel (y, x) is
//   source pixel ((y + ky - 1) >> 1, (x + kx - 1) >> 1); the super-tile is then the SOURCE pixel range the tile's rows touch
//   ((rows / 2 + 2) source rows + 2 pixels: smaller than the plain form's) and the fragment row is computed per lane and tap
//   instead of being a constant shift.  Loader-wave form only; the tile must lie inside one image (H W a multiple of BM).
template <int BM, int BN, int WAVES_M, int WAVES_N, int NL = 0, int VAR = 3, bool UPS = false>
__global__ __launch_bounds__(64 * (WAVES_M * WAVES_N + NL)) void conv3x3_halo_kernel(const IefGemmParams p) {
    static_assert(!UPS || NL > 0, "the upsample form exists with loader waves only");
    constexpr int BK = 64;
    constexpr int NW = WAVES_M * WAVES_N, NT = 64 * (NW + NL);
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
    constexpr int WMAX = 64;                                    // widest image row the super-tile is sized for
    constexpr int NPA = (BM + 2 * WMAX + 2 + 7) / 8;            // super-tile pieces (8 rows x 128 B each) at that width
    constexpr int ABUF = ((NPA + 1) / 2) * 2 * 1024;            // bytes per super-tile buffer
    constexpr int NPB = BN / 8, BPW = (NPB + NW - 1) / NW;      // weight-tile pieces, per wave
    constexpr int NSB = 5, BBUF = NPB * 1024;                   // weight ring
    constexpr int BOFF = 2 * ABUF, ZOFF = BOFF + NSB * BBUF;    // ZOFF: 128 zero bytes, what an out-of-image tap reads
    constexpr int DUMP = ZOFF + 128;                            // NL > 0: 1 KiB where the loaders' filler pieces land
    constexpr int LDS_BYTES = ZOFF + 128 + (NL > 0 ? 1024 : 0);
    constexpr int LDS_N = BN + 4;
    static_assert(WM % 16 == 0 && WN % 16 == 0 && BM % 64 == 0 && BN % 8 == 0, "tile");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(128))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WAVES_N, wc = wave % WAVES_N;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    // consecutive logical tiles run on one XCD and share its L2: walk N first (they share the A panel) where the activations are
    // the larger operand, M first (they share the weight tile) where the weights are — the 16x16 / 8x8 levels, whose weights
    // every XCD would otherwise fetch in full
    const int tiles_m = (p.M + BM - 1) / BM;
    const bool m_first = p.N > p.M;
    const int m0 = (m_first ? lid % tiles_m : lid / tiles_n) * BM, n0 = (m_first ? lid / tiles_m : lid % tiles_n) * BN;
    const int W = p.Wd, H = p.H;
    const int Ctot = p.C1 + p.C2;
    const char* __restrict__ zp = (const char*)p.zeros;
    // UPS: source geometry and the source pixel (flattened over the batch) that super-tile row 0 holds
    const int Wi = W >> 1, Hi = H >> 1;
    const int ups_img = m0 / (H * W), ups_y0 = (m0 - ups_img * H * W) / W;
    const int ups_rel = ((ups_y0 - 1) >> 1) * Wi - 1;                  // relative to the image's first source pixel
    const int src_pixels = UPS ? p.batch_images * Hi * Wi : p.M;       // pixels of a source tensor
    const int st_row0 = UPS ? ups_img * Hi * Wi + ups_rel : m0 - (W + 1);

    // channel blocks of this K slice
    const int ncb = Ctot / BK;
    int cb_lo = 0, cb_hi = ncb;
    if (p.splits > 1) {
        const int per = (ncb + p.splits - 1) / p.splits;
        cb_lo = min(ncb, (int)blockIdx.y * per);
        cb_hi = min(ncb, cb_lo + per);
    }
    const int nsteps = (cb_hi - cb_lo) * 9;

    if (tid < 32) ((float*)(smem + ZOFF))[tid] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the first raw s_barrier publishes it

    // ---- LDS images: rows of 128 B (64 channels); row r keeps 16-B chunk c in slot c ^ (r & 6).  That swizzle is free of
    // bank conflicts for ds_read_b128 fragment reads of 16 CONSECUTIVE rows starting at ANY row (the tap shift moves the
    // start), which (r >> 1) & 7 of igemm_f16_kernel is only for starts that are multiples of 4.  It is applied to the
    // SOURCE address of the LDS-DMA (the LDS image of a wave-instruction is lane-linear): lane -> (row 8q + lane/8, slot
    // lane%8) of piece q fetches chunk slot ^ (row & 6) = (lane & 7) ^ ((lane >> 3) & 6), the same for every piece.
    const unsigned st_chunk = (unsigned)(((lane & 7) ^ ((lane >> 3) & 6)) * 16);
    // LDS-DMA schedule, the same in every step so that the vmcnt counts are compile-time constants:
    //   every wave      : weight piece `wave` of tile t+4; during taps 0..5 super-tile piece wave + NW*tap of the next block
    //   waves < XB only : weight piece NW + wave; at tap 0 also super-tile piece wave + 6 NW
    // (XB = NPB - NW = 2 waves; 6 NW + XB = 50 pieces >= the 49 a 64-pixel row needs).  What has no real source — a piece past
    // the super-tile of a narrower image, a tile past the end of the K slice, the block after the last — is fetched from the
    // zero page all the same: an L1-hot kilobyte costs less than a per-wave count.
    constexpr int XB = NPB - NW;
    static_assert(NL > 0 || (XB >= 0 && XB <= NW && BPW <= 2 && 6 * NW + XB >= NPA && (6 * NW + XB) * 1024 <= ABUF), "LDS-DMA schedule");
    const bool xw = wave < XB;
    const bool late = (VAR & 1) && wave >= NW / 2;
    const int a_ms0 = m0 - (W + 1) + 8 * wave + (lane >> 3);          // pixel fetched for super-tile piece wave + NW*j: + 8 NW j
    // source of the NEXT channel block's super-tile (set per block)
    const char* an_src = zp; unsigned an_cs = 0, an_c0 = 0; bool an_on = false;
    auto set_next_block = [&](int cb, bool on) {
        const bool first = cb * BK < p.C1;
        an_src = (const char*)(first ? p.A : p.A2);
        an_cs = (unsigned)(first ? p.C1 : p.C2);
        an_c0 = (unsigned)(first ? cb * BK : cb * BK - p.C1);
        an_on = on;
    };
    auto issue_a = [&](int j, int buf_off) {              // super-tile piece wave + NW j of the next block
        const int ms = a_ms0 + 8 * NW * j;
        const bool ok = an_on && (unsigned)ms < (unsigned)p.M;
        const char* g = ok ? an_src + ((unsigned long long)((unsigned)ms * an_cs + an_c0) * 2ull + st_chunk) : zp;
        glds16(g, (half_t*)(smem + buf_off + (wave + NW * j) * 1024));
    };
    // weight pieces: rows 8 wave + lane/8 (and 8 (NW + wave) + lane/8 for waves < XB) of the N tile
    unsigned w_off[2];
    bool w_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 8 * (wave + NW * j) + (lane >> 3);
        w_ok[j] = n < p.N;
        w_off[j] = (unsigned)n * (unsigned)p.K * 2u + st_chunk;
    }
    auto issue_b = [&](int cb, int tap, bool on, int slot_off) {   // weight tile of (channel block, tap) into the ring slot at slot_off
        const unsigned k0 = (unsigned)(tap * Ctot + cb * BK) * 2u;
        const char* g0 = (w_ok[0] && on) ? (const char*)p.W + ((unsigned long long)w_off[0] + k0) : zp;
        glds16(g0, (half_t*)(smem + BOFF + slot_off + wave * 1024));
        if (xw) {
            const char* g1 = (w_ok[1] && on) ? (const char*)p.W + ((unsigned long long)w_off[1] + k0) : zp;
            glds16(g1, (half_t*)(smem + BOFF + slot_off + (NW + wave) * 1024));
        }
    };
    // LDS-DMA instructions a wave issues in the step of tap `tap`
    auto n_issued = [](int tap, bool x) constexpr -> int { return 1 + (x ? 1 : 0) + (tap < 6 ? 1 : 0) + (x && tap == 0 ? 1 : 0); };

    if constexpr (NL > 0) {
        if (wave >= NW) {
            // ---------------- loader waves: loader l issues, per step, three weight pieces (l, l + NL, l + 2 NL of tile t+4) and,
            // during taps 0..5, three super-tile pieces of the next block (9 tap + l, + NL, + 2 NL); a piece that does not exist is
            // fetched from the zero page into a scratch kilobyte, so every loader issues the same count and the vmcnt waits
            // are constants.
            static_assert(3 * NL >= NPB && 6 * 9 >= NPA && 3 * NL >= 9, "loader schedule");
            const int l = wave - NW;
            const int ms_lane = st_row0 + (lane >> 3);
            const char* an_src = zp; unsigned an_cs = 0, an_c0 = 0; bool an_on = false;
            auto set_next_block = [&](int cb, bool on) {
                const bool first = cb * BK < p.C1;
                an_src = (const char*)(first ? p.A : p.A2);
                an_cs = (unsigned)(first ? p.C1 : p.C2);
                an_c0 = (unsigned)(first ? cb * BK : cb * BK - p.C1);
                an_on = on;
            };
            auto issue_a = [&](int q, bool exists, int buf_off) {     // super-tile piece q of the next block (q wave-uniform)
                const int ms = ms_lane + 8 * q;
                const bool ok = exists && an_on && (unsigned)ms < (unsigned)src_pixels;
                const char* g = ok ? an_src + ((unsigned long long)((unsigned)ms * an_cs + an_c0) * 2ull + st_chunk) : zp;
                glds16(g, (half_t*)(smem + (exists ? buf_off + q * 1024 : DUMP)));
            };
            const unsigned w_lane = (unsigned)(n0 + (lane >> 3)) * (unsigned)p.K * 2u + st_chunk;
            auto issue_b = [&](int cb, int tap, bool on, int slot_off) {
                const unsigned k0 = (unsigned)(tap * Ctot + cb * BK) * 2u;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int q = l + NL * j;
                    const bool exists = q < NPB;
                    const bool ok = exists && on && n0 + 8 * q + (lane >> 3) < p.N;
                    const char* g = ok ? (const char*)p.W + ((unsigned long long)w_lane + (unsigned long long)((unsigned)(8 * q) * (unsigned)p.K * 2u) + k0) : zp;
                    glds16(g, (half_t*)(smem + (exists ? BOFF + slot_off + q * 1024 : DUMP)));
                }
            };
            auto n_issued = [](int tap) constexpr -> int { return 3 + (tap < 6 ? 3 : 0); };
            if (nsteps > 0) {
                set_next_block(cb_lo, true);
                for (int q = l; q < ABUF / 1024; q += NL) issue_a(q, true, 0);
                issue_b(cb_lo, 0, true, 0);
                issue_b(cb_lo, 1, true, BBUF);
                issue_b(cb_lo, 2, true, 2 * BBUF);
                issue_b(cb_lo, 3, true, 3 * BBUF);
                asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
            }
            int bs_prev = 4 * BBUF, bs = 0, bs_next = BBUF;
            for (int cbi = 0; cbi < cb_hi - cb_lo; ++cbi) {
                const int cb = cb_lo + cbi;
                const int abuf_n = ABUF - (cbi & 1) * ABUF;
                set_next_block(min(cb + 1, cb_hi - 1), cb + 1 < cb_hi);
                const int steps_left = nsteps - cbi * 9;
                auto one_step = [&](auto tap_tag) {
                    constexpr int tap = decltype(tap_tag)::value;
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_issued((tap + 8) % 9) + n_issued((tap + 7) % 9)) : "memory");
                    asm volatile("s_barrier" ::: "memory");
                    if constexpr (tap < 6) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const int k = l + NL * j, q = 9 * tap + k;
                            issue_a(q, k < 9 && q < ABUF / 1024, abuf_n);
                        }
                    }
                    issue_b(cb + (tap + 4) / 9, (tap + 4) % 9, tap + 4 < steps_left, bs_prev);
                    bs_prev = bs; bs = bs_next; bs_next = bs_next == (NSB - 1) * BBUF ? 0 : bs_next + BBUF;
                };
                one_step(IntTag<0>{}); one_step(IntTag<1>{}); one_step(IntTag<2>{});
                one_step(IntTag<3>{}); one_step(IntTag<4>{}); one_step(IntTag<5>{});
                one_step(IntTag<6>{}); one_step(IntTag<7>{}); one_step(IntTag<8>{});
            }
        }
    }
    // ---- fragment addressing
    const int fr = lane & 15, fq = lane >> 4;
    int a_row[TM], a_x[TM], a_y[TM];
    unsigned a_edge[TM];                 // bit 0: x == 0, 1: x == W-1, 2: y == 0, 3: y == H-1, 4: row past M
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wr * WM + i * 16 + fr, m = m0 + r;
        const int x = m % W, y = (m / W) % H;
        a_row[i] = r; a_x[i] = x; a_y[i] = y;
        a_edge[i] = (x == 0 ? 1u : 0u) | (x == W - 1 ? 2u : 0u) | (y == 0 ? 4u : 0u) | (y == H - 1 ? 8u : 0u) | (m >= p.M ? 16u : 0u);
    }
    int b_off[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wc * WN + j * 16 + fr;
        b_off[j] = BOFF + row * 128 + ((fq ^ (row & 6)) << 4);
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // One K tile = two k-steps of 32 channels (16-B chunks fq and fq + 4 of a row: LDS addresses 64 B apart).
    // Registers hold ONE set of fragments per k-step: a0/b0 (k-step 0) and a1/b1 (k-step 1).
    half8 a0[TM], a1[TM], b0[TN], b1[TN];
    auto a_addr = [&](int i, int abuf, int ky, int kx) -> int {
        const unsigned tmask = 16u | (kx == 0 ? 1u : 0u) | (kx == 2 ? 2u : 0u) | (ky == 0 ? 4u : 0u) | (ky == 2 ? 8u : 0u);
        int sr;
        if constexpr (UPS) sr = ((a_y[i] + ky - 1) >> 1) * Wi + ((a_x[i] + kx - 1) >> 1) - ups_rel;
        else sr = a_row[i] + ky * W + kx;
        const int ad = abuf + sr * 128 + ((fq ^ (sr & 6)) << 4);
        return (a_edge[i] & tmask) ? ZOFF : ad;
    };
    auto mma = [&](const half8 (&af)[TM], const half8 (&bf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    };

    // Step t = (channel block, tap).  Its k-step-0 fragments were read during step t-1; the step itself is
    //     wait + barrier | LDS-DMA issues | read k-step 1 of t  || MFMAs of k-step 0 | read k-step 0 of t+1 || MFMAs of k-step 1
    // so every fragment read has ten MFMAs' time to land.  LDS-DMA groups: group g = what a wave issues in step g = weight
    // tile g+4 into ring slot (g+4) % 5 and, during taps 0..5, pieces of the next channel block's super-tile (prologue:
    // g = -4 super-tile 0 + weight 0, g = -3 .. -1 weights 1 .. 3).  The wait of step t leaves groups t-1 and t-2 in flight:
    // weight t+1 (group t-3) and, at tap 8, the next super-tile (groups up to tap 5) have landed.  The barrier publishes them
    // and says every wave has finished the reads of step t-1: ring slot (t-1) % 5 = (t+4) % 5 and, at tap 0, the other
    // super-tile buffer are free.
    const bool computes = NL == 0 || wave < NW;
    if (nsteps > 0 && computes) {
        if constexpr (NL == 0) {
            set_next_block(cb_lo, true);
#pragma unroll
            for (int j = 0; j < 6; ++j) issue_a(j, 0);
            if (xw) issue_a(6, 0);
            issue_b(cb_lo, 0, true, 0);
            issue_b(cb_lo, 1, true, BBUF);
            issue_b(cb_lo, 2, true, 2 * BBUF);
            issue_b(cb_lo, 3, true, 3 * BBUF);
            if (xw) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; ++i) a0[i] = *(const half8*)(smem + a_addr(i, 0, 0, 0));
#pragma unroll
        for (int j = 0; j < TN; ++j) b0[j] = *(const half8*)(smem + b_off[j]);
        if constexpr (!(VAR & 2)) __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    int bs_prev = 4 * BBUF, bs = 0, bs_next = BBUF;     // ring slots (byte offsets) of steps t-1 (= t+4), t, t+1
    for (int cbi = 0; computes && cbi < cb_hi - cb_lo; ++cbi) {
        const int cb = cb_lo + cbi;
        const int abuf = (cbi & 1) * ABUF, abuf_n = ABUF - abuf;
        if constexpr (NL == 0) set_next_block(min(cb + 1, cb_hi - 1), cb + 1 < cb_hi);
        const int steps_left = nsteps - cbi * 9;          // weight tile t+4 exists while tap + 4 < steps_left
        auto one_step = [&](auto tap_tag) {
            constexpr int tap = decltype(tap_tag)::value;
            constexpr int ky = tap / 3, kx = tap - ky * 3;
            if constexpr (NL == 0) {   // groups t-1 and t-2 stay in flight
                constexpr int p1 = (tap + 8) % 9, p2 = (tap + 7) % 9;
                if (xw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_issued(p1, true) + n_issued(p2, true)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_issued(p1, false) + n_issued(p2, false)) : "memory");
            }
            asm volatile("s_barrier" ::: "memory");
            auto issue_group = [&]() {
                if constexpr (NL > 0) return;
                if (tap < 6) {
                    issue_a(tap, abuf_n);
                    if (tap == 0 && xw) issue_a(6, abuf_n);
                }
                issue_b(cb + (tap + 4) / 9, (tap + 4) % 9, tap + 4 < steps_left, bs_prev);
            };
            // the two waves of a SIMD (w, w + NW/2) would otherwise both spend the head of the step issuing LDS-DMA with the
            // matrix pipe idle: the second half of the workgroup issues its group at the END of the step instead
            if (!late) issue_group();
            __builtin_amdgcn_sched_barrier(0);
            // each half: the fragments it multiplies are back (one lgkmcnt(0), said with the builtin so that the compiler
            // does not add its own — it would put that after the new reads and wait for them too), then the other
            // k-step's reads go out, then ten MFMAs run while they land
            if constexpr (VAR & 2) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) a1[i] = *(const half8*)(smem + (a_addr(i, abuf, ky, kx) ^ 64));    // k-step 1 of this step
#pragma unroll
            for (int j = 0; j < TN; ++j) b1[j] = *(const half8*)(smem + bs + (b_off[j] ^ 64));
            mma(a0, b0);
            if constexpr (VAR & 4) {
                __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (VAR & 2) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
            }
            {   // k-step 0 of the next step (next tap, or tap 0 of the next channel block)
                const int nky = tap == 8 ? 0 : (tap + 1) / 3, nkx = tap == 8 ? 0 : (tap + 1) % 3;
                const int ab = tap == 8 ? abuf_n : abuf;
#pragma unroll
                for (int i = 0; i < TM; ++i) a0[i] = *(const half8*)(smem + a_addr(i, ab, nky, nkx));
#pragma unroll
                for (int j = 0; j < TN; ++j) b0[j] = *(const half8*)(smem + bs_next + b_off[j]);
            }
            mma(a1, b1);
            if constexpr (VAR & 4) {
                __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(VAR & 2)) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (late) issue_group();
            bs_prev = bs; bs = bs_next; bs_next = bs_next == (NSB - 1) * BBUF ? 0 : bs_next + BBUF;
        };
        one_step(IntTag<0>{}); one_step(IntTag<1>{}); one_step(IntTag<2>{});
        one_step(IntTag<3>{}); one_step(IntTag<4>{}); one_step(IntTag<5>{});
        one_step(IntTag<6>{}); one_step(IntTag<7>{}); one_step(IntTag<8>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // all fragment reads done, nothing in flight: the super-tile buffer becomes the epilogue's scratch

    // ---------------- epilogue: the whole BM x BN tile goes through LDS at once (fp32 [BM][BN+4] in the super-tile buffers).
    // Every thread first issues the residual loads of all its output chunks, then adds bias + the image's time-embedding row +
    // residual, rounds and stores 16 B per chunk; the GroupNorm column statistics (sum, sum of squares of the ROUNDED outputs
    // per channel) are taken by NT / BN row groups in parallel and folded in group order: fixed order, bit-reproducible.
    constexpr int CH = BN / 8, NIT = (BM * CH + NT - 1) / NT;
    constexpr int NG = NT / BN, RPG = (BM + NG - 1) / NG;          // cstat: row groups, rows per group
    static_assert(BM * LDS_N * 4 <= 2 * ABUF && NG * BN * 2 * 4 <= NSB * BBUF, "epilogue LDS");
    float* stage = (float*)smem;
    float* part = (float*)(smem + BOFF);                            // [NG][BN][2]
    half_t* __restrict__ Out = p.Out;
    if (computes) {
        const int rb0 = wr * WM;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    stage[(rb0 + i * 16 + fq * 4 + r) * LDS_N + wc * WN + j * 16 + fr] = acc[i][j][r];
    }
    half8 rs[NIT];
    if (p.residual && p.splits <= 1) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int c = tid + k * NT, row = c / CH, nc = c - row * CH;
            const int m = m0 + row, n = n0 + nc * 8;
            rs[k] = (c < BM * CH && m < p.M && n < p.N) ? *(const half8*)(p.residual + (long long)m * p.ldr + n) : (half8){0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int c = tid + k * NT;
        if (c >= BM * CH) break;
        const int row = c / CH, nc = c - row * CH;
        const int m = m0 + row, n = n0 + nc * 8;
        float* sp = stage + row * LDS_N + nc * 8;
        if (m < p.M && n < p.N) {
            const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
            if (p.splits > 1) {
                float* w = p.ws + ((long long)blockIdx.y * p.M + m) * p.N + n;
                *(f32x4*)w = s0;
                *(f32x4*)(w + 4) = s1;
                continue;
            }
            float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            if (p.bias) {
                const f32x4 b0 = *(const f32x4*)(p.bias + n), b1 = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
            }
            if (p.rowvec) {
                const float* rv = p.rowvec + (long long)(m / p.rows_per_batch) * p.N + n;
                const f32x4 b0 = *(const f32x4*)rv, b1 = *(const f32x4*)(rv + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
            }
            if (p.residual) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rs[k][e];
            }
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(v[e] * p.out_scale);
            *(half8*)(Out + (long long)m * p.ldo + n) = o;
            if (p.cstat_out) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sp[e] = (float)o[e];
            }
        } else if (p.cstat_out) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sp[e] = 0.f;
        }
    }
    if (p.cstat_out && p.splits <= 1) {
        __syncthreads();
        if (tid < NG * BN) {
            const int g = tid / BN, col = tid - g * BN;
            float cs = 0.f, cq = 0.f;
            const int r1 = min(BM, (g + 1) * RPG);
            for (int r = g * RPG; r < r1; ++r) {
                const float v = stage[r * LDS_N + col];
                cs += v;
                cq += v * v;
            }
            part[(g * BN + col) * 2] = cs;
            part[(g * BN + col) * 2 + 1] = cq;
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float cs = 0.f, cq = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) { cs += part[(g * BN + tid) * 2]; cq += part[(g * BN + tid) * 2 + 1]; }
            float* co = p.cstat_out + ((long long)(m0 / BM) * p.N + n0 + tid) * 2;
            co[0] = cs; co[1] = cq;
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NS, int NL, bool CONV>
static int launch_ns(const IefGemmParams& p, int tiles, int splits, int batch, hipStream_t st) {
    constexpr int NT = 64 * (WAVES_M * WAVES_N + NL), RP = (NL > 0 ? 64 * NL : NT) / 8;
    constexpr long long lds = 2ll * NS * (((BM + RP - 1) / RP) * RP + ((BN + RP - 1) / RP) * RP) * 64;
    if constexpr (lds > 160 * 1024) {
        return IEF_ESHAPE;   // this ring depth does not fit the 160 KiB LDS for this tile
    } else {
        hipLaunchKernelGGL((igemm_f16_kernel<BM, BN, WAVES_M, WAVES_N, NS, NL, CONV>), dim3(tiles, splits, batch), dim3(NT), 0,
                           st, p);
        return IEF_OK;
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NL, bool CONV>
static int launch_igemm(const IefGemmParams& p, int batch, hipStream_t st) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int splits = p.splits > 1 ? p.splits : 1;
    int rc;
    switch (p.stages) {
        case 3: rc = launch_ns<BM, BN, WAVES_M, WAVES_N, 3, NL, CONV>(p, tiles, splits, batch, st); break;
        case 4: rc = launch_ns<BM, BN, WAVES_M, WAVES_N, 4, NL, CONV>(p, tiles, splits, batch, st); break;
        default: rc = launch_ns<BM, BN, WAVES_M, WAVES_N, 2, NL, CONV>(p, tiles, splits, batch, st); break;
    }
    if (rc) return rc;
    IEF_LAUNCH_CHECK();
    if (splits > 1 && !p.cnt) {
        const long long total = (long long)p.M * (p.N / 8);
        int grid = (int)((total + 255) / 256);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(grid), dim3(256), 0, st, p);
        IEF_LAUNCH_CHECK();
    }
    return IEF_OK;
}

static void launch_splitk_reducer(const IefGemmParams& p, hipStream_t st) {
    const long long total = (long long)p.M * (p.N / 8);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(grid), dim3(256), 0, st, p);
}

// tile 14: conv3x3_halo_kernel<256, 80>; plain 3x3 / stride 1 / pad 1 convolutions on rows of at most 64 pixels
static int launch_conv_halo(IefGemmParams p, hipStream_t st) {
    if (p.stride != 1 || p.pad_hi_only || p.CE1 || p.CE2 || p.Wd < 2 || p.H < 2) return IEF_ESHAPE;
    if (p.ups) {   // fused nearest-2x upsample: loader-wave form, tiles inside one image, whole output rows per tile
        if (p.tile_hint != 15 || p.Wd > 128 || (p.H * p.Wd) % 256 || 256 % p.Wd) return IEF_ESHAPE;
    } else if (p.Wd > 64) return IEF_ESHAPE;
    if ((p.C1 + p.C2) % 64) return IEF_ESHAPE;
    const int ncb = (p.C1 + p.C2) / 64;
    if (p.splits > ncb) return IEF_ESHAPE;
    if (p.cstat_out && p.splits > 1) return IEF_EINVAL;
    p.cnt = nullptr;                                   // split-K slabs are always summed by the reducer launch here
    constexpr int BM = 256, BN = 80;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int splits = p.splits > 1 ? p.splits : 1;
    if (p.ups) hipLaunchKernelGGL((conv3x3_halo_kernel<BM, BN, 8, 1, 4, 3, true>), dim3(tiles, splits, 1), dim3(768), 0, st, p);
    else if (p.tile_hint == 15) hipLaunchKernelGGL((conv3x3_halo_kernel<BM, BN, 8, 1, 4>), dim3(tiles, splits, 1), dim3(768), 0, st, p);
    else hipLaunchKernelGGL((conv3x3_halo_kernel<BM, BN, 8, 1, 0>), dim3(tiles, splits, 1), dim3(512), 0, st, p);
    IEF_LAUNCH_CHECK();
    if (splits > 1) {
        launch_splitk_reducer(p, st);
        IEF_LAUNCH_CHECK();
    }
    return IEF_OK;
}

// tile ids (IefGemmParams.tile_hint); the host binding picks one per layer shape
//   1: 128x128 (2x2 waves)   2: 64x128 (2x2)    3: 64x64 (2x2)     4: 128x64 (2x2)
//   5: 64x160 (2x2)          6: 128x160 (2x2)   7: 128x160 (4x2)   8: 256x128 (4x2)   9: 128x128 (4x2)
//   with loader waves (NL): 16: 128x160 (4x2)+4   17: 128x128 (4x2)+4   18: 256x128 (4x2)+4   19: 64x160 (2x2)+2   20: 64x64 (2x2)+2
//   21: 128x160 (2x2)+2
//   14: conv3x3_halo_kernel 256x80 (8x1), convolutions only (launch_conv_halo); 15: the same with four loader waves
template <bool CONV>
static int dispatch_igemm(const IefGemmParams& p, int batch, hipStream_t st) {
    switch (p.tile_hint) {
        case 1: return launch_igemm<128, 128, 2, 2, 0, CONV>(p, batch, st);
        case 2: return launch_igemm<64, 128, 2, 2, 0, CONV>(p, batch, st);
        case 3: return launch_igemm<64, 64, 2, 2, 0, CONV>(p, batch, st);
        case 4: return launch_igemm<128, 64, 2, 2, 0, CONV>(p, batch, st);
        case 5: return launch_igemm<64, 160, 2, 2, 0, CONV>(p, batch, st);
        case 6: return launch_igemm<128, 160, 2, 2, 0, CONV>(p, batch, st);
        case 7: return launch_igemm<128, 160, 4, 2, 0, CONV>(p, batch, st);
        case 8: return launch_igemm<256, 128, 4, 2, 0, CONV>(p, batch, st);
        case 9: return launch_igemm<128, 128, 4, 2, 0, CONV>(p, batch, st);
        case 16: return launch_igemm<128, 160, 4, 2, 4, CONV>(p, batch, st);
        case 17: return launch_igemm<128, 128, 4, 2, 4, CONV>(p, batch, st);
        case 18: return launch_igemm<256, 128, 4, 2, 4, CONV>(p, batch, st);
        case 19: return launch_igemm<64, 160, 2, 2, 2, CONV>(p, batch, st);
        case 20: return launch_igemm<64, 64, 2, 2, 2, CONV>(p, batch, st);
        case 21: return launch_igemm<128, 160, 2, 2, 2, CONV>(p, batch, st);
        default: break;
    }
    auto nblk = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * batch; };
    if (nblk(128, 128) >= 384) return launch_igemm<128, 128, 2, 2, 0, CONV>(p, batch, st);
    if (nblk(64, 128) >= 256) return launch_igemm<64, 128, 2, 2, 0, CONV>(p, batch, st);
    return launch_igemm<64, 64, 2, 2, 0, CONV>(p, batch, st);
}

extern "C" int ief_gemm_tile_bm(int tile_hint) {
    switch (tile_hint) {
        case 1: case 4: case 6: case 7: case 9: case 16: case 17: case 21: return 128;
        case 2: case 3: case 5: case 19: case 20: return 64;
        case 8: case 14: case 15: case 18: return 256;
        default: return 0;
    }
}

extern "C" int ief_gemm_tile_bn(int tile_hint) {
    switch (tile_hint) {
        case 1: case 2: case 8: case 9: case 17: case 18: return 128;
        case 3: case 4: case 20: return 64;
        case 5: case 6: case 7: case 16: case 19: case 21: return 160;
        case 14: case 15: return 80;
        default: return 0;
    }
}

static int check_common(const IefGemmParams& p) {
    if (!p.A || !p.W || !p.Out || !p.zeros) return IEF_EINVAL;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return IEF_ESHAPE;
    if ((p.N & 7) || (p.K & 7) || (p.ldw & 7) || (p.ldo & 7)) return IEF_EALIGN;
    if (p.residual && (p.ldr & 7)) return IEF_EALIGN;
    if (p.rowvec && p.rows_per_batch <= 0) return IEF_ESHAPE;
    if (p.splits > 1 && !p.ws) return IEF_EINVAL;
    if (p.splits > 64) return IEF_ESHAPE;
    if (p.stages != 0 && (p.stages < 2 || p.stages > 4)) return IEF_ESHAPE;
    if (p.flags & 2) {  // fused GEGLU epilogue
        if (!p.bias || p.rowvec || p.residual || p.splits > 1 || (p.N & 15) || p.rstat_out) return IEF_EINVAL;
    }
    if (p.rstat_in && (!p.colsum || p.rstat_slots <= 0 || p.splits > 1 || !(p.ln_eps > 0.f))) return IEF_EINVAL;
    if (p.rstat_out && p.splits > 1) return IEF_EINVAL;
    if (p.cstat_out && ((p.splits > 1 && !p.cnt) || (p.flags & 2) || ief_gemm_tile_bm(p.tile_hint) == 0)) return IEF_EINVAL;
    // 32-bit byte offsets inside each operand
    if ((long long)p.N * p.ldw * 2 >= (1ll << 32)) return IEF_ESHAPE;
    return IEF_OK;
}

extern "C" int ief_gemm_f16(const IefGemmParams* pp, int batch, void* stream) {
    if (!pp) return IEF_EINVAL;
    IefGemmParams p = *pp;
    int rc = check_common(p);
    if (rc) return rc;
    if (p.lda & 7) return IEF_EALIGN;
    if (batch <= 0) return IEF_ESHAPE;
    if (p.cstat_out && batch != 1) return IEF_EINVAL;
    if (p.splits > 1 && batch != 1) return IEF_ESHAPE;
    if ((long long)p.M * p.lda * 2 >= (1ll << 32)) return IEF_ESHAPE;
    p.ups = 0; p.H = p.Wd = 1;
    return dispatch_igemm<false>(p, batch, (hipStream_t)stream);
}

extern "C" int ief_conv3x3_f16(const IefGemmParams* pp, void* stream) {
    if (!pp) return IEF_EINVAL;
    IefGemmParams p = *pp;
    const int Ctot = p.C1 + p.C2;
    if (p.C1 <= 0 || p.C2 < 0 || (p.C1 % 64) || (p.C2 % 64)) return IEF_ESHAPE;
    if (p.C2 > 0 && !p.A2) return IEF_EINVAL;
    if (p.stride != 1 && p.stride != 2) return IEF_ESHAPE;
    if (p.ups != 0 && p.ups != 1) return IEF_ESHAPE;
    if (p.ups && ((p.H & 1) || (p.Wd & 1))) return IEF_ESHAPE;
    const int pad_total = p.pad_hi_only ? 1 : 2;
    p.Ho = (p.H + pad_total - 3) / p.stride + 1;
    p.Wo = (p.Wd + pad_total - 3) / p.stride + 1;
    if (p.CE1 < 0 || p.CE2 < 0 || (p.CE1 % 64) || (p.CE2 % 64)) return IEF_ESHAPE;
    if ((p.CE1 > 0 && !p.E1) || (p.CE2 > 0 && !p.E2) || (p.CE2 > 0 && p.CE1 == 0)) return IEF_EINVAL;
    if ((p.CE1 + p.CE2) > 0 && (p.stride != 1 || p.ups != 0)) return IEF_ESHAPE;
    p.K = 9 * Ctot + p.CE1 + p.CE2;
    p.ldw = p.K;
    if (p.batch_images <= 0) return IEF_ESHAPE;
    p.M = p.batch_images * p.Ho * p.Wo;
    if (p.rowvec && p.rows_per_batch <= 0) p.rows_per_batch = p.Ho * p.Wo;  // caller may share one row across the batch
    int rc = check_common(p);
    if (rc) return rc;
    const long long in_pix = (long long)p.batch_images * (p.H >> p.ups) * (p.Wd >> p.ups);
    const int cmax = p.C1 > p.C2 ? p.C1 : p.C2;
    const int emax = p.CE1 > p.CE2 ? p.CE1 : p.CE2;
    if (in_pix * cmax * 2 >= (1ll << 32) || (long long)p.M * emax * 2 >= (1ll << 32)) return IEF_ESHAPE;
    p.strideA = p.strideW = p.strideO = p.strideR = 0;
    if (p.tile_hint == 14 || p.tile_hint == 15) return launch_conv_halo(p, (hipStream_t)stream);
    return dispatch_igemm<true>(p, 1, (hipStream_t)stream);
}
