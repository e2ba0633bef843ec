// 3x3 / stride 1 / pad 1 convolution on operand planes with the input tile RESIDENT in LDS across the nine taps (the split-operand
// form of conv3x3_halo_kernel, gemm_conv.hip).
//
// Measured on igemm_x3p_kernel (ablation builds, tests/abl_x3p.sh): the 128 x 160 x 32 K tile needs 36 KiB of LDS-DMA for 960
// matrix-pipe cycles per SIMD -- at the ~32 B/clk a CU fills its LDS from L2 the two are equal, and the loop runs at about half the
// matrix rate.  The lever is fewer staged bytes per MFMA.  A workgroup here owns 256 CONSECUTIVE output pixels of the flattened
// [B H W] axis x 80 output channels and, per 32-channel block, stages the pixel range m0 - (W + 1) .. m0 + 255 + W + 1 (the
// super-tile, <= 386 rows of 64 B per plane) ONCE; tap (ky, kx) of output row r is super-tile row r + ky W + kx, a lane whose tap
// falls outside the image reads a zero row.  Per step (one tap of one channel block: 256 x 80 x 32, the same 30 MFMAs per wave as
// the implicit GEMM's K tile) 10 KiB of weights + 5.6 KiB of input are staged instead of 36 KiB.
//
//   12 waves: 8 compute (wave tile 32 x 80: fragments + MFMAs only) + 4 LOADER waves (every LDS-DMA and every vmcnt wait).
//   LDS: [hi buf0 | hi buf1 | zero row | lo buf0 | lo buf1 | zero row] (the two planes of a super-tile lie a constant AD apart, so
//   does the pair of zero rows: one address select serves both reads), weight ring of 5 slots [hi 5 KiB | lo 5 KiB], 1 KiB where
//   filler pieces land.  Rows are 64 B, chunk c of row r in slot c ^ xp_swz(r) (x3p_common.h: conflict-free from any start row).
//   Loader schedule per step (constant counts -> compile-time vmcnt): weight tile of step t + 4 (3 pieces per loader, 2 fillers in
//   all), and during taps 0..4 three pieces of the NEXT channel block's super-tile; the wait of step t leaves groups t - 1, t - 2
//   in flight, so at the barrier of step t the weights of step t + 1 and (at tap 8) the whole next super-tile are in LDS.
//   Compute waves, step t: barrier | request the A fragments of step t + 1 | for each of the 5 column blocks: 6 MFMAs of step t,
//   then request that block's weight fragments of step t + 1 into the registers just consumed.  One set of weight-fragment
//   registers, two of input fragments (alternating by step parity; the loop body covers two channel blocks = 18 steps so the
//   parity is a compile-time constant).
//   UPS: the nearest-2x upsample of Upsample2D fused in: the super-tile is the SOURCE pixel range the tile touches, the fragment
//   row is computed per lane and tap; tiles lie inside one image (H W a multiple of 256, W | 256).
//
// Epilogue: a lane holds one output row and four consecutive columns of each block (x3p_common.h, xp_store); split-K = ranges of
// channel blocks, fp32 slabs summed by x3p_reduce_kernel.
// Replaces: ResnetBlock2D.conv1 / conv2 (/root/reference/pnp/model/register.py:139-175), Upsample2D.conv.
#include "x3p_common.h"

template <int N>
struct IntTag { static constexpr int value = N; };

template <bool UPS>
__global__ __launch_bounds__(768) void conv3x3_halo_x3p_kernel(const IefGemmX3pParams p) {
    constexpr int BM = 256, BN = 80, BK = XP_BK, NW = 8, NL = 4;
    constexpr int TM = 2, TN = 5;
    constexpr int WMAX = 64;
    constexpr int NPA = (BM + 2 * WMAX + 2 + 15) / 16;      // 25 pieces (16 rows x 64 B) per plane of a super-tile
    constexpr int APL = NPA * 1024;                          // bytes per plane of one super-tile buffer
    constexpr int AD = 2 * APL + 128;                        // hi -> lo distance inside the A region
    constexpr int ZA = 2 * APL;                              // zero row of the hi plane; its lo twin lies at ZA + AD
    constexpr int NPB = BN / 16;                             // 5 pieces per plane of a weight tile
    constexpr int BPL = NPB * 1024, BBUF = 2 * BPL, NSB = 5;
    constexpr int BOFF = 2 * AD, DUMP = BOFF + NSB * BBUF;
    constexpr int LDS_BYTES = DUMP + 1024;
    constexpr int NU = 2 * NPA;                              // super-tile pieces of a channel block (both planes)
    constexpr int ATAPS = 5, APT = 3;                        // they are issued during taps 0 .. ATAPS-1, APT per loader and step
    static_assert(ATAPS * APT * NL >= NU && 3 * NL >= 2 * NPB, "loader schedule");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    // tile order as igemm_x3p_kernel: each XCD (its own L2) takes a contiguous run of logical ids; inside it, groups of XP_GROUP_M
    // row blocks are walked column by column -- the workgroups resident on an XCD cover a patch of the output, so an input
    // super-tile and a weight tile are fetched into that L2 once per patch (PMC, 32x32x640 convolution: N-first order made every
    // XCD fetch all 14.7 MB of weights: 4.5x the algorithmic bytes)
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = lid / (XP_GROUP_M * tiles_n), within = lid - grp * (XP_GROUP_M * tiles_n);
    const int gsz = min(XP_GROUP_M, tiles_m - grp * XP_GROUP_M);
    const int tn = within / gsz, tm = grp * XP_GROUP_M + (within - tn * gsz);
    const int m0 = tm * BM, n0 = tn * BN;
    const int W = p.Wd, H = p.H;
    const int Ctot = p.C1 + p.C2;
    const char* __restrict__ zp = (const char*)p.zeros;
    const int Wi = W >> 1, Hi = H >> 1;
    const int ups_img = m0 / (H * W), ups_y0 = (m0 - ups_img * H * W) / W;
    const int ups_rel = ((ups_y0 - 1) >> 1) * Wi - 1;
    const int src_pixels = UPS ? p.batch_images * Hi * Wi : p.M;
    const int st_row0 = UPS ? ups_img * Hi * Wi + ups_rel : m0 - (W + 1);

    const int ncb = Ctot / BK;
    int cb_lo = 0, cb_hi = ncb;
    if (p.splits > 1) {
        const int per = (ncb + p.splits - 1) / p.splits;
        cb_lo = min(ncb, (int)blockIdx.y * per);
        cb_hi = min(ncb, cb_lo + per);
    }
    const int nblk = cb_hi - cb_lo, nsteps = nblk * 9;

    if (tid < 16) { ((float*)(smem + ZA))[tid] = 0.f; ((float*)(smem + ZA + AD))[tid] = 0.f; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the first s_barrier publishes the zero rows

    if (wave >= NW) {
        // ------------------------------------------------------------------------------------------------ loader waves
        const int l = wave - NW;
        bool abl_prologue = true;                 // (ablation builds only: XP_ABL & 1 keeps the prologue's LDS-DMA, drops the loop's)
        const unsigned st_chunk = (unsigned)(XP_LANE_CHUNK(lane) * 16);
        const int ms_lane = st_row0 + (lane >> 2);
        const char* an_src = zp; long long an_pl = 0; unsigned an_cs = 0, an_c0 = 0; bool an_on = false;
        auto set_next_block = [&](int cb, bool on) {
            const bool first = cb * BK < p.C1;
            an_src = (const char*)(first ? p.A : p.A2);
            an_pl = (first ? p.planeA : p.planeA2) * 2;
            an_cs = (unsigned)(first ? p.C1 : p.C2);
            an_c0 = (unsigned)(first ? cb * BK : cb * BK - p.C1);
            an_on = on;
        };
        auto issue_a = [&](int u, int buf_off) {              // super-tile piece u (wave-uniform) of the next block: plane u / NPA
            const bool exists = u < NU;
            const int pl = u >= NPA ? 1 : 0, q = u - pl * NPA;
            const int ms = ms_lane + 16 * q;
            const bool ok = exists && an_on && (unsigned)ms < (unsigned)src_pixels;
            const char* g = ok ? an_src + ((long long)pl * an_pl + (long long)((unsigned)ms * an_cs + an_c0) * 2 + st_chunk) : zp;
            if (!(XP_ABL & 1) || abl_prologue) glds16(g, (half_t*)(smem + (exists ? pl * AD + buf_off + q * 1024 : DUMP)));
        };
        const long long w_lane = (long long)(n0 + (lane >> 2)) * p.K * 2 + st_chunk;
        auto issue_b = [&](int cb, int tap, bool on, int slot_off) {
            const long long k0 = (long long)(tap * Ctot + cb * BK) * 2;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int v = l + NL * j;                     // weight piece: plane v / NPB, rows 16 (v % NPB) ..
                const bool exists = v < 2 * NPB;
                const int pl = v >= NPB ? 1 : 0, q = v - pl * NPB;
                const bool ok = exists && on && n0 + 16 * q + (lane >> 2) < p.N;
                const char* g = ok ? (const char*)p.W + ((long long)pl * p.planeW * 2 + w_lane + (long long)(16 * q) * p.K * 2 + k0) : zp;
                if (!(XP_ABL & 1) || abl_prologue) glds16(g, (half_t*)(smem + (exists ? BOFF + slot_off + pl * BPL + q * 1024 : DUMP)));
            }
        };
        auto n_issued = [](int tap) constexpr -> int { return 3 + (tap < ATAPS ? APT : 0); };
        if (nsteps > 0) {
            set_next_block(cb_lo, true);
            for (int u = l; u < ((NU + NL - 1) / NL) * NL; u += NL) issue_a(u, 0);
            issue_b(cb_lo, 0, true, 0);
            issue_b(cb_lo, 1, nsteps > 1, BBUF);
            issue_b(cb_lo, 2, nsteps > 2, 2 * BBUF);
            issue_b(cb_lo, 3, nsteps > 3, 3 * BBUF);
            asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            XP_BARRIER();
        }
        abl_prologue = false;
        int bs_prev = 4 * BBUF, bs = 0, bs_next = BBUF;
        for (int cbi = 0; cbi < nblk; ++cbi) {
            const int cb = cb_lo + cbi;
            const int abuf_n = APL - (cbi & 1) * APL;
            set_next_block(min(cb + 1, cb_hi - 1), cb + 1 < cb_hi);
            const int steps_left = nsteps - cbi * 9;
            auto one_step = [&](auto tap_tag) {
                constexpr int tap = decltype(tap_tag)::value;
                if (!(XP_ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n_issued((tap + 8) % 9) + n_issued((tap + 7) % 9)) : "memory");
                XP_BARRIER();
                if constexpr (tap < ATAPS) {
#pragma unroll
                    for (int j = 0; j < APT; ++j) issue_a(APT * NL * tap + l + NL * j, abuf_n);
                }
                issue_b(cb + (tap + 4) / 9, (tap + 4) % 9, tap + 4 < steps_left, bs_prev);
                bs_prev = bs; bs = bs_next; bs_next = bs_next == (NSB - 1) * BBUF ? 0 : bs_next + BBUF;
            };
            one_step(IntTag<0>{}); one_step(IntTag<1>{}); one_step(IntTag<2>{});
            one_step(IntTag<3>{}); one_step(IntTag<4>{}); one_step(IntTag<5>{});
            one_step(IntTag<6>{}); one_step(IntTag<7>{}); one_step(IntTag<8>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of this launch may still be landing when the LDS is released
        return;
    }

    // ---------------------------------------------------------------------------------------------------- compute waves
    const int fr = lane & 15, fq = lane >> 4;
    int a_row[TM], a_x[TM], a_y[TM];
    unsigned a_edge[TM];                 // bit 0: x == 0, 1: x == W-1, 2: y == 0, 3: y == H-1, 4: row past M
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wave * 32 + i * 16 + fr, m = m0 + r;
        const int x = m % W, y = (m / W) % H;
        a_row[i] = r; a_x[i] = x; a_y[i] = y;
        a_edge[i] = (x == 0 ? 1u : 0u) | (x == W - 1 ? 2u : 0u) | (y == 0 ? 4u : 0u) | (y == H - 1 ? 8u : 0u) | (m >= p.M ? 16u : 0u);
    }
    int b_off[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = j * 16 + fr;
        b_off[j] = BOFF + row * 64 + ((fq ^ xp_swz(row)) << 4);
    }
    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto a_addr = [&](int i, int abuf, int ky, int kx) -> int {       // byte address of the hi fragment (lo: + AD)
        const unsigned tmask = 16u | (kx == 0 ? 1u : 0u) | (kx == 2 ? 2u : 0u) | (ky == 0 ? 4u : 0u) | (ky == 2 ? 8u : 0u);
        int sr;
        if constexpr (UPS) sr = ((a_y[i] + ky - 1) >> 1) * Wi + ((a_x[i] + kx - 1) >> 1) - ups_rel;
        else sr = a_row[i] + ky * W + kx;
        const int ad = abuf + sr * 64 + ((fq ^ xp_swz(sr)) << 4);
        return (a_edge[i] & tmask) ? ZA : ad;
    };
    half8 ah[2][TM], al[2][TM], bh[TN], bl[TN];
    bool abl_read_done = false;                   // (ablation builds only: XP_ABL & 2 keeps the first fragment reads, drops the loop's)
    auto read_a = [&](half8 (&h)[TM], half8 (&l)[TM], int abuf, int ky, int kx) {
        if ((XP_ABL & 2) && abl_read_done) return;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ad = a_addr(i, abuf, ky, kx);
            h[i] = *(const half8*)(smem + ad);
            l[i] = *(const half8*)(smem + ad + AD);
        }
    };
    if (nsteps > 0) {
        XP_BARRIER();                                        // super-tile 0 and weight tile 0 have landed
        read_a(ah[0], al[0], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) { bh[j] = *(const half8*)(smem + b_off[j]); bl[j] = *(const half8*)(smem + b_off[j] + BPL); }
    }
    int bs_next = BBUF;                                      // ring slot (byte offset) of step t + 1
    // one step; PAR = parity of the step (which input-fragment set holds step t)
    auto one_step = [&](auto tap_tag, auto par_tag, int abuf, int abuf_n) {
        constexpr int tap = decltype(tap_tag)::value, PAR = decltype(par_tag)::value;
        XP_BARRIER();
        {   // input fragments of step t + 1: next tap of this block, or tap 0 of the next block (other buffer)
            constexpr int nky = tap == 8 ? 0 : (tap + 1) / 3, nkx = tap == 8 ? 0 : (tap + 1) % 3;
            read_a(ah[PAR ^ 1], al[PAR ^ 1], tap == 8 ? abuf_n : abuf, nky, nkx);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (!(XP_ABL & 4)) {
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[j], ah[PAR][i], acc[j][i], 0, 0, 0);
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], al[PAR][i], acc[j][i], 0, 0, 0);
                }
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[j], ah[PAR][i], acc[j][i], 0, 0, 0);
            }
            // this block's weight fragments of step t + 1 take the registers just consumed
            if (!(XP_ABL & 2)) {
                bh[j] = *(const half8*)(smem + bs_next + b_off[j]);
                bl[j] = *(const half8*)(smem + bs_next + b_off[j] + BPL);
            }
        }
        abl_read_done = true;
        bs_next = bs_next == (NSB - 1) * BBUF ? 0 : bs_next + BBUF;
    };
    for (int cbi = 0; cbi < nblk; cbi += 2) {
        {
            const int abuf = 0, abuf_n = APL;                 // even blocks live in buffer 0
            one_step(IntTag<0>{}, IntTag<0>{}, abuf, abuf_n); one_step(IntTag<1>{}, IntTag<1>{}, abuf, abuf_n);
            one_step(IntTag<2>{}, IntTag<0>{}, abuf, abuf_n); one_step(IntTag<3>{}, IntTag<1>{}, abuf, abuf_n);
            one_step(IntTag<4>{}, IntTag<0>{}, abuf, abuf_n); one_step(IntTag<5>{}, IntTag<1>{}, abuf, abuf_n);
            one_step(IntTag<6>{}, IntTag<0>{}, abuf, abuf_n); one_step(IntTag<7>{}, IntTag<1>{}, abuf, abuf_n);
            one_step(IntTag<8>{}, IntTag<0>{}, abuf, abuf_n);
        }
        if (cbi + 1 < nblk) {
            const int abuf = APL, abuf_n = 0;
            one_step(IntTag<0>{}, IntTag<1>{}, abuf, abuf_n); one_step(IntTag<1>{}, IntTag<0>{}, abuf, abuf_n);
            one_step(IntTag<2>{}, IntTag<1>{}, abuf, abuf_n); one_step(IntTag<3>{}, IntTag<0>{}, abuf, abuf_n);
            one_step(IntTag<4>{}, IntTag<1>{}, abuf, abuf_n); one_step(IntTag<5>{}, IntTag<0>{}, abuf, abuf_n);
            one_step(IntTag<6>{}, IntTag<1>{}, abuf, abuf_n); one_step(IntTag<7>{}, IntTag<0>{}, abuf, abuf_n);
            one_step(IntTag<8>{}, IntTag<1>{}, abuf, abuf_n);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the trailing fragment requests of the last step (never used)

    // ---------------- epilogue: a lane holds row m, columns n .. n + 3 of each 16 x 16 block
    const float inv = p.inv_scale;
    if (p.splits > 1) {
        float* slab = p.ws + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wave * 32 + i * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + j * 16 + 4 * fq;
                if (n < p.N) *(f32x4*)(slab + (long long)m * p.N + n) = acc[j][i] * inv;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wave * 32 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + j * 16 + 4 * fq;
            if (n < p.N) xp_store(p, acc[j][i] * inv, m, n);
        }
    }
}

// tiles 11 (plain) / 12 (nearest-2x fused); called by ief_gemm_x3p after its argument checks
int ief_conv_halo_x3p_dispatch(const IefGemmX3pParams& p, hipStream_t st) {
    if (!p.conv || p.stride != 1 || p.pad_hi_only || p.CE1 || p.CE2 || p.Wd < 2 || p.H < 2 || p.geglu) return IEF_ESHAPE;
    if (p.ups) {
        if (p.tile != 12 || p.Wd > 128 || (p.H * p.Wd) % 256 || 256 % p.Wd) return IEF_ESHAPE;
    } else if (p.tile != 11 || p.Wd > 64) return IEF_ESHAPE;
    const int ncb = (p.C1 + p.C2) / XP_BK;
    if (p.splits > ncb) return IEF_ESHAPE;
    const int tiles = ((p.M + 255) / 256) * ((p.N + 79) / 80);
    const int splits = p.splits > 1 ? p.splits : 1;
    if (p.ups) hipLaunchKernelGGL(conv3x3_halo_x3p_kernel<true>, dim3(tiles, splits), dim3(768), 0, st, p);
    else hipLaunchKernelGGL(conv3x3_halo_x3p_kernel<false>, dim3(tiles, splits), dim3(768), 0, st, p);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
