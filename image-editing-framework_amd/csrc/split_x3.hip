// Split-operand contractions ("f16x3"): fp32 storage, every product on THREE fp16 MFMAs.
//
// The reference computes in fp32 (`/root/reference/p2p/edit_syn.py:38`).  The fp32-input MFMA of exact_f32.hip reproduces
// that bit for bit up to summation order, at 1/16 of the fp16 matrix rate.  Here every fp32 operand element x is split
// once, while its tile is staged into LDS, into two fp16 numbers
//      hi = fp16(s x),   lo = fp16(s x - hi)            (s = a power of two, exact)
// so that s x = hi + lo up to 2^-22 |s x| (round-to-nearest twice), and a product of two such operands is
//      A B = (Ah Bh + Al Bh + Ah Bl) / (sa sb)   +   Al Bl / (sa sb)   <- dropped, 2^-22 relative
// three `v_mfma_f32_16x16x32_f16` into ONE fp32 accumulator (products of two fp16 numbers are exact in fp32, so the MFMA
// adds exact terms with fp32 rounding, as the fp32 MFMA does).  ~21 operand bits at 1/3 of the fp16 rate instead of 24 bits
// at 1/16; measured against fp64 the results are as close as the fp32 MFMA's (tests/test_gpu_x3.py).  The scales keep `lo`
// out of the fp16 subnormal range for elements of ordinary size (|lo| <= 2^-11 |hi|: a normal number once |s x| >= 2^-3;
// smaller elements keep an absolute resolution of 2^-25 / s — gfx950's MFMA does not flush fp16 subnormals); the host
// picks them per call (activations 2^2, weights 2^8, softmax maps 2^14), the epilogue divides them out.  |s x| must stay
// below 65504 (the fp16 range, as on the fp16-storage path): beyond it `hi` is infinite and the output NaN, not silently wrong.
//
//   igemm_x3_kernel<WM, WN, TM, TN, KIND, TRANSB>
//       the operator set and parameter struct of igemm_f32_kernel (linear / 1x1 / 3x3 implicit GEMM with concat sources,
//       nearest-2x, stride 2, fused 1x1 shortcut sources, the two batched attention products, split-K slabs).
//       Tile (16 WM TM) x (16 WN TN) x 32, WM x WN waves, each TM x TN MFMA blocks of 16 x 16.  The MFMA's A operand is the
//       WEIGHT block, its B operand the activation block: a lane then owns one output row m and four consecutive columns
//       n — one 16-byte store (and 16-byte bias / residual loads) per block.  Operand tiles are fetched TWO K tiles ahead
//       into registers (two register sets) through buffer descriptors (out-of-range lanes read zeros: no branch around a
//       load), split and written to a double-buffered LDS image one tile ahead; LDS rows of 48 halves (96 B): conflict-free
//       ds_read_b128 fragments of the 16x16x32 layout; 8-lane groups of the staging writes alternate rows r, r + 2
//       (96 B x 2 = 64 mod 128: the two rows of a 16-lane ds_write_b64 group fall into different bank halves).
//       KIND: 0 linear, 1 linear with A rows that are not 16-byte chunked (element loads), 2 3x3 convolution whose channel
//       counts are multiples of 32 (a K tile is ONE tap of ONE source: per row the tile's offset is a precomputed pixel
//       offset + a uniform term, its validity one bit of a 9-bit tap mask), 3 the same with the nearest-2x upsample folded
//       in, 4 any other convolution (per-lane tap decode).
//   attn_flash_x3_kernel<D, KS>         fused attention without materialised maps (below)
#include "ief_common.h"
#include "ief_params.h"

#include "x3_common.h"

// fp32 weights -> the two fp16 planes [2][n] the GEMM's B operand is staged from when it is a WEIGHT (static: split once per
// tensor by the host wrapper, not once per launch and workgroup): planes[0] = hi = fp16(s w), planes[1] = lo = fp16(s w - hi)
__global__ __launch_bounds__(256) void x3_split_weights_kernel(const float* __restrict__ w, half_t* __restrict__ planes, long long n4,
                                                               float s) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        half4 h, l;
        split4(((const f32x4*)w)[i], s, h, l);
        ((half4*)planes)[i] = h;
        ((half4*)planes)[n4 + i] = l;
    }
}
extern "C" int ief_x3_split_weights(const float* w, void* planes, long long n, float scale, void* stream) {
    if (!w || !planes) return IEF_EINVAL;
    if (n <= 0 || (n & 3) || !(scale > 0.f)) return IEF_ESHAPE;
    long long grid = (n / 4 + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(x3_split_weights_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, w, (half_t*)planes, n / 4, scale);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// BPRE: the B operand comes as pre-split fp16 planes (p.Wp: [2][N][K] contiguous, ief_x3_split_weights); otherwise it is
// fp32 and split while staged, like A (the batched attention products: both operands are activations).
template <int WM, int WN, int TM, int TN, int KIND, bool TRANSB, bool BPRE>
__global__ __launch_bounds__(64 * WM * WN, (2 * 2 * 16 * (WM * TM + WN * TN) * YLD * 2 > 80 * 1024) ? (WM * WN) / 4 : (WM * WN) / 2)
void igemm_x3_kernel(const IefGemmF32Params p) {
    constexpr bool CONV = KIND >= X3_CONV;
    constexpr bool HASFAST = KIND == X3_LIN || KIND == X3_CONV;      // kinds with the predicate-free main-loop loader
    constexpr int NTH = 64 * WM * WN;
    constexpr int BM = 16 * WM * TM, BN = 16 * WN * TN;
    constexpr int ROWS = BM + BN;
    constexpr int NA = (BM * 8 + NTH - 1) / NTH;            // 16-byte chunks of the A tile per thread
    constexpr int NBC = BPRE ? 2 * BN * 4 : BN * 8;          // 16-byte chunks of the B tile (BPRE: 8 halves each, two planes)
    constexpr int NB = (NBC + NTH - 1) / NTH;
    constexpr bool A_EXACT = (BM * 8) % NTH == 0, B_EXACT = NBC % NTH == 0;
    static_assert(!(BPRE && TRANSB), "pre-split planes are [N][K]");
    constexpr unsigned OOB = 0xFFFFFFF0u;        // general loader: an offset past every descriptor
    constexpr unsigned OOBF = 0x80000000u;       // fast loader: stays out of range when a K offset (< 2^31) is added
    // per buffer: [A hi][A lo][B hi][B lo], rows of YLD halves
    __shared__ __attribute__((aligned(16))) half_t smem_y[2 * 2 * ROWS * YLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid - wm * WN;
    const int lr = lane & 15, lg = lane >> 4;
    const int ntn = (p.N + BN - 1) / BN;
    const int ntiles = ((p.M + BM - 1) / BM) * ntn;
    // tile order: each XCD (its own L2) takes a contiguous run of logical ids (xcd_remap); inside a run, groups of X3_GROUP_M
    // row blocks are walked column by column, so the ~64 workgroups resident on an XCD cover a near-square patch of the
    // output: an A row block and a weight column block are each fetched into that L2 once per patch instead of the whole
    // weight once per row block (PMC: FeedForward.net[0] 4.9x its algorithmic bytes in row-major order)
    const int bid = xcd_remap(blockIdx.x, ntiles);
    const int ntm = (p.M + BM - 1) / BM;
    const int grp = bid / (X3_GROUP_M * ntn), within = bid - grp * (X3_GROUP_M * ntn);
    const int gsz = min(X3_GROUP_M, ntm - grp * X3_GROUP_M);
    const int tn = within / gsz, tm = grp * X3_GROUP_M + (within - tn * gsz);
    const int m0 = tm * BM, n0 = tn * BN;
    const float* A = p.A;
    const float* W = p.W;
    float* Out = p.Out;
    if (p.heads > 0) {                         // batched product: blockIdx.z = batch row * heads + head
        const int b = blockIdx.z / p.heads, h = blockIdx.z - b * p.heads;
        const int ba = p.a_src ? p.a_src[b] : b, bw = p.w_src ? p.w_src[b] : b;
        A += (long long)ba * p.sAb + (long long)h * p.sAh;
        W += (long long)bw * p.sWb + (long long)h * p.sWh;
        Out += (long long)b * p.sOb + (long long)h * p.sOh;
    }
    const int M = p.M, N = p.N, K = p.K;
    const float sa = p.sa, sb = p.sb;
    // ---- loader assignment.  A (and an fp32 B): chunk (row, 4 floats at k = 4 (tid & 7)); thread t takes rows
    // rsw(t >> 3) + RSTEP i, rsw swapping the two low bits so that lanes 8-15 of a 16-lane ds_write_b64 group sit two rows
    // below lanes 0-7 (96 B x 2 = 64 mod 128: different bank halves).  Pre-split B: chunk c = tid + NTH i of [plane][row][4
    // chunks of 8 halves], rows swizzled the same way for the 8-lane groups of ds_write_b128.
    const int kc4 = (tid & 7) * 4;
    const int rq = tid >> 3;
    const int r0 = (rq & ~3) | ((rq & 1) << 1) | ((rq >> 1) & 1);
    constexpr int RSTEP = NTH / 8;
    RowCoordY rc[NA];
    unsigned pixo1[NA], pixo2[NA], tmask[NA];   // KIND 2: byte offset of the centre tap's pixel in source 1 / 2; valid taps
    if (CONV) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + r0 + RSTEP * i;
            rc[i].ok = m < M && (A_EXACT || r0 + RSTEP * i < BM);
            const int mm = rc[i].ok ? m : 0;
            const int hw = p.Ho * p.Wo;
            rc[i].b = mm / hw;
            const int rem = mm - rc[i].b * hw;
            rc[i].oy = rem / p.Wo;
            rc[i].ox = rem - rc[i].oy * p.Wo;
            if (KIND == X3_CONV) {
                const int pl = p.pad_hi_only ? 0 : 1;
                const unsigned pix = (unsigned)((rc[i].b * p.H + rc[i].oy * p.stride) * p.Wd + rc[i].ox * p.stride);
                pixo1[i] = pix * (unsigned)p.C1 * 4u; pixo2[i] = pix * (unsigned)p.C2 * 4u;
                unsigned mk = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = rc[i].oy * p.stride + t / 3 - pl, ix = rc[i].ox * p.stride + t % 3 - pl;
                    mk |= (rc[i].ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd) ? (1u << t) : 0u;
                }
                tmask[i] = mk;
            }
        }
    }
    const int Ct = p.C1 + p.C2, K9 = 9 * Ct;
    const int pad_lo = p.pad_hi_only ? 0 : 1;
    const int Hs = p.ups ? (p.H >> 1) : p.H, Ws = p.ups ? (p.Wd >> 1) : p.Wd;   // dims of the stored source

    // split-K (grid.y): this workgroup's K tiles [kt0, kt0 + nk); the partial tile goes to an fp32 slab, a second launch sums
    // the slabs in slab order and applies the epilogue
    const int nk_all = (K + YBK - 1) / YBK;
    int kt0 = 0, nk = nk_all;
    if (p.splits > 1) {
        const int per = (nk_all + p.splits - 1) / p.splits;
        kt0 = blockIdx.y * per;
        nk = min(per, nk_all - kt0);
        if (nk < 0) nk = 0;
    }
    const int kend = min(K, (kt0 + nk) * YBK);       // loads past this workgroup's K range read zeros

    // Every operand is read through a buffer descriptor (wave-uniform: kernel arguments and blockIdx only) with a per-lane
    // byte offset: an element outside the tensor, the K range or the image (3x3 padding) is a lane whose offset lies past the
    // descriptor's size and reads zeros -- no branch around any load, so the loads of a tile issue back to back and stay in
    // flight under the MFMAs of two K tiles.  Two loaders: the GENERAL one predicates everything per lane (tensor edge, K
    // range, K tail, tap decode per call) and serves the first two and the last tiles; the FAST one (KIND 0 / 2, the tiles that
    // lie wholly inside the K range) adds a uniform K offset to per-thread bases computed once -- rows outside the tensor /
    // taps outside the image carry the base 2^31, out of range whatever is added (every buffer is < 2^31 bytes, else p.fast_ok
    // is 0) -- so the steady-state loop spends its vector instructions on the split, not on addresses.
    const rsrc_t rA = make_rsrc(A, p.bytesA), rW = make_rsrc(BPRE ? (const void*)p.Wp : (const void*)W, p.bytesW);
    const rsrc_t rA2 = make_rsrc(p.A2, p.bytesA2), rE1 = make_rsrc(p.E1, p.bytesE1), rE2 = make_rsrc(p.E2, p.bytesE2);
    // per-thread constant parts of the offsets
    unsigned aoff[NA], boff[NB];
    bool aval[NA], bval[NB];
    int bq8[NB];                                   // BPRE: k offset (halves) of the chunk inside the tile
    int bldsoff[NB];                               // LDS offset (halves, within a buffer's B area) of the B chunk
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int ml = r0 + RSTEP * i, m = m0 + ml;
        aval[i] = (m < M) & (A_EXACT || ml < BM);
        aoff[i] = (unsigned)((m * p.lda + kc4) * 4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        bq8[i] = 0;
        if (BPRE) {
            const int c = tid + NTH * i;
            const int plane = c >= BN * 4 ? 1 : 0, cc = c - plane * BN * 4;
            const int rr = cc >> 2, q = cc & 3;
            const int nl = (rr & ~3) | ((rr & 1) << 1) | ((rr >> 1) & 1), n = n0 + nl;
            bval[i] = (c < NBC) & (n < N);
            bq8[i] = q * 8;
            boff[i] = (unsigned)(((long long)plane * N * K + (long long)n * K + q * 8) * 2);
            bldsoff[i] = plane * BN * YLD + nl * YLD + q * 8;
        } else if (!TRANSB) {
            const int nl = r0 + RSTEP * i, n = n0 + nl;
            bval[i] = (n < N) & (B_EXACT || nl < BN);
            boff[i] = (unsigned)((n * p.ldw + kc4) * 4);
            bldsoff[i] = nl * YLD + kc4;
        } else {                        // W [K][N]: chunk c = tid + NTH i -> k row c & 31, 4 columns at 4 (c >> 5)
            const int c = tid + NTH * i;
            const int nl = (c >> 5) * 4, n = n0 + nl;
            bval[i] = (n < N) & (B_EXACT || nl < BN);                                  // N % 4 == 0 (host-checked)
            boff[i] = (unsigned)(((c & 31) * p.ldw + n) * 4);
            bldsoff[i] = nl * YLD + (c & 31);
        }
    }

    // ------------------------------------------------------------------------------------------ the general loader
    auto load_tile = [&](int k0, f32x4 (&ra)[NA], f32x4 (&rb)[NB]) {
        const int kk = k0 + kc4;
        if (KIND == X3_CONV || KIND == X3_CONV_UPS) {
            // uniform decode of the tile: a 3x3 tap (ky, kx) of source 1 / 2, or (tap 9) the 1x1 extra source 1 / 2
            int tap = 9, ch = k0 - K9;
            if (k0 < K9) { tap = k0 / Ct; ch = k0 - tap * Ct; }
            rsrc_t rs = rA;
            int cs = p.C1, chs = ch, ky = 0, kx = 0;
            const bool tapm = tap < 9;
            bool second = false;
            if (tapm) {
                ky = (tap * 11) >> 5; kx = tap - 3 * ky;
                if (ch >= p.C1) { rs = rA2; cs = p.C2; chs = ch - p.C1; second = true; }
            } else if (ch < p.CE1) { rs = rE1; cs = p.CE1; }
            else { rs = rE2; cs = p.CE2; chs = ch - p.CE1; }
            const int kin = k0 < kend ? 1 : 0;
            if (KIND == X3_CONV) {
                const int uni = tapm ? (((ky - pad_lo) * p.Wd + (kx - pad_lo)) * cs + chs) * 4 : chs * 4;
                const unsigned bit = tapm ? (1u << tap) : 0u;
                const unsigned cs4 = (unsigned)cs * 4u;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const unsigned base = tapm ? (second ? pixo2[i] : pixo1[i]) : (unsigned)(m0 + r0 + RSTEP * i) * cs4;
                    const bool ok = tapm ? (tmask[i] & bit) != 0 : rc[i].ok != 0;
                    ra[i] = bload(rs, (ok & (kin != 0)) ? base + (unsigned)uni + (unsigned)(kc4 * 4) : OOB);
                }
            } else {                       // nearest-2x upsample folded in: the tap samples the virtual 2H x 2W image
                const int dy = tapm ? ky - pad_lo : 0, dx = tapm ? kx - pad_lo : 0;
                const unsigned Hc = tapm ? p.H : p.Ho, Wc = tapm ? p.Wd : p.Wo;
                const int sh = tapm ? 1 : 0;
                const int hs = tapm ? Hs : p.Ho, ws = tapm ? Ws : p.Wo;
                const int cb = chs + kc4;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int iy = rc[i].oy + dy, ix = rc[i].ox + dx;
                    const int ok = rc[i].ok & kin & ((unsigned)iy < Hc ? 1 : 0) & ((unsigned)ix < Wc ? 1 : 0);
                    const unsigned off = (unsigned)((((rc[i].b * hs + (iy >> sh)) * ws + (ix >> sh)) * cs + cb) * 4);
                    ra[i] = bload(rs, ok ? off : OOB);
                }
            }
        } else if (KIND == X3_CONV_SLOW) {
            const float* src = nullptr;
            int cs = 0, chs = 0, ky = 0, kx = 0, mode = 0;       // mode 0: zero, 1: 3x3 tap, 2: 1x1 extra source
            if (kk < K9 && kk < kend) {
                const int tap = kk / Ct, ch = kk - tap * Ct;
                ky = tap / 3; kx = tap - 3 * ky;
                if (ch < p.C1) { src = p.A; cs = p.C1; chs = ch; } else { src = p.A2; cs = p.C2; chs = ch - p.C1; }
                mode = 1;
            } else if (kk < kend) {
                const int ch2 = kk - K9;
                if (ch2 < p.CE1) { src = p.E1; cs = p.CE1; chs = ch2; } else { src = p.E2; cs = p.CE2; chs = ch2 - p.CE1; }
                mode = 2;
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (rc[i].ok && mode == 1) {
                    int iy = rc[i].oy * p.stride + ky - pad_lo, ix = rc[i].ox * p.stride + kx - pad_lo;
                    if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd) {
                        if (p.ups) { iy >>= 1; ix >>= 1; }
                        v = *(const f32x4*)(src + (((long long)rc[i].b * Hs + iy) * Ws + ix) * cs + chs);
                    }
                } else if (rc[i].ok && mode == 2) {
                    v = *(const f32x4*)(src + (((long long)rc[i].b * p.Ho + rc[i].oy) * p.Wo + rc[i].ox) * cs + chs);
                }
                ra[i] = v;
            }
        } else if (KIND == X3_LIN) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = bload(rA, (aval[i] & (kk < kend)) ? aoff[i] + (unsigned)(k0 * 4) : OOB);
        } else {                        // rows of 77 keys: neither the row stride nor K is a multiple of 4 floats
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int m = m0 + r0 + RSTEP * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < M && kk < kend && (A_EXACT || r0 + RSTEP * i < BM)) {
                    const float* ap = A + (long long)m * p.lda + kk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (kk + j < K) v[j] = ap[j];
                }
                ra[i] = v;
            }
        }
        if (BPRE) {
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = bload(rW, (bval[i] & (k0 + bq8[i] < kend)) ? boff[i] + (unsigned)(k0 * 2) : OOB);
        } else if (!TRANSB) {
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = bload(rW, (bval[i] & (kk < kend)) ? boff[i] + (unsigned)(k0 * 4) : OOB);
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int kr = k0 + ((tid + NTH * i) & 31);
                rb[i] = bload(rW, (bval[i] & (kr < kend)) ? boff[i] + (unsigned)(k0 * p.ldw * 4) : OOB);
            }
        }
    };

    // ------------------------------------------------------------------------------------------ the fast loader
    // state of the convolution's tile stream (tiles are requested in order): current segment = one tap of one source (or a
    // 1x1 extra source): its descriptor, the rows' bases (2^31 where the tap falls outside the image), the running byte
    // offset inside the segment, tiles left in it
    unsigned afast[NA], bfast[NB];
    rsrc_t f_rs = rA;
    int f_tap = 0, f_ch = 0, f_left = 0, f_uni = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) afast[i] = aval[i] ? aoff[i] : OOBF;
#pragma unroll
    for (int i = 0; i < NB; ++i) bfast[i] = bval[i] ? boff[i] : OOBF;
    auto seg_setup = [&]() {            // (f_tap, f_ch) -> f_rs, afast[], f_uni, f_left; uniform, runs once per segment
        int cs, chs, src_c;
        const bool tapm = f_tap < 9;
        bool second = false;
        if (tapm) {
            if (f_ch >= p.C1) { f_rs = rA2; cs = p.C2; chs = f_ch - p.C1; second = true; src_c = p.C2; }
            else { f_rs = rA; cs = p.C1; chs = f_ch; src_c = p.C1; }
        } else if (f_ch < p.CE1) { f_rs = rE1; cs = p.CE1; chs = f_ch; src_c = p.CE1; }
        else { f_rs = rE2; cs = p.CE2; chs = f_ch - p.CE1; src_c = p.CE2; }
        f_left = (src_c - chs) / YBK;
        f_uni = chs * 4;
        const int ky = (f_tap * 11) >> 5, kx = f_tap - 3 * ky;
        const int tapoff = tapm ? ((ky - pad_lo) * p.Wd + (kx - pad_lo)) * cs * 4 : 0;
        const unsigned bit = tapm ? (1u << f_tap) : 0u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned base = tapm ? (second ? pixo2[i] : pixo1[i]) + (unsigned)tapoff
                                       : (unsigned)(m0 + r0 + RSTEP * i) * (unsigned)cs * 4u;
            const bool ok = tapm ? (tmask[i] & bit) != 0 : rc[i].ok != 0;
            afast[i] = ok ? base + (unsigned)(kc4 * 4) : OOBF;
        }
    };
    auto load_fast = [&](int k0, f32x4 (&ra)[NA], f32x4 (&rb)[NB]) {
        if (KIND == X3_CONV) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = bload(f_rs, afast[i] + (unsigned)f_uni);
            f_uni += YBK * 4;
            f_ch += YBK;
            if (--f_left == 0) {                   // next segment: the other source, the next tap, or the 1x1 sources
                if (f_tap < 9 && f_ch >= Ct) { f_ch = 0; ++f_tap; }
                seg_setup();
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = bload(rA, afast[i] + (unsigned)(k0 * 4));
        }
        const unsigned kb = BPRE ? (unsigned)(k0 * 2) : TRANSB ? (unsigned)(k0 * p.ldw * 4) : (unsigned)(k0 * 4);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = bload(rW, bfast[i] + kb);
    };

    // split + LDS write of ONE staged chunk (idx < NA: A chunk, else B chunk idx - NA) of the tile held in (ra, rb)
    auto store_chunk = [&](int buf, int idx, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB]) {
        half_t* ah = smem_y + buf * (2 * ROWS * YLD);
        half_t* al = ah + BM * YLD;
        half_t* bh = al + BM * YLD;
        half_t* bl = bh + BN * YLD;
        if (idx < NA) {
            const int i = idx;
            if (!A_EXACT && r0 + RSTEP * i >= BM) return;
            half4 h, l;
            split4(ra[i], sa, h, l);
            const int off = (r0 + RSTEP * i) * YLD + kc4;
            *(half4*)(ah + off) = h;
            *(half4*)(al + off) = l;
        } else if (BPRE) {
            const int i = idx - NA;
            if (!B_EXACT && tid + NTH * i >= NBC) return;
            *(f32x4*)(bh + bldsoff[i]) = rb[i];                  // 8 halves of one plane, already split
        } else if (!TRANSB) {
            const int i = idx - NA;
            if (!B_EXACT && r0 + RSTEP * i >= BN) return;
            half4 h, l;
            split4(rb[i], sb, h, l);
            *(half4*)(bh + bldsoff[i]) = h;
            *(half4*)(bl + bldsoff[i]) = l;
        } else {
            const int i = idx - NA;
            if (!B_EXACT && ((tid + NTH * i) >> 5) * 4 >= BN) return;
            half4 h, l;
            split4(rb[i], sb, h, l);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bh[bldsoff[i] + j * YLD] = h[j];
                bl[bldsoff[i] + j * YLD] = l[j];
            }
        }
    };
    auto store_tile = [&](int buf, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB]) {
#pragma unroll
        for (int c = 0; c < NA + NB; ++c) store_chunk(buf, c, ra, rb);
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int a = 0; a < TM; ++a) acc[b][a] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the MFMAs of the tile in LDS buffer `buf`, then the split + LDS write of the NEXT tile (registers ra, rb -> buffer buf ^ 1)
    auto mma_tile = [&](int buf, const f32x4 (&ra)[NA], const f32x4 (&rb)[NB]) {
        const half_t* ah = smem_y + buf * (2 * ROWS * YLD) + (wm * TM * 16 + lr) * YLD + 8 * lg;
        const half_t* al = ah + BM * YLD;
        const half_t* bh = smem_y + buf * (2 * ROWS * YLD) + 2 * BM * YLD + (wn * TN * 16 + lr) * YLD + 8 * lg;
        const half_t* bl = bh + BN * YLD;
        half8_t fah[TM], fal[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            fah[a] = *(const half8_t*)(ah + a * 16 * YLD);
            fal[a] = *(const half8_t*)(al + a * 16 * YLD);
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const half8_t fbh = *(const half8_t*)(bh + b * 16 * YLD);
            const half8_t fbl = *(const half8_t*)(bl + b * 16 * YLD);
#pragma unroll
            for (int a = 0; a < TM; ++a) {
                acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbl, fah[a], acc[b][a], 0, 0, 0);
                acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh, fal[a], acc[b][a], 0, 0, 0);
                acc[b][a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fbh, fah[a], acc[b][a], 0, 0, 0);
            }
        }
        store_tile(buf ^ 1, ra, rb);
    };

    // register set s holds tile t with (t & 1) == s; LDS buffer (t & 1) holds tile t.  Tiles past the K range load zeros and
    // multiply zeros (an odd tile count costs one idle pass).
    f32x4 ra0[NA], rb0[NB], ra1[NA], rb1[NB];
    load_tile(kt0 * YBK, ra0, rb0);
    load_tile((kt0 + 1) * YBK, ra1, rb1);
    store_tile(0, ra0, rb0);
    __syncthreads();
    // NOTE the order inside a half step: the loads of tile kt + 2 overwrite register set (kt & 1), whose previous content
    // (tile kt) went to LDS during the previous half step; the tile converted now is kt + 1 from the OTHER set
    int kt = 0;
    if (HASFAST && p.fast_ok) {
        const int nfast = kend / YBK - kt0;                  // tiles [0, nfast) of this workgroup lie wholly inside its K range
        if (KIND == X3_CONV && nfast > 3) {                  // the stream of the fast loader starts at tile 2
            const int k0 = (kt0 + 2) * YBK;
            if (k0 < K9) { f_tap = k0 / Ct; f_ch = k0 - f_tap * Ct; } else { f_tap = 9; f_ch = k0 - K9; }
            seg_setup();
        }
        for (; kt + 3 < nfast; kt += 2) {
            load_fast((kt0 + kt + 2) * YBK, ra0, rb0);
            mma_tile(0, ra1, rb1);
            __syncthreads();
            load_fast((kt0 + kt + 3) * YBK, ra1, rb1);
            mma_tile(1, ra0, rb0);
            __syncthreads();
        }
    }
    for (; kt < nk; kt += 2) {
        load_tile((kt0 + kt + 2) * YBK, ra0, rb0);
        mma_tile(0, ra1, rb1);
        __syncthreads();
        load_tile((kt0 + kt + 3) * YBK, ra1, rb1);
        mma_tile(1, ra0, rb0);
        __syncthreads();
    }
    const float inv = 1.0f / (sa * sb);
    // lane: output row m (lane & 15 of the block), columns n .. n + 3 (4 (lane >> 4) of the block)
    const bool vec = p.vec_out != 0;
    if (p.splits > 1) {       // raw partial sums (already in output units); ws rows are N floats
        float* slab = p.ws + (long long)blockIdx.y * M * N;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int m = m0 + (wm * TM + a) * 16 + lr;
            if (m >= M) continue;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int n = n0 + (wn * TN + b) * 16 + 4 * lg;
                if (n >= N) continue;
                const f32x4 v = acc[b][a] * inv;
                if ((N & 3) == 0) *(f32x4*)(slab + (long long)m * N + n) = v;
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (n + j < N) slab[(long long)m * N + n + j] = v[j];
                }
            }
        }
        return;
    }
    if (p.geglu) {
        // FeedForward.net[0] with its GEGLU fused (weight rows interleaved [8 hidden | 8 gate] per 16-column block, unet.py):
        // lanes 0-31 of a block hold hidden columns 0-7 (4 per lane), lanes 32-63 the matching gate columns -- one exchange
        // with lane + 32, then out[m][8 block + 4 lg .. + 3] = hidden * gelu(gate): the 2x wider pre-activation never reaches HBM
        // (host-checked: N % 16 == 0, bias only, 16-byte aligned output rows)
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int m = m0 + (wm * TM + a) * 16 + lr;
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                const int n = n0 + (wn * TN + b) * 16 + 4 * lg;
                f32x4 v = acc[b][a] * inv;
                if (p.bias && n < N) v += *(const f32x4*)(p.bias + n);
                f32x4 g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = __shfl_xor(v[j], 32);
                if (lg < 2 && m < M && n < N) {
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = v[j] * gelu_f(g[j]) * p.out_scale;
                    *(f32x4*)(Out + (long long)m * p.ldo + (n0 + (wn * TN + b) * 16) / 2 + 4 * lg) = o;
                }
            }
        }
        return;
    }
    // ---- epilogue: (acc + bias[n] + rowvec[m / rows_per_batch][n] + residual[m][n]) * out_scale, fp32
    const float* R = p.residual;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int m = m0 + (wm * TM + a) * 16 + lr;
        if (m >= M) continue;
        const float* rv = p.rowvec ? p.rowvec + (long long)(m / p.rows_per_batch) * N : nullptr;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n = n0 + (wn * TN + b) * 16 + 4 * lg;
            if (n >= N) continue;
            f32x4 v = acc[b][a] * inv;
            if (vec) {
                if (p.bias) v += *(const f32x4*)(p.bias + n);
                if (rv) v += *(const f32x4*)(rv + n);
                if (R) v += *(const f32x4*)(R + (long long)m * p.ldr + n);
                *(f32x4*)(Out + (long long)m * p.ldo + n) = v * p.out_scale;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (n + j >= N) break;
                    float s = v[j];
                    if (p.bias) s += p.bias[n + j];
                    if (rv) s += rv[n + j];
                    if (R) s += R[(long long)m * p.ldr + n + j];
                    Out[(long long)m * p.ldo + n + j] = s * p.out_scale;
                }
            }
        }
    }
}

// column-tile width for N output columns: 80 where it divides N (every SD width is a multiple of 320), else 64
// 128 x 160 tiles on 8 waves (one workgroup per CU) for the launches they cover; 0 switches them off (A/B runs: IEF_X3_WIDE=0)
static int g_x3_wide = 1, g_flash_ks2 = 1;
extern "C" void ief_gemm_x3_set_variant(int v) { g_x3_wide = v & 1; g_flash_ks2 = (v & 2) ? 0 : 1; }     // bit 1: 32-key flash tiles
// which launches take the wide tile.  Measured on the SD1.5 batch-4 shapes (gpurun, HIP events per launch): every 3x3
// convolution gains 8-12 % (fewer L1 fills and LDS stores per MFMA: an activation row block is staged for 160 columns
// instead of 80), linears gain where N <= 1280 or K >= 1280 and lose 5-30 % on the wide, shallow ones (FeedForward.net[0]
// at the 64x64 level: 32 column tiles of a 10-tile K loop, where two independent workgroups per CU hide each other's
// prologue and epilogue)
static bool x3_wide_ok(int conv, int N, int K) { return g_x3_wide && N % 160 == 0 && (conv || N <= 1280 || K >= 1280); }
// column-tile width the host's split-K policy counts tiles with
extern "C" int ief_gemm_x3_bn_k(int conv, int N, int K) { return x3_wide_ok(conv, N, K) ? 160 : (N % 80 == 0) ? 80 : 64; }
extern "C" int ief_gemm_x3_bn(int N) { return (N % 80 == 0) ? 80 : 64; }
extern "C" int ief_gemm_x3_bm(int M, int N) {
    (void)M; (void)N;
    return 128;       // a 64 x 80 tile (4 waves of 16 rows) without split-K measured slower than 128 x 80 with it: not instantiated
}

template <int WM, int WN, int TM, int TN, int KIND, bool TRANSB, bool BPRE>
static int launch_igemm_x3(const IefGemmF32Params& p, hipStream_t st) {
    constexpr int BM = 16 * WM * TM, BN = 16 * WN * TN;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int z = p.heads > 0 ? p.batch * p.heads : 1;
    const int splits = p.splits > 1 ? p.splits : 1;
    hipLaunchKernelGGL((igemm_x3_kernel<WM, WN, TM, TN, KIND, TRANSB, BPRE>), dim3(tiles, splits, z), dim3(64 * WM * WN), 0, st, p);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// called by ief_gemm_f32 (exact_f32.hip) after its argument checks when p.x3 != 0; the split-K reducer launch is the caller's
int ief_gemm_x3_dispatch(const IefGemmF32Params& pin, hipStream_t st) {
    IefGemmF32Params p = pin;
    // 16-byte epilogue accesses need 16-byte aligned rows of every operand the epilogue touches
    const bool al = ((p.N & 3) == 0) && ((p.ldo & 3) == 0) && (((uintptr_t)p.Out & 15) == 0) &&
                    (!p.bias || ((uintptr_t)p.bias & 15) == 0) && (!p.rowvec || ((uintptr_t)p.rowvec & 15) == 0) &&
                    (!p.residual || (((uintptr_t)p.residual & 15) == 0 && (p.ldr & 3) == 0)) &&
                    (p.heads == 0 || (((p.sOb | p.sOh) & 3) == 0));
    p.vec_out = al ? 1 : 0;
    // pre-split weight planes: [2][N][K] fp16 contiguous, 16-byte chunks of 8 halves
    if (p.geglu && (p.conv || p.transb || p.heads > 0 || p.splits > 1 || (p.N & 15) || p.residual || p.rowvec || !al ||
                    (((uintptr_t)p.Out | (uintptr_t)(p.bias ? p.bias : p.Out)) & 15) || (p.ldo & 3)))
        return IEF_EINVAL;
    const bool bpre = p.Wp != nullptr;
    if (bpre && (p.transb || p.heads > 0 || (p.K & 7) || ((uintptr_t)p.Wp & 15) || (!p.conv && p.ldw != p.K))) return IEF_EINVAL;
    // descriptor sizes (bytes reachable from each operand's base; batched: from the (batch row, head) base)
    const unsigned long long lim = 0xFFFFFFF0ull;
    unsigned long long bA, bW, bA2 = 0, bE1 = 0, bE2 = 0;
    if (p.conv) {
        const unsigned long long px = (unsigned long long)p.batch_images * (p.ups ? (p.H >> 1) * (p.Wd >> 1) : p.H * p.Wd);
        bA = px * p.C1 * 4; bA2 = px * p.C2 * 4;
        bE1 = (unsigned long long)p.M * p.CE1 * 4; bE2 = (unsigned long long)p.M * p.CE2 * 4;
        p.al32 = ((p.C1 | p.C2 | p.CE1 | p.CE2) & 31) == 0 ? 1 : 0;
    } else {
        bA = ((unsigned long long)(p.M - 1) * p.lda + p.K) * 4;
        p.al32 = 0;
    }
    bW = bpre ? 4ull * p.N * p.K
              : p.transb ? ((unsigned long long)(p.K - 1) * p.ldw + p.N) * 4 : ((unsigned long long)(p.N - 1) * p.ldw + p.K) * 4;
    if (bA >= lim || bW >= lim || bA2 >= lim || bE1 >= lim || bE2 >= lim) return IEF_ESHAPE;      // 32-bit buffer offsets
    p.bytesA = (unsigned)bA; p.bytesW = (unsigned)bW; p.bytesA2 = (unsigned)bA2; p.bytesE1 = (unsigned)bE1; p.bytesE2 = (unsigned)bE2;
    // the fast loader marks out-of-range rows with the base 2^31 and adds K offsets below 2^30 to it
    const unsigned long long half = 0x40000000ull;
    const unsigned long long kspan = p.transb ? 4ull * p.K * p.ldw : 4ull * p.K;
    p.fast_ok = (bA < half && bW < half && bA2 < half && bE1 < half && bE2 < half && kspan < half) ? 1 : 0;
    if (x3_wide_ok(p.conv, p.N, p.K) && bpre && !p.a_scalar && (!p.conv || (p.al32 && !p.ups)))
        return p.conv ? launch_igemm_x3<4, 2, 2, 5, X3_CONV, false, true>(p, st) : launch_igemm_x3<4, 2, 2, 5, X3_LIN, false, true>(p, st);
    const bool n80 = p.N % 80 == 0;
#define X3_GO(KIND_, TRANSB_, BPRE_) (n80 ? launch_igemm_x3<4, 1, 2, 5, KIND_, TRANSB_, BPRE_>(p, st) \
                                          : launch_igemm_x3<2, 2, 4, 2, KIND_, TRANSB_, BPRE_>(p, st))
    if (p.conv) {
        if (!p.al32) return bpre ? X3_GO(X3_CONV_SLOW, false, true) : X3_GO(X3_CONV_SLOW, false, false);
        if (p.ups) return bpre ? X3_GO(X3_CONV_UPS, false, true) : X3_GO(X3_CONV_UPS, false, false);
        return bpre ? X3_GO(X3_CONV, false, true) : X3_GO(X3_CONV, false, false);
    }
    if (p.transb) return p.a_scalar ? X3_GO(X3_LIN_SLOW, true, false) : X3_GO(X3_LIN, true, false);
    if (p.a_scalar) return bpre ? X3_GO(X3_LIN_SLOW, false, true) : X3_GO(X3_LIN_SLOW, false, false);
    return bpre ? X3_GO(X3_LIN, false, true) : X3_GO(X3_LIN, false, false);
#undef X3_GO
}

// --------------------------------------------------------------------------------------------- fused attention, split operands
// out[b] = softmax(scale * q[q_src[b]] k[k_src[b]]^T) v[v_src[b]] without materialising the maps, both products on split
// operands (same role as attn_flash_f32_kernel, same parameter struct).  One workgroup = 128 queries of one (batch row, head),
// a wave = 32 of them; 32-key tiles.
//   S^T = K Q^T on v_mfma_f32_32x32x16_f16 (A = K tile hi / lo from LDS, B = this lane's query hi / lo in registers for the
//   whole kernel): a lane owns one query COLUMN, 16 keys of the tile in its registers, the other 16 in lane + 32 -- running
//   maximum / sum in-lane plus one cross-half exchange, in fp32 (exp2 of log2e-scaled scores).
//   O^T += V^T P^T: the P registers of k-step s (registers 8s .. 8s+7, split into hi / lo with scale 2^14: P <= 1) are
//   directly the B operand; its k order is key 16 s + 8 (j >> 2) + 4 h + (j & 3) for element j of lane half h, so the V^T
//   fragment (A operand, row = head-dim index) is two 8-byte reads of a V tile kept TRANSPOSED in LDS ([d][key], hi and lo).
// K and V tiles are fetched one tile ahead into registers through buffer descriptors (keys past L read zeros and are masked
// to -inf), split once per workgroup and written to LDS (K rows of 16 DG + 8 halves, V^T rows of 36 halves: conflict-free
// fragment reads).  Head dims are padded to 16 for the scores (d = 40: 48) and to 32 for O^T (d = 40: 64).
// KS: 32-key sub-tiles staged and consumed per barrier pair (2: one online-softmax update, one rescale and two barriers per
// 64 keys instead of per 32)
template <int D, int KS>
__global__ __launch_bounds__(256, (D > 80) ? 1 : 2) void attn_flash_x3_kernel(const IefAttnF32Params p) {
    constexpr int DG = (D + 15) / 16;            // 16-deep groups of the score product
    constexpr int DT = (D + 31) / 32;            // 32-row tiles of O^T
    constexpr int KLD = DG * 16 + 8;             // halves per K row (bytes = 16 mod 32: conflict-free b128 fragments)
    constexpr int KT = 32 * KS;                  // keys per iteration
    // V stays ROW-major in LDS ([key][VRS] per plane, 8-byte stores of four head-dim values): the V^T fragments of O^T += V^T P^T
    // come from TRANSPOSING reads (ds_read_b64_tr_b16, as the fp16 kernel csrc/attention.hip) -- the former [d][key] image was
    // written by 16-bit scatter stores whose bank conflicts were 29 % of the LDS-active cycles (profiles/r03_pmc_sq_x3.txt).
    // Row stride 64 or 192 (mod 256) bytes: the four rows a 32-lane half touches fall into distinct 64-byte bank slots.
    constexpr int VRS = D <= 32 ? 32 : (D <= 96 ? 96 : 160);
    static_assert(VRS >= 32 * DT, "V row covers the padded head dim");
    constexpr float SQ = 1.f, SK = 1.f, SV = 1.f, SP = 16384.f;     // activations: scale 1 (the fp16 range itself); maps <= 1: 2^14
    __shared__ __attribute__((aligned(16))) half_t smem_f[2 * KT * KLD + 2 * KT * VRS];
    half_t* Kh = smem_f;
    half_t* Kl = Kh + KT * KLD;
    half_t* Vh = Kl + KT * KLD;                  // [KT][VRS]
    half_t* Vl = Vh + KT * VRS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // (head, query block) from the XCD-contiguous logical id: the query blocks of one head run on ONE XCD, whose L2 then
    // holds that head's K / V once (plain (x, y) order spreads them over all eight: PMC 4.5x the algorithmic bytes)
    const int nqb = gridDim.x;
    const int lid = xcd_remap(blockIdx.x + nqb * blockIdx.y, nqb * gridDim.y);
    const int bh = lid / nqb, qb = lid - bh * nqb, b = bh / p.heads, h = bh - b * p.heads;
    const int bq = p.q_src ? p.q_src[b] : b, bk = p.k_src ? p.k_src[b] : b, bv = p.v_src ? p.v_src[b] : b;
    const float* Q = p.Q + (long long)bq * p.sQb + (long long)h * D;
    const float* Kp = p.K + (long long)bk * p.sKb + (long long)h * D;
    const float* Vp = p.V + (long long)bv * p.sVb + (long long)h * D;
    float* O = p.Out ? p.Out + (long long)b * p.sOb + (long long)h * D : nullptr;
    half_t* OP = p.OutP ? p.OutP + (long long)b * p.sOPb + (long long)h * D : nullptr;
    const int q0 = qb * 128 + wid * 32;
    const int qi = q0 + li;
    // zero the LDS once: padding columns of K (d >= D) and of V (d >= D, up to VRS) stay zero (staging never writes them)
    for (int c = tid; c < (int)(sizeof(smem_f) / 16); c += 256) ((f32x4*)smem_f)[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this lane's query, split: for every 16-deep d group the 8 values d = 16 g + 8 lh + j
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
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = p.scale * 1.44269504088896341f / (SQ * SK);      // scores in log2 units
    constexpr int KCH = KT * (D / 4);            // 16-byte chunks of one K (or V) tile
    constexpr int NLD = (KCH + 255) / 256;
    const rsrc_t rK = make_rsrc(Kp, (unsigned)(((long long)(p.L - 1) * p.ldk + D) * 4));
    const rsrc_t rV = make_rsrc(Vp, (unsigned)(((long long)(p.L - 1) * p.ldv + D) * 4));
    f32x4 rk[NLD], rv[NLD];
    auto load_kv = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 256 * i;
            const int row = c / (D / 4), ch = c - row * (D / 4);
            const bool ok = (c < KCH) & (k0 + row < p.L);
            rk[i] = bload(rK, ok ? (unsigned)(((k0 + row) * p.ldk + ch * 4) * 4) : 0xFFFFFFF0u);
            rv[i] = bload(rV, ok ? (unsigned)(((k0 + row) * p.ldv + ch * 4) * 4) : 0xFFFFFFF0u);
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 256 * i;
            if (c < KCH) {
                const int row = c / (D / 4), ch = c - row * (D / 4);
                half4 hh, ll;
                split4(rk[i], SK, hh, ll);
                *(half4*)(Kh + row * KLD + ch * 4) = hh;
                *(half4*)(Kl + row * KLD + ch * 4) = ll;
                split4(rv[i], SV, hh, ll);
                *(half4*)(Vh + row * VRS + ch * 4) = hh;
                *(half4*)(Vl + row * VRS + ch * 4) = ll;
            }
        }
    };
    const int L16 = lane & 15;
    const int v_lane = (4 * lh + (L16 >> 2)) * VRS + 16 * ((lane >> 4) & 1) + 4 * (L16 & 3);      // transposing V reads
    const int nt = (p.L + KT - 1) / KT;
    load_kv(0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();                          // everybody is done with the previous tile (first pass: the zero fill)
        store_kv();
        __syncthreads();
        load_kv((t + 1) * KT);                    // past L: zeros, never stored
        // ---- S^T tiles: KS x (32 keys x 32 queries)
        f32x16 sacc[KS];
#pragma unroll
        for (int u = 0; u < KS; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[u][r] = 0.f;
#pragma unroll
            for (int g = 0; g < DG; ++g) {
                const half8_t kh = *(const half8_t*)(Kh + (u * 32 + li) * KLD + g * 16 + 8 * lh);
                const half8_t kl = *(const half8_t*)(Kl + (u * 32 + li) * KLD + g * 16 + 8 * lh);
                sacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[g], sacc[u], 0, 0, 0);
                sacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[g], sacc[u], 0, 0, 0);
                sacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[g], sacc[u], 0, 0, 0);
            }
        }
        // ---- online softmax over the key rows of this tile (register r <-> key (r&3) + 8 (r>>2) + 4 lh).  The loop is bound by
        // vector issue, not by the 21 MFMAs (289 vector instructions per tile before the three trims below, ~150 after): keys
        // are masked only in the tile that crosses L, exp2 is the bare v_exp_f32 (arguments <= 0: the library form's
        // denormal-range rescue costs five instructions per call and guards results that round to 0 here anyway), and the
        // accumulators are rescaled only when some lane's running maximum moved (alpha == 1 exactly otherwise)
        // (the scale rides in the exponent's FMA: exp2(s sc2 - m) is one v_fma + one v_exp per score; the maximum is taken on the raw
        // scores -- sc2 > 0 -- by v_max3 over the two sub-tiles)
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            if (t * KT + (u + 1) * 32 > p.L) {    // wave-uniform: only the tile that crosses L is masked
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KT + u * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= p.L) sacc[u][r] = -INFINITY;
                }
            }
        }
        float mxa = -INFINITY, mxb = -INFINITY;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KS == 2) {
                mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, sacc[0][r]), sacc[KS - 1][r]);
                mxb = __builtin_fmaxf(__builtin_fmaxf(mxb, sacc[0][8 + r]), sacc[KS - 1][8 + r]);
            } else {
                mxa = __builtin_fmaxf(mxa, sacc[0][r]);
                mxb = __builtin_fmaxf(mxb, sacc[0][8 + r]);
            }
        }
        float mx = fmaxf(mxa, mxb) * sc2;
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // exp2(-inf) = 0 on the first tile
        float ls = 0.f;
#pragma unroll
        for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[u][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[u][r], sc2, -m_new)); ls += sacc[u][r]; }
        l_run = l_run * alpha + ls;
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
#pragma unroll
            for (int tt = 0; tt < DT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[tt][r] *= alpha;
        }
        m_run = m_new;
        // ---- O^T += V^T P^T, two 16-key steps per sub-tile; P split with scale 2^14
#pragma unroll
        for (int u = 0; u < KS; ++u) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8_t ph, pl;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const f32x4 v = {sacc[u][8 * s + 4 * c2], sacc[u][8 * s + 4 * c2 + 1], sacc[u][8 * s + 4 * c2 + 2], sacc[u][8 * s + 4 * c2 + 3]};
                    half4 hh, ll;
                    split4(v, SP, hh, ll);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ph[4 * c2 + j] = hh[j]; pl[4 * c2 + j] = ll[j]; }
                }
#pragma unroll
                for (int tt = 0; tt < DT; ++tt) {
                    // fragment = head-dim rows 32 tt .., keys {4 lh .. + 3} and {8 + 4 lh .. + 3} of this 16-key step
                    const int vo = v_lane + (u * 32 + 16 * s) * VRS + 32 * tt;
                    const half4 a0 = x3_lds_tr_read(Vh + vo), a1 = x3_lds_tr_read(Vh + vo + 8 * VRS);
                    const half4 b0 = x3_lds_tr_read(Vl + vo), b1 = x3_lds_tr_read(Vl + vo + 8 * VRS);
                    half8_t vh, vl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { vh[j] = a0[j]; vh[4 + j] = a1[j]; vl[j] = b0[j]; vl[4 + j] = b1[j]; }
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[tt], 0, 0, 0);
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[tt], 0, 0, 0);
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[tt], 0, 0, 0);
                }
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / (l_tot * SV * SP);
    if (p.lse && qi < p.N && lh == 0) p.lse[((long long)b * p.heads + h) * p.N + qi] = m_run + __log2f(l_tot);     // for ief_attn_bwd_x3
    if (qi < p.N) {
#pragma unroll
        for (int tt = 0; tt < DT; ++tt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = tt * 32 + 8 * g + 4 * lh;       // registers 4g .. 4g+3 <-> d .. d+3
                if (d < D) {
                    const f32x4 v = {o[tt][4 * g] * inv, o[tt][4 * g + 1] * inv, o[tt][4 * g + 2] * inv, o[tt][4 * g + 3] * inv};
                    if (O) *(f32x4*)(O + (long long)qi * p.ldo + d) = v;
                    if (OP) {                 // operand planes for to_out's GEMM (csrc/gemm_x3p.hip)
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


#ifndef X3P_FLASH_BATCH_READS
#define X3P_FLASH_BATCH_READS 1
#endif
// --------------------------------------------------------------------------------------------- fused attention on OPERAND PLANES
// The same product as attn_flash_x3_kernel with Q / K / V arriving as pre-split hi / lo fp16 planes (written by the q|k|v GEMM's
// epilogue, csrc/gemm_x3p.hip).  What that buys, measured on the fp32-input kernel at N = 4096, d = 40 (profiles/r04_pmc_attn40x3.txt):
// every workgroup re-split the whole K / V of its head (20 % of the vector instructions) and kept the prefetched tiles in 24
// registers, which held the kernel at 162 VGPRs = 3 workgroups per CU = 768 slots for 1024 workgroups: a second round one third
// full.  Here K / V tiles are staged by LDS-DMA (global_load_lds, wave w stages plane w of {K hi, K lo, V hi, V lo}: no staging
// registers, no split, no LDS stores), the kernel fits 128 VGPRs and 40 KiB of LDS: four workgroups per CU, one round.
//   LDS: two buffers of [K hi | K lo | V hi | V lo], each KT rows of D halves, UNPADDED (an LDS-DMA piece is lane-linear); D = 64:
//   16-byte chunk c of row r lives in slot c ^ (r & 7) (128-byte rows would otherwise put every row on the same banks); D = 40
//   (80-byte rows) and D = 80 need none.  Nothing is read past a row: the score product's padding lanes (d >= D) and the
//   transposing V reads of head-dim columns >= D take their address from column 0 -- their operand is multiplied by the zero
//   padding of Q, respectively lands in output rows >= D that are never stored.
//   One barrier per tile: [own DMA of tile t landed] barrier [issue tile t + 1 into the other buffer] scores, softmax, P V.
//   NWV waves per workgroup (32 queries each).  d = 40: 8 waves / 256 queries: two workgroups (80 KiB of LDS, 16 waves) per CU --
//   four 40-KiB workgroups of 4 waves are exactly the CU's 160 KiB and were NOT co-resident (measured: still a second round);
//   a tile is then also staged once for 256 queries instead of 128.
//   Tried and dropped (round 4): an in-wave software pipeline -- 32-key sub-steps, O^T += V^T(s-1) P^T(s-1) chunk by chunk beside
//   the exp / split of sub-step s, transposing reads by inline asm with counted lgkmcnt, two P sets swapping roles.  Inside the 128
//   registers of four waves per SIMD it spills (O^T 32 + Q 24 + S 16 + two P sets 32 + fragments 16); at two waves per SIMD
//   (177 registers, one 8-wave workgroup per CU) it was parity-green and SLOWER: N = 4096 425 us vs 360, N = 16384 5.72 ms vs 5.11
//   (SQ counters: 30 % more vector instructions per key from the operand-tuple moves the register allocator inserts, and half
//   the waves to hide the rest behind).  What did pay on this loop: the lazy maximum, the split scale in the exponent and raised
//   priority in the matrix phases (360 -> 347 us, 5.11 -> 5.01 ms).
template <int D, int KS, int NWV>
__global__ __launch_bounds__(64 * NWV, D <= 40 ? 4 : 2) void attn_flash_x3p_kernel(const IefAttnF32Params p) {
    constexpr int DG = (D + 15) / 16, DT = (D + 31) / 32;
    constexpr int KT = 32 * KS;
    constexpr int CPR = D / 8;                   // 16-byte chunks per row
    constexpr int PL = KT * D;                   // halves per plane of a tile
    constexpr int NP = PL * 2 / 1024;            // LDS-DMA pieces per plane
    static_assert((PL * 2) % 1024 == 0, "a plane of a tile is a whole number of LDS-DMA pieces");
    constexpr bool SWZ = D == 64;
    constexpr float LOGSP = 10.f, THR = 5.f;
    constexpr bool BATCH_READS = X3P_FLASH_BATCH_READS != 0;
    __shared__ __attribute__((aligned(1024))) half_t smem_p[2 * 4 * PL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int nqb = gridDim.x;
    const int lid = xcd_remap(blockIdx.x + nqb * blockIdx.y, nqb * gridDim.y);
    const int bh = lid / nqb, qb = lid - bh * nqb, b = bh / p.heads, h = bh - b * p.heads;
    const int bq = p.q_src ? p.q_src[b] : b, bk = p.k_src ? p.k_src[b] : b, bv = p.v_src ? p.v_src[b] : b;
    const half_t* Qh = p.Qp + (long long)bq * p.sQb + (long long)h * D;
    // the four planes of a tile: 0 K hi, 1 K lo, 2 V hi, 3 V lo
    const char* srcs[4];
#pragma unroll
    for (int pl = 0; pl < 4; ++pl)
        srcs[pl] = (const char*)(pl < 2 ? p.Kp + (long long)bk * p.sKb + (pl & 1) * p.planeK
                                        : p.Vp + (long long)bv * p.sVb + (pl & 1) * p.planeV) + (long long)h * D * 2;
    const char* zp = (const char*)p.zeros;
    float* O = p.Out ? p.Out + (long long)b * p.sOb + (long long)h * D : nullptr;
    half_t* OP = p.OutP ? p.OutP + (long long)b * p.sOPb + (long long)h * D : nullptr;
    const int q0 = qb * (32 * NWV) + wid * 32;
    const int qi = q0 + li;
    // the lane's query: for every 16-deep d group the 8 values d = 16 g + 8 lh + j, hi and lo straight from the planes
    half8_t qh[DG], ql[DG];
#pragma unroll
    for (int g = 0; g < DG; ++g) {
        const int d0 = g * 16 + 8 * lh;
        if (qi < p.N && d0 < D) {
            qh[g] = *(const half8_t*)(Qh + (long long)qi * p.ldq + d0);
            ql[g] = *(const half8_t*)(Qh + p.planeQ + (long long)qi * p.ldq + d0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { qh[g][j] = 0; ql[g][j] = 0; }
        }
    }
    // staging: the 4 NP pieces of a tile are dealt round-robin to the waves; piece q = plane q / NP, chunks 64 (q % NP) + lane of it,
    // chunk c -> (row c / CPR, 16-byte chunk c % CPR)
    constexpr int NPW = (4 * NP + NWV - 1) / NWV;
    int s_row[NPW];
    unsigned s_off[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int q = wid + NWV * i, j = q % NP;
        const int c = 64 * j + lane;
        s_row[i] = c / CPR;
        const int ch = c - s_row[i] * CPR;
        s_off[i] = (unsigned)((SWZ ? (ch ^ (s_row[i] & 7)) : ch) * 16);     // the swizzle is applied to the chunk FETCHED for this LDS slot
    }
    auto issue = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int q = wid + NWV * i;                     // wave-uniform
            if (q < 4 * NP) {
                const int pl = q / NP, j = q - pl * NP;
                const int row = t * KT + s_row[i];
                const int ldr = pl < 2 ? p.ldk : p.ldv;
                const char* g = row < p.L ? srcs[pl] + (long long)row * ldr * 2 + s_off[i] : zp;
                x3_glds16(g, smem_p + buf * (4 * PL) + pl * PL + j * 512);
            }
        }
    };
    // fragment offsets (halves inside a plane of the tile)
    int koff[DG];                                // K row li of a 32-key sub-tile, chunk 2 g + lh (clamped to chunk 0 in the d padding)
#pragma unroll
    for (int g = 0; g < DG; ++g) {
        const int chunk = (g * 16 + 8 * lh < D) ? 2 * g + lh : 0;
        koff[g] = li * D + (SWZ ? ((chunk ^ (li & 7)) << 3) : (chunk << 3));
    }
    const int L16 = lane & 15;
    const int vrow = 4 * lh + (L16 >> 2);        // row of this lane's 8-byte granule inside a 16-key step
    int voff[DT];
#pragma unroll
    for (int tt = 0; tt < DT; ++tt) {
        int col = 32 * tt + 16 * ((lane >> 4) & 1) + 4 * (L16 & 3);
        if (col >= D) col = 0;                   // feeds output rows >= D only (never stored)
        voff[tt] = col;
    }
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc2 = p.scale * 1.44269504088896341f;      // scores in log2 units (operand scale 1)
    const int nt = (p.L + KT - 1) / KT;
    issue(0, 0);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's plane of tile t has landed
        __syncthreads();                                    // everybody's has; everybody is done with the other buffer
        if (t + 1 < nt) issue(t + 1, (t + 1) & 1);
        const half_t* Kh = smem_p + (t & 1) * (4 * PL);
        const half_t* Kl = Kh + PL;
        const half_t* Vh = Kl + PL;
        const half_t* Vl = Vh + PL;
        __builtin_amdgcn_s_setprio(1);                      // matrix phases run at raised priority: at equal priority the vector streams of
        f32x16 sacc[KS];                                    // the SIMD's other waves take the issue port and the MFMAs of this one starve
#pragma unroll
        for (int u = 0; u < KS; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[u][r] = 0.f;
            // all 2 DG fragment reads of the sub-tile go out before the first MFMA (left alone the compiler waits for each read in turn:
            // 2 DG exposed LDS latencies per sub-tile on the wave's critical path)
            half8_t kh[DG], kl[DG];
#pragma unroll
            for (int g = 0; g < DG; ++g) {
                kh[g] = *(const half8_t*)(Kh + u * 32 * D + koff[g]);
                kl[g] = *(const half8_t*)(Kl + u * 32 * D + koff[g]);
            }
            if constexpr (BATCH_READS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < DG; ++g) {
                sacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[g], qh[g], sacc[u], 0, 0, 0);
                sacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[g], ql[g], sacc[u], 0, 0, 0);
                sacc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[g], qh[g], sacc[u], 0, 0, 0);
            }
            if constexpr (BATCH_READS) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        // online softmax (register r <-> key (r & 3) + 8 (r >> 2) + 4 lh of its sub-tile)
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            if (t * KT + (u + 1) * 32 > p.L) {    // wave-uniform: only the tile that crosses L is masked
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KT + u * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (key >= p.L) sacc[u][r] = -INFINITY;
                }
            }
        }
        // LAZY running maximum (as the fp16 kernel, csrc/attention.hip): it moves only when some score of the tile exceeds it by more
        // than 2^THR in the exponent -- a wave-uniform test, no cross-lane exchange and no rescale on the common path; P <= 2^THR then,
        // and the hi / lo split scale 2^LOGSP rides in the exponent (P' = 2^10 P <= 2^15: inside fp16; O and l carry it alike)
        float mxa = -INFINITY, mxb = -INFINITY;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KS == 2) {
                mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, sacc[0][r]), sacc[KS - 1][r]);
                mxb = __builtin_fmaxf(__builtin_fmaxf(mxb, sacc[0][8 + r]), sacc[KS - 1][8 + r]);
            } else {
                mxa = __builtin_fmaxf(mxa, sacc[0][r]);
                mxb = __builtin_fmaxf(mxb, sacc[0][8 + r]);
            }
        }
        float mx = fmaxf(mxa, mxb) * sc2;
        if (__builtin_amdgcn_ballot_w64(mx > m_run + THR) != 0) {
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // first tile: exp2(-inf) = 0, O and l are 0
#pragma unroll
            for (int tt = 0; tt < DT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[tt][r] *= alpha;
            l_run *= alpha;
            m_run = m_new;
        }
        const float mb = LOGSP - m_run;
        float ls = 0.f;
#pragma unroll
        for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[u][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[u][r], sc2, mb)); ls += sacc[u][r]; }
        l_run += ls;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int u = 0; u < KS; ++u) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                half8_t ph, pl;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    const f32x4 v = {sacc[u][8 * s2 + 4 * c2], sacc[u][8 * s2 + 4 * c2 + 1], sacc[u][8 * s2 + 4 * c2 + 2], sacc[u][8 * s2 + 4 * c2 + 3]};
                    half4 hh, ll;
                    split4(v, 1.0f, hh, ll);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ph[4 * c2 + j] = hh[j]; pl[4 * c2 + j] = ll[j]; }
                }
                // the 4 DT transposing reads of this 16-key step go out before its first MFMA (as the K fragments above)
                half8_t vh[DT], vl[DT];
#pragma unroll
                for (int tt = 0; tt < DT; ++tt) {
                    // keys {4 lh .. + 3} and {8 + 4 lh .. + 3} of this 16-key step: rows r0, r0 + 8 of the tile
                    const int r0 = u * 32 + 16 * s2 + vrow;
                    const int c = voff[tt];
                    const int o0 = r0 * D + (SWZ ? ((((c >> 3) ^ (r0 & 7)) << 3) | (c & 7)) : c);
                    const int o1 = (r0 + 8) * D + (SWZ ? ((((c >> 3) ^ ((r0 + 8) & 7)) << 3) | (c & 7)) : c);
                    const half4 a0 = x3_lds_tr_read(Vh + o0), a1 = x3_lds_tr_read(Vh + o1);
                    const half4 b0 = x3_lds_tr_read(Vl + o0), b1 = x3_lds_tr_read(Vl + o1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { vh[tt][j] = a0[j]; vh[tt][4 + j] = a1[j]; vl[tt][j] = b0[j]; vl[tt][4 + j] = b1[j]; }
                }
                if constexpr (BATCH_READS) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tt = 0; tt < DT; ++tt) {
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[tt], ph, o[tt], 0, 0, 0);
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[tt], pl, o[tt], 0, 0, 0);
                    o[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[tt], ph, o[tt], 0, 0, 0);
                }
                if constexpr (BATCH_READS) __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;                          // O and l carry the same 2^10
    if (qi < p.N) {
#pragma unroll
        for (int tt = 0; tt < DT; ++tt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = tt * 32 + 8 * g + 4 * lh;
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

// called by ief_attn_flash_f32 (exact_f32.hip) after its argument checks when p.x3 != 0
int ief_attn_flash_x3_dispatch(const IefAttnF32Params& p, hipStream_t st) {
    if (p.Out && ((p.ldo & 3) || (p.sOb & 3) || ((uintptr_t)p.Out & 15))) return IEF_EALIGN;
    if (p.OutP && ((p.ldp & 3) || (p.sOPb & 3) || (p.planeO & 3) || ((uintptr_t)p.OutP & 7))) return IEF_EALIGN;
    const unsigned long long lim = 0xFFFFFFF0ull;
    if (((unsigned long long)(p.L - 1) * p.ldk + p.d) * 4 >= lim || ((unsigned long long)(p.L - 1) * p.ldv + p.d) * 4 >= lim) return IEF_ESHAPE;
    dim3 grid((p.N + 127) / 128, p.B * p.heads);
    if (p.Qp) {         // operand planes in: K / V tiles by LDS-DMA (attn_flash_x3p_kernel)
        const dim3 grid8((p.N + 255) / 256, p.B * p.heads);
        if (!p.Kp || !p.Vp || !p.zeros) return IEF_EINVAL;
        if ((p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.sQb & 7) || (p.sKb & 7) || (p.sVb & 7) || (p.planeQ & 7) || (p.planeK & 7) ||
            (p.planeV & 7) || (((uintptr_t)p.Qp | (uintptr_t)p.Kp | (uintptr_t)p.Vp) & 15)) return IEF_EALIGN;
        switch (p.d) {
            case 40: hipLaunchKernelGGL((attn_flash_x3p_kernel<40, 2, 8>), grid8, dim3(512), 0, st, p); break;
            case 64: hipLaunchKernelGGL((attn_flash_x3p_kernel<64, 1, 4>), grid, dim3(256), 0, st, p); break;
            case 80: hipLaunchKernelGGL((attn_flash_x3p_kernel<80, 1, 4>), grid, dim3(256), 0, st, p); break;
            default: return IEF_ESHAPE;
        }
        IEF_LAUNCH_CHECK();
        return IEF_OK;
    }
    // 64 keys per barrier pair where there are that many (self-attention); the 77-key cross maps keep 32 (3 tiles, not 2 x 64)
    const bool ks2 = g_flash_ks2 && p.L >= 128;
#define FLASH_GO(D_) do { if (ks2) hipLaunchKernelGGL((attn_flash_x3_kernel<D_, 2>), grid, dim3(256), 0, st, p); \
                          else hipLaunchKernelGGL((attn_flash_x3_kernel<D_, 1>), grid, dim3(256), 0, st, p); } while (0)
    switch (p.d) {
        case 32: FLASH_GO(32); break;
        case 40: FLASH_GO(40); break;
        case 64: FLASH_GO(64); break;
        case 80: FLASH_GO(80); break;
        case 160: FLASH_GO(160); break;
        default: return IEF_ESHAPE;
    }
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
