// Reference-precision ("exact") mode of the denoising path: fp32 storage, fp32 operands, fp32 accumulation.
//
// The reference computes in fp32 (`/root/reference/p2p/edit_syn.py:38`).  The default path stores activations and
// weights in fp16; its ~2e-3 error per UNet forward comes from the OPERAND rounding of every contraction (fp16 weights
// 1.0e-3 + fp16 inputs 0.9e-3; an fp32 residual stream alone changes nothing — DESIGN.md §4), so a mode that meets
// north_star's 1e-3 bound on edited images needs full-precision operands.  gfx950 has an fp32-input MFMA
// (`v_mfma_f32_32x32x2_f32`, 64 FLOP/clk/SIMD = 157 TFLOP/s, bitwise an fp32 fma chain): every contraction of this
// mode runs on it, every other op is fp32 VALU.
//
//   igemm_f32_kernel<CONV, TRANSB, NT>  linear / 1x1 / 3x3 implicit GEMM (concat sources, nearest-2x, stride 2, fused 1x1
//                                       shortcut sources, bias + per-image row vector + residual epilogue) and the two
//                                       batched products of a MATERIALISED attention (scores = Q K^T, out = P V with
//                                       TRANSB), with batch-row indirection for Q / K / V (P2P self-replace, MasaCtrl,
//                                       Plug-and-Play)
//   softmax_rows_f32, p2p_cross_edit_f32   the map pipeline between the two products (`register.py:47-48`,
//                                       `attention_base.py:118-121`): maps live in HBM in this mode, as in the reference
//   groupnorm / layernorm / geglu / add / silu / timestep embedding / boundary convs in fp32
//
// Tile: 128 x (64 NT) x 32, 4 waves (2 x 2), each wave 2 x NT MFMA tiles of 32 x 32; operands staged through LDS with
// 36-float rows (conflict-free ds_read_b128: consecutive rows start 4 banks apart mod 64), next K tile prefetched into
// registers while the current one is multiplied.  The k index INSIDE an 8-deep group is permuted identically for A and B
// (lane half h takes k = 4h..4h+3), which lets a lane fetch its four operands of four MFMAs with one 16-byte LDS read.
#include "ief_common.h"
#include "ief_params.h"
#include "x3_common.h"

#define XBM 128
#define XBK 32
#define XLD 36

__device__ __forceinline__ float silu_x(float x) { return x / (1.0f + expf(-x)); }

struct RowCoord { int b, oy, ox, ok; };

template <bool CONV, bool TRANSB, int NT>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const IefGemmF32Params p) {
    constexpr int XBN = 64 * NT;
    __shared__ __attribute__((aligned(16))) float smem_x[2 * (XBM + XBN) * XLD];
    float* As = smem_x;                       // [2][XBM][XLD]
    float* Bs = smem_x + 2 * XBM * XLD;       // [2][XBN][XLD]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int ntn = (p.N + XBN - 1) / XBN;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x - tm * ntn;
    const int m0 = tm * XBM, n0 = tn * XBN;
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
    // ---- loader assignment: A tile = 128 rows x 8 chunks of 4 floats; thread -> 4 rows (32 apart), one chunk column
    const int a_kc = tid & 7, a_r0 = tid >> 3;
    RowCoord rc[4];
    if (CONV) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + a_r0 + 32 * i;
            rc[i].ok = m < M;
            const int mm = rc[i].ok ? m : 0;
            const int hw = p.Ho * p.Wo;
            rc[i].b = mm / hw;
            const int rem = mm - rc[i].b * hw;
            rc[i].oy = rem / p.Wo;
            rc[i].ox = rem - rc[i].oy * p.Wo;
        }
    }
    const int Ct = p.C1 + p.C2, K9 = 9 * Ct;
    const int pad_lo = p.pad_hi_only ? 0 : 1;
    const int Hs = p.ups ? (p.H >> 1) : p.H, Ws = p.ups ? (p.Wd >> 1) : p.Wd;   // dims of the stored source

    f32x4 ra[4], rb[2 * NT];
    auto load_tile = [&](int k0) {
        const int kk = k0 + a_kc * 4;
        if (CONV) {
            const float* src = nullptr;
            int cs = 0, chs = 0, ky = 0, kx = 0, mode = 0;       // mode 0: zero, 1: 3x3 tap, 2: 1x1 extra source
            if (kk < K9) {
                const int tap = kk / Ct, ch = kk - tap * Ct;
                ky = tap / 3; kx = tap - 3 * ky;
                if (ch < p.C1) { src = p.A; cs = p.C1; chs = ch; } else { src = p.A2; cs = p.C2; chs = ch - p.C1; }
                mode = 1;
            } else if (kk < K) {
                const int ch2 = kk - K9;
                if (ch2 < p.CE1) { src = p.E1; cs = p.CE1; chs = ch2; } else { src = p.E2; cs = p.CE2; chs = ch2 - p.CE1; }
                mode = 2;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
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
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + a_r0 + 32 * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < M && kk < K) {
                    const float* ap = A + (long long)m * p.lda + kk;
                    if (!p.a_scalar) v = *(const f32x4*)ap;
                    else {                      // rows of 77 keys: neither the row stride nor K is a multiple of 4 floats
#pragma unroll
                        for (int j = 0; j < 4; ++j) if (kk + j < K) v[j] = ap[j];
                    }
                }
                ra[i] = v;
            }
        }
        if (!TRANSB) {                  // W [N][K]: rows n0 + a_r0 + 32 i
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) {
                const int n = n0 + a_r0 + 32 * i;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < N && kk < K) v = *(const f32x4*)(W + (long long)n * p.ldw + kk);
                rb[i] = v;
            }
        } else {                        // W [K][N]: thread -> k row (tid & 31), chunk columns (tid >> 5) + 8 i of 4 n each
            const int kr = k0 + (tid & 31);
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) {
                const int n = n0 + ((tid >> 5) + 8 * i) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kr < K && n < N) v = *(const f32x4*)(W + (long long)kr * p.ldw + n);   // N % 4 == 0 (host-checked)
                rb[i] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* as = As + buf * XBM * XLD;
        float* bs = Bs + buf * XBN * XLD;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(f32x4*)(as + (a_r0 + 32 * i) * XLD + a_kc * 4) = ra[i];
        if (!TRANSB) {
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) *(f32x4*)(bs + (a_r0 + 32 * i) * XLD + a_kc * 4) = rb[i];
        } else {
            const int kr = tid & 31;
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) {
                const int nl = ((tid >> 5) + 8 * i) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) bs[(nl + j) * XLD + kr] = rb[i][j];
            }
        }
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // split-K (grid.y): this workgroup's K tiles [kt0, kt0 + nk); the partial tile goes to an fp32 slab, a second launch sums
    // the slabs in slab order and applies the epilogue (small-M levels would otherwise leave most of the chip idle)
    const int nk_all = (K + XBK - 1) / XBK;
    int kt0 = 0, nk = nk_all;
    if (p.splits > 1) {
        const int per = (nk_all + p.splits - 1) / p.splits;
        kt0 = blockIdx.y * per;
        nk = min(per, nk_all - kt0);
        if (nk < 0) nk = 0;
    }
    if (nk > 0) {
        load_tile(kt0 * XBK);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt0 + kt + 1) * XBK);
        const float* as = As + buf * XBM * XLD + (wm * 64 + li) * XLD + 4 * lh;
        const float* bs = Bs + buf * XBN * XLD + (wn * 32 * NT + li) * XLD + 4 * lh;
#pragma unroll
        for (int g = 0; g < XBK / 8; ++g) {
            f32x4 fa[2], fb[NT];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = *(const f32x4*)(as + a * 32 * XLD + g * 8);
#pragma unroll
            for (int b = 0; b < NT; ++b) fb[b] = *(const f32x4*)(bs + b * 32 * XLD + g * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][s], fb[b][s], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    if (p.splits > 1) {       // raw partial sums
        float* slab = p.ws + (long long)blockIdx.y * M * N;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const int n = n0 + wn * 32 * NT + b * 32 + li;
                if (n >= N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < M) slab[(long long)m * N + n] = acc[a][b][r];
                }
            }
        return;
    }
    // ---- epilogue: (acc + bias[n] + rowvec[m / rows_per_batch][n] + residual[m][n]) * out_scale, fp32
    const float* R = p.residual;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int n = n0 + wn * 32 * NT + b * 32 + li;
            if (n >= N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= M) continue;
                float v = acc[a][b][r] + bv;
                if (p.rowvec) v += p.rowvec[(long long)(m / p.rows_per_batch) * N + n];
                if (R) v += R[(long long)m * p.ldr + n];
                Out[(long long)m * p.ldo + n] = v * p.out_scale;
            }
        }
}

// out = (sum of the split-K slabs, in slab order, + bias + rowvec + residual) * out_scale
__global__ __launch_bounds__(256) void splitk_reduce_f32_kernel(const IefGemmF32Params p) {
    const long long total = (long long)p.M * p.N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / p.N), n = (int)(i - (long long)m * p.N);
        float v = p.ws[i];
        for (int s = 1; s < p.splits; ++s) v += p.ws[(long long)s * total + i];
        if (p.bias) v += p.bias[n];
        if (p.rowvec) v += p.rowvec[(long long)(m / p.rows_per_batch) * p.N + n];
        if (p.residual) v += p.residual[(long long)m * p.ldr + n];
        p.Out[(long long)m * p.ldo + n] = v * p.out_scale;
    }
}

template <bool CONV, bool TRANSB, int NT>
static int launch_igemm_f32(const IefGemmF32Params& p, hipStream_t st) {
    constexpr int XBN = 64 * NT;
    const int tiles = ((p.M + XBM - 1) / XBM) * ((p.N + XBN - 1) / XBN);
    const int z = p.heads > 0 ? p.batch * p.heads : 1;
    const int splits = p.splits > 1 ? p.splits : 1;
    hipLaunchKernelGGL((igemm_f32_kernel<CONV, TRANSB, NT>), dim3(tiles, splits, z), dim3(256), 0, st, p);
    IEF_LAUNCH_CHECK();
    if (splits > 1) {
        const long long total = (long long)p.M * p.N;
        int grid = (int)((total + 255) / 256);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(splitk_reduce_f32_kernel, dim3(grid), dim3(256), 0, st, p);
        IEF_LAUNCH_CHECK();
    }
    return IEF_OK;
}

// column-tile width the launcher picks for N output columns: 64 when that wastes less of the last tile than 128
extern "C" int ief_gemm_f32_bn(int N) { return (N <= 64 || ((N % 128) != 0 && (N % 128) <= 64)) ? 64 : 128; }

int ief_gemm_x3_dispatch(const IefGemmF32Params& p, hipStream_t st);      // split_x3.hip

static int launch_x3(const IefGemmF32Params& p, hipStream_t st) {
    if (!(p.sa > 0.f) || !(p.sb > 0.f)) return IEF_EINVAL;
    const int rc = ief_gemm_x3_dispatch(p, st);
    if (rc != IEF_OK) return rc;
    if (p.splits > 1) {
        const long long total = (long long)p.M * p.N;
        int grid = (int)((total + 255) / 256);
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(splitk_reduce_f32_kernel, dim3(grid), dim3(256), 0, st, p);
        IEF_LAUNCH_CHECK();
    }
    return IEF_OK;
}

extern "C" int ief_gemm_f32(const IefGemmF32Params* pp, void* stream) {
    if (!pp || !pp->A || !pp->W || !pp->Out) return IEF_EINVAL;
    IefGemmF32Params p = *pp;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return IEF_ESHAPE;
    if (p.rowvec && p.rows_per_batch <= 0) return IEF_ESHAPE;
    if (p.heads > 0 && p.batch <= 0) return IEF_ESHAPE;
    if (p.splits > 1 && (!p.ws || p.heads > 0 || p.splits > 64)) return IEF_EINVAL;
    if (p.heads == 0 && (p.a_src || p.w_src)) return IEF_ESHAPE;
    if (p.geglu && !p.x3) return IEF_EINVAL;                   // the fused GEGLU epilogue exists in the split-operand kernel only
    hipStream_t st = (hipStream_t)stream;
    if (p.conv) {
        if (p.transb || p.heads > 0) return IEF_ESHAPE;
        if ((p.C1 & 3) || (p.C2 & 3) || (p.CE1 & 3) || (p.CE2 & 3) || p.C1 <= 0) return IEF_ESHAPE;
        if ((p.C2 > 0 && !p.A2) || (p.CE1 > 0 && !p.E1) || (p.CE2 > 0 && !p.E2)) return IEF_EINVAL;
        if (p.K != 9 * (p.C1 + p.C2) + p.CE1 + p.CE2 || p.M != p.batch_images * p.Ho * p.Wo) return IEF_ESHAPE;
        if (p.stride != 1 && p.stride != 2) return IEF_ESHAPE;
        if (p.ups && ((p.H | p.Wd) & 1)) return IEF_ESHAPE;
        if (p.x3) return launch_x3(p, st);
        return ief_gemm_f32_bn(p.N) == 64 ? launch_igemm_f32<true, false, 1>(p, st) : launch_igemm_f32<true, false, 2>(p, st);
    }
    p.a_scalar = ((p.lda & 3) || (p.K & 3)) ? 1 : 0;          // A rows not 16-byte chunked: element loads for A
    if (p.ldw & 3) return IEF_EALIGN;
    if (!p.transb && (p.K & 3)) return IEF_ESHAPE;             // W [N][K] rows are read in 16-byte chunks along K
    if (p.transb && (p.N & 3)) return IEF_ESHAPE;
    if (p.x3) return launch_x3(p, st);
    if (p.transb) {
        return ief_gemm_f32_bn(p.N) == 64 ? launch_igemm_f32<false, true, 1>(p, st) : launch_igemm_f32<false, true, 2>(p, st);
    }
    return ief_gemm_f32_bn(p.N) == 64 ? launch_igemm_f32<false, false, 1>(p, st) : launch_igemm_f32<false, false, 2>(p, st);
}

// --------------------------------------------------------------------------------------------- fused attention, fp32
// out[b] = softmax(scale * q[q_src[b]] k[k_src[b]]^T) v[v_src[b]] without materialising the maps (self-attention and
// un-edited cross-attention of the reference-precision mode; the P2P cross edit and the generic controller path keep the
// materialised pipeline below).  One workgroup = 128 queries of one (batch row, head), a wave = 32 of them; 32-key tiles.
//   S^T = K Q^T on v_mfma_f32_32x32x2_f32: a lane owns one query COLUMN (16 keys of the tile in its registers, the other
//   16 in lane + 32), so the running maximum / sum are in-lane plus one cross-half exchange, and the P registers are
//   directly the B operand of O^T += V^T P^T (the k order of that product is whatever the accumulator layout gives; the
//   V^T fragment is read from LDS in the matching order).  d is padded to 32-row tiles of O^T (d = 40: 64 rows).
template <int D>
__global__ __launch_bounds__(256) void attn_flash_f32_kernel(const IefAttnF32Params p) {
    constexpr int DT = (D + 31) / 32;            // 32-row tiles of O^T
    constexpr int KLD = D + 4;                   // K tile row stride (floats): rows start 4 banks apart mod 64 -> b128 reads conflict-free
    constexpr int VLD = DT * 32 + 4;             // V tile row stride; columns D..DT*32 are zero
    __shared__ __attribute__((aligned(16))) float smem_a[32 * KLD + 32 * VLD];
    float* Ks = smem_a;
    float* Vs = smem_a + 32 * KLD;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int bh = blockIdx.y, b = bh / p.heads, h = bh - b * p.heads;
    const int bq = p.q_src ? p.q_src[b] : b, bk = p.k_src ? p.k_src[b] : b, bv = p.v_src ? p.v_src[b] : b;
    const float* Q = p.Q + (long long)bq * p.sQb + (long long)h * D;
    const float* K = p.K + (long long)bk * p.sKb + (long long)h * D;
    const float* V = p.V + (long long)bv * p.sVb + (long long)h * D;
    float* O = p.Out + (long long)b * p.sOb + (long long)h * D;
    const int q0 = blockIdx.x * 128 + wid * 32;
    const int qi = q0 + li;
    // this lane's share of its query: for every 8-deep d group the 4 values of its lane half
    f32x4 qf[D / 8];
#pragma unroll
    for (int g = 0; g < D / 8; ++g) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (qi < p.N) v = *(const f32x4*)(Q + (long long)qi * p.ldq + g * 8 + 4 * lh);
        qf[g] = v;
    }
    f32x16 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    constexpr int KCH = 32 * (D / 4);            // 16-byte chunks of one K (or V) tile
    constexpr int NLD = (KCH + 255) / 256;
    f32x4 rk[NLD], rv[NLD];
    auto load_kv = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 256 * i;
            const int row = c / (D / 4), ch = c - row * (D / 4);
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, bb = {0.f, 0.f, 0.f, 0.f};
            if (c < KCH && k0 + row < p.L) {
                a = *(const f32x4*)(K + (long long)(k0 + row) * p.ldk + ch * 4);
                bb = *(const f32x4*)(V + (long long)(k0 + row) * p.ldv + ch * 4);
            }
            rk[i] = a; rv[i] = bb;
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + 256 * i;
            if (c < KCH) {
                const int row = c / (D / 4), ch = c - row * (D / 4);
                *(f32x4*)(Ks + row * KLD + ch * 4) = rk[i];
                *(f32x4*)(Vs + row * VLD + ch * 4) = rv[i];
            }
        }
    };
    if (DT * 32 > D) {                            // zero the padding columns of the V tile once
        for (int c = tid; c < 32 * (DT * 32 - D); c += 256) {
            const int row = c / (DT * 32 - D), col = D + c - row * (DT * 32 - D);
            Vs[row * VLD + col] = 0.f;
        }
    }
    const int nt = (p.L + 31) / 32;
    load_kv(0);
    for (int t = 0; t < nt; ++t) {
        __syncthreads();                          // everybody is done with the previous tile
        store_kv();
        __syncthreads();
        if (t + 1 < nt) load_kv((t + 1) * 32);
        // ---- S^T tile: 32 keys x 32 queries
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < D / 8; ++g) {
            const f32x4 kf = *(const f32x4*)(Ks + li * KLD + g * 8 + 4 * lh);
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s2], qf[g][s2], sacc, 0, 0, 0);
        }
        // ---- online softmax over the key rows of this tile (register r <-> key (r&3) + 8 (r>>2) + 4 lh)
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            sacc[r] = key < p.L ? sacc[r] * p.scale : -INFINITY;
            mx = fmaxf(mx, sacc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = expf(sacc[r] - m_new); ls += sacc[r]; }
        l_run = l_run * alpha + ls;
        m_run = m_new;
#pragma unroll
        for (int tt = 0; tt < DT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[tt][r] *= alpha;
        // ---- O^T += V^T P^T: step r multiplies keys {(r&3) + 8 (r>>2), + 4}: lane half lh supplies / reads its own key row
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vrow = Vs + ((r & 3) + 8 * (r >> 2) + 4 * lh) * VLD + li;
#pragma unroll
            for (int tt = 0; tt < DT; ++tt) o[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[tt * 32], sacc[r], o[tt], 0, 0, 0);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (qi < p.N) {
#pragma unroll
        for (int tt = 0; tt < DT; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (d < D) O[(long long)qi * p.ldo + d] = o[tt][r] * inv;
            }
    }
}

int ief_attn_flash_x3_dispatch(const IefAttnF32Params& p, hipStream_t st);      // split_x3.hip

extern "C" int ief_attn_flash_f32(const IefAttnF32Params* pp, void* stream) {
    if (!pp || (!pp->Qp && (!pp->Q || !pp->K || !pp->V)) || (!pp->Out && !pp->OutP)) return IEF_EINVAL;
    const IefAttnF32Params p = *pp;
    if ((p.OutP || p.Qp) && !p.x3) return IEF_EINVAL;    // operand planes are read / written by the split-operand kernels only
    if (!p.x3 && (!p.Out || (p.ldo & 3) || (p.sOb & 3))) return IEF_EINVAL;
    if (p.B <= 0 || p.heads <= 0 || p.N <= 0 || p.L <= 0) return IEF_ESHAPE;
    if ((p.ldq & 3) || (p.ldk & 3) || (p.ldv & 3) || (p.sQb & 3) || (p.sKb & 3) || (p.sVb & 3)) return IEF_EALIGN;
    dim3 grid((p.N + 127) / 128, p.B * p.heads);
    hipStream_t st = (hipStream_t)stream;
    if (p.x3) return ief_attn_flash_x3_dispatch(p, st);
    switch (p.d) {
        case 32: hipLaunchKernelGGL(attn_flash_f32_kernel<32>, grid, dim3(256), 0, st, p); break;
        case 40: hipLaunchKernelGGL(attn_flash_f32_kernel<40>, grid, dim3(256), 0, st, p); break;
        case 64: hipLaunchKernelGGL(attn_flash_f32_kernel<64>, grid, dim3(256), 0, st, p); break;
        case 80: hipLaunchKernelGGL(attn_flash_f32_kernel<80>, grid, dim3(256), 0, st, p); break;
        case 160: hipLaunchKernelGGL(attn_flash_f32_kernel<160>, grid, dim3(256), 0, st, p); break;
        default: return IEF_ESHAPE;
    }
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// --------------------------------------------------------------------------------------------- attention maps
// in-place softmax over rows of length L (fp32), one wave per row; three passes over a row that stays in L1/L2
// rows of up to 64 * SMX floats stay in registers: one read, one write
template <int SMX>
__global__ __launch_bounds__(256) void softmax_rows_f32_reg_kernel(float* __restrict__ x, long long rows, int L) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = x + row * L;
    const int lane = threadIdx.x & 63;
    float v[SMX];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < SMX; ++i) {
        const int k = lane + 64 * i;
        v[i] = k < L ? r[k] : -INFINITY;
        m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SMX; ++i) { v[i] = expf(v[i] - m); s += v[i]; }      // exp(-inf) = 0 for the padding
    const float inv = 1.0f / wave_sum(s);
#pragma unroll
    for (int i = 0; i < SMX; ++i) {
        const int k = lane + 64 * i;
        if (k < L) r[k] = v[i] * inv;
    }
}

__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(float* __restrict__ x, long long rows, int L) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = x + row * L;
    const int lane = threadIdx.x & 63;
    float m = -INFINITY;
    for (int i = lane; i < L; i += 64) m = fmaxf(m, r[i]);
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < L; i += 64) s += expf(r[i] - m);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int i = lane; i < L; i += 64) r[i] = expf(r[i] - m) * inv;
}
extern "C" int ief_softmax_rows_f32(float* x, long long rows, int L, void* stream) {
    if (!x) return IEF_EINVAL;
    if (rows <= 0 || L <= 0) return IEF_ESHAPE;
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (L <= 128) hipLaunchKernelGGL(softmax_rows_f32_reg_kernel<2>, grid, dim3(256), 0, st, x, rows, L);
    else if (L <= 1024) hipLaunchKernelGGL(softmax_rows_f32_reg_kernel<16>, grid, dim3(256), 0, st, x, rows, L);
    else if (L <= 4096) hipLaunchKernelGGL(softmax_rows_f32_reg_kernel<64>, grid, dim3(256), 0, st, x, rows, L);
    else hipLaunchKernelGGL(softmax_rows_f32_kernel, grid, dim3(256), 0, st, x, rows, L);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// Prompt-to-Prompt cross-attention edit on materialised maps [B*heads][N][L] (L <= 96), in place on the target rows:
//   P'[w] = c1[w] * sum_v P_src[v] * M[v][w] + c2[w] * P_tgt[w]       (`attention_base.py:118-121`, `attention_control.py:15-46`)
// edit_src[b] = batch row holding the source maps (or < 0: row untouched), edit_slot[b] = which (M, c1, c2);
// MT fp32 [slots][96][96] = M transposed and zero-padded, coef fp32 [slots][2][96].
// The product P_src M runs on the fp32-input MFMA (exact fp32 fma chains): a wave owns 32 map rows of one (target batch row,
// head) and all 96 output columns (three 32 x 32 accumulators); A = the source rows straight from global memory (lane: row
// lane & 31, k = 2 step + (lane >> 5)), B = M rows from an LDS copy of the slot's table (M[v][w] = MT[w][v]; padded columns
// are zero).  One workgroup = 4 waves = 128 rows.
__global__ __launch_bounds__(256) void p2p_cross_edit_f32_kernel(float* __restrict__ P, const int* __restrict__ edit_src,
                                                                 const int* __restrict__ edit_slot, const float* __restrict__ MT,
                                                                 const float* __restrict__ coef, int B, int heads, int N, int L) {
    __shared__ float sm[96 * 97];                        // sm[v * 97 + w] = M[v][w]
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
    const int sb = edit_src[b];
    if (sb < 0) return;                                  // uniform over the workgroup
    const int slot = edit_slot[b];
    const float* mt = MT + (long long)slot * 96 * 96;
    for (int i = threadIdx.x; i < 96 * 96; i += 256) { const int w = i / 96, v = i - w * 96; sm[v * 97 + w] = mt[i]; }
    __syncthreads();
    const int q0 = blockIdx.x * 128 + wv * 32;
    if (q0 >= N) return;
    const int q = min(q0 + li, N - 1);                   // rows past N: computed on a duplicate, never stored
    const float* ps = P + (((long long)sb * heads + h) * N + q) * L;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int ksteps = (L + 1) >> 1;
    for (int s = 0; s < ksteps; ++s) {
        const int v = 2 * s + lh;
        const float a = v < L ? ps[v] : 0.f;
        const float* mrow = sm + min(v, 95) * 97;
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, mrow[t * 32 + li], acc[t], 0, 0, 0);
    }
    // accumulator: column (output token) 32 t + li, rows (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32
    const float* c1 = coef + (long long)slot * 2 * 96;
    const float* c2 = c1 + 96;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int w = t * 32 + li;
        if (w >= L) continue;
        const float k1 = c1[w], k2 = c2[w];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = q0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (qq < N) {
                float* pt = P + (((long long)b * heads + h) * N + qq) * L + w;
                *pt = k1 * acc[t][r] + k2 * *pt;
            }
        }
    }
}
extern "C" int ief_p2p_cross_edit_f32(float* P, const int* edit_src, const int* edit_slot, const float* MT, const float* coef,
                                      int B, int heads, int N, int L, void* stream) {
    if (!P || !edit_src || !edit_slot || !MT || !coef) return IEF_EINVAL;
    if (B <= 0 || heads <= 0 || N <= 0 || L <= 0 || L > 96) return IEF_ESHAPE;
    hipLaunchKernelGGL(p2p_cross_edit_f32_kernel, dim3((N + 127) / 128, B * heads), dim3(256), 0, (hipStream_t)stream, P,
                       edit_src, edit_slot, MT, coef, B, heads, N, L);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// --------------------------------------------------------------------------------------------- norms
// GroupNorm (+ SiLU) over [B][HW][C1 (+ C2 concat)] fp32.  KS workgroups per (batch, group): thread (py, j) keeps channel
// pair j of the group (8-byte accesses, no index arithmetic in the loops); every workgroup streams the whole slab for the
// mean and then for the centred second moment (two-pass: no cancellation; identical sums in all KS of them, so no
// hand-off), and normalises every KS-th run of pixels.  cpg even (host-checked; odd cpg: the scalar kernel below).
__global__ __launch_bounds__(512) void groupnorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ x2, int C1, int C2,
                                                            float* __restrict__ out, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int HW, int groups, float eps, int silu,
                                                            int PY, int KS, half_t* __restrict__ outp = nullptr, long long plane = 0) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    __shared__ float red[8];
    __shared__ float bc;
    const int C = C1 + C2, cpg = C / groups, cp2 = cpg >> 1;
    const int nbg = gridDim.x / KS;
    const int ks = blockIdx.x / nbg, id = blockIdx.x - ks * nbg;
    const int b = id / groups, g = id - b * groups;
    const int j = threadIdx.x % cp2, py = threadIdx.x / cp2;
    const bool live = py < PY;
    const int c = g * cpg + 2 * j;
    const float* src = c < C1 ? x : x2;
    const int cs = c < C1 ? C1 : C2, cc = c < C1 ? c : c - C1;
    const float* base = src + (long long)b * HW * cs + cc;
    const int nw = (blockDim.x + 63) >> 6;
    auto block_sum = [&](float v) -> float {
        v = wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) { float t = 0.f; for (int w = 0; w < nw; ++w) t += red[w]; bc = t; }
        __syncthreads();
        return bc;
    };
    constexpr int GU = 8;
    float s = 0.f;
    if (live) {
        for (int p0 = py; p0 < HW; p0 += PY * GU) {
            f32x2 v[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) { const int p = p0 + u * PY; v[u] = p < HW ? *(const f32x2*)(base + (long long)p * cs) : (f32x2){0.f, 0.f}; }
#pragma unroll
            for (int u = 0; u < GU; ++u) s += v[u][0] + v[u][1];
        }
    }
    const float n = (float)cpg * (float)HW;
    const float mean = block_sum(s) / n;
    float q = 0.f;
    if (live) {
        for (int p0 = py; p0 < HW; p0 += PY * GU) {
            f32x2 v[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) { const int p = p0 + u * PY; v[u] = p < HW ? *(const f32x2*)(base + (long long)p * cs) : (f32x2){mean, mean}; }
#pragma unroll
            for (int u = 0; u < GU; ++u) { const float d0 = v[u][0] - mean, d1 = v[u][1] - mean; q += d0 * d0 + d1 * d1; }
        }
    }
    const float rstd = 1.0f / sqrtf(block_sum(q) / n + eps);
    if (!live) return;
    const float sc0 = rstd * gamma[c], sc1 = rstd * gamma[c + 1];
    const float sh0 = beta[c] - mean * sc0, sh1 = beta[c + 1] - mean * sc1;
    float* ob = out + (long long)b * HW * C + c;
    for (int p0 = py + ks * PY * GU; p0 < HW; p0 += PY * GU * KS) {
        f32x2 v[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) { const int p = p0 + u * PY; v[u] = p < HW ? *(const f32x2*)(base + (long long)p * cs) : (f32x2){0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int p = p0 + u * PY;
            if (p < HW) {
                float y0 = v[u][0] * sc0 + sh0, y1 = v[u][1] * sc1 + sh1;
                if (silu) { y0 = silu_x(y0); y1 = silu_x(y1); }
                if (out) *(f32x2*)(ob + (long long)p * C) = (f32x2){y0, y1};
                if (outp) {          // operand planes (csrc/gemm_x3p.hip): hi = fp16(y), lo = fp16(y - hi)
                    const half2_t h = __builtin_convertvector((f32x2){y0, y1}, half2_t);
                    const half2_t l = __builtin_convertvector((f32x2){y0 - (float)h[0], y1 - (float)h[1]}, half2_t);
                    half_t* o = outp + ((long long)b * HW + p) * C + c;
                    *(half2_t*)o = h;
                    *(half2_t*)(o + plane) = l;
                }
            }
        }
    }
}

// any cpg (odd included): one workgroup per (batch, group), element-wise indexing
__global__ __launch_bounds__(256) void groupnorm_f32_scalar_kernel(const float* __restrict__ x, const float* __restrict__ x2, int C1,
                                                                   int C2, float* __restrict__ out, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, int HW, int groups, float eps,
                                                                   int silu) {
    __shared__ float red[4];
    __shared__ float bc;
    const int C = C1 + C2, cpg = C / groups;
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int c0 = g * cpg;
    const long long n = (long long)HW * cpg;
    auto at = [&](long long i) -> float {
        const long long pix = i / cpg;
        const int c = c0 + (int)(i - pix * cpg);
        return c < C1 ? x[((long long)b * HW + pix) * C1 + c] : x2[((long long)b * HW + pix) * C2 + (c - C1)];
    };
    auto block_sum = [&](float v) -> float {
        v = wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) bc = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
        return bc;
    };
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) s += at(i);
    const float mean = block_sum(s) / (float)n;
    float q = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) { const float d = at(i) - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(block_sum(q) / (float)n + eps);
    for (long long i = threadIdx.x; i < n; i += 256) {
        const long long pix = i / cpg;
        const int c = c0 + (int)(i - pix * cpg);
        float v = (at(i) - mean) * rstd * gamma[c] + beta[c];
        if (silu) v = silu_x(v);
        out[((long long)b * HW + pix) * C + c] = v;
    }
}
// ---- GroupNorm in three launches that stream whole rows (every access a 16-byte piece of a contiguous channel run):
//   gn3_stats:    workgroup = (image, run of P pixels); a thread keeps one channel quad over the run: per-channel mean, then
//                 the centred second moment about it (two passes over a run that stays in L1 / L2: no cancellation)
//   gn3_finalize: one wave per (image, group): Chan's pairwise merge of the (count, mean, M2) of its channels x runs in a
//                 fixed order (bit-reproducible), then per-channel scale = rstd gamma, shift = beta - mean scale
//   gn3_apply:    y = x scale + shift (+ SiLU), same thread layout as the statistics pass
// (the one-launch kernel above reads a 40-byte slice of every 1280-byte pixel row: 0.74 TB/s; this form: see DESIGN.md)
#define GN3_U 12          // pixels a thread keeps in registers: a run is GN3_U x (pixel lanes of the workgroup) pixels
static inline void gn3_chunks(int HW, int C, int* nch, int* P) {
    const int CQ = C >> 2, CQB = CQ < 256 ? CQ : 256, PY = 256 / CQB;
    int p = GN3_U * PY;
    if (p > HW) p = HW;
    *P = p;
    *nch = (HW + p - 1) / p;
}
extern "C" long long ief_groupnorm_f32_ws_floats(int B, int HW, int C) {
    int nch, P;
    gn3_chunks(HW, C, &nch, &P);
    return 2ll * B * nch * C + 2ll * B * C;
}

__device__ __forceinline__ const float* gn3_src(const float* x, const float* x2, int C1, int C2, int b, int HW, int c, int* cs) {
    if (c < C1) { *cs = C1; return x + (long long)b * HW * C1 + c; }
    *cs = C2;
    return x2 + (long long)b * HW * C2 + (c - C1);
}

__global__ __launch_bounds__(256) void gn3_stats_kernel(const float* __restrict__ x, const float* __restrict__ x2, int C1, int C2,
                                                        int HW, int P, float* __restrict__ pmean, float* __restrict__ pm2) {
    __shared__ f32x4 red[256];
    const int C = C1 + C2, CQ = C >> 2;
    const int CQB = CQ < 256 ? CQ : 256, PY = 256 / CQB;
    const int tid = threadIdx.x, cq = tid % CQB, py = tid / CQB;
    const bool lane_ok = py < PY;
    const int chunk = blockIdx.x, b = blockIdx.y, nch = gridDim.x;
    const int p0 = chunk * P, pn = min(P, HW - p0);
    for (int qb = 0; qb * CQB < CQ; ++qb) {
        const int q = qb * CQB + cq;
        const bool ok = lane_ok && q < CQ;
        int cs = 4;
        const float* src = gn3_src(x, x2, C1, C2, b, HW, ok ? q * 4 : 0, &cs) + (long long)p0 * cs;
        // the thread's pixels py, py + PY, ... of the run: all loads issued before the first use
        f32x4 v[GN3_U];
#pragma unroll
        for (int u = 0; u < GN3_U; ++u) {
            const int pp = py + u * PY;
            v[u] = (ok && pp < pn) ? *(const f32x4*)(src + (long long)pp * cs) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < GN3_U; ++u) s += v[u];
        __syncthreads();
        red[tid] = s;
        __syncthreads();
        f32x4 mean = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            for (int y = 0; y < PY; ++y) mean += red[y * CQB + cq];        // fixed order: every pixel lane holds the same mean
            mean = mean * (1.0f / (float)pn);
        }
        f32x4 m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < GN3_U; ++u) {
            if (py + u * PY < pn) { const f32x4 d = v[u] - mean; m2 += d * d; }
        }
        __syncthreads();
        red[tid] = m2;
        __syncthreads();
        if (ok && py == 0) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int y = 0; y < PY; ++y) t += red[y * CQB + cq];
            const long long o = ((long long)b * nch + chunk) * C + q * 4;
            *(f32x4*)(pmean + o) = mean;
            *(f32x4*)(pm2 + o) = t;
        }
    }
}

// (n, mean, M2) <- merge with (nb, mb, M2b)
__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float mb, float m2b) {
    if (nb == 0.f) return;
    const float nn = n + nb, d = mb - mean;
    mean += d * (nb / nn);
    m2 += m2b + d * d * (n * nb / nn);
    n = nn;
}

__global__ __launch_bounds__(256) void gn3_finalize_kernel(const float* __restrict__ pmean, const float* __restrict__ pm2, int C, int HW,
                                                           int P, int nch, int groups, float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift,
                                                           float* __restrict__ gstats) {
    __shared__ float sn[4], sm[4], sq[4];
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups, cpg = C / groups, tid = threadIdx.x, lane = tid & 63;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    // entries = runs x channels of the group; a thread walks runs (stride 256 / cpg-rounded) for ONE channel: no division in
    // the loop, every load independent of the merges before it
    const int cl = tid % cpg, r0 = tid / cpg, rstep = 256 / cpg;
    if (r0 < rstep) {
        const float* pm = pmean + (long long)b * nch * C + g * cpg + cl;
        const float* pq = pm2 + (long long)b * nch * C + g * cpg + cl;
        for (int ch = r0; ch < nch; ch += rstep) {
            const float cnt = (float)min(P, HW - ch * P);
            chan_merge(n, mean, m2, cnt, pm[(long long)ch * C], pq[(long long)ch * C]);
        }
    }
    auto pair = [&](float& n_, float& me_, float& q_, int off) {      // fixed pairing: identical in every run
        const float nb = __shfl_xor(n_, off), mb = __shfl_xor(me_, off), qb = __shfl_xor(q_, off);
        float n1 = n_, me1 = me_, q1 = q_, n2 = nb, me2 = mb, q2 = qb;
        if (lane & off) { n1 = nb; me1 = mb; q1 = qb; n2 = n_; me2 = me_; q2 = q_; }     // lower lane's triple first
        chan_merge(n1, me1, q1, n2, me2, q2);
        n_ = n1; me_ = me1; q_ = q1;
    };
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pair(n, mean, m2, off);
    if (lane == 0) { sn[tid >> 6] = n; sm[tid >> 6] = mean; sq[tid >> 6] = m2; }
    __syncthreads();
    n = sn[0]; mean = sm[0]; m2 = sq[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) chan_merge(n, mean, m2, sn[w], sm[w], sq[w]);
    const float rstd = 1.0f / sqrtf(m2 / n + eps);
    if (gstats && tid == 0) { gstats[(long long)blockIdx.x * 2] = mean; gstats[(long long)blockIdx.x * 2 + 1] = rstd; }     // the reverse pass reads these
    for (int j = tid; j < cpg; j += 256) {
        const int c = g * cpg + j;
        const float sc = rstd * gamma[c];
        scale[(long long)b * C + c] = sc;
        shift[(long long)b * C + c] = beta[c] - mean * sc;
    }
}

// outp != nullptr: the result leaves as the two fp16 planes of the split-operand mode (hi = fp16(y), lo = fp16(y - hi); the lo
// plane `plane` elements after hi) -- the GroupNorm is the PRODUCER of its consumer GEMM's / convolution's operand planes
// (csrc/gemm_x3p.hip); out may then be null
__global__ __launch_bounds__(256) void gn3_apply_kernel(const float* __restrict__ x, const float* __restrict__ x2, int C1, int C2, int HW,
                                                        int P, const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ out, int silu, half_t* __restrict__ outp, long long plane) {
    const int C = C1 + C2, CQ = C >> 2;
    const int CQB = CQ < 256 ? CQ : 256, PY = 256 / CQB;
    const int tid = threadIdx.x, cq = tid % CQB, py = tid / CQB;
    if (py >= PY) return;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int p0 = chunk * P, pn = min(P, HW - p0);
    for (int qb = 0; qb * CQB < CQ; ++qb) {
        const int q = qb * CQB + cq;
        if (q >= CQ) continue;
        int cs = 0;
        const float* src = gn3_src(x, x2, C1, C2, b, HW, q * 4, &cs) + (long long)p0 * cs;
        const long long doff = ((long long)b * HW + p0) * C + q * 4;
        f32x4 v[GN3_U];
#pragma unroll
        for (int u = 0; u < GN3_U; ++u) {
            const int pp = py + u * PY;
            v[u] = pp < pn ? *(const f32x4*)(src + (long long)pp * cs) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const f32x4 sc = *(const f32x4*)(scale + (long long)b * C + q * 4), sh = *(const f32x4*)(shift + (long long)b * C + q * 4);
#pragma unroll
        for (int u = 0; u < GN3_U; ++u) {
            const int pp = py + u * PY;
            if (pp < pn) {
                f32x4 y = v[u] * sc + sh;
                if (silu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = silu_x(y[j]);
                }
                if (out) *(f32x4*)(out + doff + (long long)pp * C) = y;
                if (outp) {
                    half4 h, l;
                    split4(y, 1.0f, h, l);
                    half_t* o = outp + doff + (long long)pp * C;
                    *(half4*)o = h;
                    *(half4*)(o + plane) = l;
                }
            }
        }
    }
}

extern "C" int ief_groupnorm_silu_f32_ws(const float* x, const float* x2, int C1, int C2, float* out, const float* gamma,
                                         const float* beta, int B, int HW, int groups, float eps, int silu, float* ws,
                                         long long ws_floats, void* stream) {
    if (!x || !out || !gamma || !beta || !ws || (C2 > 0 && !x2)) return IEF_EINVAL;
    if (B <= 0 || HW <= 0 || groups <= 0 || C1 <= 0 || C2 < 0 || (C1 + C2) % groups || (C1 & 3) || (C2 & 3)) return IEF_ESHAPE;
    const int C = C1 + C2;
    if (C / groups > 256) return IEF_ESHAPE;
    if (ws_floats < ief_groupnorm_f32_ws_floats(B, HW, C)) return IEF_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)ws | (uintptr_t)(x2 ? x2 : x)) & 15) return IEF_EALIGN;
    int nch, P;
    gn3_chunks(HW, C, &nch, &P);
    float* pmean = ws;
    float* pm2 = pmean + (long long)B * nch * C;
    float* scale = pm2 + (long long)B * nch * C;
    float* shift = scale + (long long)B * C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn3_stats_kernel, dim3(nch, B), dim3(256), 0, st, x, x2, C1, C2, HW, P, pmean, pm2);
    IEF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn3_finalize_kernel, dim3(B * groups), dim3(256), 0, st, pmean, pm2, C, HW, P, nch, groups, eps, gamma, beta,
                       scale, shift, (float*)nullptr);
    IEF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn3_apply_kernel, dim3(nch, B), dim3(256), 0, st, x, x2, C1, C2, HW, P, scale, shift, out, silu, (half_t*)nullptr, 0ll);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
// the same three launches with the result written as operand planes (and, optionally, as fp32 too)
extern "C" int ief_groupnorm_silu_x3p_ws(const float* x, const float* x2, int C1, int C2, float* out, ief_half* outp, long long plane,
                                         const float* gamma, const float* beta, int B, int HW, int groups, float eps, int silu,
                                         float* ws, long long ws_floats, void* stream) {
    if (!x || !outp || !gamma || !beta || !ws || (C2 > 0 && !x2)) return IEF_EINVAL;
    if (B <= 0 || HW <= 0 || groups <= 0 || C1 <= 0 || C2 < 0 || (C1 + C2) % groups || (C1 & 3) || (C2 & 3) || (plane & 3)) return IEF_ESHAPE;
    const int C = C1 + C2;
    if (C / groups > 256) return IEF_ESHAPE;
    if (ws_floats < ief_groupnorm_f32_ws_floats(B, HW, C)) return IEF_EINVAL;
    if ((((uintptr_t)x | (uintptr_t)ws | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)(out ? out : x)) & 15) || ((uintptr_t)outp & 7)) return IEF_EALIGN;
    int nch, P;
    gn3_chunks(HW, C, &nch, &P);
    float* pmean = ws;
    float* pm2 = pmean + (long long)B * nch * C;
    float* scale = pm2 + (long long)B * nch * C;
    float* shift = scale + (long long)B * C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn3_stats_kernel, dim3(nch, B), dim3(256), 0, st, x, x2, C1, C2, HW, P, pmean, pm2);
    IEF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn3_finalize_kernel, dim3(B * groups), dim3(256), 0, st, pmean, pm2, C, HW, P, nch, groups, eps, gamma, beta,
                       scale, shift, (float*)nullptr);
    IEF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn3_apply_kernel, dim3(nch, B), dim3(256), 0, st, x, x2, C1, C2, HW, P, scale, shift, out, silu, (half_t*)outp, plane);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---- GroupNorm (+SiLU) BACKWARD in the same row-streaming form (the one-workgroup-per-(image, group) kernel of
// backward_f32.hip runs 32 workgroups at UNet batch 1 and walks 40-byte slices: 197 us per call in the null-text loop).
//   xh = (x - mean) rstd, z = xh gamma + beta, g = dy silu'(z) gamma, dx = rstd (g - mean(g) - xh mean(g xh)) + add
// Five launches: statistics of x (gn3_stats + gn3_finalize, as the forward), per-run per-channel sums of g and g xh, their
// per-group means (fixed order), the result.  A channel quad may straddle two groups (10 channels per group at C = 320): every
// component carries its own group's (mean, rstd, moments).
#define GN3B_U 6
static inline void gn3b_chunks(int HW, int C, int* nch, int* P) {
    const int CQ = C >> 2, CQB = CQ < 256 ? CQ : 256, PY = 256 / CQB;
    int p = GN3B_U * PY;
    if (p > HW) p = HW;
    *P = p;
    *nch = (HW + p - 1) / p;
}
extern "C" long long ief_groupnorm_bwd_f32_ws_floats(int B, int HW, int C) {
    int nch, P;
    gn3b_chunks(HW, C, &nch, &P);
    return ief_groupnorm_f32_ws_floats(B, HW, C) + 2ll * B * nch * C + 4ll * B * C;
}
__device__ __forceinline__ float dsilu_z(float z) {
    const float sg = 1.0f / (1.0f + expf(-z));
    return sg * (1.0f + z * (1.0f - sg));
}
// MODE 0: partial sums of g and g xh per (image, run, channel); MODE 1: the result (mom = per-group means of g and g xh)
template <int MODE>
__global__ __launch_bounds__(256) void gn3_bwd_kernel(const float* __restrict__ x, const float* __restrict__ x2, int C1, int C2,
                                                      const float* __restrict__ dy, const float* __restrict__ add, int HW, int P,
                                                      const float* __restrict__ gstats, const float* __restrict__ mom, int groups,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                                                      float* __restrict__ p1, float* __restrict__ p2, float* __restrict__ dx,
                                                      float* __restrict__ dx2) {
    __shared__ f32x4 red[256];
    const int C = C1 + C2, CQ = C >> 2, cpg = C / groups;
    const int CQB = CQ < 256 ? CQ : 256, PY = 256 / CQB;
    const int tid = threadIdx.x, cq = tid % CQB, py = tid / CQB;
    const bool lane_ok = py < PY;
    const int chunk = blockIdx.x, b = blockIdx.y, nch = gridDim.x;
    const int p0 = chunk * P, pn = min(P, HW - p0);
    for (int qb = 0; qb * CQB < CQ; ++qb) {
        const int q = qb * CQB + cq;
        const bool ok = lane_ok && q < CQ;
        const int c = ok ? q * 4 : 0;
        int cs = 4;
        const float* src = gn3_src(x, x2, C1, C2, b, HW, c, &cs) + (long long)p0 * cs;
        const float* dsrc = dy + ((long long)b * HW + p0) * C + c;
        f32x4 mu, rs, m1 = {0.f, 0.f, 0.f, 0.f}, m2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = (c + j) / cpg;
            mu[j] = gstats[((long long)b * groups + g) * 2];
            rs[j] = gstats[((long long)b * groups + g) * 2 + 1];
            if (MODE == 1) { m1[j] = mom[((long long)b * groups + g) * 2]; m2[j] = mom[((long long)b * groups + g) * 2 + 1]; }
        }
        const f32x4 ga = *(const f32x4*)(gamma + c), be = *(const f32x4*)(beta + c);
        f32x4 v[GN3B_U], d[GN3B_U];
#pragma unroll
        for (int u = 0; u < GN3B_U; ++u) {
            const int pp = py + u * PY;
            const bool in = ok && pp < pn;
            v[u] = in ? *(const f32x4*)(src + (long long)pp * cs) : f32x4{0.f, 0.f, 0.f, 0.f};
            d[u] = in ? *(const f32x4*)(dsrc + (long long)pp * C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < GN3B_U; ++u) {
            const int pp = py + u * PY;
            if (!(ok && pp < pn)) continue;
            const f32x4 xh = (v[u] - mu) * rs;
            f32x4 gv = d[u] * ga;
            if (silu) {
#pragma unroll
                for (int j = 0; j < 4; ++j) gv[j] *= dsilu_z(xh[j] * ga[j] + be[j]);
            }
            if (MODE == 0) { s1 += gv; s2 += gv * xh; }
            else {
                f32x4 r = rs * (gv - m1 - xh * m2);
                if (add) r += *(const f32x4*)(add + ((long long)b * HW + p0 + pp) * C + c);
                float* dst = c < C1 ? dx + ((long long)b * HW + p0 + pp) * C1 + c : dx2 + ((long long)b * HW + p0 + pp) * C2 + (c - C1);
                *(f32x4*)dst = r;
            }
        }
        if (MODE == 0) {
            __syncthreads();
            red[tid] = s1;
            __syncthreads();
            f32x4 t1 = {0.f, 0.f, 0.f, 0.f};
            if (ok && py == 0) for (int y = 0; y < PY; ++y) t1 += red[y * CQB + cq];       // fixed order
            __syncthreads();
            red[tid] = s2;
            __syncthreads();
            if (ok && py == 0) {
                f32x4 t2 = {0.f, 0.f, 0.f, 0.f};
                for (int y = 0; y < PY; ++y) t2 += red[y * CQB + cq];
                const long long o = ((long long)b * nch + chunk) * C + c;
                *(f32x4*)(p1 + o) = t1;
                *(f32x4*)(p2 + o) = t2;
            }
        }
    }
}
// per (image, group): mom = (sum p1, sum p2) / (HW cpg), runs x channels in a fixed order
__global__ __launch_bounds__(256) void gn3_bwd_finalize_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int C, int HW,
                                                               int nch, int groups, float* __restrict__ mom) {
    __shared__ float sa[4], sb[4];
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups, cpg = C / groups, tid = threadIdx.x, lane = tid & 63;
    const int cl = tid % cpg, r0 = tid / cpg, rstep = 256 / cpg;
    float a = 0.f, q = 0.f;
    if (r0 < rstep) {
        const float* pa = p1 + (long long)b * nch * C + g * cpg + cl;
        const float* pq = p2 + (long long)b * nch * C + g * cpg + cl;
        for (int ch = r0; ch < nch; ch += rstep) { a += pa[(long long)ch * C]; q += pq[(long long)ch * C]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); q += __shfl_xor(q, off); }
    if (lane == 0) { sa[tid >> 6] = a; sb[tid >> 6] = q; }
    __syncthreads();
    if (tid == 0) {
        const float inv = 1.0f / ((float)HW * (float)cpg);
        mom[(long long)blockIdx.x * 2] = ((sa[0] + sa[1]) + (sa[2] + sa[3])) * inv;
        mom[(long long)blockIdx.x * 2 + 1] = ((sb[0] + sb[1]) + (sb[2] + sb[3])) * inv;
    }
}
extern "C" int ief_groupnorm_bwd_f32_ws(const float* x, const float* x2, int C1, int C2, const float* dy, const float* add, float* dx,
                                        float* dx2, const float* gamma, const float* beta, int B, int HW, int groups, float eps,
                                        int silu, float* ws, long long ws_floats, void* stream) {
    if (!x || !dy || !dx || !gamma || !beta || !ws || (C2 > 0 && (!x2 || !dx2))) return IEF_EINVAL;
    if (B <= 0 || HW <= 0 || groups <= 0 || C1 <= 0 || C2 < 0 || (C1 + C2) % groups || (C1 & 3) || (C2 & 3)) return IEF_ESHAPE;
    const int C = C1 + C2;
    if (C / groups > 256) return IEF_ESHAPE;
    if (ws_floats < ief_groupnorm_bwd_f32_ws_floats(B, HW, C)) return IEF_EINVAL;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)ws | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)(dx2 ? dx2 : dx) |
         (uintptr_t)(add ? add : dy) | (uintptr_t)gamma | (uintptr_t)beta) & 15) return IEF_EALIGN;
    int nch, P, nchb, Pb;
    gn3_chunks(HW, C, &nch, &P);
    gn3b_chunks(HW, C, &nchb, &Pb);
    float* pmean = ws;
    float* pm2 = pmean + (long long)B * nch * C;
    float* scale = pm2 + (long long)B * nch * C;
    float* shift = scale + (long long)B * C;
    float* q1 = shift + (long long)B * C;
    float* q2 = q1 + (long long)B * nchb * C;
    float* gstats = q2 + (long long)B * nchb * C;
    float* mom = gstats + 2ll * B * groups;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn3_stats_kernel, dim3(nch, B), dim3(256), 0, st, x, x2, C1, C2, HW, P, pmean, pm2);
    IEF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn3_finalize_kernel, dim3(B * groups), dim3(256), 0, st, pmean, pm2, C, HW, P, nch, groups, eps, gamma, beta,
                       scale, shift, gstats);
    IEF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn3_bwd_kernel<0>, dim3(nchb, B), dim3(256), 0, st, x, x2, C1, C2, dy, add, HW, Pb, gstats, (const float*)nullptr,
                       groups, gamma, beta, silu, q1, q2, dx, dx2);
    IEF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn3_bwd_finalize_kernel, dim3(B * groups), dim3(256), 0, st, q1, q2, C, HW, nchb, groups, mom);
    IEF_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn3_bwd_kernel<1>, dim3(nchb, B), dim3(256), 0, st, x, x2, C1, C2, dy, add, HW, Pb, gstats, mom, groups, gamma,
                       beta, silu, q1, q2, dx, dx2);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

extern "C" int ief_groupnorm_silu_f32(const float* x, const float* x2, int C1, int C2, float* out, const float* gamma,
                                      const float* beta, int B, int HW, int groups, float eps, int silu, void* stream) {
    if (!x || !out || !gamma || !beta || (C2 > 0 && !x2)) return IEF_EINVAL;
    if (B <= 0 || HW <= 0 || groups <= 0 || C1 <= 0 || C2 < 0 || (C1 + C2) % groups) return IEF_ESHAPE;
    const int cpg = (C1 + C2) / groups;
    if ((cpg & 1) || (C1 & 1) || (cpg >> 1) > 256) {
        hipLaunchKernelGGL(groupnorm_f32_scalar_kernel, dim3(B * groups), dim3(256), 0, (hipStream_t)stream, x, x2, C1, C2, out,
                           gamma, beta, HW, groups, eps, silu);
        IEF_LAUNCH_CHECK();
        return IEF_OK;
    }
    const int cp2 = cpg >> 1;
    int PY = 512 / cp2;
    if (PY > HW) PY = HW;
    int threads = ((cp2 * PY + 63) / 64) * 64;
    if (threads > 512) { PY -= 1; threads = ((cp2 * PY + 63) / 64) * 64; }
    const int rounds = (HW + PY * 8 - 1) / (PY * 8);
    int KS = 1;
    while (KS * 2 <= rounds && B * groups * KS < 1024) KS *= 2;
    hipLaunchKernelGGL(groupnorm_f32_kernel, dim3(B * groups * KS), dim3(threads), 0, (hipStream_t)stream, x, x2, C1, C2, out,
                       gamma, beta, HW, groups, eps, silu, PY, KS);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// one launch (KS workgroups per (image, group)) writing operand planes: the small levels, where three row-streaming launches are
// three dispatch latencies for a few hundred KB
extern "C" int ief_groupnorm_silu_x3p_small(const float* x, const float* x2, int C1, int C2, ief_half* outp, long long plane,
                                            const float* gamma, const float* beta, int B, int HW, int groups, float eps, int silu,
                                            void* stream) {
    if (!x || !outp || !gamma || !beta || (C2 > 0 && !x2)) return IEF_EINVAL;
    if (B <= 0 || HW <= 0 || groups <= 0 || C1 <= 0 || C2 < 0 || (C1 + C2) % groups || (plane & 1)) return IEF_ESHAPE;
    const int cpg = (C1 + C2) / groups;
    if ((cpg & 1) || (C1 & 1) || (cpg >> 1) > 256) return IEF_ESHAPE;
    if ((((uintptr_t)x | (uintptr_t)(x2 ? x2 : x)) & 7) || ((uintptr_t)outp & 3)) return IEF_EALIGN;
    const int cp2 = cpg >> 1;
    int PY = 512 / cp2;
    if (PY > HW) PY = HW;
    int threads = ((cp2 * PY + 63) / 64) * 64;
    if (threads > 512) { PY -= 1; threads = ((cp2 * PY + 63) / 64) * 64; }
    const int rounds = (HW + PY * 8 - 1) / (PY * 8);
    int KS = 1;
    while (KS * 2 <= rounds && B * groups * KS < 1024) KS *= 2;
    hipLaunchKernelGGL(groupnorm_f32_kernel, dim3(B * groups * KS), dim3(threads), 0, (hipStream_t)stream, x, x2, C1, C2, (float*)nullptr,
                       gamma, beta, HW, groups, eps, silu, PY, KS, (half_t*)outp, plane);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// ---- GroupNorm (+SiLU) with the (image, group) slab held in REGISTERS: one workgroup of 1024 threads per (image, group) reads its
// HW x cpg slab ONCE (every load requested before the first use), takes the mean and the centred second moment from registers
// (two-pass: no cancellation), normalises and writes fp32 and / or operand planes.  For SMALL BATCHES: at UNet batch 1 (DDIM
// inversion, the null-text loop) every GroupNorm of the step ran the KS-workgroup kernel above, whose workgroups stream the slab
// three times with dependent 8-byte loads: 23 us per call whatever the size, 59 calls = 16 % of the batch-1 forward.  At batch 4 the
// row-streaming forms are as fast (a workgroup per group reads 40-byte slices of 1280-byte pixel rows: 1.23 vs 1.28 ms per step)
// and keep the step.  Slabs of up to 40 Ki floats (20 channel pairs per thread: a 32-pair build spilled hundreds of bytes per lane
// at 1024 threads per workgroup and paid ~190 us of scratch set-up per launch on this runtime).
template <int E>
__global__ __launch_bounds__(1024) void groupnorm_reg_kernel(const float* __restrict__ x, const float* __restrict__ x2, int C1, int C2,
                                                             float* __restrict__ out, half_t* __restrict__ outp, long long plane,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int HW,
                                                             int groups, float eps, int silu) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    __shared__ float red[16];
    __shared__ float bc;
    __shared__ float sc_s[512], sh_s[512];
    const int C = C1 + C2, cpg = C / groups, cp2 = cpg >> 1;
    const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
    const int npairs = HW * cp2;
    const int tid = threadIdx.x;
    const int dq = 1024 / cp2, dr = 1024 - dq * cp2;       // (pixel, pair) advance of 1024 elements
    int p = tid / cp2, j = tid - p * cp2;
    f32x2 v[E];
    const int p00 = p, j00 = j;
    const float* xb = x + (long long)b * HW * C1;
    const float* x2b = x2 ? x2 + (long long)b * HW * C2 : x;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const bool ok = tid + k * 1024 < npairs;               // slots past the slab re-read pair 0 and are zeroed: no branch around a load
        const int c = g * cpg + (ok ? 2 * j : 0), pp = ok ? p : 0;
        const float* src = c < C1 ? xb + (unsigned)(pp * C1 + c) : x2b + (unsigned)(pp * C2 + (c - C1));
        const f32x2 t = *(const f32x2*)src;
        v[k] = ok ? t : (f32x2){0.f, 0.f};
        p += dq; j += dr;
        if (j >= cp2) { j -= cp2; ++p; }
        if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // addresses are formed four at a time (they are dead once the load is out): the
    }                                                            // scheduler otherwise forms all E first and spills
    auto block_sum = [&](float t) -> float {
        t = wave_sum(t);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = t;
        __syncthreads();
        if (tid == 0) { float a = 0.f; for (int w = 0; w < 16; ++w) a += red[w]; bc = a; }
        __syncthreads();
        return bc;
    };
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) s += v[k][0] + v[k][1];      // slots past the slab hold zeros
    const float n = (float)cpg * (float)HW;
    const float mean = block_sum(s) / n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        if (tid + k * 1024 < npairs) { const float d0 = v[k][0] - mean, d1 = v[k][1] - mean; q += d0 * d0 + d1 * d1; }
    }
    const float rstd = 1.0f / sqrtf(block_sum(q) / n + eps);
    if (tid < cpg) {
        const float sc = rstd * gamma[g * cpg + tid];
        sc_s[tid] = sc;
        sh_s[tid] = beta[g * cpg + tid] - mean * sc;
    }
    __syncthreads();
    p = p00; j = j00;
    float* ob = out ? out + (long long)b * HW * C : nullptr;
    half_t* opb = outp ? outp + (long long)b * HW * C : nullptr;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        const int idx = tid + k * 1024;
        if (idx < npairs) {
            float y0 = v[k][0] * sc_s[2 * j] + sh_s[2 * j], y1 = v[k][1] * sc_s[2 * j + 1] + sh_s[2 * j + 1];
            if (silu) { y0 = silu_x(y0); y1 = silu_x(y1); }
            const unsigned o = (unsigned)(p * C + g * cpg + 2 * j);
            if (out) *(f32x2*)(ob + o) = (f32x2){y0, y1};
            if (outp) {          // operand planes (csrc/gemm_x3p.hip): hi = fp16(y), lo = fp16(y - hi)
                const half2_t h = __builtin_convertvector((f32x2){y0, y1}, half2_t);
                const half2_t l = __builtin_convertvector((f32x2){y0 - (float)h[0], y1 - (float)h[1]}, half2_t);
                *(half2_t*)(opb + o) = h;
                *(half2_t*)(opb + plane + o) = l;
            }
        }
        p += dq; j += dr;
        if (j >= cp2) { j -= cp2; ++p; }
        if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}
// 1: the register-resident form takes this shape (channels per group even and <= 512, C1 even, HW x channels per group <= 40960)
extern "C" int ief_groupnorm_reg_fits(int C1, int C2, int HW, int groups) {
    if (groups <= 0 || C1 <= 0 || C2 < 0 || (C1 + C2) % groups) return 0;
    const int cpg = (C1 + C2) / groups;
    return !(cpg & 1) && !(C1 & 1) && cpg <= 512 && (long long)HW * cpg <= 40960;
}
extern "C" int ief_groupnorm_silu_reg(const float* x, const float* x2, int C1, int C2, float* out, ief_half* outp, long long plane,
                                      const float* gamma, const float* beta, int B, int HW, int groups, float eps, int silu,
                                      void* stream) {
    if (!x || (!out && !outp) || !gamma || !beta || (C2 > 0 && !x2)) return IEF_EINVAL;
    if (B <= 0 || HW <= 0 || !ief_groupnorm_reg_fits(C1, C2, HW, groups) || (plane & 1)) return IEF_ESHAPE;
    if ((((uintptr_t)x | (uintptr_t)(x2 ? x2 : x) | (uintptr_t)(out ? out : x)) & 7) || ((uintptr_t)outp & 3)) return IEF_EALIGN;
    const int cpg = (C1 + C2) / groups;
    const int need = (HW * (cpg >> 1) + 1023) / 1024;
    const dim3 grid(B * groups), blk(1024);
    hipStream_t st = (hipStream_t)stream;
#define GNR_GO(E_) hipLaunchKernelGGL(groupnorm_reg_kernel<E_>, grid, blk, 0, st, x, x2, C1, C2, out, (half_t*)outp, plane, gamma, beta, HW, groups, eps, silu)
    if (need <= 4) GNR_GO(4);
    else if (need <= 8) GNR_GO(8);
    else if (need <= 16) GNR_GO(16);
    else GNR_GO(20);
#undef GNR_GO
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// LayerNorm over rows of C fp32, one wave per row, two-pass statistics
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            long long rows, int C, float eps) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* r = x + row * C;
    float* o = out + row * C;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int i = lane; i < C; i += 64) s += r[i];
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int i = lane; i < C; i += 64) { const float d = r[i] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    for (int i = lane; i < C; i += 64) o[i] = (r[i] - mean) * rstd * gamma[i] + beta[i];
}
// C % 4 == 0, C <= 256 NV: the row stays in registers (NV 16-byte pieces per lane): one read, one write
template <int NV>
__global__ __launch_bounds__(256) void layernorm_f32_vec_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                long long rows, int C, float eps, half_t* __restrict__ outp = nullptr,
                                                                long long plane = 0) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* r = x + row * C;
    float* o = out + row * C;
    const int lane = threadIdx.x & 63, CQ = C >> 2;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int qd = lane + 64 * j;
        v[j] = qd < CQ ? *(const f32x4*)(r + qd * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        if (lane + 64 * j < CQ) {
            const f32x4 d = v[j] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int qd = lane + 64 * j;
        if (qd < CQ) {
            const f32x4 y = (v[j] - mean) * rstd * *(const f32x4*)(gamma + qd * 4) + *(const f32x4*)(beta + qd * 4);
            if (out) *(f32x4*)(o + qd * 4) = y;
            if (outp) {          // operand planes of the split-operand mode (the consumer GEMM stages them by LDS-DMA)
                half4 h, l;
                split4(y, 1.0f, h, l);
                half_t* op = outp + row * C + qd * 4;
                *(half4*)op = h;
                *(half4*)(op + plane) = l;
            }
        }
    }
}
extern "C" int ief_layernorm_x3p(const float* x, ief_half* outp, long long plane, const float* gamma, const float* beta, long long rows,
                                 int C, float eps, void* stream) {
    if (!x || !outp || !gamma || !beta) return IEF_EINVAL;
    if (rows <= 0 || C <= 0 || (C & 3) || C > 2560 || (plane & 3)) return IEF_ESHAPE;
    if ((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15) || ((uintptr_t)outp & 7)) return IEF_EALIGN;
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    float* none = nullptr;
    if (C <= 512) hipLaunchKernelGGL(layernorm_f32_vec_kernel<2>, grid, dim3(256), 0, st, x, none, gamma, beta, rows, C, eps, (half_t*)outp, plane);
    else if (C <= 1280) hipLaunchKernelGGL(layernorm_f32_vec_kernel<5>, grid, dim3(256), 0, st, x, none, gamma, beta, rows, C, eps, (half_t*)outp, plane);
    else hipLaunchKernelGGL(layernorm_f32_vec_kernel<10>, grid, dim3(256), 0, st, x, none, gamma, beta, rows, C, eps, (half_t*)outp, plane);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
extern "C" int ief_layernorm_f32(const float* x, float* out, const float* gamma, const float* beta, long long rows, int C,
                                 float eps, void* stream) {
    if (!x || !out || !gamma || !beta) return IEF_EINVAL;
    if (rows <= 0 || C <= 0) return IEF_ESHAPE;
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (C & 3) == 0 && ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0);
    if (vec && C <= 512) hipLaunchKernelGGL(layernorm_f32_vec_kernel<2>, grid, dim3(256), 0, st, x, out, gamma, beta, rows, C, eps);
    else if (vec && C <= 1280) hipLaunchKernelGGL(layernorm_f32_vec_kernel<5>, grid, dim3(256), 0, st, x, out, gamma, beta, rows, C, eps);
    else if (vec && C <= 2560) hipLaunchKernelGGL(layernorm_f32_vec_kernel<10>, grid, dim3(256), 0, st, x, out, gamma, beta, rows, C, eps);
    else hipLaunchKernelGGL(layernorm_f32_kernel, grid, dim3(256), 0, st, x, out, gamma, beta, rows, C, eps);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// --------------------------------------------------------------------------------------------- elementwise
// which: 0 add (a + b), 1 silu(a), 2 GEGLU on the interleaved FF1 layout [8 hidden | 8 gate] groups: out[.., Ch]
__global__ __launch_bounds__(256) void ew_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                     long long n, int which, int Ch) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (which == 0) out[i] = a[i] + b[i];
        else if (which == 1) out[i] = silu_x(a[i]);
        else {
            const long long row = i / Ch;
            const int c = (int)(i - row * Ch);
            const float* pr = a + row * 2 * Ch + (c >> 3) * 16 + (c & 7);
            out[i] = pr[0] * gelu_f(pr[8]);
        }
    }
}
// GEGLU on the interleaved FF1 layout, 16-byte accesses: output quad (row, c .. c+3) = hidden quad x gelu(gate quad), the
// gate 8 floats behind the hidden values inside their group of 16
__global__ __launch_bounds__(256) void geglu_il_f32_vec_kernel(const float* __restrict__ pre, float* __restrict__ out, long long nq, int Chq) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long long)gridDim.x * 256) {
        const long long row = i / Chq;
        const int cq = (int)(i - row * Chq);             // output quad inside the row: columns 4 cq .. 4 cq + 3
        const float* pr = pre + row * (8ll * Chq) + (cq >> 1) * 16 + (cq & 1) * 4;
        const f32x4 hv = *(const f32x4*)pr, gv = *(const f32x4*)(pr + 8);
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = hv[j] * gelu_f(gv[j]);
        *(f32x4*)(out + i * 4) = y;
    }
}
static int launch_ew(const float* a, const float* b, float* out, long long n, int which, int Ch, void* stream) {
    if (!a || !out || (which == 0 && !b)) return IEF_EINVAL;
    if (n <= 0) return IEF_ESHAPE;
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(ew_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, out, n, which, Ch);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
extern "C" int ief_add_f32(const float* a, const float* b, float* out, long long n, void* stream) { return launch_ew(a, b, out, n, 0, 1, stream); }
extern "C" int ief_silu_f32(const float* x, float* out, long long n, void* stream) { return launch_ew(x, nullptr, out, n, 1, 1, stream); }
extern "C" int ief_geglu_il_f32(const float* pre, float* out, long long rows, int Ch, void* stream) {
    if (Ch <= 0 || (Ch & 7)) return IEF_ESHAPE;
    if (pre && out && rows > 0 && (((uintptr_t)pre | (uintptr_t)out) & 15) == 0) {
        const long long nq = rows * (Ch / 4);
        long long grid = (nq + 255) / 256;
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(geglu_il_f32_vec_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, pre, out, nq, Ch / 4);
        IEF_LAUNCH_CHECK();
        return IEF_OK;
    }
    return launch_ew(pre, nullptr, out, rows * Ch, 2, Ch, stream);
}

__global__ void timestep_embedding_f32_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim) {
    const int half_dim = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half_dim) return;
    const int b = i / half_dim, k = i - b * half_dim;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half_dim);  // ln(10000)
    const float a = t[b] * freq;
    out[(long long)b * dim + k] = cosf(a);
    out[(long long)b * dim + half_dim + k] = sinf(a);
}
extern "C" int ief_timestep_embedding_f32(const float* t, float* out, int B, int dim, void* stream) {
    if (!t || !out) return IEF_EINVAL;
    if (B <= 0 || dim <= 0 || (dim & 1)) return IEF_ESHAPE;
    const int n = B * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, B, dim);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// out[b][:] = in[src[b]][:] for rows of `row_elems` floats (row_elems % 4 == 0)
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out,
                                                              const int* __restrict__ src, int B, long long row4) {
    const long long total = (long long)B * row4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / row4);
        out[i] = in[(long long)src[b] * row4 + (i - (long long)b * row4)];
    }
}
extern "C" int ief_gather_rows_f32(const float* in, float* out, const int* src, int B, long long row_elems, void* stream) {
    if (!in || !out || !src) return IEF_EINVAL;
    if (B <= 0 || row_elems <= 0 || (row_elems & 3)) return IEF_ESHAPE;
    if (((uintptr_t)in | (uintptr_t)out) & 15) return IEF_EALIGN;
    const long long total = (long long)B * (row_elems / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(gather_rows_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const f32x4*)in, (f32x4*)out, src,
                       B, row_elems / 4);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// --------------------------------------------------------------------------------------------- boundary convolutions
// conv_in:  fp32 NCHW [B,Cin<=8,H,W] -> fp32 NHWC [B,H,W,Cout]; w fp32 [3][3][Cin][Cout].  One thread per (pixel, 4 channels).
// CIN > 0: the channel count is a compile-time constant and every load of a tap ROW (3 taps x CIN activations, 3 x CIN weight
// vectors) is requested before the first is used -- the plain loop below issued 9 x Cin dependent load pairs per thread (40 us for
// the 0.38 GFLOP of SD's 4 -> 320 at 64 x 64, batch 4).  The accumulation order (tap, channel) is the loop's: same bits.
template <int CIN>
__global__ __launch_bounds__(256) void conv_in_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, int B, int Cin,
                                                          int H, int Wd, int Cout) {
    const int C4 = Cout >> 2;
    const long long total = (long long)B * H * Wd * C4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long pix = i / C4;
        const int c4 = (int)(i - pix * C4);
        const int b = (int)(pix / (H * Wd));
        const int rem = (int)(pix - (long long)b * H * Wd);
        const int oy = rem / Wd, ox = rem - oy * Wd;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (CIN > 0) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy + ky - 1;
                const bool yok = (unsigned)iy < (unsigned)H;
                float v[3][CIN];
                f32x4 wv[3][CIN];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox + kx - 1;
                    const bool ok = yok && (unsigned)ix < (unsigned)Wd;
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) {
                        v[kx][ci] = ok ? x[(((long long)b * CIN + ci) * H + iy) * Wd + ix] : 0.f;
                        wv[kx][ci] = *(const f32x4*)(w + ((long long)((ky * 3 + kx) * CIN + ci)) * Cout + c4 * 4);
                    }
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox + kx - 1;
                    if (!(yok && (unsigned)ix < (unsigned)Wd)) continue;       // a padded tap adds nothing (as the plain loop: skipped, not + 0)
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] += v[kx][ci] * wv[kx][ci][e];
                }
            }
        } else {
            for (int tap = 0; tap < 9; ++tap) {
                const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
                if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)Wd) continue;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float v = x[(((long long)b * Cin + ci) * H + iy) * Wd + ix];
                    const f32x4 wv = *(const f32x4*)(w + ((long long)(tap * Cin + ci)) * Cout + c4 * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += v * wv[e];
                }
            }
        }
        if (bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += bias[c4 * 4 + e];
        }
        *(f32x4*)(out + pix * Cout + c4 * 4) = acc;
    }
}
extern "C" int ief_conv_in_f32act(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H, int Wd,
                                  int Cout, void* stream) {
    if (!x || !w || !out) return IEF_EINVAL;
    if (B <= 0 || H <= 0 || Wd <= 0 || Cin <= 0 || Cin > 8 || Cout <= 0 || (Cout & 3)) return IEF_ESHAPE;
    const long long total = (long long)B * H * Wd * (Cout / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (Cin == 4) hipLaunchKernelGGL(conv_in_f32_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, B, Cin, H, Wd, Cout);
    else hipLaunchKernelGGL(conv_in_f32_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, B, Cin, H, Wd, Cout);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// conv_out: fp32 NHWC [B,H,W,C] -> fp32 NCHW [B,Cout<=8,H,W]; w fp32 [Cout][3][3][C]; 16 lanes per output pixel split 9*C
__global__ __launch_bounds__(256) void conv_out_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int B, int C,
                                                           int H, int Wd, int Cout) {
    const int sub = threadIdx.x & 15;
    const long long pix = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long long total = (long long)B * H * Wd;
    const bool live = pix < total;
    const long long pc = live ? pix : 0;
    const int b = (int)(pc / (H * Wd));
    const int rem = (int)(pc - (long long)b * H * Wd);
    const int oy = rem / Wd, ox = rem - oy * Wd;
    const int C4 = C >> 2;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
    if (live) {
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)Wd) continue;
            const float* xp = x + (((long long)b * H + iy) * Wd + ix) * C;
            for (int c4 = sub; c4 < C4; c4 += 16) {
                const f32x4 v = *(const f32x4*)(xp + c4 * 4);
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    if (o < Cout) {
                        const f32x4 wv = *(const f32x4*)(w + ((long long)o * 9 + tap) * C + c4 * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[o] += v[e] * wv[e];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc[o] += __shfl_xor(acc[o], off);
    }
    if (live && sub == 0) {
        for (int o = 0; o < Cout; ++o) out[(((long long)b * Cout + o) * H + oy) * Wd + ox] = acc[o] + (bias ? bias[o] : 0.f);
    }
}
// The same with the weights ([Cout][9][C] fp32: 46 KB for SD's 320 -> 4) staged in LDS once per workgroup and, per pixel, the
// activation chunks of a whole tap ROW (3 taps x CH chunks of 4 channels per lane) requested before the first is used: the loop
// above issues one dependent activation + Cout weight loads per chunk (45 round trips per pixel: 65 us for 0.38 GFLOP).  Per lane
// and output channel the products are added in the loop's order (tap, chunk, element): same bits.
template <int CH>            // chunks of 4 channels per lane and tap: C <= 64 CH
__global__ __launch_bounds__(256) void conv_out_f32_lds_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ out, int B, int C,
                                                               int H, int Wd, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float wl_f32[];          // [Cout][9][C]
    const int sub = threadIdx.x & 15;
    const long long pix = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long long total = (long long)B * H * Wd;
    const bool live = pix < total;
    const long long pc = live ? pix : 0;
    const int b = (int)(pc / (H * Wd));
    const int rem = (int)(pc - (long long)b * H * Wd);
    const int oy = rem / Wd, ox = rem - oy * Wd;
    const int C4 = C >> 2;
    const int nw4 = (Cout * 9 * C) >> 2;
    for (int i = threadIdx.x; i < nw4; i += 256) ((f32x4*)wl_f32)[i] = ((const f32x4*)w)[i];
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + ky - 1;
        const bool yok = live && (unsigned)iy < (unsigned)H;
        f32x4 v[3][CH];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox + kx - 1;
            const bool ok = yok && (unsigned)ix < (unsigned)Wd;
            const float* xp = x + (((long long)b * H + (ok ? iy : 0)) * Wd + (ok ? ix : 0)) * C;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int c4 = sub + 16 * j;
                v[kx][j] = (ok && c4 < C4) ? *(const f32x4*)(xp + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox + kx - 1;
            if (!(yok && (unsigned)ix < (unsigned)Wd)) continue;               // a padded tap adds nothing (skipped, as in the loop above)
            const int tap = ky * 3 + kx;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int c4 = sub + 16 * j;
                if (c4 < C4) {
#pragma unroll
                    for (int o = 0; o < 8; ++o) {
                        if (o < Cout) {
                            const f32x4 wv = *(const f32x4*)(wl_f32 + (o * 9 + tap) * C + c4 * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[o] += v[kx][j][e] * wv[e];
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc[o] += __shfl_xor(acc[o], off);
    }
    if (live && sub == 0) {
        for (int o = 0; o < Cout; ++o) out[(((long long)b * Cout + o) * H + oy) * Wd + ox] = acc[o] + (bias ? bias[o] : 0.f);
    }
}
extern "C" int ief_conv_out_f32act(const float* x, const float* w, const float* bias, float* out, int B, int C, int H, int Wd,
                                   int Cout, void* stream) {
    if (!x || !w || !out) return IEF_EINVAL;
    if (B <= 0 || H <= 0 || Wd <= 0 || C <= 0 || (C & 3) || Cout <= 0 || Cout > 8) return IEF_ESHAPE;
    const long long total = (long long)B * H * Wd;
    const size_t lds = (size_t)Cout * 9 * C * sizeof(float);
    const dim3 grid((unsigned)((total + 15) / 16));
    if (lds <= 48 * 1024 && C <= 320 && Cout <= 4) {
        if (C <= 128) hipLaunchKernelGGL(conv_out_f32_lds_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias, out, B, C, H, Wd, Cout);
        else hipLaunchKernelGGL(conv_out_f32_lds_kernel<5>, grid, dim3(256), lds, (hipStream_t)stream, x, w, bias, out, B, C, H, Wd, Cout);
    } else {
        hipLaunchKernelGGL(conv_out_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, out, B, C, H, Wd, Cout);
    }
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// uint8 image epilogue of `latent2image` (`/root/reference/p2p/model/sd_utils.py:85-88`): decoder output fp32 NCHW in [-1, 1]
// -> uint8 NHWC, (x / 2 + 0.5).clamp(0, 1) * 255 truncated, so a quarter of the bytes cross PCIe
__global__ __launch_bounds__(256) void image_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int B, int C, int H,
                                                       int Wd) {
    const long long total = (long long)B * H * Wd * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long pix = i / C;
        const int b = (int)(pix / ((long long)H * Wd));
        const long long rem = pix - (long long)b * H * Wd;
        float v = x[((long long)b * C + c) * H * Wd + rem] / 2.0f + 0.5f;
        v = fminf(fmaxf(v, 0.f), 1.f);
        out[i] = (uint8_t)(v * 255.0f);
    }
}
extern "C" int ief_image_u8(const float* x, unsigned char* out, int B, int C, int H, int Wd, void* stream) {
    if (!x || !out) return IEF_EINVAL;
    if (B <= 0 || C <= 0 || H <= 0 || Wd <= 0) return IEF_ESHAPE;
    const long long total = (long long)B * H * Wd * C;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(image_u8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, out, B, C, H, Wd);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}
