// Split-operand contractions ("f16x3"): fp32 storage, every product on THREE fp16 MFMAs.
//
// The reference computes in fp32 (`/root/reference/p2p/edit_syn.py:38`).  The fp32-input MFMA of exact_f32.hip reproduces
// that bit for bit up to summation order, at 1/16 of the fp16 matrix rate.  Here every fp32 operand element x is split
// once, while its tile is staged into LDS, into two fp16 numbers
//      hi = fp16(s x),   lo = fp16(s x - hi)            (s = a power of two, exact)
// so that s x = hi + lo up to 2^-22 |s x| (round-to-nearest twice), and a product of two such operands is
//      A B = (Ah Bh + Al Bh + Ah Bl) / (sa sb)   +   Al Bl / (sa sb)   <- dropped, 2^-22 relative
// three `v_mfma_f32_32x32x16_f16` into ONE fp32 accumulator (products of two fp16 numbers are exact in fp32, so the MFMA
// adds exact terms with fp32 rounding, as the fp32 MFMA does).  ~21 operand bits at 1/3 of the fp16 rate instead of 24 bits
// at 1/16.  The scales keep `lo` out of the fp16 subnormal range (|lo| <= 2^-11 |hi|: with s x >= 2^-3 it is a normal
// number) whatever the MFMA does with subnormal inputs; the host picks them per call (activations 2^4: exact up to 4096,
// weights 2^8, softmax maps 2^14), the epilogue divides them out.  |s x| >= 65504 saturates `hi` (clamped), `lo` then carries the
// excess up to another 65504.
//
//   igemm_x3_kernel<CONV, TRANSB, NT>   same operator set, parameter struct and tile grid as igemm_f32_kernel
//                                       (linear / 1x1 / 3x3 implicit GEMM with concat sources, nearest-2x, stride 2, fused
//                                       1x1 shortcut sources, the two batched attention products, split-K slabs)
//   attn_flash_x3_kernel<D>             fused attention without materialised maps: S^T = K Q^T and O^T += V^T P^T on split
//                                       operands, softmax in fp32 registers
#include "ief_common.h"
#include "ief_params.h"

#define YBM 128
#define YBK 32
#define YLD 40          // halves per LDS row (80 B): b128 fragment reads and b64 staging writes are conflict-free

struct RowCoordY { int b, oy, ox, ok; };

// s x -> (hi, lo) for four consecutive k
__device__ __forceinline__ void split4(const f32x4 v, const float s, half4& hi, half4& lo) {
    const f32x4 x = v * s;
    f32x4 c;
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_fmed3f(x[j], -65504.f, 65504.f);     // hi never overflows to inf
    hi = __builtin_convertvector(c, half4);
    const f32x4 r = x - __builtin_convertvector(hi, f32x4);
    lo = __builtin_convertvector(r, half4);
}

template <bool CONV, bool TRANSB, int NT>
__global__ __launch_bounds__(256, 2) void igemm_x3_kernel(const IefGemmF32Params p) {
    constexpr int YBN = 64 * NT;
    constexpr int ROWS = YBM + YBN;
    // per buffer: [A hi][A lo][B hi][B lo], rows of YLD halves
    __shared__ __attribute__((aligned(16))) half_t smem_y[2 * 2 * ROWS * YLD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int ntn = (p.N + YBN - 1) / YBN;
    const int ntiles = ((p.M + YBM - 1) / YBM) * ntn;
    const int bid = xcd_remap(blockIdx.x, ntiles);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * YBM, n0 = tn * YBN;
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
    // ---- loader assignment: A tile = 128 rows x 8 chunks of 4 floats; thread -> 4 rows (32 apart), one chunk column
    const int a_kc = tid & 7, a_r0 = tid >> 3;
    RowCoordY rc[4];
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
        half_t* ah = smem_y + buf * (2 * ROWS * YLD);
        half_t* al = ah + YBM * YLD;
        half_t* bh = al + YBM * YLD;
        half_t* bl = bh + YBN * YLD;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            half4 h, l;
            split4(ra[i], sa, h, l);
            const int off = (a_r0 + 32 * i) * YLD + a_kc * 4;
            *(half4*)(ah + off) = h;
            *(half4*)(al + off) = l;
        }
        if (!TRANSB) {
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) {
                half4 h, l;
                split4(rb[i], sb, h, l);
                const int off = (a_r0 + 32 * i) * YLD + a_kc * 4;
                *(half4*)(bh + off) = h;
                *(half4*)(bl + off) = l;
            }
        } else {
            const int kr = tid & 31;
#pragma unroll
            for (int i = 0; i < 2 * NT; ++i) {
                half4 h, l;
                split4(rb[i], sb, h, l);
                const int nl = ((tid >> 5) + 8 * i) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bh[(nl + j) * YLD + kr] = h[j];
                    bl[(nl + j) * YLD + kr] = l[j];
                }
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
    // the slabs in slab order and applies the epilogue
    const int nk_all = (K + YBK - 1) / YBK;
    int kt0 = 0, nk = nk_all;
    if (p.splits > 1) {
        const int per = (nk_all + p.splits - 1) / p.splits;
        kt0 = blockIdx.y * per;
        nk = min(per, nk_all - kt0);
        if (nk < 0) nk = 0;
    }
    if (nk > 0) {
        load_tile(kt0 * YBK);
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile((kt0 + kt + 1) * YBK);
        const half_t* ah = smem_y + buf * (2 * ROWS * YLD) + (wm * 64 + li) * YLD + 8 * lh;
        const half_t* al = ah + YBM * YLD;
        const half_t* bh = smem_y + buf * (2 * ROWS * YLD) + 2 * YBM * YLD + (wn * 32 * NT + li) * YLD + 8 * lh;
        const half_t* bl = bh + YBN * YLD;
#pragma unroll
        for (int ks = 0; ks < YBK / 16; ++ks) {
            half8 fah[2], fal[2], fbh[NT], fbl[NT];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                fah[a] = *(const half8*)(ah + a * 32 * YLD + ks * 16);
                fal[a] = *(const half8*)(al + a * 32 * YLD + ks * 16);
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                fbh[b] = *(const half8*)(bh + b * 32 * YLD + ks * 16);
                fbl[b] = *(const half8*)(bl + b * 32 * YLD + ks * 16);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[a], fbh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[a], fbl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[a], fbh[b], acc[a][b], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }
    const float inv = 1.0f / (sa * sb);
    if (p.splits > 1) {       // raw partial sums (already in output units)
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
                    if (m < M) slab[(long long)m * N + n] = acc[a][b][r] * inv;
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
                float v = acc[a][b][r] * inv + bv;
                if (p.rowvec) v += p.rowvec[(long long)(m / p.rows_per_batch) * N + n];
                if (R) v += R[(long long)m * p.ldr + n];
                Out[(long long)m * p.ldo + n] = v * p.out_scale;
            }
        }
}

template <bool CONV, bool TRANSB, int NT>
static int launch_igemm_x3(const IefGemmF32Params& p, hipStream_t st) {
    constexpr int YBN = 64 * NT;
    const int tiles = ((p.M + YBM - 1) / YBM) * ((p.N + YBN - 1) / YBN);
    const int z = p.heads > 0 ? p.batch * p.heads : 1;
    const int splits = p.splits > 1 ? p.splits : 1;
    hipLaunchKernelGGL((igemm_x3_kernel<CONV, TRANSB, NT>), dim3(tiles, splits, z), dim3(256), 0, st, p);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

// called by ief_gemm_f32 (exact_f32.hip) after its argument checks when p.x3 != 0; the split-K reducer launch is the caller's
int ief_gemm_x3_dispatch(const IefGemmF32Params& p, hipStream_t st) {
    const bool wide = ief_gemm_f32_bn(p.N) != 64;
    if (p.conv) return wide ? launch_igemm_x3<true, false, 2>(p, st) : launch_igemm_x3<true, false, 1>(p, st);
    if (p.transb) return wide ? launch_igemm_x3<false, true, 2>(p, st) : launch_igemm_x3<false, true, 1>(p, st);
    return wide ? launch_igemm_x3<false, false, 2>(p, st) : launch_igemm_x3<false, false, 1>(p, st);
}
