// MFMA GEMM and NHWC implicit-GEMM 3x3 convolution for gfx950 (fp16 in, fp32 accumulate).
//
//   Out[m][n] = epilogue( sum_k A(m,k) * W[n][k] )
//
// DENSE : A is [M][K] row-major (linear / 1x1-conv over tokens-major activations).
// CONV  : A(m,k) is gathered on the fly from an NHWC activation: m = (b, oy, ox),
//         k = (ky, kx, c); zero padding 1; stride 1 or 2; optional nearest-2x upsample of the
//         input fused into the gather; optional channel concat of two sources (the UNet's
//         skip connections) fused as two K ranges.  Nothing is materialised (no im2col).
//         W is [Cout][3][3][C1+C2] so both operands are K-major, the layout MFMA fragments want.
//
// Tile: BM x BN x 64, 256 threads = 2x2 waves, v_mfma_f32_16x16x32_f16, LDS double buffer
// with register prefetch of the next K tile (global loads issued before the MFMA block, LDS
// writes after it, one barrier per K tile).  LDS rows are 128 B with the 16-B chunk index
// XOR-ed by (row>>1)&7: conflict-free for the ds_read_b128 fragment reads and the staging writes.
// Epilogue: accumulators -> LDS (fp32) -> rows of 8 outputs per thread: + bias[n]
// + rowvec[batch(m)][n] (time-embedding) + residual[m][n], * scale, one 16-B fp16 store.
//
// Replaces on the reference path (all executed by diffusers/PyTorch eager there):
//   ResnetBlock2D.conv1/conv2 + temb add + skip add   /root/reference/pnp/model/register.py:139-175
//   Attention.to_q/to_k/to_v/to_out                   /root/reference/p2p/model/register.py:33-54
#include "ief_common.h"
#include "ief_params.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
// one wave-instruction: 64 lanes x 16 B from per-lane global addresses -> 1 KiB of LDS at a wave-uniform base
__device__ __forceinline__ void glds16(const half_t* g, half_t* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, bool CONV, bool GLDS>
__global__ __launch_bounds__(256) void igemm_f16_kernel(const IefGemmParams p) {
    constexpr int BK = 64;
    constexpr int NA = BM * 8 / 256, NB = BN * 8 / 256;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    __shared__ __attribute__((aligned(16))) half_t smem[2 * (BM + BN) * BK];
    half_t* As = smem;
    half_t* Bs = smem + 2 * BM * BK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
    const long long z = blockIdx.z;
    const half_t* __restrict__ A = p.A + z * p.strideA;
    const half_t* __restrict__ A2 = p.A2;
    const half_t* __restrict__ Wt = p.W + z * p.strideW;

    const int rbase = tid >> 3;   // 0..31: tile row (+32 i) this thread stages
    // 16-byte chunk of the 128-byte K row this thread FETCHES.  Register staging: chunk tid&7, written to
    // the XOR-swizzled LDS slot.  LDS-DMA: the LDS slot is fixed by the lane (tid&7), so the swizzle is
    // applied to the source chunk instead (same involution; rows rbase+32i share (row>>1)&7).
    const int kc = GLDS ? ((tid & 7) ^ ((rbase >> 1) & 7)) : (tid & 7);
    long long a_off[NA];
    bool a_ok[NA];
    int a_b[NA], a_y[NA], a_x[NA];
    long long w_off[NB];
    bool w_ok[NB];
    const int Hp = p.H >> p.ups, Wp = p.Wd >> p.ups;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + rbase + 32 * i;
        a_ok[i] = m < p.M;
        if constexpr (CONV) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_b[i] = b;
            a_y[i] = oy * p.stride - 1;
            a_x[i] = ox * p.stride - 1;
            a_off[i] = m;  // output pixel index: the address of the fused-1x1 extra K range
        } else {
            a_off[i] = (long long)m * p.lda + kc * 8;
            a_b[i] = a_y[i] = a_x[i] = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + rbase + 32 * i;
        w_ok[i] = n < p.N;
        w_off[i] = (long long)n * p.ldw + kc * 8;
    }

    const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half8 ra[NA], rb[NB];

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if constexpr (CONV) {
            const int Ctot = p.C1 + p.C2;
            if (k0 < 9 * Ctot) {
                const int tap = k0 / Ctot, c0 = k0 - tap * Ctot;
                const int ky = tap / 3, kx = tap - ky * 3;
                const half_t* src = A;
                int cs = p.C1, cc = c0;
                if (c0 >= p.C1) { src = A2; cs = p.C2; cc = c0 - p.C1; }
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                    const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                    const int py = iy >> p.ups, px = ix >> p.ups;
                    const long long off = ((long long)(a_b[i] * Hp + py) * Wp + px) * cs + cc + kc * 8;
                    ra[i] = ok ? *(const half8*)(src + off) : zero8;
                }
            } else {  // fused 1x1 over the extra sources, sampled at the output pixel
                const int ke = k0 - 9 * Ctot;
                const half_t* src = p.E1;
                int cs = p.CE1, cc = ke;
                if (ke >= p.CE1) { src = p.E2; cs = p.CE2; cc = ke - p.CE1; }
#pragma unroll
                for (int i = 0; i < NA; ++i)
                    ra[i] = a_ok[i] ? *(const half8*)(src + a_off[i] * cs + cc + kc * 8) : zero8;
            }
        } else {
            const bool kok = k0 + kc * 8 < p.K;
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = (a_ok[i] && kok) ? *(const half8*)(A + a_off[i] + k0) : zero8;
        }
        const bool kok = k0 + kc * 8 < p.K;
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = (w_ok[i] && kok) ? *(const half8*)(Wt + w_off[i] + k0) : zero8;
    };
    // LDS-DMA staging of K tile kt into buffer buf (asynchronous; tracked by vmcnt)
    auto stage_tile = [&](int buf, int kt) {
        const int k0 = kt * BK;
        const half_t* zp = p.zeros;
        half_t* la = As + buf * BM * BK + (wave * 8) * BK;
        half_t* lb = Bs + buf * BN * BK + (wave * 8) * BK;
        if constexpr (CONV) {
            const int Ctot = p.C1 + p.C2;
            if (k0 < 9 * Ctot) {
                const int tap = k0 / Ctot, c0 = k0 - tap * Ctot;
                const int ky = tap / 3, kx = tap - ky * 3;
                const half_t* src = A;
                int cs = p.C1, cc = c0;
                if (c0 >= p.C1) { src = A2; cs = p.C2; cc = c0 - p.C1; }
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                    const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.Wd;
                    const int py = iy >> p.ups, px = ix >> p.ups;
                    const long long off = ((long long)(a_b[i] * Hp + py) * Wp + px) * cs + cc + kc * 8;
                    glds16(ok ? src + off : zp, la + 32 * i * BK);
                }
            } else {
                const int ke = k0 - 9 * Ctot;
                const half_t* src = p.E1;
                int cs = p.CE1, cc = ke;
                if (ke >= p.CE1) { src = p.E2; cs = p.CE2; cc = ke - p.CE1; }
#pragma unroll
                for (int i = 0; i < NA; ++i) glds16(a_ok[i] ? src + a_off[i] * cs + cc + kc * 8 : zp, la + 32 * i * BK);
            }
        } else {
            const bool kok = k0 + kc * 8 < p.K;
#pragma unroll
            for (int i = 0; i < NA; ++i) glds16((a_ok[i] && kok) ? A + a_off[i] + k0 : zp, la + 32 * i * BK);
        }
        const bool kok = k0 + kc * 8 < p.K;
#pragma unroll
        for (int i = 0; i < NB; ++i) glds16((w_ok[i] && kok) ? Wt + w_off[i] + k0 : zp, lb + 32 * i * BK);
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = rbase + 32 * i;
            *(half8*)(As + buf * BM * BK + row * BK + ((kc ^ ((row >> 1) & 7)) << 3)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = rbase + 32 * i;
            *(half8*)(Bs + buf * BN * BK + row * BK + ((kc ^ ((row >> 1) & 7)) << 3)) = rb[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk_all = (p.K + BK - 1) / BK;
    int kt_lo = 0, nk = nk_all;
    if (p.splits > 1) {  // this block's K slice
        const int per = (nk_all + p.splits - 1) / p.splits;
        kt_lo = blockIdx.y * per;
        nk = min(nk_all, kt_lo + per);
    }
    const int fr = lane & 15, fq = lane >> 4;
    if (kt_lo < nk) {
        if constexpr (GLDS) {
            stage_tile(0, kt_lo);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            load_tile(kt_lo);
            store_tile(0);
        }
    }
    __syncthreads();
    for (int kt = kt_lo; kt < nk; ++kt) {
        const int cur = (kt - kt_lo) & 1;
        if (kt + 1 < nk) {
            if constexpr (GLDS) stage_tile(cur ^ 1, kt + 1);
            else load_tile(kt + 1);
        }
        const half_t* Ac = As + cur * BM * BK;
        const half_t* Bc = Bs + cur * BN * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 af[TM], bf[TN];
            const int chunk = ks * 4 + fq;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wr * WM + i * 16 + fr;
                af[i] = *(const half8*)(Ac + row * BK + ((chunk ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wc * WN + j * 16 + fr;
                bf[j] = *(const half8*)(Bc + row * BK + ((chunk ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if constexpr (GLDS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next tile has landed in LDS
        } else {
            if (kt + 1 < nk) store_tile(cur ^ 1);
        }
        __syncthreads();
    }

    // ---------------- epilogue through LDS: 64 output rows per pass
    constexpr int LDS_N = BN + 4;
    constexpr int NPASS = BM / 64;
    constexpr int CH = BN / 8;
    float* stage = (float*)smem;
    half_t* __restrict__ Out = p.Out + z * p.strideO;
    for (int pass = 0; pass < NPASS; ++pass) {
        if (NPASS == 1 || wr == pass) {
            const int rb0 = (NPASS == 1) ? wr * WM : 0;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        stage[(rb0 + i * 16 + fq * 4 + r) * LDS_N + wc * WN + j * 16 + fr] = acc[i][j][r];
        }
        __syncthreads();
        for (int c = tid; c < 64 * CH; c += 256) {
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
                    const half8 rs = *(const half8*)(p.residual + z * p.strideR + (long long)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)rs[e];
                }
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)(v[e] * p.out_scale);
                *(half8*)(Out + (long long)m * p.ldo + n) = o;
            }
        }
        __syncthreads();
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
        for (int s = 1; s < p.splits; ++s) {
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

template <int BM, int BN, bool CONV>
static int launch_igemm(const IefGemmParams& p, int batch, hipStream_t st) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int splits = p.splits > 1 ? p.splits : 1;
    if (p.flags & 1)
        hipLaunchKernelGGL((igemm_f16_kernel<BM, BN, CONV, true>), dim3(tiles, splits, batch), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((igemm_f16_kernel<BM, BN, CONV, false>), dim3(tiles, splits, batch), dim3(256), 0, st, p);
    IEF_LAUNCH_CHECK();
    if (splits > 1) {
        const long long total = (long long)p.M * (p.N / 8);
        int grid = (int)((total + 255) / 256);
        if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(grid), dim3(256), 0, st, p);
        IEF_LAUNCH_CHECK();
    }
    return IEF_OK;
}

// Tile choice: fill 256 CUs.  Big-M layers take 128x128; when that yields fewer than ~1.5
// waves of blocks fall back to 64-row / 64-col tiles (more, smaller blocks).
template <bool CONV>
static int dispatch_igemm(const IefGemmParams& p, int batch, hipStream_t st) {
    auto nblk = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * batch; };
    if (p.tile_hint == 1) return launch_igemm<128, 128, CONV>(p, batch, st);
    if (p.tile_hint == 2) return launch_igemm<64, 128, CONV>(p, batch, st);
    if (p.tile_hint == 3) return launch_igemm<64, 64, CONV>(p, batch, st);
    if (p.tile_hint == 4) return launch_igemm<128, 64, CONV>(p, batch, st);
    if (nblk(128, 128) >= 384) return launch_igemm<128, 128, CONV>(p, batch, st);
    if (nblk(64, 128) >= 256) return launch_igemm<64, 128, CONV>(p, batch, st);
    return launch_igemm<64, 64, CONV>(p, batch, st);
}

static int check_common(const IefGemmParams& p) {
    if (!p.A || !p.W || !p.Out) return IEF_EINVAL;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return IEF_ESHAPE;
    if ((p.N & 7) || (p.K & 7) || (p.ldw & 7) || (p.ldo & 7)) return IEF_EALIGN;
    if (p.residual && (p.ldr & 7)) return IEF_EALIGN;
    if (p.rowvec && p.rows_per_batch <= 0) return IEF_ESHAPE;
    if (p.splits > 1 && !p.ws) return IEF_EINVAL;
    if ((p.flags & 1) && !p.zeros) return IEF_EINVAL;
    if (p.splits > 64) return IEF_ESHAPE;
    return IEF_OK;
}

extern "C" int ief_gemm_f16(const IefGemmParams* pp, int batch, void* stream) {
    if (!pp) return IEF_EINVAL;
    IefGemmParams p = *pp;
    int rc = check_common(p);
    if (rc) return rc;
    if (p.lda & 7) return IEF_EALIGN;
    if (batch <= 0) return IEF_ESHAPE;
    if (p.splits > 1 && batch != 1) return IEF_ESHAPE;
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
    p.Ho = (p.H + 2 - 3) / p.stride + 1;
    p.Wo = (p.Wd + 2 - 3) / p.stride + 1;
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
    p.strideA = p.strideW = p.strideO = p.strideR = 0;
    return dispatch_igemm<true>(p, 1, (hipStream_t)stream);
}
