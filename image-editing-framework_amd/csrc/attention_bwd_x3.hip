// Attention backward of the split-operand ("f16x3") mode: dQ, dK, dV of out = softmax(scale q k^T) v from fp32 q, k, v, dO and the
// forward's row log-sum-exp, WITHOUT materialising the maps -- P and dS are recomputed tile by tile and never leave registers.
// What it replaces on the fp32-storage reverse pass (hip._attn_bwd_f32: scores, dP, softmax backward, three batched map GEMMs):
// at the 64 x 64 level of SD1.5 one layer's maps are 2 x 537 MB per image; the null-text inner iteration spent 8.4 of its 27.5 ms
// there (profiles/r04_nti_kernel_stats_f16x3.csv).  Differentiates `/root/reference/p2p/model/register.py:43-51` for
// `/root/reference/p2p/inversion/nti.py:22-29` (loss.backward() through the UNet w.r.t. the unconditional embedding).
//
// The structure is csrc/attention_bwd.hip's (one template, two halves; a workgroup of 4 waves owns 128 columns X, a lane owns one
// column; 64-row tiles Y of the other side stream through LDS):
//
//     T1[y][x] = Y1[y] . X1[x]     (log2-unit scores: X1 is multiplied by scale log2e in fp32 BEFORE its split)
//     T2[y][x] = Y2[y] . X2[x]     (dP)
//     P  = exp2(T1 - lse[query])   dS = P (T2 - delta[query])            per lane, fp32
//     acc1[:, x] += Y1^T . dS      (dQ^T = K^T dS^T   |   dK^T = Q^T dS)
//     acc2[:, x] += Y2^T . P       (                      dV^T = dO^T P)     dK / dV half only
//
//              X1  X2   Y1  Y2
//     dQ       Q   dO   K   V      lse / delta per COLUMN (two scalars per lane)
//     dK/dV    K   V    Q   dO     lse / delta per ROW (staged with the tile)
//
// with EVERY product on split operands: each fp32 operand element is hi + lo (two fp16 halves, csrc/x3_common.h), each product
// three v_mfma_f32_32x32x16_f16 (lo hi + hi lo + hi hi) into one fp32 accumulator.  The column operands are split once per
// workgroup (registers), the tiles once per staging (fp32 from global -> split4 -> hi / lo images in LDS), P (<= 1: scale 2^14) and
// dS (times ds_mul; beyond the fp16 range it turns non-finite, which the drivers' guard reports) per 16-row step just before the MFMAs that consume them.  No atomics: every output
// element has one writer; results are deterministic.
#include "ief_common.h"
#include "ief_params.h"
#include "x3_common.h"

#define BX3_LOG2E 1.4426950408889634f

// delta[b][h][n] = sum_d dO[b][n][h D + d] * O[b][n][h D + d]   (fp32 in, fp32 out)
__global__ __launch_bounds__(256) void attn_bwd_delta_x3_kernel(const float* __restrict__ O, const float* __restrict__ dO,
                                                                float* __restrict__ delta, int B, int heads, int N, int d, int ldo,
                                                                int lddo) {
    const long long total = (long long)B * heads * N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i % N);
        const int h = (int)((i / N) % heads);
        const int b = (int)(i / ((long long)N * heads));
        const float* o = O + ((long long)b * N + n) * ldo + h * d;
        const float* g = dO + ((long long)b * N + n) * lddo + h * d;
        float s = 0.f;
        for (int c = 0; c < d; c += 4) {
            const f32x4 a = *(const f32x4*)(o + c), e = *(const f32x4*)(g + c);
            s += (a[0] * e[0] + a[1] * e[1]) + (a[2] * e[2] + a[3] * e[3]);
        }
        delta[i] = s;
    }
}

// eight fp32 values (two f32x4) -> hi / lo half8 (the B operand of one 16-deep MFMA step)
__device__ __forceinline__ void bx3_split8(const f32x4 a, const f32x4 b, half8_t& hi, half8_t& lo) {
    half4 h0, l0, h1, l1;
    split4(a, 1.0f, h0, l0);
    split4(b, 1.0f, h1, l1);
#pragma unroll
    for (int j = 0; j < 4; ++j) { hi[j] = h0[j]; hi[4 + j] = h1[j]; lo[j] = l0[j]; lo[4 + j] = l1[j]; }
}

template <int D, bool DKV>
__global__ __launch_bounds__(256) void attn_bwd_x3_kernel(const IefAttnBwdF32Params p) {
    constexpr int D16 = (D + 15) / 16;
    constexpr int DT = (D + 31) / 32;
    constexpr int RS = 32 * DT + 8;          // LDS row stride (halves): natural 16-byte reads and transposing reads conflict-free
    constexpr int CPR = D / 4;               // 16-byte fp32 chunks per row
    constexpr int NCH = (64 * CPR + 255) / 256;
    constexpr float SPP = 16384.f;           // split scale of P (<= 1)
    __shared__ __attribute__((aligned(16))) half_t Y1h[64 * RS], Y1l[64 * RS], Y2h[64 * RS], Y2l[64 * RS];
    __shared__ __attribute__((aligned(16))) float ylse[64], ydel[64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nx = DKV ? p.L : p.N, ny = DKV ? p.N : p.L;
    const int xblocks = (nx + 127) / 128;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int xblk = lid % xblocks, head = (lid / xblocks) % p.heads, b = lid / (xblocks * p.heads);
    const int col = xblk * 128 + wave * 32 + r;
    const bool col_ok = col < nx;

    const float* X1 = DKV ? p.K : p.Q;
    const float* X2 = DKV ? p.V : p.dO;
    const int ldx1 = DKV ? p.ldk : p.ldq, ldx2 = DKV ? p.ldv : p.ldo;
    const int ldy1 = DKV ? p.ldq : p.ldk, ldy2 = DKV ? p.ldo : p.ldv;
    const float* Y1 = (DKV ? p.Q : p.K) + (long long)b * ny * ldy1 + head * D;
    const float* Y2 = (DKV ? p.dO : p.V) + (long long)b * ny * ldy2 + head * D;
    const float* lse = p.lse + ((long long)b * p.heads + head) * p.N;
    const float* del = p.delta + ((long long)b * p.heads + head) * p.N;

    for (int i = tid; i < 64 * RS / 8; i += 256) {       // the padding columns (d >= D) stay zero: staging never writes them
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        ((half8*)Y1h)[i] = z; ((half8*)Y1l)[i] = z; ((half8*)Y2h)[i] = z; ((half8*)Y2l)[i] = z;
    }

    // column operands (B fragments): row `col` of X1 (scaled, then split) and X2 -- d = 16 s + 8 h .. + 7 for every 16-deep group
    half8_t x1h[D16], x1l[D16], x2h[D16], x2l[D16];
    {
        const float sc = p.scale * BX3_LOG2E;
        const long long o1 = ((long long)b * nx + col) * ldx1 + head * D, o2 = ((long long)b * nx + col) * ldx2 + head * D;
#pragma unroll
        for (int s = 0; s < D16; ++s) {
            const int dc = 16 * s + 8 * h;
            const bool ok = col_ok && dc < D;
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
            if (ok) {
                a0 = *(const f32x4*)(X1 + o1 + dc); a1 = *(const f32x4*)(X1 + o1 + dc + 4);
                b0 = *(const f32x4*)(X2 + o2 + dc); b1 = *(const f32x4*)(X2 + o2 + dc + 4);
            }
            bx3_split8(a0 * sc, a1 * sc, x1h[s], x1l[s]);
            bx3_split8(b0, b1, x2h[s], x2l[s]);
        }
    }
    float lse_c = 0.f, del_c = 0.f;
    if (!DKV && col_ok) { lse_c = lse[col]; del_c = del[col]; }

    // staging map (fixed): chunk c -> row c / CPR, 4 floats at (c % CPR) * 4; rows past the end are clamped (masked later)
    int st_row[NCH], st_ch[NCH], st_o[NCH];
    bool st_ok[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + 256 * i;
        const int row = c / CPR, ch = c - row * CPR;
        st_ok[i] = c < 64 * CPR;
        st_row[i] = st_ok[i] ? row : 0;
        st_ch[i] = st_ok[i] ? ch * 4 : 0;
        st_o[i] = row * RS + ch * 4;
    }
    f32x4 y1r[NCH], y2r[NCH];
    float lr = 0.f, dr = 0.f;
    auto load_tile = [&](int y0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int row = min(y0 + st_row[i], ny - 1);
            y1r[i] = *(const f32x4*)(Y1 + (long long)row * ldy1 + st_ch[i]);
            y2r[i] = *(const f32x4*)(Y2 + (long long)row * ldy2 + st_ch[i]);
        }
        if (DKV && tid < 64) {
            const int row = min(y0 + tid, ny - 1);
            lr = lse[row]; dr = del[row];
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (st_ok[i]) {
                half4 hh, ll;
                split4(y1r[i], 1.0f, hh, ll);
                *(half4*)(Y1h + st_o[i]) = hh; *(half4*)(Y1l + st_o[i]) = ll;
                split4(y2r[i], 1.0f, hh, ll);
                *(half4*)(Y2h + st_o[i]) = hh; *(half4*)(Y2l + st_o[i]) = ll;
            }
        }
        if (DKV && tid < 64) { ylse[tid] = lr; ydel[tid] = dr; }
    };

    f32x16 acc1[DT], acc2[DKV ? DT : 1];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc1[t][i] = 0.f;
            if constexpr (DKV) acc2[t][i] = 0.f;
        }

    const int nat_lane = r * RS + 8 * h;                                                       // natural fragment of row r
    const int L16 = lane & 15;
    const int tr_lane = (4 * h + (L16 >> 2)) * RS + 16 * ((lane >> 4) & 1) + 4 * (L16 & 3);  // transposing reads
    const float mul = p.ds_mul;

    const int nt = (ny + 63) / 64;
    __syncthreads();
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int j = 0; j < nt; ++j) {
        const int y0 = j * 64;
        if (j + 1 < nt) load_tile(y0 + 64);
        f32x16 t1[2], t2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) { t1[u][i] = 0.f; t2[u][i] = 0.f; }
#pragma unroll
        for (int s = 0; s < D16; ++s) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int o = nat_lane + 32 * u * RS + 16 * s;
                const half8_t a1h = *(const half8_t*)(Y1h + o), a1l = *(const half8_t*)(Y1l + o);
                const half8_t a2h = *(const half8_t*)(Y2h + o), a2l = *(const half8_t*)(Y2l + o);
                t1[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, x1h[s], t1[u], 0, 0, 0);
                t1[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, x1l[s], t1[u], 0, 0, 0);
                t1[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, x1h[s], t1[u], 0, 0, 0);
                t2[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2l, x2h[s], t2[u], 0, 0, 0);
                t2[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2h, x2l[s], t2[u], 0, 0, 0);
                t2[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2h, x2h[s], t2[u], 0, 0, 0);
            }
        }
        // P and dS in place (t1 <- P, t2 <- dS ds_mul); rows of this lane: (i & 3) + 8 (i >> 2) + 4 h of each 32-row half
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int rb = 32 * u + 8 * g4 + 4 * h;
                f32x4 l4 = {lse_c, lse_c, lse_c, lse_c}, d4 = {del_c, del_c, del_c, del_c};
                if constexpr (DKV) { l4 = *(const f32x4*)(ylse + rb); d4 = *(const f32x4*)(ydel + rb); }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g4 + e;
                    const bool ok = y0 + rb + e < ny;
                    const float pr = ok ? __builtin_amdgcn_exp2f(t1[u][i] - l4[e]) : 0.f;
                    t1[u][i] = pr;
                    t2[u][i] = pr * (t2[u][i] - d4[e]) * mul;
                }
            }
        // second products, one 16-row step (kk) at a time: split its 8 dS (and P) values, then the MFMAs of every 32-row tile of D
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int u = kk >> 1, base = 8 * (kk & 1);
            half8_t dh, dl, ph, pl;
            // no clamp: |dS ds_mul| beyond the fp16 range becomes inf and the gradient non-finite, which the drivers' non-finite guard
            // reports (as on the materialised path, whose map GEMM splits dS with the same scale) -- a silent saturation would not be seen
            bx3_split8(f32x4{t2[u][base], t2[u][base + 1], t2[u][base + 2], t2[u][base + 3]},
                       f32x4{t2[u][base + 4], t2[u][base + 5], t2[u][base + 6], t2[u][base + 7]}, dh, dl);
            if constexpr (DKV)
                bx3_split8(f32x4{t1[u][base], t1[u][base + 1], t1[u][base + 2], t1[u][base + 3]} * SPP,
                           f32x4{t1[u][base + 4], t1[u][base + 5], t1[u][base + 6], t1[u][base + 7]} * SPP, ph, pl);
            // the transposing reads of ALL D tiles of this step go out before its first MFMA (left alone the compiler waits for every group
            // of four in turn: one exposed LDS latency per three MFMAs on a kernel that runs one wave per SIMD)
            {
                half8_t ah[DT], al[DT];
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const int o = tr_lane + (16 * kk) * RS + 32 * t;
                    const half4 h0 = x3_lds_tr_read(Y1h + o), h1 = x3_lds_tr_read(Y1h + o + 8 * RS);
                    const half4 l0 = x3_lds_tr_read(Y1l + o), l1 = x3_lds_tr_read(Y1l + o + 8 * RS);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ah[t][q] = h0[q]; ah[t][4 + q] = h1[q]; al[t][q] = l0[q]; al[t][4 + q] = l1[q]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], dh, acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], dl, acc1[t], 0, 0, 0);
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], dh, acc1[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DKV) {
                half8_t ah[DT], al[DT];
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const int o = tr_lane + (16 * kk) * RS + 32 * t;
                    const half4 h0 = x3_lds_tr_read(Y2h + o), h1 = x3_lds_tr_read(Y2h + o + 8 * RS);
                    const half4 l0 = x3_lds_tr_read(Y2l + o), l1 = x3_lds_tr_read(Y2l + o + 8 * RS);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ah[t][q] = h0[q]; ah[t][4 + q] = h1[q]; al[t][q] = l0[q]; al[t][4 + q] = l1[q]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t], ph, acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], pl, acc2[t], 0, 0, 0);
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t], ph, acc2[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                       // everyone is done reading this tile
        if (j + 1 < nt) store_tile();
        __syncthreads();
    }

    if (col_ok) {
        const float f1 = p.scale / mul, f2 = 1.0f / SPP;
        float* o1 = (DKV ? p.dK : p.dQ) + ((long long)b * nx + col) * (DKV ? p.lddk : p.lddq) + head * D;
        float* o2 = DKV ? p.dV + ((long long)b * nx + col) * p.lddv + head * D : nullptr;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dbase = t * 32 + 8 * g + 4 * h;
                if (dbase < D) {
                    *(f32x4*)(o1 + dbase) = f32x4{acc1[t][4 * g], acc1[t][4 * g + 1], acc1[t][4 * g + 2], acc1[t][4 * g + 3]} * f1;
                    if constexpr (DKV)
                        *(f32x4*)(o2 + dbase) = f32x4{acc2[t][4 * g], acc2[t][4 * g + 1], acc2[t][4 * g + 2], acc2[t][4 * g + 3]} * f2;
                }
            }
    }
}

extern "C" int ief_attn_bwd_delta_f32in(const float* O, const float* dO, float* delta, int B, int heads, int N, int d, int ldo,
                                        int lddo, void* stream) {
    if (!O || !dO || !delta) return IEF_EINVAL;
    if (B <= 0 || heads <= 0 || N <= 0 || d <= 0 || (d & 3) || (ldo & 3) || (lddo & 3)) return IEF_ESHAPE;
    if (((uintptr_t)O | (uintptr_t)dO) & 15) return IEF_EALIGN;
    const long long total = (long long)B * heads * N;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(attn_bwd_delta_x3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, O, dO, delta, B, heads, N, d, ldo, lddo);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

extern "C" int ief_attn_bwd_x3(const IefAttnBwdF32Params* pp, int what, void* stream) {
    if (!pp) return IEF_EINVAL;
    const IefAttnBwdF32Params p = *pp;
    if (!p.Q || !p.K || !p.V || !p.dO || !p.lse || !p.delta) return IEF_EINVAL;
    if ((what & 1) && !p.dQ) return IEF_EINVAL;
    if ((what & 2) && (!p.dK || !p.dV)) return IEF_EINVAL;
    if (!(what & 3)) return IEF_EINVAL;
    if (p.B <= 0 || p.heads <= 0 || p.N <= 0 || p.L <= 0 || (p.d != 40 && p.d != 64)) return IEF_ESHAPE;
    if (!(p.ds_mul > 0.f)) return IEF_EINVAL;
    if ((p.ldq | p.ldk | p.ldv | p.ldo | p.lddq | p.lddk | p.lddv) & 3) return IEF_EALIGN;
    if (((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V | (uintptr_t)p.dO | (uintptr_t)(p.dQ ? p.dQ : p.Q) | (uintptr_t)(p.dK ? p.dK : p.Q) |
         (uintptr_t)(p.dV ? p.dV : p.Q)) & 15) return IEF_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (what & 1) {
        const dim3 grid(((p.N + 127) / 128) * p.heads * p.B);
        if (p.d == 40) hipLaunchKernelGGL((attn_bwd_x3_kernel<40, false>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((attn_bwd_x3_kernel<64, false>), grid, dim3(256), 0, st, p);
        IEF_LAUNCH_CHECK();
    }
    if (what & 2) {
        const dim3 grid(((p.L + 127) / 128) * p.heads * p.B);
        if (p.d == 40) hipLaunchKernelGGL((attn_bwd_x3_kernel<40, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((attn_bwd_x3_kernel<64, true>), grid, dim3(256), 0, st, p);
        IEF_LAUNCH_CHECK();
    }
    return IEF_OK;
}
