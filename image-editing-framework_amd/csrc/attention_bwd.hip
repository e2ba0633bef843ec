// Attention backward for gfx950 — activation gradients only (dQ, dK, dV), what null-text inversion
// needs to differentiate the UNet w.r.t. `encoder_hidden_states`
// (/root/reference/p2p/inversion/nti.py:22-29: `loss.backward()` through `model.unet(latent_cur, t, uncond_embeddings)`).
//
// One kernel template serves both halves.  A workgroup (4 waves) OWNS 128 columns X — queries for the
// dQ half, keys for the dK/dV half — a lane owns ONE column (its 32x32 MFMA column, as in the forward
// kernel), and streams 64-row tiles Y of the other side through LDS:
//
//     T1[y][x] = Y1[y] . X1[x]     (log2-unit scores: X1 is pre-multiplied by scale*log2e)
//     T2[y][x] = Y2[y] . X2[x]     (dP)
//     P  = exp2(T1 - lse[query])   dS = P * (T2 - delta[query])          per lane, fp32
//     acc1[:, x] += Y1^T . dS      (dQ^T = K^T dS^T   |   dK^T = Q^T dS)
//     acc2[:, x] += Y2^T . P       (                      dV^T = dO^T P)     dK/dV half only
//
//              X1  X2   Y1  Y2
//     dQ       Q   dO   K   V      lse/delta are per COLUMN (two scalars per lane)
//     dK/dV    K   V    Q   dO     lse/delta are per ROW (staged with the tile)
//
// P and dS never leave registers: the 32x32 accumulators are re-used as the B operand of the second
// MFMA with the k-permutation applied to the transposing LDS reads of Y^T (same idiom as attn_flash).
// No atomics: every output element has exactly one writer, results are deterministic.
#include "ief_common.h"
#include "ief_params.h"

#define LOG2E 1.4426950408889634f

typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fp16x4_t lds_fp16x4_t;
__device__ __forceinline__ half4 lds_tr_read_b(const half_t* p) {
    const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fp16x4_t*)p);
    return __builtin_bit_cast(half4, v);
}

__device__ __forceinline__ half_t sat_half(float v) { return (half_t)fminf(fmaxf(v, -65504.f), 65504.f); }

__device__ __forceinline__ half8 pack8s(const f32x16& p, int base) {
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = sat_half(p[base + j]);
    return o;
}

// delta[b][h][n] = sum_d dO[b][n][h*D + d] * O[b][n][h*D + d]
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const half_t* __restrict__ O, const half_t* __restrict__ dO,
                                                             float* __restrict__ delta, int B, int heads, int N, int d,
                                                             int ldo, int lddo) {
    const long long total = (long long)B * heads * N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i % N);
        const int h = (int)((i / N) % heads);
        const int b = (int)(i / ((long long)N * heads));
        const half_t* o = O + ((long long)b * N + n) * ldo + h * d;
        const half_t* g = dO + ((long long)b * N + n) * lddo + h * d;
        float s = 0.f;
        for (int c = 0; c < d; c += 8) {
            const half8 a = *(const half8*)(o + c), e = *(const half8*)(g + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)a[j] * (float)e[j];
        }
        delta[i] = s;
    }
}

template <int D, bool DKV>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const IefAttnBwdParams p) {
    constexpr int D16 = (D + 15) / 16;
    constexpr int DT = (D + 31) / 32;
    constexpr int RS = 32 * DT + 8;          // LDS row stride (halves): natural 16-B reads conflict-free
    constexpr int CPR = D / 8;               // 16-B chunks per row
    constexpr int NCH = (64 * CPR + 255) / 256;
    __shared__ __attribute__((aligned(16))) half_t Y1s[64 * RS];
    __shared__ __attribute__((aligned(16))) half_t Y2s[64 * RS];
    __shared__ __attribute__((aligned(16))) float ylse[64], ydel[64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nx = DKV ? p.L : p.N, ny = DKV ? p.N : p.L;
    const int xblocks = (nx + 127) / 128;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ysplits = DKV ? max(p.kv_splits, 1) : 1;
    const int split = lid % ysplits, lid2 = lid / ysplits;
    const int xblk = lid2 % xblocks, head = (lid2 / xblocks) % p.heads, b = lid2 / (xblocks * p.heads);
    const int col = xblk * 128 + wave * 32 + r;
    const bool col_ok = col < nx;

    const half_t* X1 = DKV ? p.K : p.Q;
    const half_t* X2 = DKV ? p.V : p.dO;
    const int ldx1 = DKV ? p.ldk : p.ldq, ldx2 = DKV ? p.ldv : p.ldo;
    const half_t* Y1 = (DKV ? p.Q : p.K) + (long long)b * ny * (DKV ? p.ldq : p.ldk) + head * D;
    const half_t* Y2 = (DKV ? p.dO : p.V) + (long long)b * ny * (DKV ? p.ldo : p.ldv) + head * D;
    const int ldy1 = DKV ? p.ldq : p.ldk, ldy2 = DKV ? p.ldo : p.ldv;
    const float* lse = p.lse + ((long long)b * p.heads + head) * p.N;
    const float* del = p.delta + ((long long)b * p.heads + head) * p.N;

    for (int i = tid; i < 64 * RS / 8; i += 256) {
        ((half8*)Y1s)[i] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
        ((half8*)Y2s)[i] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
    }

    // column operands (B fragments): row `col` of X1 (pre-scaled) and X2
    half8 x1f[D16], x2f[D16];
    {
        const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
        const float sc = p.scale * LOG2E;
        const long long o1 = ((long long)b * nx + col) * ldx1 + head * D, o2 = ((long long)b * nx + col) * ldx2 + head * D;
#pragma unroll
        for (int s = 0; s < D16; ++s) {
            const int dc = 16 * s + 8 * h;
            const bool ok = col_ok && dc < D;
            x1f[s] = ok ? *(const half8*)(X1 + o1 + dc) : zero8;
            x2f[s] = ok ? *(const half8*)(X2 + o2 + dc) : zero8;
#pragma unroll
            for (int e = 0; e < 8; ++e) x1f[s][e] = (half_t)((float)x1f[s][e] * sc);
        }
    }
    float lse_c = 0.f, del_c = 0.f;
    if (!DKV && col_ok) { lse_c = lse[col]; del_c = del[col]; }

    // staging map (fixed): chunk c -> row c / CPR, 8 halves at (c % CPR) * 8; rows past the end are clamped (masked later)
    int st_row[NCH], st_ch[NCH], st_o[NCH];
    bool st_ok[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + 256 * i;
        const int row = c / CPR, ch = c - row * CPR;
        st_ok[i] = c < 64 * CPR;
        st_row[i] = st_ok[i] ? row : 0;
        st_ch[i] = st_ok[i] ? ch * 8 : 0;
        st_o[i] = row * RS + ch * 8;
    }
    half8 y1r[NCH], y2r[NCH];
    float lr = 0.f, dr = 0.f;
    auto load_tile = [&](int y0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int row = min(y0 + st_row[i], ny - 1);
            y1r[i] = *(const half8*)(Y1 + (long long)row * ldy1 + st_ch[i]);
            y2r[i] = *(const half8*)(Y2 + (long long)row * ldy2 + st_ch[i]);
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
                *(half8*)(Y1s + st_o[i]) = y1r[i];
                *(half8*)(Y2s + st_o[i]) = y2r[i];
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

    const int nt_all = (ny + 63) / 64;
    const int tps = (nt_all + ysplits - 1) / ysplits;          // tiles per split (a trailing split may be empty: writes zeros)
    const int jb = split * tps, nt = min(nt_all, jb + tps);
    __syncthreads();
    load_tile(jb * 64);
    store_tile();
    __syncthreads();
    for (int j = jb; j < nt; ++j) {
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
                const half8 a1 = *(const half8*)(Y1s + nat_lane + 32 * u * RS + 16 * s);
                const half8 a2 = *(const half8*)(Y2s + nat_lane + 32 * u * RS + 16 * s);
                t1[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, x1f[s], t1[u], 0, 0, 0);
                t2[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, x2f[s], t2[u], 0, 0, 0);
            }
        }
        // P and dS in place (t1 <- P, t2 <- dS * ds_mul); rows of this lane: (i&3) + 8(i>>2) + 4h of each 32-row half
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
        const half8 db[4] = {pack8s(t2[0], 0), pack8s(t2[0], 8), pack8s(t2[1], 0), pack8s(t2[1], 8)};
        half8 pb[4];
        if constexpr (DKV) { pb[0] = pack8s(t1[0], 0); pb[1] = pack8s(t1[0], 8); pb[2] = pack8s(t1[1], 0); pb[3] = pack8s(t1[1], 8); }
#pragma unroll
        for (int t = 0; t < DT; ++t) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const half_t* a1p = Y1s + tr_lane + (16 * kk) * RS + 32 * t;
                const half4 lo = lds_tr_read_b(a1p), hi = lds_tr_read_b(a1p + 8 * RS);
                const half8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, db[kk], acc1[t], 0, 0, 0);
                if constexpr (DKV) {
                    const half_t* a2p = Y2s + tr_lane + (16 * kk) * RS + 32 * t;
                    const half4 lo2 = lds_tr_read_b(a2p), hi2 = lds_tr_read_b(a2p + 8 * RS);
                    const half8 a2 = {lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
                    acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, pb[kk], acc2[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();                       // everyone is done reading this tile
        if (j + 1 < nt) store_tile();
        __syncthreads();
    }

    if (DKV && ysplits > 1) {
        // raw fp32 partials: ws[which][split][b][col][heads*D]; attn_bwd_reduce_kernel sums the splits
        if (col_ok) {
            const long long per = (long long)p.B * nx * p.heads * D;
            float* w1 = p.ws + (long long)split * per + ((long long)b * nx + col) * p.heads * D + head * D;
            float* w2 = w1 + (long long)ysplits * per;
#pragma unroll
            for (int t = 0; t < DT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dbase = t * 32 + 8 * g + 4 * h;
                    if (dbase < D) {
                        *(f32x4*)(w1 + dbase) = (f32x4){acc1[t][4 * g], acc1[t][4 * g + 1], acc1[t][4 * g + 2], acc1[t][4 * g + 3]};
                        if constexpr (DKV)
                            *(f32x4*)(w2 + dbase) = (f32x4){acc2[t][4 * g], acc2[t][4 * g + 1], acc2[t][4 * g + 2], acc2[t][4 * g + 3]};
                    }
                }
        }
        return;
    }
    if (col_ok) {
        const float f1 = p.scale / mul;
        half_t* o1 = (DKV ? p.dK : p.dQ) + ((long long)b * nx + col) * (DKV ? p.lddk : p.lddq) + head * D;
        half_t* o2 = DKV ? p.dV + ((long long)b * nx + col) * p.lddv + head * D : nullptr;
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dbase = t * 32 + 8 * g + 4 * h;
                if (dbase < D) {
                    half4 v = {sat_half(acc1[t][4 * g] * f1), sat_half(acc1[t][4 * g + 1] * f1),
                               sat_half(acc1[t][4 * g + 2] * f1), sat_half(acc1[t][4 * g + 3] * f1)};
                    *(half4*)(o1 + dbase) = v;
                    if constexpr (DKV) {
                        half4 w = {sat_half(acc2[t][4 * g]), sat_half(acc2[t][4 * g + 1]), sat_half(acc2[t][4 * g + 2]),
                                   sat_half(acc2[t][4 * g + 3])};
                        *(half4*)(o2 + dbase) = w;
                    }
                }
            }
    }
}

// sums the query-split partials of dK / dV (fixed order: deterministic) and applies the fp32 factors
__global__ __launch_bounds__(256) void attn_bwd_reduce_kernel(const IefAttnBwdParams p) {
    const int C = p.heads * p.d, C4 = C >> 2;
    const long long rows = (long long)p.B * p.L, total = rows * C4;
    const long long per = rows * C;
    const float f1 = p.scale / p.ds_mul;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / C4;
        const int c = (int)(i % C4) * 4;
        f32x4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
        for (int s = 0; s < p.kv_splits; ++s) {
            const f32x4 x = *(const f32x4*)(p.ws + (long long)s * per + row * C + c);
            const f32x4 y = *(const f32x4*)(p.ws + ((long long)p.kv_splits + s) * per + row * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] += x[e]; b[e] += y[e]; }
        }
        half4 ka = {sat_half(a[0] * f1), sat_half(a[1] * f1), sat_half(a[2] * f1), sat_half(a[3] * f1)};
        half4 va = {sat_half(b[0]), sat_half(b[1]), sat_half(b[2]), sat_half(b[3])};
        *(half4*)(p.dK + row * p.lddk + c) = ka;
        *(half4*)(p.dV + row * p.lddv + c) = va;
    }
}

#define BWD_DISPATCH(DD, KV)                                                                                        \
    hipLaunchKernelGGL((attn_bwd_kernel<DD, KV>), dim3(grid), dim3(256), 0, st, p)

template <bool KV>
static int launch_bwd(const IefAttnBwdParams& p, hipStream_t st) {
    const int nx = KV ? p.L : p.N;
    const int grid = ((nx + 127) / 128) * p.heads * p.B * (KV && p.kv_splits > 1 ? p.kv_splits : 1);
    switch (p.d) {
        case 32: BWD_DISPATCH(32, KV); break;
        case 40: BWD_DISPATCH(40, KV); break;
        case 64: BWD_DISPATCH(64, KV); break;
        case 80: BWD_DISPATCH(80, KV); break;
        case 160: BWD_DISPATCH(160, KV); break;
        default: return IEF_ESHAPE;
    }
    IEF_LAUNCH_CHECK();
    if (KV && p.kv_splits > 1) {
        const long long total = (long long)p.B * p.L * (p.heads * p.d / 4);
        int rg = (int)((total + 255) / 256);
        if (rg > 2048) rg = 2048;
        hipLaunchKernelGGL(attn_bwd_reduce_kernel, dim3(rg), dim3(256), 0, st, p);
        IEF_LAUNCH_CHECK();
    }
    return IEF_OK;
}

extern "C" int ief_attn_bwd_delta_f32(const ief_half* O, const ief_half* dO, float* delta, int B, int heads, int N, int d,
                                      int ldo, int lddo, void* stream) {
    if (!O || !dO || !delta) return IEF_EINVAL;
    if (B <= 0 || heads <= 0 || N <= 0 || d <= 0 || (d & 7)) return IEF_ESHAPE;
    if ((ldo & 7) || (lddo & 7)) return IEF_EALIGN;
    const long long total = (long long)B * heads * N;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, O, dO, delta, B, heads, N, d,
                       ldo, lddo);
    IEF_LAUNCH_CHECK();
    return IEF_OK;
}

extern "C" int ief_attn_bwd_f16(const IefAttnBwdParams* pp, int what, void* stream) {
    if (!pp) return IEF_EINVAL;
    const IefAttnBwdParams p = *pp;
    if (!p.Q || !p.K || !p.V || !p.dO || !p.lse || !p.delta) return IEF_EINVAL;
    if (what < 1 || what > 3) return IEF_ESHAPE;
    if ((what & 1) && !p.dQ) return IEF_EINVAL;
    if ((what & 2) && (!p.dK || !p.dV)) return IEF_EINVAL;
    if (p.B <= 0 || p.heads <= 0 || p.N <= 0 || p.L <= 0) return IEF_ESHAPE;
    if ((p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.ldo & 7)) return IEF_EALIGN;
    if ((what & 1) && (p.lddq & 3)) return IEF_EALIGN;
    if ((what & 2) && ((p.lddk & 3) || (p.lddv & 3))) return IEF_EALIGN;
    if (!(p.ds_mul > 0.f)) return IEF_ESHAPE;
    if (p.kv_splits < 0 || p.kv_splits > 256 || (p.kv_splits > 1 && !p.ws)) return IEF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int rc = IEF_OK;
    if (what & 1) rc = launch_bwd<false>(p, st);
    if (rc == IEF_OK && (what & 2)) rc = launch_bwd<true>(p, st);
    return rc;
}
