/* C-ABI of libief_hip.so — the native boundary under the attention-controlled denoising path.
 *
 * The reference (AY-Liu/Image-Editing-Framework) has no native code and no FFI: its per-step
 * compute is PyTorch/diffusers eager CUDA (last row).  This header is therefore
 * the boundary a maintainer binds INSTEAD of those eager calls; each entry point cites the
 * reference call site whose arithmetic it replaces.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm tensors in our host
 *     code); outputs are caller-allocated; no entry point allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it (graph-capturable);
 *   - return 0 on success, <0 for rejected arguments (IEF_E*), >0 = hipError_t of the launch;
 *     nothing throws;
 *   - ief_half = IEEE fp16 storage; all accumulation and statistics are fp32;
 *   - activations are channels-last: [B, H, W, C] == tokens-major [B, H*W, C].
 */
#ifndef IEF_HIP_H
#define IEF_HIP_H
#include <stdint.h>

#ifdef __HIPCC__
typedef _Float16 ief_half;
#else
typedef uint16_t ief_half;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define IEF_ABI_VERSION 4
int ief_abi_version(void);
/* name of the code object's target, e.g. "gfx950" */
const char* ief_target_arch(void);
/* sizeof the parameter structs as compiled (0: IefGemmParams, 1: IefAttnParams, 2: IefCrossParams, 3: IefAttnBwdParams,
 * 4: IefMapLossParams, 5: IefGemmF32Params, 6: IefAttnF32Params, 7: IefGemmX3pParams) */
int ief_struct_size(int which);

/* ------------------------------------------------------------------ GEMM / conv3x3
 * Out[m][n] = ( sum_k A(m,k) W[n][k] + bias[n] + rowvec[m / rows_per_batch][n] + residual[m][n] ) * out_scale
 * replaces: Attention.to_q/to_k/to_v/to_out, proj_in/proj_out, FF linears
 *           (/root/reference/p2p/model/register.py:33,40,41,54) and ResnetBlock2D.conv1/conv2/
 *           conv_shortcut + temb add + skip add (/root/reference/pnp/model/register.py:139-175).
 */
typedef struct IefGemmParams {
    const ief_half* A;        /* dense: [M][lda];  conv: NHWC source 1, C1 channels      */
    const ief_half* A2;       /* conv only: NHWC source 2 (channel concat), C2 channels  */
    const ief_half* W;        /* [N][ldw], K contiguous; conv: [Cout][3][3][C1+C2]        */
    ief_half* Out;            /* [M][ldo]                                                 */
    const float* bias;        /* [N] or NULL                                              */
    const float* rowvec;      /* [M / rows_per_batch][N] or NULL                          */
    const ief_half* residual; /* [M][ldr] or NULL                                         */
    int M, N, K;
    int lda, ldw, ldo, ldr;
    long long strideA, strideW, strideO, strideR; /* per batch index (dense, blockIdx.z) */
    /* conv3x3 geometry (pad 1): logical input H x Wd (after the optional 2x upsample) */
    int H, Wd, C1, C2, Ho, Wo, stride, ups, batch_images;
    int rows_per_batch;
    float out_scale;
    int tile_hint;            /* 0 auto; otherwise a tile id (BM x BN, waves): 1: 128x128, 2: 64x128, 3: 64x64, 4: 128x64,
                               * 5: 64x160, 6: 128x160 (2x2), 7: 128x160 (4x2), 8: 256x128, 9: 128x128 (4x2); 16-21: 7, 9, 8, 5, 3, 6
                               * with 4 / 4 / 4 / 2 / 2 / 2 LOADER waves (extra waves that stage the operands, the others only
                               * multiply); conv3x3 only: 14 / 15 = the halo kernel (256x80; 3x3, stride 1, pad 1, no fused 1x1
                               * range, rows of <= 64 pixels; the input tile stays in LDS across the nine taps), 15 with four
                               * loader waves; 15 also takes ups = 1 (rows of <= 128 pixels, H*Wd a multiple of 256, Wd | 256).  ief_gemm_tile_bm / _bn give BM / BN of an id. */
    /* conv only: extra K range appended after the 9 taps, read at the OUTPUT pixel itself
     * (a fused 1x1 convolution over up to two more NHWC sources, e.g. ResnetBlock2D.conv_shortcut
     * over the un-concatenated [x | skip]); W rows are then [9*(C1+C2) + CE1 + CE2] long.
     * Requires stride 1, no upsample. */
    const ief_half* E1; const ief_half* E2;
    int CE1, CE2;
    /* split-K: `splits` > 1 cuts the K loop into that many slices (grid.y); each slice writes an fp32
     * partial slab ws[slice][M][N] and a second launch sums the slabs and applies the epilogue.
     * ws must hold splits*M*N floats.  Used where M*N alone yields too few tiles for 256 CUs. */
    int splits;
    float* ws;
    /* flags bit 1 (value 2): fused GEGLU epilogue — W rows interleaved in groups of 8 ([8 hidden | 8 gate] ...),
     * Out[m][j] = (h_j + bias) * gelu(g_j + bias), Out has N/2 columns (FeedForward.net[0] of diffusers' GEGLU);
     * `zeros` must point at >= 16 bytes of device zeros (source of padded / out-of-range chunks). */
    int flags;
    const ief_half* zeros;
    int stages;               /* depth of the LDS operand ring: 0/2 (double buffer), 3 or 4 K tiles in flight */
    int pad_hi_only;          /* conv: 1 = zero padding only on the bottom/right edge (pad (0,1,0,1)), as the VAE encoder's
                               * stride-2 Downsample2D uses; 0 = symmetric padding 1 */
    /* LayerNorm folded into the NEXT linear (BasicTransformerBlock.norm1/2/3 -> to_q / to_qkv / ff.net[0]):
     *   LN(x) W^T = rstd (x (W gamma)^T - mu colsum) + beta W^T,   colsum[n] = sum_k (W gamma)[n][k]
     * so the consumer GEMM runs on the RAW residual stream with W gamma as weights and beta W^T folded into its bias, and
     * corrects per row in its epilogue; the row moments come from the epilogue of the GEMM that PRODUCED x.
     * rstat_out: producer side, [M][tiles_n][2] fp32 = (sum, sum of squares) of the fp16-rounded outputs over each N tile
     *            (tiles_n = ceil(N / BN) of the tile this launch uses; ief_gemm_tile_bn(tile_hint) gives BN);
     * rstat_in / rstat_slots / colsum / ln_eps: consumer side, rstat_in = the producer's rstat_out, rstat_slots its tiles_n;
     *            the LayerNorm width is this GEMM's K.  Neither side may use split-K; rstat_out excludes the GEGLU epilogue. */
    float* rstat_out;
    const float* rstat_in;
    int rstat_slots;
    const float* colsum;
    float ln_eps;
    /* GroupNorm statistics from the PRODUCER of its input: cstat_out [ceil(M / BM)][N][2] fp32 = per output channel the
     * (sum, sum of squares) of the fp16-rounded outputs over the BM rows of each M tile, summed in a fixed order
     * (ief_gemm_tile_bm(tile_hint) gives BM; the consumer is ief_groupnorm_cstat_f16).  No split-K, no GEGLU epilogue,
     * dense batch 1. */
    float* cstat_out;
    /* split-K combined INSIDE the launch: cnt points at one zeroed int per output tile (ceil(M/BM) * ceil(N/BN)).  Each
     * K-slice workgroup stores its fp32 slab tile, releases it at agent scope and draws a ticket from cnt[tile]; the
     * workgroup that draws the last ticket acquires, sums the slabs IN SLAB ORDER (so the sum does not depend on who came
     * last), applies the epilogue (and cstat_out, now allowed with splits > 1) and stores 0 back into cnt[tile]: the
     * counters are zero again when the launch ends.  NULL: the slabs are summed by a second launch, as before.  Launches
     * that may run concurrently must not share counters. */
    int* cnt;
} IefGemmParams;

int ief_gemm_f16(const IefGemmParams* p, int batch, void* stream);
/* BN (output columns per workgroup) of a tile id, 0 for an unknown id */
int ief_gemm_tile_bn(int tile_hint);
/* fills M, K, ldw, Ho, Wo, rows_per_batch itself from the geometry fields */
int ief_conv3x3_f16(const IefGemmParams* p, void* stream);

/* conv_in: latent NCHW fp32 [B,Cin,H,W] (Cin<=8) -> NHWC fp16 [B,H,W,Cout]; W [3][3][Cin][Cout] fp16 (k-major).
 * conv_out: NHWC fp16 [B,H,W,C] -> NCHW fp32 [B,Cout,H,W] (Cout<=8); W [Cout][3][3][C] fp16.
 * (UNet2DConditionModel.conv_in / conv_out, called at /root/reference/p2p/model/sd_utils.py:73) */
int ief_conv_in_f32(const float* x, const ief_half* w, const float* bias, ief_half* out,
                    int B, int Cin, int H, int Wd, int Cout, void* stream);
int ief_conv_out_f32(const ief_half* x, const ief_half* w, const float* bias, float* out,
                     int B, int C, int H, int Wd, int Cout, void* stream);

/* ------------------------------------------------------------------ normalisation
 * GroupNorm over NHWC (+ optional SiLU): ResnetBlock2D.norm1/norm2 + nonlinearity
 * (/root/reference/pnp/model/register.py:105-110,149-158), Transformer2DModel.norm, conv_norm_out.
 * `partial` is scratch of >= B * (splits + 1) * groups * 2 floats (splits = ief_gn_splits(HW)); on completion its
 * LAST B * groups * 2 floats hold (mean, rstd) per (batch, group) — what ief_groupnorm_bwd_f16 takes as `stats`.
 * Two sources (x, x2) = channel concat [C1 | C2] normalised as one tensor, written to out [.., C1+C2].
 */
int ief_gn_splits(int HW);
int ief_groupnorm_silu_f16(const ief_half* x, const ief_half* x2, int C1, int C2, ief_half* out,
                           const float* gamma, const float* beta, float* partial,
                           int B, int HW, int groups, float eps, int apply_silu, void* stream);
/* LayerNorm over the last dim of [rows][C] (BasicTransformerBlock.norm1/2/3) */
int ief_layernorm_f16(const ief_half* x, ief_half* out, const float* gamma, const float* beta,
                      int rows, int C, float eps, void* stream);
/* GEGLU: in [rows][2*Ch] = [hidden | gate] -> out [rows][Ch] = hidden * gelu(gate) (erf form) */
int ief_geglu_f16(const ief_half* in, ief_half* out, int rows, int Ch, void* stream);

/* ------------------------------------------------------------------ attention
 * Fused softmax(Q K^T * scale) V, never materialising the map (flash-style), with per-batch
 * indirection of the Q / K / V source rows.  Identity maps = plain attention
 * (/root/reference/p2p/model/register.py:47-50 when the controller leaves the map alone).
 * q_src/k_src/v_src: DEVICE int32 [B] or NULL; out[b] = softmax(Q[q_src[b]] K[k_src[b]]^T) V[v_src[b]]:
 *   - P2P self-attention replace (attention_base.py:123,132-136): target rows take q_src = k_src = source row;
 *   - MasaCtrl mutual self-attention (/root/reference/masactrl/model/attention_control.py:59-66): k_src = v_src = source row.
 * Q [B][N][ldq], K/V [B][L][ldk/ldv], head h at columns [h*d, (h+1)*d); out [B][N][ldo].  d in {32,40,64,80,160}.
 */
typedef struct IefAttnParams {
    const ief_half* Q; const ief_half* K; const ief_half* V; ief_half* Out;
    int B, heads, N, L, d;
    int ldq, ldk, ldv, ldo;
    float scale;
    const int* q_src; const int* k_src; const int* v_src;
    float* lse;   /* ief_attn_flash_f16 only, may be NULL: fp32 [B][heads][N] row log-sum-exp (log2 units) for ief_attn_bwd_f16 */
    int variant;  /* ief_attn_flash_f16 only: 0 = default (software-pipelined 4-wave kernel), 1 = plain 4-wave kernel,
                     2 = 8-wave ping-pong kernel; same results within fp16 rounding, kept for A/B measurement */
} IefAttnParams;
int ief_attn_flash_f16(const IefAttnParams* p, void* stream);

/* Cross-attention with the Prompt-to-Prompt edit fused in (L <= 96 keys):
 *   P = softmax(Q K^T scale);  for batch rows b with edit_src[b] >= 0:
 *   P'[q][n] = c1[n] * sum_w P_src[q][w] M[w][n] + c2[n] * P_b[q][n];   out = P' V_b
 * which covers AttentionControlEdit.forward's cross branch (attention_base.py:118-121) for
 * replace (M = mapper, c1 = alpha_t, c2 = 1 - alpha_t; attention_control.py:15-16),
 * refine  (M = one-hot gather of mapper, c1 = alpha_t a, c2 = 1 - alpha_t a; :28-31) and
 * reweight (M = diag(equalizer) [x prev edit]; :42-46).
 * edit_src/edit_slot: DEVICE int32 [B]; MT: fp16 [slots][96][96] = M transposed, zero padded;
 * coef: fp32 [slots][2][96] = c1 | c2 of the CURRENT step.  All device memory so a captured
 * graph follows the step-dependent gates without re-capture.
 */
typedef struct IefCrossParams {
    const ief_half* Q; const ief_half* K; const ief_half* V; ief_half* Out;
    int B, heads, N, L, d;
    int ldq, ldk, ldv, ldo;
    float scale;
    const int* edit_src; const int* edit_slot;
    const ief_half* MT; const float* coef;
} IefCrossParams;
int ief_attn_cross_p2p_f16(const IefCrossParams* p, void* stream);

/* Generic hook path (materialised maps for arbitrary Python controllers / AttentionStore):
 * probs [B*heads][N][L] fp16 = softmax(Q K^T scale)  (Attention.get_attention_scores,
 * register.py:47) and out = probs V (torch.bmm + batch_to_head_dim, register.py:50-51). */
int ief_attn_probs_f16(const IefAttnParams* p, ief_half* probs, void* stream);
int ief_attn_apply_f16(const IefAttnParams* p, const ief_half* probs, void* stream);

/* ------------------------------------------------------------------ sampler step
 * Classifier-free guidance + DDIM update (or its inverse) in one launch:
 *   eps = eps_u + g (eps_c - eps_u);  x0 = (x - sqrt(1-a_from) eps)/sqrt(a_from);
 *   x' = sqrt(a_to) x0 + sqrt(1-a_to) eps
 * (/root/reference/p2p/model/sd_utils.py:74-76; inverse: /root/reference/p2p/inversion/ddim.py:9-18).
 * eps_u/eps_c/x/x_out fp32, n elements each; eps_u == NULL => eps = eps_c (no guidance).
 * coef: DEVICE fp32 [3] = {a_from, a_to, guidance}; x0_out optional.
 */
int ief_cfg_ddim_step_f32(const float* eps_u, const float* eps_c, const float* x, float* x_out,
                          float* x0_out, const float* coef, long long n, void* stream);
/* sinusoidal timestep embedding, flip_sin_to_cos, shift 0: out fp16 [B][dim] = [cos | sin] */
int ief_timestep_embedding_f16(const float* t, ief_half* out, int B, int dim, void* stream);
/* out = a + b elementwise on fp16 (residual add when a hook owns Attention.forward) */
int ief_add_f16(const ief_half* a, const ief_half* b, ief_half* out, long long n, void* stream);
/* out[b][:] = in[src[b]][:] for fp16 [B][row_elems], src a DEVICE int32 [B] with values in [0, B): Plug-and-Play's
 * feature injection (/root/reference/pnp/model/register.py:161-166) as a gather in front of conv2 */
int ief_gather_rows_f16(const ief_half* in, ief_half* out, const int* src, int B, long long row_elems, void* stream);
/* in-place row softmax over the last dim of fp16 [rows][L] (materialised attention of the VAE mid block, d = 512) */
int ief_softmax_rows_f16(ief_half* x, int rows, int L, void* stream);
/* out[c][r] = in[r][c] for fp16 [R][C] (V^T for the P.V GEMM of the VAE attention) */
int ief_transpose_f16(const ief_half* in, ief_half* out, int R, int C, void* stream);
/* 1x1 convolution on a small fp32 NCHW tensor, Cin, Cout <= 8 (AutoencoderKL.quant_conv / post_quant_conv) */
int ief_pointwise_f32(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int HW,
                      void* stream);
/* y = silu(x) elementwise on fp16 (ResnetBlock2D time_emb_proj input) */
int ief_silu_f16(const ief_half* x, ief_half* out, long long n, void* stream);
/* fp16 <-> fp32 casts and row gather used to stage per-step tables inside a captured graph:
 * out[r][:] = table[idx[0] (device int32)][r][:], element size `esize` bytes, row bytes `rowbytes` */
int ief_cast_f32_to_f16(const float* x, ief_half* out, long long n, void* stream);
int ief_cast_f16_to_f32(const ief_half* x, float* out, long long n, void* stream);
/* out[:] = table[min(max(step[0], 0), n_rows - 1)][:]: the row index is read from device memory and CLAMPED to the table,
 * so a captured step graph replayed past the end of its schedule re-reads the last row instead of foreign memory */
int ief_select_step(const void* table, void* out, const int* step, long long bytes_per_step, int n_rows, void* stream);
int ief_advance_step(int* step, void* stream);

/* ------------------------------------------------------------------ reference-precision ("exact") mode: fp32 end to end
 * The reference computes in fp32 (/root/reference/p2p/edit_syn.py:38).  These entry points are the same operators on
 * fp32 activations and fp32 weights, contractions on the fp32-input MFMA (csrc/exact_f32.hip); the host picks them by
 * the dtype of the tensors it is handed (hip.py).  Attention in this mode MATERIALISES its maps in HBM as the
 * reference does (/root/reference/p2p/model/register.py:43-51): scores and P.V are two batched launches of ief_gemm_f32
 * with ief_softmax_rows_f32 (and the P2P edit, ief_p2p_cross_edit_f32) between them. */
typedef struct IefGemmF32Params {
    const float* A;         /* [M][K] rows (lda) | conv: NHWC source 1 */
    const float* A2;        /* conv: NHWC source 2 of a channel concat, or NULL */
    const float* W;         /* [N][K] rows (ldw); transb: [K][N] rows (ldw) */
    float* Out;             /* [M][N] rows (ldo) */
    const float* bias;      /* [N] or NULL */
    const float* rowvec;    /* [M / rows_per_batch][N] or NULL */
    const float* residual;  /* [M][N] rows (ldr) or NULL */
    int M, N, K;
    int lda, ldw, ldo, ldr;
    int rows_per_batch;
    float out_scale;        /* out = (acc + bias + rowvec + residual) * out_scale */
    /* 3x3 implicit GEMM (conv != 0): same meaning as the fields of IefGemmParams */
    int conv, H, Wd, C1, C2, Ho, Wo, stride, ups, batch_images, pad_hi_only;
    const float* E1;
    const float* E2;
    int CE1, CE2;
    /* batched product (heads > 0): grid.z = batch * heads; element strides per batch row / head; optional batch-row
     * indirection of the A and W operands (P2P self-replace, MasaCtrl, Plug-and-Play source rows) */
    int batch, heads;
    long long sAb, sAh, sWb, sWh, sOb, sOh;
    const int* a_src;
    const int* w_src;
    int transb;
    int a_scalar;           /* set by the library: A rows are not 16-byte chunked */
    /* split-K (not batched): K cut `splits` ways over grid.y, fp32 slabs ws[splits][M][N], summed in slab order by a second
     * launch that applies the epilogue */
    int splits;
    float* ws;
    /* ABI 3: split-operand contraction (csrc/split_x3.hip).  x3 != 0: every fp32 operand element is split into two fp16
     * numbers (hi + lo of sa * a, of sb * w; sa, sb powers of two) while its tile is staged, the product runs on three
     * fp16 MFMAs per k-step (Ah Bh + Al Bh + Ah Bl, fp32 accumulate) and is divided by sa * sb; same operators, operands,
     * outputs and error codes as x3 == 0 (the fp32-input MFMA); |sa * a|, |sb * w| must stay below 1.3e5 */
    int x3;
    float sa, sb;
    /* set by the library (x3): every epilogue operand allows 16-byte accesses; conv channel counts are all multiples of
     * 32; bytes reachable from A / W / A2 / E1 / E2 (buffer descriptor sizes: each must stay below 4 GiB) */
    int vec_out, al32, fast_ok;
    unsigned bytesA, bytesW, bytesA2, bytesE1, bytesE2;
    /* x3: optional pre-split planes of a WEIGHT operand (made once per tensor by ief_x3_split_weights with scale sb): fp16
     * [2][N][K] contiguous (hi plane, lo plane); W is then not read.  Not for transb / batched products. */
    const void* Wp;
    /* x3, linear only: GEGLU fused into the epilogue -- the weight's rows are interleaved [8 hidden | 8 gate] per 16 columns
     * (the layout of ief_geglu_il_f32); Out is [M][N / 2] = hidden * gelu(gate) of (a . w + bias); no residual / rowvec / split-K */
    int geglu;
} IefGemmF32Params;
int ief_gemm_f32(const IefGemmF32Params* p, void* stream);
/* planes[0][i] = fp16(scale w[i]), planes[1][i] = fp16(scale w[i] - planes[0][i]); n % 4 == 0 */
int ief_x3_split_weights(const float* w, void* planes, long long n, float scale, void* stream);
int ief_gemm_x3_bn(int N);    /* x3 != 0: output-tile width (80 or 64) for N columns; ief_gemm_x3_bm(M, N): its row count (128) */
int ief_gemm_x3_bm(int M, int N);
/* the width the launch (conv != 0: 3x3 convolution) actually takes: 160 (8 waves, one workgroup per CU) where that tile is
   the faster one, else ief_gemm_x3_bn(N); the host's split-K policy counts tiles with it */
int ief_gemm_x3_bn_k(int conv, int N, int K);
void ief_gemm_x3_set_variant(int bits);   /* A/B measurements: bit 0 clear = never the 160-wide tile; bit 1 set = 32-key tiles in the fused attention */
int ief_gemm_f32_bn(int N);   /* output-tile width (64 or 128) the library uses for N columns; the M tile is 128 rows */
/* fused fp32 attention (maps never written): out[b] = softmax(scale q[q_src[b]] k[k_src[b]]^T) v[v_src[b]]; q [B][N][heads*d]
 * (row stride ldq, batch stride sQb; likewise k, v over L keys and out); d in {32, 40, 64, 80, 160}; *_src NULL = identity */
typedef struct IefAttnF32Params {
    const float* Q; const float* K; const float* V; float* Out;
    int B, heads, N, L, d;
    int ldq, ldk, ldv, ldo;
    long long sQb, sKb, sVb, sOb;
    float scale;
    const int* q_src; const int* k_src; const int* v_src;
    int x3;                 /* ABI 3: != 0 -> both products on split fp16 operands (csrc/split_x3.hip), softmax in fp32 */
    /* ABI 4 (split-operand kernels only): the output ALSO / ONLY as operand planes for the to_out GEMM (csrc/gemm_x3p.hip):
     * OutP hi plane [B][N][ldp] (batch stride sOPb), lo plane planeO elements further; Out may then be NULL */
    ief_half* OutP;
    long long planeO, sOPb;
    int ldp;
    /* ief_attn_cross_p2p_f32: power-of-two scale of the EDITED maps' hi / lo split, chosen by the host from the plan's
     * coefficients so that scale * max |c1| + |c2| stays below the fp16 range (AttentionReweight: c1 = alpha * equalizer);
     * 0 = 2^14 (maps <= 1) */
    float p_scale;
    /* ABI 4, ief_attn_flash_f32 with x3 != 0 only: Q / K / V given as OPERAND PLANES (the q|k|v GEMM's epilogue writes them) --
     * Qp / Kp / Vp hi planes with the layout of Q / K / V (same ld* and batch strides, in elements), lo planes planeQ / planeK /
     * planeV elements further; Q / K / V are then ignored.  The kernel stages K / V tiles by LDS-DMA and splits nothing
     * (attn_flash_x3p_kernel, csrc/split_x3.hip); d in {40, 64, 80}. */
    const ief_half* Qp; const ief_half* Kp; const ief_half* Vp;
    long long planeQ, planeK, planeV;
    const void* zeros;      /* >= 16 bytes of device zeros (source of key rows past L) when Qp is set */
    /* x3 != 0, fp32 Q / K / V (not planes) only: optional [B][heads][N] receiving the row log-sum-exp in log2 units
     * (max + log2 sum exp2) that ief_attn_bwd_x3 consumes; null: not written */
    float* lse;
} IefAttnF32Params;
int ief_attn_flash_f32(const IefAttnF32Params* p, void* stream);
int ief_softmax_rows_f32(float* x, long long rows, int L, void* stream);
/* P'[w] = c1[w] * sum_v P_src[v] M[v][w] + c2[w] * P_tgt[w] in place on maps [B*heads][N][L], L <= 96; MT fp32
 * [slots][96][96] (M transposed, zero padded), coef fp32 [slots][2][96]; edit_src / edit_slot as IefCrossParams */
int ief_p2p_cross_edit_f32(float* P, const int* edit_src, const int* edit_slot, const float* MT, const float* coef, int B,
                           int heads, int N, int L, void* stream);
/* the four launches above (scores, softmax, edit, apply) as ONE: cross-attention over L <= 96 keys with the map edit applied to
 * maps that stay in registers; split-operand arithmetic (p->x3 is not consulted), d in {40, 64, 80, 160}, no batch-row indirection
 * (q_src / k_src / v_src NULL); edit_src NULL: plain attention (csrc/cross_p2p_x3.hip) */
int ief_attn_cross_p2p_f32(const IefAttnF32Params* p, const int* edit_src, const int* edit_slot, const float* MT, const float* coef,
                           void* stream);
int ief_groupnorm_silu_f32(const float* x, const float* x2, int C1, int C2, float* out, const float* gamma, const float* beta,
                           int B, int HW, int groups, float eps, int silu, void* stream);
/* the same operator in three row-streaming launches (per-run channel statistics, Chan merge per group, apply); C1, C2
 * multiples of 4; ws: ief_groupnorm_f32_ws_floats(B, HW, C1 + C2) floats of scratch the caller owns until the stream passes */
long long ief_groupnorm_f32_ws_floats(int B, int HW, int C);
int ief_groupnorm_silu_f32_ws(const float* x, const float* x2, int C1, int C2, float* out, const float* gamma, const float* beta,
                              int B, int HW, int groups, float eps, int silu, float* ws, long long ws_floats, void* stream);
/* ABI 4 (added late in round 4; additive): the same operator with the (image, group) slab held in registers: ONE launch, one workgroup
 * of 1024 threads per (image, group), the input read once; out (fp32, nullable) and / or outp (operand planes hi / lo, lo plane
 * `plane` elements further; nullable).  Shapes: ief_groupnorm_reg_fits(C1, C2, HW, groups) != 0 (channels per group even and <= 512,
 * C1 even, HW x channels per group <= 40960).  The small-batch form (UNet batch 1 / 2: DDIM inversion, the null-text loop) of
 * /root/reference's GroupNorm + SiLU of ResnetBlock2D (pnp/model/register.py:149-158 runs them through torch). */
int ief_groupnorm_reg_fits(int C1, int C2, int HW, int groups);
int ief_groupnorm_silu_reg(const float* x, const float* x2, int C1, int C2, float* out, ief_half* outp, long long plane,
                           const float* gamma, const float* beta, int B, int HW, int groups, float eps, int silu, void* stream);
/* dx (, dx2) of the GroupNorm (+SiLU) above given dy; `add` (nullable) is summed into the result; row-streaming, five launches;
 * same shape rules as ief_groupnorm_silu_f32_ws, every pointer 16-byte aligned; ws: ief_groupnorm_bwd_f32_ws_floats floats */
long long ief_groupnorm_bwd_f32_ws_floats(int B, int HW, int C);
int ief_groupnorm_bwd_f32_ws(const float* x, const float* x2, int C1, int C2, const float* dy, const float* add, float* dx, float* dx2,
                             const float* gamma, const float* beta, int B, int HW, int groups, float eps, int silu, float* ws,
                             long long ws_floats, void* stream);
int ief_layernorm_f32(const float* x, float* out, const float* gamma, const float* beta, long long rows, int C, float eps,
                      void* stream);
/* ABI 4: the two normalisations as PRODUCERS of operand planes (hi = fp16(y), lo = fp16(y - hi), lo `plane` elements after hi;
 * csrc/gemm_x3p.hip consumes them): GroupNorm (+SiLU) in the three row-streaming launches (out: optional fp32 copy), LayerNorm
 * over rows of C <= 2560, C % 4 == 0 */
int ief_groupnorm_silu_x3p_ws(const float* x, const float* x2, int C1, int C2, float* out, ief_half* outp, long long plane,
                              const float* gamma, const float* beta, int B, int HW, int groups, float eps, int silu, float* ws,
                              long long ws_floats, void* stream);
/* one launch (a few workgroups per (image, group), each streaming the group's slab for the statistics): small tensors */
int ief_groupnorm_silu_x3p_small(const float* x, const float* x2, int C1, int C2, ief_half* outp, long long plane, const float* gamma,
                                 const float* beta, int B, int HW, int groups, float eps, int silu, void* stream);
int ief_layernorm_x3p(const float* x, ief_half* outp, long long plane, const float* gamma, const float* beta, long long rows, int C,
                      float eps, void* stream);
int ief_add_f32(const float* a, const float* b, float* out, long long n, void* stream);
int ief_silu_f32(const float* x, float* out, long long n, void* stream);
/* hidden * gelu(gate) on the interleaved FF1 layout ([8 hidden | 8 gate] groups): pre [rows][2 Ch] -> out [rows][Ch] */
int ief_geglu_il_f32(const float* pre, float* out, long long rows, int Ch, void* stream);
int ief_timestep_embedding_f32(const float* t, float* out, int B, int dim, void* stream);
int ief_gather_rows_f32(const float* in, float* out, const int* src, int B, long long row_elems, void* stream);
/* boundary convolutions with fp32 activations: fp32 NCHW latents <-> fp32 NHWC; w fp32 [3][3][Cin][Cout] / [Cout][3][3][C] */
int ief_conv_in_f32act(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int H, int Wd, int Cout,
                       void* stream);
int ief_conv_out_f32act(const float* x, const float* w, const float* bias, float* out, int B, int C, int H, int Wd, int Cout,
                        void* stream);
/* uint8 image epilogue of latent2image (/root/reference/p2p/model/sd_utils.py:85-88): fp32 NCHW in [-1, 1] -> uint8 NHWC,
 * (x / 2 + 0.5).clamp(0, 1) * 255 truncated */
int ief_image_u8(const float* x, unsigned char* out, int B, int C, int H, int Wd, void* stream);

/* ------------------------------------------------------------------ ABI 4: split-operand contractions on PRE-SPLIT PLANES
 * (csrc/gemm_x3p.hip).  An activation tensor x [rows][C] of the f16x3 mode travels between kernels as two fp16 planes
 *     hi = fp16(x),  lo = fp16(x - hi)            (activation scale 1; weights: scale 2^8, ief_x3_split_weights)
 * the same 4 bytes per element as fp32: the PRODUCER of an activation (GroupNorm / LayerNorm apply, the GEGLU epilogue, the
 * attention epilogues, a GEMM epilogue) writes the planes, so the GEMM that consumes it stages BOTH operands by LDS-DMA
 * (global_load_lds) and its K loop holds no vector work at all: per 32-deep K tile three v_mfma_f32_16x16x32_f16 per fragment
 * pair (Al Bh + Ah Bl + Ah Bh, fp32 accumulate).  A plane pair is addressed as (hi pointer, element offset of the lo plane).
 * Same operator set as IefGemmParams (dense; 3x3 implicit GEMM with channel concat, nearest-2x, stride 2, fused 1x1 range),
 * replaces the same reference call sites: /root/reference/p2p/model/register.py:33-54 (linears),
 * /root/reference/pnp/model/register.py:139-175 (ResnetBlock2D convolutions + shortcut + temb / skip adds). */
typedef struct IefGemmX3pParams {
    const ief_half* A;  long long planeA;     /* dense: [M][lda] hi plane; conv: NHWC source 1 (C1 channels)            */
    const ief_half* A2; long long planeA2;    /* conv: NHWC source 2 of a channel concat (C2 channels) or NULL            */
    const ief_half* E1; long long planeE1;    /* conv: fused 1x1 range, sources sampled at the output pixel (CE1, CE2)     */
    const ief_half* E2; long long planeE2;
    const ief_half* W;  long long planeW;     /* [N][ldw] hi plane of the weight (scale w_scale)                           */
    float* Out;                               /* fp32 [M][ldo] or NULL                                                     */
    ief_half* OutP; long long planeO;         /* planes of the result, [M][ldp] (scale 1) or NULL; at least one of the two */
    const float* bias;                        /* [N] or NULL */
    const float* rowvec;                      /* [M / rows_per_batch][N] or NULL */
    const float* residual;                    /* fp32 [M][ldr] or NULL */
    int M, N, K;
    int lda, ldw, ldo, ldp, ldr;
    int conv, H, Wd, C1, C2, Ho, Wo, stride, ups, batch_images, pad_hi_only, CE1, CE2;
    int rows_per_batch;
    float out_scale;                          /* out = (acc * inv_scale + bias + rowvec + residual) * out_scale */
    float inv_scale;                          /* 1 / (activation scale * weight scale) */
    int tile;                                 /* 1: 128x160 (8 waves), 2: the same + 4 loader waves, 3: 128x80 (4 waves), 4: 256x160 (8 waves),
                                                 5: 64x160 (4 waves), 6: 128x64 (4 waves; widths that are multiples of 64 only), 7: 128x160 on 4 waves (two workgroups per CU), 8: 256x320 (8 waves of 64 x 160; the fewest staged bytes per FLOP: wide N only); 11 / 12: conv3x3_halo_x3p (256x80, 8 waves + 4 loader waves;
                                                 plain 3x3 stride 1 pad 1, rows of <= 64 pixels; 12: nearest-2x fused) */
    int splits;                               /* split-K over grid.y: fp32 slabs ws[splits][M][N] summed in slab order by a second launch */
    float* ws;
    int geglu;                                /* weight rows interleaved [8 hidden | 8 gate]: out [M][N / 2] = hidden * gelu(gate) */
    const void* zeros;                        /* >= 16 bytes of device zeros (source of out-of-range rows / padded taps) */
    /* LayerNorm folded into the NEXT linear (BasicTransformerBlock.norm1/2/3 -> to_q|k|v / to_q / ff.net[0]), in the (count, mean,
     * M2) form merged Chan-style (no E[x^2] - mean^2 cancellation):
     *   producer: rstat_out [M][slots][2] = (mean, M2) of each output row over every 80- (tile 6: 64-) column slice one wave owns;
     *             slots = N / slice width (ief_gemm_x3p_rstat_slots); needs an output (fp32 or planes) and no split-K / GEGLU;
     *   consumer: rstat_in = the producer's rstat_out over THIS launch's K (rstat_slots slices of rstat_cnt columns each); the
     *             launch runs on the planes of the RAW residual stream with W gamma as weight and computes
     *             rstd (acc - mean colsum[n]) + bias[n], colsum[n] = sum_k (W gamma)[n][k] as the planes hold it, bias = b + W beta.
     * cstat_out [ceil(M / BM)][N][2] = (mean, M2) of each output channel over the rows of an M tile (GroupNorm statistics from
     * the producer; M % BM == 0, no split-K). */
    float* rstat_out;
    float* cstat_out;
    const float* rstat_in;
    const float* colsum;
    int rstat_slots, rstat_cnt;
    float ln_eps;
} IefGemmX3pParams;
int ief_gemm_x3p(const IefGemmX3pParams* p, void* stream);
int ief_gemm_x3p_tile_bm(int tile);
int ief_gemm_x3p_tile_bn(int tile);
/* columns one wave owns in a tile (the width of a row-statistics slice): 80, tile 6: 64 */
int ief_gemm_x3p_tile_wn(int tile);
/* fp32 x [rows][ldx] -> planes hi / lo [rows][ldp] (lo plane `plane` elements after hi); C % 4 == 0, 16-byte aligned rows */
int ief_x3_split_act(const float* x, ief_half* planes, long long plane, long long rows, int C, int ldx, int ldp, float scale,
                     void* stream);

/* ------------------------------------------------------------------ activation gradients of the fp32-storage modes
 * (csrc/backward_f32.hip): the reverse pass of null-text inversion / Pix2Pix-zero at the reference's precision
 * (/root/reference/p2p/inversion/nti.py:15-33 and /root/reference/pix2pix-zero/model/sd_utils.py:160-174 run torch autograd in
 * fp32).  Same formulas as the fp16 entry points below; attention gradients are assembled by the host from ief_gemm_f32
 * products on MATERIALISED fp32 maps plus ief_softmax_bwd_rows_f32 / ief_transpose_batched_f32. */
int ief_groupnorm_bwd_f32(const float* x, const float* x2, int C1, int C2, const float* dy, const float* add, float* dx, float* dx2,
                          const float* gamma, const float* beta, int B, int HW, int groups, float eps, int silu, void* stream);
int ief_layernorm_bwd_f32(const float* x, const float* dy, const float* add, float* dx, const float* gamma, long long rows, int C,
                          float eps, void* stream);
int ief_geglu_il_bwd_f32(const float* pre, const float* dy, float* dpre, long long rows, int Ch, void* stream);
int ief_zero_insert2x_f32(const float* in, float* out, int B, int H, int W, int C, void* stream);   /* [B,H,W,C] -> [B,2H,2W,C] */
int ief_pool2x2_sum_f32(const float* in, float* out, int B, int H, int W, int C, void* stream);     /* [B,2H,2W,C] -> [B,H,W,C] */
int ief_conv_out_bwd_f32w(const float* d_eps, const float* w, float* dh, int B, int C, int H, int W, int Cout, void* stream);
/* in place on dP [rows][L]: dS = scale * P o (dP - sum_j dP_j P_j) */
int ief_softmax_bwd_rows_f32(const float* P, float* dP, long long rows, int L, float scale, void* stream);
int ief_transpose_batched_f32(const float* in, float* out, int R, int N, int L, void* stream);        /* [R][N][L] -> [R][L][N] */
/* Pix2Pix-zero map objective on materialised maps: dP = gcoef (P - ref); loss[block] = loss_coef * sum (P - ref)^2 over the
 * block's rows (ief_map_loss_rows_blocks(rows) blocks, fixed order) */
int ief_map_loss_rows_blocks(long long rows);
int ief_map_loss_rows_f32(const float* P, const float* ref, float* dP, float* loss, long long rows, int L, float gcoef,
                          float loss_coef, void* stream);
/* torch.optim.Adam step with an fp32 gradient (g = grad * stats[1]); hyper = {lr, beta1, beta2, eps}; step[0] += 1 */
int ief_nti_adam_f32g(float* param, float* m, float* v, const float* grad, const float* stats, const float* hyper, int* step, int n,
                      void* stream);

/* ------------------------------------------------------------------ null-text inversion: activation gradients
 * The reference optimises the unconditional embedding with torch autograd through the whole UNet
 * (/root/reference/p2p/inversion/nti.py:15-33: `uncond_embeddings.requires_grad`, `loss.backward()`,
 * `optimizer.step()`), weights frozen.  Only gradients w.r.t. ACTIVATIONS are therefore needed:
 *   - linear / conv3x3 data gradients are ief_gemm_f16 / ief_conv3x3_f16 on transposed (and tap-flipped) weights;
 *   - the entry points below are the backward of everything else on the path.
 * Gradients are fp16 and carry a global scale chosen by ief_nti_loss_grad_f32 (undone in ief_nti_adam_f32).
 */
typedef struct IefAttnBwdParams {
    const ief_half* Q; const ief_half* K; const ief_half* V; const ief_half* dO;
    const float* lse;     /* [B][heads][N] from ief_attn_flash_f16 */
    const float* delta;   /* [B][heads][N] from ief_attn_bwd_delta_f32 */
    ief_half* dQ; ief_half* dK; ief_half* dV;
    int B, heads, N, L, d;
    int ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    float scale;
    float ds_mul;         /* dS is multiplied by this before its fp16 rounding (undone in fp32); > 0 */
    /* dK/dV with few keys (cross-attention: 77) own too few workgroups: kv_splits > 1 cuts the query range into that many
     * slices, each writing fp32 partials to ws (>= 2 * kv_splits * B * L * heads * d floats), summed in a fixed order. */
    int kv_splits;
    float* ws;
} IefAttnBwdParams;
/* delta[b][h][n] = sum_d dO * O  (the softmax-backward row term) */
int ief_attn_bwd_delta_f32(const ief_half* O, const ief_half* dO, float* delta, int B, int heads, int N, int d,
                           int ldo, int lddo, void* stream);
/* what: 1 = dQ, 2 = dK and dV, 3 = all.  d in {32,40,64,80,160}; any N, L >= 1. */
int ief_attn_bwd_f16(const IefAttnBwdParams* p, int what, void* stream);
/* ---- the same gradients in the split-operand ("f16x3") mode: fp32 q / k / v / dO in, fp32 dQ / dK / dV out, P and dS recomputed per
 * tile from the forward's lse (IefAttnF32Params.lse) and never written to HBM; every product on hi / lo fp16 halves (three MFMAs);
 * csrc/attention_bwd_x3.hip.  Rows of batch b start at b * rows * ld (as IefAttnBwdParams).  d in {40, 64}.  Replaces, on the
 * fp32-storage reverse pass, the materialised maps of /root/reference/p2p/model/register.py:43-51's backward. */
typedef struct IefAttnBwdF32Params {
    const float* Q; const float* K; const float* V; const float* dO;
    const float* lse;     /* [B][heads][N], log2 units */
    const float* delta;   /* [B][heads][N] from ief_attn_bwd_delta_f32in */
    float* dQ; float* dK; float* dV;
    int B, heads, N, L, d;
    int ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    float scale;
    float ds_mul;         /* dS is multiplied by this before its hi / lo split (undone in fp32; |dS ds_mul| > 65504 gives non-finite gradients); > 0 */
} IefAttnBwdF32Params;
int ief_attn_bwd_delta_f32in(const float* O, const float* dO, float* delta, int B, int heads, int N, int d, int ldo, int lddo,
                             void* stream);
/* what: 1 = dQ, 2 = dK and dV, 3 = all */
int ief_attn_bwd_x3(const IefAttnBwdF32Params* p, int what, void* stream);
/* Pix2Pix-zero cross-attention guidance (/root/reference/pix2pix-zero/model/sd_utils.py:166-173): for one cross-attention
 * module, loss = mean over (batch, head) of sum_{n,j} (P - ref)^2 with P = softmax(scale Q K^T), and its gradient w.r.t.
 * Q:  dQ (+)= gcoef * scale * dS K,  dS = P * (e - sum_j e_j P_j),  e = P - ref.  The caller passes
 * gcoef = 2 / (B heads) * gradient scale.  ref: fp16 [B*heads][N][L] contiguous (what ief_attn_probs_f16 wrote during
 * the reference pass); loss (optional): one partial per workgroup, [B][heads][ief_map_loss_blocks(N, d)] floats, each
 * already multiplied by loss_coef; L <= 96; d in {32,40,64,80,160}. */
typedef struct IefMapLossParams {
    const ief_half* Q; const ief_half* K; const ief_half* ref;
    ief_half* dQ;
    float* loss;
    int B, heads, N, L, d;
    int ldq, ldk, lddq;
    float scale, gcoef, loss_coef;
    int accumulate;       /* 1: add to the gradient already in dQ; 0: overwrite */
} IefMapLossParams;
int ief_attn_map_loss_bwd_f16(const IefMapLossParams* p, void* stream);
/* workgroups per (batch, head) of the kernel above = loss partials per (batch, head) */
int ief_map_loss_blocks(int N, int d);
/* y[i] += a * x[i] (fp32): the plain SGD step on the UNet input (sd_utils.py:160,174) */
int ief_axpy_f32(float* y, const float* x, float a, long long n, void* stream);
/* GroupNorm(+SiLU) whose statistics were left by the kernels that produced x (and x2): cstat1 / cstat2 are those launches'
 * IefGemmParams.cstat_out, bm1 / bm2 their M-tile heights (HW % bm == 0: no tile straddles two images).  One small
 * launch folds the per-tile partials into (mean, rstd) per (batch, group) in `stats` (B * groups * 2 floats), one applies:
 * the statistics pass over x is gone.  replaces the same reference lines as ief_groupnorm_silu_f16. */
int ief_groupnorm_cstat_f16(const ief_half* x, const ief_half* x2, int C1, int C2, ief_half* out, const float* gamma,
                            const float* beta, const float* cstat1, int bm1, const float* cstat2, int bm2, float* stats,
                            int B, int HW, int groups, float eps, int apply_silu, void* stream);
/* BM of a tile id (see ief_gemm_tile_bn) */
int ief_gemm_tile_bm(int tile_hint);
/* GroupNorm(+SiLU) backward w.r.t. the input: x (+x2 channel concat) is the forward INPUT, dy [B][HW][C1+C2] the gradient
 * of the output, `add` (optional, same shape as dy) is added to the result; dx [B][HW][C1], dx2 [B][HW][C2].
 * stats: (mean, rstd) [B][groups][2] saved by the forward, or NULL (recomputed); partial: scratch of
 * >= B * ief_gn_splits(HW) * groups * 2 floats, or NULL (forces the one-workgroup-per-group kernel). */
int ief_groupnorm_bwd_f16(const ief_half* x, const ief_half* x2, int C1, int C2, const ief_half* dy, const ief_half* add,
                          ief_half* dx, ief_half* dx2, const float* gamma, const float* beta, const float* stats,
                          float* partial, int B, int HW, int groups, float eps, int apply_silu, void* stream);
/* LayerNorm backward w.r.t. the input (+ optional `add`) */
int ief_layernorm_bwd_f16(const ief_half* x, const ief_half* dy, const ief_half* add, ief_half* dx, const float* gamma,
                          int rows, int C, float eps, void* stream);
/* GEGLU on the interleaved projection layout of the fused FF1 GEMM ([8 hidden | 8 gate] column groups):
 * forward out[r][8g+e] = pre[r][16g+e] * gelu(pre[r][16g+8+e]) and its backward w.r.t. pre */
int ief_geglu_il_f16(const ief_half* pre, ief_half* out, int rows, int Ch, void* stream);
int ief_geglu_il_bwd_f16(const ief_half* pre, const ief_half* dy, ief_half* dpre, int rows, int Ch, void* stream);
/* stride-2 conv data gradient = stride-1 conv of the zero-inserted gradient: out[b][2i][2j] = in[b][i][j], else 0 */
int ief_zero_insert2x_f16(const ief_half* in, ief_half* out, int B, int H, int W, int C, void* stream);
/* nearest-2x upsample backward: out[b][i][j] = sum of the 2x2 block of in [B][2H][2W][C] */
int ief_pool2x2_sum_f16(const ief_half* in, ief_half* out, int B, int H, int W, int C, void* stream);
/* conv_out data gradient: d_eps fp32 NCHW [B][Cout][H][W], w fp16 [Cout][3][3][C] -> dh fp16 NHWC [B][H][W][C] */
int ief_conv_out_bwd_f32(const float* d_eps, const ief_half* w, ief_half* dh, int B, int C, int H, int W, int Cout,
                         void* stream);
/* NTI objective (nti.py:24-27): rec = DDIM step of x with eps = eps_u + g (eps_c - eps_u); loss = mean((rec - target)^2).
 * coef: DEVICE fp32 {a_from, a_to, g} as for ief_cfg_ddim_step_f32.  Writes stats[0] = loss, stats[1] = factor, and the NORMALISED gradient
 * d_eps = (rec - target) / max|rec - target| * grad_scale, with d loss / d eps_u = d_eps * factor.  n <= 2^20. */
int ief_nti_loss_grad_f32(const float* eps_u, const float* eps_c, const float* x, const float* target, const float* coef,
                          float* d_eps, float* stats, int n, float grad_scale, void* stream);
/* torch.optim.Adam step (nti.py:17,29-30) on fp32 param [n]: g = grad16 * stats[1]; state m, v, DEVICE int step count
 * (incremented here), hyper: DEVICE fp32 {lr, beta1, beta2, eps}; also writes the fp16 copy the next forward reads. */
int ief_nti_adam_f32(float* param, float* m, float* v, const ief_half* grad16, const float* stats, const float* hyper,
                     int* step, ief_half* param16, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IEF_HIP_H */
