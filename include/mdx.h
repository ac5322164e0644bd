/*
 * mdx.h — C-ABI of libmdx.so, the MI355X (gfx950) kernel library behind
 * magicdrive_amd's drop-in for MagicDrive's per-step denoiser.
 *
 * Boundary: the reference has no FFI of its own — its hot path is
 * Python calling torch ATen / xformers.  The lowest replaceable interface is
 *   - xformers.ops.memory_efficient_attention(q, k, v, attn_bias, p, scale, op)
 *       third_party/xformers/xformers/ops/fmha/__init__.py:115-196, called from
 *       third_party/diffusers/src/diffusers/models/attention_processor.py:1165-1171
 *   - ATen conv2d / addmm / native_group_norm / native_layer_norm / gelu / silu as
 *       invoked by third_party/diffusers/src/diffusers/models/{resnet.py:590-640,
 *       attention.py:259-280, transformer_2d.py:276-315, embeddings.py:24-64}
 *   - the scheduler update third_party/diffusers/src/diffusers/schedulers/
 *       scheduling_ddim.py:325-445 and CFG combine magicdrive/pipeline/
 *       pipeline_bev_controlnet.py:426-431.
 * Every entry point below replaces one of those call sites (cited per function).
 *
 * Conventions
 *   - plain C: pointers, 64-bit integers and doubles only; no torch types.  Every
 *     descriptor field is 8 bytes wide so a ctypes.Structure mirrors it 1:1.
 *   - all device pointers are caller-owned HBM; nothing is allocated or freed here
 *     except graph handles; no implicit synchronisation: work is enqueued on the
 *     hipStream_t passed as `stream` (void*), so calls are hipGraph-capturable.
 *   - activations are channels-last: a feature map is [B][H][W][C] == a token
 *     matrix [B*H*W][C]; "ld*" are row (pixel/token) strides in ELEMENTS.
 *   - bf16 = upper 16 bits of IEEE fp32, round-to-nearest-even on store; all
 *     accumulation, softmax, normalisation statistics and scheduler math in fp32.
 *   - return 0 on success, negative MDX_E* otherwise; never throws.
 *     mdx_last_error() returns a thread-local message for the last failure.
 */
#ifndef MDX_H_
#define MDX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDX_OK 0
#define MDX_EINVAL (-1)      /* bad descriptor (shape/alignment/unsupported size) */
#define MDX_ELAUNCH (-2)     /* hip launch / runtime error */
#define MDX_EUNSUPPORTED (-3)

#define MDX_ABI_VERSION 10

/* ---- epilogue flags shared by GEMM / conv ------------------------------- */
#define MDX_EPI_NONE 0
#define MDX_EPI_GEGLU 1      /* W rows interleaved [32 value | 32 gate]; C has N/2 cols:
                                C = value * gelu_erf(gate)   (attention.py:259-280) */
#define MDX_EPI_SILU 2       /* C = silu(acc + bias)  (map_embedder.py:66-76, bbox_embedder.py:145-152) */

/*
 * mdx_gemm_bf16 — C[M,N] = epi(A[M,K] · W[N,K]^T + bias[N] + temb[row(m),N]) + R[M,N]
 * Replaces ATen addmm/mm for every nn.Linear on the path and every 1x1 conv
 * (attention_processor.py:141-157 to_q/k/v/out, attention.py:200-216 FeedForward,
 * transformer_2d.py:155-165 proj_in/out, blocks.py:81-83 connector,
 * unet_addon_rawbox.py:221-272 zero-convs).
 * batch > 1: independent problems, pointer p advances by s*[batch index] elements
 * (used to emit V^T = W_v · X_b^T per view with sA = 0).
 * temb: optional fp32 table; row added to output row m is
 *        temb[sel*temb_sel_stride + (m / rows_per_b)*temb_b_stride + n]
 * where sel = *sel_ptr (device int32, 0 if null) — lets a captured graph pick the
 * current step's row of a table precomputed for all DDIM steps.
 * splitk: 0 = let the library choose (it splits K for small-M/huge-K shapes so the launch fills
 * 256 CUs), 1 = never, >1 = forced.  Splitting needs ws: fp32 [splitk][M][N] scratch of ws_bytes.
 * Requirements: K % 8 == 0, lda/ldw % 8 == 0, 16-byte aligned A/W; ldc,ldr % 4 == 0.
 */
typedef struct MdxGemmDesc {
    const void* A; const void* W; void* C; const void* R;
    const float* bias; const float* temb; const int32_t* sel_ptr; float* ws;
    int64_t M, N, K;
    int64_t lda, ldw, ldc, ldr;
    int64_t batch, sA, sW, sC, sR;
    int64_t temb_sel_stride, temb_b_stride, rows_per_b;
    int64_t epilogue, splitk;
    int64_t c_is_f32;           /* 1: C (and R) are fp32 instead of bf16 */
    int64_t ws_bytes;           /* size of ws; split-K is reduced (or disabled) to fit */
    /* Optional transposed second output (fused q/k/v projection: the attention kernel wants V^T[view][channel][token]):
     * raw columns n >= vt_from are NOT written to C but to Vt[b][n - vt_from][t] with b = m / vt_T, t = m % vt_T
     * (element strides vt_stride per view, vt_ld per channel).  C then has vt_from columns.  Needs K == 320, vt_from % 128 == 0,
     * vt_T % 8 == 0, M % 8 == 0, 16-byte aligned Vt rows, no bias / epilogue on those columns (diffusers' to_v has none) —
     * it is implemented by the weight-stationary kernel only; anything else is rejected with MDX_EINVAL.  Vt = NULL: off. */
    void* Vt;
    int64_t vt_from, vt_T, vt_ld, vt_stride;
    /* Optional LayerNorm of the A rows, fused into the projection that consumes them (BasicTransformerBlock: norm1 -> attn1.to_q/k/v,
     * norm2 -> attn2.to_q, attention.py:85-120; BasicMultiviewTransformerBlock norm4 -> attn4, blocks.py:190-205).  ln_eps > 0: A holds
     * the RAW rows (K = the whole row); the caller has folded the affine part into the operands,
     *     W' = W diag(gamma),  bias' = bias + W beta,  ln_csum[n] = sum_k W'[n][k]   (sum of the 16-bit W' values, fp32),
     * and the result is  C[m][n] = rstd_m (sum_k A[m][k] W'[n][k] - mean_m ln_csum[n]) + bias'[n]  (+ R) with mean / rstd of row m over K
     * (biased variance, as torch.nn.LayerNorm).  The K = 320 weight-stationary kernel takes the statistics from the rows it streams
     * (the tokens are read once; no normalised copy exists); every other route first writes (A - mean) rstd to ln_scratch ([M][lda]
     * 16-bit, 16-byte aligned) and multiplies that — same weights, same result up to the rounding of the normalised copy.
     * The fused route takes a plain epilogue (optionally Vt); one batch, no split-K.  ln_eps = 0: off. */
    double ln_eps;
    const float* ln_csum;
    void* ln_scratch;
    /* Row statistics out of the PRODUCER's epilogue (ABI 9).  The LayerNorms of a transformer block normalise tensors that a
     * C x C projection has just written (proj_in -> norm1, attn1.to_out + residual -> norm2, attn2.to_out + residual -> norm4 / norm3,
     * connector(attn4.to_out) + residual -> norm3: attention.py:123-200, blocks.py:190-222).  rowstat_out != NULL: besides C the GEMM writes
     *     rowstat_out[(p * M + m) * 2 + {0, 1}] = (sum, sum of squares) of the STORED values C[m][n] (rounded to the 16-bit type, residual added)
     * over the columns n of part p, for rowstat_parts parts that together cover [0, N); parts a route does not need are written as zeros, so a
     * consumer simply adds all parts.  Fixed summation order (deterministic).  Plain epilogue, one batch, 16-bit C, no Vt; rowstat_parts >= 1.
     * The K = 320 weight-stationary kernel emits them from its store phase (one part per 128-column tile); every other route runs a small
     * statistics kernel over C behind the GEMM (part 0 = the whole row).
     * ln_stats != NULL (with ln_eps > 0): the fused LayerNorm takes mean / rstd of row m from these sums — written by the producer of A through
     * its rowstat_out, ln_stats_parts parts of M rows — instead of recomputing them from the rows it streams (in every N-tile's workgroup: 8x for
     * a fused q/k/v projection).  With given statistics the GEGLU epilogue can carry the LayerNorm too (norm3 -> ff.net.0).  Routes that do not
     * normalise in-kernel ignore ln_stats and normalise into ln_scratch as before. */
    float* rowstat_out;
    int64_t rowstat_parts;
    const float* ln_stats;
    int64_t ln_stats_parts;
    /* Optional second copy of W in MFMA-fragment order (ABI 10), for the W-direct persistent kernel (csrc/gemm_xd.hip: the weights go
     * global -> registers past the LDS, the finished tile is stored under the next tile's main loop):
     *     Wq[n / 16][k / 32][lane][8]  with lane = ((k % 32) / 8) * 16 + n % 16,  element = W[n][k + (0..7)],
     * i.e. the 64 lanes' operands of one 16-column x 32-deep block are one contiguous KiB; N padded with zero blocks to a multiple of 256
     * (roundup(N, 256) * K elements; magicdrive_amd/packing.py: pack_wq).  GEGLU: W (and so Wq) rows in the [32 value | 32 gate] order.
     * The library uses it when the shape goes to the 256-wide persistent tile and K % 128 == 0, K >= 640 (plain / GEGLU epilogue, optional
     * residual, no temb / Vt / split-K); otherwise it is ignored and W is read.  W must be given either way.  NULL: off. */
    const void* Wq;
} MdxGemmDesc;
int mdx_gemm_bf16(const MdxGemmDesc* d, void* stream);

/*
 * mdx_conv2d_bf16 — channels-last implicit-GEMM convolution on MFMA:
 *   Y[b,oy,ox,co] = epi( Σ_{ky,kx,ci} X[b, oy*sh-ph+ky, ox*sw-pw+kx, ci] · Wt[co][ky][kx][ci]
 *                        + bias[co] + temb[b,co] ) + R[b,oy,ox,co]
 * Replaces ATen conv2d in ResnetBlock2D.conv1/conv2 (resnet.py:590-640, temb add :617-618,
 * residual add :638), Downsample2D (resnet.py:198-222), Upsample2D.conv (resnet.py:165-170),
 * the BEV map encoder (map_embedder.py:66-76).  Weights are pre-packed [Cout][kh][kw][Cin].
 * ldx/ldy/ldr: pixel strides in elements (a tensor may be a channel slice of a wider one).
 * Cin % 8 == 0 required (smaller Cin goes through mdx_conv2d_direct).
 */
typedef struct MdxConvDesc {
    const void* X; const void* Wt; void* Y; const void* R;
    const float* bias; const float* temb; const int32_t* sel_ptr; float* ws;
    int64_t B, Hi, Wi, Cin, Ho, Wo, Cout;
    int64_t kh, kw, sh, sw, ph, pw;
    int64_t ldx, ldy, ldr;
    int64_t temb_sel_stride, temb_b_stride;
    int64_t epilogue, splitk;
    int64_t ws_bytes, reserved1;
} MdxConvDesc;
int mdx_conv2d_bf16(const MdxConvDesc* d, void* stream);

/*
 * mdx_conv2d_direct — small-channel / odd-K convolution or linear on the vector ALU
 * (fp32 accumulate): conv_in (Cin=4, unet_2d_condition.py:262-265), conv_out (Cout=4, :497-500),
 * cam2token (K=189, unet_addon_rawbox.py:106), bbox_proj (K=216, bbox_embedder.py:72).
 * x_is_f32 / y_is_f32 select fp32 I/O (latents and eps stay fp32).
 */
typedef struct MdxConvDirectDesc {
    const void* X; const void* Wt; void* Y; const void* R;
    const float* bias; const float* temb; const int32_t* sel_ptr; void* reserved_p;
    int64_t B, Hi, Wi, Cin, Ho, Wo, Cout;
    int64_t kh, kw, sh, sw, ph, pw;
    int64_t ldx, ldy, ldr;
    int64_t temb_sel_stride, temb_b_stride;
    int64_t epilogue, x_is_f32, y_is_f32, reserved0;
} MdxConvDirectDesc;
int mdx_conv2d_direct(const MdxConvDirectDesc* d, void* stream);

/*
 * mdx_attention_bf16 — fused softmax(Q K^T * scale) V, flash style (no T×T matrix in HBM).
 * Replaces xformers.ops.memory_efficient_attention (fmha/__init__.py:115-196; CUTLASS kernel
 * kernel_forward.h) / F.scaled_dot_product_attention (attention_processor.py:1193-1272) for
 * attn1 (self), attn2 (text+camera+box context) and attn4 (cross-view, blocks.py:106-222).
 *   Q : [B][Tq][..]  element (b,t,h,j) at Q + b*sQ + t*ldq + h*d + j
 *   K : [Bkv][Tk][..] same addressing with sK, ldk
 *   Vt: [Bkv][H*d][ldv]  V transposed — element (b,h,j,t) at Vt + b*sV + (h*d+j)*ldv + t
 *   O : [B][Tq][H*d] with sO, ldo
 * nsrc key/value sources per query batch: for query batch b they are the batches kvmap[b*nsrc + s] (identity if kvmap == NULL).
 *   joint == 0, nsrc ∈ {1,2}: every source has its own softmax and the normalised outputs are SUMMED — neighboring_attn_type "add",
 *     blocks.py:112-121, 213-217 (left + right neighbour; the doubled out-bias is the caller's business);
 *   joint == 1, nsrc ∈ {1..8}: ONE softmax over the concatenation of the sources' keys — "concat" (the two neighbours, blocks.py:122-134)
 *     and "self" (all cameras of the scene, blocks.py:135-138).
 * d % 8 == 0, d <= 160; ldq,ldk,ldv,sQ,sK,sV % 8 == 0; ldo % 4 == 0.
 */
typedef struct MdxAttnDesc {
    const void* Q; const void* K; const void* Vt; void* O;
    const int32_t* kvmap; void* reserved_p;
    int64_t B, H, Tq, Tk, d, nsrc;
    int64_t ldq, sQ, ldk, sK, ldv, sV, ldo, sO;
    double scale;
    int64_t joint;
    int64_t q_prescaled;   /* 1: Q already carries scale * log2(e) (folded into the to_q weights when they were packed): `scale` is ignored,
                            * Q K^T is used as the base-2 exponent directly — lets the head-dim-40 kernel subtract the running maximum
                            * inside the QK MFMA (attention2.hip: FOLD).  0: plain Q, softmax(scale * Q K^T) as in the reference. */
} MdxAttnDesc;
int mdx_attention_bf16(const MdxAttnDesc* d, void* stream);

/*
 * mdx_groupnorm_bf16 — GroupNorm(+SiLU) over channels-last [B][HW][C]
 * (ATen native_group_norm + silu: resnet.py:596-598, 626-630; transformer_2d.py:278;
 *  unet_2d_condition_multiview.py:519-521).  Statistics in fp32, pivot-shifted / Chan-combined (no E[x^2]-E[x]^2
 * cancellation); with a workspace, large maps take a two-stage fully coalesced path (deterministic, no atomics).
 */
typedef struct MdxGroupNormDesc {
    const void* X; void* Y; const float* gamma; const float* beta;
    int64_t B, HW, C, G, ldx, ldy;
    double eps;
    int64_t silu;
    float* ws; int64_t ws_bytes;   /* optional scratch for the two-stage (coalesced) path: B*chunks*G*3 floats */
} MdxGroupNormDesc;
int mdx_groupnorm_bf16(const MdxGroupNormDesc* d, void* stream);

/* mdx_layernorm_bf16 — LayerNorm over the last dim of [M][C] (attention.py:85,104,120; blocks.py:67-71). */
typedef struct MdxLayerNormDesc {
    const void* X; void* Y; const float* gamma; const float* beta;
    int64_t M, C, ldx, ldy;
    double eps;
    int64_t reserved0;
} MdxLayerNormDesc;
int mdx_layernorm_bf16(const MdxLayerNormDesc* d, void* stream);

/* mdx_softmax_rows — Y[r][0..T) = softmax(scale * X[r][0..T)) row by row, fp32 in (the fp32 scores of a Q K^T GEMM), bf16 out;
 * columns T..ldy-1 of Y are written as zeros (so Y can be the K-padded A operand of the P V GEMM).  Used by the VAE decoder's
 * single-head, 512-channel mid-block attention (attention_processor.py:495-558 with upcast_softmax; unet_2d_blocks.py:433-445),
 * whose head dim is outside the fused attention kernel's range. */
typedef struct MdxSoftmaxDesc {
    const float* X; void* Y;
    int64_t rows, T, ldx, ldy;
    double scale;
    int64_t reserved0;
} MdxSoftmaxDesc;
int mdx_softmax_rows(const MdxSoftmaxDesc* d, void* stream);

/* ---- small element-wise kernels ---------------------------------------- */
#define MDX_EW_ADD 1          /* Y[m, :C] += X[m, :C]            (unet_2d_condition_multiview.py:464-488) */
#define MDX_EW_COPY 2         /* Y[m, :C]  = X[m, :C]  (concat halves: unet_2d_blocks.py:1990, 2090) */
#define MDX_EW_UPSAMPLE 3     /* nearest resize [B,Hi,Wi,C] -> [B,Ho,Wo,C] (resnet.py:154-163) */
#define MDX_EW_NCHW_TO_NHWC 4 /* X fp32/bf16 NCHW -> Y NHWC */
#define MDX_EW_NHWC_TO_NCHW 5
#define MDX_EW_SILU 6
#define MDX_EW_SCALE 7        /* Y = X * alpha */
typedef struct MdxEwDesc {
    const void* X; void* Y; const int32_t* ymap; const int32_t* xmap;
    int64_t kind, M, C, ldx, ldy;
    int64_t B, Hi, Wi, Ho, Wo;
    int64_t x_is_f32, y_is_f32;
    double alpha;
} MdxEwDesc;
int mdx_elementwise(const MdxEwDesc* d, void* stream);

/*
 * mdx_fourier_embed — NeRF embedding [x, sin(2^k x), cos(2^k x)]_{k<F} of 3-vectors
 * (magicdrive/networks/embedder.py:15-40) for camera columns (unet_addon_rawbox.py:288-305) and
 * box corners with the masked null blend pos*m + null*(1-m) (bbox_embedder.py:165-176).
 *   X fp32 [n][P][3]  ->  Y bf16 [n][P*(3+6F)] ; mask (uint8 [n]) and null (fp32 [P*(3+6F)]) optional.
 */
typedef struct MdxFourierDesc {
    const float* X; void* Y; const uint8_t* mask; const float* null_feat;
    int64_t n, P, F, ldy;
} MdxFourierDesc;
int mdx_fourier_embed(const MdxFourierDesc* d, void* stream);

/*
 * mdx_gather_rows — Y[i,:] = m[i] ? T[idx[i],:] : null[:]  (class-token lookup with null blend,
 * bbox_embedder.py:179-180; idx may be -1 where mask is 0).
 */
typedef struct MdxGatherDesc {
    const void* T; void* Y; const int64_t* idx; const uint8_t* mask; const void* null_row; void* reserved_p;
    int64_t n, C, ldt, ldy, n_rows, reserved0;
} MdxGatherDesc;
int mdx_gather_rows(const MdxGatherDesc* d, void* stream);

/*
 * mdx_timestep_embedding — sinusoidal timestep features, fp32 math
 * (diffusers/models/embeddings.py:24-64 with flip_sin_to_cos, downscale_freq_shift):
 *   Y[i, :] = [cos(t_i * f_j) | sin(t_i * f_j)]  (flip) , f_j = exp(-ln(max_period) * j / (half - shift))
 */
typedef struct MdxTimeEmbDesc {
    const float* t; float* Y;   /* Y fp32 [n][ldy] */
    int64_t n, dim, flip_sin_to_cos, ldy;
    double freq_shift, max_period;
} MdxTimeEmbDesc;
int mdx_timestep_embedding(const MdxTimeEmbDesc* d, void* stream);

/*
 * mdx_cfg_ddim_step — fused classifier-free-guidance combine + DDIM update, fp32:
 *   eps = eps_u + g (eps_c - eps_u)                      pipeline_bev_controlnet.py:426-431
 *   x0  = (x - sqrt(1-a_t) eps) / sqrt(a_t) ; x <- sqrt(a_prev) x0 + sqrt(1-a_prev) eps
 *                                                        scheduling_ddim.py:379-425 (eta = 0)
 * coef: fp32 [n_steps][4] = {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)}; row = *step_ptr.
 * eps holds [uncond | cond] halves of n elements each when cfg != 0.  After the update the
 * kernel increments *step_ptr (single thread) so a replayed graph walks the table, and also
 * refreshes the model-input copy(ies) `x_in` (duplicated [uncond | cond] for CFG):
 *   xin_ld == 0 : x_in is fp32, flat, same layout as x;
 *   xin_ld  > 0 : x_in is bf16 channels-last with pixel stride xin_ld >= xin_c (x has xin_c channels per
 *                 pixel; the pad channels are never written) — the layout conv_in's MFMA path reads.
 */
typedef struct MdxDdimDesc {
    float* x; const float* eps; const float* coef; int32_t* step_ptr; void* x_in; void* reserved_p;
    int64_t n, cfg;
    double guidance;
    int64_t xin_c, xin_ld;
    /* Given views (StableDiffusionBEVControlNetGivenViewPipeline, magicdrive/pipeline/pipeline_bev_controlnet_given_view.py:263-291,
     * :380-390): gv_mask[i / gv_view_elems] != 0 marks the views whose clean latents gv_cond are known; gv_noise is the initial
     * noise of every view (same layout as x).
     *   gv_mode 1 (conditional_latents_change_every_input): after the update of step s < gv_last_step a given view is replaced by
     *     add_noise(cond, noise, next timestep) = coef[s][2] * cond + coef[s][3] * noise (what the reference does at the top of
     *     the next iteration); the last step's output is kept.
     *   gv_mode 2: the combined noise prediction of a given view is replaced by gv_noise before the update.
     *   gv_mode 0 / gv_mask NULL: off. */
    const float* gv_cond; const float* gv_noise; const uint8_t* gv_mask;
    int64_t gv_mode, gv_view_elems, gv_last_step;
} MdxDdimDesc;
int mdx_cfg_ddim_step(const MdxDdimDesc* d, void* stream);

/*
 * mdx_cfg_unipc_step — fused classifier-free-guidance combine + one UniPC (order <= 2, B(h), predict-x0)
 * predictor-corrector update, fp32 (scheduling_unipc_multistep.py:256-300, 302-405, 407-516, 518-600 — the
 * sampler tools/test.py really uses, misc/test_utils.py:129).  Every quantity of the update is a linear
 * combination with per-step scalar coefficients, computed on the host from the timestep list:
 *   e   = eps_u + g (eps_c - eps_u)
 *   m_t = a x + b e                                   (x0 prediction)
 *   x_c = corr ? cl x_last + c1 m1 + c2 m2 + ct m_t : x   (UniC with the previous step's order)
 *   x  <- px x_c + pt m_t + p1 m1                      (UniP with this step's order)
 *   x_last <- x_c ; m2 <- m1 ; m1 <- m_t
 * coef: fp32 [n_steps][12] = {a, b, corr, cl, c1, c2, ct, px, pt, p1, an, sn}; row = *step_ptr, incremented after.
 * x_last, m1, m2: fp32 state buffers of n elements (zeroed by the caller before the first step).
 * x_in / xin_c / xin_ld as in MdxDdimDesc.
 * gv_* (ABI 8): given views exactly as in MdxDdimDesc — what demo/run_cond_on_view.py runs (its pipe comes from build_pipe, which
 * installs UniPC: magicdrive/misc/test_utils.py:129).  Re-noising is scheduler-independent (scheduling_unipc_multistep.py add_noise:
 * x = sqrt(acp_t) cond + sqrt(1 - acp_t) noise), so gv_mode 1 replaces a given view's PREDICTOR output of step s < gv_last_step by
 * an * cond + sn * noise with (an, sn) = coef[s][10..11] = (alpha, sigma) of the NEXT timestep; the multistep history (x_last, m1,
 * m2) keeps the values computed from the overwritten samples, as the reference's scheduler state does.  gv_mode 2 replaces the
 * combined noise prediction by gv_noise.
 */
typedef struct MdxUniPCDesc {
    float* x; const float* eps; const float* coef; int32_t* step_ptr; void* x_in; float* x_last; float* m1; float* m2;
    int64_t n, cfg;
    double guidance;
    int64_t xin_c, xin_ld;
    const float* gv_cond; const float* gv_noise; const uint8_t* gv_mask;
    int64_t gv_mode, gv_view_elems, gv_last_step;
} MdxUniPCDesc;
int mdx_cfg_unipc_step(const MdxUniPCDesc* d, void* stream);

/* ---- program = array of ops, executed in order on one stream ------------ */
#define MDX_OP_GEMM 1
#define MDX_OP_CONV 2
#define MDX_OP_CONV_DIRECT 3
#define MDX_OP_ATTN 4
#define MDX_OP_GROUPNORM 5
#define MDX_OP_LAYERNORM 6
#define MDX_OP_EW 7
#define MDX_OP_FOURIER 8
#define MDX_OP_GATHER 9
#define MDX_OP_TIMEEMB 10
#define MDX_OP_DDIM 11
#define MDX_OP_UNIPC 12
#define MDX_OP_SOFTMAX 13

/* 16-bit storage / MFMA operand type of an op's activations and weights.  The reference samples in fp16 (magicdrive/misc/test_utils.py:95
 * `weight_dtype = torch.float16`); BASELINE.json's benchmark configuration names bf16.  Every op entry point exists in both builds:
 * mdx_<op>_bf16 / mdx_<op> (bf16) and mdx_<op>_f16 (IEEE fp16) — same descriptors, fp32 accumulation, fp32 side inputs (bias, temb,
 * norm affine, latents).  A program carries the choice per op. */
#define MDX_DTYPE_BF16 0
#define MDX_DTYPE_F16 1

#define MDX_OP_BYTES 512
typedef struct MdxOp {
    int64_t opcode;
    int64_t dtype;                           /* MDX_DTYPE_* */
    unsigned char desc[MDX_OP_BYTES - 16];   /* one of the Mdx*Desc above, zero padded */
} MdxOp;

/* the fp16 build of the op entry points above (identical descriptors and semantics; 16-bit tensors hold IEEE fp16) */
int mdx_gemm_f16(const MdxGemmDesc* d, void* stream);
int mdx_conv2d_f16(const MdxConvDesc* d, void* stream);
int mdx_conv2d_direct_f16(const MdxConvDirectDesc* d, void* stream);
int mdx_attention_f16(const MdxAttnDesc* d, void* stream);
int mdx_groupnorm_f16(const MdxGroupNormDesc* d, void* stream);
int mdx_layernorm_f16(const MdxLayerNormDesc* d, void* stream);
int mdx_elementwise_f16(const MdxEwDesc* d, void* stream);
int mdx_fourier_embed_f16(const MdxFourierDesc* d, void* stream);
int mdx_gather_rows_f16(const MdxGatherDesc* d, void* stream);
int mdx_timestep_embedding_f16(const MdxTimeEmbDesc* d, void* stream);
int mdx_cfg_ddim_step_f16(const MdxDdimDesc* d, void* stream);
int mdx_cfg_unipc_step_f16(const MdxUniPCDesc* d, void* stream);
int mdx_softmax_rows_f16(const MdxSoftmaxDesc* d, void* stream);

/* Run ops[0..n) in order on `stream`.  Stops at the first failing op (returns its code;
 * mdx_last_error() names the op index). */
int mdx_program_run(const MdxOp* ops, int64_t n, void* stream);

/* Capture ops[0..n) into a hipGraph (stream capture on an internal stream) and instantiate it.
 * The pointers inside the ops are baked into the graph; per-replay variation comes only from
 * device memory (sel_ptr / step_ptr tables). */
int mdx_graph_create(const MdxOp* ops, int64_t n, void** graph_out);
int mdx_graph_launch(void* graph, void* stream);
int mdx_graph_destroy(void* graph);

int mdx_abi_version(void);
/* Hash of the kernel sources this library was built from (csrc/Makefile: sha256 over every .hip / .h, first 16 hex digits).  Counter
 * summaries under profiles/ record it; bench.py drops counters whose build differs from the library it is timing. */
const char* mdx_build_id(void);
const char* mdx_last_error(void);
/* Name of the kernel the calling thread's last op launched (the one doing the op's work; which GEMM / conv main loop a
 * descriptor is routed to is decided inside the library by shape).  Measurement aid: bench.py groups its per-launch HIP-event
 * timings by this name so they can be compared with rocprofv3's kernel statistics. */
const char* mdx_last_kernel(void);
/* Tuning / routing switches (magicdrive_amd/csrc/options.h lists every key with its default and meaning, e.g. "GEMM_XL", "XL_BN",
 * "ATTN2"): process-wide named integers that the launchers read on every call.  mdx_set_option returns MDX_EINVAL for an unknown
 * key.  The reference has no counterpart (its kernel choice lives inside xformers' dispatch, ops/fmha/dispatch.py, and cuDNN's
 * heuristics); they exist so that parity tests can force every main loop in-process and a profile can A/B a route.  mdx_option_name(i)
 * enumerates the keys (NULL past the end).  Captured graphs keep the route they were captured with. */
int mdx_set_option(const char* key, int64_t value);
int mdx_get_option(const char* key, int64_t* value_out);
const char* mdx_option_name(int64_t index);

/* Device facts for bench/roofline bookkeeping: out[0]=CU count, out[1]=clock kHz, out[2]=HBM bytes. */
int mdx_device_info(int64_t* out3);

#ifdef __cplusplus
}
#endif
#endif /* MDX_H_ */
