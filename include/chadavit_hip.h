/* chadavit_hip.h -- C ABI of libchadavit_hip.so (gfx950 / MI355X).
 *
 * The reference (nicoboou/chadavit) has no native code: every entry point below replaces a stock
 * PyTorch op *call site* on the DINO pretraining hot path; the
 * reference file:line each one stands in for is cited next to it (paths relative to the reference
 * root).  Conventions:
 *   - plain pointers to DEVICE memory owned by the caller, sizes as int / long long, no torch types;
 *   - `stream` is a hipStream_t (passed as void*); every call is asynchronous on that stream;
 *   - return 0 on success, non-zero on error (1 = bad argument, 2 = unsupported shape,
 *     1000+hipError_t = launch failure); nothing is allocated, nothing throws;
 *   - bf16 tensors are raw uint16 storage ("bf16*" = unsigned short*), row-major, leading dimension in
 *     ELEMENTS; fp32 accumulate everywhere;
 *   - token tensors are RAGGED-PACKED: image i owns rows [cu_seqlens[i], cu_seqlens[i+1]) =
 *     [CLS, ch0 patch0..p-1, ch1 ..., ch(C_i-1) ...]; no padded tokens exist (the reference pads every
 *     image to 10 channels and masks, chada_vit.py:226-239 -- dropping them is exact).
 */
#ifndef CHADAVIT_HIP_H
#define CHADAVIT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned short chada_bf16;

/* ABI version; bumped on any signature change. */
int chadavit_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM  Out[M,N] = epilogue( X[M,K] * W[N,K]^T )     bf16 in, fp32 accumulate  (MFMA 16x16x32)
 * replaces: nn.Linear / in_proj / out_proj / linear1 / linear2 addmm call sites
 *           (chada_vit.py:106-110,115; dino.py:108-110) and their autograd dX products.
 * epilogue:
 *   0 NONE      out = acc + bias
 *   1 RELU      out = relu(acc + bias)                                   (chada_vit.py:115)
 *   2 GELU      out = gelu_erf(acc + bias); aux_out(bf16) <- acc + bias   (dino.py:59-73 nn.GELU)
 *   3 RESID     out = acc + bias + aux[m,n]                               (chada_vit.py:99-100 x + ...)
 *   4 RELUMASK  out = acc * (aux[m,n] > 0)                                (threshold_backward)
 *   5 GELUBWD   out = acc * gelu'(aux[m,n])                               (gelu_backward)
 * bias may be NULL.  out_fp32 != 0 -> Out is float*, else bf16*.  K % 64 == 0, N % 64 == 0.
 * --------------------------------------------------------------------------------------------- */
int chadavit_gemm_nt(const chada_bf16* X, int ldx, const chada_bf16* W, int ldw, void* Out, int ldo,
                     int M, int N, int K, const float* bias, int epilogue, const chada_bf16* aux, int ldaux,
                     chada_bf16* aux_out, int out_fp32, void* stream);

/* Tokenizer GEMM: patches[Mp,K=patch*patch] * Wp[D,K]^T + bias + pos[(m % p), :] + chan[chan_idx[m / p], :]
 * written to packed row m + chan_img[m / p] + 1.     replaces chada_vit.py:128-133 (Conv2d 16/16 +
 * flatten/transpose), :226-236 (split/pad/stack), :245 (+pos), :248-250 (+channel token).
 * pos: fp32 [p, D] (already bicubic-resized for non-224 crops, chada_vit.py:202-217);
 * chan: fp32 [max_channels, D] or NULL (skipped when max_channels != model.max_channels, :248). */
int chadavit_tokenizer_gemm(const chada_bf16* patches, const chada_bf16* Wp, const float* bias, const float* pos,
                            const float* chan, const int* chan_img, const int* chan_idx, chada_bf16* tokens,
                            int Mp, int D, int K, int p, void* stream);

/* im2col for the 1->D, k=stride=patch conv: x fp32 [n_chan, S, S] -> patches bf16 [n_chan*(S/patch)^2, patch*patch]
 * (row = chan*p + r*g + q, col = u*patch + v).  replaces the unfold inside Conv2d (chada_vit.py:128). */
int chadavit_im2col(const float* x, chada_bf16* patches, int n_chan, int S, int patch, void* stream);
/* tokenizer GEMM with the conv unfold folded into its operand staging (no patch buffer in HBM): x [n_chan, S, S] fp32 (16-byte
 * aligned), 16 x 16 patches, p = (S/16)^2; same epilogue / outputs as chadavit_tokenizer_gemm.  Replaces chada_vit.py:128-133 +
 * :226-268 in one launch (+ chadavit_write_cls). */
int chadavit_tokenizer_fused(const float* x, const chada_bf16* Wp, const float* bias, const float* pos, const float* chan,
                             const int* chan_img, const int* chan_idx, chada_bf16* tokens, int n_chan, int S, int D, int p,
                             void* stream);

/* CLS rows: tokens[cu_seqlens[i], :] = cls + pos0        (chada_vit.py:256-265) */
int chadavit_write_cls(chada_bf16* tokens, const int* cu_seqlens, const float* cls, const float* pos0, int B, int D,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * GEMM  C[I,J] (+)= A[T,I]^T * B[T,J]   (reduction over token rows; weight gradients)
 * replaces the autograd dW products of every nn.Linear on the path.  Split over T into `splits`
 * fp32 partial slabs (workspace >= splits*(I*J + I) floats), combined deterministically:
 *   C = (accumulate ? C : 0) + sum_s slab_s ;  colsumA[i] (+)= sum_t A[t,i]   (bias gradient; may be NULL)
 * I % 64 == 0, J % 64 == 0.
 * --------------------------------------------------------------------------------------------- */
int chadavit_gemm_tn(const chada_bf16* A, int lda, const chada_bf16* B, int ldb, float* C, int ldc, float* colsumA,
                     int T, int I, int J, int accumulate, float* workspace, long long workspace_floats, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (biased variance), fp32 statistics.   replaces native_layer_norm
 * (chada_vit.py:96,99,100 eps 1e-5; :281 eps 1e-6) and native_layer_norm_backward.
 * fwd:  y = (x - mean) * rstd * gamma + beta ; mean/rstd [T] saved when non-NULL.
 * bwd:  dx = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma ; dx += dres when dres != NULL;
 *       dgamma/dbeta (+)= column sums, through `workspace` (>= 2*D*ln_bwd_partials floats).
 * D % 4 == 0, D <= 1024.
 * --------------------------------------------------------------------------------------------- */
int chadavit_layernorm_fwd(const chada_bf16* x, const float* gamma, const float* beta, chada_bf16* y, float* mean,
                           float* rstd, int T, int D, float eps, void* stream);
/* two chained LayerNorms in one pass: y1 = LN_a(x); y2 = LN_b(y1)  (a block's norm2 followed by the next block's norm1,
 * chada_vit.py:100 then :96 of the next layer); bit-identical to two chadavit_layernorm_fwd calls. */
int chadavit_layernorm_fwd2(const chada_bf16* x, const float* gamma_a, const float* beta_a, const float* gamma_b,
                            const float* beta_b, chada_bf16* y1, chada_bf16* y2, float* mean1, float* rstd1, float* mean2,
                            float* rstd2, int T, int D, float eps_a, float eps_b, void* stream);
/* The two forward entry points above with the (second) output ALSO emitted as an OCP-MX fp8 operand for the fp8 weight path
 * (yq [T, D] e4m3, ys [D/32, lds] e8m0 with lds >= T): bit-identical to chadavit_mx8_quantize applied to y / y2, without the pass over it.
 * D in {192, 384, 768} (rc 2 otherwise). */
int chadavit_layernorm_fwd_q(const chada_bf16* x, const float* gamma, const float* beta, chada_bf16* y, float* mean, float* rstd,
                             void* yq, void* ys, int lds, int T, int D, float eps, void* stream);
int chadavit_layernorm_fwd2_q(const chada_bf16* x, const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b,
                              chada_bf16* y1, chada_bf16* y2, float* mean1, float* rstd1, float* mean2, float* rstd2, void* y2q,
                              void* y2s, int lds, int T, int D, float eps_a, float eps_b, void* stream);
int chadavit_layernorm_bwd(const chada_bf16* dy, const chada_bf16* x, const float* mean, const float* rstd,
                           const float* gamma, const chada_bf16* dres, chada_bf16* dx, float* dgamma, float* dbeta,
                           int accumulate, int T, int D, float* workspace, void* stream);
int chadavit_layernorm_bwd_partials(void); /* number of partial rows the bwd workspace must hold */
/* Two chained LayerNorm backward passes in one sweep -- the boundary between two post-norm blocks in the backward (autograd of
 * chada_vit.py:96 of layer i and :100 of layer i-1):   dx = LN_a'(dy; x) + dres   then   dz = LN_b'(dx; z)
 * where x = LN_b(z) is layer i's input -- x may be NULL when beta_b is given: it is then rebuilt from z exactly as the forward kernels
 * produce it (one [T, D] read less).  dx is not written (it is rounded to bf16 in registers, as the two-call chain would store
 * it).  (dgamma, dbeta) of both norms are produced (accumulate_* != 0: added).  D in {192, 384, 768} (rc 2 otherwise);
 * workspace >= 2 * chadavit_layernorm_bwd_partials() * 2 * D floats. */
int chadavit_layernorm_bwd_pair(const chada_bf16* dy, const chada_bf16* x, const float* mean_a, const float* rstd_a,
                                const float* gamma_a, const chada_bf16* dres, const chada_bf16* z, const float* mean_b,
                                const float* rstd_b, const float* gamma_b, const float* beta_b, chada_bf16* dz, float* dgamma_a,
                                float* dbeta_a, int accumulate_a, float* dgamma_b, float* dbeta_b, int accumulate_b, int T, int D,
                                float* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Variable-length multi-head self-attention over packed sequences (flash style, never materialises
 * the N x N scores).    replaces nn.MultiheadAttention -> scaled_dot_product_attention with the
 * key-padding mask (chada_vit.py:105-111): softmax(Q K^T / sqrt(dh)) V per image, per head.
 * qkv: bf16 [T, 3*D] rows = [q | k | v], head h uses columns [h*dh, (h+1)*dh) of each third
 * (in_proj_weight row order).  out: bf16 [T, D].  lse: fp32 [H, T] (natural log).
 * dh = D / H: 32, 64, 96, 128, 192, 256 or 384 (forward also 16; its backward is chadavit_attn_bwd_dh16).  96 / 192 / 384 are the
 * benchmark's widths (LDS-DMA kernels); 128 / 256 (embed_dim 256 / 512 with two heads) run the register-staged kernels.
 * work: int32 [n_work, 2] = (image, tile index) built by the host from cu_seqlens; n_work must be a multiple of 8 and entry j
 * is executed on XCD j % 8: keep all tiles of an image at indices of one residue class (they then share that XCD's L2);
 * entries with image < 0 are padding.
 * --------------------------------------------------------------------------------------------- */
int chadavit_attn_fwd(const chada_bf16* qkv, chada_bf16* out, float* lse, const int* cu_seqlens, const int* work,
                      int n_work, int T, int D, int H, void* stream);
/* The same operation on v_mfma_f32_32x32x16_bf16 tiles for head widths 96 / 192 (csrc/attention_m32.hip): half the MFMA
 * instructions per FLOP and a leaner softmax -- the row maximum is taken once per row, from the first key tile, and kept as the
 * exponent reference; rows that leave its range re-run with the online recurrence (same result).  variant: 0 = that (default),
 * 1 = textbook online softmax, 2 = lean softmax with the scale folded into Q.  chadavit_attn_fwd dispatches here for dh 96 / 192.
 * replaces chada_vit.py:105-111 forward. */
int chadavit_attn_fwd_m32(const chada_bf16* qkv, chada_bf16* out, float* lse, const int* cu_seqlens, const int* work, int n_work, int T,
                          int D, int H, int variant, void* stream);

/* Attention of ONE query row per sequence -- the CLS row (first row of each sequence) -- against all keys of its sequence.
 * With return_all_tokens = False (args/pretrain.py:147) only norm(x)[:, 0] leaves ChAdaViT.forward (chada_vit.py:272-289): of the
 * last encoder block's attention (chada_vit.py:105-111) only that row is ever read.  out_cls [B, D] bf16; lse_cls [H, B] fp32.
 * _bwd: gradients of autograd when only the CLS rows carry one: dqkv [T, 3D] is written completely (dQ = 0 off the CLS rows,
 * dK / dV of every row = the CLS query's contribution).  scale = the softmax scale 1/sqrt(D/H); (D/H) % 8 == 0. */
int chadavit_attn_cls_fwd(const chada_bf16* qkv, const int* cu_seqlens, chada_bf16* out_cls, float* lse_cls, int B, int D, int H,
                          float scale, void* stream);
int chadavit_attn_cls_bwd(const chada_bf16* qkv, const int* cu_seqlens, const chada_bf16* out_cls, const chada_bf16* dout_cls,
                          const float* lse_cls, chada_bf16* dqkv, int B, int D, int H, float scale, void* stream);
/* bwd: dqkv [T,3D] from dout [T,D]; delta workspace fp32 [H, T]. */
int chadavit_attn_bwd(const chada_bf16* qkv, const chada_bf16* out, const chada_bf16* dout, const float* lse,
                      chada_bf16* dqkv, float* delta, const int* cu_seqlens, const int* work, int n_work, int T,
                      int D, int H, void* stream);
/* same, selecting pieces: parts bit 1 = delta[h][t] = sum_d dO*O, 2 = dQ kernel, 4 = dK/dV kernel (dQ and dK/dV are independent
 * given delta: the host may issue them on two streams).  chadavit_attn_bwd == parts 7. */
int chadavit_attn_bwd_parts(const chada_bf16* qkv, const chada_bf16* out, const chada_bf16* dout, const float* lse,
                            chada_bf16* dqkv, float* delta, const int* cu_seqlens, const int* work, int n_work, int T,
                            int D, int H, int parts, void* stream);
/* bwd for dh = D/H = 16 -- the reference's DEFAULT constructor (12 heads at D = 192, src/backbones/vit/chada_vit.py:138-139; the
 * notebook's model): heads are widened to 32 lanes with zeros in `workspace` (>= chadavit_attn_bwd_dh16_workspace_bytes, 16-byte
 * aligned, caller-owned) and the dh = 32 kernels run with the model's softmax scale 1/sqrt(16).  Replaces autograd of
 * nn.MultiheadAttention at chada_vit.py:105-111 for that construction. */
long long chadavit_attn_bwd_dh16_workspace_bytes(int T, int H);
int chadavit_attn_bwd_dh16(const chada_bf16* qkv, const chada_bf16* out, const chada_bf16* dout, const float* lse,
                           chada_bf16* dqkv, float* delta, const int* cu_seqlens, const int* work, int n_work, int T,
                           int D, int H, void* workspace, long long workspace_bytes, void* stream);
int chadavit_attn_tile_rows(void); /* rows per work tile (q tile == kv tile) */

/* ---------------------------------------------------------------------------------------------
 * Row gather / scatter helpers around the CLS select (chada_vit.py:283-289) and its backward.
 * --------------------------------------------------------------------------------------------- */
int chadavit_gather_rows(const chada_bf16* src, const int* rows, chada_bf16* dst, int n_rows, int D, void* stream);
int chadavit_scatter_rows_zero(const chada_bf16* src, const int* rows, chada_bf16* dst, int n_rows, int T, int D,
                               void* stream); /* dst[T,D] = 0; dst[rows[i]] = src[i] */

/* Tokenizer backward reductions (autograd of chada_vit.py:245-265):
 *   dpatch_tok bf16 [Mp, D] <- dtok rows of patch tokens (compacted, for the dWp GEMM);
 *   dpos fp32 [p, D] = sum over channels/images; dchan fp32 [maxC, D]; dcls fp32 [D] = sum_i dtok[cu[i]]. */
int chadavit_tokenizer_bwd(const chada_bf16* dtok, const int* cu_seqlens, const int* chan_img, const int* chan_idx,
                           chada_bf16* dpatch_tok, float* dpos, float* dchan, float* dcls, float* workspace, int B,
                           int n_chan, int p, int D, int max_channels, void* stream);
long long chadavit_tokenizer_bwd_workspace_floats(int p, int D, int max_channels); /* workspace size in floats (16-byte aligned) */

/* ---------------------------------------------------------------------------------------------
 * DINO head pieces (dino.py:98-111): row L2 normalise (F.normalize eps 1e-12) and weight-norm of the
 * prototype matrix (nn.utils.weight_norm, dim=0: w = g * v / ||v||_row).
 * --------------------------------------------------------------------------------------------- */
int chadavit_l2norm_fwd(const float* x, chada_bf16* y, float* inv_norm, int M, int N, void* stream);
int chadavit_l2norm_bwd(const float* dy, const float* x, const float* inv_norm, chada_bf16* dx, int M, int N,
                        void* stream);
int chadavit_weightnorm_fwd(const float* v, const float* g, chada_bf16* w, chada_bf16* w_t, float* inv_norm, int P,
                            int K, void* stream);
int chadavit_weightnorm_bwd(const float* dw, const float* v, const float* g, const float* inv_norm, float* dv,
                            int accumulate, int P, int K, void* stream);
/* ... with the gradient of the magnitudes as well: dg[p] (+)= <dw[p], v[p]> / ||v[p]||  (method_kwargs.norm_last_layer = False leaves
 * last_layer.weight_g trainable, dino.py:83-84); dg may be NULL. */
int chadavit_weightnorm_bwd_g(const float* dw, const float* v, const float* g, const float* inv_norm, float* dv, float* dg,
                              int accumulate, int P, int K, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DINO loss forward+backward in one pass (losses/dino.py:69-101) and centre statistics (:103-118).
 * student, teacher: fp32 [2B, P] (two global views stacked).  loss_rows: fp32 [B] per-image loss terms;
 * dstudent (bf16 or NULL) = dL/dstudent for L = mean over rows and the 2 cross pairs.
 * teacher_colsum fp32 [P] = sum over the 2B rows of the raw teacher logits (centre update input).
 * --------------------------------------------------------------------------------------------- */
int chadavit_dino_loss(const float* student, const float* teacher, const float* center, float student_temp,
                       float teacher_temp, float* loss_rows, chada_bf16* dstudent, float* teacher_colsum, int B,
                       int P, void* stream);
/* The same loss over V >= 2 student views, student [V*B, P] view-major (losses/dino.py:69-100 with `student_out.chunk(V)`: teacher view
 * iq < 2 against every student view v != iq, 2V - 2 terms): the STANDARD-DINO multi-crop form.  The reference's DINO.training_step never
 * feeds the local crops to its loss (src/methods/dino.py:300-325, base.py:566-620) -- this serves the separately flagged option
 * `method_kwargs.standard_multicrop_loss`, not the parity path.  dstudent bf16 [V*B, P] (may be NULL). */
int chadavit_dino_loss_multicrop(const float* student, const float* teacher, const float* center, float student_temp, float teacher_temp,
                                 float* loss_rows, chada_bf16* dstudent, float* teacher_colsum, int B, int V, int P, void* stream);
int chadavit_center_ema(float* center, const float* colsum, float inv_count, float momentum, int P, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Flat-buffer parameter kernels.
 *   ema:    teacher = tau*teacher + (1-tau)*student                       (momentum.py:63-74)
 *   adamw:  torch.optim.AdamW single update on a flat slice                (base.py:67-72)
 *   cast:   bf16 copy of fp32 weights; cast_transpose also writes W^T (for the dX GEMMs)
 *   clip:   per-tensor g *= min(1, clip/(||g||+1e-6))                     (dino.py:249-261)
 * --------------------------------------------------------------------------------------------- */
int chadavit_ema_update(float* teacher, const float* student, float tau, long long n, void* stream);
int chadavit_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float bias_corr1, float bias_corr2, long long n,
                        void* stream);
/* The reference's other two optimiser choices (base.py:67-72 `_OPTIMIZERS`): torch.optim.Adam (weight decay added to the gradient) and
 * torch.optim.SGD (momentum / dampening / Nesterov), each one launch over a run of the flat parameter slab. */
int chadavit_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1, float beta2,
                       float eps, float weight_decay, float bias_corr1, float bias_corr2, long long n, void* stream);
int chadavit_sgd_step(float* param, const float* grad, float* momentum_buf, float lr, float momentum, float dampening,
                      float weight_decay, int nesterov, int first, long long n, void* stream);
/* The per-step scalars of the three launches above read from DEVICE memory instead of being passed by value, so that a hipGraph of
 * the whole training step (chadavit_amd.graphed.GraphedTrainStep) can be replayed with a new learning rate / bias correction / tau /
 * temperature each step:  hyper = {lr, 1 - beta1^t, sqrt(1 - beta2^t)};  tau, teacher_temp = one float each.
 * replaces the same reference lines as their by-value twins (base.py:67-72, momentum.py:63-74, losses/dino.py:69-118). */
int chadavit_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const float* hyper, float beta1,
                            float beta2, float eps, float weight_decay, long long n, void* stream);
int chadavit_ema_update_dev(float* teacher, const float* student, const float* tau, long long n, void* stream);
int chadavit_dino_loss_dev(const float* student, const float* teacher, const float* center, float student_temp,
                           const float* teacher_temp, float* loss_rows, chada_bf16* dstudent, float* teacher_colsum, int B, int P,
                           void* stream);
/* Per-channel intensity jitter of the collated crop tensor x [n_channel_images, 1, S, S] fp32, in place:
 * x <- clamp(gamma_c * (x + shift_c), 0, 1) (CustomColorJitter.apply, src/data/custom_transforms.py:301-351), with an
 * optional per-channel-image horizontal flip (flip may be NULL) in the same pass. */
int chadavit_channel_jitter(float* x, const float* shift, const float* gamma, const unsigned char* flip, int n_channel_images,
                            int S, void* stream);

/* Attention-map export (get_last_selfattention, src/backbones/vit/chada_vit.py:313-320; need_weights path of
 * nn.MultiheadAttention, :105-110): probs + prob_offsets[b] receives image b's [H, len_b, len_b] fp32 softmax(QK^T/sqrt(dh)).
 * max_len = longest sequence (<= 2048). */
int chadavit_attn_probs(const chada_bf16* qkv, float* probs, const int* cu_seqlens, const long long* prob_offsets, int B, int T,
                        int D, int H, int max_len, void* stream);

/* Fused feed-forward for D = 192: Out = resid + b2 + relu(X W1^T + b1) W2^T, the hidden activation never leaving the chip
 * unless H != NULL (then relu(.) is also stored, M x FF, for the backward).  Replaces linear1 -> relu -> linear2 (+ residual
 * add) of torch.nn.TransformerEncoderLayer as built at src/backbones/vit/chada_vit.py:256-264.
 * `packed` is the fragment-major weight stream chadavit_ffn_pack builds from W1 [FF,D] and W2 [D,FF] (bf16);
 * chadavit_ffn_packed_bytes gives its size (-1 if the shape is unsupported).  rows_per_wave: kept in the signature; the kernel's own row
 * tiling is used (32 rows per wave at D = 192, 16 at D = 384), 64 -- an instance removed in round 4 -- returns 2. */
long long chadavit_ffn_packed_bytes(int D, int FF);
int chadavit_ffn_pack(const chada_bf16* W1, const chada_bf16* W2, void* packed, int D, int FF, void* stream);
int chadavit_ffn_fwd(const chada_bf16* X, int ldx, const void* packed, const float* b1, const float* b2,
                     const chada_bf16* resid, int ldr,
                     chada_bf16* Out, int ldo, chada_bf16* H, int ldh, void* relu_bits, int M, int D, int FF, int rows_per_wave,
                     void* stream);
/* ReLU pattern of the FFN's hidden activation, recorded by the forward INSTEAD of the [M x FF] hidden tensor (1 bit per element:
 * M * FF / 8 bytes, opaque lane-order records of 1 KiB; relu_bits buffers are 16-byte aligned, caller-owned, >= this many bytes;
 * FF % 256 == 0).  Consumed by chadavit_ffn_bwd_dx below. */
long long chadavit_relu_bits_bytes(int M, int FF);
/* dX pass of the FFN backward (autograd of linear2(relu(linear1(x))) + x w.r.t. x, chada_vit.py:113-115) in ONE launch with nothing
 * FF-wide in HBM:  dX1 = dZ + ((dZ W2) * [H > 0]) W1.  packed_bwd = chadavit_ffn_pack / _pack_batched applied to the TRANSPOSED bf16
 * copies (W2^T [FF x D] in the W1 slot, W1^T [D x FF] in the W2 slot).  dPre (optional, [M x FF] bf16) also receives
 * (dZ W2) * [H > 0] for a stand-alone dW1 GEMM. */
int chadavit_ffn_bwd_dx(const chada_bf16* dZ, int lddz, const void* packed_bwd, const void* relu_bits, chada_bf16* dX1, int lddx,
                        chada_bf16* dPre, int lddp, int M, int D, int FF, void* stream);
/* Weighted k-NN vote of the evaluation path (WeightedKNNClassifier.compute, src/utils/knn.py:141-161): sims [n_test, ld_sims]
 * fp32 similarities to the n_train bank samples (cosine dot products, or 1/(dist+eps)); the k most similar vote for
 * train_targets[j] with weight exp(sim/temperature) (use_exp) or sim; top_classes [n_test, top] = classes by vote mass, best
 * first; votes (optional) [n_test, num_classes]. */
int chadavit_knn_vote(const float* sims, long long ld_sims, const int* train_targets, int n_test, int n_train, int k,
                      float temperature, int use_exp, int num_classes, int top, int* top_classes, float* votes, void* stream);

/* The same fused feed-forward with the block's LayerNorm tail applied before anything leaves the chip:
 *   X2 = LN_a(z), z = resid + b2 + relu(X W1^T + b1) W2^T   (norm2, chada_vit.py:100)
 *   Hn = LN_b(X2) when Hn != NULL                            (the NEXT block's norm1, chada_vit.py:96)
 * Z (optional) receives z for the backward, H (optional) relu(.); statistics (optional pairs) are fp32 per row.  Statistics are
 * taken over the bf16-rounded z / X2, as a separate LayerNorm pass would see them.  relu_bits (optional): see
 * chadavit_relu_bits_bytes.  D = 192 (Tiny) or 384 (Small: 8 waves x 16 rows per block; also chadavit_ffn_fwd / _pack* / _bwd_dx). */
int chadavit_ffn_ln_fwd(const chada_bf16* X, int ldx, const void* packed, const float* b1, const float* b2, const chada_bf16* resid,
                        int ldr, chada_bf16* Z, int ldz, chada_bf16* H, int ldh, const float* gamma_a, const float* beta_a, float eps_a,
                        chada_bf16* X2, float* mean_a, float* rstd_a, const float* gamma_b, const float* beta_b, float eps_b,
                        chada_bf16* Hn, float* mean_b, float* rstd_b, void* relu_bits, int M, int D, int FF, void* stream);

/* LARS (src/utils/lars.py:112-167) on a flat slab: tensor t = [offsets[t], offsets[t]+sizes[t]); flags[t] bit 0 = layer-wise
 * scaling + weight decay apply (p.ndim != 1 or not exclude_bias_n_norm), bit 1 = momentum buffer already initialised. */
int chadavit_lars_step(float* params, const float* grads, float* momentum_bufs, const long long* offsets,
                       const long long* sizes, const int* flags, int n_tensors, float lr, float momentum, float dampening,
                       float weight_decay, float eta, float eps, int clip_lr, int nesterov, void* stream);
int chadavit_cast_bf16(const float* src, chada_bf16* dst, long long n, void* stream);
int chadavit_cast_transpose_bf16(const float* src, chada_bf16* dst, chada_bf16* dst_t, int rows, int cols,
                                 void* stream);
/* W^T shadows of many 2-D weights of one fp32 slab in ONE launch (the per-step refresh of FlatParams): desc[4 t ..] =
 * {src offset (floats), dst_t offset (bf16 elements), rows, cols} on the device; max_tiles = the largest matrix's 32x32 tiles. */
int chadavit_cast_transpose_batched(const float* src, chada_bf16* dst_t, const long long* desc, int n_mats, int max_tiles,
                                    void* stream);
/* One transformer block from the attention output to the next block's normalised input in ONE launch (D = 192):
 *   y = x + a Wo^T + bo ; x1 = norm1(y) ; z = x1 + b2 + relu(x1 W1^T + b1) W2^T ; x2 = norm2(z) ; hn = norm1_next(x2)
 * replaces self_attn.out_proj + residual + norm1 + the feed-forward + norm2 of nn.TransformerEncoderLayer (post-norm,
 * chada_vit.py:96-100, 256-264) and the next layer's norm1.  `packed` = [3 Wo blocks | FFN blocks] written by
 * chadavit_ffn_pack_proj_batched (desc[5 t ..] = {W1, W2, Wo, next in_proj (or -1) offsets into the bf16 slab, packed offset};
 * chadavit_ffn_proj_packed_bytes per layer).  Y, Z, H, the statistics and Hn are optional (NULL) as in chadavit_ffn_ln_fwd;
 * X1 (needed by the backward) is optional too: the FFN takes its input and its residual from registers. */
long long chadavit_ffn_proj_packed_bytes(int D, int FF);
int chadavit_ffn_pack_proj_batched(const chada_bf16* slab, void* packed, const long long* desc, int n_layers, int D, int FF,
                                   void* stream);
/* The same plus, optionally, the NEXT block's QKV projection as a postlogue: QKV[M, 3 D] = hn Wqkv_next^T + bqkv (replaces the
 * in_proj of the next layer's nn.MultiheadAttention, chada_vit.py:105-111); needs gamma_b / beta_b; Hn becomes optional.  The
 * packed stream then carries 9 more blocks behind the FFN ones (chadavit_ffn_pack_proj_batched: desc[5 t ..] = {W1, W2, Wo,
 * next in_proj weight or -1, packed offset}). */
int chadavit_block_fwd(const chada_bf16* A, int lda, const chada_bf16* Xres, int ldxr, const void* packed, const float* bo,
                       const float* gamma1, const float* beta1, float eps1, chada_bf16* Y, int ldy, chada_bf16* X1, int ldx1,
                       float* mean1, float* rstd1, const float* b1, const float* b2, chada_bf16* Z, int ldz, chada_bf16* H, int ldh,
                       const float* gamma_a, const float* beta_a, float eps_a, chada_bf16* X2, float* mean_a, float* rstd_a,
                       const float* gamma_b, const float* beta_b, float eps_b, chada_bf16* Hn, float* mean_b, float* rstd_b,
                       chada_bf16* QKV, int ldqkv, const float* bqkv, void* relu_bits, int M, int D, int FF, void* stream);
int chadavit_proj_ffn_ln_fwd(const chada_bf16* A, int lda, const chada_bf16* Xres, int ldxr, const void* packed, const float* bo,
                             const float* gamma1, const float* beta1, float eps1, chada_bf16* Y, int ldy, chada_bf16* X1, int ldx1,
                             float* mean1, float* rstd1, const float* b1, const float* b2, chada_bf16* Z, int ldz, chada_bf16* H,
                             int ldh, const float* gamma_a, const float* beta_a, float eps_a, chada_bf16* X2, float* mean_a,
                             float* rstd_a, const float* gamma_b, const float* beta_b, float eps_b, chada_bf16* Hn, float* mean_b,
                             float* rstd_b, int M, int D, int FF, void* stream);
/* chadavit_ffn_pack for all layers in one launch: desc[3 t ..] = {W1 offset, W2 offset (into the bf16 slab), packed offset}. */
int chadavit_ffn_pack_batched(const chada_bf16* slab, void* packed, const long long* desc, int n_layers, int D, int FF, void* stream);
int chadavit_clip_tensors(float* grads, const long long* offsets, const long long* sizes, int n_tensors, float clip,
                          void* stream);
int chadavit_sum_rows_f32(const float* x, float* out, int rows, int cols, float scale, void* stream);

/* ---- BatchNorm1d of the DINO head's projector (reference src/methods/dino.py:59-77, `use_bn_in_head`; torch.nn.BatchNorm1d in
 * training mode: statistics over the N rows of each of the C columns, biased variance for the normalisation, unbiased for the
 * running estimate).  z: the Linear's output [N, C], fp32 (z_f32 != 0) or bf16; pre / act / dy / dz: bf16 [N, C]; row-major, C % 4 == 0;
 * workspace: 2 C floats.
 * chadavit_bn_stats: mean[C], rstd[C] = 1 / sqrt(var + eps); running_mean / running_var (both or neither) updated with `momentum`.
 * chadavit_bn_apply_gelu: pre = (z - mean) rstd gamma + beta, act = gelu(pre) (the nn.GELU that follows, dino.py:68).
 * chadavit_bn_bwd: dy = gradient w.r.t. the BatchNorm output; dgamma, dbeta (+= when accumulate), dz. */
int chadavit_bn_stats(const void* z, int z_f32, int N, int C, float eps, float* mean, float* rstd, float* running_mean, float* running_var,
                      float momentum, float* workspace, void* stream);
int chadavit_bn_apply_gelu(const void* z, int z_f32, const float* mean, const float* rstd, const float* gamma, const float* beta,
                           chada_bf16* pre, chada_bf16* act, int N, int C, void* stream);
int chadavit_bn_bwd(const chada_bf16* dy, const void* z, int z_f32, const float* mean, const float* rstd, const float* gamma, float* dgamma,
                    float* dbeta, int accumulate, chada_bf16* dz, int N, int C, float* workspace, void* stream);

/* ---- fp8 weight path (BASELINE.json configs[4]: ChAda-ViT-Base) -------------------------------------------------------------
 * OCP MX fp8: e4m3fn elements + one E8M0 power-of-two scale per 32 consecutive k of a row (the form gfx950's
 * v_mfma_scale_f32_16x16x128_f8f6f4 multiplies at twice the bf16 MFMA rate).  Replaces the nn.Linear forwards of the encoder block
 * (src/backbones/vit/chada_vit.py:95-116) when ChAdaViT.weight_dtype == "fp8".
 * chadavit_mx8_quantize: x bf16 [R, K] (row stride ldx) -> q [R, K] e4m3 bytes (16-byte aligned), scales [K/32, lds] E8M0 bytes
 *   (row stride lds >= R, a multiple of 4; 4-byte aligned); relu != 0 applies max(x, 0) first.  K % 32 == 0. */
int chadavit_mx8_quantize(const chada_bf16* x, int ldx, void* q, void* scales, int lds, int R, int K, int relu, void* stream);
/* Out[M,N] (bf16) = epilogue(deq(Xq, xs) deq(Wq, ws)^T + bias); epilogue 0 = none, 1 = ReLU, 3 = + aux (bf16 [M, N], ld ldaux),
 * 4 = masked by aux > 0 (threshold_backward: the dX GEMM behind the ReLU).
 * xs [K/32, lds_x], ws [K/32, lds_w] as written by chadavit_mx8_quantize.  N % 128 == 0, K % 128 == 0. */
int chadavit_gemm_nt_mx8(const void* Xq, const void* xs, int lds_x, const void* Wq, const void* ws, int lds_w, chada_bf16* Out, int ldo,
                         int M, int N, int K, const float* bias, int epilogue, const chada_bf16* aux, int ldaux, void* stream);
/* The same GEMM with its result ALSO (or, Out == NULL, ONLY) written as the next GEMM's OCP-MX operand: OutQ [M, N] e4m3 and
 * out_scales [N/32, lds_o] e8m0 (lds_o >= M), bit-identical to chadavit_mx8_quantize applied to Out -- linear1's epilogue feeding linear2
 * (chada_vit.py:113-115) without the quantise pass over the hidden activation. */
int chadavit_gemm_nt_mx8_q(const void* Xq, const void* xs, int lds_x, const void* Wq, const void* ws, int lds_w, chada_bf16* Out, int ldo,
                           void* OutQ, void* out_scales, int lds_o, int M, int N, int K, const float* bias, int epilogue,
                           const chada_bf16* aux, int ldaux, void* stream);

/* ---- device side of the multi-crop augmentation contract (build_transform_pipeline, src/data/pretrain_dataloader.py:272-328) -------
 * chadavit_crop_resize: RandomResizedCrop / Resize with cv2.INTER_CUBIC (+ CustomColorJitter, src/data/custom_transforms.py:301-351,
 *   + HorizontalFlip) of n_channel_images source planes into out [n_channel_images, S, S] fp32 = the (sum C, 1, S, S) collate layout of
 *   src/data/channels_strategies.py:31-85.  desc: 8 long long per OUTPUT channel image = {element offset of the source plane in src,
 *   H, W, x0, y0, crop_w, crop_h, flip}; shift / gamma: per channel image (gamma < 0: no jitter for it), or both NULL.  S <= 1024
 *   (return 2 above that); a crop window must span less than 2^32 bytes of its source plane (32-bit offsets from its corner).
 * chadavit_blur_finish: GaussianBlur (BORDER_REFLECT_101) -> Solarize -> Normalize, out != in; fin: 12 floats per channel image =
 *   {ksize (0/1 = no blur, <= 7), 7 centred 1-D taps, solarize threshold (+inf = off), solarize max, mean * max_pixel_value,
 *   1 / (std * max_pixel_value)}.  4 <= S <= 1024. */
int chadavit_crop_resize(const float* src, const long long* desc, const float* shift, const float* gamma, float* out,
                         int n_channel_images, int S, void* stream);
/* The same with source planes of src_kind 0 = float32, 1 = uint8, 2 = uint16: the integer kinds are what the image files store; the
 * reference reader casts them to float32 on the host (src/data/custom_datasets.py:181-190), here they are uploaded as stored (4x / 2x
 * less staging and PCIe traffic) and converted -- exactly -- on the device.  desc offsets are in ELEMENTS of the source type. */
int chadavit_crop_resize_src(const void* src, int src_kind, const long long* desc, const float* shift, const float* gamma, float* out,
                             int n_channel_images, int S, void* stream);
int chadavit_blur_finish(const float* in, const float* fin, float* out, int n_channel_images, int S, void* stream);
/* HOST function (no GPU work): the random parameters of one crop of n samples, drawn in the order albumentations 1.3.1's Compose consumes
 * CPython's `random` stream for the reference's transform list (pretrain_dataloader.py:281-326): crop p + RandomResizedCrop parameters
 * (10 attempts, central fallback), [gray p], [blur p, kernel size, sigma], [solarize p, threshold], [flip p], ToTensorV2 p, [normalize p]
 * -- a bracketed draw happens only when its probability is non-zero (norm_on for the last).  mt_state: the generator as
 * random.Random.getstate()[1] holds it (624 MT19937 words + index), continued in place.  hw: n x (H, W) of the samples' planes.
 * Outputs per sample: boxes (y0, x0, h, w), gray / flip / normed flags, blur_k (-1 = not fired) + blur_sigma, sol_on + sol_value.
 * Bit-identical to chadavit_amd/data/device_pipeline.py::_draw on the same state (tests/test_augment_cpu.py). */
int chadavit_draw_crop_params(unsigned int* mt_state, int n, const long long* hw, int rrc_enabled, double scale_lo, double scale_hi,
                              double ratio_lo, double ratio_hi, double gray_p, double blur_p, int blur_lo, int blur_hi, double sigma_lo,
                              double sigma_hi, double sol_p, double sol_threshold, double flip_p, int norm_on, double norm_p,
                              long long* boxes, int* gray, int* blur_k, double* blur_sigma, int* sol_on, double* sol_value, int* flip,
                              int* normed);

#ifdef __cplusplus
}
#endif
#endif /* CHADAVIT_HIP_H */
