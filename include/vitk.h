/* vitk.h -- C ABI of libvitk.so: the MI355X (gfx950) kernels behind the ViT encoder hot path of
 * dinkofranceschi/ViT.
 *
 * The reference has no FFI / operator interface of its own (it is pure PyTorch); each entry
 * point below cites the reference code (file:line under /root/reference) whose aten-op sequence
 * it replaces.  Conventions:
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless stated otherwise;
 *   - dtype codes: VITK_F32 = 0, VITK_BF16 = 1 (the element type of the "T" tensors); biases,
 *     LayerNorm affine parameters, statistics and parameter gradients are always fp32;
 *   - a row-major matrix is (pointer, leading dimension in ELEMENTS); rows must be 16-byte aligned
 *     where stated; nothing is allocated, nothing synchronises, work is queued on `stream`
 *     (a hipStream_t passed as void*);
 *   - return 0 on success, nonzero on error (1 bad argument, 2 HIP launch error, 3 workspace);
 *     vitk_last_error() returns the calling thread's last message.  No C++ exception crosses.
 */
#ifndef VITK_H
#define VITK_H

#ifdef __cplusplus
extern "C" {
#endif

#define VITK_F32 0
#define VITK_BF16 1

/* GEMM epilogue activations */
#define VITK_ACT_NONE 0
#define VITK_ACT_GELU 1   /* erf GELU (nn.GELU, timm Mlp); aux_out receives the pre-activation */
#define VITK_ACT_RELU 2
#define VITK_ACT_DGELU 3  /* out = (A.B) * gelu'(aux_in),  aux_in = saved pre-activation */
#define VITK_ACT_DRELU 4  /* out = (A.B) * (aux_in > 0),   aux_in = saved activation output */

/* ABI version: bumped whenever an entry point's argument list changes; vitk_version() of the loaded library must
 * equal the VITK_ABI_VERSION the caller was compiled / written against (vit_amd/_lib.py checks exactly that). */
#define VITK_ABI_VERSION 3
int vitk_version(void);
const char* vitk_last_error(void);

/* CUs that the persistent GEMM launches (one workgroup per CU holding its whole LDS) leave free, 0..128, process-wide.
 * For the data-parallel step: DDP's NCCL kernels share the GPU with backward (train.py:635, utils/ops.py:124-133);
 * RCCL's reduction kernels need CUs that a 160 KiB-LDS workgroup would otherwise hold until it exits.  Default 0, or
 * the environment's VITK_RESERVED_CUS.  The grid stays a multiple of 8 (XCD mapping). */
int vitk_set_reserved_cus(int n);
int vitk_get_reserved_cus(void);

/* Zero fills under HIP graph capture.  Entry points that accumulate with atomics (vitk_gemm with split-K or
 * accumulate, vitk_attn_bwd's table gradients, vitk_colsum, the FAVOR+ state) need their targets zeroed first; where the
 * library does that itself it launches a kernel of its own (vitk_zero_async in csrc/runtime.hip), never hipMemsetAsync.
 * Reason: with hipMemsetAsync nodes in a training step captured by torch.cuda.graph (ROCm 7.2, PyTorch 2.10), the
 * second replay accumulated onto stale data (logits 5e36); the same step with kernel zero fills replays bit-stably.
 * The cause was not isolated: tools/micro/memset_graph.hip captures hundreds of (memset, accumulate, check) triples
 * with plain HIP -- hipMemsetAsync on large, tiny and pitched targets, hipMemsetD32Async -- and every replay is correct, so
 * memset nodes as such are not broken; the failure needed the torch capture (its private memory pool and capture mode)
 * around ctypes calls.  A caller that captures these entry points and zeroes the accumulation targets itself should do it
 * with a kernel (torch's zero_() is one) rather than a memset node, or verify two replays against eager as
 * tests/test_gpu_graph.py does. */
/* C = epilogue(alpha * op(A) op(B) + bias) [+ residual]      (MFMA, fp32 accumulate)
 * Replaces nn.Linear forward/backward everywhere on the path: qkv / proj / Mlp fc1,fc2
 * (timm_classes/timm_locality_aware_transformer.py:48,77,94), query/key/value_proj,
 * dense_AttnOut, dense_intermediate (+GELU), dense_out (disentangled_transformer.py:854-856,
 * 188,184-185,194), linear1/linear2 (model.py:202), head (model.py:65), and the patch-embed
 * Conv2d(k = s = P) as a GEMM over gathered patches (model.py:92-99).
 * op(A) is M x K: transA = 0 -> A stored [M][lda], 1 -> stored [K][lda] (dW = dY^T X).
 * op(B) is K x N: transB = 1 -> B stored [N][ldb] (nn.Linear weight), 0 -> stored [K][ldb] (dX = dY W).
 * c_dtype: output type (bf16 inputs may write fp32: parameter gradients).
 * bias [N] fp32 or NULL.  aux_in / aux_out / residual: T matrices (ld_aux, ldr) or NULL.
 * res_period > 0: residual row = res_off + (row % res_period) (position embedding broadcast).
 * grp_rows > 0: output row r goes to (r / grp_rows) * grp_stride + grp_off + r % grp_rows
 *               (patch rows into the [B][N][D] token matrix, skipping each image's class row).
 * accumulate: C += (fp32 C only).  splitk: 0 = auto, > 1 forces split-K (fp32 C, linear epilogue). */
int vitk_gemm(int dtype, int c_dtype, int transA, int transB, int M, int N, int K, float alpha,
              const void* A, long lda, const void* B, long ldb, void* C, long ldc,
              const float* bias, int act, const void* aux_in, void* aux_out, long ld_aux,
              const void* residual, long ldr, int res_period, int res_off,
              int grp_rows, int grp_stride, int grp_off, int accumulate, int splitk, void* stream);

/* nn.LayerNorm over rows of length D (D % 8 == 0, D <= 2048), one wave per row.
 * timm_locality_aware_transformer.py:87,92,123 (eps 1e-6); disentangled_transformer.py:99,173,180
 * (eps 1e-5); model.py:22,44.  mean / rstd [rows] fp32 are saved for backward (may be NULL). */
int vitk_layernorm_fwd(int dtype, int rows, int D, const void* x, long ldx, const float* gamma,
                       const float* beta, float eps, void* y, long ldy, float* mean, float* rstd,
                       void* stream);
/* dx = LN'(dy) [+ add] (add = the residual branch's gradient); dgamma / dbeta [D] fp32 are
 * ACCUMULATED into (zero them for a fresh sum).  dxsum (may be NULL): [D] fp32, += column sums of
 * dx = the bias gradient of the Linear feeding a post-LN residual (dense_AttnOut / dense_out,
 * disentangled_transformer.py:187-196), accumulated like dgamma. */
int vitk_layernorm_bwd(int dtype, int rows, int D, const void* dy, long lddy, const void* x, long ldx,
                       const float* mean, const float* rstd, const float* gamma, const void* add,
                       long ldadd, void* dx, long lddx, float* dgamma, float* dbeta, float* dxsum,
                       void* stream);

/* out[n] += sum_m X[m][n]: bias gradients of every nn.Linear above (autograd of addmm). */
int vitk_colsum(int dtype, int M, int N, const void* X, long ldx, float* out, void* stream);

/* direction 0: fp32 -> bf16, 1: bf16 -> fp32, n elements (weight shadows for the bf16 path). */
int vitk_cast(int direction, long n, const void* src, void* dst, void* stream);

/* One torch.optim.AdamW step (the optimizer train.py:275 builds and :549 steps; torch defaults
 * betas 0.9/0.999, eps 1e-8, weight_decay 1e-2) over one fp32 parameter viewed as [rows, cols]
 * (cols % 4 == 0 unless no copy is written), `step` counting from 1:
 *   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *   p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)
 * and, in the same pass, the bf16 copies of the new p that the GEMMs read: shadow [rows][ld_shadow]
 * (may be NULL) and shadow_t [cols][ld_shadow_t] = the transpose (may be NULL).  SURVEY 8(f)-2. */
int vitk_adamw(int rows, int cols, float* p, const float* g, float* m, float* v, double lr, double beta1,
               double beta2, double eps, double weight_decay, int step, void* shadow, long ld_shadow,
               void* shadow_t, long ld_shadow_t, void* stream);
/* The same step with the step-dependent scalars read from device memory when the kernel runs:
 * dyn[3] fp32 = {1 - lr*wd, lr/(1-b1^step), 1/sqrt(1-b2^step)}.  No counterpart in the reference: it
 * lets the whole train step of train.py:530-554 be captured once in a HIP graph and replayed (the
 * host rewrites dyn before each replay) -- for the launch-bound small configurations. */
int vitk_adamw_dyn(int rows, int cols, float* p, const float* g, float* m, float* v, double beta1,
                   double beta2, double eps, const float* dyn, void* shadow, long ld_shadow,
                   void* shadow_t, long ld_shadow_t, void* stream);
/* optimizer.step() of train.py:549 for a whole parameter group in ONE launch (the multi-tensor pass SURVEY
 * 8(f)-2 names).  desc: DEVICE memory, n + 1 records of vitk_adamw_desc_bytes() = 72 bytes, each 9 int64 words
 * {p, g, m, v, shadow, shadow_t, ld_shadow | ld_shadow_t << 32, rows | cols << 32, tile0 | tiles_x << 32}
 * (64 x 64 tiles: tiles_x = ceil(cols / 64), tile0 = running sum of tiles_x * ceil(rows / 64); record n holds
 * tile0 = total_tiles only).  dyn as in vitk_adamw_dyn.  Per-element arithmetic identical to vitk_adamw. */
int vitk_adamw_multi(int n, const void* desc, long total_tiles, double beta1, double beta2, double eps,
                     const float* dyn, void* stream);
int vitk_adamw_desc_bytes(void);
/* dst[0..n) = {v0..v3}[0..n), n <= 4, written by a kernel from its ARGUMENTS: how the host hands dyn to a replayed
 * graph without keeping a host buffer alive (no counterpart in the reference; plumbing of vitk_adamw_dyn/_multi). */
int vitk_set_f32(float* dst, int n, float v0, float v1, float v2, float v3, void* stream);
/* bytes (a multiple of 8, <= 1 MiB) from host memory to device memory THROUGH KERNEL ARGUMENTS, 2 KiB per launch:
 * stream-ordered, legal inside a stream capture, and the host buffer is free again when the call returns.  No
 * counterpart in the reference: how vitk_adamw_multi's descriptor table reaches the device. */
int vitk_upload_args(void* dst, const void* src_host, size_t bytes, void* stream);

/* Patch gather / scatter for Conv2d(kernel = stride = P): out[(b*nph+ph)*npw+pw][(c*P+py)*P+px] =
 * img[b][c][ph*P+py][pw*P+px]  (model.py:92-99: proj(x).flatten(2).transpose(1,2)). */
int vitk_patchify(int img_dtype, int out_dtype, int B, int C, int H, int W, int P, const void* img,
                  void* out, void* stream);
int vitk_unpatchify(int dtype, int B, int C, int H, int W, int P, const void* dpatch, float* dimg,
                    void* stream);
/* The same gather from raw uint8 pixels with the loader's ToTensor + Normalize(mean, std) folded in
 * (train.py:340-341, 362-363, 385: x / 255 then (x - mean[c]) / std[c]).  img: [B][C][H][W], or
 * [B][H][W][C] with channels_last != 0; mean / std: C (<= 4) floats in HOST memory.  fp32 output equals
 * torchvision's arithmetic bit for bit. */
int vitk_patchify_u8(int out_dtype, int B, int C, int H, int W, int P, int channels_last, const unsigned char* img,
                     const float* mean, const float* std, void* out, void* stream);
/* x[b][0][:] = cls + pos[0] (pos may be NULL): torch.cat((cls_tokens, x)) [+ pos_embed]
 * (model.py:54-55; timm_locality_aware_transformer.py:136-138). */
int vitk_cls_rows(int dtype, int B, int N, int D, const float* cls, const float* pos, void* x, void* stream);
/* dpos[n][d] += sum_b dx[b][n][d];  dcls[d] += sum_b dx[b][0][d]  (either may be NULL). */
int vitk_token_grads(int dtype, int B, int N, int D, const void* dx, float* dpos, float* dcls, void* stream);

/* Fused attention forward: out = softmax(scale * (Q K^T + c2p + p2c) restricted to |i-j| <= band_w) V.
 * Replaces AttentionLAI.forward's q@k^T / masked_fill / softmax / @v
 * (timm_locality_aware_transformer.py:51-76 with get_attn_mask :11-21) and
 * Disentangled[LAI]SelfAttention.forward + disentangled_attention_bias
 * (disentangled_transformer.py:324-347,357-426 / :875-919,929-998).
 * q, k, v: element (b, n, h, d) at ((b*N + n) * ld + h*dh + d) (views into one packed qkv buffer).
 * pk / pq: the projected relative tables restricted to the 2N-1 reachable rows,
 *          row r <-> i - j = r - (N-1), element (r, h, d) at r * ld_p + h*dh + d; both NULL for
 *          plain / LAI attention.  score = Q_i.K_j + Q_i.pk[i-j] + K_j.pq[i-j].
 * band_w < 0: no mask.  out [B*N][ldo]; lse [B][H][N] fp32 (log-sum-exp, saved for backward).
 * drop_p > 0: dropout on the probabilities after the softmax (attn_drop, timm_locality_aware_transformer.py:74;
 *          self.dropout, disentangled_transformer.py:915): probability (b, h, i, j) is element
 *          ((b*H + h)*N + i)*N + j of the counter-based keep stream (drop_seed, drop_offset) -- see vitk_dropout;
 *          the backward regenerates the mask from the same three numbers. */
int vitk_attn_fwd(int dtype, int B, int H, int N, int dh, const void* q, const void* k, const void* v,
                  long ld, const void* pk, const void* pq, long ld_p, float scale, int band_w,
                  void* out, long ldo, float* lse, float drop_p, unsigned long long drop_seed,
                  unsigned long long drop_offset, void* stream);

/* Fused attention backward (autograd of the above): dq / dk / dv (T, same addressing with ld_d)
 * and, with tables, dpk / dpq fp32 [2N-1][ld_dp] ACCUMULATED into.  `workspace` must hold
 * vitk_attn_bwd_workspace_bytes(...) bytes (16-byte aligned). */
long vitk_attn_bwd_workspace_bytes(int dtype, int B, int H, int N, int dh);
int vitk_attn_bwd(int dtype, int B, int H, int N, int dh, const void* q, const void* k, const void* v,
                  long ld, const void* pk, const void* pq, long ld_p, float scale, int band_w,
                  const void* out, const void* dout, long ldo, const float* lse,
                  void* dq, void* dk, void* dv, long ld_d, float* dpk, float* dpq, long ld_dp,
                  void* workspace, float drop_p, unsigned long long drop_seed, unsigned long long drop_offset,
                  float* dbias, void* stream);
/* dbias (may be NULL): fp32 [3 * H * dh] (q | k | v), += the column sums of dq / dk / dv over all B*N tokens: the bias
 * gradient of the fused q/k/v projection (query_proj / key_proj / value_proj.bias, disentangled_transformer.py:854-856;
 * qkv.bias, timm_locality_aware_transformer.py:48) straight from the kernel that holds dQ / dK / dV in registers. */

/* FAVOR+ (Performer) attention: feature maps + linear-attention contraction + normaliser, between
 * the qkv projection and the output projection.  Replaces PerformerAttention.forward :36-59 with
 * softmax_kernel_transformation :69-91 / relu_kernel_transformation :93-101
 * (timm_classes/timm_vision_performer.py) and MultiheadAttentionPerformer.forward :283-313 with
 * :320-352 (model.py), quirks included: diag = plain sum of the features / 2 * dh^-1/2, the key
 * stabiliser is the max over the WHOLE [B,L,H,m] tensor, eps inside the ratio.
 * kind 0 = softmax kernel (eps 1e-6), 1 = relu kernel (eps 1e-3); ratio = m^-1/4 (timm variant,
 * relu of model.py) or m^-1/2 (softmax of model.py).  w [m][dh] in T, contiguous, m <= 256, dh <= 64.
 * q / k / v / out addressing as vitk_attn_fwd.  `workspace` (vitk_favor_workspace_bytes) receives
 * KV, ksum and the key max / argmax in the forward and is read by the backward of the same step
 * (which also routes the gradient through both maxima). */
long vitk_favor_workspace_bytes(int dtype, int B, int H, int dh, int m);
int vitk_favor_fwd(int dtype, int kind, int B, int H, int L, int dh, int m, const void* q, const void* k,
                   const void* v, long ld, const void* w, float ratio, float eps, void* out, long ldo,
                   void* workspace, void* stream);
int vitk_favor_bwd(int dtype, int kind, int B, int H, int L, int dh, int m, const void* q, const void* k,
                   const void* v, long ld, const void* w, float ratio, float eps, const void* out,
                   const void* dout, long ldo, void* dq, void* dk, void* dv, long ld_d, void* workspace,
                   void* stream);

/* FAVOR+ with position features: DisentangledSelfAttentionPerformer.forward
 * (disentangled_transformer/disentangled_transformer.py:496-610) with
 * disentangled_attention_perf_bias :612-644 and the class's own feature maps :717-749 (the same
 * quirks as above: unscaled projection, diag from the plain sum, global key max, eps inside the ratio):
 *     new_Q = q' + Pq,  new_K = k' + Pk,  out = new_Q (new_K^T v) / (new_Q . sum_l new_K)
 * Pq / Pk = query- / key-form feature maps of pos_query_proj / pos_key_proj(pos_embed), one [L][H][m]
 * table each, the same for every batch element.
 * vitk_favor_feature_pitch(m): row pitch mp (floats) of every table below.
 * vitk_favor_pos_fwd: x [L, ld] (T) -> feat fp32 [L][H][mp]; is_query picks the row max (:732) or the
 *   max over the whole table (:736) as stabiliser; `state` (vitk_favor_pos_state_bytes(H, L) bytes, 16-byte
 *   aligned) keeps the key max / argmax for the backward, which also parks partial sums there.
 * vitk_favor_pos_attn_fwd / _bwd: vitk_favor_fwd / _bwd with the tables added; the backward also fills
 *   dnq / dnk (fp32 [B][L][H][mp]) with d loss / d new_Q and d new_K per batch element.
 * vitk_favor_pos_bwd: sums dfeat over its first dim (nsum = B) and goes back through the feature map,
 *   both maxima included: dx [L, ld_d] (T). */
int vitk_favor_feature_pitch(int m);
long vitk_favor_pos_state_bytes(int H, int L);
int vitk_favor_pos_fwd(int dtype, int kind, int is_query, int H, int L, int dh, int m, const void* x, long ld,
                       const void* w, float ratio, float eps, float* feat, void* state, void* stream);
int vitk_favor_pos_bwd(int dtype, int kind, int is_query, int nsum, int H, int L, int dh, int m, const void* x,
                       long ld, const void* w, float ratio, float eps, const float* dfeat, void* dx, long ld_d,
                       void* state, void* stream);
int vitk_favor_pos_attn_fwd(int dtype, int kind, int B, int H, int L, int dh, int m, const void* q, const void* k,
                            const void* v, long ld, const void* w, float ratio, float eps, const float* pq,
                            const float* pk, void* out, long ldo, void* workspace, void* stream);
int vitk_favor_pos_attn_bwd(int dtype, int kind, int B, int H, int L, int dh, int m, const void* q, const void* k,
                            const void* v, long ld, const void* w, float ratio, float eps, const float* pq,
                            const float* pk, const void* out, const void* dout, long ldo, void* dq, void* dk,
                            void* dv, long ld_d, float* dnq, float* dnk, void* workspace, void* stream);

/* nn.Dropout in training mode (disentangled_transformer.py:189,195,883; timm_locality_aware_transformer.py:78 and timm's
 * Mlp / pos_drop; timm_vision_performer.py:61; model.py:201-203): y = x * keep / (1 - p) [+ res] on [rows, cols] rows.
 * keep(r, c) = element r * cols + c of the counter-based stream (seed, offset): a 32-bit hash of (seed, offset, index)
 * compared with p * 2^32 -- stateless, so the backward is the same call on dy (res NULL) and nothing is stored.  The
 * reference draws its masks from torch's Philox stream, which a fused kernel cannot reproduce bit for bit: parity at
 * p > 0 is defined with this mask handed to the oracle (vitk_dropout_mask, a test hook: out[i] = keep(i)). */
int vitk_dropout(int dtype, int rows, int cols, const void* x, long ldx, const void* res, long ldr, void* y, long ldy,
                 float p, unsigned long long seed, unsigned long long offset, void* stream);
int vitk_dropout_mask(long n, float p, unsigned long long seed, unsigned long long offset, unsigned char* out,
                      void* stream);

/* DropPath / stochastic depth in training mode (timm_locality_aware_transformer.py:94: self.drop_path = DropPath(p),
 * applied to both residual branches :97-98; timm 0.4.5 layers/drop.py: x / (1 - p) * floor(1 - p + rand(B, 1, 1))):
 * y = x * keep[r / rows_per_sample] / (1 - p) [+ res] -- one decision per SAMPLE, sample s = element s of the stream
 * (seed, offset), the same counter-based stream as vitk_dropout (vitk_dropout_mask over B elements returns the
 * decisions).  The backward is the same call on dy (res NULL).  rows must be a whole number of samples. */
int vitk_drop_path(int dtype, int rows, int cols, int rows_per_sample, const void* x, long ldx, const void* res, long ldr,
                   void* y, long ldy, float p, unsigned long long seed, unsigned long long offset, void* stream);

/* Label-smoothing cross entropy, mean over the batch: criterion = LabelSmoothingCrossEntropy(smoothing)
 * (train.py:273, applied :541; timm 0.4.5 loss/cross_entropy.py): (1-s) * NLL + s * mean_c(-log p).
 * logits [B, C] fp32 (row pitch ld / ld_d), target [B] int64.  fwd: loss[0] (scalar), loss_rows [B], lse [B];
 * bwd: dlogits = gout * d loss / d logits with gout one fp32 value in device memory.  A target outside
 * [0, C) makes the loss NaN (torch raises there). */
int vitk_ce_smooth_fwd(int B, int C, const float* logits, long ld, const long long* target, float smoothing,
                       float* loss_rows, float* lse, float* loss, void* stream);
int vitk_ce_smooth_bwd(int B, int C, const float* logits, long ld, const long long* target, float smoothing,
                       const float* lse, const float* gout, float* dlogits, long ld_d, void* stream);
/* vitk_ce_smooth_fwd for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  It has no counterpart in the reference.  Same kernels and results; the per-row partial
 * losses live in a static device array of the library (256 KiB per GPU, held until the process ends) instead of a
 * caller buffer, so the caller passes inputs and outputs only: lse [B], loss [1].  B <= vitk_ce_static_max_rows()
 * (65536), larger B is refused with an error.  Calls that overlap in time on ONE GPU share the array and race;
 * consecutive calls on a stream do not (each writes the rows it reads). */
int vitk_ce_smooth_fwd_static(int B, int C, const float* logits, long ld, const long long* target, float smoothing,
                              float* lse, float* loss, void* stream);
int vitk_ce_static_max_rows(void);

#ifdef __cplusplus
}
#endif
#endif /* VITK_H */
