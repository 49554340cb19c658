/* libdino_hip.so -- C ABI of the MI355X-native DINO training-step kernels.
 *
 * The reference (clemsgrs/dino) is pure Python on stock torch operators: it has no FFI of its
 * own.  The drop-in boundary is therefore its Python nn.Module API (kept by the `dino_amd`
 * package) with this C ABI immediately beneath it.  Each entry point names the reference call
 * site (path:line relative to the reference repository) whose stock operator(s) it replaces.
 *
 * Conventions: every pointer is a DEVICE pointer unless marked "host"; PyTorch owns all memory
 * (the library never allocates); `stream` is a hipStream_t; return value 0 = OK, > 0 = a
 * hipError_t, < 0 = DINO_ERR_*.  Entry points are re-entrant and use no thread-local state.
 * dtype: DINO_F32 (everything fp32) or DINO_BF16 (activations / weight shadows bf16, fp32
 * accumulation, fp32 master parameters, gradients and statistics).
 */
#ifndef DINO_HIP_H
#define DINO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DINO_OK 0
#define DINO_ERR_BADARG (-1)
#define DINO_ERR_UNSUPPORTED (-2)
#define DINO_ERR_TOO_LARGE (-3)   /* a size beyond the fixed capacity of an allocation-free entry point */

#define DINO_F32 0
#define DINO_BF16 1

/* library / ABI version (bumped on any signature change) */
int dino_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Dense contraction with fused epilogue.  Replaces every nn.Linear / nn.Conv2d-as-GEMM on the
 * path and their autograd backward:
 *   PatchEmbed.proj  dino/models/vision_transformer.py:238-245   (on the im2col matrix)
 *   Attention.qkv / .proj  :117-119,124-136     Mlp.fc1 / .fc2  :88-97
 *   DINOHead.mlp.{0,2,4} / last_layer  :691-705,716-719
 * trans == 0 ("NT"):  acc[m,n] = sum_k A[m*lda + k] * B[n*ldb + k]     (A: MxK, B: NxK)
 * trans == 1 ("TN"):  acc[m,n] = sum_k A[k*lda + m] * B[k*ldb + n]     (A: KxM, B: KxN)
 * epilogue, in this order (rs = rowscale ? rowscale[m / rows_per_group] : 1):
 *   v = acc * (prescale ? prescale[m] : 1) (+ bias[n] * (bias_rowscale ? rs : 1))
 *   act == 1: { if (aux) aux[m,n] = v;  v = gelu(v) }          (exact erf, nn.GELU default)
 *   act == 2: v = v * gelu'(aux[m,n])                           (aux = saved pre-activation)
 *   act == 3: { if (aux) aux[m,n] = gelu'(v);  v = gelu(v) }    (saves the derivative instead:
 *   act == 4: v = v * aux[m,n]                                   the backward epilogue is one multiply)
 *   if (out_rowscale) v *= rs
 *   if (residual)     v += residual[rrow*ldres + n],  rrow = res_mod ? m % res_mod
 *                                                          : row_map_n ? 1 + m % row_map_n : m
 *   orow = row_map_n ? m + m / row_map_n + 1 : m               (token rows skipping each CLS)
 *   out_mode 0: out[orow,n] = (T)v   1: out_f32[orow,n] = v   2: atomicAdd(out_f32[orow,n], v)
 *            3: out_f32[(s*M + orow), n] = v for k-split s: `out` holds splitk slabs of M x ldo floats,
 *               every slab written (or zeroed) by the call; dino_sum_slabs adds them up (bf16 NT only)
 *   if (colsum) colsum[n] += sum_m v   (atomicAdd; the bias gradient of the producing layer)
 * splitk > 1 splits the contraction over gridDim.y and requires out_mode 2 or 3 with no
 * bias/act/residual (the prototype layer's input gradient, K = out_dim = 65 536 deep: 22 atomic
 * splits cost 1.1 ms, 16 slabs + dino_sum_slabs 0.2 ms).  Requirements: K, N, lda, ldb, ldo (ldaux, ldres) multiples of 8
 * elements, 16-byte aligned base pointers.
 * ------------------------------------------------------------------------------------------ */
typedef struct dino_gemm_args {
    int32_t dtype, trans;
    int32_t M, N, K;
    int32_t splitk;
    const void* A; int64_t lda;
    const void* B; int64_t ldb;
    void* out; int64_t ldo;
    int32_t out_mode;
    int32_t act;
    const float* bias;
    void* aux; int64_t ldaux;
    const void* residual; int64_t ldres;
    const float* rowscale; int32_t rows_per_group;
    int32_t bias_rowscale, out_rowscale;
    int32_t row_map_n;
    int32_t res_mod;
    float* colsum;   /* optional.  NT: fp32 [N] += column sums of the values written to `out`.
                      * TN (trans): fp32 [M] += sum_k A[k, m] -- the bias gradient beside dW = A^T B */
    void* workspace; /* optional device scratch (split-K partial tiles of the TN kernel) */
    int64_t workspace_bytes;
    /* LayerNorm folded into the GEMMs around it (bf16 wide-tile NT kernel only: ask
     * dino_gemm_wide_tile_supported first, other kernels answer DINO_ERR_UNSUPPORTED).  Block.forward's
     * x + proj(...) -> norm2 -> fc1 (vision_transformer.py:227-228) without a LayerNorm pass over the tokens:
     *   ln_stats (producer, out_mode 0, no act, no colsum): fp32 [ceil(N/96)][M][2] = per output row and
     *     96-column slice, (sum, sum of squares) of the values written to `out` (before rounding);
     *     dino_ln_stats_finalize turns them into mean / rstd;
     *   prescale (consumer): fp32 [M], multiplies the accumulator before the bias: with B = the folded weight
     *     gamma * W - rowmean(gamma * W) (dino_ln_fold_weights) and bias = b + W beta this GEMM on the RAW
     *     residual stream equals Linear(LayerNorm(x)). */
    const float* prescale;
    float* ln_stats;
    /* consumer side without a finalize launch: instead of `prescale`, the partials themselves (ln_stats_in, as a
     * producer's ln_stats wrote them for the rows of A: [ceil(ln_dim/96)][M][2]); the epilogue forms
     * mean = sum / ln_dim, rstd = 1 / sqrt(sumsq / ln_dim - mean^2 + ln_eps) per row, multiplies the accumulator
     * by rstd and (column tile 0 only) leaves mean / rstd in ln_mean_out / ln_rstd_out (fp32 [M], nullable) for
     * the LayerNorm backward. */
    const float* ln_stats_in;
    float* ln_mean_out;
    float* ln_rstd_out;
    int32_t ln_dim;
    float ln_eps;
} dino_gemm_args;
/* Work queue of the one-workgroup-per-CU GEMMs (no counterpart in the reference; a knob for multi-GPU runs, where
 * DistributedDataParallel's bucket all-reduces -- dino.py:190-193 -- run under backward).  The wide-tile NT kernel and
 * the grouped weight-gradient kernel keep a whole CU per workgroup; with static tile shares, a workgroup whose CU is
 * held by another stream's kernel (RCCL) when the GEMM starts delays the launch by a whole share (1.5 x / 1.4 x with
 * 8 of 256 CUs held: tools/bench_nt_occupied.py, tools/bench_occupied_others.py).  With the queue on, an NT workgroup
 * owns only its first tile and draws the rest from per-XCD counters, so the late ones find the queue empty (1.03 x),
 * and the weight-gradient launch runs two rounds of half-length workgroups instead of one (0.96 x).  It is not free:
 * alone on the chip the step is 0.7 ms (1.4 %) slower with it, and on the one-GPU stand-in for overlapped
 * all-reduces (bench.py --occupy) it only pays from ~8 ms of held CUs per 48 ms step on -- so it is OFF unless asked
 * for.  on: 1 / 0, or -1 to query; returns the previous state.  Initial state: env DINO_NT_QUEUE (default 0).  NT
 * results are bit-identical either way (every tile is computed once, by whichever workgroup); the weight gradients
 * differ by fp32 summation order.  The counters live in a device array owned by the library (64 KiB per GPU), one
 * slot per launch taken round robin; nothing is allocated.  (A launch captured into a HIP graph keeps its slot:
 * replays of one graph must not overlap each other.) */
int dino_gemm_work_queue(int on);
/* 1 if dino_gemm would run these arguments on the wide-tile bf16 NT kernel (the only one that implements
 * prescale / ln_stats), else 0.  Host-side, launches nothing. */
int dino_gemm_wide_tile_supported(const dino_gemm_args* args /* host */);
/* mean[m] = sum / D, rstd[m] = 1 / sqrt(sumsq / D - mean^2 + eps) from dino_gemm's ln_stats partials
 * (nparts = ceil(D / 96) slices).  nn.LayerNorm's statistics (vision_transformer.py:191,211). */
int dino_ln_stats_finalize(const float* stats, int nparts, int64_t rows, int D, float eps, float* mean, float* rstd,
                           void* stream);
/* The same statistics straight from the rows (the first block's norm1, whose input comes from the token
 * assembly, not from a GEMM epilogue): two-pass, as dino_layernorm_fwd computes them, no output row. */
int dino_layernorm_stats(int dtype, const void* x, int64_t x_stride, float* mean, float* rstd, int64_t rows, int D,
                         float eps, void* stream);
/* Folded weights for the consumer side, `count` linear layers per launch.  table (device, int64 [count][8]) =
 * (W offset, gamma offset, beta offset, bias offset -- elements into `master` (fp32) --, folded-W offset into
 * `wfold` (bf16), folded-bias offset into `bfold` (fp32), N, K):
 *   wfold[n,k] = bf16(gamma[k] * W[n,k] - mean_k(gamma * W[n,:])),   bfold[n] = bias[n] + sum_k beta[k] * W[n,k]. */
int dino_ln_fold_weights(const float* master, const int64_t* table, int count, void* wfold, float* bfold, int maxN,
                         void* stream);
int dino_gemm(const dino_gemm_args* args /* host */, void* stream);
/* out[i] (+)= sum_s slabs[s*n + i], s < nslab (fp32; n % 4 == 0, 16-byte aligned): the reduction of
 * dino_gemm's out_mode 3 (autograd of DINOHead.last_layer, vision_transformer.py:719). */
int dino_sum_slabs(const float* slabs, int nslab, int64_t n, float* out, int accumulate, void* stream);
/* `count` independent GEMMs (host array), same results as `count` dino_gemm calls.  The four weight
 * gradients of one transformer block (autograd of the nn.Linear layers at
 * vision_transformer.py:93-99,122-138: dW = dY^T X over all tokens) share the token dimension and
 * run as one split-over-tokens launch. */
int dino_gemm_group(const dino_gemm_args* args /* host, count entries */, int count, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm(eps) forward/backward over the last dim D (rows may be strided: the final norm
 * only touches the CLS rows).  Replaces nn.LayerNorm at vision_transformer.py:191,211,311
 * (applied :222-228,382).  mean/rstd: fp32 [rows].  Backward: dx = (dres ? dres : 0) +
 * LN'(dy); dgamma/dbeta are ACCUMULATED (atomicAdd) into fp32 [D]; dx_colsum (nullable, fp32
 * [D]) += sum_rows rs(row) * dx[row] with rs = rowscale ? rowscale[row / rows_per_group] : 1 --
 * the bias gradient of the linear layer whose output was added to this residual stream.
 * D % 8 == 0, D <= 2048.
 * ------------------------------------------------------------------------------------------ */
int dino_layernorm_fwd(int dtype, const void* x, int64_t x_stride, void* y, int64_t y_stride,
                       const float* gamma, const float* beta, float* mean, float* rstd,
                       int64_t rows, int D, float eps, void* stream);
int dino_layernorm_bwd(int dtype, const void* dy, int64_t dy_stride, const void* x,
                       int64_t x_stride, const float* gamma, const float* mean,
                       const float* rstd, const void* dres, int64_t dres_stride, void* dx,
                       int64_t dx_stride, float* dgamma, float* dbeta, float* dx_colsum,
                       const float* rowscale, int rows_per_group, int64_t rows, int D,
                       void* workspace, int64_t workspace_bytes, void* stream);
/* dino_layernorm_bwd that also writes the normalised rows xn = (x - mean) * rstd * gamma + beta (T [rows, D],
 * stride xn_stride): with LayerNorm folded into the GEMMs the forward pass never materialises them, and the
 * weight gradient of the following linear layer (dW = dY^T xn) reads them from here. */
int dino_layernorm_bwd_xn(int dtype, const void* dy, int64_t dy_stride, const void* x,
                          int64_t x_stride, const float* gamma, const float* beta, const float* mean,
                          const float* rstd, const void* dres, int64_t dres_stride, void* dx,
                          int64_t dx_stride, void* xn, int64_t xn_stride, float* dgamma, float* dbeta,
                          float* dx_colsum, const float* rowscale, int rows_per_group, int64_t rows, int D,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* workspace (nullable, >= 1024*3*D*4 bytes): the column sums (dgamma, dbeta, dx_colsum) leave the
 * main kernel as per-workgroup partials and a second small kernel adds them up; without it every
 * workgroup adds into the same 3*D addresses with float atomics. */

/* dino_layernorm_bwd for callers that cannot allocate or bring a workspace (code under HIP graph
 * capture, tight serving / fine-tuning loops, harnesses that forbid hipMalloc).  It has NO counterpart
 * in the reference (which has no native code at all); same arguments and results as
 * dino_layernorm_bwd given a workspace, but the per-workgroup column-sum partials live in a
 * device-resident array owned by the library (12 MiB per GPU, held until the process ends) and the
 * call allocates nothing.  D <= DINO_LN_STATIC_MAX_D, else DINO_ERR_TOO_LARGE.  The device that owns
 * the buffers must be the current device.  All calls on one GPU share that GPU's array: two calls that
 * overlap in time on one GPU (different streams or threads) race; calls on one stream do not. */
#define DINO_LN_STATIC_MAX_D 1024
int dino_layernorm_bwd_static(int dtype, const void* dy, int64_t dy_stride, const void* x,
                              int64_t x_stride, const float* gamma, const float* mean,
                              const float* rstd, const void* dres, int64_t dres_stride, void* dx,
                              int64_t dx_stride, float* dgamma, float* dbeta, float* dx_colsum,
                              const float* rowscale, int rows_per_group, int64_t rows, int D,
                              void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused multi-head self-attention, head_dim = 64, for Bc crops of N tokens.
 * Replaces Attention.forward's reshape/permute, q@k^T*scale, softmax, attn@v, transpose
 * (vision_transformer.py:124-135) without materialising the [Bc,H,N,N] probabilities.
 * qkv: [Bc*N, 3*H*64] (Q | K | V, head-major: Appendix A of SURVEY.md), out: [Bc*N, H*64],
 * lse: fp32 [Bc, H, N] (natural-log-sum-exp of the scaled scores).  rowscale (nullable,
 * fp32 [Bc]) is the per-crop drop-path factor: out' = rowscale[b] * out is what is stored.
 * Backward takes dout' = d(loss)/d(out') ALREADY multiplied by rowscale[b] and writes dqkv.
 * delta_ws (nullable): fp32 [Bc, H, N] scratch; when given, the dQ kernel leaves
 * rowsum(dout * out) there and the dK/dV kernel reads it instead of reading `out` again.
 * (bf16 with N <= 208 runs ONE fused backward kernel -- every operand read once -- which neither
 * needs nor touches delta_ws; DINO_ATTN_FUSED=0 selects the two-kernel form.)
 * ------------------------------------------------------------------------------------------ */
int dino_attn_fwd(int dtype, const void* qkv, void* out, float* lse, int Bc, int N, int H,
                  const float* rowscale, void* stream);
int dino_attn_bwd(int dtype, const void* qkv, const void* out, const void* dout,
                  const float* lse, void* dqkv, int Bc, int N, int H, const float* rowscale,
                  float* delta_ws, void* stream);

/* Attention of the CLS query alone: row 0 of Attention.forward (vision_transformer.py:122-138) for every (crop, head).
 * VisionTransformer.forward returns x[:, 0] of the last block (:382-383), so in that block only the CLS row of
 * attention / proj / MLP is consumed and only it carries a gradient: the reference computes the other rows and drops
 * them; results are identical without them.  qkv as above; out_cls [Bc, H*64] (times rowscale[b], nullable),
 * lse_cls fp32 [Bc, H].  Backward: dout_cls [Bc, H*64] already multiplied by rowscale[b] (as dino_attn_bwd); writes
 * ALL of dqkv [Bc*N, 3*H*64]: dq in the CLS rows and zeros in the other rows' Q columns, dk_j and dv_j in every row.
 * Any N with 8 * 4 * N bytes of LDS (N <= 2048). */
int dino_attn_cls_fwd(int dtype, const void* qkv, void* out_cls, float* lse_cls, int Bc, int N, int H,
                      const float* rowscale, void* stream);
int dino_attn_cls_bwd(int dtype, const void* qkv, const void* dout_cls, const float* lse_cls, void* dqkv,
                      int Bc, int N, int H, void* stream);

/* Attention probabilities softmax(q k^T / 8) of one block, [Bc, H, N, N] in dtype: what
 * Block.forward(return_attention=True) returns (vision_transformer.py:131-132,220-226) for
 * VisionTransformer.get_last_selfattention (:385-392).  Inspection helper, N <= 1024. */
int dino_attn_probs(int dtype, const void* qkv, void* probs, int Bc, int N, int H, void* stream);

/* ------------------------------------------------------------------------------------------
 * DINO loss forward + backward in one pass (dino/components/loss.py:42-60) and the batch
 * half of update_center (:69).  student: [ncrops*B, K] crop-major, teacher: [2*B, K],
 * center: fp32 [K].  Outputs: loss_parts fp32 [ncrops*B] (per student row; loss = sum(parts),
 * already divided by B*(2*ncrops-2)); row_stats: fp32 [2*B] scratch, dstudent [ncrops*B, K] = d loss / d student (for upstream
 * grad 1), center_acc fp32 [K] += column sums of teacher (caller zeroes it).
 * ------------------------------------------------------------------------------------------ */
int dino_loss_fwd_bwd(int dtype, const void* student, const void* teacher, const float* center,
                      void* dstudent, float* loss_parts, float* center_acc, float* row_stats,
                      int B, int ncrops, int K, float inv_student_temp, float inv_teacher_temp,
                      void* stream);
/* loss.py:74-79 after the (optional) all-reduce: center = center*m + center_acc*inv_count*(1-m),
 * center_acc zeroed for the next step; and/or loss_out[0] = sum(loss_parts[nparts]).  Either half may
 * be left out (center == NULL: fold only -- the loss scalar does not wait for the cross-rank
 * all-reduce of center_acc; loss_out == NULL: centre only). */
int dino_center_update(float* center, float* center_acc, int K, float momentum, float inv_count,
                       const float* loss_parts, int nparts, float* loss_out, void* stream);

/* Prototype layer + loss without ever holding the [ncrops*B, K] student logits or their gradient
 * (DINOHead.last_layer + DINOLoss.forward, vision_transformer.py:719 and loss.py:42-60).  The caller walks
 * the K prototypes in chunks of Kc columns; s_chunk is the chunk's logits [ncrops*B, Kc] (row stride lds)
 * from dino_gemm.  phase 0: teacher_lse[2B] = log-sum-exp of (teacher - center) / tau_t per row, run cleared.
 * phase 1: run[row] = {running log-sum-exp of s / tau_s, running sum_k q[k] * s[k] / tau_s} updated with the
 * chunk at columns [k0, k0 + Kc).  phase 2: loss_parts[row] (summed into the loss by dino_center_update).
 * phase 3: s_chunk (recomputed by the GEMM) is overwritten by d loss / d logits * gout[0] (gout nullable = 1),
 * the operand of the two prototype-gradient GEMMs.  teacher: full [2B, K] logits (row stride ldt). */
int dino_proto_loss_chunk(int dtype, int phase, void* s_chunk, int64_t lds, const void* teacher, int64_t ldt,
                          const float* center, float* teacher_lse, float* run, int B, int ncrops, int K, int k0,
                          int Kc, float inv_student_temp, float inv_teacher_temp, const float* gout,
                          float* loss_parts, void* stream);

/* ------------------------------------------------------------------------------------------
 * Flat-buffer optimiser kernels.  All student parameters live in one fp32 buffer (each
 * nn.Parameter is a view), likewise grads, AdamW moments and the teacher copy.
 * chunk table: int64 [nchunks, 3] = (tensor index, start element, length<=DINO_CHUNK).
 * ------------------------------------------------------------------------------------------ */
#define DINO_CHUNK 8192
/* per-chunk sum of squares of grad (compute_gradient_stats, dino/utils/utils.py:78-83) */
int dino_grad_sqnorm(const float* grad, const int64_t* chunks, int nchunks, float* chunk_sq,
                     void* stream);
/* per-tensor norm -> clip coefficient (utils.py:85-90) + stats[0..2] = total norm, max norm,
 * clipped-tensor count.  tensor_chunk_begin: int32 [ntensors+1]; active: int32 [ntensors]
 * (0 = grad is None for that tensor -> excluded from the statistics, coef 1). */
int dino_clip_coef(const float* chunk_sq, const int32_t* tensor_chunk_begin, int ntensors,
                   const int32_t* has_grad, float clip, float* norms, float* coef, float* stats,
                   void* stream);
/* in-place grad *= coef[tensor] (the p.grad.data.mul_ of utils.py:88) */
int dino_grad_scale(float* grad, const int64_t* chunks, int nchunks, const float* coef,
                    void* stream);
/* torch.optim.AdamW step (dino.py:251, train_utils.py:131) fused with the per-tensor clip
 * scale, the momentum-teacher EMA (train_utils.py:152-162) and the bf16 shadow refresh.
 * hp: fp32 [ntensors, 8] = (lr, weight_decay, 1-beta1^t, 1-beta2^t, active, 0,0,0); tensors
 * with active == 0 (grad cancelled, utils.py:101-106) skip AdamW but still get the EMA.
 * coef may be null (no clipping); shadow pointers may be null (fp32 mode); teacher may be null
 * (plain AdamW, no EMA). */
int dino_adamw_ema(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                   float* teacher, void* param_shadow, void* teacher_shadow,
                   const int64_t* chunks, int nchunks, const float* hp, const float* coef,
                   float beta1, float beta2, float eps, float ema_m, float ema_one_minus_m,
                   void* stream);
/* stand-alone EMA over a flat range (same reference lines) + optional shadow refresh */
int dino_ema(float* teacher, const float* student, void* teacher_shadow, int64_t n, float m,
             float one_minus_m, void* stream);
/* fp32 -> bf16 shadow of a flat range */
int dino_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
/* batched 2-D transpose src[r,c] (fp32 master) -> dst[c,r] (dtype): the W^T shadows that turn
 * every dgrad into an NT GEMM.  table: int64 [nmat, 4] = (src offset, dst offset, rows, cols) */
int dino_transpose_batched(int dtype, const float* src_base, void* dst_base, const int64_t* table,
                           int nmat, void* stream);

/* ------------------------------------------------------------------------------------------
 * Token assembly (PatchEmbed + prepare_tokens, vision_transformer.py:242-245,361-375)
 * ------------------------------------------------------------------------------------------ */
/* im2col of fp32 NCHW crops into the patch matrix (k = c*P*P + dy*P + dx, K padded to a multiple
 * of 8 with zeros).  cls_row = 0: [Bc*gh*gw, Kp].  cls_row = 1: [Bc*(1+gh*gw), Kp] with an all-zero
 * first row per crop, so the matrix is row-aligned with the token matrix (CLS slot included). */
int dino_im2col(int dtype, const float* img, void* out, int Bc, int H, int W, int P, int cls_row,
                void* stream);
/* pos table for one resolution: out[0] = pos[0]; out[1+i] = sum_t w[i,t] * pos[1 + idx[i,t]]
 * (the bicubic resize of :345-353 as a fixed 16-tap gather; ntaps == 0 -> plain copy) */
int dino_pos_table(int dtype, const float* pos, const int32_t* idx, const float* w, int ntaps,
                   void* out, int n_dst, int D, void* stream);
/* CLS rows: tokens[b*N + 0, :] = cls + pos[0]   (:369-373) */
int dino_cls_rows(int dtype, void* tokens, const float* cls, const float* pos0, int Bc, int N,
                  int D, void* stream);
/* out_f32[i, d] += sum_b x[(b*N + i), d]: gradient of the broadcast pos/cls add */
int dino_fold_rows(int dtype, const void* x, float* out, int Bc, int N, int D, void* stream);
/* dpos[1 + idx[i,t], :] += w[i,t] * dpos_local[1+i, :]; dpos[0] += dpos_local[0] (fp32) */
/* n_dst may be 0 (only row 0 is scattered: the cls_token gradient) */
int dino_pos_table_bwd(const float* dpos_local, const int32_t* idx, const float* w, int ntaps,
                       float* dpos, int n_dst, int D, void* stream);

/* Stochastic depth for the whole depth in one launch (drop_path, vision_transformer.py:52-62):
 * scales[l, w, c] = floor(keep[l] + u) / keep[l], u ~ U[0,1) from a counter-based generator keyed by
 * (seed, l, w, c) -- l < L live blocks, w = attention / MLP branch, c < Bct crops.  keep: device fp32 [L].
 * row_scales (nullable, fp32 [L, 2, M]): the same values per token row, row_crop[m] = crop of row m. */
int dino_drop_path_scales(uint64_t seed, const float* keep, int L, int Bct, const int64_t* row_crop, int64_t M,
                          float* scales, float* row_scales, void* stream);

/* ToTensor + Normalize of uint8 crops (dino/data/augmentations.py:102-107; the arithmetic is
 * torchvision's to_tensor / normalize, absent from the reference tree: x = float(u8) / 255, then
 * (x - mean[c]) / std[c], each operation rounded once in fp32).  src: uint8 [B,H,W,C] (src_hwc = 1,
 * the PIL / numpy layout ToTensor takes) or [B,C,H,W] (src_hwc = 0); dst: fp32 [B,C,H,W];
 * mean, std: HOST arrays of C floats (C <= 4), copied into the launch. */
int dino_to_tensor_normalize(const uint8_t* src, float* dst, int64_t B, int C, int H, int W,
                             int src_hwc, const float* mean, const float* std, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-crop augmentation on uint8 images resident in HBM (dino/data/augmentations.py:75-157; the
 * reference runs these as torchvision -> Pillow calls on PIL images in DataLoader workers).  Results are
 * bit-identical to Pillow 12's for the same parameters.  The random draws happen on the host; every crop
 * is described by one row of DINO_AUG_PARAM_WORDS int32 words:
 *   0 source image index   1 top  2 left  3 height  4 width of the crop box (RandomResizedCrop, :112-116)
 *   5 flip (RandomHorizontalFlip :90)
 *   6..11  op[slot], slot < DINO_AUG_SLOTS: 0 none, 1 brightness, 2 contrast, 3 saturation, 4 hue
 *          (ColorJitter :93-95, in its drawn order), 5 grey-scale (RandomGrayscale :99), 6 solarize (:60-72)
 *   12..17 factor[slot]: the float factor's bits (ops 1-3) or the uint8 hue shift (op 4)
 *   18 blur on/off, 19 box radius (int), 20 ww, 21 fw  (GaussianBlur :38-57 -> BoxBlur.c's constants for
 *          the drawn radius, computed on the host in single precision as Pillow does)   22, 23 unused
 * Crops are uint8 [ncrops, S, S, 3]; ncrops <= 65535.
 * ------------------------------------------------------------------------------------------ */
#define DINO_AUG_PARAM_WORDS 24
#define DINO_AUG_SLOTS 6
/* F.resized_crop(img, top, left, h, w, (S, S), BICUBIC) + hflip: Pillow's two-pass fixed-point resample
 * (horizontal pass into a uint8 intermediate, then vertical) of the cropped box.  src: uint8 [B, Hs, Ws, 3].
 * Scratch owned by the caller: bounds int32 [ncrops, 2, S, 2], coeffs int32 [ncrops, 2, S, kmax] (kmax >= the
 * largest tap count, ceil(2 * max(box / S, 1)) * 2 + 1), tmp uint8 [ncrops, tmp_rows, S, 3] with tmp_rows >=
 * the tallest box. */
int dino_aug_resized_crop(const uint8_t* src, int B, int Hs, int Ws, const int32_t* params, int ncrops, int S,
                          int kmax, int32_t* bounds, int32_t* coeffs, uint8_t* tmp, int tmp_rows, uint8_t* out,
                          void* stream);
/* sums[crop] = sum over the crop of convert("L") (ImageEnhance.Contrast's mean, ColorJitter :93).  With
 * params != NULL only the crops whose op[slot] is contrast are summed (the others read 0). */
int dino_aug_gray_sums(const uint8_t* img, int ncrops, int S, uint32_t* sums, const int32_t* params, int slot,
                       void* stream);
/* applies op[slot] of every crop in place (ImageEnhance.Brightness/Contrast/Color = Image.blend with a
 * degenerate image; hue through Pillow's uint8 HSV; convert("L"); ImageOps.solarize).  sums: from
 * dino_aug_gray_sums on the CURRENT state of img (read by op 2 only). */
int dino_aug_color_op(uint8_t* img, int ncrops, int S, const int32_t* params, int slot, const uint32_t* sums,
                      void* stream);
/* one box-blur pass (ImageFilter.GaussianBlur = 3 passes along x, then 3 along y, each rounded to uint8)
 * of the crops whose blur word is set; the others are not touched (dst keeps its contents). */
int dino_aug_box_blur(const uint8_t* src, uint8_t* dst, int ncrops, int S, const int32_t* params, int axis,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * Head pieces (DINOHead.forward, vision_transformer.py:716-720)
 * ------------------------------------------------------------------------------------------ */
/* F.normalize(p=2, eps=1e-12) over D (:718); norm: fp32 [rows] saved for backward */
int dino_l2norm_fwd(int dtype, const void* x, void* y, float* norm, int64_t rows, int D,
                    void* stream);
/* dy is fp32 (split-K atomics output of the prototype dgrad); dx in dtype */
int dino_l2norm_bwd(int dtype, const float* dy, const void* y, const float* norm, void* dx,
                    int64_t rows, int D, void* stream);
/* weight_norm (dim=0) effective weight W[k] = g[k] v[k]/||v[k]|| (:703-706) written as the
 * dtype shadow w [K,D] and its transpose wt [D,K]; vnorm: fp32 [K] */
int dino_weightnorm_fwd(int dtype, const float* g, const float* v, void* w, void* wt,
                        float* vnorm, int K, int D, void* stream);
/* dW fp32 [K,D] -> dg [K], dv [K,D] (ACCUMULATED into fp32 grads) */
int dino_weightnorm_bwd(const float* dW, const float* g, const float* v, const float* vnorm,
                        float* dg, float* dv, int K, int D, void* stream);

/* out_f32[n] += sum_m rs(m) * x[m, n]  (bias gradients); rowscale nullable */
int dino_colsum(int dtype, const void* x, int64_t ldx, float* out, int64_t M, int N,
                const float* rowscale, int rows_per_group, void* stream);
/* x *= *scalar (device scalar), skipped when *scalar == 1 (loss.backward() upstream grad) */
int dino_scale_by_scalar(int dtype, void* x, int64_t n, const float* scalar, void* stream);

/* ------------------------------------------------------------------------------------------
 * Diagnostics (no counterpart in the reference)
 * ------------------------------------------------------------------------------------------ */
/* Holds `workgroups` compute units (one 256-thread workgroup each) for `micros` microseconds on `stream` and does
 * nothing else: a one-GPU stand-in for the kernels of another stream -- RCCL's all-reduce of a gradient bucket
 * (dino.py:190-193's DistributedDataParallel, running under backward) -- that the one-workgroup-per-CU GEMM and
 * attention kernels meet on a multi-GPU node.  bench.py --occupy n,us; tests/test_gpu_ops.py.  Bounded: the grid
 * drains after at most ~4x the requested time.  workgroups <= 1024, micros <= 100000. */
int dino_diag_occupy_cus(int workgroups, int micros, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DINO_HIP_H */
