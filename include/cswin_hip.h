/* libcswin_hip -- C ABI of the MI355X-native (gfx950) CSWin-UNet hot path.
 *
 * The reference (BoloniniD/CSWin-UNet) has no FFI of its own: its seam is the nn.Module surface
 * of networks/cswin_unet.py.  Each entry point below replaces the device work behind one piece of
 * that surface (reference file:line cited per function); cswin_unet_amd/_lib.py binds them with
 * ctypes and cswin_unet_amd/networks/cswin_unet.py calls them from modules that keep the
 * reference's class names, constructor signatures and state_dict keys.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch's caching allocator in practice), fp32 unless stated, densely packed row-major.
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*), never synchronise, never
 *     allocate, keep no pointer past return; entry points are re-entrant (no global mutable state
 *     except the thread-local error string) and hipGraph-capturable.
 *   - return 0 on success, a negative CSWIN_ERR_* code otherwise; cswin_last_error() gives the
 *     message for the calling thread.  No exceptions, no exit() (the reference print+exit(0)s on a
 *     bad stripe mode, cswin_unet.py:50-51; here that is CSWIN_ERR_SHAPE).
 *   - "tokens" = the (B, L, C) layout the reference keeps between blocks; L = H*W row-major.
 */
#ifndef CSWIN_HIP_H
#define CSWIN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a prototype or struct below changes (1: round 1; 2: round 2 -- stream / precision / storage arguments; 3: round 3 --
 * cswin_attn_fwd writes y0, cswin_attn_bwd reads it).  cswin_abi_version() returns the value the library was built with: a consumer
 * compiled against another header must refuse to call it. */
#define CSWIN_ABI_VERSION 4

#define CSWIN_OK 0
#define CSWIN_ERR_SHAPE (-1)
#define CSWIN_ERR_ALIGN (-2)
#define CSWIN_ERR_WORKSPACE (-3)
#define CSWIN_ERR_HIP (-4)
#define CSWIN_ERR_UNSUPPORTED (-5)

/* A slab reduction left pending by a producer called with `deferred` != NULL; batch up to 48 of them into one launch with
 * cswin_rows_sum_multi (a CSWinBlock backward has six to eight: four weight gradients, two LayerNorm dgamma/dbeta, the LePE conv
 * gradients; cswin_unet_amd.ops queues the jobs of a whole backward pass and reduces them in one or two launches).
 * conv_kk / conv_cin != 0: columns [0, n_first) are a convolution weight gradient in the implicit-GEMM order [Cout][k*k][Cin] and
 * are stored to `out` in the nn.Conv2d order [Cout][Cin][k][k]. */
typedef struct cswin_reduce_job {
    const float* part;
    float* out;
    float* out2;
    long long n_first, n, stride;
    int rows, reserved;
    int conv_kk, conv_cin;
} cswin_reduce_job;

const char* cswin_last_error(void);
int cswin_abi_version(void);
int cswin_device_ok(void); /* 1 if the current HIP device is gfx950 */
/* `precision` argument of the Linear and convolution entry points (and field of cswin_wgrad_desc): 0 = exact fp32 MFMA (the
 * parity path of BASELINE configs[1]); 1 = operands rounded to bf16 while staged into LDS, bf16 MFMA, fp32 accumulation, fp32
 * tensors in HBM (the "bf16" of BASELINE configs[2..4] as far as the GEMMs go; torch.autocast(bfloat16) on the reference's
 * Linear / Conv2d is the closest reference-side equivalent).  The library keeps NO precision state: entry points are
 * re-entrant and may be called with different precisions from different threads / streams. */

/* ---- LePEAttention (cswin_unet.py:31-109), both branches of a CSWinBlock in one launch (:171-176) ----
 * qkv (B, L, 3C) = output of the qkv Linear, channel layout [q | k | v] (:169).
 * nbranch = 2: branch i works on channels [i*C/2, (i+1)*C/2) of each of q,k,v with stripe mode idx[i]
 *   (0: H_sp=reso, W_sp=split; 1: H_sp=split, W_sp=reso; :43-48) and heads[i] heads;
 * nbranch = 1: whole C, idx[0] = -1 (window = whole map).  Head dim 8, 16, 24 or 32 (equal in both branches); windows of up
 *   to 288 tokens.
 * lepe_w[i] (Cb, 9) / lepe_b[i] (Cb) = get_v depthwise 3x3 weight/bias of branch i (:55).
 * y (B, L, C): x = softmax(scale q k^T) v + lepe, scattered by windows2img and concatenated (:98-107, :174).
 * y0 (B, L, C) or NULL: the same without the lepe term (attn @ v of :103), in y's storage format.  Saved for the backward only:
 *   rowsum(dO o y0) is the row term of the softmax gradient, so the backward neither recomputes LePE(v) nor reduces P o dP
 *   across keys.  Pass NULL when no backward follows.
 * lse (B, sum(heads), L): row log-sum-exp saved for backward.  scale <= 0 selects head_dim^-0.5 (:42). */
int cswin_attn_fwd(const float* qkv, const float* const* lepe_w, const float* const* lepe_b, float* y, float* y0, float* lse,
                   int B, int reso, int C, int nbranch, const int* heads, const int* idx, int split, float scale,
                   float drop_p, unsigned long long drop_seed, const unsigned long long* drop_epoch, int qkv_bf16, void* stream);
/* drop_p in [0, 1) (0 = off): nn.Dropout on the attention probabilities (cswin_unet.py:101, attn_drop_rate): y = ((P o M) v) + lepe
 * with M = keep / (1 - drop_p), keep a counter-based hash of (drop_seed, batch, head, window, query, key).  cswin_attn_bwd called
 * with the same (drop_p, drop_seed) regenerates the mask; the softmax statistics (lse) are those of the undropped P.
 * drop_epoch: NULL, or a DEVICE counter whose value is added to drop_seed when the kernel runs: a captured hipGraph (seed frozen
 * at capture) then draws a new mask on every replay if a kernel in the graph advances the counter once per step. */
size_t cswin_attn_bwd_workspace(int B, int reso, int C, int nbranch, const int* heads, const int* idx, int split);
/* autograd backward of the above: dqkv (B, L, 3C), dlepe_w[i] (Cb, 9), dlepe_b[i] (Cb) are overwritten.
 * y0 = the forward's y0 output (NOT y).  The per-window partial slabs of the LePE conv weight / bias gradient are reduced by
 * one extra launch, or left in deferred[0..nbranch) for cswin_rows_sum_multi. */
int cswin_attn_bwd(const float* qkv, const float* const* lepe_w, const float* const* lepe_b, const float* lse,
                   const float* y0, const float* dy, float* dqkv, float* const* dlepe_w, float* const* dlepe_b,
                   void* workspace, size_t ws_bytes, int B, int reso, int C, int nbranch, const int* heads, const int* idx,
                   int split, float scale, cswin_reduce_job* deferred, float drop_p, unsigned long long drop_seed,
                   const unsigned long long* drop_epoch, int qkv_bf16, void* stream);
/* qkv_bf16: storage mode of both attention entry points.  0: every tensor fp32.  1: qkv (and dqkv) are STORED as bf16 -- the
 * output format of cswin_linear_fwd(io_bf16 bit 1).  3: additionally y and y0 (the forward outputs) are stored
 * as bf16 -- the input format of the proj Linear's io_bf16 bit 0.  In modes 0 - 3 the arithmetic of the attention kernels is
 * fp32 throughout (v_mfma_f32_16x16x4_f32).  7: mode 3 with bf16 MATRIX instructions (v_mfma_f32_16x16x32_bf16 for QK^T and
 * dO V^T, v_mfma_f32_16x16x16_bf16 for P V, dV, dK, dQ): operands (scaled q, k, v, P, dS, dO) are rounded to bf16 on their way
 * into the matrix pipe -- what the reference's softmax(dtype=attn.dtype) @ v does under a bf16 config (cswin_unet.py:100) --,
 * accumulators, softmax statistics, LePE and everything stored are as in mode 3.  dy, lse and the LePE parameters / gradients
 * are fp32. */

/* ---- img2windows / windows2img (cswin_unet.py:184-202): index-only, bit-exact ----
 * img (B, C, H, W) -> out (B*nH*nW, H_sp*W_sp, C);   win (B*nH*nW, H_sp*W_sp, C) -> out (B, H, W, C) */
int cswin_img2windows(const float* img, float* out, int B, int C, int H, int W, int H_sp, int W_sp, void* stream);
int cswin_windows2img(const float* win, float* out, int B, int C, int H, int W, int H_sp, int W_sp, void* stream);

/* ---- nn.LayerNorm over C (cswin_unet.py:168,179,218,341,497,533); C in {32,64,128,256,512,1024} ---- */
int cswin_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                        int M, int C, float eps, int y_bf16, void* stream);
/* y_bf16 != 0: y is STORED as bf16 (bf16 activation storage: the input format of cswin_linear_fwd io_bf16 bit 0 and of
 * cswin_wgrad_desc io_bf16 bit 1); mean / rstd stay fp32. */
size_t cswin_layernorm_bwd_workspace(int M, int C);
/* dx = dres (optional residual-path gradient, may alias dx) + LN backward; dgamma/dbeta overwritten (by the returned
 * job when `deferred` is given, immediately otherwise).  dx_bf16: NULL, or M * C bf16 that receive a rounded copy of dx -- the
 * bf16 mode's GEMMs read that twin (cswin_linear_bwd_data io_bf16 bit 0, cswin_wgrad_desc io_bf16 bit 0) while the residual
 * path keeps the fp32 dx. */
int cswin_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        const float* dres, float* dx, float* dgamma, float* dbeta, void* workspace, size_t ws_bytes,
                        int M, int C, cswin_reduce_job* deferred, void* dx_bf16, void* stream);

/* ---- nn.Linear family: qkv / proj / Mlp.fc1+GELU / fc2 (cswin_unet.py:125,134,17-27), concat_linear{4,3,2}
 *      (:404,417,428 with the torch.cat of :509,518,526 fused as a two-source K loop), 1x1 convs of CARAFE ----
 * acc = [x | x2] (M, K) @ w (N, K)^T + bias.
 *   y_act != NULL : y = acc (pre-activation), y_act = GELU_erf(acc)
 *   residual != NULL : y = residual + row_scale[m / rows_per_sample] * acc     (x + drop_path(f(x)), :178-179)
 * x2 == NULL: single source.  row_scale may be NULL (= 1). */
int cswin_linear_fwd(const float* x, const float* x2, int k_split, const float* w, const float* bias, float* y,
                     float* y_act, const float* residual, const float* row_scale, int rows_per_sample, int M, int N,
                     int K, int precision, int io_bf16, void* stream);
/* io_bf16 (precision 1 only; 0 = every tensor fp32): tensors STORED as bf16 in HBM, fp32 accumulation as before.
 *   cswin_linear_fwd:      bit 0 = x, bit 1 = y and y_act, bit 2 = w;
 *   cswin_linear_bwd_data: bit 0 = dy, bit 1 = dx, bit 2 = w, bit 3 = gelu_pre.
 * Bit 2 reads the weights' bf16 SHADOW (same [N][K] layout; cswin_sgd_flat keeps it current): the GEMM rounds fp32 weights
 * to bf16 while staging them anyway, so results are bit-identical and the weight traffic halves.  Concat / split / add
 * forms accept bit 2 only.
 * dx (M, K) = add + row_scale * ((dy (M, N) @ w (N, K)) * gelu'(gelu_pre));  columns >= k_split go to dx2 if given */
int cswin_linear_bwd_data(const float* dy, const float* w, float* dx, float* dx2, int k_split, const float* gelu_pre,
                          const float* row_scale, int rows_per_sample, const float* add, int M, int N, int K,
                          int precision, int io_bf16, void* stream);
size_t cswin_linear_bwd_weight_workspace(int M, int N, int K);
/* dw (N, K) = (row_scale * dy)^T @ [x | x2];  dbias (N) = column sums (may be NULL); `deferred` as for layernorm_bwd */
int cswin_linear_bwd_weight(const float* dy, const float* x, const float* x2, int k_split, const float* row_scale,
                            int rows_per_sample, float* dw, float* dbias, void* workspace, size_t ws_bytes, int M,
                            int N, int K, cswin_reduce_job* deferred, int precision, void* stream);
/* One problem of cswin_linear_bwd_weight_batch: dw (N, K) = (row_scale * dy)^T @ x, dbias (N) = column sums of dy (or NULL). */
typedef struct cswin_wgrad_desc {
    const float* dy;         /* (M, N) */
    const float* x;          /* (M, K) */
    const float* row_scale;  /* per-sample multiplier of the dy rows, or NULL */
    float* dw;               /* (N, K) */
    float* dbias;            /* (N) or NULL */
    void* workspace;         /* cswin_linear_bwd_weight_workspace(M, N, K) bytes */
    size_t ws_bytes;
    int rows_per_sample, M, N, K;
    int precision;           /* 0 = exact fp32 MFMA, 1 = bf16 operands (all problems of one launch agree) */
    int io_bf16;             /* precision 1 only: bit 0 = dy is stored as bf16, bit 1 = x is stored as bf16 */
} cswin_wgrad_desc;
/* Up to 4 independent weight gradients (the four nn.Linear of a CSWinBlock, cswin_unet.py:125,134,17-19) in ONE launch;
 * deferred[0..n) receive their slab reductions (required: run them with cswin_rows_sum_multi, or hand them to a later call as
 * `pending`).  pending[0..npending), npending <= 16 (may be NULL / 0): reductions left pending by EARLIER calls; they are run by
 * this launch's last workgroups -- memory-bound work beside matrix-pipe-bound work instead of a launch of its own -- or, where the
 * batch does not go out as one launch, by a cswin_rows_sum_multi launch; either way they must not be run again. */
int cswin_linear_bwd_weight_batch(const cswin_wgrad_desc* problems, int n, cswin_reduce_job* deferred, const cswin_reduce_job* pending,
                                  int npending, void* stream);
/* The tail of a CSWinBlock's backward (cswin_unet.py:171 / :125 backward): dx (M, K) = dy (M, N) @ w (N, K) -- the qkv Linear's data
 * gradient, plain fp32 operands -- together with the block's weight gradients (`problems`, `deferred` exactly as above).  Both
 * only wait for dqkv and neither needs the other: in fp32 with 16-B aligned operands they share ONE launch, otherwise the data
 * gradient is launched first and the batch follows; the results are those of cswin_linear_bwd_data + cswin_linear_bwd_weight_batch.
 * pending / npending as for cswin_linear_bwd_weight_batch. */
int cswin_linear_bwd_tail(const float* dy, const float* w, float* dx, int M, int N, int K, const cswin_wgrad_desc* problems, int n,
                          cswin_reduce_job* deferred, const cswin_reduce_job* pending, int npending, void* stream);
/* jobs: host array of 1..48 pending reductions (the workspaces they point into must still be alive) */
int cswin_rows_sum_multi(const cswin_reduce_job* jobs, int njobs, void* stream);

/* ---- convolutions on tokens (NHWC) as implicit GEMM: stage1_conv_embed 7x7 s4 p2 (cswin_unet.py:339),
 *      Merge_Block 3x3 s2 p1 (:208,214-217), CARAFE encoder 3x3 s1 p1 (:228-229,241) ----
 * x (B, H*W, Cin) -> y (B, OH*OW, Cout).  Weights are given in the implicit-GEMM images made by
 * cswin_conv_weight_permute from the nn.Conv2d parameter [Cout][Cin][ks][ks]. Cin % 4 == 0. */
int cswin_conv_tok_fwd(const float* x, const float* w_perm, const float* bias, float* y, int B, int H, int W, int Cin,
                       int Cout, int ks, int stride, int pad, int precision, void* stream);
int cswin_conv_tok_bwd_data(const float* dy, const float* w_permT, float* dx, int B, int H, int W, int Cin, int Cout,
                            int ks, int stride, int pad, int precision, void* stream);
size_t cswin_conv_tok_bwd_weight_workspace(int B, int H, int W, int Cin, int Cout, int ks, int stride, int pad);
/* dw: [Cout][ks*ks][Cin] (torch_layout 0, the image cswin_conv_weight_unpermute takes) or directly the nn.Conv2d parameter
 * layout [Cout][Cin][ks][ks] (torch_layout 1: the slab reduction writes it, no separate unpermute launch).  deferred: NULL, or
 * the slot that receives the slab reduction instead of its launch (as for the Linear weight gradients). */
int cswin_conv_tok_bwd_weight(const float* dy, const float* x, float* dw, float* dbias, void* workspace,
                              size_t ws_bytes, int B, int H, int W, int Cin, int Cout, int ks, int stride, int pad,
                              int torch_layout, cswin_reduce_job* deferred, int precision, void* stream);
/* w [Cout][Cin][ks][ks] -> w_perm [Cout][ks*ks][Cpad] and/or w_permT [ks*ks][Cout][Cpad] (zero padded channels) */
int cswin_conv_weight_permute(const float* w, float* w_perm, float* w_permT, int Cout, int Cin, int ks, int Cpad,
                              void* stream);
int cswin_conv_weight_unpermute(const float* dw_perm, float* dw, int Cout, int Cin, int ks, int Cpad, void* stream);
/* w [Cout][Cin][ks][ks] -> wf [Cin][ks*ks (mirrored)][Cout]: with it the data gradient of a stride-1, pad = ks/2 convolution is
 * cswin_conv_tok_fwd(dy, wf, NULL, dx, B, H, W, Cout, Cin, ks, 1, pad) (CARAFE encoder, cswin_unet.py:228,241) */
int cswin_conv_weight_flipT(const float* w, float* wf, int Cout, int Cin, int ks, void* stream);

/* ---- layout adapters at the ends of the token pipeline (Rearrange 'b c h w -> b (h w) c', cswin_unet.py:340;
 *      view/permute of up_x4, :540-541) ---- */
int cswin_nchw_to_tokens(const float* x, float* y, int B, int C, int H, int W, int Cpad, void* stream);
int cswin_tokens_to_nchw(const float* x, float* y, int B, int C, int H, int W, int Cpad, void* stream);

/* ---- CARAFE / CARAFE4 reassembly (cswin_unet.py:242-264, 292-314): softmax over the 9 taps + weighted
 *      3x3 neighbourhood sum + pixel_shuffle, on tokens.  e (B, H*W, 9*S*S) = encoder output, channel
 *      k*S*S + s;  z (B, H*W, Cz) = features to reassemble (the `out` 1x1 conv is applied BEFORE, at low
 *      resolution: it commutes with the reassembly);  out (B, (S*H)*(S*W), Cz) = bias + reassembly. ---- */
int cswin_carafe_fwd(const float* e, const float* z, const float* bias, float* out, float* wt_save, int B, int H,
                     int W, int Cz, int S, void* stream);
size_t cswin_carafe_bwd_workspace(int B, int H, int W, int Cz, int S);
/* dbias (Cz, may be NULL) = column sums of dout, formed from per-workgroup partial sums in `workspace`; `deferred` as for
 * layernorm_bwd (NULL: reduced immediately; else the job is returned, zeroed when there is no bias) */
int cswin_carafe_bwd(const float* dout, const float* z, const float* wt_save, float* de, float* dz, float* dbias,
                     void* workspace, size_t ws_bytes, int B, int H, int W, int Cz, int S, cswin_reduce_job* deferred, void* stream);

/* ---- loss of the training step: 0.4*CE + 0.6*Dice (trainer.py:55-57, utils.py:9-45) ----
 * logits (B, ncls, HW) fp32, labels (B, HW) int64.  sums[1 + 3*ncls] = {sum -log p[label], intersect_c, y_sum_c,
 * z_sum_c}: all-reduce these across data-parallel ranks for the reference's global-batch Dice. */
size_t cswin_loss_workspace(int B, int ncls, long HW);
/* inputs_are_probs != 0: `logits` already holds class probabilities (DiceLoss(..., softmax=False), utils.py:32-34): no softmax
 * is applied, the gradient is d/d(probabilities) and only the Dice term is meaningful (call with w_ce = 0).
 * class_weight: ncls device floats or NULL (= all 1): utils.py:44 `loss += dice * weight[i]`.
 * A label outside [0, ncls) makes sums[0] (hence the loss) NaN: the device-side counterpart of CrossEntropyLoss raising. */
int cswin_loss_sums(const float* logits, const long long* labels, float* sums, void* workspace, size_t ws_bytes, int B,
                    int ncls, long HW, int inputs_are_probs, void* stream);
int cswin_loss_finalize(const float* sums, float* out3, float* coef, double n_pixels, int ncls, float w_ce,
                        float w_dice, const float* class_weight, void* stream);
int cswin_loss_bwd(const float* logits, const long long* labels, const float* coef, const float* grad_out,
                   float* dlogits, float ce_scale, float dice_scale, int B, int ncls, long HW, int inputs_are_probs,
                   void* stream);

/* ---- optimiser: torch.optim.SGD(momentum, weight_decay) (trainer.py:42,60) on one flat buffer ---- */
int cswin_sgd_flat(float* p, const float* g, float* m, long n, const float* lr_dev, float momentum,
                   float weight_decay, float grad_scale, void* shadow_bf16, void* stream);
/* shadow_bf16: NULL, or n bf16 that receive the updated parameters rounded to nearest even (the bf16 mode's working copy of
 * the fp32 master weights, read by the Linears' io_bf16 bit 2). */
/* table: device array of {const float* src; float* dst; long long n;} (24-byte records), one workgroup each */
int cswin_multi_copy(const void* table, int nchunks, void* stream);

/* ---- nn.Dropout(p) of the reference (cswin_unet.py:20,25,27 Mlp.drop; :135 proj_drop; :346 pos_drop), fused with the residual
 * add + DropPath row factor that follows it where there is one (:178-179):
 *   keep(i) = counter-based hash of (seed, i) >= p   (no mask tensor: backward regenerates it from the same seed)
 *   forward : y[i]  = (residual ? residual[i] : 0) + row_scale[i / (rows_per_sample * C)] * keep(i) / (1 - p) * x[i]
 *   backward: dx[i] =                               row_scale[...]                          * keep(i) / (1 - p) * dy[i]
 * x / y / dy / dx: n floats, 16-B aligned; row_scale: per-sample floats or NULL; elems_per_sample = L * C.  In place (y == x) is
 * allowed.  The random stream is this library's own: torch's Philox stream cannot be reproduced (parity unpinned, as for
 * DropPath); tests extract the mask by running the kernel on ones. */
int cswin_dropout(const float* x, const float* residual, const float* row_scale, float* y, long n, long elems_per_sample,
                  float p, unsigned long long seed, const unsigned long long* seed_epoch, void* stream);
/* seed_epoch: as drop_epoch of cswin_attn_fwd (device counter added to the seed; NULL: none). */
/* bf16 gradient wire for the data-parallel all-reduce (replaces DataParallel's fp32 reduce_add, trainer.py:37-38):
   fp32 -> bf16 round-to-nearest-even / bf16 -> fp32 over n elements (src of pack, dst of unpack 16-B aligned) */
int cswin_pack_bf16(const float* src, void* dst_bf16, long n, void* stream);
/* dst = bf16(scale * src): the trainer packs gradient buckets pre-divided by the world size, so the collective's bf16 sum IS the
   mean (eight ranks' sum would otherwise spend three of bf16's eight mantissa bits on the factor 8 it is divided by afterwards) */
int cswin_pack_bf16_scaled(const float* src, void* dst_bf16, long n, float scale, void* stream);
int cswin_unpack_bf16(const void* src_bf16, float* dst, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CSWIN_HIP_H */
