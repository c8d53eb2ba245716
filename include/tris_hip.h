/* tris_hip.h -- C ABI of libtris_hip.so: the MI355X (gfx950) kernels of the TRIS Stage-1 hot path.
 *
 * The reference (fawnliu/TRIS) has no native layer and no FFI: its hot path is PyTorch ops called from Python
 * (SURVEY.md 8b).  This header is therefore the boundary a maintainer binds *beneath* the reference's Python
 * call sites: each entry point below names the reference op (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (ids: int64) unless noted; no torch types, no hidden allocation:
 *     scratch is passed in (`workspace`, size from the matching *_workspace_bytes helper or documented inline);
 *   - activations are channels-last: an NHWC tensor is the row-major matrix [B*H*W, C];
 *   - conv weights are [Cout][kh][kw][Cin] (= torch.channels_last memory of the reference's [Cout,Cin,kh,kw]);
 *   - `stream` is a hipStream_t (0 = default stream); launches are asynchronous and re-entrant.  The ONLY state the
 *     library keeps is host-side configuration read at launch time: the arithmetic mode of the dense products (a
 *     process-wide default, tris_set_gemm_mode, plus a per-thread override, tris_set_gemm_mode_thread), the autotune
 *     switch with its per-process cache of tuned (tile, split-K) choices (tris_set_autotune), and the developer options of
 *     tris_set_option (read from the environment once, when the library is loaded).  No entry point keeps
 *     device state between calls (the SyncBN mailboxes of tris_mbox_* are caller-owned buffers);
 *   - return value: 0 on success, otherwise a hipError_t.
 */
#ifndef TRIS_HIP_H
#define TRIS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dense products -------------------------------------------------------------------------------------------
 * C[b] = act(alpha * opA(A[b]) . opB(B[b]) + bias) + resid[b]        fp32 in / fp32 out; evaluated in the arithmetic mode
 *   in force for the launching thread (tris_set_gemm_mode below; default split-bf16 "x3", fp32-class accuracy)
 *   opA: transA=0 -> A[m*lda+k], 1 -> A[k*lda+m];  opB: transB=0 -> B[k*ldb+n], 1 -> B[n*ldb+k]
 *   bias_mode 1: bias[n], 2: bias[m];  act 0 none, 1 ReLU, 2 QuickGELU x*sigmoid(1.702x)
 *   workspace (optional, batch==1): enables split-K for small output grids; any size, used opportunistically.
 * Replaces nn.Linear / 1x1 nn.Conv2d / torch.bmm / matmul on the path: CLIP/clip/model.py:17,27,45 (1x1 convs),
 * :369-376 (MHA projections, MLP), :562 (text_projection); model/model_stage1.py:36-37,61-63,75; model/attn.py:73-109,
 * 118-128. */
int tris_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long lda, long ldb, long ldc,
                  int transA, int transB, int batch, long strideA, long strideB, long strideC, const float* bias,
                  int bias_mode, const float* resid, long ldr, long strideR, int act, float alpha, float* workspace,
                  long workspace_bytes, void* stream);

/* One-shot: the NEXT tris_gemm_f32 launched from the calling thread (batch 1) also
 *   - stores the value BEFORE the activation to pre_out (layout of C) -- with act = 2 one launch yields QuickGELU(x W^T + b) and the
 *     pre-activation its backward needs (CLIP/clip/model.py:361-376, the transformer MLP);
 *   - multiplies what it stores by quickgelu'(dact_x[m, n]) (layout of C; applied last) -- the data gradient of the Linear behind a
 *     QuickGELU comes out as the gradient of the pre-activation.
 * Either pointer may be NULL.  The armed product runs without split-K and returns TRIS_DECLINED (nothing launched, arming consumed)
 * when the fast kernel does not serve its operands. */
int tris_gemm_epilogue_next(float* pre_out, const float* dact_x);

/* One-shot: the NEXT tris_gemm_f32 launched from the calling thread may finish a split-K product INSIDE its own launch: each tile's
 * k slices take a ticket from tickets[tile], and the block that arrives last sums the tile's slabs in slice order (the result is
 * the separate reduce launch's, bit for bit, whichever block that is), applies bias / activation / residual and writes C.  Used for
 * up to 8 slices (developer option FUSE_SPLITK = n, 0 = never), one batch, the fast kernels.  `tickets`: `count` ints in device
 * memory, ZERO when handed over; every launch leaves them zero; private to the stream of the launch (two products in flight at
 * once must not share an array; a captured launch keeps reading the same array at every replay).  Without it, or with fewer ints
 * than the product has tiles, the slabs are summed by a second launch.  (The small products of the transformer towers and of the
 * late trunk stages -- reference CLIP/clip/model.py:366-397 -- are ~190 such pairs of launches per training step.) */
int tris_splitk_tickets_next(int* tickets, int count);
/* how many product launches of this process took the fused finish so far (tests, bench.py) */
long tris_splitk_fused_launches(void);

/* Arithmetic of the dense-product kernels.  tris_set_gemm_mode sets the process-wide DEFAULT; tris_set_gemm_mode_thread
 * sets an override for the calling thread only (-1 = none) -- e.g. the autograd thread running the weight-gradient
 * products in another mode -- so that concurrent launches from other threads are unaffected.  Both are host-side values
 * read when a product is launched; tris_get_gemm_mode returns the mode in force for the calling thread.  0 = v_mfma_f32_32x32x2_f32 (f32 in, bit-equal to an fmaf chain);
 * 1 = split-bf16 "x3": every fp32 operand is the exact sum of three bf16 pieces, the six significant piece products
 * run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -> fp32-class accuracy (measured: error vs fp64 <= the f32-MFMA
 * path's) at up to 2.6x the f32-MFMA peak;  3 = "h2": two fp16 pieces per operand, three f16 MFMAs, power-of-two operand
 * scales (see tris_h2_next below; selected process-wide it runs with unit scales: tests).  Default: 1. */
int tris_set_gemm_mode(int mode);
int tris_set_gemm_mode_thread(int mode);
/* Developer options (tests, tools): name = FORCE_TILE (128x128|128x64|64x64|128x32|256x128), FORCE_PIPE (0|1), PIPE (0|1),
 * CONV_DIRECT (0 = implicit GEMM only, 1..6 = that direct configuration where it applies), WGRAD_DIRECT (0, 1..5), BN_FOLD (0|1),
 * STEM_CONV1 (0|1), WG_BLOCKS (n), STREAM_FORM (0|1: streaming form of the element-wise passes over more than the memory-side cache), COL_BLOCKS (n),
 * TUNE_LOG (file); a leading "TRIS_" is accepted; value NULL or "" restores the default.  Initial
 * values come from the environment variables TRIS_<name>, read once when the library is loaded -- nothing reads the environment
 * per call.  tris_set_conv_direct_thread: CONV_DIRECT for the products of the calling thread only (-1 = none): batch-invariant
 * evaluation pins the implicit 3x3 kernels this way. */
int tris_set_option(const char* name, const char* value);
int tris_set_conv_direct_thread(int value);

/* (tile, split-K) selection of the dense-product kernels: 1 (default; env TRIS_AUTOTUNE) = every admissible pair is timed
 * once per product shape on first use and the fastest is cached for the process; 0 = the static cycle model (deterministic
 * run-to-run rounding). */
int tris_set_autotune(int on);
int tris_get_gemm_mode(void);
/* Diagnostics: launches so far of the direct 3x3 convolution kernels (kind 0: forward / data gradient) and of the direct 3x3
 * weight-gradient kernel (kind 1) -- the tests use it to prove which kernel ran.  Host-side counters, not thread-safe. */
long tris_direct_launches(int kind);

/* 3x3 convolution, pad 1, no im2col buffer.  CLIP/clip/model.py:21 (Bottleneck.conv2), :212-229 (stem).
 * fwd: stride 1 or 2.  dgrad: stride 1 only (the only strided conv on the path, the stem's conv1, reads the image and
 * needs no input gradient).  wgrad: split over output pixels; workspace >= 2 slabs of Cout*9*Cin floats, more = faster.
 * Each entry point picks, per shape (timed once like the GEMM tile choice; static table when autotuning is off), between
 * the implicit GEMM and a direct kernel that splits every input value once per window instead of once per tap (x3
 * arithmetic, stride 1, Cin % 16 == 0).  Same result up to fp32 summation order. */
int tris_conv3x3_fwd_f32(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout, int stride,
                         void* stream);
int tris_conv3x3_dgrad_f32(const float* dY, const float* Wt, float* dX, int B, int H, int W, int Cin, int Cout,
                           void* stream);
int tris_conv3x3_wgrad_f32(const float* X, const float* dY, float* dW, int B, int H, int W, int Cin, int Cout,
                           int stride, float* workspace, long workspace_bytes, void* stream);

/* BatchNorm + ReLU folded into the consuming 3x3 convolution (CLIP/clip/model.py:45-46 `relu(bn1(conv1(x)))` -> conv2, and the
 * stem's conv -> bn -> relu -> conv chain :255-258): conv3x3(relu((X - mean) * invstd * gamma + beta)), stride 1, with the
 * normalised tensor formed inside the direct kernels' window staging (tris_bn_apply_f32's expression) -- it never exists in HBM.
 * tris_conv3x3_bnin_ok (host query, x3 arithmetic only) says whether direct kernels serve BOTH the forward product and the weight
 * gradient of the shape; when it returns 0 the caller materialises the BatchNorm output as usual.  fwd: stat_part / stat_rows
 * as in tris_conv3x3_fwd_bnstat_f32 (NULL: no statistics).  The data gradient is tris_conv3x3_dgrad_f32 (it does not read X). */
int tris_conv3x3_bnin_ok(int B, int H, int W, int Cin, int Cout);
int tris_conv3x3_fwd_bnin_f32(const float* X, const float* mean, const float* invstd, const float* gamma, const float* beta,
                              const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout, double* stat_part,
                              int* stat_rows, void* stream);
int tris_conv3x3_wgrad_bnin_f32(const float* X, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                const float* dY, float* dW, int B, int H, int W, int Cin, int Cout, float* workspace,
                                long workspace_bytes, void* stream);

/* Forward conv / 1x1 conv (A[M,K] . B[N,K]^T) with the train-mode BatchNorm statistics of the OUTPUT fused into the
 * epilogue: stat_part <- [rows][2][N] fp64 partial (sum, sum of squares), *stat_rows (HOST int) <- rows, or 0 when the
 * shape is not eligible (then call tris_bn_stats_f32).  stat_part capacity: ceil(M/128)*2*N doubles (rows <= ceil(M/128):
 * one per M tile of the kernel that ran).  Finish with
 * tris_bn_finalize_f32.  (CLIP/clip/model.py:17-29,45-48: conv -> bn pairs of Bottleneck / stem) */
int tris_gemm_bnstat_f32(const float* A, const float* B, float* C, int M, int N, int K, double* stat_part,
                         int* stat_rows, void* stream);
int tris_conv3x3_fwd_bnstat_f32(const float* X, const float* Wt, float* Y, int B, int H, int W, int Cin, int Cout,
                                int stride, double* stat_part, int* stat_rows, void* stream);
int tris_bn_finalize_f32(const double* part, int rows, long M, int C, float eps, float momentum, float* stats,
                         float* running_mean, float* running_var, void* stream);

/* ---- BatchNorm2d (training: batch statistics; CLIP/clip/model.py:18,22,28,39; train_stage1.py:288) ---------------------
 * stats = [mean | invstd | biased var], 3*C floats.  running_* may be NULL (no update).  workspace:
 * tris_col_workspace_bytes(M, C).  Eval mode = tris_bn_apply with mean=running_mean, invstd=rsqrt(running_var+eps). */
long tris_col_workspace_bytes(long M, int C);
int tris_bn_stats_f32(const float* X, long M, int C, float eps, float momentum, float* stats, float* running_mean,
                      float* running_var, float* workspace, void* stream);
/* SyncBatchNorm (train_stage1.py:69): gathered = [world][3*C] = every rank's `stats` block [mean | invstd | var] as written
 * by tris_bn_stats_f32 / tris_bn_finalize_f32 (all_gather it as is); each rank contributed count_per_rank rows. */
int tris_bn_sync_combine_f32(const float* gathered, int world, int C, long count_per_rank, float eps, float momentum,
                             float* stats, float* running_mean, float* running_var, void* stream);
/* Y = (X-mean)*invstd*gamma+beta (+resid) (ReLU) -- also the residual add + relu3 of Bottleneck.forward :52-54 */
int tris_bn_apply_f32(const float* X, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      const float* resid, float* Y, long M, int C, int relu, void* stream);
/* backward: dz = dY * (Y > 0) when Y != NULL.  sum_dz = dbeta, sum_dzx = dgamma.  apply: dZ (optional) receives dz,
 * the gradient of the fused residual branch (Bottleneck identity path).  gamma_mask / beta_mask != NULL (BatchNorm + ReLU
 * without a residual input): the ReLU mask is recomputed from X with tris_bn_apply_f32's own expression -- same sign bit
 * for bit -- and Y is not read (one 4-byte stream less in each of the two passes).  reduce, dz_out != NULL: the masked
 * gradient dz is also written; tris_bn_bwd_apply_f32 may then be given dY = dz_out, Y = NULL, dZ = NULL (dz is already
 * the residual branch's gradient): 7 activation-sized streams over the two passes instead of 8. */
/* The reduce pass fused into the PRODUCER of dY where that is a 1x1-convolution / Linear data gradient (the next layer's
 * backward, CLIP/clip/model.py:42-55 read in reverse): dZ[M,N] = mask(dY[M,K] . Wt[K,N] (+ resid)) with the mask of the
 * BatchNorm(+ReLU) whose raw input is bn_x [M,N] -- from its output bn_y (residual form), or, bn_y == NULL, recomputed from
 * bn_x / gamma / beta -- and part <- [rows][2][N] fp64 partial (sum dz, sum dz*xhat), *part_rows (HOST int) <- rows, or 0 when
 * the shape is not eligible (nothing launched: run tris_gemm_f32 + tris_bn_bwd_reduce_f32).  part capacity: ceil(M/128)*2*N
 * doubles.  Finish with tris_part_finalize_f32 -> sum_dz, sum_dzx, then tris_bn_bwd_apply_f32(dY = dZ, Y = NULL, dZ = NULL,
 * beta_mask = NULL).  One activation-sized read and one write less than the separate reduce pass. */
int tris_gemm_bnbwd_f32(const float* dY, const float* Wt, float* dZ, int M, int N, int K, const float* resid, long ldr,
                        const float* bn_x, const float* bn_y, const float* mean, const float* invstd, const float* gamma,
                        const float* beta, double* part, int* part_rows, void* stream);
/* the same for a 3x3 convolution's data gradient (stride 1, pad 1; bn1 -> conv2 of Bottleneck, the stem's bn1/bn2): the input
 * of the convolution is relu(bn(bn_x)), no residual, mask recomputed from bn_x. */
int tris_conv3x3_dgrad_bnbwd_f32(const float* dY, const float* Wt, float* dZ, int B, int H, int W, int Cin, int Cout,
                                 const float* bn_x, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, double* part, int* part_rows, void* stream);
int tris_part_finalize_f32(const double* part, int rows, int C, float* out0, float* out1, void* stream);

/* Arithmetic "h2" for single dense products: an operand x, multiplied by a power of two s that brings the
 * largest magnitude of its TENSOR (or an upper bound of it) to [2^13, 2^14), is held as two fp16 pieces, x s = hi + lo' 2^-11
 * (hi = fp16(x s), lo' = fp16((x s - hi) 2^11): 22 significand bits + sign); a product is three f16 MFMAs -- hi x hi into one fp32
 * accumulator, hi x lo' and lo' x hi into a second that joins with the weight 2^-11 in the epilogue -- half the matrix work of the
 * x3 default.  Because lo' is stored pre-scaled it stays a normal fp16 number as long as hi does: an element keeps all 22 bits
 * down to 2^-27 of the tensor's maximum; below that the error is absolute, <= 2^-49 of the maximum.  Error of a product:
 * fp32-class relative term + K amaxA amaxB 2^-49.
 * An amax word is 2048 unsigned words (8 KB: 8 XCDs x 16 cache lines with one used word each; the consumer takes the max).
 * tris_amax_bits_f32: atomic max of |x| as a bit pattern into that block (zero it first).  tris_h2_next: arms the CALLING THREAD -- the
 * next dense product it launches (tris_gemm_f32, tris_gemm_bnstat_f32, tris_gemm_bnbwd_f32, tris_conv3x3_{fwd,fwd_bnstat,dgrad,
 * dgrad_bnbwd,wgrad,fwd_bnin,wgrad_bnin}_f32) runs in h2 with operand A scaled from *amaxA (or, amaxA == NULL, by scaleA; 0 = 1.0)
 * and B likewise; products the fast kernels do not serve run as usual.  One shot: an arming never survives the call it was made for.
 * Operand order: forward / data gradient A = activation, B = weights; weight gradients A = dY, B = the convolution's input (for the
 * *_bnin forms: the bound word of tris_bn_out_bound_f32, since relu(bn(X)) is never materialised). */
int tris_amax_bits_f32(const float* x, long n, unsigned* out, void* stream);
/* the same for nseg tensors base + offs[i] (sizes[i] floats; device arrays) in one launch -> slots[i]: the weights of an arena */
int tris_amax_segments_f32(const float* base, const long* offs, const long* sizes, int nseg, unsigned* slots, void* stream);
/* one shot, calling thread: the next tris_bn_apply_f32 / tris_bn_apply_pool_f32 / tris_bn_bwd_apply_f32 /
 * tris_bn_bwd_apply_pool_f32 / tris_layernorm_fwd_f32 / tris_layernorm_bwd_f32 (dX) / tris_elementwise_f32 / tris_gemm_f32 (C) /
 * tris_mha_fwd_f32 / tris_mha_mfma_fwd_f32 (out) / tris_mha_bwd_f32 / tris_mha_mfma_bwd_f32 (dqkv) also maxes the magnitude bits of
 * what it WRITES into *out (zeroed by the caller): the amax of an h2 operand as a by-product of the pass that produces the tensor,
 * instead of a pass of its own.  Each of these entry points TAKES the arming first thing, whether or not it then launches anything. */
int tris_amax_next(unsigned* out);
int tris_h2_next(const unsigned* amaxA, const unsigned* amaxB, float scaleA, float scaleB);
/* Operand planes ("P8", csrc/planes.h; DESIGN.md "operand planes"): a tensor in the byte geometry of its fp32 original whose every 8
 * consecutive elements of the contiguous dimension are 16 bytes of fp16 hi pieces followed by 16 bytes of pre-scaled fp16 lo pieces,
 * x s = hi + lo' 2^-11 with the power-of-two scale s that *word implies (an amax word holding the tensor's largest magnitude or an
 * upper bound of it, final BEFORE the tensor is written).  tris_h2_planes_f32 / tris_h2_unplanes_f32 convert (n % 8 == 0);
 * tris_h2_planes_segments_f32 converts nseg tensors base + offs[i] (sizes[i] floats, word slots + 2048 slot_index[i]; device arrays)
 * of one flat buffer into out_base + offs[i] in one launch (the convolution weights of an optimiser arena, once per step).
 * tris_h2_next_planes arms the calling thread like tris_h2_next, for a product whose operands A and B are BOTH such plane tensors
 * (same pointers, shapes and leading dimensions as the fp32 form; contiguous dimensions multiples of 8): the kernels store the
 * pieces as they arrive instead of splitting fp32 values, and the result is bit-identical to the h2 product of the fp32 originals at
 * the same scales.  flags bit 0: the bn_y operand of tris_gemm_bnbwd_f32 is a plane tensor as well; bit 1: the weight operand of
 * tris_gemm_bnbwd_f32 is given TRANSPOSED ([N][K] = W^T, tris_h2_planes_t_segments_f32).  A product the fast kernels do not
 * serve FAILS (hipErrorInvalidValue): there is no fp32 operand to fall back on; the *_bnin forms refuse planes. */
/* Bit masks (round 6): tris_bn_mask_next(mask) arms the calling thread's NEXT tris_bn_apply_pl_f32 to write, beside its plane output
 * [M, C], the ReLU mask of that output as one BYTE per 8 channels (M C / 8 bytes; bit t <=> channel 8 g + t is positive, decided as a
 * reader of the planes would); tris_h2_next_planes flags bit 2 tells tris_gemm_bnbwd_f32 that its bn_y argument IS that byte array
 * (1 bit per element read in the fused BatchNorm-backward epilogue instead of the 4-byte plane element: model.py:42-55's
 * relu(bn3(..) + identity) outputs are the widest tensors of a Bottleneck). */
int tris_bn_mask_next(unsigned char* mask);
int tris_h2_planes_f32(const float* x, float* planes_out, long n, const unsigned* word, void* stream);
int tris_h2_planes_segments_f32(const float* base, const long* offs, const long* sizes, const long* slot_index, int nseg,
                                const unsigned* slots, float* out_base, void* stream);
/* the same for nseg row-major matrices [rows[i]][cols[i]] (multiples of 64), written TRANSPOSED: planes of W^T [cols][rows] -- the 1x1
 * convolution weights as the B^T operand of their data gradient */
int tris_h2_planes_t_segments_f32(const float* base, const long* offs, const long* rows, const long* cols, const long* slot_index,
                                  int nseg, const unsigned* slots, float* out_base, void* stream);
int tris_h2_unplanes_f32(const float* planes, float* out, long n, const unsigned* word, void* stream);
int tris_h2_next_planes(const unsigned* amaxA, const unsigned* amaxB, int flags);
/* The RN50 trunk's element-wise passes with plane OUTPUT (csrc/planes.hip; reference: CLIP/clip/model.py:42-55 Bottleneck.forward,
 * the BatchNorm2d / ReLU / AvgPool2d calls between its convolutions).  Same arithmetic as tris_bn_apply_f32 / tris_bn_apply_pool_f32 /
 * tris_avgpool2_fwd_f32 / tris_bn_bwd_apply_f32 / tris_bn_bwd_apply_pool_f32 on groups of 8 channels (C % 8 == 0); the result is split
 * with the scale of *out_word, which must hold an upper bound of the output's magnitude BEFORE the launch:
 *   tris_bn_out_bound2_f32: *out <- bits of max_c(|gamma[c]| xhat_max + |beta[c]|) + (add_word ? value of *add_word : 0) -- Samuelson's
 *     bound of a train-mode BatchNorm output (xhat_max = sqrt(rows - 1)) plus the bound of the residual it is added to;
 *   tris_bn_bwd_bound_f32: *out <- bits of max_c |gamma invstd| (amax_dz + |sum_dz| inv_count + xhat_max |sum_dzx| inv_count), the
 *     bound of dx = gamma invstd (dz - mean(dz) - xhat mean(dz xhat)); *dz_word: amax of the masked upstream gradient, left by the pass
 *     that reduced it (tris_amax_next before tris_bn_bwd_reduce_f32 / _pool_f32 / tris_gemm_bnbwd_f32 / tris_conv3x3_dgrad_bnbwd_f32).
 * resid_kind: 0 none, 1 fp32, 2 planes scaled by *resid_word.  Ypl of the backward: the BatchNorm's OUTPUT as planes (its ReLU mask is
 * "a piece is non-zero"), or NULL with beta_mask (mask recomputed from X) or without (no ReLU).  tris_avgpool2_fwd_pl_f32: planes in,
 * planes out at the SAME scale word (a mean never exceeds the bound of its terms). */
int tris_bn_apply_pl_f32(const float* X, const float* mean, const float* invstd, const float* gamma, const float* beta,
                         const float* resid, int resid_kind, const unsigned* resid_word, float* Ypl, const unsigned* out_word, long M,
                         int C, int relu, void* stream);
int tris_bn_bwd_reduce_pl_f32(const float* dY, const float* Ypl, const float* X, const float* mean, const float* invstd, long M, int C,
                              float* sum_dz, float* sum_dzx, float* workspace, float* dz_out, void* stream);
int tris_bn_apply_pool_pl_f32(const float* X, const float* mean, const float* invstd, const float* gamma, const float* beta,
                              float* Ypl, const unsigned* out_word, int B, int H, int W, int C, void* stream);
int tris_avgpool2_fwd_pl_f32(const float* Xpl, float* Ypl, const unsigned* word, int B, int H, int W, int C, void* stream);
int tris_bn_bwd_apply_pl_f32(const float* dY, const float* Ypl, const float* X, const float* mean, const float* invstd,
                             const float* gamma, const float* sum_dz, const float* sum_dzx, float inv_count, float* dXpl,
                             const unsigned* out_word, float* dZ, long M, int C, const float* beta_mask, void* stream);
int tris_bn_bwd_apply_pool_pl_f32(const float* dYp, const float* X, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, const float* sum_dz, const float* sum_dzx, float inv_count, float* dXpl,
                                  const unsigned* out_word, int B, int H, int W, int C, void* stream);
int tris_bn_out_bound2_f32(const float* gamma, const float* beta, int C, float xhat_max, const unsigned* add_word, unsigned* out,
                           void* stream);
/* tris_bn_finalize_f32 + tris_bn_out_bound2_f32, and tris_part_finalize_f32 + tris_bn_bwd_bound_f32, as ONE launch each (the bound is
 * atomically maxed into the zeroed word by the threads that finish the channels) */
int tris_bn_finalize_bound_f32(const double* part, int rows, long M, int C, float eps, float momentum, float* stats,
                               float* running_mean, float* running_var, const float* gamma, const float* beta, float xhat_max,
                               const unsigned* add_word, unsigned* bound_out, void* stream);
int tris_part_finalize_bound_f32(const double* part, int rows, int C, float* out0, float* out1, const float* gamma, const float* invstd,
                                 float inv_count, float xhat_max, const unsigned* dz_word, unsigned* bound_out, void* stream);
int tris_bn_bwd_bound_f32(const float* gamma, const float* invstd, const float* sum_dz, const float* sum_dzx, int C, float inv_count,
                          float xhat_max, const unsigned* dz_word, unsigned* out, void* stream);
/* *out (an amax word, zeroed by the caller) <- bits of max over c of |gamma[c]| * xhat_max + |beta[c]|: an upper bound of
 * |bn(x)| (and of relu(bn(x))) for a train-mode BatchNorm over `count` rows when xhat_max = sqrt(count - 1) (Samuelson's inequality:
 * no element lies further than sqrt(n - 1) standard deviations from the mean) */
int tris_bn_out_bound_f32(const float* gamma, const float* beta, int C, float xhat_max, unsigned* out, void* stream);
int tris_bn_bwd_reduce_f32(const float* dY, const float* Y, const float* X, const float* mean, const float* invstd,
                           long M, int C, float* sum_dz, float* sum_dzx, float* workspace, const float* gamma_mask,
                           const float* beta_mask, float* dz_out, void* stream);
int tris_bn_bwd_apply_f32(const float* dY, const float* Y, const float* X, const float* mean, const float* invstd,
                          const float* gamma, const float* sum_dz, const float* sum_dzx, float inv_count, float* dX,
                          float* dZ, long M, int C, const float* beta_mask, void* stream);
/* out[n] = sum_m X[m*ld+n]  (bias gradients).  workspace: tris_col_workspace_bytes(M, N) */
int tris_colsum_f32(const float* X, long M, int N, long ld, float* out, float* workspace, void* stream);

/* ---- InstanceNorm2d(affine) [+ReLU] on [B, P, C]  (model/attn.py:75,80,85,106) ----------------------------------------
 * (both honour tris_amax_next: the amax word of Y / dX as a by-product) */
int tris_instnorm_fwd_f32(const float* X, const float* gamma, const float* beta, float* Y, float* mean, float* invstd,
                          int B, int P, int C, float eps, int relu, void* stream);
int tris_instnorm_bwd_f32(const float* dY, const float* Y, const float* X, const float* gamma, const float* mean,
                          const float* invstd, float* dX, float* dgamma_part /*[B,C]*/, float* dbeta_part /*[B,C]*/,
                          int B, int P, int C, int relu, void* stream);

/* ---- LayerNorm over the last dim W <= 1024  (CLIP/clip/model.py:352-358) ------------------------------------------------ */
int tris_layernorm_fwd_f32(const float* X, const float* gamma, const float* beta, float* Y, float* mean, float* rstd,
                           long rows, int W, float eps, void* stream);
long tris_layernorm_bwd_workspace_bytes(long rows, int W);
int tris_layernorm_bwd_f32(const float* dY, const float* X, const float* gamma, const float* mean, const float* rstd,
                           float* dX, float* dgamma, float* dbeta, long rows, int W, float* workspace, const float* extra /* optional: added to dX */, void* stream);

/* ---- pooling / elementwise / layout -------------------------------------------------------------------------------- */
/* BatchNorm + ReLU + AvgPool2d(2) as ONE op (the stem's bn3 -> avgpool, model.py:29-31,231-237; bn2 -> avgpool of the stride-2
 * Bottlenecks, :21-25,46-49): Yp [B,H/2,W/2,C] = avgpool2(relu(bn(X))) -- the full-size activation is never written; backward:
 * the two passes read the POOLED upstream gradient dYp (a row's gradient is a quarter of its pooled pixel's), the ReLU mask is
 * recomputed from X.  H and W even. */
int tris_bn_apply_pool_f32(const float* X, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           float* Yp, int B, int H, int W, int C, void* stream);
int tris_bn_bwd_reduce_pool_f32(const float* dYp, const float* X, const float* mean, const float* invstd, int B, int H, int W,
                                int C, float* sum_dz, float* sum_dzx, float* workspace, const float* gamma, const float* beta,
                                void* stream);
int tris_bn_bwd_apply_pool_f32(const float* dYp, const float* X, const float* mean, const float* invstd, const float* gamma,
                               const float* beta, const float* sum_dz, const float* sum_dzx, float inv_count, float* dX, int B,
                               int H, int W, int C, void* stream);
int tris_avgpool2_fwd_f32(const float* X, float* Y, int B, int H, int W, int C, void* stream); /* model.py:25,37,231 */
int tris_avgpool2_bwd_f32(const float* dY, float* dX, int B, int H, int W, int C, void* stream);
#define TRIS_EW_ADD 0       /* O = A + B */
#define TRIS_EW_AXPY 1      /* O = s*A + B          (model_stage1.py:73-74, the 0.1 residual mix) */
#define TRIS_EW_RELU_BWD 2  /* O = A * (B > 0)      A = dY, B = Y */
#define TRIS_EW_QGELU 3     /* O = A*sigmoid(1.702A) (CLIP/clip/model.py:361-363) */
#define TRIS_EW_QGELU_BWD 4 /* O = A * d/dB quickgelu(B) */
#define TRIS_EW_MUL 5       /* O = A * B */
#define TRIS_EW_SCALE 6     /* O = s * A */
#define TRIS_EW_RELU 7      /* O = max(A, 0) */
int tris_elementwise_f32(int op, const float* A, const float* B, float* O, long n, float s, void* stream);
/* the same with B broadcast: B has nb elements and is read modulo nb (A, O: [n / nb, nb]; n % nb == 0, nb % 4 == 0).
 * model_stage1.py:74: the sentence features of the step against every image's attended ones (0.1 * new_lan + norm_lan) */
int tris_elementwise_bcast_f32(int op, const float* A, const float* B, float* O, long n, long nb, float s, void* stream);
/* O = X * exp(ls[0]), e_out[0] = exp(ls[0])  (model_stage1.py:77-78: score * logit_scale.exp(), ls a device scalar).
 * bwd: dX = dO * exp(ls[0]) (dX may be NULL), dls[0] = sum(dO * O) (written; deterministic two-stage sum);
 * workspace: tris_scale_exp_workspace_bytes() */
int tris_scale_exp_fwd_f32(const float* X, const float* ls, float* O, float* e_out, long n, void* stream);
long tris_scale_exp_workspace_bytes(void);
int tris_scale_exp_bwd_f32(const float* dO, const float* O, const float* ls, float* dX, float* dls, float* workspace, long n,
                           void* stream);
int tris_nchw_to_nhwc_f32(const float* X, float* Y, int B, int C, int H, int W, void* stream);

/* ---- text / ViT transformer pieces --------------------------------------------------------------------------------- */
/* nn.MultiheadAttention core on packed QKV [N, L, 3W] -> [N, L, W]; head dim 64, L <= 64; causal = the additive
 * upper-triangular -inf mask of CLIP/clip/model.py:537-543.  (model.py:380-382) */
int tris_mha_fwd_f32(const float* qkv, float* out, int N, int L, int W, int heads, int causal, void* stream);
int tris_mha_bwd_f32(const float* qkv, const float* dout, float* dqkv, int N, int L, int W, int heads, int causal,
                     void* stream);
/* The same attention for ANY sequence length, flash-style on the f32 MFMA (no L x L matrix in memory): used for L > 64
 * (a ViT-B/16 trunk at 320 px has L = 401) and, where faster, for the short sequences too.  lse [N, heads, L] receives
 * the per-query log-sum-exp (saved for backward); delta [N, heads, L] is backward scratch (rowsum(dO * O)). */
int tris_mha_mfma_fwd_f32(const float* qkv, float* out, float* lse, int N, int L, int W, int heads, int causal,
                          void* stream);
int tris_mha_mfma_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* delta,
                          float* dqkv, int N, int L, int W, int heads, int causal, void* stream);
/* The same flash-style attention in the h2 arithmetic on the 16-bit MFMA (csrc/attn_h2.hip): every product on two fp16 pieces per
 * operand (three v_mfma_f32_16x16x32_f16), one power-of-two scale per tensor from the amax words of the packed qkv and of dout
 * (2048 unsigned each, as tris_h2_next), 2^13 for the probabilities, the wave's block maximum for dS.  Same arguments, saved
 * log-sum-exp, delta and amax by-products (tris_amax_next) as the f32-MFMA entry points above. */
int tris_mha_h2_fwd_f32(const float* qkv, float* out, float* lse, const unsigned* amax_qkv, int N, int L, int W, int heads,
                        int causal, void* stream);
int tris_mha_h2_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv,
                        const unsigned* amax_qkv, const unsigned* amax_dout, int N, int L, int W, int heads, int causal,
                        void* stream);
/* Packed text rows (round 6).  A causal text tower is read at the EOT row of every sentence (reference CLIP/clip/model.py:552-564: the row at
 * argmax(ids)); under the causal mask (:537-543) no position behind it can reach that row, so the packed pass keeps rows 0 .. eot_n of
 * sentence n back to back and drops the rest (42 % of a RefCOCOg-shaped batch).  Extents live in DEVICE memory -- a captured pass is
 * valid for any ids.  Every packed buffer has R = N L rounded up to 256 rows.  plan (3 + N + R ints) <- tris_text_pack_plan_i64: [0] rows
 * in use P, [1] P rounded up to 256 = the row limit (<= R), [2 + n] first row of sentence n (n = 0 .. N), then the source token of
 * every packed row (-1 behind P).
 * tris_embed_packed_fwd_f32 writes rows < plan[1] of out [R, W] (zeros for P .. plan[1] - 1); tris_mha_packed_fwd_f32 is
 * tris_mha_fwd_f32 over the sentences' row ranges of the packed qkv [R, 3 W] (L <= 64; it keeps rows P .. plan[1] - 1 of out zero);
 * tris_eot_gather_packed_f32 reads each sentence's last row.  tris_rows_limit_thread(limit) makes the calling thread's following
 * tris_gemm_f32 (row-major A, one batch) and tris_layernorm_fwd_f32 launches skip rows >= *limit (a device word, normally &plan[1];
 * a multiple of 256 so that tiles are whole or absent); NULL clears it.  Forward only (the frozen aux tower). */
int tris_text_pack_plan_i64(const long* ids, int N, int L, int* plan, void* stream);
int tris_embed_packed_fwd_f32(const long* ids, const float* tok, const float* pos, const int* plan, float* out, int N, int L, int W,
                              void* stream);
int tris_mha_packed_fwd_f32(const float* qkv, float* out, const int* plan, int N, int Lmax, int W, int heads, int causal, void* stream);
int tris_eot_gather_packed_f32(const float* x, const int* plan, float* out, int N, int W, void* stream);
int tris_rows_limit_thread(const int* limit);
/* token_embedding(ids) + positional_embedding[:L]  (model.py:553-554).  bwd: dtok must be zero-filled by the caller */
int tris_embed_fwd_f32(const long* ids, const float* tok, const float* pos, float* out, int N, int L, int W,
                       void* stream);
int tris_embed_bwd_f32(const long* ids, const float* dout, float* dtok, float* dpos, int N, int L, int W, void* stream);
/* token-embedding gradient from a row list, deterministic (fixed summation order, no atomics): dtok[ids[t]] = scale * sum of
 * rows[t'] over the positions t' with ids[t'] == ids[t]; dtok zero-filled by the caller.  The list is this rank's N*L positions
 * (then tris_embed_bwd_f32 is called with dtok = NULL for the positional part) or, data-parallel, the all-gathered lists of every
 * rank with scale = 1 / world: <= world * B * L rows of W floats travel instead of the dense [vocab, W] table (101 MB at
 * 49408 x 512; reference: DistributedDataParallel all-reduces it densely, train_stage1.py:70). */
int tris_embed_rows_bwd_f32(const long* ids, const float* rows, float* dtok, int R, int W, float scale, void* stream);
/* out = [a; b]: int64 rows of two lists in one (positive + negative queries of a step, train_stage1.py:342-347: batched here) */
int tris_concat_i64(const long* a, long na, const long* b, long nb, long* out, void* stream);
/* x[arange(N), ids.argmax(-1)]  (model.py:562); ids NULL: x[:, 0] -- the ViT class token (model.py:443) */
int tris_eot_gather_fwd_f32(const long* ids, const float* x, float* out, int N, int L, int W, void* stream);
int tris_eot_gather_bwd_f32(const long* ids, const float* dout, float* dx, int N, int L, int W, void* stream);

/* ---- Stage-1 heads ------------------------------------------------------------------------------------------------- */
/* x / x.norm(dim=-1)  (model_stage1.py:68-69; train_stage1.py:268-269) */
int tris_l2norm_fwd_f32(const float* X, float* Y, float* inv_norm, long rows, int C, void* stream);
int tris_l2norm_bwd_f32(const float* dY, const float* Y, const float* inv_norm, float* dX, long rows, int C,
                        void* stream);
/* softmax(scale * x) over rows of length n  (model/attn.py:119,122) */
int tris_softmax_fwd_f32(const float* X, float* Y, long rows, int n, float scale, void* stream);
int tris_softmax_bwd_f32(const float* dY, const float* Y, float* dX, long rows, int n, float scale, void* stream);
/* Fused image<->text cross attention of bilateral_prompt (model/attn.py:117-128), all images in two (x3 arithmetic) or
 * three (f32 arithmetic) launches.
 * Qv,Kv,Vv [B,P,C] (pixels, channels-last), Qt,Kt,Vt [N,C] (sentences, one set shared by every image -- the reference
 * repeats it, model_stage1.py:66), scale = 1/sqrt(C); N <= 64, C % 64 == 0.
 *   new_vis[b] = softmax_n(Qv[b].Kt^T*scale).Vt      [B,P,C]
 *   new_lan[b] = softmax_p(Qt.Kv[b]^T*scale).Vv[b]   [B,N,C]
 * probs [B,4,P,N] (scratch + saved for backward): plane 0 = Av, plane 1 = Kv.Qt^T logits, plane 2 = AtT (pixel-major),
 * plane 3 = Qv.Kt^T logits (x3 path only). */
int tris_xattn_fwd_f32(const float* Qv, const float* Kv, const float* Vv, const float* Qt, const float* Kt,
                       const float* Vt, float* new_vis, float* new_lan, float* probs, int B, int P, int N, int C,
                       void* stream);
/* The same forward as ONE persistent launch (csrc/xattn_fused.hip; split-bf16 arithmetic, C = 512 | 1024, 8 <= P <= 104,
 * N <= 64): eight workgroups per image, the pixel-softmax coupling resolved by an in-kernel exchange of the logit blocks
 * (write-through stores + per-(image, slot) flags, bounded spins; all B * 8 workgroups must be co-resident, otherwise the call
 * declines).  The sentence operands are pre-split once per call into bf16 planes in MFMA
 * fragment order by a small preparation launch on the same stream.  probs: planes 0 (Av) and 2 (AtT) are written, 1 and 3 are
 * not touched.  ws: scratch of tris_xattn_fused_ws_bytes(B, N, C) bytes (0 = shape not supported); sync: caller-owned device
 * words (tris_xattn_fused_sync_words(B) of them), ZEROED once at allocation and then only passed back -- word 0 counts the
 * completed launches (the epoch that tags the exchange flags lives on the device, so a captured launch replays correctly),
 * word 2 != 0 after a launch means a wait timed out (outputs undefined).  One sync buffer per stream that may run the kernel.
 * Returns TRIS_DECLINED (-2) without launching anything when the shape / arithmetic mode is outside its domain. */
#define TRIS_DECLINED (-2)
long tris_xattn_fused_ws_bytes(int B, int N, int C);
long tris_xattn_fused_sync_words(int B);
int tris_xattn_fused_fwd_f32(const float* Qv, const float* Kv, const float* Vv, const float* Qt, const float* Kt,
                             const float* Vt, float* new_vis, float* new_lan, float* probs, int B, int P, int N, int C,
                             float* ws, long ws_bytes, unsigned* sync, void* stream);
/* The same forward as ONE persistent launch cut by PIXEL ROWS (csrc/xattn_px.hip; split-bf16 arithmetic, C = 512 | 1024,
 * P <= 104, N <= 64): S workgroups of 512 threads per image (S = min(8, CUs / B), one workgroup per CU), workgroup s owns pixels
 * [s P / S, (s + 1) P / S).  The pixel -> sentence direction (model/attn.py:118-119, 124) never leaves the workgroup; the
 * sentence -> pixel direction (attn.py:121-122, 127) needs ONE in-kernel hand-off of N x own-pixels logits, hidden behind the
 * former, and is then finished per 32-channel unit.  Same arguments, scratch / sync conventions, probs planes (0 and 2) and
 * TRIS_DECLINED behaviour as tris_xattn_fused_fwd_f32; the sync words may be the same buffer. */
long tris_xattn_px_ws_bytes(int B, int N, int C);
long tris_xattn_px_sync_words(int B);
/* h2 arithmetic for ONE call of the pixel-row launch: arms the calling thread with the amax words (2048 unsigned each, as
 * tris_h2_next) of Qv, Kv, Vv, Qt, Kt, Vt; the next tris_xattn_px_fwd_f32 of the thread computes its four products on two fp16
 * pieces per operand (three MFMAs per product, two sentence planes instead of three) with one power-of-two scale per tensor, the
 * probabilities with the fixed scale 2^13, and disarms.  Unarmed calls run the split-bf16 form. */
/* Backward of the pair on the saved probabilities, cut by pixel rows like the forward (csrc/xattn_px.hip): ONE persistent launch
 * (+ one preparation launch that splits the sentence-side operands into piece planes) produces dQv, dKv, dVv [B, P, C] and leaves
 * dS [3][B, P, N] = (dS1 = soft-max backward of the pixel -> sentence direction, dS2 = of the sentence -> pixel direction, a copy of
 * Av) for the three [N, C] gradients that sum over images and pixels (dVt = Av^T d_vis, dKt = dS1^T Qv, dQt = dS2^T Kv: split-K
 * products of the GEMM core).  probs is the forward's [B][4][P][N] buffer (planes 0 and 2).  Only the column sums of the
 * sentence -> pixel soft-max cross workgroups (N floats each, the forward's sync words and protocol).  Domain, TRIS_DECLINED and
 * the h2 arming (tris_xattn_amax_next with the words of d_vis, Vv, d_lan, Qt, Kt, Vt -- in that order) as for the forward.
 * Replaces the chain of six batched products and two soft-max backward launches of rounds 1-4 (reference model/attn.py:117-128). */
long tris_xattn_px_bwd_ws_bytes(int B, int N, int C);
int tris_xattn_px_bwd_f32(const float* d_vis, const float* d_lan, const float* Vv, const float* Qt, const float* Kt,
                          const float* Vt, const float* probs, float* dQv, float* dKv, float* dVv, float* dS, int B, int P,
                          int N, int C, float* ws, long ws_bytes, unsigned* sync, unsigned* amax_dQv, unsigned* amax_dKv,
                          unsigned* amax_dVv, void* stream);   /* amax_*: NULL or zeroed amax words the launch raises for its outputs */
long tris_xattn_px_last_form(void);   /* arithmetic of the last pixel-row launch: 0 none yet, 1 split-bf16, 2 h2 */
int tris_xattn_amax_next(const unsigned* qv, const unsigned* kv, const unsigned* vv, const unsigned* qt, const unsigned* kt,
                         const unsigned* vt);
/* host-side plan of the pixel-row launch: workgroups per image S on a device of `cus` compute units (0 = the call would decline:
 * B * S workgroups must be resident one per CU, a workgroup owns <= 32 pixels and <= 8 of the C / 32 channel units).  Workgroup s
 * owns pixels [s P / S, (s + 1) P / S) and units [s U / S, (s + 1) U / S), U = C / 32 (integer division). */
long tris_xattn_px_slots(int B, int P, int C, int cus);
int tris_xattn_px_fwd_f32(const float* Qv, const float* Kv, const float* Vv, const float* Qt, const float* Kt,
                          const float* Vt, float* new_vis, float* new_lan, float* probs, int B, int P, int N, int C,
                          float* ws, long ws_bytes, unsigned* sync, void* stream);
/* backward of a softmax taken over the P axis of [B,P,N]: dX = scale*Y*(dY - sum_p Y*dY)  (model/attn.py:122) */
int tris_softmax_col_bwd_f32(const float* dY, const float* Y, float* dX, int B, int P, int N, float scale,
                             void* stream);
/* training cls head on score [B,P,N]: bg channel + channel softmax + mean/max pooling + focal term
 * (model_stage1.py:80-108, focal_loss :122-123).  cls_fg may be NULL. */
int tris_cls_head_fwd_f32(const float* score, float* cls_out, float* cls_fg, int B, int P, int N, float focal_p,
                          float focal_c, void* stream);
int tris_cls_head_bwd_f32(const float* score, const float* g_cls_out, float* dscore, int B, int P, int N, float focal_p,
                          float focal_c, void* stream);
/* diagonal response maps score[i,:,i] -> bilinear (align_corners=False) to SxS -> relu / sigmoid
 * (model_stage1.py:110-119, model/utils.py:5-10).  sig_map may be NULL (eval).  bwd ACCUMULATES into dscore. */
int tris_maps_fwd_f32(const float* score, float* relu_map, float* sig_map, int B, int h, int w, int N, int S,
                      void* stream);
int tris_maps_bwd_f32(const float* score, const float* d_relu, const float* d_sig, float* dscore, int B, int h, int w,
                      int N, int S, void* stream);
/* F.interpolate(mode='bilinear') on [planes, Hi, Wi] (train_stage1.py:328-329, validate.py:180) */
int tris_resize_bilinear_fwd_f32(const float* X, float* Y, int planes, int Hi, int Wi, int Ho, int Wo,
                                 int align_corners, void* stream);
int tris_resize_bilinear_bwd_f32(const float* dY, float* dX, int planes, int Hi, int Wi, int Ho, int Wo,
                                 int align_corners, void* stream);
/* fg = cam * img written directly as the aux ViT's patch-GEMM operand [B, (R/ps)^2, C*ps*ps]
 * (train_stage1.py:333-338 + CLIP/clip/model.py:405,420-422) */
int tris_fg_patch_fwd_f32(const float* cam, const float* img, float* patches, int B, int C, int R, int ps,
                          void* stream);
int tris_fg_patch_bwd_f32(const float* dpatches, const float* img, float* dcam, int B, int C, int R, int ps,
                          void* stream);
/* class token + positional embedding (CLIP/clip/model.py:423-424) */
int tris_vit_assemble_fwd_f32(const float* emb, const float* cls, const float* pos, float* x, int B, int T, int W,
                              void* stream);
int tris_vit_assemble_bwd_f32(const float* dx, float* demb, int B, int T, int W, void* stream);
/* The whole loss block: fg_loss = MaxLoss(clip_forward) (train_stage1.py:263-284,340), cbs_loss (:342-353),
 * cls_loss = multilabel_soft_margin_loss(cls, eye) (:354), loss = w1*l1 + w4*l4 + w5*l5 (:364).
 * per_img: scratch [B,3]; rowloss: scratch [B]; losses[4] = {total, l1, l4, l5}.  fneg may be NULL when K == 0.
 * bwd: g = device gradient of losses[4]; the kernels form dL/dl1 = g[0]*w1 + g[1], dL/dl4 = g[0]*w4 + g[2], dL/dl5 = g[0]*w5 + g[3]. */
int tris_stage1_loss_fwd_f32(const float* cls, const float* fi, const float* ft, const float* fneg, int B, int N, int E,
                             int K, float w1, float w4, float w5, float* per_img, float* rowloss, float* losses,
                             void* stream);
int tris_stage1_loss_bwd_f32(const float* cls, const float* fi, const float* ft, const float* fneg, const float* g,
                             float w1, float w4, float w5, int B, int N, int E, int K, float* dcls, float* dfi, void* stream);

/* ---- optimiser ----------------------------------------------------------------------------------------------------- */
/* torch.optim.AdamW step over a flat arena (train_stage1.py:135-139, 370); step_count is the 1-based t */
int tris_adamw_f32(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step_count, void* stream);
/* the same update with the step-dependent scalars in DEVICE memory: hyper = {lr, 1 - beta1^t, sqrt(1 - beta2^t)} (written by
   the host before the launch) -- the form a captured training step replays (tris_amd/graphs.py GraphedTrainStep) */
int tris_adamw_dev_f32(float* p, const float* g, float* m, float* v, long n, const float* hyper, float beta1, float beta2,
                       float eps, float weight_decay, void* stream);

/* ---- evaluation post-processing (validate.py:180-190, utils/util.py:9-15) ------------------------------------------- */
/* relu_map [S,S] of ONE (image, sentence) -> cam [oH,oW] = bilinear(align_corners=True)/(max+1e-5); mask = cam > 1e-9;
 * out_iu (int64[3], device) = {I, U, argmax index of cam}; target: uint8 [oH,oW].  cam may be NULL. */
int tris_eval_post_f32(const float* relu_map, int S, const unsigned char* target, int oH, int oW, float* cam,
                       long* out_iu, float* workspace /* >= 2*1024+8 floats */, void* stream);

/* ---- input pipeline (dataset/transform.py:23-63, dataset/ReferDataset.py:125-252) ---------------------
 * The decoded dataset is kept in HBM as uint8; everything below is bit-exact against Pillow / torchvision semantics. */
/* Pillow's antialiased two-pass resampling of an 8-bit image [Hin][Win][C] -> [Hout][Wout][C] (F.resize on a PIL image,
 * dataset/transform.py:29).  bounds_* int32[out][2] = {first source index, tap count}, kk_* int32[out][ksize] taps in
 * 22-bit fixed point (built on the host exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do); a NULL
 * bounds pointer skips that pass (size unchanged along it).  tmp: Hin*Wout*C bytes when both passes run. */
int tris_resample_u8(const unsigned char* in, int Hin, int Win, int C, const int* bounds_h, const int* kk_h, int ksize_h,
                     const int* bounds_v, const int* kk_v, int ksize_v, int Hout, int Wout, unsigned char* tmp,
                     unsigned char* out, void* stream);
/* Pillow NEAREST resize (dataset/transform.py:32, ReferDataset.py:187): out[y][x] = in[yidx[y]][xidx[x]] (index < 0 -> 0) */
int tris_gather2d_u8(const unsigned char* in, int Hin, int Win, int C, const int* yidx, const int* xidx, int Hout,
                     int Wout, unsigned char* out, void* stream);
/* batch assembly: out[b] = lut[c][cache[index[b]][p][c]]  (to_tensor + normalize, dataset/transform.py:41-53, with the
 * 3x256 float table computed by the host with the reference's float ops).  cache: uint8 [N][HW][3]; index: int64[B];
 * out: fp32 [B][HW][3] (planar=0, channels-last) or [B][3][HW] (planar=1, the reference's NCHW).  HW % 4 == 0. */
int tris_u8_gather_normalize_f32(const unsigned char* cache, const long* index, int B, long HW, const float* lut,
                                 float* out, int planar, void* stream);
/* out[r] = table[index[r]]  for rows of row_bytes (multiple of 4) bytes: token ids of the sampled sentences
 * (ReferDataset.py:172-229) */
int tris_gather_rows(const void* table, const long* index, long rows, long row_bytes, void* out, void* stream);

/* ---- SyncBatchNorm statistics exchange over xGMI peer memory (csrc/comm.hip) --------------------------------------
 * Replaces the per-layer collectives of nn.SyncBatchNorm (reference: train_stage1.py:69 convert_sync_batchnorm; torch's
 * SyncBatchNorm all_gathers [mean|invstd|count] in forward and all_reduces [sum_dy|sum_dy_xmu] in backward) for ranks on
 * ONE node.  Every rank owns a mailbox in uncached device memory (tris_mbox_alloc; capacity `cap_floats` per sender block,
 * a multiple of 4) and maps its peers' mailboxes through HIP IPC (tris_mbox_ipc_handle -> 64-byte handle, exchanged by the
 * host; tris_mbox_ipc_open).  tris_mbox_exchange_f32 is one single-workgroup launch on `stream`: store the block
 * [src0[n0] | src1[n1]] (n = n0 + n1; src1 may be NULL with n1 = 0) into every
 * peer's mailbox, publish per-sender flags (system-scope release), wait for the `world` flags of the own mailbox (bounded
 * spin: after spin_limit polls *err is set to seq, the wait is abandoned and `out` is filled with NaN instead of stale slot
 * contents -- tris_mbox_bn_combine_f32 likewise writes NaN statistics and leaves the running statistics alone), then mode 0:
 * out[world][n] = the gathered
 * blocks in rank order; mode 1: out[n] = their sum in rank order (bit-identical on every rank).  `boxes` is a DEVICE array of
 * the `world` mailbox pointers as mapped in this process (own mailbox at index `rank`).  `seq` points at ONE caller-owned DEVICE
 * word, zeroed once: the number of the exchange is read from it and advanced in it by the kernel itself (exchanges of a rank run
 * on one stream; every rank issues the same sequence, so the counters agree) -- nothing about an exchange depends on host state,
 * the launches can be captured into a hipGraph and replayed.  World size <= TRIS_MBOX_MAX_WORLD. */
#define TRIS_MBOX_MAX_WORLD 16
long tris_mbox_bytes(int cap_floats);
int tris_mbox_alloc(void** ptr, int cap_floats);
int tris_mbox_free(void* ptr);
int tris_mbox_ipc_handle(void* ptr, void* handle64);
int tris_mbox_ipc_open(const void* handle64, void** ptr);
int tris_mbox_ipc_close(void* ptr);
int tris_mbox_exchange_f32(const float* src0, int n0, const float* src1, int n1, float* out, void* const* boxes, int world,
                           int rank, unsigned* seq, int cap_floats, int mode, long spin_limit, int* err, void* stream);
/* SyncBatchNorm forward in ONE launch: exchange the [mean | invstd | biased var] block (3 C floats, what tris_bn_finalize_f32 /
 * tris_bn_stats_f32 write) and combine the `world` blocks into the global statistics stats[3C] + running statistics -- the
 * arithmetic of tris_bn_sync_combine_f32.  Every rank contributes count_per_rank rows: the SAME number on every rank (the global
 * count is count_per_rank * world, as with torch's DistributedSampler, which pads the shards to equal length; ranks with
 * unequal per-step batches are not representable here -- nor in tris_bn_sync_combine_f32 -- and must use equal shards). */
int tris_mbox_bn_combine_f32(const float* local_stats, int C, long count_per_rank, float eps, float momentum, float* stats,
                             float* running_mean, float* running_var, void* const* boxes, int world, int rank, unsigned* seq,
                             int cap_floats, long spin_limit, int* err, void* stream);
/* The same two exchanges with the amax WORD of the plane tensor the following pass writes as a by-product (h2 with operand
 * planes, csrc/planes.h) -- what tris_bn_out_bound2_f32 / tris_bn_bwd_bound_f32 compute in launches of their own:
 * forward bound = max_c |gamma_c| xhat_max + |beta_c| (+ the residual's bound, resid_word or NULL); backward: out[2C] = the sums
 * [sum dz | sum dz xhat] over all ranks and bound = max_c |gamma_c invstd_c| (amax dz + |sum dz| inv_count + xhat_max |sum dz xhat|
 * inv_count), amax dz from dz_word.  Two launches less per SyncBatchNorm layer and pass. */
int tris_mbox_bn_combine_bound_f32(const float* local_stats, int C, long count_per_rank, float eps, float momentum, float* stats,
                                   float* running_mean, float* running_var, void* const* boxes, int world, int rank, unsigned* seq,
                                   int cap_floats, long spin_limit, int* err, const float* gamma, const float* beta, float xhat_max,
                                   const unsigned* resid_word, unsigned* bound_word, void* stream);
int tris_mbox_bn_bwd_exchange_f32(const float* sum_dz, const float* sum_dzx, int C, float* out, void* const* boxes, int world, int rank,
                                  unsigned* seq, int cap_floats, long spin_limit, int* err, const float* gamma, const float* invstd,
                                  float inv_count, float xhat_max, const unsigned* dz_word, unsigned* bound_word, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TRIS_HIP_H */
