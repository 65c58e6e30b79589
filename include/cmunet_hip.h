/*
 * cmunet_hip.h -- C-ABI of the MI355X (gfx950) CM-UNet hot-path library (libcmunet_hip.so).
 *
 * The reference (CamilleChallier/Contrastive-Masked-UNet) is 100 % Python: the arithmetic of its hot
 * path is dispatched by torch.nn modules into ATen/cuDNN.  There is no FFI in the reference; the
 * entry points below are what a binding for this path replaces, and each one cites the reference
 * call site whose ATen dispatch it stands for (paths relative to /root/reference).
 *
 * Conventions
 *   - plain C: device pointers, sizes, a dtype enum and a hipStream_t (passed as void*); no torch types.
 *   - every function returns 0 on success or a negative CMU_ERR_* code; cmu_last_error() returns a
 *     thread-local message.  Nothing is allocated or freed inside the library; workspaces are passed in
 *     and sized with the cmu_*_ws_bytes() queries.  No global mutable state; re-entrant.
 *   - activations are NHWC with an explicit pixel stride `ld` (elements), so a tensor may be a channel
 *     slice of a wider buffer (the concat-free decoder: UpBlock's torch.cat, Finetuning/model.py:80,
 *     never materialises).  dtype of activations/packed weights = `dt`; parameters, statistics,
 *     gradients of parameters and logits are fp32.
 *   - a "pending transform" (in_scale, in_shift, relu_from) is a training-mode BatchNorm+ReLU that the
 *     producer left un-applied: the consumer applies  v = x*scale[c]+shift[c]; if (c>=relu_from) v=max(v,0)
 *     (relu_from = -n < 0, the 3x3 conv / weight-gradient entries: the channels c < n are the activated ones instead --
 *     a concat view whose activated skip comes FIRST, cmu_conv1x1_nchw_fwd excepted)
 *     while staging its input tile (BatchNorm2d + ReLU of Finetuning/model.py:18-19,21-22 fused into the
 *     next conv / pool / convT load).  in_scale == NULL means identity.
 */
#ifndef CMUNET_HIP_H
#define CMUNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { CMU_F32 = 0, CMU_F16 = 1, CMU_BF16 = 2 } cmu_dtype;

#define CMU_OK 0
#define CMU_ERR_ARG (-1)      /* bad shape / alignment / null pointer */
#define CMU_ERR_UNSUPPORTED (-2)
#define CMU_ERR_WORKSPACE (-3)
#define CMU_ERR_LAUNCH (-4)   /* hipGetLastError() != hipSuccess after a launch */

const char* cmu_last_error(void);
/* name of the compute kernel the calling thread's last GEMM-shaped entry launched ("" if none): for profilers */
const char* cmu_last_kernel(void);
int cmu_version(void);
/* TEST HOOK -- the one piece of process-global mutable state in this library (every other entry is re-entrant and keeps no state
 * between calls: SURVEY section 8b).  Forces one of the dispatch switches that A/B two kernel forms -- "CMU_CONV_NARROW", "CMU_CONV_SLIM",
 * "CMU_CONV_PERSIST_PART", "CMU_WGRAD_SQUARE", "CMU_WGRAD_WIDE_F32" (bit-identical results either way), "CMU_CONV_V5" (the 16x16x32 conv
 * kernel of round 5 against the 32x32x16 family: equal to rounding) -- to 0 / 1, or back to the environment variable of the same name
 * (value -1; the environment is read once, never on the launch path).  The override applies to EVERY thread of the process from the
 * moment it is set: set it while no other thread is launching (the tests are single-threaded); stores and loads of the switch are atomic,
 * calls are serialised by a mutex, a launch in flight on another thread may see either value.  Product code never calls it.
 * Unknown name: CMU_ERR_ARG. */
int cmu_set_dispatch_override(const char* name, int value);
/* element size in bytes of a cmu_dtype */
int cmu_dtype_size(int dt);

/* ---------------------------------------------------------------------------------------------
 * Weight packing (parameters stay in the reference's layouts; packed copies are per-step scratch)
 * ------------------------------------------------------------------------------------------- */
/* nn.Conv2d weight (Cout,Cin,3,3) fp32 -> [chunk][tap][CoutPad][KC] dt for cmu_conv3x3_fwd
 * (Finetuning/model.py:17,20).  transpose_flip=1 packs the data-gradient form
 * w'[tap'][c][n] = w[n][c][2-kh][2-kw] so that cmu_conv3x3_fwd computes dX from dY.            */
int64_t cmu_pack_conv3x3_elems(int Cin, int Cout, int dt, int transpose_flip);
int cmu_pack_conv3x3(const float* w, void* out, int Cin, int Cout, int dt, int transpose_flip, void* stream);
/* nn.ConvTranspose2d weight (Cin,Cout,2,2) fp32 (Finetuning/model.py:60).
 * mode 0: forward form  [chunk(Cin)][1][(ij*Cout+co)Pad][KC]
 * mode 1: data-grad form [chunk(4*Cout: ij*Cout+co)][1][CinPad][KC]                              */
int64_t cmu_pack_convT2x2_elems(int Cin, int Cout, int dt, int mode);
int cmu_pack_convT2x2(const float* w, void* out, int Cin, int Cout, int dt, int mode, void* stream);
/* Every pack of a training step in one launch.  descs_dev: device array of ndesc records of cmu_pack_desc_bytes() bytes
 * { const float* w; void* out; int32 Cin, Cout, mode, kind (0 conv3x3: mode = transpose_flip, 1 convT2x2); int64 total
 * (elements of the packed array); int64 block0 (first of its cmu_pack_desc_blocks(...) workgroups) }, block0 ascending from 0;
 * total_blocks = sum of the workgroup counts.  Same layouts as cmu_pack_conv3x3 / cmu_pack_convT2x2.               */
int cmu_pack_desc_bytes(void);
int64_t cmu_pack_desc_blocks(int kind, int Cin, int Cout, int dt, int mode);
int cmu_pack_batch(const void* descs_dev, int ndesc, int64_t total_blocks, int dt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Forward
 * ------------------------------------------------------------------------------------------- */
/* (B,H,W) fp32 image -> NHWC 1-channel is the same memory; this entry is the first conv of the
 * network (Cin == 1): Conv2d(1,C,3,p=1) of model.py:96 (down_conv1) computed directly, with the
 * optional patch-mask multiply x*(1-mask) of UNet_encoder.py:156 / spark.py:93-94 fused into the load.
 * mask: uint8 (mask_per_sample ? B : 1, H, W) or NULL.  No bias (BatchNorm follows; see cmu_bn_finalize).
 * stats: [cmu_conv_ntiles][2][Cout] fp32 partial sums (sum, sum of squares) or NULL.               */
int cmu_conv3x3_c1_fwd(const float* x, const uint8_t* mask, int mask_per_sample, const float* w /*(Cout,1,3,3)*/,
                       void* y, int64_t ldy, float* stats, int B, int H, int W, int Cout, int dt, void* stream);

/* Conv2d(Cin,Cout,3,padding=1) as implicit GEMM on MFMA (model.py:17,20), bias-free, with the producer's
 * pending BN+ReLU applied on load and per-tile channel statistics of the raw output in the epilogue. */
int cmu_conv3x3_fwd(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from,
                    const void* wpacked, void* y, int64_t ldy, float* stats,
                    int B, int H, int W, int Cin, int Cout, int dt, void* stream);
/* number of spatial tiles (rows of the stats slab) for a (B,H,W) problem */
int cmu_conv_ntiles(int B, int H, int W);

/* BatchNorm2d training statistics (model.py:18,21; eps 1e-5, momentum 0.1): reduces the slab written by
 * a conv into batch mean / biased var, produces the pending transform scale=gamma*invstd,
 * shift=beta-mean*scale, saves mean/invstd for backward and updates running_mean (with the conv bias
 * added back) / running_var (unbiased).  training=0: scale/shift from the running statistics.
 * ws: cmu_bn_finalize_ws_bytes(C) bytes.                                                          */
int64_t cmu_bn_finalize_ws_bytes(int C);
int cmu_bn_finalize(const float* stats, int ntiles, int64_t count, const float* conv_bias, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                    int training, float* scale, float* shift, float* save_mean, float* save_invstd,
                    int C, void* ws, void* stream);

/* BN+ReLU apply + MaxPool2d(2) (model.py:40,44): raw conv output -> activated pooled tensor. */
int cmu_bnrelu_maxpool_fwd(const void* y, int64_t ldy, const float* scale, const float* shift,
                           void* out, int64_t ldo, int B, int H, int W, int C, int dt, void* stream);

/* ConvTranspose2d(Cin,Cout,2,stride=2) (model.py:60,78) as GEMM M=B*H*W, N=4*Cout, K=Cin with the
 * pending transform on load, bias added, and a pixel-shuffle store into out (B,2H,2W,ldo) -- normally the
 * left half of the decoder's concat buffer.                                                        */
int cmu_convT2x2_fwd(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from,
                     const void* wpacked, const float* bias, void* out, int64_t ldo,
                     int B, int H, int W, int Cin, int Cout, int dt, void* stream);

/* Conv2d(C,K,1) head (model.py:108,130) with pending transform on load: logits (B,K,H,W) fp32 NCHW. */
int cmu_conv1x1_head_fwd(const void* x, int64_t ldx, const float* in_scale, const float* in_shift,
                         const float* w /*(K,C)*/, const float* bias, float* logits,
                         int B, int H, int W, int C, int K, int dt, void* stream);

/* pending transform applied, NHWC dt -> NCHW fp32 (module boundary of DoubleConv/DownBlock/UpBlock). */
int cmu_apply_to_nchw(const void* y, int64_t ldy, const float* scale, const float* shift, int relu_from,
                      float* out, int B, int H, int W, int C, int dt, void* stream);
/* NCHW fp32 -> NHWC dt (module boundary, inputs with C > 1) and back for gradients. */
int cmu_nchw_to_nhwc(const float* x, void* out, int64_t ldo, int B, int H, int W, int C, int dt, void* stream);
int cmu_nhwc_to_nchw(const void* x, int64_t ldx, float* out, int B, int H, int W, int C, int dt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward
 * ------------------------------------------------------------------------------------------- */
/* BatchNorm2d+ReLU backward, phase 1: per-channel sums of dz = dA*[y*scale+shift>0] and dz*xhat.
 * Produces dgamma, dbeta (fp32, overwritten) and coef[2][C] = (sum dz / N, sum dz*xhat / N).      */
int64_t cmu_bn_bwd_ws_bytes(int C);
int cmu_bn_bwd_reduce(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                      const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef,
                      int B, int H, int W, int C, int dt, void* ws, void* stream);
/* finalisation of a slab written by a fused producer (cmu_maxpool_bwd / cmu_conv1x1_head_bwd with bn_ws):
 * dgamma, dbeta (nullable) and coef[2][C], summed in row order (bitwise reproducible).                          */
int cmu_bn_bwd_finalize(const void* bn_ws, int64_t count, float* dgamma, float* dbeta, float* coef, int C, void* stream);
/* Same phase-1 result from a per-tile slab [cmu_conv_ntiles][2][C] written by cmu_conv3x3_dgrad_bn /
 * cmu_convT2x2_dgrad_bn.  ws: cmu_bn_finalize_ws_bytes(C).                                                       */
int cmu_bn_bwd_finalize_tiles(const float* bstats, int ntiles, int64_t count, float* dgamma, float* dbeta, float* coef, int C,
                              void* ws, void* stream);
/* Data gradient of Conv2d 3x3 (autograd of model.py:17,20): cmu_conv3x3_fwd on dY with the flipped pack, plus -- in the
 * epilogue, from the stored dX and the raw output ``yraw`` of the conv+BN+ReLU layer that produced the conv's input --
 * that layer's BatchNorm+ReLU backward partial sums (replaces a separate cmu_bn_bwd_reduce pass over dX and yraw).
 * K = channels of dY, N = channels of dX.                                                                        */
int cmu_conv3x3_dgrad_bn(const void* dY, int64_t ldd, const void* wpacked_flip, void* dX, int64_t ldx, const void* yraw, int64_t ldy,
                         const float* scale, const float* shift, const float* save_mean, const float* save_invstd, float* bstats,
                         int B, int H, int W, int K, int N, int dt, void* stream);
/* phase 2: dY = scale * (dz - coef0 - xhat*coef1), written to dY (may alias dA). */
int cmu_bn_bwd_apply(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                     const float* save_mean, const float* save_invstd, const float* coef, void* dY, int64_t ldo,
                     int B, int H, int W, int C, int dt, void* stream);

/* weight gradient of Conv2d 3x3 (autograd of model.py:17,20): dW (Cout,Cin,3,3) fp32, overwritten.
 * x is the layer's input with its pending transform (applied on load).  ws: cmu_conv3x3_wgrad_ws_bytes. */
int64_t cmu_conv3x3_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int dt);
int cmu_conv3x3_wgrad(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from,
                      const void* dY, int64_t ldd, float* dW, int B, int H, int W, int Cin, int Cout, int dt,
                      void* ws, void* stream);
/* first layer (Cin == 1): dW (Cout,1,3,3) from the fp32 image (+ optional mask) and dY. */
int64_t cmu_conv3x3_c1_wgrad_ws_bytes(int B, int H, int W, int Cout);
int cmu_conv3x3_c1_wgrad(const float* x, const uint8_t* mask, int mask_per_sample, const void* dY, int64_t ldd,
                         float* dW, int B, int H, int W, int Cout, int dt, void* ws, void* stream);
/* Same with the layer's BatchNorm+ReLU backward applied on the fly: dA is the gradient w.r.t. the activated output, yraw the
 * layer's raw conv output, coef the phase-1 coefficients (cmu_bn_bwd_finalize*).  The first layer has no data gradient, so
 * its dY never has to exist in memory (no cmu_bn_bwd_apply pass for it).                                               */
int cmu_conv3x3_c1_wgrad_bn(const float* x, const uint8_t* mask, int mask_per_sample, const void* dA, int64_t ldd,
                            const void* yraw, int64_t ldy, const float* scale, const float* shift, const float* save_mean,
                            const float* save_invstd, const float* coef, float* dW, int B, int H, int W, int Cout, int dt,
                            void* ws, void* stream);
/* Same, with the raw output recomputed from the image and the layer's forward weights w (Cout,1,3,3) instead of read: the bits
 * cmu_conv3x3_c1_fwd stored (same FMA order, rounded through the storage type), half the HBM traffic of the pass.           */
int cmu_conv3x3_c1_wgrad_bn_w(const float* x, const uint8_t* mask, int mask_per_sample, const void* dA, int64_t ldd,
                              const float* w, const float* scale, const float* shift, const float* save_mean,
                              const float* save_invstd, const float* coef, float* dW, int B, int H, int W, int Cout, int dt,
                              void* ws, void* stream);

/* MaxPool2d(2) backward fused with the skip-branch add: dA = unpool(dP) + dSkip (dSkip may be NULL).
 * The arg-max is recomputed from the raw output + transform (first max in row-major 2x2 order, as ATen). */
/* bn_ws != NULL additionally fuses phase 1 of this layer's BatchNorm backward: the partial sums of
 * dz and dz*xhat over the dA values just written go to the slab bn_ws (cmu_bn_bwd_ws_bytes(C) bytes; a device-side
 * header word holds the row count) -- follow with cmu_bn_bwd_finalize instead of cmu_bn_bwd_reduce.          */
int cmu_maxpool_bwd(const void* dP, int64_t ldp, const void* dSkip, int64_t lds, const void* y, int64_t ldy,
                    const float* scale, const float* shift, void* dA, int64_t lda,
                    const float* save_mean, const float* save_invstd, void* bn_ws,
                    int B, int H, int W, int C, int dt, void* stream);
/* cmu_maxpool_bwd2: the same with a second skip gradient dSkip2 (or NULL) added in fp32 inside the pass -- the skips of CM_UNet's online
 * encoder feed the pixel AND the feature decoder (cmunet.py:121-124 of the reference: autograd sums the two gradients).               */
int cmu_maxpool_bwd2(const void* dP, int64_t ldp, const void* dSkip, int64_t lds, const void* dSkip2, int64_t lds2, const void* y,
                     int64_t ldy, const float* scale, const float* shift, void* dA, int64_t lda,
                     const float* save_mean, const float* save_invstd, void* bn_ws,
                     int B, int H, int W, int C, int dt, void* stream);
/* Round 3 -- the pooled gradient never stored.  cmu_maxpool_bwd2 with dA == NULL (bn_ws required) leaves only the BatchNorm-backward
 * sums; after cmu_bn_bwd_finalize, cmu_maxpool_bwd_apply recomputes dA = unpool(dP) + dSkip (+ dSkip2), rounds it to the storage type
 * as the stored form would have been, and writes dY = scale * (gate * dA - coef[0] - xhat * coef[1]) -- bit-identical to
 * cmu_maxpool_bwd2 + cmu_bn_bwd_apply (the BatchNorm2d + ReLU + MaxPool2d backward of model.py:20-25,42-45), one tensor pass less. */
int cmu_maxpool_bwd_apply(const void* dP, int64_t ldp, const void* dSkip, int64_t lds, const void* dSkip2, int64_t lds2, const void* y,
                          int64_t ldy, const float* scale, const float* shift, const float* save_mean, const float* save_invstd,
                          const float* coef, void* dY, int64_t ldo, int B, int H, int W, int C, int dt, void* stream);

/* SparK's sparse encoder (Pretraining/Spark/encoder.py:20-36 + models/custom.py:152-182: conv * mask -> sparse BN -> ReLU -> MaxPool2d):
 * mask-aware pools, so that the activated + masked copy of a level's second conv output is never materialised.  A pool window lies in
 * one patch (patch side >= 2 px at every pooled level): cmu_bnrelu_maxpool_fwd_masked writes zero for masked windows without reading
 * y, cmu_maxpool_bwd_masked leaves their dA unwritten (the masked BatchNorm-backward passes that follow never read masked positions and
 * write zeros there).  active: (B,f,f) uint8; H = W = f << s, s >= 1.                                                          */
int cmu_bnrelu_maxpool_fwd_masked(const void* y, int64_t ldy, const float* scale, const float* shift, const uint8_t* active, int f,
                                  void* out, int64_t ldo, int B, int H, int W, int C, int dt, void* stream);
int cmu_maxpool_bwd_masked(const void* dP, int64_t ldp, const void* dSkip, int64_t lds, const void* y, int64_t ldy, const float* scale,
                           const float* shift, const uint8_t* active, int f, void* dA, int64_t lda, int B, int H, int W, int C, int dt,
                           void* stream);

/* ConvTranspose2d 2x2 s2 backward.  data: dX (B,H,W,Cin) from dOut (B,2H,2W,ldd) (GEMM K = 4*Cout).
 * weight: dW (Cin,Cout,2,2) and dbias (Cout) fp32, overwritten.                                    */
int cmu_convT2x2_dgrad(const void* dOut, int64_t ldd, const void* wpacked_dgrad, void* dX, int64_t ldx,
                       int B, int H, int W, int Cin, int Cout, int dt, void* stream);
/* cmu_convT2x2_dgrad + the BatchNorm+ReLU backward partial sums of the layer that produced the ConvTranspose's input
 * (see cmu_conv3x3_dgrad_bn).                                                                                     */
int cmu_convT2x2_dgrad_bn(const void* dOut, int64_t ldd, const void* wpacked_dgrad, void* dX, int64_t ldx, const void* yraw,
                          int64_t ldy, const float* scale, const float* shift, const float* save_mean, const float* save_invstd,
                          float* bstats, int B, int H, int W, int Cin, int Cout, int dt, void* stream);
int64_t cmu_convT2x2_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int dt);
int cmu_convT2x2_wgrad(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from,
                       const void* dOut, int64_t ldd, float* dW, float* dbias,
                       int B, int H, int W, int Cin, int Cout, int dt, void* ws, void* stream);

/* 1x1 head backward: dX (NHWC dt) = dLogits^T W ; dW (K,C), dbias (K) fp32 overwritten. */
int64_t cmu_conv1x1_head_bwd_ws_bytes(int B, int H, int W, int C, int K);
int cmu_conv1x1_head_bwd(const float* dlogits, const void* x, int64_t ldx, const float* in_scale, const float* in_shift,
                         const float* w, void* dX, int64_t ldo, float* dW, float* dbias,
                         const float* save_mean, const float* save_invstd, void* bn_ws /* optional, as cmu_maxpool_bwd */,
                         int B, int H, int W, int C, int K, int dt, void* ws, void* stream);
/* Second half of the fused form: cmu_conv1x1_head_bwd with dX = NULL and bn_ws set leaves only the parameter gradients and the
 * BatchNorm-backward partial sums of the layer that fed the head (the rank-K input gradient is never stored); after
 * cmu_bn_bwd_finalize this pass recomputes it per pixel from dlogits and writes that layer's dY = scale * (gate * dA - coef[0] -
 * xhat * coef[1]) -- the same bits as cmu_conv1x1_head_bwd (dX stored) + cmu_bn_bwd_apply, 3.2 instead of 5.4 GB at the bench size. */
int cmu_conv1x1_head_bn_apply(const float* dlogits, const void* x, int64_t ldx, const float* in_scale, const float* in_shift,
                              const float* w, const float* save_mean, const float* save_invstd, const float* coef, void* dY,
                              int64_t ldo, int B, int H, int W, int C, int K, int dt, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Losses / pretraining heads
 * ------------------------------------------------------------------------------------------- */
/* CM-UNet masked reconstruction (cmunet_head.py:62-70): target = per-row normalised image (unbiased var,
 * eps 1e-6), loss = sum((pred-target)^2*mask)/sum(mask), pred = logits[:,channel].  Writes loss (1 fp32)
 * and, if dlogits != NULL, d loss*loss_scale / d logits (B,K,H,W) (zero on the other channels).
 * mask uint8 (B,H,W).  ws: cmu_masked_mse_ws_bytes(B,H).                                            */
int64_t cmu_masked_mse_ws_bytes(int B, int H);
/* amp_state (nullable): a cmu_amp_* state whose current scale multiplies loss_scale in dlogits (loss stays unscaled).  */
int cmu_masked_mse_fwd_bwd(const float* logits, int K, int channel, const float* img, const uint8_t* mask,
                           float* loss, float* dlogits, float loss_scale, const void* amp_state, int B, int H, int W, void* ws,
                           void* stream);

/* Finetune criterion (train.py:455; metrics.py:135-180,503): softmax over 2 classes, CE with one-hot
 * (probability) targets averaged over B*H*W, Dice/IoU counters on softmax[:,1] > 0.5 (no gradient: A-4).
 * out[0]=ce, out[1]=dice_loss, out[2]=iou_loss, out[3..5]=tp, sum_pr, sum_gt.  y1h fp64 (B,2,H,W).
 * dlogits (nullable) = d ce / d logits * loss_scale.                                               */
int64_t cmu_softmax_ce_dice_ws_bytes(int B, int H, int W);
int cmu_softmax_ce_dice_fwd_bwd(const float* logits, const double* y1h, float* out, float* dlogits, float loss_scale,
                                int B, int H, int W, void* ws, void* stream);

/* CM-UNet in-batch InfoNCE (cmunet_head.py:72-88): pred (B,D) raw predictor output (L2-normalised inside),
 * keys (N,D) gathered, already normalised target projections; label i + B*rank; loss = ct_w*2*t*CE.
 * dpred (nullable) = d loss / d pred (B,D).  loss: 1 + B floats (loss[0] total, loss[1+b] per-row terms).     */
int cmu_infonce_inbatch_fwd_bwd(const float* pred, const float* keys, float* loss, float* dpred,
                                int B, int N, int D, int rank, float temperature, float ct_weight, void* stream);

/* MoCo-v2 InfoNCE + momentum-queue update in ONE launch (moco2_module.py:256-285, 160-175):
 * q_raw,k_raw (B,D) encoder outputs (normalised inside); queue (D,K) fp32; logits = [q.k, q@queue]/t,
 * CE with label 0 on the PRE-enqueue queue; then queue[:, ptr:ptr+Nk] = keys_all^T (keys_all (Nk,D),
 * the all-gathered normalised keys; may be NULL on one rank = normalised k_raw) and *ptr=(ptr+Nk)%K.
 * dq (nullable) = d loss / d q_raw.  k_norm_out (nullable, (B,D)) receives normalised keys.
 * One persistent launch of five phases with grid-wide barriers (row norms; logits = q . queue as a skinny MFMA GEMM; row
 * softmax + loss; gradient = p . queue^T as a second skinny GEMM split over K; dq, enqueue, pointer), every phase spread over
 * all workgroups, fixed-order sums.  K % 4 == 0.  ws: cmu_moco_ws_bytes(B, D, K) bytes, 16-byte aligned.                  */
int64_t cmu_moco_ws_bytes(int B, int D, int K);
int cmu_moco_infonce_enqueue(const float* q_raw, const float* k_raw, const float* keys_all, int Nk,
                             float* queue, int64_t* queue_ptr, float* loss, float* dq, float* k_norm_out,
                             int B, int D, int K, float temperature, void* ws, void* stream);
/* L2-normalise rows (F.normalize(dim=1), eps 1e-12) -- used before the key all-gather. */
int cmu_l2_normalize_rows(const float* x, float* out, int B, int D, void* stream);
/* The NON-fused MoCo API (moco2_module.py:224-285 ``forward`` / ``_compute_l_s``, :311-329 ``validation_step``; round 5: no ATen op left in
 * those paths).  All fp32, one workgroup per row, fixed-order sums.
 *   cmu_l2_normalize_rows_bwd: backward of F.normalize(x, dim=1) (:256, :259): dx = dy / n - x (x . dy) / n^3, n = |x| (dy / eps below the clamp)
 *   cmu_moco_logits_assemble:  logits (B, 1 + K) = [ q[b] . k[b] | lneg[b][:] ] * inv_t   (torch.cat([l_pos, l_neg], 1) / T, :258-267)
 *   cmu_moco_logits_split / _addpos: its backward -- dlneg (B, K) = dlogits[:, 1:] * inv_t (operand of dq = dlneg @ queue^T), then
 *                              dq[b] += dlogits[b][0] * inv_t * k[b]
 *   cmu_row_cross_entropy:     F.cross_entropy(logits (B, N), target int64 (B)) with mean reduction (:283, :324): loss[0], row_loss[B],
 *                              dlogits (optional) = (softmax - onehot) / B, rank (optional) [b] = number of logits strictly above the
 *                              target's (pl_bolts precision_at_k, metrics/aggregation.py:19-32: hit iff rank < k)
 *   cmu_scale_by_device_scalar: v[i] *= s[0], s on the device (the incoming gradient of a scalar loss) */
int cmu_l2_normalize_rows_bwd(const float* x, const float* dy, float* dx, int B, int D, void* stream);
int cmu_moco_logits_assemble(const float* q, const float* k, const float* lneg, float* logits, int B, int D, int K, float inv_t, void* stream);
int cmu_moco_logits_split(const float* dlogits, float* dlneg, int B, int K, float inv_t, void* stream);
int cmu_moco_logits_addpos(const float* dlogits, const float* k, float* dq, int B, int D, int K, float inv_t, void* stream);
int cmu_row_cross_entropy(const float* logits, const int64_t* target, float* loss, float* row_loss, float* dlogits, int* rank, int B, int N,
                          void* stream);
int cmu_scale_by_device_scalar(float* v, const float* s, int64_t n, void* stream);
/* SparK.patchify / unpatchify (Pretraining/Spark/spark.py:133-148), fp32: inverse 0: src (B, C, h*p, w*p) -> dst (B, h*w, p*p*C) with
 * dst[b][hy*w + wx][(py*p + px)*C + c] = src[b][c][hy*p + py][wx*p + px]; inverse 1: the other way round (src is the patch tensor). */
int cmu_patchify(const float* src, float* dst, int B, int C, int h, int w, int p, int inverse, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SparK sparse (masked) convolution support (Pretraining/Spark/encoder.py:12-56, spark.py:88-131).
 * `active` = (B,f,f) uint8 patch map; a (H,W) level with H = f << s looks up active[b][y>>s][x>>s].
 * ------------------------------------------------------------------------------------------- */
/* per-channel sum / sum-of-squares over the selected pixels (active, or non-active if invert): the statistics of
 * sp_bn_forward (encoder.py:26-36) and the mask-token gradient.  slab: [cmu_masked_stats_rows()][2][C] fp32,
 * fully written; feed it to cmu_bn_finalize with count = number of selected pixels.                       */
int cmu_masked_stats_rows(void);
int cmu_masked_channel_stats(const void* x, int64_t ldx, const uint8_t* active, int f, int invert, float* slab,
                             int B, int H, int W, int C, int dt, void* stream);
/* out = selected ? (relu ? max(x*scale+shift,0) : x*scale+shift) : fill[c]   (scale/shift/fill nullable = 1/0/0):
 * sparse BN apply + ReLU with zeros at masked positions, `x *= active` (encoder.py:20-23), densify with mask
 * tokens torch.where(active, feat, token) (spark.py:103-107), and their gradients.                         */
int cmu_mask_select(const void* x, int64_t ldx, const float* scale, const float* shift, int relu, const uint8_t* active,
                    int f, int invert, const float* fill, void* out, int64_t ldo, int B, int H, int W, int C, int dt,
                    void* stream);
/* BatchNorm backward restricted to the active positions (autograd of sp_bn_forward): count = number of active
 * pixels; masked positions contribute nothing and receive dY = 0.                                          */
int cmu_bn_bwd_reduce_masked(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                             const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef,
                             const uint8_t* active, int f, int64_t count, int B, int H, int W, int C, int dt, void* ws,
                             void* stream);
int cmu_bn_bwd_apply_masked(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                            const float* save_mean, const float* save_invstd, const float* coef, void* dY, int64_t ldo,
                            const uint8_t* active, int f, int B, int H, int W, int C, int dt, void* stream);
/* Tile skipping for the sparse encoder (K17; Spark/encoder.py:20-23 computes the dense op and multiplies by the mask --
 * here the masked tiles are never computed).  cmu_sparse_tile_list writes, in ascending order, the indices (dense numbering
 * (b * tilesY + ty) * tilesX + tx) of the tile_h x tile_w pixel tiles of a (B, H, W) level that overlap an active patch,
 * and their number to count[0] -- both on the device: the consumers read the count there, nothing synchronises with the host.
 * list: B * ceil(H / tile_h) * ceil(W / tile_w) ints.
 *   cmu_conv3x3_fwd_tiles   cmu_conv3x3_fwd (forward and, with the flipped pack, data gradient) over a 16 x 32 tile list;
 *                           y outside the listed tiles is left untouched (its consumers select by the mask).  Shapes the
 *                           persistent kernel serves (cmu_conv3x3_tiles_supported != 0): whole tiles, whole channel blocks.
 *   cmu_conv3x3_wgrad_tiles cmu_conv3x3_wgrad with the contraction restricted to a tile list (dY is zero elsewhere): 16 x 16 tiles
 *                           (tile_h 16, the first kernel) or 8 x 16 tiles (tile_h 8, the wide kernel: cmu_conv3x3_wgrad_tile_h).
 *
 * Gather form for the levels whose patches are smaller than a tile (every dense tile holds an active pixel there): the
 * convolution over the LIST of active pixels -- GEMM rows = active pixels (rows[r] = dense pixel index (b*H + y)*W + x, patch-
 * major, written with its count by cmu_sparse_pixel_list), K = 9 taps x Cin gathered per tap from the dense NHWC input (masked
 * neighbours hold zeros there), outputs scattered to y at the listed pixels, the rest of y untouched.  FLOPs = active fraction
 * of the dense launch.  Same packed weights as cmu_conv3x3_fwd (flipped pack: data gradient).
 *   cmu_conv3x3_rows_supported  Cout % 128 == 0, Cin a whole number of 128-byte steps, input tensor below 2 GiB
 *   cmu_sparse_pixel_list       rows[0 .. capacity) (entries past the end = -1), count[0]; ws: cmu_sparse_pixel_list_ws_bytes(B, f)
 *   cmu_conv3x3_fwd_rows        max_rows: upper bound of *n_rows known to the host (sizes the grid; surplus workgroups exit)   */
int cmu_sparse_tile_list(const uint8_t* active, int f, int B, int H, int W, int tile_h, int tile_w, int* list, int* count, void* stream);
/* cmu_masked_channel_stats / cmu_bn_bwd_reduce_masked over the list of active pixels: only the listed pixels are visited. */
int cmu_rows_channel_stats(const void* x, int64_t ldx, const int* rows, const int* n_rows, float* slab, int B, int H, int W, int C, int dt,
                           void* stream);
int cmu_bn_bwd_reduce_rows(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                           const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef, const int* rows,
                           const int* n_rows, int64_t max_rows, int64_t count, int B, int H, int W, int C, int dt, void* ws, void* stream);
int64_t cmu_sparse_pixel_list_ws_bytes(int B, int f);
int cmu_sparse_pixel_list(const uint8_t* active, int f, int B, int H, int W, int* rows, int64_t capacity, int* count, void* ws, void* stream);
/* Every list of a step in one launch each (the mask is known before the step starts): cmu_sparse_tile_lists builds n <=
 * cmu_sparse_tile_lists_max() tile lists (entry i: level side H[i] = f << s, tiles tile_h[i] x tile_w[i], lists[i] / counts[i] as in
 * cmu_sparse_tile_list -- element for element the same lists); cmu_sparse_pixel_lists expands ONE patch list (a tile list with tiles
 * of one patch, e.g. H = f and 1 x 1 tiles) into the pixel lists of n levels, exactly as cmu_sparse_pixel_list writes them.
 * H, tile_h, tile_w, lists, counts, rows, capacity are HOST arrays of length n.                                              */
int cmu_sparse_tile_lists_max(void);
int cmu_sparse_tile_lists(const uint8_t* active, int f, int B, int n, const int* H, const int* tile_h, const int* tile_w, int* const* lists,
                          int* const* counts, void* stream);
int cmu_sparse_pixel_lists(const int* patches, const int* patch_count, int f, int B, int n, const int* H, int* const* rows,
                           const int64_t* capacity, int* const* counts, void* stream);
/* Patch-organised forms of the sparse encoder's element-wise passes (csrc/sparse_elem.hip; Spark/encoder.py:12-56, spark.py:98-111):
 * the unit of work is a group of patch rows, the branch on the patch's bit is uniform per workgroup, masked patches cost no loads.
 * Outputs at active positions are bit-identical to the pixel-organised entries named beside them.
 *   cmu_cells_supported        square level with H = f << s and a power-of-two number (<= 256) of 16-byte chunks per pixel
 *   cmu_bn_bwd_apply_cells     = cmu_bn_bwd_apply_masked; ring = 1: zeros are written only to the one-pixel border frame of each masked
 *                              patch (enough when every consumer of dY is list-driven: they read active patches + a one-pixel halo;
 *                              the interior of masked patches stays unwritten)
 *   cmu_mask_select_cells      = cmu_mask_select(invert 0): relu?(x * scale + shift) in active patches, elsewhere zeros (ring as above) or fill[c]
 *                              (nullable; the densify step's mask tokens, spark.py:103-107)
 *   cmu_maxpool_bwd_cells      = cmu_maxpool_bwd_masked: active patches only, dA elsewhere unwritten (H / f >= 2)
 *   cmu_cells_channel_sum      out[c] = sum of x over the pixels of the active (invert 0) / masked (invert 1) patches -- the mask-token
 *                              gradient of spark.py:104-108; fixed-order fp32 partial sums, double final; ws: cmu_cells_channel_sum_ws_bytes(C) */
/* The first layer (Conv2d(1, Cout, 3, p=1), Finetuning/model.py:17) over a list of 16 x 16 tiles (cmu_sparse_tile_list numbering) for the
 * sparse encoder: cmu_conv3x3_c1_fwd_tiles computes the listed tiles only (y elsewhere untouched); stats, if given, is
 * [cmu_conv3x3_c1_fwd_tiles_rows(max_tiles)][2][Cout], fully written -- when the patches are multiples of 16 pixels these are the
 * sparse BatchNorm statistics.  cmu_conv3x3_c1_wgrad_bn_tiles = cmu_conv3x3_c1_wgrad_bn (yraw) / cmu_conv3x3_c1_wgrad_bn_w (w) with the
 * contraction restricted to the listed tiles: no masked BatchNorm-backward apply pass, no dY tensor.  max_tiles: host-side upper bound
 * of tile_count[0].  ws: cmu_conv3x3_c1_wgrad_ws_bytes.                                                                       */
int cmu_conv3x3_c1_fwd_tiles_rows(int64_t max_tiles);
int cmu_conv3x3_c1_fwd_tiles(const float* x, const uint8_t* mask, int mask_per_sample, const float* w, void* y, int64_t ldy, float* stats,
                             const int* tile_list, const int* tile_count, int64_t max_tiles, int B, int H, int W, int Cout, int dt,
                             void* stream);
int cmu_conv3x3_c1_wgrad_bn_tiles(const float* x, const uint8_t* mask, int mask_per_sample, const void* dA, int64_t ldd, const void* yraw,
                                  int64_t ldy, const float* w, const float* scale, const float* shift, const float* save_mean,
                                  const float* save_invstd, const float* coef, const int* tile_list, const int* tile_count,
                                  int64_t max_tiles, float* dW, int B, int H, int W, int Cout, int dt, void* ws, void* stream);
int cmu_cells_supported(int B, int H, int W, int f, int C, int dt);
int cmu_bn_bwd_apply_cells(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                           const float* save_mean, const float* save_invstd, const float* coef, void* dY, int64_t ldo,
                           const uint8_t* active, int f, int ring, int B, int H, int W, int C, int dt, void* stream);
int cmu_mask_select_cells(const void* x, int64_t ldx, const float* scale, const float* shift, int relu, const uint8_t* active, int f,
                          int ring, const float* fill, void* out, int64_t ldo, int B, int H, int W, int C, int dt, void* stream);
int cmu_maxpool_bwd_cells(const void* dP, int64_t ldp, const void* dSkip, int64_t lds, const void* y, int64_t ldy, const float* scale,
                          const float* shift, const uint8_t* active, int f, void* dA, int64_t lda, int B, int H, int W, int C, int dt,
                          void* stream);
/*   cmu_cells_channel_stats    = cmu_masked_channel_stats(invert 0) / cmu_rows_channel_stats: slab [cmu_cells_stats_rows()][2][C], fully written
 *   cmu_bn_bwd_reduce_cells    = cmu_bn_bwd_reduce_masked / cmu_bn_bwd_reduce_rows (same sums in another fixed order); ws: cmu_bn_bwd_ws_bytes(C) */
int cmu_cells_stats_rows(void);
int cmu_cells_channel_stats(const void* x, int64_t ldx, const uint8_t* active, int f, float* slab, int B, int H, int W, int C, int dt,
                            void* stream);
int cmu_bn_bwd_reduce_cells(const void* dA, int64_t ldd, const void* y, int64_t ldy, const float* scale, const float* shift,
                            const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef,
                            const uint8_t* active, int f, int64_t count, int B, int H, int W, int C, int dt, void* ws, void* stream);
int64_t cmu_cells_channel_sum_ws_bytes(int C);
int cmu_cells_channel_sum(const void* x, int64_t ldx, const uint8_t* active, int f, int invert, float* out, void* ws, int B, int H, int W,
                          int C, int dt, void* stream);
int cmu_conv3x3_rows_supported(int B, int H, int W, int Cin, int Cout, int dt);
int cmu_conv3x3_fwd_rows(const void* x, int64_t ldx, const void* wpacked, void* y, int64_t ldy, const int* rows, const int* n_rows,
                         int64_t max_rows, int B, int H, int W, int Cin, int Cout, int dt, void* stream);
int cmu_conv3x3_tiles_supported(int B, int H, int W, int Cin, int Cout, int dt);
int cmu_conv3x3_fwd_tiles(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from,
                          const void* wpacked, void* y, int64_t ldy, const int* tile_list, const int* tile_count,
                          int B, int H, int W, int Cin, int Cout, int dt, void* stream);
/* tile height of the list cmu_conv3x3_wgrad_tiles wants for this shape: 8 (8 x 16 pixel tiles: the wide weight-gradient kernel
 * serves it) or 16 (16 x 16 tiles: the first kernel)                                                                       */
int cmu_conv3x3_wgrad_tile_h(int B, int H, int W, int Cin, int Cout, int dt);
int cmu_conv3x3_wgrad_tiles(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from,
                            const void* dY, int64_t ldd, float* dW, const int* tile_list, const int* tile_count, int tile_h,
                            int B, int H, int W, int Cin, int Cout, int dt, void* ws, void* stream);

/* SparK loss (spark.py:112-123): p x p patches, target patch normalised with its own mean / unbiased variance
 * (eps 1e-6), l2 = patch mean of (rec-target)^2, loss = sum over NON-active patches / (count + 1e-8).
 * rec, img (B,f*p,f*p) fp32; drec nullable.  ws: cmu_spark_loss_ws_bytes(B,f).                              */
int64_t cmu_spark_loss_ws_bytes(int B, int f);
int cmu_spark_loss_fwd_bwd(const float* rec, const float* img, const uint8_t* active, float* loss, float* drec,
                           float loss_scale, int B, int f, int p, void* ws, void* stream);

/* Global average pool of the activated latent (moco_data_module.py:65, x.mean([2,3])): out (B,C) fp32 from the
 * raw NHWC tensor + pending transform; backward broadcasts dout/(H*W) into dA (gradient w.r.t. the activation). */
int cmu_gap_fwd(const void* y, int64_t ldy, const float* in_scale, const float* in_shift, float* out,
                int B, int H, int W, int C, int dt, void* stream);
int cmu_gap_bwd(const float* dout, void* dA, int64_t ldd, int B, int H, int W, int C, int dt, void* stream);

/* EMA p_t = m*p_t + (1-m)*p_o over a flat fp32 arena (cmunet.py:78-92, moco2_module.py:153-158). */
int cmu_ema_update(float* target, const float* online, int64_t n, float momentum, void* stream);

/* Fused Adam/AdamW step over a flat fp32 arena (torch.optim.Adam, train.py:341; AdamW cmunet_config.py:79-83).
 * decoupled != 0: AdamW (p *= 1 - lr*wd) else L2 (g += wd*p).  wd_mask (nullable, uint8 per element):
 * weight decay applies where mask != 0 (bias / norm parameters are excluded, cmunet_config.py:84-91).
 * grad_scale multiplies the gradient first (loss-scale removal / DP mean).
 * amp_state (nullable, cmu_amp_*): the update is skipped when the state's found_inf flag is set, the gradient is divided
 * by the state's scale, and the step number of the bias corrections is the state's count of updates taken + 1 (``step``
 * is ignored) -- GradScaler.step / unscale_ without a host round trip.                               */
int cmu_adam_step(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled,
                  int64_t step, float grad_scale, const void* amp_state, void* stream);

/* cmu_adam_step + the EMA of the momentum networks in ONE pass over the arena: what the reference runs as AdamW.step
 * (cmunet_config.py:76-91) followed by MomentumUpdateHook.after_train_iter -> CM_UNet.momentum_update (cmunet.py:78-92,
 * momentum_update_hook.py:42-47).  Up to two ascending, disjoint element ranges [seg_lo[k], seg_hi[k]) of the online arena
 * (multiples of 4) have a target array seg_target[k] (16-byte aligned, hi - lo elements): after the element's AdamW update,
 * target = target*m + p_new*(1-m).  A step skipped by amp_state still runs the EMA with the unchanged parameters (the hook
 * runs behind a skipped optimiser step too).  n must be a multiple of 4, all arenas 16-byte aligned, wd_mask 4-byte aligned.
 * Bit-identical to cmu_adam_step followed by cmu_ema_update per segment.                                             */
int cmu_adam_ema_step(float* p, const float* g, float* m, float* v, const uint8_t* wd_mask, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int decoupled,
                      int64_t step, float grad_scale, const void* amp_state, int nseg, const int64_t* seg_lo,
                      const int64_t* seg_hi, float* const* seg_target, float ema_momentum, void* stream);

/* Dynamic loss scaling: AmpOptimWrapper(loss_scale='dynamic') of Pretraining/CM-UNet/configs/cmunet_config.py:76-78, i.e.
 * torch.cuda.amp.GradScaler(init_scale 2^16, growth 2, backoff 0.5, growth_interval 2000), with the state on the device
 * (cmu_amp_state_bytes() = 32 bytes: float scale, float found_inf, int32 growth_tracker, int32 good_steps,
 * int32 skipped_steps, pad): no host synchronisation per step.  Per step: the loss kernel scales its gradient by
 * state.scale; after the gradient exchange cmu_amp_check_finite raises found_inf if any gradient is inf / nan; the
 * optimiser kernel skips or unscales (see cmu_adam_step); cmu_amp_update halves the scale after a skipped step or doubles
 * it after growth_interval clean ones, and clears found_inf.                                                      */
int cmu_amp_state_bytes(void);
int cmu_amp_init(void* state, float init_scale, void* stream);
int cmu_amp_check_finite(const float* g, int64_t n, void* state, void* stream);
int cmu_amp_update(void* state, float growth_factor, float backoff_factor, int growth_interval, void* stream);

/* soft-clDice building blocks (Finetuning/metrics.py:401-492): soft skeleton of an fp32 (planes, H, W) stack by num_iter
 * rounds of min/max pooling (the reference uses 10), and the four sums sum(skel_pred*y_true), sum(skel_pred),
 * sum(skel_true*y_pred), sum(skel_true) -> out4 (device), from which clDice = 1 - 2*tprec*tsens/(tprec+tsens).        */
int cmu_softmax2_threshold(const float* logits /*(B,2,H,W)*/, float threshold, float* out /*(B,H,W) 0/1*/, int B, int H, int W, void* stream);
int64_t cmu_soft_skeleton_ws_bytes(int64_t n);
int cmu_soft_skeleton(const float* img, float* skel, int planes, int H, int W, int num_iter, void* ws, void* stream);
int64_t cmu_cldice_sums_ws_bytes(void);
int cmu_cldice_sums(const float* skel_pred, const float* y_true, const float* skel_true, const float* y_pred, int64_t n,
                    float* out4, void* ws, void* stream);

/* Fused SGD step over a flat fp32 arena (torch.optim.SGD; MoCo: moco2_module.py:339-344, momentum 0.9, weight decay 1e-4).
 * g' = g*grad_scale + wd*p (where wd_mask != 0 or wd_mask == NULL); buf = g' on step 1, else momentum*buf + (1-dampening)*g';
 * p -= lr * (nesterov ? g' + momentum*buf : buf); momentum == 0: p -= lr*g' (buf may be NULL).                        */
int cmu_sgd_step(float* p, const float* g, float* buf, const uint8_t* wd_mask, int64_t n, float lr, float momentum,
                 float dampening, float weight_decay, int nesterov, int64_t step, float grad_scale, void* stream);

/* Fused LAMB step (Pretraining/Spark/utils/lamb.py:67-159) over a flat fp32 arena cut into blocks of at most
 * cmu_lamb_block_elems() elements that never straddle a parameter tensor: blk_start / blk_count / blk_tensor [nblocks]
 * (device), t_blk0 [ntensors+1] first block of each tensor, t_wd [ntensors] its weight decay (0 = excluded; such tensors
 * skip the trust ratio unless always_adapt).  u: scratch arena of the parameters' size; ws: cmu_lamb_ws_bytes.
 * Global gradient-norm clip (max_grad_norm <= 0: off), moments, per-tensor trust ratio, update -- no host sync.          */
int cmu_lamb_block_elems(void);
int64_t cmu_lamb_ws_bytes(int nblocks, int ntensors);
int cmu_lamb_step(float* p, const float* g, float* m, float* v, float* u, const int64_t* blk_start, const int* blk_count,
                  const int* blk_tensor, int nblocks, const int* t_blk0, const float* t_wd, int ntensors, float lr, float beta1,
                  float beta2, float eps, int bias_correction, int grad_averaging, float max_grad_norm, int trust_clip,
                  int always_adapt, int64_t step, float grad_scale, void* ws, void* stream);

/* ---- input pipeline on the device (Pretraining/CM-UNet/cmae/datasets/cmunet_dataset.py:60-88) -------------
 * cmu_resize_bicubic: Pillow-convention bicubic resize (mode 'F': Keys a = -0.5, antialiased support, double accumulation,
 * float32 store after each of the two passes, horizontal first) of a per-sample integer crop window to (Ho, Wo), then an
 * optional horizontal flip -- cmunet_dataset.py:74-75 (boxes == NULL: whole image) and RandomResizedCrop + RandomFlip of
 * configs/cmunet_config.py:49-50.  src (B,Hs,Ws) f32; boxes (B,4) int32 device (x0, y0, w, h) or NULL; flip (B) u8 device
 * or NULL; out (B,Ho,Wo) f32; ws: cmu_resize_bicubic_ws_bytes.  Bit-identical to Pillow 12.2 on finite inputs.             */
int64_t cmu_resize_bicubic_ws_bytes(int B, int Hs, int Ws, int Ho, int Wo);
int cmu_resize_bicubic(const float* src, int B, int Hs, int Ws, const int* boxes, const uint8_t* flip, float* out, int Ho, int Wo,
                       void* ws, void* stream);
/* cmu_resize_bicubic_u8: the same call on 8-bit images (PIL mode 'L', what Image.fromarray makes of a uint8 .npy:
 * Finetuning/dataset.py:44-46, Pretraining/Spark/utils/dataset.py:25-27, cmunet_dataset.py:74-75): Pillow's 8-bit resampler -- the
 * double coefficients as 22-bit fixed point (rounded half away from zero), int32 accumulation from 2^21, arithmetic shift, clip to
 * 0..255, a uint8 image between the passes.  src / out uint8, otherwise as cmu_resize_bicubic.  Bit-identical to Pillow 12.2.
 * cmu_resize_nearest_u8: Image.resize(size, NEAREST) of the label masks (Finetuning/dataset.py:47): source index int(x) of Pillow's
 * running double x = scale / 2, += scale per output index.  ws: cmu_resize_nearest_u8_ws_bytes (the two index tables).          */
int64_t cmu_resize_bicubic_u8_ws_bytes(int B, int Hs, int Ws, int Ho, int Wo);
int cmu_resize_bicubic_u8(const uint8_t* src, int B, int Hs, int Ws, const int* boxes, const uint8_t* flip, uint8_t* out, int Ho, int Wo,
                          void* ws, void* stream);
int64_t cmu_resize_nearest_u8_ws_bytes(int Ho, int Wo);
int cmu_resize_nearest_u8(const uint8_t* src, int B, int Hs, int Ws, uint8_t* out, int Ho, int Wo, void* ws, void* stream);
/* cmu_two_view: 'img' = src[b, :out, :out] (ShiftPixel(0), pipelines/processing.py:97-127); 'img_t' = src[b, dy:dy+out,
 * dx:dx+out] + (max of that crop / 10) * z evaluated in float64 and cast to float32 (GaussNoise,
 * pipelines/auto_augment.py:1136-1153).  shifts (B,2) int32 device (dy, dx); z = noise (B,out,out) f64 device, or, when
 * noise == NULL, the in-kernel generator: Philox4x32-10 with counter (element index, 0, 0) and key `seed`, Box-Muller on
 * its first two words (cmu_philox_normal returns the same draws).                                                        */
int cmu_two_view(const float* src, int B, int S, const int* shifts, const double* noise, uint64_t seed, float* img, float* img_t,
                 int out, void* stream);
int cmu_philox_normal(double* out, int64_t n, uint64_t offset, uint64_t seed, void* stream);
/* Random patch mask of backbones/UNet_encoder.py:106-139 on the device: mask (B,H,W) u8, 1 = masked; per sample a uniformly
 * random n_mask-subset of the (H/patch)*(W/patch) patches (<= 4096): patch p of sample b draws the first word of
 * Philox4x32-10(counter = offset + b*P + p, key = seed) and is masked iff (key, p) is among the n_mask smallest.            */
int cmu_random_patch_mask(uint8_t* mask, int B, int H, int W, int patch, int n_mask, uint64_t seed, uint64_t offset, void* stream);

/* ---- skinny (weight-streaming) GEMMs of the projector / predictor necks (nonlinear_neck.py:63-66, 95-101 with
 * cmunet_config.py:18-38: Linear(H*W -> 1536)) -- nn.Linear and its autograd, fp32, M <= 256 rows per GPU (the reference's batch
 * size, cmunet_config.py:55; moco2_module.py:91,262 for the queue logits), taken in groups of 32 rows (one 32-row MFMA tile):
 *   fwd    y (M,N) = x (M,K) . w (N,K)^T + bias (or NULL)      K % 8 == 0; ws: cmu_skinny_gemm_ws_bytes (split-K slab)
 *   dgrad  dx (M,K) = dy (M,N) . w (N,K)                        K % 4 == 0; ws: cmu_skinny_gemm_bwd_ws_bytes (dy transposed)
 *   wgrad  dw (N,K) = dy^T . x, dbias (N) = sum_m dy (or NULL)  K % 4 == 0
 * x, w, dx, dw 16-byte aligned, rows contiguous.  fwd / dgrad: one pass over the weights per group of 32 rows; wgrad: one
 * launch contracting over all rows; v_mfma_f32_32x32x2_f32.  M > 256 is an argument error (there is no library GEMM behind
 * these entries: the host side raises).                                                                                   */
int64_t cmu_skinny_gemm_ws_bytes(int M, int N, int64_t K);
int cmu_skinny_gemm_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int64_t K, void* ws, void* stream);
int64_t cmu_skinny_gemm_bwd_ws_bytes(int M, int N);
int cmu_skinny_gemm_dgrad(const float* dy, const float* w, float* dx, int M, int N, int64_t K, void* ws, void* stream);
int cmu_skinny_gemm_wgrad(const float* dy, const float* x, float* dw, float* dbias, int M, int N, int64_t K, void* stream);

/* 16-bit-operand variants for the AMP configuration (cmunet_config.py:76-78: nn.Linear under autocast multiplies fp16
 * operands into fp32): x, w, dy, dx, dw stay fp32 in memory (the master weights of the optimiser), operands are rounded to
 * dt (CMU_F16 / CMU_BF16) in registers, accumulation fp32 -- v_mfma_f32_32x32x16.  Same contracts as above; fwd needs
 * K % 16 == 0; dgrad reads dy in place (no workspace).                                                                  */
int64_t cmu_skinny16_gemm_ws_bytes(int M, int N, int64_t K);
int cmu_skinny16_gemm_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int64_t K, int dt, void* ws,
                          void* stream);
int cmu_skinny16_gemm_dgrad(const float* dy, const float* w, float* dx, int M, int N, int64_t K, int dt, void* stream);
int cmu_skinny16_gemm_wgrad(const float* dy, const float* x, float* dw, float* dbias, int M, int N, int64_t K, int dt, void* stream);

/* ---- BatchNorm1d (+ ReLU) of the necks' hidden layer (nonlinear_neck.py:58-60, 95-98: (Sync)BN eps 1e-6 then ReLU), rows M x
 * columns N fp32.  Statistics per column over the rows; on more than one rank the caller exchanges the column sums
 * (SyncBN):
 *   cmu_bn1d_colsums      sums[2][N] = (sum x, sum x^2)                       -> all-reduce -> cmu_bn1d_relu_fwd(sums, count)
 *   cmu_bn1d_relu_fwd     y = [relu](gamma * (x - mean) * invstd + beta); sums == NULL: statistics of the M rows (two-pass
 *                         variance, as F.batch_norm), else mean = sums[0]/count, var = sums[1]/count - mean^2; training:
 *                         running statistics updated (unbiased variance, momentum); training == 0: running statistics used.
 *                         save_mean / save_invstd (nullable) are written in both modes (eval: running_mean and
 *                         1/sqrt(running_var + eps): what cmu_bn1d_relu_bwd needs for the fixed affine map, with zero sums).
 *                         gamma / beta nullable (affine=False).
 *   cmu_bn1d_bwd_colsums  sums[2][N] = (sum dz, sum dz * xhat), dz = dy * [y > 0] when relu   -> all-reduce
 *   cmu_bn1d_relu_bwd     dx = gamma * invstd * (dz - S0/count - xhat * S1/count) with (S0, S1) = sums or, if NULL, the local
 *                         sums (count = M); dgamma / dbeta (nullable) = LOCAL sums (the gradient exchange averages them).   */
int cmu_bn1d_colsums(const float* x, float* sums, int M, int N, void* stream);
int cmu_bn1d_relu_fwd(const float* x, const float* sums, int64_t count, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, int training, int relu, float* y, float* save_mean,
                      float* save_invstd, int M, int N, void* stream);
int cmu_bn1d_bwd_colsums(const float* dy, const float* x, const float* y, const float* save_mean, const float* save_invstd, int relu,
                         float* sums, int M, int N, void* stream);
int cmu_bn1d_relu_bwd(const float* dy, const float* x, const float* y, const float* save_mean, const float* save_invstd,
                      const float* gamma, int relu, const float* sums, int64_t count, float* dx, float* dgamma, float* dbeta, int M,
                      int N, void* stream);

/* Conv2d(K, N, 1) from an NHWC dt tensor with its pending transform to an NCHW fp32 tensor: the per-call `reduce_channels`
 * of the target latent (Pretraining/CM-UNet/cmae/models/algorithms/cmunet.py:128-131; the reference then re-views the NCHW
 * memory as a (B,1,H,W) image).  w (N,K) fp32 (rounded to dt in registers), bias (N) or NULL.  K a whole number of 32-byte
 * slices, H*W % 4 == 0.                                                                                                   */
int cmu_conv1x1_nchw_fwd(const void* x, int64_t ldx, const float* in_scale, const float* in_shift, int relu_from, const float* w,
                         const float* bias, float* out, int B, int H, int W, int K, int N, int dt, void* stream);

/* Measurement support (bench.py's roofline block; no reference counterpart): the 16-bit MFMA rate this chip SUSTAINS.
 * Every wave of a one-workgroup-per-CU grid issues iters x 8 back-to-back 32x32x16 MFMAs from registers (no LDS, no
 * memory) or -- lds_fed = 1 -- the same MFMAs fed from LDS at the persistent conv kernel's ratio (6 fragment reads of 1 KB
 * per 8 MFMAs, register double buffer); operands are generated in the kernel: pattern 0 = dense ~N(0,1), 1 = ReLU'd
 * (half zeros), 2 = all zeros.
 * The power management lowers the shader clock under this load by an amount that depends on the operand data
 * (csrc/probe.hip); *tflops and *clock_mhz (s_memtime / s_memrealtime of one wave) come back
 * after a synchronisation of the stream.  scratch64: 64 bytes of device memory.  Host-blocking; not for the hot path. */
int cmu_mfma_sustained_rate(int dt, int pattern, int lds_fed, int iters, void* scratch64, double* tflops, double* clock_mhz,
                            void* stream);
/* Diagnostic (tools/exchange_probe.py): a stand-in for a collective's reduction kernel -- `grid` workgroups of 256
 * threads compute out = a + b over n floats (n % 4 == 0) `passes` times on `stream`.  Launched on a side stream where a trainer announces a
 * gradient bucket (reference: DDP's bucketed all-reduce, Pretraining/Spark/main.py:102), it shows on ONE GPU whether such a kernel runs beside
 * the persistent conv kernels and what the backward pass loses.  Not used by the product path. */
int cmu_probe_stream_reduce(const float* a, const float* b, float* out, int64_t n, int grid, int passes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CMUNET_HIP_H */
