/*
 * robosat_hip.h -- C ABI of librobosat_hip.so: the MI355X (gfx950) implementation of the RoboSat U-Net hot path.
 *
 * The reference (mapbox/robosat v1.2.0) has no FFI of its own: its hot path is Python calling torch ops
 * (SURVEY.md section 0.1).  This header is therefore the boundary a maintainer binds *beneath* the reference's
 * Python operator surface; every entry point names the reference call it replaces (file:line under
 * /root/reference).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain device pointers + sizes, no torch types; `stream` is a hipStream_t passed as void*.
 *   - every function only ENQUEUES work on `stream` (no allocation, no implicit sync) and returns a hipError_t
 *     as int (0 = success); argument errors return RS_EINVAL (-22) before anything is enqueued.
 *   - activations are NHWC fp32 ("channels last"): x[n][y][x][c].  Convolution weights are KRSC fp32
 *     (w[cout][ky][kx][cin]) == a torch [Cout,Cin,kh,kw] tensor in channels_last memory format.
 *   - logits / probabilities / loss inputs are NCHW fp32, label maps are [N][H][W] int64: exactly the tensors the
 *     reference's losses and predict tool see (robosat/losses.py, robosat/tools/predict.py:87).
 */
#ifndef ROBOSAT_HIP_H
#define ROBOSAT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_EINVAL (-22)

typedef void* rs_stream_t;

/* Activation storage types of the *_dt / *_bf16 entry points (the bf16 training path, BASELINE configs[2]).
 * bf16 tensors are passed as raw 16-bit words (torch.bfloat16 storage); arithmetic and accumulation stay fp32. */
typedef uint16_t rs_bf16;
#define RS_F32 0
#define RS_BF16 1

/* Version of this ABI (bumped on any signature change). */
int rs_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Convolution (implicit GEMM on v_mfma_f32_32x32x2_f32; exact fp32).
 *
 * One descriptor covers every convolution in UNet.forward (robosat/unet.py:122-141):
 *   - resnet.conv1 7x7/2 stem                      (unet.py:122)   stem = 1
 *   - Bottleneck 1x1, 3x3 (stride 1|2), downsample (unet.py:127-130; torchvision 0.3.0 resnet50)
 *   - ConvRelu 3x3 + ReLU                          (unet.py:32,44)
 *   - DecoderBlock = interpolate(nearest, x2) then ConvRelu (unet.py:73), with the torch.cat of the skip tensor
 *     (unet.py:134-137) folded in: src1 = skip (first C1 channels), src2 = previous decoder output (next C2).
 *     Neither the upsampled nor the concatenated tensor is ever materialised.
 * and, by symmetry, every data-gradient convolution of loss.backward() (robosat/tools/train.py:186):
 *   ups = 2 reads the source through a zero-inserted x2 grid (the adjoint of a stride-2 convolution).
 *
 * out[n][oy][ox][co] = epilogue( sum_{ky,kx,ci} in[n][oy*stride-pad+ky][ox*stride-pad+kx][ci] * w[co][ky][kx][ci] )
 * epilogue(v) = relu?( v * scale[co] + shift[co] + residual[n][oy][ox][co] )   (each part optional / NULL)
 * and, when relu_mask != NULL, the result is zeroed wherever relu_mask[n][oy][ox][co] <= 0 (the ReLU backward of the
 * layer whose output `relu_mask` is, fused into the data-gradient convolution that produces its gradient).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct rs_conv_desc {
  int32_t N;          /* batch */
  int32_t Hs, Ws;     /* spatial size of the source tensor(s) as stored */
  int32_t C1, C2;     /* channels of src1 / src2 (C2 = 0: single source); C1, C2 multiples of 32 unless stem */
  int32_t ups;        /* 0: none; 1: nearest x2 upsample fused into the gather; 2: zero-insert x2 (stride-2 dgrad) */
  int32_t kh, kw, stride, pad;
  int32_t Ho, Wo;     /* output spatial size */
  int32_t Cout;       /* multiple of 32 */
  int32_t relu;       /* 1: ReLU in the epilogue */
  int32_t stem;       /* 1: src1 is NHWC with 4 channels, weights packed [Cout][kh][8][4] (see rs_pack_stem_weight): resnet.conv1,
                         7x7 / stride 2 / pad 3 -> 64 (stem_f32.hip); 3: ... and the 4th channel of src1 and the filter's c = 3
                         entries are zeros (an RGB image through rs_nchw_to_nhwc4): they are skipped, not multiplied */
} rs_conv_desc;

int rs_conv2d_fwd(const rs_conv_desc* d, const float* src1, const float* src2, const float* weight,
                  const float* scale, const float* shift, const float* residual, const float* relu_mask, float* out,
                  rs_stream_t stream);

/* Which tile configuration rs_conv2d_fwd picks for `d` (index into rs_conv2d_tile_name); for the roofline report. */
int rs_conv2d_tile(const rs_conv_desc* d);
const char* rs_conv2d_tile_name(int tile);

/* resnet.conv1 weight [Cout][kh][kw][Cin<=4] (KRSC) -> [Cout][kh][8][4], zero padded (unet.py:122). */
int rs_pack_stem_weight(const float* w_krsc, float* packed, int Cout, int kh, int kw, int Cin, rs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Layout / elementwise / pooling
 * ---------------------------------------------------------------------------------------------------------- */

/* images.to(device) as UNet.forward receives them (train.py:172, predict.py:83): NCHW [N][C<=4][H][W] -> NHWC4,
 * channel 3 zero-filled when C == 3. */
int rs_nchw_to_nhwc4(const float* x, float* y, int N, int C, int H, int W, rs_stream_t stream);

/* F.max_pool2d on NHWC: resnet.maxpool 3x3/2 pad 1 (unet.py:125) and the 2x2/2 before `center` (unet.py:132).
 * Padding behaves as -inf (torch semantics).  C multiple of 4.  `argmax` (optional, uint8 [N][Ho][Wo][C]) receives
 * the winning tap ky*k+kx (first maximum in row-major window order, as torch picks) for the backward pass. */
int rs_maxpool2d_fwd(const float* x, float* y, uint8_t* argmax, int N, int H, int W, int C, int k, int stride, int pad,
                     int Ho, int Wo, rs_stream_t stream);

/* Eval-mode BatchNorm2d folded to per-channel scale/shift for the conv epilogue (resnet bn*, eps as torch):
 * scale = gamma / sqrt(var + eps), shift = beta - mean * scale. */
int rs_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
               float* shift, int C, rs_stream_t stream);

/* self.final (unet.py:108,141): 1x1 conv Cin=32..128 (multiple of 4, <= 128) -> C classes (<= 8) with bias, reading
 * NHWC and writing NCHW.  softmax = 1 additionally applies nn.functional.softmax(dim=1) (predict.py:87). */
int rs_final_conv1x1(const float* x, const float* w, const float* bias, float* out, int N, int H, int W, int Cin,
                     int C, int softmax, rs_stream_t stream);


/* ------------------------------------------------------------------------------------------------------------
 * Training path: what autograd synthesises for loss.backward() (robosat/tools/train.py:186) and train-mode
 * BatchNorm (net.train(), tools/train.py:169).  Workspaces are caller-provided; query sizes with *_workspace_bytes.
 * ---------------------------------------------------------------------------------------------------------- */

/* Filter gradient of the convolution described by `d` (same descriptor as the forward call; `relu` is ignored):
 *   dw[co][ky][kx][ci] = sum_{n,oy,ox} dy[n][oy][ox][co] * in[n][oy*stride-pad+ky][ox*stride-pad+kx][ci]
 * with `in` read through the forward gather (upsample / concat / stem packing).  dw is KRSC (packed [Cout][kh][8][4]
 * for the stem -> rs_unpack_stem_weight).  Split-P partials live in `workspace`; deterministic (no atomics). */
long rs_conv2d_wgrad_workspace_bytes(const rs_conv_desc* d);
/* Which form rs_conv2d_wgrad runs this fp32 launch in (a pure host decision on the geometry and the knobs): 0 the direct form,
 * 2 the phase form of DecoderBlock (unet.py:63-73; 4/9 of the multiply-adds), 3 the same in the Winograd domain of the forward's
 * F(2x2, 2x2) form (1/4; conv_wgrad_wino_f32.hip, round 6), 4 a stride-1 3x3 / pad-1 convolution (torchvision Bottleneck.conv2) in the
 * Winograd domain of F(2x2, 3x3) (16/36; conv_wgrad_wino33_f32.hip, round 6); RS_EINVAL for a descriptor rs_conv2d_wgrad refuses. */
int rs_conv2d_wgrad_form(const rs_conv_desc* d);
int rs_conv2d_wgrad(const rs_conv_desc* d, const float* dy, const float* src1, const float* src2, float* dw,
                    void* workspace, rs_stream_t stream);
int rs_unpack_stem_weight(const float* packed, float* w_krsc, int Cout, int kh, int kw, int Cin, rs_stream_t stream);

/* Forward KRSC weights [Cout][kh][kw][Cin] -> data-gradient weights [Cin][kh][kw][Cout] with flipped taps.  The data
 * gradient of a convolution is rs_conv2d_fwd on dy with these weights, pad' = k-1-pad, and ups = 2 when the forward
 * stride was 2. */
int rs_pack_dgrad_weight(const float* w_krsc, float* out, int Cout, int kh, int kw, int Cin, rs_stream_t stream);

/* Train-mode BatchNorm2d over y [M = N*H*W][C] (torchvision resnet50 bn*, eps 1e-5, momentum 0.1):
 * rs_bn_train_stats: batch mean / biased variance -> mean, invstd = 1/sqrt(var+eps), scale = gamma*invstd,
 *   shift = beta - mean*scale; running_mean/var (optional, both or neither) updated with the UNBIASED variance;
 *   *num_batches_tracked (optional, int64) += 1.
 * rs_bn_apply: out = relu?( y*scale + shift (+ residual) ).
 * rs_bn_bwd: g = dz * (zmask > 0) (zmask optional = the ReLU output); dgamma = sum g*xhat, dbeta = sum g,
 *   dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M); dmasked (optional) receives g (gradient of the residual branch).
 * workspace: rs_bn_workspace_bytes(M, C) + 3*C*sizeof(float) bytes. */
long rs_bn_workspace_bytes(long M, int C);
int rs_bn_train_stats(const float* y, long M, int C, float eps, float momentum, const float* gamma, const float* beta,
                      float* mean, float* invstd, float* scale, float* shift, float* running_mean, float* running_var,
                      long long* num_batches_tracked, void* workspace, rs_stream_t stream);
int rs_bn_apply(const float* y, const float* scale, const float* shift, const float* residual, float* out, long M, int C,
                int relu, rs_stream_t stream);
int rs_bn_bwd(const float* dz, const float* zmask, const float* y, const float* mean, const float* invstd,
              const float* gamma, float* dy, float* dmasked, float* dgamma, float* dbeta, long M, int C, void* workspace,
              rs_stream_t stream);

/* Backward of F.max_pool2d given the argmax taps recorded by rs_maxpool2d_fwd; accumulate = 1 adds into dx. */
int rs_maxpool2d_bwd(const float* dy, const uint8_t* argmax, float* dx, int N, int H, int W, int C, int k, int stride,
                     int pad, int Ho, int Wo, int accumulate, rs_stream_t stream);

/* Backward of interpolate(nearest, x2) + torch.cat split: dup [N][2H][2W][C1+C2] -> d1 [N][H][W][C1] (+= when
 * accumulate1), d2 [N][H][W][C2]; mask1/mask2 (optional): ReLU outputs whose backward is fused (zero where <= 0). */
int rs_upsample2x_bwd(const float* dup, float* d1, float* d2, const float* mask1, const float* mask2, int N, int H, int W,
                      int C1, int C2, int accumulate1, rs_stream_t stream);

/* Backward of self.final (unet.py:108,141): x NHWC [N][H][W][Cin<=64], dlogits NCHW -> dx (zeroed where x <= 0 when
 * relu_mask: x is the ReLU output dec5), dw [C][Cin], db [C]. */
long rs_final_conv1x1_bwd_workspace_bytes(int Cin, int C);
int rs_final_conv1x1_bwd(const float* x, const float* w, const float* dlogits, float* dx, float* dw, float* db, int N, int H,
                         int W, int Cin, int C, int relu_mask, void* workspace, rs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Losses (robosat/losses.py) and metrics (robosat/metrics.py) on NCHW logits + int64 [N][H][W] targets
 * ---------------------------------------------------------------------------------------------------------- */

/* mode 0: CrossEntropyLoss2d (losses.py:8-25); mode 1: FocalLoss2d with `gamma` (losses.py:28-50).  `weight` [C]
 * optional.  loss[0] = sum w*l / sum w; stats[0..1] = (loss, sum w) is consumed by the backward:
 * dlogits = grad_out[0] * w[t] * dl/dx / sum w  (grad_out NULL = 1). */
long rs_nll_loss_workspace_bytes(void);
int rs_nll_loss_fwd(const float* logits, const long long* targets, const float* weight, float* loss, float* stats, int N,
                    int C, int H, int W, int mode, float gamma, void* workspace, rs_stream_t stream);
int rs_nll_loss_bwd(const float* logits, const long long* targets, const float* weight, const float* stats,
                    const float* grad_out, float* dlogits, int N, int C, int H, int W, int mode, float gamma,
                    rs_stream_t stream);

/* mIoULoss2d (losses.py:53-83): max(1 - mean_{c,n} softIoU, weighted NLL), gradient through the larger branch.
 * stats needs 5 + 2*N*C floats: loss, sum w, chosen branch (0 miou / 1 nll), per-(n,c) gradient coefficients, then the
 * miou branch's value and the nll branch's numerator sum_i w_i l_i.  rs_miou_loss_bwd reads [1] (the NLL denominator) and
 * [2] (the branch) back: data-parallel ranks overwrite them with the global batch's (the reference evaluates
 * max(miou, nll) ONCE over the gathered batch, losses.py:83 under tools/train.py:69). */
long rs_miou_loss_workspace_bytes(int N, int C);
int rs_miou_loss_fwd(const float* logits, const long long* targets, const float* weight, float* loss, float* stats, int N,
                     int C, int H, int W, void* workspace, rs_stream_t stream);
int rs_miou_loss_bwd(const float* logits, const long long* targets, const float* weight, const float* stats,
                     const float* grad_out, float* dlogits, int N, int C, int H, int W, rs_stream_t stream);

/* LovaszLoss2d (losses.py:86-119): batched radix sort + scans.  loss[0] = mean over images; grad_unit (optional,
 * NCHW like logits) receives d loss / d logits for grad_out = 1 (multiply with rs_scale_by_scalar). */
long rs_lovasz_workspace_bytes(int N, int C, int H, int W);
int rs_lovasz_fwd(const float* logits, const long long* targets, float* loss, float* grad_unit, int N, int C, int H, int W,
                  void* workspace, rs_stream_t stream);
int rs_scale_by_scalar(const float* src, const float* scalar, float* dst, long n, rs_stream_t stream);

/* Metrics.add over a whole batch (metrics.py:27-41): counts[0..3] += (tn, fn, fp, tp) in the reference's naming. */
int rs_confusion_counts(const float* scores, const long long* targets, unsigned long long* counts, int N, int C, int H,
                        int W, rs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * bf16 path (BASELINE configs[2]: rs train bf16): the same operators with activations stored as bf16 NHWC, fp32
 * master weights cast per step, fp32 accumulation (v_mfma_f32_32x32x16_bf16), fp32 statistics / gradients of
 * parameters / logits / losses.  The image is cast to bf16 on upload (rs_nchw_to_nhwc4_bf16); the 7x7 stem has its own
 * bf16 kernels (rs_stem_conv_*_bf16).
 *
 * `_dt` variants: same contract as the fp32 entry point of the same name, activations typed by `dtype`
 * (RS_F32 | RS_BF16); per-channel vectors, parameter gradients, logits and workspaces are always fp32.
 * ---------------------------------------------------------------------------------------------------------- */

/* rs_conv2d_fwd with bf16 sources / weights (KRSC bf16) / residual / relu_mask / output; scale, shift fp32.
 * stem = 1 is not supported here.  Same fused gather (ups, concat) and epilogue. */
int rs_conv2d_fwd_bf16(const rs_conv_desc* d, const rs_bf16* src1, const rs_bf16* src2, const rs_bf16* weight,
                       const float* scale, const float* shift, const rs_bf16* residual, const rs_bf16* relu_mask,
                       rs_bf16* out, rs_stream_t stream);
int rs_conv2d_tile_bf16(const rs_conv_desc* d);
/* Tile index (rs_conv2d_tile_name order) and K-chunk row size (64 | 128 bytes) the dispatcher picks for `d` with `es`-byte
 * activations, direct (phase4 = 0) or phase form: kernel names in reports then map 1:1 to the launched symbols. */
int rs_conv2d_config(const rs_conv_desc* d, int es, int form, int* tile, int* rowb);
/* (`form`: bit 0 = phase form; bit 1 = a launch with a fused epilogue beyond scale / shift / residual / ReLU -- BatchNorm statistics,
 * a ReLU mask, two destinations -- which never takes the plain-epilogue fp32 1x1 kernel.) */
/* Dispatcher override for the parity tests and A/B measurements (process-global; not a tuning API for callers): force the
 * tile (index as above; honoured for every launch that tile can run, -1 = the measured heuristics) and the K-chunk row
 * size (64 | 128; 0 = heuristics).  rs_conv2d_config reports what a launch will then use, so a test can assert that the
 * symbol it means to cover is the one that ran.  The reference has no counterpart: cuDNN's autotuner
 * (torch.backends.cudnn.benchmark, tools/train.py:72-73) is its equivalent of the heuristics this overrides.
 * Tile 8 = the bf16 HALO-ONCE forms (3x3 / stride 1 / pad 1; DecoderBlock phase form; its 4x4 / stride-2 data gradient):
 * the block's rows are a 2-D patch of the output grid (8 x 32 pixels; rowb = 64: 16 x 32 pixels with 32-channel chunks),
 * its source halo is copied to LDS once per channel chunk and every filter tap reads it at a row offset.  rs_conv2d_config
 * reports them as tile 8 with the N tile (128 | 64, + 0x1000 for the 512-pixel patch) in *rowb. */
int rs_conv2d_set_tuning(int tile, int rowb);
/* The library's measurement / A-B switches by name (process-global; robosat_amd/csrc/knobs.hip lists them with the environment
 * variable that seeds each ONCE, at the first use: "conv1x1_ew" <- RS_CONV1X1_EW, "conv_halo" <- RS_CONV_HALO,
 * "wgrad_f32_phase" <- RS_WGRAD_F32_PHASE, ...).  No dispatcher reads the environment per launch: a workspace query and the
 * launch it sizes always see the same settings.  rs_set_knob returns 0, or RS_EINVAL for an unknown name; rs_get_knob writes
 * the current value.  No reference counterpart (torch.backends.cudnn.* flags are the closest). */
int rs_set_knob(const char* name, int value);
int rs_get_knob(const char* name, int* value);
const char* rs_conv2d_tile_name_bf16(int tile);

/* rs_conv2d_wgrad with bf16 dy / sources; dw is fp32 KRSC (the optimizer's master gradient). */
long rs_conv2d_wgrad_bf16_workspace_bytes(const rs_conv_desc* d);
int rs_conv2d_wgrad_bf16_form(const rs_conv_desc* d); /* 0 tap-per-block, 1 all-taps thin kernel, 2 phase form (4/9 MACs) */
int rs_conv2d_wgrad_bf16_tile(const rs_conv_desc* d); /* (couts << 16) | (cins of a second per-source launch << 8) | cins of the tile; 0 = thin kernel */
int rs_conv2d_wgrad_bf16(const rs_conv_desc* d, const rs_bf16* dy, const rs_bf16* src1, const rs_bf16* src2, float* dw,
                         void* workspace, rs_stream_t stream);

/* fp32 master weights -> bf16 compute copies: plain cast (KRSC stays KRSC), and the data-gradient packing of
 * rs_pack_dgrad_weight with a bf16 result. */
int rs_cast_f32_to_bf16(const float* src, rs_bf16* dst, long n, rs_stream_t stream);
/* dst = (float)src * scale.  The gradient exchange of robosat/tools/train.py:69 (DataParallel's reduce-add of 149.4 MB
 * fp32 onto device 0) done in bf16 on the wire (`[model] grad_dtype = "bf16"`, 74.7 MB): the bucket is cast with
 * rs_cast_f32_to_bf16, summed by RCCL, and comes back through this call with scale = 1 / world into the fp32 arena the
 * optimizer reads.  src / dst 16-byte aligned. */
int rs_cast_bf16_to_f32_scaled(const rs_bf16* src, float* dst, long n, float scale, rs_stream_t stream);
/* dst = bf16(src * scale): the way OUT of the same exchange with scale = 1 / world, so that every rank puts an already
 * averaged contribution on the wire and the ring's bf16 partial sums stay at the magnitude of one gradient (summing WORLD
 * unscaled values and dividing afterwards loses mantissa and overflows earlier).  src / dst 16-byte aligned. */
int rs_cast_f32_to_bf16_scaled(const float* src, rs_bf16* dst, long n, float scale, rs_stream_t stream);
int rs_pack_dgrad_weight_bf16(const float* w_krsc, rs_bf16* out, int Cout, int kh, int kw, int Cin, rs_stream_t stream);

/* The 7x7/2 stem (resnet.conv1, unet.py:122) in bf16: x NHWC4 bf16 [N][H][W][4] (rs_nchw_to_nhwc4_bf16), weights packed
 * bf16 [64][7][8][4] (rs_pack_stem_weight_bf16 from the fp32 KRSC master), out [N][H/2][W/2][64] bf16 with the optional
 * eval-BatchNorm scale/shift + ReLU epilogue; rs_stem_conv_wgrad_bf16 returns the PACKED fp32 gradient [64][7][8][4]
 * (-> rs_unpack_stem_weight).  H, W multiples of 32.  One block owns an output patch and all 49 taps (stem_bf16.hip). */
int rs_nchw_to_nhwc4_bf16(const float* x, rs_bf16* y, int N, int C, int H, int W, rs_stream_t stream);
int rs_pack_stem_weight_bf16(const float* w_krsc, rs_bf16* packed, int Cout, int kh, int kw, int Cin, rs_stream_t stream);
int rs_stem_conv_fwd_bf16(const rs_bf16* x, const rs_bf16* w_packed, const float* scale, const float* shift, rs_bf16* out,
                          int N, int H, int W, int relu, rs_stream_t stream);
long rs_stem_conv_wgrad_bf16_workspace_bytes(int N, int H, int W);
int rs_stem_conv_wgrad_bf16(const rs_bf16* dy, const rs_bf16* x, float* dw_packed, int N, int H, int W, void* workspace,
                            rs_stream_t stream);

int rs_maxpool2d_fwd_dt(const void* x, int x_dtype, void* y, int y_dtype, uint8_t* argmax, int N, int H, int W, int C, int k,
                        int stride, int pad, int Ho, int Wo, rs_stream_t stream);
int rs_maxpool2d_bwd_dt(const void* dy, int dy_dtype, const uint8_t* argmax, void* dx, int dx_dtype, int N, int H, int W,
                        int C, int k, int stride, int pad, int Ho, int Wo, int accumulate, rs_stream_t stream);
int rs_final_conv1x1_dt(const void* x, int x_dtype, const float* w, const float* bias, float* out, int N, int H, int W,
                        int Cin, int C, int softmax, rs_stream_t stream);
int rs_final_conv1x1_bwd_dt(const void* x, const float* w, const float* dlogits, void* dx, float* dw, float* db, int dtype,
                            int N, int H, int W, int Cin, int C, int relu_mask, void* workspace, rs_stream_t stream);
int rs_bn_train_stats_dt(const void* y, int dtype, long M, int C, float eps, float momentum, const float* gamma,
                         const float* beta, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                         float* running_var, long long* num_batches_tracked, void* workspace, rs_stream_t stream);
int rs_bn_apply_dt(const void* y, const float* scale, const float* shift, const void* residual, void* out, int dtype, long M,
                   int C, int relu, rs_stream_t stream);
int rs_bn_bwd_dt(const void* dz, const void* zmask, const void* y, const float* mean, const float* invstd, const float* gamma,
                 void* dy, void* dmasked, float* dgamma, float* dbeta, int dtype, long M, int C, void* workspace,
                 rs_stream_t stream);
int rs_upsample2x_bwd_dt(const void* dup, void* d1, void* d2, const void* mask1, const void* mask2, int dtype, int N, int H,
                         int W, int C1, int C2, int accumulate1, rs_stream_t stream);

/* Train-mode convolution that also produces the BatchNorm forward statistics of its output (the conv -> bn pairs of
 * torchvision's Bottleneck, unet.py:127-130): out = conv(gather(src1|src2)) with NO epilogue, plus per-M-tile partial
 * sums stats_partial[tile][0][co] = sum, [tile][1][co] = sum of squares of the values as stored (tiles =
 * rs_conv2d_bnstats_rows(d)); rs_bn_finalize_stats turns them into what rs_bn_train_stats returns, saving that
 * kernel's read pass over the activation.  Not for the stem. */
long rs_conv2d_bnstats_rows(const rs_conv_desc* d);
/* ... for activations of `dtype` (RS_F32 | RS_BF16): the bf16 halo-once forms of the kernel tile the output into 8 x 32
 * pixel patches (one partial row per patch) where they run; rs_conv2d_bnstats_rows is the fp32 answer. */
long rs_conv2d_bnstats_rows_dt(const rs_conv_desc* d, int dtype);
int rs_conv2d_fwd_bnstats_dt(const rs_conv_desc* d, int dtype, const void* src1, const void* src2, const void* weight,
                             void* out, float* stats_partial, rs_stream_t stream);
int rs_bn_finalize_stats(const float* partial, long rows, long M, int C, float eps, float momentum, const float* gamma,
                         const float* beta, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                         float* running_var, long long* num_batches_tracked, void* workspace, rs_stream_t stream);
/* (workspace: 64 * 2 * C doubles, optional -- enables the parallel first-level reduction when rows > 256) */

/* Phase form of DecoderBlock (unet.py:73: conv3x3(pad 1) over interpolate(nearest, x2)): each output parity (oy&1, ox&1)
 * sees a 2x2 convolution on the SOURCE grid whose taps are sums of the 3x3 taps that fall on the same source pixel, so
 * the layer is four 2x2 convolutions with 4/9 of the multiply-adds and no duplicated gathers -- same result up to fp32
 * summation order.  `d` describes the original layer (ups = 1, 3x3, stride 1, pad 1, Ho = 2*Hs); `weight_phase` is
 * [4][Cout][2][2][C1+C2] from rs_pack_phase_weight_dt (fp32 KRSC master in, `dtype` out).  Epilogue as rs_conv2d_fwd. */
int rs_pack_phase_weight_dt(const float* w_krsc, void* out, int dtype, int Cout, int Cin, rs_stream_t stream);
/* The DATA gradient of a 3x3 / stride-2 / pad-1 convolution (torchvision Bottleneck.conv2 of layer2..layer4's first block, under
 * loss.backward(), tools/train.py:186) in the same form (round 6): rs_conv2d_fwd_phase_dt over dy [N][Hs][Ws][Cout] with this pack
 * [4][Cin][2][2][Cout] (fp32 KRSC master [Cout][3][3][Cin] in) gives d input [N][2 Hs][2 Ws][Cin] -- on each input parity the gradient
 * is a 2x2 convolution over dy with at most four of the nine taps (zeros elsewhere): 16 multiply-adds per four pixels (9 in the
 * Winograd form of rs_conv2d_fwd_phase_wino) instead of the 36 of the zero-insertion launch (ups = 2) it replaces. */
int rs_pack_s2_dgrad_phase_weight_dt(const float* w_krsc, void* out, int dtype, int Cout, int Cin, rs_stream_t stream);
int rs_conv2d_fwd_phase_dt(const rs_conv_desc* d, int dtype, const void* src1, const void* src2, const void* weight_phase,
                           const float* scale, const float* shift, const void* residual, const void* relu_mask, void* out,
                           rs_stream_t stream);

/* DecoderBlock (unet.py:63-73) in fp32 as a Winograd F(2x2, 2x2) convolution on the phase form: each parity's 2x2
 * convolution on the source grid produces 2x2 blocks of outputs from 3x3 blocks of inputs with 9 multiplies instead of 16
 * (transforms with 0 / +-1 coefficients only): 9/16 of the phase form's MFMA work, 1/4 of the reference-shape count -- for
 * the fp32 path, whose matrix cores (157 TFLOP/s) are the bottleneck of these layers (conv_wino_f32.hip).  `d` describes the
 * layer like rs_conv2d_fwd_phase_dt (ups = 1, 3x3, pad 1, Ho = 2 Hs); the epilogue is the model's: optional ReLU (d->relu).
 * `u` = [4][9][Cout][C1+C2] from rs_pack_wino_phase_weight(phase pack [4][Cout][2][2][Cin] of rs_pack_phase_weight_dt).
 * rs_conv2d_phase_wino_ok: 0 if this form cannot run `d` (needs >= 4 tiles per image side, channel counts % 16, Cout % 32,
 * 32-bit offsets): use rs_conv2d_fwd_phase_dt; 1 if it can and should; 2 if it can but should not (fewer than 8 tiles per image
 * side: the generic kernel is faster).  The answer depends on the layer's geometry only, never on N: the two forms differ in
 * summation order, and a tile's output must not depend on the size of the batch it travels in.  rs_conv2d_phase_wino_name: the launched instantiation, for reports. */
int rs_conv2d_phase_wino_ok(const rs_conv_desc* d);
const char* rs_conv2d_phase_wino_name(const rs_conv_desc* d);
int rs_pack_wino_phase_weight(const float* w_phase, float* u, int Cout, int Cin, rs_stream_t stream);
int rs_conv2d_fwd_phase_wino(const rs_conv_desc* d, const float* src1, const float* src2, const float* u, float* out,
                             rs_stream_t stream);
/* The DATA gradient of that DecoderBlock (autograd of unet.py:63-73 under tools/train.py:186) in the same Winograd machinery: the
 * 4x4 / stride-2 convolution over dz is four 2x2 correlations over dz's parity planes that accumulate into one source-resolution
 * tile, each of them the forward's parity item on its plane -- 9/16 of rs_conv2d_fwd_dt's multiply-adds on that launch.
 * `d` = the FORWARD layer's descriptor (as for rs_conv2d_fwd_phase_wino); dz [N][2 Hs][2 Ws][Cout] fp32; `u` = [4][9][C1+C2][Cout]
 * from rs_pack_wino_dgrad_weight(wd = [C1+C2][4][4][Cout] of rs_combine_dgrad_phase_weight_dt); result d cat[skip, prev]
 * [N][Hs][Ws][C1+C2] in `out`, or split at channel `csplit` (% 64 == 0) into `out` [..][csplit] and `out2` [..][C1+C2-csplit]
 * (torch.cat's backward fused into the store; out2 = NULL, csplit = 0: one tensor); `mask` / `mask2`: optional tensors shaped like
 * their destination, the result is zeroed where they are <= 0 (the ReLU of the layer that produced skip / prev).
 * rs_conv2d_dgrad_phase_wino_ok: 1 if this form runs the layer's gradient (>= 8 tiles per image side, C1 + C2 a multiple of 64,
 * Cout % 16 == 0, 32-bit offsets) -- geometry only, never N; else use rs_conv2d_fwd_split_dt / rs_conv2d_fwd_dt on the 4x4 form. */
int rs_conv2d_dgrad_phase_wino_ok(const rs_conv_desc* d);
const char* rs_conv2d_dgrad_phase_wino_name(const rs_conv_desc* d);
int rs_pack_wino_dgrad_weight(const float* wd4x4, float* u, int Cin, int Cout, rs_stream_t stream);
int rs_conv2d_dgrad_phase_wino(const rs_conv_desc* d, const float* dz, const float* u, float* out, const float* mask, float* out2,
                               const float* mask2, int csplit, rs_stream_t stream);

/* The stride-1 3x3 / pad-1 convolutions of the fp32 predict path -- Bottleneck.conv2 (torchvision resnet50 via unet.py:94,
 * 122-130) with its eval-mode BatchNorm folded into scale / shift, and dec5's ConvRelu (unet.py:32-44,139) -- as a Winograd
 * F(2x2, 3x3) convolution: 16 multiplies per 2x2 outputs instead of 36 (conv_wino33_f32.hip).  `d`: kh = kw = 3, stride 1,
 * pad 1, ups 0, C2 = 0, Ho = Hs, Wo = Ws; out = relu?(conv(src) * scale + shift), scale / shift optional.  `u` = [16][Cout][Cin]
 * from rs_pack_wino33_weight(KRSC fp32 weight).  rs_conv2d_wino33_ok: 1 if this form runs `d` (H, W >= 15, Cin % 16 == 0 and
 * >= 32, Cout % 16 == 0): a function of the layer's geometry only, never of N (see rs_conv2d_phase_wino_ok). */
int rs_conv2d_wino33_ok(const rs_conv_desc* d);
const char* rs_conv2d_wino33_name(const rs_conv_desc* d);
int rs_pack_wino33_weight(const float* w_krsc, float* u, int Cout, int Cin, rs_stream_t stream);
int rs_conv2d_fwd_wino33(const rs_conv_desc* d, const float* src, const float* u, const float* scale, const float* shift, float* out,
                         rs_stream_t stream);
/* dec5 + self.final (+ softmax / quantise / argmax) of the fp32 predict path in ONE launch (reference unet.py:139-141:
 * `dec5 = self.dec5(dec4); return self.final(dec5)`; tools/predict.py:87-103; tools/serve.py:160-164): rs_conv2d_fwd_wino33 on
 * a 32-cout layer whose block keeps its 32 channels on the CU, applies the 1x1 convolution final_w [C][32] + final_b [C] (C <= 8)
 * and then does what rs_final_conv1x1_dt (mode 0 logits / 1 softmax -> `out` fp32 NCHW [N][C][H][W]),
 * rs_final_conv1x1_quantize_dt (mode 2 -> `qout`, needs `anchors` and `overlap`) or rs_final_conv1x1_argmax_dt (mode 3 ->
 * `qout` [N][H][W]) do with a pixel's logits -- same operations in the same order; the logits themselves differ from the
 * two-launch form by fp32 summation order over the 32 channels.  The layer's own output is never written (537 MB per
 * bs-16 512^2 batch that the two-launch form writes and reads back).  rs_conv2d_wino33_head_ok: 1 if this form runs `d`
 * with C classes (rs_conv2d_wino33_ok, Cout == 32, 1 <= C <= 8). */
int rs_conv2d_wino33_head_ok(const rs_conv_desc* d, int C);
const char* rs_conv2d_wino33_head_name(void);
int rs_conv2d_fwd_wino33_head(const rs_conv_desc* d, const float* src, const float* u, const float* scale, const float* shift,
                              const float* final_w, const float* final_b, int C, int mode, const double* anchors, int overlap,
                              float* out, uint8_t* qout, rs_stream_t stream);
/* The tail of a layer1 Bottleneck and the head of the next one in ONE launch, fp32 eval mode (torchvision Bottleneck.forward under
 * reference unet.py:127; round 6, bottleneck_tail_f32.hip):
 *     out = relu(conv1x1(x; w3) * scale3 + shift3 + identity)     x [M][C1], w3 [Cmid][C1], identity / out [M][Cmid]
 *     z   = relu(conv1x1(out; w1) * scale1 + shift1)              w1 [C2][Cmid], z [M][C2]
 * (the folded BatchNorms of rs_bn_fold).  The second product reads the first one's accumulator registers: `out` is written once and
 * never read back.  C1 = 64, Cmid = 256, C2 = 64 (layer1's widths), M % 32 == 0; anything else: RS_EINVAL (the caller then runs the two
 * rs_conv2d_fwd launches). */
int rs_bottleneck_tail_f32(const float* x, const float* w3, const float* scale3, const float* shift3, const float* identity,
                           const float* w1, const float* scale1, const float* shift1, float* out, float* z, long M, int C1, int Cmid,
                           int C2, rs_stream_t stream);
/* The same kernel's first stage alone: out = [relu](conv1x1(x; w) * scale + shift [+ residual]), C1 = 64 -> Cout = 256, M % 32 == 0 -- layer1's
 * downsample convolution (97 against 109 us at bs 16 / 512^2; with a residual the generic launch is the faster one); `residual` may be NULL.  Same sums in
 * the same order as the chained form's `out`. */
int rs_conv1x1_wave_f32(const float* x, const float* w, const float* scale, const float* shift, const float* residual, int relu, float* out,
                        long M, int C1, int Cout, rs_stream_t stream);
/* The same layers in the TRAIN-mode forward (torchvision Bottleneck.conv2 -> BatchNorm2d under tools/train.py:169, fp32): the raw
 * convolution output plus the per-block partial sums of the BatchNorm statistics (sum y, sum y^2 over the block's pixels, in a fixed
 * order), `stats` [rs_conv2d_wino33_stats_rows(d)][2][Cout] fp32 -- the input of rs_bn_finalize_stats, as rs_conv2d_fwd_bnstats_dt's
 * rows are.  `d` / `u` as for rs_conv2d_fwd_wino33 (rs_conv2d_wino33_ok decides; d->relu ignored). */
long rs_conv2d_wino33_stats_rows(const rs_conv_desc* d);
int rs_conv2d_fwd_wino33_stats(const rs_conv_desc* d, const float* src, const float* u, float* out, float* stats, rs_stream_t stream);
/* The DATA gradient of such a layer in the same form (round 6): `d` describes the gradient's convolution (3x3 / stride 1 / pad 1 over
 * dy; C1 = dy's channels, Cout = the gradient's, Cout % 32 == 0), `u` = rs_pack_wino33_weight of rs_pack_dgrad_weight's filters.  The
 * gradient arrives at a ReLU output -- `mask` (the forward activation) or `mask_bits` (rs_bn_apply_bits_dt's; Cout % 8 == 0) or
 * neither -- and, with `stats` [rs_conv2d_wino33_stats_rows(d)][2][Cout], at a BatchNorm output: the per-block partial sums
 * (sum g, sum g * (bn_y - bn_mean) * bn_invstd) of the masked gradient g, rs_bn_bwd_from_partials_dt's input.  Replaces
 * rs_conv2d_dgrad_bnstats[_bits]_dt / rs_conv2d_fwd(relu_mask) on these layers in fp32 (autograd of torchvision Bottleneck.conv2 and of
 * ConvRelu, robosat/unet.py:28-41, under tools/train.py:186) at 4/9 of the multiply-adds. */
int rs_conv2d_dgrad_wino33(const rs_conv_desc* d, const float* dy, const float* u, const float* mask, const uint8_t* mask_bits,
                           const float* bn_y, const float* bn_mean, const float* bn_invstd, float* out, float* stats,
                           rs_stream_t stream);

/* ... and its data gradient: d loss / d (pre-upsample input) is ONE 4x4 / stride-2 / pad-1 convolution over dz with
 * pre-summed taps (rs_conv2d_fwd[_bf16] with kh = kw = 4 and these weights, [Cin][4][4][Cout]): the gradient lands at
 * the source resolution with 4/9 of the multiply-adds and the 2x2 sum of interpolate's backward already inside;
 * rs_cat_split_bwd_dt then only splits torch.cat's channels, applies the ReLU masks and accumulates (rs_upsample2x_bwd
 * without the 2x2 sum). */
int rs_pack_dgrad_phase_weight_dt(const float* w_krsc, void* out, int dtype, int Cout, int Cin, rs_stream_t stream);
/* bf16 compute copies of MANY convolution weights in one launch -- what rs_cast_f32_to_bf16 + rs_pack_dgrad_weight_bf16
 * produce per tensor, bit for bit (the per-step re-cast of the fp32 master weights after Adam.step, robosat/tools/
 * train.py:188; the reference has no counterpart: it computes in fp32).  `items_dev` is a DEVICE array of n items ordered
 * by tile_begin; item i covers tiles [tile_begin_i, tile_begin_i + taps * ceil(Cin/32) * ceil(Cout/32)); total_tiles = their
 * sum.  cast / dgrad may be NULL. */
typedef struct rs_wprep_item {
  const float* w; /* fp32 KRSC [Cout][taps][Cin] */
  rs_bf16* cast;  /* bf16 KRSC copy, or NULL */
  rs_bf16* dgrad; /* bf16 [Cin][taps, flipped][Cout] (the layout of rs_pack_dgrad_weight_bf16), or NULL */
  int Cout, taps, Cin;
  int tile_begin;
} rs_wprep_item;
int rs_weight_prep_bf16(const rs_wprep_item* items_dev, int n, int total_tiles, rs_stream_t stream);
/* The fp32 twin: every item's `dgrad` points at a FLOAT buffer [Cin][taps, flipped][Cout] (the layout of rs_pack_dgrad_weight),
 * `cast` must be NULL (the fp32 KRSC master is its own compute copy): the data-gradient weights of a whole fp32 training step
 * (tools/train.py:180-188 in the reference's arithmetic) in one launch instead of one per convolution. */
int rs_weight_prep_f32(const rs_wprep_item* items_dev, int n, int total_tiles, rs_stream_t stream);
/* The same weights from the already transposed, tap-flipped fp32 weights of rs_pack_dgrad_weight ([Cin][3][3][Cout]):
 * both sides contiguous along Cout (the one-step pack reads with a 9*Cin stride). */
int rs_combine_dgrad_phase_weight_dt(const float* w_dgrad, void* out, int dtype, int Cout, int Cin, rs_stream_t stream);
/* rs_conv2d_fwd with torch.cat's backward fused into the store: output channels [0, csplit) -> out1 (row stride csplit,
 * zeroed where mask1 <= 0 if mask1 != NULL), [csplit, Cout) -> out2 (row stride Cout - csplit, mask2); csplit must be a
 * multiple of the tile's cout width (64 | 128 for these layers). */
int rs_conv2d_fwd_split_dt(const rs_conv_desc* d, int dtype, const void* src1, const void* weight, void* out1, const void* mask1,
                           void* out2, const void* mask2, int csplit, rs_stream_t stream);
int rs_cat_split_bwd_dt(const void* dcat, void* d1, void* d2, const void* mask1, const void* mask2, int dtype, int N, int H,
                        int W, int C1, int C2, int accumulate1, rs_stream_t stream);
/* out[n][2a][2b][:] += t[n][a][b][:] (t [N][Hs][Ws][C], out [N][Ho][Wo][C], Ho >= 2 Hs - 1; fp32: C % 4 == 0, bf16: C % 8 == 0): the
 * data gradient of a 1x1 / stride-2 convolution -- torchvision Bottleneck.downsample[0] of layer2..layer4 under loss.backward(),
 * tools/train.py:186 -- is rs_conv2d_fwd of its transposed filters on the low-resolution grid, added here onto the even positions
 * of the gradient the tensor already has (round 6; replaces the zero-insertion form, ups = 2, of that launch). */
int rs_scatter_add_stride2_dt(const void* t, void* out, int dtype, int N, int Hs, int Ws, int Ho, int Wo, int C, rs_stream_t stream);

/* The same fusion for BatchNorm's BACKWARD (conv -> bn -> relu read right to left): the data-gradient convolution that
 * produces g = d loss / d z (rs_conv2d_fwd semantics on `dy` with rs_pack_dgrad_weight weights, optional residual,
 * relu_mask = z) also accumulates, per M tile, sum g and sum g * xhat with xhat = (bn_y - bn_mean) * bn_invstd;
 * rs_bn_bwd_from_partials_dt then needs one streaming pass (dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)))
 * instead of rs_bn_bwd's two.  workspace: 64*2*C doubles + 3*C floats. */
int rs_conv2d_dgrad_bnstats_dt(const rs_conv_desc* d, int dtype, const void* dy, const void* weight, const void* residual,
                               const void* relu_mask, const void* bn_y, const float* bn_mean, const float* bn_invstd,
                               void* out, float* stats_partial, rs_stream_t stream);
int rs_bn_bwd_from_partials_dt(const void* g, const void* y, const float* mean, const float* invstd, const float* gamma,
                               void* dy, float* dgamma, float* dbeta, const float* partial, long rows, int dtype, long M,
                               int C, void* workspace, rs_stream_t stream);
/* The ReLU mask as one bit per element (round 2).  autograd's ReLU backward (threshold_backward on the block output of
 * torchvision's Bottleneck, `out = relu(bn3(...) + identity)`) only needs the SIGN of z; the data-gradient epilogue of a
 * bottleneck's conv1 moves four Cout-wide operands (g out, residual gradient, bn_y, z) and is HBM-bound.
 * rs_bn_apply_bits_dt = rs_bn_apply_dt that also writes `bits` (M*C/8 bytes; bit e of byte i: element 8*i + e of `out` is > 0;
 * C must divide 2048, else RS_EINVAL); rs_conv2d_dgrad_bnstats_bits_dt = rs_conv2d_dgrad_bnstats_dt reading those bits
 * instead of z (Cout % 8 == 0).  Same results bit for bit. */
int rs_bn_apply_bits_dt(const void* y, const float* scale, const float* shift, const void* residual, void* out,
                        unsigned char* bits, int dtype, long M, int C, int relu, rs_stream_t stream);
int rs_conv2d_dgrad_bnstats_bits_dt(const rs_conv_desc* d, int dtype, const void* dy, const void* weight, const void* residual,
                                    const unsigned char* relu_mask_bits, const void* bn_y, const float* bn_mean,
                                    const float* bn_invstd, void* out, float* stats_partial, rs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Device-side input / output of `rs predict`: only bytes cross PCIe.
 * ---------------------------------------------------------------------------------------------------------- */

/* Decoded tiles as uint8 HWC [N][H][W][C<=4] -> normalised NHWC4 fp32, i.e. ToTensor + Normalize of the reference's
 * transform chain (tools/predict.py:71: x/255 then (x - mean[c])/std[c], fp32, IEEE division: bit-identical to the
 * host ops) + the stem's layout.  `mean`, `std`: HOST arrays of C floats. */
int rs_u8_to_nhwc4_norm(const uint8_t* img, float* out, const float* mean, const float* std, int N, int H, int W, int C,
                        rs_stream_t stream);

/* self.final + softmax + un-buffer crop + 8-bit quantisation in one pass (tools/predict.py:87,96-103):
 * out[n][y][x][c-1] = uint8(np.digitize(p_c, anchors)) for every non-background class c = 1..C-1 over the central
 * (H-2*overlap) x (W-2*overlap) window, with `anchors` = the 256 float64 values of np.linspace(0, 1, 256) in device memory
 * (1-based bins, 256 wraps to 0 -- the reference's behaviour).  C = 2 is the reference's single-channel PNG payload, byte for
 * byte; C > 2 (the reference asserts a binary model, predict.py:98) stores the same encoding per foreground class,
 * interleaved.  x: NHWC [N][H][W][Cin] of `x_dtype`; w [C][Cin], bias [C]. */
int rs_final_conv1x1_quantize_dt(const void* x, int x_dtype, const float* w, const float* bias, const double* anchors,
                                 uint8_t* out, int N, int H, int W, int Cin, int C, int overlap, rs_stream_t stream);

/* self.final + argmax over the classes -> one byte per pixel: Predictor.segment of `rs serve`
 * (tools/serve.py:160-164: output.argmax(axis=0).astype(np.uint8); first maximum on ties). */
int rs_final_conv1x1_argmax_dt(const void* x, int x_dtype, const float* w, const float* bias, uint8_t* out, int N, int H,
                               int W, int Cin, int C, rs_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * The callers either side of the network, on the device (csrc/postproc.hip)
 * ---------------------------------------------------------------------------------------------------------- */

/* Metrics.add generalised to C classes (robosat/metrics.py:27-41; its Todo at :87-88): counts[actual * C + predicted] += 1
 * per pixel, predicted = argmax over the C scores (first maximum).  counts: int64 [C*C] in device memory, accumulated.
 * For C = 2 the reference's counters are tn = counts[0], "fn" = counts[1], "fp" = counts[2], tp = counts[3].  C <= 8. */
int rs_confusion_matrix(const float* scores, const int64_t* targets, int64_t* counts, int N, int C, int H, int W,
                        rs_stream_t stream);

/* np.bincount over label tiles (robosat/tools/weights.py:41-47): counts256[v] += number of labels equal to v.
 * labels: n bytes in device memory, 16-byte aligned; counts256: int64 [256], accumulated. */
int rs_label_histogram_u8(const uint8_t* labels, long n, int64_t* counts256, rs_stream_t stream);

/* `rs masks` (robosat/tools/masks.py:42-84): K models' quantised probabilities q [K][P][C-1] (the bytes rs predict writes)
 * -> class index per pixel: np.argmax(np.average(probs, axis=0, weights), axis=0) with probs_k = [1 - sum_c f_c, f_1, ..],
 * f_c = anchors[q]; float64, models accumulated in order, first maximum.  weights: K doubles in device memory or NULL. */
int rs_softvote_masks(const uint8_t* q, const double* weights, const double* anchors, uint8_t* out, int K, long P, int C,
                      rs_stream_t stream);

/* Training-set augmentation from a cache of decoded tiles in HBM (robosat/tools/train.py:248-260,
 * robosat/transforms.py:127-221): for batch item n take tile index[n] of images [T][S][S][C] (uint8 HWC) / masks [T][S][S],
 * apply op[n] = f + 2*k (PIL FLIP_LEFT_RIGHT when f, then k x ROTATE_90), ToTensor + Normalize ((v/255 - mean)/std, fp32),
 * and write what the reference's loader yields: images NCHW fp32 [N][C][S][S], masks int64 [N][S][S].
 * mean / std: HOST arrays of C floats.  masks / out_masks may be NULL together. */
int rs_augment_tiles(const uint8_t* images, const uint8_t* masks, const int32_t* index, const int32_t* op, const float* mean,
                     const float* std, float* out_images, int64_t* out_masks, int N, int S, int C, rs_stream_t stream);


#ifdef __cplusplus
}
#endif
#endif /* ROBOSAT_HIP_H */
