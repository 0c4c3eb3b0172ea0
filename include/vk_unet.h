/* vk_unet.h — C ABI of libvkunet.so: the MI355X (gfx950) implementation of the one compute path of
 * ZooMEISTER/vickers-hardness-Unet: ResNet-34-encoder U-Net forward / BCE+Dice loss / backward /
 * AdamW on NHWC tensors, as hand-written HIP kernels.
 *
 * The reference has no FFI layer of its own: the path sits behind the Python nn.Module / loss /
 * optimizer protocol.  Each entry point below names the reference call it replaces:
 *
 *   vk_unet_create / vk_unet_param_info ....... smp.Unet(...) construction   train.py:372-378, infer_pth_gui.py:31-33
 *   vk_unet_forward ........................... model(x)                     train.py:436, 510, 693; infer_pth_gui.py:51
 *   vk_unet_loss .............................. bce(logits,y)+dice(logits,y) train.py:438, 513 (600-601)
 *   vk_unet_backward .......................... loss.backward()              train.py:443, 448
 *   vk_seg_metrics ............................ dice_coef / iou_coef (validate) train.py:230-281, 518-522
 *   vk_comm_* / vk_allreduce_bucket ........... (no reference counterpart: the 8-GPU data-parallel exchange)
 *   vk_adamw_step ............................. optimizer.step()/zero_grad   train.py:428, 449 (606)
 *   vk_amp_check_inf / vk_amp_unscale_check /
 *   vk_adamw_step_amp ......................... GradScaler unscale + inf check + skipped step  train.py:441-445 (610-611)
 *   vk_conv_fwd / vk_conv_wgrad / ... ......... the ATen operators the reference dispatches to
 *                                               (conv2d, batch_norm, relu, max_pool2d, interpolate, cat)
 *   vk_conv_fwd_splitk ........................ the same convolutions at batch 1 (predict_mask / Segmenter.infer)
 *   vk_letterbox_preprocess ................... letterbox + BGR->RGB + normalise   infer_pth_gui.py:17-24, 46-49;
 *                                               ui_infer_quadrilateral.py:197-216, 662-678; ui_infer_rectangle.py:225-245, 520-535
 *   vk_letterbox_postprocess_mask ............. sigmoid, threshold, un-letterbox  infer_pth_gui.py:26-29, 50-53
 *   vk_letterbox_postprocess_prob ............. sigmoid, un-letterbox, clip       ui_infer_quadrilateral.py:219-231, 705-711
 *   vk_geom_minarearect ....................... postprocess_minarearect_multi      ui_infer_rectangle.py:291-381
 *   vk_geom_quadrilateral ..................... postprocess_minarearect_multi + robust_quadrilateral_from_contour  ui_infer_quadrilateral.py:262-530
 *   vk_letterbox_u8 / _mask_u8 / vk_augment_batch  VickersDataset.__getitem__ + albumentations pipeline  train.py:67-113, 173-200
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary.
 *   - every device buffer is owned by the caller (PyTorch's allocator in the Python host);
 *     the library allocates nothing on the device and keeps no pointer after a call returns,
 *     except those bound to a vk_unet handle by vk_unet_bind (valid until rebind / destroy).
 *   - all work is enqueued on the hipStream_t passed in (void* stream); no hidden synchronisation.
 *   - return value: 0 = success, <0 = argument/shape error found on the host (VK_ERR_*),
 *     >0 = hipError_t from a launch.  vk_last_error_string() gives the text.  Nothing throws.
 *   - activations are NHWC; weights are KRSC ([K_out][R][S][C_in], i.e. torch channels_last of OIHW).
 */
#ifndef VK_UNET_H
#define VK_UNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VK_ABI_VERSION 1
/* BatchNorm partial sums are spread over this many replicas ([R][2][K] doubles) so that tens of thousands of
 * workgroups do not serialise on the same 2K addresses; vk_bn_finalize adds the replicas up. */
#define VK_STATS_REPLICAS 32

typedef enum { VK_F32 = 0, VK_BF16 = 1, VK_F16 = 2 } vk_dtype;

enum { VK_OK = 0, VK_ERR_ARG = -1, VK_ERR_STATE = -2, VK_ERR_UNSUPPORTED = -3 };

int vk_version(void);
const char* vk_last_error_string(void);
/* 1 if the code object for gfx950 is present in this build (always true for a product build) */
int vk_has_gfx950_code(void);
/* Measurement aid (bench.py): a bare loop of independent v_mfma_f32_16x16x32_bf16 on register operands, `waves_per_simd` waves on
 * every SIMD of the chip, `iters` x 8 MFMAs each — what the matrix pipes deliver at the clock the chip sustains under that load.
 * FLOPs issued = *flops_out (when not null); time it with events on `stream`.  sink: 4 floats of device scratch. */
int vk_probe_mfma_rate(int iters, int waves_per_simd, float* sink, double* flops_out, void* stream);
/* Data-parallel runs: while a communication library's kernels hold compute units, launches sized to exactly one workgroup per CU
 * (the weight-gradient kernels: <= 256 persistent workgroups that split the pixel tiles among themselves) fall into a second round.
 * n > 0 sizes those grids for (CUs - n) instead; 0 (default) restores the full chip.  Process-wide, read per launch; the Python host
 * sets it only for the backward stages that overlap a collective (parallel.py).  Returns the previous value. */
int vk_set_reserved_cus(int n);
/* Measurement aid (tests/diag/cu_hold.py): occupy compute units the way a communication library's channel kernels do while a step
 * runs on another stream.  Launches `workgroups` workgroups of `threads` threads (64..1024) with `lds_bytes` of LDS each
 * (lds_bytes = 163840 takes a whole CU: no tile kernel fits beside it; a small value lets other workgroups co-reside and only
 * competes for issue slots / memory queues) that spin for `microseconds` of wall time (s_memrealtime deadline: every wave exits by
 * itself) and, when `traffic` != 0, keep streaming 16-byte loads from it (traffic_bytes, a power of two >= 64 KiB) meanwhile.
 * sink: 4 floats of device scratch. */
int vk_debug_hold_cus(int workgroups, int threads, int lds_bytes, int microseconds, const void* traffic, size_t traffic_bytes,
                      float* sink, void* stream);

/* Per-launch timing (off by default): while enabled every launch made through this library is bracketed
 * by two hipEvents on its own stream.  vk_prof_collect synchronises those events and writes one line per
 * kernel family into buf: "tag count total_ms total_algorithmic_flops total_algorithmic_bytes\n";
 * returns the number of bytes written (or <0).  Used by bench.py for the live roofline figure. */
/* diagnostic builds only (make stamp): buffer receiving per-wave s_memtime segment totals of the tile kernels */
int vk_debug_set_stamp_buffer(void* device_buffer);
int vk_prof_enable(int on);
int vk_prof_collect(char* buf, size_t buflen);

/* ------------------------------------------------------------------------------------------------
 * Operator level (used by the engine below and by the per-kernel parity tests)
 * ---------------------------------------------------------------------------------------------- */

/* One input source of a convolution gather.  The convolution reads a *virtual* input
 *   V[n][h][w][c] = act( src[n][h >> up][w >> up][c] * scale[c] + shift[c] )
 * i.e. nearest-x2 upsample (up=1), train/eval BatchNorm apply (scale/shift != NULL) and ReLU (relu=1)
 * are fused into the operand load; zero padding is applied AFTER that transform. */
typedef struct {
  const void* ptr;      /* NHWC, dtype of the conv, [N][H>>up][W>>up][C] */
  int C;                /* channels of this source (multiple of 8; 16 allowed) */
  int up;               /* 0 or 1 */
  const float* scale;   /* [C] or NULL (identity) */
  const float* shift;   /* [C] or NULL */
  int relu;             /* apply max(.,0) after the affine */
} vk_src;

typedef struct {
  vk_dtype dtype;       /* element type of activations and packed weights */
  int N, H, W;          /* virtual input spatial size (after the optional upsample) */
  int Ho, Wo;           /* output spatial size */
  int K;                /* output channels */
  int R, S, stride, pad;
  int transposed;       /* 0: y[p] = sum_r V[p*stride - pad + r] w[r]   (forward)
                           1: y[p] = sum_r V[(p + pad - r)/stride] w[r] (data gradient; V = dz) */
  vk_src src0, src1;    /* src1.ptr == NULL when there is no channel concat; C = src0.C + src1.C */
} vk_conv_desc;

/* y = conv(V, w).  w: [K][R][S][C] of `dtype`.  y: [N][Ho][Wo][K] of `dtype`.
 * If K >= split_k1 > 0 the output channels [split_k1, K) go to y1 (leading dim K - split_k1) and
 * [0, split_k1) to y (leading dim split_k1): the two halves of a concat gradient.
 * accumulate: y (+)= result.  stats: optional double[VK_STATS_REPLICAS][2][K] receiving (spread over the replicas) sum / sum of squares over
 * N*Ho*Wo of the stored (rounded) outputs — train-mode BatchNorm partials (must be zeroed by caller). */
int vk_conv_fwd(const vk_conv_desc* d, const void* w, void* y, void* y1, int split_k1, int accumulate,
                double* stats, void* stream);

/* Data gradient of a decoder conv1 with the nearest-x2 upsample backward fused in: like vk_conv_fwd with
 * transposed=1, but the first output part (channels [0, split_k1), or all K when split_k1 == 0) is summed over
 * 2x2 pixel groups and written to y_half [N][Ho/2][Wo/2][.]; the skip part still goes to y1 at full resolution.
 * Returns VK_ERR_UNSUPPORTED for shapes outside the 3x3 stride-1 tile kernels (caller then uses
 * vk_conv_fwd + vk_upsample2x_bwd). */
int vk_conv_dgrad_pool2(const vk_conv_desc* d, const void* w, void* y_half, void* y1, int split_k1, int accumulate,
                        void* stream);

/* The 3x3 stride-1 tile kernels read their weights in the "halo pack": [red/CK][9 taps][rows][CK] (CK = 64 bytes of
 * channels), 16-byte pieces pre-swizzled for the LDS image, so that a pipeline stage is three linear 1-KiB-per-wave copies.
 * vk_halo_pack builds it from plain [rows][3][3][red] weights of `dtype` (forward: rows = K, red = C; data gradient: the
 * transposed weights, rows = C, red = K).  vk_conv_uses_halo_pack tells whether a descriptor runs on those kernels
 * (then vk_conv_fwd_packed / vk_conv_dgrad_fused / vk_conv_dgrad_pool2 expect the pack; layers with C == 16 in a 16-bit
 * type keep plain weights).  vk_conv_fwd with plain weights always works (tap-by-tap kernel) but is slower. */
int vk_halo_pack(vk_dtype dtype, int rows, int red, const void* src, void* dst, void* stream);
int vk_conv_uses_halo_pack(const vk_conv_desc* d);
int vk_conv_fwd_packed(const vk_conv_desc* d, const void* w_halo, void* y, void* y1, int split_k1, int accumulate, double* stats,
                       void* stream);

/* vk_conv_fwd_packed without statistics for SMALL grids (batch-1 inference, predict_mask / Segmenter.infer:
 * infer_pth_gui.py:51, ui_infer_quadrilateral.py:705-707): when the layer has fewer than 128 output tiles the channel
 * reduction is cut into up to 32 slices that run as separate workgroups, write fp32 partial tiles into `workspace`
 * ([slices][N*H*W][K] floats; fewer slices when it is smaller, none when NULL) and are added up in slice order by a second
 * launch — same result on every run.  Larger grids run exactly as vk_conv_fwd_packed. */
#define VK_SPLITK_WORKSPACE_BYTES (32u << 20)
int vk_conv_fwd_splitk(const vk_conv_desc* d, const void* w_halo, void* y, void* workspace, size_t workspace_bytes, void* stream);

/* BatchNorm+ReLU backward reduce fused into the kernel that produces the gradient: with y the gradient w.r.t. the
 * activated tensor relu(z*scale+shift), the kernel stores g = y * [z*scale+shift > 0] instead of y and adds sum(g),
 * sum(g*z) into sums [VK_STATS_REPLICAS][2][C] (caller zeroes).  Phase 2 is vk_bn_bwd_apply_fused with mask_mode 0. */
typedef struct {
  const void* z;        /* raw conv output the gradient belongs to, same shape as the gradient */
  const float* scale;
  const float* shift;
  double* sums;
  /* r04 — the tail of a residual block, out = relu(bn2(z) + shortcut) (reference: torchvision BasicBlock.forward behind train.py:436):
   * mask != NULL: the ReLU mask is [mask > 0] (mask = the block's stored output, same shape and type as the gradient) instead of
   * [z*scale+shift > 0]; scale / shift may then be NULL.  accumulate != 0: the convolution result is ADDED to the previous content
   * of y first (the shortcut gradient that is already there), then masked, summed and stored.  Together they move the BatchNorm
   * backward reduce of block b into the data gradient of block b+1's conv1, which produces block b's output gradient. */
  const void* mask;
  int accumulate;
} vk_bnr;

/* Data gradient with optional fusions on the first output part (channels [0, split_k1), or all of them):
 * pool2 (see vk_conv_dgrad_pool2) and/or bnr.  VK_ERR_UNSUPPORTED outside the 3x3 stride-1 tile kernels. */
int vk_conv_dgrad_fused(const vk_conv_desc* d, const void* w, void* y, void* y1, int split_k1, int pool2, const vk_bnr* bnr,
                        void* stream);

/* Stem: 7x7 stride-2 pad-3 convolution of x4 [N][H][W][4] (channel 3 is zero padding) with packed
 * weights wp [64][7][32] (tap row r, 8 columns x 4 channels, zero padded). */
int vk_stem_fwd(vk_dtype dtype, int N, int H, int W, const void* x4, const void* wp, void* y, double* stats,
                void* stream);

/* dw[K][R][S][C] (fp32, +=; caller zeroes once per step) = sum_pixels dz[n][p][q][k] * V[n][p*stride-pad+r][..][c].
 * workspace (optional, 16-byte aligned, VK_WGRAD_WORKSPACE_BYTES is always enough; dw 16-byte aligned): when given, every
 * kernel (3x3 stride-1 tile kernel, tap-by-tap kernel for stride 2 / 1x1, stem) writes per-split partial results there and
 * a second launch adds them up in a fixed order (bit-reproducible gradients); otherwise fp32 atomics. */
#define VK_WGRAD_WORKSPACE_BYTES (64u << 20)
int vk_conv_wgrad(const vk_conv_desc* d, const void* dz, float* dw, void* workspace, size_t workspace_bytes, void* stream);
/* Weight gradients of SEVERAL layers in ONE launch (r03): every layer must be of the class vk_conv_wgrad runs on its 64 x 64 tap-split
 * tile kernel — 16-bit, 3x3 stride 1 pad 1, K >= 64, every source a multiple of 64 channels (vk_conv_wgrad_batch_supports says so).
 * The (layer, output tile, 128-pixel tile) units of all layers are cut into `workgroups` equal ranges (0: one per usable CU); ranges
 * that cross an output tile leave partial tiles in `workspace`, which a second kernel adds in range order: same bits on every run.
 * At most 40 layers.  Replaces the per-layer launches of a backward stage (the reference: autograd's per-layer convolution-backward-weight calls behind
 * train.py:443 / :448).  `tables`: device scratch of VK_WGRAD_BATCH_TABLE_BYTES, filled (synchronously) by the call; dw[l] += result.
 * The call waits for `stream` (hipStreamSynchronize) before it refills `tables`, so the same tables / workspace may be passed to
 * consecutive calls; buffers shared with work on OTHER streams are the caller's to order. */
#define VK_WGRAD_BATCH_TABLE_BYTES (128u << 10)
int vk_conv_wgrad_batch_supports(const vk_conv_desc* d);
int vk_conv_wgrad_batch(const vk_conv_desc* descs, const void* const* dz, float* const* dw, int n, int workgroups, void* tables,
                        size_t tables_bytes, void* workspace, size_t workspace_bytes, void* stream);
int vk_stem_wgrad(vk_dtype dtype, int N, int H, int W, const void* x4, const void* dz, float* dw_krsc3, void* workspace,
                  size_t workspace_bytes, void* stream);
/* The same with the stem BatchNorm's backward apply folded in (16-bit types; VK_ERR_UNSUPPORTED otherwise: run vk_bn_bwd_apply +
 * vk_stem_wgrad): g = the masked upstream gradient (what vk_maxpool_bwd_bn_reduce stored), z = the stem convolution's output,
 * coef_abc = [3][64] from vk_bn_bwd_coeffs; the kernel forms dz = a*g + b*z + c while it stages its operand — the stem has no data
 * gradient, so dz has no other reader and the 3-tensor apply pass (reference: autograd's batch_norm backward node behind
 * train.py:448 -> encoder.bn1) disappears. */
int vk_stem_wgrad_bn(vk_dtype dtype, int N, int H, int W, const void* x4, const void* g, const void* z, const float* coef_abc,
                     float* dw_krsc3, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The steps either side of model(x) in the inference wrappers, one fused pass each.
 * The caller decides the geometry (the reference has two conventions: image in the top-left corner with
 * scale = min(S/h, S/w), infer_pth_gui.py:17-24; image centred with scale = min(S/max(h,w), 1),
 * ui_infer_quadrilateral.py:197-216) and passes it in the descriptor; the library does the pixel work with
 * cv2.resize's arithmetic (8-bit INTER_LINEAR in 11-bit fixed point, float INTER_LINEAR, INTER_NEAREST).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int h, w;             /* original image size */
  int src_stride;       /* bytes per row of the original BGR image (>= 3*w); unused by the post-processing calls */
  int size;             /* side S of the square network input / logit map */
  int nh, nw;           /* size of the resized image inside the square */
  int top, left;        /* its position inside the square */
  int pad_value;        /* border value 0..255, applied before the normalisation */
} vk_letterbox_desc;

/* uint8 BGR [h][w][3] (device) -> float32 [3][S][S] RGB planes, ((v/255) - mean) / std with the ImageNet constants */
int vk_letterbox_preprocess(const vk_letterbox_desc* d, const uint8_t* bgr, float* x_nchw, void* stream);
/* logits [S][S] -> uint8 [h][w] in {0,255}: (sigmoid >= thresh), crop, INTER_NEAREST back to the original size */
int vk_letterbox_postprocess_mask(const vk_letterbox_desc* d, const float* logits, float thresh, uint8_t* mask_hw, void* stream);
/* logits [S][S] -> float32 [h][w] in [0,1]: sigmoid, crop, INTER_LINEAR back to the original size (copy when equal), clip */
int vk_letterbox_postprocess_prob(const vk_letterbox_desc* d, const float* logits, float* prob_hw, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Geometry post-processing of batched probability maps: what the reference's GUIs do with
 * Segmenter.infer's output to obtain the indentation diagonals — ui_infer_rectangle.py:291-381 postprocess_minarearect_multi
 * (steps 1-3 are shared by ui_infer_quadrilateral.py:446-490): (prob >= bin_thresh) -> morphologyEx OPEN, CLOSE with the
 * MORPH_ELLIPSE k x k element -> 8-connected components, area >= min_area -> per component the minimum-area enclosing rectangle
 * (cv2.minAreaRect + boxPoints + astype(int32)) -> the two diagonals.  All steps run on the device for `batch` maps of one size.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int h, w;             /* size of every probability map (h <= 4096) */
  float bin_thresh;     /* BIN_THRESH: 0.50 (rectangle GUI) / 0.45 (quadrilateral GUI); compared in float32 */
  int morph_kernel;     /* MORPH_KERNEL: odd 1..7, side of the MORPH_ELLIPSE element (3 = cross); 1 = no morphology */
  int open_iter;        /* OPEN_ITER */
  int close_iter;       /* CLOSE_ITER */
  int min_area;         /* max(200, int(MIN_AREA_FRAC * h * w)) in the reference (ui_infer_rectangle.py:322) */
  int max_components;   /* capacity of the per-map detection list (further kept components stay in `clean` and in `counts`) */
} vk_geom_desc;

typedef struct {
  int label;            /* connected-component id as cv2 / scipy number them: 1 + raster rank of the component's first pixel */
  int area;             /* pixels */
  int box[8];           /* x0,y0 .. x3,y3: rectangle corners truncated to int32 (order around the rectangle; not TL/TR/BR/BL) */
  float cx, cy;         /* rectangle centre */
  float rw, rh;         /* side lengths: along the supporting hull edge / across it */
  float ux, uy;         /* unit direction of that edge */
  int hull_n;           /* convex-hull vertices of the component */
  int reserved;
  double d1, d2, d_mean;/* diagonals of the int32 box (longest pair first) and their mean, float64 as numpy computes them */
} vk_geom_det;

int64_t vk_geom_workspace_bytes(const vk_geom_desc* d, int batch);      /* < 0: bad descriptor */
/* prob: float32 [batch][h][w] in [0,1] (device).  clean: uint8 [batch][h][w] in {0,255}.  dets: [batch][max_components], filled in
 * label order (the host sorts by area like ui_infer_rectangle.py:379).  counts: int32 [batch] = kept components per map. */
int vk_geom_minarearect(const vk_geom_desc* d, int batch, const float* prob, uint8_t* clean, vk_geom_det* dets, int* counts,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The newer GUI's post-processing (ui_infer_quadrilateral.py:423-530 `postprocess_minarearect_multi` with its helpers :262-420):
 * steps 1-3 as above (bin_thresh 0.45 there), then per kept component: dilate by the (2 fit_outset_px + 1)^2 MORPH_ELLIPSE element
 * (fit only: `clean` and `area` are untouched), external border (cv2.findContours RETR_EXTERNAL / CHAIN_APPROX_SIMPLE), convex hull,
 * cv2.approxPolyDP epsilon bisection to exactly four vertices on both polygons, the sub-sampling and extreme-point fall-backs,
 * (quality, area) ranking of the candidates, clockwise ordering, int32 corners and the two diagonals. */
typedef struct {
  int label;            /* as vk_geom_det */
  int area;             /* pixels of the component (before the fit dilation) */
  int box[8];           /* x0,y0 .. x3,y3: the quadrilateral, ordered by _order_quad_cw (:266-277), starting at its top-most corner */
  float cx, cy;         /* mean of the four corners */
  int valid;            /* 0: no quadrilateral was found (the reference drops such a component from its list) */
  int branch;           /* which step produced the candidates: 1 epsilon bisection, 2 four consecutive vertices of the 1 % polygon, 3 extreme points */
  int n_candidates;
  int contour_n;        /* points of the CHAIN_APPROX_SIMPLE border */
  int hull_n;           /* convex-hull vertices of the dilated component */
  int flags;            /* bit 0: border longer than the 16,384-point buffer (hull candidate only); bit 1: hull > 4,096 vertices; bit 2: trace aborted */
  double quality;       /* _quad_quality of the chosen candidate */
  double d1, d2, d_mean;
} vk_geom_quad;

/* as vk_geom_minarearect (same workspace size: vk_geom_workspace_bytes); fit_outset_px 0..3 (reference default 2) */
int vk_geom_quadrilateral(const vk_geom_desc* d, int fit_outset_px, int batch, const float* prob, uint8_t* clean, vk_geom_quad* dets,
                          int* counts, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training-time augmentation on the device: VickersDataset.__getitem__ (train.py:173-200) with the
 * albumentations pipeline of train.py:67-113.  The dataset is letterboxed ONCE into uint8 tensors that stay in HBM
 * (vk_letterbox_u8 / vk_letterbox_mask_u8: LongestMaxSize + PadIfNeeded, train.py:70-75, geometry in the descriptor); every
 * step vk_augment_batch turns n dataset items + n sets of random draws (made by the host, as albumentations makes them) into
 * the network input x float32 [n][3][S][S] and target y float32 [n][1][S][S] in one fused pass.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int d4;               /* OneOf(flips, rot90), train.py:81-85: 0 none, 1 HorizontalFlip, 2 VerticalFlip, 3 + k = np.rot90(k), k = 0..3 */
  int rotate;           /* Rotate(limit=180, BORDER_CONSTANT), train.py:89: 1 = applied */
  float cos_a, sin_a;   /* of the angle (counter-clockwise, cv2.getRotationMatrix2D about (S/2 - 0.5, S/2 - 0.5)) */
  int photo;            /* OneOf, train.py:96-100: 0 none, 1 RandomBrightnessContrast, 2 CLAHE, 3 GaussianBlur */
  float alpha, beta;    /* RandomBrightnessContrast: v' = trunc(clip(v * alpha + beta * 255, 0, 255)) */
  int blur_ksize;       /* GaussianBlur: 3 or 5 (sigma 0 -> cv2's binomial kernels), BORDER_REFLECT_101 */
  float noise_scale;    /* GaussNoise, train.py:104: sigma / 65536 on the 0..255 scale; 0 = not applied */
  uint32_t noise_seed;  /* seed of the counter-based noise field of this sample */
  int clahe_limit;      /* CLAHE, train.py:98: OpenCV's integer clip limit max(1, int(clip * (S/8)^2 / 256)), clip ~ U(1, 2) */
} vk_aug_params;

/* CLAHE works on the L channel of an 8-bit RGB <-> L*a*b* conversion done in integer fixed point through three tables the
 * host builds once (vickers-hardness-unet_amd/augment.py: color_tables) and keeps on the device as int32 [VK_AUG_TABLE_INTS]:
 * LIN [256] sRGB decode x 4096 | FT [4097] Lab f(t / 4096) x 32768 | ENC [4097] sRGB encode of lin / 4096. */
#define VK_AUG_TABLE_INTS (256 + 4097 + 4097)
/* bytes of scratch vk_augment_batch needs for n samples when any of them draws CLAHE: per sample the geometric result as
 * L, a, b, mask uint8 [S][S][4] and the 8 x 8 tile LUTs uint8 [64][256] */
size_t vk_augment_workspace_bytes(int n, int size);

/* uint8 BGR [h][w][3] -> uint8 RGB [S][S][3]: cv2.resize(INTER_LINEAR) to nh x nw at (top, left), constant border */
int vk_letterbox_u8(const vk_letterbox_desc* d, const uint8_t* bgr, uint8_t* rgb_sq, void* stream);
/* uint8 mask [h][w] (row stride d->src_stride bytes) -> {0,1} [S][S]: (m > 0), cv2.resize(INTER_NEAREST), border 0 */
int vk_letterbox_mask_u8(const vk_letterbox_desc* d, const uint8_t* mask_hw, uint8_t* mask_sq, void* stream);
/* images_rgb uint8 [n_items][S][S][3], masks uint8 [n_items][S][S] in {0,1} (device); index_dev int32 [n] (device): the dataset
 * item of every sample; params_host [n]: validated on the host, then copied to params_dev (device scratch, n * sizeof).
 * color_tables (device, int32 [VK_AUG_TABLE_INTS]) and workspace (device, vk_augment_workspace_bytes(n, size)) are needed only
 * when a sample has photo == 2 and may be null otherwise; CLAHE needs size % 8 == 0 (8 x 8 tiles without padding). */
int vk_augment_batch(int n, int size, int n_items, const uint8_t* images_rgb, const uint8_t* masks, const int* index_dev,
                     const vk_aug_params* params_host, void* params_dev, const int* color_tables, void* workspace,
                     size_t workspace_bytes, float* x, float* y, void* stream);

/* NCHW fp32 [N][3][H][W] -> NHWC4 `dtype` */
int vk_input_transform(vk_dtype dtype, int N, int H, int W, const float* x, void* x4, void* stream);

/* BatchNorm statistics -> per-channel affine.  train=1: from batch sums stats[VK_STATS_REPLICAS][2][C] (count = N*H*W), updates
 * running stats (momentum 0.1, unbiased var) and writes mean/invstd for backward.  train=0: from
 * running stats.  scale = gamma*invstd, shift = beta - mean*scale. */
int vk_bn_finalize(int C, int train, const double* stats, double count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float eps, float momentum, float* scale, float* shift,
                   float* save_mean, float* save_invstd, void* stream);

/* pooled = maxpool3x3s2p1(relu(z*scale+shift)); argmax (uint8 0..8, window scan order) kept for backward */
int vk_bn_relu_maxpool(vk_dtype dtype, int N, int H, int W, int C, const void* z, const float* scale,
                       const float* shift, void* pooled, uint8_t* argmax, void* stream);
/* dy[N][H][W][C] += scatter(dpool) through argmax */
int vk_maxpool_bwd(vk_dtype dtype, int N, int H, int W, int C, const void* dpool, const uint8_t* argmax, void* dy,
                   void* stream);
/* vk_maxpool_bwd and the BatchNorm+ReLU-backward reduce of the tensor under the pool in one pass (stem tail): dy holds the other
 * gradient contributions on entry and g = (dy + maxpool backward) * [relu(z*scale+shift) > 0] on return; sum(g), sum(g*z) are added
 * into sums [VK_STATS_REPLICAS][2][C].  Phase 2 is vk_bn_bwd_apply with mask_mode 0. */
int vk_maxpool_bwd_bn_reduce(vk_dtype dtype, int N, int H, int W, int C, const void* dpool, const uint8_t* argmax, const void* z,
                             const float* scale, const float* shift, void* dy, double* sums, void* stream);

/* out = relu(z*scale+shift + (res*rscale+rshift | res)) — BasicBlock tail */
int vk_bn_add_relu(vk_dtype dtype, size_t pixels, int C, const void* z, const float* scale, const float* shift,
                   const void* res, const float* rscale, const float* rshift, void* out, void* stream);

/* BatchNorm(+ReLU) backward, two phases.  mask_mode 0: none, 1: relu(z*scale+shift) > 0, 2: mask_src > 0.
 * phase 1: sums double[VK_STATS_REPLICAS][2][C] += { sum g, sum g*z } (spread over the replicas),  g = dy * mask.
 * phase 2 (after vk_bn_bwd_coeffs): dz = a*g + b*z + c ; optional g_out (+)= g  (identity shortcut). */
int vk_bn_bwd_reduce(vk_dtype dtype, size_t pixels, int C, const void* dy, const void* z, int mask_mode,
                     const float* scale, const float* shift, const void* mask_src, double* sums, void* stream);
int vk_bn_bwd_coeffs(int C, const double* sums, double count, const float* gamma, const float* save_mean,
                     const float* save_invstd, float* dgamma, float* dbeta, float* coef_abc, void* stream);
int vk_bn_bwd_apply(vk_dtype dtype, size_t pixels, int C, const void* dy, const void* z, int mask_mode,
                    const float* scale, const float* shift, const void* mask_src, const float* coef_abc, void* dz,
                    void* g_out, int g_accumulate, void* stream);
/* phase 2 with vk_bn_bwd_coeffs folded in: coefficients are derived from `sums` inside the kernel and
 * dgamma/dbeta are accumulated by it (one launch less per BatchNorm layer). */
int vk_bn_bwd_apply_fused(vk_dtype dtype, size_t pixels, int C, const void* dy, const void* z, int mask_mode,
                          const float* scale, const float* shift, const void* mask_src, const double* sums, double count,
                          const float* gamma, const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                          void* dz, void* g_out, int g_accumulate, void* stream);

/* d_low[N][H/2][W/2][C] (+)= 2x2 sums of d_up[N][H][W][C] (nearest-x2 upsample backward) */
int vk_upsample2x_bwd(vk_dtype dtype, int N, int H, int W, int C, const void* d_up, void* d_low, int accumulate,
                      void* stream);

/* Segmentation head: 3x3 pad-1 conv C=16 -> 1 with bias on the activated decoder output; fp32 logits. */
int vk_head_fwd(vk_dtype dtype, int N, int H, int W, const vk_src* src, const float* w9x16, const float* bias,
                float* logits, void* stream);
/* Inference only, 16-bit element types (r03): decoder block 4 conv1 -> BN+ReLU -> conv2 -> BN+ReLU -> head over one overlapping tile —
 * smp's DecoderBlock + SegmentationHead behind `model(x)` in eval mode (infer_pth_gui.py:45-53) without the two 512^2 x 16 tensors
 * between them going through HBM.  src: the 32-channel output of decoder block 3 (raw z + its BatchNorm affine, up = 1); w1_pack: conv1
 * weights as vk_halo_pack(16 rows, 32 channels); w2_plain: conv2 weights [16][3][3][16] of the element type; scale / shift: the folded
 * eval BatchNorm affines; head as in vk_head_fwd.  H, W multiples of 16.  Same bits as the three separate calls. */
int vk_dec4_tail_eval(vk_dtype dtype, int N, int H, int W, const vk_src* src, const void* w1_pack, const float* scale1, const float* shift1,
                      const void* w2_plain, const float* scale2, const float* shift2, const float* head_w9x16, const float* head_bias,
                      float* logits, void* stream);
/* workspace (optional, VK_HEAD_WORKSPACE_BYTES is always enough): per-workgroup partial weight gradients that a second launch
 * adds in workgroup order -> dw / dbias are bit-reproducible; NULL: fp32 atomics */
#define VK_HEAD_WORKSPACE_BYTES (1024u * 148u * 4u)
int vk_head_bwd(vk_dtype dtype, int N, int H, int W, const vk_src* src, const float* w9x16, const float* dlogits,
                void* dy, float* dw9x16, float* dbias, void* workspace, size_t workspace_bytes, void* stream);
/* same, with the BatchNorm+ReLU backward reduce of the head's input layer fused into the dy kernel (see vk_bnr) */
int vk_head_bwd_fused(vk_dtype dtype, int N, int H, int W, const vk_src* src, const float* w9x16, const float* dlogits,
                      void* dy, float* dw9x16, float* dbias, const vk_bnr* bnr, void* workspace, size_t workspace_bytes, void* stream);

/* loss = mean BCE-with-logits + binary Dice (smp defaults: batch-global, smooth 0, eps 1e-7).
 * sums: double[8] scratch (zeroed by the call).  loss_out[0] = w_bce*bce + w_dice*dice, [1] = bce, [2] = dice.
 * dlogits (optional) = grad_scale * d(loss_out[0])/dlogits. */
int vk_bce_dice_loss(size_t count, const float* logits, const float* target, double* sums, float* loss_out,
                     float* dlogits, float grad_scale, float w_bce, float w_dice, void* stream);

/* Thresholded Dice / IoU of the reference's validate() (train.py:230-255 `dice_coef`, :259-281 `iou_coef`, :518-522):
 * per image i of `per_image` elements, pred = (p > threshold) as 0/1, I = sum pred*t, P = sum pred, T = sum t;
 * dice_i = (2 I + eps) / (P + T + eps), iou_i = (I + eps) / (P + T - I + eps) in fp32.
 * pred: probabilities, or logits when from_logits != 0 (then p = 1 / (1 + expf(-x)) first).
 * out (device, 2 + 2 n_images floats): [0] mean dice, [1] mean iou, [2 + 2i], [3 + 2i] = image i.
 * workspace: vk_seg_metrics_workspace_bytes(n_images) of 8-byte aligned device scratch; afterwards it holds the
 * fp64 sums {I, P, T} per image.  One pass over both tensors + one single-workgroup launch; bit-reproducible for 0/1 targets. */
size_t vk_seg_metrics_workspace_bytes(int n_images);
int vk_seg_metrics(int n_images, size_t per_image, const float* pred, const float* target, int from_logits, float threshold,
                   float eps, void* workspace, size_t workspace_bytes, float* out, void* stream);

/* AdamW (decoupled decay) over a flat fp32 parameter buffer; optionally emits the 16-bit working copy.
 * inv_scale multiplies the gradient first (GradScaler unscale / data-parallel averaging).
 * found_inf (optional int*): when *found_inf != 0 on the device the step is skipped. */
int vk_adamw_step(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float inv_scale,
                  const int* found_inf, void* lowp_copy, vk_dtype lowp_dtype, void* stream);
/* *found_inf |= any(!isfinite(grad)) */
int vk_amp_check_inf(size_t n, const float* grad, int* found_inf, void* stream);

/* The GradScaler protocol of the reference's CUDA branch (train.py:441-445, 610-611: scaler.scale(loss).backward();
 * scaler.step(optimizer); scaler.update()) without a host round trip.  torch.amp.GradScaler keeps its scale and its
 * found_inf flag as fp32 device scalars and hands them to an optimizer that declares _step_supports_amp_scaling:
 *
 * vk_amp_unscale_check = torch._amp_foreach_non_finite_check_and_unscale_ over the ONE flat gradient buffer:
 *   grad *= *inv_scale (device fp32; NULL or a value of exactly 1 leaves the buffer unwritten), *found_inf = 1.0f if any
 *   element is inf / nan (never cleared here: the caller zeroes it, as GradScaler does).  n % 4 == 0, grad 16-byte aligned.
 * vk_adamw_step_amp = vk_adamw_step with everything step-dependent read on the device: *found_inf != 0 skips the update and
 *   leaves *step_count (int32) as it is — GradScaler does not call optimizer.step() on an overflow; otherwise *step_count is
 *   incremented first and drives the bias corrections, and the gradient is multiplied by inv_scale / *grad_scale
 *   (grad_scale NULL: by inv_scale).  scratch4: float[4] device scratch owned by the caller. */
int vk_amp_unscale_check(size_t n, float* grad, const float* inv_scale, float* found_inf, void* stream);
int vk_adamw_step_amp(size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int* step_count, float inv_scale, const float* grad_scale,
                      const float* found_inf, float* scratch4, void* lowp_copy, vk_dtype lowp_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Engine level: the whole network as one plan
 * ---------------------------------------------------------------------------------------------- */
typedef struct vk_unet vk_unet;

typedef struct {
  int N, size;          /* batch per GPU and input HEIGHT (size % 32 == 0) */
  vk_dtype dtype;       /* compute/storage type of activations: VK_F32 (exact path) or VK_BF16 / VK_F16 */
  int training;         /* 1: plan keeps everything backward needs */
  int width;            /* input WIDTH (width % 32 == 0); 0 = size, the square inputs every reference script produces (train.py:70-75,
                           infer_pth_gui.py:17-24 letterbox to img_size x img_size) — smp itself accepts any H, W divisible by 32 (r04) */
} vk_unet_config;

typedef struct {
  char name[96];        /* smp state_dict key, e.g. "encoder.layer1.0.conv1.weight" */
  int kind;             /* 0 conv weight (KRSC in the flat buffer), 1 vector param (bn weight/bias, head bias),
                           2 fp32 buffer (running_mean/var), 3 int64 buffer (num_batches_tracked) */
  int dims[4];          /* logical torch shape: OIHW for kind 0, [C] for 1/2, [] for 3 */
  int ndim;
  int64_t offset;       /* element offset in the flat param buffer (kind 0/1), the flat fp32 buffer
                           (kind 2) or the int64 counter array (kind 3) */
  int64_t numel;
} vk_tensor_info;

int vk_unet_create(const vk_unet_config* cfg, vk_unet** out);
void vk_unet_destroy(vk_unet* h);
/* Optional: run a training plan's weight-gradient kernels on a second, library-owned HIP stream beside the caller's stream
 * (fork per layer once dz is final, join at the end of every backward stage, before the stage's gradient bucket may be
 * read); the only host-side state the library keeps.  Measured +1.3 % step throughput on one GPU, but the overlapping
 * kernels slow each other (per-kernel durations grow by 30-90 %), and with RCCL's stream as a third party the split measured
 * slower — so it is OFF by default (enable = 1, or VK_SIDE_STREAM=1 in the environment, turns it on). */
int vk_unet_set_side_stream(vk_unet* h, int enable);
int vk_unet_num_tensors(const vk_unet* h);
int vk_unet_tensor_info(const vk_unet* h, int index, vk_tensor_info* out);
int64_t vk_unet_param_numel(const vk_unet* h);        /* flat param/grad/moment buffer length (padded) */
int64_t vk_unet_buffer_numel(const vk_unet* h);       /* flat fp32 BN-buffer length */
int64_t vk_unet_workspace_bytes(const vk_unet* h);
/* gradient buckets for data-parallel all-reduce, in backward completion order */
int vk_unet_num_buckets(const vk_unet* h);
int vk_unet_bucket_range(const vk_unet* h, int bucket, int64_t* elem_begin, int64_t* elem_end);

/* params/grads: flat fp32 [param_numel]; bn_buffers: flat fp32 [buffer_numel]; nbt: int64[46];
 * workspace: workspace_bytes.  grads may be NULL for an inference plan. */
int vk_unet_bind(vk_unet* h, float* params, float* grads, float* bn_buffers, int64_t* nbt, void* workspace,
                 size_t workspace_bytes);
/* re-pack the compute copies of the weights after the fp32 master changed (load_state_dict, optimizer step) */
int vk_unet_refresh_weights(vk_unet* h, void* stream);

/* x: fp32 NCHW [N][3][S][S]; logits: fp32 [N][1][S][S].  training=1: batch statistics + running-stat update */
int vk_unet_forward(vk_unet* h, const float* x, float* logits, int training, void* stream);
/* target fp32 [N][1][S][S]; loss_out float[3]; computes dlogits for backward when the plan is a training plan */
int vk_unet_loss(vk_unet* h, const float* logits, const float* target, float* loss_out, float grad_scale,
                 float w_bce, float w_dice, void* stream);
/* dlogits: fp32 [N][1][S][S] gradient of the loss wrt the logits, or NULL to use the one vk_unet_loss
 * left in the workspace.  Runs backward stages [stage_begin, stage_end); stage i completes gradient bucket i.  Gradients are
 * accumulated into the flat grad buffer (caller zeroes it once per step, e.g. via vk_unet_zero_grad). */
int vk_unet_backward(vk_unet* h, const float* dlogits, int stage_begin, int stage_end, void* stream);
int vk_unet_zero_grad(vk_unet* h, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel collectives for a host that is not PyTorch (SURVEY.md 8(b) "DP", 8(e)): RCCL over xGMI behind plain C.  The reference
 * is single-process; this is the new capability's C boundary.  One process per GPU (call hipSetDevice first); rank 0 creates the id
 * and hands its VK_COMM_ID_BYTES to the other ranks by any out-of-band means (file, socket, MPI); vk_comm_init is collective.
 * A data-parallel step below the C boundary:
 *     vk_unet_zero_grad; vk_unet_forward; vk_unet_loss;
 *     for stage s in 0 .. num_buckets-1:  vk_unet_backward(h, NULL, s, s + 1, compute_stream)
 *     [event on compute_stream -> comm_stream waits]  vk_allreduce_bucket(comm, grads + b0, b1 - b0, comm_stream) for the finished
 *         buckets — per bucket, or (recommended, see parallel.py / DESIGN.md section 5) ONE call over buckets 0..8, which are contiguous,
 *         once stage 8 is done, then bucket 9 after the last stage;
 *     [compute_stream waits for comm_stream]  vk_adamw_step(..., inv_scale = 1 / world, ...)
 * librccl.so is opened on first use; VK_ERR_STATE when it is absent.
 * ---------------------------------------------------------------------------------------------- */
#define VK_COMM_ID_BYTES 128
typedef struct vk_comm vk_comm;
int vk_comm_unique_id(void* id_out /* VK_COMM_ID_BYTES */);
int vk_comm_init(int rank, int world, const void* id, vk_comm** out);
int vk_comm_world(const vk_comm* c);
/* in-place SUM over all ranks of `count` fp32 gradients, enqueued on `stream` (ncclAllReduce) */
int vk_allreduce_bucket(vk_comm* c, float* grads, size_t count, void* stream);
/* in-place broadcast of `bytes` bytes from rank `root` (parameters / BatchNorm buffers at start, ncclBroadcast) */
int vk_comm_broadcast(vk_comm* c, void* buf, size_t bytes, int root, void* stream);
int vk_comm_destroy(vk_comm* c);

/* debugging / parity: pointer + shape of a named intermediate ("z:encoder.layer1.0.conv1", "out:encoder.layer1.0", ...) */
int vk_unet_debug_tensor(const vk_unet* h, const char* name, void** ptr, int dims_nhwc[4]);

#ifdef __cplusplus
}
#endif
#endif /* VK_UNET_H */
