/* fsdet.h -- C ABI of libfsdet_hip.so: the MI355X (gfx950) kernels behind the few-shot
 * detection hot path (Darknet-19 meta feature extractor -> reweighting net -> channel-wise
 * reweighting -> RegionLoss).
 *
 * Boundary contract.  This is what a maintainer of the reference
 * binds (ctypes stub shown in INTEGRATION.md) in place of the PyTorch-0.3.1 / cuDNN /
 * python-loop implementations cited per entry point below.  Precedent for a native-operator
 * boundary in the reference: layers/batchnorm/src/batchnorm.h:1-6 (`bn_forward_gpu`, ...),
 * where the Python side pre-allocates every output and scratch tensor (layers/batchnorm/bn.py:16-53).
 *
 *   - plain pointers + sizes only; every pointer is DEVICE memory unless the name ends in _host
 *   - the library never allocates or frees: outputs and workspaces are caller-allocated
 *     (sizes via the *_bytes / *_elems queries)
 *   - every launcher enqueues on the hipStream_t it is given and returns immediately;
 *     re-entrant, no mutable global state
 *   - return value: 0 on success, a negative FSD_ERR_* for argument problems, or a positive
 *     hipError_t from the runtime.  Nothing aborts (the reference's cuda.c:27-49 did).
 *
 * Activation layout inside the path is NHWC fp32 ("pixel-major": channels contiguous) with an
 * explicit pixel stride so a tensor can be a channel slice of a wider (concat) buffer.
 * The public tensors of the reference API stay NCHW; fsd_transpose_batched converts.
 */
#ifndef FSDET_H_
#define FSDET_H_

#include <stddef.h>
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FSD_OK 0
#define FSD_ERR_ARG (-1)
#define FSD_ERR_UNSUPPORTED (-2)
#define FSD_ERR_WORKSPACE (-3)

/* ---- region loss ------------------------------------------------------------------------- */
/* stats[] (doubles) at the start of the region-loss workspace */
#define FSD_REGION_STATS 16
#define FSD_STAT_LOSS_X 0
#define FSD_STAT_LOSS_Y 1
#define FSD_STAT_LOSS_W 2
#define FSD_STAT_LOSS_H 3
#define FSD_STAT_LOSS_CONF 4
#define FSD_STAT_LOSS_CLS 5
#define FSD_STAT_NGT 6
#define FSD_STAT_NCORRECT 7
#define FSD_STAT_NPROPOSALS 8
#define FSD_STAT_BAD_TARGET 9 /* >0: a ground truth had no matching anchor / left the grid / bad class id */

size_t fsd_region_loss_workspace_bytes(int rows, int rows_per_image, int num_anchors, int height, int width);

/* Fused forward + gradient of RegionLossV2.forward (region_loss.py:252-366, softmax_over_rows=1,
 * rows_per_image = N episode classes, rows ordered b*N+n) and RegionLoss.forward
 * (region_loss.py:148-232, softmax_over_rows=0, rows_per_image=1), including build_targets
 * (region_loss.py:37-132).  neg_filter (region_loss.py:15-34) runs on the host because it reads
 * the host-resident target and python's RNG; its result arrives as `keep`.
 *   output      (rows, A*(5+C), H, W) fp32 NCHW      target (rows, target_len) float64
 *   keep        (rows) int32: compact index of a kept row, -1 if dropped
 *   grad_output same shape as output: d loss / d output (every element is written)
 *   loss_out    1 float: total loss (sum, not batch-normalised)
 *   dbg_targets optional, 9 planes of (rows, A, H, W) floats indexed by keep[]: coord_mask,
 *               conf_mask, cls_mask, tx, ty, tw, th, tconf, tcls (build_targets' return values)
 */
int fsd_region_loss_fwd_bwd(const float* output, const double* target, const int* keep,
                            float* grad_output, float* loss_out, void* workspace, size_t workspace_bytes,
                            int rows, int rows_per_image, int num_anchors, int num_classes,
                            int height, int width, int target_len, const double* anchors_host,
                            float coord_scale, float noobject_scale, float object_scale,
                            float class_scale, float thresh, long long seen, int max_boxes,
                            int softmax_over_rows, int zero_tcls, float* dbg_targets, hipStream_t stream);

/* Standalone build_targets (region_loss.py:37-132) for callers that hold DECODED boxes, as the reference's public
 * function does: IoU silence in float32, anchor matching and tconf in double, "later box wins" per cell.
 *   pred_boxes  (rows*A*H*W, 4) fp32: [x, y, w, h] in grid cells, cell order (row, anchor, y, x)
 *   target      (rows, target_len) float64, zero-terminated on cx
 *   targets_out 9 planes of (rows, A, H, W) floats: coord_mask, conf_mask, cls_mask, tx, ty, tw, th, tconf, tcls
 *   stats       FSD_REGION_STATS doubles (device): [FSD_STAT_NGT], [FSD_STAT_NCORRECT], [FSD_STAT_BAD_TARGET] */
int fsd_region_build_targets(const float* pred_boxes, const double* target, float* targets_out, double* stats,
                             int rows, int num_anchors, int height, int width, int target_len,
                             const double* anchors_host, float noobject_scale, float object_scale, float thresh,
                             long long seen, int max_boxes, hipStream_t stream);

/* Inference-side decode = utils.get_region_boxes_v2 (utils.py:195-290; softmax_over_rows = 1) or
 * utils.get_region_boxes (softmax_over_rows = 0): per (row, anchor, cell) sigmoid/exp decode, class
 * confidence, threshold, and compaction of the survivors on the device.
 *   boxes  [rows][cap][8] floats: key, cx, cy, w, h (normalised), det_conf, cls_conf, cls_id, where
 *          key = (cy*W + cx)*A + a is the reference's visiting order (sort by it on the host)
 *   counts [rows] int32: survivors per row (may exceed cap: the excess was dropped) */
int fsd_region_decode(const float* output, float* boxes, int* counts, int rows, int rows_per_image,
                      int num_anchors, int num_classes, int height, int width, const double* anchors_host,
                      float conf_thresh, int only_objectness, int softmax_over_rows, int cap,
                      hipStream_t stream);

/* Greedy non-maximum suppression = utils.nms (utils.py:85-104) on fsd_region_decode's output, one workgroup per
 * (image, class) row: sort by the float32 key 1 - det_conf (ties keep the reference's visiting order), then every
 * surviving box suppresses the later boxes whose IoU (centre format, double arithmetic like utils.bbox_iou) with it
 * exceeds nms_thresh.
 *   keep_idx    [rows][cap] int32: slots (second index of `boxes`) of the kept boxes, in the reference's output order
 *   keep_counts [rows] int32
 * cap <= 2048 (19x19x5 = 1805 cells at 608x608), else FSD_ERR_UNSUPPORTED. */
int fsd_region_nms(const float* boxes, const int* counts, int rows, int cap, float nms_thresh, int* keep_idx,
                   int* keep_counts, hipStream_t stream);

/* ---- convolution as implicit GEMM on the fp32 matrix cores -------------------------------- */
/* Packed weight: [round_up(rows,128)][round_up(taps*round_up(red,4), 32)] floats, K-major,
 * k = tap*red4 + r.  mode 0 (forward, replaces nn.Conv2d weight use, darknet_meta.py:236-250):
 * rows = Cout, red = Cin, tap = ky*ks+kx.  mode 1 (data gradient): rows = Cin, red = Cout and
 * taps flipped, so the same kernel computes dL/dx from dL/dy. */
size_t fsd_packed_weight_elems(int rows, int red, int ksize);
int fsd_pack_conv_weight(const float* w_oihw, float* w_packed, int cout, int cin, int ksize, int mode,
                         hipStream_t stream);

/* Number of BatchNorm partial rows fsd_conv2d_fwd* writes for this problem (= first dimension of the BN partial-sum
 * buffer): row tiles of the implicit-GEMM kernel, or one row per 8 x 16 pixel block of the halo-staged kernel of the
 * narrow 3x3 layers (which depends on the image shape, not only on the pixel count). */
int fsd_conv_row_tiles(int batch, int height, int width, int cout, int cin, int ksize);

/* y[p, co] = sum_{tap, ci} x[p + tap, ci] * w[co, tap, ci] (+ bias[co]);  stride 1,
 * pad = (ksize-1)/2, ksize in {1, 3}.  x: NHWC, cin % 4 == 0, pixel stride x_ld (floats).
 * out_nchw = 0: y NHWC with pixel stride y_ld;  out_nchw = 1: y is (B, cout, H, W) contiguous.
 * bn_partial (optional, out_nchw = 0 only): [row_tiles][cout][2] per-tile (sum, sum of squares)
 * of the raw outputs -- the batch statistics nn.BatchNorm2d needs (darknet_meta.py:245-248)
 * come out of the conv epilogue instead of a second pass over y. */
int fsd_conv2d_fwd(const float* x, long long x_ld, const float* w_packed, const float* bias, float* y,
                   long long y_ld, float* bn_partial, int batch, int height, int width, int cin,
                   int cout, int ksize, int out_nchw, hipStream_t stream);

/* First-layer 3x3 convolution (input with <= 4 channels stored as NHWC4, x_ld >= 4; cout % 32 == 0): HBM-bound
 * direct-operand MFMA kernel without LDS staging.  Takes the OIHW weights as they are (no packing).  Same results and
 * contract as fsd_conv2d_fwd(ksize = 3, out_nchw = 0); bn_partial is [fsd_conv3x3_c4_partial_rows][cout][2]. */
int fsd_conv3x3_c4_partial_rows(int batch, int height, int width);
int fsd_conv3x3_c4_fwd(const float* x, long long x_ld, const float* w_oihw, const float* bias, float* y, long long y_ld,
                       float* bn_partial, int batch, int height, int width, int cin, int cout, hipStream_t stream);

/* Winograd form of the fp32 3x3 convolution: input transform -> (tile+2)^2 batched GEMMs on the fp32 MFMA kernel ->
 * output transform (+bias, + BatchNorm partial sums [fsd_wino_partial_rows][cout][2]).
 *   tile = 2: F(2x2,3x3), 16 positions, 2.25x fewer multiplications, round-off ~1e-6 of the output magnitude
 *   tile = 4: F(4x4,3x3), 36 positions, 4x fewer multiplications (points 0,+-1,+-2,inf), round-off ~1.5e-5
 * u_packed = G g G^T from fsd_wino_pack_weight (mode 0 forward, mode 1 data gradient; same tile), cin % 32 == 0,
 * cout % 4 == 0.  Result equals fsd_conv2d_fwd up to that fp32 round-off. */
size_t fsd_wino_packed_weight_elems(int rows, int red, int tile);
int fsd_wino_pack_weight(const float* w_oihw, float* u_packed, int cout, int cin, int mode, int tile,
                         hipStream_t stream);
size_t fsd_wino_workspace_bytes(int batch, int height, int width, int cin, int cout, int tile);
int fsd_wino_partial_rows(int batch, int height, int width, int tile);
/* v_keep (nullable): if given, the transformed input B^T d B (fsd_wino_v_elems floats) is written there instead of
 * into the workspace, so that the weight gradient can reuse it (v_kept of fsd_wino_conv3x3_wgrad). */
size_t fsd_wino_v_elems(int batch, int height, int width, int cin, int tile);
/* v_in (nullable): an already transformed input (fsd_wino_grad_transforms); when given, x is not read. */
int fsd_wino_conv3x3_fwd(const float* x, long long x_ld, const float* u_packed, const float* bias, float* y,
                         long long y_ld, float* bn_partial, void* workspace, size_t workspace_bytes, float* v_keep,
                         const float* v_in, int batch, int height, int width, int cin, int cout, int tile,
                         hipStream_t stream);

/* First-layer 3x3 weight gradient (input with <= 4 channels stored as NHWC4, cout % 32 == 0) with the BatchNorm
 * backward fused into the operand load: dy = c1*(dt - c2 - xhat*c3) is formed in registers from dt (gradient w.r.t.
 * the BN output, fsd_bn_act_pool_bwd) and the raw conv output y, coef = [3][cout] from fsd_bn_bwd_finalize.  Replaces
 * fsd_bn_bwd_apply + fsd_conv2d_wgrad for a layer whose input needs no gradient. */
size_t fsd_conv3x3_wgrad_c4_bnfused_workspace_bytes(int batch, int height, int width, int cout);
int fsd_conv3x3_wgrad_c4_bnfused(const float* dt, long long dt_ld, const float* y, long long y_ld, const float* coef,
                                 const float* mean, const float* invstd, const float* x, long long x_ld, float* dw_oihw,
                                 void* workspace, size_t workspace_bytes, int batch, int height, int width, int cin,
                                 int cout, hipStream_t stream);

/* Inference forms of the three convolution entry points: y = leaky_slope(conv(x) + bias), NHWC store, no BatchNorm
 * statistics.  With eval-mode BatchNorm folded into the operands (w * scale per output channel, bias = shift) one launch
 * replaces conv + fsd_bn_finalize + fsd_bn_act_pool_fwd of a block without max pool (reference: conv / BatchNorm2d(eval) /
 * LeakyReLU modules, darknet_meta.py:236-256).  slope = 1 is exactly the plain entry point. */
int fsd_conv2d_fwd_act(const float* x, long long x_ld, const float* w_packed, const float* bias, float* y, long long y_ld,
                       float* bn_partial, int batch, int height, int width, int cin, int cout, int ksize, int out_nchw,
                       float slope, hipStream_t stream);
/* Activation on load (fp32): the input x is the RAW output of the producing convolution and the kernel forms
 * leaky(x * in_scale[c] + in_shift[c]) (per input channel, slope in_slope) in its staging registers -- the producer's
 * BatchNorm + leaky pass (fsd_bn_act_pool_fwd, pool 0) is then never run and its result never written or read.  Bit-identical
 * to running that pass and the plain entry point.  in_scale == in_shift == NULL: the plain entry point.
 *   fsd_conv2d_fwd_ex:        cin % 32 == 0 (else FSD_ERR_UNSUPPORTED).
 *   fsd_wino_conv3x3_fwd_ex:  tile == 4, only when the input transform runs (v_in == NULL).
 *   fsd_conv2d_wgrad_ex:      ksize == 1: the x operand of the weight gradient of a 1x1 convolution. */
int fsd_conv2d_fwd_ex(const float* x, long long x_ld, const float* w_packed, const float* bias, float* y, long long y_ld,
                      float* bn_partial, int batch, int height, int width, int cin, int cout, int ksize, int out_nchw,
                      float slope, const float* in_scale, const float* in_shift, float in_slope, hipStream_t stream);
int fsd_wino_conv3x3_fwd_ex(const float* x, long long x_ld, const float* u_packed, const float* bias, float* y,
                            long long y_ld, float* bn_partial, void* workspace, size_t workspace_bytes, float* v_keep,
                            const float* v_in, int batch, int height, int width, int cin, int cout, int tile, float slope,
                            const float* in_scale, const float* in_shift, float in_slope, hipStream_t stream);
int fsd_conv2d_wgrad_ex(const float* dy, long long dy_ld, const float* x, long long x_ld, float* dw_oihw, void* workspace,
                        size_t workspace_bytes, int batch, int height, int width, int cin, int cout, int ksize,
                        const float* x_scale, const float* x_shift, float x_slope, hipStream_t stream);
int fsd_wino_conv3x3_fwd_act(const float* x, long long x_ld, const float* u_packed, const float* bias, float* y,
                             long long y_ld, float* bn_partial, void* workspace, size_t workspace_bytes, float* v_keep,
                             const float* v_in, int batch, int height, int width, int cin, int cout, int tile, float slope,
                             hipStream_t stream);
int fsd_conv2d_fwd_act_h(const void* x_bf16, long long x_ld, const void* w_packed_bf16, const float* bias, void* y,
                         long long y_ld, float* bn_partial, int batch, int height, int width, int cin, int cout, int ksize,
                         int out_nchw_f32, float slope, hipStream_t stream);

/* Whole backward pass of a FIRST conv block (input of <= 4 channels stored as NHWC4, 3x3, BatchNorm, leaky, 2x2 / 2
 * max pool; height and width even, cout % 32 == 0) in one sweep over its activation: replaces fsd_bn_act_pool_bwd +
 * fsd_conv3x3_wgrad_c4_bnfused (reference: autograd through darknet_meta.py:219-268 for block 0).  The BatchNorm backward
 * is affine in dt, so  dW = c1 (S1 - c2 S2 - c3 S3)  with S1 = sum dt x, S2 = sum x, S3 = sum xhat x; _accum forms dt from
 * (dz, y) on the fly (never written) and emits per-workgroup partials of S1, S2, S3 (workspace) and of (sum dt,
 * sum dt xhat) (`partial`, [fsd_first_layer_bwd_rows][cout][2], the input of fsd_bn_bwd_finalize); _fold applies the
 * coefficients fsd_bn_bwd_finalize returns.  dz: gradient w.r.t. the pooled block output (batch, height/2, width/2, cout). */
int fsd_first_layer_bwd_rows(int batch, int height, int width);
size_t fsd_first_layer_bwd_workspace_bytes(int batch, int height, int width, int cout);
int fsd_first_layer_bwd_accum(const float* dz, long long dz_ld, const float* y, long long y_ld, const float* scale,
                              const float* shift, const float* mean, const float* invstd, float slope, const float* x,
                              long long x_ld, void* workspace, size_t workspace_bytes, float* partial, int batch,
                              int height, int width, int cin, int cout, hipStream_t stream);
int fsd_first_layer_bwd_fold(const void* workspace, size_t workspace_bytes, const float* coef, float* dw_oihw, int batch,
                             int height, int width, int cin, int cout, hipStream_t stream);

/* Which kernel variants the Winograd entry points will launch for a shape (pure host queries; the parity tests use them
 * to prove that a test shape really exercises a given variant).
 *   fsd_wino_fwd_plan:   plan4 = {BM, BN, dma_staged(0/1), m_tiles} of the batched position GEMM of
 *                        fsd_wino_conv3x3_fwd (forward and, with mode-1 weights, data gradient).
 *   fsd_wino_wgrad_plan: plan4 = {dma_128x128(0/1), row_splits, tail_rows (< 32, side launch on the 64x64 kernel),
 *                        workspace_slots} of the batched reduction GEMM of fsd_wino_conv3x3_wgrad. */
int fsd_wino_fwd_plan(int batch, int height, int width, int cin, int cout, int tile, int* plan4);
int fsd_wino_wgrad_plan(int batch, int height, int width, int cin, int cout, int tile, int* plan4);

/* Winograd form of the fp32 weight gradient of a 3x3 convolution, F(3x3, tile x tile): dW = sum over tiles,
 * (tile+2)^2 batched reduction GEMMs over tiles instead of 9 taps x pixels (tile 2: 2.25x, tile 4: 4x fewer
 * multiplications).  v_kept (nullable): the forward pass's transformed input of the same tile size; when given, x is
 * not read. */
size_t fsd_wino_wgrad_workspace_bytes(int batch, int height, int width, int cin, int cout, int tile);
/* wt_in (nullable): the already transformed gradient (fsd_wino_grad_transforms); when given, dy is not read. */
int fsd_wino_conv3x3_wgrad(const float* dy, long long dy_ld, const float* x, long long x_ld, const float* v_kept,
                           const float* wt_in, float* dw_oihw, void* workspace, size_t workspace_bytes, int batch,
                           int height, int width, int cin, int cout, int tile, hipStream_t stream);

/* BatchNorm backward fused into the weight-gradient transform of a Winograd(tile 4) layer: dt (gradient w.r.t. the BN
 * output) becomes dy = c1*(dt - c2 - xhat*c3) IN PLACE (what fsd_bn_bwd_apply would write) and wt_out = G4 dy G4^T
 * (wt_in of fsd_wino_conv3x3_wgrad) in the same pass; the data gradient then reads dy from dt as usual. */
int fsd_wino_dy_bn_transform(float* dt, long long dt_ld, const float* y, long long y_ld, const float* coef,
                             const float* mean, const float* invstd, float* wt_out, int batch, int height, int width,
                             int channels, int tile, hipStream_t stream);
/* The same after a statistics-only first pass (fsd_bn_act_pool_bwd with dt == NULL): dt is re-formed from dz (+ dz_full)
 * through the maxpool (pool 0 / 1) and the leaky activation, dy (dense (pixels, channels)) and wt_out are written.
 * Bit-identical to fsd_bn_act_pool_bwd + fsd_wino_dy_bn_transform. */
int fsd_wino_dy_bn_transform_g(const float* dz, long long dz_ld, const float* dz_full, long long dz_full_ld, const float* y,
                               long long y_ld, const float* scale, const float* shift, float slope, int pool,
                               const float* coef, const float* mean, const float* invstd, float* dy, float* wt_out,
                               int batch, int height, int width, int channels, int tile, hipStream_t stream);

/* Backward of a BatchNorm + Winograd(tile 4) layer in one pass over the gradient: forms dy = c1*(dt - c2 - xhat*c3)
 * (what fsd_bn_bwd_apply computes; coef from fsd_bn_bwd_finalize) in registers and writes both transformed operands the
 * layer's gradients need: v_out = B^T dy B (v_in of fsd_wino_conv3x3_fwd with the mode-1 weights = data gradient) and
 * wt_out = G4 dy G4^T (wt_in of fsd_wino_conv3x3_wgrad); fsd_wino_v_elems(batch, h, w, channels, 4) floats each. */
int fsd_wino_grad_transforms(const float* dt, long long dt_ld, const float* y, long long y_ld, const float* coef,
                             const float* mean, const float* invstd, float* v_out, float* wt_out, int batch, int height,
                             int width, int channels, int tile, hipStream_t stream);

/* Packed bf16 conv weights of the bf16 storage mode (operand of fsd_conv2d_fwd_h): [round_up(rows,128)]
 * [round_up(taps*round_up(red,4), 64)] bfloat16, K-major, round-to-nearest-even; mode 0 = forward (rows = Cout, red = Cin),
 * mode 1 = data gradient (rows = Cin, red = Cout, taps rotated 180 degrees). */
size_t fsd_packed_weight_elems_bf16(int rows, int red, int ksize);
int fsd_pack_conv_weight_bf16(const float* w_oihw, void* w_packed_bf16, int cout, int cin, int ksize, int mode,
                              hipStream_t stream);
/* Both packings (mode 0 and mode 1) of one weight tensor in one pass over it.  The two buffers
 * (fsd_packed_weight_elems_bf16(cout, cin, k) and (cin, cout, k) elements) must have been zero-filled ONCE by the
 * caller: only the non-padding elements are written. */
int fsd_pack_conv_weight_bf16_pair(const float* w_oihw, void* w_fwd_bf16, void* w_dgrad_bf16, int cout, int cin, int ksize,
                                   hipStream_t stream);
/* ---- batch norm (training statistics) + activation + pooling ------------------------------ */
/* Reduce the per-tile partials, produce the per-channel affine (scale = gamma*invstd,
 * shift = beta - mean*scale), save mean / invstd for the backward pass and update the running
 * statistics like nn.BatchNorm2d(momentum) in training mode (biased variance normalises,
 * unbiased variance feeds running_var).  training = 0: scale/shift from the running statistics. */
size_t fsd_bn_finalize_workspace_bytes(int channels);
int fsd_bn_finalize(const float* bn_partial, int row_tiles, long long count, int channels,
                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, int training, float* scale, float* shift,
                    float* save_mean, float* save_invstd, void* workspace, hipStream_t stream);

/* z = pool(act(y*scale + shift)).  y: (B,H,W,C) NHWC stride y_ld.  slope: 0.1 leaky, 0 relu,
 * 1 linear.  pool: 0 none, 1 = 2x2 stride 2 (floor), 2 = 2x2 stride 1 with replicate pad
 * (MaxPoolStride1, darknet_meta.py:47-53).  z: NHWC with pixel stride z_ld (channel-slice writes
 * implement [route] concatenation in place).  scale/shift may be NULL (identity). */
int fsd_bn_act_pool_fwd(const float* y, long long y_ld, const float* scale, const float* shift,
                        float slope, int pool, float* z, long long z_ld, int batch, int height,
                        int width, int channels, hipStream_t stream);

/* ---- data movement ------------------------------------------------------------------------ */
/* dst[b][c][r] = src[b][r][c] for r < rows, c < cols; element strides given per matrix.
 * NCHW -> NHWC: rows = C, cols = H*W;  NHWC -> NCHW: rows = H*W, cols = C. */
int fsd_transpose_batched(const float* src, long long src_batch_stride, long long src_row_stride,
                          float* dst, long long dst_batch_stride, long long dst_row_stride,
                          int batch, int rows, int cols, hipStream_t stream);
/* Network input: (batch, channels <= 4, hw) NCHW planes -> (batch*hw, 4) NHWC4 pixels, missing channels zero
 * (darknet_meta.py:117-118 concatenates image and mask along channels; the kernels read 16-byte pixels). */
int fsd_nchw_to_nhwc4(const float* src, float* dst, int batch, int channels, long long hw, hipStream_t stream);
int fsd_fill(float* dst, float value, long long count, hipStream_t stream);

/* Reorg (darknet_meta.py:55-74): out[b,i,j,(di*s+dj)*C + c] = x[b, s*i+di, s*j+dj, c], NHWC. */
int fsd_reorg_fwd(const float* x, long long x_ld, float* out, long long out_ld, int batch, int height,
                  int width, int channels, int stride, hipStream_t stream);

/* GlobalMaxPool2d (pooling.py:8-27): (B,H,W,C) -> (B,C).  argmax (optional) records the pixel. */
int fsd_global_maxpool_fwd(const float* x, long long x_ld, float* out, int* argmax, int batch,
                           int height, int width, int channels, hipStream_t stream);
/* GlobalAvgPool2d (pooling.py:29-45, F.adaptive_avg_pool2d(x, 1); the [globalavg] / [avgpool] cfg block): (B,H,W,C) -> (B,C),
 * fp64 sum over the map, one rounding.  x float (x_bf16 = 0) or bfloat16 bits (1).  _bwd: dx[b,p,c] = dout[b,c] / (H*W). */
int fsd_global_avgpool_fwd(const void* x, int x_bf16, long long x_ld, float* out, int batch, int height, int width,
                           int channels, hipStream_t stream);
int fsd_global_avgpool_bwd(const float* dout, void* dx, int dx_bf16, long long dx_ld, int batch, int height, int width,
                           int channels, hipStream_t stream);

/* Host -> device hand-over of per-step data (the (B, N, 250) float64 target tensor the reference's loss receives on the CPU,
 * train_meta.py:211 / region_loss.py:252): `words` 4-byte words from src to dst by a KERNEL.  src may be page-locked HOST
 * memory (hipHostMalloc: device-readable); the launch never blocks the host, unlike a memcpy on a busy stream. */
int fsd_upload_words(const void* src, void* dst, long long words, hipStream_t stream);

/* ---- channel-wise reweighting (dynamic_conv.py:125-164) ----------------------------------- */
/* Materialising form, NCHW like the reference module: out[b*N+n, c, hw] = x[b, c, hw] * w[n, c]. */
int fsd_dynamic_conv_fwd(const float* x, const float* w, float* out, int batch, int n_cls, int channels,
                         int hw, hipStream_t stream);
/* Fused form: fold the reweighting vectors into the 1x1 head so the (B*N, C, H, W) tensor never
 * exists:  w_eff[(n*O + o), c] = head_w[o, c] * dyn[n, c], bias_eff[n*O + o] = head_b[o], written
 * in the packed-weight layout of fsd_conv2d_fwd (rows = N*O, red = C, ksize = 1). */
int fsd_fold_reweight_head(const float* head_w, const float* head_b, const float* dyn, float* w_eff_packed,
                           float* bias_eff, int n_cls, int out_ch, int channels, hipStream_t stream);

/* ---- backward pass ------------------------------------------------------------------------- */
/* Weight gradient dW[co][ci][ky][kx] = sum_p dy[p][co] * x[p+tap][ci] (replaces autograd through
 * nn.Conv2d, darknet_meta.py:236-250).  dy: (pixels, dy_ld) NHWC gradient of the raw conv output
 * (dy_ld % 4 == 0, >= round_up(cout,4)); x: the NHWC input the convolution read.  Split-K over
 * pixels with a deterministic second-stage reduction; dw_oihw is fully overwritten. */
size_t fsd_conv2d_wgrad_workspace_bytes(int batch, int height, int width, int cin, int cout, int ksize);
int fsd_conv2d_wgrad(const float* dy, long long dy_ld, const float* x, long long x_ld, float* dw_oihw,
                     void* workspace, size_t workspace_bytes, int batch, int height, int width, int cin,
                     int cout, int ksize, hipStream_t stream);
/* (The data gradient is fsd_conv2d_fwd[_bf16] on dy with weights packed in mode 1.) */

/* Gradient through pool(act(y*scale+shift)): dt = d loss / d (y*scale+shift), dense (pixels, C),
 * plus per-block partial sums [fsd_bn_act_pool_bwd_rows(...)][C][2] of (dt, dt*xhat) for the BatchNorm
 * backward.  dz: grad of the block output (pooled grid if pool != 0); dz_full: optional grad of the
 * un-pooled activation.  The max-pool argmax is recomputed from y (first maximum in scan order).
 * dt == NULL: statistics only -- the partial sums are produced, dt is not written; the second pass then forms
 * dt again itself (fsd_bn_bwd_apply_g / fsd_wino_dy_bn_transform_g), which saves one write and one read of a full-resolution
 * tensor per layer. */
int fsd_act_bwd_rows(long long pixels);
int fsd_bn_act_pool_bwd_rows(int batch, int height, int width, int pool);   /* rows of `partial` below */
size_t fsd_reduce_workspace_bytes(int channels);
int fsd_bn_act_pool_bwd(const float* dz, long long dz_ld, const float* dz_full, long long dz_full_ld,
                        const float* y, long long y_ld, const float* scale, const float* shift,
                        const float* mean, const float* invstd, float slope, int pool, float* dt,
                        float* partial, int batch, int height, int width, int channels, hipStream_t stream);
/* Reduce partials -> dgamma (= sum dt*xhat), dbeta (= sum dt; also the bias gradient of a conv
 * without BN) and coef[3][C] = (scale, dbeta/count, dgamma/count) for fsd_bn_bwd_apply. */
int fsd_bn_bwd_finalize(const float* partial, int rows, long long count, int channels, const float* scale,
                        float* dgamma, float* dbeta, float* coef, void* workspace, hipStream_t stream);
/* In place: dt <- dy = scale * (dt - mean(dt) - xhat * mean(dt*xhat))  (training-mode BatchNorm). */
int fsd_bn_bwd_apply(float* dt, const float* y, long long y_ld, const float* coef, const float* mean,
                     const float* invstd, long long pixels, int channels, hipStream_t stream);
/* The second pass after a statistics-only first pass: dy (dense (pixels, C), written once) = scale * (dt - mean(dt) - xhat *
 * mean(dt*xhat)) with dt re-formed from dz (+ dz_full) through the 2x2 / stride-2 maxpool (pool == 1; pool == 0: none) and the
 * leaky activation exactly as fsd_bn_act_pool_bwd forms it.  Bit-identical to fsd_bn_act_pool_bwd + fsd_bn_bwd_apply. */
int fsd_bn_bwd_apply_g(const float* dz, long long dz_ld, const float* dz_full, long long dz_full_ld, const float* y,
                       long long y_ld, const float* scale, const float* shift, float slope, int pool, const float* coef,
                       const float* mean, const float* invstd, float* dy, int batch, int height, int width, int channels,
                       hipStream_t stream);
/* bf16 storage (fsd_bn_act_pool_bwd_h with dt == NULL is the statistics-only first pass): channels % 8 == 0, 16-byte aligned
 * rows.  dt is formed in fp32 and not rounded to bf16 on the way; dy is rounded once. */
int fsd_bn_bwd_apply_g_h(const void* dz, long long dz_ld, const void* dz_full, long long dz_full_ld, const void* y,
                         long long y_ld, const float* scale, const float* shift, float slope, int pool, const float* coef,
                         const float* mean, const float* invstd, void* dy, int batch, int height, int width, int channels,
                         hipStream_t stream);
/* Column sums of a (rows, ld) matrix as partials [fsd_act_bwd_rows(rows)][C][2] (any C). */
int fsd_colsum_partials(const float* m, long long ld, float* partial, long long rows, int channels,
                        hipStream_t stream);
int fsd_reorg_bwd(const float* dout, long long dout_ld, float* dx, long long dx_ld, int batch, int height,
                  int width, int channels, int stride, hipStream_t stream);
int fsd_global_maxpool_bwd(const float* dout, const int* argmax, float* dx, long long dx_ld, int batch,
                           int height, int width, int channels, hipStream_t stream);
int fsd_add_inplace(float* dst, long long dst_ld, const float* src, long long src_ld, long long rows,
                    int channels, hipStream_t stream);
/* Un-fold the gradient of the fused head weights w_eff[(n*O+o), c] = head_w[o,c] * dyn[n,c]. */
int fsd_head_unfold_bwd(const float* dweff, const float* head_w, const float* dyn, float* d_head_w,
                        float* d_dyn, int n_cls, int out_ch, int channels, hipStream_t stream);
/* SGD with momentum and L2 weight decay on one flat fp32 buffer (train_meta.py:143-147 uses
 * torch.optim.SGD): d = g + wd*w; buf = first ? d : momentum*buf + d; w -= lr*buf. */
int fsd_sgd_step(float* w, const float* grad, float* momentum_buf, float lr, float momentum,
                 float weight_decay, int first_step, long long count, hipStream_t stream);
/* The same step over MANY tensors of one flat parameter buffer in ONE launch, with the bf16 storage mode's per-step
 * re-packing of the conv operands folded in (train_meta.py:143-147 + what fsd_pack_conv_weight_bf16_pair does afterwards).
 * table_dev: device array [n_entries][8] of int64: {element offset of the tensor in the three flat buffers, element count,
 * cout, cin, taps (9 / 1: an OIHW conv weight with packed copies; 0: a plain range), first workgroup block of the entry
 * (prefix sum of fsd_sgd_multi_blocks), address of the bf16 forward operand, address of the bf16 data-gradient operand (both
 * as laid out by fsd_pack_conv_weight_bf16_pair; zero-filled once by the caller)}.  Entries must not overlap.  elements = sum
 * of the counts (profiling only). */
long long fsd_sgd_multi_blocks(long long count, int cout, int cin, int taps);
int fsd_sgd_step_multi(float* w_flat, const float* grad_flat, float* momentum_flat, const long long* table_dev,
                       int n_entries, long long total_blocks, long long elements, float lr, float momentum,
                       float weight_decay, int first_step, hipStream_t stream);

/* ---- bf16 storage mode (BASELINE configs[2] / [4]) --------------------------------------------------------------
 * The `_h` entry points are the bfloat16-storage twins of the functions above: every activation / activation-gradient
 * tensor (`void*`) holds raw bfloat16 bits in the same NHWC layout with leading dimensions counted in ELEMENTS; all
 * arithmetic, the BatchNorm partial sums and every parameter / parameter gradient stay float.  Arguments, shapes,
 * return codes and the kernels behind them are otherwise identical (one kernel template, two instantiations). */
int fsd_conv_row_tiles_h(long long pixels);          /* upper bound of the rows of any bn_partial array (128-row tiles) */
/* rows of the bn_partial array fsd_conv2d_fwd_h fills for this layer (one row per row tile of the tile it will pick; one
 * per workgroup of the persistent halo kernel of the 32 -> 64 / 64 -> 32 3x3 layers, conv_halo_h.hip) */
int fsd_conv2d_h_partial_rows(int batch, int height, int width, int cin, int cout, int ksize);
/* ... for exactly these operands: the halo-staged kernel additionally needs 16-byte aligned pixel rows on both sides
 * (x_ld % 8 == 0, y_ld % 8 == 0, aligned base pointers); a launch whose strides it cannot take runs on the implicit-GEMM
 * kernel and fills one row per 128-row tile.  fsd_conv2d_fwd[_act]_h follows the same rule, so a bn_partial array sized by
 * this query is always the one the launch fills (with the default dense views both queries agree). */
int fsd_conv2d_h_partial_rows_at(int batch, int height, int width, int cin, int cout, int ksize, const void* x_bf16,
                                 long long x_ld, const void* y, long long y_ld);
/* which tile fsd_conv2d_fwd_h will run this layer on: 0 = 128x128 (4 waves, two workgroups per CU), 1 = 256x256,
 * 2 = 192x256, 3 = 256x128 (8 waves, one workgroup per CU), 4 = 128x64, 5 = 128x32, 6 = 192x128 (4 waves, two workgroups
 * per CU).  Tests assert through it that the timed shapes really take the tiles they are meant to; FSD_CONV_H_TILE=0..6
 * forces one. */
int fsd_conv2d_h_plan(long long pixels, int cin, int cout, int ksize, int out_nchw_f32, int has_partial);
/* bf16 activations x packed bf16 weights (fsd_pack_conv_weight_bf16) -> bf16 NHWC y (or float NCHW when out_nchw_f32),
 * fp32 accumulation on v_mfma_f32_32x32x16_bf16, operands staged global -> LDS by DMA.  cin % 32 == 0, cout even. */
int fsd_conv2d_fwd_h(const void* x_bf16, long long x_ld, const void* w_packed_bf16, const float* bias, void* y,
                     long long y_ld, float* bn_partial, int batch, int height, int width, int cin, int cout, int ksize,
                     int out_nchw_f32, hipStream_t stream);
size_t fsd_conv2d_wgrad_h_workspace_bytes(int batch, int height, int width, int cin, int cout, int ksize);
/* which tile fsd_conv2d_wgrad_h will reduce this layer on: rows * 1000 + columns of the dW tile (256256 = the
 * 256 x 256 tile with 64-pixel chunks; 128128, 128064, 64128, 64064, 128032 = the 4-wave kernel).  Tests assert through it
 * that their shapes reach the kernel they mean to check. */
int fsd_conv2d_wgrad_h_plan(long long pixels, int cin, int cout, int ksize);
/* dW (float, OIHW) from bf16 dy and bf16 x: K = pixels, fragments through the LDS transpose read. cin, cout % 8 == 0. */
int fsd_conv2d_wgrad_h(const void* dy_bf16, long long dy_ld, const void* x_bf16, long long x_ld, float* dw_oihw,
                       void* workspace, size_t workspace_bytes, int batch, int height, int width, int cin, int cout,
                       int ksize, hipStream_t stream);
int fsd_conv3x3_c4_fwd_h(const float* x, long long x_ld, const float* w_oihw, const float* bias, void* y_bf16,
                         long long y_ld, float* bn_partial, int batch, int height, int width, int cin, int cout,
                         hipStream_t stream);
int fsd_conv3x3_wgrad_c4_bnfused_h(const void* dt, long long dt_ld, const void* y, long long y_ld, const float* coef,
                                   const float* mean, const float* invstd, const float* x, long long x_ld, float* dw_oihw,
                                   void* workspace, size_t workspace_bytes, int batch, int height, int width, int cin,
                                   int cout, hipStream_t stream);
int fsd_first_layer_bwd_accum_h(const void* dz, long long dz_ld, const void* y, long long y_ld, const float* scale,
                                const float* shift, const float* mean, const float* invstd, float slope, const float* x,
                                long long x_ld, void* workspace, size_t workspace_bytes, float* partial, int batch,
                                int height, int width, int cin, int cout, hipStream_t stream);
int fsd_bn_act_pool_fwd_h(const void* y, long long y_ld, const float* scale, const float* shift, float slope, int pool,
                          void* z, long long z_ld, int batch, int height, int width, int channels, hipStream_t stream);
/* src / dst each float (flag 0) or bf16 (flag 1) */
int fsd_transpose_batched_h(const void* src, int src_bf16, long long src_batch_stride, long long src_row_stride, void* dst,
                            int dst_bf16, long long dst_batch_stride, long long dst_row_stride, int batch, int rows,
                            int cols, hipStream_t stream);
int fsd_reorg_fwd_h(const void* x, long long x_ld, void* out, long long out_ld, int batch, int height, int width,
                    int channels, int stride, hipStream_t stream);
int fsd_global_maxpool_fwd_h(const void* x, long long x_ld, float* out, int* argmax, int batch, int height, int width,
                             int channels, hipStream_t stream);
int fsd_bn_act_pool_bwd_h(const void* dz, long long dz_ld, const void* dz_full, long long dz_full_ld, const void* y,
                          long long y_ld, const float* scale, const float* shift, const float* mean, const float* invstd,
                          float slope, int pool, void* dt, float* partial, int batch, int height, int width, int channels,
                          hipStream_t stream);
int fsd_bn_bwd_apply_h(void* dt, const void* y, long long y_ld, const float* coef, const float* mean, const float* invstd,
                       long long pixels, int channels, hipStream_t stream);
int fsd_colsum_partials_h(const void* m, long long ld, float* partial, long long rows, int channels, hipStream_t stream);
int fsd_reorg_bwd_h(const void* dout, long long dout_ld, void* dx, long long dx_ld, int batch, int height, int width,
                    int channels, int stride, hipStream_t stream);
int fsd_global_maxpool_bwd_h(const float* dout, const int* argmax, void* dx, long long dx_ld, int batch, int height,
                             int width, int channels, hipStream_t stream);
int fsd_add_inplace_h(void* dst, long long dst_ld, const void* src, long long src_ld, long long rows, int channels,
                      hipStream_t stream);

/* Episode input pipeline on the device = the image half of the reference's CPU loader: image.data_augmentation
 * (jitter crop, NEAREST resize, horizontal flip, HSV distortion; image.py:13-87) + ToTensor (train_meta.py:176-178).
 * One gather kernel from packed uint8 RGB (HWC) source images to the float network input.
 *   src / img_off [B] / img_w [B]   packed source images, byte offset and width of image b
 *   xtab [B][out_w], ytab [B][out_h] int32: source column / row of every output column / row, or -1 = outside the
 *        image (black); crop offset, Pillow's nearest-neighbour arithmetic and the flip are folded in by the host
 *        (episode.index_tables)
 *   luts [B][3][256] uint8 (nullable): the H, S, V tables of image.distort_image; null = no colour distortion
 *   distort [B] int32 (nullable): 0 = image b skips the colour distortion (data_augmentation(flag=False) inside a
 *        batch that otherwise distorts: the RGB->HSV->RGB round trip is not the identity); null = every image distorts
 *   mask_box [B][4] int32 x1, y1, x2, y2 (nullable, layout 1 only): support-mask rectangle (dataset.py:378-398)
 *   layout 0: out = (B, 3, out_h, out_w) float NCHW;  layout 1: out = (B, out_h, out_w, 4) float, channel 3 = mask / 0
 * Bit-exact with the reference run on Pillow with the 2018 defaults (tests/golden/augment.npz). */
int fsd_augment_batch(const unsigned char* src, const long long* img_off, const int* img_w, const int* xtab,
                      const int* ytab, const unsigned char* luts, const int* distort, const int* mask_box, float* out,
                      int batch, int out_h,
                      int out_w, int layout, hipStream_t stream);

/* Measurement aid (bench.py): while enabled, every launch of the kernel classes below is bracketed by HIP events on the
 * stream it is launched on and booked with its work figure; fsd_profile_collect waits for the recorded events, returns
 * per class the summed kernel time [ms], the summed work and the number of launches, and clears the records.
 *   class 0 conv_gemm_kernel (fp32 MFMA forward / data-gradient / Winograd position GEMMs)   work = MFMA FLOPs issued
 *   class 1 wgrad_kernel (fp32 MFMA weight-gradient reduction GEMMs)                          work = MFMA FLOPs issued
 *   class 2 Winograd input / output / gradient transforms                                     work = algorithmic bytes
 *   class 3 BatchNorm / leaky / maxpool backward passes                                       work = algorithmic bytes
 *   class 4 BatchNorm + leaky + maxpool forward pass                                          work = algorithmic bytes
 *   class 5 region-loss kernels                                                               work = algorithmic bytes
 *   class 6 fused SGD step                                                                    work = algorithmic bytes
 *   class 7 first-layer direct-operand kernels                                                work = algorithmic bytes
 *   class 8 bf16-operand MFMA GEMM kernels                                                    work = MFMA FLOPs issued
 * Arrays must hold fsd_profile_num_classes() entries.  Off by default: one relaxed load per launch. */
void fsd_profile_enable(int on);
int fsd_profile_num_classes(void);
int fsd_profile_collect(double* ms, double* work, long long* launches, int n_classes);
/* Number of kernel launches the library has issued since the process started (or since the last call with reset != 0, which
 * also zeroes the counter): every launch, whatever its class, recording on or off.  How many kernels one eval-mode
 * detect_forward takes (reference shape: valid_ensemble.py:137-148) is read off this. */
long long fsd_launch_count(int reset);
/* Shader clock (MHz) sustained under matrix-core load: one wave per SIMD runs iters x 16 dependent 64-cycle fp32 MFMAs;
 * synchronises.  scratch: any device buffer of >= 4 bytes (never written). */
int fsd_clock_probe(float* scratch, int iters, double* mhz_out, hipStream_t stream);

/* ---- arithmetic of the fp32 GEMMs (forward / data-gradient / weight-gradient kernels of the fp32 mode) ------------------
 * mode 0: native fp32 matrix instruction (v_mfma_f32_32x32x2_f32).
 * mode 1: "split" -- every fp32 operand element is split, inside the kernel on its way into LDS, into three bfloat16 planes
 *         x = x1 + x2 + x3 (exact), and a product is accumulated in fp32 from the six cross terms down to 2^-16 relative on
 *         the bf16 matrix instruction (v_mfma_f32_32x32x16_bf16).  Storage, accumulation and results stay fp32; the error
 *         against a float64 reference is at or below the native instruction's (tests/test_gpu_split.py).
 *         Non-finite and out-of-range operands differ from mode 0: an infinite operand element gives NaN (inf - bf16(inf)),
 *         where the fp32 instruction gives +-inf or NaN depending on its partner, and finite fp32 values above bfloat16's
 *         largest (|x| > 3.3895e38) round to inf the same way; NaN stays NaN.  Finite operands below that bound behave as in
 *         mode 0 over the whole fp32 exponent range (bfloat16 has fp32's exponent).
 * Any other value only queries.  Returns the previous mode.  Takes effect for kernels LAUNCHED afterwards -- a hipGraph
 * captured earlier replays the arithmetic it was captured in (the Python layer keys its inference graphs on the mode); the
 * first use reads the environment variable FSD_F32_SPLIT (0 / 1). */
int fsd_f32_gemm_mode(int mode);

const char* fsd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FSDET_H_ */
