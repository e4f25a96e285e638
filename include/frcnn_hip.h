/*
 * frcnn_hip.h -- C ABI of libfrcnn_hip.so, the MI355X (gfx950) Faster R-CNN hot path: inference (forward / predict),
 * and the operators of the train step (section "Training path" below).
 *
 * The reference (trzy/FasterRCNN, pytorch tree) has no FFI: its hot path is Python calling
 * torch / torchvision native kernels.  Each entry point below replaces one of those call sites;
 * the reference file:line it stands in for is cited per function (paths relative to
 * /root/reference/pytorch/FasterRCNN/).  INTEGRATION.md shows the ctypes stub a maintainer of
 * the reference would add to bind them.
 *
 * Conventions (all functions):
 *   - return int: 0 = FRCNN_OK, negative = error code below; never throws, never aborts.
 *   - every pointer named d_* / marked "device" is a DEVICE pointer owned by the caller
 *     (e.g. a torch allocation); nothing is allocated behind the caller's back except inside
 *     an frcnn_ctx (frcnn_ctx_create / frcnn_ctx_destroy).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued
 *     asynchronously on it, no host synchronisation unless stated.
 *   - activations are float32, NHWC ("pixel-major, channel-contiguous") unless stated; the
 *     network input is the reference's NCHW float32 image.
 *   - box layout is the reference's (y1, x1, y2, x2) in image pixels; anchors are
 *     (center_y, center_x, height, width).
 */
#ifndef FRCNN_HIP_H
#define FRCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRCNN_OK            0
#define FRCNN_EINVAL       -1   /* bad argument (null pointer, shape not supported by the kernel) */
#define FRCNN_EHIP         -2   /* a HIP runtime call failed; see frcnn_last_hip_error() */
#define FRCNN_ENOMEM       -3   /* workspace allocation failed */
#define FRCNN_EUNSUPPORTED -4   /* valid request outside what this build implements */
#define FRCNN_ENODEVICE    -5   /* no gfx950 device visible */

#define FRCNN_ABI_VERSION 16  /* 2: training entry points, frcnn_forward_params.conv_blocks_target; 3: Winograd F(2x2,3x3) layers; 4: one-launch Winograd layers;
                                 5: bf16 gradient GEMMs (the *_math entry points); 6: x6t GEMM, x6 Winograd layers, frcnn_forward_params.winograd_x6_mask,
                                 timing classes 8 / 9; 7: batched feature extractor (frcnn_resnet_backbone, frcnn_resnet_forward_features,
                                 frcnn_ctx_create_backbone, frcnn_conv3x3_nhwc_winograd_fused_maps); 8: the f32x3 arithmetic (frcnn_*_x3t, frcnn_*_winograd_x3,
                                 frcnn_forward_params.winograd_x3_mask, FRCNN_FC_F32X3T, frcnn_bottleneck_weights.x3_mask); 9: one-launch f32x3 Winograd layers
                                 in the forward (frcnn_forward_params.winograd_x3f_mask, timing class 10), frcnn_roi_pool_x3t; 10: frcnn_conv3x3_c3_cmax, bits 1 (conv1_2)
                                 and 13 (RPN trunk) of winograd_x3f_mask; 11: frcnn_conv_nhwc_x3g, frcnn_tensor_absmax, frcnn_bottleneck_weights.g3 / .wmax;
                                 12: frcnn_x3_saturation_events, FRCNN_X3F_WAVES4 / FRCNN_X3F_WAVES8; 13: REMOVED the round-2 f32x6 kernels that no table has used since round 3
                                 (frcnn_pack_conv3x3_x6, frcnn_conv3x3_nhwc_x6, frcnn_split_rows_x6, frcnn_linear_x6(_workspace_bytes), math mode 1 =
                                 FRCNN_MATH_F32X6 and fc mode 1 = FRCNN_FC_F32X6 are FRCNN_EINVAL); the f32x6 arithmetic stays as gemm_x6t / wino_x6; 14: FRCNN_X3F_PAIR, frcnn_conv3x3_winograd_x3_pair_workspace_bytes, frcnn_forward_params.winograd_x3p_mask,
                                 frcnn_resnet_rpn_roipool / frcnn_ctx_create_head / frcnn_resnet_head; 15: frcnn_conv_nhwc_x3g_tickets (split reductions finished inside the kernel), frcnn_pack_conv_x3g_weights + FRCNN_X3G_WSPLIT + frcnn_bottleneck_weights.g3 == 2 (pre-split weight packs);
                                 16: frcnn_train_conv, frcnn_bottleneck_backward(_workspace_bytes): the backward of one trainable bottleneck as ONE call, weight gradients on a second stream */

/* flags for frcnn_conv3x3_nhwc / frcnn_linear */
#define FRCNN_RELU   1u
#define FRCNN_POOL2  2u   /* fuse MaxPool2d(2, stride 2, floor) into the conv epilogue */
/* ABI 12, frcnn_conv3x3_nhwc_winograd_x3_fused / _chain only: force one of the two forms of the one-launch f32x3 layer (both bits clear: the
 * launcher chooses by cin).  The forms give the same results bit for bit (tests/test_gemm_x3t_gpu.py); the bits exist for that test and for
 * tools/x3f_bench.py / tools/xd_clocks.py. */
#define FRCNN_X3F_WAVES4 0x100u   /* wino_x3d_kernel: four waves, one per SIMD, 256 accumulator registers (csrc/wino_x3f.hip) */
#define FRCNN_X3F_WAVES8 0x200u   /* wino_x3e_kernel: eight waves, two per SIMD, 128 accumulator registers (csrc/wino_x3e.hip) */
/* ABI 14: the TWO-PASS form of the one-launch f32x3 layer, 128 output channels per block (wino_x3p_kernel, csrc/wino_x3p.hip): cin % 32 == 0,
 * cin >= 64, cout % 128 == 0; same results bit for bit; d_ws >= frcnn_conv3x3_winograd_x3_pair_workspace_bytes (channel maxima + the
 * block-private scratch the first pass's accumulators rest in). */
#define FRCNN_X3G_WSPLIT 0x800u   /* frcnn_conv_nhwc_x3g(_tickets): d_w_packed is the PRE-SPLIT image of the float32 pack (frcnn_pack_conv_x3g_weights, round 6) */
#define FRCNN_X3F_PAIR   0x400u

int         frcnn_abi_version(void);
const char* frcnn_error_string(int code);
/* hipGetErrorString of the last HIP failure seen by this thread ("" if none). */
const char* frcnn_last_hip_error(void);
/* Number of visible devices whose gcnArchName starts with "gfx950"; <0 on error. */
int         frcnn_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Anchors.  Replaces models/anchors.py:43-135 generate_anchor_maps (float64 math, one final
 * cast to float32 -- reproduced bit-exactly).
 *   d_anchor_map : float32 [fh][fw][9*4]  (cy, cx, h, w) per anchor, k = area-major/aspect-minor
 *   d_valid_map  : float32 [fh][fw][9]    1.0 if the anchor lies inside the image else 0.0
 * ---------------------------------------------------------------------------------------- */
int frcnn_anchors(int image_h, int image_w, int fh, int fw, int feature_pixels,
                  float* d_anchor_map, float* d_valid_map, void* stream);

/* Image preprocessing.  Replaces datasets/image.py:92-100 and :43-57: PIL
 * `Image.resize((out_w, out_h), BILINEAR)` of the decoded 8-bit RGB image (optionally after
 * FLIP_LEFT_RIGHT), reproduced bit for bit (22-bit fixed-point two-pass resampler), then channel
 * order / scaling / (x - mean) / std in float32 into the CHW tensor the model consumes.
 *   d_rgb    : uint8 [H][W][3] RGB (what imageio / PIL decode to)
 *   means, stds: HOST float[3] in OUTPUT channel order (PreprocessingParams.means / .stds)
 *   d_out    : float32 [3][out_h][out_w];  d_out_u8: optional uint8 [out_h][out_w][3] resized RGB
 *   d_ws     : scratch of frcnn_preprocess_workspace_bytes(H, W, out_h, out_w) bytes */
size_t frcnn_preprocess_workspace_bytes(int H, int W, int out_h, int out_w);
int frcnn_preprocess(const unsigned char* d_rgb, int H, int W, int out_h, int out_w, int bgr_order,
                     int horizontal_flip, float scaling, const float* means, const float* stds,
                     float* d_out, unsigned char* d_out_u8, void* d_ws, size_t ws_bytes, void* stream);

/* RPN ground-truth labelling (anchor <-> GT IoU matching).  Replaces models/anchors.py:137-262
 * generate_rpn_map: float64 IoU of every anchor with every ground-truth box (float32 corners
 * (y1,x1,y2,x2)), invalid anchors excluded; object if IoU >= object_thr or the anchor attains a
 * GT box's maximum IoU, background if max IoU < background_thr, ignored otherwise; float32
 * regression targets of the best-IoU box.
 *   d_rpn_map        : float32 [A][6] = (trainable, object, ty, tx, th, tw), A = fh*fw*9
 *   d_object_idx     : int32 [A] flat anchor indices of object anchors, ascending; d_counts[0] of them
 *   d_background_idx : int32 [A] likewise for background anchors; d_counts[1]
 *   d_ws             : scratch, at least 8*n_gt bytes.  n_gt >= 1 (the reference cannot label an
 *                      image without boxes either). */
int frcnn_rpn_targets(const float* d_anchor_map, const float* d_valid_map, int n_anchors,
                      const float* d_gt_boxes, int n_gt, double object_thr, double background_thr,
                      float* d_rpn_map, int32_t* d_object_idx, int32_t* d_background_idx, int32_t* d_counts,
                      void* d_ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight repacking (run once at load; device -> device).
 * ---------------------------------------------------------------------------------------- */
/* OIHW [cout][cin][3][3] -> tap-major [9][cout][cin] (cin contiguous) for frcnn_conv3x3_nhwc. */
int frcnn_pack_conv3x3(const float* d_w_oihw, float* d_w_packed, int cout, int cin, void* stream);
/* OIHW [cout][3][3][3] -> [27][cout] (k = ci*9 + r*3 + s major) for frcnn_conv3x3_c3. */
int frcnn_pack_conv3x3_c3(const float* d_w_oihw, float* d_w_packed, int cout, void* stream);
/* fc1 [out][c*ph*pw] (reference flatten order C,7,7: vgg16.py:129) -> [out][(p)*C + c]
 * so that it consumes the NHWC RoI-pool output directly. */
int frcnn_pack_fc_chw_to_hwc(const float* d_w, float* d_w_packed, int out_features, int channels,
                             int pooled_hw, void* stream);
/* Stack row-major matrices [n1][k] and [n2][k] into [n_pad][k] (rows >= n1+n2 zero) and the
 * biases into [n_pad].  Used for the RPN 1x1 heads (rpn.py:40-41) and the detector heads
 * (detector.py:29-30). */
int frcnn_pack_stack_rows(const float* d_w1, const float* d_b1, int n1,
                          const float* d_w2, const float* d_b2, int n2,
                          int k, int n_pad, float* d_w_out, float* d_b_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolutions.  Replace nn.Conv2d(3x3, stride 1, "same") + F.relu (+ nn.MaxPool2d(2,2)) at
 * models/vgg16.py:76-96 and models/rpn.py:88.
 * ---------------------------------------------------------------------------------------- */
/* First layer: Cin = 3, input NCHW float32 [3][H][W], output NHWC [H][W][cout]; cout % 16 == 0.
 * d_w_packed from frcnn_pack_conv3x3_c3. */
int frcnn_conv3x3_c3(const float* d_x_chw, const float* d_w_packed, const float* d_bias,
                     float* d_y, int H, int W, int cout, unsigned flags, void* stream);
/* The same layer (models/vgg16.py:76 conv1_1 + ReLU; cout = 64 only) that also leaves the per-pixel maximum |y| over the 64 output channels in d_cmax_out [H][W]: the scale
 * source of an f32x3 layer that consumes y (d_cmax_in of frcnn_conv3x3_nhwc_winograd_x3_chain).  A plain store per pixel from the lanes
 * that hold its channels (no atomics, nothing to zero); y is bit-identical to frcnn_conv3x3_c3's.  ABI 10. */
int frcnn_conv3x3_c3_cmax(const float* d_x_chw, const float* d_w_packed, const float* d_bias,
                          float* d_y, int H, int W, int cout, unsigned flags, float* d_cmax_out, void* stream);
/* General layer on the f32 MFMA pipe: x NHWC [H][W][cin], y NHWC [H][W][cout] or, with
 * FRCNN_POOL2, [H/2][W/2][cout].  Requires cin % 16 == 0, cout % 64 == 0.
 * d_w_packed from frcnn_pack_conv3x3.  Layers whose output grid cannot fill the chip (the 37x62
 * maps of block 5 / the RPN trunk) run split-K over input channels with a deterministic
 * fixed-order finish; `d_ws` is scratch of at least frcnn_conv3x3_workspace_bytes() bytes
 * (NULL or too small = no split, same results up to fp32 summation order). */
size_t frcnn_conv3x3_workspace_bytes(int H, int W, int cin, int cout);
int frcnn_conv3x3_nhwc(const float* d_x, const float* d_w_packed, const float* d_bias,
                       float* d_y, int H, int W, int cin, int cout, unsigned flags,
                       void* d_ws, size_t ws_bytes, void* stream);
/* The same layer (models/vgg16.py:36-47, models/rpn.py:39; with n_maps > 1 also the stride-1 3x3 of a ResNet
 * bottleneck on the per-RoI maps, models/resnet.py:110) as Winograd F(2x2,3x3) in float32: three launches
 * (input transform, 16 batched exact-f32 MFMA GEMMs over the channels, output transform + bias + ReLU +
 * optional 2x2 max-pool), 2.25x fewer multiplies than the direct form.  No reduced-precision operands: every
 * value is float32 (the filter transform G g G^T is evaluated in float64 and rounded once); results differ
 * from the direct kernel by fp32 rounding only (a few 1e-7 relative per layer).
 *   d_x    : float32 NHWC [n_maps][H][W][cin];  d_y : [n_maps][H][W][cout] ([n_maps][H/2][W/2][cout] with FRCNN_POOL2)
 *   d_u    : float32 [16][cout][cin] from frcnn_pack_conv3x3_winograd (OIHW weights in; d_row_scale, optional,
 *            multiplies filter row `cout` in float32 first = the frozen-BatchNorm fold of frcnn_fold_bn_pack)
 *   d_ws   : scratch of frcnn_conv3x3_winograd_workspace_bytes() = 16*n_maps*ceil(H/2)*ceil(W/2)*(cin+cout)*4 bytes
 * Requires cin % 16 == 0 and cout % 128 == 0 (FRCNN_EUNSUPPORTED otherwise).  The fused forwards use it in
 * math mode FRCNN_MATH_F32_WINOGRAD: VGG-16 / RPN trunk layers where frcnn_conv3x3_uses_winograd(cin, cout) != 0
 * (cin >= 128 and cout >= 256: for narrower layers the transformed tensors cost more HBM traffic than the
 * matrix pipe saves -- measured per VGG-16 layer) and ResNet bottlenecks where
 * frcnn_resnet_block_uses_winograd(width, stride) != 0 (the stride-1 blocks of layer3 and layer4: width >= 256). */
int frcnn_conv3x3_uses_winograd(int cin, int cout);
int frcnn_resnet_block_uses_winograd(int width, int stride);
int frcnn_pack_conv3x3_winograd(const float* d_w_oihw, const float* d_row_scale, float* d_u, int cout, int cin, void* stream);
/* The same filter bank from the direct kernels' tap-major pack [9][cout][cin] (frcnn_pack_conv3x3: the train step's master
 * weights).  data_gradient = 0: the layer's own bank [16][cout][cin]; data_gradient = 1: the bank [16][cin][cout] of the
 * convolution that maps the output gradient (cout channels) to the input gradient (cin channels), i.e. of the 180-degree rotated,
 * channel-transposed filter that frcnn_pack_conv3x3_dgrad builds for the direct kernel (autograd of models/vgg16.py:84-96). */
int frcnn_pack_conv3x3_winograd_taps(const float* d_w_packed, float* d_u, int cout, int cin, int data_gradient, void* stream);
size_t frcnn_conv3x3_winograd_workspace_bytes(int n_maps, int H, int W, int cin, int cout);
int frcnn_conv3x3_nhwc_winograd(const float* d_x, const float* d_u, const float* d_bias,
                                float* d_y, int n_maps, int H, int W, int cin, int cout, unsigned flags,
                                void* d_ws, size_t ws_bytes, void* stream);
/* The same Winograd F(2x2,3x3) float32 layer for ONE map as ONE launch with no V / M scratch (csrc/winofused.hip): a
 * block owns 2 x 16 tiles x 64 output channels and all 16 Winograd positions in MFMA accumulators
 * (v_mfma_f32_16x16x4_f32; the position rows are split over wave pairs); the input transform B^T d B is evaluated on the LDS-staged halo when the operand is formed
 * (same float32 operation order as the three-launch form), the output transform A^T M A + bias + ReLU + optional
 * 2x2 max-pool runs on the accumulators.  HBM traffic of a layer = its input, filter bank and output, so the form pays for
 * every 3x3 stride-1 layer with cin % 16 == 0 and cout % 64 == 0 (FRCNN_EUNSUPPORTED otherwise; H*W*cin*4 and 64*cin*cout < 2^31).
 *   d_u : float32 [cin/16][cout/64][16][64][16] (the 16 positions in the kernel's slab order) from frcnn_pack_conv3x3_winograd_fused (OIHW in; optional per-cout
 *         d_row_scale as above) or frcnn_pack_conv3x3_winograd_fused_taps (tap-major [9][cout][cin] in; data_gradient = 1
 *         packs the bank of the data-gradient convolution cout -> cin channels).  16*cout*cin floats, the values of
 *         frcnn_pack_conv3x3_winograd's bank in another order.
 * The fused forwards use it in math mode FRCNN_MATH_F32_WINOGRAD for every single-map 3x3 stride-1 layer
 * (frcnn_conv3x3_uses_winograd_fused(cin, cout) != 0); the three-launch form stays for ResNet's per-RoI 4 x 4 maps. */
int frcnn_conv3x3_uses_winograd_fused(int cin, int cout);
/* ResNet bottleneck 3x3 (width -> width, n_maps maps in one call): != 0 for ONE map, stride 1, width >= 64 -- the blocks of
 * layer1..3 at inference; frcnn_resnet_forward then expects frcnn_pack_conv3x3_winograd_fused's bank in w2 (frozen-BatchNorm
 * scale as d_row_scale).  The per-RoI maps of layer4 keep frcnn_resnet_block_uses_winograd / the three-launch form. */
int frcnn_resnet_block_uses_winograd_fused(int n_maps, int width, int stride);
int frcnn_pack_conv3x3_winograd_fused(const float* d_w_oihw, const float* d_row_scale, float* d_u, int cout, int cin, void* stream);
int frcnn_pack_conv3x3_winograd_fused_taps(const float* d_w_packed, float* d_u, int cout, int cin, int data_gradient, void* stream);
int frcnn_conv3x3_nhwc_winograd_fused(const float* d_x, const float* d_u, const float* d_bias, float* d_y,
                                      int H, int W, int cin, int cout, unsigned flags, void* stream);
/* The same launch over n_maps maps [n_maps][H][W][cin] -> [n_maps][Ho][Wo][cout] (a batch of images through one layer): the tile
 * blocks of the maps follow each other in ONE grid, every map's result is bit-identical to its own single-map call. */
int frcnn_conv3x3_nhwc_winograd_fused_maps(const float* d_x, const float* d_u, const float* d_bias, float* d_y, int n_maps,
                                           int H, int W, int cin, int cout, unsigned flags, void* stream);

/* ------------------------------------------------------------------------------------------
 * Round 3: the f32x6 arithmetic as a general batched GEMM on TILE records ("x6t", csrc/gemm_x6t.hip) and the Winograd layers built
 * on it (csrc/wino_x6.hip).  Replaces the multiply-accumulate of the 512-channel 3x3 convolutions, models/vgg16.py:89-96 and
 * models/rpn.py:88 (cuDNN fp32 in the reference), with fp32-class accuracy on the bf16 matrix pipe.
 *
 * x6t records of a row-major float32 matrix X[R][K], K % 16 == 0, rows padded to `rows_padded` (% 32 == 0):
 *   [K/16][rows_padded/32][3 = hi, mid, lo][1024 B], 1024 B = [k-half 2][row 32][8 bf16]; x = hi + mid + lo exactly.
 *   frcnn_x6t_record_bytes(rows_padded, K)  : bytes of one record array
 *   frcnn_split_rows_x6t                    : [batches][rows][lda] float32 -> [batches] record arrays (rows beyond `rows` zero)
 *   frcnn_gemm_x6t                          : C_b[m][n] = act(bias[n] + residual_b[m][n] + sum_k A_b[m][k] B_b[n][k]), b < batches
 *       (bias / residual may be NULL; the residual has C's layout); A records padded
 *       to a_rows (% FRCNN_X6T_ROW_TILE == 0, >= M), B records padded to b_rows (% FRCNN_X6T_COL_TILE == 0, >= N); batch strides
 *       of the record arrays in BYTES (0: shared by all batches), of C in floats; N % 4 == 0, ldc % 4 == 0; deterministic
 *       (fixed-order split-K when the grid would not cover the chip: d_ws >= frcnn_gemm_x6t_workspace_bytes(M, N, K, batches)).
 * Winograd F(2x2,3x3) layer in this arithmetic (three launches: input transform + exact split -> 16 batched GEMMs -> output
 * transform + bias + ReLU + optional fused 2x2 max-pool), NHWC float32 in and out, cin % 16 == 0, cout % 4 == 0:
 *   frcnn_pack_conv3x3_winograd_x6          : OIHW float32 (optional per-cout scale, as frcnn_pack_conv3x3_winograd) -> the
 *       transformed filter bank as x6t records, frcnn_conv3x3_winograd_x6_pack_bytes(cout, cin) bytes
 *   frcnn_conv3x3_nhwc_winograd_x6          : the layer over n_maps maps [n_maps][H][W][cin]; d_ws >= frcnn_conv3x3_winograd_x6_workspace_bytes(n_maps, H, W, cin, cout)
 *   frcnn_conv3x3_uses_winograd_x6(cin,cout): the layers the fused VGG-16 forward runs this way in math mode
 *       FRCNN_MATH_F32_WINOGRAD when frcnn_forward_params.winograd_x6 != 0 (cin >= 256, cout % 256 == 0).
 * ---------------------------------------------------------------------------------------- */
#define FRCNN_X6T_ROW_TILE 320
#define FRCNN_X6T_COL_TILE 256
size_t frcnn_x6t_record_bytes(int rows_padded, int K);
int frcnn_split_rows_x6t(const float* d_a, int lda, size_t a_batch_floats, void* d_rec, int rows, int rows_padded, int K, int batches,
                         void* stream);
size_t frcnn_gemm_x6t_workspace_bytes(int M, int N, int K, int batches);
int frcnn_gemm_x6t(const void* d_a_rec, int a_rows, size_t a_batch_bytes, const void* d_b_rec, int b_rows, size_t b_batch_bytes,
                   const float* d_bias, const float* d_residual, float* d_c, int ldc, size_t c_batch_floats, int M, int N, int K,
                   int batches, unsigned flags, void* d_ws, size_t ws_bytes, void* stream);
/* ------------------------------------------------------------------------------------------
 * The same batched GEMM in the "f32x3" arithmetic (csrc/gemm_x3t.hip): HALF the matrix instructions of the f32x6 form.  Every operand
 * row carries a power-of-two scale that puts its largest magnitude into [2^14, 2^15); the scaled float32 value is split into two fp16
 * terms x 2^e = hi + lo (|remainder| <= 2^-22 |x 2^e|) and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with float32
 * accumulation; the epilogue multiplies by the two 2^-e (exact).  Operands are held to 22-23 bits relative to their ROW's largest
 * element -- a perturbation below the float32 accumulation error of either matrix pipe (tests/test_gemm_x3t_gpu.py: error against
 * float64 within the exact-f32 kernel's).
 * x3t records of X[R][K]: [K/16][rows_padded/32][2 = hi, lo][1024 B], 1024 B = [k-half 2][row 32][8 fp16]; scales: float32 2^-e per row.
 *   frcnn_rows_scale_x3t  : [batches][rows][lda] float32 -> d_inv_scale [batches][rows_padded] (2^-e; 1 for zero / padding rows)
 *   frcnn_split_rows_x3t  : the records of the rows scaled by 1 / d_inv_scale
 *   frcnn_gemm_x3t        : as frcnn_gemm_x6t, plus the two scale arrays (batch strides in floats, 0 = shared by all batches)
 * ---------------------------------------------------------------------------------------- */
/* A PACKED x3t operand ("blob") = its record arrays followed by its scale arrays: [batches][frcnn_x3t_record_bytes(rows_padded, K)] bytes, then
 * [batches][rows_padded] float32; frcnn_pack_rows_x3t = frcnn_rows_scale_x3t + frcnn_split_rows_x3t into one such allocation of
 * frcnn_x3t_blob_bytes(rows_padded, K, batches) bytes (weights: fc1_w / fc2_w with fc_math_mode FRCNN_FC_F32X3T, rows_padded = 4096).
 * Winograd F(2x2,3x3) layer in this arithmetic (csrc/wino_x3.hip: channel-maximum pass + input transform with one scale per tile,
 * 16 batched GEMMs, output transform), the f32x3 counterpart of the frcnn_*_winograd_x6 entries:
 *   frcnn_pack_conv3x3_winograd_x3 : d_u_f32 = the float32 bank [16][cout][cin] of frcnn_pack_conv3x3_winograd -> blob of
 *                                    frcnn_conv3x3_winograd_x3_pack_bytes(cout, cin) bytes
 *   frcnn_conv3x3_nhwc_winograd_x3 : the layer; d_ws >= frcnn_conv3x3_winograd_x3_workspace_bytes(n_maps, H, W, cin, cout) */
/* A-side producers for convolutions as f32x3 GEMMs (the f32x3 counterparts of frcnn_split_pixels_x6t / frcnn_split_patches3x3_x6t): the row
 * of an output pixel is scaled by the power of two that the channel maxima of its input pixel(s) give.
 *   frcnn_pixel_absmax           : x [pixels][c] -> d_cmax [pixels] = max_c |x|
 *   frcnn_split_pixels_x3t       : 1x1 convolution, stride 1 / 2: records + d_inv_scale [rows_padded]; d_cmax = NULL: the launch reduces the
 *                                  channel maxima of the rows it needs itself (one launch instead of two, the same bytes)
 *   frcnn_split_patches3x3_x3t   : 3x3 / padding 1, stride 1 / 2 (im2col rows, K = 9 c) */
int frcnn_pixel_absmax(const float* d_x, float* d_cmax, long long pixels, int c, void* stream);
int frcnn_split_pixels_x3t(const float* d_x, const float* d_cmax, void* d_rec, float* d_inv_scale, int n_maps, int H, int W, int c, int stride,
                           int rows_padded, void* stream);
int frcnn_split_patches3x3_x3t(const float* d_x, const float* d_cmax, void* d_rec, float* d_inv_scale, int n_maps, int H, int W, int c,
                               int stride, int rows_padded, void* stream);
size_t frcnn_x3t_blob_bytes(int rows_padded, int K, int batches);
int frcnn_pack_rows_x3t(const float* d_a, int lda, size_t a_batch_floats, void* d_blob, int rows, int rows_padded, int K, int batches,
                        void* stream);
size_t frcnn_conv3x3_winograd_x3_pack_bytes(int cout, int cin);
int frcnn_pack_conv3x3_winograd_x3(const float* d_u_f32, void* d_blob, int cout, int cin, void* stream);
size_t frcnn_conv3x3_winograd_x3_workspace_bytes(int n_maps, int H, int W, int cin, int cout);
int frcnn_conv3x3_nhwc_winograd_x3(const float* d_x, const void* d_blob, const float* d_bias, float* d_y, int n_maps, int H, int W, int cin,
                                   int cout, unsigned flags, void* d_ws, size_t ws_bytes, void* stream);
/* The same layer as ONE launch (csrc/wino_x3f.hip, round 4: 64 tiles x 64 output channels x all 16 positions per block; the operand formed in
 * registers from an LDS-staged halo, the filter fragments loaded straight from L2 into registers, output transform in the epilogue; no
 * V / M scratch), for the layers whose scratch does not fit the Infinity Cache (frcnn_forward_params.winograd_x3f_mask: conv2_2 .. conv3_3
 * of VGG-16; with several images in flight also the 512-channel layers).  The same operands, products and accumulation order as
 * frcnn_conv3x3_nhwc_winograd_x3 on the same blob; the output transform combines columns before rows, so the two forms differ by the
 * float32 rounding of that transform only (<= 2e-6 of max|y|, tests/test_gemm_x3t_gpu.py).  cin % 32 == 0, cout % 64 == 0;
 * d_ws >= frcnn_conv3x3_winograd_x3_fused_workspace_bytes (the channel maxima of the input). */
size_t frcnn_conv3x3_winograd_x3_fused_workspace_bytes(int n_maps, int H, int W);
size_t frcnn_conv3x3_winograd_x3_pair_workspace_bytes(int n_maps, int H, int W, int cout);   /* with FRCNN_X3F_PAIR (0: shape not supported) */
int frcnn_conv3x3_nhwc_winograd_x3_fused(const float* d_x, const void* d_blob, const float* d_bias, float* d_y, int n_maps, int H, int W,
                                         int cin, int cout, unsigned flags, void* d_ws, size_t ws_bytes, void* stream);
/* The same two layers CHAINED (round 4): an f32x3 layer takes its scales from the per-pixel channel maximum of its INPUT.  d_cmax_in (optional):
 * those maxima [n_maps][H][W] if a producer already left them -- the layer then does not read its input once more (frcnn_pixel_absmax).
 * d_cmax_out (optional; FRCNN_RELU required, three-launch form: cout % 256 == 0): the channel maxima of the OUTPUT [n_maps][Ho][Wo],
 * accumulated with atomic maxima into a buffer the CALLER ZEROED -- the next layer's d_cmax_in.  Outputs are bit-identical with or
 * without either pointer.  one_launch != 0: csrc/wino_x3f.hip (d_ws as frcnn_conv3x3_nhwc_winograd_x3_fused), else the three-launch form.
 * frcnn_vgg16_forward chains conv2_2 ... conv5_3, the RPN trunk and the RoI pooling this way (11 of 12 channel-maximum passes disappear). */
int frcnn_conv3x3_nhwc_winograd_x3_chain(const float* d_x, const void* d_blob, const float* d_bias, float* d_y, int n_maps, int H, int W, int cin,
                                         int cout, unsigned flags, int one_launch, void* d_ws, size_t ws_bytes, const float* d_cmax_in,
                                         float* d_cmax_out, void* stream);
size_t frcnn_x3t_record_bytes(int rows_padded, int K);
int frcnn_rows_scale_x3t(const float* d_a, int lda, size_t a_batch_floats, float* d_inv_scale, int rows, int rows_padded, int K, int batches,
                         void* stream);
int frcnn_split_rows_x3t(const float* d_a, int lda, size_t a_batch_floats, const float* d_inv_scale, void* d_rec, int rows, int rows_padded,
                         int K, int batches, void* stream);
size_t frcnn_gemm_x3t_workspace_bytes(int M, int N, int K, int batches);
int frcnn_gemm_x3t(const void* d_a_rec, const float* d_a_inv, int a_rows, size_t a_batch_bytes, size_t a_inv_batch_floats,
                   const void* d_b_rec, const float* d_b_inv, int b_rows, size_t b_batch_bytes, size_t b_inv_batch_floats,
                   const float* d_bias, const float* d_residual, float* d_c, int ldc, size_t c_batch_floats, int M, int N, int K,
                   int batches, unsigned flags, void* d_ws, size_t ws_bytes, void* stream);

/* NHWC float32 [N][H][W][C] -> the x6t records of the [N Ho Wo][C] matrix of its pixels taken with `stride` in y and x
 * (Ho = (H - 1) / stride + 1): the A operand of a 1x1 convolution (stride 1 or 2) as a GEMM.  C % 16 == 0. */
int frcnn_split_pixels_x6t(const float* d_x, void* d_rec, int N, int H, int W, int C, int stride, int rows_padded, void* stream);
int frcnn_conv3x3_uses_winograd_x6(int cin, int cout);
size_t frcnn_conv3x3_winograd_x6_pack_bytes(int cout, int cin);
int frcnn_pack_conv3x3_winograd_x6(const float* d_w_oihw, const float* d_row_scale, void* d_u_rec, int cout, int cin, void* stream);
size_t frcnn_conv3x3_winograd_x6_workspace_bytes(int n_maps, int H, int W, int cin, int cout);
int frcnn_conv3x3_nhwc_winograd_x6(const float* d_x, const void* d_u_rec, const float* d_bias, float* d_y, int n_maps, int H, int W, int cin,
                                   int cout, unsigned flags, void* d_ws, size_t ws_bytes, void* stream);
/* NHWC float32 [N][H][W][C] -> the x6t records of the im2col matrix [N Ho Wo][9 C] of a 3x3 convolution with padding 1 and `stride`
 * (Ho = (H - 1) / stride + 1; column index = tap * C + c, tap = 3 r + s; zeros outside the map): the A operand of a strided 3x3
 * convolution as a GEMM against the records of the [cout][9 cin] filter matrix (ResNet layer4.0.conv2).  C % 16 == 0. */
int frcnn_split_patches3x3_x6t(const float* d_x, void* d_rec, int N, int H, int W, int C, int stride, int rows_padded, void* stream);
/* Stand-alone 2x2/stride-2 floor max-pool on NHWC (vgg16.py:78,82,87,92), c % 4 == 0. */
int frcnn_maxpool2x2_nhwc(const float* d_x, float* d_y, int H, int W, int c, void* stream);

/* ------------------------------------------------------------------------------------------
 * ResNet building blocks.  Replace what torchvision.models.resnet{50,101,152} executes under
 * models/resnet.py:33-118 (conv1/bn1/relu/maxpool/layer1-3 as feature extractor, layer4 + spatial
 * mean as the per-RoI head); BatchNorm is always in eval mode there (:58-77,:100-107) and is
 * folded into the preceding convolution at pack time.
 * ---------------------------------------------------------------------------------------- */
/* Fold a frozen BatchNorm2d(gamma, beta, running_mean, running_var, eps) into conv weights
 * OIHW [cout][cin][k][k]: d_w_packed is [k*k][cout][cin] (or [cin*k*k][cout] when cin == 3),
 * d_b_packed [cout]. */
int frcnn_fold_bn_pack(const float* d_w_oihw, const float* d_gamma, const float* d_beta,
                       const float* d_mean, const float* d_var, float eps, int cout, int cin, int ksize,
                       float* d_w_packed, float* d_b_packed, void* stream);
/* Generic NHWC convolution (gather implicit GEMM, f32 MFMA): x [N][H][W][cin] ->
 * y [N][Ho][Wo][cout], square kernel `ksize` (1 or 3), any stride / padding;
 * y = act(conv(x) + bias (+ residual)), residual [N][Ho][Wo][cout] or NULL (Bottleneck's
 * `out += identity`).  cin % 16 == 0, cout % 4 == 0.  d_ws: split-K scratch
 * (frcnn_conv_workspace_bytes; NULL = un-split). */
size_t frcnn_conv_workspace_bytes(int N, int H, int W, int cin, int cout, int ksize, int stride, int pad);
int frcnn_conv_nhwc(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual,
                    float* d_y, int N, int H, int W, int cin, int cout, int ksize, int stride, int pad,
                    unsigned flags, void* d_ws, size_t ws_bytes, void* stream);
/* Stem: 7x7 stride-2 pad-3 conv of the NCHW image [3][H][W] (+folded bn1, ReLU) -> NHWC
 * [(H-1)/2+1][(W-1)/2+1][cout]; d_w_packed [147][cout] from frcnn_fold_bn_pack(cin = 3). */
int frcnn_conv7x7_s2_c3(const float* d_x_chw, const float* d_w_packed, const float* d_bias, float* d_y,
                        int H, int W, int cout, unsigned flags, void* stream);
/* MaxPool2d(3, stride 2, padding 1) on NHWC; output [(H-1)/2+1][(W-1)/2+1][c]. */
int frcnn_maxpool3x3_s2_nhwc(const float* d_x, float* d_y, int H, int W, int c, void* stream);
/* y[n][c] = mean_y(mean_x(x[n][y][x][c]))  (models/resnet.py:117 `.mean(-1).mean(-1)`). */
int frcnn_spatial_mean_nhwc(const float* d_x, float* d_y, int N, int H, int W, int c, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense layers on the f32 MFMA pipe.  Replace nn.Linear (+ReLU) at models/vgg16.py:130-132,
 * models/detector.py:76,78 and the two 1x1 convolutions at models/rpn.py:89-90.
 *   y[m][n] = act( sum_k a[m*lda + k] * w[n*k_dim + k] + bias[n] ),  m < M, n < N
 * w is row-major [n_rows][K] with n_rows >= N rounded up to 128 (extra rows must be readable;
 * frcnn_pack_stack_rows zero-fills them).  K % 16 == 0.  `d_ws` is split-K scratch of at least
 * frcnn_linear_workspace_bytes(M, N, K) bytes (may be NULL if that returns 0).
 * ---------------------------------------------------------------------------------------- */
size_t frcnn_linear_workspace_bytes(int M, int N, int K);
int frcnn_linear(const float* d_a, int lda, const float* d_w, const float* d_bias,
                 float* d_y, int ldy, int M, int N, int K, unsigned flags,
                 void* d_ws, size_t ws_bytes, void* stream);
/* Row softmax over the first `ncls` columns of x[m*ldx ..] -> y[m*ncls ..] (detector.py:77). */
int frcnn_softmax_rows(const float* d_x, int ldx, float* d_y, int M, int ncls, void* stream);

/* ------------------------------------------------------------------------------------------
 * RPN proposal generation.  Replaces models/rpn.py:89 (sigmoid), :98-104 (_extract_valid),
 * :118-123 + models/math_utils.py:99-128 (decode), :129-132 (argsort/flip/top-N),
 * :135-144 (clip, >=16 px filter), :147-153 (torchvision.ops.nms 0.7, first N).
 *
 *   d_head      : float32 [fh*fw][ld_head]; columns 0..8 objectness logits, 9..44 box deltas
 *                 (ty,tx,th,tw per anchor) -- the fused 1x1 head output
 *   d_anchor_map: float32 [fh*fw*9][4];  d_valid_map: float32 [fh*fw*9] or NULL
 *                 (NULL == allow_edge_proposals=True, models/faster_rcnn.py:36)
 * Outputs (device):
 *   d_scores    : float32 [A]       sigmoid objectness per anchor (A = fh*fw*9), reference order
 *   d_sorted_idx: int32   [pre_nms] flat anchor index of the top-N, score-descending; ties are
 *                 broken by HIGHER anchor index first (what stable-ascending argsort + flip gives)
 *   d_props     : float32 [post_nms][4] kept proposals (y1,x1,y2,x2), rows >= *d_n_props zeroed
 *   d_counts    : int32 [4] = { n_candidates (<=pre_nms), n_after_size_filter, n_props, 0 }
 * `ctx` supplies scratch (keys, decoded boxes, NMS mask).  pre_nms <= 16384, post_nms <= 2048.
 * ---------------------------------------------------------------------------------------- */
typedef struct frcnn_ctx frcnn_ctx;

int frcnn_rpn_proposals(frcnn_ctx* ctx, const float* d_head, int ld_head,
                        const float* d_anchor_map, const float* d_valid_map,
                        int fh, int fw, int image_h, int image_w,
                        int pre_nms, int post_nms, float nms_threshold, float min_side,
                        float* d_scores, int32_t* d_sorted_idx, float* d_props, int32_t* d_counts,
                        void* stream);

/* Stand-alone greedy NMS, float32 boxes (any consistent corner order), replaces
 * torchvision.ops.nms as called at models/rpn.py:147-151: scores are sorted stably descending,
 * box j is suppressed by an earlier kept box i iff inter/(area_i+area_j-inter) > threshold.
 * d_keep receives up to max_keep indices into the INPUT order (score-descending), d_n_keep the
 * count.  n <= 16384. */
int frcnn_nms(frcnn_ctx* ctx, const float* d_boxes, const float* d_scores, int n, float threshold,
              int max_keep, int32_t* d_keep, int32_t* d_n_keep, void* stream);

/* ------------------------------------------------------------------------------------------
 * RoI max pooling.  Replaces torchvision.ops.RoIPool((7,7), 1/16) at models/detector.py:27,72
 * including the (y1,x1,y2,x2)->(x1,y1,x2,y2) column swap at :65-69.
 *   d_fm   : float32 NHWC [fh][fw][c];   d_rois : float32 [max_rois][4] (y1,x1,y2,x2) pixels
 *   d_n_rois: device int32, rows >= *d_n_rois produce zeros
 *   d_out  : float32 [max_rois][pooled][pooled][c]  (NHWC per RoI)
 * ---------------------------------------------------------------------------------------- */
int frcnn_roi_pool(const float* d_fm, int fh, int fw, int c, const float* d_rois,
                   const int32_t* d_n_rois, int max_rois, int pooled, float spatial_scale,
                   float* d_out, void* stream);
/* The same pooling writing fc1's operand in the f32x3 arithmetic directly (csrc/roipool.hip: roi_scale_x3t_kernel, roi_pool_x3t_kernel; what
 * frcnn_vgg16_forward runs with fc_math_mode FRCNN_FC_F32X3T): x3t records of the matrix [rec_rows][pooled * pooled * c], k = (ph * pooled +
 * pw) * c + channel, one power-of-two scale per RoI taken from the maximum of |fm| over the UNION OF THE RoI'S BINS (the bins' own float32
 * floor / ceil edges, which can reach one cell past round(x2 / 16): tests/test_gemm_x3t_gpu.py).
 *   d_cmax : scratch, float32 [fh * fw] (per-cell channel maximum of |fm|, written here);  d_inv_scale : float32 [rec_rows] (2^-e, written)
 *   d_rec  : frcnn_x3t_record_bytes(rec_rows, pooled * pooled * c) bytes;  rec_rows % 32 == 0, >= max_rois;  c % 16 == 0 */
int frcnn_roi_pool_x3t(const float* d_fm, int fh, int fw, int c, const float* d_rois, const int32_t* d_n_rois, int max_rois, int pooled,
                       float spatial_scale, float* d_cmax, float* d_inv_scale, void* d_rec, int rec_rows, void* stream);
/* RoIAlign with torchvision.ops.roi_align's semantics (csrc/roialign.hip) -- the pooling BASELINE.json's north_star names; the
 * reference pools with RoIPool (models/detector.py:16,27), so this is an option beyond it (DetectorNetwork(pooling="align")).
 * Same tensors as frcnn_roi_pool; sampling_ratio = samples per bin and axis (1 or 2; <= 0: adaptive ceil(roi_size / pooled)),
 * aligned = the half-pixel shift of torchvision's `aligned=True`.  float32, the operation order of torchvision's kernel.
 * frcnn_roi_align_backward: d_dfm [fh][fw][c] (+)= gradient of d_dout [n_rois][pooled][pooled][c] with respect to the feature
 * map, gathered per cell in a fixed order (deterministic; torchvision scatters with atomicAdd).  pooled <= 14. */
int frcnn_roi_align(const float* d_fm, int fh, int fw, int c, const float* d_rois, const int32_t* d_n_rois, int max_rois,
                    int pooled, float spatial_scale, int sampling_ratio, int aligned, float* d_out, void* stream);
int frcnn_roi_align_backward(const float* d_rois, int n_rois, int fh, int fw, int c, int pooled, float spatial_scale,
                             int sampling_ratio, int aligned, const float* d_dout, float* d_dfm, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Final detections.  Replaces models/faster_rcnn.py:179-224 (the numpy float64 decode with
 * stds [.1,.1,.2,.2], clip to [0,H-1]/[0,W-1], score > threshold, per-class
 * torchvision.ops.nms(0.3) on float64 boxes) without the 20 host round trips.
 *   d_props  : float32 [max_rois][4]; d_classes: float32 [max_rois][ncls] (softmax);
 *   d_deltas : float32 [max_rois][(ncls-1)*4]; d_n_rois: device int32
 * Outputs:
 *   d_out    : float64 [(ncls-1)][max_rois][5]  rows (y1,x1,y2,x2,score) in NMS (score-desc) order
 *   d_out_cnt: int32   [(ncls-1)]               rows valid per class (class c -> index c-1)
 * max_rois <= 512.
 * ---------------------------------------------------------------------------------------- */
int frcnn_detections(const float* d_props, const float* d_classes, const float* d_deltas,
                     const int32_t* d_n_rois, int max_rois, int ncls,
                     int image_h, int image_w, float score_threshold, float nms_threshold,
                     double* d_out, int32_t* d_out_cnt, void* stream);

/* ------------------------------------------------------------------------------------------
 * Context + fused VGG-16 forward.  Replaces FasterRCNNModel.forward
 * (models/faster_rcnn.py:80-132) for the VGG-16 backbone (models/vgg16.py:22-158): one call
 * enqueues every kernel of stage 1-3 on `stream`, no host round trip.
 * One ctx per in-flight image (it owns the activation ping-pong buffers and scratch);
 * a ctx is not re-entrant, different ctxs are independent.  The slab is allocated by
 * frcnn_ctx_create; the V / M scratch of the Winograd layers (FRCNN_MATH_F32_WINOGRAD) is added by the
 * first forward of the ctx that runs in that mode (one hipMalloc, sized for the ctx's largest image:
 * 307 MB at 600x1000) and counted by frcnn_ctx_bytes from then on.
 * ---------------------------------------------------------------------------------------- */
int  frcnn_ctx_create(frcnn_ctx** out, int max_image_h, int max_image_w, int max_rois);
/* A ctx that owns ONLY the proposal scratch (~50 MB: keys, decoded boxes, NMS bit matrix) for the stage-level entry points
 * frcnn_rpn_proposals / frcnn_nms; the fused forwards return FRCNN_EINVAL on it.  Not re-entrant: one per stream. */
int  frcnn_ctx_create_proposals(frcnn_ctx** out, int max_image_h, int max_image_w);
void frcnn_ctx_destroy(frcnn_ctx* ctx);
size_t frcnn_ctx_bytes(const frcnn_ctx* ctx);

typedef struct frcnn_vgg16_weights {
    const float* conv_w[13];   /* [0]: frcnn_pack_conv3x3_c3, [1..12]: frcnn_pack_conv3x3     */
    const float* conv_b[13];
    const float* rpn_conv_w;   /* frcnn_pack_conv3x3 of _rpn_conv1                            */
    const float* rpn_conv_b;
    const float* rpn_head_w;   /* frcnn_pack_stack_rows(_rpn_class, _rpn_boxes) -> [128][512] */
    const float* rpn_head_b;   /* [128]                                                       */
    const float* fc1_w;        /* frcnn_pack_fc_chw_to_hwc(_fc1) [4096][25088]; FRCNN_FC_F32X6T / _F32X3T: its records */
    const float* fc1_b;
    const float* fc2_w;        /* [4096][4096] as stored; FRCNN_FC_F32X6T / _F32X3T: its records */
    const float* fc2_b;
    const float* head_w;       /* frcnn_pack_stack_rows(_classifier, _regressor) -> [128][4096] */
    const float* head_b;       /* [128]                                                       */
    int32_t num_classes;       /* 21 for VOC                                                  */
} frcnn_vgg16_weights;

typedef struct frcnn_forward_params {
    int32_t pre_nms;            /* 6000  (models/faster_rcnn.py:124) */
    int32_t post_nms;           /* 300   (models/faster_rcnn.py:125) */
    float   rpn_nms_threshold;  /* 0.7   (models/rpn.py:150)         */
    float   min_side;           /* 16    (models/rpn.py:142)         */
    int32_t allow_edge_proposals; /* 1   (models/faster_rcnn.py:36)  */
    int32_t math_mode;          /* FRCNN_MATH_F32 (exact f32 MFMA, direct) or FRCNN_MATH_F32_WINOGRAD;
                                   selects how the 3x3 conv weight pointers of the weights struct are interpreted:
                                   FRCNN_MATH_F32: frcnn_pack_conv3x3; FRCNN_MATH_F32_WINOGRAD: layers with
                                   frcnn_conv3x3_uses_winograd_fused(cin, cout) (every 3x3 layer from conv1_2 on, the RPN trunk;
                                   ResNet: frcnn_resnet_block_uses_winograd_fused blocks) frcnn_pack_conv3x3_winograd_fused,
                                   ResNet layer4 blocks with frcnn_resnet_block_uses_winograd: frcnn_pack_conv3x3_winograd,
                                   frcnn_pack_conv3x3 otherwise */
    int32_t conv_blocks_target; /* split-K granularity of the 3x3 layers: blocks per launch to aim for.  0 = 1280 (best
                                   latency for one image on the chip); ~320 when many images are in flight on separate
                                   streams (other images' kernels fill the tail, longer work units win) */
    int32_t fc_math_mode;       /* VGG-16 detector fc1 / fc2: FRCNN_FC_F32 (exact f32 MFMA, fc1_w / fc2_w = float32 matrices), FRCNN_FC_F32X6T or
                                   FRCNN_FC_F32X3T (record operands, see the defines below) */
    int32_t roi_op;             /* FRCNN_ROI_POOL (the reference: torchvision RoIPool, models/detector.py:27) or FRCNN_ROI_ALIGN
                                   (torchvision roi_align semantics, aligned = 0) */
    int32_t roi_sampling_ratio; /* FRCNN_ROI_ALIGN: samples per bin and axis (1 or 2; <= 0 adaptive); ignored for FRCNN_ROI_POOL */
    int32_t winograd_tile_rows; /* three-launch Winograd form only (ResNet layer4): row count of the batched GEMM's block tile.  0 = 64 (64 x 128 tiles, five
                                   blocks per CU: best latency for one image on the chip); 128 with many images in flight (the chip
                                   is then at its power limit and the tile with fewer operand bytes per MFMA wins) */
    int32_t winograd_x6_mask;   /* FRCNN_MATH_F32_WINOGRAD only: bit i set = 3x3 layer i runs as an x6 Winograd layer (csrc/wino_x6.hip: the position
                                   GEMMs in the f32x6 arithmetic on the bf16 pipe) and its weight pointer is frcnn_pack_conv3x3_winograd_x6's record bank.
                                   VGG-16: bit i = conv_w[i] (1 .. 12: conv1_2 .. conv5_3), bit 13 = the RPN trunk.  ResNet: bit 13 = the RPN trunk only.
                                   A set bit on a layer with frcnn_conv3x3_uses_winograd_x6(cin, cout) == 0 is FRCNN_EINVAL.  0 = rounds 1-2's behaviour */
    int32_t x6_gemm_tiles;      /* block tile of the x6 Winograd layers' batched GEMM (csrc/gemm_x6t.hip): 0 = chosen per shape by the cost model (best
                                   latency for one image on the chip: 320 x 256 tiles, one 8-wave block per CU, where they cover the chip in one
                                   round), 1 = 320 x 256 always, 2 = 160 x 128 always (4-wave blocks that leave registers and LDS for a second kernel on
                                   the CU: with several images in flight the transforms, epilogues and proposal kernels of the other images then
                                   overlap the GEMM -- measured +3..8 % images/sec at 3 images in flight) */
    int32_t winograd_x3_mask;   /* a subset of winograd_x6_mask (VGG-16): the layers whose position GEMMs run in the f32x3 arithmetic instead (two
                                   fp16 terms per row-scaled operand, three MFMAs per product: csrc/wino_x3.hip); their weight pointers are
                                   frcnn_pack_conv3x3_winograd_x3's blobs.  ResNet: 0 */
    int32_t winograd_x3f_mask;  /* round 4 (ABI 9), VGG-16, FRCNN_MATH_F32_WINOGRAD only: bit i set = 3x3 layer i (1 = conv1_2 .. 12, 13 = the RPN trunk; disjoint from
                                   winograd_x6_mask) runs as a ONE-LAUNCH Winograd layer in the f32x3 arithmetic (csrc/wino_x3f.hip: operand formed in
                                   registers from the LDS-staged halo, filter fragments straight from L2, all 16 positions in accumulators, no V / M
                                   scratch; the three-launch f32x3 layer's arithmetic up to the rounding order of the output transform) and its weight
                                   pointer is frcnn_pack_conv3x3_winograd_x3's blob.  For the layers whose V + M scratch does not fit the Infinity
                                   Cache (conv2_1 .. conv3_3) and, when several images are in flight and the chip is full anyway, for the 512-channel
                                   layers too (one launch instead of three, no scratch traffic).  cin % 32 == 0, cout % 64 == 0 */
    int32_t winograd_x3p_mask;  /* round 6 (ABI 14): a subset of winograd_x3f_mask -- the one-launch f32x3 layers that run in the TWO-PASS form with 128 output
                                   channels per block (FRCNN_X3F_PAIR, csrc/wino_x3p.hip; cin >= 64, cout % 128 == 0).  Same results bit for bit;
                                   the ctx keeps the scratch the first pass's accumulators rest in (256 KB per block of the layer's grid) */
} frcnn_forward_params;
#define FRCNN_X6_RPN_TRUNK_BIT 13
/* capacity of the detector heads: classifier (n) + regressor (4 n - 4) rows are stacked into one zero-padded GEMM operand of
 * ceil((5 n - 4) / 128) * 128 rows (head_w / head_b of the weight structs), at most FRCNN_HEAD_LD_MAX */
#define FRCNN_HEAD_LD_MAX 512
#define FRCNN_MAX_NUM_CLASSES 103
#define FRCNN_MATH_F32   0
/* (1 was FRCNN_MATH_F32X6, the direct f32x6 convolution of round 2: removed in ABI 13) */
#define FRCNN_MATH_F32_WINOGRAD 2
#define FRCNN_ROI_POOL  0
#define FRCNN_ROI_ALIGN 1
#define FRCNN_FC_F32   0
/* (1 was FRCNN_FC_F32X6, round 2's chunk-major-record kernel: removed in ABI 13) */
#define FRCNN_FC_F32X6T 2     /* the f32x6 arithmetic (three bf16 terms per operand, six MFMAs per product) on csrc/gemm_x6t.hip (fc1_w / fc2_w = frcnn_split_rows_x6t records of the
                                 same matrices, rows padded to FRCNN_X6T_COL_TILE; any number of RoIs the ctx holds) */
#define FRCNN_FC_F32X3T 3     /* the f32x3 arithmetic on csrc/gemm_x3t.hip (two fp16 terms per row-scaled operand, three MFMAs per product):
                                 fc1_w / fc2_w = frcnn_pack_rows_x3t blobs of the same matrices (rows_padded = 4096) */

/* d_image: float32 NCHW [3][H][W] (preprocessed as models/vgg16.py:146 prescribes).
 * d_anchor_map / d_valid_map: optional caller-provided maps (faster_rcnn.py:113-115); NULL =
 * generate (and cache per shape) inside the ctx.
 * Outputs (device, caller-owned): d_props [post_nms][4], d_classes [post_nms][ncls],
 * d_deltas [post_nms][(ncls-1)*4], d_counts int32[4] as in frcnn_rpn_proposals. */
int frcnn_vgg16_forward(frcnn_ctx* ctx, const frcnn_vgg16_weights* w, const frcnn_forward_params* p,
                        const float* d_image, int H, int W,
                        const float* d_anchor_map, const float* d_valid_map,
                        float* d_props, float* d_classes, float* d_deltas, int32_t* d_counts,
                        void* stream);

/* Fused ResNet forward: FasterRCNNModel.forward (models/faster_rcnn.py:80-132) for the
 * ResNet-50/101/152 backbones (models/resnet.py:131-185): feature map [ceil(H/16)][ceil(W/16)][1024],
 * RPN on 1024 channels, RoIPool 7x7, layer4 per RoI + spatial mean -> 2048, heads. */
#define FRCNN_RESNET_MAX_BLOCKS 64
typedef struct frcnn_bottleneck_weights {
    const float *w1, *b1;      /* 1x1 cin->width    (bn folded), [1][width][cin]   */
    const float *w2, *b2;      /* 3x3 width->width, stride on this conv (v1.5), [9][width][width]; in math mode
                                  FRCNN_MATH_F32_WINOGRAD the blocks with frcnn_resnet_block_uses_winograd(width, stride)
                                  carry frcnn_pack_conv3x3_winograd's [16][width][width] instead */
    const float *w3, *b3;      /* 1x1 width->cout, [1][cout][width] */
    const float *wd, *bd;      /* downsample 1x1 cin->cout (stride), NULL when identity */
    int32_t cin, width, cout, stride;
    int32_t x6_mask;           /* round 3: bit 0 / 1 / 2 set = w1 / w3 / wd is NOT a float32 pack but the x6t record array of the folded
                                  [cout][cin] matrix (frcnn_split_rows_x6t, rows padded to FRCNN_X6T_COL_TILE): that 1x1 convolution runs as a
                                  GEMM in the f32x6 arithmetic on the bf16 pipe (csrc/gemm_x6t.hip; the activations are split on the fly by
                                  frcnn_split_pixels_x6t).  Needs cin % 16 == 0 and cout % 4 == 0; FRCNN_MATH_F32_WINOGRAD only */
    int32_t x3_mask;           /* a subset of x6_mask: those convolutions run in the f32x3 arithmetic instead (csrc/gemm_x3t.hip, csrc/wino_x3.hip) and
                                  their weight pointers are f32x3 blobs: frcnn_pack_rows_x3t of the [cout][K] matrix (1x1 and stride-2 3x3
                                  convolutions), frcnn_pack_conv3x3_winograd_x3's blob (stride-1 3x3) */
    const float* wmax;         /* round 4 (ABI 11), with g3 != 0: device float[4] = max |w1|, |w2|, |w3|, |wd| of the packs (upper bounds; [3] unused without wd) */
    int32_t g3;                /* != 0 (2, ABI 15: w1 / w2 / w3 / wd are frcnn_pack_conv_x3g_weights images of the float32 packs, same results): the block's four convolutions run in the f32x3 arithmetic under ONE scale per tensor (frcnn_conv_nhwc_x3g's kernel,
                                  csrc/conv_gather.hip): w1 / w2 / w3 / wd are the plain float32 packs ([1][width][cin], [9][width][width], ...) whatever the
                                  math mode, x6_mask must be 0.  The activation maxima travel from one convolution's epilogue to the next one's scale inside
                                  the context.  Reference: the same Bottleneck forward, models/resnet.py:38-46 / :66-90 */
    int32_t reserved0;
} frcnn_bottleneck_weights;
#define FRCNN_X6_CONV1 1
#define FRCNN_X6_CONV3 2
#define FRCNN_X6_DOWN  4
#define FRCNN_X6_CONV2 8     /* w2 = x6t records: stride 1 -> frcnn_pack_conv3x3_winograd_x6's bank (an x6 Winograd layer over the block's maps),
                                stride 2 -> the records of the folded [cout][9 cin] matrix (tap-major columns; an im2col GEMM) */

typedef struct frcnn_resnet_weights {
    const float* stem_w;       /* [147][64] */
    const float* stem_b;
    int32_t n_blocks[4];       /* layer1..layer4: (3,4,6,3) / (3,4,23,3) / (3,8,36,3) */
    frcnn_bottleneck_weights blocks[FRCNN_RESNET_MAX_BLOCKS];   /* layer1, layer2, layer3, layer4 in order */
    const float* rpn_conv_w;   /* frcnn_pack_conv3x3 (FRCNN_MATH_F32) / frcnn_pack_conv3x3_winograd (FRCNN_MATH_F32_WINOGRAD) of
                                  _rpn_conv1 (1024 -> 1024) */
    const float* rpn_conv_b;
    const float* rpn_head_w;   /* [128][1024] */
    const float* rpn_head_b;
    const float* head_w;       /* [128][2048] */
    const float* head_b;
    int32_t num_classes;
} frcnn_resnet_weights;

int frcnn_resnet_forward(frcnn_ctx* ctx, const frcnn_resnet_weights* w, const frcnn_forward_params* p,
                         const float* d_image, int H, int W,
                         const float* d_anchor_map, const float* d_valid_map,
                         float* d_props, float* d_classes, float* d_deltas, int32_t* d_counts,
                         void* stream);

/* The same forward in two calls, for a true batch of images (BASELINE configs[2] "batch=8"; the reference asserts batch 1 at
 * models/faster_rcnn.py:108, SURVEY 8b allows lifting it):
 *   frcnn_resnet_backbone          conv1 / bn1 / relu / maxpool / layer1..3 (models/resnet.py:38-46) over n_images images
 *                                  [n][3][H][W] -> d_features [n][fh][fw][1024] (NHWC float32) with EVERY bottleneck launch covering
 *                                  the n maps: the 1x1 convolutions become GEMMs over n * h * w pixels (one map of 38 x 63 fills 38
 *                                  tiles of a 256-CU chip).  ctx: frcnn_ctx_create_backbone(max_h, max_w, max_images) -- only the
 *                                  activation rotation and the gather kernel's split-K scratch (the fused forwards refuse it) -- or any
 *                                  full ctx with n_images == 1.
 *   frcnn_resnet_forward_features  RPN + RoI pooling + layer4 + heads (models/faster_rcnn.py:116-132) of ONE image from its feature
 *                                  map (copied into the ctx's own map unless it already is frcnn_ctx_tensor(ctx, 0)); H, W = the image's
 *                                  size as in frcnn_resnet_forward.
 * frcnn_resnet_forward(image) == frcnn_resnet_backbone(1 image) + frcnn_resnet_forward_features bit for bit; with n > 1 the split-K
 * factors of the under-filled GEMMs differ (fewer splits are needed), so feature maps agree to float32 rounding (~1e-6 relative),
 * deterministically. */
int frcnn_ctx_create_backbone(frcnn_ctx** out, int max_image_h, int max_image_w, int max_images);
int frcnn_resnet_backbone(frcnn_ctx* ctx, const frcnn_resnet_weights* w, const frcnn_forward_params* p,
                          const float* d_images, int n_images, int H, int W, float* d_features, void* stream);
int frcnn_resnet_forward_features(frcnn_ctx* ctx, const frcnn_resnet_weights* w, const frcnn_forward_params* p,
                                  const float* d_feature_map, int H, int W,
                                  const float* d_anchor_map, const float* d_valid_map,
                                  float* d_props, float* d_classes, float* d_deltas, int32_t* d_counts,
                                  void* stream);
/* The same tail in THREE calls (round 6, ABI 14): frcnn_resnet_rpn_roipool = RPN + proposals + RoI pooling (models/rpn.py:88-153, detector.py:65-72) of ONE
 * image from its feature map, into d_roi_out = that image's slice [post_nms][7][7][C] of a batch buffer (its own ctx and stream, as
 * frcnn_resnet_forward_features); frcnn_resnet_head = layer4 + spatial mean + classifier / regressor (models/resnet.py:109-118, detector.py:75-78) over the
 * n_rois pooled RoIs of ALL images in one set of launches (ctx: frcnn_ctx_create_head(max_rois_total)): d_classes [n_rois][num_classes],
 * d_deltas [n_rois][4 (num_classes - 1)].  Rows are independent of each other; with the per-tensor-scaled f32x3 blocks (frcnn_bottleneck_weights.g3)
 * the scales are those of the batch, so a batch's rows differ from the per-image call's at the rounding level (as frcnn_resnet_backbone's). */
int frcnn_resnet_rpn_roipool(frcnn_ctx* ctx, const frcnn_resnet_weights* w, const frcnn_forward_params* p, const float* d_feature_map, int H,
                             int W, const float* d_anchor_map, const float* d_valid_map, float* d_props, int32_t* d_counts, float* d_roi_out,
                             void* stream);
int frcnn_ctx_create_head(frcnn_ctx** out, int max_rois_total);
int frcnn_resnet_head(frcnn_ctx* ctx, const frcnn_resnet_weights* w, const frcnn_forward_params* p, float* d_rois, int n_rois, float* d_classes,
                      float* d_deltas, void* stream);

/* ==========================================================================================
 * Training path (SURVEY.md section 8 rows f2 + f3): FasterRCNNModel.train_step,
 * models/faster_rcnn.py:228-362.  The reference relies on autograd over cuDNN/cuBLAS; here every
 * backward operator is an explicit entry point and fasterrcnn_amd/models/faster_rcnn.py chains them.
 * All arithmetic is float32 on the exact-f32 matrix pipe; loss sums are float64, rounded once.
 * ======================================================================================== */

/* models/faster_rcnn.py:421-510 _label_proposals.  Rows 0..n_props-1 are the RPN proposals
 * (n_props read from the device, clamped to max_props), followed by the n_gt ground-truth boxes
 * (:433).  For every row whose best IoU >= min_background_iou (order preserved, :476-480):
 *   d_out_props         float32 [K][4]
 *   d_out_class_idx     int32   [K]       0 = background (best IoU < min_object_iou, :483)
 *   d_out_gt_classes    float32 [K][num_classes]            one-hot (:486-488)
 *   d_out_gt_box_deltas float32 [K][2][4*(num_classes-1)]   [:,0,:] mask, [:,1,:] (ty,tx,th,tw) tiled (:497-510)
 *   d_out_count         int32   K
 * Output buffers hold max_props + n_gt rows.  IoU as math_utils.py:39-63 in float32 (eps 1e-7). */
int frcnn_label_proposals(const float* d_props, const int32_t* d_n_props, int max_props,
                          const float* d_gt_boxes, const int32_t* d_gt_class_idx, int n_gt, int num_classes,
                          float min_background_iou, float min_object_iou,
                          const float box_delta_means[4], const float box_delta_stds[4],
                          float* d_out_props, int32_t* d_out_class_idx, float* d_out_gt_classes,
                          float* d_out_gt_box_deltas, int32_t* d_out_count, void* stream);

/* dst[i][:] = src[idx[i]][:] -- the index selects of faster_rcnn.py:557-561 (_sample_proposals; the
 * random permutation itself is drawn on the host exactly as the reference draws it). */
int frcnn_gather_rows(const float* d_src, const int32_t* d_idx, int n, int row_floats, float* d_dst, void* stream);

/* models/rpn.py:176-272 class_loss + regression_loss over the anchor mini-batch, and the gradient of
 * (class_loss + regression_loss) with respect to the RPN head output.
 *   d_head    float32 [cells][ld_head]: [9 objectness logits | 36 box deltas | pad] per feature-map cell
 *   d_sample  int32 [n_sample]: flat anchor indices (y*fw + x)*9 + k marked trainable by
 *             faster_rcnn.py:364-419 _sample_rpn_minibatch (no duplicates)
 *   d_rpn_map float32 [A][6] (trainable, object, ty, tx, th, tw)  (frcnn_rpn_targets layout)
 *   d_losses  float32 [2] = (class, regression);  d_grad_head [cells][ld_head] or NULL (fully written). */
int frcnn_rpn_loss(const float* d_head, int ld_head, int cells, const int32_t* d_sample, int n_sample,
                   const float* d_rpn_map, float* d_losses, float* d_grad_head, void* stream);

/* models/detector.py:83-155 class_loss + regression_loss and the gradient with respect to the stacked
 * head output [class logits (num_classes) | regressor outputs (4*(num_classes-1)) | pad to ld_grad].
 *   d_classes [n][num_classes] softmax output, d_deltas [n][4*(num_classes-1)],
 *   d_gt_classes / d_gt_box_deltas as produced by frcnn_label_proposals. */
int frcnn_detector_loss(const float* d_classes, const float* d_deltas, const float* d_gt_classes,
                        const float* d_gt_box_deltas, int n, int num_classes, float* d_losses,
                        float* d_grad_logits, int ld_grad, void* stream);

/* C[m][n] = sum_r A[r][m] * B[r][n]  (m < M, n < N, r < R): autograd's linear backward
 * (weight gradient dY^T X; data gradient with A = dY^T).  lda, ldb multiples of 4, ldc even,
 * A/B 16-byte aligned; rows may be read up to their leading dimension.  Deterministic split-R
 * through d_ws (frcnn_gemm_tn_workspace_bytes; NULL = no split). */
size_t frcnn_gemm_tn_workspace_bytes(int M, int N, int R);
int frcnn_gemm_tn(const float* d_a, int lda, const float* d_b, int ldb, float* d_c, int ldc,
                  int M, int N, int R, void* d_ws, size_t ws_bytes, void* stream);

/* The reduced-precision train step (BASELINE configs[4]; the reference trains in float32: faster_rcnn.py:355, so this is
 * beyond it): the same three gradient GEMMs with `grad_math`
 *   FRCNN_GRAD_F32  : the entry points without the suffix (exact-f32 matrix pipe)
 *   FRCNN_GRAD_BF16 : both operands rounded to bfloat16 (round to nearest even) on their way to the bf16 matrix pipe, float32
 *                     accumulation; result = f32 GEMM of the rounded operands up to the accumulation order
 *                     (oracle/train_oracle.py grad_math="bf16").  Operand arrays < 4 GiB.
 * Workspace sizes: the functions without the suffix cover both. */
#define FRCNN_GRAD_F32  0
#define FRCNN_GRAD_BF16 1
int frcnn_gemm_tn_math(const float* d_a, int lda, const float* d_b, int ldb, float* d_c, int ldc,
                       int M, int N, int R, int grad_math, void* d_ws, size_t ws_bytes, void* stream);
int frcnn_conv3x3_wgrad_math(const float* d_x, const float* d_dz, float* d_dwp, int H, int W, int cin, int cout,
                             int grad_math, void* d_ws, size_t ws_bytes, void* stream);
int frcnn_conv_wgrad_math(const float* d_x, const float* d_dz, float* d_dwp, int N, int H, int W, int cin, int cout,
                          int ksize, int stride, int pad, int grad_math, void* d_ws, size_t ws_bytes, void* stream);
/* ABI 10: the FORWARD and DATA-GRADIENT convolutions of the ResNet train step (models/faster_rcnn.py:228-362 train_step over the Bottleneck
 * convolutions of models/resnet.py:38-46,110 and their autograd backward) with the same switch (BASELINE configs[4] as written:
 * "train step ... bf16"): frcnn_conv_nhwc / frcnn_conv_dgrad with `math` = FRCNN_GRAD_BF16 round both operands (activations or output
 * gradients, and the folded weight pack) to bfloat16 on their way into LDS and multiply on the bf16 matrix pipe (one
 * v_mfma_f32_32x32x16_bf16 per 16-channel stage where the float32 kernel issues eight float32 instructions), float32 accumulation,
 * bias / residual / ReLU in float32; master weights and every stored tensor stay float32.  Same arguments, shapes and workspace as the
 * entry points without the suffix. */
int frcnn_conv_nhwc_math(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual, float* d_y,
                         int N, int H, int W, int cin, int cout, int ksize, int stride, int pad, unsigned flags, int math,
                         void* d_ws, size_t ws_bytes, void* stream);
int frcnn_conv_dgrad_math(const float* d_dz, const float* d_wd, const float* d_residual, float* d_dx, int N, int H, int W,
                          int cin, int cout, int ksize, int stride, int pad, int math, void* d_ws, size_t ws_bytes, void* stream);

/* ABI 16: the backward of ONE trainable Bottleneck (models/resnet.py:38-46 with frozen BatchNorm folded into the convolutions; the reference
 * gets it from autograd: pytorch/FasterRCNN/__main__.py:147-152 `loss.total.backward()`) as one call instead of ~25:
 *   out = relu(conv3(t2) + identity), t2 = relu(conv2(t1)), t1 = relu(conv1(x)), identity = x or downsample(x)
 * d_g = dL/d out on entry (overwritten: the ReLU mask of `out` is applied in place).  Writes, per convolution, the gradient of its RAW
 * weight (the folded weight's gradient x the BatchNorm scale of the output channel) into frcnn_train_conv.grad, and, if d_dx is not NULL,
 * dL/dx.  d_dt2 [N][Ho][Wo][width], d_dt1 [N][H][W][width], d_dxid [N][H][W][cin] (downsample blocks) are scratch the caller owns;
 * frcnn_train_conv.wd is the convolution's data-gradient pack scratch ([k k][cin][cout] floats).  The chain ReLU mask -> data gradient ->
 * ReLU mask ... runs on `stream`; the four weight gradients (GEMM + split reduction + row scale) run on `side_stream` behind ONE event
 * recorded on `stream` after the chain (NULL: on `stream` too, each before its data gradient) -- they execute under the next block's
 * chain, and the caller orders whatever reads the gradients, or frees the scratch, behind `side_stream`.  The values are those
 * of the separate entry points (frcnn_relu_backward, frcnn_conv_wgrad_math + frcnn_scale_rows, frcnn_pack_conv_dgrad + frcnn_conv_dgrad_math)
 * bit for bit.  (H, W): the block input's size; (Ho, Wo): its output's (stride of conv2 / downsample).  math: FRCNN_GRAD_F32 / _BF16. */
typedef struct frcnn_train_conv {
    const float* folded;   /* [k k][cout][cin]: W x BN scale (what the forward convolves with) */
    const float* scale;    /* [cout] BatchNorm scale gamma / sqrt(var + eps) */
    float* grad;           /* out: [k k][cout][cin] gradient of the raw weight */
    float* wd;             /* scratch: [k k][cin][cout] */
    int32_t cin, cout, ksize, stride, pad, reserved0;
} frcnn_train_conv;
int frcnn_bottleneck_backward_workspace_bytes(const frcnn_train_conv* c1, const frcnn_train_conv* c2, const frcnn_train_conv* c3,
                                              const frcnn_train_conv* cd, int N, int H, int W, int Ho, int Wo, size_t* main_bytes,
                                              size_t* side_bytes);
int frcnn_bottleneck_backward(const frcnn_train_conv* c1, const frcnn_train_conv* c2, const frcnn_train_conv* c3, const frcnn_train_conv* cd,
                              const float* d_x, const float* d_t1, const float* d_t2, const float* d_out, float* d_g, float* d_dt2,
                              float* d_dt1, float* d_dxid, float* d_dx, int N, int H, int W, int Ho, int Wo, int math, void* d_ws_main,
                              size_t ws_main_bytes, void* d_ws_side, size_t ws_side_bytes, void* stream, void* side_stream);

/* ABI 11: the same convolution (frcnn_conv_nhwc: the frozen-BN Bottleneck convolutions of models/resnet.py:38-46, BN folded) in the f32x3
 * arithmetic under ONE power-of-two scale per tensor: every activation and weight value as two fp16 terms hi = fp16(v 2^e),
 * lo = fp16(v 2^e - hi) with max|tensor| 2^e in [2^14, 2^15), split on the way into LDS, three v_mfma_f32_32x32x16_f16 per product,
 * float32 accumulation, bias / residual / ReLU in float32.  About 22 bits per value relative to the value itself down to 2^-14 of the
 * tensor maximum, 2^-38 of the maximum absolute below.
 *   d_xmax, d_wmax : device floats, UPPER BOUNDS of max|x| and max|w_packed| (a bound below the true maximum overflows fp16 -> inf)
 *   d_ymax         : NULL, or a device float that receives max(*d_ymax, max|y|) by atomic maximum -- zero it (or leave an older maximum)
 *                    before the call; it is the next convolution's d_xmax
 * Other arguments, shapes and workspace (frcnn_conv_workspace_bytes) as frcnn_conv_nhwc. */
int frcnn_conv_nhwc_x3g(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual, float* d_y,
                        int N, int H, int W, int cin, int cout, int ksize, int stride, int pad, unsigned flags,
                        const float* d_xmax, const float* d_wmax, float* d_ymax, void* d_ws, size_t ws_bytes, void* stream);
/* ABI 15: the same call with the split reduction of a small convolution finished INSIDE the kernel, by the last block of every output tile
 * to arrive (round 6: what frcnn_resnet_forward / frcnn_resnet_backbone run; frcnn_conv_nhwc_x3g keeps the separate finishing pass).
 *   d_tile_counters : FRCNN_X3G_TILE_COUNTERS unsigned ints, ZERO before the first call; every call leaves them zero again.  They
 *                     belong to one stream at a time (two convolutions in flight on different streams need two arrays).
 * The partial planes are summed in ascending order like the separate pass: the results are the same bits (tests/test_conv_x3g_gpu.py).
 * Launches whose tile count exceeds the array, and launches that do not split, behave as frcnn_conv_nhwc_x3g. */
#define FRCNN_X3G_TILE_COUNTERS 16384
int frcnn_conv_nhwc_x3g_tickets(const float* d_x, const float* d_w_packed, const float* d_bias, const float* d_residual, float* d_y,
                                int N, int H, int W, int cin, int cout, int ksize, int stride, int pad, unsigned flags,
                                const float* d_xmax, const float* d_wmax, float* d_ymax, void* d_ws, size_t ws_bytes,
                                unsigned* d_tile_counters, void* stream);
/* ABI 15: the weight side of frcnn_conv_nhwc_x3g split ONCE, at pack time.  d_out (taps * cout * cin * 4 bytes, the size of the float32 pack
 * [taps][cout][cin]) receives every row in the kernel's own operand format -- per 32 input channels 128 bytes: [hi x 16 | lo x 16] fp16 of channels
 * 0..15, then of 16..31, hi = fp16(w 2^e), lo = fp16(w 2^e - hi) under the scale that *d_wmax gives -- so that a weight piece is a 16-byte copy
 * into LDS in every block of every launch instead of a split.  Pass it as d_w_packed with FRCNN_X3G_WSPLIT in `flags` (and the SAME d_wmax):
 * the results are the same bits as with the float32 pack.  cin % 32 == 0.  frcnn_bottleneck_weights.g3 == 2: the block's four packs are such images. */
int frcnn_pack_conv_x3g_weights(const float* d_w_packed, const float* d_wmax, void* d_out, int taps, int cout, int cin, void* stream);
/* ABI 12: frcnn_conv_nhwc_x3g CLAMPS an activation whose hi term would overflow fp16 under the tensor's scale (i.e. *d_xmax was not an
 * upper bound) instead of producing inf / NaN; every wave that clamped one adds 1 to a process-wide counter.  *out = that count since the
 * library was loaded (read it after synchronising the streams the convolutions ran on).  With the maxima the producers' epilogues leave
 * behind (frcnn_resnet_forward / frcnn_resnet_backbone chain them) the count stays 0: tests/test_stress_gpu.py asserts it on inputs built
 * against the per-tensor scale (models/resnet.py:38-46 has no such notion: its float32 convolutions cannot overflow). */
int frcnn_x3_saturation_events(unsigned long long* out);

/* d_out[0] = max(d_out[0], max_i |d_x[i]|), n floats, d_x 16-byte aligned (the scale source of a tensor no frcnn_conv_nhwc_x3g produced) */
int frcnn_tensor_absmax(const float* d_x, long long n, float* d_out, void* stream);

/* conv2d backward of the 3x3 "same" layers (vgg16.py:76-96, rpn.py:88):
 *   weight gradient  d_dwp [9][cout][cin] (the frcnn_pack_conv3x3 layout) from x [H][W][cin], dz [H][W][cout];
 *   data gradient    dx = frcnn_conv3x3_nhwc(dz, frcnn_pack_conv3x3_dgrad(wp), zero bias, flags 0). */
size_t frcnn_conv3x3_wgrad_workspace_bytes(int H, int W, int cin, int cout);
int frcnn_conv3x3_wgrad(const float* d_x, const float* d_dz, float* d_dwp, int H, int W, int cin, int cout,
                        void* d_ws, size_t ws_bytes, void* stream);
int frcnn_pack_conv3x3_dgrad(const float* d_wp, float* d_wd, int cout, int cin, void* stream);

/* General forms for the ResNet bottlenecks (models/resnet.py:38-46,110; BatchNorm frozen, so each conv+BN is one conv with a
 * folded weight): x [N][H][W][cin], y = conv(x) [N][Ho][Wo][cout], k x k taps, stride, pad.
 *   frcnn_conv_wgrad : d_dwp [k*k][cout][cin] = gradient with respect to the (folded) weight pack
 *   frcnn_conv_dgrad : d_dx [N][H][W][cin] = d_residual (or 0 if NULL) + gradient with respect to x, from d_dz [N][Ho][Wo][cout]
 *                      and d_wd = frcnn_pack_conv_dgrad(weight pack) ([tap][cin][cout]); cout % 16 == 0, cin % 4 == 0
 *   frcnn_scale_rows : dst[tap][co][ci] = src[tap][co][ci] * scale[co] (fold a frozen BN scale into a pack; chain rule back)
 *   frcnn_bn_scale_shift : scale = gamma / sqrt(var + eps), shift = beta - mean * scale (the eval-mode BatchNorm affine)
 *   frcnn_spatial_mean_backward : backward of `y.mean(-1).mean(-1)` (resnet.py:117). */
size_t frcnn_conv_wgrad_workspace_bytes(int N, int H, int W, int cin, int cout, int ksize, int stride, int pad);
int frcnn_conv_wgrad(const float* d_x, const float* d_dz, float* d_dwp, int N, int H, int W, int cin, int cout,
                     int ksize, int stride, int pad, void* d_ws, size_t ws_bytes, void* stream);
size_t frcnn_conv_dgrad_workspace_bytes(int N, int H, int W, int cin, int cout, int ksize, int stride, int pad);
int frcnn_conv_dgrad(const float* d_dz, const float* d_wd, const float* d_residual, float* d_dx, int N, int H, int W,
                     int cin, int cout, int ksize, int stride, int pad, void* d_ws, size_t ws_bytes, void* stream);
int frcnn_pack_conv_dgrad(const float* d_wp, float* d_wd, int taps, int cout, int cin, void* stream);
int frcnn_scale_rows(const float* d_src, const float* d_scale, float* d_dst, int taps, int cout, int cin, void* stream);
int frcnn_bn_scale_shift(const float* d_gamma, const float* d_beta, const float* d_mean, const float* d_var, float eps,
                         int c, float* d_scale, float* d_shift, void* stream);
int frcnn_spatial_mean_backward(const float* d_dy, float* d_dx, int N, int H, int W, int c, void* stream);

/* dy[i] = y[i] > 0 ? dy[i] : 0 (ReLU backward, in place); a[i] += b[i]. */
int frcnn_relu_backward(float* d_dy, const float* d_y, size_t n, void* stream);
int frcnn_add_inplace(float* d_a, const float* d_b, size_t n, void* stream);
/* MaxPool2d(2,2) backward: x [H][W][c] pool input, dy [H/2][W/2][c] -> dx [H][W][c] (first maximum wins). */
int frcnn_maxpool2x2_backward(const float* d_x, const float* d_dy, float* d_dx, int H, int W, int c, void* stream);
/* torchvision RoIPool backward (detector.py:72): dout [n][pooled][pooled][c] -> dfm [fh][fw][c] (gradient to
 * each bin's argmax cell).  Deterministic gather formulation; accumulate != 0 adds to dfm. */
size_t frcnn_roi_pool_backward_workspace_bytes(int n_rois, int pooled, int c);
int frcnn_roi_pool_backward(const float* d_fm, int fh, int fw, int c, const float* d_rois, int n_rois, int pooled,
                            float spatial_scale, const float* d_dout, float* d_dfm, int accumulate,
                            void* d_ws, size_t ws_bytes, void* stream);
/* y[c][r] = x[r][c]; y rows are ldo >= rows wide, the tail is zero filled. */
int frcnn_transpose(const float* d_x, int ldi, float* d_y, int ldo, int rows, int cols, void* stream);
/* torch.optim.SGD.step as built at __main__.py:98-105: g += weight_decay*w; buf = first_step ? g :
 * momentum*buf + g; w -= lr*buf (layout agnostic: applied to the packed weights). */
int frcnn_sgd_step(float* d_w, const float* d_grad, float* d_momentum_buf, size_t n, float lr, float momentum,
                   float weight_decay, int first_step, void* stream);
/* The same update (__main__.py:98-105 SGD) of a convolution master [taps][cout][cin] whose frozen BatchNorm (models/resnet.py:58-77) is folded into the convolution (ResNet bottlenecks),
 * with the folded pack rebuilt in the same launch: d_folded = w_new * d_scale[co] (frcnn_scale_rows' product; n = taps*cout*cin).  ABI 10. */
int frcnn_sgd_step_fold(float* d_w, const float* d_grad, float* d_momentum_buf, size_t n, float lr, float momentum,
                        float weight_decay, int first_step, const float* d_scale, float* d_folded, int cout, int cin, void* stream);

/* Introspection for parity tests: device pointers of intermediate tensors of the LAST forward
 * on this ctx.  which: 0 feature map NHWC [fh][fw][512], 1 RPN head [fh*fw][128],
 * 2 objectness scores [A], 3 sorted anchor indices int32 [pre_nms], 4 RoI-pool out
 * [post_nms][7][7][512], 5 fc2 output [post_nms][4096], 6 anchor map, 7 valid map,
 * 8 head logits [post_nms][128].  (ResNet: the feature map / RoI-pool tensors have 1024 channels,
 * id 5 is the pooled [post_nms][2048] vector.)  Returns FRCNN_EINVAL for unknown ids. */
int frcnn_ctx_tensor(frcnn_ctx* ctx, int which, void** d_ptr, size_t* bytes);

/* Per-kernel-class HIP-event timing for bench.py's roofline block: when enabled, every launch
 * of class `k` inside frcnn_vgg16_forward is bracketed by events on the launch stream.
 * classes: 0 conv3x3 MFMA (backbone+RPN), 1 conv first layer, 2 linear MFMA, 3 proposals,
 * 4 roi_pool, 5 other, 6 Winograd input / output transforms (three-launch float32 form), 7 float32 Winograd MFMA kernels
 * (wino_fused_kernel; the three-launch form's batched GEMM), 8 x6 Winograd input / output transforms, 9 x6 Winograd batched GEMM
 * (gemm_x6t_kernel / gemm_x3t_kernel), 10 one-launch f32x3 Winograd layers (wino_x3d_kernel + their channel-maximum pass).
 * frcnn_ctx_timing_read synchronises the recorded events and returns
 * accumulated milliseconds and launch counts since the last reset. */
#define FRCNN_NUM_KCLASS 11
int frcnn_ctx_timing_enable(frcnn_ctx* ctx, int enable);
int frcnn_ctx_timing_read(frcnn_ctx* ctx, double ms[FRCNN_NUM_KCLASS], int64_t launches[FRCNN_NUM_KCLASS],
                          int reset);

#ifdef __cplusplus
}
#endif
#endif /* FRCNN_HIP_H */
