// common.h -- shared helpers for libfrcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include "../../include/frcnn_hip.h"

typedef float f32x4  __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Experiment knobs.  The tuning / A-B variables the tools/ scripts set (FRCNN_FC_TILES, FRCNN_HX_CFG, FRCNN_WF_XCL, ...) exist ONLY in a
// library built with -DFRCNN_EXPERIMENT_KNOBS (`make KNOBS=1`, tools/build_ablate.sh).  The shipped library never reads the environment:
// a stray variable must not change split-K factors -- i.e. output bits -- behind the hipGraph keys and the forward params.
inline const char* frcnn_knob(const char* name)
{
#ifdef FRCNN_EXPERIMENT_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

namespace frcnn {

// Record the last HIP error of this thread (read by frcnn_last_hip_error()).
void set_hip_error(hipError_t e);

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_hip_error(e); return FRCNN_EHIP; }
    return FRCNN_OK;
}

#define FRCNN_HIP_TRY(expr)                                   \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) { ::frcnn::set_hip_error(_e); return FRCNN_EHIP; } \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE attribute: one flag per (kernel instantiation, device), so that a
// process which moves a model to a second GPU (`model.cuda(1)`) raises the limit there too (a process-wide flag left the
// > 64 KB launches of the second device at the default limit).  `once` is a function-local static of the launcher.
struct DeviceOnce { unsigned char done[64]; };
inline int set_max_dynamic_lds(DeviceOnce& once, const void* kern, size_t bytes)
{
    int dev = 0;
    FRCNN_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return FRCNN_EINVAL;
    if (!once.done[dev]) {
        FRCNN_HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        once.done[dev] = 1;
    }
    return FRCNN_OK;
}
#define FRCNN_MAX_LDS_ONCE(kern, bytes)                                                                  \
    do {                                                                                                 \
        static ::frcnn::DeviceOnce _once = {};                                                           \
        const int _rc = ::frcnn::set_max_dynamic_lds(_once, reinterpret_cast<const void*>(kern), (bytes)); \
        if (_rc) return _rc;                                                                             \
    } while (0)

// 3x3 layers that run as Winograd F(2x2,3x3) in math mode FRCNN_MATH_F32_WINOGRAD (VGG-16: conv3_1 ... conv5_3 and the RPN trunk)
static inline bool conv3x3_uses_winograd(int cin, int cout) { return cin >= 128 && cout >= 256 && cin % 16 == 0 && cout % 128 == 0; }
// ResNet bottlenecks (3x3 width -> width): the stride-1 blocks of layer3 (width 256, one 38 x 63 map) and layer4 (width 512,
// 300 RoIs x 4 x 4 maps).  Measured: layer4 alone ResNet-50 311 -> 360 / ResNet-101 228 -> 258 img/s, with layer3 364 / 267;
// the 128-wide blocks of layer2 are below the width where the transforms pay.  (The experiment knob that moved this bound is gone: the
// Python packer decides with the same constant -- fasterrcnn_amd/_native.py resnet_block_uses_winograd -- and a library that read the
// environment would consume packs of the other layout: ADVICE r4.)
static inline bool resnet_block_uses_winograd(int width, int stride) { return stride == 1 && width >= 256 && width % 128 == 0; }

// Per-kernel-class event timer (see frcnn_ctx_timing_* in the header).
struct KernelTimer;

}  // namespace frcnn

// ---- launchers implemented across the .hip files (internal C++ API; the C ABI in api.hip
//      validates arguments and forwards here) ------------------------------------------------
namespace frcnn {

int launch_anchors(int image_h, int image_w, int fh, int fw, int feature_pixels,
                   float* anchor_map, float* valid_map, hipStream_t s);

int launch_pack_conv3x3(const float* w, float* wp, int cout, int cin, hipStream_t s);
int launch_pack_conv3x3_c3(const float* w, float* wp, int cout, hipStream_t s);
int launch_pack_fc_chw_to_hwc(const float* w, float* wp, int out_f, int c, int phw, hipStream_t s);
int launch_pack_stack_rows(const float* w1, const float* b1, int n1, const float* w2, const float* b2,
                           int n2, int k, int n_pad, float* wo, float* bo, hipStream_t s);

int launch_conv3x3_c3(const float* x, const float* wp, const float* b, float* y, int H, int W,
                      int cout, unsigned flags, hipStream_t s, float* cmax_out = nullptr);
size_t conv3x3_workspace_bytes(int H, int W, int cin, int cout);
int conv3x3_blocks_target();
void conv3x3_set_blocks_target(int target);     // split-K work-unit target of the calling thread's next launches (0 = default)
int launch_conv3x3_nhwc(const float* x, const float* wp, const float* b, float* y, int H, int W,
                        int cin, int cout, unsigned flags, void* ws, size_t ws_bytes, hipStream_t s);
int launch_maxpool2x2(const float* x, float* y, int H, int W, int c, hipStream_t s);

// conv_gather.hip (ResNet path)
size_t conv_gather_workspace_bytes(int N, int H, int W, int cin, int cout, int R, int stride, int pad);
// math FRCNN_CONV_F32X3G: both operands as two fp16 terms under ONE power-of-two scale per tensor (conv_gather_x3_kernel); x3 names the
// device floats holding an upper bound of max|x|, max|wp| and (or null) the float that receives max|y| by atomic maximum (zero it first)
#define FRCNN_CONV_F32X3G 2
// tile_counters (optional, f32x3 mode): >= GX_TILE_COUNTERS zeroed unsigned ints owned by the caller's stream -- a split reduction is then
// finished by the LAST block of every output tile to arrive (conv_gather_x3_kernel) instead of by gather_splitk_finish_kernel; the last
// block leaves its counter at zero again
static constexpr int GX_TILE_COUNTERS = 16384;
// wsplit: wp is NOT the float32 pack but its pre-split image (launch_pack_x3g_weights: same size, the kernel's LDS row format)
// trusted (with wsplit): *xmax is the tensor's TRUE maximum (a producer's epilogue / tensor_absmax_kernel left it): no clamp, no saturation count
struct GatherX3 { const float* xmax; const float* wmax; float* ymax; unsigned* tile_counters = nullptr; bool wsplit = false; bool trusted = false; };
int launch_pack_x3g_weights(const float* wp, const float* wmax, void* out, long long rows, int cin, hipStream_t s);
int launch_conv_gather(const float* x, const float* wp, const float* bias, const float* residual, float* y,
                       int N, int H, int W, int cin, int cout, int R, int stride, int pad, unsigned flags,
                       void* ws, size_t ws_bytes, hipStream_t s, int math = FRCNN_GRAD_F32, const GatherX3* x3 = nullptr);
// host-mapped process-wide counter of the waves whose activation split saturated (conv_gather_x3_kernel); null if it could not be allocated
unsigned* x3_saturation_counter();
// out[0] = max(out[0], max |x[i]|) (out zeroed by the caller or holding an earlier maximum)
int launch_tensor_absmax(const float* x, long long n, float* out, hipStream_t s);
int launch_conv7x7_s2_c3(const float* x, const float* wp, const float* b, float* y, int H, int W, int cout,
                         unsigned flags, hipStream_t s);
int launch_maxpool3x3_s2(const float* x, float* y, int H, int W, int c, hipStream_t s);
int launch_spatial_mean(const float* x, float* y, int N, int H, int W, int c, hipStream_t s);
int launch_fold_bn_pack(const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                        float eps, int cout, int cin, int ksize, float* wp, float* bp, hipStream_t s);

size_t linear_workspace_bytes(int M, int N, int K);
int launch_linear(const float* a, int lda, const float* w, const float* bias, float* y, int ldy,
                  int M, int N, int K, unsigned flags, void* ws, size_t ws_bytes, hipStream_t s);
int launch_linear_batched(const float* a, int lda, size_t a_stride, const float* w, size_t w_stride, float* y, int ldy,
                          size_t y_stride, int M, int N, int K, int batches, hipStream_t s);
void linear_batched_set_tile(int rows);        // 64 | 128: tile rows of the calling thread's next batched launches (0 = default)
// winograd.hip: F(2x2,3x3) float32 path of the wide 3x3 layers
size_t conv3x3_winograd_workspace_bytes(int N, int H, int W, int cin, int cout);
int launch_pack_conv3x3_winograd(const float* w, const float* scale, float* u, int cout, int cin, hipStream_t s);
int launch_pack_conv3x3_winograd_taps(const float* wp, float* u, int cout, int cin, int data_gradient, hipStream_t s);
int launch_conv3x3_winograd(const float* x, const float* u, const float* b, float* y, int N, int H, int W, int cin, int cout,
                            unsigned flags, void* ws, size_t ws_bytes, hipStream_t s);
int winograd_plan(int N, int H, int W, int cin, int cout, unsigned flags, void* ws, size_t ws_bytes, float** V, float** M);
int launch_winograd_input(const float* x, float* V, int N, int H, int W, int cin, hipStream_t s);
int launch_winograd_gemm(const float* V, const float* u, float* M, int N, int H, int W, int cin, int cout, hipStream_t s);
// cmax_out (optional, cout % 256 == 0 and ReLU): the per-pixel channel maximum of the OUTPUT, accumulated with atomic maxima into a buffer the
// caller zeroed -- the next f32x3 layer's scale source, so that it need not read the tensor again (launch_pixel_absmax)
int launch_winograd_output(const float* M, const float* b, float* y, int N, int H, int W, int cout, unsigned flags, hipStream_t s,
                           float* cmax_out = nullptr);
bool winograd_output_emits_cmax(int cout, unsigned flags);
// winofused.hip: the same layer for one map as ONE launch (all 16 positions in accumulators, no V / M scratch)
static inline bool conv3x3_uses_winograd_fused(int cin, int cout) { return cin >= 64 && cin % 16 == 0 && cout >= 64 && cout % 64 == 0; }
// ResNet bottleneck 3x3 (width -> width) on ONE map (the feature extractor's layer1..3 at inference): every stride-1 block;
// the per-RoI 4 x 4 maps of layer4 (n_maps = RoIs) stay on the three-launch batched form (resnet_block_uses_winograd)
static inline bool resnet_block_uses_winograd_fused(int n_maps, int width, int stride)
{
    return n_maps == 1 && stride == 1 && conv3x3_uses_winograd_fused(width, width);
}
bool conv3x3_winograd_fused_ok(int H, int W, int cin, int cout);
int launch_pack_conv3x3_winograd_fused(const float* w, const float* scale, float* u, int cout, int cin, hipStream_t s);
int launch_pack_conv3x3_winograd_fused_taps(const float* wp, float* u, int cout, int cin, int data_gradient, hipStream_t s);
int launch_conv3x3_winograd_fused(const float* x, const float* u, const float* b, float* y, int H, int W, int cin, int cout,
                                  unsigned flags, hipStream_t s, int n_maps = 1);
// gemm_x6t.hip: batched f32x6 GEMM on tile records, LDS-DMA staged
int gemm_x6t_row_tile(int M);
int gemm_x6t_col_tile(int N);
size_t x6t_record_bytes(int rows_padded, int K);
int launch_split_rows_x6t(const float* a, int lda, size_t a_batch_floats, void* rec, int R, int rows_padded, int K, int batches, hipStream_t s);
bool gemm_x6t_shape_ok(int M, int N, int K, int batches);
size_t gemm_x6t_workspace_bytes(int M, int N, int K, int batches);
void gemm_x6t_set_tiles(int mode);          // tile choice of the calling thread's next launches: 0 cost model, 1 = 320 x 256, 2 = 160 x 128
int launch_gemm_x6t(const void* a_rec, int a_rows, size_t a_batch_bytes, const void* b_rec, int b_rows, size_t b_batch_bytes,
                    const float* bias, const float* residual, float* c, int ldc, size_t c_batch_floats, int M, int N, int K, int batches,
                    unsigned flags, void* ws, size_t ws_bytes, hipStream_t s, int tiles_mode = -1);
int launch_split_pixels_x6t(const float* x, void* rec, int N, int H, int W, int C, int stride, int rows_padded, hipStream_t s);
int gemm_x6t_get_tiles();
int launch_gemm_x6t_reduce(const float* ws, const float* bias, const float* residual, float* c, int ldc, size_t c_batch_floats, int M, int N,
                           int batches, int splits, int relu, hipStream_t s);
// gemm_x3t.hip: the same GEMM in the f32x3 arithmetic (two fp16 terms per row-scaled operand, three fp16 MFMAs per product)
size_t x3t_record_bytes(int rows_padded, int K);
int launch_rows_scale_x3t(const float* a, int lda, size_t a_batch_floats, float* inv_scale, int R, int rows_padded, int K, int batches,
                          hipStream_t s);
int launch_split_rows_x3t(const float* a, int lda, size_t a_batch_floats, const float* inv_scale, void* rec, int R, int rows_padded, int K,
                          int batches, hipStream_t s);
int launch_split_pixels_x3t(const float* x, const float* cmax, void* rec, float* inv, int N, int H, int W, int C, int stride, int rows_padded,
                            hipStream_t s);
int launch_split_patches3x3_x3t(const float* x, const float* cmax, void* rec, float* inv, int N, int H, int W, int C, int stride,
                                int rows_padded, hipStream_t s);
size_t gemm_x3t_workspace_bytes(int M, int N, int K, int batches);
int launch_gemm_x3t(const void* a_rec, const float* a_inv, int a_rows, size_t a_batch_bytes, size_t a_inv_batch, const void* b_rec,
                    const float* b_inv, int b_rows, size_t b_batch_bytes, size_t b_inv_batch, const float* bias, const float* residual,
                    float* c, int ldc, size_t c_batch_floats, int M, int N, int K, int batches, unsigned flags, void* ws, size_t ws_bytes,
                    hipStream_t s, int tiles_mode = -1);
// wino_x3.hip: the Winograd layer on gemm_x3t; packed x3t operands (records + row scales in one blob)
size_t x3t_blob_bytes(int rows_padded, int K, int batches);
int launch_pack_rows_x3t(const float* a, int lda, size_t a_batch_floats, void* blob, int R, int rows_padded, int K, int batches, hipStream_t s);
int launch_pixel_absmax(const float* x, float* cmax, long long pixels, int C, hipStream_t s);
size_t conv3x3_winograd_x3_pack_bytes(int cout, int cin);
int launch_pack_conv3x3_winograd_x3(const float* u_f32, void* ublob, int cout, int cin, hipStream_t s);
size_t conv3x3_winograd_x3_workspace_bytes(int N, int H, int W, int cin, int cout);
int winograd_x3_plan(int N, int H, int W, int cin, int cout, unsigned flags, void* ws, size_t ws_bytes, void** V, float** vinv, float** cmax,
                     float** M, void** G, size_t* g_bytes);
int launch_winograd_x3_input(const float* x, float* cmax, void* vrec, float* vinv, int N, int H, int W, int cin, hipStream_t s,
                             const float* cmax_ready = nullptr);   // cmax_ready: the channel maxima of x are already there (skips the pass)
int launch_winograd_x3_gemm(const void* vrec, const float* vinv, const void* ublob, float* M, int N, int H, int W, int cin, int cout, void* gws,
                            size_t gws_bytes, hipStream_t s);
int launch_conv3x3_winograd_x3(const float* x, const void* ublob, const float* b, float* y, int N, int H, int W, int cin, int cout,
                               unsigned flags, void* ws, size_t ws_bytes, hipStream_t s,
                               const float* cmax_ready = nullptr, float* cmax_out = nullptr);
// wino_x3f.hip: the one-launch form of the x3 Winograd layer (frcnn_forward_params.winograd_x3f_mask: conv2_2 .. conv3_3 of VGG-16)
size_t conv3x3_winograd_x3_fused_workspace_bytes(int N, int H, int W);
int launch_conv3x3_winograd_x3_fused(const float* x, const void* ublob, const float* b, float* y, int N, int H, int W, int cin, int cout,
                                     unsigned flags, void* ws, size_t ws_bytes, hipStream_t s, const float* cmax_ready = nullptr,
                                     float* cmax_out = nullptr,    // cmax_ready / cmax_out: as above (cmax_out needs ReLU; zeroed by the caller)
                                     float* pair_spill = nullptr, size_t pair_spill_bytes = 0);   // FRCNN_X3F_PAIR: the spill scratch if it is not behind the maxima in ws
size_t conv3x3_winograd_x3_pair_spill_bytes(int N, int H, int W, int cout);       // wino_x3p.hip: the spill scratch alone (0: shape not supported)
size_t conv3x3_winograd_x3_pair_workspace_bytes(int N, int H, int W, int cout);   // FRCNN_X3F_PAIR: channel maxima + spill scratch (0: shape not supported)
int launch_roi_pool_x3t(const float* fm, int fh, int fw, int c, const float* rois, const int32_t* n_rois, int max_rois, int pooled,
                        float scale, float* cmax, float* inv, void* rec, int rec_rows, hipStream_t s, bool cmax_ready = false);
// wino_x6.hip: Winograd F(2x2,3x3) layers whose position GEMMs run on gemm_x6t
bool conv3x3_uses_winograd_x6(int cin, int cout);
size_t conv3x3_winograd_x6_pack_bytes(int cout, int cin);
size_t conv3x3_winograd_x6_workspace_bytes(int N, int H, int W, int cin, int cout);
int launch_pack_conv3x3_winograd_x6(const float* w, const float* scale, void* urec, int cout, int cin, hipStream_t s);
int launch_winograd_x6_input(const float* x, void* vrec, int N, int H, int W, int cin, hipStream_t s);
int launch_winograd_x6_gemm(const void* vrec, const void* urec, float* M, int N, int H, int W, int cin, int cout, void* gws, size_t gws_bytes, hipStream_t s);
int winograd_x6_plan(int N, int H, int W, int cin, int cout, unsigned flags, void* ws, size_t ws_bytes, void** V, float** M, void** G, size_t* g_bytes);
int launch_conv3x3_winograd_x6(const float* x, const void* urec, const float* b, float* y, int N, int H, int W, int cin, int cout,
                               unsigned flags, void* ws, size_t ws_bytes, hipStream_t s);
int launch_split_patches3x3_x6t(const float* x, void* rec, int N, int H, int W, int C, int stride, int rows_padded, hipStream_t s);
int launch_softmax_rows(const float* x, int ldx, float* y, int M, int ncls, hipStream_t s);
int launch_head_finish(const float* x, int ldx, int M, int ncls, int ndelta, float* classes,
                       float* deltas, hipStream_t s);

// proposal scratch layout is owned by the ctx; see proposals.hip
struct ProposalScratch {
    unsigned long long* keys;   // [A_cap]
    float*    boxes_all;        // [A_cap][4]
    float*    cand_boxes;       // [pre_cap][4]  clipped+filtered, score-descending
    float*    cand_scores;      // [pre_cap]
    unsigned long long* mask;   // [pre_cap][pre_cap/64]
    int32_t*  keep;             // [post_cap]
    int       a_cap, pre_cap, post_cap;
};
size_t proposal_scratch_bytes(int a_cap, int pre_cap, int post_cap);
void   proposal_scratch_carve(ProposalScratch& ps, void* base, int a_cap, int pre_cap, int post_cap);

int launch_rpn_proposals(const ProposalScratch& ps, const float* head, int ld_head,
                         const float* anchor_map, const float* valid_map, int fh, int fw,
                         int image_h, int image_w, int pre_nms, int post_nms, float nms_thr,
                         float min_side, float* scores, int32_t* sorted_idx, float* props,
                         int32_t* counts, hipStream_t s);
int launch_nms(const ProposalScratch& ps, const float* boxes, const float* scores, int n, float thr,
               int max_keep, int32_t* keep, int32_t* n_keep, hipStream_t s);

int launch_roi_pool(const float* fm, int fh, int fw, int c, const float* rois, const int32_t* n_rois,
                    int max_rois, int pooled, float scale, float* out, hipStream_t s);
int launch_roi_pool_x6t(const float* fm, int fh, int fw, int c, const float* rois, const int32_t* n_rois,
                        int max_rois, int pooled, float scale, void* rec, int rec_rows, hipStream_t s);

// roialign.hip
int launch_roi_align(const float* fm, int fh, int fw, int c, const float* rois, const int32_t* n_rois, int max_rois, int pooled,
                     float scale, int sampling_ratio, int aligned, float* out, hipStream_t s);
int launch_roi_align_backward(const float* rois, int n_rois, int fh, int fw, int c, int pooled, float scale, int sampling_ratio,
                              int aligned, const float* dout, float* dfm, int accumulate, hipStream_t s);

int launch_rpn_targets(const float* anchor_map, const float* valid_map, int A, const float* gt, int M,
                       double obj_thr, double bg_thr, float* rpn_map, int32_t* obj_idx, int32_t* bg_idx,
                       int32_t* counts, void* ws, hipStream_t s);

size_t preprocess_workspace_bytes(int H, int W, int Ho, int Wo);
int launch_preprocess(const unsigned char* rgb, int H, int W, int Ho, int Wo, int bgr, int flip, float scaling,
                      const float* means, const float* stds, float* out, unsigned char* out_u8, void* ws,
                      size_t ws_bytes, hipStream_t s);

int launch_detections(const float* props, const float* classes, const float* deltas,
                      const int32_t* n_rois, int max_rois, int ncls, int image_h, int image_w,
                      float score_thr, float nms_thr, double* out, int32_t* out_cnt, hipStream_t s);

// gemm_tn.hip / train.hip (train step)
size_t gemm_tn_workspace_bytes(int M, int N, int R, int taps);
int launch_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int R,
                   void* ws, size_t ws_bytes, hipStream_t s, int math = FRCNN_GRAD_F32);
int launch_conv3x3_wgrad(const float* x, const float* dz, float* dwp, int H, int W, int cin, int cout,
                         void* ws, size_t ws_bytes, hipStream_t s, int math = FRCNN_GRAD_F32);
int launch_label_proposals(const float* props, const int32_t* n_props, int max_props, const float* gt,
                           const int32_t* gt_cls, int M, int ncls, float bg_thr, float obj_thr,
                           const float means[4], const float stds[4], float* out_props, int32_t* out_cls,
                           float* out_onehot, float* out_deltas, int32_t* out_count, hipStream_t s);
int launch_gather_rows(const float* src, const int32_t* idx, int n, int row_floats, float* dst, hipStream_t s);
int launch_rpn_loss(const float* head, int ld, int cells, const int32_t* sample, int n_sample, const float* rpn_map,
                    float* losses, float* d_head, hipStream_t s);
int launch_detector_loss(const float* classes, const float* deltas, const float* gt_onehot, const float* gt_deltas,
                         int S, int ncls, float* losses, float* d_logits, int ld, hipStream_t s);
int launch_relu_backward(float* dy, const float* y, size_t n, hipStream_t s);
int launch_add_inplace(float* a, const float* b, size_t n, hipStream_t s);
int launch_maxpool2x2_backward(const float* x, const float* dy, float* dx, int H, int W, int C, hipStream_t s);
size_t roi_pool_backward_workspace_bytes(int n_rois, int pooled, int C);
int launch_roi_pool_backward(const float* fm, int fh, int fw, int C, const float* rois, int n_rois, int pooled,
                             float scale, const float* dout, float* dfm, int accumulate, void* ws, size_t ws_bytes,
                             hipStream_t s);
int launch_transpose(const float* x, int ldi, float* y, int ldo, int rows, int cols, hipStream_t s);
int launch_pack_conv3x3_dgrad(const float* wp, float* wd, int cout, int cin, hipStream_t s);
int launch_conv_wgrad(const float* x, const float* dz, float* dwp, int N, int H, int W, int cin, int cout, int ks, int stride,
                      int pad, void* ws, size_t ws_bytes, hipStream_t s, int math = FRCNN_GRAD_F32);
size_t conv_dgrad_workspace_bytes(int N, int H, int W, int cin, int cout, int R, int stride, int pad);
int launch_conv_dgrad(const float* dz, const float* wd, const float* residual, float* dx, int N, int H, int W, int cin,
                      int cout, int R, int stride, int pad, void* ws, size_t ws_bytes, hipStream_t s, int math = FRCNN_GRAD_F32);
int launch_pack_conv_dgrad(const float* wp, float* wd, int taps, int cout, int cin, hipStream_t s);
int launch_scale_rows(const float* src, const float* scale, float* dst, int taps, int cout, int cin, hipStream_t s);
int launch_bn_scale_shift(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int c,
                          float* scale, float* shift, hipStream_t s);
int launch_spatial_mean_backward(const float* dy, float* dx, int N, int H, int W, int c, hipStream_t s);
int launch_sgd(float* w, const float* g, float* buf, size_t n, float lr, float momentum, float weight_decay, int first,
               hipStream_t s);
int launch_sgd_fold(float* w, const float* g, float* buf, size_t n, float lr, float momentum, float weight_decay, int first,
                    const float* scale, float* folded, int cout, int cin, hipStream_t s);

}  // namespace frcnn
