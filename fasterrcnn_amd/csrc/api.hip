// api.hip -- the extern "C" surface declared in include/frcnn_hip.h: argument validation,
// the per-image context (activation ping-pong buffers + scratch, one hipMalloc slab), and the
// fused VGG-16 forward that enqueues every kernel of FasterRCNNModel.forward
// (reference: models/faster_rcnn.py:80-132) on one stream with no host round trip.
#include "common.h"
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include <new>

namespace frcnn {

static thread_local std::string g_hip_error;
void set_hip_error(hipError_t e) { g_hip_error = hipGetErrorString(e); }

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct EventRec { int cls; hipEvent_t start, stop; };

}  // namespace frcnn

using namespace frcnn;

struct frcnn_ctx {
    int max_h = 0, max_w = 0, max_rois = 0, max_images = 0;     // max_images > 0: a backbone-only ctx (frcnn_ctx_create_backbone)
    int max_fh = 0, max_fw = 0, a_cap = 0, pre_cap = 0;
    void* slab = nullptr;
    size_t slab_bytes = 0;
    // carved buffers
    float *act_a = nullptr, *act_b = nullptr;      // ping-pong activations (NHWC)
    float *fm = nullptr;                           // [fh][fw][512]
    float *rpn_trunk = nullptr;                    // [fh][fw][512]
    float *rpn_head = nullptr;                     // [fh*fw][128]
    float *scores = nullptr;                       // [A]
    int32_t* sorted_idx = nullptr;                 // [pre_cap]
    float *anchor_map = nullptr, *valid_map = nullptr;
    float *roi_out = nullptr;                      // [max_rois][7][7][512]
    float *fc1_out = nullptr, *fc2_out = nullptr;  // [max_rois][4096]
    void *roi_rec = nullptr, *fc1_rec = nullptr;   // x6t / x3t records of roi_out ([rec_rows][25088]) and fc1_out (FRCNN_FC_F32X6T / _F32X3T):
    int rec_rows = 0;                              // one allocation, made by the first forward that runs fc1 / fc2 in a record mode
    float *roi_inv = nullptr, *fc1_inv = nullptr, *fm_cmax = nullptr;   // FRCNN_FC_F32X3T: row scales of the two record arrays, channel maximum of the feature map
    float *head_logits = nullptr;                  // [max_rois][128]
    void* lin_ws = nullptr; size_t lin_ws_bytes = 0;
    void* conv_ws = nullptr; size_t conv_ws_bytes = 0;   // split-K partials of under-filled conv layers
    void* wino_ws = nullptr; size_t wino_ws_bytes = 0;   // V and M of the Winograd layers; allocated by the first forward that needs it
    void* wx_ws = nullptr; size_t wx_ws_bytes = 0;       // V records, M and split-K partials of the x6 Winograd layers; allocated on first use
    int max_head_rois = 0;      // frcnn_ctx_create_head: pooled RoIs a call of frcnn_resnet_head may bring
    float* x3p_spill = nullptr; size_t x3p_spill_bytes = 0;  // scratch of the two-pass one-launch f32x3 layers (csrc/wino_x3p.hip); first use
    float* x3f_cmax = nullptr; size_t x3f_cmax_bytes = 0;   // channel maxima of a one-launch f32x3 Winograd layer's input (csrc/wino_x3f.hip); first use
    void* rx_rec = nullptr; size_t rx_rec_bytes = 0;     // activation records of the x6 1x1 convolutions (ResNet bottlenecks); on first use
    void* rx_ws = nullptr; size_t rx_ws_bytes = 0;       // their split-K partials
    float* rx_aux = nullptr; size_t rx_aux_floats = 0;   // f32x3 form: row scales of the record array + channel maxima of the layer input
    // g3 bottlenecks (frcnn_bottleneck_weights.g3): the tensor maxima handed from one convolution's epilogue to the next one's scale.  One
    // zeroed float per convolution output of a forward (gx_next counts them; one memset per stage), gx_x = the slot bounding the CURRENT
    // block input (null: unknown, the next g3 block measures it)
    float* gx_max = nullptr; int gx_next = 0; const float* gx_x = nullptr;
    unsigned* gx_cnt = nullptr;                  // GX_TILE_COUNTERS tile tickets of the in-kernel split reductions (conv_gather_x3_kernel), zero between launches
    float* res_buf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // ResNet bottleneck rotation
    size_t res_buf_floats = 0;
    int last_c = 512, last_vec = 4096, last_head_ld = 128;
    ProposalScratch ps{};
    // anchor cache key
    int anc_h = -1, anc_w = -1, anc_fh = -1, anc_fw = -1;
    // last forward's shapes (for frcnn_ctx_tensor)
    int last_fh = 0, last_fw = 0, last_pre = 0, last_post = 0;
    // timing
    bool timing = false;
    std::vector<EventRec> recs;
    std::vector<hipEvent_t> free_events;
    double t_ms[FRCNN_NUM_KCLASS] = {0};
    int64_t t_cnt[FRCNN_NUM_KCLASS] = {0};
};

namespace {

struct Scope {
    frcnn_ctx* c; int cls; hipStream_t s; hipEvent_t e0 = nullptr, e1 = nullptr; bool on;
    Scope(frcnn_ctx* c_, int cls_, hipStream_t s_) : c(c_), cls(cls_), s(s_), on(c_ && c_->timing) {
        if (!on) return;
        e0 = take(); e1 = take();
        if (!e0 || !e1) { on = false; return; }
        (void)hipEventRecord(e0, s);
    }
    ~Scope() {
        if (!on) return;
        (void)hipEventRecord(e1, s);
        c->recs.push_back(EventRec{cls, e0, e1});
    }
    hipEvent_t take() {
        if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" {

int frcnn_abi_version(void) { return FRCNN_ABI_VERSION; }

const char* frcnn_error_string(int code)
{
    switch (code) {
        case FRCNN_OK: return "ok";
        case FRCNN_EINVAL: return "invalid argument";
        case FRCNN_EHIP: return "HIP runtime error";
        case FRCNN_ENOMEM: return "out of device memory";
        case FRCNN_EUNSUPPORTED: return "unsupported configuration";
        case FRCNN_ENODEVICE: return "no gfx950 device";
        default: return "unknown error";
    }
}

const char* frcnn_last_hip_error(void) { return g_hip_error.c_str(); }

int frcnn_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int good = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++good;
    }
    return good;
}

int frcnn_anchors(int image_h, int image_w, int fh, int fw, int feature_pixels,
                  float* d_anchor_map, float* d_valid_map, void* stream)
{
    if (!d_anchor_map || !d_valid_map || image_h < 1 || image_w < 1) return FRCNN_EINVAL;
    return launch_anchors(image_h, image_w, fh, fw, feature_pixels, d_anchor_map, d_valid_map, as_stream(stream));
}

size_t frcnn_preprocess_workspace_bytes(int H, int W, int out_h, int out_w) { return preprocess_workspace_bytes(H, W, out_h, out_w); }

int frcnn_preprocess(const unsigned char* d_rgb, int H, int W, int out_h, int out_w, int bgr_order, int horizontal_flip,
                     float scaling, const float* means, const float* stds, float* d_out, unsigned char* d_out_u8,
                     void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_rgb || !d_out || !d_ws || !means || !stds) return FRCNN_EINVAL;
    return launch_preprocess(d_rgb, H, W, out_h, out_w, bgr_order, horizontal_flip, scaling, means, stds, d_out, d_out_u8,
                             d_ws, ws_bytes, as_stream(stream));
}

int frcnn_rpn_targets(const float* d_anchor_map, const float* d_valid_map, int n_anchors, const float* d_gt_boxes,
                      int n_gt, double object_thr, double background_thr, float* d_rpn_map, int32_t* d_object_idx,
                      int32_t* d_background_idx, int32_t* d_counts, void* d_ws, void* stream)
{
    if (!d_anchor_map || !d_valid_map || !d_gt_boxes || !d_rpn_map || !d_object_idx || !d_background_idx ||
        !d_counts || !d_ws)
        return FRCNN_EINVAL;
    return launch_rpn_targets(d_anchor_map, d_valid_map, n_anchors, d_gt_boxes, n_gt, object_thr, background_thr,
                              d_rpn_map, d_object_idx, d_background_idx, d_counts, d_ws, as_stream(stream));
}

int frcnn_pack_conv3x3(const float* d_w, float* d_wp, int cout, int cin, void* stream)
{
    if (!d_w || !d_wp) return FRCNN_EINVAL;
    return launch_pack_conv3x3(d_w, d_wp, cout, cin, as_stream(stream));
}

int frcnn_pack_conv3x3_c3(const float* d_w, float* d_wp, int cout, void* stream)
{
    if (!d_w || !d_wp) return FRCNN_EINVAL;
    return launch_pack_conv3x3_c3(d_w, d_wp, cout, as_stream(stream));
}

int frcnn_pack_fc_chw_to_hwc(const float* d_w, float* d_wp, int out_features, int channels, int pooled_hw,
                             void* stream)
{
    if (!d_w || !d_wp) return FRCNN_EINVAL;
    return launch_pack_fc_chw_to_hwc(d_w, d_wp, out_features, channels, pooled_hw, as_stream(stream));
}

int frcnn_pack_stack_rows(const float* d_w1, const float* d_b1, int n1, const float* d_w2, const float* d_b2,
                          int n2, int k, int n_pad, float* d_w_out, float* d_b_out, void* stream)
{
    if (!d_w1 || !d_b1 || (n2 > 0 && (!d_w2 || !d_b2)) || !d_w_out || !d_b_out) return FRCNN_EINVAL;
    return launch_pack_stack_rows(d_w1, d_b1, n1, d_w2, d_b2, n2, k, n_pad, d_w_out, d_b_out, as_stream(stream));
}

int frcnn_conv3x3_c3(const float* d_x, const float* d_wp, const float* d_bias, float* d_y, int H, int W,
                     int cout, unsigned flags, void* stream)
{
    if (!d_x || !d_wp || !d_bias || !d_y) return FRCNN_EINVAL;
    if (flags & FRCNN_POOL2) return FRCNN_EUNSUPPORTED;
    return launch_conv3x3_c3(d_x, d_wp, d_bias, d_y, H, W, cout, flags, as_stream(stream));
}

int frcnn_conv3x3_c3_cmax(const float* d_x, const float* d_wp, const float* d_bias, float* d_y, int H, int W,
                          int cout, unsigned flags, float* d_cmax_out, void* stream)
{
    if (!d_x || !d_wp || !d_bias || !d_y || !d_cmax_out) return FRCNN_EINVAL;
    if (flags & FRCNN_POOL2) return FRCNN_EUNSUPPORTED;
    return launch_conv3x3_c3(d_x, d_wp, d_bias, d_y, H, W, cout, flags, as_stream(stream), d_cmax_out);
}

size_t frcnn_conv3x3_workspace_bytes(int H, int W, int cin, int cout) { return conv3x3_workspace_bytes(H, W, cin, cout); }

int frcnn_conv3x3_nhwc(const float* d_x, const float* d_wp, const float* d_bias, float* d_y, int H, int W,
                       int cin, int cout, unsigned flags, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_wp || !d_bias || !d_y) return FRCNN_EINVAL;
    return launch_conv3x3_nhwc(d_x, d_wp, d_bias, d_y, H, W, cin, cout, flags, d_ws, ws_bytes, as_stream(stream));
}

int frcnn_conv3x3_uses_winograd(int cin, int cout) { return conv3x3_uses_winograd(cin, cout) ? 1 : 0; }
int frcnn_resnet_block_uses_winograd(int width, int stride) { return resnet_block_uses_winograd(width, stride) ? 1 : 0; }

int frcnn_pack_conv3x3_winograd(const float* d_w, const float* d_row_scale, float* d_u, int cout, int cin, void* stream)
{
    if (!d_w || !d_u) return FRCNN_EINVAL;
    return launch_pack_conv3x3_winograd(d_w, d_row_scale, d_u, cout, cin, as_stream(stream));
}

int frcnn_pack_conv3x3_winograd_taps(const float* d_wp, float* d_u, int cout, int cin, int data_gradient, void* stream)
{
    if (!d_wp || !d_u) return FRCNN_EINVAL;
    return launch_pack_conv3x3_winograd_taps(d_wp, d_u, cout, cin, data_gradient, as_stream(stream));
}

size_t frcnn_conv3x3_winograd_workspace_bytes(int n_maps, int H, int W, int cin, int cout)
{
    return conv3x3_winograd_workspace_bytes(n_maps, H, W, cin, cout);
}

int frcnn_conv3x3_nhwc_winograd(const float* d_x, const float* d_u, const float* d_bias, float* d_y, int n_maps, int H, int W,
                                int cin, int cout, unsigned flags, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_u || !d_bias || !d_y || n_maps < 1 || H < 1 || W < 1) return FRCNN_EINVAL;
    return launch_conv3x3_winograd(d_x, d_u, d_bias, d_y, n_maps, H, W, cin, cout, flags, d_ws, ws_bytes, as_stream(stream));
}

int frcnn_conv3x3_uses_winograd_fused(int cin, int cout) { return conv3x3_uses_winograd_fused(cin, cout) ? 1 : 0; }
int frcnn_resnet_block_uses_winograd_fused(int n_maps, int width, int stride) { return resnet_block_uses_winograd_fused(n_maps, width, stride) ? 1 : 0; }

int frcnn_pack_conv3x3_winograd_fused(const float* d_w, const float* d_row_scale, float* d_u, int cout, int cin, void* stream)
{
    if (!d_w || !d_u) return FRCNN_EINVAL;
    return launch_pack_conv3x3_winograd_fused(d_w, d_row_scale, d_u, cout, cin, as_stream(stream));
}

int frcnn_pack_conv3x3_winograd_fused_taps(const float* d_wp, float* d_u, int cout, int cin, int data_gradient, void* stream)
{
    if (!d_wp || !d_u) return FRCNN_EINVAL;
    return launch_pack_conv3x3_winograd_fused_taps(d_wp, d_u, cout, cin, data_gradient, as_stream(stream));
}

int frcnn_conv3x3_nhwc_winograd_fused(const float* d_x, const float* d_u, const float* d_bias, float* d_y, int H, int W,
                                      int cin, int cout, unsigned flags, void* stream)
{
    if (!d_x || !d_u || !d_bias || !d_y || H < 1 || W < 1) return FRCNN_EINVAL;
    return launch_conv3x3_winograd_fused(d_x, d_u, d_bias, d_y, H, W, cin, cout, flags, as_stream(stream));
}

int frcnn_conv3x3_nhwc_winograd_fused_maps(const float* d_x, const float* d_u, const float* d_bias, float* d_y, int n_maps, int H, int W,
                                           int cin, int cout, unsigned flags, void* stream)
{
    if (!d_x || !d_u || !d_bias || !d_y || H < 1 || W < 1 || n_maps < 1) return FRCNN_EINVAL;
    return launch_conv3x3_winograd_fused(d_x, d_u, d_bias, d_y, H, W, cin, cout, flags, as_stream(stream), n_maps);
}

size_t frcnn_x6t_record_bytes(int rows_padded, int K)
{
    return (rows_padded > 0 && rows_padded % 32 == 0 && K >= 16 && K % 16 == 0) ? x6t_record_bytes(rows_padded, K) : 0;
}

int frcnn_split_rows_x6t(const float* d_a, int lda, size_t a_batch_floats, void* d_rec, int rows, int rows_padded, int K, int batches,
                         void* stream)
{
    if (!d_a || !d_rec) return FRCNN_EINVAL;
    return launch_split_rows_x6t(d_a, lda, a_batch_floats, d_rec, rows, rows_padded, K, batches, as_stream(stream));
}

size_t frcnn_gemm_x6t_workspace_bytes(int M, int N, int K, int batches) { return gemm_x6t_workspace_bytes(M, N, K, batches); }

int frcnn_gemm_x6t(const void* d_a_rec, int a_rows, size_t a_batch_bytes, const void* d_b_rec, int b_rows, size_t b_batch_bytes,
                   const float* d_bias, const float* d_residual, float* d_c, int ldc, size_t c_batch_floats, int M, int N, int K,
                   int batches, unsigned flags, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_a_rec || !d_b_rec || !d_c) return FRCNN_EINVAL;
    return launch_gemm_x6t(d_a_rec, a_rows, a_batch_bytes, d_b_rec, b_rows, b_batch_bytes, d_bias, d_residual, d_c, ldc, c_batch_floats,
                           M, N, K, batches, flags, d_ws, ws_bytes, as_stream(stream));
}

size_t frcnn_x3t_record_bytes(int rows_padded, int K)
{
    return rows_padded > 0 && rows_padded % 32 == 0 && K >= 16 && K % 16 == 0 ? x3t_record_bytes(rows_padded, K) : 0;
}

int frcnn_rows_scale_x3t(const float* d_a, int lda, size_t a_batch_floats, float* d_inv_scale, int rows, int rows_padded, int K, int batches,
                         void* stream)
{
    if (!d_a || !d_inv_scale) return FRCNN_EINVAL;
    return launch_rows_scale_x3t(d_a, lda, a_batch_floats, d_inv_scale, rows, rows_padded, K, batches, as_stream(stream));
}

int frcnn_split_rows_x3t(const float* d_a, int lda, size_t a_batch_floats, const float* d_inv_scale, void* d_rec, int rows, int rows_padded,
                         int K, int batches, void* stream)
{
    if (!d_a || !d_rec || !d_inv_scale) return FRCNN_EINVAL;
    return launch_split_rows_x3t(d_a, lda, a_batch_floats, d_inv_scale, d_rec, rows, rows_padded, K, batches, as_stream(stream));
}

size_t frcnn_gemm_x3t_workspace_bytes(int M, int N, int K, int batches) { return gemm_x3t_workspace_bytes(M, N, K, batches); }

int frcnn_gemm_x3t(const void* d_a_rec, const float* d_a_inv, int a_rows, size_t a_batch_bytes, size_t a_inv_batch_floats,
                   const void* d_b_rec, const float* d_b_inv, int b_rows, size_t b_batch_bytes, size_t b_inv_batch_floats,
                   const float* d_bias, const float* d_residual, float* d_c, int ldc, size_t c_batch_floats, int M, int N, int K,
                   int batches, unsigned flags, void* d_ws, size_t ws_bytes, void* stream)
{
    return launch_gemm_x3t(d_a_rec, d_a_inv, a_rows, a_batch_bytes, a_inv_batch_floats, d_b_rec, d_b_inv, b_rows, b_batch_bytes,
                           b_inv_batch_floats, d_bias, d_residual, d_c, ldc, c_batch_floats, M, N, K, batches, flags, d_ws, ws_bytes,
                           as_stream(stream));
}

size_t frcnn_conv3x3_winograd_x3_fused_workspace_bytes(int n_maps, int H, int W)
{
    return n_maps > 0 && H > 0 && W > 0 ? conv3x3_winograd_x3_fused_workspace_bytes(n_maps, H, W) : 0;
}

size_t frcnn_conv3x3_winograd_x3_pair_workspace_bytes(int n_maps, int H, int W, int cout)
{
    return n_maps > 0 && H > 0 && W > 0 && cout > 0 ? conv3x3_winograd_x3_pair_workspace_bytes(n_maps, H, W, cout) : 0;
}

int frcnn_conv3x3_nhwc_winograd_x3_fused(const float* d_x, const void* d_blob, const float* d_bias, float* d_y, int n_maps, int H, int W,
                                         int cin, int cout, unsigned flags, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_blob || !d_bias || !d_y) return FRCNN_EINVAL;
    return launch_conv3x3_winograd_x3_fused(d_x, d_blob, d_bias, d_y, n_maps, H, W, cin, cout, flags, d_ws, ws_bytes, as_stream(stream));
}

int frcnn_pixel_absmax(const float* d_x, float* d_cmax, long long pixels, int c, void* stream)
{
    if (!d_x || !d_cmax) return FRCNN_EINVAL;
    return launch_pixel_absmax(d_x, d_cmax, pixels, c, as_stream(stream));
}

int frcnn_split_pixels_x3t(const float* d_x, const float* d_cmax, void* d_rec, float* d_inv_scale, int n_maps, int H, int W, int c, int stride,
                           int rows_padded, void* stream)
{
    if (!d_x || !d_rec) return FRCNN_EINVAL;
    return launch_split_pixels_x3t(d_x, d_cmax, d_rec, d_inv_scale, n_maps, H, W, c, stride, rows_padded, as_stream(stream));
}

int frcnn_split_patches3x3_x3t(const float* d_x, const float* d_cmax, void* d_rec, float* d_inv_scale, int n_maps, int H, int W, int c,
                               int stride, int rows_padded, void* stream)
{
    if (!d_x || !d_rec) return FRCNN_EINVAL;
    return launch_split_patches3x3_x3t(d_x, d_cmax, d_rec, d_inv_scale, n_maps, H, W, c, stride, rows_padded, as_stream(stream));
}

size_t frcnn_x3t_blob_bytes(int rows_padded, int K, int batches)
{
    return rows_padded > 0 && rows_padded % 32 == 0 && K >= 16 && K % 16 == 0 && batches >= 1 ? x3t_blob_bytes(rows_padded, K, batches) : 0;
}

int frcnn_pack_rows_x3t(const float* d_a, int lda, size_t a_batch_floats, void* d_blob, int rows, int rows_padded, int K, int batches,
                        void* stream)
{
    if (!d_a || !d_blob) return FRCNN_EINVAL;
    return launch_pack_rows_x3t(d_a, lda, a_batch_floats, d_blob, rows, rows_padded, K, batches, as_stream(stream));
}

size_t frcnn_conv3x3_winograd_x3_pack_bytes(int cout, int cin) { return conv3x3_winograd_x3_pack_bytes(cout, cin); }

int frcnn_pack_conv3x3_winograd_x3(const float* d_u_f32, void* d_blob, int cout, int cin, void* stream)
{
    if (!d_u_f32 || !d_blob) return FRCNN_EINVAL;
    return launch_pack_conv3x3_winograd_x3(d_u_f32, d_blob, cout, cin, as_stream(stream));
}

size_t frcnn_conv3x3_winograd_x3_workspace_bytes(int n_maps, int H, int W, int cin, int cout)
{
    return conv3x3_winograd_x3_workspace_bytes(n_maps, H, W, cin, cout);
}

int frcnn_conv3x3_nhwc_winograd_x3(const float* d_x, const void* d_blob, const float* d_bias, float* d_y, int n_maps, int H, int W, int cin,
                                   int cout, unsigned flags, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_blob || !d_bias || !d_y) return FRCNN_EINVAL;
    return launch_conv3x3_winograd_x3(d_x, d_blob, d_bias, d_y, n_maps, H, W, cin, cout, flags, d_ws, ws_bytes, as_stream(stream));
}

int frcnn_conv3x3_nhwc_winograd_x3_chain(const float* d_x, const void* d_blob, const float* d_bias, float* d_y, int n_maps, int H, int W, int cin,
                                         int cout, unsigned flags, int one_launch, void* d_ws, size_t ws_bytes, const float* d_cmax_in,
                                         float* d_cmax_out, void* stream)
{
    if (!d_x || !d_blob || !d_bias || !d_y) return FRCNN_EINVAL;
    if (one_launch)
        return launch_conv3x3_winograd_x3_fused(d_x, d_blob, d_bias, d_y, n_maps, H, W, cin, cout, flags, d_ws, ws_bytes, as_stream(stream),
                                                d_cmax_in, d_cmax_out);
    return launch_conv3x3_winograd_x3(d_x, d_blob, d_bias, d_y, n_maps, H, W, cin, cout, flags, d_ws, ws_bytes, as_stream(stream), d_cmax_in,
                                      d_cmax_out);
}

int frcnn_split_pixels_x6t(const float* d_x, void* d_rec, int N, int H, int W, int C, int stride, int rows_padded, void* stream)
{
    if (!d_x || !d_rec) return FRCNN_EINVAL;
    return launch_split_pixels_x6t(d_x, d_rec, N, H, W, C, stride, rows_padded, as_stream(stream));
}

int frcnn_conv3x3_uses_winograd_x6(int cin, int cout) { return conv3x3_uses_winograd_x6(cin, cout) ? 1 : 0; }
size_t frcnn_conv3x3_winograd_x6_pack_bytes(int cout, int cin) { return conv3x3_winograd_x6_pack_bytes(cout, cin); }

int frcnn_pack_conv3x3_winograd_x6(const float* d_w, const float* d_row_scale, void* d_u_rec, int cout, int cin, void* stream)
{
    if (!d_w || !d_u_rec) return FRCNN_EINVAL;
    return launch_pack_conv3x3_winograd_x6(d_w, d_row_scale, d_u_rec, cout, cin, as_stream(stream));
}

size_t frcnn_conv3x3_winograd_x6_workspace_bytes(int n_maps, int H, int W, int cin, int cout)
{
    return conv3x3_winograd_x6_workspace_bytes(n_maps, H, W, cin, cout);
}

int frcnn_conv3x3_nhwc_winograd_x6(const float* d_x, const void* d_u_rec, const float* d_bias, float* d_y, int n_maps, int H, int W, int cin,
                                   int cout, unsigned flags, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_u_rec || !d_bias || !d_y) return FRCNN_EINVAL;
    return launch_conv3x3_winograd_x6(d_x, d_u_rec, d_bias, d_y, n_maps, H, W, cin, cout, flags, d_ws, ws_bytes, as_stream(stream));
}

int frcnn_split_patches3x3_x6t(const float* d_x, void* d_rec, int N, int H, int W, int C, int stride, int rows_padded, void* stream)
{
    if (!d_x || !d_rec) return FRCNN_EINVAL;
    return launch_split_patches3x3_x6t(d_x, d_rec, N, H, W, C, stride, rows_padded, as_stream(stream));
}

int frcnn_maxpool2x2_nhwc(const float* d_x, float* d_y, int H, int W, int c, void* stream)
{
    if (!d_x || !d_y) return FRCNN_EINVAL;
    return launch_maxpool2x2(d_x, d_y, H, W, c, as_stream(stream));
}

int frcnn_fold_bn_pack(const float* d_w, const float* d_gamma, const float* d_beta, const float* d_mean,
                       const float* d_var, float eps, int cout, int cin, int ksize, float* d_wp, float* d_bp, void* stream)
{
    if (!d_w || !d_gamma || !d_beta || !d_mean || !d_var || !d_wp || !d_bp) return FRCNN_EINVAL;
    return launch_fold_bn_pack(d_w, d_gamma, d_beta, d_mean, d_var, eps, cout, cin, ksize, d_wp, d_bp, as_stream(stream));
}

size_t frcnn_conv_workspace_bytes(int N, int H, int W, int cin, int cout, int ksize, int stride, int pad)
{
    return conv_gather_workspace_bytes(N, H, W, cin, cout, ksize, stride, pad);
}

int frcnn_conv_nhwc(const float* d_x, const float* d_wp, const float* d_bias, const float* d_residual, float* d_y,
                    int N, int H, int W, int cin, int cout, int ksize, int stride, int pad, unsigned flags,
                    void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_wp || !d_bias || !d_y) return FRCNN_EINVAL;
    if (flags & FRCNN_POOL2) return FRCNN_EUNSUPPORTED;
    return launch_conv_gather(d_x, d_wp, d_bias, d_residual, d_y, N, H, W, cin, cout, ksize, stride, pad, flags,
                              d_ws, ws_bytes, as_stream(stream));
}

int frcnn_conv_nhwc_math(const float* d_x, const float* d_wp, const float* d_bias, const float* d_residual, float* d_y,
                         int N, int H, int W, int cin, int cout, int ksize, int stride, int pad, unsigned flags, int math,
                         void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_wp || !d_bias || !d_y) return FRCNN_EINVAL;
    if (flags & FRCNN_POOL2) return FRCNN_EUNSUPPORTED;
    return launch_conv_gather(d_x, d_wp, d_bias, d_residual, d_y, N, H, W, cin, cout, ksize, stride, pad, flags,
                              d_ws, ws_bytes, as_stream(stream), math);
}

int frcnn_x3_saturation_events(unsigned long long* out)
{
    if (!out) return FRCNN_EINVAL;
    const unsigned* c = x3_saturation_counter();
    *out = c ? (unsigned long long)*reinterpret_cast<const volatile unsigned*>(c) : 0ull;
    return FRCNN_OK;
}

int frcnn_conv_nhwc_x3g(const float* d_x, const float* d_wp, const float* d_bias, const float* d_residual, float* d_y,
                        int N, int H, int W, int cin, int cout, int ksize, int stride, int pad, unsigned flags,
                        const float* d_xmax, const float* d_wmax, float* d_ymax, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_wp || !d_bias || !d_y || !d_xmax || !d_wmax) return FRCNN_EINVAL;
    if (flags & FRCNN_POOL2) return FRCNN_EUNSUPPORTED;
    const GatherX3 x3{d_xmax, d_wmax, d_ymax, nullptr, (flags & FRCNN_X3G_WSPLIT) != 0};
    return launch_conv_gather(d_x, d_wp, d_bias, d_residual, d_y, N, H, W, cin, cout, ksize, stride, pad, flags,
                              d_ws, ws_bytes, as_stream(stream), FRCNN_CONV_F32X3G, &x3);
}

int frcnn_pack_conv_x3g_weights(const float* d_wp, const float* d_wmax, void* d_out, int taps, int cout, int cin, void* stream)
{
    if (taps < 1 || cout < 1) return FRCNN_EINVAL;
    return launch_pack_x3g_weights(d_wp, d_wmax, d_out, (long long)taps * cout, cin, as_stream(stream));
}

int frcnn_conv_nhwc_x3g_tickets(const float* d_x, const float* d_wp, const float* d_bias, const float* d_residual, float* d_y,
                                int N, int H, int W, int cin, int cout, int ksize, int stride, int pad, unsigned flags,
                                const float* d_xmax, const float* d_wmax, float* d_ymax, void* d_ws, size_t ws_bytes,
                                unsigned* d_tile_counters, void* stream)
{
    static_assert(FRCNN_X3G_TILE_COUNTERS == GX_TILE_COUNTERS, "one bound on both sides of the ABI");
    if (!d_x || !d_wp || !d_bias || !d_y || !d_xmax || !d_wmax || !d_tile_counters) return FRCNN_EINVAL;
    if (flags & FRCNN_POOL2) return FRCNN_EUNSUPPORTED;
    const GatherX3 x3{d_xmax, d_wmax, d_ymax, d_tile_counters, (flags & FRCNN_X3G_WSPLIT) != 0};
    return launch_conv_gather(d_x, d_wp, d_bias, d_residual, d_y, N, H, W, cin, cout, ksize, stride, pad, flags,
                              d_ws, ws_bytes, as_stream(stream), FRCNN_CONV_F32X3G, &x3);
}

int frcnn_tensor_absmax(const float* d_x, long long n, float* d_out, void* stream)
{
    if (!d_x || !d_out) return FRCNN_EINVAL;
    return launch_tensor_absmax(d_x, n, d_out, as_stream(stream));
}

int frcnn_conv7x7_s2_c3(const float* d_x, const float* d_wp, const float* d_bias, float* d_y, int H, int W,
                        int cout, unsigned flags, void* stream)
{
    if (!d_x || !d_wp || !d_bias || !d_y) return FRCNN_EINVAL;
    return launch_conv7x7_s2_c3(d_x, d_wp, d_bias, d_y, H, W, cout, flags, as_stream(stream));
}

int frcnn_maxpool3x3_s2_nhwc(const float* d_x, float* d_y, int H, int W, int c, void* stream)
{
    if (!d_x || !d_y) return FRCNN_EINVAL;
    return launch_maxpool3x3_s2(d_x, d_y, H, W, c, as_stream(stream));
}

int frcnn_spatial_mean_nhwc(const float* d_x, float* d_y, int N, int H, int W, int c, void* stream)
{
    if (!d_x || !d_y) return FRCNN_EINVAL;
    return launch_spatial_mean(d_x, d_y, N, H, W, c, as_stream(stream));
}

size_t frcnn_linear_workspace_bytes(int M, int N, int K) { return linear_workspace_bytes(M, N, K); }

int frcnn_linear(const float* d_a, int lda, const float* d_w, const float* d_bias, float* d_y, int ldy,
                 int M, int N, int K, unsigned flags, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_a || !d_w || !d_bias || !d_y) return FRCNN_EINVAL;
    return launch_linear(d_a, lda, d_w, d_bias, d_y, ldy, M, N, K, flags, d_ws, ws_bytes, as_stream(stream));
}

int frcnn_softmax_rows(const float* d_x, int ldx, float* d_y, int M, int ncls, void* stream)
{
    if (!d_x || !d_y) return FRCNN_EINVAL;
    return launch_softmax_rows(d_x, ldx, d_y, M, ncls, as_stream(stream));
}

int frcnn_rpn_proposals(frcnn_ctx* ctx, const float* d_head, int ld_head, const float* d_anchor_map,
                        const float* d_valid_map, int fh, int fw, int image_h, int image_w, int pre_nms,
                        int post_nms, float nms_threshold, float min_side, float* d_scores,
                        int32_t* d_sorted_idx, float* d_props, int32_t* d_counts, void* stream)
{
    if (!ctx || !d_head || !d_anchor_map || !d_scores || !d_sorted_idx || !d_props || !d_counts) return FRCNN_EINVAL;
    FRCNN_HIP_TRY(hipMemsetAsync(d_counts, 0, 4 * sizeof(int32_t), as_stream(stream)));
    return launch_rpn_proposals(ctx->ps, d_head, ld_head, d_anchor_map, d_valid_map, fh, fw, image_h, image_w,
                                pre_nms, post_nms, nms_threshold, min_side, d_scores, d_sorted_idx, d_props,
                                d_counts, as_stream(stream));
}

int frcnn_nms(frcnn_ctx* ctx, const float* d_boxes, const float* d_scores, int n, float threshold,
              int max_keep, int32_t* d_keep, int32_t* d_n_keep, void* stream)
{
    if (!ctx || !d_boxes || !d_scores || !d_keep || !d_n_keep) return FRCNN_EINVAL;
    return launch_nms(ctx->ps, d_boxes, d_scores, n, threshold, max_keep, d_keep, d_n_keep, as_stream(stream));
}

int frcnn_roi_pool(const float* d_fm, int fh, int fw, int c, const float* d_rois, const int32_t* d_n_rois,
                   int max_rois, int pooled, float spatial_scale, float* d_out, void* stream)
{
    if (!d_fm || !d_rois || !d_n_rois || !d_out) return FRCNN_EINVAL;
    return launch_roi_pool(d_fm, fh, fw, c, d_rois, d_n_rois, max_rois, pooled, spatial_scale, d_out,
                           as_stream(stream));
}

int frcnn_roi_pool_x3t(const float* d_fm, int fh, int fw, int c, const float* d_rois, const int32_t* d_n_rois, int max_rois, int pooled,
                       float spatial_scale, float* d_cmax, float* d_inv_scale, void* d_rec, int rec_rows, void* stream)
{
    if (!d_fm || !d_rois || !d_n_rois || !d_cmax || !d_inv_scale || !d_rec) return FRCNN_EINVAL;
    return launch_roi_pool_x3t(d_fm, fh, fw, c, d_rois, d_n_rois, max_rois, pooled, spatial_scale, d_cmax, d_inv_scale, d_rec, rec_rows,
                               as_stream(stream));
}

int frcnn_roi_align(const float* d_fm, int fh, int fw, int c, const float* d_rois, const int32_t* d_n_rois, int max_rois,
                    int pooled, float spatial_scale, int sampling_ratio, int aligned, float* d_out, void* stream)
{
    if (!d_fm || !d_rois || !d_n_rois || !d_out) return FRCNN_EINVAL;
    return launch_roi_align(d_fm, fh, fw, c, d_rois, d_n_rois, max_rois, pooled, spatial_scale, sampling_ratio, aligned, d_out,
                            as_stream(stream));
}

int frcnn_roi_align_backward(const float* d_rois, int n_rois, int fh, int fw, int c, int pooled, float spatial_scale,
                             int sampling_ratio, int aligned, const float* d_dout, float* d_dfm, int accumulate, void* stream)
{
    if (!d_dfm || (n_rois > 0 && (!d_rois || !d_dout))) return FRCNN_EINVAL;
    return launch_roi_align_backward(d_rois, n_rois, fh, fw, c, pooled, spatial_scale, sampling_ratio, aligned, d_dout, d_dfm,
                                     accumulate, as_stream(stream));
}

int frcnn_detections(const float* d_props, const float* d_classes, const float* d_deltas,
                     const int32_t* d_n_rois, int max_rois, int ncls, int image_h, int image_w,
                     float score_threshold, float nms_threshold, double* d_out, int32_t* d_out_cnt, void* stream)
{
    if (!d_props || !d_classes || !d_deltas || !d_n_rois || !d_out || !d_out_cnt) return FRCNN_EINVAL;
    return launch_detections(d_props, d_classes, d_deltas, d_n_rois, max_rois, ncls, image_h, image_w,
                             score_threshold, nms_threshold, d_out, d_out_cnt, as_stream(stream));
}

// ---- training path -----------------------------------------------------------------------------
int frcnn_label_proposals(const float* d_props, const int32_t* d_n_props, int max_props,
                          const float* d_gt_boxes, const int32_t* d_gt_class_idx, int n_gt, int num_classes,
                          float min_background_iou, float min_object_iou,
                          const float box_delta_means[4], const float box_delta_stds[4],
                          float* d_out_props, int32_t* d_out_class_idx, float* d_out_gt_classes,
                          float* d_out_gt_box_deltas, int32_t* d_out_count, void* stream)
{
    if ((!d_props && max_props > 0) || !d_n_props || !d_gt_boxes || !d_gt_class_idx || !box_delta_means || !box_delta_stds ||
        !d_out_props || !d_out_class_idx || !d_out_gt_classes || !d_out_gt_box_deltas || !d_out_count)
        return FRCNN_EINVAL;
    if (!(min_background_iou < min_object_iou)) return FRCNN_EINVAL;      // faster_rcnn.py:421 assert
    return launch_label_proposals(d_props, d_n_props, max_props, d_gt_boxes, d_gt_class_idx, n_gt, num_classes,
                                  min_background_iou, min_object_iou, box_delta_means, box_delta_stds, d_out_props,
                                  d_out_class_idx, d_out_gt_classes, d_out_gt_box_deltas, d_out_count, as_stream(stream));
}

int frcnn_gather_rows(const float* d_src, const int32_t* d_idx, int n, int row_floats, float* d_dst, void* stream)
{
    if (n > 0 && (!d_src || !d_idx || !d_dst)) return FRCNN_EINVAL;
    return launch_gather_rows(d_src, d_idx, n, row_floats, d_dst, as_stream(stream));
}

int frcnn_rpn_loss(const float* d_head, int ld_head, int cells, const int32_t* d_sample, int n_sample,
                   const float* d_rpn_map, float* d_losses, float* d_grad_head, void* stream)
{
    if (!d_head || (n_sample > 0 && !d_sample) || !d_rpn_map || !d_losses) return FRCNN_EINVAL;
    return launch_rpn_loss(d_head, ld_head, cells, d_sample, n_sample, d_rpn_map, d_losses, d_grad_head, as_stream(stream));
}

int frcnn_detector_loss(const float* d_classes, const float* d_deltas, const float* d_gt_classes,
                        const float* d_gt_box_deltas, int n, int num_classes, float* d_losses,
                        float* d_grad_logits, int ld_grad, void* stream)
{
    if (!d_losses || (n > 0 && (!d_classes || !d_deltas || !d_gt_classes || !d_gt_box_deltas))) return FRCNN_EINVAL;
    return launch_detector_loss(d_classes, d_deltas, d_gt_classes, d_gt_box_deltas, n, num_classes, d_losses,
                                d_grad_logits, ld_grad, as_stream(stream));
}

size_t frcnn_gemm_tn_workspace_bytes(int M, int N, int R) { return M > 0 && N > 0 && R > 0 ? gemm_tn_workspace_bytes(M, N, R, 1) : 0; }

int frcnn_gemm_tn(const float* d_a, int lda, const float* d_b, int ldb, float* d_c, int ldc,
                  int M, int N, int R, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_a || !d_b || !d_c) return FRCNN_EINVAL;
    return launch_gemm_tn(d_a, lda, d_b, ldb, d_c, ldc, M, N, R, d_ws, ws_bytes, as_stream(stream));
}

int frcnn_gemm_tn_math(const float* d_a, int lda, const float* d_b, int ldb, float* d_c, int ldc,
                       int M, int N, int R, int grad_math, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_a || !d_b || !d_c) return FRCNN_EINVAL;
    return launch_gemm_tn(d_a, lda, d_b, ldb, d_c, ldc, M, N, R, d_ws, ws_bytes, as_stream(stream), grad_math);
}

size_t frcnn_conv3x3_wgrad_workspace_bytes(int H, int W, int cin, int cout)
{
    return H > 0 && W > 0 && cin > 0 && cout > 0 ? gemm_tn_workspace_bytes(cout, cin, H * W, 9) : 0;
}

int frcnn_conv3x3_wgrad(const float* d_x, const float* d_dz, float* d_dwp, int H, int W, int cin, int cout,
                        void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_dz || !d_dwp) return FRCNN_EINVAL;
    return launch_conv3x3_wgrad(d_x, d_dz, d_dwp, H, W, cin, cout, d_ws, ws_bytes, as_stream(stream));
}

int frcnn_conv3x3_wgrad_math(const float* d_x, const float* d_dz, float* d_dwp, int H, int W, int cin, int cout,
                             int grad_math, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_dz || !d_dwp) return FRCNN_EINVAL;
    return launch_conv3x3_wgrad(d_x, d_dz, d_dwp, H, W, cin, cout, d_ws, ws_bytes, as_stream(stream), grad_math);
}

int frcnn_pack_conv3x3_dgrad(const float* d_wp, float* d_wd, int cout, int cin, void* stream)
{
    if (!d_wp || !d_wd) return FRCNN_EINVAL;
    return launch_pack_conv3x3_dgrad(d_wp, d_wd, cout, cin, as_stream(stream));
}

size_t frcnn_conv_wgrad_workspace_bytes(int N, int H, int W, int cin, int cout, int ksize, int stride, int pad)
{
    if (N < 1 || H < 1 || W < 1 || cin < 1 || cout < 1 || ksize < 1 || stride < 1 || pad < 0) return 0;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    return Ho > 0 && Wo > 0 ? gemm_tn_workspace_bytes(cout, cin, N * Ho * Wo, ksize * ksize) : 0;
}

int frcnn_conv_wgrad(const float* d_x, const float* d_dz, float* d_dwp, int N, int H, int W, int cin, int cout,
                     int ksize, int stride, int pad, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_dz || !d_dwp) return FRCNN_EINVAL;
    return launch_conv_wgrad(d_x, d_dz, d_dwp, N, H, W, cin, cout, ksize, stride, pad, d_ws, ws_bytes, as_stream(stream));
}

int frcnn_conv_wgrad_math(const float* d_x, const float* d_dz, float* d_dwp, int N, int H, int W, int cin, int cout,
                          int ksize, int stride, int pad, int grad_math, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_x || !d_dz || !d_dwp) return FRCNN_EINVAL;
    return launch_conv_wgrad(d_x, d_dz, d_dwp, N, H, W, cin, cout, ksize, stride, pad, d_ws, ws_bytes, as_stream(stream), grad_math);
}

size_t frcnn_conv_dgrad_workspace_bytes(int N, int H, int W, int cin, int cout, int ksize, int stride, int pad)
{
    return conv_dgrad_workspace_bytes(N, H, W, cin, cout, ksize, stride, pad);
}

int frcnn_conv_dgrad(const float* d_dz, const float* d_wd, const float* d_residual, float* d_dx, int N, int H, int W,
                     int cin, int cout, int ksize, int stride, int pad, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_dz || !d_wd || !d_dx) return FRCNN_EINVAL;
    return launch_conv_dgrad(d_dz, d_wd, d_residual, d_dx, N, H, W, cin, cout, ksize, stride, pad, d_ws, ws_bytes,
                             as_stream(stream));
}

int frcnn_conv_dgrad_math(const float* d_dz, const float* d_wd, const float* d_residual, float* d_dx, int N, int H, int W,
                          int cin, int cout, int ksize, int stride, int pad, int math, void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_dz || !d_wd || !d_dx) return FRCNN_EINVAL;
    return launch_conv_dgrad(d_dz, d_wd, d_residual, d_dx, N, H, W, cin, cout, ksize, stride, pad, d_ws, ws_bytes,
                             as_stream(stream), math);
}

// ---- ABI 16: the backward of ONE trainable bottleneck as one call (fasterrcnn_amd/training.py: _TrainBlock.backward) ---------------------
namespace {
int tc_ok(const frcnn_train_conv* c)
{
    return c && c->folded && c->scale && c->grad && c->wd && c->cin > 0 && c->cout > 0 && c->ksize > 0 && c->stride > 0 && c->pad >= 0;
}
// The gradient of the RAW weight of a conv + frozen BatchNorm: the folded weight's gradient times the BN scale of its output channel.
int tc_wgrad(const frcnn_train_conv* c, const float* x, const float* dz, int N, int H, int W, int math, void* ws, size_t ws_bytes, hipStream_t s)
{
    int rc = launch_conv_wgrad(x, dz, c->grad, N, H, W, c->cin, c->cout, c->ksize, c->stride, c->pad, ws, ws_bytes, s, math);
    if (rc) return rc;
    return launch_scale_rows(c->grad, c->scale, c->grad, c->ksize * c->ksize, c->cout, c->cin, s);
}
int tc_dgrad(const frcnn_train_conv* c, const float* dz, const float* residual, float* dx, int N, int H, int W, int math, void* ws, size_t ws_bytes,
             hipStream_t s)
{
    int rc = launch_pack_conv_dgrad(c->folded, c->wd, c->ksize * c->ksize, c->cout, c->cin, s);
    if (rc) return rc;
    return launch_conv_dgrad(dz, c->wd, residual, dx, N, H, W, c->cin, c->cout, c->ksize, c->stride, c->pad, ws, ws_bytes, s, math);
}
size_t tc_wgrad_ws(const frcnn_train_conv* c, int N, int H, int W)
{
    return frcnn_conv_wgrad_workspace_bytes(N, H, W, c->cin, c->cout, c->ksize, c->stride, c->pad);
}
size_t tc_dgrad_ws(const frcnn_train_conv* c, int N, int H, int W)
{
    return conv_dgrad_workspace_bytes(N, H, W, c->cin, c->cout, c->ksize, c->stride, c->pad);
}
// the second stream runs behind everything the main stream holds at this point
// (events from a ring per device, made once: hipStreamWaitEvent takes the event's state at the call, so an entry may be recorded again
//  as soon as its wait has been enqueued; creating and destroying one per use measured ~25 us of host time per block)
int side_behind_main(hipStream_t main, hipStream_t side)
{
    if (side == main) return FRCNN_OK;
    static constexpr int RING = 16, MAXDEV = 16;
    static thread_local hipEvent_t ring[MAXDEV][RING];
    static thread_local bool made[MAXDEV];
    static thread_local unsigned next[MAXDEV];
    int dev = 0;
    FRCNN_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAXDEV) return FRCNN_EUNSUPPORTED;
    if (!made[dev]) {
        for (int i = 0; i < RING; ++i) FRCNN_HIP_TRY(hipEventCreateWithFlags(&ring[dev][i], hipEventDisableTiming));
        made[dev] = true;
    }
    hipEvent_t ev = ring[dev][next[dev]++ % RING];
    FRCNN_HIP_TRY(hipEventRecord(ev, main));
    FRCNN_HIP_TRY(hipStreamWaitEvent(side, ev, 0));
    return FRCNN_OK;
}
}  // namespace

int frcnn_bottleneck_backward_workspace_bytes(const frcnn_train_conv* c1, const frcnn_train_conv* c2, const frcnn_train_conv* c3,
                                              const frcnn_train_conv* cd, int N, int H, int W, int Ho, int Wo, size_t* main_bytes,
                                              size_t* side_bytes)
{
    if (!tc_ok(c1) || !tc_ok(c2) || !tc_ok(c3) || (cd && !tc_ok(cd)) || !main_bytes || !side_bytes || N < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1)
        return FRCNN_EINVAL;
    size_t m = std::max(std::max(tc_dgrad_ws(c3, N, Ho, Wo), tc_dgrad_ws(c2, N, H, W)), tc_dgrad_ws(c1, N, H, W));
    size_t w = std::max(std::max(tc_wgrad_ws(c3, N, Ho, Wo), tc_wgrad_ws(c2, N, H, W)), tc_wgrad_ws(c1, N, H, W));
    if (cd) { m = std::max(m, tc_dgrad_ws(cd, N, H, W)); w = std::max(w, tc_wgrad_ws(cd, N, H, W)); }
    *main_bytes = m; *side_bytes = w;
    return FRCNN_OK;
}

int frcnn_bottleneck_backward(const frcnn_train_conv* c1, const frcnn_train_conv* c2, const frcnn_train_conv* c3, const frcnn_train_conv* cd,
                              const float* d_x, const float* d_t1, const float* d_t2, const float* d_out, float* d_g, float* d_dt2,
                              float* d_dt1, float* d_dxid, float* d_dx, int N, int H, int W, int Ho, int Wo, int math, void* d_ws_main,
                              size_t ws_main_bytes, void* d_ws_side, size_t ws_side_bytes, void* stream, void* side_stream)
{
    if (!tc_ok(c1) || !tc_ok(c2) || !tc_ok(c3) || (cd && !tc_ok(cd)) || !d_x || !d_t1 || !d_t2 || !d_out || !d_g || !d_dt2 || !d_dt1 ||
        (cd && d_dx && !d_dxid) || N < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1)
        return FRCNN_EINVAL;
    size_t need_m = 0, need_s = 0;
    int rc = frcnn_bottleneck_backward_workspace_bytes(c1, c2, c3, cd, N, H, W, Ho, Wo, &need_m, &need_s);
    if (rc) return rc;
    if ((need_m && (!d_ws_main || ws_main_bytes < need_m)) || (need_s && (!d_ws_side || ws_side_bytes < need_s))) return FRCNN_EINVAL;
    hipStream_t s = as_stream(stream);
    hipStream_t w = side_stream ? as_stream(side_stream) : s;
    const size_t n_out = (size_t)N * Ho * Wo * c3->cout, n_t2 = (size_t)N * Ho * Wo * c3->cin, n_t1 = (size_t)N * H * W * c2->cin;
#define BSTEP(call) do { rc = (call); if (rc) return rc; } while (0)
    // The chain first -- out = relu(conv3(t2) + identity): the mask of the output's ReLU, conv3's data gradient, t2's mask, conv2's, t1's
    // mask, (downsample's,) conv1's -- then ONE event, and the block's four weight gradients behind it on the second stream: they run under
    // the NEXT block's chain (an event per gradient, each right behind the mask it needs, measured 80 us of host time per block and made the
    // bf16 step host-bound again).  Without a second stream the order is the separate entry points': each gradient before its data gradient.
    const bool two = w != s;
    BSTEP(launch_relu_backward(d_g, d_out, n_out, s));
    if (!two) BSTEP(tc_wgrad(c3, d_t2, d_g, N, Ho, Wo, math, d_ws_side, ws_side_bytes, s));
    BSTEP(tc_dgrad(c3, d_g, nullptr, d_dt2, N, Ho, Wo, math, d_ws_main, ws_main_bytes, s));
    BSTEP(launch_relu_backward(d_dt2, d_t2, n_t2, s));
    if (!two) BSTEP(tc_wgrad(c2, d_t1, d_dt2, N, H, W, math, d_ws_side, ws_side_bytes, s));
    BSTEP(tc_dgrad(c2, d_dt2, nullptr, d_dt1, N, H, W, math, d_ws_main, ws_main_bytes, s));
    BSTEP(launch_relu_backward(d_dt1, d_t1, n_t1, s));
    if (!two) {
        BSTEP(tc_wgrad(c1, d_x, d_dt1, N, H, W, math, d_ws_side, ws_side_bytes, s));
        if (cd) BSTEP(tc_wgrad(cd, d_x, d_g, N, H, W, math, d_ws_side, ws_side_bytes, s));
    }
    if (d_dx) {
        const float* identity_grad = d_g;
        if (cd) { BSTEP(tc_dgrad(cd, d_g, nullptr, d_dxid, N, H, W, math, d_ws_main, ws_main_bytes, s)); identity_grad = d_dxid; }
        BSTEP(tc_dgrad(c1, d_dt1, identity_grad, d_dx, N, H, W, math, d_ws_main, ws_main_bytes, s));
    }
    if (two) {
        BSTEP(side_behind_main(s, w));
        BSTEP(tc_wgrad(c3, d_t2, d_g, N, Ho, Wo, math, d_ws_side, ws_side_bytes, w));
        BSTEP(tc_wgrad(c2, d_t1, d_dt2, N, H, W, math, d_ws_side, ws_side_bytes, w));
        BSTEP(tc_wgrad(c1, d_x, d_dt1, N, H, W, math, d_ws_side, ws_side_bytes, w));
        if (cd) BSTEP(tc_wgrad(cd, d_x, d_g, N, H, W, math, d_ws_side, ws_side_bytes, w));
    }
#undef BSTEP
    return FRCNN_OK;
}

int frcnn_pack_conv_dgrad(const float* d_wp, float* d_wd, int taps, int cout, int cin, void* stream)
{
    if (!d_wp || !d_wd) return FRCNN_EINVAL;
    return launch_pack_conv_dgrad(d_wp, d_wd, taps, cout, cin, as_stream(stream));
}

int frcnn_scale_rows(const float* d_src, const float* d_scale, float* d_dst, int taps, int cout, int cin, void* stream)
{
    if (!d_src || !d_scale || !d_dst) return FRCNN_EINVAL;
    return launch_scale_rows(d_src, d_scale, d_dst, taps, cout, cin, as_stream(stream));
}

int frcnn_bn_scale_shift(const float* d_gamma, const float* d_beta, const float* d_mean, const float* d_var, float eps,
                         int c, float* d_scale, float* d_shift, void* stream)
{
    if (!d_gamma || !d_beta || !d_mean || !d_var || !d_scale || !d_shift) return FRCNN_EINVAL;
    return launch_bn_scale_shift(d_gamma, d_beta, d_mean, d_var, eps, c, d_scale, d_shift, as_stream(stream));
}

int frcnn_spatial_mean_backward(const float* d_dy, float* d_dx, int N, int H, int W, int c, void* stream)
{
    if (!d_dy || !d_dx) return FRCNN_EINVAL;
    return launch_spatial_mean_backward(d_dy, d_dx, N, H, W, c, as_stream(stream));
}

int frcnn_relu_backward(float* d_dy, const float* d_y, size_t n, void* stream)
{
    if (n > 0 && (!d_dy || !d_y)) return FRCNN_EINVAL;
    return launch_relu_backward(d_dy, d_y, n, as_stream(stream));
}

int frcnn_add_inplace(float* d_a, const float* d_b, size_t n, void* stream)
{
    if (n > 0 && (!d_a || !d_b)) return FRCNN_EINVAL;
    return launch_add_inplace(d_a, d_b, n, as_stream(stream));
}

int frcnn_maxpool2x2_backward(const float* d_x, const float* d_dy, float* d_dx, int H, int W, int c, void* stream)
{
    if (!d_x || !d_dy || !d_dx) return FRCNN_EINVAL;
    return launch_maxpool2x2_backward(d_x, d_dy, d_dx, H, W, c, as_stream(stream));
}

size_t frcnn_roi_pool_backward_workspace_bytes(int n_rois, int pooled, int c)
{
    return n_rois > 0 && pooled > 0 && c > 0 ? roi_pool_backward_workspace_bytes(n_rois, pooled, c) : 0;
}

int frcnn_roi_pool_backward(const float* d_fm, int fh, int fw, int c, const float* d_rois, int n_rois, int pooled,
                            float spatial_scale, const float* d_dout, float* d_dfm, int accumulate,
                            void* d_ws, size_t ws_bytes, void* stream)
{
    if (!d_fm || !d_dfm || (n_rois > 0 && (!d_rois || !d_dout))) return FRCNN_EINVAL;
    return launch_roi_pool_backward(d_fm, fh, fw, c, d_rois, n_rois, pooled, spatial_scale, d_dout, d_dfm, accumulate,
                                    d_ws, ws_bytes, as_stream(stream));
}

int frcnn_transpose(const float* d_x, int ldi, float* d_y, int ldo, int rows, int cols, void* stream)
{
    if (!d_x || !d_y) return FRCNN_EINVAL;
    return launch_transpose(d_x, ldi, d_y, ldo, rows, cols, as_stream(stream));
}

int frcnn_sgd_step(float* d_w, const float* d_grad, float* d_momentum_buf, size_t n, float lr, float momentum,
                   float weight_decay, int first_step, void* stream)
{
    if (n > 0 && (!d_w || !d_grad)) return FRCNN_EINVAL;
    return launch_sgd(d_w, d_grad, d_momentum_buf, n, lr, momentum, weight_decay, first_step, as_stream(stream));
}

int frcnn_sgd_step_fold(float* d_w, const float* d_grad, float* d_momentum_buf, size_t n, float lr, float momentum,
                        float weight_decay, int first_step, const float* d_scale, float* d_folded, int cout, int cin, void* stream)
{
    if (n > 0 && (!d_w || !d_grad || !d_scale || !d_folded)) return FRCNN_EINVAL;
    return launch_sgd_fold(d_w, d_grad, d_momentum_buf, n, lr, momentum, weight_decay, first_step, d_scale, d_folded, cout, cin,
                           as_stream(stream));
}

// ---- context ---------------------------------------------------------------------------------
int frcnn_ctx_create(frcnn_ctx** out, int max_image_h, int max_image_w, int max_rois)
{
    if (!out || max_image_h < 16 || max_image_w < 16 || max_rois < 1 || max_rois > 512) return FRCNN_EINVAL;
    frcnn_ctx* c = new (std::nothrow) frcnn_ctx();
    if (!c) return FRCNN_ENOMEM;
    c->max_h = max_image_h; c->max_w = max_image_w; c->max_rois = max_rois;
    c->max_fh = cdiv(max_image_h, 16); c->max_fw = cdiv(max_image_w, 16);   // ceil covers ResNet maps too
    c->a_cap = c->max_fh * c->max_fw * 9;
    if (c->a_cap < 16384) c->a_cap = 16384;      // the stand-alone NMS entry borrows this scratch
    c->pre_cap = 16384;

    const size_t act = (size_t)max_image_h * max_image_w * 64 * sizeof(float);
    const size_t fmb = (size_t)c->max_fh * c->max_fw * 1024 * sizeof(float);   // 1024 channels for the ResNet maps
    const size_t headb = (size_t)c->max_fh * c->max_fw * 128 * sizeof(float);
    size_t lin = 0;
    {
        const size_t w1 = linear_workspace_bytes(max_rois, 4096, 512 * 49);
        const size_t w2 = linear_workspace_bytes(max_rois, 4096, 4096);
        const size_t w3 = linear_workspace_bytes(max_rois, FRCNN_HEAD_LD_MAX, 4096);
        const size_t w4 = linear_workspace_bytes(c->max_fh * c->max_fw, 45, 512);
        const size_t w5 = linear_workspace_bytes(c->max_fh * c->max_fw, 45, 1024);
        const size_t w6 = linear_workspace_bytes(max_rois, FRCNN_HEAD_LD_MAX, 2048);
        lin = w1; if (w2 > lin) lin = w2; if (w3 > lin) lin = w3; if (w4 > lin) lin = w4; if (w5 > lin) lin = w5; if (w6 > lin) lin = w6;
        const size_t t1 = gemm_x6t_workspace_bytes(max_rois, 4096, 512 * 49, 1), t2 = gemm_x6t_workspace_bytes(max_rois, 4096, 4096, 1);
        if (t1 > lin) lin = t1;
        if (t2 > lin) lin = t2;
        const size_t h1 = gemm_x3t_workspace_bytes(max_rois, 4096, 512 * 49, 1), h2 = gemm_x3t_workspace_bytes(max_rois, 4096, 4096, 1);
        if (h1 > lin) lin = h1;
        if (h2 > lin) lin = h2;
    }
    // row count of the activation record arrays: the x6t GEMM's 320-row tiles over max_rois
    const int rec_rows = cdiv(max_rois, gemm_x6t_row_tile(max_rois)) * gemm_x6t_row_tile(max_rois);
    c->rec_rows = rec_rows;
    size_t cws = 0;
    {
        // the layers that may split: every VGG-16 shape at the largest image this ctx accepts
        const int dims[5][2] = {{max_image_h, max_image_w}, {max_image_h / 2, max_image_w / 2},
                                {max_image_h / 4, max_image_w / 4}, {max_image_h / 8, max_image_w / 8},
                                {max_image_h / 16, max_image_w / 16}};
        const int chans[5][2] = {{64, 64}, {128, 128}, {256, 256}, {512, 512}, {512, 512}};
        for (int i = 0; i < 5; ++i) {
            size_t b = conv3x3_workspace_bytes(dims[i][0], dims[i][1], chans[i][0], chans[i][1]);
            if (i > 0) {
                const size_t b2 = conv3x3_workspace_bytes(dims[i][0], dims[i][1], chans[i - 1][1], chans[i][1]);
                if (b2 > b) b = b2;
            }
            if (b > cws) cws = b;
        }
        // smaller images of the same ctx can pick a larger split factor: size for the worst case
        const size_t generous = (size_t)16 * c->max_fh * c->max_fw * 512 * sizeof(float);
        if (generous > cws) cws = generous;
    }
    {
        // ResNet: stem output ((H+1)/2 x (W+1)/2 x 64) == layer1 output size; per-RoI head tensors
        const size_t stem = (size_t)((max_image_h + 1) / 2) * ((max_image_w + 1) / 2) * 64;
        const size_t head = (size_t)max_rois * 49 * 512;
        const size_t head2 = (size_t)max_rois * 16 * 2048;
        size_t m = stem; if (head > m) m = head; if (head2 > m) m = head2;
        c->res_buf_floats = m;
        if (cws < ((size_t)160 << 20)) cws = (size_t)160 << 20;
    }
    struct Item { void** p; size_t bytes; };
    void* ps_base = nullptr;
    Item items[] = {
        {(void**)&c->act_a, act}, {(void**)&c->act_b, act / 2},
        {(void**)&c->fm, fmb}, {(void**)&c->rpn_trunk, fmb}, {(void**)&c->rpn_head, headb},
        {(void**)&c->scores, (size_t)c->a_cap * 4}, {(void**)&c->sorted_idx, (size_t)c->pre_cap * 4},
        {(void**)&c->anchor_map, (size_t)c->a_cap * 16}, {(void**)&c->valid_map, (size_t)c->a_cap * 4},
        {(void**)&c->roi_out, (size_t)max_rois * 49 * 1024 * 4},
        {(void**)&c->fc1_out, (size_t)max_rois * 4096 * 4}, {(void**)&c->fc2_out, (size_t)max_rois * 4096 * 4},
        {(void**)&c->head_logits, (size_t)max_rois * FRCNN_HEAD_LD_MAX * 4},
        {(void**)&c->lin_ws, lin}, {(void**)&c->conv_ws, cws},
        {(void**)&c->res_buf[0], c->res_buf_floats * 4}, {(void**)&c->res_buf[1], c->res_buf_floats * 4},
        {(void**)&c->res_buf[2], c->res_buf_floats * 4}, {(void**)&c->res_buf[3], c->res_buf_floats * 4},
        {(void**)&c->res_buf[4], c->res_buf_floats * 4},
        {&ps_base, proposal_scratch_bytes(c->a_cap, c->pre_cap, 2048)},
    };
    size_t total = 0;
    for (auto& it : items) total += align_up(it.bytes, 256);
    hipError_t e = hipMalloc(&c->slab, total);
    if (e != hipSuccess) { set_hip_error(e); delete c; return FRCNN_ENOMEM; }
    c->slab_bytes = total;
    unsigned char* p = static_cast<unsigned char*>(c->slab);
    for (auto& it : items) { *it.p = p; p += align_up(it.bytes, 256); }
    c->lin_ws_bytes = lin;
    c->conv_ws_bytes = cws;
    proposal_scratch_carve(c->ps, ps_base, c->a_cap, c->pre_cap, 2048);
    *out = c;
    return FRCNN_OK;
}

int frcnn_ctx_create_proposals(frcnn_ctx** out, int max_image_h, int max_image_w)
{
    // the ~50 MB proposal scratch only (keys, decoded boxes, NMS bit matrix): what frcnn_rpn_proposals / frcnn_nms use.  The fused
    // forwards refuse such a ctx (max_rois == 0 < post_nms).
    if (!out || max_image_h < 16 || max_image_w < 16) return FRCNN_EINVAL;
    frcnn_ctx* c = new (std::nothrow) frcnn_ctx();
    if (!c) return FRCNN_ENOMEM;
    c->max_h = max_image_h; c->max_w = max_image_w; c->max_rois = 0;
    c->max_fh = cdiv(max_image_h, 16); c->max_fw = cdiv(max_image_w, 16);
    c->a_cap = c->max_fh * c->max_fw * 9;
    if (c->a_cap < 16384) c->a_cap = 16384;
    c->pre_cap = 16384;
    const size_t total = align_up(proposal_scratch_bytes(c->a_cap, c->pre_cap, 2048), 256);
    hipError_t e = hipMalloc(&c->slab, total);
    if (e != hipSuccess) { set_hip_error(e); delete c; return FRCNN_ENOMEM; }
    c->slab_bytes = total;
    proposal_scratch_carve(c->ps, c->slab, c->a_cap, c->pre_cap, 2048);
    *out = c;
    return FRCNN_OK;
}

void frcnn_ctx_destroy(frcnn_ctx* ctx)
{
    if (!ctx) return;
    for (auto& r : ctx->recs) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    for (auto e : ctx->free_events) (void)hipEventDestroy(e);
    if (ctx->slab) (void)hipFree(ctx->slab);
    if (ctx->wino_ws) (void)hipFree(ctx->wino_ws);
    if (ctx->wx_ws) (void)hipFree(ctx->wx_ws);
    if (ctx->x3f_cmax) (void)hipFree(ctx->x3f_cmax);
    if (ctx->x3p_spill) (void)hipFree(ctx->x3p_spill);
    if (ctx->roi_rec) (void)hipFree(ctx->roi_rec);
    if (ctx->rx_rec) (void)hipFree(ctx->rx_rec);
    if (ctx->rx_ws) (void)hipFree(ctx->rx_ws);
    if (ctx->rx_aux) (void)hipFree(ctx->rx_aux);
    if (ctx->gx_max) (void)hipFree(ctx->gx_max);
    if (ctx->gx_cnt) (void)hipFree(ctx->gx_cnt);
    delete ctx;
}

size_t frcnn_ctx_bytes(const frcnn_ctx* ctx) { return ctx ? ctx->slab_bytes + ctx->wino_ws_bytes + ctx->wx_ws_bytes + ctx->rx_rec_bytes + ctx->rx_ws_bytes + (ctx->roi_rec ? (size_t)ctx->rec_rows * (49 * 512 + 4096) * 6 : 0) : 0; }

int frcnn_ctx_timing_enable(frcnn_ctx* ctx, int enable)
{
    if (!ctx) return FRCNN_EINVAL;
    ctx->timing = enable != 0;
    return FRCNN_OK;
}

int frcnn_ctx_timing_read(frcnn_ctx* ctx, double ms[FRCNN_NUM_KCLASS], int64_t launches[FRCNN_NUM_KCLASS], int reset)
{
    if (!ctx || !ms || !launches) return FRCNN_EINVAL;
    for (auto& r : ctx->recs) {
        FRCNN_HIP_TRY(hipEventSynchronize(r.stop));
        float t = 0.f;
        FRCNN_HIP_TRY(hipEventElapsedTime(&t, r.start, r.stop));
        ctx->t_ms[r.cls] += t;
        ctx->t_cnt[r.cls] += 1;
        ctx->free_events.push_back(r.start);
        ctx->free_events.push_back(r.stop);
    }
    ctx->recs.clear();
    for (int i = 0; i < FRCNN_NUM_KCLASS; ++i) { ms[i] = ctx->t_ms[i]; launches[i] = ctx->t_cnt[i]; }
    if (reset) for (int i = 0; i < FRCNN_NUM_KCLASS; ++i) { ctx->t_ms[i] = 0; ctx->t_cnt[i] = 0; }
    return FRCNN_OK;
}

int frcnn_ctx_tensor(frcnn_ctx* c, int which, void** d_ptr, size_t* bytes)
{
    if (!c || !d_ptr || !bytes) return FRCNN_EINVAL;
    const size_t fmsz = (size_t)c->last_fh * c->last_fw;
    switch (which) {
        case 0: *d_ptr = c->fm; *bytes = fmsz * c->last_c * 4; break;
        case 1: *d_ptr = c->rpn_head; *bytes = fmsz * 128 * 4; break;
        case 2: *d_ptr = c->scores; *bytes = fmsz * 9 * 4; break;
        case 3: *d_ptr = c->sorted_idx; *bytes = (size_t)c->last_pre * 4; break;
        case 4: *d_ptr = c->roi_out; *bytes = (size_t)c->last_post * 49 * c->last_c * 4; break;
        case 5: *d_ptr = c->fc2_out; *bytes = (size_t)c->last_post * c->last_vec * 4; break;
        case 6: *d_ptr = c->anchor_map; *bytes = fmsz * 9 * 16; break;
        case 7: *d_ptr = c->valid_map; *bytes = fmsz * 9 * 4; break;
        case 8: *d_ptr = c->head_logits; *bytes = (size_t)c->last_post * c->last_head_ld * 4; break;
        default: return FRCNN_EINVAL;
    }
    return FRCNN_OK;
}

namespace {
// Winograd scratch (V and M of the largest eligible layer up to the ctx's largest image), allocated once, on first use:
// contexts that never run the mode do not pay the 16*T*(cin+cout) floats.
int ensure_wino_ws(frcnn_ctx* c)
{
    if (c->wino_ws) return FRCNN_OK;
    size_t need = 0;
    // VGG-16 layer shapes (image / 4, / 8, / 16)
    const int shapes[5][3] = {{4, 128, 256}, {4, 256, 256}, {8, 256, 512}, {8, 512, 512}, {16, 512, 512}};
    for (auto& sh : shapes) {
        const size_t b = c->max_h > 0 ? conv3x3_winograd_workspace_bytes(1, c->max_h / sh[0], c->max_w / sh[0], sh[1], sh[2]) : 0;
        if (b > need) need = b;
    }
    // ResNet: the RPN trunk on the 1024-channel map, layer4's 512-wide 3x3 on max_rois 4x4 maps (a head ctx: max_head_rois)
    const int rois = c->max_rois > c->max_head_rois ? c->max_rois : c->max_head_rois;
    const size_t b1 = c->max_fh > 0 ? conv3x3_winograd_workspace_bytes(1, c->max_fh, c->max_fw, 1024, 1024) : 0;
    const size_t b2 = rois > 0 ? conv3x3_winograd_workspace_bytes(rois, 4, 4, 512, 512) : 0;
    if (b1 > need) need = b1;
    if (b2 > need) need = b2;
    if (need == 0) return FRCNN_EINVAL;
    hipError_t e = hipMalloc(&c->wino_ws, need);
    if (e != hipSuccess) { set_hip_error(e); c->wino_ws = nullptr; return FRCNN_ENOMEM; }
    c->wino_ws_bytes = need;
    return FRCNN_OK;
}

// One Winograd layer inside a fused forward, its three launches timed as classes 6 (transforms) and 7 (GEMM).
int run_winograd_layer(frcnn_ctx* c, const float* x, const float* u, const float* b, float* y, int N, int h, int w, int ci, int co,
                       unsigned flags, hipStream_t s)
{
    float *V = nullptr, *M = nullptr;
    int r = winograd_plan(N, h, w, ci, co, flags, c->wino_ws, c->wino_ws_bytes, &V, &M);
    if (r) return r;
    { Scope _t(c, 6, s); r = launch_winograd_input(x, V, N, h, w, ci, s); }
    if (r) return r;
    { Scope _g(c, 7, s); r = launch_winograd_gemm(V, u, M, N, h, w, ci, co, s); }
    if (r) return r;
    Scope _o(c, 6, s);
    return launch_winograd_output(M, b, y, N, h, w, co, flags, s);
}

// Scratch of the x6 Winograd layers (V records + M + split-K partials).  Sized on first use for the largest layer of the DEFAULT table
// up to the ctx's largest image (conv4_x on image / 8, conv5_x / RPN trunk on image / 16, ResNet's 1024-channel trunk); a layer that
// needs more (a larger map put on the table by the caller) grows it once (hipMalloc synchronises: first image only).  Zeroed, so the
// padding row blocks of V are zero for ever: nobody writes them.
int ensure_wx_ws(frcnn_ctx* c, size_t layer_need, hipStream_t s)
{
    if (c->wx_ws && layer_need <= c->wx_ws_bytes) return FRCNN_OK;
    size_t need = layer_need;
    const int shapes[4][4] = {{c->max_h / 8, c->max_w / 8, 256, 512}, {c->max_h / 8, c->max_w / 8, 512, 512},
                              {c->max_fh, c->max_fw, 512, 512}, {c->max_fh, c->max_fw, 1024, 1024}};
    for (auto& sh : shapes) {
        const size_t b = conv3x3_winograd_x6_workspace_bytes(1, sh[0], sh[1], sh[2], sh[3]);
        if (b > need) need = b;
    }
    if (need == 0) return FRCNN_EINVAL;
    if (c->wx_ws) {
        FRCNN_HIP_TRY(hipStreamSynchronize(s));
        (void)hipFree(c->wx_ws);
        c->wx_ws = nullptr; c->wx_ws_bytes = 0;
    }
    hipError_t e = hipMalloc(&c->wx_ws, need);
    if (e != hipSuccess) { set_hip_error(e); c->wx_ws = nullptr; return FRCNN_ENOMEM; }
    // zeroed ON THE CALLER'S STREAM: the slots of in-flight images run on non-blocking streams, which a null-stream hipMemset is not
    // ordered with (it could still be clearing the buffer while the first layer writes its records)
    e = hipMemsetAsync(c->wx_ws, 0, need, s);
    if (e != hipSuccess) { set_hip_error(e); (void)hipFree(c->wx_ws); c->wx_ws = nullptr; return FRCNN_EHIP; }
    c->wx_ws_bytes = need;
    return FRCNN_OK;
}

// The activation record arrays of fc1 / fc2 (x6t / x3t modes of the VGG-16 detector): 53 MB at 320 rows, allocated and zeroed by the first
// forward that needs them (ResNet models and fc_math_mode f32 never do).  The padding rows max_rois .. rec_rows - 1 stay zero.
int ensure_fc_records(frcnn_ctx* c, hipStream_t s)
{
    if (c->roi_rec) return FRCNN_OK;
    // (sized for the x6t records, 6 bytes per value; the x3t records of FRCNN_FC_F32X3T need 4)
    const size_t b1 = (size_t)c->rec_rows * 49 * 512 * 6, b2 = (size_t)c->rec_rows * 4096 * 6;
    const size_t b3 = align_up((size_t)c->rec_rows * sizeof(float), 256), b4 = align_up((size_t)c->max_fh * c->max_fw * sizeof(float), 256);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, b1 + b2 + 2 * b3 + b4);
    if (e != hipSuccess) { set_hip_error(e); return FRCNN_ENOMEM; }
    e = hipMemsetAsync(p, 0, b1 + b2 + 2 * b3 + b4, s);     // on the caller's stream (see ensure_wx_ws)
    if (e != hipSuccess) { set_hip_error(e); (void)hipFree(p); return FRCNN_EHIP; }
    c->roi_rec = p;
    c->fc1_rec = static_cast<unsigned char*>(p) + b1;
    c->roi_inv = reinterpret_cast<float*>(static_cast<unsigned char*>(p) + b1 + b2);
    c->fc1_inv = reinterpret_cast<float*>(static_cast<unsigned char*>(p) + b1 + b2 + b3);
    c->fm_cmax = reinterpret_cast<float*>(static_cast<unsigned char*>(p) + b1 + b2 + 2 * b3);
    return FRCNN_OK;
}

// One x6 Winograd layer inside a fused forward: transforms timed as class 8, the batched bf16-pipe GEMM as class 9.
int run_wino_x6_layer(frcnn_ctx* c, const float* x, const void* urec, const float* b, float* y, int h, int w, int ci, int co,
                      unsigned flags, hipStream_t s, int N = 1)
{
    if (N == 1 && !conv3x3_uses_winograd_x6(ci, co)) return FRCNN_EINVAL;
    int r = ensure_wx_ws(c, conv3x3_winograd_x6_workspace_bytes(N, h, w, ci, co), s);
    if (r) return r;
    void *V = nullptr, *G = nullptr;
    float* M = nullptr;
    size_t gb = 0;
    r = winograd_x6_plan(N, h, w, ci, co, flags, c->wx_ws, c->wx_ws_bytes, &V, &M, &G, &gb);
    if (r) return r;
    { Scope _t(c, 8, s); r = launch_winograd_x6_input(x, V, N, h, w, ci, s); }
    if (r) return r;
    { Scope _g(c, 9, s); r = launch_winograd_x6_gemm(V, urec, M, N, h, w, ci, co, G, gb, s); }
    if (r) return r;
    Scope _o(c, 8, s);
    return launch_winograd_output(M, b, y, N, h, w, co, flags, s);
}

// The same layer in the f32x3 arithmetic (csrc/wino_x3.hip; a bit of frcnn_forward_params.winograd_x3_mask): ublob = the packed
// x3t filter bank (records + row scales).  Same timing classes.
// cmax_ready: the channel maxima of x, left behind by the producing layer's epilogue (skips the pass over x); cmax_out: where this layer's
// output transform accumulates the channel maxima of y (zeroed by the caller; needs winograd_output_emits_cmax(co, flags)).
int run_wino_x3_layer(frcnn_ctx* c, const float* x, const void* ublob, const float* b, float* y, int h, int w, int ci, int co,
                      unsigned flags, hipStream_t s, int N = 1, const float* cmax_ready = nullptr, float* cmax_out = nullptr)
{
    if (N == 1 && !conv3x3_uses_winograd_x6(ci, co)) return FRCNN_EINVAL;
    int r = ensure_wx_ws(c, conv3x3_winograd_x3_workspace_bytes(N, h, w, ci, co), s);
    if (r) return r;
    void *V = nullptr, *G = nullptr;
    float *M = nullptr, *vinv = nullptr, *cmax = nullptr;
    size_t gb = 0;
    r = winograd_x3_plan(N, h, w, ci, co, flags, c->wx_ws, c->wx_ws_bytes, &V, &vinv, &cmax, &M, &G, &gb);
    if (r) return r;
    { Scope _t(c, 8, s); r = launch_winograd_x3_input(x, cmax, V, vinv, N, h, w, ci, s, cmax_ready); }
    if (r) return r;
    { Scope _g(c, 9, s); r = launch_winograd_x3_gemm(V, vinv, ublob, M, N, h, w, ci, co, G, gb, s); }
    if (r) return r;
    Scope _o(c, 8, s);
    return launch_winograd_output(M, b, y, N, h, w, co, flags, s, cmax_out);
}

// The three channel-maximum buffers of a ctx (max_h x max_w floats each): [0] scratch of a layer that has to compute its input's maxima
// itself, [1] / [2] the ping-pong pair the f32x3 layers hand the maxima of their OUTPUT to the next layer in.
int ensure_x3f_cmax(frcnn_ctx* c, size_t need, hipStream_t s)
{
    if (c->x3f_cmax && c->x3f_cmax_bytes >= need) return FRCNN_OK;
    if (c->x3f_cmax) { FRCNN_HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(c->x3f_cmax); c->x3f_cmax = nullptr; c->x3f_cmax_bytes = 0; }
    const size_t cap = std::max(need, (size_t)c->max_h * c->max_w * sizeof(float));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->x3f_cmax), 3 * cap);
    if (e != hipSuccess) { set_hip_error(e); c->x3f_cmax = nullptr; return FRCNN_ENOMEM; }
    c->x3f_cmax_bytes = cap;
    return FRCNN_OK;
}
float* x3f_cmax_buffer(frcnn_ctx* c, int which) { return reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(c->x3f_cmax) + (size_t)which * c->x3f_cmax_bytes); }

// A ONE-LAUNCH f32x3 Winograd layer inside a fused forward (csrc/wino_x3f.hip; a bit of frcnn_forward_params.winograd_x3f_mask; timing class 10,
// which includes the channel-maximum pass over the layer input).  ublob = frcnn_pack_conv3x3_winograd_x3's blob.
int run_wino_x3f_layer(frcnn_ctx* c, const float* x, const void* ublob, const float* b, float* y, int h, int w, int ci, int co,
                       unsigned flags, hipStream_t s, const float* cmax_ready = nullptr, float* cmax_out = nullptr, bool pair = false)
{
    int r = ensure_x3f_cmax(c, conv3x3_winograd_x3_fused_workspace_bytes(1, h, w), s);
    if (r) return r;
    if (pair) {                                                   // the two-pass form (csrc/wino_x3p.hip): the ctx owns the spill scratch
        const size_t need = conv3x3_winograd_x3_pair_spill_bytes(1, h, w, co);
        if (!need) return FRCNN_EUNSUPPORTED;
        if (!c->x3p_spill || c->x3p_spill_bytes < need) {
            if (c->x3p_spill) { FRCNN_HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(c->x3p_spill); c->x3p_spill = nullptr; c->x3p_spill_bytes = 0; }
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->x3p_spill), need);
            if (e != hipSuccess) { set_hip_error(e); c->x3p_spill = nullptr; return FRCNN_ENOMEM; }
            c->x3p_spill_bytes = need;
        }
        flags |= FRCNN_X3F_PAIR;
    }
    Scope _w(c, 10, s);
    return launch_conv3x3_winograd_x3_fused(x, ublob, b, y, 1, h, w, ci, co, flags, x3f_cmax_buffer(c, 0), c->x3f_cmax_bytes, s, cmax_ready, cmax_out,
                                            pair ? c->x3p_spill : nullptr, pair ? c->x3p_spill_bytes : 0);
}

// One one-launch Winograd layer inside a fused forward (timed as class 7).
// (A channel split of the small maps over more blocks -- partial outputs, arrival tickets, the last arriver sums in part order --
//  was built and measured for one image on the chip: conv5_x 87 -> 84-87 us, conv4_2 227 -> 222 us with 2 parts, slower with 4.
//  The per-block prologue / epilogue and the partial traffic cost what the shorter serial chain wins, so it is not in the tree.)
int run_wino_fused_layer(frcnn_ctx* c, bool /*latency*/, const float* x, const float* u, const float* b, float* y, int h, int w, int ci,
                         int co, unsigned flags, hipStream_t s, int n_maps = 1)
{
    Scope _w(c, 7, s);
    return launch_conv3x3_winograd_fused(x, u, b, y, h, w, ci, co, flags, s, n_maps);
}

struct BlocksTargetScope {
    explicit BlocksTargetScope(int t, int wino_rows = 0, int x6_tiles = 0) { conv3x3_set_blocks_target(t); linear_batched_set_tile(wino_rows); gemm_x6t_set_tiles(x6_tiles); }
    ~BlocksTargetScope() { conv3x3_set_blocks_target(0); linear_batched_set_tile(0); gemm_x6t_set_tiles(0); }
};
}  // namespace

// ---- fused forward ---------------------------------------------------------------------------
int frcnn_vgg16_forward(frcnn_ctx* c, const frcnn_vgg16_weights* w, const frcnn_forward_params* p,
                        const float* d_image, int H, int W, const float* d_anchor_map,
                        const float* d_valid_map, float* d_props, float* d_classes, float* d_deltas,
                        int32_t* d_counts, void* stream)
{
    if (!c || !w || !p || !d_image || !d_props || !d_classes || !d_deltas || !d_counts) return FRCNN_EINVAL;
    if (H < 16 || W < 16 || H > c->max_h || W > c->max_w) return FRCNN_EINVAL;
    if (p->post_nms < 1 || p->post_nms > c->max_rois || p->pre_nms < 1 || p->pre_nms > c->pre_cap) return FRCNN_EINVAL;
    if (w->num_classes < 2 || w->num_classes > FRCNN_MAX_NUM_CLASSES) return FRCNN_EUNSUPPORTED;   // ncls + 4 (ncls - 1) <= FRCNN_HEAD_LD_MAX
    for (int i = 0; i < 13; ++i) if (!w->conv_w[i] || !w->conv_b[i]) return FRCNN_EINVAL;
    if (!w->rpn_conv_w || !w->rpn_conv_b || !w->rpn_head_w || !w->rpn_head_b || !w->fc1_w || !w->fc1_b ||
        !w->fc2_w || !w->fc2_b || !w->head_w || !w->head_b)
        return FRCNN_EINVAL;
    hipStream_t s = as_stream(stream);
    const unsigned R = FRCNN_RELU, RP = FRCNN_RELU | FRCNN_POOL2;
    int rc;
    if (p->math_mode != FRCNN_MATH_F32 && p->math_mode != FRCNN_MATH_F32_WINOGRAD) return FRCNN_EINVAL;
    if (p->conv_blocks_target < 0) return FRCNN_EINVAL;
    if (p->fc_math_mode != FRCNN_FC_F32 && p->fc_math_mode != FRCNN_FC_F32X6T && p->fc_math_mode != FRCNN_FC_F32X3T) return FRCNN_EINVAL;
    if ((p->roi_op != FRCNN_ROI_POOL && p->roi_op != FRCNN_ROI_ALIGN) || p->roi_sampling_ratio > 2) return FRCNN_EINVAL;
    if (p->winograd_tile_rows != 0 && p->winograd_tile_rows != 64 && p->winograd_tile_rows != 128) return FRCNN_EINVAL;
    if (p->x6_gemm_tiles < 0 || p->x6_gemm_tiles > 2) return FRCNN_EINVAL;
    if ((p->winograd_x3_mask & ~p->winograd_x6_mask) != 0) return FRCNN_EINVAL;     // a subset of the x6 table
    BlocksTargetScope target_scope(p->conv_blocks_target, p->winograd_tile_rows, p->x6_gemm_tiles);
    const bool wino = p->math_mode == FRCNN_MATH_F32_WINOGRAD;
    if (p->winograd_x6_mask != 0 && (!wino || (p->winograd_x6_mask & ~0x3FFE) != 0)) return FRCNN_EINVAL;
    if (p->winograd_x3f_mask != 0 && (!wino || (p->winograd_x3f_mask & ~0x3FFE) != 0 || (p->winograd_x3f_mask & p->winograd_x6_mask) != 0)) return FRCNN_EINVAL;
    if ((p->winograd_x3p_mask & ~p->winograd_x3f_mask) != 0 || (p->winograd_x3p_mask & 0x2) != 0) return FRCNN_EINVAL;    // a subset of the one-launch table; conv1_2 has 64 output channels
    int layer_index = 0;        // 1 .. 12 = conv_w[i], 13 = the RPN trunk (frcnn_forward_params.winograd_x6_mask)
    const float* cmax_ready = nullptr;     // channel maxima of the activation tensor produced last, if its producer emitted them
    // The producers accumulate those maxima with atomic maxima into ZEROED floats.  Every emitting layer of the image gets its own region of
    // one arena (buffers [1], [2] of the ctx: 2 max_h max_w floats; the emitted maps are those of conv2_1 and later: < H W / 2 floats in
    // all), and the arena is cleared by ONE fill at the top of the forward instead of one per layer (ten 5 us launches of every image).
    size_t cmax_used = 0, cmax_cleared = 0;
    if (wino && ((p->winograd_x3f_mask | (p->winograd_x3_mask & p->winograd_x6_mask)) & 0x1FFE) != 0) {
        int r0 = ensure_x3f_cmax(c, (size_t)H * W * sizeof(float), s);
        if (r0) return r0;
        // The arena's size follows from the masks and (H, W) -- every emitting layer's output map, rounded up to 64 floats -- and is
        // validated HERE, before anything is launched (the budget used to be two constants fitted to the default tables, with a
        // per-layer fill and an EINVAL inside the layer loop for everything else).  Layers 1 .. 12 = conv1_2 .. conv5_3 (models/vgg16.py:27-47).
        static const int co_of[13] = {0, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
        static const bool pool_of[13] = {false, true, false, true, false, false, true, false, false, true, false, false, false};
        size_t want = 0;
        int eh = H, ew = W;
        for (int li = 1; li <= 12; ++li) {
            const bool f3 = ((p->winograd_x3f_mask >> li) & 1) != 0;
            const bool t3 = !f3 && ((p->winograd_x6_mask >> li) & 1) && ((p->winograd_x3_mask >> li) & 1);
            const unsigned efl = FRCNN_RELU | (pool_of[li] ? FRCNN_POOL2 : 0u);
            if (pool_of[li]) { eh /= 2; ew /= 2; }
            if ((f3 || t3) && (f3 || winograd_output_emits_cmax(co_of[li], efl))) want += ((size_t)eh * ew + 63) & ~(size_t)63;
        }
        if (want > 2 * c->x3f_cmax_bytes / sizeof(float)) return FRCNN_EINVAL;      // (< 0.75 H W floats for any table: cannot happen)
        cmax_cleared = want;
        if (want) FRCNN_HIP_TRY(hipMemsetAsync(x3f_cmax_buffer(c, 1), 0, cmax_cleared * sizeof(float), s));
    }
    auto conv3 = [&](const float* xin, const float* wgt, const float* bs, float* yout, int hh, int ww, int ci, int co,
                     unsigned fl) -> int {
        ++layer_index;
        // The f32x3 layers take their scales from the per-pixel channel maximum of their INPUT.  A layer whose producer was an f32x3 layer finds
        // those maxima ready (`cmax_ready`: the producer's epilogue accumulated them with atomic maxima) instead of reading the tensor once more
        // (pixel_absmax_kernel: 11 of the 12 passes of an image); the same values, so the same bits.  The RPN trunk's output feeds no such layer.
        const float* in_cmax = cmax_ready;
        cmax_ready = nullptr;
        const bool is_x3f = wino && ((p->winograd_x3f_mask >> layer_index) & 1);
        const bool is_x3 = wino && !is_x3f && ((p->winograd_x6_mask >> layer_index) & 1) && ((p->winograd_x3_mask >> layer_index) & 1);
        float* out_cmax = nullptr;
        if ((is_x3f || is_x3) && layer_index != FRCNN_X6_RPN_TRUNK_BIT && (fl & FRCNN_RELU) && (is_x3f || winograd_output_emits_cmax(co, fl))) {
            const int oh = (fl & FRCNN_POOL2) ? hh / 2 : hh, ow = (fl & FRCNN_POOL2) ? ww / 2 : ww;
            const size_t need = ((size_t)oh * ow + 63) & ~(size_t)63;
            out_cmax = x3f_cmax_buffer(c, 1) + cmax_used;
            cmax_used += need;
            if (cmax_used > cmax_cleared) return FRCNN_EINVAL;      // (the plan above and this walk disagree: a bug, not an input)
        }
        int r1;
        if (is_x3f)                                                  // one-launch f32x3 Winograd layer: wgt = the x3 blob (csrc/wino_x3f.hip)
            r1 = run_wino_x3f_layer(c, xin, wgt, bs, yout, hh, ww, ci, co, fl, s, in_cmax, out_cmax, ((p->winograd_x3p_mask >> layer_index) & 1) != 0);
        else if (is_x3)                                              // three-launch f32x3 layer: wgt = the x3 blob (csrc/wino_x3.hip)
            r1 = run_wino_x3_layer(c, xin, wgt, bs, yout, hh, ww, ci, co, fl, s, 1, in_cmax, out_cmax);
        else if (wino && ((p->winograd_x6_mask >> layer_index) & 1)) // x6 Winograd layer: wgt = the record bank (csrc/wino_x6.hip)
            r1 = run_wino_x6_layer(c, xin, wgt, bs, yout, hh, ww, ci, co, fl, s);
        else
            r1 = -9999;
        if (r1 != -9999) {
            if (r1 == FRCNN_OK) cmax_ready = out_cmax;
            return r1;
        }
        if (wino && conv3x3_uses_winograd_fused(ci, co)) {     // one launch, no scratch (csrc/winofused.hip); timed as class 7
            return run_wino_fused_layer(c, p->conv_blocks_target == 0, xin, wgt, bs, yout, hh, ww, ci, co, fl, s);
        }
        Scope _d(c, 0, s);
        return launch_conv3x3_nhwc(xin, wgt, bs, yout, hh, ww, ci, co, fl, c->conv_ws, c->conv_ws_bytes, s);
    };
#define STEP(cls, call) do { Scope _sc(c, cls, s); rc = (call); } while (0); if (rc) return rc
#define CONV(call) do { rc = (call); } while (0); if (rc) return rc

    // stage 1: feature extractor (models/vgg16.py:76-96)
    float *A = c->act_a, *B = c->act_b;
    int h = H, wd = W;
    // conv1_1 leaves the channel maxima of its output for conv1_2 when that is a one-launch f32x3 layer (a plain store per pixel from the
    // lanes that hold its 64 channels, into the ctx's scratch buffer [0]: no pass over the 153.6 MB tensor anywhere in the image)
    float* c11_cmax = (wino && ((p->winograd_x3f_mask >> 1) & 1)) ? x3f_cmax_buffer(c, 0) : nullptr;
    STEP(1, launch_conv3x3_c3(d_image, w->conv_w[0], w->conv_b[0], A, h, wd, 64, R, s, c11_cmax));
    cmax_ready = c11_cmax;
    CONV(conv3(A, w->conv_w[1], w->conv_b[1], B, h, wd, 64, 64, RP));   h /= 2; wd /= 2;
    CONV(conv3(B, w->conv_w[2], w->conv_b[2], A, h, wd, 64, 128, R));
    CONV(conv3(A, w->conv_w[3], w->conv_b[3], B, h, wd, 128, 128, RP)); h /= 2; wd /= 2;
    CONV(conv3(B, w->conv_w[4], w->conv_b[4], A, h, wd, 128, 256, R));
    CONV(conv3(A, w->conv_w[5], w->conv_b[5], B, h, wd, 256, 256, R));
    CONV(conv3(B, w->conv_w[6], w->conv_b[6], A, h, wd, 256, 256, RP)); h /= 2; wd /= 2;
    CONV(conv3(A, w->conv_w[7], w->conv_b[7], B, h, wd, 256, 512, R));
    CONV(conv3(B, w->conv_w[8], w->conv_b[8], A, h, wd, 512, 512, R));
    CONV(conv3(A, w->conv_w[9], w->conv_b[9], B, h, wd, 512, 512, RP)); h /= 2; wd /= 2;
    CONV(conv3(B, w->conv_w[10], w->conv_b[10], A, h, wd, 512, 512, R));
    CONV(conv3(A, w->conv_w[11], w->conv_b[11], B, h, wd, 512, 512, R));
    CONV(conv3(B, w->conv_w[12], w->conv_b[12], c->fm, h, wd, 512, 512, R));
    const float* fm_cmax_ready = cmax_ready;            // the feature map's channel maxima, if conv5_3 left them (the RoI pooling's scale source)
    const int fh = h, fw = wd;
    c->last_fh = fh; c->last_fw = fw; c->last_pre = p->pre_nms; c->last_post = p->post_nms; c->last_c = 512; c->last_vec = 4096;

    // stage 2: RPN (models/rpn.py:88-153)
    CONV(conv3(c->fm, w->rpn_conv_w, w->rpn_conv_b, c->rpn_trunk, fh, fw, 512, 512, R));
    STEP(2, launch_linear(c->rpn_trunk, 512, w->rpn_head_w, w->rpn_head_b, c->rpn_head, 128, fh * fw, 45, 512,
                          0u, c->lin_ws, c->lin_ws_bytes, s));
    const float* amap = d_anchor_map;
    const float* vmap = d_valid_map;
    if (!amap || !vmap) {
        if (c->anc_h != H || c->anc_w != W || c->anc_fh != fh || c->anc_fw != fw) {
            STEP(5, launch_anchors(H, W, fh, fw, 16, c->anchor_map, c->valid_map, s));
            c->anc_h = H; c->anc_w = W; c->anc_fh = fh; c->anc_fw = fw;
        }
        amap = c->anchor_map; vmap = c->valid_map;
    }
    FRCNN_HIP_TRY(hipMemsetAsync(d_counts, 0, 4 * sizeof(int32_t), s));
    STEP(3, launch_rpn_proposals(c->ps, c->rpn_head, 128, amap, p->allow_edge_proposals ? nullptr : vmap, fh, fw,
                                 H, W, p->pre_nms, p->post_nms, p->rpn_nms_threshold, p->min_side, c->scores,
                                 c->sorted_idx, d_props, d_counts, s));

    // stage 3: detector (models/detector.py:65-80, models/vgg16.py:129-133)
    const int R_ = p->post_nms;
    const bool fc_x6t = p->fc_math_mode == FRCNN_FC_F32X6T;
    const bool fc_x3t = p->fc_math_mode == FRCNN_FC_F32X3T;
    if (fc_x6t || fc_x3t) { rc = ensure_fc_records(c, s); if (rc) return rc; }
    if (fc_x3t) {
        // fc1 / fc2 in the f32x3 arithmetic (csrc/gemm_x3t.hip): fc1_w / fc2_w = packed x3t operands (records of the 4096-row matrices, then
        // their row scales); RoIPool writes fc1's A records with one scale per RoI; fc1's float32 output is scaled and split again for fc2
        const int rr = c->rec_rows;
        const size_t w1rec = x3t_record_bytes(4096, 49 * 512), w2rec = x3t_record_bytes(4096, 4096);
        const float* w1inv = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(w->fc1_w) + w1rec);
        const float* w2inv = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(w->fc2_w) + w2rec);
        if (p->roi_op == FRCNN_ROI_ALIGN) {
            STEP(4, launch_roi_align(c->fm, fh, fw, 512, d_props, d_counts + 2, R_, 7, 1.0f / 16.0f, p->roi_sampling_ratio, 0, c->roi_out, s));
            STEP(2, launch_rows_scale_x3t(c->roi_out, 49 * 512, 0, c->roi_inv, R_, rr, 49 * 512, 1, s));
            STEP(2, launch_split_rows_x3t(c->roi_out, 49 * 512, 0, c->roi_inv, c->roi_rec, R_, rr, 49 * 512, 1, s));
        } else {
            STEP(4, launch_roi_pool_x3t(c->fm, fh, fw, 512, d_props, d_counts + 2, R_, 7, 1.0f / 16.0f,
                                        fm_cmax_ready ? const_cast<float*>(fm_cmax_ready) : c->fm_cmax, c->roi_inv, c->roi_rec, rr, s, fm_cmax_ready != nullptr));
        }
        // (tile mode 0 = the cost model's choice in every slot: the in-flight tile override of the Winograd layers would change fc's split-K
        //  factor, and an image must give the same bits in flight and alone)
        STEP(2, launch_gemm_x3t(c->roi_rec, c->roi_inv, rr, 0, 0, w->fc1_w, w1inv, 4096, 0, 0, w->fc1_b, nullptr, c->fc1_out, 4096, 0, R_, 4096,
                                49 * 512, 1, R, c->lin_ws, c->lin_ws_bytes, s, 0));
        STEP(2, launch_rows_scale_x3t(c->fc1_out, 4096, 0, c->fc1_inv, R_, rr, 4096, 1, s));
        STEP(2, launch_split_rows_x3t(c->fc1_out, 4096, 0, c->fc1_inv, c->fc1_rec, R_, rr, 4096, 1, s));
        // fc2 (K = 4096) on the 160 x 128 tiles: 64 tiles x 4 reduction ranges of 64 chunks instead of 16 tiles x 16 ranges of 16 -- a
        // quarter of the partial planes (19.7 MB instead of 78.6 MB written and read again) and blocks that are not mostly prologue and
        // epilogue: 42.7 us against 53.0 with its reduction (tools/exp_fc_cfg.sh; the cost model scores the two within 3 %).  The same
        // in every slot (an image gives the same bits in flight and alone).
        STEP(2, launch_gemm_x3t(c->fc1_rec, c->fc1_inv, rr, 0, 0, w->fc2_w, w2inv, 4096, 0, 0, w->fc2_b, nullptr, c->fc2_out, 4096, 0, R_, 4096,
                                4096, 1, R, c->lin_ws, c->lin_ws_bytes, s, 2));
    } else
    if (fc_x6t) {
        // fc1 / fc2 as f32x6 GEMMs on tile records (csrc/gemm_x6t.hip): RoIPool writes fc1's operand records itself (the rows R_ ..
        // rec_rows - 1 were zeroed when the ctx was created); fc1's float32 output is split again for fc2 (5 MB: a 3 us launch)
        const int rr = c->rec_rows;
        if (p->roi_op == FRCNN_ROI_ALIGN) {
            STEP(4, launch_roi_align(c->fm, fh, fw, 512, d_props, d_counts + 2, R_, 7, 1.0f / 16.0f, p->roi_sampling_ratio, 0, c->roi_out, s));
            STEP(2, launch_split_rows_x6t(c->roi_out, 49 * 512, 0, c->roi_rec, R_, rr, 49 * 512, 1, s));
        } else {
            STEP(4, launch_roi_pool_x6t(c->fm, fh, fw, 512, d_props, d_counts + 2, R_, 7, 1.0f / 16.0f, c->roi_rec, rr, s));
        }
        static const int fc_tiles = []() { const char* e = frcnn_knob("FRCNN_FC_TILES"); return e ? atoi(e) : 0; }();     // experiments
        STEP(2, launch_gemm_x6t(c->roi_rec, rr, 0, w->fc1_w, 4096, 0, w->fc1_b, nullptr, c->fc1_out, 4096, 0, R_, 4096, 49 * 512, 1, R,
                                c->lin_ws, c->lin_ws_bytes, s, fc_tiles));
        STEP(2, launch_split_rows_x6t(c->fc1_out, 4096, 0, c->fc1_rec, R_, rr, 4096, 1, s));
        STEP(2, launch_gemm_x6t(c->fc1_rec, rr, 0, w->fc2_w, 4096, 0, w->fc2_b, nullptr, c->fc2_out, 4096, 0, R_, 4096, 4096, 1, R,
                                c->lin_ws, c->lin_ws_bytes, s, fc_tiles));
    } else
    {
        if (p->roi_op == FRCNN_ROI_ALIGN) {
            STEP(4, launch_roi_align(c->fm, fh, fw, 512, d_props, d_counts + 2, R_, 7, 1.0f / 16.0f, p->roi_sampling_ratio, 0, c->roi_out, s));
        } else {
            STEP(4, launch_roi_pool(c->fm, fh, fw, 512, d_props, d_counts + 2, R_, 7, 1.0f / 16.0f, c->roi_out, s));
        }
        STEP(2, launch_linear(c->roi_out, 49 * 512, w->fc1_w, w->fc1_b, c->fc1_out, 4096, R_, 4096, 49 * 512, R,
                              c->lin_ws, c->lin_ws_bytes, s));
        STEP(2, launch_linear(c->fc1_out, 4096, w->fc2_w, w->fc2_b, c->fc2_out, 4096, R_, 4096, 4096, R,
                              c->lin_ws, c->lin_ws_bytes, s));
    }
    const int ncls = w->num_classes, nd = (ncls - 1) * 4;
    const int hld = cdiv(ncls + nd, 128) * 128;                  // row count of the stacked classifier + regressor operand (zero padded)
    c->last_head_ld = hld;
    STEP(2, launch_linear(c->fc2_out, 4096, w->head_w, w->head_b, c->head_logits, hld, R_, ncls + nd, 4096, 0u,
                          c->lin_ws, c->lin_ws_bytes, s));
    STEP(5, launch_head_finish(c->head_logits, hld, R_, ncls, nd, d_classes, d_deltas, s));
#undef STEP
#undef CONV
    return FRCNN_OK;
}

// ---- fused ResNet forward ----------------------------------------------------------------------
namespace {
// One 1x1 convolution (stride 1 or 2) of a bottleneck as an f32x6 GEMM: y[N Ho Wo][cout] = act(bias + residual + x_pixels . w^T).
// The activation records live in the ctx's rx_rec scratch (grown on demand: hipMalloc synchronises, first image of a shape only).
// (ksize 3: the 3x3 / padding-1 form over im2col records, K = 9 cin -- layer4.0.conv2's stride-2 convolution)
int run_conv1x1_x6(frcnn_ctx* c, const float* x, const void* wrec, const float* bias, const float* residual, float* y, int N, int h,
                   int w, int cin, int cout, int stride, unsigned flags, int cls, hipStream_t s, int ksize = 1)
{
    if (cin % 16 != 0 || cout % 4 != 0 || (ksize != 1 && ksize != 3)) return FRCNN_EINVAL;
    const int ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
    const int K = ksize == 3 ? 9 * cin : cin;
    const long long rows = (long long)N * ho * wo;
    if (rows > 0x7fffffffLL / 8) return FRCNN_EINVAL;
    const int M = (int)rows;
    const int Mp = cdiv(M, gemm_x6t_row_tile(M)) * gemm_x6t_row_tile(M), Np = cdiv(cout, gemm_x6t_col_tile(cout)) * gemm_x6t_col_tile(cout);
    const size_t need = x6t_record_bytes(Mp, K), gneed = gemm_x6t_workspace_bytes(M, cout, K, 1);
    if (need > c->rx_rec_bytes) {
        if (c->rx_rec) { FRCNN_HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(c->rx_rec); c->rx_rec = nullptr; c->rx_rec_bytes = 0; }
        hipError_t e = hipMalloc(&c->rx_rec, need);
        if (e != hipSuccess) { set_hip_error(e); c->rx_rec = nullptr; return FRCNN_ENOMEM; }
        c->rx_rec_bytes = need;
    }
    if (gneed > c->rx_ws_bytes) {
        if (c->rx_ws) { FRCNN_HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(c->rx_ws); c->rx_ws = nullptr; c->rx_ws_bytes = 0; }
        hipError_t e = hipMalloc(&c->rx_ws, gneed);
        if (e != hipSuccess) { set_hip_error(e); c->rx_ws = nullptr; return FRCNN_ENOMEM; }
        c->rx_ws_bytes = gneed;
    }
    int rc;
    { Scope _t(c, 8, s); rc = ksize == 3 ? launch_split_patches3x3_x6t(x, c->rx_rec, N, h, w, cin, stride, Mp, s)
                                        : launch_split_pixels_x6t(x, c->rx_rec, N, h, w, cin, stride, Mp, s); }
    if (rc) return rc;
    Scope _g(c, cls, s);
    return launch_gemm_x6t(c->rx_rec, Mp, 0, wrec, Np, 0, bias, residual, y, cout, 0, M, cout, K, 1, flags, c->rx_ws, c->rx_ws_bytes, s);
}

// The same convolution in the f32x3 arithmetic (csrc/gemm_x3t.hip): wblob = frcnn_pack_rows_x3t of the [cout][K] matrix (records, then
// the 2^-e of its rows); the activation rows are scaled per output pixel from a channel-maximum pass over the input.
int run_conv1x1_x3(frcnn_ctx* c, const float* x, const void* wblob, const float* bias, const float* residual, float* y, int N, int h,
                   int w, int cin, int cout, int stride, unsigned flags, int cls, hipStream_t s, int ksize = 1)
{
    if (cin % 16 != 0 || cout % 4 != 0 || (ksize != 1 && ksize != 3)) return FRCNN_EINVAL;
    const int ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
    const int K = ksize == 3 ? 9 * cin : cin;
    const long long rows = (long long)N * ho * wo;
    if (rows > 0x7fffffffLL / 8) return FRCNN_EINVAL;
    const int M = (int)rows;
    const int Mp = cdiv(M, gemm_x6t_row_tile(M)) * gemm_x6t_row_tile(M), Np = cdiv(cout, gemm_x6t_col_tile(cout)) * gemm_x6t_col_tile(cout);
    const size_t need = x3t_record_bytes(Mp, K), gneed = gemm_x3t_workspace_bytes(M, cout, K, 1);
    const size_t aneed = (size_t)Mp + (size_t)N * h * w;
    if (need > c->rx_rec_bytes) {
        if (c->rx_rec) { FRCNN_HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(c->rx_rec); c->rx_rec = nullptr; c->rx_rec_bytes = 0; }
        hipError_t e = hipMalloc(&c->rx_rec, need);
        if (e != hipSuccess) { set_hip_error(e); c->rx_rec = nullptr; return FRCNN_ENOMEM; }
        c->rx_rec_bytes = need;
    }
    if (gneed > c->rx_ws_bytes) {
        if (c->rx_ws) { FRCNN_HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(c->rx_ws); c->rx_ws = nullptr; c->rx_ws_bytes = 0; }
        hipError_t e = hipMalloc(&c->rx_ws, gneed);
        if (e != hipSuccess) { set_hip_error(e); c->rx_ws = nullptr; return FRCNN_ENOMEM; }
        c->rx_ws_bytes = gneed;
    }
    if (aneed > c->rx_aux_floats) {
        if (c->rx_aux) { FRCNN_HIP_TRY(hipStreamSynchronize(s)); (void)hipFree(c->rx_aux); c->rx_aux = nullptr; c->rx_aux_floats = 0; }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->rx_aux), aneed * sizeof(float));
        if (e != hipSuccess) { set_hip_error(e); c->rx_aux = nullptr; return FRCNN_ENOMEM; }
        c->rx_aux_floats = aneed;
    }
    float* ainv = c->rx_aux;
    float* cmax = c->rx_aux + Mp;
    int rc;
    {
        Scope _t(c, 8, s);
        // (1x1: one launch that reduces the row maxima itself; the 3x3 patches take their scale from nine pixels' maxima: two launches)
        static const bool two_pass = frcnn_knob("FRCNN_SPLIT_TWO_PASS") != nullptr;      // experiments / the bit-identity test
        rc = (ksize == 3 || two_pass) ? launch_pixel_absmax(x, cmax, (long long)N * h * w, cin, s) : FRCNN_OK;
        if (!rc) rc = ksize == 3 ? launch_split_patches3x3_x3t(x, cmax, c->rx_rec, ainv, N, h, w, cin, stride, Mp, s)
                                 : launch_split_pixels_x3t(x, two_pass ? cmax : nullptr, c->rx_rec, ainv, N, h, w, cin, stride, Mp, s);
    }
    if (rc) return rc;
    const float* winv = reinterpret_cast<const float*>(static_cast<const unsigned char*>(wblob) + x3t_record_bytes(Np, K));
    Scope _g(c, cls, s);
    return launch_gemm_x3t(c->rx_rec, ainv, Mp, 0, 0, wblob, winv, Np, 0, 0, bias, residual, y, cout, 0, M, cout, K, 1, flags, c->rx_ws,
                           c->rx_ws_bytes, s);
}

// the next zeroed maximum slot of this context's current pass (GX_SLOTS floats, zeroed by gx_begin)
static constexpr int GX_SLOTS = 512;
int gx_begin(frcnn_ctx* c, hipStream_t s)
{
    if (!c->gx_max) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->gx_max), GX_SLOTS * sizeof(float));
        if (e != hipSuccess) { set_hip_error(e); c->gx_max = nullptr; return FRCNN_ENOMEM; }
    }
    if (!c->gx_cnt) {                                              // zeroed ONCE: the last block of a tile leaves its counter at zero again
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->gx_cnt), GX_TILE_COUNTERS * sizeof(unsigned));
        if (e != hipSuccess) { set_hip_error(e); c->gx_cnt = nullptr; return FRCNN_ENOMEM; }
        FRCNN_HIP_TRY(hipMemsetAsync(c->gx_cnt, 0, GX_TILE_COUNTERS * sizeof(unsigned), s));
    }
    FRCNN_HIP_TRY(hipMemsetAsync(c->gx_max, 0, GX_SLOTS * sizeof(float), s));
    c->gx_next = 0;
    c->gx_x = nullptr;
    return FRCNN_OK;
}
int gx_slot(frcnn_ctx* c, hipStream_t s, float** out)
{
    if (!c->gx_max || c->gx_next >= GX_SLOTS) return FRCNN_EINVAL;
    (void)s;
    *out = c->gx_max + c->gx_next++;
    return FRCNN_OK;
}

// One Bottleneck (torchvision v1.5): out = relu(conv3(relu(conv2(relu(conv1(x))))) + identity).
// x: [N][h][w][cin] in `cur`; returns the buffer index holding the output, updates h, w.
// `backbone`: the block belongs to the feature extractor (layer1..3): its 3x3 is packed for the one-launch Winograd kernel whatever the
// number of images N in the launch (the per-RoI blocks of layer4 are packed for the three-launch form).
int run_bottleneck(frcnn_ctx* c, const frcnn_bottleneck_weights& b, int N, int& h, int& w, int cur,
                   int* out_idx, hipStream_t s, int cls_conv, bool wino, bool latency, bool backbone)
{
    const int pack_maps = backbone ? 1 : N;
    // pick three scratch buffers different from `cur`
    int f[4], k = 0;
    for (int i = 0; i < 5 && k < 4; ++i) if (i != cur) f[k++] = i;
    float* X = c->res_buf[cur];
    float* T1 = c->res_buf[f[0]];
    float* T2 = c->res_buf[f[1]];
    float* ID = c->res_buf[f[2]];
    float* OUT = c->res_buf[f[3]];
    const unsigned R = FRCNN_RELU;
    int rc;
    const int ho = (h + 2 - 3) / b.stride + 1, wo = (w + 2 - 3) / b.stride + 1;
    const size_t need1 = (size_t)N * h * w * b.width, need2 = (size_t)N * ho * wo * b.cout;
    if (need1 > c->res_buf_floats || need2 > c->res_buf_floats) return FRCNN_EINVAL;
#define RSTEP(call) do { Scope _sc(c, cls_conv, s); rc = (call); } while (0); if (rc) return rc
    if (b.g3 != 0) {
        // (a split-operand block belongs to the f32_winograd table: math mode "f32" is the strict exact-f32 mode -- ADVICE r4)
        if (b.x6_mask != 0 || !b.wmax || !wino) return FRCNN_EINVAL;
        float *m1, *m2, *m3;
        if (!c->gx_x) {                                          // the block input came from a kernel that leaves no maximum behind
            float* mx;
            if ((rc = gx_slot(c, s, &mx))) return rc;
            RSTEP(launch_tensor_absmax(X, (long long)N * h * w * b.cin, mx, s));
            c->gx_x = mx;
        }
        if ((rc = gx_slot(c, s, &m1)) || (rc = gx_slot(c, s, &m2)) || (rc = gx_slot(c, s, &m3))) return rc;
        const bool ws_ = b.g3 == 2;                              // the packs are pre-split images (frcnn_pack_conv_x3g_weights)
        const GatherX3 g1{c->gx_x, b.wmax + 0, m1, c->gx_cnt, ws_, ws_}, g2{m1, b.wmax + 1, m2, c->gx_cnt, ws_, ws_}, g3{m2, b.wmax + 2, m3, c->gx_cnt, ws_, ws_}, gd{c->gx_x, b.wmax + 3, nullptr, c->gx_cnt, ws_, ws_};      // (trusted: every maximum of this chain is a producer's or tensor_absmax_kernel's)
        const int X3 = FRCNN_CONV_F32X3G;
        RSTEP(launch_conv_gather(X, b.w1, b.b1, nullptr, T1, N, h, w, b.cin, b.width, 1, 1, 0, R, c->conv_ws, c->conv_ws_bytes, s, X3, &g1));
        RSTEP(launch_conv_gather(T1, b.w2, b.b2, nullptr, T2, N, h, w, b.width, b.width, 3, b.stride, 1, R, c->conv_ws, c->conv_ws_bytes, s, X3, &g2));
        const float* identity = X;
        if (b.wd) {
            RSTEP(launch_conv_gather(X, b.wd, b.bd, nullptr, ID, N, h, w, b.cin, b.cout, 1, b.stride, 0, 0u, c->conv_ws, c->conv_ws_bytes, s, X3, &gd));
            identity = ID;
        } else if (b.cin != b.cout || b.stride != 1) {
            return FRCNN_EINVAL;
        }
        RSTEP(launch_conv_gather(T2, b.w3, b.b3, identity, OUT, N, ho, wo, b.width, b.cout, 1, 1, 0, R, c->conv_ws, c->conv_ws_bytes, s, X3, &g3));
        c->gx_x = m3;
        h = ho; w = wo;
        *out_idx = f[3];
        return FRCNN_OK;
    }
    c->gx_x = nullptr;
    if (b.x6_mask != 0 && !wino) return FRCNN_EINVAL;
    if ((b.x3_mask & ~b.x6_mask) != 0) return FRCNN_EINVAL;          // x3_mask: the subset of the split-operand convolutions in f32x3
    if (b.x6_mask & FRCNN_X6_CONV1) {
        rc = (b.x3_mask & FRCNN_X6_CONV1) ? run_conv1x1_x3(c, X, b.w1, b.b1, nullptr, T1, N, h, w, b.cin, b.width, 1, R, 9, s)
                                          : run_conv1x1_x6(c, X, b.w1, b.b1, nullptr, T1, N, h, w, b.cin, b.width, 1, R, 9, s);
        if (rc) return rc;
    } else {
        RSTEP(launch_conv_gather(X, b.w1, b.b1, nullptr, T1, N, h, w, b.cin, b.width, 1, 1, 0, R, c->conv_ws, c->conv_ws_bytes, s));
    }
    if (b.x6_mask & FRCNN_X6_CONV2) {
        // the 3x3 on the bf16 pipe: stride 1 = an x6 Winograd layer over the block's N maps, stride 2 = an im2col GEMM (K = 9 width)
        if (b.x3_mask & FRCNN_X6_CONV2)
            rc = b.stride == 1 ? run_wino_x3_layer(c, T1, b.w2, b.b2, T2, h, w, b.width, b.width, R, s, N)
                               : run_conv1x1_x3(c, T1, b.w2, b.b2, nullptr, T2, N, h, w, b.width, b.width, b.stride, R, 9, s, 3);
        else
            rc = b.stride == 1 ? run_wino_x6_layer(c, T1, b.w2, b.b2, T2, h, w, b.width, b.width, R, s, N)
                               : run_conv1x1_x6(c, T1, b.w2, b.b2, nullptr, T2, N, h, w, b.width, b.width, b.stride, R, 9, s, 3);
        if (rc) return rc;
    } else if (wino && resnet_block_uses_winograd_fused(pack_maps, b.width, b.stride)) {
        rc = run_wino_fused_layer(c, latency, T1, b.w2, b.b2, T2, h, w, b.width, b.width, R, s, N);
        if (rc) return rc;
    } else if (wino && resnet_block_uses_winograd(b.width, b.stride)) {
        rc = run_winograd_layer(c, T1, b.w2, b.b2, T2, N, h, w, b.width, b.width, R, s);
        if (rc) return rc;
    } else if (b.stride == 1 && pack_maps == 1 && b.width % 64 == 0) {
        // direct exact-f32 3x3 (math mode "f32"): one launch per image of the batch
        for (int i = 0; i < N; ++i) {
            RSTEP(launch_conv3x3_nhwc(T1 + (size_t)i * h * w * b.width, b.w2, b.b2, T2 + (size_t)i * h * w * b.width, h, w, b.width, b.width, R,
                                      c->conv_ws, c->conv_ws_bytes, s));
        }
    } else {
        RSTEP(launch_conv_gather(T1, b.w2, b.b2, nullptr, T2, N, h, w, b.width, b.width, 3, b.stride, 1, R,
                                 c->conv_ws, c->conv_ws_bytes, s));
    }
    const float* identity = X;
    if (b.wd) {
        if (b.x6_mask & FRCNN_X6_DOWN) {
            rc = (b.x3_mask & FRCNN_X6_DOWN) ? run_conv1x1_x3(c, X, b.wd, b.bd, nullptr, ID, N, h, w, b.cin, b.cout, b.stride, 0u, 9, s)
                                             : run_conv1x1_x6(c, X, b.wd, b.bd, nullptr, ID, N, h, w, b.cin, b.cout, b.stride, 0u, 9, s);
            if (rc) return rc;
        } else {
            RSTEP(launch_conv_gather(X, b.wd, b.bd, nullptr, ID, N, h, w, b.cin, b.cout, 1, b.stride, 0, 0u,
                                     c->conv_ws, c->conv_ws_bytes, s));
        }
        identity = ID;
    } else if (b.cin != b.cout || b.stride != 1) {
        return FRCNN_EINVAL;
    }
    if (b.x6_mask & FRCNN_X6_CONV3) {
        rc = (b.x3_mask & FRCNN_X6_CONV3) ? run_conv1x1_x3(c, T2, b.w3, b.b3, identity, OUT, N, ho, wo, b.width, b.cout, 1, R, 9, s)
                                          : run_conv1x1_x6(c, T2, b.w3, b.b3, identity, OUT, N, ho, wo, b.width, b.cout, 1, R, 9, s);
        if (rc) return rc;
    } else {
        RSTEP(launch_conv_gather(T2, b.w3, b.b3, identity, OUT, N, ho, wo, b.width, b.cout, 1, 1, 0, R,
                                 c->conv_ws, c->conv_ws_bytes, s));
    }
#undef RSTEP
    h = ho; w = wo;
    *out_idx = f[3];
    return FRCNN_OK;
}
}  // namespace

namespace {
int resnet_check_weights(const frcnn_resnet_weights* w, bool with_heads)
{
    int nb = 0;
    for (int i = 0; i < 4; ++i) { if (w->n_blocks[i] < 1) return FRCNN_EINVAL; nb += w->n_blocks[i]; }
    if (nb > FRCNN_RESNET_MAX_BLOCKS) return FRCNN_EINVAL;
    if (!w->stem_w || !w->stem_b) return FRCNN_EINVAL;
    if (with_heads && (!w->rpn_conv_w || !w->rpn_conv_b || !w->rpn_head_w || !w->rpn_head_b || !w->head_w || !w->head_b)) return FRCNN_EINVAL;
    for (int i = 0; i < nb; ++i) {
        const frcnn_bottleneck_weights& b = w->blocks[i];
        if (!b.w1 || !b.b1 || !b.w2 || !b.b2 || !b.w3 || !b.b3 || (b.wd && !b.bd)) return FRCNN_EINVAL;
        if (b.cin % 16 || b.width % 16 || b.cout % 64 || (b.stride != 1 && b.stride != 2)) return FRCNN_EINVAL;
    }
    return FRCNN_OK;
}

int resnet_check_params(const frcnn_forward_params* p)
{
    if ((p->winograd_x3_mask & ~p->winograd_x6_mask) != 0) return FRCNN_EINVAL;    // (only the RPN trunk bit exists for ResNet)
    if (p->math_mode != FRCNN_MATH_F32 && p->math_mode != FRCNN_MATH_F32_WINOGRAD) return FRCNN_EUNSUPPORTED;   // no f32x6 ResNet path
    if ((p->roi_op != FRCNN_ROI_POOL && p->roi_op != FRCNN_ROI_ALIGN) || p->roi_sampling_ratio > 2) return FRCNN_EINVAL;
    if (p->conv_blocks_target < 0) return FRCNN_EINVAL;
    if (p->winograd_tile_rows != 0 && p->winograd_tile_rows != 64 && p->winograd_tile_rows != 128) return FRCNN_EINVAL;
    if (p->x6_gemm_tiles < 0 || p->x6_gemm_tiles > 2) return FRCNN_EINVAL;
    return FRCNN_OK;
}

// stage 1: conv1 / bn1 / relu / maxpool / layer1..3 (models/resnet.py:38-46) over n images [n][3][H][W] in ONE pass: every bottleneck
// launch covers the n maps (the 1x1 convolutions are GEMMs over n * h * w pixels).  Leaves the [n][fh][fw][C] maps in res_buf[*cur].
int resnet_stage1(frcnn_ctx* c, const frcnn_resnet_weights* w, const frcnn_forward_params* p, const float* d_images, int n, int H, int W,
                  int* cur_out, int* fh_out, int* fw_out, int* c_out, hipStream_t s)
{
    const bool wino = p->math_mode == FRCNN_MATH_F32_WINOGRAD;
    int rc;
    int h = (H + 6 - 7) / 2 + 1, wd = (W + 6 - 7) / 2 + 1;
    if ((size_t)n * h * wd * 64 > c->res_buf_floats) return FRCNN_EINVAL;
    const int hp = (h + 2 - 3) / 2 + 1, wp = (wd + 2 - 3) / 2 + 1;
    for (int i = 0; i < n; ++i) {
        { Scope _sc(c, 1, s); rc = launch_conv7x7_s2_c3(d_images + (size_t)i * 3 * H * W, w->stem_w, w->stem_b, c->res_buf[0] + (size_t)i * h * wd * 64,
                                                       H, W, 64, FRCNN_RELU, s); }
        if (rc) return rc;
        { Scope _sc(c, 5, s); rc = launch_maxpool3x3_s2(c->res_buf[0] + (size_t)i * h * wd * 64, c->res_buf[1] + (size_t)i * hp * wp * 64, h, wd, 64, s); }
        if (rc) return rc;
    }
    h = hp; wd = wp;
    int cur = 1, bi = 0;
    {
        bool any_g3 = false;
        for (int i = 0; i < w->n_blocks[0] + w->n_blocks[1] + w->n_blocks[2]; ++i) any_g3 = any_g3 || w->blocks[i].g3 != 0;
        if (any_g3 && (rc = gx_begin(c, s))) return rc;
    }
    for (int layer = 0; layer < 3; ++layer)
        for (int k = 0; k < w->n_blocks[layer]; ++k, ++bi) {
            int out = -1;
            rc = run_bottleneck(c, w->blocks[bi], n, h, wd, cur, &out, s, 0, wino, p->conv_blocks_target == 0, true);
            if (rc) return rc;
            cur = out;
        }
    *cur_out = cur; *fh_out = h; *fw_out = wd; *c_out = w->blocks[bi - 1].cout;     // 1024
    return FRCNN_OK;
}

// stages 2 and 3 on the feature map in c->fm: RPN (models/rpn.py:88-153), RoI pooling, layer4 per RoI, spatial mean, heads
// (models/detector.py:65-80, resnet.py:109-118)
// ... in two halves (round 6): RPN + proposals + RoI pooling of ONE image into `roi_dst` ([post_nms][7][7][C]), and layer4 + spatial mean + heads over
// `n_rois` pooled RoIs -- one image's (resnet_tail) or a whole batch's (frcnn_resnet_head: the per-RoI GEMMs of 8 images as ONE launch each)
int resnet_rpn_roi(frcnn_ctx* c, const frcnn_resnet_weights* w, const frcnn_forward_params* p, int H, int W, int fh, int fw, int C,
                   const float* d_anchor_map, const float* d_valid_map, float* d_props, int32_t* d_counts, float* roi_dst, hipStream_t s)
{
    const bool wino = p->math_mode == FRCNN_MATH_F32_WINOGRAD;
    int rc;
#define STEP(cls, call) do { Scope _sc(c, cls, s); rc = (call); } while (0); if (rc) return rc
    c->last_fh = fh; c->last_fw = fw; c->last_pre = p->pre_nms; c->last_post = p->post_nms; c->last_c = C;
    if (p->winograd_x6_mask != 0 && (!wino || p->winograd_x6_mask != (1 << FRCNN_X6_RPN_TRUNK_BIT))) return FRCNN_EINVAL;
    if (wino && p->winograd_x6_mask) {
        rc = p->winograd_x3_mask ? run_wino_x3_layer(c, c->fm, w->rpn_conv_w, w->rpn_conv_b, c->rpn_trunk, fh, fw, C, C, FRCNN_RELU, s)
                                 : run_wino_x6_layer(c, c->fm, w->rpn_conv_w, w->rpn_conv_b, c->rpn_trunk, fh, fw, C, C, FRCNN_RELU, s);
        if (rc) return rc;
    } else if (wino && conv3x3_uses_winograd_fused(C, C)) {
        rc = run_wino_fused_layer(c, p->conv_blocks_target == 0, c->fm, w->rpn_conv_w, w->rpn_conv_b, c->rpn_trunk, fh, fw, C, C, FRCNN_RELU, s);
        if (rc) return rc;
    } else {
        STEP(0, launch_conv3x3_nhwc(c->fm, w->rpn_conv_w, w->rpn_conv_b, c->rpn_trunk, fh, fw, C, C, FRCNN_RELU,
                                    c->conv_ws, c->conv_ws_bytes, s));
    }
    STEP(2, launch_linear(c->rpn_trunk, C, w->rpn_head_w, w->rpn_head_b, c->rpn_head, 128, fh * fw, 45, C, 0u,
                          c->lin_ws, c->lin_ws_bytes, s));
    const float* amap = d_anchor_map;
    const float* vmap = d_valid_map;
    if (!amap || !vmap) {
        if (c->anc_h != H || c->anc_w != W || c->anc_fh != fh || c->anc_fw != fw) {
            STEP(5, launch_anchors(H, W, fh, fw, 16, c->anchor_map, c->valid_map, s));
            c->anc_h = H; c->anc_w = W; c->anc_fh = fh; c->anc_fw = fw;
        }
        amap = c->anchor_map; vmap = c->valid_map;
    }
    FRCNN_HIP_TRY(hipMemsetAsync(d_counts, 0, 4 * sizeof(int32_t), s));
    STEP(3, launch_rpn_proposals(c->ps, c->rpn_head, 128, amap, p->allow_edge_proposals ? nullptr : vmap, fh, fw,
                                 H, W, p->pre_nms, p->post_nms, p->rpn_nms_threshold, p->min_side, c->scores,
                                 c->sorted_idx, d_props, d_counts, s));

    const int R_ = p->post_nms;
    if (p->roi_op == FRCNN_ROI_ALIGN) {
        STEP(4, launch_roi_align(c->fm, fh, fw, C, d_props, d_counts + 2, R_, 7, 1.0f / 16.0f, p->roi_sampling_ratio, 0, roi_dst, s));
    } else {
        STEP(4, launch_roi_pool(c->fm, fh, fw, C, d_props, d_counts + 2, R_, 7, 1.0f / 16.0f, roi_dst, s));
    }
#undef STEP
    return FRCNN_OK;
}

int resnet_head(frcnn_ctx* c, const frcnn_resnet_weights* w, const frcnn_forward_params* p, float* roi_in, int n_rois, float* d_classes,
                float* d_deltas, hipStream_t s)
{
    const bool wino = p->math_mode == FRCNN_MATH_F32_WINOGRAD;
    int rc;
    int bi = w->n_blocks[0] + w->n_blocks[1] + w->n_blocks[2];
    const int R_ = n_rois;
#define STEP(cls, call) do { Scope _sc(c, cls, s); rc = (call); } while (0); if (rc) return rc
    // layer4 reads its input from the pooled RoIs: staged as res_buf "cur" by pointer swap
    float* saved = c->res_buf[0];
    c->res_buf[0] = roi_in;
    int cur = 0, h = 7, wd = 7;
    {
        bool any_g3 = false;
        for (int k = 0; k < w->n_blocks[3]; ++k) any_g3 = any_g3 || w->blocks[bi + k].g3 != 0;
        if (any_g3 && (rc = gx_begin(c, s))) { c->res_buf[0] = saved; return rc; }
    }
    for (int k = 0; k < w->n_blocks[3]; ++k, ++bi) {
        int out = -1;
        rc = run_bottleneck(c, w->blocks[bi], R_, h, wd, cur, &out, s, 2, wino, p->conv_blocks_target == 0, false);
        if (rc) { c->res_buf[0] = saved; return rc; }
        cur = out;
    }
    float* head_in = c->res_buf[cur];
    c->res_buf[0] = saved;
    const int V = w->blocks[bi - 1].cout;                       // 2048
    if (V > 4096) return FRCNN_EINVAL;
    c->last_vec = V;
    STEP(5, launch_spatial_mean(head_in, c->fc2_out, R_, h, wd, V, s));
    const int ncls = w->num_classes, nd = (ncls - 1) * 4;
    const int hld = cdiv(ncls + nd, 128) * 128;
    c->last_head_ld = hld;
    STEP(2, launch_linear(c->fc2_out, V, w->head_w, w->head_b, c->head_logits, hld, R_, ncls + nd, V, 0u,
                          c->lin_ws, c->lin_ws_bytes, s));
    STEP(5, launch_head_finish(c->head_logits, hld, R_, ncls, nd, d_classes, d_deltas, s));
#undef STEP
    return FRCNN_OK;
}

int resnet_tail(frcnn_ctx* c, const frcnn_resnet_weights* w, const frcnn_forward_params* p, int H, int W, int fh, int fw, int C,
                const float* d_anchor_map, const float* d_valid_map, float* d_props, float* d_classes, float* d_deltas, int32_t* d_counts,
                hipStream_t s)
{
    int rc = resnet_rpn_roi(c, w, p, H, W, fh, fw, C, d_anchor_map, d_valid_map, d_props, d_counts, c->roi_out, s);
    if (rc) return rc;
    return resnet_head(c, w, p, c->roi_out, p->post_nms, d_classes, d_deltas, s);
}
}  // namespace

int frcnn_resnet_forward(frcnn_ctx* c, const frcnn_resnet_weights* w, const frcnn_forward_params* p,
                         const float* d_image, int H, int W, const float* d_anchor_map,
                         const float* d_valid_map, float* d_props, float* d_classes, float* d_deltas,
                         int32_t* d_counts, void* stream)
{
    if (!c || !w || !p || !d_image || !d_props || !d_classes || !d_deltas || !d_counts) return FRCNN_EINVAL;
    if (H < 32 || W < 32 || H > c->max_h || W > c->max_w) return FRCNN_EINVAL;
    if (p->post_nms < 1 || p->post_nms > c->max_rois || p->pre_nms < 1 || p->pre_nms > c->pre_cap) return FRCNN_EINVAL;
    if (w->num_classes < 2 || w->num_classes > FRCNN_MAX_NUM_CLASSES) return FRCNN_EUNSUPPORTED;
    int rc = resnet_check_params(p);
    if (rc) return rc;
    rc = resnet_check_weights(w, true);
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    BlocksTargetScope target_scope(p->conv_blocks_target, p->winograd_tile_rows, p->x6_gemm_tiles);
    if (p->math_mode == FRCNN_MATH_F32_WINOGRAD) { rc = ensure_wino_ws(c); if (rc) return rc; }
    int cur = 0, fh = 0, fw = 0, C = 0;
    rc = resnet_stage1(c, w, p, d_image, 1, H, W, &cur, &fh, &fw, &C, s);
    if (rc) return rc;
    if (fh > c->max_fh || fw > c->max_fw || C > 1024 || C % 64) return FRCNN_EINVAL;
    FRCNN_HIP_TRY(hipMemcpyAsync(c->fm, c->res_buf[cur], (size_t)fh * fw * C * sizeof(float), hipMemcpyDeviceToDevice, s));
    return resnet_tail(c, w, p, H, W, fh, fw, C, d_anchor_map, d_valid_map, d_props, d_classes, d_deltas, d_counts, s);
}

// ---- the same forward in two calls: the feature extractor over a BATCH of images, then RPN + detector per image -----------------
int frcnn_ctx_create_backbone(frcnn_ctx** out, int max_image_h, int max_image_w, int max_images)
{
    // scratch of frcnn_resnet_backbone only: the five rotating activation buffers sized for max_images maps and the split-K scratch of
    // the gather kernel; Winograd / x6 scratch is allocated by the first call that needs it.  The fused forwards refuse such a ctx.
    if (!out || max_image_h < 32 || max_image_w < 32 || max_images < 1 || max_images > 64) return FRCNN_EINVAL;
    frcnn_ctx* c = new (std::nothrow) frcnn_ctx();
    if (!c) return FRCNN_ENOMEM;
    c->max_h = max_image_h; c->max_w = max_image_w; c->max_rois = 0; c->max_images = max_images;
    c->max_fh = cdiv(max_image_h, 16); c->max_fw = cdiv(max_image_w, 16);
    const size_t stem = (size_t)((max_image_h + 1) / 2) * ((max_image_w + 1) / 2) * 64;
    c->res_buf_floats = stem * max_images;
    const size_t cws = (size_t)160 << 20;
    const size_t total = 5 * align_up(c->res_buf_floats * 4, 256) + align_up(cws, 256);
    hipError_t e = hipMalloc(&c->slab, total);
    if (e != hipSuccess) { set_hip_error(e); delete c; return FRCNN_ENOMEM; }
    c->slab_bytes = total;
    unsigned char* q = static_cast<unsigned char*>(c->slab);
    for (int i = 0; i < 5; ++i) { c->res_buf[i] = reinterpret_cast<float*>(q); q += align_up(c->res_buf_floats * 4, 256); }
    c->conv_ws = q; c->conv_ws_bytes = cws;
    *out = c;
    return FRCNN_OK;
}

int frcnn_resnet_backbone(frcnn_ctx* c, const frcnn_resnet_weights* w, const frcnn_forward_params* p, const float* d_images,
                          int n_images, int H, int W, float* d_features, void* stream)
{
    if (!c || !w || !p || !d_images || !d_features || n_images < 1) return FRCNN_EINVAL;
    if (H < 32 || W < 32 || H > c->max_h || W > c->max_w || n_images > (c->max_images > 0 ? c->max_images : 1)) return FRCNN_EINVAL;
    int rc = resnet_check_params(p);
    if (rc) return rc;
    rc = resnet_check_weights(w, false);
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    BlocksTargetScope target_scope(p->conv_blocks_target, p->winograd_tile_rows, p->x6_gemm_tiles);
    int cur = 0, fh = 0, fw = 0, C = 0;
    rc = resnet_stage1(c, w, p, d_images, n_images, H, W, &cur, &fh, &fw, &C, s);
    if (rc) return rc;
    FRCNN_HIP_TRY(hipMemcpyAsync(d_features, c->res_buf[cur], (size_t)n_images * fh * fw * C * sizeof(float), hipMemcpyDeviceToDevice, s));
    return FRCNN_OK;
}

int frcnn_resnet_forward_features(frcnn_ctx* c, const frcnn_resnet_weights* w, const frcnn_forward_params* p,
                                  const float* d_feature_map, int H, int W, const float* d_anchor_map,
                                  const float* d_valid_map, float* d_props, float* d_classes, float* d_deltas,
                                  int32_t* d_counts, void* stream)
{
    if (!c || !w || !p || !d_feature_map || !d_props || !d_classes || !d_deltas || !d_counts) return FRCNN_EINVAL;
    if (H < 32 || W < 32 || H > c->max_h || W > c->max_w) return FRCNN_EINVAL;
    if (p->post_nms < 1 || p->post_nms > c->max_rois || p->pre_nms < 1 || p->pre_nms > c->pre_cap) return FRCNN_EINVAL;
    if (w->num_classes < 2 || w->num_classes > FRCNN_MAX_NUM_CLASSES) return FRCNN_EUNSUPPORTED;
    int rc = resnet_check_params(p);
    if (rc) return rc;
    rc = resnet_check_weights(w, true);
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    BlocksTargetScope target_scope(p->conv_blocks_target, p->winograd_tile_rows, p->x6_gemm_tiles);
    if (p->math_mode == FRCNN_MATH_F32_WINOGRAD) { rc = ensure_wino_ws(c); if (rc) return rc; }
    // the map's shape follows from the image's (conv1 7x7/2 pad 3, maxpool 3x3/2 pad 1, layer2 and layer3 stride 2)
    int fh = (H + 6 - 7) / 2 + 1, fw = (W + 6 - 7) / 2 + 1;
    fh = (fh + 2 - 3) / 2 + 1; fw = (fw + 2 - 3) / 2 + 1;
    for (int i = 0; i < 2; ++i) { fh = (fh + 2 - 3) / 2 + 1; fw = (fw + 2 - 3) / 2 + 1; }
    const int nb3 = w->n_blocks[0] + w->n_blocks[1] + w->n_blocks[2];
    const int C = w->blocks[nb3 - 1].cout;
    if (fh > c->max_fh || fw > c->max_fw || C > 1024 || C % 64) return FRCNN_EINVAL;
    if (d_feature_map != c->fm)
        FRCNN_HIP_TRY(hipMemcpyAsync(c->fm, d_feature_map, (size_t)fh * fw * C * sizeof(float), hipMemcpyDeviceToDevice, s));
    return resnet_tail(c, w, p, H, W, fh, fw, C, d_anchor_map, d_valid_map, d_props, d_classes, d_deltas, d_counts, s);
}

// ---- ... and in THREE calls (round 6): the per-RoI head of a whole batch as one set of launches --------------------------------------------------
// frcnn_resnet_rpn_roipool: stages 2 and the RoI pooling of ONE image (its own ctx and stream) into d_roi_out = its slice
// [post_nms][7][7][C] of a batch buffer; frcnn_resnet_head (a ctx from frcnn_ctx_create_head): layer4 + spatial mean + heads
// (models/resnet.py:109-118, detector.py:75-78) over the n_rois pooled RoIs of ALL images -- 14,700 / 4,800-row GEMMs per image become
// 117,600 / 38,400-row GEMMs per batch of 8: whole tiles, no split-K, an eighth of the launches.
int frcnn_resnet_rpn_roipool(frcnn_ctx* c, const frcnn_resnet_weights* w, const frcnn_forward_params* p, const float* d_feature_map, int H,
                             int W, const float* d_anchor_map, const float* d_valid_map, float* d_props, int32_t* d_counts, float* d_roi_out,
                             void* stream)
{
    if (!c || !w || !p || !d_feature_map || !d_props || !d_counts || !d_roi_out) return FRCNN_EINVAL;
    if (H < 32 || W < 32 || H > c->max_h || W > c->max_w) return FRCNN_EINVAL;
    if (p->post_nms < 1 || p->post_nms > c->max_rois || p->pre_nms < 1 || p->pre_nms > c->pre_cap) return FRCNN_EINVAL;
    int rc = resnet_check_params(p);
    if (rc) return rc;
    rc = resnet_check_weights(w, true);
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    BlocksTargetScope target_scope(p->conv_blocks_target, p->winograd_tile_rows, p->x6_gemm_tiles);
    if (p->math_mode == FRCNN_MATH_F32_WINOGRAD) { rc = ensure_wino_ws(c); if (rc) return rc; }
    int fh = (H + 6 - 7) / 2 + 1, fw = (W + 6 - 7) / 2 + 1;
    fh = (fh + 2 - 3) / 2 + 1; fw = (fw + 2 - 3) / 2 + 1;
    for (int i = 0; i < 2; ++i) { fh = (fh + 2 - 3) / 2 + 1; fw = (fw + 2 - 3) / 2 + 1; }
    const int nb3 = w->n_blocks[0] + w->n_blocks[1] + w->n_blocks[2];
    const int C = w->blocks[nb3 - 1].cout;
    if (fh > c->max_fh || fw > c->max_fw || C > 1024 || C % 64) return FRCNN_EINVAL;
    if (d_feature_map != c->fm)
        FRCNN_HIP_TRY(hipMemcpyAsync(c->fm, d_feature_map, (size_t)fh * fw * C * sizeof(float), hipMemcpyDeviceToDevice, s));
    return resnet_rpn_roi(c, w, p, H, W, fh, fw, C, d_anchor_map, d_valid_map, d_props, d_counts, d_roi_out, s);
}

int frcnn_ctx_create_head(frcnn_ctx** out, int max_rois_total)
{
    // scratch of frcnn_resnet_head only: the rotating layer4 buffers for max_rois_total pooled RoIs, the mean / logits rows, the split-K
    // scratch of the gather kernel; record / Winograd scratch of the split-operand head modes is allocated by the first call that needs
    // it.  Every other entry point refuses such a ctx (max_h = max_w = 0).
    if (!out || max_rois_total < 1 || max_rois_total > 512 * 64) return FRCNN_EINVAL;
    frcnn_ctx* c = new (std::nothrow) frcnn_ctx();
    if (!c) return FRCNN_ENOMEM;
    c->max_h = 0; c->max_w = 0; c->max_rois = 0; c->max_images = 0; c->max_head_rois = max_rois_total;
    const size_t n = (size_t)max_rois_total;
    c->res_buf_floats = n * 49 * 512 > n * 16 * 2048 ? n * 49 * 512 : n * 16 * 2048;
    const size_t lin = linear_workspace_bytes(max_rois_total, FRCNN_HEAD_LD_MAX, 2048);
    const size_t cws = (size_t)160 << 20;
    struct Item { void** p; size_t bytes; };
    Item items[] = {
        {(void**)&c->fc2_out, n * 4096 * 4}, {(void**)&c->head_logits, n * FRCNN_HEAD_LD_MAX * 4}, {(void**)&c->lin_ws, lin}, {(void**)&c->conv_ws, cws},
        {(void**)&c->res_buf[0], c->res_buf_floats * 4}, {(void**)&c->res_buf[1], c->res_buf_floats * 4}, {(void**)&c->res_buf[2], c->res_buf_floats * 4},
        {(void**)&c->res_buf[3], c->res_buf_floats * 4}, {(void**)&c->res_buf[4], c->res_buf_floats * 4},
    };
    size_t total = 0;
    for (auto& it : items) total += align_up(it.bytes, 256);
    hipError_t e = hipMalloc(&c->slab, total);
    if (e != hipSuccess) { set_hip_error(e); delete c; return FRCNN_ENOMEM; }
    c->slab_bytes = total;
    unsigned char* q = static_cast<unsigned char*>(c->slab);
    for (auto& it : items) { *it.p = q; q += align_up(it.bytes, 256); }
    c->lin_ws_bytes = lin;
    c->conv_ws_bytes = cws;
    *out = c;
    return FRCNN_OK;
}

int frcnn_resnet_head(frcnn_ctx* c, const frcnn_resnet_weights* w, const frcnn_forward_params* p, float* d_rois, int n_rois, float* d_classes,
                      float* d_deltas, void* stream)
{
    if (!c || !w || !p || !d_rois || !d_classes || !d_deltas) return FRCNN_EINVAL;
    if (n_rois < 1 || n_rois > c->max_head_rois) return FRCNN_EINVAL;
    if (w->num_classes < 2 || w->num_classes > FRCNN_MAX_NUM_CLASSES) return FRCNN_EUNSUPPORTED;
    int rc = resnet_check_params(p);
    if (rc) return rc;
    rc = resnet_check_weights(w, true);
    if (rc) return rc;
    hipStream_t s = as_stream(stream);
    BlocksTargetScope target_scope(p->conv_blocks_target, p->winograd_tile_rows, p->x6_gemm_tiles);
    if (p->math_mode == FRCNN_MATH_F32_WINOGRAD) { rc = ensure_wino_ws(c); if (rc) return rc; }
    return resnet_head(c, w, p, d_rois, n_rois, d_classes, d_deltas, s);
}

}  // extern "C"
