"""
Device-side plumbing shared by the model mirror: frcnn_ctx handles, per-image slots (context +
stream + persistent output buffers + pinned host staging) and packed-weight caching.
PyTorch is used only for memory, streams and events.
"""
import ctypes as C

import numpy as np
import torch as t

from . import _native as nv


class Context:
    """Owns one frcnn_ctx (activation buffers + scratch for one in-flight image)."""
    def __init__(self, device, max_h, max_w, max_rois, proposals_only=False):
        """proposals_only: just the ~50 MB proposal scratch (frcnn_ctx_create_proposals) for the stage-level RPN / NMS entry points."""
        nv.require_gpu()
        self.device = t.device(device)
        self.max_h, self.max_w, self.max_rois = int(max_h), int(max_w), 0 if proposals_only else int(max_rois)
        handle = C.c_void_p()
        with t.cuda.device(self.device):
            if proposals_only:
                nv.check(nv.lib().frcnn_ctx_create_proposals(C.byref(handle), self.max_h, self.max_w), "frcnn_ctx_create_proposals")
            else:
                nv.check(nv.lib().frcnn_ctx_create(C.byref(handle), self.max_h, self.max_w, self.max_rois),
                         "frcnn_ctx_create")
        self.handle = handle
        self.timing = False          # per-kernel event timing on: the model then launches eagerly (events are not captured)

    def fits(self, h, w, rois):
        return h <= self.max_h and w <= self.max_w and rois <= self.max_rois

    @property
    def nbytes(self):
        return int(nv.lib().frcnn_ctx_bytes(self.handle))

    def tensor(self, which, dtype=t.float32):
        """Copies intermediate tensor `which` of the last forward to a new CUDA tensor (tests only)."""
        p, n = C.c_void_p(), C.c_size_t()
        nv.check(nv.lib().frcnn_ctx_tensor(self.handle, which, C.byref(p), C.byref(n)), "frcnn_ctx_tensor")
        itemsize = t.empty((), dtype=dtype).element_size()
        out = t.empty((n.value // itemsize,), dtype=dtype, device=self.device)
        t.cuda.current_stream(self.device).synchronize()
        rc = _hip_memcpy_dtod(out.data_ptr(), p.value, n.value)
        if rc != 0:
            raise RuntimeError("hipMemcpy failed: %d" % rc)
        return out

    def timing_enable(self, on):
        self.timing = bool(on)
        nv.check(nv.lib().frcnn_ctx_timing_enable(self.handle, 1 if on else 0), "frcnn_ctx_timing_enable")

    def timing_read(self, reset=True):
        ms = (C.c_double * nv.NUM_KCLASS)()
        cnt = (C.c_int64 * nv.NUM_KCLASS)()
        nv.check(nv.lib().frcnn_ctx_timing_read(self.handle, ms, cnt, 1 if reset else 0), "frcnn_ctx_timing_read")
        return {nv.KCLASS_NAMES[i]: (ms[i], cnt[i]) for i in range(nv.NUM_KCLASS)}

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle.value:
                nv.lib().frcnn_ctx_destroy(self.handle)
                self.handle = None
            if getattr(self, "head_handle", None) is not None and self.head_handle.value:
                nv.lib().frcnn_ctx_destroy(self.head_handle)
                self.head_handle = None
        except Exception:
            pass


_hip = None


def _hip_memcpy_dtod(dst, src, nbytes):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.restype = C.c_int
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _hip.hipMemcpy(dst, src, nbytes, 3)   # hipMemcpyDeviceToDevice


_slot_streams = {}


def _lane_stream_index(lane, image):
    """The stream of image `image` of batch lane `lane` (predict_batch_async: RPN + per-RoI head + decode per image behind the lane's feature
    extractor): the in-flight slots' streams 1 .. 4, image i on stream 1 + i % 4, whatever the lane.  Round 6 (tools/exp_r50_lane_streams.py,
    ResNet-50 as batches of 8, two batches in flight, bursts of 24 / 200 images): a stream per (lane, image) = 16 streams 624-641 / 658-666
    images/sec, 8 shared streams 612-618 / 644-653, **these four 658-669 / 701-702** (four streams of their own: 677-680 / 703-715, but four more
    hardware queues of the process): the chip runs four queues side by side, more streams only spread the tails unevenly over them."""
    return 1 + (int(image) % 4)


def slot_stream(device, index):
    """
    The HIP stream of in-flight slot `index` on `device`, ONE per process: every model's slot k enqueues on the same stream.  The chip runs
    four hardware queues side by side and a process has GPU_MAX_HW_QUEUES of them (round 6, profiles/r06/exp_vgg_queues.txt): a second
    model that made its own streams (bench.py's ResNet-50 legs behind VGG-16's slots and feeder streams) found the queues taken and two of
    its slots serialised on one (599 against 680 images/sec standalone).  Models that run in turn lose nothing by sharing; two models in
    flight AT ONCE on the same slot index run one behind the other -- give them different slot indices.
    """
    if isinstance(index, tuple) and len(index) == 3 and index[0] == "lane":
        index = _lane_stream_index(index[1], index[2])
    key = (str(t.device(device)), index)
    st = _slot_streams.get(key)
    if st is None:
        st = t.cuda.Stream(device=t.device(device))
        _slot_streams[key] = st
    return st


class Slot:
    """
    Everything one in-flight image needs: a Context, a stream, device output buffers for
    forward()/predict() and pinned host buffers for the (single) D2H copy of the detections.
    """
    def __init__(self, device, max_h, max_w, max_rois, num_classes, own_stream, index=None):
        self.device = t.device(device)
        self.ctx = Context(device, max_h, max_w, max_rois)
        self.max_rois, self.num_classes = int(max_rois), int(num_classes)
        self.stream = (slot_stream(self.device, index) if index is not None else t.cuda.Stream(device=self.device)) if own_stream else None
        nfg = num_classes - 1
        d = self.device
        self.props = t.zeros((max_rois, 4), dtype=t.float32, device=d)
        self.classes = t.zeros((max_rois, num_classes), dtype=t.float32, device=d)
        self.deltas = t.zeros((max_rois, nfg * 4), dtype=t.float32, device=d)
        self.counts = t.zeros((4,), dtype=t.int32, device=d)
        self.det = t.zeros((nfg, max_rois, 5), dtype=t.float64, device=d)
        self.det_cnt = t.zeros((nfg,), dtype=t.int32, device=d)
        self.h_det = t.zeros((nfg, max_rois, 5), dtype=t.float64).pin_memory()
        self.h_det_cnt = t.zeros((nfg,), dtype=t.int32).pin_memory()
        self.h_counts = t.zeros((4,), dtype=t.int32).pin_memory()
        self.done = t.cuda.Event()
        self.roi_ready = t.cuda.Event()          # predict_batch: this image's pooled RoIs are in the lane's batch buffer
        self.graph, self.graph_key, self.graph_input, self.capture_stream = None, None, None, None    # hipGraph of the last call shape
        self.busy = False
        self.keepalive = None     # references that must outlive the enqueued work

    def use_stream(self):
        return self.stream if self.stream is not None else t.cuda.current_stream(self.device)


class BackboneLane:
    """
    A batch of images going through the ResNet feature extractor together (frcnn_resnet_backbone): a backbone-only frcnn_ctx sized for
    `max_images` maps, the lane's stream, the [max_images][fh][fw][C] feature maps the per-image slots read, and the event that says
    the maps are complete.
    """
    def __init__(self, device, max_h, max_w, max_images, channels):
        nv.require_gpu()
        self.device = t.device(device)
        self.max_h, self.max_w, self.max_images = int(max_h), int(max_w), int(max_images)
        handle = C.c_void_p()
        with t.cuda.device(self.device):
            nv.check(nv.lib().frcnn_ctx_create_backbone(C.byref(handle), self.max_h, self.max_w, self.max_images), "frcnn_ctx_create_backbone")
            self.stream = t.cuda.Stream(device=self.device)
        self.handle = handle
        fh, fw = feature_map_shape(self.max_h, self.max_w)
        self.features = t.empty((self.max_images * fh * fw * int(channels),), dtype=t.float32, device=self.device)
        self.ready = t.cuda.Event()
        self.readers = []            # the slots whose enqueued work still reads self.features
        self.keepalive = None
        # the batch's per-RoI head as ONE set of launches (frcnn_resnet_head, round 6): made by ensure_head
        self.head_handle = None
        self.head_rois = 0
        self.roi_all = self.classes_all = self.deltas_all = None
        self.head_done = t.cuda.Event()

    def ensure_head(self, max_rois, num_classes, channels=1024):
        """The head ctx and the batch buffers: pooled RoIs [max_images][max_rois][7][7][channels], class scores and box deltas per RoI."""
        total = self.max_images * int(max_rois)
        if self.head_handle is not None and self.head_rois >= total and self.classes_all.shape[1] == num_classes:
            return
        if self.head_handle is not None:
            t.cuda.synchronize(self.device)
            nv.lib().frcnn_ctx_destroy(self.head_handle)
            self.head_handle = None
        handle = C.c_void_p()
        with t.cuda.device(self.device):
            nv.check(nv.lib().frcnn_ctx_create_head(C.byref(handle), total), "frcnn_ctx_create_head")
        self.head_handle, self.head_rois = handle, total
        self.roi_all = t.empty((total * 49 * int(channels),), dtype=t.float32, device=self.device)
        self.classes_all = t.zeros((total, int(num_classes)), dtype=t.float32, device=self.device)
        self.deltas_all = t.zeros((total, 4 * (int(num_classes) - 1)), dtype=t.float32, device=self.device)

    def fits(self, h, w, n):
        return h <= self.max_h and w <= self.max_w and n <= self.max_images

    def __del__(self):
        try:
            if getattr(self, "handle", None) is not None and self.handle.value:
                nv.lib().frcnn_ctx_destroy(self.handle)
                self.handle = None
            if getattr(self, "head_handle", None) is not None and self.head_handle.value:
                nv.lib().frcnn_ctx_destroy(self.head_handle)
                self.head_handle = None
        except Exception:
            pass


def feature_map_shape(h, w):
    """(fh, fw) of the ResNet feature map of an h x w image: conv1 7x7/2 pad 3, maxpool 3x3/2 pad 1, layer2 and layer3 stride 2
    (models/resnet.py:38-46) -- each halves as (x - 1) // 2 + 1."""
    for _ in range(4):
        h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    return h, w


def param_key(params):
    """Cache key that changes whenever any of the tensors is replaced, moved or written in place."""
    return tuple((p.data_ptr(), p._version, str(p.device), tuple(p.shape)) for p in params)


def as_f32_cuda(x, what):
    if not isinstance(x, t.Tensor):
        raise TypeError("%s must be a torch.Tensor" % what)
    if not x.is_cuda:
        raise RuntimeError("%s must be a CUDA (MI355X) tensor: this path has no CPU implementation" % what)
    if x.dtype != t.float32:
        raise TypeError("%s must be float32" % what)
    return x.contiguous()


def to_device_map(x, device):
    """Accepts the reference's numpy anchor maps (or tensors) and returns a contiguous CUDA float32 tensor."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        x = t.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return x.to(device=device, dtype=t.float32).contiguous()
