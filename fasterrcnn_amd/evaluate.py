"""
Image-stream inference + mAP, the build's counterpart of the reference's evaluate() loop
(pytorch/FasterRCNN/__main__.py:62-96): for each sample `model.predict(image, score_threshold=0.05)`
then `PrecisionRecallCurveCalculator.add_image_results`, finally mAP.

Two additions:
  * several images are kept in flight on separate HIP streams (`inflight` slots; every image is
    still an independent batch-1 forward), because the late VGG-16 layers of ONE 600x1000 image
    expose fewer wave-tiles than the chip has SIMDs;
  * image-parallel multi-GPU: rank r takes images r, r+world, ...; there is NO communication
    during inference.  The only exchange is at the end: every rank's per-image records
    (image, class, score, is_true_positive) and ground-truth counts are all-gathered (one
    size exchange + one padded payload all-gather per array, RCCL over xGMI when the backend is
    "nccl"), re-ordered by global image index and merged, which reproduces the single-process
    mAP bit for bit.
"""
import numpy as np
import torch as t
import torch.distributed as dist

from .statistics import PrecisionRecallCurveCalculator


class ImageRecords:
    """Per-image mAP records of one rank, kept in a form that can be exchanged and re-ordered."""
    def __init__(self):
        self.pred = []   # rows (image_index, class_index, score, is_tp)
        self.gt = []     # rows (image_index, class_index, count)

    def add(self, image_index, scored_boxes_by_class_index, gt_boxes):
        calc = PrecisionRecallCurveCalculator()
        calc.add_image_results(scored_boxes_by_class_index=scored_boxes_by_class_index, gt_boxes=gt_boxes)
        s = calc.state()
        for c, sc, tp in zip(s["cls"].tolist(), s["score"].tolist(), s["tp"].tolist()):
            self.pred.append((float(image_index), float(c), sc, float(tp)))
        for c, n in zip(s["gt_cls"].tolist(), s["gt_cnt"].tolist()):
            self.gt.append((int(image_index), int(c), int(n)))

    def arrays(self):
        pred = np.asarray(self.pred, dtype=np.float64).reshape(-1, 4)
        gt = np.asarray(self.gt, dtype=np.int64).reshape(-1, 3)
        return pred, gt


def _all_gather_rows(x, device):
    """all-gather of a 2-D tensor whose first dimension differs per rank (sizes first, then padded payload)."""
    world = dist.get_world_size()
    n = t.tensor([x.shape[0]], dtype=t.int64, device=device)
    sizes = [t.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    pad = t.zeros((m, x.shape[1]), dtype=x.dtype, device=device)
    pad[: x.shape[0]] = x.to(device)
    bufs = [t.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return t.cat([b[:k] for b, k in zip(bufs, sizes)], dim=0).cpu()


def merged_calculator(records, device=None, force_gather=False):
    """
    Builds the global PrecisionRecallCurveCalculator from this rank's ImageRecords.  With an
    initialised process group of more than one rank the records of all ranks are exchanged first
    (every rank gets the full result); without one it is the local accumulation.
    `force_gather` runs the exchange also in a one-rank group (the RCCL path on a single GPU).
    """
    pred, gt = records.arrays()
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_gather):
        if device is None:
            device = t.device("cuda", t.cuda.current_device()) if dist.get_backend() == "nccl" else t.device("cpu")
        pred = _all_gather_rows(t.from_numpy(pred), device).numpy()
        gt = _all_gather_rows(t.from_numpy(gt), device).numpy()
    calc = PrecisionRecallCurveCalculator()
    # global image order, records of one image in their original (class, NMS) order
    order = np.argsort(pred[:, 0], kind="stable") if len(pred) else np.zeros((0,), dtype=np.int64)
    pred = pred[order]
    gorder = np.argsort(gt[:, 0], kind="stable") if len(gt) else np.zeros((0,), dtype=np.int64)
    gt = gt[gorder]
    # a class enters the calculator's dicts when first seen in ANY image, as in a sequential run:
    # walk images in order and interleave each image's predictions and ground truth
    pi = gi = 0
    images = sorted(set(pred[:, 0].astype(np.int64).tolist()) | set(gt[:, 0].tolist()))
    for img in images:
        state = {"cls": [], "score": [], "tp": [], "gt_cls": [], "gt_cnt": []}
        while pi < len(pred) and int(pred[pi, 0]) == img:
            state["cls"].append(int(pred[pi, 1])); state["score"].append(pred[pi, 2]); state["tp"].append(int(pred[pi, 3]))
            pi += 1
        while gi < len(gt) and int(gt[gi, 0]) == img:
            state["gt_cls"].append(int(gt[gi, 1])); state["gt_cnt"].append(int(gt[gi, 2]))
            gi += 1
        calc.merge_state({k: np.asarray(v) for k, v in state.items()})
    return calc


def evaluate_stream(model, samples, score_threshold=0.05, inflight=4, rank=0, world=1, on_result=None, batch=1):
    """
    samples: sequence of (image_index, image (1,3,H,W) CUDA float32 tensor, gt_boxes list[Box]).
    Processes the samples whose position p satisfies p % world == rank, `inflight` at a time, and
    returns this rank's ImageRecords.  `on_result(image_index, dict)` is called per finished image.
    batch > 1 (ResNet backbones): consecutive samples of one shape go through the feature extractor as ONE batch of up to `batch`
    images (model.predict_batch_async), max(1, inflight // batch) batches in flight (two measure best: inflight = 2 x batch); results and their order are per image as before.
    """
    records = ImageRecords()
    pending = []   # (Pending, image_index, gt_boxes)

    def collect(entry):
        handle, image_index, gt_boxes = entry
        det = handle.result()
        if on_result is not None:
            on_result(image_index, det)
        if gt_boxes is not None:
            records.add(image_index, det, gt_boxes)

    batch = max(1, int(batch))
    if batch > 1 and not getattr(model, "_is_resnet", False):
        batch = 1               # VGG-16's layers fill the chip with one image: its images go in flight one by one
    if batch > 1:
        nlanes = max(1, int(inflight) // batch)               # (inflight = 2 x batch: the second batch's feature extractor under the first's per-image tails)
        group, lanes, state = [], [], {"lane": 0}             # lanes[i]: the lane of pending[i]

        def flush():
            if not group:
                return
            while state["lane"] in lanes:                    # the lane about to be reused must have been collected (in image order)
                collect(pending.pop(0))
                lanes.pop(0)
            images = group[0][1] if len(group) == 1 else t.cat([g[1] for g in group], dim=0)
            handles = model.predict_batch_async(images, score_threshold, lane=state["lane"])
            pending.extend((hd, g[0], g[2]) for hd, g in zip(handles, group))
            lanes.extend([state["lane"]] * len(group))
            state["lane"] = (state["lane"] + 1) % nlanes
            del group[:]

        for p, (image_index, image, gt_boxes) in enumerate(samples):
            if p % world != rank:
                continue
            if group and tuple(group[0][1].shape[1:]) != tuple(image.shape[1:]):
                flush()
            group.append((image_index, image, gt_boxes))
            if len(group) == batch:
                flush()
        flush()
        while pending:
            collect(pending.pop(0))
        return records

    nslots = max(1, int(inflight))
    k = 0
    for p, (image_index, image, gt_boxes) in enumerate(samples):
        if p % world != rank:
            continue
        if len(pending) == nslots:
            collect(pending.pop(0))
        slot = 1 + (k % nslots)
        pending.append((model.predict_async(image, score_threshold, slot=slot), image_index, gt_boxes))
        k += 1
    while pending:
        collect(pending.pop(0))
    return records


class BackgroundUploader:
    """
    The upload of evaluate()'s loop (the reference's `t.from_numpy(image).unsqueeze(0).cuda()`, __main__.py:78-86) off the submitting
    thread.  A preprocessed float32 (3, 600, 1000) image is 7.2 MB; `.to(device)` from PAGEABLE memory is a synchronous, staged copy of
    ~1 ms, and with it in the loop evaluate() measured 545-730 images/sec against 960-980 from resident images
    (tools/exp_evaluate_h2d.py).  Staging through pinned buffers first is no cure: the host's copy INTO pinned (uncached) memory took 10 ms per
    image (measured: 100 images/sec).  So ONE worker thread walks the samples -- the dataset's own iteration included -- and uploads each
    image on its own stream (the copy blocks only that thread; torch releases the GIL for it), `depth` images ahead; the consuming thread
    makes its current stream wait for the image's event, so whatever it enqueues next (predict_async) is ordered behind the copy.
    A CPU `device` (the gloo tests' stand-ins): no thread, the plain conversion.
    """
    _END = object()

    def __init__(self, device, depth=4):
        self.device, self.depth = t.device(device), max(1, int(depth))

    @staticmethod
    def _as_tensor(array):
        return array if isinstance(array, t.Tensor) else t.from_numpy(np.ascontiguousarray(array, dtype=np.float32))

    def iterate(self, items):
        """items: iterable of (index, image or None, payload); yields (index, (1, 3, H, W) tensor on the device or None, payload) in order."""
        if self.device.type != "cuda":
            for i, array, payload in items:
                yield i, (None if array is None else self._as_tensor(array).unsqueeze(dim=0).to(self.device)), payload
            return
        import queue
        import threading
        q, stop = queue.Queue(maxsize=self.depth), threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def worker():
            try:
                with t.cuda.device(self.device):
                    stream = t.cuda.Stream(device=self.device)
                    for i, array, payload in items:
                        if array is None:
                            if not put((i, None, None, payload)):
                                return
                            continue
                        src = self._as_tensor(array)
                        with t.cuda.stream(stream):
                            image = src.unsqueeze(dim=0).to(self.device)
                            ready = t.cuda.Event()
                            ready.record(stream)
                        if not put((i, image, ready, payload)):
                            return
                put(self._END)
            except BaseException as e:                     # (handed to the consuming thread)
                put(e)

        th = threading.Thread(target=worker, name="frcnn-upload", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is self._END:
                    break
                if isinstance(item, BaseException):
                    raise item
                i, image, ready, payload = item
                if image is not None:
                    cur = t.cuda.current_stream(self.device)
                    cur.wait_event(ready)
                    image.record_stream(cur)
                yield i, image, payload
        finally:
            stop.set()


def default_inflight(model):
    """
    Images in flight per GPU that measured best (profiles/r06/exp_inflight_driver.txt, inflight_sweep.txt): VGG-16's layers fill the chip
    from one image, so 4 images -- one per default HIP hardware queue, 20-image bursts split 5 + 5 + 5 + 5 -- are enough to cover the serial
    proposal / detection tails (3: -1 %, 5-6: -6 ... -12 %, 8 = two per pipe: -1.5 %).  ResNet-50 measures the same pattern (4: 702, 5-6: 624-651,
    8: 693 images/sec, exp_r50_inflight.txt): one answer for every backbone.
    """
    return 4


def evaluate(model, eval_data, num_samples=None, print_average_precisions=False, class_index_to_name=None,
             inflight=None, score_threshold=0.05, force_gather=False):
    """
    The reference's evaluate() (pytorch/FasterRCNN/__main__.py:62-96): `model.predict(score_threshold=0.05)` per
    sample of `eval_data`, `PrecisionRecallCurveCalculator.add_image_results`, returns 100 x mAP.
    `eval_data` iterates samples that carry `.image_data` (numpy or tensor (3, H, W), preprocessed) and `.gt_boxes`
    (the reference's TrainingSample); the dataset itself (voc.Dataset) stays the caller's.  Images are uploaded and
    predicted `inflight` at a time (None = default_inflight(model)); under an initialised process group each rank takes every world-th sample and the
    records are merged with one all-gather (merged_calculator), so every rank returns the same value.
    """
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    device = model._device()
    if inflight is None:
        inflight = default_inflight(model)

    def samples():
        for i, sample in enumerate(eval_data):
            if num_samples is not None and i >= num_samples:
                break
            if i % world != rank:
                yield i, None, None           # evaluate_stream skips it without touching the image
                continue
            yield i, sample.image_data, sample.gt_boxes

    def stream():
        return BackgroundUploader(device, depth=int(inflight)).iterate(samples())

    records = evaluate_stream(model, stream(), score_threshold=score_threshold, inflight=inflight, rank=rank, world=world)
    calc = merged_calculator(records, force_gather=force_gather)
    if print_average_precisions:
        calc.print_average_precisions(class_index_to_name=class_index_to_name or {})
    return 100.0 * calc.compute_mean_average_precision()


def predict(model, image_data, score_threshold=0.7):
    """__main__.py:226-228 without the drawing: (3, H, W) preprocessed numpy image -> predict() dict."""
    if not isinstance(image_data, t.Tensor):
        image_data = t.from_numpy(np.ascontiguousarray(image_data, dtype=np.float32))
    return model.predict(image_data=image_data.unsqueeze(dim=0).to(model._device()), score_threshold=score_threshold)


class HostFeeder:
    """
    The host -> device leg of the reference's loops, pipelined: `__main__.py:78-86` does `t.from_numpy(image).unsqueeze(0).cuda()` then
    `predict` per image, and `predict_one` (`:237-240`) runs `load_image` (decode + PIL resize + normalise on the CPU,
    datasets/image.py:59-101) first.  Here a DECODED image in pinned host memory goes through: async H2D of the uint8 pixels (0.7 MB for a
    375x625 VOC-sized image, against 7.2 MB for the preprocessed float32 tensor the reference uploads) -> `frcnn_preprocess` (PIL-exact
    BILINEAR resize to the 600-pixel minimum side + normalisation, on the device: datasets/image.py) on a feeder stream -> `model.predict_async`
    on an in-flight slot's stream, which waits for the feeder's event on the device.  The host never blocks between stage() / submit() and
    the handle's result().

    `stage()` and `submit_staged()` are the two halves: staging the frames a few images AHEAD of their predict (`lookahead` feeder streams
    in a ring) takes the copy + resize latency (~0.1 ms) off every image's own critical path -- with three chip-filling images in flight
    that latency is not hidden by the other two.  `submit()` is stage + submit_staged back to back; `submit_preprocessed` is the
    reference's literal form: a preprocessed float32 (3, H, W) host image, uploaded as is.
    """
    def __init__(self, model, min_dimension_pixels=600, lookahead=4):
        self.model = model
        self.min_dimension_pixels = min_dimension_pixels
        self.device = model._device()
        self._streams = [t.cuda.Stream(device=self.device) for _ in range(max(1, int(lookahead)))]
        self._staging = [None] * len(self._streams)     # device copy of the frame last uploaded on ring entry k (kept: the copy is asynchronous)
        self._host = [None] * len(self._streams)        # (pinned source tensor, copy-done event) of ring entry k: the source stays referenced until its copy ran
        self._next = 0

    def stage(self, rgb_u8_host):
        """rgb_u8_host: uint8 (H, W, 3) RGB CPU tensor, pinned for a truly asynchronous copy.  Enqueues H2D + resize + normalisation on the
        next feeder stream of the ring and returns the staged handle (preprocessed CUDA tensor, event) for `submit_staged`."""
        from .datasets import image as I
        if rgb_u8_host.dtype != t.uint8 or rgb_u8_host.dim() != 3 or rgb_u8_host.shape[2] != 3:
            raise ValueError("rgb_u8_host must be a uint8 (H, W, 3) image")
        k = self._next
        self._next = (k + 1) % len(self._streams)
        feeder = self._streams[k]
        with t.cuda.device(self.device), t.cuda.stream(feeder):
            buf = self._staging[k]
            if buf is None or buf.shape != rgb_u8_host.shape:
                buf = t.empty(rgb_u8_host.shape, dtype=t.uint8, device=self.device)
                self._staging[k] = buf
            buf.copy_(rgb_u8_host, non_blocking=True)       # (ring entry k's previous frame was consumed by its own preprocess: same stream)
            copied = t.cuda.Event()
            copied.record(feeder)
            self._host[k] = (rgb_u8_host, copied)           # the asynchronous copy reads the caller's pinned tensor: keep it alive
            image, _, _ = I.preprocess_image(buf, self.model.backbone.image_preprocessing_params, self.min_dimension_pixels, False)
            ev = t.cuda.Event()
            ev.record(feeder)
        return image, ev, feeder

    def submit_staged(self, staged, score_threshold, slot):
        """predict_async of a staged frame on in-flight slot `slot` (whose previous handle must have been collected)."""
        image, ev, feeder = staged
        # the slot's stream waits for THIS frame's event, not for the feeder stream's tail (frames staged later on the same ring entry)
        with t.cuda.device(self.device):
            return self.model.predict_async(image.unsqueeze(0), score_threshold, slot=slot, wait_event=ev)

    def submit(self, rgb_u8_host, score_threshold, slot):
        """stage + submit_staged back to back.  Returns the Pending handle of predict_async (its result() is the reference's predict() dict)."""
        return self.submit_staged(self.stage(rgb_u8_host), score_threshold, slot)

    def submit_preprocessed(self, image_f32_host, score_threshold, slot):
        """image_f32_host: float32 (3, H, W) preprocessed CPU tensor (what the reference's dataset yields), pinned."""
        k = self._next
        self._next = (k + 1) % len(self._streams)
        feeder = self._streams[k]
        with t.cuda.device(self.device), t.cuda.stream(feeder):
            image = t.empty(image_f32_host.shape, dtype=t.float32, device=self.device)
            image.copy_(image_f32_host, non_blocking=True)
            return self.model.predict_async(image.unsqueeze(0), score_threshold, slot=slot)


def predict_one(model, url, score_threshold=0.7, min_dimension_pixels=600):
    """
    __main__.py:237-240 (`run_one_image` of the north star): load_image with the backbone's preprocessing at a
    600-pixel minimum side, then predict at score threshold 0.7.  Returns (scored_boxes_by_class_index, PIL image,
    scale_factor); visualisation (visualize.show_detections) is outside this build's scope.
    Decode on the host, resize + normalise on the device (datasets/image.py) -- the tensor never leaves the GPU.
    """
    from PIL import Image
    from .datasets import image as I
    with Image.open(url) as im:
        rgb = np.array(im.convert("RGB"))
    image_data, scale_factor, _, resized = I.preprocess_image(rgb, model.backbone.image_preprocessing_params,
                                                              min_dimension_pixels, False, return_resized=True)
    det = model.predict(image_data=image_data.unsqueeze(dim=0), score_threshold=score_threshold)
    return det, Image.fromarray(resized.cpu().numpy(), mode="RGB"), scale_factor
