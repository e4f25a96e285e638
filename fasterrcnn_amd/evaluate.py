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


def merged_calculator(records, device=None):
    """
    Builds the global PrecisionRecallCurveCalculator from this rank's ImageRecords.  With an
    initialised process group the records of all ranks are exchanged first (every rank gets the
    full result); without one it is the local accumulation.
    """
    pred, gt = records.arrays()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
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


def evaluate_stream(model, samples, score_threshold=0.05, inflight=4, rank=0, world=1, on_result=None):
    """
    samples: sequence of (image_index, image (1,3,H,W) CUDA float32 tensor, gt_boxes list[Box]).
    Processes the samples whose position p satisfies p % world == rank, `inflight` at a time, and
    returns this rank's ImageRecords.  `on_result(image_index, dict)` is called per finished image.
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
