"""
mAP bookkeeping, mirroring pytorch/FasterRCNN/statistics.py:65-214
(PrecisionRecallCurveCalculator).  Host-side by design: it consumes the per-image dict that
`FasterRCNNModel.predict` returns and a list of ground-truth `Box` records, exactly like the
reference's evaluate() loop (__main__.py:78-86).

Behaviour reproduced on purpose:
  * statistics.py:99 sorts the (iou, box, gt) triples with a key that is the same object for
    every element, i.e. it does NOT sort: matching proceeds gt-major, box-minor (boxes arrive in
    NMS/score-descending order), each ground-truth box taking the first still-unmatched
    prediction with IoU > 0.5 (:118-127).
  * AP = sum of (recall step) x (max precision to the right), with (0,0) and (1,0) sentinels
    (:158-197); mAP = mean over the classes that occur in the ground truth (:210-214).

Added for the image-parallel multi-GPU evaluation: `state()` / `merge()`
serialise the per-class (score, is_true_positive) records and ground-truth counts so ranks can
exchange them with one all-gather; merging in global image order reproduces the single-process
result bit for bit.
"""
from collections import defaultdict

import numpy as np

from .models.math_utils import intersection_over_union


class PrecisionRecallCurveCalculator:
    """
    Collects data over the course of a validation pass and then computes precision and recall
    (including mean average precision).
    """
    def __init__(self):
        # (confidence_score, correctness) by class for all images seen so far
        self._unsorted_predictions_by_class_index = defaultdict(list)
        # true number of objects by class for all images seen so far
        self._object_count_by_class_index = defaultdict(int)

    def _compute_correctness_of_predictions(self, scored_boxes_by_class_index, gt_boxes):
        unsorted_predictions_by_class_index = {}
        object_count_by_class_index = defaultdict(int)
        for gt_box in gt_boxes:
            object_count_by_class_index[gt_box.class_index] += 1

        for class_index, scored_boxes in scored_boxes_by_class_index.items():
            scored_boxes = np.asarray(scored_boxes)
            num_boxes = len(scored_boxes)
            gt_this_class = [gt_box for gt_box in gt_boxes if gt_box.class_index == class_index]
            is_true_positive = np.zeros(num_boxes, dtype=bool)
            if num_boxes > 0 and len(gt_this_class) > 0:
                gt_corners = np.stack([np.asarray(g.corners) for g in gt_this_class], axis=0)
                ious = intersection_over_union(boxes1=scored_boxes[:, 0:4], boxes2=gt_corners)   # (boxes, gts)
                over = ious > 0.5
                for gt_idx in range(len(gt_this_class)):
                    candidates = np.flatnonzero(over[:, gt_idx] & ~is_true_positive)
                    if candidates.size > 0:
                        is_true_positive[candidates[0]] = True
            unsorted_predictions_by_class_index[class_index] = [
                (scored_boxes[i][4], bool(is_true_positive[i])) for i in range(num_boxes)]
        return unsorted_predictions_by_class_index, object_count_by_class_index

    def add_image_results(self, scored_boxes_by_class_index, gt_boxes):
        """
        Adds one image's detections ({class_index: (n,5) rows of (y_min, x_min, y_max, x_max, score)})
        and ground-truth boxes (list of datasets.training_sample.Box) to the running tally.  Call once
        per image.
        """
        predictions, counts = self._compute_correctness_of_predictions(
            scored_boxes_by_class_index=scored_boxes_by_class_index, gt_boxes=gt_boxes)
        for class_index, preds in predictions.items():
            self._unsorted_predictions_by_class_index[class_index] += preds
        for class_index, count in counts.items():
            self._object_count_by_class_index[class_index] += count

    def _compute_average_precision(self, class_index):
        preds = self._unsorted_predictions_by_class_index[class_index]
        num_ground_truth_positives = self._object_count_by_class_index[class_index]
        scores = np.array([p[0] for p in preds], dtype=np.float64)
        correct = np.array([p[1] for p in preds], dtype=bool)
        order = np.argsort(-scores, kind="stable")          # descending, ties keep insertion order
        correct = correct[order]
        true_positives = np.cumsum(correct)
        false_positives = np.cumsum(~correct)
        recall = true_positives / num_ground_truth_positives
        precision = true_positives / np.maximum(true_positives + false_positives, 1)
        recall_array = np.concatenate([[0.0], recall, [1.0]])
        precision_array = np.concatenate([[0.0], precision, [0.0]])
        # interpolation: highest precision seen at or after each point
        precision_array = np.maximum.accumulate(precision_array[::-1])[::-1]
        steps = (recall_array[1:] - recall_array[:-1]) * precision_array[1:]
        average_precision = float(np.cumsum(steps)[-1]) if steps.size else 0.0   # cumsum = sequential adds
        return average_precision, recall_array.tolist(), precision_array.tolist()

    def compute_mean_average_precision(self):
        """mAP over all classes present in the ground truth seen so far (np.float64)."""
        average_precisions = []
        for class_index in self._object_count_by_class_index:
            average_precision, _, _ = self._compute_average_precision(class_index=class_index)
            average_precisions.append(average_precision)
        return np.mean(average_precisions)

    def print_average_precisions(self, class_index_to_name):
        labels = [class_index_to_name[c] for c in self._object_count_by_class_index]
        aps = {class_index_to_name[c]: self._compute_average_precision(class_index=c)[0]
               for c in self._object_count_by_class_index}
        width = max([len(x) for x in labels] + [1])
        print("Average Precisions")
        print("------------------")
        for label, ap in sorted(aps.items(), key=lambda kv: kv[1], reverse=True):
            print("%s: %1.1f%%" % (label.ljust(width), ap * 100.0))
        print("------------------")

    # ---- multi-process exchange ---------------------------------------------------------------
    def state(self):
        """Flat numpy view of the accumulator: records (class, score, tp) in insertion order + GT counts."""
        cls, score, tp = [], [], []
        for class_index, preds in self._unsorted_predictions_by_class_index.items():
            for s, c in preds:
                cls.append(class_index); score.append(s); tp.append(1 if c else 0)
        gt_cls = list(self._object_count_by_class_index.keys())
        return {
            "cls": np.asarray(cls, dtype=np.int64), "score": np.asarray(score, dtype=np.float64),
            "tp": np.asarray(tp, dtype=np.int64),
            "gt_cls": np.asarray(gt_cls, dtype=np.int64),
            "gt_cnt": np.asarray([self._object_count_by_class_index[c] for c in gt_cls], dtype=np.int64),
        }

    def merge_state(self, state):
        for c, s, k in zip(state["cls"].tolist(), state["score"].tolist(), state["tp"].tolist()):
            self._unsorted_predictions_by_class_index[int(c)].append((np.float64(s), bool(k)))
        for c, n in zip(state["gt_cls"].tolist(), state["gt_cnt"].tolist()):
            self._object_count_by_class_index[int(c)] += int(n)
