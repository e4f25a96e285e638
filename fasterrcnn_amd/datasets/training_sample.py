"""Ground-truth box record consumed by the mAP calculator (pytorch/FasterRCNN/datasets/training_sample.py:17-27)."""
from dataclasses import dataclass
import numpy as np


@dataclass
class Box:
    class_index: int
    class_name: str
    corners: np.ndarray   # (y1, x1, y2, x2)

    def __repr__(self):
        return "[class=%s (%f,%f,%f,%f)]" % (self.class_name, self.corners[0], self.corners[1],
                                             self.corners[2], self.corners[3])

    def __str__(self):
        return repr(self)
