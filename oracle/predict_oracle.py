"""ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy restatement of the predict-loop glue
(`/root/reference/src/predict.py:120-159, 100, 259-260`): nearest-neighbour down-size of the decoded
mask, one-hot vote paste with clipping, final arg-max.  Integer work: the product path must match
it bit for bit."""
from __future__ import annotations

import numpy as np


def nearest_resize(mask: np.ndarray, size: int) -> np.ndarray:
    """`cv2.resize(pred, (size, size), interpolation=cv2.INTER_NEAREST)` (`src/predict.py:259`):
    source index = min(floor(dst * src/dst_size), src - 1) on both axes."""
    h, w = mask.shape
    ys = np.minimum(np.floor(np.arange(size) * (h / size)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(size) * (w / size)).astype(np.int64), w - 1)
    return mask[ys[:, None], xs[None, :]]


def one_hot(pred: np.ndarray, num_classes: int) -> np.ndarray:
    """`np.eye(num_classes, dtype=np.uint8)[pred]` (`src/predict.py:260`)."""
    return np.eye(num_classes, dtype=np.uint8)[pred]


def accumulate(counter: np.ndarray, crop: tuple[int, int, int, int], one_hot_pred: np.ndarray) -> bool:
    """`Accumulator.update` vote paste (`src/predict.py:138-157`): clip the crop window to the mosaic,
    add the matching source slice into the uint8 counters (wraps at 256 like numpy uint8)."""
    h, w = counter.shape[:2]
    xmin, ymin, xmax, ymax = crop
    dy0, dy1, dx0, dx1 = max(ymin, 0), min(ymax, h), max(xmin, 0), min(xmax, w)
    sy0, sx0 = dy0 - ymin, dx0 - xmin
    sy1, sx1 = sy0 + (dy1 - dy0), sx0 + (dx1 - dx0)
    if sy1 <= sy0 or sx1 <= sx0:
        return False
    counter[dy0:dy1, dx0:dx1] += one_hot_pred[sy0:sy1, sx0:sx1]
    return True


def vote_argmax(counter: np.ndarray) -> np.ndarray:
    """`np.argmax(self.current_pred_counter, axis=2)` (`src/predict.py:100`)."""
    return np.argmax(counter, axis=2)
